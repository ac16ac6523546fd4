"""k_preprocess alone at a BASELINE scene (library event profiler): python tools/time_preprocess.py [C3] [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gs_livm_amd as G
from gs_livm_amd import synthetic as S
from helpers import to_dev
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
P, W, H, seed = S.CONFIGS[name]
sc = S.make_scene(P, W, H, seed)
dev = torch.device("cuda:0")
t = to_dev(sc, dev)
def fw():
    return G.rasterize_forward(t["bg"], t["means3D"], t["colors_precomp"], t["opacities"], t["scales"], t["rotations"], 1.0, t["cov3D_precomp"], t["viewmatrix"], t["projmatrix"], sc["tanfovx"], sc["tanfovy"], H, W, t["shs"], 0, t["campos"], False, False)
for _ in range(5): fw()
torch.cuda.synchronize()
G.profile_enable(True, only=["k_preprocess", "k_scan_offsets"])
for _ in range(n): fw()
torch.cuda.synchronize()
G.profile_enable(False)
pr = G.profile_read()
print("%s %s: k_preprocess %.1f us  k_scan_offsets %.1f us" % (os.path.basename(G.LIB_PATH), name, 1e3 * pr["k_preprocess"][0] / pr["k_preprocess"][1], 1e3 * pr["k_scan_offsets"][0] / max(1, pr["k_scan_offsets"][1])), flush=True)
