"""A/B timing of the two blend kernels on a BASELINE scene in the product's steady state (speculative near/far
forward, backward behind it): python tools/time_blend.py [C3] [iters].  GSR_LIB selects an experiment build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gs_livm_amd as G
from gs_livm_amd import synthetic as S
from helpers import to_dev
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
P, W, H, seed = S.CONFIGS[name]
sc = S.make_scene(P, W, H, seed)
dev = torch.device("cuda:0")
t = to_dev(sc, dev)
dcol, dacc = S.make_upstream_grads(W, H, seed)
dc, da = torch.from_numpy(dcol).to(dev), torch.from_numpy(dacc).to(dev)
def fw():
    return G.rasterize_forward(t["bg"], t["means3D"], t["colors_precomp"], t["opacities"], t["scales"], t["rotations"], 1.0, t["cov3D_precomp"], t["viewmatrix"], t["projmatrix"], sc["tanfovx"], sc["tanfovy"], H, W, t["shs"], 0, t["campos"], False, False)
def bw(f):
    return G.rasterize_backward(t["bg"], t["means3D"], f[4], t["colors_precomp"], t["scales"], t["rotations"], 1.0, t["cov3D_precomp"], t["viewmatrix"], t["projmatrix"], sc["tanfovx"], sc["tanfovy"], dc, da, t["shs"], 0, t["campos"], f[5], f[0], f[6], f[7], False)
for _ in range(5):
    g = bw(fw())
torch.cuda.synchronize()
G.profile_enable(True, only=["k_blend_forward", "k_blend_backward"])
for _ in range(n):
    g = bw(fw())
torch.cuda.synchronize()
G.profile_enable(False)
pr = G.profile_read()
print("%s %s: blend_forward %.1f us  blend_backward %.1f us  (split %s)  checksum %.9e" % (
    os.path.basename(G.LIB_PATH), name, 1e3 * pr["k_blend_forward"][0] / max(1, pr["k_blend_forward"][1]),
    1e3 * pr["k_blend_backward"][0] / max(1, pr["k_blend_backward"][1]), G.last_near_far()[0],
    float(sum(x.double().abs().sum() for x in g[:8]))), flush=True)
