import sys, time
import numpy as np, torch
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gs_livm_amd as G
from gs_livm_amd import synthetic as S
from helpers import hip_forward
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
P, W, H, seed = S.CONFIGS[name]
sc = S.make_scene(P, W, H, seed)
t, fwd = hip_forward(sc, dev, debug=False)
dcol, dacc = S.make_upstream_grads(W, H, seed)
dc = torch.from_numpy(dcol).to(dev); da = torch.from_numpy(dacc).to(dev)
def fw():
    return G.rasterize_forward(t["bg"], t["means3D"], t["colors_precomp"], t["opacities"], t["scales"], t["rotations"], 1.0, t["cov3D_precomp"], t["viewmatrix"], t["projmatrix"], sc["tanfovx"], sc["tanfovy"], H, W, t["shs"], 0, t["campos"], False, False)
def bw(f):
    return G.rasterize_backward(t["bg"], t["means3D"], f[4], t["colors_precomp"], t["scales"], t["rotations"], 1.0, t["cov3D_precomp"], t["viewmatrix"], t["projmatrix"], sc["tanfovx"], sc["tanfovy"], dc, da, t["shs"], 0, t["campos"], f[5], f[0], f[6], f[7], False)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f = fw(); t1a = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter(); g = bw(f); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%s iter %d: fwd %.2f ms (host return %.2f)  bwd %.2f ms" % (name, it, (t1 - t0) * 1e3, (t1a - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
G.profile_enable(True)
f = fw(); g = bw(f); torch.cuda.synchronize()
G.profile_enable(False)
for k, (ms, c) in sorted(G.profile_read().items(), key=lambda kv: -kv[1][0]):
    if c: print("  %-22s %8.3f ms  x%d" % (k, ms, c))
