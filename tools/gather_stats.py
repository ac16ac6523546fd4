"""What the gradient gather of a near/far frame has to look at: python tools/gather_stats.py [yaw_deg=20] [workload=C3]
Renders the BASELINE scene from a yawed camera (the image border then shows the edge of the synthetic map: tiles that
never saturate, so the far chain runs), prints the frame's device counters and how many Gaussians carry a gradient."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gs_livm_amd as G
from gs_livm_amd import synthetic as S
from gs_livm_amd import _capi as CAPI
from helpers import to_dev
yaw = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
name = sys.argv[2] if len(sys.argv) > 2 else "C3"
P, W, H, seed = S.CONFIGS[name]
sc = S.make_scene(P, W, H, seed)
cam = S.make_camera(W, H, yaw_deg=yaw)
for k in ("viewmatrix", "projmatrix", "campos", "tanfovx", "tanfovy"):
    sc[k] = cam[k]
dev = torch.device("cuda:0")
t = to_dev(sc, dev)
dcol, dacc = S.make_upstream_grads(W, H, seed)
dc, da = torch.from_numpy(dcol).to(dev), torch.from_numpy(dacc).to(dev)
def fw():
    return G.rasterize_forward(t["bg"], t["means3D"], t["colors_precomp"], t["opacities"], t["scales"], t["rotations"], 1.0, t["cov3D_precomp"], t["viewmatrix"], t["projmatrix"], sc["tanfovx"], sc["tanfovy"], H, W, t["shs"], 0, t["campos"], False, False)
def bw(f):
    return G.rasterize_backward(t["bg"], t["means3D"], f[4], t["colors_precomp"], t["scales"], t["rotations"], 1.0, t["cov3D_precomp"], t["viewmatrix"], t["projmatrix"], sc["tanfovx"], sc["tanfovy"], dc, da, t["shs"], 0, t["campos"], f[5], f[0], f[6], f[7], False)
for _ in range(6):
    f = fw(); g = bw(f)
torch.cuda.synchronize()
f = fw()
torch.cuda.synchronize()
v = CAPI.state_views(f[5], f[6], f[7], P, f[0], W, H)
c = v["counters"]
print("yaw %.0f: all instances %d | near %d, far %d | near Gaussians %d, far Gaussians emitted %d | tiles unfinished after the near phase %d of %d" % (
    yaw, c[0], c[6], c[8], c[7], c[10], c[9], ((W + 15) // 16) * ((H + 15) // 16)))
g = bw(f)
torch.cuda.synchronize()
print("Gaussians with a non-zero opacity gradient: %d; speculation %s" % (int((g[2] != 0).sum()), G.speculation_stats()))
tt = v["tiles_touched"].long()
order = v["depth_order"].long()
near_ids = order[:c[7]]
def dist(name, x):
    x = x.float()
    if x.numel() == 0:
        print("  %s: none" % name); return
    q = torch.quantile(x, torch.tensor([0.5, 0.9, 0.99], device=x.device)).tolist()
    print("  %s: %d Gaussians, slots sum %d, run length p50 %d p90 %d p99 %d max %d" % (name, x.numel(), int(x.sum()), q[0], q[1], q[2], int(x.max())))
dist("near Gaussians (emitted)", tt[near_ids])
touched = (g[2].view(-1) != 0).nonzero().view(-1)
isnear = torch.zeros(P, dtype=torch.bool, device=dev); isnear[near_ids] = True
dist("with gradient, near", tt[touched[isnear[touched]]])
dist("with gradient, far", tt[touched[~isnear[touched]]])
G.profile_enable(True, only=["k_gather_records"])
for _ in range(5):
    g = bw(fw())
torch.cuda.synchronize()
G.profile_enable(False)
pr = G.profile_read()
print("k_gather_records %.1f us" % (1e3 * pr["k_gather_records"][0] / max(1, pr["k_gather_records"][1])))
