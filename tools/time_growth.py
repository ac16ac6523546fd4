"""Where does a step go right after the map has grown?  python tools/time_growth.py [P=2000000] [n=20000] [K=4]
Times add_new_pointcloud, the first K-view iteration after it and the second one, and counts the device allocations
(cudaMalloc-level) the caching allocator had to make for each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import gs_livm_amd as G
from gs_livm_amd import synthetic as S
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4
W, H = 1920, 1080
dev = torch.device("cuda:0")
g = S.make_gaussians(P, 3, aspect=W / H)
model = G.GrowableGaussians(P + 64 * n, 1, dev)
model.P = P; model._bind()
with torch.no_grad():
    model._xyz.copy_(torch.from_numpy(g["means3D"])); model._features_dc.copy_(torch.from_numpy(g["shs"][:, :1]))
    model._scaling.copy_(torch.from_numpy(np.log(g["scales"]))); model._rotation.copy_(torch.from_numpy(g["rotations"]))
    model._opacity.copy_(torch.from_numpy(np.log(g["opacities"] / (1 - g["opacities"]))))
opt = G.GrowableAdam(model, eps=1e-15)
model.fused_tail = True
bg = torch.ones(3, device=dev)
rasters = []
for y in S.C4_YAWS_DEG[:K]:
    cm = S.make_camera(W, H, yaw_deg=y)
    rasters.append(G.GaussianRasterizer(G.GaussianRasterizationSettings(H, W, cm["tanfovx"], cm["tanfovy"], bg, 1.0,
        torch.from_numpy(cm["viewmatrix"]).to(dev), torch.from_numpy(cm["projmatrix"]).to(dev), 0,
        torch.from_numpy(cm["campos"]).to(dev), False)))
dcol, dacc = S.make_upstream_grads(W, H, 3)
wc, wa = torch.from_numpy(dcol).to(dev), torch.from_numpy(dacc).to(dev)
def step():
    xyz, op, sc, rot, shs = model.activated()
    outs = [r(xyz, torch.zeros((model.P, 3), device=dev, requires_grad=True), op, shs=shs, scales=sc, rotations=rot) for r in rasters]
    torch.autograd.backward([t for o in outs for t in (o[0], o[3])], [wc, wa] * K)
    opt.step_model(model)
def timed(f):
    torch.cuda.synchronize(); a0 = torch.cuda.memory_stats().get("num_device_alloc", 0); t0 = time.perf_counter()
    f(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, torch.cuda.memory_stats().get("num_device_alloc", 0) - a0
for _ in range(4): step()
print("steady step: %.2f ms, %d device allocations" % timed(step))
gen = torch.Generator().manual_seed(1)
for rep in range(3):
    xyz = torch.randn((n, 3), generator=gen).to(dev); covs = (torch.eye(3) * 1e-3).repeat(n, 1, 1).to(dev); rgb = torch.rand((n, 3), generator=gen).to(dev) * 255
    print("grow: %.2f ms, %d allocs" % timed(lambda: model.add_new_pointcloud(xyz, covs, rgb, 1.0)), "| P =", model.P)
    print("  first step after: %.2f ms, %d allocs" % timed(step))
    print("  second step after: %.2f ms, %d allocs" % timed(step))
