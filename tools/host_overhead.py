"""Host cost of one training step, piece by piece, on a scene so small that the GPU work is negligible:
    python tools/host_overhead.py [P=2000] [host=cpp|python] [steps=400]
Prints the mean / median wall time of model.activated(), the rasterizer forward (which contains the one host wait of
an unsplit frame: the instance count), autograd.backward and the fused optimiser tail, without any synchronisation
between them, and the whole step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import gs_livm_amd as G
from gs_livm_amd import synthetic as S
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
host = sys.argv[2] if len(sys.argv) > 2 else "cpp"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
W, H = 640, 512
dev = torch.device("cuda:0")
g = S.make_gaussians(P, 3, aspect=W / H)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
model = G.GaussianParameters(t(g["means3D"]), t(g["shs"][:, :1]), t(g["shs"][:, 1:]), t(np.log(g["scales"])), t(g["rotations"]),
                             t(np.log(g["opacities"] / (1 - g["opacities"]))))
opt = G.FusedAdam([gr for gr in model.param_groups() if gr["params"][0].numel()], eps=1e-15)
model.fused_tail = True
bg = torch.ones(3, device=dev)
cm = S.make_camera(W, H)
vm, pm, cc = (torch.from_numpy(cm[k]).to(dev) for k in ("viewmatrix", "projmatrix", "campos"))
if host == "cpp":
    T = G.torch_ops()
    r_cpp = T.GaussianRasterizer(T.GaussianRasterizationSettings(H, W, cm["tanfovx"], cm["tanfovy"], bg, 1.0, vm, pm, 0, cc, False))
    raster = lambda xyz, m2d, op, shs, scales, rotations: r_cpp.forward(xyz, m2d, op, shs=shs, scales=scales, rotations=rotations)
else:
    raster = G.GaussianRasterizer(G.GaussianRasterizationSettings(H, W, cm["tanfovx"], cm["tanfovy"], bg, 1.0, vm, pm, 0, cc, False))
dcol, dacc = S.make_upstream_grads(W, H, 3)
wc, wa = torch.from_numpy(dcol).to(dev), torch.from_numpy(dacc).to(dev)
m2d = torch.zeros((P, 3), device=dev, requires_grad=True)
names = ["activated", "forward", "backward", "step_model", "step"]
acc = {k: [] for k in names}
def step(record):
    t0 = time.perf_counter()
    xyz, op, sc, rot, shs = model.activated()
    t1 = time.perf_counter()
    out = raster(xyz, m2d, op, shs=shs, scales=sc, rotations=rot)
    t2 = time.perf_counter()
    m2d.grad = None
    torch.autograd.backward([out[0], out[3]], [wc, wa])
    t3 = time.perf_counter()
    opt.step_model(model)
    t4 = time.perf_counter()
    if record:
        for k, v in zip(names, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)):
            acc[k].append(v * 1e6)
for _ in range(50): step(False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): step(True)
torch.cuda.synchronize()
print("P=%d host=%s: %.1f us/step over %d steps" % (P, host, (time.perf_counter() - t0) / steps * 1e6, steps))
for k in names:
    a = np.array(acc[k])
    print("  %-11s mean %7.1f  p10 %7.1f  p50 %7.1f  p90 %7.1f us" % (k, a.mean(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90)))
