"""Times the fused photometric loss (forward + backward kernels) at one image shape (GPU):
    python tools/time_loss.py [H] [W] [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import gs_livm_amd as G

H = int(sys.argv[1]) if len(sys.argv) > 1 else 1080
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(1)
img = torch.rand((3, H, W), generator=gen).to(dev).requires_grad_(True)
gt = (0.6 * torch.rand((3, H, W), generator=gen).to(dev) + 0.4 * img.detach().roll(1, 2)).clamp(0, 1)
for _ in range(5):
    loss = G.photometric_loss(img, gt, 0.2)
    (g,) = torch.autograd.grad(loss, img)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    loss = G.photometric_loss(img, gt, 0.2)
    (g,) = torch.autograd.grad(loss, img)
e1.record()
torch.cuda.synchronize()
print("%dx%d: %.1f us per loss forward+backward (torch autograd wrapper included), loss %.6f" % (W, H, e0.elapsed_time(e1) * 1e3 / iters, float(loss)))
G.profile_enable(True)
for _ in range(10):
    loss = G.photometric_loss(img, gt, 0.2)
    (g,) = torch.autograd.grad(loss, img)
torch.cuda.synchronize()
G.profile_enable(False)
for k, (ms, c) in sorted(G.profile_read().items(), key=lambda kv: -kv[1][0]):
    if c:
        print("  %-22s %8.1f us per launch  x%d" % (k, ms * 1e3 / c, c))
