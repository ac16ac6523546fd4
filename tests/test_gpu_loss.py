"""GPU tests of the fused photometric loss (csrc/loss.hip; SURVEY.md 8(f) "next" row 2) against a plain
PyTorch f32 restatement of the reference's l1_loss / ssim (include/gs/gs/loss_utils.cuh:11-13, 43-70), built from
grouped conv2d exactly as the reference does, with the reference's own (asymmetric) window."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import gs_livm_amd as G

pytestmark = pytest.mark.gpu
C1, C2 = 0.01 * 0.01, 0.03 * 0.03


def ref_ssim(img1, img2, w1d):
    ch = img1.shape[0]
    window = (w1d[:, None] @ w1d[None, :])[None, None].expand(ch, 1, 11, 11).contiguous()
    conv = lambda x: F.conv2d(x[None], window, padding=5, groups=ch)[0]  # noqa: E731
    mu1, mu2 = conv(img1), conv(img2)
    s1 = conv(img1 * img1) - mu1 * mu1
    s2 = conv(img2 * img2) - mu2 * mu2
    s12 = conv(img1 * img2) - mu1 * mu2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))
    return m.mean()


def ref_loss(img, gt, w1d, lam):
    return (1 - lam) * (img - gt).abs().mean() + lam * (1 - ref_ssim(img, gt, w1d))


@pytest.mark.parametrize("shape,lam", [((3, 45, 67), 0.2), ((3, 300, 200), 0.2), ((1, 16, 16), 0.5), ((3, 9, 7), 0.0),
                                       ((3, 128, 130), 1.0),
                                       # the shapes the bench and the product run (54 x 32-pixel work units:
                                       # 1920 = 35.6 units wide, 1080 = 33.75 high; 640 x 512 = 11.9 x 16)
                                       ((3, 1080, 1920), 0.2), ((3, 512, 640), 0.2)])
def test_fused_loss_matches_torch(shape, lam, gpu_device):
    gen = torch.Generator().manual_seed(shape[1])
    img = torch.rand(shape, generator=gen).to(gpu_device).requires_grad_(True)
    gt = torch.rand(shape, generator=gen).to(gpu_device)
    # a structured target makes SSIM non-trivial
    gt = (0.6 * gt + 0.4 * img.detach().roll(1, 2)).clamp(0, 1)
    w = G.reference_window_1d()
    want = ref_loss(img, gt, w.to(gpu_device), lam)
    (gw,) = torch.autograd.grad(want, img)
    img2 = img.detach().clone().requires_grad_(True)
    got = G.photometric_loss(img2, gt, lam)
    (gg,) = torch.autograd.grad(3.0 * got, img2)  # upstream scale is honoured
    assert abs(float(got) - float(want)) <= 2e-6 * max(1.0, abs(float(want)))
    scale = float(gw.abs().max())
    assert float((gg / 3.0 - gw).abs().max()) <= 2e-5 * scale + 1e-9


def test_symmetric_window_and_determinism(gpu_device):
    gen = torch.Generator().manual_seed(5)
    img = torch.rand((3, 100, 90), generator=gen).to(gpu_device).requires_grad_(True)
    gt = torch.rand((3, 100, 90), generator=gen).to(gpu_device)
    x = torch.arange(11, dtype=torch.float32) - 5
    w = torch.exp(-x * x / 4.5)
    w = w / w.sum()  # the centred window of the original SSIM code
    want = ref_loss(img, gt, w.to(gpu_device), 0.2)
    a = G.photometric_loss(img, gt, 0.2, window11=w)
    b = G.photometric_loss(img, gt, 0.2, window11=w)
    assert float(a) == float(b)  # fixed-order reduction
    assert abs(float(a) - float(want)) <= 2e-6
