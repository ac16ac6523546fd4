"""Host-side mirror of the reference's image loss (include/gs/gs/loss_utils.cuh) on the fused kernels of
csrc/loss.hip (SURVEY.md section 8(f) "next" row 2)."""
import math

import torch

from . import _capi


def reference_window_1d(window_size=11, sigma=1.5):
    """The 1-D kernel of gaussian_splatting::gaussian (loss_utils.cuh:24-31), bug for bug: the exponent uses
    floor((x - window_size) / 2), not (x - window_size // 2), so the window is NOT the centred Gaussian of the
    original SSIM code.  The 2-D window of create_window (:33-37) is the outer product of this vector."""
    g = [math.exp(-(math.floor((x - window_size) / 2.0) ** 2) / (2.0 * sigma * sigma)) for x in range(window_size)]
    t = torch.tensor(g, dtype=torch.float32)
    return t / t.sum()


class PhotometricLoss(torch.autograd.Function):
    """loss = (1 - lambda_dssim) * l1_loss(img, gt) + lambda_dssim * (1 - ssim(img, gt))
    (src/liw/lioOptimization.cpp:1705-1710); gradient w.r.t. img only (gt is data)."""

    @staticmethod
    def forward(ctx, img, gt, window11, lambda_dssim):
        out3, grad = _capi.photometric_loss(img, gt, window11.tolist(), lambda_dssim, want_grad=img.requires_grad)
        ctx.save_for_backward(grad if grad is not None else torch.empty(0, device=img.device))
        ctx.parts = out3  # [loss, l1, ssim] for logging (the reference prints PSNR/SSIM every 50 iterations)
        return out3[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g if grad.numel() else None), None, None, None


def photometric_loss(img, gt, lambda_dssim=0.2, window11=None):
    """Drop-in for the reference's three lines at lioOptimization.cpp:1705-1710 (lambda_dssim:
    config/basic_common.yaml:63)."""
    if window11 is None:
        window11 = reference_window_1d()
    return PhotometricLoss.apply(img, gt, window11, float(lambda_dssim))
