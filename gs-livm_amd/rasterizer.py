"""Host-side mirror of the reference operator surface (include/gs/gs/rasterizer.cuh:8-80,
src/gs/rasterizer.cu) for Python hosts: same names, argument meaning and error behaviour.
The C++/LibTorch equivalent (what GS-LIVM itself links) is csrc/torch_binding.cpp.
"""
from typing import NamedTuple, Optional

import torch

from . import _capi


class GaussianRasterizationSettings(NamedTuple):
    """include/gs/gs/rasterizer.cuh:8-20"""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    camera_center: torch.Tensor
    prefiltered: bool = False


def _empty(device):
    return torch.empty(0, device=device)


class _RasterizeGaussians(torch.autograd.Function):
    """src/gs/rasterizer.cu:6-149.  The reference boxes its scalars into 0-dim CUDA tensors and reads
    them back with .item() (7 host syncs per render, rasterizer.cu:27-33); here they travel by value."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings):
        s = settings
        R, color, depth, acc, radii, geom, binning, img = _capi.rasterize_forward(
            s.bg, means3D.contiguous(), colors_precomp.contiguous(), opacities.contiguous(), scales.contiguous(),
            rotations.contiguous(), s.scale_modifier, cov3Ds_precomp.contiguous(), s.viewmatrix.contiguous(),
            s.projmatrix.contiguous(), s.tanfovx, s.tanfovy, s.image_height, s.image_width, sh.contiguous(),
            s.sh_degree, s.camera_center.contiguous(), s.prefiltered, False)
        ctx.settings, ctx.num_rendered = s, R
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning,
                              img)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)  # no zero image for the depth output nobody differentiates
        return color, radii, depth, acc

    @staticmethod
    def backward(ctx, grad_color, _grad_radii, _grad_depth, grad_acc):
        # the depth gradient is ignored, exactly as in the reference (rasterizer.cu:78-79, 117-118)
        s = ctx.settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img = ctx.saved_tensors
        if grad_color is None:
            grad_color = torch.zeros((3, s.image_height, s.image_width), device=means3D.device)
        if grad_acc is None:
            grad_acc = torch.zeros((1, s.image_height, s.image_width), device=means3D.device)
        (g_means2D, g_colors, g_opac, g_means3D, g_cov3D, g_sh, g_scales, g_rot) = _capi.rasterize_backward(
            s.bg, means3D.contiguous(), radii, colors_precomp.contiguous(), scales.contiguous(),
            rotations.contiguous(), s.scale_modifier, cov3Ds_precomp.contiguous(), s.viewmatrix.contiguous(),
            s.projmatrix.contiguous(), s.tanfovx, s.tanfovy, grad_color, grad_acc, sh.contiguous(), s.sh_degree,
            s.camera_center.contiguous(), geom, ctx.num_rendered, binning, img, False,
            want_cov3D=cov3Ds_precomp.numel() != 0)
        none_if_empty = lambda g, x: g if x.numel() else None  # noqa: E731
        return (g_means3D, g_means2D, none_if_empty(g_sh, sh), none_if_empty(g_colors, colors_precomp), g_opac,
                none_if_empty(g_scales, scales), none_if_empty(g_rot, rotations),
                none_if_empty(g_cov3D, cov3Ds_precomp), None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    """src/gs/rasterizer.cu:208-288: moves stray host tensors to the device, then applies the op."""
    dev = means3D.device if means3D.is_cuda else torch.device("cuda")
    mv = lambda t: t if t.device == dev else t.to(dev)  # noqa: E731
    s = raster_settings._replace(bg=mv(raster_settings.bg), viewmatrix=mv(raster_settings.viewmatrix),
                                 projmatrix=mv(raster_settings.projmatrix),
                                 camera_center=mv(raster_settings.camera_center))
    return _RasterizeGaussians.apply(mv(means3D), mv(means2D), mv(sh), mv(colors_precomp), mv(opacities), mv(scales),
                                     mv(rotations), mv(cov3Ds_precomp), s)


class GaussianRasterizer(torch.nn.Module):
    """include/gs/gs/rasterizer.cuh:51-80, src/gs/rasterizer.cu:152-206."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def mark_visible(self, positions):
        with torch.no_grad():
            s = self.raster_settings
            return _capi.mark_visible(positions.contiguous(), s.viewmatrix.contiguous(), s.projmatrix.contiguous())

    def forward(self, means3D, means2D, opacities, shs: Optional[torch.Tensor] = None,
                colors_precomp: Optional[torch.Tensor] = None, scales: Optional[torch.Tensor] = None,
                rotations: Optional[torch.Tensor] = None, cov3D_precomp: Optional[torch.Tensor] = None):
        if (shs is None) == (colors_precomp is None):
            raise ValueError("Please provide exactly one of either SHs or precomputed colors!")
        if ((scales is not None or rotations is not None) and cov3D_precomp is not None) or \
                (scales is None and rotations is None and cov3D_precomp is None):
            raise ValueError("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        # The reference additionally exits the process when shs / scales / rotations are undefined
        # (rasterizer.cu:173-190), i.e. it only ever runs the SH + scale/rotation path; the kernels
        # support the precomputed inputs, so they are accepted here.
        dev = means3D.device
        shs = _empty(dev) if shs is None else shs
        colors_precomp = _empty(dev) if colors_precomp is None else colors_precomp
        scales = _empty(dev) if scales is None else scales
        rotations = _empty(dev) if rotations is None else rotations
        cov3D_precomp = _empty(dev) if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings)
