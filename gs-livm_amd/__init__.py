"""gs_livm_amd -- MI355X-native tile rasterizer for GS-LIVM's render/optimisation hot path.

Layout (only what the hot path needs):
  csrc/            hand-written HIP kernels (gfx950) + the C ABI (include/gsraster.h) + LibTorch binding
  _capi.py         ctypes binding of the C ABI (mirrors src/gs/rasterize_points.cu)
  rasterizer.py    host-side mirror of the reference operator surface (src/gs/rasterizer.cu)
  render_utils.py  Camera (rasterizer-facing part) and render() as in include/gs/gs/render_utils.cuh
  loss.py          fused L1 + SSIM photometric loss (SURVEY.md 8(f) "next" row 2)
  model.py         fused activations + Adam for the caller's leaf tensors (SURVEY.md 8(f) "next" row 1)
  synthetic.py     synthetic scenes of SURVEY.md section 8(d) for tests and bench
  multiview.py     view-parallel sharding + the two RCCL exchange steps
  build.py         hipcc / g++ build recipe

The directory is named `gs-livm_amd`; import it as `gs_livm_amd` (alias module at the repo root).
"""
from . import multiview, synthetic  # noqa: F401
from ._capi import (GsrError, LIB_PATH, NumRendered, last_num_rendered, lib, mailbox_slow_path_last, mark_visible,
                    set_binning_capacity_hint, speculation_stats, set_near_far, set_near_far_thread, set_reference_rects_thread, async_outcomes_pending, set_near_far_hints, last_near_far, set_far_speculation, last_far_skipped, profile_enable, profile_read,  # noqa: F401
                    rasterize_backward, rasterize_forward, reference_rects, set_reference_rects, state_views)
from .loss import PhotometricLoss, photometric_loss, reference_window_1d  # noqa: F401
from .model import FusedActivations, FusedAdam, GaussianParameters, GrowableAdam, GrowableGaussians  # noqa: F401
from . import ply  # noqa: F401
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                         rasterize_gaussians)
from .render_utils import Camera, get_projection_matrix, render  # noqa: F401



def torch_ops():
    """The C++/LibTorch operator surface (csrc/torch_binding.cpp), built in-tree as _gsraster_torch.so.
    This is the code GS-LIVM itself would link; raises when it has not been built (no fallback)."""
    import importlib.util
    import os
    import torch  # noqa: F401  (libtorch must be loaded first)
    lib()  # libgsraster_hip.so must resolve before the extension that links it
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_gsraster_torch.so")
    if not os.path.exists(path):
        raise RuntimeError("_gsraster_torch.so not built: run `python gs-livm_amd/build.py`")
    spec = importlib.util.spec_from_file_location("_gsraster_torch", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


__all__ = ["PhotometricLoss", "photometric_loss", "reference_window_1d", "FusedActivations", "FusedAdam", "GaussianParameters", "GrowableAdam", "GrowableGaussians", "ply", "GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "Camera", "render", "get_projection_matrix", "rasterize_forward",
           "rasterize_backward", "mark_visible", "state_views", "set_reference_rects", "reference_rects", "last_num_rendered", "set_binning_capacity_hint", "speculation_stats", "mailbox_slow_path_last", "set_near_far", "set_near_far_thread", "set_reference_rects_thread", "async_outcomes_pending", "set_near_far_hints", "last_near_far", "set_far_speculation", "last_far_skipped", "NumRendered", "lib", "torch_ops", "synthetic", "multiview", "GsrError", "LIB_PATH"]
