"""gs_livm_amd -- MI355X-native tile rasterizer for GS-LIVM's render/optimisation hot path.

Layout (only what the hot path needs):
  csrc/            hand-written HIP kernels (gfx950) + the C ABI (include/gsraster.h) + LibTorch binding
  _capi.py         ctypes binding of the C ABI (mirrors src/gs/rasterize_points.cu)
  rasterizer.py    host-side mirror of the reference operator surface (src/gs/rasterizer.cu)
  synthetic.py     synthetic scenes of SURVEY.md section 8(d) for tests and bench
  multiview.py     view-parallel sharding + the two RCCL exchange steps
  build.py         hipcc / g++ build recipe

The directory is named `gs-livm_amd`; import it as `gs_livm_amd` (alias module at the repo root).
"""
from . import multiview, synthetic  # noqa: F401
from ._capi import (GsrError, LIB_PATH, lib, mark_visible, profile_enable, profile_read,  # noqa: F401
                    rasterize_backward, rasterize_forward, state_views)
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                         rasterize_gaussians)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "rasterize_forward",
           "rasterize_backward", "mark_visible", "state_views", "lib", "synthetic", "multiview", "GsrError", "LIB_PATH"]
