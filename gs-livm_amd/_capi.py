"""ctypes binding of the C ABI in include/gsraster.h (libgsraster_hip.so).

This is the reference-side binding a Python host would use; the C++/LibTorch operator surface
(csrc/torch_binding.cpp) goes through the same ABI.  Device memory, streams and allocation are
PyTorch-ROCm plumbing; all compute is in the HIP library.  There is NO CPU fallback: if the
library is missing or a call fails, this module raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSR_LIB", os.path.join(_HERE, "libgsraster_hip.so"))  # GSR_LIB: A/B experiment builds

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)


class GeometryView(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("depths", "radii", "splats", "cov3D", "tiles_touched", "point_offsets", "clamped", "depth_order",
                 "num_rendered", "gpack")]


class BinningView(C.Structure):
    _fields_ = [("point_list", C.c_void_p)]


class ImageView(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ranges", "final_T", "n_contrib", "quad_last", "ranges_far")]


class MailboxEvent(C.Structure):
    _fields_ = [("count", C.c_uint), ("ticket_expected", C.c_uint32), ("ticket_seen_before_query", C.c_uint32),
                ("elapsed_us", C.c_double), ("ticket_seen_after_query", C.c_uint32), ("first_query_result", C.c_int),
                ("visible_at_query", C.c_int), ("queries", C.c_uint), ("longest_query_us", C.c_double),
                ("longest_poll_gap_us", C.c_double)]


class NumRendered(int):
    """num_rendered as the reference returns it (RasterizeGaussiansCUDA's first element) -- the exact instance count --
    carrying `key`: what gsr_forward returned, the capacity the binning blob was carved for (== the count after a
    synchronous forward, >= it after a speculative one).  rasterize_backward / state_views take the key from it."""

    def __new__(cls, exact, key):
        obj = super().__new__(cls, exact)
        obj.key = int(key)
        return obj


def _key(R):
    return int(getattr(R, "key", R))


_lib = None

EXPORTS = ("gsr_forward", "gsr_backward", "gsr_mark_visible", "gsr_geometry_bytes", "gsr_image_bytes",
           "gsr_binning_bytes", "gsr_geometry_view_of", "gsr_binning_view_of", "gsr_image_view_of",
           "gsr_higher_msb", "gsr_last_error", "gsr_abi_version", "gsr_kernel_count", "gsr_kernel_name",
           "gsr_profile_enable", "gsr_profile_enable_only", "gsr_profile_read", "gsr_mailbox_slow_path_hits", "gsr_activate", "gsr_activate_backward", "gsr_adam_step",
           "gsr_photometric_loss", "gsr_photometric_loss_workspace", "gsr_init_gaussians", "gsr_ply_row_floats",
           "gsr_pack_ply_rows", "gsr_model_step", "gsr_set_reference_rects", "gsr_reference_rects",
           "gsr_last_num_rendered", "gsr_set_binning_capacity_hint", "gsr_speculative_forwards",
           "gsr_speculation_overflows", "gsr_mailbox_slow_path_last", "gsr_set_near_far", "gsr_near_far",
           "gsr_last_near_far", "gsr_set_near_far_hints", "gsr_near_far_forwards", "gsr_set_far_speculation",
           "gsr_last_far_skipped", "gsr_far_skips", "gsr_far_skip_misses", "gsr_async_far_frames",
           "gsr_near_budget_scale", "gsr_near_budget_feedback", "gsr_near_far_pause", "gsr_set_near_far_thread",
           "gsr_set_reference_rects_thread", "gsr_async_outcomes_pending", "gsr_async_outcomes_lost",
           "gsr_frame_note_misses")


def lib():
    """Loads libgsraster_hip.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libgsraster_hip.so not built: run `python gs-livm_amd/build.py` (needs hipcc). "
            "gs_livm_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, ci, cf, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    L.gsr_forward.restype = ci
    L.gsr_forward.argtypes = [ALLOC_FN, vp, ALLOC_FN, vp, ALLOC_FN, vp, ci, ci, ci, vp, ci, ci, vp, vp, vp, vp, vp, cf,
                              vp, vp, vp, vp, vp, cf, cf, ci, vp, vp, vp, vp, ci, vp]
    L.gsr_backward.restype = ci
    L.gsr_backward.argtypes = [ci, ci, ci, ci, vp, ci, ci, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp, cf, cf, vp, vp, vp,
                               vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, vp]
    L.gsr_mark_visible.restype = ci
    L.gsr_mark_visible.argtypes = [ci, vp, vp, vp, vp, vp]
    for n in ("gsr_geometry_bytes", "gsr_binning_bytes"):
        getattr(L, n).restype = sz
        getattr(L, n).argtypes = [ci]
    L.gsr_image_bytes.restype = sz
    L.gsr_image_bytes.argtypes = [ci, ci]
    L.gsr_geometry_view_of.argtypes = [vp, ci, C.POINTER(GeometryView)]
    L.gsr_binning_view_of.argtypes = [vp, ci, C.POINTER(BinningView)]
    L.gsr_image_view_of.argtypes = [vp, ci, ci, C.POINTER(ImageView)]
    L.gsr_higher_msb.restype = C.c_uint32
    L.gsr_higher_msb.argtypes = [C.c_uint32]
    L.gsr_last_error.restype = C.c_char_p
    L.gsr_abi_version.restype = ci
    L.gsr_kernel_count.restype = ci
    L.gsr_kernel_name.restype = C.c_char_p
    L.gsr_kernel_name.argtypes = [ci]
    L.gsr_profile_enable.argtypes = [ci]
    L.gsr_profile_enable_only.argtypes = [C.POINTER(ci), ci]
    L.gsr_set_reference_rects.restype = ci
    L.gsr_set_reference_rects.argtypes = [ci]
    L.gsr_reference_rects.restype = ci
    L.gsr_reference_rects.argtypes = []
    L.gsr_last_num_rendered.restype = ci
    L.gsr_last_num_rendered.argtypes = []
    L.gsr_set_binning_capacity_hint.restype = C.c_longlong
    L.gsr_set_binning_capacity_hint.argtypes = [C.c_longlong]
    for n in ("gsr_speculative_forwards", "gsr_speculation_overflows"):
        getattr(L, n).restype = C.c_ulonglong
        getattr(L, n).argtypes = []
    L.gsr_set_near_far.restype = ci
    L.gsr_set_near_far.argtypes = [ci]
    L.gsr_near_far.restype = ci
    L.gsr_near_far.argtypes = []
    L.gsr_last_near_far.restype = ci
    L.gsr_last_near_far.argtypes = [C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    L.gsr_set_near_far_hints.restype = None
    L.gsr_set_near_far_hints.argtypes = [C.c_longlong, C.c_longlong]
    L.gsr_near_far_forwards.restype = C.c_ulonglong
    L.gsr_near_far_forwards.argtypes = []
    L.gsr_set_far_speculation.restype = ci
    L.gsr_set_far_speculation.argtypes = [ci]
    L.gsr_last_far_skipped.restype = ci
    L.gsr_last_far_skipped.argtypes = []
    L.gsr_near_budget_scale.restype = C.c_uint
    L.gsr_near_budget_scale.argtypes = []
    L.gsr_near_budget_feedback.restype = C.c_uint
    L.gsr_near_budget_feedback.argtypes = [C.c_uint, C.c_uint, C.c_uint]
    L.gsr_near_far_pause.restype = ci
    L.gsr_near_far_pause.argtypes = [ci]
    for n in ("gsr_far_skips", "gsr_far_skip_misses", "gsr_async_far_frames", "gsr_async_outcomes_lost",
              "gsr_frame_note_misses"):
        getattr(L, n).restype = C.c_ulonglong
        getattr(L, n).argtypes = []
    for n in ("gsr_set_near_far_thread", "gsr_set_reference_rects_thread"):
        getattr(L, n).restype = ci
        getattr(L, n).argtypes = [ci]
    L.gsr_async_outcomes_pending.restype = ci
    L.gsr_async_outcomes_pending.argtypes = []
    L.gsr_mailbox_slow_path_last.restype = ci
    L.gsr_mailbox_slow_path_last.argtypes = [C.POINTER(MailboxEvent)]
    L.gsr_mailbox_slow_path_hits.restype = C.c_ulonglong
    L.gsr_mailbox_slow_path_hits.argtypes = []
    L.gsr_profile_read.restype = ci
    L.gsr_profile_read.argtypes = [ci, C.POINTER(C.c_double), C.POINTER(ci)]
    L.gsr_activate.restype = ci
    L.gsr_activate.argtypes = [ci, ci] + [vp] * 9 + [vp]
    L.gsr_activate_backward.restype = ci
    L.gsr_activate_backward.argtypes = [ci, ci] + [vp] * 12 + [vp]
    L.gsr_photometric_loss_workspace.restype = sz
    L.gsr_photometric_loss_workspace.argtypes = [ci, ci, ci]
    L.gsr_photometric_loss.restype = ci
    L.gsr_photometric_loss.argtypes = [ci, ci, ci, vp, vp, C.POINTER(cf), cf, vp, vp, vp, sz, vp]
    L.gsr_init_gaussians.restype = ci
    L.gsr_init_gaussians.argtypes = [ci, ci, vp, vp, vp, cf] + [vp] * 6 + [vp]
    L.gsr_ply_row_floats.restype = sz
    L.gsr_ply_row_floats.argtypes = [ci]
    L.gsr_pack_ply_rows.restype = ci
    L.gsr_pack_ply_rows.argtypes = [ci, ci] + [vp] * 7 + [vp]
    L.gsr_model_step.restype = ci
    L.gsr_model_step.argtypes = [ci, ci, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)] + [vp] * 9 + \
        [C.POINTER(cf), C.c_double, C.c_double, C.c_double, ci, vp]
    L.gsr_adam_step.restype = ci
    L.gsr_adam_step.argtypes = [ci, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(sz),
                                C.POINTER(cf), C.c_double, C.c_double, C.c_double, ci, ci, vp]
    _lib = L
    return L


class GsrError(RuntimeError):
    pass


def _check(code):
    if code < 0:
        raise GsrError("libgsraster_hip error %d: %s" % (code, lib().gsr_last_error().decode()))
    return code


def _ptr(t):
    """Device pointer of a tensor; size-0 / None -> NULL (the reference's convention,
    src/gs/rasterizer.cu:178-193)."""
    if t is None or t.numel() == 0:
        return None
    assert t.is_cuda and t.is_contiguous(), "expects contiguous device tensors"
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Blob:
    """Allocator callback target: replaces resizeFunctional (src/gs/rasterize_points.cu:36-44)."""

    def __init__(self, device):
        self.device = device
        self.tensor = torch.empty(0, dtype=torch.uint8, device=device)
        self.fn = ALLOC_FN(self._alloc)

    def _alloc(self, _ctx, nbytes):
        try:
            self.tensor = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            return self.tensor.data_ptr()
        except Exception:  # pragma: no cover - surfaces as GSR_ERR_ALLOC
            return None


def rasterize_forward(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                      viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                      prefiltered=False, debug=False, radii_out=True):
    """Mirror of RasterizeGaussiansCUDA (src/gs/rasterize_points.cu:46-130) over the C ABI.
    Returns (num_rendered, out_color, out_depth, out_acc, radii, geomBuffer, binningBuffer, imgBuffer)."""
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise ValueError("means3D must have dimensions (num_points, 3)")
    dev = means3D.device
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    f32 = dict(dtype=torch.float32, device=dev)
    out_color = torch.empty((3, H, W), **f32)
    out_depth = torch.empty((1, H, W), **f32)
    out_acc = torch.empty((1, H, W), **f32)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)  # the library writes every entry
    gb, bb, ib = _Blob(dev), _Blob(dev), _Blob(dev)
    M = int(sh.size(1)) if (sh is not None and sh.numel() != 0) else 0
    L = lib()
    R = _check(L.gsr_forward(gb.fn, None, bb.fn, None, ib.fn, None, P, int(degree), M, _ptr(background), W, H,
                             _ptr(means3D), _ptr(sh), _ptr(colors), _ptr(opacity), _ptr(scales),
                             float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), _ptr(viewmatrix),
                             _ptr(projmatrix), _ptr(campos), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)),
                             _ptr(out_color), _ptr(out_depth), _ptr(out_acc), _ptr(radii) if radii_out else None,
                             int(bool(debug)), _stream()))
    R = NumRendered(L.gsr_last_num_rendered() if P else 0, R)
    return R, out_color, out_depth, out_acc, radii, gb.tensor, bb.tensor, ib.tensor


def rasterize_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                       viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_acc, sh, degree, campos,
                       geomBuffer, R, binningBuffer, imageBuffer, debug=False, return_conic=False, want_cov3D=True):
    """Mirror of RasterizeGaussiansBackwardCUDA (src/gs/rasterize_points.cu:132-224).
    Returns (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)
    [+ dL_dconic when return_conic].  want_cov3D=False (only without cov3D_precomp): dL_dcov3D -- then an intermediate
    nobody reads -- is neither allocated nor written and comes back as None."""
    dev = means3D.device
    P = int(means3D.size(0))
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = int(sh.size(1)) if (sh is not None and sh.numel() != 0) else 0
    f32 = dict(dtype=torch.float32, device=dev)
    # the library overwrites every element, so torch.empty replaces the reference's nine torch::zeros
    # (rasterize_points.cu:173-181); P == 0 needs the zeros (nothing runs).
    mk = torch.zeros if P == 0 else torch.empty
    dL_dmeans3D = mk((P, 3), **f32)
    dL_dmeans2D = mk((P, 3), **f32)
    dL_dcolors = mk((P, 3), **f32)
    dL_dconic = mk((P, 2, 2), **f32)
    dL_dopacity = mk((P, 1), **f32)
    dL_dcov3D = mk((P, 6), **f32) if want_cov3D else None
    dL_dsh = mk((P, M, 3), **f32)
    dL_dscales = mk((P, 3), **f32)
    dL_drotations = mk((P, 4), **f32)
    if P != 0:
        dpix = dL_dout_color.contiguous()
        dacc = dL_dout_acc.contiguous()
        _check(lib().gsr_backward(P, int(degree), M, _key(R), _ptr(background), W, H, _ptr(means3D), _ptr(sh),
                                  _ptr(colors), _ptr(scales), float(scale_modifier), _ptr(rotations),
                                  _ptr(cov3D_precomp), _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos),
                                  float(tan_fovx), float(tan_fovy), _ptr(radii), _ptr(geomBuffer),
                                  _ptr(binningBuffer), _ptr(imageBuffer), _ptr(dpix), _ptr(dacc), _ptr(dL_dmeans2D),
                                  _ptr(dL_dconic), _ptr(dL_dopacity), _ptr(dL_dcolors), _ptr(dL_dmeans3D),
                                  _ptr(dL_dcov3D), _ptr(dL_dsh), _ptr(dL_dscales), _ptr(dL_drotations),
                                  int(bool(debug)), _stream()))
    out = (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)
    return out + (dL_dconic,) if return_conic else out


def last_num_rendered():
    """Exact instance count of the calling thread's last forward (no device access)."""
    return int(lib().gsr_last_num_rendered())


def set_binning_capacity_hint(capacity):
    """Test / tuning hook (include/gsraster.h): the calling thread's NEXT forward allocates its binning blob for
    `capacity` instances (smaller than the frame's count forces the overflow path); 0 = forget the history (next
    forward synchronous); None = no override."""
    return int(lib().gsr_set_binning_capacity_hint(-1 if capacity is None else int(capacity)))


def speculation_stats():
    L = lib()
    return dict(speculative_forwards=int(L.gsr_speculative_forwards()), overflows=int(L.gsr_speculation_overflows()),
                near_far_forwards=int(L.gsr_near_far_forwards()), far_skips=int(L.gsr_far_skips()),
                far_skip_misses=int(L.gsr_far_skip_misses()), async_far_frames=int(L.gsr_async_far_frames()),
                async_outcomes_lost=int(L.gsr_async_outcomes_lost()), frame_note_misses=int(L.gsr_frame_note_misses()),
                near_budget_scale_q8=int(L.gsr_near_budget_scale()),
                mailbox_slow_path_hits=int(L.gsr_mailbox_slow_path_hits()))


def mailbox_slow_path_last():
    ev = MailboxEvent()
    _check(lib().gsr_mailbox_slow_path_last(C.byref(ev)))
    return {n: getattr(ev, n) for n, _ in MailboxEvent._fields_}


def set_near_far(on):
    """Near/far frames (include/gsraster.h): speculative forwards of dense scenes bin the nearest Gaussians first and
    the rest only where a tile is still unfinished.  Returns the previous setting."""
    return bool(lib().gsr_set_near_far(int(bool(on))))


def set_near_far_thread(mode):
    """The same switch for the CALLING THREAD's forwards only: True / False, None = follow the process-wide value.
    Returns the thread's previous setting (None = none)."""
    prev = int(lib().gsr_set_near_far_thread(-1 if mode is None else int(bool(mode))))
    return None if prev < 0 else bool(prev)


def async_outcomes_pending():
    """Asynchronous frames of the calling thread whose far-chain outcome has not reached the host yet (never waits)."""
    return int(lib().gsr_async_outcomes_pending())


def set_near_far_hints(near_entries_per_tile=None, far_capacity=None):
    """Test / tuning hook: near budget in list entries per tile and the far capacity of the calling thread's next
    near/far forward (None = default / from history)."""
    lib().gsr_set_near_far_hints(-1 if near_entries_per_tile is None else int(near_entries_per_tile),
                                 -1 if far_capacity is None else int(far_capacity))


def set_far_speculation(mode):
    """Test / tuning hook (include/gsraster.h): True = the calling thread's next split forward does not enqueue its far
    chain until it has seen the live-tile count, False = never, None = automatic.  Returns the previous override."""
    return int(lib().gsr_set_far_speculation(-1 if mode is None else int(bool(mode))))


def last_far_skipped():
    return bool(lib().gsr_last_far_skipped())


def last_near_far():
    """(was the calling thread's last forward binned near/far, near instances, far instances)"""
    a, b = C.c_uint(0), C.c_uint(0)
    split = bool(lib().gsr_last_near_far(C.byref(a), C.byref(b)))
    return split, int(a.value), int(b.value)


def set_reference_rects(on):
    """Binning mode (include/gsraster.h): True = the reference's full 3-sigma tile squares (auxiliary.h:39-46), so that
    tiles_touched / num_rendered / point_list / ranges / n_contrib are the reference's own; False (default) = the
    footprint-culled rectangles.  Returns the previous setting."""
    return bool(lib().gsr_set_reference_rects(int(bool(on))))


def set_reference_rects_thread(mode):
    """Binning mode of the CALLING THREAD's forwards only: True / False, None = follow the process-wide value.  Returns
    the thread's previous setting (None = none)."""
    prev = int(lib().gsr_set_reference_rects_thread(-1 if mode is None else int(bool(mode))))
    return None if prev < 0 else bool(prev)


def reference_rects():
    return bool(lib().gsr_reference_rects())


def mark_visible(means3D, viewmatrix, projmatrix):
    """Mirror of markVisible (src/gs/rasterize_points.cu:226-241)."""
    P = int(means3D.size(0))
    present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
    if P:
        _check(lib().gsr_mark_visible(P, _ptr(means3D), _ptr(viewmatrix), _ptr(projmatrix),
                                      C.c_void_p(present.data_ptr()), _stream()))
    return present


def _sub(blob, ptr, count, dtype):
    if not ptr or count == 0:
        return None
    off = int(ptr) - blob.data_ptr()
    nbytes = count * torch.empty((), dtype=dtype).element_size()
    return blob[off:off + nbytes].view(dtype)


def state_views(geomBuffer, binningBuffer, imageBuffer, P, R, W, H):
    """Typed tensor views into the three opaque blobs (tests / tooling)."""
    L = lib()
    gv, bv, iv = GeometryView(), BinningView(), ImageView()
    T = ((W + 15) // 16) * ((H + 15) // 16)
    out = {}
    if P:
        _check(L.gsr_geometry_view_of(_ptr(geomBuffer), P, C.byref(gv)))
        splats = _sub(geomBuffer, gv.splats, P * 12, torch.float32).view(P, 12)
        gpack = _sub(geomBuffer, gv.gpack, P * 2, torch.int32).view(P, 2)
        out.update(depths=splats[:, 9],  # (culled Gaussians have no record: compare where radii > 0)
                   radii=_sub(geomBuffer, gv.radii, P, torch.int32),  # valid only if the forward got no radii array
                   splats=splats,
                   cov3D=_sub(geomBuffer, gv.cov3D, P * 6, torch.float32).view(P, 6),  # debug forwards only
                   tiles_touched=gpack[:, 0],
                   point_offsets=_sub(geomBuffer, gv.point_offsets, P, torch.int32),
                   clamped=_sub(geomBuffer, gv.clamped, P, torch.uint8),
                   depth_order=_sub(geomBuffer, gv.depth_order, P, torch.int32))
        _check(L.gsr_image_view_of(_ptr(imageBuffer), W, H, C.byref(iv)))
        out.update(ranges=_sub(imageBuffer, iv.ranges, T * 2, torch.int32).view(T, 2),
                   final_T=_sub(imageBuffer, iv.final_T, W * H, torch.float32).view(H, W),
                   n_contrib=_sub(imageBuffer, iv.n_contrib, W * H, torch.int32).view(H, W),
                   quad_last=_sub(imageBuffer, iv.quad_last, T * 4, torch.int32).view(T, 4))
        counters = _sub(geomBuffer, gv.num_rendered, 16, torch.int32).cpu().tolist()  # the forward's device counters
        split = counters[12] == 1
        exact = counters[6] + counters[8] if split else counters[0]
        out.update(num_rendered=exact, near_far=split, counters=counters)
        key = _key(R)
        if exact:
            _check(L.gsr_binning_view_of(_ptr(binningBuffer), key, C.byref(bv)))
            raw = _sub(binningBuffer, bv.point_list, key, torch.int32)
            # A tile's list = its near segment followed by its far segment (near/far frames; the far ranges are (0, 0)
            # otherwise).  The views present it as the reference does: ONE list per tile, lists back to back in tile
            # order, `ranges` indexing into `point_list`.
            ra = out["ranges"].long()
            rb = _sub(imageBuffer, iv.ranges_far, T * 2, torch.int32).view(T, 2).long()
            seg_start = torch.stack([ra[:, 0], rb[:, 0]], 1).reshape(-1)
            seg_len = torch.stack([ra[:, 1] - ra[:, 0], rb[:, 1] - rb[:, 0]], 1).reshape(-1)
            total = int(seg_len.sum())
            excl = torch.cumsum(seg_len, 0) - seg_len
            idx = torch.repeat_interleave(seg_start - excl, seg_len) + torch.arange(total, device=ra.device)
            point_list = raw[idx]
            ln = seg_len.view(T, 2).sum(1)
            ends = torch.cumsum(ln, 0)
            ranges = torch.stack([ends - ln, ends], 1)
            ranges[ln == 0] = 0  # the reference's memset leaves (0, 0) for tiles without instances
            out.update(ranges_near=out["ranges"], ranges_far=rb.int(), ranges=ranges.int())
            tile_ids = torch.repeat_interleave(torch.arange(T, device=ra.device), ln)
            # the reference's 64-bit sorted keys, recomposed: (tile << 32) | bits(depth of the instance's Gaussian)
            dbits = out["depths"].view(torch.int32)[point_list.long()].long() & 0xFFFFFFFF
            out.update(tile_ids=tile_ids, point_list=point_list, keys=(tile_ids.long() << 32) | dbits)
    return out


def profile_enable(on=True, only=None):
    """Start / stop recording a hipEvent pair around every kernel launch (bench.py's roofline line).
    `only`: iterable of kernel names -- record just those (each recorded launch costs a few us of GPU idle)."""
    L = lib()
    if on and only is not None:
        names = [L.gsr_kernel_name(i).decode() for i in range(L.gsr_kernel_count())]
        ids = [names.index(k) for k in only]
        _check(L.gsr_profile_enable_only((C.c_int * len(ids))(*ids), len(ids)))
    else:
        _check(L.gsr_profile_enable(int(bool(on))))


def profile_read():
    """{kernel name: (total_ms, launches)} since the last read; synchronises the recorded events."""
    L = lib()
    n = L.gsr_kernel_count()
    ms = (C.c_double * n)()
    cnt = (C.c_int * n)()
    _check(L.gsr_profile_read(n, ms, cnt))
    return {L.gsr_kernel_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(n)}


# ---- "next" row: fused activations + Adam (include/gsraster.h; csrc/optimizer.hip) ----
def activate(scaling_raw, rotation_raw, opacity_raw, features_dc, features_rest):
    """GaussianModel's getters in one launch (include/gs/gs/gaussian.cuh:40-54):
    returns (scales [P,3], rotations [P,4], opacities [P,1], shs [P,M,3])."""
    P = int(scaling_raw.size(0))
    M = 1 + (int(features_rest.size(1)) if features_rest is not None and features_rest.numel() else 0)
    f32 = dict(dtype=torch.float32, device=scaling_raw.device)
    scales, rot = torch.empty((P, 3), **f32), torch.empty((P, 4), **f32)
    opac, shs = torch.empty((P, 1), **f32), torch.empty((P, M, 3), **f32)
    _check(lib().gsr_activate(P, M, _ptr(scaling_raw), _ptr(rotation_raw), _ptr(opacity_raw), _ptr(features_dc),
                              _ptr(features_rest), _ptr(scales), _ptr(rot), _ptr(opac), _ptr(shs), _stream()))
    return scales, rot, opac, shs


def activate_backward(rotation_raw, scales, opacities, g_scales, g_rot, g_opac, g_shs):
    """Chain rule of `activate`; returns grads w.r.t. (_scaling, _rotation, _opacity, _features_dc, _features_rest)."""
    P, M = int(scales.size(0)), int(g_shs.size(1))
    f32 = dict(dtype=torch.float32, device=scales.device)
    gs, gr, go = torch.empty((P, 3), **f32), torch.empty((P, 4), **f32), torch.empty((P, 1), **f32)
    gdc, grest = torch.empty((P, 1, 3), **f32), torch.empty((P, M - 1, 3), **f32)
    _check(lib().gsr_activate_backward(P, M, _ptr(rotation_raw), _ptr(scales), _ptr(opacities), _ptr(g_scales),
                                       _ptr(g_rot), _ptr(g_opac), _ptr(g_shs), _ptr(gs), _ptr(gr), _ptr(go),
                                       _ptr(gdc), _ptr(grest), _stream()))
    return gs, gr, go, gdc, grest


def adam_step(params, grads, exp_avg, exp_avg_sq, lrs, beta1, beta2, eps, step, zero_grads=True):
    """torch::optim::Adam::step for up to 8 tensors in one launch (in place)."""
    n = len(params)
    VP = C.c_void_p * n
    arr = lambda ts: VP(*[t.data_ptr() if t.numel() else None for t in ts])  # noqa: E731
    numel = (C.c_size_t * n)(*[int(t.numel()) for t in params])
    lr = (C.c_float * n)(*[float(x) for x in lrs])
    for t in list(params) + list(grads) + list(exp_avg) + list(exp_avg_sq):
        assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
    _check(lib().gsr_adam_step(n, arr(params), arr(grads), arr(exp_avg), arr(exp_avg_sq), numel, lr, float(beta1),
                               float(beta2), float(eps), int(step), int(bool(zero_grads)), _stream()))


def photometric_loss(img, gt, window11, lambda_dssim, want_grad=True):
    """(1 - lambda) * L1 + lambda * (1 - SSIM) in three launches (include/gsraster.h).
    Returns (out3 = [loss, l1, ssim] device tensor, dL_dimg or None)."""
    assert img.dim() == 3 and img.shape == gt.shape and img.is_cuda
    Cn, H, W = (int(x) for x in img.shape)
    img, gt = img.contiguous(), gt.contiguous()
    nbytes = int(lib().gsr_photometric_loss_workspace(Cn, H, W))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=img.device)
    out3 = torch.empty(3, dtype=torch.float32, device=img.device)
    grad = torch.empty_like(img) if want_grad else None
    win = (C.c_float * 11)(*[float(x) for x in window11])
    _check(lib().gsr_photometric_loss(Cn, H, W, _ptr(img), _ptr(gt), win, float(lambda_dssim), _ptr(out3), _ptr(grad),
                                      _ptr(ws), nbytes, _stream()))
    return out3, grad


# ---- "next" row 4: map growth and PLY export (include/gsraster.h; csrc/growth.hip) ----
def init_gaussians(xyz, covs, rgbs, scale_factor, out_xyz, out_features_dc, out_features_rest, out_scaling,
                   out_rotation, out_opacity):
    """GaussianModel::addNewPointcloud's arithmetic (src/gs/gaussian.cu:241-313) for n new points, written in
    place into the given (contiguous, n-row) output views -- normally the tail rows of capacity buffers."""
    n = int(xyz.size(0))
    M = 1 + (int(out_features_rest.size(1)) if out_features_rest is not None and out_features_rest.numel() else 0)
    for t in (xyz, covs, rgbs, out_xyz, out_features_dc, out_scaling, out_rotation, out_opacity):
        assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
    assert covs.shape == (n, 3, 3) and rgbs.shape == (n, 3)
    _check(lib().gsr_init_gaussians(n, M, _ptr(xyz), _ptr(covs), _ptr(rgbs), float(scale_factor), _ptr(out_xyz),
                                    _ptr(out_features_dc), _ptr(out_features_rest), _ptr(out_scaling),
                                    _ptr(out_rotation), _ptr(out_opacity), _stream()))


def pack_ply_rows(xyz, features_dc, features_rest, opacity, scaling, rotation):
    """[P, 14 + 3M] f32 device tensor: the vertex rows of the reference's PLY export (one coalesced pass)."""
    P = int(xyz.size(0))
    M = 1 + (int(features_rest.size(1)) if features_rest is not None and features_rest.numel() else 0)
    rf = int(lib().gsr_ply_row_floats(M))
    rows = torch.empty((P, rf), dtype=torch.float32, device=xyz.device)
    args = [t.contiguous() for t in (xyz, features_dc, features_rest, opacity, scaling, rotation)]
    _check(lib().gsr_pack_ply_rows(P, M, *[_ptr(t) for t in args], _ptr(rows), _stream()))
    return rows


def model_step(params6, exp_avg6, exp_avg_sq6, g_xyz, g_scales, g_rot, g_opac, g_shs, lrs6, beta1, beta2, eps, step,
               want_activated=True):
    """Chain rule of the activations + Adam on the six groups + activations of the updated parameters, one launch
    (include/gsraster.h, gsr_model_step).  params6 in the order xyz, features_dc, features_rest, scaling, rotation,
    opacity.  Returns (scales, rotations, opacities [P,1], shs) of the UPDATED model, or None."""
    xyz, fdc, frest = params6[0], params6[1], params6[2]
    P = int(xyz.size(0))
    M = 1 + (int(frest.size(1)) if frest.numel() else 0)
    for t in list(params6) + list(exp_avg6) + list(exp_avg_sq6) + [g_xyz, g_scales, g_rot, g_opac, g_shs]:
        assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
    outs = None
    if want_activated:
        f32 = dict(dtype=torch.float32, device=xyz.device)
        outs = (torch.empty((P, 3), **f32), torch.empty((P, 4), **f32), torch.empty((P, 1), **f32),
                torch.empty((P, M, 3), **f32))
    VP = C.c_void_p * 6
    arr = lambda ts: VP(*[t.data_ptr() if t.numel() else None for t in ts])  # noqa: E731
    lr = (C.c_float * 6)(*[float(x) for x in lrs6])
    o = outs if outs is not None else (None, None, None, None)
    _check(lib().gsr_model_step(P, M, arr(params6), arr(exp_avg6), arr(exp_avg_sq6), _ptr(g_xyz), _ptr(g_scales),
                                _ptr(g_rot), _ptr(g_opac), _ptr(g_shs), _ptr(o[0]), _ptr(o[1]), _ptr(o[2]), _ptr(o[3]), lr,
                                float(beta1), float(beta2), float(eps), int(step), _stream()))
    return outs
