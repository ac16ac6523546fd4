"""CPU tests of the C-ABI boundary: the library loads without a GPU, exports every symbol that
include/gsraster.h declares, and rejects bad arguments before touching the device."""
import ctypes as C
import os
import re

import pytest

import gs_livm_amd as G
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    h = open(os.path.join(ROOT, "include", "gsraster.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(gsr_[a-z_0-9]+)\s*\(", h)) - {"gsr_alloc_fn"})


def test_library_exports_every_declared_symbol():
    L = G.lib()
    names = _declared()
    assert set(names) == set(G._capi.EXPORTS), (names, G._capi.EXPORTS)
    for n in names:
        assert hasattr(L, n), n
    assert L.gsr_abi_version() == 2


def test_no_cpu_fallback_when_library_missing(monkeypatch, tmp_path):
    monkeypatch.setattr(G._capi, "_lib", None)
    monkeypatch.setattr(G._capi, "LIB_PATH", str(tmp_path / "absent.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        G._capi.lib()


def test_size_helpers():
    L = G.lib()
    assert L.gsr_geometry_bytes(0) > 0 and L.gsr_binning_bytes(0) > 0
    prev = 0
    for P in (1, 256, 257, 10_000, 2_000_000):
        b = L.gsr_geometry_bytes(P)
        assert b >= prev and b % 256 == 0
        prev = b
    # per-instance footprint: point_list 4 B + max(sort buffers, 48-B gradient records) + 1-B flag
    per = (L.gsr_binning_bytes(40_000_000) - L.gsr_binning_bytes(0)) / 40e6
    assert 52.9 < per < 54.0
    assert L.gsr_image_bytes(1920, 1080) >= 1920 * 1080 * 8 + 8160 * 12


def test_higher_msb_equals_oracle():
    L = G.lib()
    for n in list(range(1, 300)) + [1200, 3600, 8160, 65535, 65536, 1 << 20]:
        assert L.gsr_higher_msb(n) == O.higher_msb(n), n


def test_argument_errors_are_reported_not_crashed():
    L = G.lib()
    noop = G._capi.ALLOC_FN(lambda ctx, n: None)
    null = C.c_void_p(None)
    rc = L.gsr_forward(noop, None, noop, None, noop, None, -1, 0, 1, null, 64, 64, null, null, null, null, null, 1.0,
                       null, null, null, null, null, 1.0, 1.0, 0, null, null, null, null, 0, null)
    assert rc == -1 and b"bad P" in L.gsr_last_error()
    one = C.c_void_p(1)  # never dereferenced: validation fails first
    rc = L.gsr_forward(noop, None, noop, None, noop, None, 5, 0, 1, one, 64, 64, null, null, null, one, one, 1.0,
                       one, null, one, one, one, 1.0, 1.0, 0, one, one, one, null, 0, null)
    assert rc == -1 and b"null required input" in L.gsr_last_error()
    rc = L.gsr_forward(noop, None, noop, None, noop, None, 5, 4, 25, one, 64, 64, one, one, null, one, one, 1.0,
                       one, null, one, one, one, 1.0, 1.0, 0, one, one, one, null, 0, null)
    assert rc == -4 and b"SH degree" in L.gsr_last_error()
    rc = L.gsr_backward(-1, 0, 1, 0, null, 64, 64, *([null] * 4), 1.0, *([null] * 5), 1.0, 1.0, *([null] * 15), 0,
                        null)
    assert rc == -1
    assert L.gsr_mark_visible(-3, null, null, null, null, null) == -1
    assert L.gsr_mark_visible(0, null, null, null, null, null) == 0


def test_adaptive_near_budget_rule():
    """include/gsraster.h, "Adaptive near budget": host-side rule, no device involved.  fb(unfinished quads, near
    instances, far instances) feeds one frame's outcome to the rule of the calling thread's current view."""
    L = G.lib()
    fb = L.gsr_near_budget_feedback
    assert L.gsr_near_budget_scale() == 256
    assert fb(10, 1000, 4000) == 256             # a miss over a HUGE remainder (a sparse scene): no budget helps
    assert fb(100, 1000, 3999) == 320            # any other miss: + a quarter of the configured budget, on probation
    assert fb(40, 1000, 10) == 384               # it finished more than a quarter of those quads: kept, and raised again
    assert fb(0, 1000, 0) == 384                 # a hit: the raise on probation is kept
    # a raise that does not help is taken back, and none is tried for 256 frames of the view
    assert fb(100, 1000, 10) == 448
    assert fb(90, 1000, 10) == 384               # 90 of 100 quads still unfinished: sky, the border of the map ...
    for k in range(255):
        assert fb(90, 1000, 10) == 384
    assert fb(90, 1000, 10) == 448               # ... then the rule tries again
    assert fb(90, 1000, 10) == 384
    assert L.gsr_near_far_pause(-1) == 0         # (misses of frames that splitting shortens never pause it)
    # raises stop at three times the configured budget
    for k in range(256):
        fb(1 << 20, 1000, 10)
    live = 1 << 20
    for want in (512, 576, 640, 704, 768, 768, 768, 768):
        live //= 2
        assert fb(live, 1000, 10) == want
    # sixty-four hits in a row take a sixteenth back, never below the configured budget
    for k in range(63):
        assert fb(0, 1000, 0) == 768
    assert fb(0, 1000, 0) == 752
    for k in range(64 * 40):
        s = fb(0, 1000, 0)
    assert s == 256
    for k in range(200):
        assert fb(0, 1000, 0) == 256
    # eight misses in a row over a far chain of four times the near chain (a sparse scene) pause the splitting
    for k in range(7):
        fb(10, 1000, 9000)
    assert L.gsr_near_far_pause(-1) == 0
    fb(10, 1000, 9000)
    assert L.gsr_near_far_pause(-1) == 256 and L.gsr_near_budget_scale() == 256
    assert L.gsr_near_far_pause(0) == 256 and L.gsr_near_far_pause(-1) == 0
    # ... in a row: a frame that splitting does shorten restarts the count
    for k in range(7):
        fb(10, 1000, 9000)
    fb(10, 1000, 500)
    for k in range(7):
        fb(10, 1000, 9000)
    assert L.gsr_near_far_pause(-1) == 0
    fb(10, 1000, 9000)
    assert L.gsr_near_far_pause(-1) == 256
    L.gsr_near_far_pause(0)
    for k in range(64 * 40):                     # (leave the thread's rule as it was found)
        fb(0, 1000, 0)
    assert L.gsr_near_budget_scale() == 256
