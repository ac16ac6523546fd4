"""ctypes front-end of the CPU oracle (oracle/gsr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.  Parity status: unpinned (see the
header of gsr_oracle.c).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgsr_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "gsr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgsr_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def build_ref():
    """oracle/_ref/check_glm: compiled from the reference's vendored GLM where /root/reference exists (the build
    container); returns its path or None."""
    subprocess.call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # + the link check of the C++ binding against the reference's own headers (oracle/ref_link/; cached)
    subprocess.call(["make", "-C", _HERE, "ref_link"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    p = os.path.join(_HERE, "_ref", "check_glm")
    return p if os.path.exists(p) else None


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.gsro_forward.restype = C.c_void_p
        _lib.gsro_num_rendered.restype = C.c_int
        _lib.gsro_num_rendered.argtypes = [C.c_void_p]
        _lib.gsro_free.argtypes = [C.c_void_p]
        _lib.gsro_higher_msb.restype = C.c_uint32
        _lib.gsro_higher_msb.argtypes = [C.c_uint32]
        for name in _FIELDS:
            fn = getattr(_lib, "gsro_" + name)
            fn.restype = C.c_void_p
            fn.argtypes = [C.c_void_p]
        # never more OpenMP workers than a GPU box's CPU share (16 cores per GPU), whatever the host exposes
        _lib.gsro_max_threads.restype = C.c_int
        _lib.gsro_set_threads(C.c_int(min(int(_lib.gsro_max_threads()), len(os.sched_getaffinity(0)), 16)))
    return _lib


# name -> (dtype, shape-lambda(P, R, W, H, T))
_FIELDS = {
    "radii": (np.int32, lambda P, R, W, H, T: (P,)),
    "means2D": (np.float32, lambda P, R, W, H, T: (P, 2)),
    "depths": (np.float32, lambda P, R, W, H, T: (P,)),
    "cov3D": (np.float32, lambda P, R, W, H, T: (P, 6)),
    "rgb": (np.float32, lambda P, R, W, H, T: (P, 3)),
    "conic_opacity": (np.float32, lambda P, R, W, H, T: (P, 4)),
    "tiles_touched": (np.uint32, lambda P, R, W, H, T: (P,)),
    "point_offsets": (np.uint32, lambda P, R, W, H, T: (P,)),
    "clamped": (np.uint8, lambda P, R, W, H, T: (P, 3)),
    "keys_unsorted": (np.uint64, lambda P, R, W, H, T: (R,)),
    "values_unsorted": (np.uint32, lambda P, R, W, H, T: (R,)),
    "keys": (np.uint64, lambda P, R, W, H, T: (R,)),
    "point_list": (np.uint32, lambda P, R, W, H, T: (R,)),
    "ranges": (np.uint32, lambda P, R, W, H, T: (T, 2)),
    "final_T": (np.float32, lambda P, R, W, H, T: (H, W)),
    "n_contrib": (np.uint32, lambda P, R, W, H, T: (H, W)),
    "out_color": (np.float32, lambda P, R, W, H, T: (3, H, W)),
    "out_depth": (np.float32, lambda P, R, W, H, T: (1, H, W)),
    "out_acc": (np.float32, lambda P, R, W, H, T: (1, H, W)),
    "fragile": (np.uint8, lambda P, R, W, H, T: (H, W)),
}


def _f32(a):
    if a is None:
        return None
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None or a.size == 0 else a.ctypes.data_as(C.c_void_p)


class Frame:
    """Owns one oracle forward pass; attributes are numpy copies of every intermediate."""

    def __init__(self, handle, P, W, H, keep_handle):
        L = lib()
        self.P, self.W, self.H = P, W, H
        self.R = L.gsro_num_rendered(handle)
        T = ((W + 15) // 16) * ((H + 15) // 16)
        for name, (dt, shp) in _FIELDS.items():
            shape = shp(P, self.R, W, H, T)
            n = int(np.prod(shape))
            if n == 0:
                arr = np.zeros(shape, dtype=dt)
            else:
                ptr = getattr(L, "gsro_" + name)(handle)
                arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dt).itemsize,))
                arr = arr.view(dt).reshape(shape).copy()
            setattr(self, name, arr)
        self._h = handle if keep_handle else None
        if not keep_handle:
            L.gsro_free(handle)

    def close(self):
        if getattr(self, "_h", None):
            lib().gsro_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass


def forward(scene, keep_handle=True, tight=False):
    """scene: dict with the arrays of gs_livm_amd.synthetic.make_scene (numpy).
    tight=False: the reference's tile rectangles (every integer stage as the reference computes it).
    tight=True: the product's culled rectangles (gsr_oracle.c, tighten_rect) -- same images and gradients,
    different tiles_touched / sorted lists / ranges / n_contrib; used to check the HIP path's integer stages."""
    s = scene
    L = lib()
    L.gsro_set_tight(C.c_int(1 if tight else 0))
    a = {k: _f32(s.get(k)) for k in ("bg", "means3D", "shs", "colors_precomp", "opacities", "scales", "rotations",
                                     "cov3D_precomp", "viewmatrix", "projmatrix", "campos")}
    P = int(a["means3D"].shape[0])
    M = 0 if a["shs"] is None or a["shs"].size == 0 else int(a["shs"].shape[1])
    h = L.gsro_forward(C.c_int(P), C.c_int(int(s["sh_degree"])), C.c_int(M), _p(a["bg"]), C.c_int(int(s["W"])),
                       C.c_int(int(s["H"])), _p(a["means3D"]), _p(a["shs"]), _p(a["colors_precomp"]),
                       _p(a["opacities"]), _p(a["scales"]), C.c_float(float(s.get("scale_modifier", 1.0))),
                       _p(a["rotations"]), _p(a["cov3D_precomp"]), _p(a["viewmatrix"]), _p(a["projmatrix"]),
                       _p(a["campos"]), C.c_float(float(s["tanfovx"])), C.c_float(float(s["tanfovy"])))
    L.gsro_set_tight(C.c_int(0))
    fr = Frame(C.c_void_p(h), P, int(s["W"]), int(s["H"]), keep_handle)
    fr._keep = a
    fr.M = M
    return fr


def backward(frame, scene, dL_dcolor, dL_dacc):
    """Returns dict of the nine gradient arrays (reference shapes, rasterize_points.cu:173-181)."""
    assert frame._h, "forward(..., keep_handle=True) required"
    s, a, L = scene, frame._keep, lib()
    P, M = frame.P, frame.M
    g = {
        "dL_dmeans2D": np.zeros((P, 3), np.float32), "dL_dconic": np.zeros((P, 2, 2), np.float32),
        "dL_dopacity": np.zeros((P, 1), np.float32), "dL_dcolors": np.zeros((P, 3), np.float32),
        "dL_dmeans3D": np.zeros((P, 3), np.float32), "dL_dcov3D": np.zeros((P, 6), np.float32),
        "dL_dsh": np.zeros((P, M, 3), np.float32), "dL_dscales": np.zeros((P, 3), np.float32),
        "dL_drotations": np.zeros((P, 4), np.float32),
    }
    dpix, dacc = _f32(dL_dcolor), _f32(dL_dacc)
    if P:
        L.gsro_backward(frame._h, _p(a["bg"]), _p(a["means3D"]), _p(a["shs"]), _p(a["colors_precomp"]),
                        _p(a["scales"]), C.c_float(float(s.get("scale_modifier", 1.0))), _p(a["rotations"]),
                        _p(a["cov3D_precomp"]), _p(a["viewmatrix"]), _p(a["projmatrix"]), _p(a["campos"]),
                        C.c_float(float(s["tanfovx"])), C.c_float(float(s["tanfovy"])), _p(dpix), _p(dacc),
                        _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dcolors"]),
                        _p(g["dL_dmeans3D"]), _p(g["dL_dcov3D"]), _p(g["dL_dsh"]), _p(g["dL_dscales"]),
                        _p(g["dL_drotations"]))
    return g


def mark_visible(means3D, viewmatrix):
    m, v = _f32(means3D), _f32(viewmatrix)
    out = np.zeros((m.shape[0],), np.uint8)
    if m.shape[0]:
        lib().gsro_mark_visible(C.c_int(m.shape[0]), _p(m), _p(v), _p(out))
    return out.astype(bool)


def higher_msb(n):
    return int(lib().gsro_higher_msb(C.c_uint32(n)))


def set_threads(n):
    lib().gsro_set_threads(C.c_int(int(n)))


def max_threads():
    return int(lib().gsro_max_threads())
