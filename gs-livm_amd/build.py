"""Build recipe for libgsraster_hip.so (gfx950 only) and the LibTorch operator binding.

    python gs-livm_amd/build.py            # HIP library + Torch binding
    python gs-livm_amd/build.py --lib-only

hipcc cross-compiles without a GPU; outputs stay in-tree (git-ignored, shipped to the GPU box).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
# experiment hooks (A/B builds): extra flags for render.hip and a suffix for the library / object names
SUFFIX = os.environ.get("GSR_LIB_SUFFIX", "")
LIB = os.path.join(HERE, "libgsraster_hip%s.so" % SUFFIX)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

COMMON = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
          "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + os.environ.get("GSR_EXTRA_FLAGS", "").split()
# translation unit -> extra flags.  preprocess.hip feeds the exact-match integer outputs and must
# not be contracted into FMAs (see its header); the blend kernels are tolerance-checked and may.
UNITS = {
    "preprocess.hip": ["-ffp-contract=off"],
    "radix_sort.hip": [],
    # SLP vectorisation packs the scalar f32 math into v_pk_* and splits the fused v_add_f32_dpp reductions
    # into mov_dpp + pk_add: measured 9 % slower on k_blend_backward, so it is off for the blend kernels.
    "render.hip": ["-ffp-contract=fast", "-fno-slp-vectorize"] + os.environ.get("GSR_EXTRA_RENDER_FLAGS", "").split(),
    "optimizer.hip": [],
    "loss.hip": ["-fno-slp-vectorize"],  # (packing the 11-tap sums costs more moves than it saves: 944 -> 732 VALU in the backward)
    "growth.hip": [],
    "api.hip": [],
}
HEADERS = [os.path.join(CSRC, "gsr_internal.hpp"), os.path.join(CSRC, "sort_core.hpp"),
           os.path.join(ROOT, "include", "gsraster.h")]


def _stale(out, deps):
    return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def build_lib(force=False, verbose=False):
    objs, jobs = [], []
    for src, extra in UNITS.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", "%s.o" % SUFFIX))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS + [os.path.abspath(__file__)]):
            jobs.append([HIPCC] + COMMON + extra + ["-c", s, "-o", o])
    with ThreadPoolExecutor(max_workers=4) as ex:
        for out in ex.map(_run, jobs):
            if verbose and out.strip():
                print(out)
    if force or jobs or _stale(LIB, objs):
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs +
             ["-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined"])
    return LIB


def build_torch_binding(force=False, verbose=False):
    """C++/LibTorch operator surface (csrc/torch_binding.cpp) -> gs-livm_amd/_gsraster_torch.so.

    Plain C++ (no device code): compiled with g++ against the LibTorch that ships with PyTorch-ROCm,
    linked to libgsraster_hip.so through its C ABI."""
    import sysconfig

    import torch
    from torch.utils import cpp_extension as ce

    out = os.path.join(HERE, "_gsraster_torch.so")
    # torch_binding.cpp: the rasterizer's operator surface (+ the pybind11 module); torch_next.cpp: the hosts of the rows
    # either side of it (fused loss, activations + Adam, growth, PLY export), declared in gsr_torch_next.hpp
    srcs = [os.path.join(CSRC, "torch_binding.cpp"), os.path.join(CSRC, "torch_next.cpp")]
    hdrs = [os.path.join(CSRC, "gsr_torch_next.hpp"), os.path.join(CSRC, "gsr_torch_surface.hpp")]
    if not (force or _stale(out, srcs + hdrs + HEADERS + [LIB, os.path.abspath(__file__)])):
        return out
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = []
    for p in ce.include_paths() + [sysconfig.get_paths()["include"], os.path.join(ROOT, "include"),
                                   "/opt/rocm/include"]:
        inc += ["-I" + p]
    cmd = (["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-Wall", "-Wno-unused-variable", "-Wno-sign-compare", "-Wno-attributes",
            "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_gsraster_torch",
            "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)] + inc +
           srcs + ["-o", out, "-L" + tlib, "-L" + HERE, "-lgsraster_hip", "-lc10", "-lc10_hip", "-ltorch_cpu",
            "-ltorch_hip", "-ltorch", "-ltorch_python", "-Wl,-rpath," + tlib, "-Wl,-rpath,$ORIGIN",
            "-Wl,-rpath,/opt/rocm/lib"])
    o = _run(cmd)
    if verbose and o.strip():
        print(o)
    return out


if __name__ == "__main__":
    force = "--force" in sys.argv
    print(build_lib(force=force, verbose=True))
    if "--lib-only" not in sys.argv:
        print(build_torch_binding(force=force, verbose=True))
