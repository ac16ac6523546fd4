"""Container-only recipe behind `make -C oracle ref_link`: compiles gs-livm_amd/csrc/torch_binding.cpp against the
reference's OWN headers (-DGSR_REFERENCE_HEADER='"gs/rasterizer.cuh"', include path /root/reference/include/gs,
-DGSR_NO_PYBIND), links it with oracle/ref_link/caller.cpp (which sees only those headers) into
oracle/_ref/link_check, runs it, and checks with nm that the five members rasterizer.cuh:22-80 declares and the three
functions of rasterize_points.cuh are defined as strong (T) symbols.  Outputs stay under oracle/_ref/ (git-ignored).
Needs /root/reference; nothing here is used by the product or on the GPU box."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("REF", "/root/reference")
OUT = os.path.join(ROOT, "oracle", "_ref")

WANT = ["_RasterizeGaussians::forward(", "_RasterizeGaussians::backward(", "GaussianRasterizer::mark_visible(",
        "GaussianRasterizer::rasterize_gaussians(", "GaussianRasterizer::forward(", "RasterizeGaussiansCUDA(",
        "RasterizeGaussiansBackwardCUDA(", "markVisible("]
# gsr_torch_next.hpp (torch_next.cpp): the hosts of the rows either side of the rasterizer
WANT_NEXT = ["gsr_torch::reference_window_1d(", "gsr_torch::photometric_loss(", "gsr_torch::photometric_loss_parts(",
             "gsr_torch::activate(", "gsr_torch::FusedAdam::FusedAdam(", "gsr_torch::FusedAdam::step(",
             "gsr_torch::FusedAdam::step_model(", "gsr_torch::FusedAdam::replace_param(", "gsr_torch::init_gaussians(",
             "gsr_torch::pack_ply_rows(", "gsr_torch::write_ply(", "gsr_torch::ply_attribute_names"]  # ([abi:cxx11] follows the name)


def main():
    hdr = os.path.join(REF, "include", "gs", "gs", "rasterizer.cuh")
    if not os.path.exists(hdr):
        print("no %s: skipping ref_link" % hdr)
        return 0
    pkg = os.path.join(ROOT, "gs-livm_amd")
    report = os.path.join(OUT, "link_check.txt")
    deps = [os.path.join(pkg, "csrc", "torch_binding.cpp"), os.path.join(pkg, "csrc", "torch_next.cpp"),
            os.path.join(pkg, "csrc", "gsr_torch_next.hpp"), os.path.join(HERE, "caller.cpp"), os.path.abspath(__file__),
            os.path.join(ROOT, "include", "gsraster.h"), hdr]  # (not the .so: the link depends on its ABI header only)
    if os.path.exists(report) and all(os.path.getmtime(d) <= os.path.getmtime(report) for d in deps):
        print(open(report).read().strip().splitlines()[-1] + " (cached: sources unchanged)")
        return 0
    import torch
    from torch.utils import cpp_extension as ce
    os.makedirs(OUT, exist_ok=True)
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = ["-I" + p for p in ce.include_paths() + [sysconfig.get_paths()["include"], os.path.join(ROOT, "include"),
                                                    "/opt/rocm/include", os.path.join(REF, "include", "gs"),
                                                    os.path.join(pkg, "csrc")]]
    common = ["g++", "-O1", "-fPIC", "-std=c++17", "-w", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
              "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)] + inc
    obj = os.path.join(OUT, "torch_binding_ref.o")
    subprocess.check_call(common + ["-DGSR_NO_PYBIND", "-DGSR_REFERENCE_HEADER=\"gs/rasterizer.cuh\"", "-c",
                                    os.path.join(pkg, "csrc", "torch_binding.cpp"), "-o", obj])
    obj_next = os.path.join(OUT, "torch_next_ref.o")
    subprocess.check_call(common + ["-c", os.path.join(pkg, "csrc", "torch_next.cpp"), "-o", obj_next])
    exe = os.path.join(OUT, "link_check")
    pylib = sysconfig.get_config_var("LDLIBRARY").replace("lib", "", 1).rsplit(".so", 1)[0]
    subprocess.check_call(common + [os.path.join(HERE, "caller.cpp"), obj, obj_next, "-o", exe, "-L" + tlib, "-L" + pkg,
                                    "-lgsraster_hip", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch",
                                    "-ltorch_python", "-l" + pylib, "-Wl,-rpath," + tlib, "-Wl,-rpath," + pkg,
                                    "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined"])
    syms = subprocess.check_output(["nm", "-C", "--defined-only", obj], text=True).splitlines()
    strong = [ln.split(" ", 2)[2] for ln in syms if len(ln.split(" ", 2)) == 3 and ln.split(" ", 2)[1] == "T"]
    syms_next = subprocess.check_output(["nm", "-C", "--defined-only", obj_next], text=True).splitlines()
    strong_next = [ln.split(" ", 2)[2] for ln in syms_next if len(ln.split(" ", 2)) == 3 and ln.split(" ", 2)[1] == "T"]
    missing = [w for w in WANT if not any(s.startswith(w) for s in strong)] + \
              [w for w in WANT_NEXT if not any(s.startswith(w) for s in strong_next)]
    if missing:
        print("not defined as strong symbols:", missing)
        return 1
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    print(r.stdout.strip())
    if r.returncode != 0 or "ref_link ok" not in r.stdout:
        return 1
    with open(report, "w") as f:
        f.write("strong symbols in torch_binding_ref.o (nm -C --defined-only, type T):\n")
        for w in WANT:
            f.write("  T %s\n" % next(s for s in strong if s.startswith(w))[:160])
        f.write("strong symbols in torch_next_ref.o:\n")
        for w in WANT_NEXT:
            f.write("  T %s\n" % next(s for s in strong_next if s.startswith(w))[:160])
        f.write(r.stdout)
    return 0


if __name__ == "__main__":
    sys.exit(main())
