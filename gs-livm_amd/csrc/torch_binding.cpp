// torch_binding.cpp -- the LibTorch operator surface GS-LIVM links against, re-pointed at the C ABI of
// libgsraster_hip.so (include/gsraster.h).  Plain C++ (no device code): PyTorch-ROCm is used only for
// tensors, the caching allocator, autograd and the current HIP stream.
//
// Same names, argument order/meaning and return shapes as the reference:
//   GaussianRasterizationSettings, GaussianRasterizer, _RasterizeGaussians   include/gs/gs/rasterizer.cuh:8-80,
//                                                                          src/gs/rasterizer.cu
//   RasterizeGaussiansCUDA / RasterizeGaussiansBackwardCUDA / markVisible  include/gs/gs/rasterize_points.cuh:18-73,
//                                                                          src/gs/rasterize_points.cu
// A maintainer drops this file in place of src/gs/rasterize_points.cu + src/gs/rasterizer.cu (see
// INTEGRATION.md); the pybind11 module at the bottom exists so the Python test-suite can drive the very
// same C++ code path.
//
// Deliberate differences from the reference, all behind the same signatures:
//   * errors from the device library are thrown as std::runtime_error (a C ABI cannot throw);
//   * the reference calls exit(1) when shs / scales / rotations are undefined (rasterizer.cu:173-190);
//     here that is a std::invalid_argument -- the precomputed-colour / covariance inputs are supported;
//   * scalars are boxed into CPU tensors, so _RasterizeGaussians::forward's seven .item() calls
//     (rasterizer.cu:27-33) no longer synchronise the device;
//   * the nine gradient tensors are torch::empty (the library overwrites every element), saving the
//     ~120 B/Gaussian of memset the reference's torch::zeros cost per backward (rasterize_points.cu:173-181).
#include <torch/extension.h>

#include <c10/hip/HIPStream.h>

#include <stdexcept>
#include <string>
#include <tuple>

#include "gsraster.h"
#include "gsr_torch_next.hpp"  // the hosts of the rows either side of the rasterizer (torch_next.cpp): bound below

// The declarations this file implements.  In the GS-LIVM tree they are the reference's OWN headers, untouched
// (-DGSR_REFERENCE_HEADER='"gs/rasterizer.cuh"', which pulls in gs/rasterize_points.cuh); standalone (this repo, the
// GPU box) the same declarations come from gsr_torch_surface.hpp.  Either way the members below are defined out of
// line, as strong symbols, exactly where src/gs/rasterizer.cu and src/gs/rasterize_points.cu defined them.
#ifdef GSR_REFERENCE_HEADER
#include GSR_REFERENCE_HEADER
#else
#include "gsr_torch_surface.hpp"
#endif

namespace {

void* current_stream() { return static_cast<void*>(c10::hip::getCurrentHIPStream().stream()); }

// size-0 tensor == "not provided" (rasterizer.cu:178-193)
const float* fptr(const torch::Tensor& t) { return t.defined() && t.numel() ? t.data_ptr<float>() : nullptr; }

void check(int code, const char* what) {
  if (code < 0) throw std::runtime_error(std::string(what) + ": " + gsr_last_error());
}

// Allocator callback: replaces resizeFunctional (rasterize_points.cu:36-44).
// The binning blob's callback can come a second time within one forward (gsr_forward, speculative path, when the
// predicted capacity was too small): the tensor is then replaced, not resized -- resize_ would copy the old contents.
char* resize_blob(void* ctx, size_t bytes) {
  auto* t = static_cast<torch::Tensor*>(ctx);
  if (t->numel() != 0)
    *t = torch::empty({static_cast<long long>(bytes)}, t->options());
  else
    t->resize_({static_cast<long long>(bytes)});
  return reinterpret_cast<char*>(t->data_ptr());
}

}  // namespace

std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                       const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
                       const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                       const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                       const int image_height, const int image_width, const torch::Tensor& sh, const int degree,
                       const torch::Tensor& campos, const bool prefiltered, const bool debug) {
  if (means3D.ndimension() != 2 || means3D.size(1) != 3) {
    AT_ERROR("means3D must have dimensions (num_points, 3)");
  }
  const int P = means3D.size(0);
  const int H = image_height;
  const int W = image_width;
  auto float_opts = means3D.options().dtype(torch::kFloat32);
  // fully written by the library (zero-filled by it when P == 0)
  torch::Tensor out_color = torch::empty({3, H, W}, float_opts);
  torch::Tensor out_depth = torch::empty({1, H, W}, float_opts);
  torch::Tensor out_acc = torch::empty({1, H, W}, float_opts);
  torch::Tensor radii = P ? torch::empty({P}, means3D.options().dtype(torch::kInt32))  // every entry is written
                          : torch::zeros({0}, means3D.options().dtype(torch::kInt32));
  auto byte_opts = means3D.options().dtype(torch::kByte);
  torch::Tensor geomBuffer = torch::empty({0}, byte_opts);
  torch::Tensor binningBuffer = torch::empty({0}, byte_opts);
  torch::Tensor imgBuffer = torch::empty({0}, byte_opts);

  int M = 0;
  if (sh.defined() && sh.numel() != 0) M = sh.size(1);
  auto bg = background.contiguous(), m3 = means3D.contiguous(), shc = sh.contiguous(), col = colors.contiguous(),
       op = opacity.contiguous(), sc = scales.contiguous(), rot = rotations.contiguous(),
       cov = cov3D_precomp.contiguous(), view = viewmatrix.contiguous(), proj = projmatrix.contiguous(),
       cam = campos.contiguous();
  const int rendered = gsr_forward(
      resize_blob, &geomBuffer, resize_blob, &binningBuffer, resize_blob, &imgBuffer, P, degree, M, fptr(bg), W, H,
      fptr(m3), fptr(shc), fptr(col), fptr(op), fptr(sc), scale_modifier, fptr(rot), fptr(cov), fptr(view), fptr(proj),
      fptr(cam), tan_fovx, tan_fovy, prefiltered ? 1 : 0, out_color.data_ptr<float>(), out_depth.data_ptr<float>(),
      out_acc.data_ptr<float>(), P ? radii.data_ptr<int>() : nullptr, debug ? 1 : 0, current_stream());
  check(rendered, "gsr_forward");
  return std::make_tuple(rendered, out_color, out_depth, out_acc, radii, geomBuffer, binningBuffer, imgBuffer);
}

// want_cov3D = false (the autograd node, when the covariance comes from scales and rotations): dL_dcov3D is an
// intermediate nobody reads -- it is neither allocated nor written (24 of the 218 bytes per Gaussian the per-Gaussian
// backward moves) and comes back undefined.  The public entry point always returns it, as the reference does.
static std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
                  torch::Tensor>
rasterize_backward_impl(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                        const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                        const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                        const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                        const torch::Tensor& dL_dout_color, const torch::Tensor& dL_dout_acc, const torch::Tensor& sh,
                        const int degree, const torch::Tensor& campos, const torch::Tensor& geomBuffer, const int R,
                        const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer, const bool debug,
                        const bool want_cov3D) {
  const int P = means3D.size(0);
  const int H = dL_dout_color.size(1);
  const int W = dL_dout_color.size(2);
  int M = 0;
  if (sh.defined() && sh.numel() != 0) M = sh.size(1);
  auto o = means3D.options();
  auto mk = [&](std::initializer_list<int64_t> shape) { return P ? torch::empty(shape, o) : torch::zeros(shape, o); };
  torch::Tensor dL_dmeans3D = mk({P, 3});
  torch::Tensor dL_dmeans2D = mk({P, 3});
  torch::Tensor dL_dcolors = mk({P, 3});
  torch::Tensor dL_dconic = mk({P, 2, 2});
  torch::Tensor dL_dopacity = mk({P, 1});
  torch::Tensor dL_dcov3D = want_cov3D ? mk({P, 6}) : torch::Tensor();
  torch::Tensor dL_dsh = mk({P, M, 3});
  torch::Tensor dL_dscales = mk({P, 3});
  torch::Tensor dL_drotations = mk({P, 4});
  if (P != 0) {
    auto bg = background.contiguous(), m3 = means3D.contiguous(), shc = sh.contiguous(), col = colors.contiguous(),
         sc = scales.contiguous(), rot = rotations.contiguous(), cov = cov3D_precomp.contiguous(),
         view = viewmatrix.contiguous(), proj = projmatrix.contiguous(), cam = campos.contiguous(),
         dpix = dL_dout_color.contiguous(), dacc = dL_dout_acc.contiguous(), rad = radii.contiguous();
    check(gsr_backward(P, degree, M, R, fptr(bg), W, H, fptr(m3), fptr(shc), fptr(col), fptr(sc), scale_modifier,
                       fptr(rot), fptr(cov), fptr(view), fptr(proj), fptr(cam), tan_fovx, tan_fovy,
                       rad.data_ptr<int>(), reinterpret_cast<char*>(geomBuffer.data_ptr()),
                       reinterpret_cast<char*>(binningBuffer.data_ptr()),
                       reinterpret_cast<char*>(imageBuffer.data_ptr()), fptr(dpix), fptr(dacc),
                       dL_dmeans2D.data_ptr<float>(), dL_dconic.data_ptr<float>(), dL_dopacity.data_ptr<float>(),
                       dL_dcolors.data_ptr<float>(), dL_dmeans3D.data_ptr<float>(),
                       dL_dcov3D.defined() ? dL_dcov3D.data_ptr<float>() : nullptr,
                       M ? dL_dsh.data_ptr<float>() : nullptr, dL_dscales.data_ptr<float>(),
                       dL_drotations.data_ptr<float>(), debug ? 1 : 0, current_stream()),
          "gsr_backward");
  }
  return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
                         dL_drotations);
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
RasterizeGaussiansBackwardCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                               const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                               const float scale_modifier, const torch::Tensor& cov3D_precomp,
                               const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
                               const float tan_fovy, const torch::Tensor& dL_dout_color,
                               const torch::Tensor& dL_dout_acc, const torch::Tensor& sh, const int degree,
                               const torch::Tensor& campos, const torch::Tensor& geomBuffer, const int R,
                               const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer, const bool debug) {
  return rasterize_backward_impl(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_acc, sh, degree, campos,
                                 geomBuffer, R, binningBuffer, imageBuffer, debug, /*want_cov3D=*/true);
}

torch::Tensor markVisible(torch::Tensor& means3D, torch::Tensor& viewmatrix, torch::Tensor& projmatrix) {
  const int P = means3D.size(0);
  torch::Tensor present = torch::full({P}, false, means3D.options().dtype(at::kBool));
  if (P != 0) {
    auto m3 = means3D.contiguous(), view = viewmatrix.contiguous(), proj = projmatrix.contiguous();
    check(gsr_mark_visible(P, fptr(m3), fptr(view), fptr(proj),
                           reinterpret_cast<unsigned char*>(present.data_ptr<bool>()), current_stream()),
          "gsr_mark_visible");
  }
  return present;
}

// ------------------------------- src/gs/rasterizer.cu ---------------------------------------------
// Out-of-line definitions of the five members include/gs/gs/rasterizer.cuh:22-80 declares without bodies -- the
// strong symbols render_utils.cuh and lioOptimization.cpp link against once src/gs/rasterizer.cu is gone.
torch::autograd::tensor_list _RasterizeGaussians::forward(
    torch::autograd::AutogradContext* ctx, torch::Tensor means3D, torch::Tensor means2D, torch::Tensor sh,
    torch::Tensor colors_precomp, torch::Tensor opacities, torch::Tensor scales, torch::Tensor rotations,
    torch::Tensor cov3Ds_precomp, torch::Tensor image_height, torch::Tensor image_width, torch::Tensor tanfovx,
    torch::Tensor tanfovy, torch::Tensor bg, torch::Tensor scale_modifier, torch::Tensor viewmatrix,
    torch::Tensor projmatrix, torch::Tensor sh_degree, torch::Tensor camera_center, torch::Tensor prefiltered) {
  // host-resident 0-dim tensors (see rasterize_gaussians below): these reads do not touch the device
  const int image_height_val = image_height.item<int>();
  const int image_width_val = image_width.item<int>();
  const float tanfovx_val = tanfovx.item<float>();
  const float tanfovy_val = tanfovy.item<float>();
  const float scale_modifier_val = scale_modifier.item<float>();
  const int sh_degree_val = sh_degree.item<int>();
  const bool prefiltered_val = prefiltered.item<bool>();
  auto [num_rendered, color, out_depth, out_acc, radii, geomBuffer, binningBuffer, imgBuffer] =
      RasterizeGaussiansCUDA(bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier_val,
                             cov3Ds_precomp, viewmatrix, projmatrix, tanfovx_val, tanfovy_val, image_height_val,
                             image_width_val, sh, sh_degree_val, camera_center, prefiltered_val, false);
  ctx->save_for_backward(
      {colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer});
  ctx->saved_data["num_rendered"] = num_rendered;
  ctx->saved_data["background"] = bg;
  ctx->saved_data["scale_modifier"] = scale_modifier_val;
  ctx->saved_data["viewmatrix"] = viewmatrix;
  ctx->saved_data["projmatrix"] = projmatrix;
  ctx->saved_data["tanfovx"] = tanfovx_val;
  ctx->saved_data["tanfovy"] = tanfovy_val;
  ctx->saved_data["image_height"] = image_height_val;
  ctx->saved_data["image_width"] = image_width_val;
  ctx->saved_data["sh_degree"] = sh_degree_val;
  ctx->saved_data["camera_center"] = camera_center;
  ctx->saved_data["prefiltered"] = prefiltered_val;
  ctx->mark_non_differentiable({radii});
  ctx->set_materialize_grads(false);  // no zero images for the outputs nobody differentiates (depth)
  return {color, radii, out_depth, out_acc};
}

torch::autograd::tensor_list _RasterizeGaussians::backward(torch::autograd::AutogradContext* ctx,
                                                           torch::autograd::tensor_list grad_outputs) {
  auto grad_out_color = grad_outputs[0];
  // grad_outputs[1] (radii) and [2] (depth) are ignored, exactly as the reference (rasterizer.cu:78-79)
  auto grad_acc = grad_outputs[3];
  const int num_rendered = ctx->saved_data["num_rendered"].to<int>();
  auto saved = ctx->get_saved_variables();
  auto colors_precomp = saved[0], means3D = saved[1], scales = saved[2], rotations = saved[3],
       cov3Ds_precomp = saved[4], radii = saved[5], sh = saved[6], geomBuffer = saved[7], binningBuffer = saved[8],
       imgBuffer = saved[9];
  const int H = ctx->saved_data["image_height"].to<int>(), W = ctx->saved_data["image_width"].to<int>();
  if (!grad_out_color.defined()) grad_out_color = torch::zeros({3, H, W}, means3D.options());
  if (!grad_acc.defined()) grad_acc = torch::zeros({1, H, W}, means3D.options());
  auto [grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
        grad_rotations] =
      rasterize_backward_impl(
          ctx->saved_data["background"].to<torch::Tensor>(), means3D, radii, colors_precomp, scales, rotations,
          ctx->saved_data["scale_modifier"].to<double>(), cov3Ds_precomp,
          ctx->saved_data["viewmatrix"].to<torch::Tensor>(), ctx->saved_data["projmatrix"].to<torch::Tensor>(),
          ctx->saved_data["tanfovx"].to<double>(), ctx->saved_data["tanfovy"].to<double>(), grad_out_color, grad_acc,
          sh, ctx->saved_data["sh_degree"].to<int>(), ctx->saved_data["camera_center"].to<torch::Tensor>(),
          geomBuffer, num_rendered, binningBuffer, imgBuffer, false, /*want_cov3D=*/cov3Ds_precomp.numel() != 0);
  auto opt = [](const torch::Tensor& g, const torch::Tensor& x) { return x.numel() ? g : torch::Tensor(); };
  return {grad_means3D,
          grad_means2D,
          opt(grad_sh, sh),
          opt(grad_colors_precomp, colors_precomp),
          grad_opacities,
          opt(grad_scales, scales),
          opt(grad_rotations, rotations),
          opt(grad_cov3Ds_precomp, cov3Ds_precomp),
          torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(),
          torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor()};
}

torch::Tensor GaussianRasterizer::mark_visible(torch::Tensor positions) {
  torch::NoGradGuard no_grad;
  return markVisible(positions, raster_settings_.viewmatrix, raster_settings_.projmatrix);
}

torch::autograd::tensor_list GaussianRasterizer::rasterize_gaussians(
    torch::Tensor means3D, torch::Tensor means2D, torch::Tensor sh, torch::Tensor colors_precomp,
    torch::Tensor opacities, torch::Tensor scales, torch::Tensor rotations, torch::Tensor cov3Ds_precomp,
    GaussianRasterizationSettings raster_settings) {
  torch::Device device = means3D.is_cuda() ? means3D.device() : torch::Device(torch::kCUDA);
  // scalars stay on the host: the reference's 0-dim CUDA tensors cost seven D2H syncs per render
  auto image_height = torch::tensor(raster_settings.image_height);
  auto image_width = torch::tensor(raster_settings.image_width);
  auto tanfovx = torch::tensor(raster_settings.tanfovx);
  auto tanfovy = torch::tensor(raster_settings.tanfovy);
  auto scale_modifier = torch::tensor(raster_settings.scale_modifier);
  auto sh_degree = torch::tensor(raster_settings.sh_degree);
  auto prefiltered = torch::tensor(raster_settings.prefiltered);
  auto mv = [&](torch::Tensor t) { return (t.defined() && t.device() != device) ? t.to(device) : t; };
  return _RasterizeGaussians::apply(mv(means3D), mv(means2D), mv(sh), mv(colors_precomp), mv(opacities), mv(scales),
                                    mv(rotations), mv(cov3Ds_precomp), image_height, image_width, tanfovx, tanfovy,
                                    mv(raster_settings.bg), scale_modifier, mv(raster_settings.viewmatrix),
                                    mv(raster_settings.projmatrix), sh_degree, mv(raster_settings.camera_center),
                                    prefiltered);
}

// (the default arguments live on the declaration)
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> GaussianRasterizer::forward(
    torch::Tensor means3D, torch::Tensor means2D, torch::Tensor opacities, torch::Tensor shs,
    torch::Tensor colors_precomp, torch::Tensor scales, torch::Tensor rotations, torch::Tensor cov3D_precomp) {
  if ((shs.defined() && colors_precomp.defined()) || (!shs.defined() && !colors_precomp.defined())) {
    throw std::invalid_argument("Please provide exactly one of either SHs or precomputed colors!");
  }
  if (((scales.defined() || rotations.defined()) && cov3D_precomp.defined()) ||
      (!scales.defined() && !rotations.defined() && !cov3D_precomp.defined())) {
    throw std::invalid_argument(
        "Please provide exactly one of either scale/rotation pair or "
        "precomputed 3D covariance!");
  }
  if ((scales.defined() != rotations.defined()) && !cov3D_precomp.defined()) {
    throw std::invalid_argument("scales and rotations must be provided together");
  }
  auto empty = [&]() { return torch::empty({0}, means3D.options().dtype(torch::kFloat32)); };
  if (!shs.defined()) shs = empty();
  if (!colors_precomp.defined()) colors_precomp = empty();
  if (!scales.defined()) scales = empty();
  if (!rotations.defined()) rotations = empty();
  if (!cov3D_precomp.defined()) cov3D_precomp = empty();
  auto result = rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                    cov3D_precomp, raster_settings_);
  return {result[0], result[1], result[2], result[3]};
}

// ------------------------- pybind11 exposure (tests drive the C++ surface) -------------------------
#ifndef GSR_NO_PYBIND
namespace py = pybind11;

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "C++/LibTorch operator surface of gs_livm_amd (mirror of GS-LIVM's src/gs/rasterizer.cu)";
  m.def("RasterizeGaussiansCUDA", &RasterizeGaussiansCUDA);
  m.def("RasterizeGaussiansBackwardCUDA", &RasterizeGaussiansBackwardCUDA);
  m.def("markVisible", [](torch::Tensor a, torch::Tensor b, torch::Tensor c) { return markVisible(a, b, c); });
  // exact instance count of this thread's last forward (RasterizeGaussiansCUDA's first element is the binning key:
  // the capacity the binning blob was carved for -- equal to the count after a synchronous forward)
  m.def("last_num_rendered", []() { return gsr_last_num_rendered(); });
  py::class_<GaussianRasterizationSettings>(m, "GaussianRasterizationSettings")
      .def(py::init([](int h, int w, float tx, float ty, torch::Tensor bg, float sm, torch::Tensor view,
                       torch::Tensor proj, int deg, torch::Tensor cam, bool pre) {
             return GaussianRasterizationSettings{h, w, tx, ty, bg, sm, view, proj, deg, cam, pre};
           }),
           py::arg("image_height"), py::arg("image_width"), py::arg("tanfovx"), py::arg("tanfovy"), py::arg("bg"),
           py::arg("scale_modifier"), py::arg("viewmatrix"), py::arg("projmatrix"), py::arg("sh_degree"),
           py::arg("camera_center"), py::arg("prefiltered") = false);
  auto opt = [](const py::object& o) { return o.is_none() ? torch::Tensor() : o.cast<torch::Tensor>(); };
  // GaussianRasterizer inherits torch::nn::Module privately (as in the reference), which pybind11 cannot
  // hold directly: the Python handle owns one by composition.
  struct PyRasterizer {
    GaussianRasterizer impl;
    explicit PyRasterizer(GaussianRasterizationSettings s) : impl(std::move(s)) {}
  };
  py::class_<PyRasterizer>(m, "GaussianRasterizer")
      .def(py::init<GaussianRasterizationSettings>())
      .def("mark_visible", [](PyRasterizer& self, torch::Tensor p) { return self.impl.mark_visible(p); })
      .def("forward",
           [opt](PyRasterizer& self, torch::Tensor means3D, torch::Tensor means2D, torch::Tensor opacities,
                 py::object shs, py::object colors_precomp, py::object scales, py::object rotations,
                 py::object cov3D_precomp) {
             const torch::Tensor a = opt(shs), b = opt(colors_precomp), c = opt(scales), d = opt(rotations),
                                 e = opt(cov3D_precomp);
             // (no Python object is touched from here on: other rendering threads of the interpreter may run -- the
             // reference's rendering threads are C++ threads and never meet a GIL)
             py::gil_scoped_release unlocked;
             return self.impl.forward(means3D, means2D, opacities, a, b, c, d, e);
           },
           py::arg("means3D"), py::arg("means2D"), py::arg("opacities"), py::arg("shs") = py::none(),
           py::arg("colors_precomp") = py::none(), py::arg("scales") = py::none(), py::arg("rotations") = py::none(),
           py::arg("cov3D_precomp") = py::none());

  // ---- gsr_torch_next.hpp: the C++ hosts of the "next" rows, so that the tests can hold them against the Python route
  namespace nx = gsr_torch;
  auto n = m.def_submodule("next", "C++/LibTorch hosts of the fused loss, activations + Adam, growth and PLY export");
  n.def("reference_window_1d", &nx::reference_window_1d, py::arg("window_size") = 11, py::arg("sigma") = 1.5f);
  n.def("photometric_loss",
        [opt](torch::Tensor image, torch::Tensor gt, float lambda_dssim, py::object window) {
          return nx::photometric_loss(image, gt, lambda_dssim, opt(window));
        },
        py::arg("image"), py::arg("gt"), py::arg("lambda_dssim") = 0.2f, py::arg("window1d") = py::none());
  n.def("photometric_loss_parts",
        [opt](torch::Tensor image, torch::Tensor gt, float lambda_dssim, py::object window) {
          return nx::photometric_loss_parts(image, gt, lambda_dssim, opt(window));
        },
        py::arg("image"), py::arg("gt"), py::arg("lambda_dssim") = 0.2f, py::arg("window1d") = py::none());
  n.def("activate", [](torch::Tensor s, torch::Tensor r, torch::Tensor o, torch::Tensor dc, torch::Tensor rest) {
    const nx::Activated a = nx::activate(s, r, o, dc, rest);
    return std::make_tuple(a.scaling, a.rotation, a.opacity, a.features);
  });
  py::class_<nx::FusedAdam>(n, "FusedAdam")
      .def(py::init<std::vector<torch::Tensor>, std::vector<double>, double, double, double>(), py::arg("params"),
           py::arg("lrs"), py::arg("beta1") = 0.9, py::arg("beta2") = 0.999, py::arg("eps") = 1e-15)
      .def("step", &nx::FusedAdam::step, py::arg("zero_grad") = true)
      .def("step_model",
           [](nx::FusedAdam& self, torch::Tensor gx, torch::Tensor gs, torch::Tensor gr, torch::Tensor go,
              torch::Tensor gf) {
             const nx::Activated a = self.step_model(gx, gs, gr, go, gf);
             return std::make_tuple(a.scaling, a.rotation, a.opacity, a.features);
           })
      .def("replace_param", &nx::FusedAdam::replace_param)
      .def("step_count", &nx::FusedAdam::step_count)
      .def("params", &nx::FusedAdam::params)
      .def("exp_avg", &nx::FusedAdam::exp_avg)
      .def("exp_avg_sq", &nx::FusedAdam::exp_avg_sq);
  n.def("init_gaussians", &nx::init_gaussians);
  n.def("pack_ply_rows", &nx::pack_ply_rows);
  n.def("write_ply", &nx::write_ply);
  n.def("ply_attribute_names", &nx::ply_attribute_names);
}
#endif  // GSR_NO_PYBIND
