// gsr_torch_surface.hpp -- declarations of the LibTorch operator surface torch_binding.cpp implements, for
// builds OUTSIDE the GS-LIVM tree (this repository's own build and tests).  They restate, member for member and
// default argument for default argument, what the reference declares in
//   include/gs/gs/rasterize_points.cuh:18-73   RasterizeGaussiansCUDA / RasterizeGaussiansBackwardCUDA / markVisible
//   include/gs/gs/rasterizer.cuh:8-80          GaussianRasterizationSettings, _RasterizeGaussians, GaussianRasterizer
// Inside GS-LIVM this header is NOT used: torch_binding.cpp is compiled against the reference's own headers
// (-DGSR_REFERENCE_HEADER); oracle/Makefile's `ref_link` target proves that build links.
#pragma once
#include <torch/extension.h>

#include <tuple>

using GsrFwdResult = std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
                                torch::Tensor, torch::Tensor>;
using GsrBwdResult = std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
                                torch::Tensor, torch::Tensor, torch::Tensor>;

// -> (num_rendered, out_color [3,H,W], out_depth [1,H,W], out_acc [1,H,W], radii [P] i32, geomBuffer, binningBuffer,
//     imgBuffer)
GsrFwdResult RasterizeGaussiansCUDA(const torch::Tensor& background, const torch::Tensor& means3D,
                                    const torch::Tensor& colors, const torch::Tensor& opacity,
                                    const torch::Tensor& scales, const torch::Tensor& rotations,
                                    const float scale_modifier, const torch::Tensor& cov3D_precomp,
                                    const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix,
                                    const float tan_fovx, const float tan_fovy, const int image_height,
                                    const int image_width, const torch::Tensor& sh, const int degree,
                                    const torch::Tensor& campos, const bool prefiltered, const bool debug);

// -> (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)
GsrBwdResult RasterizeGaussiansBackwardCUDA(const torch::Tensor& background, const torch::Tensor& means3D,
                                            const torch::Tensor& radii, const torch::Tensor& colors,
                                            const torch::Tensor& scales, const torch::Tensor& rotations,
                                            const float scale_modifier, const torch::Tensor& cov3D_precomp,
                                            const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix,
                                            const float tan_fovx, const float tan_fovy,
                                            const torch::Tensor& dL_dout_color, const torch::Tensor& dL_dout_acc,
                                            const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
                                            const torch::Tensor& geomBuffer, const int R,
                                            const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer,
                                            const bool debug);

torch::Tensor markVisible(torch::Tensor& means3D, torch::Tensor& viewmatrix, torch::Tensor& projmatrix);

struct GaussianRasterizationSettings {
  int image_height, image_width;
  float tanfovx, tanfovy;
  torch::Tensor bg;
  float scale_modifier;
  torch::Tensor viewmatrix, projmatrix;
  int sh_degree;
  torch::Tensor camera_center;
  bool prefiltered;
};

// 19 tensor inputs (the scalars boxed as 0-dim tensors), 4 outputs; backward fills the first 8 of 19 slots
class _RasterizeGaussians : public torch::autograd::Function<_RasterizeGaussians> {
 public:
  static torch::autograd::tensor_list forward(
      torch::autograd::AutogradContext* ctx, torch::Tensor means3D, torch::Tensor means2D, torch::Tensor sh,
      torch::Tensor colors_precomp, torch::Tensor opacities, torch::Tensor scales, torch::Tensor rotations,
      torch::Tensor cov3Ds_precomp, torch::Tensor image_height, torch::Tensor image_width, torch::Tensor tanfovx,
      torch::Tensor tanfovy, torch::Tensor bg, torch::Tensor scale_modifier, torch::Tensor viewmatrix,
      torch::Tensor projmatrix, torch::Tensor sh_degree, torch::Tensor camera_center, torch::Tensor prefiltered);
  static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx,
                                               torch::autograd::tensor_list grad_outputs);
};

class GaussianRasterizer : torch::nn::Module {
 public:
  GaussianRasterizer(GaussianRasterizationSettings raster_settings) : raster_settings_(raster_settings) {}
  torch::Tensor mark_visible(torch::Tensor positions);
  torch::autograd::tensor_list rasterize_gaussians(torch::Tensor means3D, torch::Tensor means2D, torch::Tensor sh,
                                                   torch::Tensor colors_precomp, torch::Tensor opacities,
                                                   torch::Tensor scales, torch::Tensor rotations,
                                                   torch::Tensor cov3Ds_precomp,
                                                   GaussianRasterizationSettings raster_settings);
  // -> (color [3,H,W], radii [P] i32, depth [1,H,W], acc [1,H,W])
  std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> forward(
      torch::Tensor means3D, torch::Tensor means2D, torch::Tensor opacities, torch::Tensor shs = torch::Tensor(),
      torch::Tensor colors_precomp = torch::Tensor(), torch::Tensor scales = torch::Tensor(),
      torch::Tensor rotations = torch::Tensor(), torch::Tensor cov3D_precomp = torch::Tensor());

 private:
  GaussianRasterizationSettings raster_settings_;
};
