// gsr_torch_next.hpp -- C++/LibTorch hosts of the rows either side of the rasterizer (SURVEY.md 8(f) "next" rows 1, 2
// and 4), for GS-LIVM's own translation units: what src/liw/lioOptimization.cpp and src/gs/gaussian.cu call instead of
// the Torch-op sequences they run today.  Definitions: torch_next.cpp (plain C++, links libgsraster_hip.so through the
// C ABI of include/gsraster.h).  The reference has no declarations for these -- its versions are header-inline Torch
// code (loss_utils.cuh) or private members of GaussianModel -- so this header is what a maintainer includes;
// INTEGRATION.md section 3 shows the call sites.  oracle/ref_link/caller.cpp odr-uses every entry point next to the
// reference's own headers (`make -C oracle ref_link`).
#pragma once
#include <torch/torch.h>

#include <string>
#include <tuple>
#include <vector>

namespace gsr_torch {

// ---- row 2: photometric loss ------------------------------------------------------------------------------------
// gaussian_splatting::gaussian(window_size, sigma) (include/gs/gs/loss_utils.cuh:24-31), bug for bug: the exponent
// uses floor((x - window_size) / 2), so the window is not the centred Gaussian.  CPU f32 [window_size].
torch::Tensor reference_window_1d(int window_size = 11, float sigma = 1.5f);

// (1 - lambda_dssim) * l1_loss(image, gt) + lambda_dssim * (1 - ssim(image, gt, create_window(11, C), 11, C))
// (loss_utils.cuh:11-13, 43-70; src/liw/lioOptimization.cpp:1705-1710) as ONE autograd node on the fused kernels of
// csrc/loss.hip.  image, gt: [C, H, W] f32 on the device; gradient w.r.t. image only.  window1d: 11 taps (CPU or
// device), undefined = reference_window_1d().  Returns the 0-dim loss.
torch::Tensor photometric_loss(const torch::Tensor& image, const torch::Tensor& gt, float lambda_dssim = 0.2f,
                               const torch::Tensor& window1d = torch::Tensor());
// [loss, l1, ssim] of the same evaluation, no graph (the reference logs PSNR / SSIM every 50 iterations)
torch::Tensor photometric_loss_parts(const torch::Tensor& image, const torch::Tensor& gt, float lambda_dssim = 0.2f,
                                     const torch::Tensor& window1d = torch::Tensor());

// ---- row 1: activations + Adam ----------------------------------------------------------------------------------
// The five getters of GaussianModel (include/gs/gs/gaussian.cuh:40-54) as one autograd node:
//   scaling = exp(_scaling), rotation = normalize(_rotation), opacity = sigmoid(_opacity),
//   features = cat({_features_dc, _features_rest}, 1)
struct Activated {
  torch::Tensor scaling, rotation, opacity, features;
};
Activated activate(const torch::Tensor& scaling_raw, const torch::Tensor& rotation_raw, const torch::Tensor& opacity_raw,
                   const torch::Tensor& features_dc, const torch::Tensor& features_rest);

// torch::optim::Adam as GaussianModel::Training_setup configures it (src/gs/gaussian.cu:396-428: one group per leaf,
// betas (0.9, 0.999), eps 1e-15, no weight decay / amsgrad) with _optimizer->step() + zero_grad()
// (lioOptimization.cpp:1831-1832) in one launch per eight tensors.  The moments are owned here.
class FusedAdam {
 public:
  // params / lrs in the reference's group order: _xyz, _features_dc, _features_rest, _scaling, _rotation, _opacity
  FusedAdam(std::vector<torch::Tensor> params, std::vector<double> lrs, double beta1 = 0.9, double beta2 = 0.999,
            double eps = 1e-15);
  // every param with a defined .grad() is stepped; zero_grad: the gradients are cleared by the same kernel
  void step(bool zero_grad = true);
  // The whole optimiser tail in ONE launch (k_model_step): chain rule of the activations applied to the gradients
  // w.r.t. the ACTIVATED tensors (what the rasterizer's backward returns), Adam on the six groups, and the activated
  // values of the updated parameters for the next forward.  g_xyz [P,3], g_scaling [P,3], g_rotation [P,4],
  // g_opacity [P,1], g_features [P,M,3].  Needs exactly the six leaves of a GaussianModel in the order above.
  Activated step_model(const torch::Tensor& g_xyz, const torch::Tensor& g_scaling, const torch::Tensor& g_rotation,
                       const torch::Tensor& g_opacity, const torch::Tensor& g_features);
  // GaussianModel::cat_tensors_to_optimizer (gaussian.cu:451-472): the leaf at `index` was replaced by a longer tensor
  // whose first rows are the old ones; its moments grow by zero rows
  void replace_param(size_t index, torch::Tensor new_param);
  int64_t step_count() const { return step_; }
  const std::vector<torch::Tensor>& params() const { return params_; }
  const std::vector<torch::Tensor>& exp_avg() const { return m_; }
  const std::vector<torch::Tensor>& exp_avg_sq() const { return v_; }

 private:
  std::vector<torch::Tensor> params_, m_, v_;
  std::vector<double> lrs_;
  double beta1_, beta2_, eps_;
  int64_t step_ = 0;
};

// ---- row 4: map growth and PLY export ---------------------------------------------------------------------------
// The tensor construction of GaussianModel::addNewPointcloud (src/gs/gaussian.cu:241-313) for n new points -- xyz
// [n,3], covs [n,3,3], rgbs [n,3] in 0..255 -- written IN PLACE into n-row views (normally the tail rows of capacity
// buffers): scaling = log(sqrt(diag(cov) * scale_factor)), rotation = (1,0,0,0), opacity = inverse_sigmoid(0.5) = 0,
// features_dc = RGB2SH(rgb / 255), features_rest = 0.
void init_gaussians(const torch::Tensor& xyz, const torch::Tensor& covs, const torch::Tensor& rgbs, float scale_factor,
                    torch::Tensor xyz_out, torch::Tensor features_dc_out, torch::Tensor features_rest_out,
                    torch::Tensor scaling_out, torch::Tensor rotation_out, torch::Tensor opacity_out);

// Vertex rows of GaussianModel::Save_ply (gaussian.cu:494-522) interleaved on the device: [P, 14 + 3 M] f32 in the
// order of construct_list_of_attributes (:474-492).
torch::Tensor pack_ply_rows(const torch::Tensor& xyz, const torch::Tensor& features_dc,
                            const torch::Tensor& features_rest, const torch::Tensor& opacity,
                            const torch::Tensor& scaling, const torch::Tensor& rotation);
// Write_output_ply (gaussian.cu:542-573) without tinyply: one D2H copy of the packed rows behind a header that is byte
// for byte what the reference's vendored tinyply writes (tests/golden/ply_*.ply).  Returns the bytes written.
size_t write_ply(const std::string& file_path, const torch::Tensor& xyz, const torch::Tensor& features_dc,
                 const torch::Tensor& features_rest, const torch::Tensor& opacity, const torch::Tensor& scaling,
                 const torch::Tensor& rotation);
std::vector<std::string> ply_attribute_names(int M);  // construct_list_of_attributes

}  // namespace gsr_torch
