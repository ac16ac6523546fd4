// torch_next.cpp -- definitions of gsr_torch_next.hpp: the C++/LibTorch hosts of the rows either side of the
// rasterizer, over the C ABI of libgsraster_hip.so (include/gsraster.h).  Plain C++ (no device code); PyTorch-ROCm
// supplies tensors, autograd and the current HIP stream.  Python mirrors with the same semantics: gs-livm_amd/loss.py,
// model.py, ply.py (the GPU tests compare the two routes bit for bit through the pybind11 module of torch_binding.cpp).
//
// Replaces, at their call sites:
//   gaussian_splatting::l1_loss + ssim + the loss line     include/gs/gs/loss_utils.cuh:11-13,43-70,
//                                                           src/liw/lioOptimization.cpp:1705-1710
//   GaussianModel's getters                                include/gs/gs/gaussian.cuh:40-54
//   _optimizer->step() / zero_grad()                       src/gs/gaussian.cu:396-428, lioOptimization.cpp:1831-1832
//   the tensor construction of addNewPointcloud            src/gs/gaussian.cu:241-313
//   Save_ply / Write_output_ply                            src/gs/gaussian.cu:494-573
#include "gsr_torch_next.hpp"

#include <c10/hip/HIPStream.h>

#include <cmath>
#include <cstdio>
#include <stdexcept>

#include "gsraster.h"

namespace gsr_torch {
namespace {

void* current_stream() { return static_cast<void*>(c10::hip::getCurrentHIPStream().stream()); }

void check(int code, const char* what) {
  if (code < 0) throw std::runtime_error(std::string(what) + ": " + gsr_last_error());
}

float* fp(const torch::Tensor& t) { return t.defined() && t.numel() ? t.data_ptr<float>() : nullptr; }

torch::Tensor dev_f32(const torch::Tensor& t, const char* name) {
  if (!t.defined() || !t.is_cuda() || t.scalar_type() != torch::kFloat32)
    throw std::invalid_argument(std::string(name) + ": expected a float32 tensor on the device");
  return t.contiguous();
}

int coefficients(const torch::Tensor& features_rest) {  // M = SH coefficients per channel
  return 1 + (features_rest.defined() && features_rest.numel() ? static_cast<int>(features_rest.size(1)) : 0);
}

void window_taps(const torch::Tensor& window1d, float out[11]) {
  const torch::Tensor w = (window1d.defined() ? window1d : reference_window_1d()).to(torch::kCPU, torch::kFloat32).contiguous();
  if (w.numel() != 11) throw std::invalid_argument("photometric_loss: the window has 11 taps");
  for (int k = 0; k < 11; k++) out[k] = w.data_ptr<float>()[k];
}

// one evaluation of the fused loss: out3 = [loss, l1, ssim], grad = dL/dimage (or undefined)
std::pair<torch::Tensor, torch::Tensor> run_loss(const torch::Tensor& image, const torch::Tensor& gt, float lambda,
                                                 const torch::Tensor& window1d, bool want_grad) {
  const torch::Tensor img = dev_f32(image, "image"), ref = dev_f32(gt, "gt");
  if (img.dim() != 3 || img.sizes() != ref.sizes()) throw std::invalid_argument("photometric_loss: [C,H,W] images of one shape");
  const int C = img.size(0), H = img.size(1), W = img.size(2);
  float taps[11];
  window_taps(window1d, taps);
  const size_t nbytes = gsr_photometric_loss_workspace(C, H, W);
  torch::Tensor ws = torch::empty({static_cast<long long>(nbytes)}, img.options().dtype(torch::kByte));
  torch::Tensor out3 = torch::empty({3}, img.options());
  torch::Tensor grad = want_grad ? torch::empty_like(img) : torch::Tensor();
  check(gsr_photometric_loss(C, H, W, fp(img), fp(ref), taps, lambda, out3.data_ptr<float>(), fp(grad),
                             reinterpret_cast<char*>(ws.data_ptr()), nbytes, current_stream()),
        "gsr_photometric_loss");
  return {out3, grad};
}

struct PhotometricLossFn : public torch::autograd::Function<PhotometricLossFn> {
  static torch::Tensor forward(torch::autograd::AutogradContext* ctx, torch::Tensor image, torch::Tensor gt,
                               double lambda, torch::Tensor window1d) {
    auto r = run_loss(image, gt, static_cast<float>(lambda), window1d, image.requires_grad());
    ctx->save_for_backward({r.second.defined() ? r.second : torch::empty({0}, image.options())});
    return r.first[0];
  }
  static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx,
                                               torch::autograd::tensor_list grad_outputs) {
    const torch::Tensor grad = ctx->get_saved_variables()[0];
    return {grad.numel() ? grad * grad_outputs[0] : torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor()};
  }
};

struct ActivateFn : public torch::autograd::Function<ActivateFn> {
  static torch::autograd::tensor_list forward(torch::autograd::AutogradContext* ctx, torch::Tensor scaling_raw,
                                              torch::Tensor rotation_raw, torch::Tensor opacity_raw,
                                              torch::Tensor features_dc, torch::Tensor features_rest) {
    const torch::Tensor s = dev_f32(scaling_raw, "_scaling"), r = dev_f32(rotation_raw, "_rotation"),
                        o = dev_f32(opacity_raw, "_opacity"), dc = dev_f32(features_dc, "_features_dc"),
                        rest = dev_f32(features_rest, "_features_rest");
    const int P = s.size(0), M = coefficients(rest);
    torch::Tensor scales = torch::empty({P, 3}, s.options()), rot = torch::empty({P, 4}, s.options()),
                  opac = torch::empty({P, 1}, s.options()), shs = torch::empty({P, M, 3}, s.options());
    check(gsr_activate(P, M, fp(s), fp(r), fp(o), fp(dc), fp(rest), fp(scales), fp(rot), fp(opac), fp(shs),
                       current_stream()),
          "gsr_activate");
    ctx->save_for_backward({r, scales, opac});
    ctx->saved_data["M"] = M;
    return {scales, rot, opac, shs};
  }
  static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx,
                                               torch::autograd::tensor_list g) {
    const auto saved = ctx->get_saved_variables();
    const torch::Tensor rotation_raw = saved[0], scales = saved[1], opac = saved[2];
    const int P = scales.size(0), M = ctx->saved_data["M"].toInt();
    auto or_zero = [&](const torch::Tensor& t, std::initializer_list<int64_t> shape) {
      return t.defined() ? t.contiguous() : torch::zeros(shape, scales.options());
    };
    const torch::Tensor gs = or_zero(g[0], {P, 3}), gr = or_zero(g[1], {P, 4}), go = or_zero(g[2], {P, 1}),
                        gsh = or_zero(g[3], {P, M, 3});
    torch::Tensor d_s = torch::empty({P, 3}, scales.options()), d_r = torch::empty({P, 4}, scales.options()),
                  d_o = torch::empty({P, 1}, scales.options()), d_dc = torch::empty({P, 1, 3}, scales.options()),
                  d_rest = torch::empty({P, M - 1, 3}, scales.options());
    check(gsr_activate_backward(P, M, fp(rotation_raw), fp(scales), fp(opac), fp(gs), fp(gr), fp(go), fp(gsh), fp(d_s),
                                fp(d_r), fp(d_o), fp(d_dc), fp(d_rest), current_stream()),
          "gsr_activate_backward");
    return {d_s, d_r, d_o, d_dc, d_rest};
  }
};

}  // namespace

torch::Tensor reference_window_1d(int window_size, float sigma) {
  torch::Tensor g = torch::empty({window_size}, torch::kFloat32);
  for (int x = 0; x < window_size; ++x)  // loss_utils.cuh:27 (floor of the halved OFFSET: the reference's quirk)
    g[x] = std::exp(-(std::pow(std::floor(static_cast<float>(x - window_size) / 2.f), 2)) / (2.f * sigma * sigma));
  return g / g.sum();
}

torch::Tensor photometric_loss(const torch::Tensor& image, const torch::Tensor& gt, float lambda_dssim,
                               const torch::Tensor& window1d) {
  return PhotometricLossFn::apply(image, gt, static_cast<double>(lambda_dssim),
                                  window1d.defined() ? window1d : reference_window_1d());
}

torch::Tensor photometric_loss_parts(const torch::Tensor& image, const torch::Tensor& gt, float lambda_dssim,
                                     const torch::Tensor& window1d) {
  torch::NoGradGuard no_grad;
  return run_loss(image, gt, lambda_dssim, window1d, false).first;
}

Activated activate(const torch::Tensor& scaling_raw, const torch::Tensor& rotation_raw, const torch::Tensor& opacity_raw,
                   const torch::Tensor& features_dc, const torch::Tensor& features_rest) {
  auto r = ActivateFn::apply(scaling_raw, rotation_raw, opacity_raw, features_dc, features_rest);
  return Activated{r[0], r[1], r[2], r[3]};
}

FusedAdam::FusedAdam(std::vector<torch::Tensor> params, std::vector<double> lrs, double beta1, double beta2, double eps)
    : params_(std::move(params)), lrs_(std::move(lrs)), beta1_(beta1), beta2_(beta2), eps_(eps) {
  if (params_.size() != lrs_.size()) throw std::invalid_argument("FusedAdam: one learning rate per parameter tensor");
  for (const auto& p : params_) {
    if (!p.is_cuda() || p.scalar_type() != torch::kFloat32 || !p.is_contiguous())
      throw std::invalid_argument("FusedAdam: contiguous float32 device tensors");
    m_.push_back(torch::zeros_like(p));
    v_.push_back(torch::zeros_like(p));
  }
}

void FusedAdam::step(bool zero_grad) {
  torch::NoGradGuard no_grad;
  std::vector<size_t> idx;
  for (size_t k = 0; k < params_.size(); k++)
    if (params_[k].numel() && params_[k].grad().defined()) idx.push_back(k);
  ++step_;
  for (size_t b = 0; b < idx.size(); b += 8) {
    float *p[8], *g[8], *m[8], *v[8];
    size_t numel[8];
    float lr[8];
    std::vector<torch::Tensor> keep;
    const int n = static_cast<int>(std::min<size_t>(8, idx.size() - b));
    for (int j = 0; j < n; j++) {
      const size_t k = idx[b + j];
      torch::Tensor grad = params_[k].grad();
      if (!grad.is_contiguous()) throw std::invalid_argument("FusedAdam: non-contiguous gradient");
      p[j] = params_[k].data_ptr<float>(); g[j] = grad.data_ptr<float>();
      m[j] = m_[k].data_ptr<float>(); v[j] = v_[k].data_ptr<float>();
      numel[j] = static_cast<size_t>(params_[k].numel());
      lr[j] = static_cast<float>(lrs_[k]);
    }
    check(gsr_adam_step(n, p, g, m, v, numel, lr, beta1_, beta2_, eps_, static_cast<int>(step_), zero_grad ? 1 : 0,
                        current_stream()),
          "gsr_adam_step");
  }
}

Activated FusedAdam::step_model(const torch::Tensor& g_xyz, const torch::Tensor& g_scaling,
                                const torch::Tensor& g_rotation, const torch::Tensor& g_opacity,
                                const torch::Tensor& g_features) {
  torch::NoGradGuard no_grad;
  if (params_.size() != 6) throw std::invalid_argument("FusedAdam::step_model: the six leaves of a GaussianModel");
  const int P = params_[0].size(0), M = coefficients(params_[2]);
  const torch::Tensor gx = dev_f32(g_xyz, "g_xyz"), gs = dev_f32(g_scaling, "g_scaling"),
                      gr = dev_f32(g_rotation, "g_rotation"), go = dev_f32(g_opacity, "g_opacity"),
                      gf = dev_f32(g_features, "g_features");
  float *p[6], *m[6], *v[6], lr[6];
  for (int k = 0; k < 6; k++) {
    p[k] = fp(params_[k]); m[k] = fp(m_[k]); v[k] = fp(v_[k]);
    lr[k] = static_cast<float>(lrs_[k]);
  }
  const auto o = params_[0].options();
  Activated a{torch::empty({P, 3}, o), torch::empty({P, 4}, o), torch::empty({P, 1}, o), torch::empty({P, M, 3}, o)};
  ++step_;
  check(gsr_model_step(P, M, p, m, v, fp(gx), fp(gs), fp(gr), fp(go), fp(gf), fp(a.scaling), fp(a.rotation),
                       fp(a.opacity), fp(a.features), lr, beta1_, beta2_, eps_, static_cast<int>(step_),
                       current_stream()),
        "gsr_model_step");
  return a;
}

void FusedAdam::replace_param(size_t index, torch::Tensor new_param) {
  torch::NoGradGuard no_grad;
  if (index >= params_.size()) throw std::out_of_range("FusedAdam::replace_param");
  const int64_t old_rows = params_[index].size(0), new_rows = new_param.size(0);
  if (new_rows < old_rows) throw std::invalid_argument("FusedAdam::replace_param: the tensor shrank");
  auto grow = [&](torch::Tensor& mom) {  // cat({old, zeros_like(extension)}), gaussian.cu:462-466
    torch::Tensor t = torch::zeros_like(new_param);
    if (old_rows) t.narrow(0, 0, old_rows).copy_(mom);
    mom = t;
  };
  grow(m_[index]);
  grow(v_[index]);
  params_[index] = std::move(new_param);
}

void init_gaussians(const torch::Tensor& xyz, const torch::Tensor& covs, const torch::Tensor& rgbs, float scale_factor,
                    torch::Tensor xyz_out, torch::Tensor features_dc_out, torch::Tensor features_rest_out,
                    torch::Tensor scaling_out, torch::Tensor rotation_out, torch::Tensor opacity_out) {
  torch::NoGradGuard no_grad;
  const torch::Tensor x = dev_f32(xyz, "xyz"), c = dev_f32(covs, "covs"), rgb = dev_f32(rgbs, "rgbs");
  const int n = x.size(0), M = coefficients(features_rest_out);
  if (c.dim() != 3 || c.size(0) != n || c.size(1) != 3 || c.size(2) != 3 || rgb.size(0) != n)
    throw std::invalid_argument("init_gaussians: xyz [n,3], covs [n,3,3], rgbs [n,3]");
  for (const torch::Tensor* t : {&xyz_out, &features_dc_out, &scaling_out, &rotation_out, &opacity_out})
    if (!t->is_cuda() || !t->is_contiguous() || t->size(0) != n)
      throw std::invalid_argument("init_gaussians: outputs are contiguous n-row device views");
  if (M > 1 && (!features_rest_out.is_contiguous() || features_rest_out.size(0) != n))
    throw std::invalid_argument("init_gaussians: features_rest output");
  check(gsr_init_gaussians(n, M, fp(x), fp(c), fp(rgb), scale_factor, fp(xyz_out), fp(features_dc_out),
                           fp(features_rest_out), fp(scaling_out), fp(rotation_out), fp(opacity_out), current_stream()),
        "gsr_init_gaussians");
}

torch::Tensor pack_ply_rows(const torch::Tensor& xyz, const torch::Tensor& features_dc,
                            const torch::Tensor& features_rest, const torch::Tensor& opacity,
                            const torch::Tensor& scaling, const torch::Tensor& rotation) {
  torch::NoGradGuard no_grad;
  const torch::Tensor x = dev_f32(xyz, "xyz"), dc = dev_f32(features_dc, "features_dc"),
                      rest = dev_f32(features_rest, "features_rest"), o = dev_f32(opacity, "opacity"),
                      s = dev_f32(scaling, "scaling"), r = dev_f32(rotation, "rotation");
  const int P = x.size(0), M = coefficients(rest);
  torch::Tensor rows = torch::empty({P, static_cast<long long>(gsr_ply_row_floats(M))}, x.options());
  check(gsr_pack_ply_rows(P, M, fp(x), fp(dc), fp(rest), fp(o), fp(s), fp(r), fp(rows), current_stream()),
        "gsr_pack_ply_rows");
  return rows;
}

std::vector<std::string> ply_attribute_names(int M) {  // construct_list_of_attributes, gaussian.cu:474-492
  std::vector<std::string> names = {"x", "y", "z", "nx", "ny", "nz"};
  for (int i = 0; i < 3; i++) names.push_back("f_dc_" + std::to_string(i));
  for (int i = 0; i < 3 * (M - 1); i++) names.push_back("f_rest_" + std::to_string(i));
  names.push_back("opacity");
  for (int i = 0; i < 3; i++) names.push_back("scale_" + std::to_string(i));
  for (int i = 0; i < 4; i++) names.push_back("rot_" + std::to_string(i));
  return names;
}

size_t write_ply(const std::string& file_path, const torch::Tensor& xyz, const torch::Tensor& features_dc,
                 const torch::Tensor& features_rest, const torch::Tensor& opacity, const torch::Tensor& scaling,
                 const torch::Tensor& rotation) {
  const torch::Tensor rows = pack_ply_rows(xyz, features_dc, features_rest, opacity, scaling, rotation).cpu();  // ONE D2H copy
  const int M = coefficients(features_rest);
  std::string header = "ply\nformat binary_little_endian 1.0\nelement vertex " + std::to_string(rows.size(0)) + "\n";
  for (const auto& n : ply_attribute_names(M)) header += "property float " + n + "\n";
  header += "end_header\n";
  FILE* f = std::fopen(file_path.c_str(), "wb");
  if (!f) throw std::runtime_error("write_ply: cannot open " + file_path);
  const size_t nbytes = static_cast<size_t>(rows.numel()) * sizeof(float);
  const bool ok = std::fwrite(header.data(), 1, header.size(), f) == header.size() &&
                  (nbytes == 0 || std::fwrite(rows.data_ptr<float>(), 1, nbytes, f) == nbytes);
  if (std::fclose(f) != 0 || !ok) throw std::runtime_error("write_ply: short write to " + file_path);
  return header.size() + nbytes;
}

}  // namespace gsr_torch
