// optimizer.hip -- the callers' per-Gaussian elementwise work either side of the rasterizer, fused
// ("next" row 1; widened into only after the hot path met its bar):
//
//   k_activate           GaussianModel's getters (reference include/gs/gs/gaussian.cuh:40-54):
//                        opacity = sigmoid(_opacity), scales = exp(_scaling),
//                        rotations = normalize(_rotation), shs = cat(_features_dc, _features_rest, 1)
//                        -- five Torch kernels + their autograd nodes in the reference, one launch here.
//   k_activate_backward  the matching chain rule back to the raw leaves, one launch.
//   k_adam               torch::optim::Adam::step for all parameter groups (reference src/gs/gaussian.cu:
//                        396-428: six groups, per-group lr, eps 1e-15, betas 0.9/0.999, no weight decay, no
//                        amsgrad) as ONE multi-tensor launch, optionally zeroing the gradients it consumed
//                        (the reference's zero_grad, src/liw/lioOptimization.cpp:1831-1832).
//
// All three are pure streaming kernels (HBM-bound): 16-byte accesses where the layout allows.
#include "gsr_internal.hpp"

namespace gsr {

// full-precision expf: these kernels are HBM-bound, and exp(_scaling) feeds the exact-match stages downstream
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void k_activate(const int P, const int M, const float* __restrict__ scaling_raw,
                                                  const float* __restrict__ rotation_raw,
                                                  const float* __restrict__ opacity_raw,
                                                  const float* __restrict__ f_dc, const float* __restrict__ f_rest,
                                                  float* __restrict__ scales, float* __restrict__ rotations,
                                                  float* __restrict__ opacities, float* __restrict__ shs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  opacities[i] = sigmoidf_(opacity_raw[i]);
#pragma unroll
  for (int k = 0; k < 3; k++) scales[3 * i + k] = expf(scaling_raw[3 * i + k]);
  // (scalar accesses: the rotation arrays may be views at an 8-byte offset of a flat parameter buffer)
  const float4 q = make_float4(rotation_raw[4 * i], rotation_raw[4 * i + 1], rotation_raw[4 * i + 2], rotation_raw[4 * i + 3]);
  // torch::nn::functional::normalize: x / max(||x||_2, 1e-12)
  const float nrm = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
  const float inv = 1.0f / nrm;
  rotations[4 * i] = q.x * inv; rotations[4 * i + 1] = q.y * inv; rotations[4 * i + 2] = q.z * inv;
  rotations[4 * i + 3] = q.w * inv;
  float* o = shs + (size_t)i * M * 3;
  o[0] = f_dc[3 * i]; o[1] = f_dc[3 * i + 1]; o[2] = f_dc[3 * i + 2];
  const float* r = f_rest + (size_t)i * (M - 1) * 3;
  for (int k = 0; k < (M - 1) * 3; k++) o[3 + k] = r[k];
}

__global__ __launch_bounds__(256) void k_activate_backward(
    const int P, const int M, const float* __restrict__ rotation_raw, const float* __restrict__ scales,
    const float* __restrict__ opacities, const float* __restrict__ g_scales, const float* __restrict__ g_rot,
    const float* __restrict__ g_opac, const float* __restrict__ g_shs, float* __restrict__ g_scaling_raw,
    float* __restrict__ g_rotation_raw, float* __restrict__ g_opacity_raw, float* __restrict__ g_f_dc,
    float* __restrict__ g_f_rest) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const float s = opacities[i];
  g_opacity_raw[i] = g_opac[i] * s * (1.0f - s);  // sigmoid'
#pragma unroll
  for (int k = 0; k < 3; k++) g_scaling_raw[3 * i + k] = g_scales[3 * i + k] * scales[3 * i + k];  // exp' = exp
  const float4 q = make_float4(rotation_raw[4 * i], rotation_raw[4 * i + 1], rotation_raw[4 * i + 2], rotation_raw[4 * i + 3]);
  const float4 g = make_float4(g_rot[4 * i], g_rot[4 * i + 1], g_rot[4 * i + 2], g_rot[4 * i + 3]);
  const float n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  const float nrm = sqrtf(n2);
  float4 o;
  if (nrm > 1e-12f) {  // y = x/|x|:  dx = (g - y (y.g)) / |x|
    const float inv = 1.0f / nrm;
    const float yx = q.x * inv, yy = q.y * inv, yz = q.z * inv, yw = q.w * inv;
    const float d = yx * g.x + yy * g.y + yz * g.z + yw * g.w;
    o = make_float4((g.x - yx * d) * inv, (g.y - yy * d) * inv, (g.z - yz * d) * inv, (g.w - yw * d) * inv);
  } else {  // clamped branch of normalize: y = x / 1e-12
    o = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);
  }
  g_rotation_raw[4 * i] = o.x; g_rotation_raw[4 * i + 1] = o.y; g_rotation_raw[4 * i + 2] = o.z;
  g_rotation_raw[4 * i + 3] = o.w;
  const float* gs = g_shs + (size_t)i * M * 3;
  g_f_dc[3 * i] = gs[0]; g_f_dc[3 * i + 1] = gs[1]; g_f_dc[3 * i + 2] = gs[2];
  float* r = g_f_rest + (size_t)i * (M - 1) * 3;
  for (int k = 0; k < (M - 1) * 3; k++) r[k] = gs[3 + k];
}

struct AdamArgs {
  static constexpr int MAXT = 8;
  float* p[MAXT];
  float* g[MAXT];
  float* m[MAXT];
  float* v[MAXT];
  unsigned long long end[MAXT];    // cumulative count of 4-element groups, ceil(numel/4) per tensor
  unsigned long long numel[MAXT];
  float step_size[MAXT];           // lr / (1 - beta1^t)
  int n;
  unsigned aligned;                // bit k: all four pointers of tensor k are 16-B aligned
  float beta1, beta2, omb1, omb2, eps, inv_bc2_sqrt;  // 1 - beta (rounded from double, as torch does), 1/sqrt(1 - beta2^t)
  int zero_grads;
};

__device__ __forceinline__ void adam1(float& p, float& g, float& m, float& v, const AdamArgs& a, int t) {
  // torch::optim::Adam::step: exp_avg = exp_avg*b1 + g*(1-b1); exp_avg_sq = exp_avg_sq*b2 + g*g*(1-b2);
  // denom = sqrt(exp_avg_sq)/sqrt(bias_correction2) + eps; p -= step_size * exp_avg / denom
  m = m * a.beta1 + g * a.omb1;
  v = v * a.beta2 + g * g * a.omb2;
  const float denom = sqrtf(v) * a.inv_bc2_sqrt + a.eps;
  p -= a.step_size[t] * (m / denom);
  if (a.zero_grads) g = 0.0f;
}

// One launch for every parameter tensor: thread #i owns 4 consecutive elements of the concatenation of all
// tensors (each tensor rounded up to a multiple of 4 in this index space).
__global__ __launch_bounds__(256) void k_adam(const AdamArgs a) {
  const unsigned long long idx = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  int t = 0;
  unsigned long long start = 0;
#pragma unroll
  for (int k = 0; k < AdamArgs::MAXT; k++) {
    if (k < a.n && idx >= a.end[k]) { t = k + 1; start = a.end[k]; }
  }
  if (t >= a.n) return;
  const unsigned long long e0 = (idx - start) * 4;
  const unsigned long long left = a.numel[t] - e0;
  if (left >= 4 && ((a.aligned >> t) & 1u)) {
    float4* p4 = reinterpret_cast<float4*>(a.p[t] + e0);
    float4* g4 = reinterpret_cast<float4*>(a.g[t] + e0);
    float4* m4 = reinterpret_cast<float4*>(a.m[t] + e0);
    float4* v4 = reinterpret_cast<float4*>(a.v[t] + e0);
    float4 p = *p4, g = *g4, m = *m4, v = *v4;
    adam1(p.x, g.x, m.x, v.x, a, t);
    adam1(p.y, g.y, m.y, v.y, a, t);
    adam1(p.z, g.z, m.z, v.z, a, t);
    adam1(p.w, g.w, m.w, v.w, a, t);
    *p4 = p; *m4 = m; *v4 = v;
    if (a.zero_grads) *g4 = g;
  } else {
    const int cnt = left < 4 ? (int)left : 4;
    for (int k = 0; k < cnt; k++)
      adam1(a.p[t][e0 + k], a.g[t][e0 + k], a.m[t][e0 + k], a.v[t][e0 + k], a, t);
  }
}

hipError_t launch_activate(int P, int M, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                           const float* f_dc, const float* f_rest, float* scales, float* rotations, float* opacities,
                           float* shs, hipStream_t s) {
  ProfScope ps(K_ACTIVATE, s);
  hipLaunchKernelGGL(k_activate, dim3((P + 255) / 256), dim3(256), 0, s, P, M, scaling_raw, rotation_raw, opacity_raw,
                     f_dc, f_rest, scales, rotations, opacities, shs);
  return hipGetLastError();
}

hipError_t launch_activate_backward(int P, int M, const float* rotation_raw, const float* scales,
                                    const float* opacities, const float* g_scales, const float* g_rot,
                                    const float* g_opac, const float* g_shs, float* g_scaling_raw,
                                    float* g_rotation_raw, float* g_opacity_raw, float* g_f_dc, float* g_f_rest,
                                    hipStream_t s) {
  ProfScope ps(K_ACTIVATE_BWD, s);
  hipLaunchKernelGGL(k_activate_backward, dim3((P + 255) / 256), dim3(256), 0, s, P, M, rotation_raw, scales,
                     opacities, g_scales, g_rot, g_opac, g_shs, g_scaling_raw, g_rotation_raw, g_opacity_raw, g_f_dc,
                     g_f_rest);
  return hipGetLastError();
}

hipError_t launch_adam(int n, float* const* params, float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const size_t* numel, const float* lr, double beta1, double beta2,
                       double eps, int step, int zero_grads, hipStream_t s) {
  AdamArgs a;
  a.n = n;
  a.beta1 = (float)beta1;
  a.beta2 = (float)beta2;
  a.omb1 = (float)(1.0 - beta1);  // the options are doubles in torch::optim: 1 - 0.999 = 0.001, not 1 - 0.999f
  a.omb2 = (float)(1.0 - beta2);
  a.eps = (float)eps;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  a.zero_grads = zero_grads;
  a.aligned = 0;
  unsigned long long cum = 0;
  for (int k = 0; k < AdamArgs::MAXT; k++) {
    if (k < n) {
      a.p[k] = params[k]; a.g[k] = grads[k]; a.m[k] = exp_avg[k]; a.v[k] = exp_avg_sq[k];
      a.step_size[k] = (float)((double)lr[k] / bc1);
      a.numel[k] = numel[k];
      cum += (numel[k] + 3) / 4;
      const uintptr_t bits = (uintptr_t)params[k] | (uintptr_t)grads[k] | (uintptr_t)exp_avg[k] | (uintptr_t)exp_avg_sq[k];
      if ((bits & 15u) == 0) a.aligned |= 1u << k;
    } else {
      a.p[k] = a.g[k] = a.m[k] = a.v[k] = nullptr;
      a.step_size[k] = 0.f;
      a.numel[k] = 0;
    }
    a.end[k] = cum;
  }
  ProfScope ps(K_ADAM, s);
  if (cum) hipLaunchKernelGGL(k_adam, dim3((unsigned)((cum + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace gsr
