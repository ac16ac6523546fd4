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
  if (M == 1) {  // shs = f_dc; longer rows are moved by k_sh_move
    shs[3 * i] = f_dc[3 * i]; shs[3 * i + 1] = f_dc[3 * i + 1]; shs[3 * i + 2] = f_dc[3 * i + 2];
  }
}

// M > 1: shs = cat(f_dc, f_rest, 1) (SPLIT = false) or its adjoint, the split of dL/dshs into the two leaves'
// gradients (SPLIT = true), as a flat copy: a thread owns 4 consecutive floats of the [P][M][3] side (one
// 16-byte access when aligned) and walks the matching elements of the [P][1][3] / [P][M-1][3] side, which are
// consecutive except at the row seams -- every wave instruction covers one contiguous span either way.
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_sh_move(const unsigned long long numel, const int C, float* __restrict__ wide,
                                                 float* __restrict__ f_dc, float* __restrict__ f_rest, const int al) {
  const unsigned long long e0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e0 >= numel) return;
  unsigned long long n;
  int c;
  if (numel <= 0xFFFFFFFFull) { const uint32_t q = (uint32_t)e0 / (uint32_t)C; n = q; c = (int)((uint32_t)e0 - q * (uint32_t)C); }
  else { n = e0 / (unsigned long long)C; c = (int)(e0 - n * (unsigned long long)C); }
  const int cnt = numel - e0 < 4 ? (int)(numel - e0) : 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (SPLIT) {
    if (al && cnt == 4) { const float4 t = *reinterpret_cast<const float4*>(wide + e0); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else for (int k = 0; k < cnt; k++) v[k] = wide[e0 + k];
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (k < cnt) {
      float* leaf = c < 3 ? f_dc + n * 3 + c : f_rest + n * (unsigned long long)(C - 3) + (c - 3);
      if (SPLIT) *leaf = v[k]; else v[k] = *leaf;
      if (++c == C) { c = 0; n++; }
    }
  }
  if (!SPLIT) {
    if (al && cnt == 4) *reinterpret_cast<float4*>(wide + e0) = make_float4(v[0], v[1], v[2], v[3]);
    else for (int k = 0; k < cnt; k++) wide[e0 + k] = v[k];
  }
}

template <bool SPLIT>
static void launch_sh_move(int P, int M, float* wide, float* f_dc, float* f_rest, hipStream_t s) {
  const unsigned long long numel = (unsigned long long)P * 3ull * (unsigned long long)M;
  const unsigned long long groups = (numel + 3) / 4;
  hipLaunchKernelGGL(k_sh_move<SPLIT>, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, numel, 3 * M, wide, f_dc,
                     f_rest, (reinterpret_cast<uintptr_t>(wide) & 15u) == 0 ? 1 : 0);
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
  if (M == 1) {  // longer rows are split by k_sh_move
    g_f_dc[3 * i] = g_shs[3 * i]; g_f_dc[3 * i + 1] = g_shs[3 * i + 1]; g_f_dc[3 * i + 2] = g_shs[3 * i + 2];
  }
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

// ------------------------------------------------------------------------------------------------
// k_model_step: the whole optimiser tail of one iteration in ONE pass over the model --
//   chain rule of the activations (= k_activate_backward) -> Adam on all six groups (= k_adam) ->
//   the activated values of the UPDATED parameters for the next forward (= k_activate),
// one thread per Gaussian.  Same arithmetic as the three separate kernels (the activated scales / opacity the
// chain rule needs are recomputed from the raw parameters: bit-identical to what the forward used), but the
// raw-space gradients never exist in memory and the parameters are read once: 436 instead of 608 bytes per
// Gaussian at M = 1, one launch instead of three.
// ------------------------------------------------------------------------------------------------
struct StepArgs {
  int P, M;
  float *xyz, *fdc, *frest, *scaling, *rot, *opac;  // raw parameters, updated in place
  float* m[6];                                      // exp_avg    in group order xyz, f_dc, f_rest, scaling, rotation, opacity
  float* v[6];                                      // exp_avg_sq
  const float *g_xyz, *g_scales, *g_rot, *g_opac, *g_shs;  // dL/d(xyz, activated scales / rotations / opacities / shs)
  float *a_scales, *a_rot, *a_opac, *a_shs;         // activated values of the updated parameters (may be null)
  float step_size[6];
  float beta1, beta2, omb1, omb2, eps, inv_bc2_sqrt;
  unsigned long long end[6];    // cumulative count of 4-element groups, ceil(numel/4) per tensor
  unsigned long long numel[6];
  unsigned aligned;             // bit k: every pointer the 16-byte path of tensor k touches is 16-B aligned
};

__device__ __forceinline__ float adam_update(float p, float g, float& m, float& v, const StepArgs& a, int grp) {
  m = m * a.beta1 + g * a.omb1;
  v = v * a.beta2 + g * g * a.omb2;
  const float denom = sqrtf(v) * a.inv_bc2_sqrt + a.eps;
  return p - a.step_size[grp] * (m / denom);
}

// Work decomposition as in k_adam: thread #idx owns 4 consecutive floats of the virtual concatenation of the six
// parameter tensors (each rounded up to a multiple of 4), so every access is a coalesced 16-byte one; the
// activation kind of the tensor decides how the gradient is pulled back and what is written for the next forward:
//   xyz: identity | f_dc, f_rest: identity, gradient / value live inside shs [P][M][3] | scaling: exp |
//   rotation: normalise (one thread = one quaternion) | opacity: sigmoid.
// The 16-byte path needs 16-byte-aligned tensors (StepArgs::aligned) and, for the two SH tensors, M == 1 (then
// shs and f_dc coincide); otherwise the thread handles its 4 elements one by one.
__device__ __forceinline__ void step_elem(const StepArgs& a, int t, unsigned long long e) {
  const int M = a.M, R1 = 3 * (M - 1);
  float* P_[6] = {a.xyz, a.fdc, a.frest, a.scaling, a.rot, a.opac};
  float p = P_[t][e], m = a.m[t][e], v = a.v[t][e], g;
  if (t == 0) g = a.g_xyz[e];
  else if (t == 1) g = a.g_shs[(e / 3) * (unsigned long long)(3 * M) + e % 3];
  else if (t == 2) g = a.g_shs[(e / R1) * (unsigned long long)(3 * M) + 3 + e % R1];
  else if (t == 3) g = a.g_scales[e] * expf(p);
  else { const float sg = sigmoidf_(p); g = a.g_opac[e] * sg * (1.0f - sg); }
  p = adam_update(p, g, m, v, a, t);
  P_[t][e] = p; a.m[t][e] = m; a.v[t][e] = v;
  if (t == 1 && a.a_shs) a.a_shs[(e / 3) * (unsigned long long)(3 * M) + e % 3] = p;
  if (t == 2 && a.a_shs) a.a_shs[(e / R1) * (unsigned long long)(3 * M) + 3 + e % R1] = p;
  if (t == 3 && a.a_scales) a.a_scales[e] = expf(p);
  if (t == 5 && a.a_opac) a.a_opac[e] = sigmoidf_(p);
}

__global__ __launch_bounds__(256) void k_model_step(const StepArgs a) {
  const unsigned long long idx = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  int t = 0;
  unsigned long long start = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    if (idx >= a.end[k]) { t = k + 1; start = a.end[k]; }
  }
  if (t >= 6) return;
  const unsigned long long e0 = (idx - start) * 4;
  const unsigned long long left = a.numel[t] - e0;
  float* P_[6] = {a.xyz, a.fdc, a.frest, a.scaling, a.rot, a.opac};
  if (t == 4) {  // one quaternion (numel is a multiple of 4)
    const bool al = (a.aligned >> 4) & 1u;  // wave-uniform
    const float4 q = al ? *reinterpret_cast<const float4*>(a.rot + e0)
                        : make_float4(a.rot[e0], a.rot[e0 + 1], a.rot[e0 + 2], a.rot[e0 + 3]);
    const float4 g = al ? *reinterpret_cast<const float4*>(a.g_rot + e0)
                        : make_float4(a.g_rot[e0], a.g_rot[e0 + 1], a.g_rot[e0 + 2], a.g_rot[e0 + 3]);
    const float4 m4 = al ? *reinterpret_cast<const float4*>(a.m[4] + e0)
                         : make_float4(a.m[4][e0], a.m[4][e0 + 1], a.m[4][e0 + 2], a.m[4][e0 + 3]);
    const float4 v4 = al ? *reinterpret_cast<const float4*>(a.v[4] + e0)
                         : make_float4(a.v[4][e0], a.v[4][e0 + 1], a.v[4][e0 + 2], a.v[4][e0 + 3]);
    const float nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    float go[4];
    if (nrm > 1e-12f) {  // y = x/|x|:  dx = (g - y (y.g)) / |x|   (as k_activate_backward)
      const float inv = 1.0f / nrm;
      const float yx = q.x * inv, yy = q.y * inv, yz = q.z * inv, yw = q.w * inv;
      const float d = yx * g.x + yy * g.y + yz * g.z + yw * g.w;
      go[0] = (g.x - yx * d) * inv; go[1] = (g.y - yy * d) * inv; go[2] = (g.z - yz * d) * inv; go[3] = (g.w - yw * d) * inv;
    } else {
      go[0] = g.x * 1e12f; go[1] = g.y * 1e12f; go[2] = g.z * 1e12f; go[3] = g.w * 1e12f;
    }
    const float qo[4] = {q.x, q.y, q.z, q.w};
    float mo[4] = {m4.x, m4.y, m4.z, m4.w}, vo[4] = {v4.x, v4.y, v4.z, v4.w}, pn[4];
#pragma unroll
    for (int k = 0; k < 4; k++) pn[k] = adam_update(qo[k], go[k], mo[k], vo[k], a, 4);
    const float inv2 = 1.0f / fmaxf(sqrtf(pn[0] * pn[0] + pn[1] * pn[1] + pn[2] * pn[2] + pn[3] * pn[3]), 1e-12f);
    if (al) {
      *reinterpret_cast<float4*>(a.rot + e0) = make_float4(pn[0], pn[1], pn[2], pn[3]);
      *reinterpret_cast<float4*>(a.m[4] + e0) = make_float4(mo[0], mo[1], mo[2], mo[3]);
      *reinterpret_cast<float4*>(a.v[4] + e0) = make_float4(vo[0], vo[1], vo[2], vo[3]);
      if (a.a_rot)
        *reinterpret_cast<float4*>(a.a_rot + e0) = make_float4(pn[0] * inv2, pn[1] * inv2, pn[2] * inv2, pn[3] * inv2);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        a.rot[e0 + k] = pn[k]; a.m[4][e0 + k] = mo[k]; a.v[4][e0 + k] = vo[k];
        if (a.a_rot) a.a_rot[e0 + k] = pn[k] * inv2;
      }
    }
    return;
  }
  const bool sh_direct = a.M == 1;  // shs == f_dc in memory
  if (left >= 4 && ((a.aligned >> t) & 1u) && (t == 0 || t == 3 || t == 5 || (t == 1 && sh_direct))) {
    float4 p = *reinterpret_cast<float4*>(P_[t] + e0);
    float4 m = *reinterpret_cast<float4*>(a.m[t] + e0);
    float4 v = *reinterpret_cast<float4*>(a.v[t] + e0);
    const float* gsrc = t == 0 ? a.g_xyz : t == 1 ? a.g_shs : t == 3 ? a.g_scales : a.g_opac;
    float4 g = *reinterpret_cast<const float4*>(gsrc + e0);
    if (t == 3) { g.x *= expf(p.x); g.y *= expf(p.y); g.z *= expf(p.z); g.w *= expf(p.w); }
    if (t == 5) {
      const float s0 = sigmoidf_(p.x), s1 = sigmoidf_(p.y), s2 = sigmoidf_(p.z), s3 = sigmoidf_(p.w);
      g.x *= s0 * (1.0f - s0); g.y *= s1 * (1.0f - s1); g.z *= s2 * (1.0f - s2); g.w *= s3 * (1.0f - s3);
    }
    p.x = adam_update(p.x, g.x, m.x, v.x, a, t);
    p.y = adam_update(p.y, g.y, m.y, v.y, a, t);
    p.z = adam_update(p.z, g.z, m.z, v.z, a, t);
    p.w = adam_update(p.w, g.w, m.w, v.w, a, t);
    *reinterpret_cast<float4*>(P_[t] + e0) = p;
    *reinterpret_cast<float4*>(a.m[t] + e0) = m;
    *reinterpret_cast<float4*>(a.v[t] + e0) = v;
    if (t == 1 && a.a_shs) *reinterpret_cast<float4*>(a.a_shs + e0) = p;
    if (t == 3 && a.a_scales)
      *reinterpret_cast<float4*>(a.a_scales + e0) = make_float4(expf(p.x), expf(p.y), expf(p.z), expf(p.w));
    if (t == 5 && a.a_opac)
      *reinterpret_cast<float4*>(a.a_opac + e0) = make_float4(sigmoidf_(p.x), sigmoidf_(p.y), sigmoidf_(p.z), sigmoidf_(p.w));
  } else if (left >= 4 && ((a.aligned >> t) & 1u) && (t == 1 || t == 2) && a.numel[t] <= 0xFFFFFFF0ull) {
    // SH tensors at M > 1: parameter and moments are contiguous (16-byte accesses); the gradient and the
    // activated copy live inside shs [P][M][3] -- 4 dword accesses each, consecutive except where the 4
    // elements cross into the next Gaussian's row.  One 32-bit division per thread (step_elem does three
    // 64-bit ones per element).
    const uint32_t W = t == 1 ? 3u : 3u * (uint32_t)(a.M - 1), C = 3u * (uint32_t)a.M, skip = t == 1 ? 0u : 3u;
    uint32_t n = (uint32_t)e0 / W, r = (uint32_t)e0 - n * W;
    unsigned long long si[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      si[k] = (unsigned long long)n * C + skip + r;
      if (++r == W) { r = 0; n++; }
    }
    float4 p = *reinterpret_cast<float4*>(P_[t] + e0);
    float4 m = *reinterpret_cast<float4*>(a.m[t] + e0);
    float4 v = *reinterpret_cast<float4*>(a.v[t] + e0);
    const float g0 = a.g_shs[si[0]], g1 = a.g_shs[si[1]], g2 = a.g_shs[si[2]], g3 = a.g_shs[si[3]];
    p.x = adam_update(p.x, g0, m.x, v.x, a, t);
    p.y = adam_update(p.y, g1, m.y, v.y, a, t);
    p.z = adam_update(p.z, g2, m.z, v.z, a, t);
    p.w = adam_update(p.w, g3, m.w, v.w, a, t);
    *reinterpret_cast<float4*>(P_[t] + e0) = p;
    *reinterpret_cast<float4*>(a.m[t] + e0) = m;
    *reinterpret_cast<float4*>(a.v[t] + e0) = v;
    if (a.a_shs) { a.a_shs[si[0]] = p.x; a.a_shs[si[1]] = p.y; a.a_shs[si[2]] = p.z; a.a_shs[si[3]] = p.w; }
  } else {
    const int cnt = left < 4 ? (int)left : 4;
    for (int k = 0; k < cnt; k++) step_elem(a, t, e0 + k);
  }
}

// k_sh_step: the SH part of the tail at M > 1.  The two leaves ([P][1][3] and [P][M-1][3]) and their moments are
// contiguous per tensor, the gradient and the activated copy are rows of shs [P][M][3]: a workgroup takes SH_ROWS
// Gaussians, brings their dL/dshs rows into LDS as one contiguous block, updates the leaves with 16-byte
// accesses (gradient read from / new value written to the LDS rows) and stores the rows back out as the next
// forward's shs -- every global access is a coalesced 16-byte one.  Needs 16-byte-aligned tensors; otherwise
// k_model_step's generic path does the same arithmetic.
__device__ __forceinline__ void sh_leaf_step(const StepArgs& a, const int t, float* buf, float* __restrict__ p_,
                                             float* __restrict__ m_, float* __restrict__ v_, const int count,
                                             const int W, const int C, const int skip) {
  const int n4 = count >> 2;
  for (int j = threadIdx.x; j < n4; j += 256) {
    float4 p = reinterpret_cast<float4*>(p_)[j], m = reinterpret_cast<float4*>(m_)[j], v = reinterpret_cast<float4*>(v_)[j];
    int n = (4 * j) / W, r = 4 * j - n * W;
    int li[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      li[k] = n * C + skip + r;
      if (++r == W) { r = 0; n++; }
    }
    p.x = adam_update(p.x, buf[li[0]], m.x, v.x, a, t);
    p.y = adam_update(p.y, buf[li[1]], m.y, v.y, a, t);
    p.z = adam_update(p.z, buf[li[2]], m.z, v.z, a, t);
    p.w = adam_update(p.w, buf[li[3]], m.w, v.w, a, t);
    reinterpret_cast<float4*>(p_)[j] = p; reinterpret_cast<float4*>(m_)[j] = m; reinterpret_cast<float4*>(v_)[j] = v;
    buf[li[0]] = p.x; buf[li[1]] = p.y; buf[li[2]] = p.z; buf[li[3]] = p.w;
  }
  const int e = 4 * n4 + (int)threadIdx.x;  // the last partial group of the last workgroup
  if (e < count) {
    const int n = e / W, li = n * C + skip + (e - n * W);
    float p = p_[e], m = m_[e], v = v_[e];
    p = adam_update(p, buf[li], m, v, a, t);
    p_[e] = p; m_[e] = m; v_[e] = v;
    buf[li] = p;
  }
}

constexpr int SH_ROWS = 64;  // 12 KB of LDS at M = 16: eight workgroups per CU keep loads, arithmetic and stores overlapped

__global__ __launch_bounds__(256) void k_sh_step(const StepArgs a) {
  extern __shared__ float sh_buf[];  // [nrows][3M]
  const int C = 3 * a.M, W = C - 3;
  const size_t row0 = (size_t)blockIdx.x * SH_ROWS;
  const int nrows = a.P - (long long)row0 < SH_ROWS ? (int)(a.P - (long long)row0) : SH_ROWS;
  const int n = nrows * C, n4 = n >> 2;
  const float* gsrc = a.g_shs + row0 * C;
  for (int j = threadIdx.x; j < n4; j += 256)
    reinterpret_cast<float4*>(sh_buf)[j] = reinterpret_cast<const float4*>(gsrc)[j];
  if (4 * n4 + (int)threadIdx.x < n) sh_buf[4 * n4 + threadIdx.x] = gsrc[4 * n4 + threadIdx.x];
  __syncthreads();
  sh_leaf_step(a, 1, sh_buf, a.fdc + row0 * 3, a.m[1] + row0 * 3, a.v[1] + row0 * 3, nrows * 3, 3, C, 0);
  sh_leaf_step(a, 2, sh_buf, a.frest + row0 * W, a.m[2] + row0 * W, a.v[2] + row0 * W, nrows * W, W, C, 3);
  if (!a.a_shs) return;  // uniform
  __syncthreads();
  float* dst = a.a_shs + row0 * C;
  for (int j = threadIdx.x; j < n4; j += 256)
    reinterpret_cast<float4*>(dst)[j] = reinterpret_cast<const float4*>(sh_buf)[j];
  if (4 * n4 + (int)threadIdx.x < n) dst[4 * n4 + threadIdx.x] = sh_buf[4 * n4 + threadIdx.x];
}

hipError_t launch_model_step(int P, int M, float* const* params, float* const* exp_avg, float* const* exp_avg_sq,
                             const float* g_xyz, const float* g_scales, const float* g_rot, const float* g_opac,
                             const float* g_shs, float* a_scales, float* a_rot, float* a_opac, float* a_shs,
                             const float* lr, double beta1, double beta2, double eps, int step, hipStream_t s) {
  StepArgs a;
  a.P = P; a.M = M;
  a.xyz = params[0]; a.fdc = params[1]; a.frest = params[2]; a.scaling = params[3]; a.rot = params[4]; a.opac = params[5];
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  for (int k = 0; k < 6; k++) {
    a.m[k] = exp_avg[k];
    a.v[k] = exp_avg_sq[k];
    a.step_size[k] = (float)((double)lr[k] / bc1);
  }
  a.g_xyz = g_xyz; a.g_scales = g_scales; a.g_rot = g_rot; a.g_opac = g_opac; a.g_shs = g_shs;
  a.a_scales = a_scales; a.a_rot = a_rot; a.a_opac = a_opac; a.a_shs = a_shs;
  a.beta1 = (float)beta1; a.beta2 = (float)beta2;
  a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
  a.eps = (float)eps;
  a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  const unsigned long long P64 = (unsigned long long)P;
  const unsigned long long ne[6] = {3 * P64, 3 * P64, 3 * (unsigned long long)(M - 1) * P64, 3 * P64, 4 * P64, P64};
  const void* extra[6][2] = {{g_xyz, nullptr}, {g_shs, a_shs}, {nullptr, nullptr}, {g_scales, a_scales},
                             {g_rot, a_rot}, {g_opac, a_opac}};
  unsigned long long cum = 0;
  a.aligned = 0;
  for (int k = 0; k < 6; k++) {
    const uintptr_t bits = (uintptr_t)params[k] | (uintptr_t)exp_avg[k] | (uintptr_t)exp_avg_sq[k] |
                           (uintptr_t)extra[k][0] | (uintptr_t)extra[k][1];
    if ((bits & 15u) == 0) a.aligned |= 1u << k;
  }
  // M > 1: the SH leaves go through k_sh_step (LDS-staged rows) when everything it touches is 16-byte aligned
  const size_t sh_lds = (size_t)SH_ROWS * 3 * (size_t)M * sizeof(float);
  const bool sh_staged = M > 1 && P > 0 && ((a.aligned >> 1) & 3u) == 3u && sh_lds <= 64 * 1024;
  for (int k = 0; k < 6; k++) {
    a.numel[k] = (sh_staged && (k == 1 || k == 2)) ? 0ull : ne[k];
    cum += (a.numel[k] + 3) / 4;
    a.end[k] = cum;
  }
  ProfScope ps(K_MODEL_STEP, s);
  hipLaunchKernelGGL(k_model_step, dim3((unsigned)((cum + 255) / 256)), dim3(256), 0, s, a);
  if (sh_staged) hipLaunchKernelGGL(k_sh_step, dim3((unsigned)((P + SH_ROWS - 1) / SH_ROWS)), dim3(256), sh_lds, s, a);
  return hipGetLastError();
}

hipError_t launch_activate(int P, int M, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                           const float* f_dc, const float* f_rest, float* scales, float* rotations, float* opacities,
                           float* shs, hipStream_t s) {
  ProfScope ps(K_ACTIVATE, s);
  hipLaunchKernelGGL(k_activate, dim3((P + 255) / 256), dim3(256), 0, s, P, M, scaling_raw, rotation_raw, opacity_raw,
                     f_dc, f_rest, scales, rotations, opacities, shs);
  if (M > 1 && P > 0) launch_sh_move<false>(P, M, shs, const_cast<float*>(f_dc), const_cast<float*>(f_rest), s);
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
  if (M > 1 && P > 0) launch_sh_move<true>(P, M, const_cast<float*>(g_shs), g_f_dc, g_f_rest, s);
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
