// growth.hip -- "next" row SURVEY.md section 8(f) #4: the data formats either side of the hot path.
//   k_init_gaussians   new map points -> the six leaf parameter rows, written IN PLACE at the tail of the caller's
//                      capacity buffers (replaces the tensor algebra of GaussianModel::addNewPointcloud,
//                      src/gs/gaussian.cu:241-313, whose results then go through six torch::cat + twelve
//                      optimiser-state torch::cat of the WHOLE model, :451-472, 524-540)
//   k_pack_ply_rows    the six leaf tensors -> the interleaved little-endian f32 rows of the reference's PLY
//                      export (construct_list_of_attributes :474-492, Save_ply :494-522): one coalesced pass on
//                      the device and ONE device-to-host copy instead of seven .cpu() copies + a host interleave.
// gfx950 only; both kernels are pure streaming (HBM-bound).
#include "gsr_internal.hpp"

namespace gsr {

// RGB2SH (include/gs/gs/sh_utils.cuh:61-63) with C0 = 0.28209479177387814 narrowed to f32, as the reference does
constexpr float SH_C0_F = 0.28209479177387814f;

__global__ __launch_bounds__(256) void k_init_gaussians(const int n, const int M, const float* __restrict__ xyz,
                                                        const float* __restrict__ covs,
                                                        const float* __restrict__ rgbs, const float scale_factor,
                                                        float* __restrict__ xyz_out, float* __restrict__ fdc_out,
                                                        float* __restrict__ frest_out, float* __restrict__ scaling_out,
                                                        float* __restrict__ rotation_out,
                                                        float* __restrict__ opacity_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    xyz_out[3 * i + k] = xyz[3 * i + k];
    // decomposeSR keeps the DIAGONAL of the 3x3 covariance (gaussian.cu:10-11);
    // _scaling = log(sqrt(diag * scale_factor)) (:280), rotation = identity quaternion (:282-283)
    scaling_out[3 * i + k] = logf(sqrtf(covs[9 * i + 4 * k] * scale_factor));
    fdc_out[3 * i + k] = (rgbs[3 * i + k] / 255.0f - 0.5f) / SH_C0_F;  // RGB2SH(rgb / 255), :289
  }
  rotation_out[4 * i + 0] = 1.0f;
  rotation_out[4 * i + 1] = 0.0f;
  rotation_out[4 * i + 2] = 0.0f;
  rotation_out[4 * i + 3] = 0.0f;
  opacity_out[i] = 0.0f;  // inverse_sigmoid(0.5) = log(0.5 / 0.5), :286
  for (int k = 0; k < 3 * (M - 1); k++) frest_out[(size_t)3 * (M - 1) * i + k] = 0.0f;  // :292-297
}

// One thread per output float; row = [x y z | nx ny nz (zeros) | f_dc (channel-major) | f_rest (channel-major) |
// opacity | scale | rot].  "channel-major" = the reference's _features.transpose(1, 2).flatten(1):
// f_dc_c = features_dc[p][0][c], f_rest_{c*(M-1)+k} = features_rest[p][k][c].
__global__ __launch_bounds__(256) void k_pack_ply_rows(const size_t total, const int M, const float* __restrict__ xyz,
                                                       const float* __restrict__ fdc, const float* __restrict__ frest,
                                                       const float* __restrict__ opacity,
                                                       const float* __restrict__ scaling,
                                                       const float* __restrict__ rotation, float* __restrict__ rows) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int RF = 14 + 3 * M, R1 = 3 * (M - 1);
  const size_t p = e / RF;
  const int c = (int)(e - p * RF);
  float v;
  if (c < 3) v = xyz[3 * p + c];
  else if (c < 6) v = 0.0f;
  else if (c < 9) v = fdc[3 * p + (c - 6)];
  else if (c < 9 + R1) {
    const int q = c - 9, ch = q / (M - 1), k = q - ch * (M - 1);
    v = frest[(size_t)R1 * p + 3 * k + ch];
  } else if (c == 9 + R1) v = opacity[p];
  else if (c < 13 + R1) v = scaling[3 * p + (c - 10 - R1)];
  else v = rotation[4 * p + (c - 13 - R1)];
  rows[e] = v;
}

hipError_t launch_init_gaussians(int n, int M, const float* xyz, const float* covs, const float* rgbs, float scale_factor,
                                 float* xyz_out, float* fdc_out, float* frest_out, float* scaling_out,
                                 float* rotation_out, float* opacity_out, hipStream_t s) {
  ProfScope ps(K_INIT_GAUSSIANS, s);
  hipLaunchKernelGGL(k_init_gaussians, dim3((n + 255) / 256), dim3(256), 0, s, n, M, xyz, covs, rgbs, scale_factor,
                     xyz_out, fdc_out, frest_out, scaling_out, rotation_out, opacity_out);
  return hipGetLastError();
}

hipError_t launch_pack_ply_rows(int P, int M, const float* xyz, const float* fdc, const float* frest,
                                const float* opacity, const float* scaling, const float* rotation, float* rows,
                                hipStream_t s) {
  const size_t total = (size_t)P * (14 + 3 * M);
  ProfScope ps(K_PACK_PLY, s);
  hipLaunchKernelGGL(k_pack_ply_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, total, M, xyz, fdc, frest,
                     opacity, scaling, rotation, rows);
  return hipGetLastError();
}

}  // namespace gsr
