// loss.hip -- the photometric loss that follows every render in GS-LIVM's optimiser, fused
// ("next" row 2):
//
//   L = (1 - lambda) * mean|img - gt|  +  lambda * (1 - mean(SSIM(img, gt)))
//
// Reference: gaussian_splatting::l1_loss and ::ssim (include/gs/gs/loss_utils.cuh:11-13, 43-70), combined at
// src/liw/lioOptimization.cpp:1705-1710 with lambda = lambda_dssim (config/basic_common.yaml:63).  The
// reference runs five grouped 11x11 conv2d + ~15 elementwise kernels forward and their autograd backward per
// view; here:
//   k_loss_forward   one pass: separable 11-tap window over (x, y, x^2, y^2, xy) in LDS, SSIM map, the three
//                    per-pixel derivative maps the backward needs, per-workgroup partial sums of |x-y| and SSIM
//   k_loss_finalize  fixed-order sum of the partials -> {loss, l1, ssim}   (deterministic: no float atomics)
//   k_loss_backward  dL/dimg = (1-lambda)/N * sign(x-y) - lambda/N * [convT(A) + 2x convT(B) + y convT(C)]
// The window is whatever 1-D kernel the caller passes (the 2-D window of the reference is its outer product,
// loss_utils.cuh:33-37) -- including the reference's own, which is NOT symmetric (gaussian() floors
// (x - window_size)/2, loss_utils.cuh:24-31), so correlation (forward) and its transpose (backward) are kept
// apart.  Zero padding of window/2, as conv2d(padding = window_size / 2).
#include "gsr_internal.hpp"

namespace gsr {

constexpr int LW_ = 11;          // window taps
constexpr int LR_ = LW_ / 2;     // halo
constexpr int LT_ = 16;          // output tile edge
constexpr int LH_ = LT_ + 2 * LR_;  // 26: input tile edge
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;  // loss_utils.cuh:8-9

struct LossWindow { float w[LW_]; };

// mu(q) = sum_k w[k] * f(q + k - 5)   (cross-correlation, zero padded): what conv2d computes
__global__ __launch_bounds__(256) void k_loss_forward(const int C, const int H, const int W,
                                                      const float* __restrict__ img, const float* __restrict__ gt,
                                                      const LossWindow win, float* __restrict__ mapA,
                                                      float* __restrict__ mapB, float* __restrict__ mapC,
                                                      float* __restrict__ partials) {
  __shared__ float sx[LH_][LH_ + 1], sy[LH_][LH_ + 1];
  __shared__ float h[5][LH_][LT_ + 1];  // horizontally filtered rows
  __shared__ float red[2][4];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.z;
  const int x0 = blockIdx.x * LT_, y0 = blockIdx.y * LT_;
  const float* X = img + (size_t)c * H * W;
  const float* Y = gt + (size_t)c * H * W;
  for (int i = threadIdx.x; i < LH_ * LH_; i += 256) {
    const int r = i / LH_, cc = i % LH_;
    const int gy = y0 + r - LR_, gx = x0 + cc - LR_;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    sx[r][cc] = in ? X[(size_t)gy * W + gx] : 0.f;
    sy[r][cc] = in ? Y[(size_t)gy * W + gx] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < LH_ * LT_; i += 256) {
    const int r = i / LT_, cc = i % LT_;
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll
    for (int k = 0; k < LW_; k++) {
      const float xv = sx[r][cc + k], yv = sy[r][cc + k], wk = win.w[k];
      a0 += wk * xv; a1 += wk * yv; a2 += wk * xv * xv; a3 += wk * yv * yv; a4 += wk * xv * yv;
    }
    h[0][r][cc] = a0; h[1][r][cc] = a1; h[2][r][cc] = a2; h[3][r][cc] = a3; h[4][r][cc] = a4;
  }
  __syncthreads();
  float l1 = 0.f, ss = 0.f;
  const int gx = x0 + tx, gy = y0 + ty;
  if (gx < W && gy < H) {
    float mu1 = 0, mu2 = 0, e11 = 0, e22 = 0, e12 = 0;
#pragma unroll
    for (int k = 0; k < LW_; k++) {
      const float wk = win.w[k];
      mu1 += wk * h[0][ty + k][tx]; mu2 += wk * h[1][ty + k][tx]; e11 += wk * h[2][ty + k][tx];
      e22 += wk * h[3][ty + k][tx]; e12 += wk * h[4][ty + k][tx];
    }
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
    const float n1 = 2.f * mu12 + SSIM_C1, n2 = 2.f * s12 + SSIM_C2;
    const float d1 = mu1_sq + mu2_sq + SSIM_C1, d2 = s1 + s2 + SSIM_C2;
    const float inv = 1.0f / (d1 * d2);
    const float s = n1 * n2 * inv;
    // partials of s w.r.t. (mu1, sigma1_sq, sigma12) at fixed img2
    const float ds_dmu1 = (2.f * mu2 * n2 * d1 - 2.f * mu1 * n1 * n2) * inv / d1;  // d/dmu1 of n1/d1 times n2/d2
    const float ds_ds1 = -s / d2;
    const float ds_ds12 = 2.f * n1 * inv;
    // total derivative through sigma1_sq = E[x^2] - mu1^2 and sigma12 = E[xy] - mu1*mu2:
    const size_t o = ((size_t)c * H + gy) * W + gx;
    mapA[o] = ds_dmu1 - 2.f * mu1 * ds_ds1 - mu2 * ds_ds12;
    mapB[o] = ds_ds1;
    mapC[o] = ds_ds12;
    ss = s;
    l1 = fabsf(sx[ty + LR_][tx + LR_] - sy[ty + LR_][tx + LR_]);
  }
  // fixed-order workgroup reduction -> one partial pair per workgroup
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    l1 += __shfl_xor(l1, o, 64);
    ss += __shfl_xor(ss, o, 64);
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = l1; red[1][threadIdx.x >> 6] = ss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    partials[2 * b] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partials[2 * b + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

__global__ __launch_bounds__(1024) void k_loss_finalize(const float* __restrict__ partials, const int nblocks,
                                                        const float inv_n, const float lambda,
                                                        float* __restrict__ out3) {
  // one workgroup, fixed association order (thread-strided partial sums in f64, then a tree): deterministic
  __shared__ double r1[1024], r2[1024];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 1024) {
    const float2 v = reinterpret_cast<const float2*>(partials)[i];
    a += v.x; b += v.y;
  }
  r1[threadIdx.x] = a; r2[threadIdx.x] = b;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float l1 = (float)(r1[0] * inv_n), ssim = (float)(r2[0] * inv_n);
    out3[0] = (1.f - lambda) * l1 + lambda * (1.f - ssim);
    out3[1] = l1;
    out3[2] = ssim;
  }
}

// dL/dx(p) = (1-lambda)/N sign(x-y) - lambda/N * sum_q w(p - q + 5) * [A(q) + 2 x(p) B(q) + y(p) C(q)]
// i.e. the transposed correlation: taps are read flipped.
__global__ __launch_bounds__(256) void k_loss_backward(const int C, const int H, const int W,
                                                       const float* __restrict__ img, const float* __restrict__ gt,
                                                       const LossWindow win, const float* __restrict__ mapA,
                                                       const float* __restrict__ mapB, const float* __restrict__ mapC,
                                                       const float inv_n, const float lambda,
                                                       float* __restrict__ dL_dimg) {
  __shared__ float s[3][LH_][LH_ + 1];
  __shared__ float h[3][LH_][LT_ + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.z;
  const int x0 = blockIdx.x * LT_, y0 = blockIdx.y * LT_;
  const size_t plane = (size_t)c * H * W;
  for (int i = threadIdx.x; i < LH_ * LH_; i += 256) {
    const int r = i / LH_, cc = i % LH_;
    const int gy = y0 + r - LR_, gx = x0 + cc - LR_;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const size_t o = plane + (size_t)gy * W + gx;
    s[0][r][cc] = in ? mapA[o] : 0.f;
    s[1][r][cc] = in ? mapB[o] : 0.f;
    s[2][r][cc] = in ? mapC[o] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < LH_ * LT_; i += 256) {
    const int r = i / LT_, cc = i % LT_;
    float a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
    for (int k = 0; k < LW_; k++) {
      const float wk = win.w[LW_ - 1 - k];  // flipped
      a0 += wk * s[0][r][cc + k]; a1 += wk * s[1][r][cc + k]; a2 += wk * s[2][r][cc + k];
    }
    h[0][r][cc] = a0; h[1][r][cc] = a1; h[2][r][cc] = a2;
  }
  __syncthreads();
  const int gx = x0 + tx, gy = y0 + ty;
  if (gx < W && gy < H) {
    float tA = 0, tB = 0, tC = 0;
#pragma unroll
    for (int k = 0; k < LW_; k++) {
      const float wk = win.w[LW_ - 1 - k];
      tA += wk * h[0][ty + k][tx]; tB += wk * h[1][ty + k][tx]; tC += wk * h[2][ty + k][tx];
    }
    const size_t o = plane + (size_t)gy * W + gx;
    const float x = img[o], y = gt[o];
    const float d = x - y;
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);  // torch: abs'(0) = 0
    dL_dimg[o] = (1.f - lambda) * inv_n * sgn - lambda * inv_n * (tA + 2.f * x * tB + y * tC);
  }
}

size_t loss_workspace_bytes(int C, int H, int W) {
  const size_t maps = 3 * align_up((size_t)C * H * W * sizeof(float));
  const size_t nblocks = (size_t)((W + LT_ - 1) / LT_) * ((H + LT_ - 1) / LT_) * C;
  return maps + align_up(nblocks * 2 * sizeof(float)) + ALIGN;
}

hipError_t launch_photometric_loss(int C, int H, int W, const float* img, const float* gt, const float* window11,
                                   float lambda, float* loss_out3, float* dL_dimg, char* workspace, hipStream_t s) {
  Carver cv(workspace);
  const size_t n = (size_t)C * H * W;
  float* mapA = cv.take<float>(n);
  float* mapB = cv.take<float>(n);
  float* mapC = cv.take<float>(n);
  const dim3 grid((W + LT_ - 1) / LT_, (H + LT_ - 1) / LT_, C);
  const int nblocks = (int)(grid.x * grid.y * grid.z);
  float* partials = cv.take<float>((size_t)nblocks * 2);
  LossWindow win;
  for (int k = 0; k < LW_; k++) win.w[k] = window11[k];
  const float inv_n = (float)(1.0 / (double)n);
  {
    ProfScope ps(K_LOSS_FWD, s);
    hipLaunchKernelGGL(k_loss_forward, grid, dim3(256), 0, s, C, H, W, img, gt, win, mapA, mapB, mapC, partials);
  }
  {
    ProfScope ps(K_LOSS_FINALIZE, s);
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(1024), 0, s, partials, nblocks, inv_n, lambda, loss_out3);
  }
  if (dL_dimg) {
    ProfScope ps(K_LOSS_BWD, s);
    hipLaunchKernelGGL(k_loss_backward, grid, dim3(256), 0, s, C, H, W, img, gt, win, mapA, mapB, mapC, inv_n, lambda,
                       dL_dimg);
  }
  return hipGetLastError();
}

}  // namespace gsr
