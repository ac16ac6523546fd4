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
constexpr int LTX_ = 32, LTY_ = 16;                      // output tile of one 256-thread workgroup
constexpr int LHX_ = LTX_ + 2 * LR_, LHY_ = LTY_ + 2 * LR_;  // 42 x 26 input tile
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;  // loss_utils.cuh:8-9

struct LossWindow { float w[LW_]; };

// Both passes are register-blocked: a thread of the horizontal pass produces 4 adjacent outputs of one row from
// 14 staged inputs (instead of 4 x 11 reads), a thread of the vertical pass 2 vertically adjacent outputs from
// 12 filtered values per moment -- the window is walked in the same tap order as a plain 11-tap sum, so the
// arithmetic is unchanged while LDS traffic per pixel drops ~2.5x and the halo (42 x 26 for 32 x 16) ~1.25x.

// mu(q) = sum_k w[k] * f(q + k - 5)   (cross-correlation, zero padded): what conv2d computes
__global__ __launch_bounds__(256) void k_loss_forward(const int C, const int H, const int W,
                                                      const float* __restrict__ img, const float* __restrict__ gt,
                                                      const LossWindow win, float* __restrict__ mapA,
                                                      float* __restrict__ mapB, float* __restrict__ mapC,
                                                      float* __restrict__ partials) {
  __shared__ float sx[LHY_][LHX_ + 1], sy[LHY_][LHX_ + 1];
  __shared__ float h[5][LHY_][LTX_ + 1];  // horizontally filtered rows
  __shared__ float red[2][4];
  const int c = blockIdx.z;
  const int x0 = blockIdx.x * LTX_, y0 = blockIdx.y * LTY_;
  const float* X = img + (size_t)c * H * W;
  const float* Y = gt + (size_t)c * H * W;
  {  // all of a thread's halo loads are issued before the first one is consumed (one round trip, not five)
    constexpr int NI = (LHY_ * LHX_ + 255) / 256;
    float vx[NI], vy[NI];
#pragma unroll
    for (int it = 0; it < NI; it++) {
      const int i = threadIdx.x + 256 * it, r = i / LHX_, cc = i % LHX_;
      const int gy = y0 + r - LR_, gx = x0 + cc - LR_;
      const bool in = i < LHY_ * LHX_ && gy >= 0 && gy < H && gx >= 0 && gx < W;
      vx[it] = in ? X[(size_t)gy * W + gx] : 0.f;
      vy[it] = in ? Y[(size_t)gy * W + gx] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < NI; it++) {
      const int i = threadIdx.x + 256 * it, r = i / LHX_, cc = i % LHX_;
      if (i < LHY_ * LHX_) { sx[r][cc] = vx[it]; sy[r][cc] = vy[it]; }
    }
  }
  __syncthreads();
  if (threadIdx.x < LHY_ * (LTX_ / 4)) {  // 26 rows x 8 groups of 4 outputs
    const int r = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 4;
    float xv[LW_ + 3], yv[LW_ + 3];
#pragma unroll
    for (int k = 0; k < LW_ + 3; k++) { xv[k] = sx[r][c0 + k]; yv[k] = sy[r][c0 + k]; }
#pragma unroll
    for (int o = 0; o < 4; o++) {
      float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll
      for (int k = 0; k < LW_; k++) {
        const float xq = xv[o + k], yq = yv[o + k], wk = win.w[k];
        a0 += wk * xq; a1 += wk * yq; a2 += wk * xq * xq; a3 += wk * yq * yq; a4 += wk * xq * yq;
      }
      h[0][r][c0 + o] = a0; h[1][r][c0 + o] = a1; h[2][r][c0 + o] = a2; h[3][r][c0 + o] = a3; h[4][r][c0 + o] = a4;
    }
  }
  __syncthreads();
  float l1 = 0.f, ss = 0.f;
  const int tx = threadIdx.x & 31, ty0 = (threadIdx.x >> 5) * 2;  // two vertically adjacent outputs
  float m[5][2];
#pragma unroll
  for (int q = 0; q < 5; q++) {
    float col[LW_ + 1];
#pragma unroll
    for (int k = 0; k < LW_ + 1; k++) col[k] = h[q][ty0 + k][tx];
    float o0 = 0, o1 = 0;
#pragma unroll
    for (int k = 0; k < LW_; k++) { o0 += win.w[k] * col[k]; o1 += win.w[k] * col[k + 1]; }
    m[q][0] = o0; m[q][1] = o1;
  }
#pragma unroll
  for (int o = 0; o < 2; o++) {
    const int gx = x0 + tx, gy = y0 + ty0 + o;
    if (gx < W && gy < H) {
      const float mu1 = m[0][o], mu2 = m[1][o], e11 = m[2][o], e22 = m[3][o], e12 = m[4][o];
      const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
      const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
      const float n1 = 2.f * mu12 + SSIM_C1, n2 = 2.f * s12 + SSIM_C2;
      const float d1 = mu1_sq + mu2_sq + SSIM_C1, d2 = s1 + s2 + SSIM_C2;
      const float inv = 1.0f / (d1 * d2);
      const float sv = n1 * n2 * inv;
      // partials of s w.r.t. (mu1, sigma1_sq, sigma12) at fixed img2
      const float ds_dmu1 = (2.f * mu2 * n2 * d1 - 2.f * mu1 * n1 * n2) * inv / d1;  // d/dmu1 of n1/d1 times n2/d2
      const float ds_ds1 = -sv / d2;
      const float ds_ds12 = 2.f * n1 * inv;
      // total derivative through sigma1_sq = E[x^2] - mu1^2 and sigma12 = E[xy] - mu1*mu2:
      const size_t oidx = ((size_t)c * H + gy) * W + gx;
      mapA[oidx] = ds_dmu1 - 2.f * mu1 * ds_ds1 - mu2 * ds_ds12;
      mapB[oidx] = ds_ds1;
      mapC[oidx] = ds_ds12;
      ss += sv;
      l1 += fabsf(sx[ty0 + o + LR_][tx + LR_] - sy[ty0 + o + LR_][tx + LR_]);
    }
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
  __shared__ float s[3][LHY_][LHX_ + 1];
  __shared__ float h[3][LHY_][LTX_ + 1];
  const int c = blockIdx.z;
  const int x0 = blockIdx.x * LTX_, y0 = blockIdx.y * LTY_;
  const size_t plane = (size_t)c * H * W;
  const int tx = threadIdx.x & 31, ty0 = (threadIdx.x >> 5) * 2;
  float px[2], py[2];  // this thread's two output pixels of img / gt, requested together with the halo
  {
    constexpr int NI = (LHY_ * LHX_ + 255) / 256;
    float va[NI], vb[NI], vc[NI];
#pragma unroll
    for (int it = 0; it < NI; it++) {
      const int i = threadIdx.x + 256 * it, r = i / LHX_, cc = i % LHX_;
      const int gy = y0 + r - LR_, gx = x0 + cc - LR_;
      const bool in = i < LHY_ * LHX_ && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const size_t o = plane + (size_t)gy * W + gx;
      va[it] = in ? mapA[o] : 0.f;
      vb[it] = in ? mapB[o] : 0.f;
      vc[it] = in ? mapC[o] : 0.f;
    }
#pragma unroll
    for (int o = 0; o < 2; o++) {
      const int gx = x0 + tx, gy = y0 + ty0 + o;
      const bool in = gx < W && gy < H;
      px[o] = in ? img[plane + (size_t)gy * W + gx] : 0.f;
      py[o] = in ? gt[plane + (size_t)gy * W + gx] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < NI; it++) {
      const int i = threadIdx.x + 256 * it, r = i / LHX_, cc = i % LHX_;
      if (i < LHY_ * LHX_) { s[0][r][cc] = va[it]; s[1][r][cc] = vb[it]; s[2][r][cc] = vc[it]; }
    }
  }
  __syncthreads();
  if (threadIdx.x < LHY_ * (LTX_ / 4)) {  // register-blocked as the forward: 4 adjacent outputs per thread
    const int r = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 4;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      float v[LW_ + 3];
#pragma unroll
      for (int k = 0; k < LW_ + 3; k++) v[k] = s[q][r][c0 + k];
#pragma unroll
      for (int o = 0; o < 4; o++) {
        float a = 0;
#pragma unroll
        for (int k = 0; k < LW_; k++) a += win.w[LW_ - 1 - k] * v[o + k];  // flipped taps
        h[q][r][c0 + o] = a;
      }
    }
  }
  __syncthreads();
  float t[3][2];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    float col[LW_ + 1];
#pragma unroll
    for (int k = 0; k < LW_ + 1; k++) col[k] = h[q][ty0 + k][tx];
    float o0 = 0, o1 = 0;
#pragma unroll
    for (int k = 0; k < LW_; k++) { o0 += win.w[LW_ - 1 - k] * col[k]; o1 += win.w[LW_ - 1 - k] * col[k + 1]; }
    t[q][0] = o0; t[q][1] = o1;
  }
#pragma unroll
  for (int o = 0; o < 2; o++) {
    const int gx = x0 + tx, gy = y0 + ty0 + o;
    if (gx < W && gy < H) {
      const size_t oi = plane + (size_t)gy * W + gx;
      const float x = px[o], y = py[o];
      const float d = x - y;
      const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);  // torch: abs'(0) = 0
      dL_dimg[oi] = (1.f - lambda) * inv_n * sgn - lambda * inv_n * (t[0][o] + 2.f * x * t[1][o] + y * t[2][o]);
    }
  }
}

size_t loss_workspace_bytes(int C, int H, int W) {
  const size_t maps = 3 * align_up((size_t)C * H * W * sizeof(float));
  const size_t nblocks = (size_t)((W + LTX_ - 1) / LTX_) * ((H + LTY_ - 1) / LTY_) * C;
  return maps + align_up(nblocks * 2 * sizeof(float)) + ALIGN;
}

hipError_t launch_photometric_loss(int C, int H, int W, const float* img, const float* gt, const float* window11,
                                   float lambda, float* loss_out3, float* dL_dimg, char* workspace, hipStream_t s) {
  Carver cv(workspace);
  const size_t n = (size_t)C * H * W;
  float* mapA = cv.take<float>(n);
  float* mapB = cv.take<float>(n);
  float* mapC = cv.take<float>(n);
  const dim3 grid((W + LTX_ - 1) / LTX_, (H + LTY_ - 1) / LTY_, C);
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
