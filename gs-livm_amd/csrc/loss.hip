// loss.hip -- the photometric loss that follows every render in GS-LIVM's optimiser, fused
// ("next" row 2):
//
//   L = (1 - lambda) * mean|img - gt|  +  lambda * (1 - mean(SSIM(img, gt)))
//
// Reference: gaussian_splatting::l1_loss and ::ssim (include/gs/gs/loss_utils.cuh:11-13, 43-70), combined at
// src/liw/lioOptimization.cpp:1705-1710 with lambda = lambda_dssim (config/basic_common.yaml:63).  The
// reference runs five grouped 11x11 conv2d + ~15 elementwise kernels forward and their autograd backward per
// view; here:
//   k_loss_forward   one pass: separable 11-tap window over (x, y, x^2 + y^2, xy) -- vertical pass in registers
//                    straight from coalesced loads, horizontal pass through LDS --, SSIM map, the three per-pixel
//                    derivative maps the backward needs, per-workgroup partial sums of |x-y| and SSIM
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
// Work unit of one 256-thread workgroup: 54 x 32 outputs.  Wave w owns the 8 output rows 8w .. 8w+7, lane l the input
// column x0 - 5 + l (64 columns = 54 outputs + the halo), so every global load of the vertical pass is one coalesced
// 256-byte row segment and the pass runs in registers straight from those loads (18 rows -> 8 outputs per moment, taps
// in the order of a plain 11-tap sum); its results cross the workgroup's ONE barrier through LDS and the horizontal pass
// is register-blocked seven outputs to a thread (17 staged values per moment).
constexpr int LTX_ = 54, LTY_ = 32;
constexpr int LSEG_ = 8;                    // output rows per wave
constexpr int LIN_ = LSEG_ + 2 * LR_;       // 18 input rows per wave
constexpr int LHO_ = 7;                     // horizontal pass: outputs per thread (8 threads x 7 >= 54)
constexpr int LFLAT_ = (LTX_ * LTY_ + 255) / 256;  // 7 elements per thread in the tile's flattened order
constexpr int LSTRIDE_ = 72;                // LDS row stride: (8 r + 7 g + k) mod 64 is conflict-free over a wave's 8 x 8
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;  // loss_utils.cuh:8-9

struct LossWindow { float w[LW_]; };

// mu(q) = sum_k w[k] * f(q + k - 5)   (cross-correlation, zero padded): what conv2d computes.  SSIM needs the two
// variances only through their sum (d2 = sigma1^2 + sigma2^2 + C2), so four windowed moments are carried, not five:
// E[x], E[y], E[x^2 + y^2], E[xy].
__global__ __launch_bounds__(256) void k_loss_forward(const int C, const int H, const int W,
                                                      const float* __restrict__ img, const float* __restrict__ gt,
                                                      const LossWindow win, float* __restrict__ mapA,
                                                      float* __restrict__ mapB, float* __restrict__ mapC,
                                                      float* __restrict__ partials) {
  __shared__ float hv[4][LTY_][LSTRIDE_];  // vertically filtered moments
  __shared__ float red[2][4];
  const int c = blockIdx.z;
  const int x0 = blockIdx.x * LTX_, y0 = blockIdx.y * LTY_;
  const size_t plane = (size_t)c * H * W;
  const float* X = img + plane;
  const float* Y = gt + plane;
  const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
  float l1 = 0.f, ss = 0.f;
  {
    const int gx = x0 - LR_ + lane;
    const bool col_in = gx >= 0 && gx < W;
    const int gxc = min(max(gx, 0), W - 1);
    float vx[LIN_], vy[LIN_];
#pragma unroll
    for (int i = 0; i < LIN_; i++) {  // all 36 loads are in flight before the first is consumed: clamped addresses,
      const int gy = y0 + LSEG_ * seg - LR_ + i;  // unconditional loads, zero padding by select (no branch per load)
      const int off = min(max(gy, 0), H - 1) * W + gxc;
      const float a = X[off], b = Y[off];
      const bool in = col_in && gy >= 0 && gy < H;
      vx[i] = in ? a : 0.f;
      vy[i] = in ? b : 0.f;
    }
    float acc[LSEG_][4];
#pragma unroll
    for (int o = 0; o < LSEG_; o++) acc[o][0] = acc[o][1] = acc[o][2] = acc[o][3] = 0.f;
#pragma unroll
    for (int i = 0; i < LIN_; i++) {
      const float x = vx[i], y = vy[i];
      const float p2 = __builtin_fmaf(y, y, x * x), p3 = x * y;
#pragma unroll
      for (int o = 0; o < LSEG_; o++) {
        const int k = i - o;
        if (k >= 0 && k < LW_) {
          const float wk = win.w[k];
          acc[o][0] += wk * x; acc[o][1] += wk * y; acc[o][2] += wk * p2; acc[o][3] += wk * p3;
        }
      }
    }
#pragma unroll
    for (int o = 0; o < LSEG_; o++) {
#pragma unroll
      for (int q = 0; q < 4; q++) hv[q][LSEG_ * seg + o][lane] = acc[o][q];
    }
    if (lane >= LR_ && lane < LR_ + LTX_ && col_in) {  // the L1 term of this thread's own eight output pixels
#pragma unroll
      for (int o = 0; o < LSEG_; o++)
        if (y0 + LSEG_ * seg + o < H) l1 += fabsf(vx[o + LR_] - vy[o + LR_]);
    }
  }
  __syncthreads();
  {
    const int r = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * LHO_;
    float m[4][LHO_];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      float v[LHO_ + LW_ - 1];
#pragma unroll
      for (int k = 0; k < LHO_ + LW_ - 1; k++) v[k] = hv[q][r][c0 + k];  // (the last group reads into the row padding)
#pragma unroll
      for (int o = 0; o < LHO_; o++) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < LW_; k++) a += win.w[k] * v[o + k];
        m[q][o] = a;
      }
    }
    const int gy = y0 + r;
    float oa[LHO_], ob[LHO_], oc[LHO_];
#pragma unroll
    for (int o = 0; o < LHO_; o++) {
      const int gx = x0 + c0 + o;
      oa[o] = ob[o] = oc[o] = 0.f;
      if (c0 + o < LTX_ && gx < W && gy < H) {
        const float mu1 = m[0][o], mu2 = m[1][o], e_sum = m[2][o], e12 = m[3][o];
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float s12 = e12 - mu12;
        const float n1 = 2.f * mu12 + SSIM_C1, n2 = 2.f * s12 + SSIM_C2;
        const float d1 = mu1_sq + mu2_sq + SSIM_C1, d2 = (e_sum - mu1_sq - mu2_sq) + SSIM_C2;
        const float r1 = __builtin_amdgcn_rcpf(d1), r2 = __builtin_amdgcn_rcpf(d2);  // (1 ulp; d1, d2 >= C1, C2 > 0 up to rounding)
        const float inv = r1 * r2;
        const float sv = n1 * n2 * inv;
        // partials of s w.r.t. (mu1, sigma1_sq, sigma12) at fixed img2
        const float ds_dmu1 = 2.f * n2 * (mu2 * d1 - mu1 * n1) * inv * r1;  // d/dmu1 of n1/d1 times n2/d2
        const float ds_ds1 = -sv * r2;
        const float ds_ds12 = 2.f * n1 * inv;
        // total derivative through sigma1_sq = E[x^2] - mu1^2 and sigma12 = E[xy] - mu1*mu2:
        oa[o] = ds_dmu1 - 2.f * mu1 * ds_ds1 - mu2 * ds_ds12;
        ob[o] = ds_ds1;
        oc[o] = ds_ds12;
        ss += sv;
      }
    }
    // A thread's seven outputs are adjacent in a row: stored from here, a wave's store would be 64 separate 4-byte
    // accesses (measured: the kernel was bound by them, not by its arithmetic).  The maps change hands through LDS
    // and leave in the flattened order of the tile, 54-float row segments to consecutive lanes.
    __syncthreads();  // (everyone has read hv)
#pragma unroll
    for (int o = 0; o < LHO_; o++) { hv[0][r][c0 + o] = oa[o]; hv[1][r][c0 + o] = ob[o]; hv[2][r][c0 + o] = oc[o]; }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < LFLAT_; it++) {
    const int e = it * 256 + (int)threadIdx.x;
    const int row = e / LTX_, col = e - row * LTX_;
    const int gx = x0 + col, gy = y0 + row;
    if (e < LTX_ * LTY_ && gx < W && gy < H) {
      const size_t oidx = plane + (size_t)gy * W + gx;
      mapA[oidx] = hv[0][row][col];
      mapB[oidx] = hv[1][row][col];
      mapC[oidx] = hv[2][row][col];
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
// i.e. the transposed correlation: taps are read flipped.  Same work unit and passes as the forward, over the three
// derivative maps.
__global__ __launch_bounds__(256) void k_loss_backward(const int C, const int H, const int W,
                                                       const float* __restrict__ img, const float* __restrict__ gt,
                                                       const LossWindow win, const float* __restrict__ mapA,
                                                       const float* __restrict__ mapB, const float* __restrict__ mapC,
                                                       const float inv_n, const float lambda,
                                                       float* __restrict__ dL_dimg) {
  __shared__ float hv[3][LTY_][LSTRIDE_];
  const int c = blockIdx.z;
  const int x0 = blockIdx.x * LTX_, y0 = blockIdx.y * LTY_;
  const size_t plane = (size_t)c * H * W;
  const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int r = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * LHO_;
  float px[LFLAT_], py[LFLAT_];  // img / gt at this thread's output pixels (flattened order), requested with the maps
  {
    const int gx = x0 - LR_ + lane;
    const bool col_in = gx >= 0 && gx < W;
    const int gxc = min(max(gx, 0), W - 1);
    const float* MA = mapA + plane;
    const float* MB = mapB + plane;
    const float* MC = mapC + plane;
    float va[LIN_], vb[LIN_], vc[LIN_];
#pragma unroll
    for (int i = 0; i < LIN_; i++) {  // (clamped addresses, unconditional loads, zero padding by select)
      const int gy = y0 + LSEG_ * seg - LR_ + i;
      const int off = min(max(gy, 0), H - 1) * W + gxc;
      const float a = MA[off], b = MB[off], cc = MC[off];
      const bool in = col_in && gy >= 0 && gy < H;
      va[i] = in ? a : 0.f;
      vb[i] = in ? b : 0.f;
      vc[i] = in ? cc : 0.f;
    }
#pragma unroll
    for (int it = 0; it < LFLAT_; it++) {  // (pixels outside the image or the tile are never stored: any valid address)
      const int e = it * 256 + (int)threadIdx.x;
      const int row = e / LTX_, col = e - row * LTX_;
      const int off = min(y0 + row, H - 1) * W + min(x0 + col, W - 1);
      px[it] = img[plane + off];
      py[it] = gt[plane + off];
    }
    float acc[LSEG_][3];
#pragma unroll
    for (int o = 0; o < LSEG_; o++) acc[o][0] = acc[o][1] = acc[o][2] = 0.f;
#pragma unroll
    for (int i = 0; i < LIN_; i++) {
#pragma unroll
      for (int o = 0; o < LSEG_; o++) {
        const int k = i - o;
        if (k >= 0 && k < LW_) {
          const float wk = win.w[LW_ - 1 - k];  // flipped taps
          acc[o][0] += wk * va[i]; acc[o][1] += wk * vb[i]; acc[o][2] += wk * vc[i];
        }
      }
    }
#pragma unroll
    for (int o = 0; o < LSEG_; o++) {
#pragma unroll
      for (int q = 0; q < 3; q++) hv[q][LSEG_ * seg + o][lane] = acc[o][q];
    }
  }
  __syncthreads();
  float t[3][LHO_];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    float v[LHO_ + LW_ - 1];
#pragma unroll
    for (int k = 0; k < LHO_ + LW_ - 1; k++) v[k] = hv[q][r][c0 + k];
#pragma unroll
    for (int o = 0; o < LHO_; o++) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < LW_; k++) a += win.w[LW_ - 1 - k] * v[o + k];
      t[q][o] = a;
    }
  }
  // the filtered maps change hands through LDS (as the forward's outputs do): the image is read and the gradient
  // written in the flattened order of the tile, 54-float row segments to consecutive lanes
  __syncthreads();  // (everyone has read hv)
#pragma unroll
  for (int o = 0; o < LHO_; o++) { hv[0][r][c0 + o] = t[0][o]; hv[1][r][c0 + o] = t[1][o]; hv[2][r][c0 + o] = t[2][o]; }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < LFLAT_; it++) {
    const int e = it * 256 + (int)threadIdx.x;
    const int row = e / LTX_, col = e - row * LTX_;
    const int gx = x0 + col, gy = y0 + row;
    if (e < LTX_ * LTY_ && gx < W && gy < H) {
      const float x = px[it], y = py[it];
      const float d = x - y;
      const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);  // torch: abs'(0) = 0
      dL_dimg[plane + (size_t)gy * W + gx] =
          (1.f - lambda) * inv_n * sgn -
          lambda * inv_n * (hv[0][row][col] + 2.f * x * hv[1][row][col] + y * hv[2][row][col]);
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
  if (dL_dimg) {  // (first: the render backward waits for this one, nobody on the stream for the three scalars)
    ProfScope ps(K_LOSS_BWD, s);
    hipLaunchKernelGGL(k_loss_backward, grid, dim3(256), 0, s, C, H, W, img, gt, win, mapA, mapB, mapC, inv_n, lambda,
                       dL_dimg);
  }
  {
    ProfScope ps(K_LOSS_FINALIZE, s);
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(1024), 0, s, partials, nblocks, inv_n, lambda, loss_out3);
  }
  return hipGetLastError();
}

}  // namespace gsr
