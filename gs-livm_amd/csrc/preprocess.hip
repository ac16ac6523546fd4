// preprocess.hip -- per-Gaussian and per-instance bookkeeping stages of the MI355X rasterizer (gfx950):
//   F1  k_preprocess        project / cull / EWA / SH->RGB, footprint-box tile rectangle, 48-B splat record,
//                           depth-sort key + the depth sort's digit histograms and the tile counts per top key
//                           byte; publishes num_rendered to the host mailbox (last workgroup)
//   F2' k_compact_near      partial depth sort: the near candidates of a near/far frame, compacted in id order
//   F3  k_scan_offsets      single-launch (look-back) scan of the tile counts in depth order; 16-byte emission
//                           descriptors in depth order, slotinfo, emission chunk table; near phase of a near/far
//                           frame: up to the Gaussian in whose slot run the budget falls
//       k_scan_offsets_far  the far Gaussians whose rectangle still holds an unfinished tile
//   F4  k_emit              output-centric emission of (tile id, Gaussian id) pairs in depth order; with 16-bit
//                           tile ids it only counts the digits of the sort's first pass, and
//       k_emit_scatter      generates the pairs inside that pass (they never reach HBM unsorted)
//   B2a k_compact_touched / k_gather_records   per-Gaussian sums of the per-instance gradient records (no atomics,
//                           fixed order)
//   B2b k_gaussian_backward EWA / projection / SH / covariance chain rule
//   V1  k_mark_visible;  debug only: k_point_offsets
//
// THIS FILE IS COMPILED WITH -ffp-contract=off.  radii, tile rectangles and the depth bits of the
// sort keys are exact-match targets: every f32 operation below that feeds them is written in the
// evaluation order of the reference (GLM column-major products, accumulated left to right) so the
// rounding matches the unfused CPU oracle bit for bit.  Divisions and sqrt are IEEE (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt).
#include <cstdlib>

#include "gsr_internal.hpp"
#include "sort_core.hpp"

namespace gsr {

// CUDA's min/max on floats are fminf/fmaxf (NaN-ignoring); v_min_f32/v_max_f32 have the same semantics.
__device__ __forceinline__ float fmin_(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float fmax_(float a, float b) { return fmaxf(a, b); }

// ndc2Pix (reference auxiliary.h:35-37) is evaluated in f64 there (double literals) and rounded once.
__device__ __forceinline__ float ndc_to_pix(float v, int S) {
  return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Inclusive scan across the 64 lanes of a wave.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

constexpr float SH0 = 0.28209479177387814f;
constexpr float SH1 = 0.4886025119029199f;
__device__ const float SH2c[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                  -1.0925484305920792f, 0.5462742152960396f};
__device__ const float SH3c[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                  -0.4570457994644658f, 1.445305721320277f,  -0.5900435899266435f};

// Symmetric 3D covariance (6 floats) from scale and quaternion (r,x,y,z), used as given.
// Restates forward.cu:138-176: M = S*R (column-major), Sigma = M^T M.
__device__ __forceinline__ void cov3d_from_scale_rot(const float s0, const float s1, const float s2, const float4 q,
                                                     float* c6) {
  const float r = q.x, x = q.y, y = q.z, z = q.w;
  // R[c][r]: column c, row r
  const float R00 = 1.f - 2.f * (y * y + z * z), R01 = 2.f * (x * y - r * z), R02 = 2.f * (x * z + r * y);
  const float R10 = 2.f * (x * y + r * z), R11 = 1.f - 2.f * (x * x + z * z), R12 = 2.f * (y * z - r * x);
  const float R20 = 2.f * (x * z - r * y), R21 = 2.f * (y * z + r * x), R22 = 1.f - 2.f * (x * x + y * y);
  // M[c][r] = s_r * R[c][r]
  const float M00 = s0 * R00, M01 = s1 * R01, M02 = s2 * R02;
  const float M10 = s0 * R10, M11 = s1 * R11, M12 = s2 * R12;
  const float M20 = s0 * R20, M21 = s1 * R21, M22 = s2 * R22;
  // Sigma[c][r] = M[r][0]*M[c][0] + M[r][1]*M[c][1] + M[r][2]*M[c][2]
  c6[0] = M00 * M00 + M01 * M01 + M02 * M02;  // Sigma[0][0]
  c6[1] = M10 * M00 + M11 * M01 + M12 * M02;  // Sigma[0][1]
  c6[2] = M20 * M00 + M21 * M01 + M22 * M02;  // Sigma[0][2]
  c6[3] = M10 * M10 + M11 * M11 + M12 * M12;  // Sigma[1][1]
  c6[4] = M20 * M10 + M21 * M11 + M22 * M12;  // Sigma[1][2]
  c6[5] = M20 * M20 + M21 * M21 + M22 * M22;  // Sigma[2][2]
}

struct Ewa {
  float T00, T01, T02, T10, T11, T12;  // T[c][r] for c = 0,1 (third column is zero)
  float tx, ty, tz, txtz, tytz;
  float cxx, cxy, cyy;  // cov2D before the +0.3 low-pass
};

// Restates forward.cu:79-126 / backward.cu:159-200: T = W*J, cov = T^T Vrk^T T.
__device__ __forceinline__ Ewa ewa_project(const float mx, const float my, const float mz, const FrameParams& fp,
                                           const float* __restrict__ c6, const float* __restrict__ V) {
  Ewa e;
  float t0 = V[0] * mx + V[4] * my + V[8] * mz + V[12];
  float t1 = V[1] * mx + V[5] * my + V[9] * mz + V[13];
  const float t2 = V[2] * mx + V[6] * my + V[10] * mz + V[14];
  const float limx = 1.3f * fp.tan_fovx, limy = 1.3f * fp.tan_fovy;
  e.txtz = t0 / t2;
  e.tytz = t1 / t2;
  t0 = fmin_(limx, fmax_(-limx, e.txtz)) * t2;
  t1 = fmin_(limy, fmax_(-limy, e.tytz)) * t2;
  e.tx = t0; e.ty = t1; e.tz = t2;
  const float J00 = fp.focal_x / t2, J02 = -(fp.focal_x * t0) / (t2 * t2);
  const float J11 = fp.focal_y / t2, J12 = -(fp.focal_y * t1) / (t2 * t2);
  // W[k][r] = V[k + 4r];  T[0][r] = W[0][r]*J00 + W[2][r]*J02;  T[1][r] = W[1][r]*J11 + W[2][r]*J12
  e.T00 = V[0] * J00 + V[2] * J02;
  e.T01 = V[4] * J00 + V[6] * J02;
  e.T02 = V[8] * J00 + V[10] * J02;
  e.T10 = V[1] * J11 + V[2] * J12;
  e.T11 = V[5] * J11 + V[6] * J12;
  e.T12 = V[9] * J11 + V[10] * J12;
  // A[c][r] = T[r][0]*S(0,c) + T[r][1]*S(1,c) + T[r][2]*S(2,c), S symmetric from c6
  const float S00 = c6[0], S01 = c6[1], S02 = c6[2], S11 = c6[3], S12 = c6[4], S22 = c6[5];
  const float A00 = e.T00 * S00 + e.T01 * S01 + e.T02 * S02;
  const float A10 = e.T00 * S01 + e.T01 * S11 + e.T02 * S12;
  const float A20 = e.T00 * S02 + e.T01 * S12 + e.T02 * S22;
  const float A01 = e.T10 * S00 + e.T11 * S01 + e.T12 * S02;
  const float A11 = e.T10 * S01 + e.T11 * S11 + e.T12 * S12;
  const float A21 = e.T10 * S02 + e.T11 * S12 + e.T12 * S22;
  e.cxx = A00 * e.T00 + A10 * e.T01 + A20 * e.T02;  // cov[0][0]
  e.cxy = A01 * e.T00 + A11 * e.T01 + A21 * e.T02;  // cov[0][1]
  e.cyy = A01 * e.T10 + A11 * e.T11 + A21 * e.T12;  // cov[1][1]
  return e;
}

// ------------------------------------------------------------------------------------------------
// SH rows at M > 1 (the degree-3 stress configuration: 192 B per Gaussian).  A thread reading or writing
// its own [M][3] row straight from global memory touches 64 different 128-byte lines per wave
// instruction and the 48 KB a workgroup covers does not survive in the vector L1.  So the workgroup moves its
// 256 consecutive rows as ONE contiguous block with 16-byte accesses, through LDS with an odd row stride
// (3M | 1 words: per-thread row walks are then bank-conflict free), and the arithmetic reads / writes LDS.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int sh_row_stride(int C) { return C | 1; }

__device__ __forceinline__ void rows_to_lds(float* lds, const float* __restrict__ src, const int nrows, const int C) {
  const int n = nrows * C, S = sh_row_stride(C);
  if ((C & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {  // block-uniform
    const int C4 = C >> 2, n4 = n >> 2;
    const int dr = PRE_BLOCK / C4, dc = PRE_BLOCK - dr * C4;
    int r = (int)threadIdx.x / C4, c4 = (int)threadIdx.x - r * C4;
    for (int j = threadIdx.x; j < n4; j += PRE_BLOCK) {
      const float4 v = reinterpret_cast<const float4*>(src)[j];
      float* d = lds + r * S + 4 * c4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      r += dr; c4 += dc;
      if (c4 >= C4) { c4 -= C4; r++; }
    }
  } else {
    const int dr = PRE_BLOCK / C, dc = PRE_BLOCK - dr * C;
    int r = (int)threadIdx.x / C, c = (int)threadIdx.x - r * C;
    for (int e = threadIdx.x; e < n; e += PRE_BLOCK) {
      lds[r * S + c] = src[e];
      r += dr; c += dc;
      if (c >= C) { c -= C; r++; }
    }
  }
}

__device__ __forceinline__ void lds_to_rows(float* __restrict__ dst, const float* lds, const int nrows, const int C) {
  const int n = nrows * C, S = sh_row_stride(C);
  if ((C & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
    const int C4 = C >> 2, n4 = n >> 2;
    const int dr = PRE_BLOCK / C4, dc = PRE_BLOCK - dr * C4;
    int r = (int)threadIdx.x / C4, c4 = (int)threadIdx.x - r * C4;
    for (int j = threadIdx.x; j < n4; j += PRE_BLOCK) {
      const float* d = lds + r * S + 4 * c4;
      reinterpret_cast<float4*>(dst)[j] = make_float4(d[0], d[1], d[2], d[3]);
      r += dr; c4 += dc;
      if (c4 >= C4) { c4 -= C4; r++; }
    }
  } else {
    const int dr = PRE_BLOCK / C, dc = PRE_BLOCK - dr * C;
    int r = (int)threadIdx.x / C, c = (int)threadIdx.x - r * C;
    for (int e = threadIdx.x; e < n; e += PRE_BLOCK) {
      dst[e] = lds[r * S + c];
      r += dr; c += dc;
      if (c >= C) { c -= C; r++; }
    }
  }
}

// The rows of the listed Gaussians only (block-local row numbers in `list`, `nlist` of them): the backward needs the SH
// coefficients of the Gaussians that carry a gradient record -- a few per cent of a dense scene -- and writes zeros for
// the rest.  Row-major chunks of 16 bytes when the rows allow it, single floats otherwise.
__device__ __forceinline__ void listed_rows_to_lds(float* lds, const float* __restrict__ src, const uint32_t* list,
                                                   const int nlist, const int C) {
  const int S = sh_row_stride(C);
  if ((C & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {  // block-uniform
    const int C4 = C >> 2;
    for (int k = threadIdx.x; k < nlist * C4; k += PRE_BLOCK) {
      const int r = (int)list[k / C4], c4 = k % C4;
      const float4 v = reinterpret_cast<const float4*>(src + (size_t)r * C)[c4];
      float* d = lds + r * S + 4 * c4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  } else {
    for (int k = threadIdx.x; k < nlist * C; k += PRE_BLOCK) {
      const int r = (int)list[k / C], c = k % C;
      lds[r * S + c] = src[(size_t)r * C + c];
    }
  }
}

// bytes of dynamic LDS the staged variants need; 0 = do not stage (M == 1, or rows too long for 64 KB)
static inline size_t sh_stage_bytes(int M) {
  if (M <= 1) return 0;
  const size_t b = (size_t)PRE_BLOCK * (size_t)((3 * M) | 1) * sizeof(float);
  return b <= 64 * 1024 ? b : 0;
}

// ------------------------------------------------------------------------------------------------
// F1.  One thread per Gaussian.  Replaces preprocessCUDA (reference forward.cu:179-286).
// ------------------------------------------------------------------------------------------------
// PLAIN = the product's configuration, fixed at compile time: SH degree 0 read from `shs`, scales + rotations given, no
// precomputed covariance or colour.  The general variant keeps those inputs behind run-time branches, and every such
// branch with a load in it ends in an `s_waitcnt vmcnt(0)` at the join -- vector loads return in order, so that wait
// also waits for the NEXT block's inputs requested at the top of the iteration, and the prefetch never overlaps the
// arithmetic.  With the branches compiled away no load is issued between the prefetch and the next iteration.
// PIN ("plain inputs", implied by PLAIN) = scales + rotations + SH given, nothing precomputed, no debug copy of cov3D,
// at any SH degree: the same compile-time removal of optional loads for the LDS-staged variant.  ROWK > 0 (PIN, STAGED,
// rows of 4 ROWK floats at a 16-byte aligned base: degree 1 and 3): the SH rows take part in the software pipeline --
// every WAVE requests the rows of its own 64 Gaussians of the NEXT block as ROWK fully coalesced 16-byte loads per lane
// before it works on this one, and moves them into its quarter of the LDS image when it gets there (wave-private: fences
// instead of the two workgroup barriers per block of the block-wide copy, whose loads nothing overlapped).
template <bool STAGED, bool PLAIN = false, bool PIN = PLAIN, int ROWK = 0>
__global__ __launch_bounds__(PRE_BLOCK) void k_preprocess(
    const FrameParams fp, const float* __restrict__ means3D, const float* __restrict__ scales,
    const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs,
    const float* __restrict__ cov3D_precomp, const float* __restrict__ colors_precomp,
    const float* __restrict__ V, const float* __restrict__ Pm, const float* __restrict__ campos, GeomState g,
    int* __restrict__ radii_out, const bool write_cov3D, unsigned long long* __restrict__ done_word,
    unsigned long long* __restrict__ publish,
    const uint32_t ticket, uint32_t* __restrict__ ghist_acc, uint32_t* __restrict__ ghist_clear) {
  // Persistent-style grid: the launcher sizes the grid to ONE resident round of workgroups (preprocess_grid) and
  // workgroup b walks the 256-Gaussian blocks b, b + grid, b + 2 grid, ...: no second, mostly empty round of
  // workgroups at the end, every workgroup does the same number of blocks (+-1), and the instance count below
  // costs one same-address atomic per workgroup (they retire one at a time, ~5 ns each) instead of one per block.
  // The two camera matrices are read ONCE, here, before the first store of the kernel: uniform addresses that nothing
  // has clobbered yet become scalar loads into SGPRs.  Read where they are used -- inside the block loop, behind the
  // previous block's stores -- the compiler had to fetch every entry with a vector load per thread (25 per Gaussian), and
  // since vector loads return in order, waiting for one of them also waited for the NEXT block's inputs requested
  // just before: the prefetch never overlapped the arithmetic.
  if (blockIdx.x == 0 && threadIdx.x == 0) { g.total[2] = 0u; g.total[11] = 0u; }  // ([11]: no near budget crossed yet)
  float Vc[16], Pc[16];
#pragma unroll
  for (int k = 0; k < 16; k++) { Vc[k] = V[k]; Pc[k] = Pm[k]; }
  const float cam0 = campos[0], cam1 = campos[1], cam2 = campos[2];
  extern __shared__ float sh_rows[];  // STAGED: the sub-block's SH rows (rows_to_lds)
  // The depth sort's four digit histograms are counted here, where the keys are made (LDS atomics, one flush of
  // the non-empty bins per workgroup): the sort needs no histogram pass of its own.  ghist_acc is library-owned
  // and zero on entry; ghist_clear is the buffer the NEXT forward will count into.
  // (the STAGED variant too: 5 KB beside its SH rows -- 12 to 50 KB -- and the record images; a near/far frame's
  // near limit and partial depth sort hang on these counts)
  // (row 4: the tile counts summed by the keys' TOP byte -- with row 3 it tells k_scan_offsets how far into the depth
  // order the near phase of a near/far frame can reach)
  __shared__ uint32_t dhist[5][256];
  __shared__ float4 s_rec[PRE_BLOCK / 64][64 * SPLAT_F4];  // per wave: the records of its 64 Gaussians on their way out
  if (ghist_acc) {
#pragma unroll
    for (int k = 0; k < 5; k++) dhist[k][threadIdx.x] = 0u;
    if (blockIdx.x == 0)
#pragma unroll
      for (int k = 0; k < 5; k++) ghist_clear[k * 256 + threadIdx.x] = 0u;
    __syncthreads();
  }
  // Every input of a Gaussian is fetched up front and unconditionally (one round trip instead of one per cull
  // stage: means -> scales -> rotation -> opacity / colour), and the NEXT sub-block's inputs are requested before
  // this one is worked on, so the arithmetic of a wave overlaps its own loads.
  struct In { float mx, my, mz, s0, s1, s2, op, dc0, dc1, dc2; float4 q; };
  const bool dc_direct = PLAIN || (!STAGED && shs && !colors_precomp);  // first SH coefficient read with the other inputs
  const int D = PLAIN ? 0 : fp.D;
  // (always executed, on an index clamped into the array: a load behind a branch is merged with the "not loaded" value
  // by a register copy placed right after it, and that copy makes the wave wait for the load at once)
  auto fetch = [&](const int i_) {
    In v;
    const int i = i_ < fp.P ? i_ : fp.P - 1;
    v.s0 = v.s1 = v.s2 = v.dc0 = v.dc1 = v.dc2 = 0.f;
    v.q = make_float4(0.f, 0.f, 0.f, 0.f);
    v.mx = means3D[3 * i]; v.my = means3D[3 * i + 1]; v.mz = means3D[3 * i + 2];
    if (PIN || scales) { v.s0 = scales[3 * i]; v.s1 = scales[3 * i + 1]; v.s2 = scales[3 * i + 2]; }
    if (PIN || !cov3D_precomp) v.q = reinterpret_cast<const float4*>(rotations)[i];
    v.op = opacities[i];
    if (dc_direct) {
      const float* r = shs + (size_t)i * fp.M * 3;
      v.dc0 = r[0]; v.dc1 = r[1]; v.dc2 = r[2];
    }
    return v;
  };
  uint32_t tiles_wg = 0;
  const int nblk = (fp.P + PRE_BLOCK - 1) / PRE_BLOCK;
  // one block of 256 Gaussians whose inputs are in registers
  // Every vector-memory instruction of a block is issued UNCONDITIONALLY: the hardware counts loads and stores in one
  // in-order counter, and as soon as a store sits behind a branch the compiler no longer knows how many are in flight
  // and waits for ALL of them (vmcnt(0)) wherever it needs an earlier load -- i.e. for the stores it has just issued,
  // every block.  So the lanes of the last block that lie beyond P act as duplicates of Gaussian P - 1 (fetch() clamps
  // the same way): same inputs, same results, stored to the same place (a benign same-value race), not counted.
  auto work = [&](const In& in, const int blk, auto&& before_stores) {
  const int idx_raw = blk * PRE_BLOCK + threadIdx.x;
  const bool dup = idx_raw >= fp.P;
  const int idx = dup ? fp.P - 1 : idx_raw;
  if (STAGED && !(ROWK > 0 && (blk + 1) * PRE_BLOCK <= fp.P)) {  // (ROWK: only the last, partial block -- its duplicate
    const int row0 = blk * PRE_BLOCK;                            // lanes read a row another wave has staged)
    if (blk != (int)blockIdx.x) __syncthreads();  // the previous block's rows have been read
    if (row0 < fp.P) rows_to_lds(sh_rows, shs + (size_t)row0 * fp.M * 3, min(PRE_BLOCK, fp.P - row0), fp.M * 3);
    __syncthreads();
  }
  uint32_t tiles = 0, rect_packed = 0, dkey = 0xFFFFFFFFu;
  int radius = 0;
  float4 rec0 = make_float4(0.f, 0.f, 0.f, 0.f), rec1 = rec0, rec2 = rec0;  // splat record (zeros for a culled Gaussian)
  uint8_t clamp_out = 0;
  {
    const float mx = in.mx, my = in.my, mz = in.mz;
    const float pvz = Vc[2] * mx + Vc[6] * my + Vc[10] * mz + Vc[14];
    bool alive = !(pvz <= 0.2f);  // near cull only (forward.cu:221-225)
    float s0 = 0, s1 = 0, s2 = 0;
    if (alive && (PIN || scales)) {  // scale cull (forward.cu:19-25)
      s0 = fp.scale_modifier * in.s0;
      s1 = fp.scale_modifier * in.s1;
      s2 = fp.scale_modifier * in.s2;
      alive = !(s0 > 0.3f || s1 > 0.3f || s2 > 0.3f);
    }
    if (alive) {
      const float ph0 = Pc[0] * mx + Pc[4] * my + Pc[8] * mz + Pc[12];
      const float ph1 = Pc[1] * mx + Pc[5] * my + Pc[9] * mz + Pc[13];
      const float ph3 = Pc[3] * mx + Pc[7] * my + Pc[11] * mz + Pc[15];
      const float pw = 1.0f / (ph3 + 0.0000001f);
      const float ppx = ph0 * pw, ppy = ph1 * pw;
      float c6[6];
      if (!PIN && cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) c6[k] = cov3D_precomp[6 * idx + k];
      } else {
        cov3d_from_scale_rot(s0, s1, s2, in.q, c6);
        if (!PIN && write_cov3D) {  // debug forwards only (the views): the backward recomputes it with the same arithmetic
#pragma unroll
          for (int k = 0; k < 6; k++) g.cov3D[6 * (size_t)idx + k] = c6[k];
        }
      }
      const Ewa e = ewa_project(mx, my, mz, fp, c6, Vc);
      const float cx = e.cxx + 0.3f, cy = e.cxy, cz = e.cyy + 0.3f;
      const float det = cx * cz - cy * cy;
      if (det != 0.0f) {
        const float det_inv = 1.f / det;
        const float conx = cz * det_inv, cony = -cy * det_inv, conz = cx * det_inv;
        const float mid = 0.5f * (cx + cz);
        const float disc = sqrtf(fmax_(0.1f, mid * mid - det));
        const float lambda1 = mid + disc, lambda2 = mid - disc;
        const float my_radius = ceilf(3.f * sqrtf(fmax_(lambda1, lambda2)));
        const float pixx = ndc_to_pix(ppx, fp.W), pixy = ndc_to_pix(ppy, fp.H);
        int x0, y0, x1, y1;
        tile_rect(pixx, pixy, (int)my_radius, fp.gx, fp.gy, x0, y0, x1, y1);
        const int area = (x1 - x0) * (y1 - y0);
        if (area != 0) {
          float rgb[3];
          uint8_t clampbits = 0;
          if (!PIN && colors_precomp) {
            rgb[0] = colors_precomp[3 * idx];
            rgb[1] = colors_precomp[3 * idx + 1];
            rgb[2] = colors_precomp[3 * idx + 2];
          } else {  // SH -> RGB (forward.cu:29-76)
            float x = 0.f, y = 0.f, z = 0.f;
            if (D > 0) {  // the view direction only enters from degree 1 on (the product runs degree 0)
              const float d0 = mx - cam0, d1 = my - cam1, d2 = mz - cam2;
              const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
              x = d0 / len; y = d1 / len; z = d2 / len;
            }
            // (STAGED: the block's rows sit in LDS in block order; a duplicate lane reads Gaussian P - 1's row)
            const float* sh = STAGED ? sh_rows + (idx - blk * PRE_BLOCK) * sh_row_stride(fp.M * 3)
                                     : shs + (size_t)idx * fp.M * 3;
            const float dc[3] = {dc_direct ? in.dc0 : sh[0], dc_direct ? in.dc1 : sh[1], dc_direct ? in.dc2 : sh[2]};
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
              float res = SH0 * dc[ch];
              if (D > 0) {
                res = res - SH1 * y * sh[3 + ch] + SH1 * z * sh[6 + ch] - SH1 * x * sh[9 + ch];
                if (D > 1) {
                  const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                  res = res + SH2c[0] * xy * sh[12 + ch] + SH2c[1] * yz * sh[15 + ch] +
                        SH2c[2] * (2.0f * zz - xx - yy) * sh[18 + ch] + SH2c[3] * xz * sh[21 + ch] +
                        SH2c[4] * (xx - yy) * sh[24 + ch];
                  if (D > 2) {
                    res = res + SH3c[0] * y * (3.0f * xx - yy) * sh[27 + ch] + SH3c[1] * xy * z * sh[30 + ch] +
                          SH3c[2] * y * (4.0f * zz - xx - yy) * sh[33 + ch] +
                          SH3c[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + ch] +
                          SH3c[4] * x * (4.0f * zz - xx - yy) * sh[39 + ch] + SH3c[5] * z * (xx - yy) * sh[42 + ch] +
                          SH3c[6] * x * (xx - 3.0f * yy) * sh[45 + ch];
                  }
                }
              }
              res += 0.5f;
              if (res < 0) clampbits |= (uint8_t)(1u << ch);
              rgb[ch] = fmax_(res, 0.0f);
            }
          }
          const float op = in.op;
          // Exact-conservative footprint: a pixel can only receive alpha = min(.99, op*exp(power)) >= 1/255
          // if -power <= ln(255*op); the axis-aligned box of that ellipse (plus slack for rounding in
          // the blend kernels) bounds every contributing pixel.  op < 1/255 never contributes.
          float hx, hy;
          if (op < 1.0f / 255.0f) {
            hx = hy = -1e30f;  // box test can never pass
          } else {
            const float tau = footprint_tau(op);
            const float dc = conx * conz - cony * cony;
            if (conx > 0.0f && conz > 0.0f && dc > 0.0f) {
              hx = sqrtf(2.0f * tau * conz / dc) + 0.05f;
              hy = sqrtf(2.0f * tau * conx / dc) + 0.05f;
            } else {
              hx = hy = 1e30f;  // indefinite conic: no culling
            }
          }
          // Tiles: the reference's square of the 3-sigma radius (auxiliary.h:39-46), cut down to the tiles the
          // footprint box overlaps -- an instance outside it has alpha < 1/255 at every pixel of its tile and the
          // reference skips it pixel by pixel (forward.cu:343-345, backward.cu:476-478), so images and gradients
          // are unchanged while ~30 % of the instances are never emitted, sorted or walked.  radii stays the
          // reference's value.
          // gsr_set_reference_rects(1) keeps the reference's square as it is: tiles_touched, num_rendered, the sorted
          // lists, ranges and n_contrib are then the reference's own, bit for bit (auxiliary.h:39-46,
          // rasterizer_impl.cu:64-125).
          if (fp.ref_rects) {
          } else if (hx < 0.0f) {
            x1 = x0;
          } else if (hx < 1e6f) {
            const float lim = 1e6f;
            const int bx0 = (int)ceilf(fmax_(-lim, fmin_(lim, (pixx - hx - 15.0f) * 0.0625f)));
            const int bx1 = (int)floorf(fmax_(-lim, fmin_(lim, (pixx + hx) * 0.0625f))) + 1;
            const int by0 = (int)ceilf(fmax_(-lim, fmin_(lim, (pixy - hy - 15.0f) * 0.0625f)));
            const int by1 = (int)floorf(fmax_(-lim, fmin_(lim, (pixy + hy) * 0.0625f))) + 1;
            x0 = max(x0, bx0); x1 = min(x1, bx1);
            y0 = max(y0, by0); y1 = min(y1, by1);
          }
          radius = (int)my_radius;
          tiles = (x1 > x0 && y1 > y0) ? (uint32_t)((x1 - x0) * (y1 - y0)) : 0u;
          rect_packed = (uint32_t)x0 | ((uint32_t)y0 << 10) | ((uint32_t)(x1 - x0) << 20);
          if (tiles) dkey = __float_as_uint(pvz);  // depth > 0.2: the bit pattern orders like the value
          rec0 = make_float4(pixx, pixy, conx, cony);
          rec1 = make_float4(conz, op, rgb[0], rgb[1]);
          rec2 = make_float4(rgb[2], pvz, hx, hy);
          clamp_out = clampbits;

        }
      }
    }
  }
  before_stores();  // (the loop takes over the next block's inputs here, see below)
  radii_out[idx] = radius;  // the caller's array, or the blob's own when it passed none (launcher)
  g.gpack[idx] = make_uint2(tiles, tiles ? rect_packed : 0u);
  g.touched[idx] = 0;  // backward bookkeeping starts clean (the backward clears what it sets)
  g.clamped[idx] = clamp_out;
  // key of the per-Gaussian depth sort (its value is the position, idx: no id array is written); Gaussians without
  // instances sort to the end
  g.dkeysA[idx] = dkey;
  if (dup) { tiles = 0u; dkey = 0xFFFFFFFFu; }  // a duplicate lane is not counted below
  {
    // The 48-byte records of a wave's 64 Gaussians leave as three fully contiguous 1-KB stores: each lane parks its record
    // in the wave's LDS image (stride 48 B: conflict-free) and stores float4 number lane + 64 k of the image.  Stored
    // straight from the lane that computed it a record is three 16-byte pieces at a 48-byte stride -- every store
    // instruction half-fills 64 sectors -- and the kernel spent 21 of its 74 us on these 83 MB (measured by leaving
    // them out).  Wave-private, in program order: no barrier.
    float4* img = s_rec[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    img[3 * lane + 0] = rec0;
    img[3 * lane + 1] = rec1;
    img[3 * lane + 2] = rec2;
    // lanes exchange data here: without a (wave-scope) release / acquire pair the compiler reasons per thread -- a
    // thread never reads its own img[3 lane + 1] back -- and deletes that store.  No instruction is emitted for it.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the wave's first Gaussian (a wave wholly beyond P holds 64 copies of Gaussian P - 1's record: it re-stores that)
    const int first = min(blk * PRE_BLOCK + (int)(threadIdx.x & ~63u), fp.P - 1);
    const int nrec = min(64, fp.P - first) * SPLAT_F4;  // float4s of the image that belong to Gaussians (>= 3)
    float4* dst = g.splats + (size_t)first * SPLAT_F4;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int j = min(lane + 64 * k, nrec - 1);       // (beyond the last record: its last float4 once more)
      dst[j] = img[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the image is rewritten by the wave's next block)
    __builtin_amdgcn_wave_barrier();
  }
  if (ghist_acc) {
    // Gaussians without instances all carry the key 0xFFFFFFFF: counted per wave, not per lane (one address)
    const bool has = !dup, none = has && dkey == 0xFFFFFFFFu;
    const uint64_t nm = __ballot(none);
    if (has && !none) {
#pragma unroll
      for (int k = 0; k < 4; k++) atomicAdd(&dhist[k][(dkey >> (8 * k)) & 255u], 1u);
      atomicAdd(&dhist[4][dkey >> 24], tiles);
    }
    if (nm != 0ull && (threadIdx.x & 63) == 0) {
      const uint32_t c = (uint32_t)__popcll(nm);
#pragma unroll
      for (int k = 0; k < 4; k++) atomicAdd(&dhist[k][255], c);
    }
  }
  tiles_wg += tiles;
  };
  // Software pipeline over the blocks: the next block's inputs are requested before this block is worked on, and taken
  // over -- the one place that has to wait for them -- AFTER this block's arithmetic and BEFORE its stores are issued.
  // Loads and stores share one in-order counter: a wait placed behind the stores (where the compiler puts the loop-
  // carried register copies by itself) would wait for the stores just issued, every block; placed here it waits for
  // loads requested a whole block ago and stores issued a block ago.  The empty asm pins the take-over to this spot.
  {
    auto pin = [](In& v) {
      asm volatile("" : "+v"(v.mx), "+v"(v.my), "+v"(v.mz), "+v"(v.s0), "+v"(v.s1), "+v"(v.s2), "+v"(v.op));
      asm volatile("" : "+v"(v.dc0), "+v"(v.dc1), "+v"(v.dc2), "+v"(v.q.x), "+v"(v.q.y), "+v"(v.q.z), "+v"(v.q.w));
    };
    In a = fetch(blockIdx.x * PRE_BLOCK + threadIdx.x);
    pin(a);  // (waited for here, so that the loop itself starts with nothing pending on these registers)
    // ROWK: this wave's 64 SH rows of a block = 64 ROWK consecutive float4s; lane l holds numbers l, l + 64, ...
    constexpr int NPRE = ROWK > 0 ? ROWK : 1;
    float4 pre[NPRE];
    const int lane_ = threadIdx.x & 63, wave0 = (int)(threadIdx.x & ~63u);
    auto rows_fetch = [&](const int blk) {  // unconditional, on addresses clamped into the array (see fetch)
      const int first = min(blk * PRE_BLOCK + wave0, fp.P - 1);
      const int n4 = min(64, fp.P - first) * NPRE;  // >= ROWK
      const float4* src = reinterpret_cast<const float4*>(shs + (size_t)first * (4 * NPRE));
#pragma unroll
      for (int k = 0; k < NPRE; k++) pre[k] = src[min(lane_ + 64 * k, n4 - 1)];
    };
    auto rows_commit = [&]() {  // float4 number j of the wave's image is columns 4 (j % ROWK) .. + 3 of its row j / NPRE
      float* img = sh_rows + wave0 * sh_row_stride(4 * NPRE);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the previous block's rows have been read)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < NPRE; k++) {
        const int j = lane_ + 64 * k, r = j / NPRE, c4 = j - r * NPRE;
        float* d = img + r * sh_row_stride(4 * NPRE) + 4 * c4;
        d[0] = pre[k].x; d[1] = pre[k].y; d[2] = pre[k].z; d[3] = pre[k].w;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // lanes read each other's pieces: see the record image below
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    if constexpr (ROWK > 0) rows_fetch(blockIdx.x);
    for (int blk = blockIdx.x; blk < nblk; blk += (int)gridDim.x) {
      if constexpr (ROWK > 0) {
        if ((blk + 1) * PRE_BLOCK <= fp.P) rows_commit();  // (no memory instruction behind this branch)
      }
      const In n = fetch((blk + (int)gridDim.x) * PRE_BLOCK + threadIdx.x);
      if constexpr (ROWK > 0) rows_fetch(blk + (int)gridDim.x);
      work(a, blk, [&] {
        a = n;
        pin(a);
      });
    }
  }
  // side job: clear the depth sort's histograms, tickets and look-back status words
  for (size_t w = (size_t)blockIdx.x * PRE_BLOCK + threadIdx.x; w < g.dsort.nwords; w += (size_t)gridDim.x * PRE_BLOCK)
    g.dsort.words[w] = 0u;
  __shared__ uint32_t wsum[PRE_BLOCK / 64];
  const uint32_t ws = wave_sum_u32(tiles_wg);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = ws;
  __syncthreads();
  if (ghist_acc) {
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const uint32_t c = dhist[k][threadIdx.x];
      if (c) (void)__hip_atomic_fetch_add(ghist_acc + k * 256 + threadIdx.x, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // num_rendered without a scan launch and without a release fence (an agent-scope fence writes back the XCD's
  // L2: ~75 ns per workgroup when 7813 of them do it): every workgroup adds (1 << 40 | its sum) to ONE 64-bit
  // word, so the count of finished workgroups and the running total travel in the same atomic.  The workgroup
  // that sees count == grid - 1 in the returned value is last: it knows the total, publishes it to the host
  // mailbox (a page-locked, host-mapped word the host polls: no copy engine, no interrupt) and resets the word,
  // which is library-owned device memory, for the next call.
  if (threadIdx.x == 0) {
    const uint32_t mine = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const unsigned long long old = atomicAdd(done_word, (1ull << 40) | (unsigned long long)mine);
    if ((old >> 40) == (unsigned long long)gridDim.x - 1ull) {
      const unsigned long long tot = (old & ((1ull << 40) - 1ull)) + mine;
      const uint32_t r = tot > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)tot;  // the host rejects R >= 2^31
      g.total[0] = r;
      __hip_atomic_store(done_word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (publish)
        __hip_atomic_store(publish, ((unsigned long long)ticket << 32) | r, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);  // self-contained value: no release (= L2 write-back) needed
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Debug only (the views of a debug forward): point_offsets = inclusive scan of tiles_touched in
// Gaussian-id order, the reference's array (rasterizer_impl.cu:270-273).  The pipeline itself never
// reads it (slots are assigned in depth order by k_scan_offsets), so this is one workgroup walking
// the array: simple, and off the production path.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_point_offsets(const int P, const uint2* __restrict__ gpack,
                                                        uint32_t* __restrict__ point_offsets) {
  __shared__ uint32_t wtot[16];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < P; base += 1024) {
    const int i = base + tid;
    const uint32_t v = i < P ? gpack[i].x : 0u;
    const uint32_t inc = wave_incl_scan_u32(v, lane);
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < w; k++) woff += wtot[k];
    const uint32_t carry = carry_s;
    if (i < P) point_offsets[i] = carry + woff + inc;
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + inc;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// F4.  Replaces duplicateWithKeys (reference rasterizer_impl.cu:64-101) and the InclusiveSum before it.
// Gaussians are visited in depth order; Gaussian order[i] gets the contiguous slot run
// [sdesc[i].x, sdesc[i+1].x) and its instances are emitted in the reference's row-major tile order with
// key = tile id, value = Gaussian id.  The depth part of the reference's 64-bit key is implied by the
// emission order, which the stable tile sort preserves.  A slot is also the index of the instance's
// gradient record in the backward.
//
// k_scan_offsets: ONE launch for the exclusive scan in depth order (decoupled look-back over workgroups of
// 4096 Gaussians: ticketed tiles, one 64-bit status word each, wave-wide look-back window) that also
//   * gathers (tiles_touched, rect) into depth order with one 8-byte load per Gaussian and writes the 16-byte
//     descriptor (slot, id, rect, 1/width) the emitters read coalesced, one load each; slotinfo for the backward,
//   * writes, for every EMIT_CHUNK boundary inside a Gaussian's run, the depth-order index of that
//     Gaussian (chunk_first: saves k_emit two dependent searches per workgroup),
//   * zeroes the tile ranges (the reference's cudaMemset, rasterizer_impl.cu:311).
// k_emit is OUTPUT-centric: every workgroup owns EMIT_CHUNK consecutive slots, stages the descriptors
// of the Gaussians covering them in LDS and each thread turns its slots into (tile, id) by bisection +
// an exact multiply-high division.  Work per workgroup is constant however skewed the tile counts are
// (the nearest Gaussians own >1000 tiles each and sit next to each other in depth order), and all
// stores are coalesced.
// ------------------------------------------------------------------------------------------------
constexpr unsigned long long SC_GLOBAL = 2ull << 32, SC_LOCAL = 1ull << 32;

// How far into the depth order can the near budget of a near/far frame reach?  k_preprocess left, per TOP BYTE of the
// depth keys, the number of Gaussians (top_hist[0..255]) and their tile counts (top_hist[256..511]); lanes own four
// consecutive byte values.  The budget falls into the first byte value at which the running tile count reaches it:
// `limit` = the Gaussians up to and including that value (0xFFFFFFFF: the budget is never reached), `top_end` = the
// first top-byte value behind it.  One wave; shared by k_scan_offsets and k_compact_near so that both draw the same line.
__device__ __forceinline__ void wave_near_limit(const uint32_t* __restrict__ top_hist, const uint32_t budget, const int lane,
                                                uint32_t& limit, uint32_t& top_end) {
  uint32_t cn = 0, sl = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) { cn += top_hist[4 * lane + q]; sl += top_hist[256 + 4 * lane + q]; }
  const uint32_t cn_inc = wave_incl_scan_u32(cn, lane), sl_inc = wave_incl_scan_u32(sl, lane);
  const uint64_t m = __ballot(sl_inc >= budget);
  // inside the lane's four values: the first one at which the running tile count reaches the budget (one top-byte value
  // = a factor of four in depth; the line is drawn behind it)
  uint32_t c = cn_inc - cn, t = sl_inc - sl, my_limit = 0xFFFFFFFFu, my_end = 256u;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    c += top_hist[4 * lane + q];
    t += top_hist[256 + 4 * lane + q];
    if (t >= budget && my_end == 256u) { my_limit = c; my_end = (uint32_t)(4 * lane + q + 1); }
  }
  limit = 0xFFFFFFFFu;
  top_end = 256u;
  if (m) {
    const int L = __builtin_ctzll(m);
    limit = __shfl(my_limit, L, 64);
    top_end = __shfl(my_end, L, 64);
  }
}

// Partial depth sort (api.hip).  A near/far frame whose far chain is expected to stay idle needs the depth order of
// the NEAR candidates only -- at 2 M Gaussians / 1080p some 50 000 of them -- so instead of sorting all P (key, id)
// pairs (4 passes over 2 M pairs: 0.11 ms) the candidates are compacted here, in id order (a stable sort of them
// then yields exactly the first `limit` entries of the full depth order), and only they are sorted; the full sort is
// left to the far chain, should it run.  One launch: decoupled look-back over tiles of 16 384 keys (64 per thread: a
// quarter of k_scan_offsets' chain length -- 27 -> .. us at 2 M), digit counts of the candidates by LDS atomics for the
// near sort's four passes.
constexpr int COMPACT_ITEMS = 64, COMPACT_TILE = PRE_BLOCK * COMPACT_ITEMS;
__global__ __launch_bounds__(PRE_BLOCK) void k_compact_near(const int P, const uint32_t* __restrict__ keys,
                                                            const uint32_t* __restrict__ top_hist, const uint32_t budget,
                                                            uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                            uint32_t* __restrict__ n_out, uint32_t* __restrict__ ghist_near,
                                                            unsigned long long* __restrict__ st, uint32_t* __restrict__ ticket,
                                                            unsigned long long* __restrict__ publish,
                                                            const uint32_t frame_ticket) {
  __shared__ uint32_t wtot[PRE_BLOCK / 64];
  __shared__ uint32_t s_tile, s_prefix, s_top_end;
  __shared__ uint32_t dh[4][256];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#pragma unroll
  for (int k = 0; k < 4; k++) dh[k][tid] = 0u;
  if (tid == 0) s_tile = atomicAdd(ticket, 1u);  // every lower tile is already running
  if (w == 0) {
    uint32_t limit, top_end;
    wave_near_limit(top_hist, budget, lane, limit, top_end);
    if (lane == 0) s_top_end = top_end;
  }
  __syncthreads();
  const uint32_t tile = s_tile, top_end = s_top_end;
  // a wave owns COMPACT_ITEMS x 64 consecutive keys and takes them 64 at a time, lane = consecutive key: coalesced loads,
  // and a candidate's rank among its wave's is a ballot and a popcount of the lower lanes (stable by construction)
  const int wbase = (int)(tile * COMPACT_TILE) + w * (COMPACT_ITEMS * 64);
  uint32_t key[COMPACT_ITEMS];
#pragma unroll
  for (int r = 0; r < COMPACT_ITEMS; r++) {
    const int i = wbase + 64 * r + lane;
    key[r] = i < P ? keys[i] : 0xFFFFFFFFu;
  }
  auto is_near = [&](const int r) { return wbase + 64 * r + lane < P && (key[r] >> 24) < top_end; };
  uint32_t wcount = 0;  // wave-uniform
#pragma unroll
  for (int r = 0; r < COMPACT_ITEMS; r++) wcount += (uint32_t)__popcll(__ballot(is_near(r)));
  if (lane == 0) wtot[w] = wcount;
  __syncthreads();
  const uint32_t agg = wtot[0] + wtot[1] + wtot[2] + wtot[3];
  if (w == 0) {
    uint32_t prefix = 0;
    if (tile == 0) {
      if (lane == 0) __hip_atomic_store(st, SC_GLOBAL | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0) __hip_atomic_store(st + tile, SC_LOCAL | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int base = (int)tile - 1;  // lane L inspects tile base - L; beyond tile 0 counts as a known prefix of 0
      for (;;) {
        const int t = base - lane;
        const unsigned long long v =
            t >= 0 ? __hip_atomic_load(st + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : SC_GLOBAL;
        const uint32_t flag = (uint32_t)(v >> 32);
        const uint64_t mg = __ballot(flag == 2u), mn = __ballot(flag == 0u);
        const int fg = mg ? __builtin_ctzll(mg) : 64;
        const uint64_t nearer = fg == 64 ? ~0ull : ((1ull << fg) - 1ull);
        if (mn & nearer) {  // a nearer tile has not published yet
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        prefix += wave_sum_u32(lane <= fg ? (uint32_t)v : 0u);
        if (fg < 64) break;
        base -= 64;
      }
      if (lane == 0)
        __hip_atomic_store(st + tile, SC_GLOBAL | (unsigned long long)(prefix + agg), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) s_prefix = prefix;
  }
  __syncthreads();
  if (tile == gridDim.x - 1u && tid == 0) {
    *n_out = s_prefix + agg;  // the candidates of the whole frame
    // (the host's copy: a thread whose near candidates turn out to be most of the scene sorts everything up front again)
    if (publish)
      __hip_atomic_store(publish, ((unsigned long long)frame_ticket << 32) | (s_prefix + agg), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (agg == 0u) return;  // (workgroup-uniform: nothing to write or count)
  uint32_t pos = s_prefix;
  for (int k = 0; k < w; k++) pos += wtot[k];
  if (wcount) {
#pragma unroll
    for (int r = 0; r < COMPACT_ITEMS; r++) {
      const bool take = is_near(r);
      const uint64_t m = __ballot(take);
      if (take) {
        const uint32_t p = pos + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        keys_out[p] = key[r];
        vals_out[p] = (uint32_t)(wbase + 64 * r + lane);  // (the id: k_preprocess numbers the pairs 0 .. P-1)
#pragma unroll
        for (int d = 0; d < 4; d++) atomicAdd(&dh[d][(key[r] >> (8 * d)) & 255u], 1u);
      }
      pos += (uint32_t)__popcll(m);
    }
  }
  __syncthreads();
#pragma unroll
  for (int d = 0; d < 4; d++) {
    const uint32_t c = dh[d][tid];
    if (c) (void)__hip_atomic_fetch_add(ghist_near + d * 256 + tid, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// `budget` < 0xFFFFFFFF (near/far frames, api.hip): only the NEAR Gaussians -- those whose first slot lies below the
// budget -- get descriptors, slots and chunk entries; the Gaussian in whose run the budget falls ends the near phase:
// it publishes the phase's instance count (total[6]) and the depth-order index of the first far Gaussian (total[7]),
// and the scan tile it sits in (total[11]): scan tiles beyond it have nothing to do and leave at once (they publish a
// saturated prefix so that the look-back of still later tiles ends there).
__global__ __launch_bounds__(PRE_BLOCK) void k_scan_offsets(const FrameParams fp, GeomState g, const Count cnt,
                                                            uint32_t* __restrict__ chunk_first,
                                                            uint2* __restrict__ ranges, uint2* __restrict__ rangesB,
                                                            uint32_t* __restrict__ counts0, const size_t ncounts0,
                                                            const uint32_t budget,
                                                            unsigned long long* __restrict__ publish_near,
                                                            const uint32_t ticket,
                                                            const uint32_t* __restrict__ top_hist,
                                                            const uint32_t* __restrict__ order,
                                                            const bool order_is_near_list) {
  __shared__ uint32_t wtot[PRE_BLOCK / 64];
  __shared__ uint32_t s_tile, s_prefix, s_limit;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const bool split = budget != 0xFFFFFFFFu;
  // R: the count this chain works with.  Whole frame: min(num_rendered, capacity) (slots beyond the capacity are never
  // emitted, api.hip).  Near phase: `cnt` points at the near count this very kernel publishes, so the frame's total
  // comes from total[0].
  const uint32_t R_total = split ? g.total[0] : (uint32_t)cnt.get();
  const int R = (int)R_total;
  if (blockIdx.x == 0 && tid == 0) {
    g.total[12] = split ? 1u : 0u;  // marks the frame for the views (api.hip)
    g.total[8] = 0u;                // far-phase totals: a frame whose far chain is never enqueued has none
    g.total[9] = 0u;
    g.total[10] = 0u;
  }
  // side job: (0, 0) for the tiles no instance lands in
  for (int t = blockIdx.x * PRE_BLOCK + tid; t < fp.gx * fp.gy; t += gridDim.x * PRE_BLOCK) {
    ranges[t] = make_uint2(0u, 0u);
    rangesB[t] = make_uint2(0u, 0u);
  }
  // side job: clear the digit counts of the tile sort's first pass (k_emit accumulates them while emitting)
  for (size_t q = (size_t)blockIdx.x * PRE_BLOCK + tid; q < ncounts0; q += (size_t)gridDim.x * PRE_BLOCK) counts0[q] = 0u;
  if (tid == 0) s_tile = atomicAdd(g.dsort.tickets() + 4, 1u);  // every lower tile is already running
  if (w == 0) {
    // Near phase: how far into the depth order can the budget reach?  k_preprocess left, per TOP BYTE of the depth keys,
    // the number of Gaussians (top_hist[0..255]) and their tile counts (top_hist[256..511]): the budget falls into the
    // first byte value at which the running tile count reaches it, so no Gaussian beyond that value's last index is
    // needed and the scan tiles behind it leave without touching memory (at 2 M Gaussians / 1080p: 13 of 489 tiles stay).
    uint32_t limit = 0xFFFFFFFFu, top_end = 256u;
    if (split && top_hist) wave_near_limit(top_hist, budget, lane, limit, top_end);  // (top-byte granularity)
    if (lane == 0) s_limit = limit;
  }
  __syncthreads();
  const uint32_t tile = s_tile;
  if (split) {
    const uint32_t crossed = __hip_atomic_load(g.total + 11, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((crossed != 0u && tile + 1u > crossed) ||    // the near phase ended in an earlier tile
        (unsigned long long)tile * SCAN_TILE >= s_limit) {  // ... or must end before this one
      if (tid == 0)
        __hip_atomic_store(g.dsort.scan_status() + tile, SC_GLOBAL | 0xFFFFFFFFull, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
  const int i0 = (int)(tile * SCAN_TILE) + tid * SCAN_ITEMS;
  uint32_t id[SCAN_ITEMS], n[SCAN_ITEMS], rect[SCAN_ITEMS];
  // (`order` is the full depth order, or -- partial depth sort -- the sorted near candidates: its first `limit` entries)
  const int nvalid = order_is_near_list && s_limit < (uint32_t)fp.P ? (int)s_limit : fp.P;
  if (i0 + SCAN_ITEMS <= nvalid) {
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS / 4; q++) {
      const uint4 o = *reinterpret_cast<const uint4*>(order + i0 + 4 * q);
      id[4 * q] = o.x; id[4 * q + 1] = o.y; id[4 * q + 2] = o.z; id[4 * q + 3] = o.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) id[k] = i0 + k < nvalid ? order[i0 + k] : 0xFFFFFFFFu;
  }
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    const uint2 gp = id[k] != 0xFFFFFFFFu ? g.gpack[id[k]] : make_uint2(0u, 0u);
    n[k] = gp.x;
    rect[k] = gp.y;
    sum += gp.x;
  }
  const uint32_t inc = wave_incl_scan_u32(sum, lane);
  if (lane == 63) wtot[w] = inc;
  __syncthreads();
  const uint32_t agg = wtot[0] + wtot[1] + wtot[2] + wtot[3];
  if (w == 0) {
    unsigned long long* st = g.dsort.scan_status();
    uint32_t prefix = 0;
    if (tile == 0) {
      if (lane == 0) __hip_atomic_store(st, SC_GLOBAL | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0) __hip_atomic_store(st + tile, SC_LOCAL | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int base = (int)tile - 1;  // lane L inspects tile base - L; beyond tile 0 counts as a known prefix of 0
      for (;;) {
        const int t = base - lane;
        const unsigned long long v =
            t >= 0 ? __hip_atomic_load(st + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : SC_GLOBAL;
        const uint32_t flag = (uint32_t)(v >> 32);
        const uint64_t mg = __ballot(flag == 2u), mn = __ballot(flag == 0u);
        const int fg = mg ? __builtin_ctzll(mg) : 64;  // nearest tile whose inclusive prefix is known
        const uint64_t nearer = fg == 64 ? ~0ull : ((1ull << fg) - 1ull);
        if (mn & nearer) {  // a nearer tile has not published yet
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        {  // saturating: a tile that left early published 0xFFFFFFFF ("beyond the near budget")
          const unsigned long long add = wave_sum_u64(lane <= fg ? (unsigned long long)(uint32_t)v : 0ull) + prefix;
          prefix = add > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)add;
        }
        if (fg < 64) break;
        base -= 64;
      }
      if (lane == 0) {
        const unsigned long long inc_all = (unsigned long long)prefix + agg;
        __hip_atomic_store(st + tile, SC_GLOBAL | (inc_all > 0xFFFFFFFFull ? 0xFFFFFFFFull : inc_all), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (lane == 0) s_prefix = prefix;
  }
  __syncthreads();
  if (split && s_prefix >= budget) return;  // the whole tile lies beyond the near phase (status already published)
  uint32_t off = s_prefix + inc - sum;  // exclusive offset of this thread's first Gaussian
  for (int k = 0; k < w; k++) off += wtot[k];
  if (i0 >= fp.P) return;
  uint32_t offs[SCAN_ITEMS], inv[SCAN_ITEMS];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    offs[k] = off;
    const uint32_t rw = rect[k] >> 20;
    inv[k] = n[k] ? 0xFFFFFFFFu / rw + 1u : 0u;  // ceil(2^32 / rw) for rw > 1 (wraps to 0 for rw == 1: k_emit)
    if (i0 + k < fp.P && n[k] && off < budget) {
      const uint32_t end = off + n[k];
      g.slotinfo[id[k]] = make_uint2(off, rect[k]);
      // this chain's last slot: the frame's (clamped to the capacity in an overflowed speculative frame) or, in a
      // near/far frame, the end of the run the budget falls into
      const bool last_near = split && (end >= budget || end == R_total);
      const uint32_t Rc = split ? (last_near ? end : 0xFFFFFFFFu) : (uint32_t)R;
      const uint32_t end_c = end < Rc ? end : Rc;
      for (uint32_t c = (off + EMIT_CHUNK - 1) / EMIT_CHUNK; (unsigned long long)c * EMIT_CHUNK < end_c; c++)
        chunk_first[c] = (uint32_t)(i0 + k);
      if (split ? last_near : (off < (uint32_t)R && end >= (uint32_t)R))
        chunk_first[(size_t)(((split ? end : (uint32_t)R) + EMIT_CHUNK - 1) / EMIT_CHUNK)] = (uint32_t)(i0 + k);
      if (last_near) {  // exactly one Gaussian of the frame
        g.total[6] = end;
        g.total[7] = (uint32_t)(i0 + k) + 1u;
        g.sdesc[i0 + k + 1] = make_uint4(end, 0u, 0u, 0u);  // sentinel behind the last near descriptor (.x is what counts)
        __hip_atomic_store(g.total + 11, tile + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (publish_near)  // the host's copy of the near count (statistics only: the near capacity cannot overflow)
          __hip_atomic_store(publish_near, ((unsigned long long)ticket << 32) | end, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    off += n[k];
  }
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++)
    if (i0 + k < fp.P && offs[k] < budget)
      g.sdesc[i0 + k] = make_uint4(offs[k], id[k], rect[k], inv[k]);  // 256 contiguous bytes per thread
  if (!split && i0 + SCAN_ITEMS >= fp.P) g.sdesc[fp.P] = make_uint4(off, 0u, 0u, 0u);  // sentinel: .x = R
  if (split && R_total == 0u && i0 == 0) {  // nothing to bin at all
    g.total[6] = 0u;
    g.total[7] = 0u;
    if (publish_near)
      __hip_atomic_store(publish_near, (unsigned long long)ticket << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Far phase of a near/far frame (api.hip).  The near phase has been binned and blended; `sat` is the summed-area table
// of the tiles that still have an unfinished pixel.  A far Gaussian (depth-order index >= total[7]) is emitted -- with
// its whole tile rectangle, so that every slot rule of the near phase holds -- iff its rectangle contains a live tile:
// a tile that is finished can receive nothing from it (every pixel has stopped, forward.cu:380-383), exactly as if
// its instances sat in the part of the list the reference never reads.  One launch, look-back scan over the same
// 4096-Gaussian tiles of the depth order with a (slots, emitted Gaussians) pair in the 64-bit status word
// [flag:2 | Gaussians:30 | slots:32]; the emitted Gaussians' descriptors are written COMPACTED (sdescB) because the
// emitters stage the descriptors between two chunk boundaries in LDS and assume that each owns at least one slot.
// Slots are numbered from `slot_base` (the near phase's capacity) in the frame's one slot space, so gradient records,
// flags and slotinfo need no notion of a phase.  The last tile publishes the phase's totals (total[8], total[10],
// the sentinel, and the host's second mailbox word).
constexpr unsigned long long SF_GLOBAL = 2ull << 62, SF_LOCAL = 1ull << 62, SF_MASK = (1ull << 62) - 1ull;

__global__ __launch_bounds__(PRE_BLOCK) void k_scan_offsets_far(const FrameParams fp, GeomState g, const Count capB_,
                                                                const uint32_t slot_base,
                                                                const uint32_t* __restrict__ sat,
                                                                uint32_t* __restrict__ chunk_firstB,
                                                                uint32_t* __restrict__ counts0, const size_t ncounts0,
                                                                unsigned long long* __restrict__ publish,
                                                                const uint32_t ticket) {
  __shared__ unsigned long long wtot[PRE_BLOCK / 64];
  __shared__ uint32_t s_tile;
  __shared__ unsigned long long s_prefix;
  if (capB_.closed()) return;  // (asynchronous frame that needs no far chain)
  const int capB = capB_.cap;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // side job: clear the digit counts of the far tile sort's first pass
  for (size_t q = (size_t)blockIdx.x * PRE_BLOCK + tid; q < ncounts0; q += (size_t)gridDim.x * PRE_BLOCK) counts0[q] = 0u;
  if (tid == 0) s_tile = atomicAdd(g.dsort.tickets() + 5, 1u);
  __syncthreads();
  const uint32_t tile = s_tile;
  const uint32_t ntiles = gridDim.x;
  const uint32_t iA = g.total[7], live_tiles = g.total[9];
  unsigned long long* st = g.dsort.scanB_status();
  const int i0 = (int)(tile * SCAN_TILE) + tid * SCAN_ITEMS;
  const bool idle = live_tiles == 0u || (uint32_t)(tile + 1u) * SCAN_TILE <= iA;  // workgroup-uniform
  uint32_t id[SCAN_ITEMS], n[SCAN_ITEMS], rect[SCAN_ITEMS];
  unsigned long long sum = 0ull;  // (emitted Gaussians << 32) | slots of this thread's items
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) { id[k] = 0xFFFFFFFFu; n[k] = 0u; rect[k] = 0u; }
  if (!idle) {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++)
      if (i0 + k < fp.P && (uint32_t)(i0 + k) >= iA) id[k] = g.order[i0 + k];
    const int sw = fp.gx + 1;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
      if (id[k] == 0xFFFFFFFFu) continue;
      const uint2 gp = g.gpack[id[k]];
      if (!gp.x) continue;
      const int x0 = (int)(gp.y & 1023u), y0 = (int)((gp.y >> 10) & 1023u), rw = (int)(gp.y >> 20);
      const int x1 = x0 + rw, y1 = y0 + (int)(gp.x / (uint32_t)rw);
      const uint32_t live = sat[y1 * sw + x1] - sat[y0 * sw + x1] - sat[y1 * sw + x0] + sat[y0 * sw + x0];
      if (live) {
        n[k] = gp.x;
        rect[k] = gp.y;
        sum += (1ull << 32) | gp.x;
      }
    }
  }
  // inclusive scan of the packed pair over the workgroup (slots < 2^31 per frame: no carry into the upper half)
  unsigned long long inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wtot[w] = inc;
  __syncthreads();
  const unsigned long long agg = wtot[0] + wtot[1] + wtot[2] + wtot[3];
  auto pack = [](unsigned long long v) { return ((v >> 32) << 32) | (v & 0xFFFFFFFFull); };  // identity (documentation)
  if (w == 0) {
    unsigned long long prefix = 0ull;
    if (tile == 0) {
      if (lane == 0) __hip_atomic_store(st, SF_GLOBAL | pack(agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0) __hip_atomic_store(st + tile, SF_LOCAL | pack(agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int base = (int)tile - 1;
      for (;;) {
        const int t = base - lane;
        const unsigned long long v =
            t >= 0 ? __hip_atomic_load(st + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : SF_GLOBAL;
        const uint32_t flag = (uint32_t)(v >> 62);
        const uint64_t mg = __ballot(flag == 2u), mn = __ballot(flag == 0u);
        const int fg = mg ? __builtin_ctzll(mg) : 64;
        const uint64_t nearer = fg == 64 ? ~0ull : ((1ull << fg) - 1ull);
        if (mn & nearer) {
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        prefix += wave_sum_u64(lane <= fg ? (v & SF_MASK) : 0ull);
        if (fg < 64) break;
        base -= 64;
      }
      if (lane == 0)
        __hip_atomic_store(st + tile, SF_GLOBAL | (prefix + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) s_prefix = prefix;
  }
  __syncthreads();
  unsigned long long run = s_prefix + inc - sum;  // exclusive (Gaussians, slots) before this thread's first item
  for (int k = 0; k < w; k++) run += wtot[k];
  if (tile == ntiles - 1u && tid == PRE_BLOCK - 1) {  // the frame's last item: totals of the far phase
    const unsigned long long tot = s_prefix + agg;
    const uint32_t RB = (uint32_t)(tot & 0xFFFFFFFFull), nG = (uint32_t)(tot >> 32);
    g.total[8] = RB;
    g.total[10] = nG;
    g.sdescB[nG] = make_uint4(RB, 0u, 0u, 0u);  // sentinel: .x = the far phase's instance count
    if (nG && RB <= (uint32_t)capB) chunk_firstB[(size_t)((RB + EMIT_CHUNK - 1) / EMIT_CHUNK)] = nG - 1u;  // (as k_scan_offsets)
    if (publish)
      __hip_atomic_store(publish, ((unsigned long long)ticket << 32) | RB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (idle) return;
  const uint32_t RB_cap = (uint32_t)capB;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    if (!n[k]) continue;
    const uint32_t off = (uint32_t)(run & 0xFFFFFFFFull), j = (uint32_t)(run >> 32);
    const uint32_t rw = rect[k] >> 20;
    const uint32_t end = off + n[k];
    g.sdescB[j] = make_uint4(off, id[k], rect[k], 0xFFFFFFFFu / rw + 1u);
    g.slotinfo[id[k]] = make_uint2(slot_base + off, rect[k]);
    // (a frame whose far count exceeds the capacity is discarded by the host: keep the chunk table in bounds)
    const uint32_t end_c = end < RB_cap ? end : RB_cap;
    for (uint32_t c = (off + EMIT_CHUNK - 1) / EMIT_CHUNK; (unsigned long long)c * EMIT_CHUNK < end_c; c++)
      chunk_firstB[c] = j;
    if (off < RB_cap && end > RB_cap)  // overflowed frame: the emitters stop at the capacity, inside this run
      chunk_firstB[(size_t)((RB_cap + EMIT_CHUNK - 1) / EMIT_CHUNK)] = j;
    run += (1ull << 32) | n[k];
  }
}

// Staging and slot walk of the emitters, as device functions (k_emit_scatter runs them twice per workgroup).
struct EmitStage {
  uint32_t s_off[EMIT_CHUNK + 2], s_id[EMIT_CHUNK + 1], s_rect[EMIT_CHUNK + 1], s_inv[EMIT_CHUNK + 1];
};

// descriptors of the Gaussians covering slots [c0, c1) of emit chunk `e` -> LDS; returns their count S
__device__ __forceinline__ int emit_stage(const uint4* __restrict__ sdesc, const uint32_t* __restrict__ chunk_first, int e,
                                          uint32_t c1, int R, int tid, EmitStage& st) {
  const int i0 = (int)chunk_first[e];
  int i1 = (int)chunk_first[e + 1];
  if (c1 < (uint32_t)R && sdesc[i1].x >= c1) i1--;    // the Gaussian covering slot c1 starts exactly there
  const int S = i1 - i0 + 1;                          // <= EMIT_CHUNK + 1: every staged Gaussian owns >= 1 slot
  for (int j = tid; j <= S; j += 256) {               // s_off[S] = start of the first run beyond this chunk
    const uint4 d = sdesc[i0 + j];                    // (the descriptor arrays have P + 1 entries)
    st.s_off[j] = d.x;
    if (j < S) {
      st.s_id[j] = d.y;
      st.s_rect[j] = d.z;
      st.s_inv[j] = d.w;
    }
  }
  return S;
}

// (tile id, Gaussian id) of the eight slots [t0, t0 + 8); those >= c1 are meaningless
__device__ __forceinline__ void emit_walk8(uint32_t t0, uint32_t c1, int S, const EmitStage& st, uint32_t gx,
                                           uint32_t tk[8], uint32_t iv[8]) {
  int lo = 0, hi = S - 1;  // largest j with s_off[j] <= t0
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (st.s_off[mid] <= t0) lo = mid; else hi = mid - 1;
  }
  int j = lo;
  uint32_t rc = st.s_rect[j], rw = rc >> 20, id = st.s_id[j], next = st.s_off[j + 1];
  const uint32_t local = t0 - st.s_off[j];
  // local < 2^20, rw < 2^10  =>  local * (inv*rw - 2^32) < 2^32: the multiply-high quotient is exact
  uint32_t row = rw == 1u ? local : __umulhi(local, st.s_inv[j]);
  uint32_t col = local - row * rw;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    tk[k] = (((rc >> 10) & 1023u) + row) * gx + (rc & 1023u) + col;
    iv[k] = id;
    const uint32_t t = t0 + (uint32_t)k + 1u;
    if (t == next && t < c1) {  // run finished: next Gaussian (every staged Gaussian owns >= 1 slot)
      j++;
      rc = st.s_rect[j]; rw = rc >> 20; id = st.s_id[j]; next = st.s_off[j + 1];
      row = 0; col = 0;
    } else if (++col == rw) {
      col = 0;
      row++;
    }
  }
}

// Emission fused with the FIRST pass of the tile sort (16-bit keys): a workgroup generates the 4096 pairs of one
// sort tile (two emit chunks), brings them into scatter order through LDS and scatters them by their low digit --
// the unsorted pairs are never written to or read back from HBM (2 x 6 bytes per instance).  The digit counts the
// scatter needs were accumulated by the count-only emitter (k_emit<K, false>) and scanned in between.
template <bool ARANK>
__global__ __launch_bounds__(256) void k_emit_scatter(const FrameParams fp, const uint4* __restrict__ sdesc, const Count cnt,
                                                      const uint32_t* __restrict__ chunk_first,
                                                      uint16_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                      const int shift0, const int nbits0,
                                                      const uint32_t* __restrict__ counts,
                                                      const uint32_t* __restrict__ chunk_base,
                                                      const uint32_t* __restrict__ digit_total) {
  static_assert(TSORT_TILE == 2 * EMIT_CHUNK && TSORT_WAVES == 4, "one sort tile = two emit chunks on 256 threads");
  constexpr int NSTEP = TSORT_TILE / TSORT_WAVES / 64;
  union SMem {
    EmitStage st;                                         // while generating
    ScatterLds<uint16_t, TSORT_WAVES, TSORT_TILE> L;      // afterwards
  };
  __shared__ SMem sm;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (cnt.closed()) return;
  const int R = cnt.get();
  for_each_unit(units_of(R, TSORT_TILE), [&](const int tile) {
  uint32_t tk[2][8], iv[2][8];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int e = 2 * tile + r;
    const uint32_t c0 = (uint32_t)e * EMIT_CHUNK;
    const bool live = c0 < (uint32_t)R;  // workgroup-uniform
    const uint32_t c1 = c0 + EMIT_CHUNK < (uint32_t)R ? c0 + EMIT_CHUNK : (uint32_t)R;
    int S = 0;
    if (live) S = emit_stage(sdesc, chunk_first, e, c1, R, tid, sm.st);
    __syncthreads();
    const uint32_t t0 = c0 + (uint32_t)tid * 8u;
    if (live && t0 < c1) emit_walk8(t0, c1, S, sm.st, (uint32_t)fp.gx, tk[r], iv[r]);
    __syncthreads();
  }
  // slot order -> LDS (the staging arrays are dead now)
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const uint32_t sl = (uint32_t)r * EMIT_CHUNK + (uint32_t)tid * 8u;  // slot inside the tile
    const size_t t0 = (size_t)tile * TSORT_TILE + sl;
    if (t0 + 8 <= (size_t)R) {
      *reinterpret_cast<uint4*>(sm.L.lkey + sl) = make_uint4(tk[r][0] | (tk[r][1] << 16), tk[r][2] | (tk[r][3] << 16),
                                                             tk[r][4] | (tk[r][5] << 16), tk[r][6] | (tk[r][7] << 16));
      *reinterpret_cast<uint4*>(sm.L.lval + sl) = make_uint4(iv[r][0], iv[r][1], iv[r][2], iv[r][3]);
      *reinterpret_cast<uint4*>(sm.L.lval + sl + 4) = make_uint4(iv[r][4], iv[r][5], iv[r][6], iv[r][7]);
    } else {
      for (int k = 0; k < 8 && t0 + k < (size_t)R; k++) {
        sm.L.lkey[sl + k] = (uint16_t)tk[r][k];
        sm.L.lval[sl + k] = iv[r][k];
      }
    }
  }
  __syncthreads();
  // scatter order -> registers
  uint32_t key[NSTEP], val[NSTEP];
#pragma unroll
  for (int s = 0; s < NSTEP; s++) {
    const uint32_t p = (uint32_t)(w * (TSORT_TILE / TSORT_WAVES) + s * 64 + lane);
    const bool valid = (size_t)tile * TSORT_TILE + p < (size_t)R;
    key[s] = valid ? (uint32_t)sm.L.lkey[p] : 0u;
    val[s] = valid ? sm.L.lval[p] : 0u;
  }
  __syncthreads();
  scatter_core<uint16_t, false, ARANK, TSORT_WAVES, TSORT_TILE>(sm.L, key, val, tile, keys_out, vals_out, R, shift0, nbits0,
                                                                counts, chunk_base, digit_total, nullptr);
  });
}

// STORE = false: count-only emitter in front of k_emit_scatter (digit counts + inst_flag reset, no pair stores)
template <typename K, bool STORE>
__global__ __launch_bounds__(256) void k_emit(const FrameParams fp, const uint4* __restrict__ sdesc, const Count cnt,
                                              const uint32_t* __restrict__ chunk_first,
                                              K* __restrict__ tkeys_out, uint32_t* __restrict__ ivals_out,
                                              uint8_t* __restrict__ inst_flag, uint32_t* __restrict__ counts0,
                                              const uint32_t digit_shift0, const uint32_t digit_mask0) {
  __shared__ EmitStage st;
  __shared__ uint32_t hist[256];  // digit counts of the tile sort's FIRST pass for this workgroup's 2048 slots
  const int tid = threadIdx.x;
  if (cnt.closed()) return;
  const int R = cnt.get();
  for_each_unit(units_of(R, EMIT_CHUNK), [&](const int chunk) {
  hist[tid] = 0;
  const uint32_t c0 = (uint32_t)chunk * EMIT_CHUNK;
  const uint32_t c1 = c0 + EMIT_CHUNK < (uint32_t)R ? c0 + EMIT_CHUNK : (uint32_t)R;
  const int S = emit_stage(sdesc, chunk_first, chunk, c1, R, tid, st);
  __syncthreads();
  // Each thread owns EIGHT consecutive slots: one bisection for the first, then it walks (row, col) and steps to
  // the next Gaussian when a run ends -- 4x fewer LDS round trips than a search per slot, 32-byte stores.
  const uint32_t t0 = c0 + (uint32_t)tid * 8u;
  if (t0 < c1) {
  uint32_t tk[8], iv[8];
  emit_walk8(t0, c1, S, st, (uint32_t)fp.gx, tk, iv);
  if (!STORE) {
    if (t0 + 8u <= c1) {
      *reinterpret_cast<uint2*>(inst_flag + t0) = make_uint2(0u, 0u);
    } else {
      for (int k = 0; k < 8 && t0 + (uint32_t)k < c1; k++) inst_flag[t0 + k] = 0;
    }
  } else if (t0 + 8u <= c1) {
    uint4* vo = reinterpret_cast<uint4*>(ivals_out + t0);
    if (sizeof(K) == 2) {  // eight 16-bit tile ids = one 16-byte store
      *reinterpret_cast<uint4*>(tkeys_out + t0) = make_uint4(tk[0] | (tk[1] << 16), tk[2] | (tk[3] << 16),
                                                             tk[4] | (tk[5] << 16), tk[6] | (tk[7] << 16));
    } else {
      uint4* ko = reinterpret_cast<uint4*>(tkeys_out + t0);
      ko[0] = make_uint4(tk[0], tk[1], tk[2], tk[3]);
      ko[1] = make_uint4(tk[4], tk[5], tk[6], tk[7]);
    }
    vo[0] = make_uint4(iv[0], iv[1], iv[2], iv[3]);
    vo[1] = make_uint4(iv[4], iv[5], iv[6], iv[7]);
    *reinterpret_cast<uint2*>(inst_flag + t0) = make_uint2(0u, 0u);  // no gradient record in these slots yet
  } else {
    for (int k = 0; k < 8 && t0 + (uint32_t)k < c1; k++) {
      tkeys_out[t0 + k] = (K)tk[k];
      ivals_out[t0 + k] = iv[k];
      inst_flag[t0 + k] = 0;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (t0 + (uint32_t)k < c1) atomicAdd(&hist[(tk[k] >> digit_shift0) & digit_mask0], 1u);
  }
  // The sort's first pass needs digit counts per sort tile: several emitting workgroups share one, so each adds
  // its non-zero bins (k_scan_offsets cleared the array) -- this replaces a histogram pass over all the keys.
  __syncthreads();
  const uint32_t hc = hist[tid];
  if (hc) atomicAdd(&counts0[(size_t)(c0 / TSORT_TILE) * 256 + tid], hc);
  });
}

// ------------------------------------------------------------------------------------------------
// Backward gather.  The blend backward left one 9-float record per (tile, Gaussian) instance it
// touched, in the Gaussian's slot run, and flagged the Gaussian.  k_compact_touched lists the flagged
// Gaussians; k_gather_records gives each ONE WAVE: lanes stride over the run's flag bytes (coalesced),
// load the flagged records, and a fixed-tree wave reduction yields the nine sums -- every Gaussian is
// an independent unit of work (some near ones own >1000 records, most own none), and the result is
// bitwise reproducible because each sum has a fixed association order.  Replaces the 9 atomicAdds per
// (pixel, Gaussian) of the reference (backward.cu:565, 591-600).
// ------------------------------------------------------------------------------------------------
// Every thread owns sixteen flag bytes (one 16-byte load) and a workgroup appends its ids with ONE atomicAdd
// (same-address atomics retire one at a time in L2: a few hundred of them, not one per wave).  The list order
// is arbitrary; every Gaussian is gathered independently, so the results do not depend on it.
__global__ __launch_bounds__(PRE_BLOCK) void k_compact_touched(const int P, const uint8_t* __restrict__ touched,
                                                               uint32_t* __restrict__ list,
                                                               uint32_t* __restrict__ count) {
  __shared__ uint32_t wtot[PRE_BLOCK / 64];
  __shared__ uint32_t wg_base;
  const int idx0 = (blockIdx.x * PRE_BLOCK + threadIdx.x) * 16;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t f[4] = {0u, 0u, 0u, 0u};
  if (idx0 + 15 < P) {
    const uint4 v = *reinterpret_cast<const uint4*>(touched + idx0);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
    for (int k = 0; k < 16 && idx0 + k < P; k++) f[k >> 2] |= (uint32_t)touched[idx0 + k] << (8 * (k & 3));
  }
  uint32_t c = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) c += ((f[k >> 2] >> (8 * (k & 3))) & 0xFFu) ? 1u : 0u;
  const uint32_t inc = wave_incl_scan_u32(c, lane);
  if (lane == 63) wtot[w] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t tot = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    wg_base = tot ? atomicAdd(count, tot) : 0u;
  }
  __syncthreads();
  if (!c) return;
  uint32_t base = wg_base + inc - c;
  for (int k = 0; k < w; k++) base += wtot[k];
#pragma unroll
  for (int k = 0; k < 16; k++)
    if ((f[k >> 2] >> (8 * (k & 3))) & 0xFFu) list[base++] = (uint32_t)(idx0 + k);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get_(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float row_allsum_(float v) {
  v += dpp_get_<0xB1, 0xF>(v);
  v += dpp_get_<0x4E, 0xF>(v);
  v += dpp_get_<0x124, 0xF>(v);
  v += dpp_get_<0x128, 0xF>(v);
  return v;
}
__device__ __forceinline__ void swap_add32_(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap_add16_(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// (the register allocator settles on 102 VGPRs = four waves per SIMD for k_gather_records; asked for 96 it finds 86 without a spill)
#ifndef GSR_GATHER_ATTR
#define GSR_GATHER_ATTR __attribute__((amdgpu_num_vgpr(96)))
#endif
// the nine sums of one Gaussian's records, per lane (lane = slot mod 64)
struct RecordSums {
  float v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0, v6 = 0, v7 = 0, v8 = 0;
};
// Slots [base, base + 512) of the run [first, first + n): the eight flags of a lane are requested at once, then its
// flagged records in two groups of four, summed in slot order (three round trips for 512 slots; 1024 slots at a time
// with sixteen flags measured slower: registers).
constexpr uint32_t GATHER_CHUNK = 512;
constexpr uint32_t GATHER_SHARED_FROM = 2048;  // runs longer than this are shared by the workgroup's waves
__device__ __forceinline__ void gather_chunk(const size_t first, const uint32_t n, const uint32_t base, const int lane,
                                             const float4* __restrict__ grad_inst, uint8_t* __restrict__ inst_flag,
                                             RecordSums& r) {
  uint8_t f[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t k = base + 64 * j + lane;
    f[j] = k < n ? inst_flag[first + k] : (uint8_t)0;
  }
#pragma unroll
  for (int q = 0; q < 2; q++) {
    if (base + 256u * q >= n) break;  // (wave-uniform)
    float4 a[4], b[4], c[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (f[4 * q + j]) {
        const size_t slot = first + base + 64 * (4 * q + j) + lane;
        inst_flag[slot] = 0;  // consumed: the blobs are clean for another backward
        a[j] = grad_inst[slot * GRAD_F4 + 0];
        b[j] = grad_inst[slot * GRAD_F4 + 1];
        c[j] = grad_inst[slot * GRAD_F4 + 2];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (f[4 * q + j]) {
        r.v0 += a[j].x; r.v1 += a[j].y; r.v2 += a[j].z; r.v3 += a[j].w;
        r.v4 += b[j].x; r.v5 += b[j].y; r.v6 += b[j].z; r.v7 += b[j].w;
        r.v8 += (c[j].x + c[j].y) + (c[j].z + c[j].w);  // the four 16-lane-row sums of dLG the tile kernel leaves
      }
    }
  }
}
// Wave reduction of the per-lane sums and the Gaussian's four gradient rows.  The records hold RAW pixel sums (colour
// r g b | S3 = sum dLG dx, S4 = sum dLG dy | S5 = sum dLG dx^2, S6 = sum dLG dx dy, S7 = sum dLG dy^2 | sum dLG),
// dLG = G dL/dalpha.  The factors every pixel and every instance of the Gaussian share are applied here, ONCE per
// Gaussian (backward.cu:561-562, 583-597):
//   dL/dmean2D = -(conic (S3, S4)) * opacity * (W/2, H/2),  dL/dconic = -1/2 opacity (S5, S6, S7),  dL/dopacity = sum dLG
__device__ __forceinline__ void gather_finish(RecordSums r, const uint32_t id, const int lane, const float cx, const float cy,
                                              const float cz, const float op, const float ddelx_dx, const float ddely_dy,
                                              float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic,
                                              float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolor) {
  // (v0,v4) (v1,v5) (v2,v6) (v3,v7) across half-waves, then rows, then inside rows
  swap_add32_(r.v0, r.v4);
  swap_add32_(r.v1, r.v5);
  swap_add32_(r.v2, r.v6);
  swap_add32_(r.v3, r.v7);
  swap_add16_(r.v0, r.v2);  // rows hold v0, v2, v4, v6
  swap_add16_(r.v1, r.v3);  // rows hold v1, v3, v5, v7
  const float w0 = row_allsum_(r.v0), w1 = row_allsum_(r.v1);
  float v8 = row_allsum_(r.v8);
  v8 += dpp_get_<0x142, 0xA>(v8);  // row_bcast:15
  v8 += dpp_get_<0x143, 0xC>(v8);  // row_bcast:31 -> total in lane 63
  const float S3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w1), 16));
  const float S4 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w0), 32));
  const float mc = -0.5f * op;
  if (lane == 0) { dL_dcolor[3 * id] = w0; dL_dcolor[3 * id + 1] = w1; }
  if (lane == 16) { dL_dcolor[3 * id + 2] = w0; dL_dmean2D[3 * id] = -(cx * S3 + cy * S4) * (op * ddelx_dx); }
  if (lane == 32) { dL_dmean2D[3 * id + 1] = -(cz * S4 + cy * S3) * (op * ddely_dy); dL_dconic[4 * id] = w1 * mc; }
  if (lane == 48) { dL_dconic[4 * id + 1] = w0 * mc; dL_dconic[4 * id + 3] = w1 * mc; }
  if (lane == 63) dL_dopacity[id] = v8;
}

__global__ __launch_bounds__(PRE_BLOCK) GSR_GATHER_ATTR void k_gather_records(
    GeomState g, const float4* __restrict__ grad_inst, uint8_t* __restrict__ inst_flag,
    float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity,
    float* __restrict__ dL_dcolor, const float ddelx_dx, const float ddely_dy, const int P, const bool from_descriptors) {
  constexpr int NW = PRE_BLOCK / 64;
  __shared__ uint32_t s_long[PRE_BLOCK];  // Gaussians of this round whose runs the workgroup's waves share
  __shared__ uint32_t s_nlong;
  __shared__ float s_part[NW][9][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t nwaves = gridDim.x * NW;
  // Which Gaussians have records: the list k_compact_touched made of the flagged ones -- or, in a near/far frame, simply
  // the Gaussians the frame emitted (a few per cent of the scene: the near chain's descriptors [0, total[7]) and the far
  // chain's [0, total[10])), each asked for its flag: no compaction launch over all P flags
  const uint32_t nA = from_descriptors ? g.total[7] : 0u;
  const uint32_t count = from_descriptors ? nA + g.total[10] : g.total[2];
  // A wave examines 64 candidates per round -- consecutive candidates go to different WORKGROUPS first, then to the
  // workgroups' other waves, then to the waves' other lanes: the Gaussians with the longest runs sit next to each other
  // at the front of the near chain -- and gathers those that have records one after the other.  From the touched list
  // every candidate has records; from the descriptors of a near/far frame whose far chain ran most do not (far Gaussians
  // are emitted with their whole rectangle as soon as ONE tile of it is live): asked one per wave iteration, the two
  // dependent loads of each rejected candidate cost a round trip of their own (0.09 instead of 0.04 ms at 2 M Gaussians /
  // 1080p).
  // A run of up to 2048 slots is one wave's job (512 slots at a time: flags, then the flagged records).  LONGER
  // runs -- a Gaussian that comes close to the camera plane of a turned view covers the whole image, 8160 slots at
  // 1080p: a few dozen of those, walked serially by the waves they fell to, WERE the kernel (80 us,
  // tools/gather_stats.py) -- go on the workgroup's list and are shared by its four waves, 512-slot chunks in turn, the
  // waves' per-lane sums added in wave order: a fixed association order for every Gaussian, so the backward stays
  // bitwise reproducible.  (count is the same in every wave: the barriers below are reached by all of them.)
  for (uint32_t base = 0; base < count; base += 64u * nwaves) {
    if (threadIdx.x == 0) s_nlong = 0u;
    __syncthreads();
    const uint32_t qc = base + ((uint32_t)lane * NW + (uint32_t)w) * gridDim.x + blockIdx.x;
    uint32_t cand = 0xFFFFFFFFu;
    if (qc < count) cand = !from_descriptors ? g.tlist[qc] : qc < nA ? g.sdesc[qc].y : g.sdescB[qc - nA].y;
    // (emitted, but no pixel took it: no record)
    const bool has = cand < (uint32_t)P && (!from_descriptors || g.touched[cand] != 0);
    // Every lane fetches what the gather of ITS candidate will need -- first slot, run length, the splat record's conic
    // and opacity -- in one round trip for the whole batch; the wave then picks them up with v_readlane instead of
    // starting each Gaussian with two dependent loads.
    uint32_t my_first = 0u, my_n = 0u;
    float4 my_ra = make_float4(0.f, 0.f, 0.f, 0.f), my_rb = my_ra;
    if (has) {
      my_first = g.slotinfo[cand].x;
      my_n = g.gpack[cand].x;
      my_ra = g.splats[(size_t)cand * SPLAT_F4 + 0];  // (x, y, conic.x, conic.y)
      my_rb = g.splats[(size_t)cand * SPLAT_F4 + 1];  // (conic.z, opacity, r, g)
    }
    const bool is_long = has && my_n > GATHER_SHARED_FROM;
    if (is_long) s_long[atomicAdd(&s_nlong, 1u)] = cand;  // (list order is arbitrary: every entry is gathered independently)
    for (uint64_t todo = __ballot(has && !is_long); todo != 0ull; todo &= todo - 1ull) {
      const int src = __builtin_ctzll(todo);
      const uint32_t id = (uint32_t)__builtin_amdgcn_readlane((int)cand, src);
      const size_t first = (uint32_t)__builtin_amdgcn_readlane((int)my_first, src);
      const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)my_n, src);
      RecordSums r;
      for (uint32_t cb = 0; cb < n; cb += GATHER_CHUNK) gather_chunk(first, n, cb, lane, grad_inst, inst_flag, r);
      gather_finish(r, id, lane, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_ra.z), src)),
                    __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_ra.w), src)),
                    __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_rb.x), src)),
                    __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_rb.y), src)), ddelx_dx, ddely_dy, dL_dmean2D,
                    dL_dconic, dL_dopacity, dL_dcolor);
    }
    __syncthreads();
    const uint32_t nlong = s_nlong;
    for (uint32_t i = 0; i < nlong; i++) {
      const uint32_t id = s_long[i];
      const size_t first = g.slotinfo[id].x;
      const uint32_t n = g.gpack[id].x;
      RecordSums r;
      for (uint32_t cb = (uint32_t)w * GATHER_CHUNK; cb < n; cb += NW * GATHER_CHUNK)
        gather_chunk(first, n, cb, lane, grad_inst, inst_flag, r);
      s_part[w][0][lane] = r.v0; s_part[w][1][lane] = r.v1; s_part[w][2][lane] = r.v2;
      s_part[w][3][lane] = r.v3; s_part[w][4][lane] = r.v4; s_part[w][5][lane] = r.v5;
      s_part[w][6][lane] = r.v6; s_part[w][7][lane] = r.v7; s_part[w][8][lane] = r.v8;
      __syncthreads();
      if (w == 0) {
        RecordSums t;
        float* tv[9] = {&t.v0, &t.v1, &t.v2, &t.v3, &t.v4, &t.v5, &t.v6, &t.v7, &t.v8};
#pragma unroll
        for (int k = 0; k < 9; k++) {
          float acc = s_part[0][k][lane];
#pragma unroll
          for (int ww = 1; ww < NW; ww++) acc += s_part[ww][k][lane];
          *tv[k] = acc;
        }
        const float4 ra = g.splats[(size_t)id * SPLAT_F4 + 0];
        const float4 rb = g.splats[(size_t)id * SPLAT_F4 + 1];
        gather_finish(t, id, lane, ra.z, ra.w, rb.x, rb.y, ddelx_dx, ddely_dy, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor);
      }
      __syncthreads();
    }
    __syncthreads();  // (every wave has read this round's s_nlong before thread 0 clears it for the next)
  }
}

// ------------------------------------------------------------------------------------------------
// B2 + B3, fused with the gradient gather.  One thread per Gaussian.  Replaces computeCov2DCUDA
// (reference backward.cu:140-275), preprocessCUDA backward (:371-435), computeColorFromSH backward
// (:20-135) and computeCov3D backward (:279-366), and the 9 atomicAdds per (pixel, Gaussian) of the
// reference's blend backward (:565,591-600): the blend backward left one 9-float record per
// (tile, Gaussian) instance in the Gaussian's contiguous slot run; they are summed here in slot
// order, which makes the whole backward bitwise reproducible.
// Every output element of Gaussian idx is written (zeros when radii <= 0).
// ------------------------------------------------------------------------------------------------
// `row`: null = the Gaussian's SH row is read from shs and its gradient row written to dL_dsh directly (M == 1);
// else the LDS row that holds the SH coefficients on entry and the gradient on exit (staged variant, M > 1).
__device__ __forceinline__ void gaussian_backward_one(
    const int idx, float* row,
    const FrameParams& fp, GeomState& g, const int* __restrict__ radii, const float* __restrict__ means3D, const float* __restrict__ scales,
    const float* __restrict__ rotations, const float* __restrict__ shs, const float* __restrict__ cov3D_precomp,
    const float* __restrict__ V, const float* __restrict__ Pm, const float* __restrict__ campos,
    const int colors_are_precomp, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
    float* __restrict__ dL_dmean3D,
    float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh, float* __restrict__ dL_dscale,
    float* __restrict__ dL_drot, const bool skip_recordless) {
  // Two round trips: the Gaussian's radius and "touched" flag first -- a Gaussian without a gradient record (culled, or
  // visible but taken by no pixel: 97 % of the scene in a dense near/far frame) gets its zeros written and reads nothing
  // else -- then, for the others, every remaining input at once (record sums, mean, covariance inputs, clamp bits)
  // rather than one round trip per stage.  (Round 2 requested everything up front for every Gaussian: one round trip,
  // but 94 bytes read per Gaussian that only the touched ones need.)
  const int rad_in = radii[idx];
  const uint8_t touched_in = g.touched[idx];
  const bool vis = rad_in > 0;
  const int M = fp.M;
  float* gs = row ? row : dL_dsh + (size_t)idx * fp.M * 3;
  if (idx == 0) g.total[2] = 0u;  // the touched list has been consumed (k_gather_records ran before this kernel)
  const bool was_touched = touched_in != 0;
  if (was_touched) g.touched[idx] = 0;  // leave the blobs clean for another backward over them
  const bool rec = vis && was_touched;  // sums left by k_gather_records; everything else has no record at all
  if (!rec && (skip_recordless || !vis)) {  // every gradient of this Gaussian is zero (with all-zero record sums the chain below yields zeros)
    dL_dmean2D[3 * idx] = 0.f; dL_dmean2D[3 * idx + 1] = 0.f; dL_dmean2D[3 * idx + 2] = 0.f;
    dL_dconic[4 * idx] = 0.f; dL_dconic[4 * idx + 1] = 0.f; dL_dconic[4 * idx + 2] = 0.f; dL_dconic[4 * idx + 3] = 0.f;
    dL_dopacity[idx] = 0.f;
    dL_dcolor[3 * idx] = 0.f; dL_dcolor[3 * idx + 1] = 0.f; dL_dcolor[3 * idx + 2] = 0.f;
    dL_dmean3D[3 * idx] = 0.f; dL_dmean3D[3 * idx + 1] = 0.f; dL_dmean3D[3 * idx + 2] = 0.f;
    if (dL_dcov3D) {  // (null: the 3-D covariance is not an input of the caller's graph, nobody reads its gradient)
#pragma unroll
      for (int k = 0; k < 6; k++) dL_dcov3D[6 * (size_t)idx + k] = 0.f;
    }
    for (int k = 0; k < 3 * M; k++) gs[k] = 0.f;
    dL_dscale[3 * idx] = 0.f; dL_dscale[3 * idx + 1] = 0.f; dL_dscale[3 * idx + 2] = 0.f;
    dL_drot[4 * idx] = 0.f; dL_drot[4 * idx + 1] = 0.f; dL_drot[4 * idx + 2] = 0.f; dL_drot[4 * idx + 3] = 0.f;
    return;
  }
  // (skip_recordless off -- GSR_GBWD_ALL=1, diagnostics --: a visible Gaussian without a record walks the chain with zero sums)
  const float gcol0 = rec ? dL_dcolor[3 * idx] : 0.f, gcol1 = rec ? dL_dcolor[3 * idx + 1] : 0.f, gcol2 = rec ? dL_dcolor[3 * idx + 2] : 0.f;
  const float gmx = rec ? dL_dmean2D[3 * idx] : 0.f, gmy = rec ? dL_dmean2D[3 * idx + 1] : 0.f;
  const float gca = rec ? dL_dconic[4 * idx] : 0.f, gcb = rec ? dL_dconic[4 * idx + 1] : 0.f, gcc = rec ? dL_dconic[4 * idx + 3] : 0.f;
  if (!rec) {
    dL_dmean2D[3 * idx] = 0.f; dL_dmean2D[3 * idx + 1] = 0.f;
    dL_dconic[4 * idx] = 0.f; dL_dconic[4 * idx + 1] = 0.f; dL_dconic[4 * idx + 3] = 0.f;
    dL_dopacity[idx] = 0.f;
    dL_dcolor[3 * idx] = 0.f; dL_dcolor[3 * idx + 1] = 0.f; dL_dcolor[3 * idx + 2] = 0.f;
  }
  const float mx = means3D[3 * idx], my = means3D[3 * idx + 1], mz = means3D[3 * idx + 2];
  const uint8_t clamped_in = g.clamped[idx];
  float4 q_in = make_float4(0.f, 0.f, 0.f, 0.f);
  float sc_in[3] = {0.f, 0.f, 0.f};
  if (scales) {
    q_in = reinterpret_cast<const float4*>(rotations)[idx];
    sc_in[0] = scales[3 * idx]; sc_in[1] = scales[3 * idx + 1]; sc_in[2] = scales[3 * idx + 2];
  }
  // The 3D covariance is not kept by the forward (24 B written + 24 B read per Gaussian): it is recomputed here with
  // the forward's own arithmetic (same function, same unfused file), so it is the same bits (rasterizer_impl.cu:427
  // reads geomState.cov3D instead).
  float c6[6];
  if (cov3D_precomp) {
#pragma unroll
    for (int k = 0; k < 6; k++) c6[k] = cov3D_precomp[6 * (size_t)idx + k];
  } else {
    cov3d_from_scale_rot(fp.scale_modifier * sc_in[0], fp.scale_modifier * sc_in[1], fp.scale_modifier * sc_in[2], q_in, c6);
  }
  const float* sh = row ? row : shs + (size_t)idx * fp.M * 3;
  // (the record sums stay where k_gather_records left them: they are this kernel's dL_dmean2D / dL_dconic / dL_dopacity /
  // dL_dcolor outputs as well; the unused third / fourth components are zeroed)
  dL_dmean2D[3 * idx + 2] = 0.f;
  dL_dconic[4 * idx + 2] = 0.f;
  // ---- B2: conic -> cov2D -> cov3D and mean (backward.cu:140-275) ----
  const Ewa e = ewa_project(mx, my, mz, fp, c6, V);
  const float limx = 1.3f * fp.tan_fovx, limy = 1.3f * fp.tan_fovy;
  const float xmul = (e.txtz < -limx || e.txtz > limx) ? 0.f : 1.f;
  const float ymul = (e.tytz < -limy || e.tytz > limy) ? 0.f : 1.f;
  const float a = e.cxx + 0.3f, b = e.cxy, c = e.cyy + 0.3f;
  const float denom = a * c - b * b;
  float dL_da = 0, dL_db = 0, dL_dc = 0;
  const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
  float dcov[6] = {0, 0, 0, 0, 0, 0};
  if (denom2inv != 0) {
    dL_da = denom2inv * (-c * c * gca + 2 * b * c * gcb + (denom - a * c) * gcc);
    dL_dc = denom2inv * (-a * a * gcc + 2 * a * b * gcb + (denom - a * c) * gca);
    dL_db = denom2inv * 2 * (b * c * gca - (denom + 2 * b * b) * gcb + a * b * gcc);
    dcov[0] = (e.T00 * e.T00 * dL_da + e.T00 * e.T10 * dL_db + e.T10 * e.T10 * dL_dc);
    dcov[3] = (e.T01 * e.T01 * dL_da + e.T01 * e.T11 * dL_db + e.T11 * e.T11 * dL_dc);
    dcov[5] = (e.T02 * e.T02 * dL_da + e.T02 * e.T12 * dL_db + e.T12 * e.T12 * dL_dc);
    dcov[1] = 2 * e.T00 * e.T01 * dL_da + (e.T00 * e.T11 + e.T01 * e.T10) * dL_db + 2 * e.T10 * e.T11 * dL_dc;
    dcov[2] = 2 * e.T00 * e.T02 * dL_da + (e.T00 * e.T12 + e.T02 * e.T10) * dL_db + 2 * e.T10 * e.T12 * dL_dc;
    dcov[4] = 2 * e.T02 * e.T01 * dL_da + (e.T01 * e.T12 + e.T02 * e.T11) * dL_db + 2 * e.T11 * e.T12 * dL_dc;
  }
  if (dL_dcov3D) {
#pragma unroll
    for (int k = 0; k < 6; k++) dL_dcov3D[6 * (size_t)idx + k] = dcov[k];
  }
  // Vrk[c][r] symmetric: S(c,r)
  const float S00 = c6[0], S01 = c6[1], S02 = c6[2], S11 = c6[3], S12 = c6[4], S22 = c6[5];
  const float u0 = e.T00 * S00 + e.T01 * S01 + e.T02 * S02, u1 = e.T00 * S01 + e.T01 * S11 + e.T02 * S12,
              u2 = e.T00 * S02 + e.T01 * S12 + e.T02 * S22;
  const float v0 = e.T10 * S00 + e.T11 * S01 + e.T12 * S02, v1 = e.T10 * S01 + e.T11 * S11 + e.T12 * S12,
              v2 = e.T10 * S02 + e.T11 * S12 + e.T12 * S22;
  const float dT00 = 2 * u0 * dL_da + v0 * dL_db, dT01 = 2 * u1 * dL_da + v1 * dL_db, dT02 = 2 * u2 * dL_da + v2 * dL_db;
  const float dT10 = 2 * v0 * dL_dc + u0 * dL_db, dT11 = 2 * v1 * dL_dc + u1 * dL_db, dT12 = 2 * v2 * dL_dc + u2 * dL_db;
  // W[k][r] = V[k + 4r]
  const float dJ00 = V[0] * dT00 + V[4] * dT01 + V[8] * dT02;
  const float dJ02 = V[2] * dT00 + V[6] * dT01 + V[10] * dT02;
  const float dJ11 = V[1] * dT10 + V[5] * dT11 + V[9] * dT12;
  const float dJ12 = V[2] * dT10 + V[6] * dT11 + V[10] * dT12;
  const float tz = 1.f / e.tz, tz2 = tz * tz, tz3 = tz2 * tz;
  const float hx = fp.focal_x, hy = fp.focal_y;
  const float dtx = xmul * -hx * tz2 * dJ02;
  const float dty = ymul * -hy * tz2 * dJ12;
  const float dtz = -hx * tz2 * dJ00 - hy * tz2 * dJ11 + (2 * hx * e.tx) * tz3 * dJ02 + (2 * hy * e.ty) * tz3 * dJ12;
  float dm0 = V[0] * dtx + V[1] * dty + V[2] * dtz;
  float dm1 = V[4] * dtx + V[5] * dty + V[6] * dtz;
  float dm2 = V[8] * dtx + V[9] * dty + V[10] * dtz;
  // ---- B3: projection Jacobian (backward.cu:389-407) ----
  {
    const float mh3 = Pm[3] * mx + Pm[7] * my + Pm[11] * mz + Pm[15];
    const float m_w = 1.0f / (mh3 + 0.0000001f);
    const float mul1 = (Pm[0] * mx + Pm[4] * my + Pm[8] * mz + Pm[12]) * m_w * m_w;
    const float mul2 = (Pm[1] * mx + Pm[5] * my + Pm[9] * mz + Pm[13]) * m_w * m_w;
    dm0 += (Pm[0] * m_w - Pm[3] * mul1) * gmx + (Pm[1] * m_w - Pm[3] * mul2) * gmy;
    dm1 += (Pm[4] * m_w - Pm[7] * mul1) * gmx + (Pm[5] * m_w - Pm[7] * mul2) * gmy;
    dm2 += (Pm[8] * m_w - Pm[11] * mul1) * gmx + (Pm[9] * m_w - Pm[11] * mul2) * gmy;
  }
  // ---- SH backward (backward.cu:20-135) ----
  if (shs && !colors_are_precomp) {
    const uint8_t cb = clamped_in;
    const float gr[3] = {(cb & 1) ? 0.f : gcol0, (cb & 2) ? 0.f : gcol1, (cb & 4) ? 0.f : gcol2};
    const float q0 = mx - campos[0], q1 = my - campos[1], q2 = mz - campos[2];
    const float len = sqrtf(q0 * q0 + q1 * q1 + q2 * q2);
    const float x = q0 / len, y = q1 / len, z = q2 / len;
    float ddx = 0, ddy = 0, ddz = 0;  // dL_ddir
    const int D = fp.D;
    const int used = (D + 1) * (D + 1);
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
      const float gch = gr[ch];
      // this channel's coefficients, read before its gradients are written (gs may be the same LDS row)
      float c_[16];
#pragma unroll
      for (int k = 1; k < 16; k++) c_[k] = k < used ? sh[3 * k + ch] : 0.f;
      float dx_ = 0, dy_ = 0, dz_ = 0;  // dRGB/d{x,y,z} for this channel
      gs[ch] = SH0 * gch;
      if (D > 0) {
        gs[3 + ch] = (-SH1 * y) * gch;
        gs[6 + ch] = (SH1 * z) * gch;
        gs[9 + ch] = (-SH1 * x) * gch;
        dx_ = -SH1 * c_[3];
        dy_ = -SH1 * c_[1];
        dz_ = SH1 * c_[2];
        if (D > 1) {
          const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
          gs[12 + ch] = (SH2c[0] * xy) * gch;
          gs[15 + ch] = (SH2c[1] * yz) * gch;
          gs[18 + ch] = (SH2c[2] * (2.f * zz - xx - yy)) * gch;
          gs[21 + ch] = (SH2c[3] * xz) * gch;
          gs[24 + ch] = (SH2c[4] * (xx - yy)) * gch;
          dx_ += SH2c[0] * y * c_[4] + SH2c[2] * 2.f * -x * c_[6] + SH2c[3] * z * c_[7] +
                 SH2c[4] * 2.f * x * c_[8];
          dy_ += SH2c[0] * x * c_[4] + SH2c[1] * z * c_[5] + SH2c[2] * 2.f * -y * c_[6] +
                 SH2c[4] * 2.f * -y * c_[8];
          dz_ += SH2c[1] * y * c_[5] + SH2c[2] * 2.f * 2.f * z * c_[6] + SH2c[3] * x * c_[7];
          if (D > 2) {
            gs[27 + ch] = (SH3c[0] * y * (3.f * xx - yy)) * gch;
            gs[30 + ch] = (SH3c[1] * xy * z) * gch;
            gs[33 + ch] = (SH3c[2] * y * (4.f * zz - xx - yy)) * gch;
            gs[36 + ch] = (SH3c[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * gch;
            gs[39 + ch] = (SH3c[4] * x * (4.f * zz - xx - yy)) * gch;
            gs[42 + ch] = (SH3c[5] * z * (xx - yy)) * gch;
            gs[45 + ch] = (SH3c[6] * x * (xx - 3.f * yy)) * gch;
            dx_ += (SH3c[0] * c_[9] * 3.f * 2.f * xy + SH3c[1] * c_[10] * yz +
                    SH3c[2] * c_[11] * -2.f * xy + SH3c[3] * c_[12] * -3.f * 2.f * xz +
                    SH3c[4] * c_[13] * (-3.f * xx + 4.f * zz - yy) + SH3c[5] * c_[14] * 2.f * xz +
                    SH3c[6] * c_[15] * 3.f * (xx - yy));
            dy_ += (SH3c[0] * c_[9] * 3.f * (xx - yy) + SH3c[1] * c_[10] * xz +
                    SH3c[2] * c_[11] * (-3.f * yy + 4.f * zz - xx) + SH3c[3] * c_[12] * -3.f * 2.f * yz +
                    SH3c[4] * c_[13] * -2.f * xy + SH3c[5] * c_[14] * -2.f * yz +
                    SH3c[6] * c_[15] * -3.f * 2.f * xy);
            dz_ += (SH3c[1] * c_[10] * xy + SH3c[2] * c_[11] * 4.f * 2.f * yz +
                    SH3c[3] * c_[12] * 3.f * (2.f * zz - xx - yy) + SH3c[4] * c_[13] * 4.f * 2.f * xz +
                    SH3c[5] * c_[14] * (xx - yy));
          }
        }
      }
      ddx += dx_ * gch;
      ddy += dy_ * gch;
      ddz += dz_ * gch;
    }
    // coefficients beyond the active degree get zero gradient
    for (int k = used * 3; k < M * 3; k++) gs[k] = 0.f;
    const float sum2 = q0 * q0 + q1 * q1 + q2 * q2;
    const float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    dm0 += ((+sum2 - q0 * q0) * ddx - q1 * q0 * ddy - q2 * q0 * ddz) * inv32;
    dm1 += (-q0 * q1 * ddx + (sum2 - q1 * q1) * ddy - q2 * q1 * ddz) * inv32;
    dm2 += (-q0 * q2 * ddx - q1 * q2 * ddy + (sum2 - q2 * q2) * ddz) * inv32;
  } else {
    for (int k = 0; k < M * 3; k++) gs[k] = 0.f;
  }
  dL_dmean3D[3 * idx] = dm0; dL_dmean3D[3 * idx + 1] = dm1; dL_dmean3D[3 * idx + 2] = dm2;
  // ---- covariance -> scale / rotation (backward.cu:279-366) ----
  if (scales) {
    const float4 q = q_in;
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    const float s[3] = {fp.scale_modifier * sc_in[0], fp.scale_modifier * sc_in[1], fp.scale_modifier * sc_in[2]};
    float R[3][3];  // R[c][r]
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
    // dSig[c][r]
    const float dS[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                            {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                            {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
    // dM = (2*M) * dSig, M[c][r] = s_r R[c][r];  dM[c][r] = sum_k 2M[k][r] * dS[c][k]
    float dMt[3][3];  // dMt[c][r] = dM[r][c]
#pragma unroll
    for (int cc = 0; cc < 3; cc++)
#pragma unroll
      for (int rr = 0; rr < 3; rr++) {
        const float v = (2.0f * (s[rr] * R[0][rr])) * dS[cc][0] + (2.0f * (s[rr] * R[1][rr])) * dS[cc][1] +
                        (2.0f * (s[rr] * R[2][rr])) * dS[cc][2];
        dMt[rr][cc] = v;
      }
    // Rt[c][r] = R[r][c];  dL_dscale_k = dot(Rt[k], dMt[k])
    float ds[3];
#pragma unroll
    for (int k = 0; k < 3; k++) ds[k] = R[0][k] * dMt[k][0] + R[1][k] * dMt[k][1] + R[2][k] * dMt[k][2];
    dL_dscale[3 * idx] = ds[0]; dL_dscale[3 * idx + 1] = ds[1]; dL_dscale[3 * idx + 2] = ds[2];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
      for (int rr = 0; rr < 3; rr++) dMt[k][rr] *= s[k];
#define Dm(ci, ri) dMt[ci][ri]
    const float dq0 = 2 * z * (Dm(0, 1) - Dm(1, 0)) + 2 * y * (Dm(2, 0) - Dm(0, 2)) + 2 * x * (Dm(1, 2) - Dm(2, 1));
    const float dq1 = 2 * y * (Dm(1, 0) + Dm(0, 1)) + 2 * z * (Dm(2, 0) + Dm(0, 2)) + 2 * r * (Dm(1, 2) - Dm(2, 1)) -
                      4 * x * (Dm(2, 2) + Dm(1, 1));
    const float dq2 = 2 * x * (Dm(1, 0) + Dm(0, 1)) + 2 * r * (Dm(2, 0) - Dm(0, 2)) + 2 * z * (Dm(1, 2) + Dm(2, 1)) -
                      4 * y * (Dm(2, 2) + Dm(0, 0));
    const float dq3 = 2 * r * (Dm(0, 1) - Dm(1, 0)) + 2 * x * (Dm(2, 0) + Dm(0, 2)) + 2 * y * (Dm(1, 2) + Dm(2, 1)) -
                      4 * z * (Dm(1, 1) + Dm(0, 0));
#undef Dm
    dL_drot[4 * idx] = dq0; dL_drot[4 * idx + 1] = dq1; dL_drot[4 * idx + 2] = dq2; dL_drot[4 * idx + 3] = dq3;
  } else {
    dL_dscale[3 * idx] = 0.f; dL_dscale[3 * idx + 1] = 0.f; dL_dscale[3 * idx + 2] = 0.f;
    dL_drot[4 * idx] = 0.f; dL_drot[4 * idx + 1] = 0.f; dL_drot[4 * idx + 2] = 0.f; dL_drot[4 * idx + 3] = 0.f;
  }
}

template <bool STAGED>
__global__ __launch_bounds__(PRE_BLOCK) void k_gaussian_backward(
    const FrameParams fp, GeomState g, const int* __restrict__ radii, const float* __restrict__ means3D, const float* __restrict__ scales,
    const float* __restrict__ rotations, const float* __restrict__ shs, const float* __restrict__ cov3D_precomp,
    const float* __restrict__ V, const float* __restrict__ Pm, const float* __restrict__ campos,
    const int colors_are_precomp, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
    float* __restrict__ dL_dmean3D,
    float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh, float* __restrict__ dL_dscale,
    float* __restrict__ dL_drot, const bool skip_recordless) {
  extern __shared__ float sh_rows[];  // STAGED: SH rows in, dL_dsh rows out (listed_rows_to_lds / lds_to_rows)
  __shared__ uint32_t s_rec[PRE_BLOCK];  // STAGED: block-local rows of the Gaussians with a gradient record
  __shared__ uint32_t s_nrec;
  const int row0 = blockIdx.x * PRE_BLOCK, idx = row0 + threadIdx.x;
  const int C = fp.M * 3, nrows = min(PRE_BLOCK, fp.P - row0);
  if (STAGED) {
    // only the rows gaussian_backward_one will read (same test as there; the flags are cleared by that function, later)
    if (threadIdx.x == 0) s_nrec = 0u;
    __syncthreads();
    if (idx < fp.P && radii[idx] > 0 && (g.touched[idx] != 0 || !skip_recordless)) s_rec[atomicAdd(&s_nrec, 1u)] = threadIdx.x;
    __syncthreads();
    listed_rows_to_lds(sh_rows, shs + (size_t)row0 * C, s_rec, (int)s_nrec, C);
    __syncthreads();
  }
  if (idx < fp.P)
    gaussian_backward_one(idx, STAGED ? sh_rows + threadIdx.x * sh_row_stride(C) : nullptr, fp, g, radii, means3D, scales,
                          rotations, shs, cov3D_precomp, V, Pm, campos, colors_are_precomp, dL_dmean2D, dL_dconic,
                          dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, skip_recordless);
  if (STAGED) {
    __syncthreads();
    lds_to_rows(dL_dsh + (size_t)row0 * C, sh_rows, nrows, C);
  }
}

// V1.  Replaces checkFrustum / in_frustum (reference rasterizer_impl.cu:52-60, auxiliary.h:120-144).
__global__ __launch_bounds__(PRE_BLOCK) void k_mark_visible(int P, const float* __restrict__ means3D,
                                                            const float* __restrict__ V,
                                                            unsigned char* __restrict__ present) {
  const int idx = blockIdx.x * PRE_BLOCK + threadIdx.x;
  if (idx >= P) return;
  const float z = V[2] * means3D[3 * idx] + V[6] * means3D[3 * idx + 1] + V[10] * means3D[3 * idx + 2] + V[14];
  present[idx] = !(z <= 0.2f);
}

// ---------------------------------- launchers ----------------------------------------------------
// whether k_preprocess counts the depth sort's digit histograms for this call: not in the LDS-staged SH variant,
// and only when its workgroups walk several blocks each (beyond 1 M Gaussians) -- every workgroup flushes ~770
// bins with global atomics, which a one-block workgroup cannot hide (500 k Gaussians: 29 -> 51 us, against 10 us
// for the sort's own histogram pass)
bool preprocess_counts_depth_digits(const FrameParams& fp, const float* shs, const float* colors_precomp) {
  // (GSR_PRE_HIST_MIN_P: tests run the paths that hang on these histograms -- the near limit, the partial depth sort --
  // on small scenes)
  // From 128 k Gaussians (with one workgroup per CU up to 1 M, so that a workgroup walks several blocks between its ~770
  // flush atomics: 640x512 / 300 k: k_preprocess +2.5 us, k_sort_hist_all's 8.6 us launch gone; 500 k: +4.8 / -9.6).
  static const long min_p = getenv("GSR_PRE_HIST_MIN_P") ? atol(getenv("GSR_PRE_HIST_MIN_P")) : (1 << 17);
  // (with SH rows staged through LDS the kernel keeps three workgroups per CU -- it has no prefetch to cover a lone
  // workgroup's loads --, so the digits are counted there only where a workgroup still walks several blocks: > 1 M)
  static const bool min_p_given = getenv("GSR_PRE_HIST_MIN_P") != nullptr;
  const bool staged = shs && !colors_precomp && sh_stage_bytes(fp.M) != 0;
  return fp.P > (staged && !min_p_given ? (1L << 20) : min_p);
}

hipError_t launch_preprocess(const FrameParams& fp, const float* means3D, const float* scales, const float* rotations,
                             const float* opacities, const float* shs, const float* cov3D_precomp,
                             const float* colors_precomp, const float* view, const float* proj, const float* campos,
                             GeomState g, int* radii_out, bool write_cov3D, unsigned long long* done_word,
                             unsigned long long* publish, uint32_t ticket, uint32_t* ghist_acc, uint32_t* ghist_clear,
                             hipStream_t s) {
  if (!radii_out) radii_out = g.radii;  // rasterizer_impl.cu:217-219
  const int nb = (fp.P + PRE_BLOCK - 1) / PRE_BLOCK;
  ProfScope ps_k_preprocess(K_PREPROCESS, s);
  const size_t stage = (shs && !colors_precomp) ? sh_stage_bytes(fp.M) : 0;  // M > 1: SH rows go through LDS
  // one resident round: 3 workgroups of 4 waves per CU x 256 CUs, the blocks spread evenly over them.  The kernel's
  // time does not depend on the occupancy between 2 and 5 waves per SIMD (C3, one box: 2, 3, 4, 5, 6 per CU = 63, 63,
  // 64, 67, 70 us; with the arithmetic compiled out it still takes 56 us for its 252 MB: it is bound by the memory
  // path, not by latency or VALU), and every workgroup ends with ~770 histogram flush atomics and one same-address count
  // atomic, so fewer and longer-lived workgroups win
  static const int wg_per_cu = getenv("GSR_PRE_WG_PER_CU") ? atoi(getenv("GSR_PRE_WG_PER_CU")) : 3;  // experiment knob
  // (up to 1 M Gaussians a single workgroup per CU when the digits are counted here: see preprocess_counts_depth_digits)
  static const bool wg_forced = getenv("GSR_PRE_WG_PER_CU") != nullptr;
  const bool few_wg = ghist_acc != nullptr && fp.P <= (1 << 20) && !wg_forced;
  const int max_wg = (stage ? 3 : few_wg ? 1 : wg_per_cu) * 256, rounds = (nb + max_wg - 1) / max_wg;
  const dim3 grid(rounds ? (nb + rounds - 1) / rounds : 1);
  // (the staged variants: SH degree 1 and 3 with plain inputs pipeline their rows per wave, see the kernel's header)
  const bool plain_in = shs && scales && rotations && !cov3D_precomp && !colors_precomp && !write_cov3D;
  const bool rows16 = plain_in && (reinterpret_cast<uintptr_t>(shs) & 15u) == 0;
  static const bool env_block_rows = getenv("GSR_PRE_BLOCK_ROWS") != nullptr;  // diagnostics: the block-wide copy
  if (stage && rows16 && fp.M == 16 && !env_block_rows)
    hipLaunchKernelGGL((k_preprocess<true, false, true, 12>), grid, dim3(PRE_BLOCK), stage, s, fp, means3D, scales, rotations,
                       opacities, shs, cov3D_precomp, colors_precomp, view, proj, campos, g, radii_out, write_cov3D,
                       done_word, publish, ticket, ghist_acc, ghist_clear);
  else if (stage && rows16 && fp.M == 4 && !env_block_rows)
    hipLaunchKernelGGL((k_preprocess<true, false, true, 3>), grid, dim3(PRE_BLOCK), stage, s, fp, means3D, scales, rotations,
                       opacities, shs, cov3D_precomp, colors_precomp, view, proj, campos, g, radii_out, write_cov3D,
                       done_word, publish, ticket, ghist_acc, ghist_clear);
  else if (stage)
    hipLaunchKernelGGL(k_preprocess<true>, grid, dim3(PRE_BLOCK), stage, s, fp, means3D, scales, rotations, opacities,
                       shs, cov3D_precomp, colors_precomp, view, proj, campos, g, radii_out, write_cov3D, done_word, publish, ticket,
                       ghist_acc, ghist_clear);
  else if (fp.D == 0 && shs && scales && rotations && !cov3D_precomp && !colors_precomp && !write_cov3D)  // the product's configuration
    hipLaunchKernelGGL((k_preprocess<false, true>), grid, dim3(PRE_BLOCK), 0, s, fp, means3D, scales, rotations, opacities,
                       shs, cov3D_precomp, colors_precomp, view, proj, campos, g, radii_out, write_cov3D, done_word, publish, ticket,
                       ghist_acc, ghist_clear);
  else
    hipLaunchKernelGGL(k_preprocess<false>, grid, dim3(PRE_BLOCK), 0, s, fp, means3D, scales, rotations, opacities,
                       shs, cov3D_precomp, colors_precomp, view, proj, campos, g, radii_out, write_cov3D, done_word, publish, ticket,
                       ghist_acc, ghist_clear);
  return hipGetLastError();
}

// debug forwards only: the reference's point_offsets array for the views
hipError_t launch_point_offsets(const FrameParams& fp, GeomState g, hipStream_t s) {
  ProfScope ps(K_POINT_OFFSETS, s);
  hipLaunchKernelGGL(k_point_offsets, dim3(1), dim3(1024), 0, s, fp.P, g.gpack, g.point_offsets);
  return hipGetLastError();
}

hipError_t launch_scan_offsets(const FrameParams& fp, GeomState g, Count R, uint32_t* chunk_first, uint2* ranges,
                               uint2* rangesB, uint32_t* counts0, uint32_t near_budget,
                               unsigned long long* publish_near, uint32_t ticket, const uint32_t* top_hist,
                               const uint32_t* order, bool order_is_near_list, hipStream_t s) {
  const size_t ncounts0 = (size_t)((R.cap + TSORT_TILE - 1) / TSORT_TILE) * 256;
  ProfScope ps(K_SCAN_OFFSETS, s);
  hipLaunchKernelGGL(k_scan_offsets, dim3((fp.P + SCAN_TILE - 1) / SCAN_TILE), dim3(PRE_BLOCK), 0, s, fp, g, R,
                     chunk_first, ranges, rangesB, counts0, ncounts0, near_budget, publish_near, ticket, top_hist, order,
                     order_is_near_list);
  return hipGetLastError();
}

hipError_t launch_compact_near(const FrameParams& fp, GeomState g, const uint32_t* top_hist, uint32_t near_budget,
                               uint32_t* keys_out, uint32_t* vals_out, uint32_t* n_out, uint32_t* ghist_near,
                               unsigned long long* publish, uint32_t ticket, hipStream_t s) {
  ProfScope ps(K_COMPACT_NEAR, s);
  hipLaunchKernelGGL(k_compact_near, dim3((fp.P + COMPACT_TILE - 1) / COMPACT_TILE), dim3(PRE_BLOCK), 0, s, fp.P, g.dkeysA,
                     top_hist, near_budget, keys_out, vals_out, n_out, ghist_near, g.dsort.scanC_status(),
                     g.dsort.tickets() + 6, publish, ticket);
  return hipGetLastError();
}

hipError_t launch_scan_offsets_far(const FrameParams& fp, GeomState g, Count capB, uint32_t slot_base, const uint32_t* sat,
                                   uint32_t* chunk_firstB, uint32_t* counts0, unsigned long long* publish,
                                   uint32_t ticket, hipStream_t s) {
  const size_t ncounts0 = (size_t)((capB.cap + TSORT_TILE - 1) / TSORT_TILE) * 256;
  ProfScope ps(K_SCAN_OFFSETS, s);
  hipLaunchKernelGGL(k_scan_offsets_far, dim3((fp.P + SCAN_TILE - 1) / SCAN_TILE), dim3(PRE_BLOCK), 0, s, fp, g, capB,
                     slot_base, sat, chunk_firstB, counts0, ncounts0, publish, ticket);
  return hipGetLastError();
}

hipError_t launch_emit(const FrameParams& fp, const uint4* sdesc, Count R, uint32_t* chunk_first, uint32_t* tkeys_out,
                       uint32_t* ivals_out, uint8_t* inst_flag, uint32_t* counts0, uint32_t digit_shift0,
                       uint32_t digit_mask0, bool key16, bool store_pairs, hipStream_t s) {
  if (R.cap <= 0) return hipSuccess;
  ProfScope ps(K_EMIT, s);
  const dim3 grid(chain_grid(R, EMIT_CHUNK));
  if (key16 && store_pairs)
    hipLaunchKernelGGL((k_emit<uint16_t, true>), grid, dim3(256), 0, s, fp, sdesc, R, chunk_first,
                       reinterpret_cast<uint16_t*>(tkeys_out), ivals_out, inst_flag, counts0, digit_shift0, digit_mask0);
  else if (key16)
    hipLaunchKernelGGL((k_emit<uint16_t, false>), grid, dim3(256), 0, s, fp, sdesc, R, chunk_first,
                       reinterpret_cast<uint16_t*>(tkeys_out), ivals_out, inst_flag, counts0, digit_shift0, digit_mask0);
  else
    hipLaunchKernelGGL((k_emit<uint32_t, true>), grid, dim3(256), 0, s, fp, sdesc, R, chunk_first, tkeys_out, ivals_out,
                       inst_flag, counts0, digit_shift0, digit_mask0);
  return hipGetLastError();
}

// first pass of the 16-bit tile sort with the pairs generated in place (see k_emit_scatter)
hipError_t launch_emit_scatter(const EmitFusion& ef, uint16_t* keys_out, uint32_t* vals_out, int shift0, int nbits0,
                               const uint32_t* counts, const uint32_t* chunk_base, const uint32_t* digit_total,
                               bool arank, hipStream_t s) {
  const dim3 grid(chain_grid(ef.R, TSORT_TILE));
  if (arank)
    hipLaunchKernelGGL(k_emit_scatter<true>, grid, dim3(256), 0, s, ef.fp, ef.sdesc, ef.R, ef.chunk_first, keys_out, vals_out,
                       shift0, nbits0, counts, chunk_base, digit_total);
  else
    hipLaunchKernelGGL(k_emit_scatter<false>, grid, dim3(256), 0, s, ef.fp, ef.sdesc, ef.R, ef.chunk_first, keys_out,
                       vals_out, shift0, nbits0, counts, chunk_base, digit_total);
  return hipGetLastError();
}

hipError_t launch_gather_records(const FrameParams& fp, GeomState g, BinningState b, float* dL_dmean2D,
                                 float* dL_dconic, float* dL_dopacity, float* dL_dcolor, bool split_frame,
                                 hipStream_t s) {
  const int nb = (fp.P + PRE_BLOCK - 1) / PRE_BLOCK;
  if (!split_frame) {
    ProfScope ps(K_COMPACT_TOUCHED, s);
    hipLaunchKernelGGL(k_compact_touched, dim3((nb + 15) / 16), dim3(PRE_BLOCK), 0, s, fp.P, g.touched, g.tlist,
                       g.total + 2);
  }
  {
    ProfScope ps(K_GATHER_RECORDS, s);
    // grid-stride over the touched list, one Gaussian per wave: up to P waves' worth of workgroups (the list
    // length is only known on the device), capped where the chip is full several times over
    const int want = (fp.P + PRE_BLOCK / 64 - 1) / (PRE_BLOCK / 64);
    const int grid = want < 4096 ? want : 4096;
    hipLaunchKernelGGL(k_gather_records, dim3(grid), dim3(PRE_BLOCK), 0, s, g, b.grad_inst, b.inst_flag, dL_dmean2D,
                       dL_dconic, dL_dopacity, dL_dcolor, 0.5f * (float)fp.W, 0.5f * (float)fp.H, fp.P, split_frame);
  }
  return hipGetLastError();
}

hipError_t launch_gaussian_backward(const FrameParams& fp, GeomState g, BinningState b, const int* radii,
                                    const float* means3D, const float* scales, const float* rotations,
                                    const float* shs, const float* cov3D_precomp, const float* view, const float* proj,
                                    const float* campos, bool colors_precomp, float* dL_dmean2D, float* dL_dconic,
                                    float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                                    float* dL_dsh, float* dL_dscale, float* dL_drot, hipStream_t s) {
  const int nb = (fp.P + PRE_BLOCK - 1) / PRE_BLOCK;
  ProfScope ps_k_gaussian_bwd(K_GAUSSIAN_BWD, s);
  const size_t stage = (shs && !colors_precomp) ? sh_stage_bytes(fp.M) : 0;  // M > 1: SH / dL_dsh rows go through LDS
  static const bool skip_recordless = getenv("GSR_GBWD_ALL") == nullptr;  // (GSR_GBWD_ALL=1: diagnostics)
  if (stage)
    hipLaunchKernelGGL(k_gaussian_backward<true>, dim3(nb), dim3(PRE_BLOCK), stage, s, fp, g, radii,
                       means3D, scales, rotations, shs, cov3D_precomp, view, proj, campos, 0,
                       dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot,
                       skip_recordless);
  else
    hipLaunchKernelGGL(k_gaussian_backward<false>, dim3(nb), dim3(PRE_BLOCK), 0, s, fp, g, radii,
                       means3D, scales, rotations, shs, cov3D_precomp, view, proj, campos, colors_precomp ? 1 : 0,
                       dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot,
                       skip_recordless);
  return hipGetLastError();
}

hipError_t launch_mark_visible(int P, const float* means3D, const float* view, unsigned char* present,
                               hipStream_t s) {
  const int nb = (P + PRE_BLOCK - 1) / PRE_BLOCK;
  ProfScope ps_k_mark_visible(K_MARK_VISIBLE, s);
  hipLaunchKernelGGL(k_mark_visible, dim3(nb), dim3(PRE_BLOCK), 0, s, P, means3D, view, present);
  return hipGetLastError();
}

}  // namespace gsr
