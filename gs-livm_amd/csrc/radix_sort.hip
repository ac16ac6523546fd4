// radix_sort.hip -- stable LSD radix sort of (u64 key, u32 value) pairs and per-tile range
// identification, hand-written for gfx950 wave64.
//
// Replaces cub::DeviceRadixSort::SortPairs(begin_bit 0, end_bit 32+bit) and identifyTileRanges
// (reference rasterizer_impl.cu:298-309, :106-125).  The permutation must equal a stable sort
// (ties keep ascending Gaussian id); every step below is order-preserving.
//
// Work unit: a WAVE-TILE of 1024 consecutive pairs (64 lanes x 16 steps, lane = consecutive
// element, so loads are 512-B coalesced).  Per 8-bit pass:
//   k_sort_hist        per-wave-tile digit counts                  counts[tile][256]
//   k_sort_scan_chunks exclusive prefix over the 64 tiles of a chunk (in place) + chunk_sums[chunk][256]
//   k_sort_scan_top    one wave per digit: exclusive prefix over chunks (in place) + digit totals
//   k_sort_scatter     base(d) = digit_base[d] + chunk_base[chunk][d] + counts[tile][d]; ranks inside a
//                      step come from a wave-wide digit match (8 ballots) + popcount of lower lanes;
//                      per-wave running offsets live in LDS; waves are independent (no block barrier
//                      in the main loop, no atomics, no inter-workgroup communication).
#include "gsr_internal.hpp"

namespace gsr {

constexpr int STEPS = SORT_TILE / 64;  // 16

__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// 64-bit mask of the valid lanes holding the same 8-bit digit as this lane.
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid) {
  uint64_t m = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; b++) {
    const bool bit = (d >> b) & 1u;
    const uint64_t bm = __ballot(bit);
    m &= bit ? bm : ~bm;
  }
  return m;
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

__global__ __launch_bounds__(256) void k_sort_hist(const uint64_t* __restrict__ keys, int n, int ntiles, int shift,
                                                   uint32_t* __restrict__ counts) {
  __shared__ uint32_t hist[4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int tile = blockIdx.x * 4 + w;
#pragma unroll
  for (int k = 0; k < 4; k++) hist[w][lane + 64 * k] = 0;
  __syncthreads();
  if (tile < ntiles) {
    const size_t base = (size_t)tile * SORT_TILE;
    uint64_t key[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
      const size_t i = base + (size_t)s * 64 + lane;
      key[s] = i < (size_t)n ? keys[i] : 0ull;
    }
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
      const size_t i = base + (size_t)s * 64 + lane;
      const bool valid = i < (size_t)n;
      const uint32_t d = (uint32_t)(key[s] >> shift) & 0xFFu;
      const uint64_t m = match_digit(d, valid);
      const int rank = __popcll(m & lanemask_lt(lane));
      if (valid && rank == 0) hist[w][d] += (uint32_t)__popcll(m);  // one leader per digit: no conflicts
    }
  }
  __syncthreads();
  if (tile < ntiles) {
#pragma unroll
    for (int k = 0; k < 4; k++) counts[(size_t)tile * 256 + lane + 64 * k] = hist[w][lane + 64 * k];
  }
}

// grid = nchunks, block = 256 (thread = digit).
__global__ __launch_bounds__(256) void k_sort_scan_chunks(uint32_t* __restrict__ counts, int ntiles,
                                                          uint32_t* __restrict__ chunk_sums) {
  const int d = threadIdx.x;
  const int t0 = blockIdx.x * SORT_CHUNK;
  uint32_t v[SORT_CHUNK];
#pragma unroll
  for (int k = 0; k < SORT_CHUNK; k++) v[k] = (t0 + k < ntiles) ? counts[(size_t)(t0 + k) * 256 + d] : 0u;
  uint32_t run = 0;
#pragma unroll
  for (int k = 0; k < SORT_CHUNK; k++) {
    const uint32_t c = v[k];
    if (t0 + k < ntiles) counts[(size_t)(t0 + k) * 256 + d] = run;
    run += c;
  }
  chunk_sums[(size_t)blockIdx.x * 256 + d] = run;
}

// grid = 64, block = 256: one wave per digit scans that digit's column of chunk_sums.
__global__ __launch_bounds__(256) void k_sort_scan_top(uint32_t* __restrict__ chunk_sums, int nchunks,
                                                       uint32_t* __restrict__ digit_total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int d = blockIdx.x * 4 + w;
  uint32_t carry = 0;
  for (int c0 = 0; c0 < nchunks; c0 += 64) {
    const int c = c0 + lane;
    const uint32_t v = c < nchunks ? chunk_sums[(size_t)c * 256 + d] : 0u;
    const uint32_t inc = wave_incl_scan(v, lane);
    if (c < nchunks) chunk_sums[(size_t)c * 256 + d] = carry + inc - v;
    carry += __shfl(inc, 63, 64);
  }
  if (lane == 0) digit_total[d] = carry;
}

__global__ __launch_bounds__(256) void k_sort_scatter(const uint64_t* __restrict__ keys_in,
                                                      const uint32_t* __restrict__ vals_in,
                                                      uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                      int n, int ntiles, int shift, const uint32_t* __restrict__ counts,
                                                      const uint32_t* __restrict__ chunk_base,
                                                      const uint32_t* __restrict__ digit_total) {
  __shared__ uint32_t woff[4][256];
  __shared__ uint32_t dbase[256];
  __shared__ uint32_t wtot[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int tile = blockIdx.x * 4 + w;
  // exclusive scan of the 256 digit totals (every block recomputes it: 1 KB, L2-resident)
  {
    const uint32_t t = digit_total[threadIdx.x];
    const uint32_t inc = wave_incl_scan(t, lane);
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    uint32_t o = 0;
    for (int k = 0; k < w; k++) o += wtot[k];
    dbase[threadIdx.x] = o + inc - t;
    __syncthreads();
  }
  if (tile >= ntiles) return;  // no barrier below: waves are independent from here on
  const int chunk = tile / SORT_CHUNK;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int d = lane + 64 * k;
    woff[w][d] = dbase[d] + chunk_base[(size_t)chunk * 256 + d] + counts[(size_t)tile * 256 + d];
  }
  const size_t base = (size_t)tile * SORT_TILE;
  uint64_t key[STEPS];
  uint32_t val[STEPS];
#pragma unroll
  for (int s = 0; s < STEPS; s++) {
    const size_t i = base + (size_t)s * 64 + lane;
    const bool valid = i < (size_t)n;
    key[s] = valid ? keys_in[i] : 0ull;
    val[s] = valid ? vals_in[i] : 0u;
  }
  volatile uint32_t* my = woff[w];
#pragma unroll
  for (int s = 0; s < STEPS; s++) {
    const size_t i = base + (size_t)s * 64 + lane;
    const bool valid = i < (size_t)n;
    const uint32_t d = (uint32_t)(key[s] >> shift) & 0xFFu;
    const uint64_t m = match_digit(d, valid);
    const uint32_t rank = (uint32_t)__popcll(m & lanemask_lt(lane));
    const uint32_t off = my[d];  // same address inside a digit group: LDS broadcast
    // LDS operations of one wave execute in issue order, so every lane has read `off` before the
    // group leader publishes the advanced offset for the next step.
    if (valid && rank == 0) my[d] = off + (uint32_t)__popcll(m);
    if (valid) {
      keys_out[off + rank] = key[s];
      vals_out[off + rank] = val[s];
    }
  }
}

// Replaces identifyTileRanges (reference rasterizer_impl.cu:106-125); ranges must be zeroed first
// (the reference's cudaMemset at :311).
__global__ __launch_bounds__(256) void k_tile_ranges(const uint64_t* __restrict__ keys, int L,
                                                     uint2* __restrict__ ranges) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= L) return;
  const uint32_t cur = (uint32_t)(keys[idx] >> 32);
  if (idx == 0) {
    ranges[cur].x = 0;
  } else {
    const uint32_t prev = (uint32_t)(keys[idx - 1] >> 32);
    if (cur != prev) {
      ranges[prev].y = (uint32_t)idx;
      ranges[cur].x = (uint32_t)idx;
    }
  }
  if (idx == L - 1) ranges[cur].y = (uint32_t)L;
}

// The pairs start in (keysA, point_list) when start_in_A, else in (keysB, valsB); passes alternate
// and the caller picks start_in_A = (passes even) so the result always lands in (keysA, point_list).
hipError_t launch_sort_pairs(BinningState b, int R, int end_bit, bool start_in_A, hipStream_t s) {
  if (R <= 0) return hipSuccess;
  const int ntiles = (R + SORT_TILE - 1) / SORT_TILE;
  const int nchunks = (ntiles + SORT_CHUNK - 1) / SORT_CHUNK;
  const int nblk = (ntiles + 3) / 4;
  const int passes = sort_passes(end_bit);
  bool inA = start_in_A;
  for (int p = 0; p < passes; p++) {
    const uint64_t* kin = inA ? b.keysA : b.keysB;
    const uint32_t* vin = inA ? b.point_list : b.valsB;
    uint64_t* kout = inA ? b.keysB : b.keysA;
    uint32_t* vout = inA ? b.valsB : b.point_list;
    const int shift = 8 * p;
    {
      ProfScope ps(K_SORT_HIST, s);
      hipLaunchKernelGGL(k_sort_hist, dim3(nblk), dim3(256), 0, s, kin, R, ntiles, shift, b.counts);
    }
    {
      ProfScope ps(K_SORT_SCAN_CHUNKS, s);
      hipLaunchKernelGGL(k_sort_scan_chunks, dim3(nchunks), dim3(256), 0, s, b.counts, ntiles, b.chunk_sums);
    }
    {
      ProfScope ps(K_SORT_SCAN_TOP, s);
      hipLaunchKernelGGL(k_sort_scan_top, dim3(64), dim3(256), 0, s, b.chunk_sums, nchunks, b.digit_base);
    }
    {
      ProfScope ps(K_SORT_SCATTER, s);
      hipLaunchKernelGGL(k_sort_scatter, dim3(nblk), dim3(256), 0, s, kin, vin, kout, vout, R, ntiles, shift,
                         b.counts, b.chunk_sums, b.digit_base);
    }
    inA = !inA;
  }
  return hipGetLastError();
}

hipError_t launch_tile_ranges(const uint64_t* keys, int R, uint2* ranges, int tiles, hipStream_t s) {
  hipError_t e = hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)tiles, s);
  if (e != hipSuccess) return e;
  if (R > 0) {
    ProfScope ps(K_TILE_RANGES, s);
    hipLaunchKernelGGL(k_tile_ranges, dim3((R + 255) / 256), dim3(256), 0, s, keys, R, ranges);
  }
  return hipGetLastError();
}

}  // namespace gsr
