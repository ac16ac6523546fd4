// radix_sort.hip -- stable LSD radix sort of (u32 key, u32 value) pairs and per-tile range
// identification, hand-written for gfx950 wave64.
//
// Replaces cub::DeviceRadixSort::SortPairs(begin_bit 0, end_bit 32+bit) on 64-bit (tile|depth) keys and
// identifyTileRanges (reference rasterizer_impl.cu:298-309, :106-125).  The reference's permutation --
// by tile, then depth bits, then ascending Gaussian id (stable sort) -- is
// produced here by TWO stable sorts on 32-bit keys: the P Gaussians by depth bits (4 passes over P
// pairs), then, emitted in that order, the R instances by tile id (2 passes at 1080p).  A stable sort
// on the minor key followed by a stable sort on the major key IS the lexicographic sort, so the result
// is bit-identical, at roughly a quarter of the HBM traffic of six passes over R 12-byte pairs.
//
// Work unit: a workgroup tile of 4096 consecutive pairs = 4 waves x (64 lanes x 16 steps), lane =
// consecutive element, so loads are coalesced.  The scatter kernel (k_sort_scatter) is shared by both sorts:
//   * ranks inside a wave come from ONE returning LDS add per 64 elements on a per-wave digit table when the
//     one-time probe k_probe_lds_atomic_order confirms lane-ordered conflict resolution (ARANK), otherwise from
//     a wave-wide digit match (<= 8 ballots) + popcount of lower lanes;
//   * the tile is then REORDERED IN LDS by digit and written out run by run, so global stores are coalesced
//     runs instead of 64 scattered dwords per instruction.
// Instance (tile-id) sort, R ~ 10^7 pairs, u16 keys: classic passes -- per-tile digit counts (from k_emit for
//   the first pass, k_sort_hist after that), k_sort_scan_chunks / k_sort_scan_top, k_sort_scatter; the first pass
//   generates its pairs in place (k_emit_scatter, preprocess.hip), the last one leaves per-tile counts instead of
//   sorted keys (COUNT, sort_core.hpp) from which k_ranges_from_counts makes the tile ranges.
// Depth sort, P ~ 10^6 pairs, u32 keys: digit histograms from the key producer (or k_sort_hist_all) and ONE launch
//   per pass with decoupled look-back (LB): ticketed tiles, one status word per (tile, digit), counts published
//   right after the key load.  For the 10^7-pair sort the look-back measured slower than the helper kernels
//   (DESIGN.md "Tried and rejected"), for the 10^6-pair sort it replaces 16 launches by 5.
// Every step is order-preserving (stable); no float atomics.
#include <cstdlib>
#include <mutex>

#include "gsr_internal.hpp"
#include "sort_core.hpp"

namespace gsr {


// Per-tile digit counts with LDS integer atomics (order-independent, so still deterministic): three
// instructions per key instead of a ballot match; neighbouring keys rarely share a digit in either sort.
template <typename K, int TILE>
__global__ __launch_bounds__(256) void k_sort_hist(const K* __restrict__ keys, const Count cnt, int shift, int nbits,
                                                   uint32_t* __restrict__ counts) {
  __shared__ uint32_t hist[256];
  const int tid = threadIdx.x;
  if (cnt.closed()) return;
  const int n = cnt.get();  // (tiles beyond it leave zero counts: the scans run over the capacity's tiles)
  const uint32_t mask = (1u << nbits) - 1u;
  for_each_unit(units_of(cnt.cap, TILE), [&](const int tile) {
  hist[tid] = 0;
  __syncthreads();
  const size_t base = (size_t)tile * TILE;
  uint32_t key[TILE / 256];
#pragma unroll
  for (int s = 0; s < TILE / 256; s++) {
    const size_t i = base + (size_t)s * 256 + tid;
    key[s] = i < (size_t)n ? (uint32_t)keys[i] : 0u;
  }
#pragma unroll
  for (int s = 0; s < TILE / 256; s++) {
    const size_t i = base + (size_t)s * 256 + tid;
    if (i < (size_t)n) atomicAdd(&hist[(key[s] >> shift) & mask], 1u);
  }
  __syncthreads();
  counts[(size_t)tile * 256 + tid] = hist[tid];
  });
}

// grid = nchunks, block = 256 (thread = digit).
__global__ __launch_bounds__(256) void k_sort_scan_chunks(uint32_t* __restrict__ counts, int ntiles,
                                                          uint32_t* __restrict__ chunk_sums, const Count gate) {
  if (gate.closed()) return;
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
                                                       uint32_t* __restrict__ digit_total, const Count gate) {
  if (gate.closed()) return;
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

// Both scans in ONE launch, for sorts of up to SCAN_COLUMNS_MAX workgroup tiles (8 M pairs: the near/far frames and
// every frame below 1080p): grid = 64, block = 256, one wave per digit scans that digit's column of the per-tile
// counts in place (exclusive), eight strided loads in flight per lane; the scatter pass then needs no chunk bases.
// Beyond that size a column no longer fits a wave's few round trips and the two-level scan above is faster.
constexpr int SCAN_COLUMNS_MAX = 2048;
__global__ __launch_bounds__(256) void k_sort_scan_columns(uint32_t* __restrict__ counts, const int ntiles,
                                                           uint32_t* __restrict__ digit_total, const Count gate) {
  if (gate.closed()) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int d = blockIdx.x * 4 + w;
  uint32_t carry = 0;
  for (int c0 = 0; c0 < ntiles; c0 += 1024) {  // (a near chain's 640 tiles: one round trip)
    uint32_t v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int t = c0 + 64 * k + lane;
      v[k] = t < ntiles ? counts[(size_t)t * 256 + d] : 0u;
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int t = c0 + 64 * k + lane;
      if (c0 + 64 * k < ntiles) {  // (wave-uniform)
        const uint32_t inc = wave_incl_scan(v[k], lane);
        if (t < ntiles) counts[(size_t)t * 256 + d] = carry + inc - v[k];
        carry += __shfl(inc, 63, 64);
      }
    }
  }
  if (lane == 0) digit_total[d] = carry;
}

// All four digit histograms of the 32-bit keys in one pass over them (single-launch-per-pass sort below):
// LDS atomics per workgroup, then one global atomicAdd per non-empty bin.  ghist[4][256] is zero on entry.
__global__ __launch_bounds__(256) void k_sort_hist_all(const uint32_t* __restrict__ keys, int n,
                                                       uint32_t* __restrict__ ghist) {
  __shared__ uint32_t hist[4][256];
  const int tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 4; p++) hist[p][tid] = 0;
  __syncthreads();
  const int ntiles = (n + SORT_TILE - 1) / SORT_TILE;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const size_t base = (size_t)tile * SORT_TILE;
    uint32_t key[SORT_TILE / 256];
#pragma unroll
    for (int s = 0; s < SORT_TILE / 256; s++) {
      const size_t i = base + (size_t)s * 256 + tid;
      key[s] = i < (size_t)n ? keys[i] : 0u;
    }
#pragma unroll
    for (int s = 0; s < SORT_TILE / 256; s++) {
      const size_t i = base + (size_t)s * 256 + tid;
      if (i < (size_t)n) {
        atomicAdd(&hist[0][key[s] & 255u], 1u);
        atomicAdd(&hist[1][(key[s] >> 8) & 255u], 1u);
        atomicAdd(&hist[2][(key[s] >> 16) & 255u], 1u);
        atomicAdd(&hist[3][key[s] >> 24], 1u);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; p++) {
    const uint32_t c = hist[p][tid];
    if (c) atomicAdd(&ghist[p * 256 + tid], c);
  }
}

// Probe for the ARANK variant of k_sort_scatter: 1 = within one ds_add_rtn_u32 instruction, lanes that hit the
// same LDS address receive their return values in ascending lane order (and successive instructions
// accumulate), for 512 digit patterns covering every conflict multiplicity from 1 to 64.
__global__ __launch_bounds__(64) void k_probe_lds_atomic_order(uint32_t* __restrict__ ok_out) {
  __shared__ uint32_t hist[128];
  const int lane = threadIdx.x;
  volatile uint32_t* vh = hist;
  bool ok = true;
  for (uint32_t trial = 0; trial < 512; trial++) {
    vh[lane] = 0;
    vh[lane + 64] = 0;
    // number of distinct digits: 1, 2, 4 .. 128, then pseudo-random in [1, 128]
    const uint32_t groups = trial < 8 ? 1u << trial : 1u + ((trial * 2654435761u) >> 25);
    for (uint32_t rep = 0; rep < 2; rep++) {  // the second instruction must continue from the first one's counts
      uint32_t h = (uint32_t)lane * 0x9E3779B1u + trial * 0x85EBCA77u + rep * 0xC2B2AE3Du;
      h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
      // odd trials: scattered digits; even trials: runs of equal digits in neighbouring lanes
      const uint32_t run = 64u / (groups > 64u ? 64u : groups);
      const uint32_t d = ((trial & 1u) ? h : (uint32_t)lane / run + rep) % groups;
      const uint64_t m = match_digit(d, true, 7);
      const uint32_t before = vh[d];  // LDS operations of one wave execute in issue order
      const uint32_t got = __hip_atomic_fetch_add(&hist[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (got != before + (uint32_t)__popcll(m & lanemask_lt(lane))) ok = false;
    }
  }
  const uint64_t all = __ballot(ok);
  if (lane == 0) *ok_out = all == ~0ull ? 1u : 0u;
}

// Once per process: may k_sort_scatter rank with returning LDS atomics (see ARANK)?  GSR_SORT_BALLOT_RANK=1 in
// the environment forces the ballot variant.
static bool lds_atomic_rank_ok(hipStream_t s) {
  // decided once per DEVICE, under a lock (the reference calls the rasterizer from several host threads); a debug
  // forward additionally verifies the order of every sorted list it produces (k_verify_sorted_lists)
  static std::mutex mu;
  static int state[64];
  static bool init = false;
  std::lock_guard<std::mutex> lk(mu);
  if (!init) {
    for (int& v : state) v = -1;
    init = true;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  if (state[dev] >= 0) return state[dev] == 1;
  state[dev] = 0;
  if (getenv("GSR_SORT_BALLOT_RANK")) return false;
  uint32_t* d = nullptr;
  uint32_t h = 0;
  if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(uint32_t)) != hipSuccess) return false;
  hipLaunchKernelGGL(k_probe_lds_atomic_order, dim3(1), dim3(64), 0, s, d);
  if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess)
    state[dev] = h == 1u ? 1 : 0;
  (void)hipFree(d);
  return state[dev] == 1;
}

// Debug forwards only: every tile's list must be ordered by (depth bits, Gaussian id) -- the order a stable sort of
// the reference's 64-bit keys produces.  Cheap self-check of the whole binning chain,
// in particular of the returning-LDS-atomic ranking whose lane order the ISA manual does not document.
__global__ __launch_bounds__(256) void k_verify_sorted_lists(const uint2* __restrict__ ranges, const int T,
                                                             const uint32_t* __restrict__ point_list,
                                                             const float4* __restrict__ splats,
                                                             uint32_t* __restrict__ violations) {
  const int tile = blockIdx.x;
  if (tile >= T) return;
  const uint2 r = ranges[tile];
  uint32_t bad = 0;
  for (uint32_t i = r.x + 1 + threadIdx.x; i < r.y; i += 256) {
    const uint32_t a = point_list[i - 1], b = point_list[i];
    const uint32_t da = __float_as_uint(splats[(size_t)a * SPLAT_F4 + 2].y), db = __float_as_uint(splats[(size_t)b * SPLAT_F4 + 2].y);
    if (da > db || (da == db && a >= b)) bad++;
  }
  if (bad) atomicAdd(violations, bad);
}

hipError_t launch_verify_sorted_lists(const uint2* ranges, int T, const uint32_t* point_list, const float4* splats,
                                      uint32_t* violations, hipStream_t s) {
  hipError_t e = hipMemsetAsync(violations, 0, sizeof(uint32_t), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_verify_sorted_lists, dim3(T), dim3(256), 0, s, ranges, T, point_list, splats, violations);
  return hipGetLastError();
}

// Second and last pass of the bucket form of the tile sort (gsr_internal.hpp, tile_sort_buckets).  The first pass has
// partitioned the pairs, stably, by the top eight bits of the 16-bit tile id: bucket b is the contiguous run of
// digit_total[b] pairs behind the buckets below it, in emission (= depth) order.  One 1024-thread workgroup per bucket
// holds up to 16 384 pairs in registers (16 per thread, lane = consecutive pair: coalesced loads, all in flight at
// once).  Ranking the pairs inside their wave -- ONE returning LDS add per 64 pairs on a per-wave table of the bucket's
// <= 256 tiles where the probe allows (sort_core.hpp, ARANK), a ballot match otherwise -- leaves the tables holding the
// per-wave counts, i.e. the bucket's histogram as well: thread = tile sums them, the bucket's 256 counts are scanned,
// the RANGES of its tiles are written -- (0, 0) for tiles without instances, the reference's memset
// (rasterizer_impl.cu:311) -- and every pair goes to segment start + (pairs of earlier waves) + rank.  A bucket that
// does not fit (sorts of several million pairs per chain) is counted in a first sweep and scattered chunk by chunk in
// a second.  What a classic pass needs k_sort_hist + k_sort_scan_columns + k_sort_scatter<COUNT> +
// k_ranges_from_counts for stays inside one workgroup.  Output = the reference's list order (tile, depth bits,
// Gaussian id), bit for bit.
constexpr int BUCKET_THREADS = 1024, BUCKET_WAVES = BUCKET_THREADS / 64, BUCKET_ITEMS = 16;
constexpr int BUCKET_CHUNK = BUCKET_THREADS * BUCKET_ITEMS;

// exclusive scan of one value per thread of the FIRST 256 threads of a 1024-thread workgroup (every thread calls it)
__device__ __forceinline__ uint32_t bucket_excl_scan_256(uint32_t v, int tid, uint32_t* wtot /*[4]*/) {
  const int lane = tid & 63, w = tid >> 6;
  const uint32_t inc = tid < 256 ? wave_incl_scan(v, lane) : 0u;
  if (tid < 256 && lane == 63) wtot[w] = inc;
  __syncthreads();
  uint32_t o = 0;
  if (tid < 256)
    for (int k = 0; k < w; k++) o += wtot[k];
  __syncthreads();
  return o + inc - v;
}

template <bool ARANK>
__global__ __launch_bounds__(BUCKET_THREADS) void k_bucket_sort(const uint16_t* __restrict__ keys,
                                                                const uint32_t* __restrict__ vals,
                                                                uint32_t* __restrict__ out, const Count cnt,
                                                                const uint32_t* __restrict__ digit_total,
                                                                const int lowbits, const int T,
                                                                uint2* __restrict__ ranges, const uint32_t list_base) {
  __shared__ uint32_t wcnt[BUCKET_WAVES][256];  // per-wave tile counts of the chunk, then per-wave write positions
  __shared__ uint32_t hist[256], run[256], wtot[4];
  // a bucket that fits one chunk is brought into its final order HERE and written out in one coalesced sweep (64
  // scattered 4-byte stores per wave instruction measured 3x the time of the whole rest of the kernel)
  __shared__ uint32_t lval[BUCKET_CHUNK];
  __shared__ uint32_t s_start, s_count;
  if (cnt.closed()) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t lowmask = (1u << lowbits) - 1u;
  for (int b = blockIdx.x; b < 256; b += gridDim.x) {
    {
      const uint32_t dt = tid < 256 ? digit_total[tid] : 0u;
      const uint32_t ex = bucket_excl_scan_256(dt, tid, wtot);
      if (tid == b) { s_start = ex; s_count = dt; }
      if (tid < 256) hist[tid] = 0u;
      __syncthreads();
    }
    const uint32_t start = s_start, count = s_count;
    const bool one_chunk = count <= (uint32_t)BUCKET_CHUNK;  // (workgroup-uniform)
    if (!one_chunk) {  // first sweep: the bucket's histogram
      for (uint32_t c0 = 0; c0 < count; c0 += BUCKET_CHUNK) {
        uint32_t k16[BUCKET_ITEMS];
#pragma unroll
        for (int s = 0; s < BUCKET_ITEMS; s++) {
          const uint32_t i = c0 + (uint32_t)(w * (BUCKET_ITEMS * 64) + s * 64 + lane);
          k16[s] = i < count ? (uint32_t)keys[start + i] & lowmask : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int s = 0; s < BUCKET_ITEMS; s++)
          if (k16[s] != 0xFFFFFFFFu) atomicAdd(&hist[k16[s]], 1u);
      }
      __syncthreads();
      const uint32_t c = tid < 256 ? hist[tid] : 0u;
      const uint32_t o = bucket_excl_scan_256(c, tid, wtot);
      if (tid < 256) {
        run[tid] = start + o;  // next free position of this tile's segment
        const uint32_t t = ((uint32_t)b << lowbits) | (uint32_t)tid;
        if ((uint32_t)tid <= lowmask && t < (uint32_t)T)
          ranges[t] = c ? make_uint2(list_base + start + o, list_base + start + o + c) : make_uint2(0u, 0u);
      }
      __syncthreads();
    }
    for (uint32_t c0 = 0; c0 < count || (one_chunk && c0 == 0); c0 += BUCKET_CHUNK) {
#pragma unroll
      for (int k = 0; k < 4; k++) wcnt[w][lane + 64 * k] = 0u;  // (this wave's own table; others read it behind a barrier)
      uint32_t d[BUCKET_ITEMS], v[BUCKET_ITEMS], lrank[BUCKET_ITEMS];
#pragma unroll
      for (int s = 0; s < BUCKET_ITEMS; s++) {
        const uint32_t i = c0 + (uint32_t)(w * (BUCKET_ITEMS * 64) + s * 64 + lane);
        const bool valid = i < count;
        d[s] = valid ? (uint32_t)keys[start + i] & lowmask : 0xFFFFFFFFu;
        v[s] = valid ? vals[start + i] : 0u;
      }
      volatile uint32_t* my = wcnt[w];
#pragma unroll
      for (int s = 0; s < BUCKET_ITEMS; s++) {
        const bool valid = d[s] != 0xFFFFFFFFu;
        if (ARANK) {  // (lanes of one ds_add_rtn that hit one address are served in lane order -- probed once per process)
          lrank[s] = valid ? __hip_atomic_fetch_add(&wcnt[w][d[s]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
        } else {
          const uint64_t m = match_digit(valid ? d[s] : 0u, valid, lowbits);
          const uint32_t rank = (uint32_t)__popcll(m & lanemask_lt(lane));
          const uint32_t prior = my[valid ? d[s] : 0u];
          if (valid && rank == 0) my[d[s]] = prior + (uint32_t)__popcll(m);
          lrank[s] = prior + rank;
        }
      }
      __syncthreads();
      if (one_chunk) {  // the per-wave counts ARE the bucket's histogram: scan it and write the ranges
        uint32_t c = 0;
        if (tid < 256)
          for (int k = 0; k < BUCKET_WAVES; k++) c += wcnt[k][tid];
        const uint32_t o = bucket_excl_scan_256(c, tid, wtot);
        if (tid < 256) {
          run[tid] = start + o;
          const uint32_t t = ((uint32_t)b << lowbits) | (uint32_t)tid;
          if ((uint32_t)tid <= lowmask && t < (uint32_t)T)
            ranges[t] = c ? make_uint2(list_base + start + o, list_base + start + o + c) : make_uint2(0u, 0u);
        }
      }
      if (tid < 256) {  // thread = tile of the bucket: where each wave's pairs of this chunk go
        uint32_t pos = run[tid];
#pragma unroll
        for (int k = 0; k < BUCKET_WAVES; k++) {
          const uint32_t n = wcnt[k][tid];
          wcnt[k][tid] = pos;
          pos += n;
        }
        run[tid] = pos;
      }
      __syncthreads();
      if (one_chunk) {
#pragma unroll
        for (int s = 0; s < BUCKET_ITEMS; s++)
          if (d[s] != 0xFFFFFFFFu) lval[wcnt[w][d[s]] + lrank[s] - start] = v[s];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < BUCKET_ITEMS; s++) {
          const uint32_t i = (uint32_t)(s * BUCKET_THREADS + tid);
          if (i < count) out[start + i] = lval[i];
        }
      } else {
#pragma unroll
        for (int s = 0; s < BUCKET_ITEMS; s++)
          if (d[s] != 0xFFFFFFFFu) out[wcnt[w][d[s]] + lrank[s]] = v[s];
      }
      __syncthreads();
    }
  }
}

bool tile_sort_buckets(int tile_bits, bool key16, int capacity) {
  static const bool lsd = getenv("GSR_TILE_SORT_LSD") != nullptr;  // diagnostics / fallback: always the two LSD passes
  return key16 && !lsd && tile_bits > 8 && sort_passes(tile_bits) == 2 && capacity <= TILE_SORT_BUCKETS_MAX;
}

// The pairs start in (keysA, valsA) when start_in_A, else in (keysB, valsB); passes alternate.  The caller
// picks start_in_A = (passes even) so the result always lands in (keysA, valsA).
template <typename K>
static hipError_t sort_pairs_impl(K* keysA, uint32_t* valsA, K* keysB, uint32_t* valsB, SortScratch sc, Count cnt,
                                  int end_bit, bool start_in_A, bool is_depth_sort, bool first_hist_done,
                                  const EmitFusion* ef, uint32_t* key_count, hipStream_t s,
                                  const BucketPass* buckets = nullptr) {
  const int kb = is_depth_sort ? (int)K_DSORT_HIST - (int)K_SORT_HIST : 0;  // profiler ids of this sort
  // tile geometry of the instance sort (both key widths use the same today; see TSORT_TILE)
  constexpr int TT = sizeof(K) == 2 ? TSORT_TILE : SORT_TILE, NWV = sizeof(K) == 2 ? TSORT_WAVES : 4;
  if (sizeof(K) != 2) first_hist_done = false;  // the emitter counts per TSORT_TILE slots
  const int n = cnt.cap;  // grids and scratch cover the capacity; the kernels read the count itself (Count)
  const int ntiles = (n + TT - 1) / TT;
  const int nchunks = (ntiles + SORT_CHUNK - 1) / SORT_CHUNK;
  const int passes = sort_passes(end_bit);
  const int nbits = sort_digit_bits(end_bit);
  const bool arank = lds_atomic_rank_ok(s);
  static const bool two_level_scan = getenv("GSR_SORT_TWO_LEVEL_SCAN") != nullptr;  // diagnostics / fallback
  bool inA = start_in_A;
  if (buckets && sizeof(K) == 2 && ef && first_hist_done) {
    // bucket form: first pass = the fused emit-scatter on the TOP eight bits (its counts came from the emitter), second
    // pass = one launch that finishes every bucket and writes the ranges
    const int lowbits = end_bit - 8;
    const bool one_scan = ntiles <= SCAN_COLUMNS_MAX && !two_level_scan;
    const uint32_t* chunk_base = one_scan ? nullptr : sc.chunk_sums;
    if (one_scan) {
      ProfScope ps(K_SORT_SCAN_CHUNKS, s);
      hipLaunchKernelGGL(k_sort_scan_columns, dim3(64), dim3(256), 0, s, sc.counts, ntiles, sc.digit_base, cnt);
    } else {
      {
        ProfScope ps(K_SORT_SCAN_CHUNKS, s);
        hipLaunchKernelGGL(k_sort_scan_chunks, dim3(nchunks), dim3(256), 0, s, sc.counts, ntiles, sc.chunk_sums, cnt);
      }
      {
        ProfScope ps(K_SORT_SCAN_TOP, s);
        hipLaunchKernelGGL(k_sort_scan_top, dim3(64), dim3(256), 0, s, sc.chunk_sums, nchunks, sc.digit_base, cnt);
      }
    }
    K* kmid = inA ? keysB : keysA;
    uint32_t* vmid = inA ? valsB : valsA;
    uint32_t* vfin = inA ? valsA : valsB;
    {
      ProfScope ps(K_SORT_SCATTER, s);
      const hipError_t e = launch_emit_scatter(*ef, reinterpret_cast<uint16_t*>(kmid), vmid, lowbits, 8, sc.counts,
                                               chunk_base, sc.digit_base, arank, s);
      if (e != hipSuccess) return e;
    }
    {
      ProfScope ps(K_TILE_RANGES, s);  // (booked where the range step was: it now holds the whole second pass)
      if (arank)
        hipLaunchKernelGGL(k_bucket_sort<true>, dim3(256), dim3(BUCKET_THREADS), 0, s, reinterpret_cast<const uint16_t*>(kmid), vmid,
                           vfin, cnt, sc.digit_base, lowbits, buckets->tiles, buckets->ranges, buckets->list_base);
      else
        hipLaunchKernelGGL(k_bucket_sort<false>, dim3(256), dim3(BUCKET_THREADS), 0, s, reinterpret_cast<const uint16_t*>(kmid), vmid,
                           vfin, cnt, sc.digit_base, lowbits, buckets->tiles, buckets->ranges, buckets->list_base);
    }
    return hipGetLastError();
  }
  for (int p = 0; p < passes; p++) {
    const K* kin = inA ? keysA : keysB;
    const uint32_t* vin = inA ? valsA : valsB;
    K* kout = inA ? keysB : keysA;
    uint32_t* vout = inA ? valsB : valsA;
    const int shift = nbits * p;
    if (!(p == 0 && first_hist_done)) {  // the emitter already left the first pass's counts in sc.counts
      ProfScope ps(K_SORT_HIST + kb, s);
      hipLaunchKernelGGL((k_sort_hist<K, TT>), dim3(chain_grid(cnt, TT)), dim3(256), 0, s, kin, cnt, shift, nbits, sc.counts);
    }
    const bool one_scan = ntiles <= SCAN_COLUMNS_MAX && !two_level_scan;
    const uint32_t* chunk_base = one_scan ? nullptr : sc.chunk_sums;
    if (one_scan) {
      ProfScope ps(K_SORT_SCAN_CHUNKS + kb, s);
      hipLaunchKernelGGL(k_sort_scan_columns, dim3(64), dim3(256), 0, s, sc.counts, ntiles, sc.digit_base, cnt);
    } else {
      {
        ProfScope ps(K_SORT_SCAN_CHUNKS + kb, s);
        hipLaunchKernelGGL(k_sort_scan_chunks, dim3(nchunks), dim3(256), 0, s, sc.counts, ntiles, sc.chunk_sums, cnt);
      }
      {
        ProfScope ps(K_SORT_SCAN_TOP + kb, s);
        hipLaunchKernelGGL(k_sort_scan_top, dim3(64), dim3(256), 0, s, sc.chunk_sums, nchunks, sc.digit_base, cnt);
      }
    }
    if (p == 0 && ef && sizeof(K) == 2) {  // the emitter generates the pairs inside the first pass
      ProfScope ps(K_SORT_SCATTER + kb, s);
      const hipError_t e = launch_emit_scatter(*ef, reinterpret_cast<uint16_t*>(kout), vout, 0, nbits, sc.counts,
                                               chunk_base, sc.digit_base, arank, s);
      if (e != hipSuccess) return e;
    } else if (key_count && p == passes - 1 && p > 0 && sizeof(K) == 2) {
      // last pass of the instance sort: per-key counts instead of the sorted keys (sort_core.hpp, COUNT)
      ProfScope ps(K_SORT_SCATTER + kb, s);
      if (arank)
        hipLaunchKernelGGL((k_sort_scatter<K, false, true, NWV, TT, true>), dim3(chain_grid(cnt, TT)), dim3(64 * NWV), 0, s, kin, vin, kout,
                           vout, cnt, shift, nbits, sc.counts, chunk_base, sc.digit_base, (uint32_t*)nullptr,
                           (uint32_t*)nullptr, key_count);
      else
        hipLaunchKernelGGL((k_sort_scatter<K, false, false, NWV, TT, true>), dim3(chain_grid(cnt, TT)), dim3(64 * NWV), 0, s, kin, vin, kout,
                           vout, cnt, shift, nbits, sc.counts, chunk_base, sc.digit_base, (uint32_t*)nullptr,
                           (uint32_t*)nullptr, key_count);
    } else {
      ProfScope ps(K_SORT_SCATTER + kb, s);
      if (arank)
        hipLaunchKernelGGL((k_sort_scatter<K, false, true, NWV, TT>), dim3(chain_grid(cnt, TT)), dim3(64 * NWV), 0, s, kin, vin, kout, vout, cnt,
                           shift, nbits, sc.counts, chunk_base, sc.digit_base, (uint32_t*)nullptr,
                           (uint32_t*)nullptr);
      else
        hipLaunchKernelGGL((k_sort_scatter<K, false, false, NWV, TT>), dim3(chain_grid(cnt, TT)), dim3(64 * NWV), 0, s, kin, vin, kout, vout, cnt,
                           shift, nbits, sc.counts, chunk_base, sc.digit_base, (uint32_t*)nullptr,
                           (uint32_t*)nullptr);
    }
    inA = !inA;
  }
  return hipGetLastError();
}

// key16: the keys are 16-bit (tile ids of images with <= 65536 tiles): a quarter less traffic per pass; the
// buffers are the same allocations, viewed as uint16_t.
hipError_t launch_sort_pairs(uint32_t* keysA, uint32_t* valsA, uint32_t* keysB, uint32_t* valsB, SortScratch sc,
                             Count n, int end_bit, bool start_in_A, bool is_depth_sort, bool key16,
                             bool first_hist_done, const EmitFusion* fused_first_pass, uint32_t* key_count,
                             hipStream_t s, const BucketPass* buckets) {
  if (n.cap <= 0) return hipSuccess;
  if (key16)
    return sort_pairs_impl<uint16_t>(reinterpret_cast<uint16_t*>(keysA), valsA, reinterpret_cast<uint16_t*>(keysB),
                                     valsB, sc, n, end_bit, start_in_A, is_depth_sort, first_hist_done, fused_first_pass,
                                     key_count, s, buckets);
  return sort_pairs_impl<uint32_t>(keysA, valsA, keysB, valsB, sc, n, end_bit, start_in_A, is_depth_sort, first_hist_done,
                                   nullptr, nullptr, s);
}

// Depth sort of the P (depth bits, Gaussian id) pairs: 32-bit keys, four 8-bit passes, one look-back scatter launch
// per pass; `ghist` = the [4][256] digit histograms if the producer of the keys counted them (k_preprocess
// does), else null and k_sort_hist_all counts them first.  Pairs start in (keysA, valsA) and end
// there.  sc.words (ghist | tickets | status) must be zero on entry: k_preprocess clears it.
template <int TILE, int NW>
static void depth_sort_passes(uint32_t* keysA, uint32_t* valsA, uint32_t* keysB, uint32_t* valsB, DepthSortScratch sc,
                              Count n, bool arank, const uint32_t* ghist, bool vals_are_positions, hipStream_t s) {
  const int ntiles = (n.cap + TILE - 1) / TILE;
  bool inA = true;
  for (int p = 0; p < 4; p++) {
    ProfScope ps(K_DSORT_SCATTER, s);
    const uint32_t* kin = inA ? keysA : keysB;
    const uint32_t* vin = (p == 0 && vals_are_positions) ? nullptr : inA ? valsA : valsB;
    uint32_t* kout = inA ? keysB : keysA;
    uint32_t* vout = inA ? valsB : valsA;
    if (arank)
      hipLaunchKernelGGL((k_sort_scatter<uint32_t, true, true, NW, TILE>), dim3(ntiles), dim3(64 * NW), 0, s, kin, vin, kout, vout,
                         n, 8 * p, 8, (const uint32_t*)nullptr, (const uint32_t*)nullptr, ghist + 256 * p,
                         sc.status(p, ntiles), sc.tickets() + p);
    else
      hipLaunchKernelGGL((k_sort_scatter<uint32_t, true, false, NW, TILE>), dim3(ntiles), dim3(64 * NW), 0, s, kin, vin, kout, vout,
                         n, 8 * p, 8, (const uint32_t*)nullptr, (const uint32_t*)nullptr, ghist + 256 * p,
                         sc.status(p, ntiles), sc.tickets() + p);
    inA = !inA;
  }
}

hipError_t launch_depth_sort(uint32_t* keysA, uint32_t* valsA, uint32_t* keysB, uint32_t* valsB, DepthSortScratch sc,
                             Count n, const uint32_t* ghist, bool vals_are_positions, hipStream_t s) {
  if (n.cap <= 0) return hipSuccess;
  if (!ghist) {  // the producer of the keys did not count the digits: one histogram pass over them (host-known counts only)
    ghist = sc.ghist();
    const int nwg = (n.cap + SORT_TILE - 1) / SORT_TILE;
    ProfScope ps(K_DSORT_HIST, s);
    hipLaunchKernelGGL(k_sort_hist_all, dim3(nwg < 256 ? nwg : 256), dim3(256), 0, s, keysA, n.cap, sc.ghist());
  }
  const bool arank = lds_atomic_rank_ok(s);
  // (the near sort of a partial depth sort -- n.dev set, a few per cent of n.cap pairs -- keeps the tile size of the
  // full sort: with 4096-pair tiles instead of 8192 at 2 M its passes measured 16.5 instead of 14.5 us)
  const size_t tile = depth_sort_tile((size_t)n.cap);
  if (tile == (size_t)SORT_TILE_SMALL)
    depth_sort_passes<SORT_TILE_SMALL, 4>(keysA, valsA, keysB, valsB, sc, n, arank, ghist, vals_are_positions, s);
  else if (tile == (size_t)SORT_TILE)
    depth_sort_passes<SORT_TILE, 4>(keysA, valsA, keysB, valsB, sc, n, arank, ghist, vals_are_positions, s);
  else
    depth_sort_passes<SORT_TILE_BIG, 8>(keysA, valsA, keysB, valsB, sc, n, arank, ghist, vals_are_positions, s);
  return hipGetLastError();
}

// Zeroes library scratch from inside a gated chain (the far chain's full depth sort re-uses the tickets and look-back
// status words the near sort has used; a memset node would run whether the chain is needed or not).
__global__ __launch_bounds__(256) void k_clear_words(const Count gate, uint32_t* __restrict__ words, const size_t n) {
  if (gate.closed()) return;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) words[i] = 0u;
}
hipError_t launch_clear_words(Count gate, uint32_t* words, size_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  const size_t wg = (n + 255) / 256;
  hipLaunchKernelGGL(k_clear_words, dim3((unsigned)(wg < 512 ? wg : 512)), dim3(256), 0, s, gate, words, n);
  return hipGetLastError();
}

// Replaces identifyTileRanges (reference rasterizer_impl.cu:106-125); ranges must be zero beforehand (the
// reference's cudaMemset at :311 -- here a side job of k_scan_offsets).  Every thread owns 16 bytes of sorted
// keys (8 x u16 or 4 x u32) plus the key before them, so the pass over the keys runs at streaming rate.
template <typename K>
__global__ __launch_bounds__(256) void k_tile_ranges(const K* __restrict__ keys, const Count cnt, uint2* __restrict__ ranges,
                                                     const uint32_t list_base) {
  constexpr int PER = 16 / (int)sizeof(K);
  if (cnt.closed()) return;
  const int L = cnt.get();
  const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * PER;
  if (i0 >= (size_t)L) return;
  uint32_t k[PER];
  if (i0 + PER <= (size_t)L) {
    const uint4 v = *reinterpret_cast<const uint4*>(keys + i0);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < PER; j++) k[j] = sizeof(K) == 2 ? (w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu : w[j];
  } else {
#pragma unroll
    for (int j = 0; j < PER; j++) k[j] = i0 + j < (size_t)L ? (uint32_t)keys[i0 + j] : 0u;
  }
  uint32_t prev = i0 ? (uint32_t)keys[i0 - 1] : 0xFFFFFFFFu;  // no tile has this id
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const size_t i = i0 + j;
    if (i < (size_t)L) {
      if (k[j] != prev) {
        if (i) ranges[prev].y = list_base + (uint32_t)i;
        ranges[k[j]].x = list_base + (uint32_t)i;
      }
      if (i == (size_t)L - 1) ranges[k[j]].y = list_base + (uint32_t)L;
      prev = k[j];
    }
  }
}

// Ranges from the per-tile instance counts the last sort pass left in ranges[t].y (ranges[t].x still zero): one
// workgroup scans them; tiles without instances keep the reference's (0, 0) (its cudaMemset, rasterizer_impl.cu:311).
__global__ __launch_bounds__(1024) void k_ranges_from_counts(uint2* __restrict__ ranges, const int T,
                                                             const uint32_t list_base, const Count gate) {
  __shared__ uint32_t wsum[16];
  if (gate.closed()) return;
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry_s = list_base;  // (the far segment of a near/far frame lives behind the near capacity)
  __syncthreads();
  constexpr int PT = 8;  // tiles per thread: a 1080p frame's 8 160 tiles in one round
  for (int t0 = 0; t0 < T; t0 += 1024 * PT) {
    const int i0 = t0 + tid * PT;
    uint32_t c[PT];
#pragma unroll
    for (int k = 0; k < PT; k++) c[k] = i0 + k < T ? ranges[i0 + k].y : 0u;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < PT; k++) sum += c[k];
    const uint32_t inc = wave_incl_scan(sum, lane);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = carry_s;
    for (int k = 0; k < w; k++) base += wsum[k];
    uint32_t run = base + inc - sum;
#pragma unroll
    for (int k = 0; k < PT; k++) {
      if (i0 + k < T) ranges[i0 + k] = c[k] ? make_uint2(run, run + c[k]) : make_uint2(0u, 0u);
      run += c[k];
    }
    __syncthreads();
    if (tid == 1023) carry_s = run;
    __syncthreads();
  }
}

hipError_t launch_ranges_from_counts(uint2* ranges, int T, uint32_t list_base, Count gate, hipStream_t s) {
  ProfScope ps(K_TILE_RANGES, s);
  hipLaunchKernelGGL(k_ranges_from_counts, dim3(1), dim3(1024), 0, s, ranges, T, list_base, gate);
  return hipGetLastError();
}

hipError_t launch_tile_ranges(const uint32_t* keys, Count R, uint2* ranges, bool key16, uint32_t list_base, hipStream_t s) {
  if (R.cap <= 0) return hipSuccess;
  ProfScope ps(K_TILE_RANGES, s);
  if (key16)
    hipLaunchKernelGGL(k_tile_ranges<uint16_t>, dim3((R.cap + 2047) / 2048), dim3(256), 0, s,
                       reinterpret_cast<const uint16_t*>(keys), R, ranges, list_base);
  else
    hipLaunchKernelGGL(k_tile_ranges<uint32_t>, dim3((R.cap + 1023) / 1024), dim3(256), 0, s, keys, R, ranges, list_base);
  return hipGetLastError();
}

}  // namespace gsr
