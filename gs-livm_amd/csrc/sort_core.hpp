// sort_core.hpp -- device code of the radix scatter pass shared by radix_sort.hip (k_sort_scatter) and
// preprocess.hip (k_emit_scatter).  gfx950 only; not part of the public ABI.
#pragma once
#include "gsr_internal.hpp"

namespace gsr {

__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// 64-bit mask of the valid lanes holding the same digit (nbits wide) as this lane.
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid, int nbits) {
  uint64_t m = __ballot(valid);
  for (int b = 0; b < nbits; b++) {
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

// Exclusive scan of one value per thread across the 256 threads of the workgroup.
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, int lane, int w, uint32_t* wtot /*[4]*/) {
  const uint32_t inc = wave_incl_scan(v, lane);
  if (lane == 63) wtot[w] = inc;
  __syncthreads();
  uint32_t o = 0;
  for (int k = 0; k < w; k++) o += wtot[k];
  __syncthreads();
  return o + inc - v;
}

// Status word of one (tile, digit) in the single-launch-per-pass sort: bit 31 = inclusive prefix over tiles
// [0, tile] known, bit 30 = only this tile's own count known, low 30 bits = the count.
constexpr int LBK = 4;  // status words fetched per look-back round trip
constexpr uint32_t ST_GLOBAL = 0x80000000u, ST_LOCAL = 0x40000000u, ST_MASK = 0x3FFFFFFFu;

// LB = false: the classic pass -- digit offsets of every tile come from k_sort_hist / k_sort_scan_*.
// LB = true: ONE launch per pass (decoupled look-back): a workgroup takes its tile from a ticket counter (so
// every lower tile is already running), publishes its digit counts, and thread d walks back over the lower
// tiles' status words of digit d until it meets a known prefix.  Used for the per-Gaussian depth sort, where
// P / 4096 ~ 500 tiles make the three helper launches per pass cost more than the pass itself; for the
// 10^7-instance tile sort the classic pass measured faster.
// NW = waves per workgroup (4 or 8) sharing one 4096-pair tile.  With returning-atomic ranking VALU issue is only
// 12 % of the pass (SQ counters) and 8 waves (twice the loads in flight per CU) did not help: at 331 MB algorithmic /
// 380 MB measured traffic in 95 us the pass moves ~4 TB/s of mixed reads and 64-128-byte write runs.
// Scatter of ONE workgroup tile whose TILE (key, value) pairs already sit in registers in scatter order -- wave w,
// step s, lane l holds element w * WTILE + 64 s + l of the tile -- shared by k_sort_scatter (pairs loaded from
// memory) and k_emit_scatter (pairs generated in place, preprocess.hip).
template <typename K, int NW, int TILE>
struct ScatterLds {
  uint32_t wcnt[NW][256];  // per-wave digit counts, then per-wave local write bases
  uint32_t gdelta[256];    // global position of local slot p holding digit d = gdelta[d] + p
  uint32_t wtot[NW];
  K lkey[TILE];
  uint32_t lval[TILE];
};

// COUNT (last pass of the instance sort, NW == 4): the sorted keys are not written at all; instead the pass counts
// the instances of every full key (= tile id) into key_count[key * 2 + 1] -- the .y of the tile's zeroed range --
// from which k_ranges_from_counts derives the per-tile ranges.  The input of the last pass is sorted by the lower
// digits and the pass is stable, so every digit group of the reordered workgroup tile is sorted by its low part
// and holds very few distinct ones: the digit's thread finds their boundaries by bisection and issues one global
// atomic per distinct key.  Saves the sorted-key write, the range kernel's read of it, and most of that kernel.
template <typename K, bool LB, bool ARANK, int NW, int TILE, bool COUNT = false>
__device__ __forceinline__ void scatter_core(ScatterLds<K, NW, TILE>& L, uint32_t (&key)[TILE / NW / 64],
                                             uint32_t (&val)[TILE / NW / 64], const int tile, K* __restrict__ keys_out,
                                             uint32_t* __restrict__ vals_out, const int n, const int shift,
                                             const int nbits, const uint32_t* __restrict__ counts,
                                             const uint32_t* __restrict__ chunk_base,
                                             const uint32_t* __restrict__ digit_total, uint32_t* __restrict__ status,
                                             uint32_t* __restrict__ key_count = nullptr) {
  constexpr int NT = 64 * NW, WTILE = TILE / NW, NSTEP = WTILE / 64;
  auto& wcnt = L.wcnt;
  auto& gdelta = L.gdelta;
  auto& wtot = L.wtot;
  auto& lkey = L.lkey;
  auto& lval = L.lval;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const size_t base = (size_t)tile * TILE + (size_t)w * WTILE;
  uint32_t lrank[NSTEP];
  const uint32_t mask = (1u << nbits) - 1u;
  // global exclusive base of every digit (every workgroup recomputes it: 1 KB, L2-resident)
  const uint32_t dbase = block_excl_scan_256(tid < 256 ? digit_total[tid] : 0u, lane, w, wtot);  // waves >= 4: unused
#pragma unroll
  for (int k = 0; k < 4; k++) wcnt[w][lane + 64 * k] = 0;
  if (LB) {  // publish this tile's digit counts as early as possible: the lower tiles' walks depend on them
    if (tid < 256) gdelta[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NSTEP; s++)
      if (base + (size_t)s * 64 + lane < (size_t)n) atomicAdd(&gdelta[(key[s] >> shift) & mask], 1u);
    __syncthreads();
    if (tid < 256)
      __hip_atomic_store(status + (size_t)tile * 256 + tid, (tile == 0 ? ST_GLOBAL : ST_LOCAL) | gdelta[tid],
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // pass A: rank of every element among the equal-digit elements of ITS WAVE that precede it
  volatile uint32_t* my = wcnt[w];
#pragma unroll
  for (int s = 0; s < NSTEP; s++) {
    const size_t i = base + (size_t)s * 64 + lane;
    const bool valid = i < (size_t)n;
    const uint32_t d = (key[s] >> shift) & mask;
    if (ARANK) {
      // One returning LDS add per step does the whole job of the ballot match below: the value returned to a lane
      // is the number of equal-digit elements of this wave that precede it, PROVIDED the LDS resolves the lanes
      // of one instruction that hit the same address in ascending lane order.  That order is not in the ISA
      // manual, so the host probes it once per process (k_probe_lds_atomic_order) and only then selects this
      // variant; the parity suite additionally checks the sorted lists bit for bit.  -17 % on the tile sort.
      lrank[s] = valid ? __hip_atomic_fetch_add(&wcnt[w][d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
      continue;
    }
    const uint64_t m = match_digit(d, valid, nbits);
    const uint32_t rank = (uint32_t)__popcll(m & lanemask_lt(lane));
    const uint32_t prior = my[d];  // same address inside a digit group: LDS broadcast
    // LDS operations of one wave execute in issue order: every lane has read `prior` before the group
    // leader publishes the advanced count for the next step.
    if (valid && rank == 0) my[d] = prior + (uint32_t)__popcll(m);
    lrank[s] = prior + rank;
  }
  __syncthreads();
  // per digit (thread = digit): tile-local base (exclusive over digits), per-wave bases, global delta
  uint32_t my_lbase = 0, my_cnt = 0;  // COUNT: this thread's digit group in the reordered tile
  {
    uint32_t cw[NW], c = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) {
      cw[k] = tid < 256 ? wcnt[k][tid] : 0u;
      c += cw[k];
    }
    const uint32_t lbase = block_excl_scan_256(c, lane, w, wtot);  // barriers inside: every thread calls it
    my_lbase = lbase;
    my_cnt = c;
    if (tid < 256) {
    uint32_t run = lbase;
#pragma unroll
    for (int k = 0; k < NW; k++) {
      wcnt[k][tid] = run;
      run += cw[k];
    }
    if (LB) {
      uint32_t* mine = status + (size_t)tile * 256 + tid;
      uint32_t excl = 0;
      if (tile != 0) {
        // walk back from tile - 1; tile 0 always ends the walk with a GLOBAL word.  LBK status words are
        // fetched per round trip (independent loads) and consumed in order up to the first unpublished one.
        int t = tile - 1;
        bool done = false;
        while (!done) {
          uint32_t v[LBK];
#pragma unroll
          for (int k = 0; k < LBK; k++)
            v[k] = t - k >= 0 ? __hip_atomic_load(status + (size_t)(t - k) * 256 + tid, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT)
                              : 0u;
          bool stalled = false;
#pragma unroll
          for (int k = 0; k < LBK; k++) {
            if (done || stalled) continue;
            if (v[k] & ST_GLOBAL) {
              excl += v[k] & ST_MASK;
              done = true;
            } else if (v[k] & ST_LOCAL) {
              excl += v[k] & ST_MASK;
              t--;
            } else {
              stalled = true;  // not published yet: look again from here
            }
          }
          if (stalled) __builtin_amdgcn_s_sleep(1);
        }
        __hip_atomic_store(mine, ST_GLOBAL | (excl + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      gdelta[tid] = dbase + excl - lbase;
    } else {
      const int chunk = tile / SORT_CHUNK;
      // (chunk_base == nullptr: the counts were scanned over all tiles in one launch, k_sort_scan_columns)
      gdelta[tid] = dbase + (chunk_base ? chunk_base[(size_t)chunk * 256 + tid] : 0u) + counts[(size_t)tile * 256 + tid] - lbase;
    }
    }
  }
  __syncthreads();
  // pass B: stable local reorder by digit
#pragma unroll
  for (int s = 0; s < NSTEP; s++) {
    const size_t i = base + (size_t)s * 64 + lane;
    if (i < (size_t)n) {
      const uint32_t d = (key[s] >> shift) & mask;
      const uint32_t p = wcnt[w][d] + lrank[s];
      lkey[p] = (K)key[s];
      lval[p] = val[s];
    }
  }
  __syncthreads();
  // write-out: consecutive local slots of one digit are consecutive in the output
  const size_t tile_base = (size_t)tile * TILE;
  const uint32_t nvalid = (size_t)n - tile_base < (size_t)TILE ? (uint32_t)((size_t)n - tile_base) : TILE;
  const uint32_t lomask = (1u << shift) - 1u;
#pragma unroll
  for (int k = 0; k < TILE / NT; k++) {
    const uint32_t p = (uint32_t)(k * NT + tid);
    if (p < nvalid) {
      const uint32_t kk = (uint32_t)lkey[p];
      const uint32_t g = gdelta[(kk >> shift) & mask] + p;
      if (!COUNT) keys_out[g] = (K)kk;
      vals_out[g] = lval[p];
    }
  }
  if (COUNT && tid < 256 && my_cnt) {
    // thread = digit: its group [my_lbase, my_lbase + my_cnt) of the reordered tile is sorted by the low part (the
    // pass is stable and its input was sorted by it), which takes very few distinct values inside one tile: count
    // each by bisection -- no atomics in LDS -- and add it to the key's global counter
    uint32_t prev = my_lbase;
    const uint32_t end = my_lbase + my_cnt;
    while (prev < end) {
      const uint32_t lo = (uint32_t)lkey[prev] & lomask;  // next distinct low part of this group
      uint32_t a = prev + 1, b = end;                     // first index with a larger low part
      while (a < b) {
        const uint32_t m = (a + b) >> 1;
        if (((uint32_t)lkey[m] & lomask) <= lo) a = m + 1; else b = m;
      }
      const uint32_t kk = ((uint32_t)tid << shift) | lo;
      (void)__hip_atomic_fetch_add(key_count + 2 * (size_t)kk + 1, a - prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      prev = a;
    }
  }
}


template <typename K, bool LB, bool ARANK, int NW, int TILE, bool COUNT = false>
__global__ __launch_bounds__(64 * NW) void k_sort_scatter(const K* __restrict__ keys_in,
                                                      const uint32_t* __restrict__ vals_in,
                                                      K* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                      const Count cnt, int shift, int nbits, const uint32_t* __restrict__ counts,
                                                      const uint32_t* __restrict__ chunk_base,
                                                      const uint32_t* __restrict__ digit_total,
                                                      uint32_t* __restrict__ status, uint32_t* __restrict__ ticket,
                                                      uint32_t* __restrict__ key_count = nullptr) {
  constexpr int WTILE = TILE / NW, NSTEP = WTILE / 64;
  __shared__ ScatterLds<K, NW, TILE> L;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (cnt.closed()) return;
  const int n = cnt.get();
  auto one_tile = [&](const int tile) {
    const size_t base = (size_t)tile * TILE + (size_t)w * WTILE;
    uint32_t key[NSTEP], val[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; s++) {
      const size_t i = base + (size_t)s * 64 + lane;
      const bool valid = i < (size_t)n;
      key[s] = valid ? (uint32_t)keys_in[i] : 0u;
      val[s] = !valid ? 0u : vals_in ? vals_in[i] : (uint32_t)i;  // (no value array: the values are the positions)
    }
    scatter_core<K, LB, ARANK, NW, TILE, COUNT>(L, key, val, tile, keys_out, vals_out, n, shift, nbits, counts, chunk_base,
                                                digit_total, status, key_count);
  };
  if (LB) {  // one ticketed tile per workgroup
    __shared__ int s_tile;
    // A pass whose digit takes ONE value over all the pairs is the identity (the sort is stable): every workgroup copies
    // its tile, nobody looks back.  The near candidates of a partial depth sort often share their keys' top byte (a
    // factor of four in depth): 14.5 -> 8 us for that pass at 2 M Gaussians / 1080p.
    if (__syncthreads_or(tid < 256 && n > 0 && digit_total[tid] == (uint32_t)n)) {
      const size_t base = (size_t)blockIdx.x * TILE + (size_t)w * WTILE;
#pragma unroll
      for (int s = 0; s < NSTEP; s++) {
        const size_t i = base + (size_t)s * 64 + lane;
        if (i < (size_t)n) {
          keys_out[i] = keys_in[i];
          vals_out[i] = vals_in ? vals_in[i] : (uint32_t)i;
        }
      }
      return;
    }
    if (tid == 0) s_tile = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    // (a grid sized for a capacity -- the near sort of a partial depth sort: tiles beyond the pairs have nothing to
    // do and nobody looks back at them)
    if ((size_t)s_tile * TILE < (size_t)n) one_tile(s_tile);
  } else {   // the tiles that hold pairs, grid stride (gsr_internal.hpp, for_each_unit)
    for_each_unit(units_of(n, TILE), one_tile);
  }
}

}  // namespace gsr
