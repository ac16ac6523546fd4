// render.hip -- per-tile alpha blending, forward (F8) and backward (B1), for gfx950 wave64.
//
// Replaces renderCUDA forward (reference forward.cu:291-407) and renderCUDA backward
// (reference backward.cu:438-603).
//
// Shape of both kernels (NOT the reference's 16x16-thread / 32-lane-warp structure):
//   * one 256-thread workgroup per 16x16 tile = 4 waves; wave w owns the 8x8 pixel QUAD
//     (w&1, w>>1) of the tile, lane l the pixel (l&7, l>>3) inside it;
//   * the tile's depth-sorted splat list is staged through LDS in chunks of 256: every thread
//     gathers ONE 48-B splat record (3 x dwordx4) and tests its exact-conservative footprint box
//     (hx, hy from the preprocess) against the four quads; four 64-bit wave ballots per loader wave
//     form per-quad hit masks in LDS -- the in-tile compaction;
//   * each wave then walks only the set bits of its own quad's masks (scalar bit scan, s_ff1), reading
//     the record by LDS broadcast, so a splat that cannot touch a quad costs that wave nothing.
//     Skipping is exact: a skipped (quad, splat) pair has alpha < 1/255 for every pixel, which the
//     reference discards as well.
// Forward stops per wave as soon as its 64 pixels are saturated and per tile when all four are.
// Backward emits, for each (tile, splat) instance, ONE 9-float gradient record into the splat's
// slot (plain 16-B stores, no atomics): DPP wave reduction over the 64 pixels of a quad, partials
// of the (up to) four quads combined in fixed order.  k_gaussian_backward sums a Gaussian's
// records.  The reference issues 9 global atomicAdds per (pixel, splat) instead (backward.cu:565,
// 591-600) and is run-to-run non-deterministic; this path is bitwise reproducible.
#include <cstdlib>

#include "gsr_internal.hpp"

namespace gsr {



template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_get(float v) {
  // lanes whose source is masked off / out of range read 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}

// Sum over the 64 lanes of a wave; the total is valid in lanes 48..63 (read it from lane 63).
// Fixed association order => deterministic.
__device__ __forceinline__ float wave_sum_to_hi(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0xB1, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x4E, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x124, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x128, 0xF, 0xF, true));
  // every lane of a 16-lane row now holds the row sum
  v += dpp_get<0x142, 0xA, 0xF>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_get<0x143, 0xC, 0xF>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}

// Sums of EIGHT per-lane values over the 64 lanes of a wave in 20 cross-lane ops instead of 48:
// v_permlane32_swap / v_permlane16_swap (gfx950) exchange half-waves / odd-even 16-lane rows between two
// registers, so one swap + one add halves the lane span of TWO values at once; four DPP steps finish
// the 16-lane rows.  On return w0 holds (in every lane of rows 0,1,2,3) the totals of v0,v2,v4,v6 and
// w1 those of v1,v3,v5,v7.  Fixed association order => deterministic.
__device__ __forceinline__ void swap_add32(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]) + __uint_as_float(r[1]);  // lanes 0-31: a[l]+a[l+32]; lanes 32-63: b[l-32]+b[l]
}
__device__ __forceinline__ void swap_add16(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]) + __uint_as_float(r[1]);  // rows 0,2: a's row pairs; rows 1,3: b's row pairs
}
// quad_perm / row_ror read a valid lane for every lane, so `old` is irrelevant: passing the source itself lets
// the compiler fold the move into a single v_add_f32_dpp (no zero-initialised temporary).
template <int CTRL>
__device__ __forceinline__ float dpp_full(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_allsum(float v) {
  v += dpp_full<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_full<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_full<0x124>(v);  // row_ror:4
  v += dpp_full<0x128>(v);  // row_ror:8
  return v;
}
__device__ __forceinline__ void wave_sum8(float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                          float v7, float& w0, float& w1) {
  swap_add32(v0, v4);  // v0: lo = v0 partials, hi = v4 partials
  swap_add32(v1, v5);
  swap_add32(v2, v6);
  swap_add32(v3, v7);
  swap_add16(v0, v2);  // v0 rows: v0, v2, v4, v6
  swap_add16(v1, v3);  // v1 rows: v1, v3, v5, v7
  w0 = row_allsum(v0);
  w1 = row_allsum(v1);
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t t = (uint32_t)__shfl_xor((int)v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}

__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// Exact-conservative ellipse-vs-quad test.  f(u, v) = 1/2 (A u^2 + C v^2) + B u v = -power of a pixel at offset (u, v)
// from the splat centre; the pixel centres of a quad fill [x0, x0 + 7] x [y0, y0 + 7].  f is convex (the caller
// only asks for positive-definite conics), so its minimum over the rectangle is 0 if the centre lies inside and
// otherwise sits on one of the four edges, where it is a clamped 1-D minimisation.  min f > tau means that no pixel
// of the quad can reach alpha >= 1/255 (tau carries the slack, footprint_tau): the quad skips the splat exactly as
// the reference skips it pixel by pixel (forward.cu:374-376, backward.cu:476-478).  The footprint BOX alone lets
// ~25 % of the (quad, splat) pairs through that this test rejects (corners of slanted ellipses).
__device__ __forceinline__ bool ellipse_reaches_quad(float cx, float cy, float A, float B, float C, float tau,
                                                     float x0, float y0) {
  const float u0 = x0 - cx, u1 = u0 + 7.0f, v0 = y0 - cy, v1 = v0 + 7.0f;
  if (u0 <= 0.0f && u1 >= 0.0f && v0 <= 0.0f && v1 >= 0.0f) return true;
  const float nBiC = -B * __builtin_amdgcn_rcpf(C), nBiA = -B * __builtin_amdgcn_rcpf(A);
  float fmin = 3.0e38f;
#pragma unroll
  for (int e = 0; e < 2; e++) {
    const float ue = e ? u1 : u0;
    const float v = fminf(fmaxf(nBiC * ue, v0), v1);
    fmin = fminf(fmin, 0.5f * (A * ue * ue + C * v * v) + B * ue * v);
    const float ve = e ? v1 : v0;
    const float u = fminf(fmaxf(nBiA * ve, u0), u1);
    fmin = fminf(fmin, 0.5f * (A * u * u + C * ve * ve) + B * u * ve);
  }
  return fmin <= tau;
}

// hit test of one splat against one quad: footprint box first (hx < 0: never, hx >= 1e6: indefinite conic, no
// culling), then the ellipse itself
__device__ __forceinline__ bool splat_hits_quad(const float4 a, const float4 b, const float4 c, float x0, float y0) {
  const bool box = (a.x + c.z >= x0) && (a.x - c.z <= x0 + 7.0f) && (a.y + c.w >= y0) && (a.y - c.w <= y0 + 7.0f);
  if (!box || c.z >= 1.0e6f) return box;
  return ellipse_reaches_quad(a.x, a.y, a.z, a.w, b.x, footprint_tau(b.y), x0, y0);
}

// Footprint box of a splat vs the four 8x8 quads of the tile at (tx0, ty0); bit q set = may touch.
__device__ __forceinline__ uint32_t quad_hits(float x, float y, float hx, float hy, float tx0, float ty0) {
  const float xl = x - hx, xh = x + hx, yl = y - hy, yh = y + hy;
  const bool cx0 = (xh >= tx0) && (xl <= tx0 + 7.0f);
  const bool cx1 = (xh >= tx0 + 8.0f) && (xl <= tx0 + 15.0f);
  const bool cy0 = (yh >= ty0) && (yl <= ty0 + 7.0f);
  const bool cy1 = (yh >= ty0 + 8.0f) && (yl <= ty0 + 15.0f);
  return (uint32_t)(cx0 && cy0) | ((uint32_t)(cx1 && cy0) << 1) | ((uint32_t)(cx0 && cy1) << 2) |
         ((uint32_t)(cx1 && cy1) << 3);
}

// ------------------------------------------------------------------------------------------------
// F8 forward blend.  ONE WAVE PER 8x8 QUAD, no workgroup barriers at all: the four quads of a tile are four
// independent 64-thread workgroups (consecutive block ids, so they share the tile's splats in L2).  A wave
// streams the tile's list in sub-chunks of 64 -- each lane gathers one 48-B record (the next sub-chunk's gather
// is issued before the current one is consumed), the footprint-box test against ITS quad and one ballot give
// the hit mask, records go to a 3-KB wave-private LDS image and are read back by broadcast -- and stops the
// moment its 64 pixels are saturated.  The redundant gathers (each record is fetched by up to four waves) hit
// L2; in exchange no wave ever waits for a slower quad, which the SQ counters showed to be the dominant cost
// of a barrier-per-chunk design (VALU active 14 % of wave cycles).
// quad_last[4*tile + q] = max n_contrib inside the quad: bounds the backward walk.
// ------------------------------------------------------------------------------------------------
template <int FW>  // quads (= waves) per workgroup: the waves never synchronise, FW only sets how many share a workgroup slot
__global__ __launch_bounds__(64 * FW) void k_blend_forward(const FrameParams fp, const uint2* __restrict__ ranges,
                                                      const uint32_t* __restrict__ point_list,
                                                      const float4* __restrict__ splats,
                                                      const float* __restrict__ bg, float* __restrict__ final_T,
                                                      uint32_t* __restrict__ n_contrib,
                                                      uint32_t* __restrict__ quad_last, float* __restrict__ out_color,
                                                      float* __restrict__ out_depth, float* __restrict__ out_acc) {
  __shared__ float4 sAll[FW][3][64];  // wave-private images: no barrier anywhere in this kernel
  const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
  float4* sA = sAll[wq][0];
  float4* sB = sAll[wq][1];
  float4* sC = sAll[wq][2];
  const int quad = blockIdx.x * FW + wq;
  if (quad >= fp.gx * fp.gy * 4) return;
  const int tile = quad >> 2, q = quad & 3;
  const int tile_x = tile % fp.gx, tile_y = tile / fp.gx;
  const int qx = tile_x * TILE + (q & 1) * 8, qy = tile_y * TILE + (q >> 1) * 8;
  const int px = qx + (lane & 7), py = qy + (lane >> 3);
  const bool inside = px < fp.W && py < fp.H;
  const float pfx = (float)px, pfy = (float)py;
  const float qx0 = (float)qx, qy0 = (float)qy;
  const uint2 range = ranges[tile];
  const int n = (int)(range.y - range.x);

  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, A = 0.f;
  uint32_t last = 0;
  bool done = !inside;
  bool wave_done = __ballot(!done) == 0ull;

  // software pipeline: (a, b, c, hit) hold the sub-chunk about to be consumed
  float4 a = make_float4(0, 0, 0, 0), b = a, c = a;
  bool hit = false;
  if (!wave_done && lane < n) {
    const uint32_t id = point_list[range.x + lane];
    a = splats[(size_t)id * SPLAT_F4 + 0];
    b = splats[(size_t)id * SPLAT_F4 + 1];
    c = splats[(size_t)id * SPLAT_F4 + 2];
    hit = splat_hits_quad(a, b, c, qx0, qy0);
  }
  for (int base = 0; base < n && !wave_done; base += 64) {
    // published conic terms carry the constant factors of power = -1/2 (A dx^2 + C dy^2) - B dx dy and the
    // log2(e) of exp(x) = exp2(x log2 e): one multiply per ENTRY here instead of three per (entry, pixel) visit
    constexpr float L2E = 1.4426950408889634f;
    sA[lane] = make_float4(a.x, a.y, a.z * (-0.5f * L2E), a.w * (-L2E));
    sB[lane] = make_float4(b.x * (-0.5f * L2E), b.y, b.z, b.w);
    sC[lane] = c;
    uint64_t m = __ballot(hit);
    // issue the next sub-chunk's gather now; it completes while this one is blended
    const int jn = base + 64 + lane;
    hit = false;
    if (jn < n) {
      const uint32_t id = point_list[range.x + jn];
      a = splats[(size_t)id * SPLAT_F4 + 0];
      b = splats[(size_t)id * SPLAT_F4 + 1];
      c = splats[(size_t)id * SPLAT_F4 + 2];
      hit = splat_hits_quad(a, b, c, qx0, qy0);
    }
    // Visit loop, software-pipelined by hand: the LDS broadcast reads of the NEXT hit are issued before the
    // current hit is blended (two register sets, no copies), so their latency hides behind ~30 VALU ops.
    auto blend = [&](const float4 ra, const float4 rb, const float4 rc, const int jj) {
      const float dx = ra.x - pfx, dy = ra.y - pfy;
      const float power = ra.z * (dx * dx) + rb.x * (dy * dy) + ra.w * (dx * dy);  // = log2(e) x the reference's power
      const float alpha = fminf(0.99f, rb.y * __builtin_amdgcn_exp2f(power));
      bool ok = !done && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
      const float test_T = T * (1.0f - alpha);
      const bool stop = ok && (test_T < 0.0001f);
      done = done || stop;
      ok = ok && !stop;
      const float wgt = ok ? alpha * T : 0.0f;
      C0 += rb.z * wgt;
      C1 += rb.w * wgt;
      C2 += rc.x * wgt;
      Dp += rc.y * wgt;
      A += wgt;
      T = ok ? test_T : T;
      last = ok ? (uint32_t)(base + jj + 1) : last;
    };
    if (m) {
      int j0 = __builtin_ctzll(m), j1;
      m &= m - 1;
      float4 a0 = sA[j0], b0 = sB[j0], c0 = sC[j0], a1, b1, c1;
      for (;;) {
        j1 = -1;
        if (m) {
          j1 = __builtin_ctzll(m);
          m &= m - 1;
          a1 = sA[j1]; b1 = sB[j1]; c1 = sC[j1];
        }
        blend(a0, b0, c0, j0);
        if (j1 < 0) {
          wave_done = __ballot(!done) == 0ull;
          break;
        }
        j0 = -1;
        if (m) {
          j0 = __builtin_ctzll(m);
          m &= m - 1;
          a0 = sA[j0]; b0 = sB[j0]; c0 = sC[j0];
        }
        blend(a1, b1, c1, j1);
        wave_done = __ballot(!done) == 0ull;  // checked once per pair of visits: a visit after saturation changes nothing
        if (wave_done || j0 < 0) break;
      }
    }
  }

  const uint32_t wl = wave_max_u32(inside ? last : 0u);
  if (lane == 0) quad_last[quad] = wl;
  if (inside) {
    const size_t pid = (size_t)fp.W * py + px;
    const size_t N = (size_t)fp.W * fp.H;
    final_T[pid] = T;
    n_contrib[pid] = last;
    out_color[pid] = C0 + T * bg[0];
    out_color[N + pid] = C1 + T * bg[1];
    out_color[2 * N + pid] = C2 + T * bg[2];
    out_depth[pid] = Dp;
    out_acc[pid] = A;
  }
}

// ------------------------------------------------------------------------------------------------
// B1 backward blend.
// ------------------------------------------------------------------------------------------------
// BCHUNK = list entries staged per round (one per thread of the first BCHUNK/64 waves).  128 instead of 256
// halves the LDS footprint (the per-quad partial sums dominate it), doubling the resident workgroups per CU:
// the kernel is latency-bound at 3 waves/SIMD (SQ counters: VALU issue active 36 % of wave cycles).
template <int BCHUNK>
__global__ __launch_bounds__(256) void k_blend_backward(
    const FrameParams fp, const uint2* __restrict__ ranges, const uint32_t* __restrict__ quad_last_in,
    const uint32_t* __restrict__ point_list, const float4* __restrict__ splats, const uint2* __restrict__ slotinfo,
    const float* __restrict__ bg, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dpix, const float* __restrict__ dL_dacc, float4* __restrict__ grad_inst,
    uint8_t* __restrict__ inst_flag, uint8_t* __restrict__ touched) {
  constexpr int LW = BCHUNK / 64;  // loader waves
  // one 48-byte image per staged entry -- (x, y, conic.x', conic.y' | conic.z', opacity, r, g | b, conic) with the
  // primed terms pre-scaled for exp2 -- so a visit
  // addresses all of it from ONE register (base + 48 jj) with immediate offsets
  __shared__ float4 sE[BCHUNK][3];
  __shared__ uint32_t sSlot[BCHUNK], sId[BCHUNK];
  __shared__ uint64_t smask[4][LW];
  // per (quad, entry): four (pair of wave totals, 16-lane-row partial of the ninth value) triples, one per row of the
  // wave: the row partials are added in the combine step (once per instance) instead of two more cross-lane steps
  // per visit
  __shared__ __attribute__((aligned(16))) float sPart[4][BCHUNK][12];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tile = blockIdx.y * fp.gx + blockIdx.x;
  const uint32_t ql0 = quad_last_in[4 * tile], ql1 = quad_last_in[4 * tile + 1], ql2 = quad_last_in[4 * tile + 2],
                 ql3 = quad_last_in[4 * tile + 3];
  const int n = (int)max(max(ql0, ql1), max(ql2, ql3));  // entries [0, n) of the tile's list can carry gradient
  if (n == 0) return;
  const uint32_t rbase = ranges[tile].x;
  const int px = blockIdx.x * TILE + (w & 1) * 8 + (lane & 7);
  const int py = blockIdx.y * TILE + (w >> 1) * 8 + (lane >> 3);
  const bool inside = px < fp.W && py < fp.H;
  const float pfx = (float)px, pfy = (float)py;
  const float tx0 = (float)(blockIdx.x * TILE), ty0 = (float)(blockIdx.y * TILE);
  const size_t pid = (size_t)fp.W * py + px, N = (size_t)fp.W * fp.H;

  const float T_final = inside ? final_T[pid] : 0.0f;
  const int lastc = inside ? (int)n_contrib[pid] : 0;
  const float dp0 = inside ? dL_dpix[pid] : 0.f, dp1 = inside ? dL_dpix[N + pid] : 0.f,
              dp2 = inside ? dL_dpix[2 * N + pid] : 0.f;
  const float dacc = inside ? dL_dacc[pid] : 0.f;  // the reference reads this unguarded (backward.cu:497)
  const float bg_dot = bg[0] * dp0 + bg[1] * dp1 + bg[2] * dp2;
  const float neg_Tf_bg = -T_final * bg_dot;  // per-pixel constant of the background term (backward.cu:578-581)
  const float ddelx_dx = 0.5f * (float)fp.W, ddely_dy = 0.5f * (float)fp.H;
  float T = T_final;
  float ar0 = 0.f, ar1 = 0.f, ar2 = 0.f;  // accum_rec
  float nacc = 1.0f;                      // 1 - accum_acc_rec: the form dL_dalpha uses; its update is one multiply

  // this lane's pair slot (first 8 floats) and row slot (last 4) inside an entry's 12 floats
  // (row r of the wave owns floats [3r, 3r + 2]: its pair of wave totals and its share of the ninth value -- one
  // 12-byte store per visit from one address register)
  float* const my_part = &sPart[w][0][0] + 3 * (lane >> 4);
  for (int base = 0; base < n; base += BCHUNK) {
    const int k = base + tid;  // k-th entry counted from the back of [0, n)
    const bool stager = tid < BCHUNK;
    uint32_t hits = 0;
    if (stager && k < n) {
      const int pos = n - 1 - k;
      const uint32_t id = point_list[rbase + pos];
      const float4 a = splats[(size_t)id * SPLAT_F4 + 0];
      const float4 b = splats[(size_t)id * SPLAT_F4 + 1];
      const float4 c = splats[(size_t)id * SPLAT_F4 + 2];
      const uint2 si = slotinfo[id];
      const int x0 = (int)(si.y & 1023u), y0 = (int)((si.y >> 10) & 1023u), rw = (int)(si.y >> 20);
      // the visits read the conic pre-scaled (power's -1/2 and the log2(e) of exp(x) = exp2(x log2 e): one multiply per
      // ENTRY here instead of two instructions per visit); the combine step reads the plain conic from the third slot
      constexpr float L2E = 1.4426950408889634f;
      sE[tid][0] = make_float4(a.x, a.y, a.z * (-0.5f * L2E), a.w * (-L2E));
      sE[tid][1] = make_float4(b.x * (-0.5f * L2E), b.y, b.z, b.w);
      sE[tid][2] = make_float4(c.x, a.z, a.w, b.x);
      sId[tid] = id;
      sSlot[tid] = si.x + (uint32_t)(((int)blockIdx.y - y0) * rw + ((int)blockIdx.x - x0));
      hits = quad_hits(a.x, a.y, c.z, c.w, tx0, ty0);
      if (hits && c.z < 1.0e6f) {  // box passed: ask the ellipse itself, quad by quad
        const float tau = footprint_tau(b.y);
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (((hits >> q) & 1u) &&
              !ellipse_reaches_quad(a.x, a.y, a.z, a.w, b.x, tau, tx0 + (float)((q & 1) * 8), ty0 + (float)((q >> 1) * 8)))
            hits &= ~(1u << q);
      }
      // entries at or beyond a quad's own last contributor cannot receive gradient from that quad
      hits &= (pos < (int)ql0 ? 1u : 0u) | (pos < (int)ql1 ? 2u : 0u) | (pos < (int)ql2 ? 4u : 0u) |
              (pos < (int)ql3 ? 8u : 0u);
    }
    if (w < LW) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint64_t m = __ballot((hits >> q) & 1u);
        if (lane == 0) smask[q][w] = m;
      }
    }
    __syncthreads();
    for (int lw = 0; lw < LW; lw++) {
      uint64_t m = uniform_u64(smask[w][lw]);
      while (m) {
        const int bpos = __builtin_ctzll(m);
        m &= m - 1;
        const int jj = lw * 64 + bpos;
        const int pos = n - 1 - (base + jj);  // 0-based index in the tile list == `contributor` after decrement
        const float4 a = sE[jj][0];
        const float4 b = sE[jj][1];
        const float blue = sE[jj][2].x;
        const float dx = a.x - pfx, dy = a.y - pfy;
        const float power = a.z * (dx * dx) + b.x * (dy * dy) + a.w * (dx * dy);  // = log2(e) x the reference's power
        const float Graw = __builtin_amdgcn_exp2f(power);
        const float araw = fminf(0.99f, b.y * Graw);
        const bool ok = (pos < lastc) && !(power > 0.0f) && !(araw < 1.0f / 255.0f);
        // Branch-free per lane: a lane that does not take this splat runs the same code with alpha = 0 and G = 0.
        // Every gradient term carries a factor alpha or G, so its partials are exact zeros, and the recurrence state
        // is left untouched bit for bit (accum + 0 * d = accum, T * rcp(1 - 0) = T) -- identical to not having
        // visited, without the save/restore traffic a divergent `if (ok)` costs.  The state update itself sits
        // OUTSIDE the wave-uniform "any lane takes it" branch below: inside it the compiler routes the five state
        // registers through copies at the loop join (5-10 v_mov per visit); as straight-line code they are updated
        // in place.
        const float alpha = ok ? araw : 0.0f;
        const float oma = 1.0f - alpha;
        const float rom = __builtin_amdgcn_rcpf(oma);
        const float Tn = T * rom;  // T / (1 - alpha)
        const float d0 = b.z - ar0, d1 = b.w - ar1, d2 = blue - ar2, da = nacc;
        if (__ballot(ok) != 0ull) {
          const float G = ok ? Graw : 0.0f;
          const float dch = alpha * Tn;
          float dL_dalpha = d0 * dp0 + d1 * dp1 + d2 * dp2 + da * dacc;
          dL_dalpha *= Tn;
          dL_dalpha += rom * neg_Tf_bg;
          // Factors common to every pixel of the splat (opacity, -0.5, 0.5*W, 0.5*H) are applied once per
          // instance when the four quads are combined, not per pixel.
          const float dLG = G * dL_dalpha;  // dL/dG up to the opacity factor; also the opacity partial itself
          const float sx = dLG * dx, sy = dLG * dy;
          const float g0 = dch * dp0, g1 = dch * dp1, g2 = dch * dp2;
          // dL/dmean2D = -(conic . (sum sx, sum sy)): the conic is the same for every pixel of the splat, so the quads
          // reduce sx and sy themselves and the 2x2 product is taken once per instance in the combine step
          const float g3 = sx, g4 = sy;
          const float g5 = sx * dx, g6 = sx * dy, g7 = sy * dy;
          float g8 = dLG;
          float w0, w1;
          wave_sum8(g0, g1, g2, g3, g4, g5, g6, g7, w0, w1);
          g8 = row_allsum(g8);
          // pin the sums here: otherwise the compiler sinks the last add of each tree into the 4-lane store block
          // below and leaves a v_mov_dpp + v_add pair where one v_add_dpp does (3 instructions per visit)
          asm volatile("" : "+v"(w0), "+v"(w1), "+v"(g8));
          if ((lane & 15) == 0) {  // lanes 0,16,32,48 hold the totals of (g0,g1),(g2,g3),(g4,g5),(g6,g7)
            float* const dst = my_part + 12u * (uint32_t)jj;  // (w0, w1, this row's share of the opacity partial)
            dst[0] = w0; dst[1] = w1; dst[2] = g8;
          }
        } else if ((lane & 15) == 0) {
          float* const dst = my_part + 12u * (uint32_t)jj;
          dst[0] = 0.f; dst[1] = 0.f; dst[2] = 0.f;
        }
        // Fold this splat into the "everything behind the next one" accumulators NOW (the reference does it at the
        // top of its next iteration from saved (last_alpha, last_color), backward.cu:533-543), in the form
        // accum + alpha (c - accum) == alpha c + (1 - alpha) accum: the differences are the ones dL_dalpha used and
        // each update is one in-place fma.
        T = Tn;
        ar0 = __builtin_fmaf(alpha, d0, ar0);
        ar1 = __builtin_fmaf(alpha, d1, ar1);
        ar2 = __builtin_fmaf(alpha, d2, ar2);
        nacc = nacc * oma;  // 1 - (acc + alpha (1 - acc)) = (1 - acc)(1 - alpha)
      }
    }
    __syncthreads();
    if (stager && k < n) {
      // hit bits of entry `tid`, quad order 0..3 fixed => reproducible sums
      const int lw = tid >> 6;
      float s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      bool any = false;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if ((smask[q][lw] >> (tid & 63)) & 1ull) {
          any = true;
#pragma unroll
          for (int r = 0; r < 4; r++) {  // row r carries the wave totals of values 2r, 2r + 1
            s[2 * r] += sPart[q][tid][3 * r];
            s[2 * r + 1] += sPart[q][tid][3 * r + 1];
          }
          s[8] += (sPart[q][tid][2] + sPart[q][tid][5]) + (sPart[q][tid][8] + sPart[q][tid][11]);
        }
      }
      if (any) {
        const size_t slot = sSlot[tid];
        const float4 cc = sE[tid][2];  // (blue, conic.x, conic.y, conic.z): the plain conic
        const float op = sE[tid][1].y;  // dL/dG = opacity * dL/dalpha; conic terms carry -0.5 (backward.cu:583-597)
        const float mx = op * ddelx_dx, my = op * ddely_dy, mc = -0.5f * op;
        const float gx = -(cc.y * s[3] + cc.z * s[4]);  // dG_ddelx, dG_ddely summed over the pixels (backward.cu:561-562)
        const float gy = -(cc.w * s[4] + cc.z * s[3]);
        grad_inst[slot * GRAD_F4 + 0] = make_float4(s[0], s[1], s[2], gx * mx);
        grad_inst[slot * GRAD_F4 + 1] = make_float4(gy * my, s[5] * mc, s[6] * mc, s[7] * mc);
        grad_inst[slot * GRAD_F4 + 2] = make_float4(s[8], 0.f, 0.f, 0.f);
        inst_flag[slot] = 1;
        touched[sId[tid]] = 1;  // same value from every writer: a benign race
      }
    }
    __syncthreads();
  }
}

hipError_t launch_blend_forward(const FrameParams& fp, GeomState g, BinningState b, ImageState im, const float* bg,
                                float* out_color, float* out_depth, float* out_acc, hipStream_t s) {
  ProfScope ps_k_blend_fwd(K_BLEND_FWD, s);
  // the four quads of a tile share a workgroup slot (they never synchronise): their redundant gathers of the same
  // records coincide in time and hit L1/L2; 1, 2 and 4 waves per workgroup measured within 3 % of each other
  const int quads = fp.gx * fp.gy * 4;
  hipLaunchKernelGGL(k_blend_forward<4>, dim3((quads + 3) / 4), dim3(256), 0, s, fp, im.ranges, b.point_list, g.splats,
                     bg, im.final_T, im.n_contrib, im.quad_last, out_color, out_depth, out_acc);
  return hipGetLastError();
}

hipError_t launch_blend_backward(const FrameParams& fp, GeomState g, BinningState b, ImageState im, const float* bg,
                                 const float* dL_dpix, const float* dL_dacc, hipStream_t s) {
  ProfScope ps_k_blend_bwd(K_BLEND_BWD, s);
  // chunks of 128 list entries (64 and 128 measured equal, 256 slower: LDS footprint)
  hipLaunchKernelGGL(k_blend_backward<128>, dim3(fp.gx, fp.gy), dim3(256), 0, s, fp, im.ranges, im.quad_last, b.point_list,
                     g.splats, g.slotinfo, bg, im.final_T, im.n_contrib, dL_dpix, dL_dacc, b.grad_inst, b.inst_flag,
                     g.touched);
  return hipGetLastError();
}

}  // namespace gsr
