// render.hip -- per-tile alpha blending, forward (F8) and backward (B1), for gfx950 wave64.
//
// Replaces renderCUDA forward (reference forward.cu:291-407) and renderCUDA backward
// (reference backward.cu:438-603).
//
// Shapes (NOT the reference's 16x16-thread / 32-lane-warp structure; details at each kernel):
//   * forward, k_blend_forward: ONE WAVE PER 8x8 PIXEL QUAD, no workgroup barrier in the blending; a wave streams its
//     tile's depth-sorted list 64 entries at a time -- one 48-B splat record per lane, exact ellipse-vs-quad test, one
//     ballot, records published to a wave-private LDS image -- and walks only the set bits (scalar bit scan, LDS
//     broadcast), so a splat that cannot touch the quad costs it nothing.  Skipping is exact: a skipped pair has
//     alpha < 1/255 for every pixel, which the reference discards as well.  A wave stops when its 64 pixels have.
//     Near/far frames (api.hip) run it as phase 1 (near segment; unfinished quads park their state; the launch's last
//     workgroup counts them and, in an asynchronous frame, opens the far chain or releases the caller's stream) and
//     phase 2 (far segment, gated);
//   * backward, k_blend_backward_tile (frames of >= 3072 tiles): ONE WAVE PER 16x16 TILE, four pixels per lane, no
//     barriers: the nine partial sums of a (tile, splat) pair are accumulated over the lane's pixels first and reduced
//     across the wave once (v_permlane32_swap / v_permlane16_swap + DPP); four lanes store the raw sums as ONE record
//     into the splat's slot (plain stores, no atomics).  k_blend_backward (smaller frames): workgroup per tile, wave
//     per quad, per-quad partials combined in fixed order.  Both take the tiles longest walk first (k_tile_order).
//   k_gather_records (preprocess.hip) sums a Gaussian's records.  The reference issues 9 global atomicAdds per
//   (pixel, splat) instead (backward.cu:565, 591-600) and is run-to-run non-deterministic; this path is bitwise
//   reproducible.
// Also here: k_live_sat (which tiles the near chain left unfinished), k_release_go (end of an asynchronous far chain).
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
// three 16-lane-row sums with the steps of the three chains interleaved: a DPP instruction that reads the VGPR the
// previous VALU instruction wrote needs two wait states, which the neighbouring chains fill
__device__ __forceinline__ void row_allsum3(float& a, float& b, float& c) {
  a += dpp_full<0xB1>(a); b += dpp_full<0xB1>(b); c += dpp_full<0xB1>(c);     // quad_perm [1,0,3,2]
  a += dpp_full<0x4E>(a); b += dpp_full<0x4E>(b); c += dpp_full<0x4E>(c);     // quad_perm [2,3,0,1]
  a += dpp_full<0x124>(a); b += dpp_full<0x124>(b); c += dpp_full<0x124>(c);  // row_ror:4
  a += dpp_full<0x128>(a); b += dpp_full<0x128>(b); c += dpp_full<0x128>(c);  // row_ror:8
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

// log2(e) x the reference's power = -1/2 (A dx^2 + C dy^2) - B dx dy of a pixel at offset (dx, dy) from the splat
// centre (forward.cu:362-366), from the pre-scaled conic terms A' = -1/2 log2(e) A, B' = -log2(e) B,
// C' = -1/2 log2(e) C.  ONE fixed sequence of roundings -- A' dx^2 + dy (C' dy + B' dx), explicit fmas -- shared by the
// forward and the backward blend so that both take bit-identical alpha / skip decisions for every (pixel, splat)
// pair; the backward evaluates it for four pixels of one column per lane and shares dx, dx^2 and A' dx^2 among them.
__device__ __forceinline__ float splat_power_shared(float Adx2, float Bp, float Cp, float dx, float dy) {
  const float Cdy = Cp * dy;
  const float t = __builtin_fmaf(Bp, dx, Cdy);
  return __builtin_fmaf(t, dy, Adx2);
}
__device__ __forceinline__ float splat_power(float Ap, float Bp, float Cp, float dx, float dy) {
  const float dx2 = dx * dx;
  return splat_power_shared(Ap * dx2, Bp, Cp, dx, dy);
}

// Exact-conservative ellipse-vs-quad test.  f(u, v) = 1/2 (A u^2 + C v^2) + B u v = -power of a pixel at offset (u, v)
// from the splat centre; the pixel centres of a quad fill [x0, x0 + 7] x [y0, y0 + 7].  f is convex (the caller
// only asks for positive-definite conics), so its minimum over the rectangle is 0 if the centre lies inside and
// otherwise sits on one of the four edges, where it is a clamped 1-D minimisation.  min f > tau means that no pixel
// of the quad can reach alpha >= 1/255 (tau carries the slack, footprint_tau): the quad skips the splat exactly as
// the reference skips it pixel by pixel (forward.cu:374-376, backward.cu:476-478).  The footprint BOX alone lets
// ~25 % of the (quad, splat) pairs through that this test rejects (corners of slanted ellipses).
template <int EXTENT = 7>  // the rectangle of pixel centres is [x0, x0 + EXTENT] x [y0, y0 + EXTENT]: 7 = quad, 15 = tile
__device__ __forceinline__ bool ellipse_reaches_quad(float cx, float cy, float A, float B, float C, float tau,
                                                     float x0, float y0) {
  const float u0 = x0 - cx, u1 = u0 + (float)EXTENT, v0 = y0 - cy, v1 = v0 + (float)EXTENT;
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
// PHASE (near/far frames, api.hip): 0 = the whole list in one launch.  1 = the NEAR segment of the tile's list
// (`ranges`): a quad all of whose pixels have stopped is finished as in phase 0 and marked quad_done; any other quad
// parks its pixels' running state in the output arrays (colour WITHOUT the background term, T, depth and silhouette
// sums, last contributor with bit 31 = "this pixel has stopped").  2 = the FAR segment (`rangesB`), only for the quads
// phase 1 left unfinished: the state is picked up, list positions continue behind the near segment's, and the pixels
// are finished.  The arithmetic per pixel is the same sequence either way: images, n_contrib and final_T are bitwise
// those of a single launch over the concatenated list.
constexpr uint32_t STOPPED_BIT = 0x80000000u;

// Lane select by an explicit 64-bit lane mask held in scalar registers: bit set -> a, else b (one v_cndmask_b32).  The
// forward blend keeps its per-pixel predicates (done, take, stop, ok) as such masks and combines them with a handful of
// scalar ALU instructions; as loop-carried `bool`s the compiler turned every wave-wide test of them into a
// v_cndmask 0/1 + v_cmp + s_cmp round trip and every `!x` of a float compare into a second compare.
__device__ __forceinline__ float sel_f(uint64_t m, float a, float b) {
  float d;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(a), "s"(m));
  return d;
}
__device__ __forceinline__ float sel_f0(uint64_t m, float a) {  // bit set -> a, else 0
  float d;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(d) : "v"(a), "s"(m));
  return d;
}
__device__ __forceinline__ uint32_t sel_u(uint64_t m, uint32_t a, uint32_t b) {
  uint32_t d;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(a), "s"(m));
  return d;
}

#ifndef GSR_FWD_ASM_VISIT
#define GSR_FWD_ASM_VISIT 1
#endif
// slots of a wave's LDS image: 64 hits, +1 null entry padding an odd count, +1 read ahead by the walk
constexpr int FWD_SLOTS = GSR_FWD_ASM_VISIT ? 66 : 64;

template <int FW, int PHASE>  // FW quads (= waves) per workgroup: the waves never synchronise, FW only sets how many share a slot
__global__ __launch_bounds__(64 * FW) void k_blend_forward(const FrameParams fp, const uint2* __restrict__ ranges,
                                                      const uint2* __restrict__ rangesB,
                                                      const uint32_t* __restrict__ point_list,
                                                      const float4* __restrict__ splats,
                                                      const float* __restrict__ bg, float* __restrict__ final_T,
                                                      uint32_t* __restrict__ n_contrib,
                                                      uint32_t* __restrict__ quad_last, uint8_t* __restrict__ quad_done,
                                                      float* __restrict__ out_color,
                                                      float* __restrict__ out_depth, float* __restrict__ out_acc,
                                                      unsigned long long* __restrict__ done_word,
                                                      unsigned long long* __restrict__ publish, const uint32_t ticket,
                                                      uint32_t* __restrict__ live_quads_out, const AsyncWords aw,
                                                      const Count gate) {
  if (PHASE == 2 && gate.closed()) return;  // (asynchronous frame whose near chain finished every quad)
  __shared__ float4 sAll[FW][3][FWD_SLOTS];  // wave-private images: no barrier in the blending itself
  // PHASE 1 counts the quads it leaves unfinished and the launch's last workgroup hands the total to the host (api.hip:
  // a frame whose far chain was not enqueued is complete iff that total is zero).  Finished waves per workgroup and
  // their unfinished quads travel in one LDS word, finished workgroups and the running total in one 64-bit global word
  // (as k_preprocess counts its instances): no fence, one same-address global atomic per workgroup.
  __shared__ uint32_t s_done;
  if (PHASE == 1) {
    if (threadIdx.x == 0) s_done = 0u;
    __syncthreads();  // the only barrier: at the start, where no wave waits for a slower one
  }
  // (wave-uniform by construction; made scalars by hand, see k_blend_backward_tile)
#ifndef GSR_FWD_EXEC_MASK
#define GSR_FWD_EXEC_MASK 1
#endif
#ifndef GSR_FWD_SCALAR
#define GSR_FWD_SCALAR 0  // (scalarising the range / wave index by hand measured +3 %: more SALU on the walk's critical path)
#endif
#if GSR_FWD_SCALAR
#define GSR_FWD_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#else
#define GSR_FWD_UNIFORM(x) (x)
#endif
  const int lane = threadIdx.x & 63, wq = GSR_FWD_UNIFORM(threadIdx.x >> 6);
  float4* sA = sAll[wq][0];
  float4* sB = sAll[wq][1];
  float4* sC = sAll[wq][2];
  // (image order: taking the tiles in the previous frame's longest-walk-first order, as the backward does with its
  // own, measured no gain here -- 32 640 quad-sized jobs over 8 192 wave slots leave no tail worth ordering)
  const int quad = blockIdx.x * FW + wq;
  if (quad >= fp.gx * fp.gy * 4) return;  // (never in PHASE 1: the grid covers exactly four quads per tile, FW = 4)
  const int tile = quad >> 2, q = quad & 3;
  const int tile_x = tile % fp.gx, tile_y = tile / fp.gx;
  const int qx = tile_x * TILE + (q & 1) * 8, qy = tile_y * TILE + (q >> 1) * 8;
  const int px = qx + (lane & 7), py = qy + (lane >> 3);
  const bool inside = px < fp.W && py < fp.H;
  const float pfx = (float)px, pfy = (float)py;
  const float qx0 = (float)qx, qy0 = (float)qy;
  if (PHASE == 2 && quad_done[quad]) return;  // finished by the near phase
  const uint2 range_v = PHASE == 2 ? rangesB[tile] : ranges[tile];
  const uint2 range = make_uint2(GSR_FWD_UNIFORM(range_v.x), GSR_FWD_UNIFORM(range_v.y));
  const int n = (int)(range.y - range.x);
  uint32_t pos0 = 0;  // list position of this segment's first entry
  if (PHASE == 2) {
    const uint2 rn = ranges[tile];
    pos0 = GSR_FWD_UNIFORM(rn.y - rn.x);
  }
  const size_t pid = (size_t)fp.W * py + px;
  const size_t N = (size_t)fp.W * fp.H;

  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, A = 0.f;
  uint32_t last = 0;
  bool done = !inside;
  if (PHASE == 2 && inside) {  // the state phase 1 parked
    T = final_T[pid];
    const uint32_t l = n_contrib[pid];
    last = l & ~STOPPED_BIT;
    done = (l & STOPPED_BIT) != 0u;
    C0 = out_color[pid]; C1 = out_color[N + pid]; C2 = out_color[2 * N + pid];
    Dp = out_depth[pid];
    A = out_acc[pid];
  }
  // (from here on the pixels' "done" flags live in one scalar lane mask)
  uint64_t done_m = __builtin_amdgcn_ballot_w64(done);
  bool wave_done = done_m == ~0ull;

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
    // The hits of the sub-chunk are published COMPACTED: hit number r (in list order: the rank of the lane among the
    // hit lanes) goes to slot r of the wave's LDS image, together with the list position it stands for, so the walk below
    // is a counted loop over slots 0 .. hits-1 with nothing but an address increment in the scalar unit.  Walking the
    // set bits of the hit mask instead (s_ff1, a 64-bit mask clear, an address multiply per visit) cost 0.57 scalar
    // instructions per vector instruction -- 50 M per frame at 2 M Gaussians / 1080p, two thirds of the one scalar
    // unit a CU's 32 waves share.
    // Published conic terms carry the constant factors of power = -1/2 (A dx^2 + C dy^2) - B dx dy and the
    // log2(e) of exp(x) = exp2(x log2 e): one multiply per ENTRY here instead of three per (entry, pixel) visit
    constexpr float L2E = 1.4426950408889634f;
    const uint64_t m = __builtin_amdgcn_ballot_w64(hit);
    const int nh = (int)__popcll(m);
    if (hit) {
      const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      sA[r] = make_float4(a.x, a.y, a.z * (-0.5f * L2E), a.w * (-L2E));
      sB[r] = make_float4(b.x * (-0.5f * L2E), b.y, b.z, b.w);
      sC[r] = make_float4(c.x, c.y, __uint_as_float(pos0 + (uint32_t)(base + lane + 1)), 0.f);  // (.z: the list position)
    }
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
#if GSR_FWD_ASM_VISIT
    // One visit = one block of assembly: 23 vector and 4 scalar instructions.  The skip / stop rules of forward.cu:367-383
    // narrow the execution mask step by step -- not-yet-stopped pixels (s_andn1_saveexec), power <= 0 and alpha >= 1/255
    // (two v_cmpx), then T (1 - alpha) >= 1e-4 (the pixels that fail it are added to the `done` mask and leave) -- and the
    // pixels still active take the splat.  As C++ the same rules cost 13 scalar instructions per visit (lane masks
    // combined with s_and / s_or / s_andn2, loop flags through s_cselect and vcc branches): 43 M per frame at 2 M Gaussians /
    // 1080p on the ONE scalar unit the CU's waves share, which was as busy as the vector units (profiles/r03e_sq_probe.txt).
    // Same arithmetic, same order of roundings as the C++ visit (`blend`, below): images and n_contrib are bit-identical.
    auto visit = [&](const float4 ra, const float4 rb, const float4 rc) {
      float t0, t1, dx, dy;
      uint64_t sv;
      asm volatile(
          "v_sub_f32_e32 %[dx], %[ax], %[pfx]\n\t"
          "v_sub_f32_e32 %[dy], %[ay], %[pfy]\n\t"
          "v_mul_f32_e32 %[t0], %[dx], %[dx]\n\t"
          "v_mul_f32_e32 %[t1], %[Cc], %[dy]\n\t"
          "v_mul_f32_e32 %[t0], %[Aa], %[t0]\n\t"
          "v_fmac_f32_e32 %[t1], %[Bb], %[dx]\n\t"
          "v_fmac_f32_e32 %[t0], %[t1], %[dy]\n\t"      // power (x log2 e)
          "v_exp_f32_e32 %[t1], %[t0]\n\t"
          // (gfx950 does not interlock a transcendental's result against the very next vector instruction -- the
          // compiler keeps one instruction between them, and so must hand-written code: the first narrowing goes here)
          "s_andn1_saveexec_b64 %[sv], %[done]\n\t"     // pixels that have not stopped
          "v_cmpx_nlt_f32_e32 vcc, 0, %[t0]\n\t"        // ... with !(power > 0)
          "v_mul_f32_e32 %[t1], %[op], %[t1]\n\t"
          "v_min_f32_e32 %[t1], 0x3f7d70a4, %[t1]\n\t"  // alpha = min(0.99, opacity G)
          "v_sub_f32_e32 %[dx], 1.0, %[t1]\n\t"
          "v_mul_f32_e32 %[dx], %[T], %[dx]\n\t"        // test_T = T (1 - alpha)
          "v_cmpx_ngt_f32_e32 vcc, 0x3b808081, %[t1]\n\t"  // ... and !(alpha < 1/255)
          "v_cmp_gt_f32_e32 vcc, 0x38d1b717, %[dx]\n\t"    // of those: test_T < 1e-4 -> the pixel stops BEFORE this splat
          "s_or_b64 %[done], %[done], vcc\n\t"
          "s_andn2_b64 exec, exec, vcc\n\t"
          "v_mul_f32_e32 %[t0], %[t1], %[T]\n\t"        // weight = alpha T
          "v_fmac_f32_e32 %[C0], %[cr], %[t0]\n\t"
          "v_fmac_f32_e32 %[C1], %[cg], %[t0]\n\t"
          "v_fmac_f32_e32 %[C2], %[cb], %[t0]\n\t"
          "v_fmac_f32_e32 %[Dp], %[dz], %[t0]\n\t"
          "v_add_f32_e32 %[A], %[A], %[t0]\n\t"
          "v_mov_b32_e32 %[T], %[dx]\n\t"
          "v_mov_b32_e32 %[last], %[pos]\n\t"
          "s_mov_b64 exec, %[sv]"
          : [t0] "=&v"(t0), [t1] "=&v"(t1), [dx] "=&v"(dx), [dy] "=&v"(dy), [sv] "=&s"(sv), [done] "+s"(done_m),
            [C0] "+v"(C0), [C1] "+v"(C1), [C2] "+v"(C2), [Dp] "+v"(Dp), [A] "+v"(A), [T] "+v"(T), [last] "+v"(last)
          : [ax] "v"(ra.x), [ay] "v"(ra.y), [Aa] "v"(ra.z), [Bb] "v"(ra.w), [Cc] "v"(rb.x), [op] "v"(rb.y), [cr] "v"(rb.z),
            [cg] "v"(rb.w), [cb] "v"(rc.x), [dz] "v"(rc.y), [pos] "v"(rc.z), [pfx] "v"(pfx), [pfy] "v"(pfy)
          : "vcc", "scc");
    };
    if (nh) {
      // (an odd count is padded with a null entry -- opacity 0, so alpha = 0 and no pixel takes it: the walk below goes
      // in pairs without a test in the middle, and reads one slot ahead, hence FWD_SLOTS)
      if (lane == 0 && (nh & 1)) {
        sA[nh] = make_float4(0.f, 0.f, 0.f, 0.f);
        sB[nh] = make_float4(0.f, 0.f, 0.f, 0.f);
        sC[nh] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const int nh2 = (nh + 1) & ~1;
      float4 a0 = sA[0], b0 = sB[0], c0 = sC[0];
      // Each entry is read from LDS one visit before it is used, and waited for right after the visit that covered its
      // latency (s_waitcnt lgkmcnt(0) only: the next chunk's global gathers stay in flight).  Left to itself the
      // compiler waits at the first use, behind the younger reads of the other register set: one exposed LDS round
      // trip per pair of visits.
      __builtin_amdgcn_s_waitcnt(0xC07F);
      for (int r = 0; r < nh2; r += 2) {
        const float4 a1 = sA[r + 1], b1 = sB[r + 1], c1 = sC[r + 1];
        visit(a0, b0, c0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        a0 = sA[r + 2]; b0 = sB[r + 2]; c0 = sC[r + 2];  // (slot nh2 <= 64: read, never used)
        visit(a1, b1, c1);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (done_m == ~0ull) {  // checked once per pair of visits: a visit after saturation changes nothing
          wave_done = true;
          break;
        }
      }
    }
#else
    // Visit loop, software-pipelined by hand: the LDS broadcast reads of the NEXT hit are issued before the
    // current hit is blended (two register sets, no copies), so their latency hides behind ~30 VALU ops.
    auto blend = [&](const float4 ra, const float4 rb, const float4 rc) {
      const float dx = ra.x - pfx, dy = ra.y - pfy;
      const float power = splat_power(ra.z, ra.w, rb.x, dx, dy);  // = log2(e) x the reference's power
      const float alpha = fminf(0.99f, rb.y * __builtin_amdgcn_exp2f(power));
      // forward.cu:367-383: skip on power > 0 and alpha < 1/255; a pixel whose T would fall below 1e-4 stops BEFORE taking
      // the splat.  Three compares straight into lane masks, the rest is scalar mask arithmetic.
      const uint64_t take = __builtin_amdgcn_ballot_w64(!(power > 0.0f)) &
                            __builtin_amdgcn_ballot_w64(!(alpha < 1.0f / 255.0f)) & ~done_m;
      const float test_T = T * (1.0f - alpha);
      const uint64_t low = __builtin_amdgcn_ballot_w64(test_T < 0.0001f);
      done_m |= take & low;
      const uint64_t ok = take & ~low;
#if GSR_FWD_EXEC_MASK
      // The pixels that take the splat update their sums, T and last contributor under the execution mask `ok`: eight
      // plain instructions.  (With selects -- weight or 0, new or old T, new or old position -- it is nine, three of
      // them v_cndmask_b32 at 1.6 times the issue cost of a multiply-add: tools/microbench/valu_probe.hip.)  The compiler cannot
      // be told to use a scalar lane mask as a branch condition without deriving a per-lane flag from it first, hence
      // the assembly; the wave's execution mask is restored before anything else runs.
      uint64_t exec_save;
      float wgt;
      asm volatile(
          "s_and_saveexec_b64 %[sv], %[ok]\n\t"
          "v_mul_f32_e32 %[w], %[al], %[T]\n\t"
          "v_fmac_f32_e32 %[C0], %[cr], %[w]\n\t"
          "v_fmac_f32_e32 %[C1], %[cg], %[w]\n\t"
          "v_fmac_f32_e32 %[C2], %[cb], %[w]\n\t"
          "v_fmac_f32_e32 %[Dp], %[dz], %[w]\n\t"
          "v_add_f32_e32 %[A], %[A], %[w]\n\t"
          "v_mov_b32_e32 %[T], %[tT]\n\t"
          "v_mov_b32_e32 %[last], %[pos]\n\t"
          "s_mov_b64 exec, %[sv]"
          : [sv] "=&s"(exec_save), [w] "=&v"(wgt), [C0] "+v"(C0), [C1] "+v"(C1), [C2] "+v"(C2), [Dp] "+v"(Dp), [A] "+v"(A),
            [T] "+v"(T), [last] "+v"(last)
          : [ok] "s"(ok), [al] "v"(alpha), [cr] "v"(rb.z), [cg] "v"(rb.w), [cb] "v"(rc.x), [dz] "v"(rc.y), [tT] "v"(test_T),
            [pos] "v"(rc.z)
          : "scc");
#else
      const float wgt = sel_f0(ok, alpha * T);
      C0 += rb.z * wgt;
      C1 += rb.w * wgt;
      C2 += rc.x * wgt;
      Dp += rc.y * wgt;
      A += wgt;
      T = sel_f(ok, test_T, T);
      last = sel_u(ok, __float_as_uint(rc.z), last);
#endif
    };
    if (nh) {
      float4 a0 = sA[0], b0 = sB[0], c0 = sC[0], a1, b1, c1;
      for (int r = 0;; r += 2) {
        const bool has1 = r + 1 < nh;
        if (has1) { a1 = sA[r + 1]; b1 = sB[r + 1]; c1 = sC[r + 1]; }
        blend(a0, b0, c0);
        if (!has1) {
          wave_done = done_m == ~0ull;
          break;
        }
        const bool has2 = r + 2 < nh;
        if (has2) { a0 = sA[r + 2]; b0 = sB[r + 2]; c0 = sC[r + 2]; }
        blend(a1, b1, c1);
        wave_done = done_m == ~0ull;  // checked once per pair of visits: a visit after saturation changes nothing
        if (wave_done || !has2) break;
      }
    }
#endif
  }

  done = ((done_m >> lane) & 1ull) != 0ull;  // (back to a per-lane flag for the epilogue)
  const uint32_t wl = wave_max_u32(inside ? last : 0u);
  if (lane == 0) quad_last[quad] = wl;
  if (PHASE == 1) {
    const bool all_stopped = done_m == ~0ull;
    if (lane == 0) {
      quad_done[quad] = all_stopped ? 1 : 0;
      const uint32_t old = atomicAdd(&s_done, 1u | (all_stopped ? 0u : 0x100u));
      if ((old & 0xFFu) == (uint32_t)FW - 1u) {  // last wave of this workgroup
        const uint32_t mine = (old >> 8) + (all_stopped ? 0u : 1u);
        const unsigned long long o = atomicAdd(done_word, (1ull << 40) | (unsigned long long)mine);
        if ((o >> 40) == (unsigned long long)gridDim.x - 1ull) {  // last workgroup of the launch
          const uint32_t live = (uint32_t)((o & ((1ull << 40) - 1ull)) + mine);
          *live_quads_out = live;
          __hip_atomic_store(done_word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (publish)
            __hip_atomic_store(publish, ((unsigned long long)ticket << 32) | live, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
          // Asynchronous frame.  Nothing left to do (live == 0): the gates stay closed, so the far chain's kernels touch
          // nothing, and the caller's stream -- ordered behind this kernel anyway -- is released from here.  Quads left
          // unfinished: NOT decided here.  Other waves of this launch may still be parking their state, and the XCDs'
          // L2s are not coherent: a far chain released by a word stored from inside this kernel could read quad_done
          // or parked pixels that have not reached memory.  k_decide_far, enqueued behind this kernel on the same
          // stream, opens the far chain after the kernel boundary has written everything back.
          if (aw.decide && live == 0u) {
            const uint32_t d = 2u * aw.seq;
            __hip_atomic_store(aw.gate_dev, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(aw.go, aw.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(aw.decide, d, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // (after the gate word)
          }
        }
      }
    }
    if (!all_stopped) {  // park the running state for phase 2
      if (inside) {
        final_T[pid] = T;
        n_contrib[pid] = last | (done ? STOPPED_BIT : 0u);
        out_color[pid] = C0; out_color[N + pid] = C1; out_color[2 * N + pid] = C2;
        out_depth[pid] = Dp;
        out_acc[pid] = A;
      }
      return;
    }
  }
  if (inside) {
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
    uint8_t* __restrict__ inst_flag, uint8_t* __restrict__ touched, const uint32_t* __restrict__ tile_order,
    const uint2* __restrict__ rangesB) {
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
  const int tile = tile_order ? (int)tile_order[blockIdx.x] : (int)blockIdx.x;  // longest walk first (k_tile_order)
  const int tile_x = tile % fp.gx, tile_y = tile / fp.gx;
  const uint32_t ql0 = quad_last_in[4 * tile], ql1 = quad_last_in[4 * tile + 1], ql2 = quad_last_in[4 * tile + 2],
                 ql3 = quad_last_in[4 * tile + 3];
  const int n = (int)max(max(ql0, ql1), max(ql2, ql3));  // entries [0, n) of the tile's list can carry gradient
  if (n == 0) return;
  // the tile's list = its near segment followed by its far segment (near/far frames; (0, 0) otherwise)
  const uint2 rng = ranges[tile];
  const uint32_t rbase = rng.x, rbaseB = rangesB[tile].x;
  const int lenA = (int)(rng.y - rng.x);
  const int px = tile_x * TILE + (w & 1) * 8 + (lane & 7);
  const int py = tile_y * TILE + (w >> 1) * 8 + (lane >> 3);
  const bool inside = px < fp.W && py < fp.H;
  const float pfx = (float)px, pfy = (float)py;
  const float tx0 = (float)(tile_x * TILE), ty0 = (float)(tile_y * TILE);
  const size_t pid = (size_t)fp.W * py + px, N = (size_t)fp.W * fp.H;

  const float T_final = inside ? final_T[pid] : 0.0f;
  const int lastc = inside ? (int)n_contrib[pid] : 0;
  const float dp0 = inside ? dL_dpix[pid] : 0.f, dp1 = inside ? dL_dpix[N + pid] : 0.f,
              dp2 = inside ? dL_dpix[2 * N + pid] : 0.f;
  const float dacc = inside ? dL_dacc[pid] : 0.f;  // the reference reads this unguarded (backward.cu:497)
  float T = T_final;
  // ONE recurrence instead of the reference's four (accum_rec[3], accum_acc_rec; backward.cu:533-571).  dL/dalpha of a
  // splat is T (q - S) with q = colour . dL/dpixel + 1 * dL/dacc of THIS splat and S = the same dot product of what lies
  // behind it (accum_rec . dL/dpixel + accum_acc_rec * dL/dacc); folding the splat in, accum <- accum + alpha (c - accum)
  // for every channel, is S <- S + alpha (q - S): the difference dL/dalpha has just used.  The background is the last
  // thing behind every splat, so S starts at bg . dL/dpixel and the reference's separate term
  // (-T_final / (1 - alpha)) bg . dL/dpixel (backward.cu:578-581) is carried by the recurrence: T_i (1 - alpha_{i+1}) ..
  // (1 - alpha_last) = T_final / (1 - alpha_i).  Five instructions per (pixel, splat) instead of twelve, one state
  // register instead of five.
  float S = bg[0] * dp0 + bg[1] * dp1 + bg[2] * dp2;

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
      const uint32_t id = pos < lenA ? point_list[rbase + pos] : point_list[rbaseB + (uint32_t)(pos - lenA)];
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
      sSlot[tid] = si.x + (uint32_t)((tile_y - y0) * rw + (tile_x - x0));
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
        const float power = splat_power(a.z, a.w, b.x, dx, dy);  // = log2(e) x the reference's power
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
        const float D = __builtin_fmaf(b.z, dp0, __builtin_fmaf(b.w, dp1, __builtin_fmaf(blue, dp2, dacc))) - S;
        if (__ballot(ok) != 0ull) {
          const float G = ok ? Graw : 0.0f;
          const float dch = alpha * Tn;
          const float dL_dalpha = Tn * D;
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
        S = __builtin_fmaf(alpha, D, S);
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
        // raw pixel sums; the splat-constant factors (conic, opacity, 1/2 W, 1/2 H) are applied once per GAUSSIAN by
        // k_gather_records, after its instances have been summed
        grad_inst[slot * GRAD_F4 + 0] = make_float4(s[0], s[1], s[2], s[3]);
        grad_inst[slot * GRAD_F4 + 1] = make_float4(s[4], s[5], s[6], s[7]);
        grad_inst[slot * GRAD_F4 + 2] = make_float4(s[8], 0.f, 0.f, 0.f);
        inst_flag[slot] = 1;
        touched[sId[tid]] = 1;  // same value from every writer: a benign race
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// B1 backward blend, ONE WAVE PER 16x16 TILE, four pixels per lane, no barriers.
//
// Why: at the 2 M-Gaussian / 1080p shape 84 % of the list entries the backward walks cover all four 8x8 quads of
// their tile and 93 % of the lanes of a (quad, splat) visit contribute (tools/analyze_blend_visits.py), and half of a
// one-pixel-per-lane visit is the nine-value cross-lane reduction.  Here lane l owns the pixels (x = l & 15,
// y = 4 k + (l >> 4)), k = 0..3, of its tile: the nine partial sums are accumulated over the lane's own four pixels in
// registers first (as fused multiply-adds: no extra instructions) and ONE reduction serves 256 pixels -- 2.9 instead
// of 7.5 reduction instructions per (quad, splat) -- and the four pixels of a lane share a column, hence dx, dx^2 and
// A' dx^2 of the power.  The wave owns the whole tile, so the per-instance record is final after the reduction:
// no per-quad partials in LDS, no combine step, no workgroup barrier; four lanes store the nine RAW pixel sums
// (the splat-constant factors are applied once per Gaussian by k_gather_records).
// The list is streamed in sub-chunks of 64 like the forward: each lane gathers one entry (record, slot), tests the
// splat's exact alpha >= 1/255 ellipse against the TILE, the wave ballots, publishes the entries to a wave-private
// LDS image and walks the set bits back to front; the next sub-chunk's gather is in flight meanwhile.
// Replaces renderCUDA backward (reference backward.cu:438-603); sums in a fixed order => bitwise reproducible.
// ------------------------------------------------------------------------------------------------
// Order in which the backward takes the tiles: longest list walk first.  A tile is one wave's serial job and the chip
// holds only about half of a 1080p frame's tiles at once, so in image order the launch ends with a tail of whatever
// long tiles happened to start late; started in descending order of work, the last tiles to start are the shortest
// ones (longest-processing-time-first list scheduling).  Counting sort by walk length, one workgroup; the order among
// tiles of equal length is arbitrary and nothing depends on it (every tile is an independent unit of work).
__global__ __launch_bounds__(1024) void k_tile_order(const uint32_t* __restrict__ quad_last, const int T,
                                                     uint32_t* __restrict__ order) {
  __shared__ uint32_t bins[1024];  // bin b = walk length min(w, 1023); scanned from the longest down
  __shared__ uint32_t wtot[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  bins[tid] = 0u;
  __syncthreads();
  auto walk = [&](int t) {
    const uint4 q = *reinterpret_cast<const uint4*>(quad_last + 4 * (size_t)t);
    const uint32_t m = max(max(q.x, q.y), max(q.z, q.w));
    return m < 1023u ? m : 1023u;
  };
  // (a tile's walk is needed twice, its loads are requested once: up to 8 tiles per thread = a 1080p frame in registers)
  uint32_t wk[8];
#pragma unroll
  for (int k = 0; k < 8; k++) wk[k] = tid + 1024 * k < T ? walk(tid + 1024 * k) : 0u;
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (tid + 1024 * k < T) atomicAdd(&bins[wk[k]], 1u);
  for (int t = tid + 8192; t < T; t += 1024) atomicAdd(&bins[walk(t)], 1u);
  __syncthreads();
  const uint32_t v = bins[1023 - tid];  // thread i owns bin 1023 - i: an exclusive scan over i = over longer walks
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = __shfl_up(inc, o, 64);
    if (lane >= o) inc += u;
  }
  if (lane == 63) wtot[w] = inc;
  __syncthreads();
  uint32_t base = inc - v;
  for (int k = 0; k < w; k++) base += wtot[k];
  bins[1023 - tid] = base;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (tid + 1024 * k < T) order[atomicAdd(&bins[wk[k]], 1u)] = (uint32_t)(tid + 1024 * k);
  for (int t = tid + 8192; t < T; t += 1024) order[atomicAdd(&bins[walk(t)], 1u)] = (uint32_t)t;
}

// experiment knobs (A/B builds: GSR_EXTRA_RENDER_FLAGS=-DGSR_BWD_...=..)
#ifndef GSR_BWD_STRIP_BRANCH
#define GSR_BWD_STRIP_BRANCH 1
#endif
#ifndef GSR_BWD_MIN_WAVES
#define GSR_BWD_MIN_WAVES 1
#endif
#ifndef GSR_BWD_EXEC_MASK
#define GSR_BWD_EXEC_MASK 1
#endif
template <int TW>  // tiles (= waves) per workgroup; the waves never synchronise
__global__ __launch_bounds__(64 * TW, GSR_BWD_MIN_WAVES) void k_blend_backward_tile(
    const FrameParams fp, const uint2* __restrict__ ranges, const uint32_t* __restrict__ quad_last_in,
    const uint32_t* __restrict__ point_list, const float4* __restrict__ splats, const uint2* __restrict__ slotinfo,
    const float* __restrict__ bg, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dpix, const float* __restrict__ dL_dacc, float* __restrict__ grad_inst,
    uint8_t* __restrict__ inst_flag, uint8_t* __restrict__ touched, const uint32_t* __restrict__ tile_order,
    const uint2* __restrict__ rangesB) {
  __shared__ float4 sE[TW][64][3];  // (x, y, A', B' | C', opacity, r, g | b, -, -, -) per staged entry
  __shared__ uint32_t sSlot[TW][64], sId[TW][64];
  // (wave-uniform values are made scalars by hand -- the compiler cannot know that a loaded value is the same in every
  // lane and would keep list positions and LDS bases in vector registers, with vector instructions to update them)
  const int lane = threadIdx.x & 63, wq = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (blockIdx.x * TW + wq >= fp.gx * fp.gy) return;
  const int tile = __builtin_amdgcn_readfirstlane(tile_order ? (int)tile_order[blockIdx.x * TW + wq]
                                                             : (int)(blockIdx.x * TW + wq));
  const uint32_t ql0 = quad_last_in[4 * tile], ql1 = quad_last_in[4 * tile + 1], ql2 = quad_last_in[4 * tile + 2],
                 ql3 = quad_last_in[4 * tile + 3];
  // entries [0, n) of the tile's list can carry gradient
  const int n = __builtin_amdgcn_readfirstlane((int)max(max(ql0, ql1), max(ql2, ql3)));
  if (n == 0) return;
  const int tile_x = tile % fp.gx, tile_y = tile / fp.gx;
  // the tile's list = its near segment followed by its far segment (near/far frames; (0, 0) otherwise)
  const uint2 rng = ranges[tile];
  const uint32_t rbase = rng.x, rbaseB = rangesB[tile].x;
  const int lenA = (int)(rng.y - rng.x);
  const int px = tile_x * TILE + (lane & 15);
  const float pfx = (float)px;
  const float tx0 = (float)(tile_x * TILE), ty0 = (float)(tile_y * TILE);
  const size_t N = (size_t)fp.W * fp.H;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];

  // per-pixel state, k = 0..3 (fully unrolled: registers)
  float pfy[4], T[4], S[4], dp0[4], dp1[4], dp2[4], dacc[4];
  int lastc[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int py = tile_y * TILE + 4 * k + (lane >> 4);
    const bool inside = px < fp.W && py < fp.H;
    const size_t pid = (size_t)fp.W * py + px;
    pfy[k] = (float)py;
    T[k] = inside ? final_T[pid] : 0.0f;
    lastc[k] = inside ? (int)n_contrib[pid] : 0;  // a pixel outside the image never takes a splat
    dp0[k] = inside ? dL_dpix[pid] : 0.f;
    dp1[k] = inside ? dL_dpix[N + pid] : 0.f;
    dp2[k] = inside ? dL_dpix[2 * N + pid] : 0.f;
    dacc[k] = inside ? dL_dacc[pid] : 0.f;  // the reference reads this unguarded (backward.cu:497)
    // the one recurrence of k_blend_backward (see there): what lies behind the last contributor is the background
    S[k] = bg0 * dp0[k] + bg1 * dp1[k] + bg2 * dp2[k];
  }

  // staging of list entry `k` counted from the back of [0, n): record, gradient slot, tile-level hit test
  float4 a = make_float4(0, 0, 0, 0), b = a, c = a;
  uint32_t eid = 0, eslot = 0;
  bool hit = false;
  auto gather = [&](const int k) {
    hit = false;
    if (k < n) {
      const int pos = n - 1 - k;
      const uint32_t id = pos < lenA ? point_list[rbase + (uint32_t)pos] : point_list[rbaseB + (uint32_t)(pos - lenA)];
      a = splats[(size_t)id * SPLAT_F4 + 0];
      b = splats[(size_t)id * SPLAT_F4 + 1];
      c = splats[(size_t)id * SPLAT_F4 + 2];
      const uint2 si = slotinfo[id];
      const int x0 = (int)(si.y & 1023u), y0 = (int)((si.y >> 10) & 1023u), rw = (int)(si.y >> 20);
      eid = id;
      eslot = si.x + (uint32_t)((tile_y - y0) * rw + (tile_x - x0));
      // footprint box against the tile, then the ellipse itself (hx < 0: never; hx >= 1e6: indefinite conic, no culling)
      hit = (a.x + c.z >= tx0) && (a.x - c.z <= tx0 + 15.0f) && (a.y + c.w >= ty0) && (a.y - c.w <= ty0 + 15.0f);
      if (hit && c.z < 1.0e6f)
        hit = ellipse_reaches_quad<15>(a.x, a.y, a.z, a.w, b.x, footprint_tau(b.y), tx0, ty0);
    }
  };
  gather(lane);
  float* const rec_lane = grad_inst + 2 * (lane >> 4);  // lanes 0, 16, 32, 48 store floats (2r, 2r + 1) of a record
  for (int base = 0; base < n; base += 64) {
    constexpr float L2E = 1.4426950408889634f;
    sE[wq][lane][0] = make_float4(a.x, a.y, a.z * (-0.5f * L2E), a.w * (-L2E));
    sE[wq][lane][1] = make_float4(b.x * (-0.5f * L2E), b.y, b.z, b.w);
    sE[wq][lane][2] = make_float4(c.x, 0.f, 0.f, 0.f);
    sSlot[wq][lane] = eslot;
    sId[wq][lane] = eid;
    uint64_t m = __ballot(hit);
    gather(base + 64 + lane);  // the next sub-chunk's gather completes while this one is walked
    while (m) {
      const int jj = __builtin_ctzll(m);
      m &= m - 1;
      const int pos = n - 1 - (base + jj);  // 0-based index in the tile list == `contributor` after decrement
      const float4 ea = sE[wq][jj][0];
      const float4 eb = sE[wq][jj][1];
      const float blue = sE[wq][jj][2].x;
      const float dx = ea.x - pfx;
      const float dx2 = dx * dx;
      const float Adx2 = ea.z * dx2;
      // lane-local sums over the lane's four pixels: colour (3), dLG, dLG dy, dLG dy^2 (accumulated as fused
      // multiply-adds: combining the four pixels costs no instruction of its own)
      float c0 = 0.f, c1 = 0.f, c2 = 0.f, sG = 0.f, sGy = 0.f, sGyy = 0.f;
#if GSR_BWD_EXEC_MASK
      uint64_t anym = 0ull;  // lanes one of whose four pixels takes the splat
#else
      uint32_t abits = 0u;  // OR of the lane's four alphas: non-zero iff one of its pixels takes the splat
#endif
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const float dy = ea.y - pfy[k];
        const float power = splat_power_shared(Adx2, ea.w, eb.x, dx, dy);  // = log2(e) x the reference's power
        const float Graw = __builtin_amdgcn_exp2f(power);
        const float araw = fminf(0.99f, eb.y * Graw);
        const bool ok = (pos < lastc[k]) && !(power > 0.0f) && !(araw < 1.0f / 255.0f);
#if GSR_BWD_EXEC_MASK
        // The pixels that take this splat run the update under the execution mask; the others keep their recurrence
        // state and add nothing -- no select instructions (a v_cndmask / v_cmp / v_min costs 1.6 plain multiply-adds on
        // this chip, tools/microbench/valu_probe.hip), and a strip no pixel of which takes the splat costs the test only.
        // (a ballot of the conjunction would be materialised as a select and a compare: and the three masks instead)
        anym |= __builtin_amdgcn_ballot_w64(pos < lastc[k]) & __builtin_amdgcn_ballot_w64(!(power > 0.0f)) &
                __builtin_amdgcn_ballot_w64(!(araw < 1.0f / 255.0f));
        {
          if (ok) {
            const float rom = __builtin_amdgcn_rcpf(1.0f - araw);
            const float Tn = T[k] * rom;  // T / (1 - alpha)
            const float D =
                __builtin_fmaf(eb.z, dp0[k], __builtin_fmaf(eb.w, dp1[k], __builtin_fmaf(blue, dp2[k], dacc[k]))) - S[k];
            const float dch = araw * Tn;
            const float dL_dalpha = Tn * D;
            const float dLG = Graw * dL_dalpha;  // dL/dG up to the opacity factor; also the opacity partial itself
            const float sy = dLG * dy;
            c0 = __builtin_fmaf(dch, dp0[k], c0);
            c1 = __builtin_fmaf(dch, dp1[k], c1);
            c2 = __builtin_fmaf(dch, dp2[k], c2);
            sG += dLG;
            sGy += sy;
            sGyy = __builtin_fmaf(sy, dy, sGyy);
            // Fold this splat into the "everything behind the next one" accumulators (the reference does it at the top
            // of its next iteration from saved (last_alpha, last_color), backward.cu:533-543): accum + alpha (c - accum)
            T[k] = Tn;
            S[k] = __builtin_fmaf(araw, D, S[k]);
          }
        }
      }
#else
        // Branch-free per lane: a pixel that does not take this splat runs with alpha = 0 and G = 0 -- exact zero
        // partials, recurrence state untouched bit for bit (see k_blend_backward)
        const float alpha = ok ? araw : 0.0f;
        const float oma = 1.0f - alpha;
        const float rom = __builtin_amdgcn_rcpf(oma);
        const float Tn = T[k] * rom;  // T / (1 - alpha)
        const float D =
            __builtin_fmaf(eb.z, dp0[k], __builtin_fmaf(eb.w, dp1[k], __builtin_fmaf(blue, dp2[k], dacc[k]))) - S[k];
        if (!GSR_BWD_STRIP_BRANCH || __ballot(ok) != 0ull) {  // some pixel of this 16 x 4 strip takes the splat
          const float G = ok ? Graw : 0.0f;
          const float dch = alpha * Tn;
          const float dL_dalpha = Tn * D;
          const float dLG = G * dL_dalpha;  // dL/dG up to the opacity factor; also the opacity partial itself
          const float sy = dLG * dy;
          c0 = __builtin_fmaf(dch, dp0[k], c0);
          c1 = __builtin_fmaf(dch, dp1[k], c1);
          c2 = __builtin_fmaf(dch, dp2[k], c2);
          sG += dLG;
          sGy += sy;
          sGyy = __builtin_fmaf(sy, dy, sGyy);
        }
        abits |= __float_as_uint(alpha);
        // Fold this splat into the "everything behind the next one" accumulators (the reference does it at the top of
        // its next iteration from saved (last_alpha, last_color), backward.cu:533-543): accum + alpha (c - accum)
        T[k] = Tn;
        S[k] = __builtin_fmaf(alpha, D, S[k]);
      }
      const uint64_t anym = __ballot(abits != 0u);
#endif
      if (anym != 0ull) {
        // the lane's pixels share dx: sum dLG dx = dx sum dLG, sum dLG dx^2 = dx^2 sum dLG, sum dLG dx dy = dx sum dLG dy
        float v0 = c0, v1 = c1, v2 = c2, v3 = dx * sG, v4 = sGy, v5 = dx2 * sG, v6 = dx * sGy, v7 = sGyy, g8 = sG;
        swap_add32(v0, v4);
        swap_add32(v1, v5);
        swap_add32(v2, v6);
        swap_add32(v3, v7);
        swap_add16(v0, v2);  // rows hold the half-wave-pair sums of v0, v2, v4, v6
        swap_add16(v1, v3);  // ... of v1, v3, v5, v7
        row_allsum3(v0, v1, g8);
        asm volatile("" : "+v"(v0), "+v"(v1), "+v"(g8));
        const size_t slot = sSlot[wq][jj];
        // record = the RAW pixel sums: floats 0..7 = colour r g b | dLG dx, dLG dy | dLG dx^2, dLG dx dy, dLG dy^2 (row r
        // of the wave holds the wave totals of values 2r, 2r + 1), floats 8..11 = the four 16-lane-row sums of dLG
        // (k_gather_records adds them: two cross-row steps fewer here, once per instance)
        if ((lane & 15) == 0) {
          *reinterpret_cast<float2*>(rec_lane + slot * (GRAD_F4 * 4)) = make_float2(v0, v1);
          grad_inst[slot * (GRAD_F4 * 4) + 8 + (lane >> 4)] = g8;
        }
        if (lane == 63) {
          inst_flag[slot] = 1;
          touched[sId[wq][jj]] = 1;  // same value from every writer: a benign race
        }
      }
    }
  }
}

// Near/far frames: summed-area table of the tiles that still have an unfinished quad after the near phase, and their
// number (total_live).  sat[(y + 1)(gx + 1) + (x + 1)] = live tiles in [0, x] x [0, y]; row 0 and column 0 are zero.
// One workgroup.  IN_LDS (the table fits 150 KB: every frame up to ~3000 x 3000 px): flags in with coalesced loads,
// row sums by wave scans and column sums by a thread per column, all in LDS, table out coalesced (5 us at 1080p;
// the global-memory variant below walks rows and columns with dependent loads: 54 us).
template <bool IN_LDS>
__global__ __launch_bounds__(1024) void k_live_sat(const uint8_t* __restrict__ quad_done, const int gx, const int gy,
                                                   uint32_t* __restrict__ sat, uint32_t* __restrict__ total_live,
                                                   const Count gate) {
  extern __shared__ uint32_t s_sat[];
  if (gate.closed()) return;
  uint32_t* const S = IN_LDS ? s_sat : sat;
  const int sw = gx + 1, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (IN_LDS) {
    for (int i = tid; i < sw * (gy + 1); i += 1024) {
      const int y = i / sw - 1, x = i % sw - 1;
      uint32_t live = 0;
      if (x >= 0 && y >= 0) {
        const uchar4 q = *reinterpret_cast<const uchar4*>(quad_done + 4 * ((size_t)y * gx + x));
        live = (q.x & q.y & q.z & q.w) ? 0u : 1u;
      }
      S[i] = live;
    }
    __syncthreads();
    for (int y = w; y < gy; y += 16) {  // a wave per row: inclusive scan in chunks of 64
      uint32_t carry = 0;
      for (int x0 = 0; x0 < gx; x0 += 64) {
        const int x = x0 + lane;
        uint32_t v = x < gx ? S[(y + 1) * sw + x + 1] : 0u;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t u = __shfl_up(v, o, 64);
          if (lane >= o) v += u;
        }
        if (x < gx) S[(y + 1) * sw + x + 1] = carry + v;
        carry += __shfl(v, 63, 64);
      }
    }
    __syncthreads();
  } else {
    for (int i = tid; i < sw; i += 1024) S[i] = 0u;
    for (int y = tid; y < gy; y += 1024) {
      uint32_t run = 0;
      S[(y + 1) * sw] = 0u;
      for (int x = 0; x < gx; x++) {
        const uchar4 q = *reinterpret_cast<const uchar4*>(quad_done + 4 * ((size_t)y * gx + x));
        run += (q.x & q.y & q.z & q.w) ? 0u : 1u;
        S[(y + 1) * sw + x + 1] = run;
      }
    }
    __syncthreads();
  }
  for (int x = tid; x < gx; x += 1024) {
    uint32_t run = 0;
    for (int y = 0; y < gy; y++) {
      run += S[(y + 1) * sw + x + 1];
      S[(y + 1) * sw + x + 1] = run;
    }
  }
  __syncthreads();
  if (IN_LDS)
    for (int i = tid; i < sw * (gy + 1); i += 1024) sat[i] = S[i];
  if (tid == 0) *total_live = S[gy * sw + gx];
}

// Asynchronous near/far frame whose near blend left quads unfinished: opens the far chain that waits on the library's
// second stream.  A launch of its own behind the near blend on the caller's stream, so that the kernel boundary -- not a
// fence inside a kernel whose other waves are still storing -- orders the near chain's results (quad_done, the parked
// pixel state, quad_last) before everything the far chain reads.  One thread; a frame with nothing left open was
// decided by the near blend itself and this kernel does nothing.
__global__ void k_decide_far(const uint32_t* __restrict__ live_quads, const AsyncWords aw) {
  if (threadIdx.x != 0 || *live_quads == 0u) return;
  const uint32_t d = 2u * aw.seq + 1u;
  __hip_atomic_store(aw.gate_dev, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(aw.decide, d, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // (after the gate word)
}
hipError_t launch_decide_far(const uint32_t* live_quads, AsyncWords aw, hipStream_t s) {
  hipLaunchKernelGGL(k_decide_far, dim3(1), dim3(64), 0, s, live_quads, aw);
  return hipGetLastError();
}

__global__ void k_release_go(const Count gate, uint32_t* __restrict__ go, const uint32_t seq) {
  if (threadIdx.x == 0 && !gate.closed()) __hip_atomic_store(go, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// last launch of an asynchronous frame's far chain: lets the caller's stream go on (if the chain ran; otherwise the
// near blend has released it long ago)
hipError_t launch_release_go(Count gate, uint32_t* go, uint32_t seq, hipStream_t s) {
  hipLaunchKernelGGL(k_release_go, dim3(1), dim3(64), 0, s, gate, go, seq);
  return hipGetLastError();
}

hipError_t launch_live_sat(const FrameParams& fp, ImageState im, uint32_t* total_live, Count gate, hipStream_t s) {
  ProfScope ps(K_LIVE_SAT, s);
  const size_t bytes = (size_t)(fp.gx + 1) * (fp.gy + 1) * sizeof(uint32_t);
  if (bytes <= 150 * 1024) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_live_sat<true>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(k_live_sat<true>, dim3(1), dim3(1024), bytes, s, im.quad_done, fp.gx, fp.gy, im.live_sat,
                       total_live, gate);
  } else {
    hipLaunchKernelGGL(k_live_sat<false>, dim3(1), dim3(1024), 0, s, im.quad_done, fp.gx, fp.gy, im.live_sat, total_live,
                       gate);
  }
  return hipGetLastError();
}

hipError_t launch_blend_forward(const FrameParams& fp, GeomState g, BinningState b, ImageState im, const float* bg,
                                float* out_color, float* out_depth, float* out_acc, int phase,
                                unsigned long long* done_word, unsigned long long* publish, uint32_t ticket,
                                AsyncWords aw, Count gate, hipStream_t s) {
  ProfScope ps_k_blend_fwd(K_BLEND_FWD, s);
  // the four quads of a tile share a workgroup slot (they never synchronise): their redundant gathers of the same
  // records coincide in time and hit L1/L2; 1, 2 and 4 waves per workgroup measured within 3 % of each other
  const int quads = fp.gx * fp.gy * 4;
#define GSR_LAUNCH_FWD(PH)                                                                                            \
  hipLaunchKernelGGL((k_blend_forward<4, PH>), dim3((quads + 3) / 4), dim3(256), 0, s, fp, im.ranges, im.rangesB,      \
                     b.point_list, g.splats, bg, im.final_T, im.n_contrib, im.quad_last, im.quad_done, out_color,      \
                     out_depth, out_acc, done_word, publish, ticket, g.total + 13, aw, gate)
  if (phase == 1) GSR_LAUNCH_FWD(1);
  else if (phase == 2) GSR_LAUNCH_FWD(2);
  else GSR_LAUNCH_FWD(0);
#undef GSR_LAUNCH_FWD
  return hipGetLastError();
}

// Tile order of the backward (k_tile_order) as a launch of its own: a forward that has nothing else left to enqueue
// computes it at its end (api.hip), where it fills the gap until the host has enqueued the next kernels.
static bool backward_image_order() {
  static const bool image_order = getenv("GSR_BWD_IMAGE_ORDER") != nullptr;  // experiment knob
  return image_order;
}
hipError_t launch_tile_order(const FrameParams& fp, ImageState im, hipStream_t s) {
  if (backward_image_order()) return hipSuccess;
  ProfScope ps(K_TILE_ORDER, s);
  hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, s, im.quad_last, fp.gx * fp.gy, im.tile_order);
  return hipGetLastError();
}

hipError_t launch_blend_backward(const FrameParams& fp, GeomState g, BinningState b, ImageState im, const float* bg,
                                 const float* dL_dpix, const float* dL_dacc, bool have_tile_order, hipStream_t s) {
  if (!have_tile_order) {
    const hipError_t e = launch_tile_order(fp, im, s);
    if (e != hipSuccess) return e;
  }
  ProfScope ps_k_blend_bwd(K_BLEND_BWD, s);
  // Two kernels, chosen by the number of tiles.  One wave per tile (four pixels per lane, one reduction per 256
  // pixels) needs a frame with at least ~3000 tiles to fill the chip -- a tile is one wave's serial job -- and then wins
  // (1080p: 0.21 vs 0.28 ms; 1280x720: 0.141 vs 0.162); on the small frames of the product (640x512 = 1280 tiles) four
  // waves per tile keep four times as many waves in flight (0.121 vs 0.327 ms).  Both take the tiles longest walk
  // first.  GSR_BLEND_BACKWARD_QUADS=1 / GSR_BLEND_BACKWARD_TILES=1 force one of them (diagnostics, tests).
  static const bool force_quad = getenv("GSR_BLEND_BACKWARD_QUADS") != nullptr;
  static const bool force_tile = getenv("GSR_BLEND_BACKWARD_TILES") != nullptr;
  const int tiles = fp.gx * fp.gy;
  const bool per_quad = force_quad || (!force_tile && tiles < 3072);
  const uint32_t* order = backward_image_order() ? nullptr : im.tile_order;
  if (per_quad) {
    // chunks of 128 list entries (64 and 128 measured equal, 256 slower: LDS footprint)
    hipLaunchKernelGGL(k_blend_backward<128>, dim3(tiles), dim3(256), 0, s, fp, im.ranges, im.quad_last, b.point_list,
                       g.splats, g.slotinfo, bg, im.final_T, im.n_contrib, dL_dpix, dL_dacc, b.grad_inst, b.inst_flag,
                       g.touched, order, im.rangesB);
  } else {
    constexpr int TW = 4;  // (1, 2 and 4 tiles per workgroup measured equal)
    hipLaunchKernelGGL(k_blend_backward_tile<TW>, dim3((tiles + TW - 1) / TW), dim3(64 * TW), 0, s, fp, im.ranges,
                       im.quad_last, b.point_list, g.splats, g.slotinfo, bg, im.final_T, im.n_contrib, dL_dpix, dL_dacc,
                       reinterpret_cast<float*>(b.grad_inst), b.inst_flag, g.touched, order, im.rangesB);
  }
  return hipGetLastError();
}

}  // namespace gsr
