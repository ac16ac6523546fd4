/*
 * gsr_oracle.c -- CPU restatement of GS-LIVM's tile rasterizer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (gs-livm_amd/, the
 * C-ABI library, the Torch binding) may include, link or call this file.  It is
 * used by tests/, by __graft_entry__.smoke() as the checker and by bench.py's
 * cpu_baseline leg.
 *
 * PARITY STATUS: "parity unpinned".  The reference (CUDA, needs nvcc + CUB +
 * CUDA headers) cannot be built in this image and ships no tests, golden
 * vectors or fixtures for this path.  What IS pinned:
 *  - the single-Gaussian known answer the survey recorded from the reference
 *    (radius 33, xy (340.8333,196.8333), conic
 *    (0.008766,0.000005,0.053982), 25 tiles): tests/test_oracle.py;
 *  - the 3x3 product / transpose / dot / length helpers below, bit for bit
 *    against the reference's own vendored GLM (oracle/ref_glm/check_glm.cpp is
 *    compiled straight from /root/reference/external/glm into oracle/_ref/).
 * Everything else (culling rules, EWA, SH, sort, blending, the whole backward)
 * is a restatement checked only against itself.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference).  Arithmetic is IEEE f32 with the reference's evaluation
 * order and NO fused multiply-add (build with -ffp-contract=off): the integer
 * results (radii, tile rects, sort keys, tile ranges) depend on that rounding.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE 16 /* BLOCK_X == BLOCK_Y == 16: include/gs/cuda_rasterizer/config.h:16-17 */

/* ---- small dense helpers; 3x3 matrices are stored column-major (m[col][row]),
 * the convention of the vendored GLM that the reference computes with. ---- */
typedef struct { float c[3][3]; } m3;

/* external/glm/glm/detail/type_mat3x3.inl:486-518: out(r,c) = sum_k a(r,k) b(k,c),
 * accumulated left to right over k = 0,1,2. */
static m3 m3_mul(const m3* a, const m3* b) {
  m3 o;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++)
      o.c[c][r] = a->c[0][r] * b->c[c][0] + a->c[1][r] * b->c[c][1] + a->c[2][r] * b->c[c][2];
  return o;
}
static m3 m3_t(const m3* a) {
  m3 o;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) o.c[c][r] = a->c[r][c];
  return o;
}
static m3 m3_cols(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2) {
  m3 o = {{{a0, a1, a2}, {b0, b1, b2}, {c0, c1, c2}}};
  return o;
}
/* CUDA's min/max on floats are fminf/fmaxf (the non-NaN operand wins) */
static inline float fminf_(float a, float b) { return fminf(a, b); }
static inline float fmaxf_(float a, float b) { return fmaxf(a, b); }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* include/gs/cuda_rasterizer/auxiliary.h:48-64 (transformPoint4x3 / 4x4): matrices
 * are column-major 4x4 in memory. */
static void xform43(const float* p, const float* m, float* o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static void xform44(const float* p, const float* m, float* o) {
  xform43(p, m, o);
  o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* auxiliary.h:35-37: the literals are doubles, so this is evaluated in f64 and
 * rounded to f32 once on return. */
static float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

/* auxiliary.h:39-46 (getRect): float arithmetic with int->float conversions at each
 * step, truncating casts, clamp to [0, grid]. */
static void tile_rect(float px, float py, int radius, int gx, int gy, int* x0, int* y0, int* x1, int* y1) {
  float r = (float)radius;
  *x0 = imin(gx, imax(0, (int)((px - r) / (float)TILE)));
  *y0 = imin(gy, imax(0, (int)((py - r) / (float)TILE)));
  *x1 = imin(gx, imax(0, (int)((((px + r) + (float)TILE) - 1.0f) / (float)TILE)));
  *y1 = imin(gy, imax(0, (int)((((py + r) + (float)TILE) - 1.0f) / (float)TILE)));
}

/* SH basis constants: auxiliary.h:22-33 */
static const float C0 = 0.28209479177387814f;
static const float C1 = 0.4886025119029199f;
static const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                            0.5462742152960396f};
static const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                            -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

/* src/cuda_rasterizer/forward.cu:29-76 (computeColorFromSH, forward).  sh layout
 * [M][3]; result per channel; clamp flags recorded. */
static void sh_to_rgb(int deg, int M, const float* mean, const float* campos, const float* sh, float* rgb,
                      uint8_t* clamped) {
  float d[3] = {mean[0] - campos[0], mean[1] - campos[1], mean[2] - campos[2]};
  float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]); /* glm::length: (x*x + y*y) + z*z */
  float x = d[0] / len, y = d[1] / len, z = d[2] / len;
  (void)M;
  for (int ch = 0; ch < 3; ch++) {
#define SH(k) sh[(k) * 3 + ch]
    float res = C0 * SH(0);
    if (deg > 0) {
      res = res - C1 * y * SH(1) + C1 * z * SH(2) - C1 * x * SH(3);
      if (deg > 1) {
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        res = res + C2[0] * xy * SH(4) + C2[1] * yz * SH(5) + C2[2] * (2.0f * zz - xx - yy) * SH(6) +
              C2[3] * xz * SH(7) + C2[4] * (xx - yy) * SH(8);
        if (deg > 2) {
          res = res + C3[0] * y * (3.0f * xx - yy) * SH(9) + C3[1] * xy * z * SH(10) +
                C3[2] * y * (4.0f * zz - xx - yy) * SH(11) + C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + C3[5] * z * (xx - yy) * SH(14) +
                C3[6] * x * (xx - 3.0f * yy) * SH(15);
        }
      }
    }
#undef SH
    res += 0.5f;
    clamped[ch] = (res < 0);
    rgb[ch] = fmaxf_(res, 0.0f);
  }
}

/* forward.cu:138-176 (computeCov3D): quaternion used as given (r,x,y,z), M = S*R,
 * Sigma = M^T M, upper triangle stored. */
static void cov3d_from_scale_rot(const float* scale, float mod, const float* q, float* out6) {
  m3 S = m3_cols(mod * scale[0], 0, 0, 0, mod * scale[1], 0, 0, 0, mod * scale[2]);
  float r = q[0], x = q[1], y = q[2], z = q[3];
  m3 R = m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y), 2.f * (x * y + r * z),
                 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x), 2.f * (x * z - r * y), 2.f * (y * z + r * x),
                 1.f - 2.f * (x * x + y * y));
  m3 Mm = m3_mul(&S, &R);
  m3 Mt = m3_t(&Mm);
  m3 Sg = m3_mul(&Mt, &Mm);
  out6[0] = Sg.c[0][0];
  out6[1] = Sg.c[0][1];
  out6[2] = Sg.c[0][2];
  out6[3] = Sg.c[1][1];
  out6[4] = Sg.c[1][2];
  out6[5] = Sg.c[2][2];
}

/* Shared by forward.cu:79-133 (computeCov2D) and backward.cu:159-200: builds T = W*J
 * and cov = T^T Vrk^T T (before the +0.3 low-pass).  t_out = clamped view-space mean. */
static void ewa_project(const float* mean, float fx, float fy, float tanx, float tany, const float* c6,
                        const float* V, m3* T, m3* Vrk, m3* W, m3* cov, float* t_out, float* txtz, float* tytz) {
  float t[3];
  xform43(mean, V, t);
  const float limx = 1.3f * tanx, limy = 1.3f * tany;
  *txtz = t[0] / t[2];
  *tytz = t[1] / t[2];
  t[0] = fminf_(limx, fmaxf_(-limx, *txtz)) * t[2];
  t[1] = fminf_(limy, fmaxf_(-limy, *tytz)) * t[2];
  m3 J = m3_cols(fx / t[2], 0.0f, -(fx * t[0]) / (t[2] * t[2]), 0.0f, fy / t[2], -(fy * t[1]) / (t[2] * t[2]), 0, 0, 0);
  *W = m3_cols(V[0], V[4], V[8], V[1], V[5], V[9], V[2], V[6], V[10]);
  *T = m3_mul(W, &J);
  *Vrk = m3_cols(c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]);
  m3 Tt = m3_t(T), Vt = m3_t(Vrk);
  m3 A = m3_mul(&Tt, &Vt);
  *cov = m3_mul(&A, T);
  t_out[0] = t[0];
  t_out[1] = t[1];
  t_out[2] = t[2];
}

typedef struct gsro_frame {
  int P, D, M, W, H, gx, gy, R;
  /* per Gaussian (GeometryState, rasterizer_impl.cu:137-151) */
  int* radii;
  float* means2D;       /* [P][2] */
  float* depths;        /* [P] */
  float* cov3D;         /* [P][6] */
  float* rgb;           /* [P][3] */
  float* conic_opacity; /* [P][4] */
  uint32_t* tiles_touched;
  uint32_t* point_offsets;
  uint8_t* clamped; /* [P][3] */
  /* per instance (BinningState, rasterizer_impl.cu:161-177) */
  uint64_t* keys_unsorted;
  uint32_t* values_unsorted;
  uint64_t* keys;
  uint32_t* point_list;
  /* per image (ImageState, rasterizer_impl.cu:153-159) */
  uint32_t* ranges; /* [T][2] */
  float* final_T;
  uint32_t* n_contrib;
  float *out_color, *out_depth, *out_acc;
  uint8_t* fragile; /* [H*W] 1 = some threshold test of this pixel sat within rounding distance (see blend_tile) */
  int own_cov3D, own_rgb;
  const float* colors_used; /* rgb or colors_precomp */
  const float* cov3D_used;
} gsro_frame;

void gsro_free(gsro_frame* f) {
  if (!f) return;
  free(f->radii); free(f->means2D); free(f->depths); free(f->cov3D); free(f->rgb); free(f->conic_opacity);
  free(f->tiles_touched); free(f->point_offsets); free(f->clamped); free(f->keys_unsorted);
  free(f->values_unsorted); free(f->keys); free(f->point_list); free(f->ranges); free(f->final_T);
  free(f->n_contrib); free(f->out_color); free(f->out_depth); free(f->out_acc); free(f->fragile);
  free(f);
}

/* ---- NOT reference behaviour: the product's tile culling, restated so its integer stages can be checked ----
 * With g_tight != 0 (gsro_set_tight; default 0 = the reference's rectangles) a Gaussian's tile rectangle is
 * cut down to the tiles overlapped by the axis-aligned box that contains every pixel with
 * alpha = op * exp(power) >= 1/255 (gs-livm_amd/csrc/preprocess.hip, k_preprocess).  Instances outside that box
 * are skipped pixel by pixel by the reference (forward.cu:343-345), so colour/depth/acc and every gradient
 * are identical in both modes (tests/test_oracle.py checks that bit for bit); tiles_touched, point_offsets,
 * the sorted lists, ranges and n_contrib are the ones that differ.  Pure +,*,/ arithmetic: -ffp-contract=off
 * here and in preprocess.hip make it bit-identical on both sides. */
static int g_tight = 0;
void gsro_set_tight(int on) { g_tight = on != 0; }
int gsro_get_tight(void) { return g_tight; }

static float ln_upper(float x) {
  uint32_t b;
  memcpy(&b, &x, 4);
  int e = (int)(b >> 23) - 127;
  uint32_t mb = (b & 0x007FFFFFu) | 0x3F800000u;
  float m;
  memcpy(&m, &mb, 4);
  float t = (m - 1.0f) / (m + 1.0f);
  float t2 = t * t;
  float s = t * (2.0f + t2 * (0.6666667f + t2 * 0.4f));
  return (float)e * 0.6931472f + s + 3e-4f;
}

/* narrows [x0,x1) x [y0,y1); an empty result has x1 <= x0 or y1 <= y0 */
static void tighten_rect(float px, float py, const float conic[3], float op, int* x0, int* y0, int* x1, int* y1) {
  float hx, hy;
  if (op < 1.0f / 255.0f) {
    *x1 = *x0;
    return;
  }
  float tau = ln_upper(255.0f * op) * 1.01f + 0.02f;
  float dc = conic[0] * conic[2] - conic[1] * conic[1];
  if (!(conic[0] > 0.0f && conic[2] > 0.0f && dc > 0.0f)) return; /* indefinite conic: no culling */
  hx = sqrtf(2.0f * tau * conic[2] / dc) + 0.05f;
  hy = sqrtf(2.0f * tau * conic[0] / dc) + 0.05f;
  if (!(hx < 1e6f)) return;
  const float lim = 1e6f;
  int bx0 = (int)ceilf(fmaxf_(-lim, fminf_(lim, (px - hx - 15.0f) * 0.0625f)));
  int bx1 = (int)floorf(fmaxf_(-lim, fminf_(lim, (px + hx) * 0.0625f))) + 1;
  int by0 = (int)ceilf(fmaxf_(-lim, fminf_(lim, (py - hy - 15.0f) * 0.0625f)));
  int by1 = (int)floorf(fmaxf_(-lim, fminf_(lim, (py + hy) * 0.0625f))) + 1;
  if (bx0 > *x0) *x0 = bx0;
  if (bx1 < *x1) *x1 = bx1;
  if (by0 > *y0) *y0 = by0;
  if (by1 < *y1) *y1 = by1;
}

/* forward.cu:179-286 (preprocessCUDA), one Gaussian. */
static void preprocess_one(gsro_frame* f, int idx, const float* means3D, const float* scales, float mod,
                           const float* rotations, const float* opacities, const float* shs,
                           const float* cov3D_precomp, const float* colors_precomp, const float* V,
                           const float* Pm, const float* campos, float tanx, float tany, float fx, float fy) {
  f->radii[idx] = 0;
  f->tiles_touched[idx] = 0;
  const float* p = means3D + 3 * idx;
  float pv[3];
  xform43(p, V, pv);
  if (pv[2] <= 0.2f) return; /* near cull only: forward.cu:221-225 */
  if (scales) {               /* scale cull: forward.cu:19-25,227-229 */
    const float* s = scales + 3 * idx;
    if (mod * s[0] > 0.3f || mod * s[1] > 0.3f || mod * s[2] > 0.3f) return;
  }
  float ph[4];
  xform44(p, Pm, ph);
  float pw = 1.0f / (ph[3] + 0.0000001f);
  float ppx = ph[0] * pw, ppy = ph[1] * pw;
  const float* c6;
  if (cov3D_precomp) {
    c6 = cov3D_precomp + 6 * idx;
  } else {
    cov3d_from_scale_rot(scales + 3 * idx, mod, rotations + 4 * idx, f->cov3D + 6 * idx);
    c6 = f->cov3D + 6 * idx;
  }
  m3 T, Vrk, Wm, cov;
  float t[3], a, b;
  ewa_project(p, fx, fy, tanx, tany, c6, V, &T, &Vrk, &Wm, &cov, t, &a, &b);
  float cx = cov.c[0][0] + 0.3f, cy = cov.c[0][1], cz = cov.c[1][1] + 0.3f; /* low-pass: forward.cu:130-131 */
  float det = cx * cz - cy * cy;
  if (det == 0.0f) return;
  float det_inv = 1.f / det;
  float conic[3] = {cz * det_inv, -cy * det_inv, cx * det_inv};
  float mid = 0.5f * (cx + cz);
  float lambda1 = mid + sqrtf(fmaxf_(0.1f, mid * mid - det));
  float lambda2 = mid - sqrtf(fmaxf_(0.1f, mid * mid - det));
  float my_radius = ceilf(3.f * sqrtf(fmaxf_(lambda1, lambda2)));
  float pix[2] = {ndc2pix(ppx, f->W), ndc2pix(ppy, f->H)};
  int x0, y0, x1, y1;
  tile_rect(pix[0], pix[1], (int)my_radius, f->gx, f->gy, &x0, &y0, &x1, &y1);
  if ((x1 - x0) * (y1 - y0) == 0) return;
  if (!colors_precomp) sh_to_rgb(f->D, f->M, p, campos, shs + (size_t)idx * f->M * 3, f->rgb + 3 * idx, f->clamped + 3 * idx);
  f->depths[idx] = pv[2];
  f->radii[idx] = (int)my_radius;
  f->means2D[2 * idx] = pix[0];
  f->means2D[2 * idx + 1] = pix[1];
  f->conic_opacity[4 * idx + 0] = conic[0];
  f->conic_opacity[4 * idx + 1] = conic[1];
  f->conic_opacity[4 * idx + 2] = conic[2];
  f->conic_opacity[4 * idx + 3] = opacities[idx];
  if (g_tight) {
    tighten_rect(pix[0], pix[1], conic, opacities[idx], &x0, &y0, &x1, &y1);
    f->tiles_touched[idx] = (x1 > x0 && y1 > y0) ? (uint32_t)((y1 - y0) * (x1 - x0)) : 0u;
  } else {
    f->tiles_touched[idx] = (uint32_t)((y1 - y0) * (x1 - x0));
  }
}

/* Stable LSD radix sort of (u64 key, u32 value) pairs: the semantics of
 * cub::DeviceRadixSort::SortPairs(begin_bit=0, end_bit=32+bit) at
 * rasterizer_impl.cu:298-309 (bits above end_bit are zero in every key, so sorting
 * on the whole key gives the same permutation). */
static void stable_sort_pairs(int n, const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout) {
  uint64_t* ka = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n ? n : 1));
  uint32_t* va = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n ? n : 1));
  uint64_t* kb = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n ? n : 1));
  uint32_t* vb = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n ? n : 1));
  memcpy(ka, kin, sizeof(uint64_t) * (size_t)n);
  memcpy(va, vin, sizeof(uint32_t) * (size_t)n);
  for (int pass = 0; pass < 4; pass++) {
    size_t cnt[65537];
    memset(cnt, 0, sizeof(cnt));
    int sh = 16 * pass;
    for (int i = 0; i < n; i++) cnt[((ka[i] >> sh) & 0xFFFF) + 1]++;
    for (int d = 0; d < 65536; d++) cnt[d + 1] += cnt[d];
    for (int i = 0; i < n; i++) {
      size_t o = cnt[(ka[i] >> sh) & 0xFFFF]++;
      kb[o] = ka[i];
      vb[o] = va[i];
    }
    uint64_t* tk = ka; ka = kb; kb = tk;
    uint32_t* tv = va; va = vb; vb = tv;
  }
  memcpy(kout, ka, sizeof(uint64_t) * (size_t)n);
  memcpy(vout, va, sizeof(uint32_t) * (size_t)n);
  free(ka); free(va); free(kb); free(vb);
}

/* forward.cu:291-407 (renderCUDA forward), one tile.  The reference's rounds of 256
 * with block-wide early exit do not change any pixel's result, so each pixel simply
 * walks the tile's list front to back. */
static void blend_tile(gsro_frame* f, int tx, int ty, const float* bg) {
  const int W = f->W, H = f->H;
  const uint32_t b = f->ranges[2 * (ty * f->gx + tx)], e = f->ranges[2 * (ty * f->gx + tx) + 1];
  const float* feat = f->colors_used;
  for (int ly = 0; ly < TILE; ly++)
    for (int lx = 0; lx < TILE; lx++) {
      int px = tx * TILE + lx, py = ty * TILE + ly;
      if (px >= W || py >= H) continue;
      float pfx = (float)px, pfy = (float)py;
      float T = 1.0f, C[3] = {0, 0, 0}, Dp = 0, A = 0;
      uint32_t contributor = 0, last = 0;
      uint8_t fragile = 0;
      float t_unc = 0.0f;
      for (uint32_t i = b; i < e; i++) {
        contributor++;
        uint32_t g = f->point_list[i];
        float dx = f->means2D[2 * g] - pfx, dy = f->means2D[2 * g + 1] - pfy;
        const float* co = f->conic_opacity + 4 * g;
        float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (power > 0.0f) continue;
        float alpha = fminf_(0.99f, co[3] * expf(power));
        /* Not part of the reference: flag pixels where a discontinuous test is decided by less than
         * the rounding differences any other implementation has (FMA contraction, exp ulps): the
         * 1/255 alpha cut, the T < 1e-4 stop (T carries accumulated error) and power > 0.  A flip
         * there legitimately changes the pixel by up to alpha*T, so parity tests compare such
         * pixels with the looser bound. */
        /* The power is a quadratic form whose three terms can cancel (a nearly singular conic along its long
         * axis, hundreds of pixels from the centre: terms of 1e3 summing to ~5): its f32 rounding uncertainty is
         * ~ a few ulp of the LARGEST term, and alpha inherits it as a RELATIVE error (alpha = op * e^power).  The
         * window of the alpha cut and of power > 0 therefore scales with that uncertainty. */
        float unc = 6e-7f * (fabsf(co[0] * dx * dx) + fabsf(co[2] * dy * dy) + 2.0f * fabsf(co[1] * dx * dy));
        if (fabsf(alpha * 255.0f - 1.0f) < 2e-5f + unc || fabsf(power) < 1e-6f + unc) fragile = 1;
        if (alpha < 1.0f / 255.0f) continue;
        float test_T = T * (1 - alpha);
        t_unc += alpha * unc / (1.0f - alpha); /* relative uncertainty T inherits from the alphas before it */
        if (fabsf(test_T * 10000.0f - 1.0f) < 2e-4f + t_unc) fragile = 1;
        if (test_T < 0.0001f) break; /* done = true */
        for (int ch = 0; ch < 3; ch++) C[ch] += feat[3 * g + ch] * alpha * T;
        Dp += f->depths[g] * alpha * T; /* forward.cu:386 */
        A += alpha * T;                 /* forward.cu:387 */
        T = test_T;
        last = contributor;
      }
      size_t pid = (size_t)W * py + px;
      f->final_T[pid] = T;
      f->n_contrib[pid] = last;
      for (int ch = 0; ch < 3; ch++) f->out_color[(size_t)ch * H * W + pid] = C[ch] + T * bg[ch];
      f->out_depth[pid] = Dp;
      f->out_acc[pid] = A;
      f->fragile[pid] = fragile;
    }
}

/* rasterizer_impl.cu:181-342 (Rasterizer::forward): F1 preprocess, F2 inclusive scan,
 * F4 duplicateWithKeys (:64-101), F5 stable sort, F6/F7 tile ranges (:106-125, :311),
 * F8 blend.  Returns a frame that owns every intermediate. */
gsro_frame* gsro_forward(int P, int D, int M, const float* background, int W, int H, const float* means3D,
                         const float* shs, const float* colors_precomp, const float* opacities,
                         const float* scales, float scale_modifier, const float* rotations,
                         const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                         const float* cam_pos, float tan_fovx, float tan_fovy) {
  gsro_frame* f = (gsro_frame*)calloc(1, sizeof(gsro_frame));
  f->P = P; f->D = D; f->M = M; f->W = W; f->H = H;
  f->gx = (W + TILE - 1) / TILE;
  f->gy = (H + TILE - 1) / TILE;
  const float fy = H / (2.0f * tan_fovy), fx = W / (2.0f * tan_fovx); /* rasterizer_impl.cu:210-211 */
  size_t Pn = (size_t)(P ? P : 1), N = (size_t)W * H, Tn = (size_t)f->gx * f->gy;
  f->radii = (int*)calloc(Pn, sizeof(int));
  f->means2D = (float*)calloc(Pn * 2, sizeof(float));
  f->depths = (float*)calloc(Pn, sizeof(float));
  f->cov3D = (float*)calloc(Pn * 6, sizeof(float));
  f->rgb = (float*)calloc(Pn * 3, sizeof(float));
  f->conic_opacity = (float*)calloc(Pn * 4, sizeof(float));
  f->tiles_touched = (uint32_t*)calloc(Pn, sizeof(uint32_t));
  f->point_offsets = (uint32_t*)calloc(Pn, sizeof(uint32_t));
  f->clamped = (uint8_t*)calloc(Pn * 3, 1);
  f->ranges = (uint32_t*)calloc((Tn ? Tn : 1) * 2, sizeof(uint32_t));
  f->final_T = (float*)calloc(N ? N : 1, sizeof(float));
  f->n_contrib = (uint32_t*)calloc(N ? N : 1, sizeof(uint32_t));
  f->out_color = (float*)calloc((N ? N : 1) * 3, sizeof(float));
  f->out_depth = (float*)calloc(N ? N : 1, sizeof(float));
  f->out_acc = (float*)calloc(N ? N : 1, sizeof(float));
  f->fragile = (uint8_t*)calloc(N ? N : 1, 1);
  f->colors_used = colors_precomp ? colors_precomp : f->rgb;
  f->cov3D_used = cov3D_precomp ? cov3D_precomp : f->cov3D;

#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++)
    preprocess_one(f, i, means3D, scales, scale_modifier, rotations, opacities, shs, cov3D_precomp, colors_precomp,
                   viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, fx, fy);

  uint32_t run = 0; /* cub::DeviceScan::InclusiveSum, rasterizer_impl.cu:270-273 */
  for (int i = 0; i < P; i++) {
    run += f->tiles_touched[i];
    f->point_offsets[i] = run;
  }
  int R = P ? (int)f->point_offsets[P - 1] : 0;
  f->R = R;
  size_t Rn = (size_t)(R ? R : 1);
  f->keys_unsorted = (uint64_t*)calloc(Rn, sizeof(uint64_t));
  f->values_unsorted = (uint32_t*)calloc(Rn, sizeof(uint32_t));
  f->keys = (uint64_t*)calloc(Rn, sizeof(uint64_t));
  f->point_list = (uint32_t*)calloc(Rn, sizeof(uint32_t));

#pragma omp parallel for schedule(dynamic, 1024)
  for (int i = 0; i < P; i++) { /* duplicateWithKeys */
    if (f->radii[i] <= 0) continue;
    uint32_t off = i == 0 ? 0 : f->point_offsets[i - 1];
    int x0, y0, x1, y1;
    tile_rect(f->means2D[2 * i], f->means2D[2 * i + 1], f->radii[i], f->gx, f->gy, &x0, &y0, &x1, &y1);
    if (g_tight) {
      if (!f->tiles_touched[i]) continue;
      tighten_rect(f->means2D[2 * i], f->means2D[2 * i + 1], f->conic_opacity + 4 * i, f->conic_opacity[4 * i + 3],
                   &x0, &y0, &x1, &y1);
    }
    uint32_t dbits;
    memcpy(&dbits, &f->depths[i], 4);
    for (int y = y0; y < y1; y++)
      for (int x = x0; x < x1; x++) {
        uint64_t key = (uint64_t)(uint32_t)(y * f->gx + x);
        key <<= 32;
        key |= dbits;
        f->keys_unsorted[off] = key;
        f->values_unsorted[off] = (uint32_t)i;
        off++;
      }
  }
  stable_sort_pairs(R, f->keys_unsorted, f->values_unsorted, f->keys, f->point_list);

  for (int i = 0; i < R; i++) { /* identifyTileRanges on zeroed ranges */
    uint32_t cur = (uint32_t)(f->keys[i] >> 32);
    if (i == 0)
      f->ranges[2 * cur] = 0;
    else {
      uint32_t prev = (uint32_t)(f->keys[i - 1] >> 32);
      if (cur != prev) {
        f->ranges[2 * prev + 1] = (uint32_t)i;
        f->ranges[2 * cur] = (uint32_t)i;
      }
    }
    if (i == R - 1) f->ranges[2 * cur + 1] = (uint32_t)R;
  }

#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int ty = 0; ty < f->gy; ty++)
    for (int tx = 0; tx < f->gx; tx++) blend_tile(f, tx, ty, background);
  return f;
}

static inline void addf(float* p, float v, int par) {
  if (par) {
#pragma omp atomic
    *p += v;
  } else {
    *p += v;
  }
}

/* backward.cu:438-603 (renderCUDA backward), one tile; accumulation order with one
 * thread = tiles row-major, pixels row-major inside a tile, list back to front
 * (f32 adds, like the reference's atomicAdd; the reference's own order is arbitrary). */
static void blend_tile_bwd(const gsro_frame* f, int tx, int ty, const float* bg, const float* dL_dpix,
                           const float* dL_dacc, float* g_mean2D, float* g_conic, float* g_opacity,
                           float* g_color, int par) {
  const int W = f->W, H = f->H;
  const uint32_t b = f->ranges[2 * (ty * f->gx + tx)], e = f->ranges[2 * (ty * f->gx + tx) + 1];
  const float* feat = f->colors_used;
  const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H); /* backward.cu:505-506 */
  for (int ly = 0; ly < TILE; ly++)
    for (int lx = 0; lx < TILE; lx++) {
      int px = tx * TILE + lx, py = ty * TILE + ly;
      if (px >= W || py >= H) continue;
      size_t pid = (size_t)W * py + px;
      float pfx = (float)px, pfy = (float)py;
      const float T_final = f->final_T[pid];
      float T = T_final;
      uint32_t contributor = e - b;
      const uint32_t last = f->n_contrib[pid];
      float accum_rec[3] = {0, 0, 0}, dpix[3], last_color[3] = {0, 0, 0};
      for (int ch = 0; ch < 3; ch++) dpix[ch] = dL_dpix[(size_t)ch * H * W + pid];
      const float dacc = dL_dacc[pid];
      float accum_acc_rec = 0, last_alpha = 0, last_acc = 0;
      for (uint32_t k = 0; k < e - b; k++) {
        contributor--;
        if (contributor >= last) continue;
        uint32_t g = f->point_list[e - k - 1];
        float dx = f->means2D[2 * g] - pfx, dy = f->means2D[2 * g + 1] - pfy;
        const float* co = f->conic_opacity + 4 * g;
        float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (power > 0.0f) continue;
        const float G = expf(power);
        const float alpha = fminf_(0.99f, co[3] * G);
        if (alpha < 1.0f / 255.0f) continue;
        T = T / (1.f - alpha);
        const float dchannel_dcolor = alpha * T;
        float dL_dalpha = 0.0f;
        for (int ch = 0; ch < 3; ch++) {
          const float c = feat[3 * g + ch];
          accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
          last_color[ch] = c;
          dL_dalpha += (c - accum_rec[ch]) * dpix[ch];
          addf(&g_color[3 * g + ch], dchannel_dcolor * dpix[ch], par);
        }
        const float c_d = 1.0f; /* acc path with constant 1: backward.cu:567-571 */
        accum_acc_rec = last_alpha * last_acc + (1.f - last_alpha) * accum_acc_rec;
        last_acc = c_d;
        dL_dalpha += (c_d - accum_acc_rec) * dacc;
        dL_dalpha *= T;
        last_alpha = alpha;
        float bg_dot = 0;
        for (int ch = 0; ch < 3; ch++) bg_dot += bg[ch] * dpix[ch];
        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
        const float dL_dG = co[3] * dL_dalpha;
        const float gdx = G * dx, gdy = G * dy;
        const float dG_ddelx = -gdx * co[0] - gdy * co[1];
        const float dG_ddely = -gdy * co[2] - gdx * co[1];
        addf(&g_mean2D[3 * g + 0], dL_dG * dG_ddelx * ddelx_dx, par);
        addf(&g_mean2D[3 * g + 1], dL_dG * dG_ddely * ddely_dy, par);
        addf(&g_conic[4 * g + 0], -0.5f * gdx * dx * dL_dG, par);
        addf(&g_conic[4 * g + 1], -0.5f * gdx * dy * dL_dG, par);
        addf(&g_conic[4 * g + 3], -0.5f * gdy * dy * dL_dG, par);
        addf(&g_opacity[g], G * dL_dalpha, par);
      }
    }
}

/* backward.cu:140-275 (computeCov2DCUDA), one Gaussian: assigns dL_dcov3D and the
 * covariance-induced part of dL_dmean3D. */
static void cov2d_bwd_one(const gsro_frame* f, int idx, const float* means3D, float fx, float fy, float tanx,
                          float tany, const float* V, const float* g_conic, float* g_mean3D, float* g_cov3D) {
  const float* c6 = f->cov3D_used + 6 * idx;
  float dcx = g_conic[4 * idx], dcy = g_conic[4 * idx + 1], dcz = g_conic[4 * idx + 3];
  m3 T, Vrk, Wm, cov;
  float t[3], txtz, tytz;
  ewa_project(means3D + 3 * idx, fx, fy, tanx, tany, c6, V, &T, &Vrk, &Wm, &cov, t, &txtz, &tytz);
  const float limx = 1.3f * tanx, limy = 1.3f * tany;
  const float xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
  const float ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
  float a = cov.c[0][0] + 0.3f, b = cov.c[0][1], c = cov.c[1][1] + 0.3f;
  float denom = a * c - b * b;
  float dL_da = 0, dL_db = 0, dL_dc = 0;
  float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
  float* o = g_cov3D + 6 * idx;
#define Tm(ci, ri) T.c[ci][ri]
  if (denom2inv != 0) {
    dL_da = denom2inv * (-c * c * dcx + 2 * b * c * dcy + (denom - a * c) * dcz);
    dL_dc = denom2inv * (-a * a * dcz + 2 * a * b * dcy + (denom - a * c) * dcx);
    dL_db = denom2inv * 2 * (b * c * dcx - (denom + 2 * b * b) * dcy + a * b * dcz);
    o[0] = (Tm(0, 0) * Tm(0, 0) * dL_da + Tm(0, 0) * Tm(1, 0) * dL_db + Tm(1, 0) * Tm(1, 0) * dL_dc);
    o[3] = (Tm(0, 1) * Tm(0, 1) * dL_da + Tm(0, 1) * Tm(1, 1) * dL_db + Tm(1, 1) * Tm(1, 1) * dL_dc);
    o[5] = (Tm(0, 2) * Tm(0, 2) * dL_da + Tm(0, 2) * Tm(1, 2) * dL_db + Tm(1, 2) * Tm(1, 2) * dL_dc);
    o[1] = 2 * Tm(0, 0) * Tm(0, 1) * dL_da + (Tm(0, 0) * Tm(1, 1) + Tm(0, 1) * Tm(1, 0)) * dL_db +
           2 * Tm(1, 0) * Tm(1, 1) * dL_dc;
    o[2] = 2 * Tm(0, 0) * Tm(0, 2) * dL_da + (Tm(0, 0) * Tm(1, 2) + Tm(0, 2) * Tm(1, 0)) * dL_db +
           2 * Tm(1, 0) * Tm(1, 2) * dL_dc;
    o[4] = 2 * Tm(0, 2) * Tm(0, 1) * dL_da + (Tm(0, 1) * Tm(1, 2) + Tm(0, 2) * Tm(1, 1)) * dL_db +
           2 * Tm(1, 1) * Tm(1, 2) * dL_dc;
  } else {
    for (int i = 0; i < 6; i++) o[i] = 0;
  }
#define Vk(ci, ri) Vrk.c[ci][ri]
  float dT00 = 2 * (Tm(0, 0) * Vk(0, 0) + Tm(0, 1) * Vk(0, 1) + Tm(0, 2) * Vk(0, 2)) * dL_da +
               (Tm(1, 0) * Vk(0, 0) + Tm(1, 1) * Vk(0, 1) + Tm(1, 2) * Vk(0, 2)) * dL_db;
  float dT01 = 2 * (Tm(0, 0) * Vk(1, 0) + Tm(0, 1) * Vk(1, 1) + Tm(0, 2) * Vk(1, 2)) * dL_da +
               (Tm(1, 0) * Vk(1, 0) + Tm(1, 1) * Vk(1, 1) + Tm(1, 2) * Vk(1, 2)) * dL_db;
  float dT02 = 2 * (Tm(0, 0) * Vk(2, 0) + Tm(0, 1) * Vk(2, 1) + Tm(0, 2) * Vk(2, 2)) * dL_da +
               (Tm(1, 0) * Vk(2, 0) + Tm(1, 1) * Vk(2, 1) + Tm(1, 2) * Vk(2, 2)) * dL_db;
  float dT10 = 2 * (Tm(1, 0) * Vk(0, 0) + Tm(1, 1) * Vk(0, 1) + Tm(1, 2) * Vk(0, 2)) * dL_dc +
               (Tm(0, 0) * Vk(0, 0) + Tm(0, 1) * Vk(0, 1) + Tm(0, 2) * Vk(0, 2)) * dL_db;
  float dT11 = 2 * (Tm(1, 0) * Vk(1, 0) + Tm(1, 1) * Vk(1, 1) + Tm(1, 2) * Vk(1, 2)) * dL_dc +
               (Tm(0, 0) * Vk(1, 0) + Tm(0, 1) * Vk(1, 1) + Tm(0, 2) * Vk(1, 2)) * dL_db;
  float dT12 = 2 * (Tm(1, 0) * Vk(2, 0) + Tm(1, 1) * Vk(2, 1) + Tm(1, 2) * Vk(2, 2)) * dL_dc +
               (Tm(0, 0) * Vk(2, 0) + Tm(0, 1) * Vk(2, 1) + Tm(0, 2) * Vk(2, 2)) * dL_db;
#undef Vk
#undef Tm
  float dJ00 = Wm.c[0][0] * dT00 + Wm.c[0][1] * dT01 + Wm.c[0][2] * dT02;
  float dJ02 = Wm.c[2][0] * dT00 + Wm.c[2][1] * dT01 + Wm.c[2][2] * dT02;
  float dJ11 = Wm.c[1][0] * dT10 + Wm.c[1][1] * dT11 + Wm.c[1][2] * dT12;
  float dJ12 = Wm.c[2][0] * dT10 + Wm.c[2][1] * dT11 + Wm.c[2][2] * dT12;
  float tz = 1.f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
  float dtx = xmul * -fx * tz2 * dJ02;
  float dty = ymul * -fy * tz2 * dJ12;
  float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * t[0]) * tz3 * dJ02 + (2 * fy * t[1]) * tz3 * dJ12;
  /* transformVec4x3Transpose, auxiliary.h:75-82 */
  g_mean3D[3 * idx + 0] = V[0] * dtx + V[1] * dty + V[2] * dtz;
  g_mean3D[3 * idx + 1] = V[4] * dtx + V[5] * dty + V[6] * dtz;
  g_mean3D[3 * idx + 2] = V[8] * dtx + V[9] * dty + V[10] * dtz;
}

/* backward.cu:20-135 (computeColorFromSH backward). */
static void sh_bwd_one(int deg, int M, const float* mean, const float* campos, const float* sh, const uint8_t* clamped,
                       const float* g_rgb_in, float* g_mean3D, float* g_sh) {
  float d0[3] = {mean[0] - campos[0], mean[1] - campos[1], mean[2] - campos[2]};
  float len = sqrtf(d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2]);
  float x = d0[0] / len, y = d0[1] / len, z = d0[2] / len;
  float g[3];
  for (int ch = 0; ch < 3; ch++) g[ch] = g_rgb_in[ch] * (clamped[ch] ? 0.f : 1.f);
  float ddir[3] = {0, 0, 0};
  float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
  (void)M;
#define SH(k) sh[(k) * 3 + ch]
#define GS(k) g_sh[(k) * 3 + ch]
  for (int ch = 0; ch < 3; ch++) {
    GS(0) = C0 * g[ch];
    if (deg > 0) {
      GS(1) = (-C1 * y) * g[ch];
      GS(2) = (C1 * z) * g[ch];
      GS(3) = (-C1 * x) * g[ch];
      dRGBdx[ch] = -C1 * SH(3);
      dRGBdy[ch] = -C1 * SH(1);
      dRGBdz[ch] = C1 * SH(2);
      if (deg > 1) {
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        GS(4) = (C2[0] * xy) * g[ch];
        GS(5) = (C2[1] * yz) * g[ch];
        GS(6) = (C2[2] * (2.f * zz - xx - yy)) * g[ch];
        GS(7) = (C2[3] * xz) * g[ch];
        GS(8) = (C2[4] * (xx - yy)) * g[ch];
        dRGBdx[ch] += C2[0] * y * SH(4) + C2[2] * 2.f * -x * SH(6) + C2[3] * z * SH(7) + C2[4] * 2.f * x * SH(8);
        dRGBdy[ch] += C2[0] * x * SH(4) + C2[1] * z * SH(5) + C2[2] * 2.f * -y * SH(6) + C2[4] * 2.f * -y * SH(8);
        dRGBdz[ch] += C2[1] * y * SH(5) + C2[2] * 2.f * 2.f * z * SH(6) + C2[3] * x * SH(7);
        if (deg > 2) {
          GS(9) = (C3[0] * y * (3.f * xx - yy)) * g[ch];
          GS(10) = (C3[1] * xy * z) * g[ch];
          GS(11) = (C3[2] * y * (4.f * zz - xx - yy)) * g[ch];
          GS(12) = (C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * g[ch];
          GS(13) = (C3[4] * x * (4.f * zz - xx - yy)) * g[ch];
          GS(14) = (C3[5] * z * (xx - yy)) * g[ch];
          GS(15) = (C3[6] * x * (xx - 3.f * yy)) * g[ch];
          dRGBdx[ch] += (C3[0] * SH(9) * 3.f * 2.f * xy + C3[1] * SH(10) * yz + C3[2] * SH(11) * -2.f * xy +
                         C3[3] * SH(12) * -3.f * 2.f * xz + C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                         C3[5] * SH(14) * 2.f * xz + C3[6] * SH(15) * 3.f * (xx - yy));
          dRGBdy[ch] += (C3[0] * SH(9) * 3.f * (xx - yy) + C3[1] * SH(10) * xz +
                         C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + C3[3] * SH(12) * -3.f * 2.f * yz +
                         C3[4] * SH(13) * -2.f * xy + C3[5] * SH(14) * -2.f * yz + C3[6] * SH(15) * -3.f * 2.f * xy);
          dRGBdz[ch] += (C3[1] * SH(10) * xy + C3[2] * SH(11) * 4.f * 2.f * yz +
                         C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) + C3[4] * SH(13) * 4.f * 2.f * xz +
                         C3[5] * SH(14) * (xx - yy));
        }
      }
    }
  }
#undef SH
#undef GS
  ddir[0] = dRGBdx[0] * g[0] + dRGBdx[1] * g[1] + dRGBdx[2] * g[2]; /* glm::dot: (a+b)+c */
  ddir[1] = dRGBdy[0] * g[0] + dRGBdy[1] * g[1] + dRGBdy[2] * g[2];
  ddir[2] = dRGBdz[0] * g[0] + dRGBdz[1] * g[1] + dRGBdz[2] * g[2];
  /* dnormvdv, auxiliary.h:91-100 */
  float vx = d0[0], vy = d0[1], vz = d0[2];
  float sum2 = vx * vx + vy * vy + vz * vz;
  float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
  g_mean3D[0] += ((+sum2 - vx * vx) * ddir[0] - vy * vx * ddir[1] - vz * vx * ddir[2]) * inv32;
  g_mean3D[1] += (-vx * vy * ddir[0] + (sum2 - vy * vy) * ddir[1] - vz * vy * ddir[2]) * inv32;
  g_mean3D[2] += (-vx * vz * ddir[0] - vy * vz * ddir[1] + (sum2 - vz * vz) * ddir[2]) * inv32;
}

/* backward.cu:279-366 (computeCov3D backward): dL_dscale, raw dL_dq (no quaternion
 * normalisation Jacobian, :360-365). */
static void cov3d_bwd_one(const float* scale, float mod, const float* q, const float* g6, float* g_scale, float* g_rot) {
  float r = q[0], x = q[1], y = q[2], z = q[3];
  m3 R = m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y), 2.f * (x * y + r * z),
                 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x), 2.f * (x * z - r * y), 2.f * (y * z + r * x),
                 1.f - 2.f * (x * x + y * y));
  float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
  m3 S = m3_cols(s[0], 0, 0, 0, s[1], 0, 0, 0, s[2]);
  m3 Mm = m3_mul(&S, &R);
  m3 dSig = m3_cols(g6[0], 0.5f * g6[1], 0.5f * g6[2], 0.5f * g6[1], g6[3], 0.5f * g6[4], 0.5f * g6[2], 0.5f * g6[4], g6[5]);
  m3 M2; /* 2.0f * M (scalar * matrix, element-wise) */
  for (int c = 0; c < 3; c++)
    for (int rr = 0; rr < 3; rr++) M2.c[c][rr] = 2.0f * Mm.c[c][rr];
  m3 dM = m3_mul(&M2, &dSig);
  m3 Rt = m3_t(&R), dMt = m3_t(&dM);
  for (int k = 0; k < 3; k++)
    g_scale[k] = Rt.c[k][0] * dMt.c[k][0] + Rt.c[k][1] * dMt.c[k][1] + Rt.c[k][2] * dMt.c[k][2];
  for (int k = 0; k < 3; k++)
    for (int rr = 0; rr < 3; rr++) dMt.c[k][rr] *= s[k];
#define D(ci, ri) dMt.c[ci][ri]
  g_rot[0] = 2 * z * (D(0, 1) - D(1, 0)) + 2 * y * (D(2, 0) - D(0, 2)) + 2 * x * (D(1, 2) - D(2, 1));
  g_rot[1] = 2 * y * (D(1, 0) + D(0, 1)) + 2 * z * (D(2, 0) + D(0, 2)) + 2 * r * (D(1, 2) - D(2, 1)) -
             4 * x * (D(2, 2) + D(1, 1));
  g_rot[2] = 2 * x * (D(1, 0) + D(0, 1)) + 2 * r * (D(2, 0) - D(0, 2)) + 2 * z * (D(1, 2) + D(2, 1)) -
             4 * y * (D(2, 2) + D(0, 0));
  g_rot[3] = 2 * r * (D(0, 1) - D(1, 0)) + 2 * x * (D(2, 0) + D(0, 2)) + 2 * y * (D(1, 2) + D(2, 1)) -
             4 * z * (D(1, 1) + D(0, 0));
#undef D
}

/* rasterizer_impl.cu:346-457 (Rasterizer::backward): B1 blend backward, then
 * BACKWARD::preprocess = B2 computeCov2DCUDA + B3 preprocessCUDA (backward.cu:605-673).
 * All nine gradient arrays must be zero on entry (rasterize_points.cu:173-181). */
void gsro_backward(const gsro_frame* f, const float* background, const float* means3D, const float* shs,
                   const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                   const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
                   float tan_fovx, float tan_fovy, const float* dL_dpix, const float* dL_dacc, float* dL_dmean2D,
                   float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                   float* dL_dsh, float* dL_dscale, float* dL_drot) {
  const int P = f->P, W = f->W, H = f->H;
  const float fy = H / (2.0f * tan_fovy), fx = W / (2.0f * tan_fovx);
  (void)colors_precomp;
  (void)cov3D_precomp;
  int par = 0;
#ifdef _OPENMP
  par = omp_get_max_threads() > 1;
#endif
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int ty = 0; ty < f->gy; ty++)
    for (int tx = 0; tx < f->gx; tx++)
      blend_tile_bwd(f, tx, ty, background, dL_dpix, dL_dacc, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, par);

#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++) {
    if (!(f->radii[i] > 0)) continue;
    cov2d_bwd_one(f, i, means3D, fx, fy, tan_fovx, tan_fovy, viewmatrix, dL_dconic, dL_dmean3D, dL_dcov3D);
  }
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++) { /* backward.cu:371-435 */
    if (!(f->radii[i] > 0)) continue;
    const float* m = means3D + 3 * i;
    const float* pr = projmatrix;
    float mh[4];
    xform44(m, pr, mh);
    float m_w = 1.0f / (mh[3] + 0.0000001f);
    float mul1 = (pr[0] * m[0] + pr[4] * m[1] + pr[8] * m[2] + pr[12]) * m_w * m_w;
    float mul2 = (pr[1] * m[0] + pr[5] * m[1] + pr[9] * m[2] + pr[13]) * m_w * m_w;
    float gx2 = dL_dmean2D[3 * i], gy2 = dL_dmean2D[3 * i + 1];
    float dm[3];
    dm[0] = (pr[0] * m_w - pr[3] * mul1) * gx2 + (pr[1] * m_w - pr[3] * mul2) * gy2;
    dm[1] = (pr[4] * m_w - pr[7] * mul1) * gx2 + (pr[5] * m_w - pr[7] * mul2) * gy2;
    dm[2] = (pr[8] * m_w - pr[11] * mul1) * gx2 + (pr[9] * m_w - pr[11] * mul2) * gy2;
    dL_dmean3D[3 * i + 0] += dm[0];
    dL_dmean3D[3 * i + 1] += dm[1];
    dL_dmean3D[3 * i + 2] += dm[2];
    if (shs)
      sh_bwd_one(f->D, f->M, m, campos, shs + (size_t)i * f->M * 3, f->clamped + 3 * i, dL_dcolor + 3 * i,
                 dL_dmean3D + 3 * i, dL_dsh + (size_t)i * f->M * 3);
    if (scales) cov3d_bwd_one(scales + 3 * i, scale_modifier, rotations + 4 * i, dL_dcov3D + 6 * i, dL_dscale + 3 * i, dL_drot + 4 * i);
  }
}

/* rasterizer_impl.cu:52-60,128-135 + auxiliary.h:120-144: present = z_view > 0.2 */
void gsro_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present) {
  for (int i = 0; i < P; i++) {
    float pv[3];
    xform43(means3D + 3 * i, viewmatrix, pv);
    present[i] = !(pv[2] <= 0.2f);
  }
}

/* rasterizer_impl.cu:35-48 (getHigherMsb) */
uint32_t gsro_higher_msb(uint32_t n) {
  uint32_t msb = sizeof(n) * 4, step = msb;
  while (step > 1) {
    step /= 2;
    if (n >> msb) msb += step; else msb -= step;
  }
  if (n >> msb) msb++;
  return msb;
}

/* Test hooks: the small-matrix helpers, exposed so tests/test_oracle.py can compare them bit for bit with the
 * reference's vendored GLM (oracle/ref_glm/check_glm.cpp).  Matrices are 9 floats, [col][row]. */
void gsro_test_m3_mul(const float* a, const float* b, float* o) {
  m3 A, B;
  memcpy(A.c, a, sizeof(A.c));
  memcpy(B.c, b, sizeof(B.c));
  m3 P = m3_mul(&A, &B);
  memcpy(o, P.c, sizeof(P.c));
}
void gsro_test_m3_ttm(const float* a, const float* b, float* o) { /* transpose(A) * transpose(B) * A */
  m3 A, B;
  memcpy(A.c, a, sizeof(A.c));
  memcpy(B.c, b, sizeof(B.c));
  m3 At = m3_t(&A), Bt = m3_t(&B);
  m3 L = m3_mul(&At, &Bt);
  m3 P = m3_mul(&L, &A);
  memcpy(o, P.c, sizeof(P.c));
}
float gsro_test_dot3(const float* u, const float* v) { return u[0] * v[0] + u[1] * v[1] + u[2] * v[2]; }
float gsro_test_dot4(const float* p, const float* q) { return (p[0] * q[0] + p[1] * q[1]) + (p[2] * q[2] + p[3] * q[3]); }
float gsro_test_len3(const float* u) { return sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]); }

/* accessors for the ctypes wrapper (oracle/oracle.py) */
int gsro_num_rendered(const gsro_frame* f) { return f->R; }
#define ACC(name, type) type* gsro_##name(gsro_frame* f) { return f->name; }
ACC(radii, int) ACC(means2D, float) ACC(depths, float) ACC(cov3D, float) ACC(rgb, float) ACC(conic_opacity, float)
ACC(tiles_touched, uint32_t) ACC(point_offsets, uint32_t) ACC(clamped, uint8_t) ACC(keys_unsorted, uint64_t)
ACC(values_unsorted, uint32_t) ACC(keys, uint64_t) ACC(point_list, uint32_t) ACC(ranges, uint32_t)
ACC(final_T, float) ACC(n_contrib, uint32_t) ACC(out_color, float) ACC(out_depth, float) ACC(out_acc, float)
ACC(fragile, uint8_t)
int gsro_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void gsro_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}
