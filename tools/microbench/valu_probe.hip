// valu_probe.hip -- what does one VALU instruction cost on this chip, per SIMD, at 1..8 waves per SIMD?
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/valu_probe.hip -o gpurun_out/valu_probe && gpurun_out/valu_probe
// Every wave runs ITER x 64 instructions of one kind on eight independent accumulators; the grid is 256 CUs x W
// workgroups of 256 threads (one wave per SIMD each).  Reported: cycles per wave-instruction per SIMD, from the shader
// clock (s_memtime) and from hipEvents at an assumed 2.4 GHz.  A measurement tool: nothing in the product uses it.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

enum Kind { FMA, PK_FMA, PK_MUL, MUL_SGPR, CNDMASK_VCC, CNDMASK_SGPR, EXP, RCP, DPP_ADD, MAX3, FMA_DEP, FMAC, MOV, CMP_VCC, CMP_SGPR,
            PERMLANE32_SWAP, MIN, LDS_B128_BCAST, LDS_B32, N_KIND };
static const char* kind_name[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_mul_f32 (sgpr src)", "v_cndmask_b32 (vcc)",
                                 "v_cndmask_b32_e64 (sgpr mask)", "v_exp_f32", "v_rcp_f32", "v_add_f32 dpp row_shr:1",
                                 "v_max3_f32", "v_fma_f32 (one dependent chain)", "v_fmac_f32 (VOP2)", "v_mov_b32", "v_cmp_lt_f32 (vcc)",
                                 "v_cmp_lt_f32_e64 (sgpr pair)", "v_permlane32_swap_b32", "v_min_f32", "ds_read_b128 (one address)",
                                 "ds_read_b32 (lane-linear)"};

template <int KIND>
__global__ __launch_bounds__(256) void k_probe(float* out, unsigned long long* cycles, int iters, float seed, unsigned long long mask) {
  __shared__ float lds[4096];
  if (KIND == LDS_B128_BCAST || KIND == LDS_B32) { for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed; __syncthreads(); }
  float a[8];
  f2 p[8];
  for (int i = 0; i < 8; i++) {
    a[i] = seed + threadIdx.x * 1e-3f + i;
    p[i] = f2{a[i], a[i] + 0.5f};
  }
  const float b = 1.0000001f, c = 1e-9f;
  const f2 pb = f2{b, b}, pc = f2{c, c};
  const float sb = __builtin_amdgcn_readfirstlane(seed * 0.0f + 1.0000001f);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (KIND == FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP64(X)
#undef X
    } else if (KIND == PK_FMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
      REP64(X)
#undef X
    } else if (KIND == PK_MUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
      REP64(X)
#undef X
    } else if (KIND == MUL_SGPR) {
#define X(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sb));
      REP64(X)
#undef X
    } else if (KIND == CNDMASK_VCC) {
      asm volatile("s_mov_b64 vcc, %0" ::"s"(mask) : "vcc");
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
      REP64(X)
#undef X
    } else if (KIND == CNDMASK_SGPR) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(mask));
      REP64(X)
#undef X
    } else if (KIND == EXP) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      REP64(X)
#undef X
    } else if (KIND == RCP) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      REP64(X)
#undef X
    } else if (KIND == DPP_ADD) {
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      REP64(X)
#undef X
    } else if (KIND == MAX3) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP64(X)
#undef X
    } else if (KIND == FMAC) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP64(X)
#undef X
    } else if (KIND == MOV) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
      REP64(X)
#undef X
    } else if (KIND == CMP_VCC) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
      REP64(X)
#undef X
    } else if (KIND == CMP_SGPR) {
      unsigned long long m[8];
#define X(i) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m[i]) : "v"(a[i]), "v"(b));
      REP64(X)
#undef X
      if (it == iters - 1) a[0] += (float)(m[0] ^ m[1] ^ m[2] ^ m[3] ^ m[4] ^ m[5] ^ m[6] ^ m[7]);
    } else if (KIND == PERMLANE32_SWAP) {
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 4) & 7]));
      REP64(X)
#undef X
    } else if (KIND == MIN) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP64(X)
#undef X
    } else if (KIND == LDS_B128_BCAST) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 q[8];
      const unsigned addr = (unsigned)(it & 7) * 16u;
#define X(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i]) : "v"(addr), "n"(i * 64));
      REP64(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)");
      if (it == iters - 1) for (int i = 0; i < 8; i++) a[i] += q[i].x + q[i].w;
    } else if (KIND == LDS_B32) {
      float q[8];
      const unsigned addr = (threadIdx.x & 63) * 4u;
#define X(i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(q[i]) : "v"(addr), "n"(i * 256));
      REP64(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)");
      if (it == iters - 1) for (int i = 0; i < 8; i++) a[i] += q[i];
    } else if (KIND == FMA_DEP) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
      REP64(X)
#undef X
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(int W, int iters, float* out, unsigned long long* cyc, std::vector<unsigned long long>& h) {
  const int blocks = 256 * W;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k_probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 16, 1.0f, 0x5555555555555555ull);  // warm-up
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f, 0x5555555555555555ull);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < blocks * 4; i++) mean += (double)h[i];
  mean /= blocks * 4;
  const double n = (double)iters * 64;  // instructions per wave
  // W waves share a SIMD: SIMD cycles per wave-instruction = wave's cycles / (W * n)
  printf("  %-34s W=%d  %6.2f cyc/instr/SIMD by s_memtime (wave: %6.2f per instr)   %6.2f by events @2.4GHz  (%.3f ms)\n",
         kind_name[KIND], W, mean / (W * n), mean / n, ms * 1e-3 * 2.4e9 / (W * n), ms);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * 256 * 8 * 256);
  hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 8 * 4);
  std::vector<unsigned long long> h(256 * 8 * 4);
  const int iters = 2000;
  for (int W : {1, 2, 5, 8}) {
    printf("W = %d waves per SIMD\n", W);
    run<FMA>(W, iters, out, cyc, h);
    run<FMA_DEP>(W, iters, out, cyc, h);
    run<PK_FMA>(W, iters, out, cyc, h);
    run<PK_MUL>(W, iters, out, cyc, h);
    run<MUL_SGPR>(W, iters, out, cyc, h);
    run<CNDMASK_VCC>(W, iters, out, cyc, h);
    run<CNDMASK_SGPR>(W, iters, out, cyc, h);
    run<EXP>(W, iters, out, cyc, h);
    run<RCP>(W, iters, out, cyc, h);
    run<DPP_ADD>(W, iters, out, cyc, h);
    run<MAX3>(W, iters, out, cyc, h);
    run<FMAC>(W, iters, out, cyc, h);
    run<MOV>(W, iters, out, cyc, h);
    run<MIN>(W, iters, out, cyc, h);
    run<CMP_VCC>(W, iters, out, cyc, h);
    run<CMP_SGPR>(W, iters, out, cyc, h);
    run<PERMLANE32_SWAP>(W, iters, out, cyc, h);
    run<LDS_B128_BCAST>(W, iters, out, cyc, h);
    run<LDS_B32>(W, iters, out, cyc, h);
  }
  return 0;
}
