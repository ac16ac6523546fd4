// valu_rate.hip -- how many SIMD cycles does one wave64 f32 VALU instruction cost on gfx950, as a function of the
// waves resident per SIMD?  (MI355X_MICROARCH.md quotes 2 cycles on the SIMD-32 and 4 for a wave alone; the blend
// kernels' launch time works out to ~4.2 cycles per VALU instruction at 5 waves per SIMD.)
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
// Every thread runs ITER x 16 independent v_fma_f32 (8 accumulators x 2), or the same count of v_add_f32_dpp /
// v_exp_f32; grid = 256 CUs x 4 SIMDs x W waves, one workgroup of 64 threads per wave so W is exact.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ __launch_bounds__(64) void k_rate(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      if (KIND == 0) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
      } else if (KIND == 1) {
        asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                     "v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                     "v_add_f32_dpp %4, %4, %4 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                     "v_add_f32_dpp %6, %6, %6 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_ror:4 row_mask:0xf bank_mask:0xf"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      } else if (KIND == 2) {
        asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                     "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
      } else if (KIND == 4) {
        asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n"
                     "v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n"
                     "v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      } else if (KIND == 5) {
        asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n"
                     "v_cndmask_b32_e64 %3, %3, %8, %9\n v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n"
                     "v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                     : "v"(a), "s"(0x5555555555555555ull));
      } else if (KIND == 6) {
        asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n"
                     "v_permlane16_swap_b32 %6, %7\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n"
                     "v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      } else {
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                     "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      }
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int KIND>
static void run(const char* name, float* out) {
  const int iters = 4096;
  for (int W = 1; W <= 8; W *= 2) {
    const int grid = 256 * 4 * W;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(64), 0, 0, out, 16, 1.0001f, 0.5f);  // warm-up
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 16 * W;           // wave-instructions issued on one SIMD
    const double cycles = ms * 1e-3 * 2.4e9;                        // at the 2.4 GHz the guide quotes
    printf("%-16s %d wave(s)/SIMD: %.3f ms  -> %.2f cycles per wave-instruction\n", name, W, ms, cycles / instr_per_simd);
  }
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float));
  run<0>("v_fma_f32", out);
  run<2>("v_mul_f32", out);
  run<1>("v_add_f32_dpp", out);
  run<3>("v_exp_f32", out);
  run<4>("permlane32_swap", out);
  run<6>("permlane16_swap", out);
  run<5>("v_cndmask_b32", out);
  (void)hipFree(out);
  return 0;
}
