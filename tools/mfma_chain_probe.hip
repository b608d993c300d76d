// Micro-probe (dev tool, not product): issue rate of v_mfma_f32_32x32x16_f16 on gfx950 with ONE wave per SIMD as a function of the number
// of independent accumulator chains (1, 2, 4) and of where the B operand comes from (registers / LDS read per MFMA).
// Prints cycles per MFMA (wall clock of a 256-CU-filling launch x 2.4 GHz nominal; compare the rows, not the absolute value).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int CH, bool LDS>
__global__ __launch_bounds__(256, 1) void probe(float* out, int iters) {
  __shared__ uint4 buf[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = make_uint4(i, 1, 2, 3);
  __syncthreads();
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  f16x8 a = {1, 2, 3, 4, 5, 6, 7, (_Float16)(threadIdx.x & 7)}, b = {1, 1, 2, 2, 3, 3, 4, (_Float16)(threadIdx.x & 3)};
  const uint4* lp = buf + (threadIdx.x & 63);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 48; ++m) {
      if constexpr (LDS) b = __builtin_bit_cast(f16x8, lp[((m + it) & 63) * 64]);
      acc[m % CH] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % CH], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int c = 0; c < CH; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CH, bool LDS>
static void run(const char* name, float* out) {
  const int iters = 2000, grid = 256 * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<CH, LDS>), dim3(grid), dim3(256), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<CH, LDS>), dim3(grid), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)grid / 256.0 * iters * 48;  // one workgroup per CU at a time, 4 rounds
  printf("%-28s %.3f ms  %.1f cycles per MFMA at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / mfma_per_simd);
}

int main() {
  float* out;
  hipMalloc(&out, 1024 * 256 * 4);
  run<1, false>("1 chain, regs", out);
  run<2, false>("2 chains, regs", out);
  run<4, false>("4 chains, regs", out);
  run<1, true>("1 chain, LDS read per MFMA", out);
  run<2, true>("2 chains, LDS read per MFMA", out);
  run<4, true>("4 chains, LDS read per MFMA", out);
  return 0;
}
