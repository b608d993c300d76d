// Micro-probe (dev tool, not product): does a SECOND wave on the same SIMD hide the issue cost of the first wave's MFMAs?
// Every wave runs the scan's inner pattern: one f16 32x32x16 MFMA (B operand in AGPRs, two alternating accumulators) followed
// by K med3 fillers; WAVES = 4 (one per SIMD) or 8 (two per SIMD). Prints cycles per MFMA per WAVE and per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int K, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void probe(float* out, long long* cyc, int iters, const u32x4* src) {
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
  float l[16];
  for (int i = 0; i < 16; ++i) l[i] = (float)(threadIdx.x + i);
  u32x4 b[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    b[i] = src[threadIdx.x + 256 * i];
    asm volatile("" : "=a"(b[i]) : "0"(b[i]));
  }
  u32x4 a = src[threadIdx.x + 7];
  float x = (float)threadIdx.x * 0.5f;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      __builtin_amdgcn_sched_barrier(0);
      if (m & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "a"(b[m]));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "a"(b[m]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < K; ++k) l[15 - ((k + 3 * m) % 15)] = __builtin_amdgcn_fmed3f(l[14 - ((k + 3 * m) % 15)], l[15 - ((k + 3 * m) % 15)], x);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += l[i] + acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int K, int WAVES>
void run() {
  float* out; long long* cyc; u32x4* src;
  hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&cyc, 32); hipMemset(cyc, 0, 32);
  hipMalloc(&src, 512 * 80 * 16);
  {  // RANDOM f16 operands in [-1, 1): dense random data draws far more matrix-pipe power than a constant fill
    unsigned short* h = (unsigned short*)malloc(512 * 80 * 16);
    unsigned x = 12345u;
    for (int i = 0; i < 512 * 80 * 8; ++i) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(((x >> 16) & 0x83ff) | 0x3800 | ((x >> 3) & 0x0400)); }
    hipMemcpy(src, h, 512 * 80 * 16, hipMemcpyHostToDevice); free(h);
  }
  const int iters = 4000;
  hipLaunchKernelGGL((probe<K, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, out, cyc, iters, src);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<K, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, out, cyc, iters, src);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("  wall %.1f us -> %.1f TFLOP/s; ", ms * 1e3, 256.0 * WAVES * iters * 16 * 32768.0 / (ms * 1e-3) / 1e12);
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double per_wave = (double)h / (iters * 16.0);
  printf("K=%2d waves/SIMD=%d: %6.1f cycles per MFMA per wave, %6.1f per SIMD-MFMA\n", K, WAVES / 4, per_wave, per_wave / (WAVES / 4));
  hipFree(out); hipFree(cyc); hipFree(src);
}

int main() {
  run<0, 4>(); run<0, 8>();
  run<5, 4>(); run<5, 8>();
  run<6, 4>(); run<6, 8>();
  run<7, 4>(); run<7, 8>();
  run<9, 4>(); run<9, 8>();
  run<10, 4>(); run<10, 8>();
  return 0;
}
