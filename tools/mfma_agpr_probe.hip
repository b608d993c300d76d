// Micro-probe (dev tool, not product): cost of VALU fillers between inline-asm bf16 32x32x16 MFMAs on gfx950 with
// one wave per SIMD, by operand placement. BREG: 0 = B operand in VGPRs, 1 = in AGPRs. CREG: accumulators in VGPRs / AGPRs.
// K med3 fillers (the key-list insertion pattern) after every MFMA, order pinned by sched_barrier(0).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int BREG, int CREG>
__device__ __forceinline__ void mf(f32x16& acc, const u32x4& a, const u32x4& b) {
  if constexpr (BREG == 0 && CREG == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  if constexpr (BREG == 1 && CREG == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
  if constexpr (BREG == 0 && CREG == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  if constexpr (BREG == 1 && CREG == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(b));
}

template <int K, int BREG, int CREG, int NB>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* cyc, int iters, const u32x4* src) {
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
  if constexpr (CREG) { asm volatile("" : "+a"(acc0)); asm volatile("" : "+a"(acc1)); }
  float l[16];
  for (int i = 0; i < 16; ++i) l[i] = (float)(threadIdx.x + i);
  u32x4 b[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    b[i] = src[threadIdx.x + 256 * i];
    if constexpr (BREG) asm volatile("" : "=a"(b[i]) : "0"(b[i]));
    else asm volatile("" : "+v"(b[i]));
  }
  u32x4 a = src[threadIdx.x + 7];
  float x = (float)threadIdx.x * 0.5f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NB; ++m) {
      __builtin_amdgcn_sched_barrier(0);
      if (m & 1) mf<BREG, CREG>(acc1, a, b[m]); else mf<BREG, CREG>(acc0, a, b[m]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < K; ++k) l[15 - ((k + 3 * m) % 15)] = __builtin_amdgcn_fmed3f(l[14 - ((k + 3 * m) % 15)], l[15 - ((k + 3 * m) % 15)], x);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  if constexpr (CREG) { asm volatile("" : "+a"(acc0)); asm volatile("" : "+a"(acc1)); }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += l[i] + acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int K, int BREG, int CREG, int NB>
void run() {
  float* out; long long* cyc; u32x4* src;
  hipMalloc(&out, 256 * 256 * sizeof(float)); hipMalloc(&cyc, 32); hipMemset(cyc, 0, 32);
  hipMalloc(&src, 256 * 80 * 16); hipMemset(src, 0x3c, 256 * 80 * 16);
  const int iters = 4000 * 16 / NB;
  hipLaunchKernelGGL((probe<K, BREG, CREG, NB>), dim3(256), dim3(256), 0, 0, out, cyc, iters, src);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL((probe<K, BREG, CREG, NB>), dim3(256), dim3(256), 0, 0, out, cyc, iters, src); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("B in %s, acc in %s, %2d B regs, K=%d: %7.1f us, %.1f cycles/MFMA (memtime), %.2f ns/MFMA wall\n", BREG ? "AGPR" : "VGPR", CREG ? "AGPR" : "VGPR",
         NB, K, ms * 1e3, (double)h / (iters * (double)NB), ms * 1e6 / (iters * (double)NB));
  hipFree(out); hipFree(cyc); hipFree(src);
}

int main() {
  run<0, 0, 0, 16>(); run<3, 0, 0, 16>(); run<5, 0, 0, 16>();
  run<0, 1, 0, 16>(); run<3, 1, 0, 16>(); run<5, 1, 0, 16>();
  run<0, 0, 1, 16>(); run<3, 0, 1, 16>(); run<5, 0, 1, 16>();
  run<0, 1, 1, 16>(); run<3, 1, 1, 16>(); run<5, 1, 1, 16>();
  run<0, 1, 0, 64>(); run<3, 1, 0, 64>(); run<3, 1, 1, 64>();
  return 0;
}
