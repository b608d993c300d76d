// Micro-probe (dev tool, not product): how much VALU work hides behind MFMAs on gfx950?
//   mode 0: one wave per SIMD, K independent med3 between consecutive bf16 32x32x16 MFMAs (same wave)
//   mode 1: two waves per SIMD: wave w<4 MFMA only, wave w>=4 VALU only (K med3 per "MFMA slot"), no sync
//   mode 2: as mode 0 with the f32 32x32x2 MFMA
//   mode 3: as mode 1 but MFMA = even waves, VALU = odd waves
// Prints cycles per MFMA (s_memtime based) for K in {0,2,4,6,8,12}.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K, int MODE>
__global__ __launch_bounds__(512, 2) void probe(float* out, long long* cyc, int iters) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: role branches are real branches
  f32x16 acc0 = {0}, acc1 = {0};
  float l[16];
  for (int i = 0; i < 16; ++i) l[i] = (float)(threadIdx.x + i);
  uint4 a = make_uint4(threadIdx.x, 2, 3, 4), b = make_uint4(5, 6, 7, threadIdx.x);
  float x = (float)threadIdx.x * 0.5f;
  const bool do_mfma = (MODE == 1) ? (wave < 4) : (MODE == 3) ? !(wave & 1) : true;
  const bool do_valu = (MODE == 1) ? (wave >= 4) : (MODE == 3) ? (wave & 1) : true;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (do_mfma) {
        if (MODE == 2) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, l[0], (m & 1) ? acc1 : acc0, 0, 0, 0);
          if (m & 1) { acc1 = acc0; }
        } else {
          if ((m & 1) && MODE != 4) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc1, 0, 0, 0);
          else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc0, 0, 0, 0);
        }
      }
      if (do_valu) {
        if (MODE == 4 || MODE == 5) {  // the kernel's pattern: s'[i] = med3(s[i-1], s[i], x), i descending: independent ops
#pragma unroll
          for (int k = 0; k < K; ++k) l[15 - (k % 15)] = __builtin_amdgcn_fmed3f(l[14 - (k % 15)], l[15 - (k % 15)], x);
        } else {
#pragma unroll
          for (int k = 0; k < K; ++k) l[(k + 1) & 15] = __builtin_amdgcn_fmed3f(l[k & 15], l[(k + 1) & 15], x);
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += l[i] + acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  if (threadIdx.x == 256 && blockIdx.x == 0) cyc[1] = t1 - t0;
  if (threadIdx.x == 64 && blockIdx.x == 0) cyc[2] = t1 - t0;
}

template <int K, int MODE>
void run(const char* name, int threads) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&cyc, 32); hipMemset(cyc, 0, 32);
  const int iters = 2000;
  hipLaunchKernelGGL((probe<K, MODE>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL((probe<K, MODE>), dim3(256), dim3(threads), 0, 0, out, cyc, iters); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[3]; hipMemcpy(h, cyc, 24, hipMemcpyDeviceToHost);
  printf("%s K=%2d: %.1f us, cycles/slot wave0 %.1f wave1 %.1f wave4 %.1f\n", name, K, ms * 1e3, (double)h[0] / (iters * 16.0), (double)h[2] / (iters * 16.0), (double)h[1] / (iters * 16.0));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0, 0>("same-wave bf16", 256); run<2, 0>("same-wave bf16", 256); run<4, 0>("same-wave bf16", 256);
  run<6, 0>("same-wave bf16", 256); run<8, 0>("same-wave bf16", 256); run<12, 0>("same-wave bf16", 256);
  run<0, 1>("two-wave  bf16", 512); run<2, 1>("two-wave  bf16", 512); run<4, 1>("two-wave  bf16", 512);
  run<6, 1>("two-wave  bf16", 512); run<8, 1>("two-wave  bf16", 512); run<12, 1>("two-wave  bf16", 512);
  run<0, 4>("1chain list   ", 256); run<4, 4>("1chain list   ", 256); run<6, 4>("1chain list   ", 256); run<8, 4>("1chain list   ", 256);
  run<0, 5>("2chain list   ", 256); run<4, 5>("2chain list   ", 256); run<6, 5>("2chain list   ", 256); run<8, 5>("2chain list   ", 256);
  run<0, 3>("even/odd  bf16", 512); run<4, 3>("even/odd  bf16", 512); run<8, 3>("even/odd  bf16", 512); run<12, 3>("even/odd  bf16", 512);
  run<0, 2>("same-wave f32 ", 256); run<4, 2>("same-wave f32 ", 256); run<8, 2>("same-wave f32 ", 256); run<12, 2>("same-wave f32 ", 256);
  return 0;
}
