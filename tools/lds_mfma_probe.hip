// Micro-probe (dev tool): bf16 MFMA chain (3 per k-step) + 2 ds_read_b128 per k-step (prefetch depth 4) + K list-style med3
// per MFMA, 16 k-steps per "tile", optional barrier per tile. Prints cycles per tile.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K, int S>
__device__ __forceinline__ void steps(const char* tb, f32x16& acc, float (&l)[16], uint4 (&ah)[4], uint4 (&al)[4], uint4 b, float x) {
  if constexpr (S < 16) {
    const uint4 a_hi = ah[S & 3], a_lo = al[S & 3];
    if constexpr (S + 4 < 16) {
      ah[S & 3] = *reinterpret_cast<const uint4*>(tb + 16 * (S + 4));
      al[S & 3] = *reinterpret_cast<const uint4*>(tb + 512 + 16 * (S + 4));
    }
    const bf16x8 vh = __builtin_bit_cast(bf16x8, a_hi), vl = __builtin_bit_cast(bf16x8, a_lo), vb = __builtin_bit_cast(bf16x8, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, vb, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, vb, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, vb, acc, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 3 * K; ++k) l[15 - (k % 15)] = __builtin_amdgcn_fmed3f(l[14 - (k % 15)], l[15 - (k % 15)], x);
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
    }
    if constexpr (S + 4 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    steps<K, S + 1>(tb, acc, l, ah, al, b, x);
  }
}

template <int K, int LDS, int BAR>
__global__ __launch_bounds__(256, 2) void probe(float* out, long long* cyc, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2 * 33280 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = (float)i;
  __syncthreads();
  f32x16 acc = {0};
  float l[16];
  for (int i = 0; i < 16; ++i) l[i] = (float)(threadIdx.x + i);
  uint4 b = make_uint4(5, 6, 7, threadIdx.x);
  float x = (float)threadIdx.x * 0.5f;
  const char* tb0 = smem + (lane & 31) * 1040 + (lane >> 5) * 256;
  long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < tiles; ++t) {
    if (BAR) __syncthreads();
    const char* tb = tb0 + (t & 1) * 33280;
    uint4 ah[4], al[4];
    if (LDS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { ah[i] = *reinterpret_cast<const uint4*>(tb + 16 * i); al[i] = *reinterpret_cast<const uint4*>(tb + 512 + 16 * i); }
      steps<K, 0>(tb, acc, l, ah, al, b, x);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) { ah[i] = make_uint4(t, i, 3, lane); al[i] = make_uint4(i, t, lane, 4); }
      steps<K, 12>(tb, acc, l, ah, al, b, x); steps<K, 12>(tb, acc, l, ah, al, b, x); steps<K, 12>(tb, acc, l, ah, al, b, x); steps<K, 12>(tb, acc, l, ah, al, b, x);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += l[i] + acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int K, int LDS, int BAR>
void run(const char* name, int blocks) {
  float* out; long long* cyc;
  hipMalloc(&out, 1024 * 256 * sizeof(float)); hipMalloc(&cyc, 16);
  const int tiles = 400;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<K, LDS, BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 66560);
  hipLaunchKernelGGL((probe<K, LDS, BAR>), dim3(blocks), dim3(256), 66560, 0, out, cyc, tiles);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%s blocks=%d K=%d lds=%d bar=%d: %.0f cycles/tile\n", name, blocks, K, LDS, BAR, (double)h / tiles);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0, 0, 0>("mfma only      ", 256); run<6, 0, 0>("mfma+valu      ", 256);
  run<0, 1, 0>("mfma+lds       ", 256); run<6, 1, 0>("mfma+valu+lds  ", 256); run<6, 1, 1>("..+barrier     ", 256);
  run<0, 1, 0>("mfma+lds       ", 512); run<6, 1, 0>("mfma+valu+lds  ", 512); run<6, 1, 1>("..+barrier     ", 512);
  return 0;
}
