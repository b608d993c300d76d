// 32-row f32 MFMA building blocks shared by pointnet.hip and fine.hip (gfx950 only): the "half-split" weight packing and the
// matching dot-product loop with an explicit 4-deep prefetch ring on the weight stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "gemm_f32.h"

namespace t2l {

using train::f32x16;

// [rows/32][kp/8][64 lanes] float4: lane (i, kh) holds Wf[tile*32+i][kh*kp/2 + 4*s4 + 0..3]; Wf = [W | bias column | 0] of width kp
static inline std::vector<float> pack_half_split(const std::vector<float>& W, const std::vector<float>* bias, int rows, int cin, int kp) {
  std::vector<float> out((size_t)rows * kp, 0.f);
  const int half = kp / 2;
  for (int t = 0; t < rows / 32; ++t)
    for (int s4 = 0; s4 < kp / 8; ++s4)
      for (int lane = 0; lane < 64; ++lane)
        for (int c = 0; c < 4; ++c) {
          const int row = t * 32 + (lane & 31), k = (lane >> 5) * half + 4 * s4 + c;
          float v = 0.f;
          if (k < cin) v = W[(size_t)row * cin + k];
          else if (bias && k == cin) v = (*bias)[row];
          out[(((size_t)t * (kp / 8) + s4) * 64 + lane) * 4 + c] = v;
        }
  return out;
}

// acc += A[32 x 8*S4] (this lane's row half, LDS, float4 per 4 k-steps) * B (packed weights, global: wp[s4*64], lane folded in),
// with 4 weight loads in flight (one wave per SIMD: the L2 latency has to be hidden inside the wave)
template <int S4>
__device__ __forceinline__ void mm32_dot(const float* __restrict__ ar, const float4* __restrict__ wp, f32x16& acc) {
  float4 b0 = wp[0], b1 = wp[64 * (1 < S4 ? 1 : 0)], b2 = wp[64 * (2 < S4 ? 2 : 0)], b3 = wp[64 * (3 < S4 ? 3 : 0)];
#define GA_STEP(B, S)                                                                         \
  {                                                                                           \
    const float4 a = *reinterpret_cast<const float4*>(ar + 4 * (S));                          \
    const float4 w = B;                                                                       \
    if ((S) + 4 < S4) B = wp[64 * ((S) + 4)];                                                 \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w.x, acc, 0, 0, 0);                       \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w.y, acc, 0, 0, 0);                       \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w.z, acc, 0, 0, 0);                       \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w.w, acc, 0, 0, 0);                       \
  }
  int s = 0;
  for (; s + 4 <= S4; s += 4) {
    GA_STEP(b0, s) GA_STEP(b1, s + 1) GA_STEP(b2, s + 2) GA_STEP(b3, s + 3)
  }
  if (s < S4) GA_STEP(b0, s)
  if (s + 1 < S4) GA_STEP(b1, s + 1)
  if (s + 2 < S4) GA_STEP(b2, s + 2)
#undef GA_STEP
}


}  // namespace t2l
