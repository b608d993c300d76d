// Split-f16 MFMA building blocks shared by encode.hip, fine.hip and pointnet.hip (gfx950 only).
//
// An f32 value a is used as hi + lo with hi = f16(a) (RNE), lo = f16(a - hi): 22 significand bits (gfx950's MFMA honours
// f16 denormals, so small values keep their precision: |a - hi - lo| <= max(2^-22 |a|, 2^-25)). A product is
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with f32 accumulation: ~5e-7 relative at 3 MFMAs of 32 cycles per
// 32x32x16 block, against 8 MFMAs of 64 cycles for v_mfma_f32_32x32x2_f32. f16 overflows at 65504: callers bound every
// activation that enters a split product (statically from the weights, or with a run-time guard on raw inputs) and keep an
// all-f32 path for the rest.
//
// Weight packing: [n_tile(32 rows)][K/16 steps][64 lanes][hi 16 B | lo 16 B]; lane (i, kh) of step s holds
// W[tile*32 + i][kh*K/2 + 8 s .. +7] (the "half-split" k order of the f32 packing, 8 values per step).
// Activations: a lane's LDS row half, 8 consecutive floats per step, split on the fly (20 VALU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <vector>

namespace t2l {

typedef float h3_f32x16 __attribute__((ext_vector_type(16)));
typedef float h3_f32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h3_f16x8 __attribute__((ext_vector_type(8)));

constexpr float kSplitF16Safe = 3.0e4f;  // bound below which an operand may enter a split-f16 product

struct HFrag {
  h3_f16x8 hi, lo;
};
// SINGLE: the plain-f16 variant (one product per pair of operands, |error| <= 2^-10 relative per product): no low part
template <bool SINGLE = false>
__device__ __forceinline__ HFrag split_h(const float* __restrict__ p) {  // 8 consecutive floats (16-byte aligned)
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  const h3_f32x8 v = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  HFrag f;
  f.hi = __builtin_convertvector(v, h3_f16x8);
  if constexpr (SINGLE) f.lo = f.hi;  // (never read)
  else f.lo = __builtin_convertvector(v - __builtin_convertvector(f.hi, h3_f32x8), h3_f16x8);
  return f;
}
__device__ __forceinline__ HFrag load_h(const uint4* __restrict__ wp) {
  // packed weights live in global memory: say so (a pointer that went through an opaque asm would otherwise load as flat_*)
  const __attribute__((address_space(1))) uint4* gp = (const __attribute__((address_space(1))) uint4*)wp;
  HFrag f;
  f.hi = __builtin_bit_cast(h3_f16x8, gp[0]);
  f.lo = __builtin_bit_cast(h3_f16x8, gp[1]);
  return f;
}
// acc += A * B with A, B split fragments (a = A operand, b = B operand of the MFMA)
template <bool SINGLE = false>
__device__ __forceinline__ void mfma_h3(h3_f32x16& acc, const HFrag& a, const HFrag& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.hi, acc, 0, 0, 0);
  if constexpr (!SINGLE) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo, b.hi, acc, 0, 0, 0);
  }
}
template <bool SINGLE = false>
__device__ __forceinline__ HFrag load_h1(const uint4* __restrict__ wp) {  // SINGLE: the high half only (half the weight bytes)
  const __attribute__((address_space(1))) uint4* gp = (const __attribute__((address_space(1))) uint4*)wp;
  HFrag f;
  f.hi = __builtin_bit_cast(h3_f16x8, gp[0]);
  if constexpr (SINGLE) f.lo = f.hi;
  else f.lo = __builtin_bit_cast(h3_f16x8, gp[1]);
  return f;
}
// acc += A[32 x 16*STEPS] (this lane's LDS row half) * W^T (one packed weight tile, offset to its first step and to this
// lane: 2 uint4 per lane and step, 128 uint4 per step)
// dev experiment (make exp_fine / exp_enc EXPFLAG=-DT2L_EXP_HOTW; WRONG results, timing only): every weight fragment of a tile pass comes
// from the pass's first two k-steps — the instruction stream stays, the packed-weight stream out of the L2 disappears
#ifdef T2L_EXP_HOTW
#define T2L_HOT(s) ((s) & 1)
#else
#define T2L_HOT(s) (s)
#endif
#ifndef T2L_DOT_RING
#define T2L_DOT_RING 2
#endif
constexpr int kDotRing = T2L_DOT_RING;
template <int STEPS, bool SINGLE = false>
__device__ __forceinline__ void mm32_dot_h(const float* __restrict__ arow, const uint4* __restrict__ wp, h3_f32x16& acc) {
  // the weight fragments through a register ring kDotRing steps deep, each request issued before the MFMAs of the step that frees its
  // slot and pinned there (the compiler otherwise sinks every load to its use; encode.hip: stream_weights). fine_match, 40,960 pairs:
  // no ring 5.90 ms, depth 2 / 4 / 8: 5.66 / 5.70 / 5.84 ms. Same arithmetic in the same order: bit-identical results.
  constexpr int D = STEPS < kDotRing ? STEPS : kDotRing;
  HFrag ring[D];
#pragma unroll
  for (int i = 0; i < D; ++i) ring[i] = load_h1<SINGLE>(wp + T2L_HOT(i) * 128);
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const HFrag wf = ring[s % D];
    if (s + D < STEPS) ring[s % D] = load_h1<SINGLE>(wp + T2L_HOT(s + D) * 128);
    __builtin_amdgcn_sched_barrier(0);
    mfma_h3<SINGLE>(acc, split_h<SINGLE>(arow + 8 * s), wf);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// W [rows][cin] row-major (optionally [W | bias column | 0] of width kp, as pack_half_split) -> split-f16 fragments; bit
// patterns carried in a float vector (8 floats per lane and step)
static inline std::vector<float> pack_split_f16(const float* W, const float* bias, int rows, int cin, int kp) {
  std::vector<float> p((size_t)rows * kp, 0.f);
  const int steps = kp / 16;
  uint16_t* out = reinterpret_cast<uint16_t*>(p.data());
  for (int nt = 0; nt < rows / 32; ++nt)
    for (int st = 0; st < steps; ++st)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int row = nt * 32 + (lane & 31), k = (lane >> 5) * (kp / 2) + 8 * st + e;
          float w = 0.f;
          if (k < cin) w = W[(size_t)row * cin + k];
          else if (bias && k == cin) w = bias[row];
          const _Float16 hi = (_Float16)w;
          const _Float16 lo = (_Float16)(w - (float)hi);
          const size_t base = (((size_t)nt * steps + st) * 64 + lane) * 16;
          memcpy(out + base + e, &hi, 2);
          memcpy(out + base + 8 + e, &lo, 2);
        }
  return p;
}
static inline float h3_max_abs(const float* v, size_t n) {
  float m = 0.f;
  for (size_t i = 0; i < n; ++i) m = fmaxf(m, fabsf(v[i]));
  return m;
}
static inline float h3_max_row_norm(const float* W, int rows, int cols) {
  double m = 0.0;
  for (int r = 0; r < rows; ++r) {
    double ss = 0.0;
    for (int c = 0; c < cols; ++c) ss += (double)W[(size_t)r * cols + c] * W[(size_t)r * cols + c];
    m = fmax(m, sqrt(ss));
  }
  return (float)m;
}

}  // namespace t2l
