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
// v_fma_mix_f32 reads f16 operands out of either half of a packed register, so the two f32 <-> (hi, lo) conversions that every
// split epilogue and every residual read-back is made of need no unpacking v_cvt_f32_f16 (this toolchain never selects the
// instruction for these patterns). Both results are the exact f32 values the cvt + sub / cvt + cvt + add forms produce.
//   h3_minus_half<HI>(x, p):  x - f16 half HI of p      (the low part of a split: x - hi is exact in f32)
//   h3_sum_halves<HI>(h, l):  f16 half HI of h + f16 half HI of l  (hi + lo back to f32: one rounding, as the f32 add)
// Inline asm: the compiler pads no MFMA hazard for these reads. x / p always come out of a compiler-visible VALU op on the
// same registers (the v_cvt_pk that made p from x) or out of a load; tests/test_mfma_hazard_scan.py checks every kernel.
template <int HI>
__device__ __forceinline__ float h3_minus_half(float x, unsigned p) {
  float d;
  if constexpr (HI) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(p), "v"(x));
  else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(p), "v"(x));
  return d;
}
template <int HI>
__device__ __forceinline__ float h3_sum_halves(unsigned h, unsigned l) {
  float d;
  if constexpr (HI) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(h), "v"(l));
  else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(h), "v"(l));
  return d;
}
typedef _Float16 h3_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h3_f16x2 __attribute__((ext_vector_type(2)));
typedef float h3_f32x4 __attribute__((ext_vector_type(4)));
typedef float h3_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned h3_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned h3_u32x4 __attribute__((ext_vector_type(4)));
// four / two f32 values -> packed hi halves and packed lo halves (lo of the split form)
__device__ __forceinline__ void h3_split4(h3_f32x4 v, h3_f16x4& hi, h3_f16x4& lo) {
  hi = __builtin_convertvector(v, h3_f16x4);
  const h3_u32x2 p = __builtin_bit_cast(h3_u32x2, hi);
  const h3_f32x4 d = {h3_minus_half<0>(v[0], p[0]), h3_minus_half<1>(v[1], p[0]), h3_minus_half<0>(v[2], p[1]), h3_minus_half<1>(v[3], p[1])};
  lo = __builtin_convertvector(d, h3_f16x4);
}
__device__ __forceinline__ h3_f32x4 h3_join4(h3_f16x4 hi, h3_f16x4 lo) {
  const h3_u32x2 h = __builtin_bit_cast(h3_u32x2, hi), l = __builtin_bit_cast(h3_u32x2, lo);
  return h3_f32x4{h3_sum_halves<0>(h[0], l[0]), h3_sum_halves<1>(h[0], l[0]), h3_sum_halves<0>(h[1], l[1]), h3_sum_halves<1>(h[1], l[1])};
}
__device__ __forceinline__ void h3_split2(h3_f32x2 v, h3_f16x2& hi, h3_f16x2& lo) {
  hi = __builtin_convertvector(v, h3_f16x2);
  const unsigned p = __builtin_bit_cast(unsigned, hi);
  const h3_f32x2 d = {h3_minus_half<0>(v[0], p), h3_minus_half<1>(v[1], p)};
  lo = __builtin_convertvector(d, h3_f16x2);
}
__device__ __forceinline__ h3_f32x2 h3_join2(h3_f16x2 hi, h3_f16x2 lo) {
  const unsigned h = __builtin_bit_cast(unsigned, hi), l = __builtin_bit_cast(unsigned, lo);
  return h3_f32x2{h3_sum_halves<0>(h, l), h3_sum_halves<1>(h, l)};
}
// SINGLE: the plain-f16 variant (one product per pair of operands, |error| <= 2^-10 relative per product): no low part
template <bool SINGLE = false>
__device__ __forceinline__ HFrag split_v8(h3_f32x8 v) {  // 8 register values -> fragment
  HFrag f;
  f.hi = __builtin_convertvector(v, h3_f16x8);
  if constexpr (SINGLE) {
    f.lo = f.hi;  // (never read)
  } else {
    const h3_u32x4 p = __builtin_bit_cast(h3_u32x4, f.hi);
    const h3_f32x8 d = {h3_minus_half<0>(v[0], p[0]), h3_minus_half<1>(v[1], p[0]), h3_minus_half<0>(v[2], p[1]), h3_minus_half<1>(v[3], p[1]),
                        h3_minus_half<0>(v[4], p[2]), h3_minus_half<1>(v[5], p[2]), h3_minus_half<0>(v[6], p[3]), h3_minus_half<1>(v[7], p[3])};
    f.lo = __builtin_convertvector(d, h3_f16x8);
  }
  return f;
}
// registers 8 m .. 8 m + 7 of an MFMA accumulator tile as a fragment: two tiles in the same layout, cut the same way, give
// the two operands of a product the same k order (whatever rows of the tiles those registers are)
template <bool SINGLE = false>
__device__ __forceinline__ HFrag split_acc8(const h3_f32x16& t, int m) {
  return split_v8<SINGLE>(h3_f32x8{t[8 * m], t[8 * m + 1], t[8 * m + 2], t[8 * m + 3], t[8 * m + 4], t[8 * m + 5], t[8 * m + 6], t[8 * m + 7]});
}
template <bool SINGLE = false>
__device__ __forceinline__ HFrag split_h(const float* __restrict__ p) {  // 8 consecutive floats (16-byte aligned)
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  return split_v8<SINGLE>(h3_f32x8{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w});
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
