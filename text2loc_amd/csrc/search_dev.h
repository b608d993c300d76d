// Device-side building blocks shared by the search kernels (search.hip, search_stream.hip): key lists, split-bf16
// helpers, the MFMA+selection tile body, wave reductions, the float64 exact scan. gfx950 only.
#pragma once
#include <float.h>
#include <limits.h>

#include "t2l_internal.h"

namespace t2l {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define T2L_NEG_INF (-__builtin_inff())

// Sorted (descending) register list of keys. Element I of the NEW list depends only on OLD values:
// s'[I] = med3(s[I-1], s[I], x) for a descending s, s'[0] = max(s[0], x). Updating I = L-1 .. 0 in place
// therefore needs no temporaries and no compares.
template <int L, int I = L - 1>
__device__ __forceinline__ void ins_key(float (&s)[L], float x) {
  if constexpr (I == 0) {
    s[0] = __builtin_amdgcn_fmed3f(s[0], x, __builtin_inff());  // max without a canonicalising extra op
  } else {
    s[I] = __builtin_amdgcn_fmed3f(s[I - 1], s[I], x);
    ins_key<L, I - 1>(s, x);
  }
}

__device__ __forceinline__ float make_key(float v, int mask, int code) {
  return __int_as_float((__float_as_int(v) & mask) | code);
}


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned bf16_rne_bits(float x) {
  const unsigned u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
  const unsigned h0 = bf16_rne_bits(x0), h1 = bf16_rne_bits(x1);
  const unsigned l0 = bf16_rne_bits(x0 - __uint_as_float(h0 << 16)), l1 = bf16_rne_bits(x1 - __uint_as_float(h1 << 16));
  hi = h0 | (h1 << 16);
  lo = l0 | (l1 << 16);
}
__device__ __forceinline__ void split8(const float4 a, const float4 b, uint4& hi, uint4& lo) {
  split_pair(a.x, a.y, hi.x, lo.x);
  split_pair(a.z, a.w, hi.y, lo.y);
  split_pair(b.x, b.y, hi.z, lo.z);
  split_pair(b.z, b.w, hi.w, lo.w);
}


template <int L, int I = L - 1>
__device__ __forceinline__ void ins_key_sat(float (&s)[L], float x, float pinf) {
  if constexpr (I == 0) {
    s[0] = __builtin_amdgcn_fmed3f(s[0], x, pinf);  // max(s0, x) as ONE op (pinf is a run-time +inf: no folding to
  } else {                                          // the canonicalising v_max pair)
    s[I] = __builtin_amdgcn_fmed3f(s[I - 1], s[I], x);
    ins_key_sat<L, I - 1>(s, x, pinf);
  }
}

// 48 MFMAs of a tile on ONE accumulator chain, 3 per k-step, with one score of the previous tile inserted per
// k-step (measured: for the bf16 MFMA a single chain with ~6 interleaved VALU per MFMA beats two alternating
// chains, which cost 32 more VGPRs and a spill at two waves per SIMD; issuing the next tile's LDS-DMA rows one per
// k-step inside this phase instead of in a burst after the barrier was also measured: no net gain).
template <int L, int VPM, int S, int S_END>
__device__ __forceinline__ void tile_mfma_bf16_sel(const char* tb, const uint4 (&qh)[16], const uint4 (&ql)[16],
                                                   f32x16& cur, const f32x16& prev, int vmask, int code0, float pinf,
                                                   float (&ls)[L], uint4 (&ah)[4], uint4 (&al)[4]) {
  if constexpr (S < S_END) {
    if constexpr (S == 0) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);  // prologue LDS reads first
    const uint4 a_hi = ah[S & 3], a_lo = al[S & 3];
    if constexpr (S + 4 < 16) {
      ah[S & 3] = *reinterpret_cast<const uint4*>(tb + 16 * (S + 4));
      al[S & 3] = *reinterpret_cast<const uint4*>(tb + 512 + 16 * (S + 4));
    }
    const bf16x8 vh = __builtin_bit_cast(bf16x8, a_hi), vl = __builtin_bit_cast(bf16x8, a_lo);
    const bf16x8 bh = __builtin_bit_cast(bf16x8, qh[S]), bl = __builtin_bit_cast(bf16x8, ql[S]);
    cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, bh, cur, 0, 0, 0);
    cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, bl, cur, 0, 0, 0);
    cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, bh, cur, 0, 0, 0);
    {  // score S of the previous tile enters the list: 2 + L VALU
      const int code = __builtin_amdgcn_readfirstlane(code0 + S);
      const float key = __int_as_float((__float_as_int(prev[S]) & vmask) | code);
      ins_key_sat<L>(ls, key, pinf);
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
    }
    if constexpr (S + 4 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    tile_mfma_bf16_sel<L, VPM, S + 1, S_END>(tb, qh, ql, cur, prev, vmask, code0, pinf, ls, ah, al);
  }
}


// ---- wave-wide all-reduces on the VALU (DPP + v_permlane{16,32}_swap), no LDS traffic: __shfl_xor lowers to
// ds_bpermute_b32, and the re-rank is shuffle-bound (hundreds of shuffles per query).
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;

__device__ __forceinline__ float wave_max_f32(float v, float pinf) {  // every lane gets the maximum
#define T2L_MAXSTEP(o) v = __builtin_amdgcn_fmed3f(v, (o), pinf)
  T2L_MAXSTEP(__uint_as_float(dpp_u<kDppXor1>(__float_as_uint(v))));
  T2L_MAXSTEP(__uint_as_float(dpp_u<kDppXor2>(__float_as_uint(v))));
  T2L_MAXSTEP(__uint_as_float(dpp_u<kDppHalfMirror>(__float_as_uint(v))));
  T2L_MAXSTEP(__uint_as_float(dpp_u<kDppMirror>(__float_as_uint(v))));
  {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __builtin_amdgcn_fmed3f(__uint_as_float(r[0]), __uint_as_float(r[1]), pinf);
  }
  {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __builtin_amdgcn_fmed3f(__uint_as_float(r[0]), __uint_as_float(r[1]), pinf);
  }
#undef T2L_MAXSTEP
  return v;
}

template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  const unsigned long long b = __double_as_longlong(v);
  const unsigned lo = dpp_u<CTRL>((unsigned)b), hi = dpp_u<CTRL>((unsigned)(b >> 32));
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {  // every lane gets the sum (fixed tree order)
  v += dpp_d<kDppXor1>(v);
  v += dpp_d<kDppXor2>(v);
  v += dpp_d<kDppHalfMirror>(v);
  v += dpp_d<kDppMirror>(v);
  {
    const unsigned long long b = __double_as_longlong(v);
    const auto l = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    v = __longlong_as_double(((unsigned long long)h[0] << 32) | l[0]) +
        __longlong_as_double(((unsigned long long)h[1] << 32) | l[1]);
  }
  {
    const unsigned long long b = __double_as_longlong(v);
    const auto l = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    v = __longlong_as_double(((unsigned long long)h[0] << 32) | l[0]) +
        __longlong_as_double(((unsigned long long)h[1] << 32) | l[1]);
  }
  return v;
}

// key -> local DB row. part = 2*split + half.
__device__ __forceinline__ int key_row(float key, int part, int per, int code_bits) {
  const int code = __float_as_int(key) & ((1 << code_bits) - 1);
  const int r = code & 15;
  return (((part >> 1) * per + (code >> 4)) << 5) + (r & 3) + 8 * (r >> 2) + 4 * (part & 1);
}

// float64 dot of DB row `row` with the query fragment held by the wave (lane owns dims 4*lane..4*lane+3)
__device__ __forceinline__ double wave_dot64(const float* __restrict__ db, int row, const float4 qv, int lane) {
  const float4 dv = reinterpret_cast<const float4*>(db + (size_t)row * kD)[lane];
  double d = (double)dv.x * qv.x + (double)dv.y * qv.y + (double)dv.z * qv.z + (double)dv.w * qv.w;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) d += __shfl_xor(d, off);
  return d;
}

// bound on (float64 score - key) for any row whose key is <= g: truncation of `code_bits` mantissa bits
// (relative 2^(code_bits-23), doubled for slack) plus the f32 dot-product rounding error eps32.
__device__ __forceinline__ double key_slack(float g, int code_bits, double eps32) {
  return fabs((double)g) * ldexp(1.0, code_bits - 22) + eps32;
}


template <int KMAX>
__device__ void exact_scan(const float* __restrict__ db, int n_rows, const double* qs, int K, int row_offset,
                           int32_t* __restrict__ out_idx, double* __restrict__ out_score, double* red_s, int* red_i,
                           int* red_t) {
  const int tid = threadIdx.x;
  double ls[KMAX];
  int li[KMAX];
#pragma unroll
  for (int i = 0; i < KMAX; ++i) {
    ls[i] = -__builtin_inf();
    li[i] = INT_MAX;
  }
  for (int row = tid; row < n_rows; row += 256) {
    const float4* rp = reinterpret_cast<const float4*>(db + (size_t)row * kD);
    double d = 0.0;
    for (int k = 0; k < kD / 4; ++k) {
      const float4 v = rp[k];
      d += (double)v.x * qs[4 * k] + (double)v.y * qs[4 * k + 1] + (double)v.z * qs[4 * k + 2] +
           (double)v.w * qs[4 * k + 3];
    }
    if (d > ls[KMAX - 1]) {  // rows ascend per thread: strict > keeps the lower row ahead among equals
#pragma unroll
      for (int i = KMAX - 1; i >= 1; --i) {
        const bool c_prev = d > ls[i - 1];
        const bool c_cur = d > ls[i];
        li[i] = c_prev ? li[i - 1] : (c_cur ? row : li[i]);
        ls[i] = c_prev ? ls[i - 1] : (c_cur ? d : ls[i]);
      }
      const bool c0 = d > ls[0];
      li[0] = c0 ? row : li[0];
      ls[0] = c0 ? d : ls[0];
    }
  }
  for (int r = 0; r < K; ++r) {
    red_s[tid] = ls[0];
    red_i[tid] = li[0];
    red_t[tid] = tid;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
      if (tid < st) {
        const double os = red_s[tid + st];
        const int oi = red_i[tid + st];
        if (os > red_s[tid] || (os == red_s[tid] && oi < red_i[tid])) {
          red_s[tid] = os;
          red_i[tid] = oi;
          red_t[tid] = red_t[tid + st];
        }
      }
      __syncthreads();
    }
    const int win = red_t[0];
    if (tid == 0) {
      const bool ok = red_i[0] != INT_MAX;
      out_idx[r] = ok ? red_i[0] + row_offset : -1;
      if (out_score) out_score[r] = ok ? red_s[0] : -__builtin_inf();
    }
    if (tid == win) {  // pop the winner's head
#pragma unroll
      for (int i = 0; i < KMAX - 1; ++i) {
        ls[i] = ls[i + 1];
        li[i] = li[i + 1];
      }
      ls[KMAX - 1] = -__builtin_inf();
      li[KMAX - 1] = INT_MAX;
    }
    __syncthreads();
  }
}


}  // namespace t2l
