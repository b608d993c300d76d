// f32 MFMA GEMM shared by the training step (train.hip) and the PointNet++ head (pointnet.hip). gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace t2l {
namespace train {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------------------
// GEMM: C[M,N] (+)= A(m,k) B(k,n) (+ bias[n]) (relu), f32 MFMA 32x32x2.
//   A_KC: A[m*lda + k] (k contiguous)  else A[k*lda + m]
//   B_KC: B[n*ldb + k] (k contiguous)  else B[k*ldb + n]
// These problems are small (M <= a few thousand tokens, N,K <= 1024) and live in L2, so the kernel is built for
// PARALLELISM, not for operand reuse: one workgroup owns ONE 32x32 output tile and its four waves split the reduction
// range four ways; every wave streams its operands straight from global memory into the MFMA operand registers — no
// LDS staging, no barrier in the loop. The k index inside a 16-step is permuted (lane half kh owns k0+8*kh .. +7) so
// that k-contiguous operands are two float4 loads per lane; any permutation is valid as long as A and B share it. The
// four partial tiles meet in LDS once, at the end. M and the reduction range may be ragged; N must be a multiple of 32
// (and M too when !A_KC).
// ---------------------------------------------------------------------------------------------------------------
struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K, lda, ldb, ldc, relu, accumulate, kchunk;
  float* colsum;  // !A_KC only: colsum[m] += sum_k A(m,k)  (bias gradient of the same dY), or nullptr
  int bf16;       // 1: operands rounded to bf16 (RNE) on the fly, ONE v_mfma_f32_32x32x16_bf16 per 16-step instead of eight
                  // v_mfma_f32_32x32x2_f32 — the "bf16" training variant of BASELINE config 4 (f32 accumulation, f32 master
                  // weights, f32 everything else). 2: split-bf16 — every operand as hi + lo bf16 (lo = bf16(v - hi)) and three
                  // MFMAs per 16-step (hi*hi + hi*lo + lo*hi): relative product error <= 2^-16 + 2^-18 with f32's exponent
                  // range (no magnitude guards needed, unlike split-f16), 2.7x fewer matrix-pipe cycles than f32
  // fused element-wise links of the training step (all optional; element index = row * ldc + column, as the standalone kernels):
  float* C2;               // epi == 1: C = relu output (saved for backward), C2 = the same after dropout (the next GEMM's input)
  const float* mask_src;   // epi == 2: value *= (mask_src[idx] > 0) and the dropout factor (ReLU + dropout backward)
  uint32_t drop_key, drop_thr;  // counter-based dropout (train_kernels.h: keep_bit); drop_thr == 0: identity
  float drop_scale;
  int epi;
};

// Workgroups are dealt to the 8 XCDs round robin in launch order, and that order is kept: every product of the step has a multiple of 8
// column tiles (N = 256 / 512 / 768), so block b = bx + gx * by lands on XCD bx % 8 — each L2 holds an eighth of the WEIGHT operand and
// streams the activations. (Round 3 measured the alternative — per-XCD row bands, option "train_xcd_map": 0.589 -> 0.630 ms per step
// with f32 operands, neutral with bf16; removed in round 5)

__device__ __forceinline__ bool gemm_keep_bit(uint32_t key, uint32_t idx, uint32_t thr) {  // == train_kernels.h: keep_bit
  uint32_t x = idx * 0x9E3779B1u + key;
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return (x >> 8) >= thr;
}

typedef __bf16 gemm_bf16x8 __attribute__((ext_vector_type(8)));
typedef float gemm_f32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ gemm_bf16x8 gemm_to_bf16(const float (&v)[8]) {
  return __builtin_convertvector(gemm_f32x8{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]}, gemm_bf16x8);  // v_cvt_pk_bf16_f32
}
__device__ __forceinline__ void gemm_split_bf16(const float (&v)[8], gemm_bf16x8& hi, gemm_bf16x8& lo) {
  const gemm_f32x8 x = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
  hi = __builtin_convertvector(x, gemm_bf16x8);
  lo = __builtin_convertvector(x - __builtin_convertvector(hi, gemm_f32x8), gemm_bf16x8);
}

template <bool KC>
__device__ __forceinline__ void gemm_load(const float* __restrict__ P, int ld, int row, bool row_ok, int k0, int kh, int kend,
                                          float (&v)[8]) {
  if (KC) {
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x;
    if (row_ok) {
      const float* p = P + (size_t)row * ld + k0 + 8 * kh;
      x = *reinterpret_cast<const float4*>(p);
      y = *reinterpret_cast<const float4*>(p + 4);
    }
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + 8 * kh + j;
      v[j] = k < kend ? P[(size_t)k * ld + row] : 0.f;
    }
  }
}

template <bool A_KC, bool B_KC>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, int bx, int by, int bz, float* red, float* cred) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 31, kh = lane >> 5;
  const int n0 = bx * 32, m0 = by * 32;
  const int kb = bz * g.kchunk, ke = min(g.K, kb + g.kchunk);
  const int slice = ((((ke - kb) + 3) / 4) + 15) & ~15;  // per-wave share of the reduction range, whole 16-steps
  const int wk0 = kb + w * slice, wk1 = min(ke, wk0 + slice);
  const bool a_ok = !A_KC || (m0 + i) < g.M;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // operand ring, kRing 16-steps deep: a wave's share of the reduction is short (4-16 steps) and every step's operands come
  // from L2 (~1 us away), so the loads of several steps have to be in flight at once — nothing else hides that latency
  constexpr int kRing = 4;
  float a[kRing][8], b[kRing][8], csum = 0.f;
  const bool do_colsum = !A_KC && g.colsum && bx == 0;
#pragma unroll
  for (int d = 0; d < kRing; ++d)
    if (wk0 + 16 * d < wk1) {
      gemm_load<A_KC>(g.A, g.lda, m0 + i, a_ok, wk0 + 16 * d, kh, wk1, a[d]);
      gemm_load<B_KC>(g.B, g.ldb, n0 + i, true, wk0 + 16 * d, kh, wk1, b[d]);
    }
  for (int k0 = wk0; k0 < wk1; k0 += 16 * kRing) {
#pragma unroll
    for (int d = 0; d < kRing; ++d) {
      if (k0 + 16 * d < wk1) {
        if (g.bf16 == 2) {  // split-bf16: three products per 16-step
          gemm_bf16x8 ah, al, bh, bl;
          gemm_split_bf16(a[d], ah, al);
          gemm_split_bf16(b[d], bh, bl);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
          if (do_colsum) {
#pragma unroll
            for (int j = 0; j < 8; ++j) csum += a[d][j];
          }
        } else if (g.bf16) {  // (workgroup-uniform) lane (i, kh) holds k0 + 8*kh + j, j = 0..7: exactly the 32x32x16 operand layout
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gemm_to_bf16(a[d]), gemm_to_bf16(b[d]), acc, 0, 0, 0);
          if (do_colsum) {
#pragma unroll
            for (int j = 0; j < 8; ++j) csum += a[d][j];  // the bias gradient stays an f32 sum
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][j], b[d][j], acc, 0, 0, 0);
            if (do_colsum) csum += a[d][j];
          }
        }
        if (k0 + 16 * (d + kRing) < wk1) {  // refill this slot with the step one ring ahead
          gemm_load<A_KC>(g.A, g.lda, m0 + i, a_ok, k0 + 16 * (d + kRing), kh, wk1, a[d]);
          gemm_load<B_KC>(g.B, g.ldb, n0 + i, true, k0 + 16 * (d + kRing), kh, wk1, b[d]);
        }
      }
    }
  }
  // (see gemm_rows2.h: no VALU read of an accumulator right behind the last MFMA of a loop)
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(w * 16 + r) * 64 + lane] = acc[r];
  if (do_colsum) cred[w * 64 + lane] = csum;
  __syncthreads();
  if (do_colsum && tid < 32) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) s += cred[q * 64 + tid] + cred[q * 64 + 32 + tid];
    unsafeAtomicAdd(g.colsum + m0 + tid, s);
  }
  // epilogue in two phases — every load of the four elements a thread owns first (bias: one column for all four; the ReLU mask),
  // then the stores: a load next to each guarded store made every store wait for the previous one (s_waitcnt vmcnt(0) each)
  {
    const int l = tid & 63, cg = n0 + (l & 31);
    const float bv = (g.bias && bz == 0) ? g.bias[cg] : 0.f;
    float v[4], mk[4];
    size_t idx[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = (tid >> 6) + 4 * q;
      const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      ok[q] = row < g.M;
      idx[q] = (size_t)min(row, g.M - 1) * g.ldc + cg;
      mk[q] = g.epi == 2 ? g.mask_src[idx[q]] : 1.f;
      v[q] = red[(0 * 16 + r) * 64 + l] + red[(1 * 16 + r) * 64 + l] + red[(2 * 16 + r) * 64 + l] + red[(3 * 16 + r) * 64 + l] + bv;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float x = v[q];
      if (g.relu) x = fmaxf(x, 0.f);
      if (g.epi == 2) {  // ReLU + dropout backward of the layer whose output fed this product's left operand
        x = mk[q] > 0.f ? x : 0.f;
        if (g.drop_thr) x = gemm_keep_bit(g.drop_key, (uint32_t)idx[q], g.drop_thr) ? x * g.drop_scale : 0.f;
      }
      if (ok[q]) {
        float* dst = g.C + idx[q];
        if (g.accumulate)
          unsafeAtomicAdd(dst, x);
        else
          *dst = x;
        if (g.epi == 1) g.C2[idx[q]] = gemm_keep_bit(g.drop_key, (uint32_t)idx[q], g.drop_thr) ? x * g.drop_scale : 0.f;
      }
    }
  }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  __shared__ float red[4 * 16 * 64];
  __shared__ float cred[4 * 64];
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  gemm_body<A_KC, B_KC>(g, bx, by, bz, red, cred);
}

// ---------------------------------------------------------------------------------------------------------------
// The same product on a 64 x 64 output block: every wave holds 2 x 2 accumulator tiles over its quarter of the reduction (two A and
// two B fragments feed four MFMA chains per 16-step), so the workgroup reads (64 + 64) rows of operands for four output tiles
// instead of (32 + 32) for one — HALF the L2 -> L1 operand traffic, which is what bounds these GEMMs (rocprofv3, round 3: 14-25 us
// for 0.3-0.9 GFLOP; 86 MB of operand reads for the QKV product). The four partial blocks meet in LDS so that wave t ends up with
// the complete tile t (tm = t >> 1, tn = t & 1) in its registers and stores it itself (48 KiB: three foreign partials per tile).
// Needs every output dimension to be a multiple of 64 where it is not ragged-checked (N; and M when !A_KC); M may be ragged when A_KC.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kGemm4RedFloats = 4 * 3 * 16 * 64;  // 48 KiB
template <bool A_KC, bool B_KC>
__device__ __forceinline__ void gemm_body4(const GemmArgs& g, int bx, int by, int bz, float* red, float* cred) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 31, kh = lane >> 5;
  const int n0 = bx * 64, m0 = by * 64;
  const int kb = bz * g.kchunk, ke = min(g.K, kb + g.kchunk);
  const int slice = ((((ke - kb) + 3) / 4) + 15) & ~15;
  const int wk0 = kb + w * slice, wk1 = min(ke, wk0 + slice);
  const bool a_ok[2] = {!A_KC || (m0 + i) < g.M, !A_KC || (m0 + 32 + i) < g.M};
  f32x16 acc[2][2];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
  constexpr int kRing = 2;  // (four steps of 2 x 2 fragments are 128 registers: with the 64 accumulators they spill at two waves per SIMD)
  float a[kRing][2][8], b[kRing][2][8], csum[2] = {0.f, 0.f};
  const bool do_colsum = !A_KC && g.colsum && bx == 0;
  auto load = [&](int k0, float (&av)[2][8], float (&bv)[2][8]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      gemm_load<A_KC>(g.A, g.lda, m0 + 32 * t + i, a_ok[t], k0, kh, wk1, av[t]);
      gemm_load<B_KC>(g.B, g.ldb, n0 + 32 * t + i, true, k0, kh, wk1, bv[t]);
    }
  };
#pragma unroll
  for (int d = 0; d < kRing; ++d)
    if (wk0 + 16 * d < wk1) load(wk0 + 16 * d, a[d], b[d]);
  for (int k0 = wk0; k0 < wk1; k0 += 16 * kRing) {
#pragma unroll
    for (int d = 0; d < kRing; ++d) {
      if (k0 + 16 * d < wk1) {
        if (g.bf16 == 2) {
          gemm_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            gemm_split_bf16(a[d][t], ah[t], al[t]);
            gemm_split_bf16(b[d][t], bh[t], bl[t]);
          }
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
            }
        } else if (g.bf16) {
          gemm_bf16x8 ah[2], bh[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            ah[t] = gemm_to_bf16(a[d][t]);
            bh[t] = gemm_to_bf16(b[d][t]);
          }
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
              for (int tn = 0; tn < 2; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][tm][j], b[d][tn][j], acc[tm][tn], 0, 0, 0);
        }
        if (do_colsum) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) csum[t] += a[d][t][j];
        }
        if (k0 + 16 * (d + kRing) < wk1) load(k0 + 16 * (d + kRing), a[d], b[d]);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  // wave w parks its partials of the three tiles it does not own: tile t's slot (w - t - 1) & 3 in {0, 1, 2}
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t != w) {
      const int slot = (w - t - 1) & 3;
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((t * 3 + slot) * 16 + r) * 64 + lane] = acc[t >> 1][t & 1][r];
    }
  }
  if (do_colsum) {
    cred[(w * 2 + 0) * 64 + lane] = csum[0];
    cred[(w * 2 + 1) * 64 + lane] = csum[1];
  }
  __syncthreads();
  if (do_colsum && tid < 64) {
    const int t = tid >> 5, c = tid & 31;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) s += cred[(q * 2 + t) * 64 + c] + cred[(q * 2 + t) * 64 + 32 + c];
    unsafeAtomicAdd(g.colsum + m0 + 32 * t + c, s);
  }
  // wave w completes and stores tile w (every accumulator index below is a compile-time constant after the switch)
  f32x16 full;
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (t == w) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        full[r] = acc[t >> 1][t & 1][r] + red[((t * 3 + 0) * 16 + r) * 64 + lane] + red[((t * 3 + 1) * 16 + r) * 64 + lane] +
                  red[((t * 3 + 2) * 16 + r) * 64 + lane];
    }
  const int tm = w >> 1, tn = w & 1;
  const int cg = n0 + 32 * tn + i;
  const float bv = (g.bias && bz == 0) ? g.bias[cg] : 0.f;
  // (element offsets as 32-bit: the training step's tensors are far below 2^31 elements; the dropout counter is 32-bit anyway)
  const int rbase = m0 + 32 * tm + 4 * kh;
  float mk[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = min(rbase + (r & 3) + 8 * (r >> 2), g.M - 1);
    mk[r] = g.epi == 2 ? g.mask_src[(unsigned)row * (unsigned)g.ldc + cg] : 1.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = rbase + (r & 3) + 8 * (r >> 2);
    const unsigned idx = (unsigned)min(row, g.M - 1) * (unsigned)g.ldc + cg;
    float x = full[r] + bv;
    if (g.relu) x = fmaxf(x, 0.f);
    if (g.epi == 2) {
      x = mk[r] > 0.f ? x : 0.f;
      if (g.drop_thr) x = gemm_keep_bit(g.drop_key, idx, g.drop_thr) ? x * g.drop_scale : 0.f;
    }
    if (row < g.M) {
      float* dst = g.C + idx;
      if (g.accumulate)
        unsafeAtomicAdd(dst, x);
      else
        *dst = x;
      if (g.epi == 1) g.C2[idx] = gemm_keep_bit(g.drop_key, idx, g.drop_thr) ? x * g.drop_scale : 0.f;
    }
  }
}
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm4_kernel(GemmArgs g) {
  __shared__ float red[kGemm4RedFloats];
  __shared__ float cred[8 * 64];
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  gemm_body4<A_KC, B_KC>(g, bx, by, bz, red, cred);
}
// Two products that depend on the same dY and on nothing of each other — dW += dY^T X (with the bias gradient) and
// dX = dY W — as ONE launch: workgroups [0, tn_blocks) take the first, the rest the second (a dependent launch costs ~5 us
// whatever its size: 11 of them per training step).
struct GemmPair {
  GemmArgs tn, nn;
  int tn_gx, tn_gy, tn_blocks, nn_gx;
};
static __global__ __launch_bounds__(256) void gemm_pair_kernel(GemmPair p) {
  __shared__ float red[4 * 16 * 64];
  __shared__ float cred[4 * 64];
  const int b = (int)blockIdx.x;
  if (b < p.tn_blocks) {
    const int bx = b % p.tn_gx, by = (b / p.tn_gx) % p.tn_gy, bz = b / (p.tn_gx * p.tn_gy);
    gemm_body<false, false>(p.tn, bx, by, bz, red, cred);
  } else {
    const int c = b - p.tn_blocks;
    gemm_body<true, false>(p.nn, c % p.nn_gx, c / p.nn_gx, 0, red, cred);
  }
}

// the dW + dX pair on 64 x 64 blocks (GemmPair's tn_gx / tn_gy / nn_gx count 64-wide blocks here)
static __global__ __launch_bounds__(256, 2) void gemm4_pair_kernel(const GemmPair p) {
  __shared__ float red[kGemm4RedFloats];
  __shared__ float cred[8 * 64];
  const int b = (int)blockIdx.x;
  if (b < p.tn_blocks) {
    const int bx = b % p.tn_gx, by = (b / p.tn_gx) % p.tn_gy, bz = b / (p.tn_gx * p.tn_gy);
    gemm_body4<false, false>(p.tn, bx, by, bz, red, cred);
  } else {
    const int c = b - p.tn_blocks;
    gemm_body4<true, false>(p.nn, c % p.nn_gx, c / p.nn_gx, 0, red, cred);
  }
}

// The same products for up to three independent jobs of identical shape in ONE launch (the feature branches of the training step):
// blockIdx.z = job for the plain product (its own z is 1), blockIdx.y = job for the dW + dX pair (whose blocks are flattened in x).
struct GemmMulti {
  GemmArgs j[3];
};
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_multi_kernel(GemmMulti m) {
  __shared__ float red[4 * 16 * 64];
  __shared__ float cred[4 * 64];
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  gemm_body<A_KC, B_KC>(m.j[bz], bx, by, 0, red, cred);
}
struct GemmPairMulti {
  GemmPair p[3];
};
static __global__ __launch_bounds__(256) void gemm_pair_multi_kernel(GemmPairMulti m) {
  __shared__ float red[4 * 16 * 64];
  __shared__ float cred[4 * 64];
  const int job = blockIdx.y, b = blockIdx.x;
  const GemmPair& p = m.p[job];
  if (b < p.tn_blocks) {
    const int bx = b % p.tn_gx, by = (b / p.tn_gx) % p.tn_gy, bz = b / (p.tn_gx * p.tn_gy);
    gemm_body<false, false>(p.tn, bx, by, bz, red, cred);
  } else {
    const int c = b - p.tn_blocks;
    gemm_body<true, false>(p.nn, c % p.nn_gx, c / p.nn_gx, 0, red, cred);
  }
}

}  // namespace train
}  // namespace t2l
