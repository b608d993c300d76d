// Row-streaming GEMMs of the PointNet++ training path, second version (pointnet_train.h). gfx950 only.
//
// The edge-MLP products are TALL: M = 10^5 .. 3*10^6 rows, N and K <= 1024. The first version (round 3's gemm_rows_kernel, removed in round 5,
// gemm_kernel<false,false>) loaded every weight fragment from L2 right in front of the MFMAs that consume it — a dependent
// ~1 us load per 8 MFMAs — and ran at 0.3-0.4 of the f32 MFMA peak whatever the operand type (rocprofv3, r03f: 50-63 TFLOP/s
// in f32, the same wall time with bf16 operands). Here the SMALL operand lives in LDS:
//
//   rows2_kernel   C[M,N] = f(A)[M,K] W[N,K]^T (+ bias): persistent workgroups of 8 waves (two per SIMD); the workgroup
//                  converts a chunk of W (up to 8 column tiles, <= 128 KiB) ONCE per pass into LDS in MFMA fragment order
//                  (f32 / bf16 / bf16 hi+lo planes) and every wave then streams a CONTIGUOUS range of rows against it: one A
//                  fragment per 16-step from HBM (4-deep register ring), fragment reads of W are conflict-free ds_read_b128.
//                  Fused links: (a) BatchNorm+ReLU of the PREVIOUS layer applied to A as it is loaded (AFUSE: the post-ReLU
//                  activations a1 of a block's first layer never exist in memory), (b) the per-(cell, channel) BatchNorm
//                  partial sums of THIS layer's output in the epilogue (float32 per wave range, float64 atomics per cell).
//   tn2_kernel     dW[N,K] += dY[M,N]^T f(X)[M,K] (+ db[n] += sum_m dY): a workgroup owns a contiguous chunk of rows and the
//                  WHOLE N x K result in registers (<= 64 tiles of 32x32 over 8 waves); the rows of the chunk go through LDS
//                  once per 32-row step (dY tile and X tile, transposed by the fragment reads), float atomics at the end.
#pragma once
#include <type_traits>

#include "gemm_f32.h"

namespace t2l {
namespace train {

// ---- storage type of the EDGE-ROW tensors (X, y1, a1, y2 and their gradients): float32, or bf16 when the GEMMs run on bf16 operands
// (MODE 1, BASELINE config 4's arithmetic — what torch.autocast(bfloat16) stores between its Linear and BatchNorm modules too). With bf16
// operands every product and every element-wise pass of the step is HBM-bound (round 3: 35 GB per step): halving the bytes of the
// tensors that ARE the traffic is the lever. Accumulation, the BatchNorm sums (taken from the f32 accumulators in the product's
// epilogue, before the rounding), the per-cell tables and every group-level tensor (xout, dxout, arg, ysel) stay float32. A value is
// rounded ONCE, when it is stored (RNE, v_cvt_pk_bf16_f32); every consumer — forward and backward — reads the same rounded value,
// so the ReLU signs and arg-max rows the backward replays are the forward's.
typedef unsigned short pn_bf16;
template <int MODE>
using pn_store_t = std::conditional_t<MODE == 1, pn_bf16, float>;
typedef __bf16 pn_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pn_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pn_pack2(float a, float b) {  // low half = a, high half = b, round to nearest even
  return __builtin_bit_cast(unsigned, __builtin_convertvector(pn_f32x2{a, b}, pn_bf16x2));
}
__device__ __forceinline__ float4 pn_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 pn_ld4(const pn_bf16* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
}
__device__ __forceinline__ void pn_st4(float* p, const float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void pn_st4(pn_bf16* p, const float4 v) { *reinterpret_cast<uint2*>(p) = make_uint2(pn_pack2(v.x, v.y), pn_pack2(v.z, v.w)); }
__device__ __forceinline__ float pn_ld1(const float* p) { return *p; }
__device__ __forceinline__ float pn_ld1(const pn_bf16* p) { return __uint_as_float((unsigned)*p << 16); }
__device__ __forceinline__ void pn_st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void pn_st1(pn_bf16* p, float v) { *p = (pn_bf16)(pn_pack2(v, 0.f) & 0xFFFFu); }
__device__ __forceinline__ float pn_round(float v, const float*) { return v; }
__device__ __forceinline__ float pn_round(float v, const pn_bf16*) { return __uint_as_float(pn_pack2(v, 0.f) << 16); }  // the value a bf16 store keeps

constexpr int kRows2Lds = 128 * 1024;   // LDS bytes of the weight chunk
constexpr int kRows2Threads = 512;
constexpr int kRows2MaxT = 4;  // column tiles per pass (64 accumulator registers: two waves per SIMD without spills)
constexpr int kTn2MaxT = 4;    // k tiles per workgroup of tn2_kernel

struct Rows2Args {  // (A and C are pn_store_t<MODE> arrays: float32, or bf16 in MODE 1 — the pointers are typed float for the host's sake)
  const float* A;      // [M][lda]
  const float* W;      // [N][ldw] (k contiguous)
  const float* bias;   // [N] or nullptr
  float* C;            // [M][ldc]
  int M, N, K, lda, ldw, ldc;
  int tp;              // column tiles per pass (= the TP the kernel was instantiated with; the last pass may be partly past N)
  int rows_per_wave;   // contiguous rows per wave (multiple of 32)
  // AFUSE: A := relu((A - mean[cell][k]) * rg[cell][k] + beta[k]) (BatchNorm in batch-statistics mode + ReLU of the layer below)
  const float *a_mean, *a_rg, *a_beta;
  const int32_t* row_cell;   // AFUSE and/or stats: cell of every row (rows are sorted by cell)
  double* acc;               // stats: acc[cell][0][n] += sum C, acc[cell][1][n] += sum C^2 ([cell][2][1024] doubles) or nullptr
  // scattered output (scat_dst != nullptr; the input gradient of a set-abstraction level): instead of C[row][n],
  // scat_dst[scat_src[row]][n] += value for n < scat_cols (float atomics: several edge rows share a source row) — the [E, kp]
  // gradient of the edge inputs is never stored and no scatter launch reads it back; columns >= scat_cols (the relative positions and
  // the padding) have no consumer
  const int32_t* scat_src;
  float* scat_dst;
  int scat_cols;
};

// W chunk -> LDS in fragment order. Item (t, i, ks, kh): the 8 values W[n0 + 32 t + i][16 ks + 8 kh + j], j = 0..7 — what lane
// (i, kh) of a 32x32 MFMA consumes in 16-step ks for column tile t (the k permutation of gemm_f32.h: lane half kh owns
// k0 + 8 kh + j). Threads walk (kh, ks) fastest, so a row of W is read as one contiguous stream.
//   MODE 0 (f32):   [t][ks][q = 0,1][lane][4 floats]      (two conflict-free ds_read_b128 per fragment)
//   MODE 1 (bf16):  [t][ks][lane][8 bf16]
//   MODE 2 (split): hi plane as MODE 1, lo plane behind it (+ plane bytes)
template <int MODE>
__device__ __forceinline__ void rows2_fill(char* lds, const float* __restrict__ W, int ldw, int n0, int tp, int K, int N) {
  const int KS = K >> 4;
  const int items = tp * 32 * KS * 2;
  const int plane = tp * KS * 64 * 16;
  for (int it = threadIdx.x; it < items; it += kRows2Threads) {
    const int kh = it & 1, ks = (it >> 1) % KS, i = ((it >> 1) / KS) & 31, t = (it >> 1) / (KS * 32);
    const float* p = W + (size_t)min(n0 + 32 * t + i, N - 1) * ldw + 16 * ks + 8 * kh;  // (tiles past N repeat the last row: never stored)
    const float4 x = *reinterpret_cast<const float4*>(p), y = *reinterpret_cast<const float4*>(p + 4);
    const int lane = kh * 32 + i;
    if (MODE == 0) {
      float4* d = reinterpret_cast<float4*>(lds) + ((size_t)(t * KS + ks) * 2) * 64 + lane;
      d[0] = x;
      d[64] = y;
    } else {
      const float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
      gemm_bf16x8 hi, lo;
      if (MODE == 2) gemm_split_bf16(v, hi, lo);
      else hi = gemm_to_bf16(v);
      gemm_bf16x8* d = reinterpret_cast<gemm_bf16x8*>(lds) + (size_t)(t * KS + ks) * 64 + lane;
      *d = hi;
      if (MODE == 2) *reinterpret_cast<gemm_bf16x8*>(reinterpret_cast<char*>(d) + plane) = lo;
    }
  }
}

// TP (column tiles per pass) is a template parameter and the k loop has no data-dependent branch between the issue of a ring
// slot's loads and their use: with run-time tile / step guards the compiler's wait-count pass gives up and puts s_waitcnt
// vmcnt(0) in front of every MFMA group — the ring then hides nothing (first build of this kernel: 44 % of the wave cycles
// waiting on VMEM, 0.41 of the f32 peak). K / 16 is even for every layer (K = 32 .. 512), so the loop runs whole rounds of the
// 4-slot ring plus at most one half round; refills past the end re-load the last step (clamped address, never used).
template <int MODE, bool AFUSE, bool STATS, int TP>
__global__ __launch_bounds__(kRows2Threads) void rows2_kernel(const Rows2Args g) {
  extern __shared__ __attribute__((aligned(16))) char r2_lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 31, kh = lane >> 5;
  const int KS = g.K >> 4;
  const long gw = (long)blockIdx.x * 8 + w;
  const long row_lo = gw * g.rows_per_wave, row_hi = min((long)g.M, row_lo + g.rows_per_wave);
  constexpr int kRing = 4;
  const int plane = TP * KS * 64 * 16;
  typedef pn_store_t<MODE> ST;
  const ST* __restrict__ gA = reinterpret_cast<const ST*>(g.A);
  ST* __restrict__ gC = reinterpret_cast<ST*>(g.C);
  for (int n0 = 0; n0 < g.N; n0 += 32 * TP) {
    __syncthreads();  // the previous pass's readers are done
    rows2_fill<MODE>(r2_lds, g.W, g.ldw, n0, TP, g.K, g.N);
    __syncthreads();
    // running BatchNorm sums of this wave's rows for the cell `cur` (flushed when the cell changes)
    float run1[TP], run2[TP], bv[TP];
    int cur = -1;
#pragma unroll
    for (int t = 0; t < TP; ++t) {
      run1[t] = run2[t] = 0.f;
      bv[t] = g.bias && n0 + 32 * t + i < g.N ? g.bias[n0 + 32 * t + i] : 0.f;
    }
    auto flush = [&]() {
      if (cur < 0) return;
#pragma unroll
      for (int t = 0; t < TP; ++t)
        if (n0 + 32 * t < g.N && kh == 0) {
          double* p = g.acc + ((size_t)cur * 2) * 1024 + n0 + 32 * t + i;
          atomicAdd(p, (double)run1[t]);
          atomicAdd(p + 1024, (double)run2[t]);
          run1[t] = run2[t] = 0.f;
        }
    };
    for (long m0 = row_lo; m0 < row_hi; m0 += 32) {
      const long arow = min(m0 + i, (long)g.M - 1);  // rows past the end repeat the last one (never stored, never counted)
      const ST* ap = gA + (size_t)arow * g.lda + 8 * kh;
      int acell = 0;
      if (AFUSE) acell = g.row_cell[arow];
      const float* mp = AFUSE ? g.a_mean + (size_t)acell * g.K + 8 * kh : nullptr;
      const float* rp = AFUSE ? g.a_rg + (size_t)acell * g.K + 8 * kh : nullptr;
      const float* bp = AFUSE ? g.a_beta + 8 * kh : nullptr;
      f32x16 acc[TP];
#pragma unroll
      for (int t = 0; t < TP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      float a[kRing][8];
      auto load_a = [&](int ks, float (&d)[8]) {
#ifdef T2L_EXP_R2_NOLOAD  // dev experiment (wrong results): every k-step re-reads the row's first 16 bytes (L1-hot): what does the A stream cost?
        const int k0 = 0 * ks;
#else
        const int k0 = 16 * min(ks, KS - 1);
#endif
        if constexpr (MODE == 1) {  // bf16 rows: the lane's 8 k values are ONE 16-byte load
          const uint4 u = *reinterpret_cast<const uint4*>(ap + k0);
          d[0] = __uint_as_float(u.x << 16); d[1] = __uint_as_float(u.x & 0xFFFF0000u); d[2] = __uint_as_float(u.y << 16);
          d[3] = __uint_as_float(u.y & 0xFFFF0000u); d[4] = __uint_as_float(u.z << 16); d[5] = __uint_as_float(u.z & 0xFFFF0000u);
          d[6] = __uint_as_float(u.w << 16); d[7] = __uint_as_float(u.w & 0xFFFF0000u);
        } else {
          const float4 x = pn_ld4(ap + k0), y = pn_ld4(ap + k0 + 4);
          d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w; d[4] = y.x; d[5] = y.y; d[6] = y.z; d[7] = y.w;
        }
        if (AFUSE) {
          const float4 m0v = *reinterpret_cast<const float4*>(mp + k0), m1v = *reinterpret_cast<const float4*>(mp + k0 + 4);
          const float4 r0v = *reinterpret_cast<const float4*>(rp + k0), r1v = *reinterpret_cast<const float4*>(rp + k0 + 4);
          const float4 b0v = *reinterpret_cast<const float4*>(bp + k0), b1v = *reinterpret_cast<const float4*>(bp + k0 + 4);
          const float mm[8] = {m0v.x, m0v.y, m0v.z, m0v.w, m1v.x, m1v.y, m1v.z, m1v.w};
          const float rr[8] = {r0v.x, r0v.y, r0v.z, r0v.w, r1v.x, r1v.y, r1v.z, r1v.w};
          const float bb[8] = {b0v.x, b0v.y, b0v.z, b0v.w, b1v.x, b1v.y, b1v.z, b1v.w};
#pragma unroll
          for (int j = 0; j < 8; ++j) d[j] = fmaxf(__fmaf_rn(__fsub_rn(d[j], mm[j]), rr[j], bb[j]), 0.f);
        }
      };
      auto step = [&](int ks, const float (&av)[8]) {
        gemm_bf16x8 ah, al;
        if (MODE == 2) gemm_split_bf16(av, ah, al);
        else if (MODE == 1) ah = gemm_to_bf16(av);
#pragma unroll
        for (int t = 0; t < TP; ++t) {
          if (MODE == 0) {
            const float4* f = reinterpret_cast<const float4*>(r2_lds) + ((size_t)(t * KS + ks) * 2) * 64 + lane;
            const float4 x = f[0], y = f[64];
            const float b[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b[j], acc[t], 0, 0, 0);
          } else {
            const gemm_bf16x8* f = reinterpret_cast<const gemm_bf16x8*>(r2_lds) + (size_t)(t * KS + ks) * 64 + lane;
            const gemm_bf16x8 bh = *f;
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
            if (MODE == 2) {
              const gemm_bf16x8 bl = *reinterpret_cast<const gemm_bf16x8*>(reinterpret_cast<const char*>(f) + plane);
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
            }
          }
        }
      };
#pragma unroll
      for (int d = 0; d < kRing; ++d) load_a(d, a[d]);
      int kb = 0;
      for (; kb + kRing <= KS; kb += kRing) {
#pragma unroll
        for (int d = 0; d < kRing; ++d) {
          step(kb + d, a[d]);
          load_a(kb + d + kRing, a[d]);  // refill this slot with the step one ring ahead (clamped past the end)
        }
      }
      if (kb < KS) {  // K / 16 = 2 mod 4: the half round
        step(kb, a[0]);
        step(kb + 1, a[1]);
      }
      // (the same guard as in tn2_kernel: no VALU read of an accumulator closer than a 16-pass MFMA's latency behind its issue)
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // epilogue: bias, store, BatchNorm partial sums
      bool uniform = true;
      if (STATS) {
        const int c_first = g.row_cell[m0], c_last = g.row_cell[min(m0 + 31, (long)g.M - 1)];
        uniform = c_first == c_last;
        if (uniform && c_first != cur) {
          flush();
          cur = c_first;
        }
      }
      // (full tiles — all but a wave's last — store without per-row branches: a guarded store makes the compiler wait for
      //  vmcnt(0), i.e. for the previous STORE, in front of every store)
      const bool full = m0 + 32 <= (long)g.M;
#pragma unroll
      for (int t = 0; t < TP; ++t) {
        const int cg = n0 + 32 * t + i;
        if (cg < g.N) {
          float v[16];  // (the BatchNorm sums below take these float32 values: the statistics of the product, not of its rounded copy)
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = acc[t][r] + bv[t];
          if (g.scat_dst) {  // (workgroup-uniform)
            if (cg < g.scat_cols) {
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const long row = m0 + 4 * kh + (r & 3) + 8 * (r >> 2);
                if (row < (long)g.M) unsafeAtomicAdd(g.scat_dst + (size_t)g.scat_src[row] * g.scat_cols + cg, v[r]);
              }
            }
            continue;
          }
          ST* cp = gC + (size_t)(m0 + 4 * kh) * g.ldc + cg;
#ifdef T2L_EXP_R2_NOSTORE  // dev experiment (wrong results): what do the epilogue's strided stores cost?
          if (v[0] == 1.2345e30f)
#endif
          if (full) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pn_st1(cp + (size_t)((r & 3) + 8 * (r >> 2)) * g.ldc, v[r]);
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (m0 + 4 * kh + (r & 3) + 8 * (r >> 2) < (long)g.M) pn_st1(cp + (size_t)((r & 3) + 8 * (r >> 2)) * g.ldc, v[r]);
          }
          if (STATS) {
            if (uniform && full) {
              float s1 = 0.f, s2 = 0.f;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                s1 += v[r];
                s2 += v[r] * v[r];
              }
              s1 += __shfl_xor(s1, 32);
              s2 += __shfl_xor(s2, 32);
              run1[t] += s1;
              run2[t] += s2;
            } else {  // a tile that straddles two cells (one per cell boundary) or the ragged last tile: element-wise
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const long row = m0 + 4 * kh + (r & 3) + 8 * (r >> 2);
                if (row < (long)g.M) {
                  double* p = g.acc + ((size_t)g.row_cell[row] * 2) * 1024 + cg;
                  atomicAdd(p, (double)v[r]);
                  atomicAdd(p + 1024, (double)v[r] * (double)v[r]);
                }
              }
            }
          }
        }
      }
    }
    if (STATS) flush();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// tn2_kernel: dW[N,K] += dY[M,N]^T f(X)[M,K], db[n] += sum_m dY[m][n]. Workgroup (bx, by, bz): rows [bx * rows_per_wg, +rows_per_wg),
// output block n in [256 by, +NB), k in [32 kb_tiles bz, +KB) (NB <= 256, KB <= 128: <= 8 x 4 tiles of 32 x 32). Wave w: n tile nt = w % NT and
// row group rg = w / NT of the RG = 8 / NT groups of 32 rows a step stages (NT = NB / 32 is 1, 2, 4 or 8), ALL k tiles: one A
// fragment (dY, n x m) feeds KT MFMAs (chains) per 16-step; partial results of the row groups and of the workgroups meet in the
// float atomics at the end (gradients accumulate anyway). A step's rows go global -> registers -> LDS (double buffered, one
// barrier per step; f32 row-major, row stride = 4 mod 8 floats so the column-wise fragment reads of both lane halves are
// conflict-free); XFUSE applies the BatchNorm + ReLU of the layer that produced X while staging (a1 is never stored).
// ---------------------------------------------------------------------------------------------------------------
struct Tn2Args {   // (dY and X are pn_store_t<MODE> arrays, as Rows2Args' A and C)
  const float* dY;   // [M][ldy]
  const float* X;    // [M][ldx]
  float* dW;         // [N][ldw]
  float* db;         // [N] or nullptr
  int M, N, K, ldy, ldx, ldw, rows_per_wg;
  int kb_tiles;      // k tiles (of 32) per workgroup block along K (<= kTn2MaxT); blockIdx.z walks the blocks
  int k_real;        // columns k >= k_real of the product are padding of X and are not written (dW has ldw >= k_real columns)
  const float *x_mean, *x_rg, *x_beta;  // XFUSE: [cell][K], [cell][K], [K]
  const int32_t* row_cell;
};
// NT and KT are template parameters: the LDS row strides are compile-time constants (every fragment read is base + immediate),
// the staging loops are exact and branch-free (rows past the chunk are loaded from a clamped address and zeroed by a select).
template <int MODE, bool XFUSE, int NT, int KT>
__global__ __launch_bounds__(kRows2Threads) void tn2_kernel(const Tn2Args g) {
  extern __shared__ __attribute__((aligned(16))) char t2_lds[];
  constexpr int NB = 32 * NT, KB = 32 * KT, RG = 8 / NT, SR = RG * 32;
  constexpr int ldyl = NB + 4, ldxl = KB + 4;  // LDS row strides (floats)
  constexpr int buf_floats = SR * (ldyl + ldxl);
  constexpr int UY = SR * (NB / 4) / kRows2Threads;                        // = 4: float4 units of dY per thread and step
  constexpr int UXT = SR * (KB / 4), UX = (UXT + kRows2Threads - 1) / kRows2Threads;  // ... of X (the last one may be partial)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 31, kh = lane >> 5;
  const int n_base = 256 * blockIdx.y, k_base = 32 * g.kb_tiles * blockIdx.z;
  const int nt = w % NT, rg = w / NT;
  float* lds = reinterpret_cast<float*>(t2_lds);
  const long m_lo = (long)blockIdx.x * g.rows_per_wg, m_hi = min((long)g.M, m_lo + g.rows_per_wg);
  if (m_hi <= m_lo) return;
  const int steps = (int)((m_hi - m_lo + SR - 1) / SR);
  f32x16 acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float csum = 0.f;
  const bool do_db = g.db && blockIdx.z == 0;
  float4 sy[UY], sx[UX];
  typedef pn_store_t<MODE> ST;
  const ST* __restrict__ gY = reinterpret_cast<const ST*>(g.dY);
  const ST* __restrict__ gX = reinterpret_cast<const ST*>(g.X);
  auto load_regs = [&](int s) {
    const long m0 = m_lo + (long)s * SR;
#pragma unroll
    for (int q = 0; q < UY; ++q) {
      const int u = tid + q * kRows2Threads, r = u / (NB / 4), c = (u % (NB / 4)) * 4;
      const long row = m0 + r;
      const float4 v = pn_ld4(gY + (size_t)min(row, m_hi - 1) * g.ldy + n_base + c);
      sy[q] = row < m_hi ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < UX; ++q) {
      const int u = min(tid + q * kRows2Threads, UXT - 1), r = u / (KB / 4), c = (u % (KB / 4)) * 4;
      const long row = m0 + r, rowc = min(row, m_hi - 1);
      float4 v = pn_ld4(gX + (size_t)rowc * g.ldx + k_base + c);
      if (XFUSE) {
        const size_t o = (size_t)g.row_cell[rowc] * g.K + k_base + c;
        const float4 mm = *reinterpret_cast<const float4*>(g.x_mean + o), rr = *reinterpret_cast<const float4*>(g.x_rg + o),
                     bb = *reinterpret_cast<const float4*>(g.x_beta + k_base + c);
        v.x = fmaxf(__fmaf_rn(__fsub_rn(v.x, mm.x), rr.x, bb.x), 0.f);
        v.y = fmaxf(__fmaf_rn(__fsub_rn(v.y, mm.y), rr.y, bb.y), 0.f);
        v.z = fmaxf(__fmaf_rn(__fsub_rn(v.z, mm.z), rr.z, bb.z), 0.f);
        v.w = fmaxf(__fmaf_rn(__fsub_rn(v.w, mm.w), rr.w, bb.w), 0.f);
      }
      sx[q] = row < m_hi ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto write_lds = [&](int b) {
    float* yb = lds + (size_t)b * buf_floats;
    float* xb = yb + SR * ldyl;
#pragma unroll
    for (int q = 0; q < UY; ++q) {
      const int u = tid + q * kRows2Threads, r = u / (NB / 4), c = (u % (NB / 4)) * 4;
      *reinterpret_cast<float4*>(yb + r * ldyl + c) = sy[q];
    }
#pragma unroll
    for (int q = 0; q < UX; ++q) {
      const int u = tid + q * kRows2Threads, r = u / (KB / 4), c = (u % (KB / 4)) * 4;
      if (UXT % kRows2Threads == 0 || u < UXT) *reinterpret_cast<float4*>(xb + r * ldxl + c) = sx[q];
    }
  };
  load_regs(0);
  write_lds(0);
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    if (s + 1 < steps) load_regs(s + 1);
    const float* yb = lds + (size_t)(s & 1) * buf_floats;
    const float* xb = yb + SR * ldyl;
    if (do_db && tid < NB) {
      float c = 0.f;
#pragma unroll 8
      for (int r = 0; r < SR; ++r) c += yb[r * ldyl + tid];
      csum += c;
    }
    const float* ya = yb + (rg * 32 + 8 * kh) * ldyl + nt * 32 + i;
    const float* xa = xb + (rg * 32 + 8 * kh) * ldxl + i;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float a[8];
#pragma unroll
#ifdef T2L_EXP_TN2_NOLDS  // dev experiment (wrong results): fragments from registers instead of the transposed LDS reads
      for (int j = 0; j < 8; ++j) a[j] = csum + (float)j;
#else
      for (int j = 0; j < 8; ++j) a[j] = ya[(16 * h + j) * ldyl];
#endif
      gemm_bf16x8 ah, al;
      if (MODE == 2) gemm_split_bf16(a, ah, al);
      else if (MODE == 1) ah = gemm_to_bf16(a);
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        float b[8];
#pragma unroll
#ifdef T2L_EXP_TN2_NOLDS
        for (int j = 0; j < 8; ++j) b[j] = csum + (float)(j + t);
#else
        for (int j = 0; j < 8; ++j) b[j] = xa[(16 * h + j) * ldxl + t * 32];
#endif
        if (MODE == 0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[t], 0, 0, 0);
        } else if (MODE == 1) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gemm_to_bf16(b), acc[t], 0, 0, 0);
        } else {
          gemm_bf16x8 bh, bl;
          gemm_split_bf16(b, bh, bl);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
        }
      }
    }
    // ROCm 7.2 / gfx950: without this wait, when the loop is left the register allocator's copy of the last accumulator register
    // follows the final 16-pass MFMA by five scalar instructions (branch, compare, wait, barrier, branch) and reads it BEFORE the
    // MFMA has written it (observed: output rows 27 and 31 of tile 0 — accumulator register 15 — off by 4 % in
    // tn2_kernel<0, true, 2, 1>, nothing else wrong). 32 idle cycles per step against thousands of MFMA cycles.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < steps) write_lds((s + 1) & 1);
    __syncthreads();
  }
#ifdef T2L_EXP_TN2_NOATOMIC  // dev experiment (wrong results): what do the final float atomics cost?
  if (acc[0][0] != 1.2345e30f) return;
#endif
  // The RG row groups of the workgroup hold partial sums of the SAME output tiles: they meet in LDS (the staging buffers are free
  // now) and only the first group's waves go to memory — the float atomics of all 256 workgroups land on the same N x K addresses, and
  // for the small first-level gradients (32 x 32: 1,024 addresses) 8 x as many of them cost 174 us of a 411 us launch (round 6).
  if constexpr (RG >= 4) {  // (two row groups of a wide block: the LDS round costs more than the uncontended atomics it saves — measured)
    static_assert((size_t)8 * KT * 16 * 64 <= (size_t)2 * buf_floats, "the partial tiles fit the staging buffers");
    float* red = lds + (size_t)(w * KT) * 1024 + lane;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64] = acc[t][r];
    __syncthreads();
    if (rg != 0) return;
#pragma unroll
    for (int q = 1; q < RG; ++q) {
      const float* o = lds + (size_t)((q * NT + nt) * KT) * 1024 + lane;
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] += o[(t * 16 + r) * 64];
    }
  }
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const int k = k_base + t * 32 + i;
    if (k < g.k_real) {
      float* dp = g.dW + (size_t)(n_base + nt * 32 + 4 * kh) * g.ldw + k;
#pragma unroll
      for (int r = 0; r < 16; ++r) unsafeAtomicAdd(dp + (size_t)((r & 3) + 8 * (r >> 2)) * g.ldw, acc[t][r]);
    }
  }
  if (do_db && tid < NB) unsafeAtomicAdd(g.db + n_base + tid, csum);
}

}  // namespace train
}  // namespace t2l
