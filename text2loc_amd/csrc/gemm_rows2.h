// Row-streaming GEMMs of the PointNet++ training path, second version (pointnet_train.h). gfx950 only.
//
// The edge-MLP products are TALL: M = 10^5 .. 3*10^6 rows, N and K <= 1024. The first version (gemm_f32.h: gemm_rows_kernel,
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
#include "gemm_f32.h"

namespace t2l {
namespace train {

constexpr int kRows2Lds = 128 * 1024;   // LDS bytes of the weight chunk
constexpr int kRows2Threads = 512;
constexpr int kRows2MaxT = 4;  // column tiles per pass (64 accumulator registers: two waves per SIMD without spills)
constexpr int kTn2MaxT = 4;    // k tiles per workgroup of tn2_kernel

struct Rows2Args {
  const float* A;      // [M][lda]
  const float* W;      // [N][ldw] (k contiguous)
  const float* bias;   // [N] or nullptr
  float* C;            // [M][ldc]
  int M, N, K, lda, ldw, ldc;
  int tp;              // column tiles per pass (host: min(kRows2MaxT, LDS budget))
  int rows_per_wave;   // contiguous rows per wave (multiple of 32)
  // AFUSE: A := relu((A - mean[cell][k]) * rg[cell][k] + beta[k]) (BatchNorm in batch-statistics mode + ReLU of the layer below)
  const float *a_mean, *a_rg, *a_beta;
  const int32_t* row_cell;   // AFUSE and/or stats: cell of every row (rows are sorted by cell)
  double* acc;               // stats: acc[cell][0][n] += sum C, acc[cell][1][n] += sum C^2 ([cell][2][1024] doubles) or nullptr
};

// W chunk -> LDS in fragment order. Item (t, i, ks, kh): the 8 values W[n0 + 32 t + i][16 ks + 8 kh + j], j = 0..7 — what lane
// (i, kh) of a 32x32 MFMA consumes in 16-step ks for column tile t (the k permutation of gemm_f32.h: lane half kh owns
// k0 + 8 kh + j). Threads walk (kh, ks) fastest, so a row of W is read as one contiguous stream.
//   MODE 0 (f32):   [t][ks][q = 0,1][lane][4 floats]      (two conflict-free ds_read_b128 per fragment)
//   MODE 1 (bf16):  [t][ks][lane][8 bf16]
//   MODE 2 (split): hi plane as MODE 1, lo plane behind it (+ plane bytes)
template <int MODE>
__device__ __forceinline__ void rows2_fill(char* lds, const float* __restrict__ W, int ldw, int n0, int tp, int K) {
  const int KS = K >> 4;
  const int items = tp * 32 * KS * 2;
  const int plane = tp * KS * 64 * 16;
  for (int it = threadIdx.x; it < items; it += kRows2Threads) {
    const int kh = it & 1, ks = (it >> 1) % KS, i = ((it >> 1) / KS) & 31, t = (it >> 1) / (KS * 32);
    const float* p = W + (size_t)(n0 + 32 * t + i) * ldw + 16 * ks + 8 * kh;
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

template <int MODE, bool AFUSE, bool STATS>
__global__ __launch_bounds__(kRows2Threads) void rows2_kernel(const Rows2Args g) {
  extern __shared__ __attribute__((aligned(16))) char r2_lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 31, kh = lane >> 5;
  const int KS = g.K >> 4;
  const long gw = (long)blockIdx.x * 8 + w;
  const long row_lo = gw * g.rows_per_wave, row_hi = min((long)g.M, row_lo + g.rows_per_wave);
  constexpr int kRing = 4;
  for (int n0 = 0; n0 < g.N; n0 += 32 * g.tp) {
    const int tp = min(g.tp, (g.N - n0) >> 5);
    __syncthreads();  // the previous pass's readers are done
    rows2_fill<MODE>(r2_lds, g.W, g.ldw, n0, tp, g.K);
    __syncthreads();
    const int plane = tp * KS * 64 * 16;
    // running BatchNorm sums of this wave's rows for the cell `cur` (flushed when the cell changes)
    float run1[kRows2MaxT], run2[kRows2MaxT];
    int cur = -1;
    if (STATS) {
#pragma unroll
      for (int t = 0; t < kRows2MaxT; ++t) run1[t] = run2[t] = 0.f;
    }
    auto flush = [&]() {
      if (cur < 0) return;
#pragma unroll
      for (int t = 0; t < kRows2MaxT; ++t)
        if (t < tp && kh == 0) {
          double* p = g.acc + ((size_t)cur * 2) * 1024 + n0 + 32 * t + i;
          atomicAdd(p, (double)run1[t]);
          atomicAdd(p + 1024, (double)run2[t]);
          run1[t] = run2[t] = 0.f;
        }
    };
    for (long m0 = row_lo; m0 < row_hi; m0 += 32) {
      const long arow = min(m0 + i, (long)g.M - 1);  // rows past the end repeat the last one (never stored, never counted)
      const float* ap = g.A + (size_t)arow * g.lda + 8 * kh;
      int acell = 0;
      if (AFUSE) acell = g.row_cell[arow];
      const float* mp = AFUSE ? g.a_mean + (size_t)acell * g.K + 8 * kh : nullptr;
      const float* rp = AFUSE ? g.a_rg + (size_t)acell * g.K + 8 * kh : nullptr;
      const float* bp = AFUSE ? g.a_beta + 8 * kh : nullptr;
      f32x16 acc[kRows2MaxT];
#pragma unroll
      for (int t = 0; t < kRows2MaxT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      float a[kRing][8];
      auto load_a = [&](int k0, float (&d)[8]) {
        const float4 x = *reinterpret_cast<const float4*>(ap + k0), y = *reinterpret_cast<const float4*>(ap + k0 + 4);
        d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w; d[4] = y.x; d[5] = y.y; d[6] = y.z; d[7] = y.w;
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
#pragma unroll
      for (int d = 0; d < kRing; ++d)
        if (d < KS) load_a(16 * d, a[d]);
      for (int kb = 0; kb < KS; kb += kRing) {
#pragma unroll
        for (int d = 0; d < kRing; ++d) {
          const int ks = kb + d;
          if (ks < KS) {
            gemm_bf16x8 ah, al;
            if (MODE == 2) gemm_split_bf16(a[d], ah, al);
            else if (MODE == 1) ah = gemm_to_bf16(a[d]);
#pragma unroll
            for (int t = 0; t < kRows2MaxT; ++t) {
              if (t < tp) {
                if (MODE == 0) {
                  const float4* f = reinterpret_cast<const float4*>(r2_lds) + ((size_t)(t * KS + ks) * 2) * 64 + lane;
                  const float4 x = f[0], y = f[64];
                  const float b[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
#pragma unroll
                  for (int j = 0; j < 8; ++j) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][j], b[j], acc[t], 0, 0, 0);
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
            }
            if (ks + kRing < KS) load_a(16 * (ks + kRing), a[d]);  // refill this slot with the step one ring ahead
          }
        }
      }
      // epilogue: bias, store, BatchNorm partial sums
      bool uniform = true;
      int c_first = 0;
      if (STATS) {
        c_first = g.row_cell[m0];
        const int c_last = g.row_cell[min(m0 + 31, (long)g.M - 1)];
        uniform = c_first == c_last;
        if (uniform && c_first != cur) {
          flush();
          cur = c_first;
        }
      }
#pragma unroll
      for (int t = 0; t < kRows2MaxT; ++t) {
        if (t < tp) {
          const int cg = n0 + 32 * t + i;
          const float bv = g.bias ? g.bias[cg] : 0.f;
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const long row = m0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (row < g.M) {
              const float v = acc[t][r] + bv;
              g.C[(size_t)row * g.ldc + cg] = v;
              if (STATS) {
                if (uniform) {
                  s1 += v;
                  s2 += v * v;
                } else {  // a tile that straddles two cells (one per cell boundary): element-wise
                  double* p = g.acc + ((size_t)g.row_cell[row] * 2) * 1024 + cg;
                  atomicAdd(p, (double)v);
                  atomicAdd(p + 1024, (double)v * (double)v);
                }
              }
            }
          }
          if (STATS && uniform) {
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            run1[t] += s1;
            run2[t] += s2;
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
struct Tn2Args {
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
constexpr int kTn2MaxUnits = 8;  // float4 units a thread stages per step: <= 256 rows x 64 or 64 rows x 256 or 32 rows x 384 floats / 4 / 512 threads

template <int MODE, bool XFUSE>
__global__ __launch_bounds__(kRows2Threads) void tn2_kernel(const Tn2Args g) {
  extern __shared__ __attribute__((aligned(16))) char t2_lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 31, kh = lane >> 5;
  const int n_base = 256 * blockIdx.y, k_base = 32 * g.kb_tiles * blockIdx.z;
  const int NB = min(256, g.N - n_base), KB = min(32 * g.kb_tiles, g.K - k_base), NT = NB >> 5, KT = KB >> 5, RG = 8 / NT;
  const int nt = w % NT, rg = w / NT;
  const int SR = RG * 32;                      // rows per step
  const int ldyl = NB + 4, ldxl = KB + 4;      // LDS row strides (floats)
  const int buf_floats = SR * (ldyl + ldxl);
  float* lds = reinterpret_cast<float*>(t2_lds);
  const long m_lo = (long)blockIdx.x * g.rows_per_wg, m_hi = min((long)g.M, m_lo + g.rows_per_wg);
  const int steps = m_hi > m_lo ? (int)((m_hi - m_lo + SR - 1) / SR) : 0;
  const int uy = SR * (NB >> 2), units = uy + SR * (KB >> 2);
  f32x16 acc[kTn2MaxT];
#pragma unroll
  for (int t = 0; t < kTn2MaxT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float csum = 0.f;
  const bool do_db = g.db && blockIdx.z == 0;
  float4 st[kTn2MaxUnits];
  auto load_regs = [&](int s) {
    const long m0 = m_lo + (long)s * SR;
#pragma unroll
    for (int q = 0; q < kTn2MaxUnits; ++q) {
      const int u = tid + q * kRows2Threads;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (u < units) {
        if (u < uy) {
          const int r = u / (NB >> 2), c = (u % (NB >> 2)) << 2;
          const long row = m0 + r;
          if (row < m_hi) v = *reinterpret_cast<const float4*>(g.dY + (size_t)row * g.ldy + n_base + c);
        } else {
          const int u2 = u - uy, r = u2 / (KB >> 2), c = (u2 % (KB >> 2)) << 2;
          const long row = m0 + r;
          if (row < m_hi) {
            v = *reinterpret_cast<const float4*>(g.X + (size_t)row * g.ldx + k_base + c);
            if (XFUSE) {
              const size_t o = (size_t)g.row_cell[row] * g.K + k_base + c;
              const float4 mm = *reinterpret_cast<const float4*>(g.x_mean + o), rr = *reinterpret_cast<const float4*>(g.x_rg + o),
                           bb = *reinterpret_cast<const float4*>(g.x_beta + k_base + c);
              v.x = fmaxf(__fmaf_rn(__fsub_rn(v.x, mm.x), rr.x, bb.x), 0.f);
              v.y = fmaxf(__fmaf_rn(__fsub_rn(v.y, mm.y), rr.y, bb.y), 0.f);
              v.z = fmaxf(__fmaf_rn(__fsub_rn(v.z, mm.z), rr.z, bb.z), 0.f);
              v.w = fmaxf(__fmaf_rn(__fsub_rn(v.w, mm.w), rr.w, bb.w), 0.f);
            }
          }
        }
      }
      st[q] = v;
    }
  };
  auto write_lds = [&](int b) {
    float* yb = lds + (size_t)b * buf_floats;
    float* xb = yb + SR * ldyl;
#pragma unroll
    for (int q = 0; q < kTn2MaxUnits; ++q) {
      const int u = tid + q * kRows2Threads;
      if (u < units) {
        if (u < uy) {
          const int r = u / (NB >> 2), c = (u % (NB >> 2)) << 2;
          *reinterpret_cast<float4*>(yb + r * ldyl + c) = st[q];
        } else {
          const int u2 = u - uy, r = u2 / (KB >> 2), c = (u2 % (KB >> 2)) << 2;
          *reinterpret_cast<float4*>(xb + r * ldxl + c) = st[q];
        }
      }
    }
  };
  if (steps > 0) {
    load_regs(0);
    write_lds(0);
  }
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    if (s + 1 < steps) load_regs(s + 1);
    const float* yb = lds + (size_t)(s & 1) * buf_floats;
    const float* xb = yb + SR * ldyl;
    if (do_db && tid < NB) {
      float c = 0.f;
      for (int r = 0; r < SR; ++r) c += yb[r * ldyl + tid];
      csum += c;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int rb = rg * 32 + 16 * h + 8 * kh;
      float a[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = yb[(rb + j) * ldyl + nt * 32 + i];
      gemm_bf16x8 ah, al;
      if (MODE == 2) gemm_split_bf16(a, ah, al);
      else if (MODE == 1) ah = gemm_to_bf16(a);
#pragma unroll
      for (int t = 0; t < kTn2MaxT; ++t) {
        if (t < KT) {
          float b[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) b[j] = xb[(rb + j) * ldxl + t * 32 + i];
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
    }
    if (s + 1 < steps) write_lds((s + 1) & 1);
    __syncthreads();
  }
  if (steps == 0) return;
#pragma unroll
  for (int t = 0; t < kTn2MaxT; ++t) {
    if (t < KT) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n_base + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (k_base + t * 32 + i < g.k_real) unsafeAtomicAdd(g.dW + (size_t)n * g.ldw + k_base + t * 32 + i, acc[t][r]);
      }
    }
  }
  if (do_db && tid < NB) unsafeAtomicAdd(g.db + n_base + tid, csum);
}

}  // namespace train
}  // namespace t2l
