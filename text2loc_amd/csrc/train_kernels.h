// Device kernels of the object-branch TRAINING step: forward in model.train() mode with saved
// activations, backward, Adam. gfx950 only. All arithmetic is f32 (the reference trains in f32); contractions run on
// v_mfma_f32_32x32x2_f32. The step is small (B=64 cells -> 1,792 tokens, ~12 GFLOP fwd+bwd) and latency-bound, so the
// kernels are modular (one per op of the autograd graph) rather than fused per cell like the eval encoder.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_f32.h"

namespace t2l {
namespace train {

constexpr int kTD = 256;    // embed dim
constexpr int kTS = 28;     // object slots (tokens) per cell
constexpr int kTH = 4;      // heads
constexpr int kTHd = 64;    // head dim
constexpr float kBnEps = 1e-5f, kLnEps = 1e-5f, kNormEps = 1e-12f;

// all-reduce sum over the 64 lanes on the VALU (DPP + v_permlane swaps): __shfl_xor lowers to ds_bpermute_b32 — six dependent
// LDS round trips per sum
template <int CTRL>
__device__ __forceinline__ float wsum_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wsum(float v) {
  v += wsum_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
  v += wsum_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
  v += wsum_dpp<0x141>(v);  // row_half_mirror
  v += wsum_dpp<0x140>(v);  // row_mirror
  {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  return v;
}

// Counter-based dropout: keep element `idx` of site `site` iff the top 24 bits of lowbias32(idx*0x9E3779B1 + key) >= thr,
// key = seed ^ site*0x85EBCA77, thr = p*2^24. oracle/t2l_oracle_train.py:dropout_keep is the same function.
__device__ __forceinline__ bool keep_bit(uint32_t key, uint32_t idx, uint32_t thr) {
  uint32_t x = idx * 0x9E3779B1u + key;
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return (x >> 8) >= thr;
}
struct Drop {
  uint32_t key, thr;
  float scale;  // 1/(1-p); thr == 0 -> identity
};

// ---------------------------------------------------------------------------------------------------------------
// first Linear of the small branches (K = 1 or 3 inputs -> 64): y[m][c] = b[c] + sum_k x[m][k] w[c][k]
// standardize: x = (n_pts - mean)/std evaluated in f32 exactly as the reference's tensor expression
// (models/object_encoder.py:141-144).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float small_in(const float* x, int m, int K, int k, int standardize, float mean, float stdv) {
  float v = x[(size_t)m * K + k];
  return standardize ? (v - mean) / stdv : v;
}
// (every kernel of the feature branches exists as a body + a plain launch + a MULTI launch: the two or three small branches —
//  position, point count, colour — are independent chains of identical shape, so each stage of all of them is ONE launch with the
//  branch in an extra grid dimension: a dependent launch costs ~5 us whatever it does, and the branches ran one after the other)
constexpr int kMaxJobs = 3;
__device__ __forceinline__ void smallk_fwd_body(const float* __restrict__ x, int M, int K, const float* __restrict__ w,
                                                const float* __restrict__ b, int standardize, float mean, float stdv,
                                                float* __restrict__ y, unsigned bx) {
  const int i = bx * blockDim.x + threadIdx.x;
  if (i >= M * 64) return;
  const int m = i >> 6, c = i & 63;
  float s = b[c];
  for (int k = 0; k < K; ++k) s += small_in(x, m, K, k, standardize, mean, stdv) * w[c * K + k];
  y[i] = s;
}
__global__ void smallk_fwd_kernel(const float* __restrict__ x, int M, int K, const float* __restrict__ w,
                                  const float* __restrict__ b, int standardize, float mean, float stdv,
                                  float* __restrict__ y) {
  smallk_fwd_body(x, M, K, w, b, standardize, mean, stdv, y, blockIdx.x);
}
struct SmallkJob {
  const float* x;
  int K, standardize;
  const float *w, *b;   // forward
  float* y;             // forward
  const float* dy;      // backward
  float *dW, *db;       // backward
};
struct SmallkMulti {
  SmallkJob j[kMaxJobs];
  int M, rows_per_block;
  float mean, stdv;
};
__global__ void smallk_fwd_multi_kernel(SmallkMulti m) {
  const SmallkJob& j = m.j[blockIdx.y];
  smallk_fwd_body(j.x, m.M, j.K, j.w, j.b, j.standardize, m.mean, m.stdv, j.y, blockIdx.x);
}
// dW[c][k] += sum_m dy[m][c] x[m][k]   grid = row chunks, 256 threads = 64 channels x 4 row lanes
__device__ __forceinline__ void smallk_bwd_body(const float* __restrict__ x, int M, int K, const float* __restrict__ dy, int standardize,
                                                float mean, float stdv, int rows_per_block, float* __restrict__ dW,
                                                float* __restrict__ db, unsigned bx) {
  __shared__ float red[4][256];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int lo = bx * rows_per_block, hi = min(M, lo + rows_per_block);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int m = lo + g; m < hi; m += 4) {
    const float d = dy[(size_t)m * 64 + c];
    s[3] += d;
    for (int k = 0; k < K; ++k) s[k] += d * small_in(x, m, K, k, standardize, mean, stdv);
  }
  for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = s[k];
  __syncthreads();
  if (g == 0) {
    for (int k = 0; k < K; ++k)
      unsafeAtomicAdd(dW + c * K + k, red[k][c] + red[k][c + 64] + red[k][c + 128] + red[k][c + 192]);
    unsafeAtomicAdd(db + c, red[3][c] + red[3][c + 64] + red[3][c + 128] + red[3][c + 192]);
  }
}
__global__ __launch_bounds__(256) void smallk_bwd_kernel(const float* __restrict__ x, int M, int K,
                                                         const float* __restrict__ dy, int standardize, float mean,
                                                         float stdv, int rows_per_block, float* __restrict__ dW,
                                                         float* __restrict__ db) {
  smallk_bwd_body(x, M, K, dy, standardize, mean, stdv, rows_per_block, dW, db, blockIdx.x);
}
__global__ __launch_bounds__(256) void smallk_bwd_multi_kernel(SmallkMulti m) {
  const SmallkJob& j = m.j[blockIdx.y];
  smallk_bwd_body(j.x, m.M, j.K, j.dy, j.standardize, m.mean, m.stdv, m.rows_per_block, j.dW, j.db, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm1d in training mode (+ ReLU). Two stages so that the whole chip takes part although there are only 64-1024
// channels: (1) per (64 channels x 64 rows) workgroup partial sums, float64 atomics into acc[2*C]; (2) elementwise apply.
// var = E[x^2] - mean^2 in float64 (inputs are f32; the cancellation is harmless there). Running statistics get the
// unbiased variance (momentum 0.1), as torch does.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kBnRows = 64;
// One accumulator slot: [2 * 1024] channel sums + the ROW COUNT at kBnCount (cross-rank BatchNorm only, `sync`: the slot is summed over
// the ranks between the statistics and the apply launch — t2l_train_sync_bn — and the apply side then divides by the global count).
constexpr int kBnCount = 2048;
constexpr int kBnStride = 2056;
// MODE 0: acc[c] += sum y, acc[C+c] += sum y^2.  MODE 1: dv = out>0 ? d : 0; acc[c] += sum dv, acc[C+c] += sum dv*xhat
template <int MODE>
__device__ __forceinline__ void bn_stats_body(const float* __restrict__ y, const float* __restrict__ d, const float* __restrict__ out, int M,
                                              int C, const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
                                              double* __restrict__ acc, unsigned bx, unsigned by, int sync = 0,
                                              float* __restrict__ dgamma = nullptr, float* __restrict__ dbeta = nullptr) {
  __shared__ float r1[256], r2[256];
  const int c = bx * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  const int lo = by * kBnRows, hi = min(M, lo + kBnRows);
  float mean = 0.f, rstd = 0.f;
  if (MODE == 1) {
    mean = save_mean[c];
    rstd = save_rstd[c];
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll 8  // 16 rows per thread: the loads of eight of them in flight (each row's loads otherwise wait for the previous row's)
  for (int m = lo + g; m < hi; m += 4) {
    const size_t i = (size_t)m * C + c;
    if (MODE == 0) {
      const float v = y[i];
      s1 += v;
      s2 += v * v;
    } else {
      const float dv = out[i] > 0.f ? d[i] : 0.f;
      s1 += dv;
      s2 += dv * (y[i] - mean) * rstd;
    }
  }
  r1[threadIdx.x] = s1;
  r2[threadIdx.x] = s2;
  __syncthreads();
  if (g == 0) {
    const int t = threadIdx.x;
    const double t1 = (double)r1[t] + (double)r1[t + 64] + (double)r1[t + 128] + (double)r1[t + 192];
    const double t2 = (double)r2[t] + (double)r2[t + 64] + (double)r2[t + 128] + (double)r2[t + 192];
    atomicAdd(acc + c, t1);
    atomicAdd(acc + C + c, t2);
    if (sync) {
      // cross-rank statistics: the slot will hold GLOBAL sums when the apply launch reads it, and the parameter gradients are this
      // rank's own sums (the gradient all_reduce adds the ranks' up) — so they are taken here, not in the apply launch
      if (MODE == 1) {
        unsafeAtomicAdd(dgamma + c, (float)t2);
        unsafeAtomicAdd(dbeta + c, (float)t1);
      }
      if (bx == 0 && by == 0 && t == 0) atomicAdd(acc + kBnCount, (double)M);
    }
  }
}
template <int MODE>
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ y, const float* __restrict__ d,
                                                       const float* __restrict__ out, int M, int C,
                                                       const float* __restrict__ save_mean,
                                                       const float* __restrict__ save_rstd, double* __restrict__ acc, int sync,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
  bn_stats_body<MODE>(y, d, out, M, C, save_mean, save_rstd, acc, blockIdx.x, blockIdx.y, sync, dgamma, dbeta);
}
// one BatchNorm layer of one branch: everything the statistics / apply kernels of either direction need
struct BnJob {
  const float* y;       // pre-BatchNorm Linear output
  float* out;           // forward: post-ReLU output (written); backward: the same (read: ReLU mask)
  float* d;             // backward: gradient w.r.t. out, overwritten with the gradient w.r.t. y
  double* acc;
  const float *gamma, *beta;
  float *run_mean, *run_var, *save_mean, *save_rstd, *dgamma, *dbeta;
};
struct BnMulti {
  BnJob j[kMaxJobs];
  int M, C;
  float momentum;
  int sync;
};
template <int MODE>
__global__ __launch_bounds__(256) void bn_stats_multi_kernel(BnMulti m) {
  const BnJob& j = m.j[blockIdx.z];
  bn_stats_body<MODE>(j.y, j.d, j.out, m.M, m.C, j.save_mean, j.save_rstd, j.acc, blockIdx.x, blockIdx.y, m.sync, j.dgamma, j.dbeta);
}
__device__ __forceinline__ void bn_apply_fwd_body(const float* __restrict__ y, int M, int C, const double* __restrict__ acc,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  float* __restrict__ run_mean, float* __restrict__ run_var, float momentum,
                                                  float* __restrict__ out, float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                  unsigned bx, int sync = 0) {
  const size_t i = (size_t)bx * 256 + threadIdx.x;
  const double Mg = sync ? acc[kBnCount] : (double)M;  // rows behind the sums: this rank's, or every rank's
  if (i < (size_t)M * C) {
    const int c = (int)(i % C);
    const double mean = acc[c] / Mg;
    const double var = fmax(acc[C + c] / Mg - mean * mean, 0.0);
    const float rstd = 1.0f / sqrtf((float)var + kBnEps);
    out[i] = fmaxf((y[i] - (float)mean) * rstd * gamma[c] + beta[c], 0.f);
  }
  if (bx == 0)
    for (int c = threadIdx.x; c < C; c += 256) {
      const double mean = acc[c] / Mg;
      const double var = fmax(acc[C + c] / Mg - mean * mean, 0.0);
      save_mean[c] = (float)mean;
      save_rstd[c] = 1.0f / sqrtf((float)var + kBnEps);
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)(var * (Mg / fmax(Mg - 1.0, 1.0)));
    }
}
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const float* __restrict__ y, int M, int C,
                                                           const double* __restrict__ acc,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ run_mean, float* __restrict__ run_var,
                                                           float momentum, float* __restrict__ out,
                                                           float* __restrict__ save_mean, float* __restrict__ save_rstd, int sync) {
  bn_apply_fwd_body(y, M, C, acc, gamma, beta, run_mean, run_var, momentum, out, save_mean, save_rstd, blockIdx.x, sync);
}
__global__ __launch_bounds__(256) void bn_apply_fwd_multi_kernel(BnMulti m) {
  const BnJob& j = m.j[blockIdx.y];
  bn_apply_fwd_body(j.y, m.M, m.C, j.acc, j.gamma, j.beta, j.run_mean, j.run_var, m.momentum, j.out, j.save_mean, j.save_rstd, blockIdx.x, m.sync);
}
// d: gradient w.r.t. the ReLU output (in), overwritten with the gradient w.r.t. the Linear output y.
__device__ __forceinline__ void bn_apply_bwd_body(float* __restrict__ d, const float* __restrict__ out, const float* __restrict__ y, int M,
                                                  int C, const double* __restrict__ acc, const float* __restrict__ gamma,
                                                  const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
                                                  float* __restrict__ dgamma, float* __restrict__ dbeta, unsigned bx, int sync = 0) {
  const size_t i = (size_t)bx * 256 + threadIdx.x;
  if (i < (size_t)M * C) {
    const int c = (int)(i % C);
    const float Mg = sync ? (float)acc[kBnCount] : (float)M;
    const float rstd = save_rstd[c], s1 = (float)acc[c], s2 = (float)acc[C + c];
    const float dv = out[i] > 0.f ? d[i] : 0.f;
    d[i] = gamma[c] * rstd / Mg * (Mg * dv - s1 - (y[i] - save_mean[c]) * rstd * s2);
  }
  if (bx == 0 && !sync)  // (sync: the statistics launch took this rank's own sums)
    for (int c = threadIdx.x; c < C; c += 256) {
      unsafeAtomicAdd(dgamma + c, (float)acc[C + c]);
      unsafeAtomicAdd(dbeta + c, (float)acc[c]);
    }
}
__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(float* __restrict__ d, const float* __restrict__ out,
                                                           const float* __restrict__ y, int M, int C,
                                                           const double* __restrict__ acc, const float* __restrict__ gamma,
                                                           const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_rstd, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int sync) {
  bn_apply_bwd_body(d, out, y, M, C, acc, gamma, save_mean, save_rstd, dgamma, dbeta, blockIdx.x, sync);
}
__global__ __launch_bounds__(256) void bn_apply_bwd_multi_kernel(BnMulti m) {
  const BnJob& j = m.j[blockIdx.y];
  bn_apply_bwd_body(j.d, j.out, j.y, m.M, m.C, j.acc, j.gamma, j.save_mean, j.save_rstd, j.dgamma, j.dbeta, blockIdx.x, m.sync);
}

// ---------------------------------------------------------------------------------------------------------------
// F.normalize over 256-wide rows: one wave per row, 4 floats per lane.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 norm_fwd(float4 v, float& n) {
  n = fmaxf(sqrtf(wsum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w)), kNormEps);
  return make_float4(v.x / n, v.y / n, v.z / n, v.w / n);
}
__device__ __forceinline__ float4 norm_bwd(float4 dy, float4 y, float n) {
  if (n <= kNormEps) return make_float4(dy.x / kNormEps, dy.y / kNormEps, dy.z / kNormEps, dy.w / kNormEps);
  const float t = wsum(dy.x * y.x + dy.y * y.y + dy.z * y.z + dy.w * y.w);
  return make_float4((dy.x - y.x * t) / n, (dy.y - y.y * t) / n, (dy.z - y.z * t) / n, (dy.w - y.w * t) / n);
}
// src row = table[idx[m]] when idx != nullptr (embedding lookup) else src[m]; dst row stride ldd (a 256-wide slot of cat)
__device__ __forceinline__ void rownorm_fwd_body(const float* __restrict__ src, const int32_t* __restrict__ idx, int M, float* __restrict__ dst,
                                                 int ldd, float* __restrict__ save_n, unsigned bx) {
  const int m = (bx * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (m >= M) return;
  const size_t row = idx ? (size_t)idx[m] : (size_t)m;
  const float4 v = *reinterpret_cast<const float4*>(src + row * kTD + lane * 4);
  float n;
  const float4 y = norm_fwd(v, n);
  *reinterpret_cast<float4*>(dst + (size_t)m * ldd + lane * 4) = y;
  if (lane == 0) save_n[m] = n;
}
__global__ void rownorm_fwd_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, int M,
                                   float* __restrict__ dst, int ldd, float* __restrict__ save_n) {
  rownorm_fwd_body(src, idx, M, dst, ldd, save_n, blockIdx.x);
}
struct RownormJob {
  const float* src;      // forward: rows to normalise (or the embedding table); backward: dy (a slot of dcat)
  const int32_t* idx;    // forward: embedding lookup or nullptr
  float* dst;            // forward: slot of cat; backward: dx [M][256]
  const float* y;        // backward: the normalised rows (slot of cat)
  float* save_n;
};
struct RownormMulti {
  RownormJob j[kMaxJobs];
  int M, ld;
};
__global__ void rownorm_fwd_multi_kernel(RownormMulti m) {
  const RownormJob& j = m.j[blockIdx.y];
  rownorm_fwd_body(j.src, j.idx, m.M, j.dst, m.ld, j.save_n, blockIdx.x);
}
// dx[m] = normalize_bwd(dy[m], y[m], n[m])
__device__ __forceinline__ void rownorm_bwd_body(const float* __restrict__ dy, const float* __restrict__ y, int ld,
                                                 const float* __restrict__ save_n, int M, float* __restrict__ dx, unsigned bx) {
  const int m = (bx * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (m >= M) return;
  const float4 d = *reinterpret_cast<const float4*>(dy + (size_t)m * ld + lane * 4);
  const float4 yy = *reinterpret_cast<const float4*>(y + (size_t)m * ld + lane * 4);
  *reinterpret_cast<float4*>(dx + (size_t)m * kTD + lane * 4) = norm_bwd(d, yy, save_n[m]);
}
__global__ void rownorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, int ld,
                                   const float* __restrict__ save_n, int M, float* __restrict__ dx) {
  rownorm_bwd_body(dy, y, ld, save_n, M, dx, blockIdx.x);
}
__global__ void rownorm_bwd_multi_kernel(RownormMulti m) {
  const RownormJob& j = m.j[blockIdx.y];
  rownorm_bwd_body(j.src, j.y, m.ld, j.save_n, m.M, j.dst, blockIdx.x);
}
// Embedding-table gradient, stage 2 (stage 1 = rownorm_bwd_kernel writing g[M,256] = normalize_bwd per object):
// dtable[r] += sum over the objects with idx == r of g[m]. grid (table rows - 1, kEmbSplit): row 0 = padding_idx never
// receives gradient (models/object_encoder.py:33,37). Each workgroup lists in LDS the matching objects with
// m % kEmbSplit == blockIdx.y, its 4 waves add their rows (independent coalesced 1 KiB loads), then 256 float atomics.
constexpr int kEmbSplit = 8;
__device__ __forceinline__ void embed_sum_body(const float* __restrict__ g, const int32_t* __restrict__ idx, int M, float* __restrict__ dtable,
                                               unsigned bx, unsigned by) {
  __shared__ float4 red[256];
  __shared__ int list[2048];
  __shared__ int cnt;
  const int r = bx + 1, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int base = 0; base < M; base += 2048) {
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    for (int m = base + threadIdx.x; m < min(M, base + 2048); m += 256)
      if (idx[m] == r && (m % kEmbSplit) == (int)by) list[atomicAdd(&cnt, 1)] = m;  // this workgroup's share
    __syncthreads();
    const int n = cnt;
#pragma unroll 4
    for (int i = w; i < n; i += 4) {
      const float4 v = *reinterpret_cast<const float4*>(g + (size_t)list[i] * kTD + lane * 4);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    __syncthreads();
  }
  red[threadIdx.x] = a;
  __syncthreads();
  if (w == 0) {
    for (int i = 1; i < 4; ++i) {
      const float4 b = red[lane + 64 * i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float* t = dtable + (size_t)r * kTD + lane * 4;
    unsafeAtomicAdd(t + 0, a.x); unsafeAtomicAdd(t + 1, a.y); unsafeAtomicAdd(t + 2, a.z); unsafeAtomicAdd(t + 3, a.w);
  }
}
__global__ __launch_bounds__(256) void embed_sum_kernel(const float* __restrict__ g, const int32_t* __restrict__ idx, int M,
                                                        float* __restrict__ dtable) {
  embed_sum_body(g, idx, M, dtable, blockIdx.x, blockIdx.y);
}
struct EmbedSumMulti {
  const float* g[kMaxJobs];
  const int32_t* idx[kMaxJobs];
  float* dtable[kMaxJobs];
  int rows[kMaxJobs];  // table rows (row 0 = padding_idx receives nothing); grid.x = max(rows) - 1
  int M;
};
__global__ __launch_bounds__(256) void embed_sum_multi_kernel(EmbedSumMulti m) {
  const int j = blockIdx.z;
  if ((int)blockIdx.x + 1 >= m.rows[j]) return;  // (workgroup-uniform: the barriers inside are not reached by anyone)
  embed_sum_body(m.g[j], m.idx[j], m.M, m.dtable[j], blockIdx.x, blockIdx.y);
}
// tokens: X0[b*28+s] = normalize(feats[offsets[b]+s]) for s < min(count,28), zeros otherwise (cell_retrieval.py:85-98)
__global__ void scatter_norm_fwd_kernel(const float* __restrict__ feats, const int32_t* __restrict__ offsets, int B,
                                        float* __restrict__ X0, float* __restrict__ save_n) {
  const int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (t >= B * kTS) return;
  const int b = t / kTS, s = t - b * kTS;
  const int lo = offsets[b], cnt = offsets[b + 1] - lo;
  float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
  if (s < cnt) {
    float n;
    y = norm_fwd(*reinterpret_cast<const float4*>(feats + (size_t)(lo + s) * kTD + lane * 4), n);
    if (lane == 0) save_n[lo + s] = n;
  }
  *reinterpret_cast<float4*>(X0 + (size_t)t * kTD + lane * 4) = y;
}
// dfeats[o] = normalize_bwd(dX0[token of o]) for the kept objects, 0 for objects beyond slot 27
__global__ void scatter_norm_bwd_kernel(const float* __restrict__ dX0, const float* __restrict__ X0,
                                        const float* __restrict__ save_n, const int32_t* __restrict__ offsets, int B,
                                        int M, float* __restrict__ dfeats) {
  const int o = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (o >= M) return;
  int lo = 0, hi = B;  // largest b with offsets[b] <= o
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (offsets[mid] <= o) lo = mid; else hi = mid;
  }
  const int s = o - offsets[lo];
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  if (s < kTS) {
    const size_t t = (size_t)lo * kTS + s;
    g = norm_bwd(*reinterpret_cast<const float4*>(dX0 + t * kTD + lane * 4),
                 *reinterpret_cast<const float4*>(X0 + t * kTD + lane * 4), save_n[o]);
  }
  *reinterpret_cast<float4*>(dfeats + (size_t)o * kTD + lane * 4) = g;
}

// ---------------------------------------------------------------------------------------------------------------
// self-attention over the 28 slots of one cell, one workgroup per (cell, head). No padding mask (cell_retrieval.py:102).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kAS = kTHd + 1;  // LDS row stride of q/k/v tiles
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ P,
                                                       float* __restrict__ O, Drop dr) {
  __shared__ float q[kTS * kAS], k[kTS * kAS], v[kTS * kAS], p[kTS * (kTS + 1)];
  const int b = blockIdx.x >> 2, h = blockIdx.x & 3, tid = threadIdx.x;
#pragma unroll  // (7 trips: every global load of the tile issued before the first LDS write waits for one)
  for (int i = tid; i < kTS * kTHd; i += 256) {
    const int s = i >> 6, d = i & 63;
    const float* base = qkv + (size_t)(b * kTS + s) * (3 * kTD) + h * kTHd + d;
    q[s * kAS + d] = base[0];
    k[s * kAS + d] = base[kTD];
    v[s * kAS + d] = base[2 * kTD];
  }
  __syncthreads();
  for (int e = tid; e < kTS * kTS; e += 256) {
    const int i = e / kTS, j = e - i * kTS;
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < kTHd; ++d) s += q[i * kAS + d] * k[j * kAS + d];
    p[i * (kTS + 1) + j] = s * 0.125f;
  }
  __syncthreads();
  if (tid < kTS) {
    float* row = p + tid * (kTS + 1);
    float mx = row[0];
    for (int j = 1; j < kTS; ++j) mx = fmaxf(mx, row[j]);
    float sum = 0.f;
    for (int j = 0; j < kTS; ++j) {
      row[j] = expf(row[j] - mx);
      sum += row[j];
    }
    for (int j = 0; j < kTS; ++j) row[j] /= sum;
  }
  __syncthreads();
  const size_t pbase = (size_t)blockIdx.x * kTS * kTS;
  for (int e = tid; e < kTS * kTS; e += 256) {
    const int i = e / kTS, j = e - i * kTS;
    float pv = p[i * (kTS + 1) + j];
    P[pbase + e] = pv;  // probabilities BEFORE dropout (softmax backward needs them)
    if (dr.thr) pv = keep_bit(dr.key, (uint32_t)(pbase + e), dr.thr) ? pv * dr.scale : 0.f;
    p[i * (kTS + 1) + j] = pv;
  }
  __syncthreads();
  for (int e = tid; e < kTS * kTHd; e += 256) {
    const int i = e >> 6, d = e & 63;
    float s = 0.f;
    for (int j = 0; j < kTS; ++j) s += p[i * (kTS + 1) + j] * v[j * kAS + d];
    O[(size_t)(b * kTS + i) * kTD + h * kTHd + d] = s;
  }
}
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ P,
                                                       const float* __restrict__ dO, float* __restrict__ dqkv, Drop dr) {
  __shared__ float q[kTS * kAS], k[kTS * kAS], v[kTS * kAS], go[kTS * kAS];
  __shared__ float p[kTS * (kTS + 1)], pd[kTS * (kTS + 1)], ds[kTS * (kTS + 1)];
  const int b = blockIdx.x >> 2, h = blockIdx.x & 3, tid = threadIdx.x;
#pragma unroll  // (7 trips: every global load of the tile issued before the first LDS write waits for one)
  for (int i = tid; i < kTS * kTHd; i += 256) {
    const int s = i >> 6, d = i & 63;
    const float* base = qkv + (size_t)(b * kTS + s) * (3 * kTD) + h * kTHd + d;
    q[s * kAS + d] = base[0];
    k[s * kAS + d] = base[kTD];
    v[s * kAS + d] = base[2 * kTD];
    go[s * kAS + d] = dO[(size_t)(b * kTS + s) * kTD + h * kTHd + d];
  }
  const size_t pbase = (size_t)blockIdx.x * kTS * kTS;
  for (int e = tid; e < kTS * kTS; e += 256) {
    const int i = e / kTS, j = e - i * kTS;
    const float pv = P[pbase + e];
    const float m = dr.thr ? (keep_bit(dr.key, (uint32_t)(pbase + e), dr.thr) ? dr.scale : 0.f) : 1.f;
    p[i * (kTS + 1) + j] = pv;
    pd[i * (kTS + 1) + j] = pv * m;
    ds[i * (kTS + 1) + j] = m;  // mask factor for now
  }
  __syncthreads();
  // dV[j][d] = sum_i Pd[i][j] dO[i][d]
  for (int e = tid; e < kTS * kTHd; e += 256) {
    const int j = e >> 6, d = e & 63;
    float s = 0.f;
    for (int i = 0; i < kTS; ++i) s += pd[i * (kTS + 1) + j] * go[i * kAS + d];
    dqkv[(size_t)(b * kTS + j) * (3 * kTD) + 2 * kTD + h * kTHd + d] = s;
  }
  // dP[i][j] = mask * sum_d dO[i][d] V[j][d]
  for (int e = tid; e < kTS * kTS; e += 256) {
    const int i = e / kTS, j = e - i * kTS;
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < kTHd; ++d) s += go[i * kAS + d] * v[j * kAS + d];
    ds[i * (kTS + 1) + j] *= s;
  }
  __syncthreads();
  if (tid < kTS) {  // dS = P * (dP - sum_j dP*P) / sqrt(hd)
    float t = 0.f;
    for (int j = 0; j < kTS; ++j) t += ds[tid * (kTS + 1) + j] * p[tid * (kTS + 1) + j];
    for (int j = 0; j < kTS; ++j)
      ds[tid * (kTS + 1) + j] = p[tid * (kTS + 1) + j] * (ds[tid * (kTS + 1) + j] - t) * 0.125f;
  }
  __syncthreads();
  for (int e = tid; e < kTS * kTHd; e += 256) {
    const int i = e >> 6, d = e & 63;
    float sq = 0.f, sk = 0.f;
    for (int j = 0; j < kTS; ++j) {
      sq += ds[i * (kTS + 1) + j] * k[j * kAS + d];  // dQ[i] = sum_j dS[i][j] K[j]
      sk += ds[j * (kTS + 1) + i] * q[j * kAS + d];  // dK[i] = sum_j dS[j][i] Q[j]
    }
    float* base = dqkv + (size_t)(b * kTS + i) * (3 * kTD) + h * kTHd + d;
    base[0] = sq;
    base[kTD] = sk;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// out = LayerNorm(x + dropout(y)); one wave per token. Saves xhat and rstd.
// ---------------------------------------------------------------------------------------------------------------
__global__ void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, int T,
                              const float* __restrict__ gamma, const float* __restrict__ beta, Drop dr,
                              float* __restrict__ out, float* __restrict__ xhat, float* __restrict__ save_rstd) {
  const int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (t >= T) return;
  const size_t o = (size_t)t * kTD + lane * 4;
  float4 a = *reinterpret_cast<const float4*>(x + o);
  float4 f = *reinterpret_cast<const float4*>(y + o);
  if (dr.thr) {
    f.x = keep_bit(dr.key, (uint32_t)o + 0, dr.thr) ? f.x * dr.scale : 0.f;
    f.y = keep_bit(dr.key, (uint32_t)o + 1, dr.thr) ? f.y * dr.scale : 0.f;
    f.z = keep_bit(dr.key, (uint32_t)o + 2, dr.thr) ? f.z * dr.scale : 0.f;
    f.w = keep_bit(dr.key, (uint32_t)o + 3, dr.thr) ? f.w * dr.scale : 0.f;
  }
  a.x += f.x; a.y += f.y; a.z += f.z; a.w += f.w;
  const float mu = wsum(a.x + a.y + a.z + a.w) * (1.f / kTD);
  a.x -= mu; a.y -= mu; a.z -= mu; a.w -= mu;
  const float var = wsum(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w) * (1.f / kTD);
  const float rstd = 1.0f / sqrtf(var + kLnEps);
  const float4 h = make_float4(a.x * rstd, a.y * rstd, a.z * rstd, a.w * rstd);
  const float4 g = *reinterpret_cast<const float4*>(gamma + lane * 4);
  const float4 be = *reinterpret_cast<const float4*>(beta + lane * 4);
  *reinterpret_cast<float4*>(xhat + o) = h;
  *reinterpret_cast<float4*>(out + o) = make_float4(h.x * g.x + be.x, h.y * g.y + be.y, h.z * g.z + be.z, h.w * g.w + be.w);
  if (lane == 0) save_rstd[t] = rstd;
}
// dz = LN backward of dout; d_res = dz (gradient of the residual input), d_y = dz * dropout mask (gradient of y).
// dgamma/dbeta: per-workgroup column partials, then atomics. grid = any; each workgroup strides over tokens.
// Workgroups of 16 waves: the dgamma / dbeta atomics of a workgroup land on the same 512 addresses as everybody else's (~75 ns per
// same-address atomic: 64 workgroups of 4 waves spent ~5 of their 10.7 us there, 128 workgroups took 15.5 us), so the rows are
// spread over MORE WAVES per workgroup instead of more workgroups.
__global__ __launch_bounds__(1024) void ln_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ xhat,
                                                     const float* __restrict__ save_rstd, int T,
                                                     const float* __restrict__ gamma, Drop dr, float* __restrict__ d_res,
                                                     float* __restrict__ d_y, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta) {
  __shared__ float4 rg[1024], rb[1024];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const float4 g = *reinterpret_cast<const float4*>(gamma + lane * 4);
  float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag;
  // a wave walks ~7 token rows: the next row's operands are loaded before the current row is reduced and stored (written as
  // load -> reduce -> store per row, every row paid its own L2 round trip behind the previous row's stores: 11.4 us per launch)
  const int stride = gridDim.x * nw;
  int t = blockIdx.x * nw + w;
  float4 d = make_float4(0.f, 0.f, 0.f, 0.f), h = d;
  float rstd = 0.f;
  if (t < T) {
    const size_t o = (size_t)t * kTD + lane * 4;
    d = *reinterpret_cast<const float4*>(dout + o);
    h = *reinterpret_cast<const float4*>(xhat + o);
    rstd = save_rstd[t];
  }
  while (t < T) {
    const int tn = min(t + stride, T - 1);  // (clamped: the last prefetch re-reads a valid row and is dropped)
    const size_t on = (size_t)tn * kTD + lane * 4;
    const float4 nd = *reinterpret_cast<const float4*>(dout + on);
    const float4 nh = *reinterpret_cast<const float4*>(xhat + on);
    const float nrstd = save_rstd[tn];
    const size_t o = (size_t)t * kTD + lane * 4;
    ag.x += d.x * h.x; ag.y += d.y * h.y; ag.z += d.z * h.z; ag.w += d.w * h.w;
    ab.x += d.x; ab.y += d.y; ab.z += d.z; ab.w += d.w;
    const float4 dh = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
    const float m1 = wsum(dh.x + dh.y + dh.z + dh.w) * (1.f / kTD);
    const float m2 = wsum(dh.x * h.x + dh.y * h.y + dh.z * h.z + dh.w * h.w) * (1.f / kTD);
    float4 dz = make_float4(rstd * (dh.x - m1 - h.x * m2), rstd * (dh.y - m1 - h.y * m2), rstd * (dh.z - m1 - h.z * m2),
                            rstd * (dh.w - m1 - h.w * m2));
    *reinterpret_cast<float4*>(d_res + o) = dz;
    if (dr.thr) {
      dz.x = keep_bit(dr.key, (uint32_t)o + 0, dr.thr) ? dz.x * dr.scale : 0.f;
      dz.y = keep_bit(dr.key, (uint32_t)o + 1, dr.thr) ? dz.y * dr.scale : 0.f;
      dz.z = keep_bit(dr.key, (uint32_t)o + 2, dr.thr) ? dz.z * dr.scale : 0.f;
      dz.w = keep_bit(dr.key, (uint32_t)o + 3, dr.thr) ? dz.w * dr.scale : 0.f;
    }
    *reinterpret_cast<float4*>(d_y + o) = dz;
    d = nd;
    h = nh;
    rstd = nrstd;
    t += stride;
  }
  rg[threadIdx.x] = ag;
  rb[threadIdx.x] = ab;
  __syncthreads();
  if (w == 0) {
    for (int i = 1; i < nw; ++i) {
      const float4 a = rg[lane + 64 * i], c = rb[lane + 64 * i];
      ag.x += a.x; ag.y += a.y; ag.z += a.z; ag.w += a.w;
      ab.x += c.x; ab.y += c.y; ab.z += c.z; ab.w += c.w;
    }
    float* pg = dgamma + lane * 4;
    float* pb = dbeta + lane * 4;
    unsafeAtomicAdd(pg + 0, ag.x); unsafeAtomicAdd(pg + 1, ag.y); unsafeAtomicAdd(pg + 2, ag.z); unsafeAtomicAdd(pg + 3, ag.w);
    unsafeAtomicAdd(pb + 0, ab.x); unsafeAtomicAdd(pb + 1, ab.y); unsafeAtomicAdd(pb + 2, ab.z); unsafeAtomicAdd(pb + 3, ab.w);
  }
}

// feed-forward hidden dropout: hd = h * mask (forward); dh = dhd * mask * (h > 0) (backward, in place)
__global__ void drop_fwd_kernel(const float* __restrict__ h, size_t n, Drop dr, float* __restrict__ hd) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) hd[i] = keep_bit(dr.key, (uint32_t)i, dr.thr) ? h[i] * dr.scale : 0.f;
}
__global__ void relu_drop_bwd_kernel(float* __restrict__ d, const float* __restrict__ h, size_t n, Drop dr) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = h[i] > 0.f ? d[i] : 0.f;
  if (dr.thr) v = keep_bit(dr.key, (uint32_t)i, dr.thr) ? v * dr.scale : 0.f;
  d[i] = v;
}

// ---------------------------------------------------------------------------------------------------------------
// max over the 28 slots (first maximal slot wins) + F.normalize: one workgroup (256 threads = columns) per cell
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wsum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void pool_norm_fwd_kernel(const float* __restrict__ X, float* __restrict__ out,
                                                            int32_t* __restrict__ arg, float* __restrict__ save_n,
                                                            float* __restrict__ out2,  // out2: the caller's copy (no memcpy launch)
                                                            double* __restrict__ zero, int zero_n) {
  __shared__ float red[4];
  const int b = blockIdx.x, c = threadIdx.x;
  // the forward's last launch clears the BatchNorm accumulators of this step's backward and of the next forward (every
  // forward-side reader is behind it on the stream): no memset launch in the step
  for (int i = b * 256 + c; i < zero_n; i += gridDim.x * 256) zero[i] = 0.0;
  float mx = X[(size_t)b * kTS * kTD + c];
  int am = 0;
#pragma unroll
  for (int s = 1; s < kTS; ++s) {
    const float v = X[((size_t)b * kTS + s) * kTD + c];
    if (v > mx) { mx = v; am = s; }
  }
  const float n = fmaxf(sqrtf(block_sum256(mx * mx, red)), kNormEps);
  out[(size_t)b * kTD + c] = mx / n;
  out2[(size_t)b * kTD + c] = mx / n;
  arg[(size_t)b * kTD + c] = am;
  if (c == 0) save_n[b] = n;
}
__global__ __launch_bounds__(256) void pool_norm_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ out,
                                                            const int32_t* __restrict__ arg,
                                                            const float* __restrict__ save_n, float* __restrict__ dX) {
  __shared__ float red[4];
  const int b = blockIdx.x, c = threadIdx.x;
  const float g = gout[(size_t)b * kTD + c], y = out[(size_t)b * kTD + c], n = save_n[b];
  const float t = block_sum256(g * y, red);
  const float d = n <= kNormEps ? g / kNormEps : (g - y * t) / n;
  const int am = arg[(size_t)b * kTD + c];
  for (int s = 0; s < kTS; ++s) dX[((size_t)b * kTS + s) * kTD + c] = (s == am) ? d : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------
// torch.optim.Adam (defaults: no weight decay, no amsgrad) over all bound tensors in ONE launch:
// chunk table entry = (tensor, first element); 1,024 elements per workgroup.
// ---------------------------------------------------------------------------------------------------------------
struct AdamTensor {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t numel;
};
struct AdamChunk {
  int32_t tensor;
  int32_t first;  // element offset / 1024
};
__global__ __launch_bounds__(256) void adam_kernel(const AdamTensor* __restrict__ ts, const AdamChunk* __restrict__ cs,
                                                   float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt) {
  const AdamChunk c = cs[blockIdx.x];
  const AdamTensor t = ts[c.tensor];
  const int64_t base = (int64_t)c.first * 1024;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t i = base + j * 256 + threadIdx.x;
    if (i < t.numel) {
      const float g = t.g[i];
      const float m = b1 * t.m[i] + (1.f - b1) * g;
      const float v = b2 * t.v[i] + (1.f - b2) * g * g;
      t.m[i] = m;
      t.v[i] = v;
      t.p[i] -= (lr / bc1) * m / (sqrtf(v) / bc2_sqrt + eps);
    }
  }
}
__global__ void zero_kernel(const AdamTensor* __restrict__ ts, const AdamChunk* __restrict__ cs) {
  const AdamChunk c = cs[blockIdx.x];
  const AdamTensor t = ts[c.tensor];
  float* g = const_cast<float*>(t.g);
  const int64_t base = (int64_t)c.first * 1024;
  for (int j = 0; j < 4; ++j) {
    const int64_t i = base + j * 256 + threadIdx.x;
    if (i < t.numel) g[i] = 0.f;
  }
}

// ===============================================================================================================
// Generic forms for the TEXT head's training step (train.hip: text_train_*): the same ops at the head's shapes —
// d_model 1024 / head_dim 256 over L <= 32 tokens per sentence, and d_model 256 / head_dim 64 over S <= 32 sentences
// per description (models/language_encoder.py:97-101,127-147).
// ===============================================================================================================
// self-attention over the S rows of one group (sentence / description), one workgroup per (group, head); qkv [rows][3 D],
// D = 4 HD. Dynamic LDS: q, k, v tiles [S][HD + 1] + p [S][S + 1] (forward), + go and two more [S][S + 1] (backward).
template <int HD>
__global__ __launch_bounds__(256) void attn_fwd_g_kernel(const float* __restrict__ qkv, float* __restrict__ P, float* __restrict__ O,
                                                         int S, Drop dr) {
  extern __shared__ float sm[];
  constexpr int AS = HD + 1, D = 4 * HD;
  float *q = sm, *k = q + S * AS, *v = k + S * AS, *p = v + S * AS;
  const int b = blockIdx.x >> 2, h = blockIdx.x & 3, tid = threadIdx.x, SP = S + 1;
  const float scale = 1.0f / sqrtf((float)HD);
  for (int i = tid; i < S * HD; i += 256) {
    const int s = i / HD, d = i - s * HD;
    const float* base = qkv + (size_t)(b * S + s) * (3 * D) + h * HD + d;
    q[s * AS + d] = base[0];
    k[s * AS + d] = base[D];
    v[s * AS + d] = base[2 * D];
  }
  __syncthreads();
  for (int e = tid; e < S * S; e += 256) {
    const int i = e / S, j = e - i * S;
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < HD; ++d) s += q[i * AS + d] * k[j * AS + d];
    p[i * SP + j] = s * scale;
  }
  __syncthreads();
  if (tid < S) {
    float* row = p + tid * SP;
    float mx = row[0];
    for (int j = 1; j < S; ++j) mx = fmaxf(mx, row[j]);
    float sum = 0.f;
    for (int j = 0; j < S; ++j) {
      row[j] = expf(row[j] - mx);
      sum += row[j];
    }
    for (int j = 0; j < S; ++j) row[j] /= sum;
  }
  __syncthreads();
  const size_t pbase = (size_t)blockIdx.x * S * S;
  for (int e = tid; e < S * S; e += 256) {
    const int i = e / S, j = e - i * S;
    float pv = p[i * SP + j];
    P[pbase + e] = pv;  // probabilities BEFORE dropout (softmax backward needs them)
    if (dr.thr) pv = keep_bit(dr.key, (uint32_t)(pbase + e), dr.thr) ? pv * dr.scale : 0.f;
    p[i * SP + j] = pv;
  }
  __syncthreads();
  for (int e = tid; e < S * HD; e += 256) {
    const int i = e / HD, d = e - i * HD;
    float s = 0.f;
    for (int j = 0; j < S; ++j) s += p[i * SP + j] * v[j * AS + d];
    O[(size_t)(b * S + i) * D + h * HD + d] = s;
  }
}
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_g_kernel(const float* __restrict__ qkv, const float* __restrict__ P,
                                                         const float* __restrict__ dO, float* __restrict__ dqkv, int S, Drop dr) {
  extern __shared__ float sm[];
  constexpr int AS = HD + 1, D = 4 * HD;
  const int SP = S + 1;
  float *q = sm, *k = q + S * AS, *v = k + S * AS, *go = v + S * AS, *p = go + S * AS, *pd = p + S * SP, *ds = pd + S * SP;
  const int b = blockIdx.x >> 2, h = blockIdx.x & 3, tid = threadIdx.x;
  const float scale = 1.0f / sqrtf((float)HD);
  for (int i = tid; i < S * HD; i += 256) {
    const int s = i / HD, d = i - s * HD;
    const float* base = qkv + (size_t)(b * S + s) * (3 * D) + h * HD + d;
    q[s * AS + d] = base[0];
    k[s * AS + d] = base[D];
    v[s * AS + d] = base[2 * D];
    go[s * AS + d] = dO[(size_t)(b * S + s) * D + h * HD + d];
  }
  const size_t pbase = (size_t)blockIdx.x * S * S;
  for (int e = tid; e < S * S; e += 256) {
    const int i = e / S, j = e - i * S;
    const float pv = P[pbase + e];
    const float m = dr.thr ? (keep_bit(dr.key, (uint32_t)(pbase + e), dr.thr) ? dr.scale : 0.f) : 1.f;
    p[i * SP + j] = pv;
    pd[i * SP + j] = pv * m;
    ds[i * SP + j] = m;  // mask factor for now
  }
  __syncthreads();
  for (int e = tid; e < S * HD; e += 256) {  // dV[j][d] = sum_i Pd[i][j] dO[i][d]
    const int j = e / HD, d = e - j * HD;
    float s = 0.f;
    for (int i = 0; i < S; ++i) s += pd[i * SP + j] * go[i * AS + d];
    dqkv[(size_t)(b * S + j) * (3 * D) + 2 * D + h * HD + d] = s;
  }
  for (int e = tid; e < S * S; e += 256) {  // dP[i][j] = mask * sum_d dO[i][d] V[j][d]
    const int i = e / S, j = e - i * S;
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < HD; ++d) s += go[i * AS + d] * v[j * AS + d];
    ds[i * SP + j] *= s;
  }
  __syncthreads();
  if (tid < S) {  // dS = P * (dP - sum_j dP*P) / sqrt(hd)
    float t = 0.f;
    for (int j = 0; j < S; ++j) t += ds[tid * SP + j] * p[tid * SP + j];
    for (int j = 0; j < S; ++j) ds[tid * SP + j] = p[tid * SP + j] * (ds[tid * SP + j] - t) * scale;
  }
  __syncthreads();
  for (int e = tid; e < S * HD; e += 256) {
    const int i = e / HD, d = e - i * HD;
    float sq = 0.f, sk = 0.f;
    for (int j = 0; j < S; ++j) {
      sq += ds[i * SP + j] * k[j * AS + d];
      sk += ds[j * SP + i] * q[j * AS + d];
    }
    float* base = dqkv + (size_t)(b * S + i) * (3 * D) + h * HD + d;
    base[0] = sq;
    base[D] = sk;
  }
}

// out = LayerNorm(x + dropout(y)) over D columns, one wave per row (D / 256 float4 per lane, each wave-load 1 KiB contiguous).
template <int D>
__global__ void ln_fwd_g_kernel(const float* __restrict__ x, const float* __restrict__ y, int T, const float* __restrict__ gamma,
                                const float* __restrict__ beta, Drop dr, float* __restrict__ out, float* __restrict__ xhat,
                                float* __restrict__ save_rstd) {
  constexpr int NV = D / 256;
  const int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (t >= T) return;
  float4 a[NV];
  float s1 = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const size_t o = (size_t)t * D + j * 256 + lane * 4;
    a[j] = *reinterpret_cast<const float4*>(x + o);
    float4 f = *reinterpret_cast<const float4*>(y + o);
    if (dr.thr) {
      f.x = keep_bit(dr.key, (uint32_t)o + 0, dr.thr) ? f.x * dr.scale : 0.f;
      f.y = keep_bit(dr.key, (uint32_t)o + 1, dr.thr) ? f.y * dr.scale : 0.f;
      f.z = keep_bit(dr.key, (uint32_t)o + 2, dr.thr) ? f.z * dr.scale : 0.f;
      f.w = keep_bit(dr.key, (uint32_t)o + 3, dr.thr) ? f.w * dr.scale : 0.f;
    }
    a[j].x += f.x; a[j].y += f.y; a[j].z += f.z; a[j].w += f.w;
    s1 += (a[j].x + a[j].y) + (a[j].z + a[j].w);
  }
  const float mu = wsum(s1) * (1.f / D);
  float s2 = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    a[j].x -= mu; a[j].y -= mu; a[j].z -= mu; a[j].w -= mu;
    s2 += (a[j].x * a[j].x + a[j].y * a[j].y) + (a[j].z * a[j].z + a[j].w * a[j].w);
  }
  const float rstd = 1.0f / sqrtf(wsum(s2) * (1.f / D) + kLnEps);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const size_t o = (size_t)t * D + j * 256 + lane * 4;
    const float4 h = make_float4(a[j].x * rstd, a[j].y * rstd, a[j].z * rstd, a[j].w * rstd);
    const float4 g = *reinterpret_cast<const float4*>(gamma + j * 256 + lane * 4);
    const float4 be = *reinterpret_cast<const float4*>(beta + j * 256 + lane * 4);
    *reinterpret_cast<float4*>(xhat + o) = h;
    *reinterpret_cast<float4*>(out + o) = make_float4(h.x * g.x + be.x, h.y * g.y + be.y, h.z * g.z + be.z, h.w * g.w + be.w);
  }
  if (lane == 0) save_rstd[t] = rstd;
}
// LN backward (see ln_bwd_kernel): d_res = dz, d_y = dz * dropout mask; dgamma / dbeta by per-workgroup partials + atomics.
// 256-thread workgroups (4 waves), each wave strides over rows.
template <int D>
__global__ __launch_bounds__(256) void ln_bwd_g_kernel(const float* __restrict__ dout, const float* __restrict__ xhat,
                                                       const float* __restrict__ save_rstd, int T, const float* __restrict__ gamma,
                                                       Drop dr, float* __restrict__ d_res, float* __restrict__ d_y,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
  constexpr int NV = D / 256;
  __shared__ float4 rg[256 * NV], rb[256 * NV];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4 g[NV], ag[NV], ab[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    g[j] = *reinterpret_cast<const float4*>(gamma + j * 256 + lane * 4);
    ag[j] = ab[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int t = blockIdx.x * 4 + w; t < T; t += gridDim.x * 4) {
    const float rstd = save_rstd[t];
    float4 d[NV], h[NV], dh[NV];
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const size_t o = (size_t)t * D + j * 256 + lane * 4;
      d[j] = *reinterpret_cast<const float4*>(dout + o);
      h[j] = *reinterpret_cast<const float4*>(xhat + o);
      ag[j].x += d[j].x * h[j].x; ag[j].y += d[j].y * h[j].y; ag[j].z += d[j].z * h[j].z; ag[j].w += d[j].w * h[j].w;
      ab[j].x += d[j].x; ab[j].y += d[j].y; ab[j].z += d[j].z; ab[j].w += d[j].w;
      dh[j] = make_float4(d[j].x * g[j].x, d[j].y * g[j].y, d[j].z * g[j].z, d[j].w * g[j].w);
      m1 += (dh[j].x + dh[j].y) + (dh[j].z + dh[j].w);
      m2 += (dh[j].x * h[j].x + dh[j].y * h[j].y) + (dh[j].z * h[j].z + dh[j].w * h[j].w);
    }
    m1 = wsum(m1) * (1.f / D);
    m2 = wsum(m2) * (1.f / D);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const size_t o = (size_t)t * D + j * 256 + lane * 4;
      float4 dz = make_float4(rstd * (dh[j].x - m1 - h[j].x * m2), rstd * (dh[j].y - m1 - h[j].y * m2),
                              rstd * (dh[j].z - m1 - h[j].z * m2), rstd * (dh[j].w - m1 - h[j].w * m2));
      *reinterpret_cast<float4*>(d_res + o) = dz;
      if (dr.thr) {
        dz.x = keep_bit(dr.key, (uint32_t)o + 0, dr.thr) ? dz.x * dr.scale : 0.f;
        dz.y = keep_bit(dr.key, (uint32_t)o + 1, dr.thr) ? dz.y * dr.scale : 0.f;
        dz.z = keep_bit(dr.key, (uint32_t)o + 2, dr.thr) ? dz.z * dr.scale : 0.f;
        dz.w = keep_bit(dr.key, (uint32_t)o + 3, dr.thr) ? dz.w * dr.scale : 0.f;
      }
      *reinterpret_cast<float4*>(d_y + o) = dz;
    }
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    rg[threadIdx.x * NV + j] = ag[j];
    rb[threadIdx.x * NV + j] = ab[j];
  }
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      float4 sg = ag[j], sb = ab[j];
      for (int i = 1; i < 4; ++i) {
        const float4 a = rg[(lane + 64 * i) * NV + j], c = rb[(lane + 64 * i) * NV + j];
        sg.x += a.x; sg.y += a.y; sg.z += a.z; sg.w += a.w;
        sb.x += c.x; sb.y += c.y; sb.z += c.z; sb.w += c.w;
      }
      float* pg = dgamma + j * 256 + lane * 4;
      float* pb = dbeta + j * 256 + lane * 4;
      unsafeAtomicAdd(pg + 0, sg.x); unsafeAtomicAdd(pg + 1, sg.y); unsafeAtomicAdd(pg + 2, sg.z); unsafeAtomicAdd(pg + 3, sg.w);
      unsafeAtomicAdd(pb + 0, sb.x); unsafeAtomicAdd(pb + 1, sb.y); unsafeAtomicAdd(pb + 2, sb.z); unsafeAtomicAdd(pb + 3, sb.w);
    }
  }
}

// out[b][c] = max over the S rows of group b of (X + R) (R optional: the residual AROUND the inter-sentence layer); first maximal
// row wins, as torch.max does. arg keeps the row for the backward scatter.
__global__ __launch_bounds__(256) void seq_max_fwd_kernel(const float* __restrict__ X, const float* __restrict__ R, int B, int S, int D,
                                                          float* __restrict__ out, int32_t* __restrict__ arg) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)B * D) return;
  const int b = (int)(i / D), c = (int)(i - (size_t)b * D);
  float mx = 0.f;
  int am = 0;
  for (int s = 0; s < S; ++s) {
    const size_t o = ((size_t)b * S + s) * D + c;
    const float v = R ? X[o] + R[o] : X[o];
    if (s == 0 || v > mx) { mx = v; am = s; }
  }
  out[i] = mx;
  arg[i] = am;
}
// dX[b][s][c] = g[b][c] at s = arg, else 0 (the gradient of X and, with a residual, of R alike)
__global__ __launch_bounds__(256) void seq_max_bwd_kernel(const float* __restrict__ g, const int32_t* __restrict__ arg, int B, int S, int D,
                                                          float* __restrict__ dX) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)B * S * D) return;
  const int c = (int)(i % D);
  const size_t bs = i / D;
  const int s = (int)(bs % S);
  const size_t b = bs / S;
  dX[i] = arg[b * D + c] == s ? g[b * D + c] : 0.f;
}

// BatchNorm1d in training mode WITHOUT a ReLU behind it (inter_mlp = get_mlp2([1024, D]): Linear + BatchNorm1d,
// models/language_encoder.py:43-74,99) over M rows (a few hundred sentences) x C <= 256 columns: one workgroup per 4 columns, 64 row
// lanes each (one thread per column looping over all rows was 150 us per pass: a third of a millisecond per step).
__device__ __forceinline__ void bn4_reduce(double s1, double s2, double (*red)[4][2], int cl, int rl, double& t1, double& t2) {
  red[rl][cl][0] = s1;
  red[rl][cl][1] = s2;
  __syncthreads();
  t1 = t2 = 0.0;
  for (int i = 0; i < 64; ++i) {
    t1 += red[i][cl][0];
    t2 += red[i][cl][1];
  }
  __syncthreads();
}
// PHASE 0: statistics + apply in one launch. Cross-rank statistics (t2l_train_sync_bn) split it: PHASE 1 stores this rank's sums and row
// count into the slot `acc` (plain stores: a column has one owner), the host sums the slot over the ranks, PHASE 2 applies from it.
template <int PHASE = 0>
__global__ __launch_bounds__(256) void bn_plain_fwd_kernel(const float* __restrict__ y, int M, int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ run_mean,
                                                           float* __restrict__ run_var, float momentum, float* __restrict__ out,
                                                           float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                           double* __restrict__ acc = nullptr) {
  __shared__ double red[64][4][2];
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2, c = blockIdx.x * 4 + cl;
  double t1, t2, Mg = (double)M;
  if (PHASE != 2) {
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
      for (int m = rl; m < M; m += 64) {
        const double v = y[(size_t)m * C + c];
        s1 += v;
        s2 += v * v;
      }
    bn4_reduce(s1, s2, red, cl, rl, t1, t2);
  }
  if (PHASE == 1) {
    if (c < C && rl == 0) {
      acc[c] = t1;
      acc[C + c] = t2;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) acc[kBnCount] = (double)M;
    return;
  }
  if (c >= C) return;
  if (PHASE == 2) {
    t1 = acc[c];
    t2 = acc[C + c];
    Mg = acc[kBnCount];
  }
  const double mean = t1 / Mg, var = fmax(t2 / Mg - mean * mean, 0.0);
  const float rstd = 1.0f / sqrtf((float)var + kBnEps), g = gamma[c], be = beta[c];
  for (int m = rl; m < M; m += 64) out[(size_t)m * C + c] = (y[(size_t)m * C + c] - (float)mean) * rstd * g + be;
  if (rl == 0) {
    save_mean[c] = (float)mean;
    save_rstd[c] = rstd;
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)(var * (Mg / fmax(Mg - 1.0, 1.0)));
  }
}
// d: gradient w.r.t. the BatchNorm output (in), overwritten with the gradient w.r.t. its input y. PHASE as above; the parameter
// gradients are this rank's own sums in every phase form (the gradient all_reduce adds the ranks' up).
template <int PHASE = 0>
__global__ __launch_bounds__(256) void bn_plain_bwd_kernel(float* __restrict__ d, const float* __restrict__ y, int M, int C,
                                                           const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_rstd, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, double* __restrict__ acc = nullptr) {
  __shared__ double red[64][4][2];
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2, c = blockIdx.x * 4 + cl;
  const float mean = c < C ? save_mean[c] : 0.f, rstd = c < C ? save_rstd[c] : 0.f;
  double t1, t2;
  float Mg = (float)M;
  if (PHASE != 2) {
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
      for (int m = rl; m < M; m += 64) {
        const float dv = d[(size_t)m * C + c];
        s1 += dv;
        s2 += dv * (y[(size_t)m * C + c] - mean) * rstd;
      }
    bn4_reduce(s1, s2, red, cl, rl, t1, t2);
    if (c < C && rl == 0) {
      dgamma[c] += (float)t2;
      dbeta[c] += (float)t1;
    }
  }
  if (PHASE == 1) {
    if (c < C && rl == 0) {
      acc[c] = t1;
      acc[C + c] = t2;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) acc[kBnCount] = (double)M;
    return;
  }
  if (c >= C) return;
  if (PHASE == 2) {
    t1 = acc[c];
    t2 = acc[C + c];
    Mg = (float)acc[kBnCount];
  }
  const float f1 = (float)t1, f2 = (float)t2, g = gamma[c];
  for (int m = rl; m < M; m += 64) {
    const size_t i = (size_t)m * C + c;
    d[i] = g * rstd / Mg * (Mg * d[i] - f1 - (y[i] - mean) * rstd * f2);
  }
}

}  // namespace train
}  // namespace t2l
