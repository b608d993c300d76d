// Coarse retrieval: all-pairs query x submap-DB similarity + top-k, fused (gfx950 / CDNA4).
//
// Replaces the host loop of training/coarse.py:119-125 — per query a float64 `cell_encodings @ t`
// (N x 256) and a full argsort — by three launches on one stream, no host round trip:
//
//   scan_kernel    f32 MFMA (v_mfma_f32_32x32x2_f32, bit-exact f32 FMA chains) over [128 queries] x
//                  [DB split]; every lane owns (1 query, half of each 32-row tile) and keeps a sorted
//                  top-L list in registers; scores are never written to HBM.
//   rerank_kernel  one wave per query: merges the 2*nsplit sorted lists to the f32 top-L, re-scores
//                  those L rows in float64 (what the reference ranks by), orders them by
//                  (score desc, row asc), and CERTIFIES the result: every row that was not re-scored
//                  has f32 score <= g_L, so if  s64_K - g_L > eps(f32 error bound)  the top-K equals
//                  the float64 ranking exactly.
//   exact_kernel   only for queries whose certificate failed: float64 scan of the whole shard.
//
// Data layout in HBM: DB f32[n_pad,256] row-major (n_pad = n rounded up to 32, tail rows zero and
// masked by row >= n); queries f32[Q,256]; candidates f32/i32 [Q][2*nsplit][L].
#include <float.h>
#include <limits.h>

#include "t2l_internal.h"

namespace t2l {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define T2L_NEG_INF (-__builtin_inff())

// Sorted (descending) register list; x goes after equal elements, so among equal scores the one seen
// first (lower row id within a lane's ascending scan) stays ahead.
template <int L>
__device__ __forceinline__ void list_insert(float (&s)[L], int (&id)[L], float x, int xi) {
#pragma unroll
  for (int i = L - 1; i >= 1; --i) {
    const bool c_prev = x > s[i - 1];
    const bool c_cur = x > s[i];
    id[i] = c_prev ? id[i - 1] : (c_cur ? xi : id[i]);
    s[i] = c_prev ? s[i - 1] : (c_cur ? x : s[i]);
  }
  const bool c0 = x > s[0];
  id[0] = c0 ? xi : id[0];
  s[0] = c0 ? x : s[0];
}

// ------------------------------------------------------------------------------------------------
// scan: grid = n_qblocks * nsplit workgroups of 256 threads; block b -> split b % nsplit so that
// with nsplit a multiple of 8 every XCD (b % 8) keeps re-reading the same DB split from its own L2.
// LDS: 2 x [32 rows x 260 f32] DB tiles (glds double buffer) + per-lane candidate staging.
// ------------------------------------------------------------------------------------------------
template <int L>
__global__ __launch_bounds__(256, 1) void scan_kernel(const float* __restrict__ db, int n_rows, int n_tiles,
                                                      const float* __restrict__ q, int Q, int nsplit,
                                                      float* __restrict__ cand_s, int* __restrict__ cand_i) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tiles = smem;
  float* stage_s = smem + 2 * kTileFloats;
  int* stage_i = reinterpret_cast<int*>(stage_s + kScanWaves * kStageCap * 64);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int sp = blockIdx.x % nsplit, qb = blockIdx.x / nsplit;
  const int per = (n_tiles + nsplit - 1) / nsplit;
  const int t0 = sp * per;
  const int t1 = min(n_tiles, t0 + per);
  const int qrow = qb * kQPerBlock + wave * kQPerWave + col;
  const int qload = min(qrow, Q - 1);

  // B operand (queries), register resident for the whole scan. The MFMA sums over k in any order as
  // long as A and B agree: lane (col, half) owns k in [128*half, 128*half+128), one contiguous
  // 512-byte half row, so both operands load as float4.
  float qa[128];
  {
    const float4* qp = reinterpret_cast<const float4*>(q + (size_t)qload * kD + half * 128);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const float4 v = qp[i];
      qa[4 * i + 0] = v.x;
      qa[4 * i + 1] = v.y;
      qa[4 * i + 2] = v.z;
      qa[4 * i + 3] = v.w;
    }
  }

  float ls[L];
  int li[L];
#pragma unroll
  for (int i = 0; i < L; ++i) {
    ls[i] = T2L_NEG_INF;
    li[i] = -1;
  }
  float tau = T2L_NEG_INF;
  int cnt = 0;
  float* my_s = stage_s + wave * kStageCap * 64 + lane;
  int* my_i = stage_i + wave * kStageCap * 64 + lane;

  auto issue = [&](int t, int buf) {
    const float* src = db + (size_t)t * kTileRows * kD + lane * 4;
    float* dst = tiles + buf * kTileFloats;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = wave * 8 + i;  // one wave-instruction moves one 1 KiB DB row into LDS
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + row * kD),
                                       (__attribute__((address_space(3))) void*)(dst + row * kRowStrideF), 16, 0, 0);
    }
  };
  auto compact = [&]() {
    for (int slot = 0; __any(slot < cnt); ++slot) {
      const bool ok = slot < cnt;
      const float v = ok ? my_s[slot * 64] : T2L_NEG_INF;
      const int vi = ok ? my_i[slot * 64] : -1;
      list_insert<L>(ls, li, v, vi);
    }
    cnt = 0;
    tau = ls[L - 1];
  };

  if (t0 < t1) issue(t0, 0);
  for (int t = t0; t < t1; ++t) {
    const int buf = (t - t0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile t landed for every wave; every wave is done reading buffer buf^1
    if (t + 1 < t1) issue(t + 1, buf ^ 1);

    const float* tb = tiles + buf * kTileFloats + col * kRowStrideF + half * 128;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(tb + 4 * s);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qa[4 * s + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qa[4 * s + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qa[4 * s + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qa[4 * s + 3], acc, 0, 0, 0);
    }
    // D[row][col]: this lane holds query `col`, DB rows (r&3) + 8*(r>>2) + 4*half of the tile.
    const int row0 = t * kTileRows + 4 * half;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + (r & 3) + 8 * (r >> 2);
      const float v = acc[r];
      if (v > tau && row < n_rows) {
        my_s[cnt * 64] = v;
        my_i[cnt * 64] = row;
        ++cnt;
      }
    }
    if (__any(cnt > kStageCap - 16)) compact();
  }
  compact();

  if (qrow < Q) {
    const size_t base = (((size_t)qrow * nsplit + sp) * 2 + half) * L;
#pragma unroll
    for (int i = 0; i < L; ++i) {
      cand_s[base + i] = ls[i];
      cand_i[base + i] = li[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// rerank: one wave per query, 4 queries per 256-thread block.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool better(float s, int i, float bs, int bi) { return s > bs || (s == bs && i < bi); }

template <int L>
__global__ __launch_bounds__(256) void rerank_kernel(const float* __restrict__ db, const float* __restrict__ q, int Q,
                                                     int K, int parts, const float* __restrict__ cand_s,
                                                     const int* __restrict__ cand_i, int row_offset, float eps_rel,
                                                     const float* __restrict__ db_norm_max,
                                                     int32_t* __restrict__ out_idx, double* __restrict__ out_score,
                                                     int32_t* __restrict__ flags, int32_t* __restrict__ fb_count) {
  const int lane = threadIdx.x & 63;
  const int qid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qid >= Q) return;

  // ---- merge the `parts` sorted lists into the f32 top-L (lane c ends up with the c-th best)
  const size_t base = ((size_t)qid * parts + lane) * L;
  int ptr = 0;
  float head_s = T2L_NEG_INF;
  int head_i = INT_MAX;
  if (lane < parts) {
    head_s = cand_s[base];
    head_i = cand_i[base];
    if (head_i < 0) head_i = INT_MAX;
  }
  float my_s = T2L_NEG_INF;
  int my_i = INT_MAX;
  for (int r = 0; r < L; ++r) {
    float bs = head_s;
    int bi = head_i;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float os = __shfl_xor(bs, off);
      const int oi = __shfl_xor(bi, off);
      if (better(os, oi, bs, bi)) {
        bs = os;
        bi = oi;
      }
    }
    if (lane == r) {
      my_s = bs;
      my_i = bi;
    }
    if (bi != INT_MAX && head_i == bi) {  // the winner advances its list
      ++ptr;
      head_s = T2L_NEG_INF;
      head_i = INT_MAX;
      if (ptr < L) {
        head_s = cand_s[base + ptr];
        head_i = cand_i[base + ptr];
        if (head_i < 0) head_i = INT_MAX;
      }
    }
  }
  const float g_L = __shfl(my_s, L - 1);  // every row that is NOT re-scored has f32 score <= g_L

  // ---- float64 re-score of the L selected rows (products of f32 values are exact in f64)
  const float4 qv = reinterpret_cast<const float4*>(q + (size_t)qid * kD)[lane];
  double qn = (double)qv.x * qv.x + (double)qv.y * qv.y + (double)qv.z * qv.z + (double)qv.w * qv.w;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) qn += __shfl_xor(qn, off);
  double my_d = -__builtin_inf();
  for (int c = 0; c < L; ++c) {
    const int row = __shfl(my_i, c);
    if (row == INT_MAX) continue;  // wave-uniform
    const float4 dv = reinterpret_cast<const float4*>(db + (size_t)row * kD)[lane];
    double d = (double)dv.x * qv.x + (double)dv.y * qv.y + (double)dv.z * qv.z + (double)dv.w * qv.w;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) d += __shfl_xor(d, off);
    if (lane == c) my_d = d;
  }

  // ---- order by (float64 score desc, row asc); lane c computes its rank among the L
  int rank = 0;
  for (int j = 0; j < L; ++j) {
    const double dj = __shfl(my_d, j);
    const int ij = __shfl(my_i, j);
    rank += (dj > my_d || (dj == my_d && ij < my_i)) ? 1 : 0;
  }
  const bool valid = lane < L && my_i != INT_MAX;
  if (lane < K) {
    out_idx[(size_t)qid * K + lane] = -1;
    if (out_score) out_score[(size_t)qid * K + lane] = -__builtin_inf();
  }
  if (valid && rank < K) {
    out_idx[(size_t)qid * K + rank] = my_i + row_offset;
    if (out_score) out_score[(size_t)qid * K + rank] = my_d;
  }

  // ---- certificate
  const unsigned long long kth = __ballot(valid && rank == K - 1);
  bool certified = true;
  if (g_L != T2L_NEG_INF) {
    if (kth == 0ull) {
      certified = false;  // cannot happen (>= K valid candidates whenever an L-th one exists); stay safe
    } else {
      const double dK = __shfl(my_d, __ffsll((long long)kth) - 1);
      const double eps = (double)eps_rel * sqrt(qn) * (double)(*db_norm_max);
      certified = (dK - (double)g_L) > eps;
    }
  }
  if (lane == 0) {
    flags[qid] = certified ? 0 : 1;
    if (!certified) atomicAdd(fb_count, 1);
  }
}

// ------------------------------------------------------------------------------------------------
// exact fallback: one workgroup per flagged query; float64 scan of every row + top-K selection.
// ------------------------------------------------------------------------------------------------
template <int KMAX>
__global__ __launch_bounds__(256) void exact_kernel(const float* __restrict__ db, int n_rows,
                                                    const float* __restrict__ q, int K,
                                                    const int32_t* __restrict__ flags, int row_offset,
                                                    int32_t* __restrict__ out_idx, double* __restrict__ out_score) {
  const int qid = blockIdx.x;
  if (!flags[qid]) return;
  __shared__ double qs[kD];
  __shared__ double red_s[256];
  __shared__ int red_i[256];
  __shared__ int red_t[256];
  const int tid = threadIdx.x;
  qs[tid] = (double)q[(size_t)qid * kD + tid];
  __syncthreads();

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
    if (d > ls[KMAX - 1]) {
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
      out_idx[(size_t)qid * K + r] = ok ? red_i[0] + row_offset : -1;
      if (out_score) out_score[(size_t)qid * K + r] = ok ? red_s[0] : -__builtin_inf();
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

// max row 2-norm of the shard (bounds the f32 dot-product error in the certificate)
__global__ __launch_bounds__(256) void db_norm_kernel(const float* __restrict__ db, int n_rows, float* out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const float4 v = reinterpret_cast<const float4*>(db + (size_t)row * kD)[lane];
  float s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) atomicMax(reinterpret_cast<int*>(out), __float_as_int(sqrtf(s) * 1.0001f));
}

// ------------------------------------------------------------------------------------------------
// merge of per-shard top-k lists (the one exchange step of the row-sharded DB):
// idx/score [parts][Q][K] (as all-gathered over RCCL) -> [Q][K] by (score desc, row id asc).
// One wave per query; parts*K <= 256.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void merge_kernel(const int32_t* __restrict__ idx, const double* __restrict__ score,
                                                    int parts, int Q, int K, int32_t* __restrict__ out_idx,
                                                    double* __restrict__ out_score) {
  const int lane = threadIdx.x & 63;
  const int qid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qid >= Q) return;
  const int total = parts * K;
  double s[4];
  int id[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = lane + 64 * e;
    s[e] = -__builtin_inf();
    id[e] = INT_MAX;
    if (c < total) {
      const size_t off = ((size_t)(c / K) * Q + qid) * K + (c % K);
      const int v = idx[off];
      if (v >= 0) {
        id[e] = v;
        s[e] = score[off];
      }
    }
  }
  for (int r = 0; r < K; ++r) {
    double bs = s[0];
    int bi = id[0];
#pragma unroll
    for (int e = 1; e < 4; ++e)
      if (s[e] > bs || (s[e] == bs && id[e] < bi)) {
        bs = s[e];
        bi = id[e];
      }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double os = __shfl_xor(bs, off);
      const int oi = __shfl_xor(bi, off);
      if (os > bs || (os == bs && oi < bi)) {
        bs = os;
        bi = oi;
      }
    }
    if (lane == 0) {
      out_idx[(size_t)qid * K + r] = bi == INT_MAX ? -1 : bi;
      if (out_score) out_score[(size_t)qid * K + r] = bs;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (bi != INT_MAX && id[e] == bi) {  // row ids are unique across shards
        s[e] = -__builtin_inf();
        id[e] = INT_MAX;
      }
  }
}

int merge_impl(t2l_ctx* ctx, const int32_t* idx, const double* score, int parts, int Q, int K, int32_t* out_idx,
               double* out_score, hipStream_t s) {
  if (parts * K > 256) return fail(ctx, T2L_EINVAL, "t2l_merge_topk: parts * k must be <= 256");
  hipLaunchKernelGGL(merge_kernel, dim3((Q + 3) / 4), dim3(256), 0, s, idx, score, parts, Q, K, out_idx, out_score);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

int db_norm_impl(t2l_ctx* ctx, hipStream_t s) {
  T2L_HIP(ctx, hipMemsetAsync(ctx->db_norm_max, 0, sizeof(float), s));
  if (ctx->db_rows > 0) {
    const int blocks = (int)((ctx->db_rows + 3) / 4);
    hipLaunchKernelGGL(db_norm_kernel, dim3(blocks), dim3(256), 0, s, ctx->db, (int)ctx->db_rows, ctx->db_norm_max);
    T2L_HIP(ctx, hipGetLastError());
  }
  return T2L_OK;
}

static size_t scan_lds_bytes() {
  return (size_t)2 * kTileFloats * sizeof(float) + (size_t)kScanWaves * kStageCap * 64 * 8;
}

template <int L>
static int launch_search(t2l_ctx* ctx, const float* q, int Q, int K, int nsplit, int32_t* out_idx, double* out_score,
                         hipStream_t s) {
  const int n_rows = (int)ctx->db_rows;
  const int n_tiles = (int)(ctx->db_pad / kTileRows);
  const int n_qblocks = (Q + kQPerBlock - 1) / kQPerBlock;
  const int parts = 2 * nsplit;
  const size_t lds = scan_lds_bytes();
  static bool attr_done = false;
  if (!attr_done) {
    T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_kernel<L>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  event_begin(ctx, "search_scan", s);
  hipLaunchKernelGGL(scan_kernel<L>, dim3(n_qblocks * nsplit), dim3(256), lds, s, ctx->db, n_rows, n_tiles, q, Q,
                     nsplit, ctx->cand_score, ctx->cand_idx);
  event_end(ctx, "search_scan", s);
  T2L_HIP(ctx, hipGetLastError());
  // f32 dot-product error bound: gamma_n * |a||b| with n = 256 terms (+ slack for the MFMA's k order)
  const float eps_rel = (float)(ctx->eps_scale * (kD + 8) * 5.9604644775390625e-08);
  event_begin(ctx, "search_rerank", s);
  hipLaunchKernelGGL(rerank_kernel<L>, dim3((Q + 3) / 4), dim3(256), 0, s, ctx->db, q, Q, K, parts, ctx->cand_score,
                     ctx->cand_idx, (int)ctx->row_offset, eps_rel, ctx->db_norm_max, out_idx, out_score, ctx->flags,
                     ctx->fb_count);
  T2L_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(exact_kernel<32>, dim3(Q), dim3(256), 0, s, ctx->db, n_rows, q, K, ctx->flags,
                     (int)ctx->row_offset, out_idx, out_score);
  event_end(ctx, "search_rerank", s);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

int search_impl(t2l_ctx* ctx, const float* q, int Q, int K, int32_t* out_idx, double* out_score, hipStream_t s) {
  if (Q == 0) return T2L_OK;
  const int n_tiles = (int)(ctx->db_pad / kTileRows);
  const int n_qblocks = (Q + kQPerBlock - 1) / kQPerBlock;
  int nsplit = ctx->nsplit_override;
  if (nsplit <= 0) {
    // >= 1 workgroup per CU (256 CUs); multiples of 8 keep a split on one XCD's L2
    nsplit = (256 + n_qblocks - 1) / n_qblocks;
    nsplit = ((nsplit + 7) / 8) * 8;
  }
  nsplit = max(1, min(nsplit, kMaxParts / 2));
  nsplit = max(1, min(nsplit, max(1, n_tiles)));
  const int L = (K <= 10) ? 16 : 32;
  const size_t need = (size_t)n_qblocks * kQPerBlock * 2 * nsplit * L;
  if (need > ctx->cand_cap) {
    if (ctx->cand_score) (void)hipFree(ctx->cand_score);
    if (ctx->cand_idx) (void)hipFree(ctx->cand_idx);
    ctx->cand_score = nullptr;
    ctx->cand_idx = nullptr;
    ctx->cand_cap = 0;
    T2L_HIP(ctx, hipMalloc(&ctx->cand_score, need * sizeof(float)));
    T2L_HIP(ctx, hipMalloc(&ctx->cand_idx, need * sizeof(int32_t)));
    ctx->cand_cap = need;
  }
  if ((size_t)Q > ctx->flag_cap) {
    if (ctx->flags) (void)hipFree(ctx->flags);
    ctx->flags = nullptr;
    ctx->flag_cap = 0;
    T2L_HIP(ctx, hipMalloc(&ctx->flags, (size_t)Q * sizeof(int32_t)));
    ctx->flag_cap = Q;
  }
  T2L_HIP(ctx, hipMemsetAsync(ctx->fb_count, 0, sizeof(int32_t), s));
  if (L == 16) return launch_search<16>(ctx, q, Q, K, nsplit, out_idx, out_score, s);
  return launch_search<32>(ctx, q, Q, K, nsplit, out_idx, out_score, s);
}

}  // namespace t2l
