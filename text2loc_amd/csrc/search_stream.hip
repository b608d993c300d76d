// Small query batches against a LARGE shard: the HBM-streaming form of the search (SURVEY.md §8d config 2').
//
// With few queries (<= 64) the arithmetic is far below the matrix pipe's capacity and the job is to pull the DB
// through the chip once at HBM speed. The batched scan (search.hip) gives every 256-query block its own pass over
// the DB and would leave most CUs idle here. This variant turns the decomposition around:
//
//   scanq_kernel       ALL waves of the chip hold the SAME (<= 32) queries as register-resident f16 fragments (scaled by
//                      a power of two, as scanh_kernel) and each wave streams its OWN 32-row tiles of the f16 DB plane
//                      HBM -> LDS (LDS-DMA into a private 2 x 16 KiB double buffer: the next tile is in flight while the
//                      current one multiplies; no workgroup barrier anywhere), one f16 MFMA per product, per-lane sorted
//                      key lists. At the end a workgroup merges its 8 lists per query into one (key, row) list: the
//                      candidate set is [32 queries][G workgroups][L].
//   rerank_rows_kernel one workgroup per query: top-L of the G lists by key, float64 re-score, (score desc, row asc)
//                      order, the same certificate as the batched path.
//   exact_only_kernel  float64 scan of the shard for queries whose certificate failed.
//
// Algorithmic bytes per launch: the f16 DB plane once = 512 B per row (+ G*32*L*8 B of candidates).
#include "search_dev.h"

namespace t2l {

constexpr int kStreamQ = 32;  // queries per scanq launch

// 16 k-steps of one tile: fragment ring of 4 (within the tile), one MFMA and the insertion of one score of the previous
// tile per k-step (1 + L VALU, pinned behind the MFMA)
template <int L, int S>
__device__ __forceinline__ void tileq_steps(const char* tb, const u32x4 (&qf)[16], f32x16& cur,
                                            const f32x16& prev, int vmask, int code0, float pinf, float (&ls)[L],
                                            u32x4 (&ring)[4]) {
  if constexpr (S < 16) {
    const u32x4 a = ring[S & 3];
    const int code = __builtin_amdgcn_readfirstlane(code0 + S);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S == 0) mfma_f16_first(cur, a, qf[S]); else mfma_f16_acc(cur, a, qf[S]);
    if constexpr (S + 4 < 16) ring[S & 3] = *reinterpret_cast<const u32x4*>(tb + (S + 4) * 512);
    __builtin_amdgcn_sched_barrier(0);
    ins_key_sat<L>(ls, __int_as_float((__float_as_int(prev[S]) & vmask) | code), pinf);
    __builtin_amdgcn_sched_barrier(0);
    tileq_steps<L, S + 1>(tb, qf, cur, prev, vmask, code0, pinf, ls, ring);
  }
}

template <int L>
__global__ __launch_bounds__(256, 1) void scanq_kernel(const uint4* __restrict__ dbh, int n_rows, int n_tiles, int per,
                                                       int code_bits, const float* __restrict__ q, int q0, int Q,
                                                       float* __restrict__ cand_key, int* __restrict__ cand_row,
                                                       float pinf) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform_wave_id();
  const int half = lane >> 5, col = lane & 31;
  const int G = gridDim.x, wg = blockIdx.x;
  char* tiles = reinterpret_cast<char*>(smem) + wave * (2 * kHalfTileBytes);  // this wave's private double buffer
  const unsigned tiles_lds = lds_addr_of(tiles);
  const int mask = ~((1 << code_bits) - 1);
  int vmask = mask;
  asm volatile("" : "+v"(vmask));

  // every wave of the chip: the same 32 queries q0 .. q0+31 (clamped), scaled and rounded to f16 like scanh_kernel
  u32x4 qf[16];
  {
    const int qrow = min(q0 + col, Q - 1);
    const float4* qp = reinterpret_cast<const float4*>(q + (size_t)qrow * kD + half * 128);
    float4 v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = qp[i];
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      m = fmaxf(fmaxf(m, fabsf(v[i].x)), fabsf(v[i].y));
      m = fmaxf(fmaxf(m, fabsf(v[i].z)), fabsf(v[i].w));
    }
    {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
      m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    int shift;
    half_shift_of(m, shift);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float4 a = v[2 * s], b = v[2 * s + 1];
      qf[s] = pin_agpr(u32x4{pack_f16x2(a.x, a.y, shift), pack_f16x2(a.z, a.w, shift), pack_f16x2(b.x, b.y, shift),
                             pack_f16x2(b.z, b.w, shift)});
    }
  }
  float ls[L];
#pragma unroll
  for (int i = 0; i < L; ++i) ls[i] = T2L_NEG_INF;
  f32x16 accA, accB;
#pragma unroll
  for (int r = 0; r < 16; ++r) accA[r] = accB[r] = T2L_NEG_INF;

  // LDS tile image == global tile image (tile-chunk-major f16 plane, search_dev.h): 16 contiguous 1 KiB LDS-DMA pieces per
  // tile; lane (col, half) reads chunk half*16 + S of row col at  half*8192 + col*16 + S*512.
  const unsigned lane16 = lane * 16;
  const int frag_off = half * 8192 + col * 16;

  // wave (wg, wave) owns tiles  (wg*per + j)*4 + wave,  j = 0 .. per-1  (interleaved so neighbours stream neighbours)
  const int tbase = wg * per * 4 + wave;
  auto tile_of = [&](int j) { return tbase + 4 * j; };
  auto fetch = [&](int t, int buf) {
    const char* src = reinterpret_cast<const char*>(dbh) + (size_t)t * kHalfTileBytes;
    const unsigned dst = tiles_lds + buf * kHalfTileBytes;
#pragma unroll
    for (int i = 0; i < 16; ++i) lds_dma_row(dst + i * 1024, lane16, src + i * 1024);
  };
  auto step = [&](int j, int buf, bool more, f32x16& cur, const f32x16& prev) {
    // tile j has landed (its 16 pieces are older than the 16 of tile j+1 that may still be in flight)
    if (more) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char* tb = tiles + buf * kHalfTileBytes + frag_off;
    u32x4 ring[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ring[i] = *reinterpret_cast<const u32x4*>(tb + i * 512);
    tileq_steps<L, 0>(tb, qf, cur, prev, vmask, (j - 1) << 4, pinf, ls, ring);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // all LDS reads of this tile are done: the buffer may refill
  };
  int nj = 0;
  while (nj < per && tile_of(nj) < n_tiles) ++nj;  // tiles this wave really has
  if (nj > 0) fetch(tile_of(0), 0);
  for (int j = 0; j < nj; j += 2) {
    if (j + 1 < nj) fetch(tile_of(j + 1), 1);  // buffer 1 was last read by tile j-1 (lgkmcnt(0) above)
    step(j, 0, j + 1 < nj, accA, accB);
    if (j + 1 < nj) {
      if (j + 2 < nj) fetch(tile_of(j + 2), 0);
      step(j + 1, 1, j + 2 < nj, accB, accA);
    }
  }
  if (nj > 0) {  // the last tile's scores are still in registers; only the last tile of the shard can be partial
    const int row0 = tile_of(nj - 1) * kTileRows + 4 * half;
    const int code0 = (nj - 1) << 4;
    const bool odd = nj & 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + (r & 3) + 8 * (r >> 2);
      ins_key<L>(ls, row < n_rows ? make_key(odd ? accA[r] : accB[r], mask, code0 + r) : T2L_NEG_INF);
    }
  }

  // ---- workgroup merge: 8 sorted lists per query (4 waves x 2 halves) -> one (key, row) list of L
  __syncthreads();  // every wave is done with its tile buffers: reuse LDS
  float* lists = smem;  // [32 queries][8 lists][L]
#pragma unroll
  for (int i = 0; i < L; ++i) lists[(col * 8 + wave * 2 + half) * L + i] = ls[i];
  __syncthreads();
  {
    const int qi = tid >> 3, j = tid & 7;  // 8 consecutive lanes own the 8 lists of one query
    const float* mine = lists + (qi * 8 + j) * L;
    const int jw = j >> 1, jh = j & 1;     // list j came from wave jw, half jh
    int ptr = 0;
    float head = mine[0];
    for (int r = 0; r < L; ++r) {
      float bk = head;
      int bj = j;
#pragma unroll
      for (int off = 4; off >= 1; off >>= 1) {
        const float ok = __shfl_xor(bk, off);
        const int oj = __shfl_xor(bj, off);
        if (ok > bk || (ok == bk && oj < bj)) {
          bk = ok;
          bj = oj;
        }
      }
      if (j == bj) {  // the winner emits and advances
        int row = -1;
        if (bk != T2L_NEG_INF) {
          const int code = __float_as_int(bk) & ~mask;
          const int rr = code & 15;
          row = plane_row(((wg * per + (code >> 4)) * 4 + jw) * kTileRows + (rr & 3) + 8 * (rr >> 2) + 4 * jh, n_rows);
        }
        const size_t o = ((size_t)qi * G + wg) * L + r;
        cand_key[o] = bk;
        cand_row[o] = row;
        ++ptr;
        head = ptr < L ? mine[ptr] : T2L_NEG_INF;
      }
    }
  }
}

// One workgroup per query: G sorted (key,row) lists -> top-L by key -> float64 re-score -> order + certificate.
template <int L>
__global__ __launch_bounds__(256) void rerank_rows_kernel(const float* __restrict__ db, const float* __restrict__ q,
                                                          int q0, int Q, int K, int G, int code_bits,
                                                          const float* __restrict__ cand_key,
                                                          const int* __restrict__ cand_row, int row_offset,
                                                          float eps_rel, const float* __restrict__ db_norm_max,
                                                          int32_t* __restrict__ out_idx, double* __restrict__ out_score,
                                                          int32_t* __restrict__ flags, float pinf) {
  const int qi = blockIdx.x, qid = q0 + qi;
  if (qid >= Q) return;
  __shared__ float sel_key[32];
  __shared__ int sel_row[32];
  __shared__ double sel_d[32];
  __shared__ float red_k[4];
  __shared__ int red_t[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // thread t walks the lists of workgroups t, t+256, ... merged on the fly: it keeps ONE current head
  // (the best not-yet-taken key among its lists) by re-scanning its lists' heads (G/256 <= 4 lists per thread)
  constexpr int MAXL = 4;
  int ptr[MAXL];
#pragma unroll
  for (int i = 0; i < MAXL; ++i) ptr[i] = 0;
  const float* keys = cand_key + (size_t)qi * G * L;
  const int* rows = cand_row + (size_t)qi * G * L;
  auto my_head = [&](int& which) {
    float best = T2L_NEG_INF;
    which = -1;
#pragma unroll
    for (int i = 0; i < MAXL; ++i) {
      const int g = tid + 256 * i;
      if (g < G && ptr[i] < L) {
        const float k = keys[(size_t)g * L + ptr[i]];
        if (k > best) {
          best = k;
          which = i;
        }
      }
    }
    return best;
  };
  int which;
  float head = my_head(which);
  for (int r = 0; r < L; ++r) {
    const float wk = wave_max_f32(head, pinf);
    const unsigned long long who = __ballot(head == wk);
    if (lane == 0) {
      red_k[wave] = wk;
      red_t[wave] = wave * 64 + (__ffsll((long long)who) - 1);
    }
    __syncthreads();
    float bk = red_k[0];
    int bt = red_t[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (red_k[w] > bk) {
        bk = red_k[w];
        bt = red_t[w];
      }
    if (tid == bt) {
      int row = INT_MAX;
      if (bk != T2L_NEG_INF) {
        const int g = tid + 256 * which;
        row = rows[(size_t)g * L + ptr[which]];
#pragma unroll
        for (int i = 0; i < MAXL; ++i)
          if (i == which) ++ptr[i];
        head = my_head(which);
      }
      sel_key[r] = bk;
      sel_row[r] = row < 0 ? INT_MAX : row;
    }
    __syncthreads();
  }

  // float64 re-score: wave w takes selected rows w, w+4, ...
  const float4 qv = reinterpret_cast<const float4*>(q + (size_t)qid * kD)[lane];
  const double qn = wave_sum_f64((double)qv.x * qv.x + (double)qv.y * qv.y + (double)qv.z * qv.z + (double)qv.w * qv.w);
  for (int c = wave; c < L; c += 4) {
    const int row = sel_row[c];
    double d = -__builtin_inf();
    if (row != INT_MAX) {
      const float4 dv = reinterpret_cast<const float4*>(db + (size_t)row * kD)[lane];
      d = wave_sum_f64((double)dv.x * qv.x + (double)dv.y * qv.y + (double)dv.z * qv.z + (double)dv.w * qv.w);
    }
    if (lane == 0) sel_d[c] = d;
  }
  __syncthreads();
  if (tid < 64) {
    const double my_d = lane < L ? sel_d[lane] : -__builtin_inf();
    const int my_row = lane < L ? sel_row[lane] : INT_MAX;
    int rank = 0;
    for (int j = 0; j < L; ++j) {
      const double dj = sel_d[j];
      const int ij = sel_row[j];
      rank += (dj > my_d || (dj == my_d && ij < my_row)) ? 1 : 0;
    }
    const bool valid = lane < L && my_row != INT_MAX;
    if (lane < K) {
      out_idx[(size_t)qid * K + lane] = -1;
      if (out_score) out_score[(size_t)qid * K + lane] = -__builtin_inf();
    }
    if (valid && rank < K) {
      out_idx[(size_t)qid * K + rank] = my_row + row_offset;
      if (out_score) out_score[(size_t)qid * K + rank] = my_d;
    }
    const float g = sel_key[L - 1];
    // keys are true scores times 2^(shift_db + shift_q) (f16 scan): undo that exactly; inputs without an f16 image are
    // scanned exactly
    int sq, sd;
    const float m = wave_max_f32(fmaxf(fmaxf(fabsf(qv.x), fabsf(qv.y)), fmaxf(fabsf(qv.z), fabsf(qv.w))), pinf);
    bool representable = half_shift_of(m, sq);
    representable = half_shift_of(db_norm_max[1], sd) && representable;
    const double kscale = ldexp(1.0, -(sq + sd));
    bool certified = true;
    if (g != T2L_NEG_INF) {
      certified = false;
      const unsigned long long kth = __ballot(valid && rank == K - 1);
      if (K <= L && kth != 0ull) {
        const double dK = sel_d[__ffsll((long long)kth) - 1];
        const double eps32 = (double)eps_rel * sqrt(qn) * (double)(*db_norm_max);
        certified = dK > (double)g * kscale + key_slack(g, code_bits, eps32, kscale);
      }
    }
    if (lane == 0) flags[qid] = (certified && representable) ? 0 : 1;
  }
}

__global__ __launch_bounds__(256) void exact_only_kernel(const float* __restrict__ db, int n_rows,
                                                         const float* __restrict__ q, int q0, int nq, int K,
                                                         const int32_t* __restrict__ flags, int row_offset,
                                                         int32_t* __restrict__ out_idx, double* __restrict__ out_score,
                                                         int32_t* __restrict__ fb_count) {
  __shared__ double qs[kD];
  __shared__ double red_s[256];
  __shared__ int red_i[256];
  __shared__ int red_t[256];
  for (int qid = q0 + blockIdx.x; qid < q0 + nq; qid += gridDim.x) {
    if (!flags[qid]) continue;
    __syncthreads();
    qs[threadIdx.x] = (double)q[(size_t)qid * kD + threadIdx.x];
    if (threadIdx.x == 0) atomicAdd(&fb_count[0], 1);
    __syncthreads();
    exact_scan<32>(db, n_rows, qs, K, row_offset, out_idx + (size_t)qid * K,
                   out_score ? out_score + (size_t)qid * K : nullptr, red_s, red_i, red_t);
  }
}

static int grow_buf(t2l_ctx* ctx, void** p, size_t* cap, size_t need_bytes) {
  if (need_bytes <= *cap) return T2L_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  T2L_HIP(ctx, hipMalloc(p, need_bytes));
  *cap = need_bytes;
  return T2L_OK;
}

template <int L>
static int stream_launch(t2l_ctx* ctx, const float* q, int Q, int K, int32_t* out_idx, double* out_score, hipStream_t s) {
  const int n_rows = (int)ctx->db_rows;
  const int n_tiles = (int)(ctx->db_pad / kTileRows);
  int G = 256;  // one workgroup (4 waves, 4 private 2 x 16 KiB tile buffers = 128 KiB of LDS) per CU
  while (G > 1 && (n_tiles + G * 4 - 1) / (G * 4) < 2) G >>= 1;
  const int per = (n_tiles + G * 4 - 1) / (G * 4);
  int code_bits = 4;
  while ((1 << code_bits) < per * 16) ++code_bits;
  if (code_bits > 13) return fail(ctx, T2L_EINVAL, "t2l_search: shard too large for one streaming launch (16.7M rows)");
  int rc;
  if ((rc = grow_buf(ctx, (void**)&ctx->cand_score, &ctx->cand_cap, (size_t)kStreamQ * G * L * sizeof(float))) != T2L_OK ||
      (rc = grow_buf(ctx, (void**)&ctx->seg_idx, &ctx->seg_idx_cap, (size_t)kStreamQ * G * L * sizeof(int32_t))) != T2L_OK ||
      (rc = grow_buf(ctx, (void**)&ctx->flags, &ctx->flag_cap, (size_t)Q * sizeof(int32_t))) != T2L_OK)
    return rc;
  const size_t lds = (size_t)4 * 2 * kHalfTileBytes;
  static PerDeviceOnce attr_done;
  if (attr_done.need(ctx->device)) {
    T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&scanq_kernel<L>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done.mark(ctx->device);
  }
  const float eps_rel = (float)(ctx->eps_scale * ((kD + 8) * 5.9604644775390625e-08 + 9.85e-4));  // f16 operands (search.hip)
  T2L_HIP(ctx, hipMemsetAsync(ctx->fb_count, 0, 2 * sizeof(int32_t), s));
  for (int q0 = 0; q0 < Q; q0 += kStreamQ) {
    const int nq = min(kStreamQ, Q - q0);
    event_begin(ctx, "search_scan", s);
    hipLaunchKernelGGL(scanq_kernel<L>, dim3(G), dim3(256), lds, s, ctx->db_half, n_rows, n_tiles, per, code_bits, q, q0,
                       Q, ctx->cand_score, ctx->seg_idx, __builtin_inff());
    event_end(ctx, "search_scan", s);
    T2L_HIP(ctx, hipGetLastError());
    event_begin(ctx, "search_rerank", s);
    hipLaunchKernelGGL(rerank_rows_kernel<L>, dim3(nq), dim3(256), 0, s, ctx->db, q, q0, Q, K, G, code_bits,
                       ctx->cand_score, ctx->seg_idx, (int)ctx->row_offset, eps_rel, ctx->db_norm_max, out_idx, out_score,
                       ctx->flags, __builtin_inff());
    T2L_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(exact_only_kernel, dim3(min(nq, 32)), dim3(256), 0, s, ctx->db, n_rows, q, q0, nq, K, ctx->flags,
                       (int)ctx->row_offset, out_idx, out_score, ctx->fb_count);
    event_end(ctx, "search_rerank", s);
    T2L_HIP(ctx, hipGetLastError());
  }
  return T2L_OK;
}

int search_stream_impl(t2l_ctx* ctx, const float* q, int Q, int K, int32_t* out_idx, double* out_score, hipStream_t s) {
  if (K <= 10) return stream_launch<16>(ctx, q, Q, K, out_idx, out_score, s);
  return stream_launch<32>(ctx, q, Q, K, out_idx, out_score, s);
}

}  // namespace t2l
