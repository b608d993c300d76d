// The one exchange step of the row-sharded search on the device: merging the ranks' all-gathered per-shard top-k
// lists, and the pack kernel of the single-buffer exchange form. Split out of search.hip in round 5 (nothing here touches the scan).
#include <float.h>
#include <limits.h>

#include "t2l_internal.h"
#include "search_dev.h"

namespace t2l {

// ------------------------------------------------------------------------------------------------
// merge of per-shard top-k lists (the one exchange step of the row-sharded DB):
// idx/score [parts][Q][K] (as all-gathered over RCCL) -> [Q][K] by (score desc, row id asc).
// One wave per query; parts*K <= 256.
// ------------------------------------------------------------------------------------------------
// PAIRS: the input is the all-gathered {score, row id as f64} records of t2l_pack_pairs (idx unused) — no unpack launch.
// Otherwise part p's ids / scores start part_stride BYTES after part p-1's (t2l_merge_topk: two separate [parts][Q][K] arrays;
// t2l_merge_gathered: every rank's contiguous {ids | scores} block as all-gathered, no pack launch either).
// Ranking by counting: the wave's <= 256 candidates sit in LDS ((score, id) = 12 bytes each), every lane counts how many
// candidates beat each of its own (broadcast reads, no cross-lane dependency chain: the previous K rounds of a 6-step f64
// butterfly cost ~10k dependent cycles per query) and the K best go straight to their output slots.
template <bool PAIRS>
__global__ __launch_bounds__(256) void merge_kernel(const char* __restrict__ idx, size_t idx_stride, const char* __restrict__ score,
                                                    size_t score_stride, int parts, int Q, int K, int32_t* __restrict__ out_idx,
                                                    double* __restrict__ out_score) {
  __shared__ double sh_s[4][256];
  __shared__ int sh_i[4][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int qid = blockIdx.x * 4 + wv;
  if (qid >= Q) return;  // (one wave per query: no workgroup barrier below)
  const int total = parts * K, ne = (total + 63) >> 6;
  const float inv_k = 1.0f / (float)K;  // c / K for c < 256, K <= 26 without an integer division (~35 instructions each, a dozen per wave)
  double s[4];
  int id[4], part_of[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = lane + 64 * e;
    s[e] = -__builtin_inf();
    id[e] = INT_MAX;
    part_of[e] = (int)(((float)c + 0.5f) * inv_k);
    if (e < ne && c < total) {
      const int part = part_of[e];
      const size_t off = (size_t)qid * K + (c - part * K);
      if constexpr (PAIRS) {
        const double2 pr = reinterpret_cast<const double2*>(score + part * score_stride)[off];
        const int v = (int)pr.y;
        if (v >= 0) {
          id[e] = v;
          s[e] = pr.x;
        }
      } else {  // (both loads issued together: the score's address does not depend on the id)
        const int v = reinterpret_cast<const int32_t*>(idx + part * idx_stride)[off];
        const double sv = reinterpret_cast<const double*>(score + part * score_stride)[off];
        if (v >= 0) {
          id[e] = v;
          s[e] = sv;
        }
      }
    }
  }
  // ---- fast path: every part is already best-first ((score desc, row id asc), invalid entries trailing) — what t2l_search writes and
  // therefore what the sharded exchange carries. Checked here, not assumed: the wave stages its candidates in LDS, every candidate
  // looks at its successor inside its part, and only if no pair is out of order the K answers come out of K rounds of an arg-max over
  // the `parts` list heads (lane p = part p; the winner pops its next entry from LDS). ~40 instructions per round instead of an
  // all-pairs count over the ~50 candidates that survive the bound below (measured: 18.7 -> see DESIGN 5 per 4,096 x 8 x 10).
  {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = lane + 64 * e;
      if (e < ne && c < total) {
        sh_s[wv][c] = s[e];
        sh_i[wv][c] = id[e];
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    bool ok = true;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = lane + 64 * e;
      if (e < ne && c + 1 < total && part_of[e] * K + K - 1 != c) {  // (c is not the last entry of its part)
        const double ns = sh_s[wv][c + 1];
        const int ni = sh_i[wv][c + 1];
        ok = ok && (s[e] > ns || (s[e] == ns && id[e] <= ni));
      }
    }
    if (__ballot(!ok) == 0ull && parts <= 64) {
      int pos = 0;
      double hs = -__builtin_inf();
      int hi = INT_MAX;
      if (lane < parts) {
        hs = sh_s[wv][lane * K];
        hi = sh_i[wv][lane * K];
      }
      double my_s = -__builtin_inf();
      int my_id = -1;
      for (int r = 0; r < K; ++r) {
        double smax = hs;
        if (parts <= 16) {  // (wave-uniform) the heads sit in the first DPP row
          smax = fmax(smax, dpp_d<kDppXor1>(smax));
          smax = fmax(smax, dpp_d<kDppXor2>(smax));
          smax = fmax(smax, dpp_d<kDppHalfMirror>(smax));
          smax = fmax(smax, dpp_d<kDppMirror>(smax));
          smax = __shfl(smax, 0);
        } else {
#pragma unroll
          for (int off = 32; off >= 1; off >>= 1) smax = fmax(smax, __shfl_xor(smax, off));
        }
        if (smax == -__builtin_inf()) break;  // no valid candidate left (wave-uniform)
        unsigned long long who = __ballot(lane < parts && hs == smax);
        if (who & (who - 1ull)) {  // equal scores: lowest row id first, then lowest part
          int idc = (lane < parts && hs == smax) ? hi : INT_MAX;
#pragma unroll
          for (int off = 32; off >= 1; off >>= 1) idc = min(idc, __shfl_xor(idc, off));
          who = __ballot(lane < parts && hs == smax && hi == idc);
        }
        const int bl = __ffsll((long long)who) - 1;
        const int wid = __shfl(hi, bl);
        if (lane == r) {
          my_s = smax;
          my_id = wid;
        }
        if (lane == bl) {
          ++pos;
          hs = pos < K ? sh_s[wv][lane * K + pos] : -__builtin_inf();
          hi = pos < K ? sh_i[wv][lane * K + pos] : INT_MAX;
          if (hi == INT_MAX) hs = -__builtin_inf();
        }
      }
      if (lane < K) {
        out_idx[(size_t)qid * K + lane] = my_id;
        if (out_score) out_score[(size_t)qid * K + lane] = my_s;
      }
      return;
    }
    __builtin_amdgcn_wave_barrier();  // (the general path below reuses sh_i)
  }
  // B = the largest, over the parts that hold K valid entries, of the part's SMALLEST entry is a lower bound of the global K-th best
  // (that part alone holds K candidates >= B): only candidates >= B can make the cut — usually K .. 2K of the parts * K — and only
  // those are ranked. No order is assumed inside a part (round 3 took the part's K-th entry, i.e. required best-first lists with the
  // invalid entries trailing — a precondition the public t2l_merge_topk never stated); sorted inputs give the same bound.
  // (All scores equal: everybody survives, the loop below is the full all-pairs count.) The bound only has to be a LOWER bound: it is
  // taken in float32 rounded towards -inf.
  // per-part minimum through LDS integer atomics on the order-preserving integer image of the f32 value (one ds_min per candidate
  // instead of K dependent reads per part: the first form of this pass cost the merge 3 us)
  int* sh_min = &sh_i[wv][0];  // (sh_i takes the survivors' ids only after this phase)
  auto ord = [](float f) { const int b = __float_as_int(f); return b ^ ((b >> 31) & 0x7fffffff); };  // monotone: f < g <=> ord(f) < ord(g)
  auto unord = [](int o) { return __int_as_float(o ^ ((o >> 31) & 0x7fffffff)); };
  for (int p = lane; p < parts; p += 64) sh_min[p] = INT_MAX;
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = lane + 64 * e;
    if (e < ne && c < total) atomicMin(&sh_min[part_of[e]], ord(id[e] != INT_MAX ? __double2float_rd(s[e]) : -__builtin_inff()));
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  float bnd32 = -__builtin_inff();
  for (int p = lane; p < parts; p += 64) bnd32 = fmaxf(bnd32, unord(sh_min[p]));  // a part with an invalid entry has minimum -inf
  const double bnd = (double)wave_max_f32(bnd32, __builtin_inff());
  __builtin_amdgcn_wave_barrier();  // (sh_f is dead: sh_i may be overwritten)
  int n_s = 0;
  bool sv[4];
  int my_pos[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sv[e] = e < ne && id[e] != INT_MAX && s[e] >= bnd;
    const unsigned long long m = __ballot(sv[e]);
    my_pos[e] = n_s + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
    if (sv[e]) {
      sh_s[wv][my_pos[e]] = s[e];
      sh_i[wv][my_pos[e]] = id[e];
    }
    n_s += __popcll(m);
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own LDS writes are visible to its reads
  int rank[4] = {0, 0, 0, 0};
#pragma unroll 4  // (the LDS reads of four candidates in flight: one read-wait per candidate made the loop latency-bound)
  for (int o = 0; o < n_s; ++o) {
    const double os = sh_s[wv][o];
    const int oi = sh_i[wv][o];
#pragma unroll
    for (int e = 0; e < 4; ++e)  // (score desc, row id asc, then position: a duplicated (score, id) pair — row ids are meant to be unique
      if (e < ne)                //  across parts — still gets two different ranks: no slot is written twice, none is left out)
        rank[e] += (os > s[e] || (os == s[e] && (oi < id[e] || (oi == id[e] && o < my_pos[e])))) ? 1 : 0;
  }
  // the survivors' ranks are a permutation of 0..n_s-1 and n_s >= min(K, valid candidates): slots [n_s, K) — and only those — take
  // the default fill (no slot has two writers)
  if (lane >= n_s && lane < K) {
    out_idx[(size_t)qid * K + lane] = -1;
    if (out_score) out_score[(size_t)qid * K + lane] = -__builtin_inf();
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (sv[e] && rank[e] < K) {
      out_idx[(size_t)qid * K + rank[e]] = id[e];
      if (out_score) out_score[(size_t)qid * K + rank[e]] = s[e];
    }
}

// (score, row id) pairs as one f64[.,2] record (row ids are exact in f64): lets the sharded search exchange ONE buffer
__global__ __launch_bounds__(256) void pack_pairs_kernel(const int32_t* __restrict__ idx, const double* __restrict__ score,
                                                         int n, double* __restrict__ pairs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    pairs[2 * i] = score[i];
    pairs[2 * i + 1] = (double)idx[i];
  }
}

int pack_impl(t2l_ctx* ctx, const int32_t* idx, const double* score, int n, double* pairs, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(pack_pairs_kernel, dim3((n + 255) / 256), dim3(256), 0, s, idx, score, n, pairs);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

int merge_pairs_impl(t2l_ctx* ctx, const double* pairs, int parts, int Q, int K, int32_t* out_idx, double* out_score,
                     hipStream_t s) {
  if (parts * K > 256) return fail(ctx, T2L_EINVAL, "t2l_merge_pairs: parts * k must be <= 256");
  hipLaunchKernelGGL((merge_kernel<true>), dim3((Q + 3) / 4), dim3(256), 0, s, (const char*)nullptr, (size_t)0, (const char*)pairs,
                     (size_t)Q * K * 16, parts, Q, K, out_idx, out_score);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

int merge_impl(t2l_ctx* ctx, const int32_t* idx, const double* score, int parts, int Q, int K, int32_t* out_idx,
               double* out_score, hipStream_t s) {
  if (parts * K > 256) return fail(ctx, T2L_EINVAL, "t2l_merge_topk: parts * k must be <= 256");
  hipLaunchKernelGGL((merge_kernel<false>), dim3((Q + 3) / 4), dim3(256), 0, s, (const char*)idx, (size_t)Q * K * 4, (const char*)score,
                     (size_t)Q * K * 8, parts, Q, K, out_idx, out_score);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

// every rank's {ids i32[Q][K] | scores f64[Q][K] at score_offset} block, all-gathered back to back: merged without a pack launch
int merge_gathered_impl(t2l_ctx* ctx, const void* blocks, int64_t block_bytes, int64_t score_offset, int parts, int Q, int K,
                        int32_t* out_idx, double* out_score, hipStream_t s) {
  if (parts * K > 256) return fail(ctx, T2L_EINVAL, "t2l_merge_gathered: parts * k must be <= 256");
  if (score_offset % 8 || block_bytes % 8 || score_offset < (int64_t)Q * K * 4 || block_bytes < score_offset + (int64_t)Q * K * 8)
    return fail(ctx, T2L_EINVAL, "t2l_merge_gathered: a block is {i32[Q][K] ids, f64[Q][K] scores at an 8-byte aligned score_offset}");
  event_begin(ctx, "merge", s);
  hipLaunchKernelGGL((merge_kernel<false>), dim3((Q + 3) / 4), dim3(256), 0, s, (const char*)blocks, (size_t)block_bytes,
                     (const char*)blocks + score_offset, (size_t)block_bytes, parts, Q, K, out_idx, out_score);
  event_end(ctx, "merge", s);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

}  // namespace t2l
