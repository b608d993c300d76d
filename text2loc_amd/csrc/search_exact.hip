// The exact stage of the search on the matrix pipe: float64 MFMA scan for the queries no certificate could settle.
//
// When a database is so tightly clustered that the scores of a query's best rows lie closer together than the error
// band of every reduced-precision scan (f16: ~1e-3 |q||c|, split-bf16: ~2e-5) — what an encoder over overlapping cells
// produces — the certificate fails for most queries and the work falls to the exact float64 ranking the reference
// computes (training/coarse.py:119-125). Round 1 did that with one VALU dot product per (query, row) pair, the query
// broadcast from LDS: 12.6 ms per 4,096 flagged queries at N = 11,259. Here:
//
//   exactd_kernel      one workgroup per 16 flagged queries; every wave holds the 16 queries as float64 B operands
//                      (64 k-steps of v_mfma_f64_16x16x4_f64, products of f32 values are exact in f64) and streams its own
//                      16-row tiles of the f32 DB HBM/L2 -> LDS (LDS-DMA, private double buffer, no workgroup barrier in
//                      the loop), two accumulator chains per tile. Scores enter per-lane sorted lists of float64 KEYS
//                      (low mantissa bits = tile ordinal and row slot). The workgroup then takes the top-L keys per
//                      query, re-scores those rows with the re-rank's arithmetic, orders by (score desc, row asc) and
//                      certifies against the (L+1)-th key / the floors of full lists with the key truncation
//                      (2^(code_bits-51) relative) and the MFMA's accumulation error (2^-44 |q| max|c|) as slack.
//   exact_list_kernel  the last resort for what is left (more than L exact ties, non-finite inputs): the float64 VALU
//                      scan of round 1 over a list of queries.
//
// Roofline: f64 MFMA (78.6 TFLOP/s dense), 2*Q_flagged*N*D FLOP; DB bytes are re-read per 16-query block from L2.
#include "search_dev.h"

namespace t2l {

typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int kExactQ = 16;    // flagged queries per workgroup pass (one 16-column MFMA block, shared by the 4 waves)
constexpr int kExactLL = 8;    // per-lane list length
constexpr int kExactRows = 16; // DB rows per MFMA tile
constexpr int kExactTileBytes = kExactRows * kD * 4;  // 16 KiB of f32

template <int L, int I = L - 1>
__device__ __forceinline__ void ins_key_f64(double (&s)[L], double x) {  // descending list, element I reads OLD neighbours
  if constexpr (I == 0) {
    s[0] = fmax(s[0], x);
  } else {
    s[I] = fmax(fmin(s[I - 1], x), s[I]);
    ins_key_f64<L, I - 1>(s, x);
  }
}

__device__ __forceinline__ double make_key_f64(double v, unsigned long long mask, unsigned code) {
  return __longlong_as_double((long long)(((unsigned long long)__double_as_longlong(v) & mask) | code));
}

__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  return v;
}

__global__ __launch_bounds__(256, 1) void exactd_kernel(const float* __restrict__ db, int n_rows, const float* __restrict__ q,
                                                        int K, int L, int row_offset, int code_bits,
                                                        const int32_t* __restrict__ list, const int32_t* __restrict__ n_list_ptr,
                                                        const float* __restrict__ db_norm_max,
                                                        int32_t* __restrict__ out_idx, double* __restrict__ out_score,
                                                        int32_t* __restrict__ list_out, int32_t* __restrict__ n_out_ptr,
                                                        int32_t* __restrict__ served) {
  extern __shared__ __attribute__((aligned(16))) char xsmem[];
  __shared__ int sel_row[4][32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform_wave_id();
  const int col = lane & 15, kq = lane >> 4;  // MFMA lane roles: A row / B column = lane & 15, k slot = lane >> 4
  const int n_list = *n_list_ptr;
  const int n_tiles = (n_rows + kExactRows - 1) / kExactRows;
  const int nj = wave < n_tiles ? (n_tiles - wave + 3) / 4 : 0;  // this wave's tiles: wave, wave + 4, ...
  const unsigned long long kmask = ~((1ull << code_bits) - 1ull);
  char* tiles = xsmem + wave * (2 * kExactTileBytes);
  const unsigned tiles_lds = lds_addr_of(tiles);
  // LDS tile: 16 rows x 1 KiB, 16-byte chunk c of row r at chunk c ^ r (low 4 bits): the 16 lanes of a k slot read 16
  // different bank groups. DMA piece i = row i: lane l lands at chunk l and fetches global chunk l ^ i.
  const unsigned read_row = col * 1024 + kq * 256;

  for (int blk = blockIdx.x; blk * kExactQ < n_list; blk += gridDim.x) {
    __syncthreads();  // the previous pass is done with the merge area
    const int qi = blk * kExactQ + col;
    const int qid = list[min(qi, n_list - 1)];
    double qb[64];  // B operand: element t <-> k = 64*kq + t of query `col`
    {
      const float4* qp = reinterpret_cast<const float4*>(q + (size_t)qid * kD + kq * 64);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float4 v = qp[i];
        qb[4 * i] = (double)v.x;
        qb[4 * i + 1] = (double)v.y;
        qb[4 * i + 2] = (double)v.z;
        qb[4 * i + 3] = (double)v.w;
      }
    }
    double ls[kExactLL];
#pragma unroll
    for (int i = 0; i < kExactLL; ++i) ls[i] = -__builtin_inf();

    auto fetch = [&](int j, int buf) {
      const char* src = reinterpret_cast<const char*>(db) + (size_t)(wave + 4 * j) * kExactTileBytes;
      const unsigned dst = tiles_lds + buf * kExactTileBytes;
#pragma unroll
      for (int i = 0; i < 16; ++i) lds_dma_row(dst + i * 1024, (unsigned)((lane ^ i) << 4), src + i * 1024);
    };
    if (nj > 0) fetch(0, 0);
    for (int j = 0; j < nj; ++j) {
      const int buf = j & 1;
      if (j + 1 < nj) {
        fetch(j + 1, buf ^ 1);  // that buffer's reads finished in the previous iteration (lgkmcnt(0) below)
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      const char* tb = tiles + buf * kExactTileBytes + read_row;
      float4 a[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) a[s] = *reinterpret_cast<const float4*>(tb + ((s ^ col) << 4));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 16; ++s) {  // two chains: consecutive MFMAs never wait on each other
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a[s].x, qb[4 * s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a[s].y, qb[4 * s + 1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a[s].z, qb[4 * s + 2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a[s].w, qb[4 * s + 3], acc1, 0, 0, 0);
      }
      // f64 C/D layout: lane (col, kq) holds rows kq + 4e of the tile, e = 0..3
      const int row0 = (wave + 4 * j) * kExactRows + kq;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double sc = acc0[e] + acc1[e];
        ins_key_f64<kExactLL>(ls, row0 + 4 * e < n_rows ? make_key_f64(sc, kmask, (unsigned)(j * 4 + e)) : -__builtin_inf());
      }
    }

    // ---- merge: [16 queries][16 lists = 4 waves x 4 k slots][8 keys] in LDS, then wave w settles queries 4w .. 4w+3
    __syncthreads();  // every wave is past its tile buffers
    double* keys = reinterpret_cast<double*>(xsmem);
#pragma unroll
    for (int i = 0; i < kExactLL; ++i) keys[(col * 16 + wave * 4 + kq) * kExactLL + i] = ls[i];
    __syncthreads();
    for (int qq = wave * 4; qq < wave * 4 + 4; ++qq) {
      if (blk * kExactQ + qq >= n_list) break;  // wave-uniform
      const int my_qid = list[blk * kExactQ + qq];
      const double* kq_keys = keys + qq * 128;
      const double k0 = kq_keys[lane], k1 = kq_keys[64 + lane];
      int r0 = 0, r1 = 0;
      for (int o = 0; o < 128; ++o) {  // rank by (key desc, slot asc); keys of different rows differ in their code bits or
        const double ko = kq_keys[o];  // come from different lists: equal keys are ordered by slot
        r0 += (ko > k0 || (ko == k0 && o < lane)) ? 1 : 0;
        r1 += (ko > k1 || (ko == k1 && o < 64 + lane)) ? 1 : 0;
      }
      auto row_of = [&](double key, int slot) {  // slot = list * 8 + i, list = wave' * 4 + kq'
        const int lst = slot >> 3;
        const unsigned code = (unsigned)((unsigned long long)__double_as_longlong(key) & ~kmask);
        return ((lst >> 2) + 4 * (int)(code >> 2)) * kExactRows + (lst & 3) + 4 * (int)(code & 3u);
      };
      const bool v0 = k0 != -__builtin_inf(), v1 = k1 != -__builtin_inf();
      if (lane < 32) sel_row[wave][lane] = INT_MAX;
      if (r0 < L) sel_row[wave][r0] = v0 ? row_of(k0, lane) : INT_MAX;
      if (r1 < L) sel_row[wave][r1] = v1 ? row_of(k1, 64 + lane) : INT_MAX;
      // g bounds the key of every row that is NOT re-scored: the (L+1)-th key, or the floor of a full list (its 8th key:
      // everything that list dropped lies at or below it)
      double g = -__builtin_inf();
      if (r0 == L) g = k0;
      if (r1 == L) g = k1;
      if ((lane & 7) == 7) g = fmax(g, fmax(k0, k1));  // slots 8m + 7: the lists' floors (-inf when the list is not full)
      g = wave_max_f64(g);
      // exact re-score with the re-rank's arithmetic (products of f32 values are exact in f64)
      const float4 qv = reinterpret_cast<const float4*>(q + (size_t)my_qid * kD)[lane];
      const double qn = wave_sum_f64((double)qv.x * qv.x + (double)qv.y * qv.y + (double)qv.z * qv.z + (double)qv.w * qv.w);
      double my_d = -__builtin_inf();
      int my_row = INT_MAX;
      for (int c = 0; c < L; ++c) {
        const int row = sel_row[wave][c];
        if (row == INT_MAX) continue;  // wave-uniform
        const float4 dv = reinterpret_cast<const float4*>(db + (size_t)row * kD)[lane];
        const double d = wave_sum_f64((double)dv.x * qv.x + (double)dv.y * qv.y + (double)dv.z * qv.z + (double)dv.w * qv.w);
        if (lane == c) {
          my_d = d;
          my_row = row;
        }
      }
      int rank = 0;
      for (int o = 0; o < L; ++o) {
        const double od = __shfl(my_d, o);
        const int orow = __shfl(my_row, o);
        rank += (od > my_d || (od == my_d && orow < my_row)) ? 1 : 0;
      }
      const bool valid = lane < L && my_row != INT_MAX;
      if (lane < K) {
        out_idx[(size_t)my_qid * K + lane] = -1;
        if (out_score) out_score[(size_t)my_qid * K + lane] = -__builtin_inf();
      }
      if (valid && rank < K) {
        out_idx[(size_t)my_qid * K + rank] = my_row + row_offset;
        if (out_score) out_score[(size_t)my_qid * K + rank] = my_d;
      }
      bool certified = true;
      if (g != -__builtin_inf()) {  // something was not re-scored
        certified = false;
        const unsigned long long kth = __ballot(valid && rank == K - 1);
        if (kth != 0ull) {
          const double dK = __shfl(my_d, __ffsll((long long)kth) - 1);
          const double slack = fabs(g) * ldexp(1.0, code_bits - 51) + ldexp(1.0, -44) * sqrt(qn) * (double)(*db_norm_max);
          certified = dK > g + slack;
        }
      }
      if (lane == 0) {
        atomicAdd(served, 1);
        if (!certified) list_out[atomicAdd(n_out_ptr, 1)] = my_qid;
      }
    }
  }
}

// last resort: the float64 VALU scan (search_dev.h) for the queries on `list`
__global__ __launch_bounds__(256) void exact_list_kernel(const float* __restrict__ db, int n_rows, const float* __restrict__ q,
                                                         int K, int row_offset, const int32_t* __restrict__ list,
                                                         const int32_t* __restrict__ n_list_ptr, int32_t* __restrict__ out_idx,
                                                         double* __restrict__ out_score, int32_t* __restrict__ fb_count) {
  __shared__ double qs[kD];
  __shared__ double red_s[256];
  __shared__ int red_i[256];
  __shared__ int red_t[256];
  const int n_list = *n_list_ptr;
  for (int i = blockIdx.x; i < n_list; i += gridDim.x) {
    const int qid = list[i];
    __syncthreads();
    qs[threadIdx.x] = (double)q[(size_t)qid * kD + threadIdx.x];
    if (threadIdx.x == 0) atomicAdd(&fb_count[0], 1);
    __syncthreads();
    exact_scan<32>(db, n_rows, qs, K, row_offset, out_idx + (size_t)qid * K, out_score ? out_score + (size_t)qid * K : nullptr,
                   red_s, red_i, red_t);
  }
}

// flags layout (search.hip): [0,Q) flag, [Q,2Q) threshold keys, [2Q,3Q) the re-rank's flagged list, [3Q,4Q) the list the
// fallback kernel defers to the exact stage, [4Q,5Q) what the exact stage could not certify
int exact_stage_impl(t2l_ctx* ctx, const float* db, int n_rows, int row_offset, const float* q, int Q, int K, int32_t* out_idx,
                     double* out_score, hipStream_t s) {
  const int L = K <= 10 ? 16 : 32;
  const int n_tiles = (n_rows + kExactRows - 1) / kExactRows;
  int code_bits = 2;
  while ((1 << code_bits) < ((n_tiles + 3) / 4) * 4) ++code_bits;
  const size_t lds = (size_t)4 * 2 * kExactTileBytes;
  static PerDeviceOnce once;
  if (once.need(ctx->device)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&exactd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)((size_t)4 * 2 * kExactTileBytes));
    once.mark(ctx->device);
  }
  int32_t* list3 = ctx->flags + (size_t)3 * Q;
  int32_t* list4 = ctx->flags + (size_t)4 * Q;
  event_begin(ctx, "search_exact", s);
  hipLaunchKernelGGL(exactd_kernel, dim3(min((Q + kExactQ - 1) / kExactQ, 256)), dim3(256), lds, s, db, n_rows, q, K, L, row_offset,
                     code_bits, list3, ctx->fb_count + 4, ctx->db_norm_max, out_idx, out_score, list4, ctx->fb_count + 6,
                     ctx->fb_count + 7);
  T2L_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(exact_list_kernel, dim3(min(Q, 128)), dim3(256), 0, s, db, n_rows, q, K, row_offset, list4,
                     ctx->fb_count + 6, out_idx, out_score, ctx->fb_count);
  event_end(ctx, "search_exact", s);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

}  // namespace t2l
