// Few queries (Q <= 16) against a shard that fits the caches: the whole search as ONE launch, exact from the start.
//
// The reference's own granularity is one query at a time (training/coarse.py:119-125: `scores = cell_encodings[:] @ text_encoding`
// in float64, full argsort) — SURVEY.md 8d's "Q_b = 1" point, HBM / cache-bandwidth bound: 4 * N * D bytes per query. Until round 4
// such a call took the batched path's two launches (f16 MFMA candidate scan for 256 query slots, float64 re-rank + certificate):
// 21.3 us for ONE query, nearly all of it dispatch, kernel-boundary and latency chains. For a handful of queries the arithmetic
// is nothing: this kernel computes the float64 scores of ALL rows directly (the re-rank's own arithmetic: exact f32 x f32
// products summed in float64 in a fixed order, so a score is bit-identical to what the batched path reports for that row) and
// ranks them — no reduced-precision scan, no certificate, no candidate lists, no second launch:
//
//   phase 1 (every workgroup): rows [wg * R, wg * R + R) of the f32 DB against NQ <= 4 queries (blockIdx.y picks the group of 4):
//            a 16-lane DPP row per DB row, four rows per wave pass, 256-byte coalesced loads, three row groups in flight; the scores
//            meet in LDS and the workgroup ranks its R <= 256 rows by counting — (score desc, row asc), a strict total order, one
//            (row, query) pair per thread — and PUBLISHES its best K as 16-byte entries {float64 score bits, int32 row, 0} with
//            write-through (sc1) buffer stores;
//   phase 2 (the workgroup that arrives LAST on the slice's ticket counter): reads the G published lists' HEADS with sc1 loads (one
//            query and K <= 16: the whole lists, same round trip), takes as bound T the K-th best head — K lists have a head at or
//            ahead of T, so at least K entries are and nothing behind T can be in the top K —, compacts the entries at or ahead of T
//            (typically K .. 2 K of the G * K) and ranks them by counting: one wave per query, the K ids / scores go out.
//
// Cross-workgroup visibility follows MI355X_MICROARCH.md / cdna_hip_programming.md (in-launch split-K reduction): 16-byte
// `raw_buffer_store_b128 ... sc1` (write-through) -> every wave `s_waitcnt vmcnt(0)` -> `__syncthreads()` -> lane 0 relaxed agent
// `fetch_add` on the ticket; the last arriver reads with sc1 buffer loads. No workgroup ever waits for another one: nothing here
// depends on co-residency. The ticket is a running total (the host passes the value the last arriver will draw), so no reset, no
// memset launch, and a grid of another size next call cannot be confused with this one.
//
// Roofline: the f32 rows once per group of 4 queries = 1 KiB per row (from L2 / Infinity Cache once resident). Measured: bench.py ->
// search_latency.
#include <limits.h>

#include "t2l_internal.h"
#include "search_dev.h"

namespace t2l {

constexpr int kSmallMaxQ = 16;    // queries per call this path takes (4 per blockIdx.y slice)
constexpr int kSmallNQ = 4;
constexpr int kSmallMaxR = 256;   // rows per workgroup (one per thread in the ranking pass)
constexpr int kSmallMaxG = 256;   // workgroups per slice = published lists the last arriver merges (one per thread)

// order-preserving image of a float64 score: larger score <-> larger key; -0 and +0 share a key (they compare equal), NaN ranks
// below everything (numpy's argsort puts NaN last, too)
__device__ __forceinline__ unsigned long long small_key(double s) {
  if (s != s) return 0ull;
  const unsigned long long b = (unsigned long long)__double_as_longlong(s + 0.0);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
// (key, row) a ranks ahead of b: score desc, row asc
__device__ __forceinline__ bool small_ahead(unsigned long long ka, int ra, unsigned long long kb, int rb) {
  return ka > kb || (ka == kb && ra < rb);
}

// Ranking by counting, the cheap way. A full comparison of two (64-bit key, row) pairs is ~7 VALU instructions, two of them 64-bit
// compares; counted one LDS read at a time it made the ranking the most expensive part of a call (measured: +14 us for four queries).
// So: count on the HIGH 32 bits of the keys (sign, exponent, 20 mantissa bits: one 32-bit compare + one add per entry, four keys per
// 16-byte LDS read); only a key whose high word occurs more than once in the list — scores within 2^-20 of each other, exact ties —
// is ranked again with the full comparison (a divergent, rare second pass).
struct SmallCount {
  int gt, eq;
};
__device__ __forceinline__ SmallCount small_count32(const unsigned* khi, int n8, unsigned mh) {  // n8: multiple of 8, 16-byte aligned
  SmallCount c{0, 0};
  for (int o = 0; o < n8; o += 8) {
    const uint4 a = *reinterpret_cast<const uint4*>(khi + o), b = *reinterpret_cast<const uint4*>(khi + o + 4);
    c.gt += (a.x > mh) + (a.y > mh) + (a.z > mh) + (a.w > mh) + (b.x > mh) + (b.y > mh) + (b.z > mh) + (b.w > mh);
    c.eq += (a.x == mh) + (a.y == mh) + (a.z == mh) + (a.w == mh) + (b.x == mh) + (b.y == mh) + (b.z == mh) + (b.w == mh);
  }
  return c;
}
// the full comparison: how many of keys[0 .. n) rank ahead of (mk, mrow), the row of keys[o] being rows ? rows[o] : o
__device__ __forceinline__ int small_rank_full(const unsigned long long* keys, const int* rows, int n, unsigned long long mk, int mrow) {
  int rank = 0;
  for (int o = 0; o < n; ++o) rank += small_ahead(keys[o], rows ? rows[o] : o, mk, mrow) ? 1 : 0;
  return rank;
}
// rank of entry (mk, mrow), itself one of the n entries (padded to n8 with high word 0 — if that collides with a real high word
// the full pass sorts it out: pads are not among the first n)
__device__ __forceinline__ int small_rank(const unsigned* khi, const unsigned long long* keys, const int* rows, int n, int n8,
                                          unsigned long long mk, int mrow) {
  const SmallCount c = small_count32(khi, n8, (unsigned)(mk >> 32));
  return c.eq == 1 ? c.gt : small_rank_full(keys, rows, n, mk, mrow);
}

// every lane gets the wave's maximum (VALU only: DPP row steps + the two permlane swaps)
__device__ __forceinline__ unsigned small_wave_max_u32(unsigned v) {
  v = max(v, dpp_u<kDppXor1>(v));
  v = max(v, dpp_u<kDppXor2>(v));
  v = max(v, dpp_u<kDppHalfMirror>(v));
  v = max(v, dpp_u<kDppMirror>(v));
  {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = max(r[0], r[1]);
  }
  {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    v = max(r[0], r[1]);
  }
  return v;
}

constexpr int kSmallSurv = 256;   // survivors per query the last arriver ranks out of LDS (beyond: straight from memory, slowly and correctly)
constexpr int kSmallFuseK = 16;   // one query per slice and K <= 16: the last arriver reads heads AND lists in one round trip
struct __attribute__((aligned(16))) SmallShared {
  unsigned khi[kSmallNQ][kSmallMaxR];            // phase 1: high words of the workgroup's score keys; phase 2: of the lists' heads
  unsigned long long key[kSmallNQ][kSmallMaxR];  // ... the full keys
  double score[kSmallNQ][kSmallMaxR];            // phase 1 only
  int hrow[kSmallNQ][kSmallMaxG];                // phase 2: rows of the heads
  int sl[kSmallNQ][kSmallMaxG];                  // phase 2: the lists whose head passed the bound
  unsigned skhi[kSmallNQ][kSmallSurv];           // phase 2: the survivors of query j (ranked by wave j)
  unsigned long long skey[kSmallNQ][kSmallSurv];
  double sscore[kSmallNQ][kSmallSurv];
  int srow[kSmallNQ][kSmallSurv];
  unsigned thr[kSmallNQ];                        // phase 2: the bound (high word of the K-th best head)
  int n_lists[kSmallNQ], n_surv[kSmallNQ];
  int last;
};

template <int NQ>
__global__ __launch_bounds__(256) void smallq_kernel(const float* __restrict__ db, int n_rows, int R, const float* __restrict__ q, int Q,
                                                     int K, int row_offset, uint4* __restrict__ part, unsigned part_bytes,
                                                     unsigned* __restrict__ ticket, unsigned last_ticket, int32_t* __restrict__ out_idx,
                                                     double* __restrict__ out_score, int32_t* __restrict__ fb_count) {
  __shared__ SmallShared sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int seg = lane & 15, grp = lane >> 4;
  const int G = gridDim.x, wg = blockIdx.x, q0 = blockIdx.y * kSmallNQ;
  const int r0 = wg * R, rn = max(0, min(R, n_rows - r0));  // this workgroup's rows [r0, r0 + rn)
  const int nqv = min(NQ, Q - q0);                          // valid queries of this slice

  // ---- phase 1: float64 scores of the workgroup's rows (the re-rank's arithmetic: rerank_query / wg_exact_scan)
  double qd[NQ][16];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const float4* qp = reinterpret_cast<const float4*>(q + (size_t)min(q0 + j, Q - 1) * kD) + seg;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = qp[16 * i];
      qd[j][4 * i] = (double)v.x;
      qd[j][4 * i + 1] = (double)v.y;
      qd[j][4 * i + 2] = (double)v.z;
      qd[j][4 * i + 3] = (double)v.w;
    }
  }
  const int n_groups = (rn + 3) / 4;
  // three row groups in flight per wave — at the default 44 rows per workgroup that is ALL of a wave's rows behind one memory round
  // trip, issued together with the query loads above
  constexpr int kFly = 3;
  for (int g = wave; g < n_groups; g += 4 * kFly) {
    float4 rv[kFly][4];
#pragma unroll
    for (int u = 0; u < kFly; ++u) {
      const int row = min(r0 + 4 * (g + 4 * u) + grp, n_rows - 1);
      const float4* rp = reinterpret_cast<const float4*>(db + (size_t)row * kD) + seg;
#pragma unroll
      for (int i = 0; i < 4; ++i) rv[u][i] = rp[16 * i];
    }
#pragma unroll
    for (int u = 0; u < kFly; ++u) {
      const int gg = g + 4 * u;
      if (gg >= n_groups) break;  // wave-uniform
      const int lrow = 4 * gg + grp;
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        double d0 = 0.0, d1 = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          d0 += (double)rv[u][i].x * qd[j][4 * i];
          d1 += (double)rv[u][i].y * qd[j][4 * i + 1];
          d0 += (double)rv[u][i].z * qd[j][4 * i + 2];
          d1 += (double)rv[u][i].w * qd[j][4 * i + 3];
        }
        const double d = row16_sum_f64(d0 + d1);
        if (seg == 0 && lrow < rn) {
          const unsigned long long k = small_key(d);
          sh.score[j][lrow] = d;
          sh.key[j][lrow] = k;
          sh.khi[j][lrow] = (unsigned)(k >> 32);
        }
      }
    }
  }
  const int rn8 = (rn + 7) & ~7;
#pragma unroll
  for (int j = 0; j < NQ; ++j)
    if (tid >= rn && tid < rn8) sh.khi[j][tid] = 0u;  // pad to a multiple of 8
  __syncthreads();
  // rank by counting — one (row, query) pair per thread while they fit — and publish the best K of this workgroup, write-through:
  // one 16-byte entry {score bits, row, 0} per store (aux 16 = sc1; narrow sc1 stores are one fabric write each)
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(part, 0, (int)part_bytes, 0x00020000);
  typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
  auto entry = [](double sc, int row) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(sc);
    return u32x4s{(unsigned)b, (unsigned)(b >> 32), (unsigned)row, 0u};
  };
  const unsigned pbase = (unsigned)((blockIdx.y * kSmallNQ * G + wg) * K);  // entry index of this workgroup's list of query 0
  for (int t = tid; t < rn * nqv; t += 256) {
    const int j = t / rn, row = t - j * rn;
    const int rank = small_rank(sh.khi[j], sh.key[j], nullptr, rn, rn8, sh.key[j][row], row);
    if (rank < K) __builtin_amdgcn_raw_buffer_store_b128(entry(sh.score[j][row], r0 + row), rsrc, (pbase + (unsigned)(j * G * K + rank)) * 16u, 0, 16);
  }
  if (rn < K)  // fewer rows than K: the tail of the list is empty (-inf, row -1)
    for (int t = tid; t < (K - rn) * nqv; t += 256) {
      const int j = t / (K - rn), e = rn + t - j * (K - rn);
      __builtin_amdgcn_raw_buffer_store_b128(entry(-__builtin_inf(), -1), rsrc, (pbase + (unsigned)(j * G * K + e)) * 16u, 0, 16);
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have left
  __syncthreads();                                     // ... and every wave's
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(ticket + blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh.last = (t + 1u == last_ticket) ? 1 : 0;
  }
  __syncthreads();
  if (!sh.last) return;  // workgroup-uniform

  // ---- phase 2 (one workgroup per slice): merge the G published lists
  // Every list is sorted and holds its workgroup's best K, so the global top K is the top K of the union. Bound T: any K-th best
  // among K or more list HEADS — those K lists have a head at or ahead of T, so at least K entries are, and nothing behind T can be
  // in the top K. Each wave takes the K-th best of ITS 64 heads and T is the best of the four — typically 2 K .. 4 K survivors.
  // (The K-th entry of the best list — what the sharded merge uses — is useless here: a list covers a few dozen rows, its K-th entry
  // is a mediocre score that most published entries pass.) The phase is a chain of dependent memory round trips (~1 us each), so
  // every stage issues ALL its loads before it consumes one, the stages run for the slice's queries together, and the last two
  // stages give every query its own wave. One query and K <= 16: heads and lists arrive in ONE round trip (FUSE).
  // (the batched path's counters are left alone: a small call between two batched calls must not erase the earlier call's report
  // card — the host answers t2l_search_fallbacks with zeros after a small call, ctx->last_search_small)
  const unsigned e0 = (unsigned)(blockIdx.y * kSmallNQ * G * K);  // first entry of this slice
  auto load_entry = [&](unsigned e) { return __builtin_amdgcn_raw_buffer_load_b128(rsrc, e * 16u, 0, 16); };  // (aux 16 = sc1: L1-bypassing)
  auto entry_bits = [](u32x4s v) { return (unsigned long long)v[0] | ((unsigned long long)v[1] << 32); };
  const bool fuse = NQ == 1 && K <= kSmallFuseK;
  u32x4s mine[NQ == 1 ? kSmallFuseK : 1];  // FUSE: this thread's whole list
  // stage A: the heads (thread = list)
  unsigned long long hk[NQ];
  int hr[NQ];
  {
    u32x4s hv[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      hv[j] = u32x4s{0u, 0u, 0xFFFFFFFFu, 0u};
      if (j < nqv && tid < G) hv[j] = load_entry(e0 + (unsigned)((j * G + tid) * K));
    }
    if constexpr (NQ == 1) {
      if (fuse) {
#pragma unroll
        for (int e = 1; e < kSmallFuseK; ++e) {
          mine[e] = u32x4s{0u, 0u, 0xFFFFFFFFu, 0u};
          if (tid < G && e < K) mine[e] = load_entry(e0 + (unsigned)(tid * K + e));
        }
      }
      mine[0] = hv[0];
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      hr[j] = (int)hv[j][2];
      hk[j] = hr[j] >= 0 ? small_key(__longlong_as_double((long long)entry_bits(hv[j]))) : 0ull;
      if (hr[j] < 0) hr[j] = INT_MAX;  // (threads beyond G, empty lists: an empty head ranks behind everything)
      sh.key[j][tid] = hk[j];
      sh.khi[j][tid] = (unsigned)(hk[j] >> 32);
      sh.hrow[j][tid] = hr[j];
    }
  }
  if (tid < kSmallNQ) sh.n_lists[tid] = sh.n_surv[tid] = 0;
  __syncthreads();
  // stage B (wave = query): the K-th best head of the query, on the high words — K rounds of a wave maximum over the G <= 256 heads
  // (four per lane), knocking out the maxima found; T = the high word at which K heads are reached. (Ranking every head among its
  // wave's 64 by counting cost ~1,100 instructions per thread for four queries; this is ~20 per round.) Heads whose high words tie
  // count together: the bound only gets looser by what shares a high word with the K-th best head.
  if (wave < nqv) {
    unsigned h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = sh.khi[wave][lane + 64 * i];
    int cnt = 0;
    unsigned T = 0u;
    for (int round = 0; round < K && cnt < K; ++round) {
      const unsigned m = small_wave_max_u32(max(max(h[0], h[1]), max(h[2], h[3])));
      if (m == 0u) break;  // (high word 0: an empty head or a NaN score — never a bound)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool hit = h[i] == m;
        cnt += __popcll(__ballot(hit));
        if (hit) h[i] = 0u;
      }
      T = m;
    }
    if (lane == 0) sh.thr[wave] = cnt >= K ? T : 0u;  // fewer than K heads in all: no bound, every entry survives (the host kept G * K small)
  }
  __syncthreads();
  // stage C (thread = list): the lists whose head reaches the bound are compacted per query (FUSE: their surviving ENTRIES, straight
  // from the registers)
  auto survive = [&](int j, unsigned long long k, double sc, int row) {
    const int slot = atomicAdd(&sh.n_surv[j], 1);
    if (slot < kSmallSurv) {
      sh.skhi[j][slot] = (unsigned)(k >> 32);
      sh.skey[j][slot] = k;
      sh.sscore[j][slot] = sc;
      sh.srow[j][slot] = row;
    }
  };
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    if (j >= nqv) continue;
    const unsigned T = sh.thr[j];
    if (hr[j] != INT_MAX && (unsigned)(hk[j] >> 32) >= T) {
      bool listed = true;
      if constexpr (NQ == 1) {
        if (fuse) {
          listed = false;
#pragma unroll
          for (int e = 0; e < kSmallFuseK; ++e) {
            const int r = (int)mine[e][2];
            if (r < 0) continue;
            const double sc = __longlong_as_double((long long)entry_bits(mine[e]));
            const unsigned long long k = small_key(sc);
            if ((unsigned)(k >> 32) >= T) survive(0, k, sc, r);
          }
        }
      }
      if (listed) sh.sl[j][atomicAdd(&sh.n_lists[j], 1)] = tid;
    }
  }
  __syncthreads();
  // stage D (wave = query): the surviving lists' K entries, all loads in flight at once, survivors compacted into the query's LDS list
  constexpr int kMaxK = T2L_MAX_TOPK;
  const int j = wave;
  if (j < nqv && !fuse) {
    const unsigned eq = e0 + (unsigned)(j * G * K);
    const unsigned T = sh.thr[j];
    const int nl = sh.n_lists[j];
    for (int base = 0; base < nl; base += 64) {
      const bool have = base + lane < nl;
      const int list = have ? sh.sl[j][base + lane] : 0;
      u32x4s ev[kMaxK];
#pragma unroll
      for (int e = 0; e < kMaxK; ++e) {
        ev[e] = u32x4s{0u, 0u, 0xFFFFFFFFu, 0u};
        if (have && e < K) ev[e] = load_entry(eq + (unsigned)(list * K + e));
      }
#pragma unroll
      for (int e = 0; e < kMaxK; ++e) {
        const int r = (int)ev[e][2];
        if (r < 0) continue;
        const double sc = __longlong_as_double((long long)entry_bits(ev[e]));
        const unsigned long long k = small_key(sc);
        if ((unsigned)(k >> 32) >= T) survive(j, k, sc, r);
      }
    }
  }
  __syncthreads();
  // stage E (wave = query): rank the survivors by counting, write the K results
  if (j < nqv) {
    const unsigned eq = e0 + (unsigned)(j * G * K);
    const int ns = sh.n_surv[j];
    int32_t* oi = out_idx + (size_t)(q0 + j) * K;
    double* os = out_score ? out_score + (size_t)(q0 + j) * K : nullptr;
    if (ns <= kSmallSurv) {
      const int ns8 = (ns + 7) & ~7;
      if (lane < ns8 - ns) sh.skhi[j][ns + lane] = 0u;  // pad (this wave ranks the list: wave-local ordering through the LDS queue)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int c = lane; c < ns; c += 64) {
        const int mr = sh.srow[j][c];
        const int rank = small_rank(sh.skhi[j], sh.skey[j], sh.srow[j], ns, ns8, sh.skey[j][c], mr);
        if (rank < K) {
          oi[rank] = mr + row_offset;
          if (os) os[rank] = sh.sscore[j][c];
        }
      }
      if (lane >= ns && lane < K) {  // fewer than K rows in the shard
        oi[lane] = -1;
        if (os) os[lane] = -__builtin_inf();
      }
    } else {
      // more survivors than the LDS list holds (hundreds of published entries tie with or beat the bound: exact ties across many
      // lists at the K-th rank; the host's choice of R excludes the other way there, G < K with G * K > 256): rank every
      // published entry against all of them, straight from memory — slow and correct
      if (lane < K) {
        oi[lane] = -1;
        if (os) os[lane] = -__builtin_inf();
      }
      for (int c = lane; c < G * K; c += 64) {
        const u32x4s vc = load_entry(eq + (unsigned)c);
        const int r = (int)vc[2];
        if (r < 0) continue;
        const unsigned long long sbc = entry_bits(vc);
        const unsigned long long k = small_key(__longlong_as_double((long long)sbc));
        int rank = 0;
        for (int o = 0; o < G * K && rank < K; ++o) {
          const u32x4s vo = load_entry(eq + (unsigned)o);
          if ((int)vo[2] < 0) continue;
          rank += small_ahead(small_key(__longlong_as_double((long long)entry_bits(vo))), (int)vo[2], k, r) ? 1 : 0;
        }
        if (rank < K) {
          oi[rank] = r + row_offset;
          if (os) os[rank] = __longlong_as_double((long long)sbc);
        }
      }
    }
  }
}

bool search_small_applies(const t2l_ctx* ctx, int Q, int K) {
  const int64_t n = ctx->db_rows;
  return ctx->search_small && Q >= 1 && Q <= kSmallMaxQ && K >= 1 && K <= T2L_MAX_TOPK && n > 0 && n <= (int64_t)kSmallMaxG * kSmallMaxR &&
         ctx->search_mode == 0 && ctx->nsplit_override == 0;
}

int search_small_impl(t2l_ctx* ctx, const float* q, int Q, int K, int32_t* out_idx, double* out_score, hipStream_t s) {
  const int n_rows = (int)ctx->db_rows;
  // rows per workgroup: a multiple of 4 (the wave pass), at least 16, as few as fill the chip with <= 256 workgroups per slice
  // workgroups (= published lists) per slice. More of them shorten phase 1 (fewer rows each) and lengthen phase 2 (more lists to
  // publish, read and bound); measured at N = 11,259, us per call issued from C, G = 256 / 192 / 128 / 96:
  //   Q = 1: 12.1 / 11.2 / 10.8 / 11.2    Q = 4: 15.0 / 15.0 / 16.6 / 17.9    Q = 8: 18.8 / 17.6 / 17.9 / 19.0    Q = 16: 27.6 / 24.2 / 21.4 / 23.4
  // (the batched two-launch path: 21.2 / 22.6 / 22.7 / 22.9)
  const int g_default = (Q <= 2 || Q > 8) ? 128 : 192;
  int G = min(kSmallMaxG, max(1, ctx->search_small_wgs > 0 ? ctx->search_small_wgs : g_default));
  G = max(G, (n_rows + kSmallMaxR - 1) / kSmallMaxR);  // (at most 256 rows per workgroup: one per thread in the ranking pass)
  int R = ((n_rows + G - 1) / G + 3) / 4 * 4;
  // small shards: at least 2 K rows per workgroup; and when that leaves fewer than K lists (no K-th best head to bound the merge:
  // every published entry is a survivor) few enough of them that G * K fits the last arriver's LDS list
  R = min(kSmallMaxR, max(R, max(16, (2 * K + 3) / 4 * 4)));
  G = (n_rows + R - 1) / R;
  if (G < K && G * K > kSmallSurv) {
    G = max(1, kSmallSurv / K);
    R = min(kSmallMaxR, ((n_rows + G - 1) / G + 3) / 4 * 4);
    G = (n_rows + R - 1) / R;
  }
  const int slices = (Q + kSmallNQ - 1) / kSmallNQ;
  const size_t n_part = (size_t)slices * kSmallNQ * G * K;
  if (!ctx->small_ticket) {
    T2L_HIP(ctx, hipMalloc(&ctx->small_ticket, sizeof(unsigned) * (kSmallMaxQ / kSmallNQ)));
    T2L_HIP(ctx, hipMemset(ctx->small_ticket, 0, sizeof(unsigned) * (kSmallMaxQ / kSmallNQ)));
    for (auto& b : ctx->small_ticket_base) b = 0u;
  }
  const size_t need = n_part * sizeof(uint4);
  if (ctx->small_part_cap < need) {
    if (ctx->small_part) (void)hipFree(ctx->small_part);
    ctx->small_part = nullptr;
    ctx->small_part_cap = 0;
    T2L_HIP(ctx, hipMalloc(&ctx->small_part, need));
    ctx->small_part_cap = need;
  }
  // the ticket a slice's LAST arriver draws + 1: every slice of this call adds G to its own running total (slices a shorter call
  // does not launch keep theirs: one base per slice)
  unsigned last_ticket = 0;
  {
    // all launched slices must agree on one value (it is a kernel argument): bring the bases of the launched slices level first
    unsigned top = 0;
    bool level = true;
    for (int y = 0; y < slices; ++y) {
      if (y && ctx->small_ticket_base[y] != top) level = false;
      if (!y) top = ctx->small_ticket_base[0];
    }
    if (!level) {  // a call with more slices than any before it since the bases diverged: re-zero (stream-ordered, rare)
      T2L_HIP(ctx, hipMemsetAsync(ctx->small_ticket, 0, sizeof(unsigned) * (kSmallMaxQ / kSmallNQ), s));
      for (auto& b : ctx->small_ticket_base) b = 0u;
      top = 0;
    }
    last_ticket = top + (unsigned)G;
    for (int y = 0; y < slices; ++y) ctx->small_ticket_base[y] = last_ticket;
  }
  const int nq = Q >= 3 ? 4 : Q;  // register budget: the kernel is instantiated for 1, 2 and 4 queries per slice
  event_begin(ctx, "search_small", s);
  const dim3 grid(G, slices);
#define T2L_SMALL(NQv)                                                                                                                         \
  hipLaunchKernelGGL(smallq_kernel<NQv>, grid, dim3(256), 0, s, (const float*)ctx->db, n_rows, R, q, Q, K, (int)ctx->row_offset,               \
                     reinterpret_cast<uint4*>(ctx->small_part), (unsigned)need, ctx->small_ticket, last_ticket, out_idx, out_score, (int32_t*)nullptr)
  if (nq == 1) T2L_SMALL(1);
  else if (nq == 2) T2L_SMALL(2);
  else T2L_SMALL(4);
#undef T2L_SMALL
  event_end(ctx, "search_small", s);
  if (const hipError_t le = hipGetLastError(); le != hipSuccess) {
    // no workgroup drew a ticket, but the host bases are G ahead: bring both back to zero (stream-ordered) so that the next call's
    // last arriver exists — otherwise every later call would return T2L_OK with unwritten outputs
    (void)hipMemsetAsync(ctx->small_ticket, 0, sizeof(unsigned) * (kSmallMaxQ / kSmallNQ), s);
    for (auto& b : ctx->small_ticket_base) b = 0u;
    return fail(ctx, T2L_EHIP, std::string("smallq_kernel launch: ") + hipGetErrorString(le));
  }
  ctx->last_search_small = true;  // this call has no certificate and no fallback: its counters are all zero by construction
  return T2L_OK;
}

}  // namespace t2l
