// Coarse retrieval: all-pairs query x submap-DB similarity + top-k, fused (gfx950 / CDNA4).
//
// Replaces the host loop of training/coarse.py:119-125 — per query a float64 `cell_encodings @ t`
// (N x 256) and a full argsort — by three launches on one stream, no host round trip:
//
//   scan_kernel      f32 MFMA (v_mfma_f32_32x32x2_f32, bit-exact f32 FMA chains) over [128 queries] x
//                    [DB split]. A lane owns (1 query, half of each 32-row tile) and keeps its top-L as a
//                    sorted register list of KEYS = f32 score with the low `code_bits` mantissa bits
//                    replaced by the row's position inside the split, so one v_med3_f32 per list element
//                    inserts a score branch-free, in the shadow of the serially dependent MFMA chain.
//                    Scores are never written to HBM.
//   rerank_kernel    one wave per query: merges the 2*nsplit sorted lists to the top-L keys, re-scores
//                    those rows in float64 (what the reference ranks by), orders them by (score desc,
//                    row asc) and CERTIFIES: every row that was not re-scored has key <= g, so
//                    s64_K > g + key-truncation + f32-rounding bound  ==> the top-K equals the float64
//                    ranking exactly.
//                    A failed certificate is usually repaired inside the wave (the lists are still in registers: re-score
//                    the few keys that can still reach the top-K); what is left — about one query in a million on
//                    unit-Gaussian data — gets the exact float64 ranking from the re-rank's own workgroup (wg_exact_scan).
//                    No third launch: an empty fallback kernel cost ~5 us of every ~55 us step in round 1.
//   (heavy mode)     databases that defeat the certificates wholesale: search_exact.hip (float64 MFMA stage).
//
// HBM layout: DB f32[n_pad,256] row-major (n_pad = n rounded up to 32, tail rows zero and masked by
// row >= n); queries f32[Q,256]; candidate keys f32 [Q][nsplit][2][L].
#include <float.h>
#include <limits.h>

#include <type_traits>

#include "t2l_internal.h"
#include "search_dev.h"

namespace t2l {

// One 32-row DB tile: 128 MFMAs (D[db row][query]) alternating between TWO accumulators (even / odd k
// steps; their sum is the score) — back-to-back MFMAs on one accumulator forward the result only when they
// are adjacent in the instruction stream, and any instruction between them costs the full write-back
// latency, so the VALU work below needs two independent chains to hide behind. After every 8th MFMA one
// score of the PREVIOUS tile is turned into a key and inserted into the lane's list: ~L+4 VALU instructions
// spread over the gaps (sched_group_barrier pins the interleave).
// fb_count (dev i32[128]), per t2l_search call: [0] queries that ended in an exact float64 VALU scan, [1] queries re-scored
// beyond the first L candidates, [2] queries the first certificate + in-wave re-score left unsettled, [3] f16-probe count
// (auto mode), [4] queries deferred to the float64 MFMA stage (heavy mode), [5] queries a WIDE in-wave repair settled,
// [6] queries the MFMA stage could not certify,
// [7] queries it served, [9] Q and [10] stat mode of the call (rerank_kernel). The first scan launch of a call
// (zero_counts) copies the finished call's [0..15] to [64..79] — rerank_kernel publishes that copy to the host's report
// card — and clears the counters.
// Two banks (round 4): a call counts in `fb_count`, which the call before it left zeroed; `fb_prev` is that earlier call's bank — its
// final counts are parked at fb_count[64..79] (the report card) and it is cleared for the call after this one. The host swaps the two
// per call. No workgroup of a call ever waits for this reset (a launch may count from its first finished query
// block on, whatever workgroup 0 is doing).
__device__ __forceinline__ void reset_counts(int32_t* fb_count, int32_t* fb_prev, int zero_counts, int tid) {
  if (zero_counts && tid < 16) {
    fb_count[64 + tid] = fb_prev[tid];
    fb_prev[tid] = 0;
  }
  if (!zero_counts && tid == 0) {  // a later segment of a multi-segment shard: the deferred lists are per segment
    fb_count[12] += fb_count[4];
    fb_count[4] = fb_count[6] = 0;
  }
}

// (Rounds 1-4 also shipped scan_kernel: the candidate scan on the exact-f32 MFMA, option search_mode = 1 — 212 us per 4,096 queries
// against 30 us for the f16 scan, with ids and float64 scores identical by construction, since the re-rank decides both. Removed in
// round 5 with the other measured losers)

// ------------------------------------------------------------------------------------------------
// Split-bf16 operands (the default scan, scan3_kernel below).
//
// Every f32 value x is stored as two bf16 planes hi = bf16(x), lo = bf16(x - hi); a product is formed as
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with f32 accumulation: |error| <= (2^-16 + 2^-18) * |a||b|
// per dot product on top of the f32 accumulation bound — the same order as the key truncation, and covered
// by the same certificate (the float64 re-rank decides the final order either way). 48 MFMAs of 32 cycles
// per 32-row tile instead of 128 of 64 cycles.
// DB layout for this kernel: bf16 [n_pad][2 planes][256] = 1 KiB per row, built once by split_db_kernel.
// ------------------------------------------------------------------------------------------------
// f32 [rows][256] -> bf16 [rows][hi 256 | lo 256]
__global__ __launch_bounds__(256) void split_db_kernel(const float* __restrict__ db, uint4* __restrict__ out, int rows) {
  const int gid = blockIdx.x * 256 + threadIdx.x;  // one thread = 8 consecutive floats
  const int row = gid >> 5, c = gid & 31;
  if (row >= rows) return;
  const float4* src = reinterpret_cast<const float4*>(db + (size_t)row * kD + 8 * c);
  uint4 hi, lo;
  split8(src[0], src[1], hi, lo);
  out[(size_t)row * 64 + c] = hi;
  out[(size_t)row * 64 + 32 + c] = lo;
}

// ------------------------------------------------------------------------------------------------
// scanw: the wide split-bf16 scan — ONE workgroup (4 waves = one wave per SIMD) per CU, 64 queries per wave.
//
// Why: at 32 queries per wave the LDS feeds the MFMAs at 2/3 of a 16-byte read per MFMA, i.e. 83 % of a CU's LDS
// bandwidth at the full MFMA rate, and scan3 hides its barrier / DMA-issue / first-read phases behind a second
// wave on the same SIMD whose VALU stream slows the first wave's MFMAs (measured; scan3 reaches 40 % MFMA busy).
// Here a wave holds two query fragments (256 operand registers; the unified 512-entry file of a single wave per
// SIMD), every DB fragment read feeds 6 MFMAs on two independent accumulator chains, and everything else is
// software-pipelined INSIDE the one wave: the fragment ring runs across the tile boundary, the tile NBUF-1 ahead
// is DMA'd while the current one multiplies, the previous tile's scores enter the per-lane lists in the MFMA
// shadow. One barrier per tile, placed where nothing waits on it (the tile it releases landed a tile-time ago).
//   grid = ceil(Q/256) * nsplit; block b -> split b % nsplit (tiles sp, sp + nsplit, ...); LDS NBUF x 33 KiB.
// ------------------------------------------------------------------------------------------------
constexpr int kWideQPerWave = 64;
constexpr int kWideQPerBlock = 4 * kWideQPerWave;

template <int LL, int NBUF>
__global__ __launch_bounds__(256, 1) void scanw_kernel(const uint4* __restrict__ dbs, int n_rows, int n_tiles, int code_bits,
                                                       const float* __restrict__ q, int Q, int nsplit,
                                                       float* __restrict__ cand, int32_t* __restrict__ fb_count, int32_t* __restrict__ fb_prev,
                                                       int zero_counts, float pinf) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int sp = blockIdx.x % nsplit, qb = blockIdx.x / nsplit;
  const int nt = sp < n_tiles ? (n_tiles - sp + nsplit - 1) / nsplit : 0;
  const int qrow0 = qb * kWideQPerBlock + wave * kWideQPerWave + col, qrow1 = qrow0 + 32;
  const int mask = ~((1 << code_bits) - 1);
  int vmask = mask;
  asm volatile("" : "+v"(vmask));
  if (blockIdx.x == 0) reset_counts(fb_count, fb_prev, zero_counts, tid);
  if (nt == 0) return;  // (the host never launches an empty split)
  const int uwave = uniform_wave_id();

  // LDS-DMA of this wave's 8 rows of the split's j-th tile (clamped: the pipeline over-issues NBUF-1 tiles at the end,
  // into buffers nobody reads any more; the final s_waitcnt keeps them from outliving the workgroup)
  auto dma_of = [&](int j, int buf) {
    const int tile = sp + min(j, nt - 1) * nsplit;
    WideDma d;
    d.src = reinterpret_cast<const char*>(dbs) + ((size_t)tile * kTileRows + uwave * 8) * 1024;
    d.dst = lds_addr_of(smem + buf * kTileFloats + uwave * 8 * kRowStrideF);
    d.lane16 = lane * 16;
    return d;
  };
#pragma unroll
  for (int b = 0; b < NBUF - 1; ++b) {
    const WideDma d = dma_of(b, b);
#pragma unroll
    for (int i = 0; i < 8; ++i) d.row(i);
  }

  u32x4 qh0[16], ql0[16], qh1[16], ql1[16];  // 256 AGPRs
  {
    const float4* qp0 = reinterpret_cast<const float4*>(q + (size_t)min(qrow0, Q - 1) * kD + half * 128);
    const float4* qp1 = reinterpret_cast<const float4*>(q + (size_t)min(qrow1, Q - 1) * kD + half * 128);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      uint4 h, l;
      split8(qp0[2 * s], qp0[2 * s + 1], h, l);
      qh0[s] = pin_agpr(as_u32x4(h));
      ql0[s] = pin_agpr(as_u32x4(l));
      split8(qp1[2 * s], qp1[2 * s + 1], h, l);
      qh1[s] = pin_agpr(as_u32x4(h));
      ql1[s] = pin_agpr(as_u32x4(l));
    }
  }
  WideLists<LL> w;
#pragma unroll
  for (int i = 0; i < LL; ++i) w.ls0[i] = w.ls1[i] = T2L_NEG_INF;
  w.key0 = w.key1 = T2L_NEG_INF;
  f32x16 accA0, accA1, accB0, accB1;  // tile j / tile j+1, per query group
#pragma unroll
  for (int r = 0; r < 16; ++r) accA0[r] = accA1[r] = accB0[r] = accB1[r] = T2L_NEG_INF;

  const int lane_off = col * (kRowStrideF * 4) + half * 256;
  const char* lbase = reinterpret_cast<const char*>(smem) + lane_off;
  u32x4 ah[4], al[4];
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(8 * (NBUF - 2)) : "memory");  // tile 0 landed for every wave
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ah[i] = *reinterpret_cast<const u32x4*>(lbase + 16 * i);
    al[i] = *reinterpret_cast<const u32x4*>(lbase + 512 + 16 * i);
  }

  int buf = 0;  // buffer of tile j; tile j+1 is in buf+1, the DMA target (tile j+NBUF-1) is buf-1 (mod NBUF)
  auto step = [&](int j, f32x16& cur0, f32x16& cur1, const f32x16& prev0, const f32x16& prev1) {
    const int nbuf = buf + 1 == NBUF ? 0 : buf + 1;
    const int dbuf = buf == 0 ? NBUF - 1 : buf - 1;
    const char* tb = lbase + buf * (kTileFloats * 4);
    const char* tbn = lbase + nbuf * (kTileFloats * 4);
    const WideDma d = dma_of(j + NBUF - 1, dbuf);
    tilew_steps<LL, 0, 12>(tb, tbn, qh0, ql0, qh1, ql1, cur0, cur1, prev0, prev1, vmask, (j - 1) << 4, pinf, w, ah, al, d);
    // tile j+1 (issued NBUF-2 tiles ago) has landed for this wave; after the barrier it has for every wave, and every
    // wave is past its last read of tile j-1, whose buffer the DMA below refills
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(8 * (NBUF - 3)) : "memory");
    tilew_steps<LL, 12, 16>(tb, tbn, qh0, ql0, qh1, ql1, cur0, cur1, prev0, prev1, vmask, (j - 1) << 4, pinf, w, ah, al, d);
    buf = nbuf;
  };
  for (int j = 0; j < nt; j += 2) {
    step(j, accA0, accA1, accB0, accB1);
    if (j + 1 < nt) step(j + 1, accB0, accB1, accA0, accA1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // over-issued DMA rows must not land in LDS after the workgroup is gone

  {  // the last tile's scores are still in registers; only here can rows be >= n_rows
    const int row0 = (sp + (nt - 1) * nsplit) * kTileRows + 4 * half;
    const int code0 = (nt - 1) << 4;
    const bool odd = nt & 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool ok = row0 + (r & 3) + 8 * (r >> 2) < n_rows;
      const float s0 = odd ? accA0[r] : accB0[r], s1 = odd ? accA1[r] : accB1[r];
      ins_key<LL>(w.ls0, ok ? make_key(s0, mask, code0 + r) : T2L_NEG_INF);
      ins_key<LL>(w.ls1, ok ? make_key(s1, mask, code0 + r) : T2L_NEG_INF);
    }
  }
  const int part = 2 * sp + half, parts = 2 * nsplit;
  if (qrow0 < Q) {
    float4* out = reinterpret_cast<float4*>(cand + ((size_t)qrow0 * parts + part) * LL);
#pragma unroll
    for (int i = 0; i < LL / 4; ++i) out[i] = make_float4(w.ls0[4 * i], w.ls0[4 * i + 1], w.ls0[4 * i + 2], w.ls0[4 * i + 3]);
  }
  if (qrow1 < Q) {
    float4* out = reinterpret_cast<float4*>(cand + ((size_t)qrow1 * parts + part) * LL);
#pragma unroll
    for (int i = 0; i < LL / 4; ++i) out[i] = make_float4(w.ls1[4 * i], w.ls1[4 * i + 1], w.ls1[4 * i + 2], w.ls1[4 * i + 3]);
  }
}

// ------------------------------------------------------------------------------------------------
// scanh: the f16 scan (default) — the wide kernel's structure with ONE f16 MFMA per product instead of three bf16 ones.
//
// The scan only has to deliver candidates whose keys bound the scores of everything it drops; the float64 re-rank
// decides the order and the certificate proves it. With exact power-of-two scaling (search_dev.h) the f16 operand
// rounding costs |error| <= ~1e-3 |q||c| — 50x the split-bf16 bound, still far below the gap between the K-th and the
// L-th best score of a real query, and a failed certificate only costs that query the float64 fallback.
// A third of the MFMA work, half the DB bytes through L2/LDS (512 B per row), half the operand registers
// (128 AGPRs), so the kernel is bound by the per-score selection VALU (9 ops) rather than by the matrix pipe.
//   DB plane: f16 [n_pad][256] (scaled by 2^shift_db), built by half_db_kernel. LDS: 4 x 16 KiB tiles, unpadded rows,
//   16-byte chunk c of row r stored at chunk c ^ r (the LDS-DMA lanes fetch the permuted chunks; ds_read_b128 of one
//   k-step then hits every bank group exactly 4 times = full LDS rate).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void half_db_kernel(const float* __restrict__ db, const float* __restrict__ norms,
                                                      uint4* __restrict__ out, int rows_pad, int n_rows) {
  // one thread = 8 consecutive floats of one row -> one 16-byte f16 chunk; consecutive threads take consecutive SLOTS of one
  // chunk column, i.e. consecutive 16-byte units of the tile-chunk-major plane (search_dev.h): coalesced writes. The slots of a
  // tile hold STRIDED rows (plane_row, search_dev.h): the reads are 32-byte pieces of rows n_rows / 32 apart — once per t2l_db_set.
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int r = gid & 31, c = (gid >> 5) & 31, tile = gid >> 10;
  const int pos = tile * kTileRows + r;
  if (pos >= rows_pad) return;
  const int row = plane_row(pos, n_rows);
  if (row >= n_rows) {  // padding slots (only the partial last tile has them, and whole tiles behind it)
    out[gid] = make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  int shift;
  half_shift_of(norms[1], shift);
  const float4* src = reinterpret_cast<const float4*>(db + (size_t)row * kD + 8 * c);
  const float4 a = src[0], b = src[1];
  out[gid] = make_uint4(pack_f16x2(a.x, a.y, shift), pack_f16x2(a.z, a.w, shift), pack_f16x2(b.x, b.y, shift),
                        pack_f16x2(b.z, b.w, shift));
}

template <int LL>
__global__ __launch_bounds__(256, 1) void scanh_kernel(const uint4* __restrict__ dbh, int n_rows, int n_tiles, int code_bits,
                                                       const float* __restrict__ q, int Q, int nsplit,
                                                       float* __restrict__ cand, int32_t* __restrict__ fb_count, int32_t* __restrict__ fb_prev,
                                                       int zero_counts, float pinf) {
  constexpr int NBUF = 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int half = lane >> 5, col = lane & 31;
  const int sp = blockIdx.x % nsplit, qb = blockIdx.x / nsplit;
  const int nt = sp < n_tiles ? (n_tiles - sp + nsplit - 1) / nsplit : 0;
  const int uwave = uniform_wave_id();
  const int qrow0 = qb * kWideQPerBlock + uwave * kWideQPerWave + col, qrow1 = qrow0 + 32;
  const int mask = ~((1 << code_bits) - 1);
  int vmask = mask;
  asm volatile("" : "+v"(vmask));
  if (blockIdx.x == 0) reset_counts(fb_count, fb_prev, zero_counts, tid);
  if (nt == 0) return;  // (the host never launches an empty split)

  // ---- LDS-DMA plan: wave w moves pieces 4w .. 4w+3 (1 KiB each, contiguous) of a tile: global image == LDS image
  const unsigned lds_base = lds_addr_of(smem);
  auto dma_of = [&](int j, int buf) {
    const int tile = sp + min(j, nt - 1) * nsplit;  // over-issue at the end is clamped (see scanw_kernel)
    HalfDma d;
    d.src = reinterpret_cast<const char*>(dbh) + (size_t)tile * kHalfTileBytes + uwave * 4096;
    d.dst = lds_base + buf * kHalfTileBytes + uwave * 4096;
    d.lane16 = lane * 16;
    return d;
  };
#pragma unroll
  for (int b = 0; b < NBUF - 1; ++b) {
    const HalfDma d = dma_of(b, b);
#pragma unroll
    for (int i = 0; i < 4; ++i) d.piece(i);
  }

  // ---- queries: scale by the power of two that puts the largest |element| into [2^14, 2^15), round to f16, pin in AGPRs
  u32x4 q0[16], q1[16];  // 128 AGPRs
  auto load_group = [&](int qrow, u32x4 (&dst)[16]) {
    const float4* qp = reinterpret_cast<const float4*>(q + (size_t)min(qrow, Q - 1) * kD + half * 128);
    float4 v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = qp[i];
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      m = fmaxf(fmaxf(m, fabsf(v[i].x)), fabsf(v[i].y));
      m = fmaxf(fmaxf(m, fabsf(v[i].z)), fabsf(v[i].w));
    }
    {  // the other half of the row lives in lane ^ 32
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
      m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    int shift;
    half_shift_of(m, shift);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float4 a = v[2 * s], b = v[2 * s + 1];
      dst[s] = pin_agpr(u32x4{pack_f16x2(a.x, a.y, shift), pack_f16x2(a.z, a.w, shift), pack_f16x2(b.x, b.y, shift),
                              pack_f16x2(b.z, b.w, shift)});
    }
  };
  load_group(qrow0, q0);
  load_group(qrow1, q1);

  WideLists<LL> w;
#pragma unroll
  for (int i = 0; i < LL; ++i) w.ls0[i] = w.ls1[i] = T2L_NEG_INF;
  w.key0 = w.key1 = T2L_NEG_INF;
  f32x16 accA0, accA1, accB0, accB1;  // tile j / tile j+1, per query group
#pragma unroll
  for (int r = 0; r < 16; ++r) accA0[r] = accA1[r] = accB0[r] = accB1[r] = T2L_NEG_INF;

  // lane (col, half) reads chunk half*16 + S of row col at k-step S: lb + S*512 (tile-chunk-major, search_dev.h)
  const char* lb = reinterpret_cast<const char*>(smem) + half * 8192 + col * 16;
  u32x4 ring[4];
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (NBUF - 2)) : "memory");  // tile 0 landed for every wave
#pragma unroll
  for (int i = 0; i < 4; ++i) ring[i] = *reinterpret_cast<const u32x4*>(lb + i * 512);

  auto step = [&](auto buf_tag, int j, f32x16& cur0, f32x16& cur1, const f32x16& prev0, const f32x16& prev1) {
    constexpr int BUF = decltype(buf_tag)::value;
    const HalfDma d = dma_of(j + NBUF - 1, (BUF + NBUF - 1) % NBUF);
    tileh_steps<LL, 0, 12, BUF, NBUF>(lb, q0, q1, cur0, cur1, prev0, prev1, vmask, (j - 1) << 4, pinf, w, ring, d);
    // tile j+1 (issued two tiles ago) has landed for this wave; after the barrier it has for every wave, and every wave
    // is past its last read of tile j-1, whose buffer the DMA below refills
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (NBUF - 3)) : "memory");
    tileh_steps<LL, 12, 16, BUF, NBUF>(lb, q0, q1, cur0, cur1, prev0, prev1, vmask, (j - 1) << 4, pinf, w, ring, d);
  };
  for (int j = 0; j < nt; j += 4) {
    step(std::integral_constant<int, 0>{}, j, accA0, accA1, accB0, accB1);
    if (j + 1 < nt) step(std::integral_constant<int, 1>{}, j + 1, accB0, accB1, accA0, accA1);
    if (j + 2 < nt) step(std::integral_constant<int, 2>{}, j + 2, accA0, accA1, accB0, accB1);
    if (j + 3 < nt) step(std::integral_constant<int, 3>{}, j + 3, accB0, accB1, accA0, accA1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // over-issued DMA pieces must not land in LDS after the workgroup is gone

  {  // the last tile's scores are still in registers; only here can rows be >= n_rows
    const int row0 = (sp + (nt - 1) * nsplit) * kTileRows + 4 * half;
    const int code0 = (nt - 1) << 4;
    const bool odd = nt & 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool ok = row0 + (r & 3) + 8 * (r >> 2) < n_rows;
      const float s0 = odd ? accA0[r] : accB0[r], s1 = odd ? accA1[r] : accB1[r];
      ins_key<LL>(w.ls0, ok ? make_key(s0, mask, code0 + r) : T2L_NEG_INF);
      ins_key<LL>(w.ls1, ok ? make_key(s1, mask, code0 + r) : T2L_NEG_INF);
    }
  }
  const int part = 2 * sp + half, parts = 2 * nsplit;
  if (qrow0 < Q) {
    float4* out = reinterpret_cast<float4*>(cand + ((size_t)qrow0 * parts + part) * LL);
#pragma unroll
    for (int i = 0; i < LL / 4; ++i) out[i] = make_float4(w.ls0[4 * i], w.ls0[4 * i + 1], w.ls0[4 * i + 2], w.ls0[4 * i + 3]);
  }
  if (qrow1 < Q) {
    float4* out = reinterpret_cast<float4*>(cand + ((size_t)qrow1 * parts + part) * LL);
#pragma unroll
    for (int i = 0; i < LL / 4; ++i) out[i] = make_float4(w.ls1[4 * i], w.ls1[4 * i + 1], w.ls1[4 * i + 2], w.ls1[4 * i + 3]);
  }
}

// ------------------------------------------------------------------------------------------------
// scanp: the PAIRED f16 scan (default) — scanh's per-wave pipeline with TWO waves per SIMD (search_dev.h).
//
//   workgroup = 512 threads = 8 waves, 256 queries (waves w and w+4 hold the same 64), DB split sp of nsplit; wave w (< 4)
//   scans the tiles of virtual split sp, wave w+4 those of sp + nsplit (of 2*nsplit: tile t belongs to virtual split
//   t % (2*nsplit)), one tile each per STEP. A step's two tiles share an LDS slot (2 x 16 KiB); the ring holds NS slots and
//   all 8 waves move every tile (4 LDS-DMA pieces per wave per step). One s_waitcnt + s_barrier per step.
//   Per-lane lists of LL = 6 keys: 4 lists per (query, split) -> 64 lists per query at nsplit = 16, so the candidate set
//   (384 keys) is larger than scanh's (32 x 8) at 7 instead of 9 selection VALU per score.
//   Prologue: the 256 queries are converted COOPERATIVELY, once per workgroup: thread (query, quarter row) loads 64
//   floats (64 staging registers instead of 128 per lane), the row maximum is a DPP quad reduction, the scaled f16 chunks
//   go to LDS and every wave reads its 2 x 16 fragments back into AGPRs — two rounds of 128 queries (round = query group),
//   64 KiB of LDS that the ring's later slots reuse.
//   grid = ceil(Q/256) * nsplit; LDS 128 KiB.
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, true));
}

// The arguments of the re-rank's per-query routine (rerank_query), as one struct.
struct RerankArgs {
  const float* db;
  const float* q;
  int Q, K, parts, code_bits;
  const float* cand;
  int row_offset;
  float eps_rel;
  const float* db_norm_max;
  int half_mode;
  int32_t* out_idx;
  double* out_score;
  int32_t* flags;
  int32_t* fb_count;
  float eps_rel_probe, pinf;
  int n_rows, defer, stat_mode;
  int32_t* host_stat;
  int seq, wide_cap;
  int rec6;  // merged records of the tile-local selection: 6 keys, B1 (a key: the largest third-best key of the record's lanes), B2
};

// MERGE: the workgroup's four lists per query (two lane halves x two waves, 24 keys) leave as ONE record of 8 floats — the best 7 keys,
// re-keyed so that the source list rides in two more code bits, + a bound on every key that did not make it (kMergedLL; the
// re-rank's MG form reads it): a third of the bytes written back at the end of the launch and read by the re-rank.
constexpr int kMergedLL = 8;
// SEL = 1: the tile-local selection (search_dev.h, TileSelLists): 86 instead of 112 selection VALU per lane-tile and query group;
// the third-best key of every group of 8 accumulator registers goes into the lane's (dA, dB), which leave in the record as B1 / B2.
#ifndef T2L_SEL_RD
#define T2L_SEL_RD 2  // fragment-ring depth of the SEL = 1 loop (the tile-local registers come out of the ring's)
#endif
template <int LL, int NS, bool MERGE = false, int SEL = 0>
__global__ __launch_bounds__(512, 1) void scanp_kernel(const uint4* __restrict__ dbt, int n_rows, int n_tiles, int code_bits,
                                                       const float* __restrict__ q, int Q, int nsplit,
                                                       float* __restrict__ cand, int32_t* __restrict__ fb_count, int32_t* __restrict__ fb_prev,
                                                       int zero_counts, float pinf, unsigned long long* __restrict__ span,
                                                       unsigned span_seq, int xcd_qgroups) {
  static_assert(NS == 4, "the step loop below is unrolled for a ring of 4 slots");
  static_assert(!MERGE || LL == 6, "the in-workgroup list merge is written for lists of 6");
  static_assert(!SEL || MERGE, "the tile-local selection's dropped-bound leaves through the merged record's bound");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kSlotBytes = 2 * kHalfTileBytes;
  constexpr unsigned kExchange = 2 * kSlotBytes;  // slots 2.. double as the query exchange area (64 KiB) in the prologue
  const int tid = threadIdx.x, lane = tid & 63;
  const int half = lane >> 5, col = lane & 31;
  // workgroup -> (query block, split): blockIdx % 8 is the XCD, so the workgroups of one XCD share SPLITS (its L2 pulls all
  // queries and 1/8 of the plane). Measured alternative — sharing query blocks instead (1/8 of the queries in the prologue, the
  // whole plane over the main loop): 47.6 vs 46.7 us per step, rejected.
  // xcd_qgroups = GQ > 1: XCD x owns the RECTANGLE {query blocks = x % GQ (mod GQ)} x {splits = x / GQ (mod 8 / GQ)} instead: its
  // L2 pulls 1/GQ of the queries in the prologue (the burst every CU waits for) and GQ/8 of the plane over the main loop.
  int sp = blockIdx.x % nsplit, qb = blockIdx.x / nsplit;
  if (xcd_qgroups > 1) {
    const int GQ = xcd_qgroups, GS = 8 / GQ, nqb = gridDim.x / nsplit;
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, per = nqb / GQ;
    qb = (j % per) * GQ + x % GQ;
    sp = (j / per) * GS + x / GQ;
  }
  const int uwave = uniform_wave_id();
  const int quad = uwave >> 2, wq = uwave & 3;
  const int vn = 2 * nsplit, vs = sp + quad * nsplit;                     // this wave's virtual split
  const int nt = vs < n_tiles ? (n_tiles - vs + vn - 1) / vn : 0;         // its tiles: vs, vs + vn, ...
  const int steps = sp < n_tiles ? (n_tiles - sp + vn - 1) / vn : 0;      // = nt of the first quad >= nt of the second
  // MERGE: keys are born in the merged record's form — code << 2 | source list (2 quad + half; the half bit is OR-ed in at the end: one
  // scalar code per accumulator register, as before) — so that the lists stay sorted through the merge
  constexpr int kCS = MERGE ? 2 : 0;
  const int mask = ~((1 << (code_bits + kCS)) - 1);
  const int code_q = MERGE ? (quad << 1) : 0;
  int vmask = mask;
  asm volatile("" : "+v"(vmask));
  if (blockIdx.x == 0) reset_counts(fb_count, fb_prev, zero_counts, tid);
  if (steps == 0) return;  // (the host never launches an empty split)
  // always-on stamps (t2l_kernel_stats "search_scan_span" / "search_scan_busy"): every workgroup stores its own start and end
  // (launch sequence << 40 | 100 MHz ticks) in its own slot of the launch's ring entry — plain stores, no packet on the stream
  // (an event pair around the kernel costs ~6 us) and no atomics (256 workgroups meeting in ONE atomic at the kernel's end
  // cost ~3 us of kernel completion, measured). The host reduces: span = max end - min start, busy = mean (end - start).
  const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();  // (scalar: lives in SGPRs)
#ifdef T2L_STAMPS
#define T2L_STAMP(k)                                                                                                   \
  if (tid == 0) {                                                                                                      \
    const long long t_ = __builtin_amdgcn_s_memrealtime();                                                             \
    if (blockIdx.x == 37) reinterpret_cast<long long*>(fb_count + 16)[k] = t_;                                         \
    if (k == 0) atomicMin(reinterpret_cast<unsigned long long*>(fb_count + 16) + 4, (unsigned long long)t_);           \
    if (k == 0) atomicMax(reinterpret_cast<unsigned long long*>(fb_count + 16) + 5, (unsigned long long)t_);           \
    if (k == 3) atomicMin(reinterpret_cast<unsigned long long*>(fb_count + 16) + 6, (unsigned long long)t_);           \
    if (k == 3) atomicMax(reinterpret_cast<unsigned long long*>(fb_count + 16) + 7, (unsigned long long)t_);           \
  }
#define T2L_STAMP2(k) if (tid == 0 && blockIdx.x == 37) reinterpret_cast<long long*>(fb_count + 32)[k] = __builtin_amdgcn_s_memrealtime()
#else
#define T2L_STAMP(k)
#define T2L_STAMP2(k)
#endif
  T2L_STAMP(0);

  const unsigned lds_base = lds_addr_of(smem);
  const unsigned lane16 = lane * 16;
  // the wave's 4 LDS-DMA pieces of step i (2 tiles x pieces 2w, 2w+1) into slot `slot`; tiles past the end are clamped to
  // the last tile (the pipeline over-issues, and the second quad may run one step more than it has tiles: nobody reads those)
  auto dma_of = [&](int i, int slot) {
    PairDma d;
    const int t0 = min(sp + (2 * i) * nsplit, n_tiles - 1), t1 = min(sp + (2 * i + 1) * nsplit, n_tiles - 1);
    d.src0 = reinterpret_cast<const char*>(dbt) + (size_t)t0 * kHalfTileBytes + uwave * 2048;
    d.src1 = reinterpret_cast<const char*>(dbt) + (size_t)t1 * kHalfTileBytes + uwave * 2048;
    d.dst = lds_base + slot * kSlotBytes + uwave * 2048;
    d.lane16 = lane16;
    return d;
  };
#pragma unroll
  for (int b = 0; b < 2; ++b) {  // steps 0 and 1 travel while the queries are converted
    const PairDma d = dma_of(b, b);
#pragma unroll
    for (int e = 0; e < 4; ++e) d.piece(e);
  }

  // ---- queries: round g converts query group g of every wave slot: local query ql = (wave slot, col) <-> global query
  // qb*256 + slot*64 + g*32 + col; thread (ql, quarter) = (tid >> 2, tid & 3). LDS exchange rows are 512 B (one f16 query),
  // 16-byte chunk c of row ql at chunk c ^ (ql & 31): conflict-free for the quarter-row writers and the fragment readers.
  u32x4 q0[16], q1[16];  // 128 AGPRs
  {
    const int ql = tid >> 2, qt = tid & 3;
    char* ex_w = reinterpret_cast<char*>(smem) + kExchange + ql * 512;
    const char* ex_r = reinterpret_cast<const char*>(smem) + kExchange + (wq * 32 + col) * 512;
    // BOTH rounds' loads are issued before anything waits on them (2 x 64 staging registers: what does not fit the 128
    // VGPRs parks in the still-empty AGPRs): the prologue is latency-bound, one round trip instead of two
    float4 v[2][16];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int qrow = min(qb * kWideQPerBlock + (ql >> 5) * kWideQPerWave + g * 32 + (ql & 31), Q - 1);
      // thread (ql, qt) owns the 16-byte chunks 4 i + qt of the row: the 4 lanes of a row read one whole 64-byte line per load
      // (a quarter-row per thread made every line arrive in four separate requests: the workgroup's 256 KB then took 4.3 us
      // to trickle through the CU's L1 — the slowest wave sets the barrier)
      const float4* qp = reinterpret_cast<const float4*>(q + (size_t)qrow * kD) + qt;
#pragma unroll
      for (int i = 0; i < 16; ++i) v[g][i] = qp[4 * i];
#ifdef T2L_EXP_HALFLOAD  // dev experiment: is the prologue bound by the bytes or by the round trip?
      if (g == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[1][i] = v[0][i];
      }
#endif
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float m = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        m = fmaxf(fmaxf(m, fabsf(v[g][i].x)), fabsf(v[g][i].y));
        m = fmaxf(fmaxf(m, fabsf(v[g][i].z)), fabsf(v[g][i].w));
      }
      if (g == 0) { T2L_STAMP2(0); } else { T2L_STAMP2(3); }
      m = fmaxf(m, dpp_f<kDppXor1>(m));  // the row's 4 quarters sit in 4 adjacent lanes
      m = fmaxf(m, dpp_f<kDppXor2>(m));
      int shift;
      half_shift_of(m, shift);
      if (g == 1) __syncthreads();  // every wave has read round 0's fragments: the exchange area may be overwritten
#pragma unroll
      for (int j = 0; j < 16; ++j) {  // f32 chunk 4 j + qt = half (qt & 1) of f16 chunk 2 j + (qt >> 1)
        const float4 a = v[g][j];
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2*>(ex_w + (((2 * j + (qt >> 1)) ^ (ql & 31)) << 4) + ((qt & 1) << 3)) =
            u32x2{pack_f16x2(a.x, a.y, shift), pack_f16x2(a.z, a.w, shift)};
      }
      __syncthreads();
      if (g == 0) { T2L_STAMP2(1); } else { T2L_STAMP2(4); }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const u32x4 f = *reinterpret_cast<const u32x4*>(ex_r + ((((half << 4) + s) ^ col) << 4));
        if (g == 0) q0[s] = pin_agpr(f); else q1[s] = pin_agpr(f);
      }
      if (g == 0) { T2L_STAMP2(2); } else { T2L_STAMP2(5); }
    }
    __syncthreads();  // the exchange area is free: it is slots 2.. of the tile ring from here on
  }
  T2L_STAMP(1);
  {
    const PairDma d = dma_of(2, 2);
#pragma unroll
    for (int e = 0; e < 4; ++e) d.piece(e);
  }

  std::conditional_t<SEL != 0, TileSelLists<LL>, WideLists<LL>> w;
#pragma unroll
  for (int i = 0; i < LL; ++i) w.ls0[i] = w.ls1[i] = T2L_NEG_INF;
  if constexpr (SEL) w.dA0 = w.dB0 = w.dA1 = w.dB1 = w.t0 = w.t1 = w.t2 = w.k = T2L_NEG_INF;
  else w.key0 = w.key1 = T2L_NEG_INF;
  f32x16 accA0, accA1, accB0, accB1;  // step i / step i+1, per query group
#pragma unroll
  for (int r = 0; r < 16; ++r) accA0[r] = accA1[r] = accB0[r] = accB1[r] = T2L_NEG_INF;

  lds_cptr lb0 = (lds_cptr)smem + quad * kHalfTileBytes + half * 8192 + col * 16;
  lds_cptr lb1 = lb0 + 65536;
  asm volatile("" : "+v"(lb1));  // opaque: otherwise every slot-2/3 fragment address (> the 16-bit ds offset) gets its own register
  constexpr int RD = SEL ? T2L_SEL_RD : 4;
  u32x4 ring[RD];
  // steps 0 and 1 landed for every wave before the query loads returned (vmcnt retires in order) and the barriers above
  // published that; only step 2's pieces may be in flight
#pragma unroll
  for (int i = 0; i < RD; ++i) ring[i] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(lb0 + i * 512);

  auto step = [&](auto slot_tag, auto first_tag, int i, f32x16& cur0, f32x16& cur1, const f32x16& prev0, const f32x16& prev1) {
    constexpr int SLOT = decltype(slot_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value && SEL;  // the launch's first step has no scores to select from yet
    const PairDma d = dma_of(i + NS - 1, (SLOT + NS - 1) % NS);
    if constexpr (SEL) tilep3_steps<LL, 0, 12, SLOT, NS, kCS, RD, FIRST>(lb0, lb1, q0, q1, cur0, cur1, prev0, prev1, vmask, ((i - 1) << (4 + kCS)) | code_q, pinf, w, ring, d);
    else tilep_steps<LL, 0, 12, SLOT, NS, kCS>(lb0, lb1, q0, q1, cur0, cur1, prev0, prev1, vmask, ((i - 1) << (4 + kCS)) | code_q, pinf, w, ring, d);
    // step i+1 (issued two steps ago) has landed for this wave; after the barrier it has for every wave, and every wave
    // is past its last read of step i-1, whose slot the DMA below refills
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (NS - 3)) : "memory");
    if constexpr (SEL) tilep3_steps<LL, 12, 16, SLOT, NS, kCS, RD, FIRST>(lb0, lb1, q0, q1, cur0, cur1, prev0, prev1, vmask, ((i - 1) << (4 + kCS)) | code_q, pinf, w, ring, d);
    else tilep_steps<LL, 12, 16, SLOT, NS, kCS>(lb0, lb1, q0, q1, cur0, cur1, prev0, prev1, vmask, ((i - 1) << (4 + kCS)) | code_q, pinf, w, ring, d);
  };
  // SEL: the first step is peeled (it has nothing to select from: NOSEL). NOT for the per-score insertion: there the first step's keys
  // are NaN (-inf accumulators OR-ed with a code) and stay out of the lists by the HARDWARE's med3 rule (a NaN operand -> min3 of the
  // others); peeled, the compiler knows the lists are -inf there and folds med3(-inf, key, +inf) to min(key, +inf) = +inf for a NaN key
  // — every list head +inf, every certificate failing (measured: all 4,096 queries in the exact scan). A loop it cannot see through
  // leaves the rule to the hardware.
  constexpr std::false_type later{};
  if constexpr (SEL) step(std::integral_constant<int, 0>{}, std::true_type{}, 0, accA0, accA1, accB0, accB1);
  for (int i = 0; i < steps; i += 4) {
    if (i || !SEL) step(std::integral_constant<int, 0>{}, later, i, accA0, accA1, accB0, accB1);
    if (i + 1 < steps) step(std::integral_constant<int, 1>{}, later, i + 1, accB0, accB1, accA0, accA1);
    if (i + 2 < steps) step(std::integral_constant<int, 2>{}, later, i + 2, accA0, accA1, accB0, accB1);
    if (i + 3 < steps) step(std::integral_constant<int, 3>{}, later, i + 3, accB0, accB1, accA0, accA1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // over-issued DMA pieces must not land in LDS after the workgroup is gone
  T2L_STAMP(2);

  if (nt == steps) {  // the wave's last tile's scores are still in registers; only here can rows be >= n_rows. (A wave with
                      // nt == steps - 1 ran its last step on a clamped tile: that step inserted the real last tile's scores.)
    const int row0 = (vs + (nt - 1) * vn) * kTileRows + 4 * half;
    const int code0 = ((nt - 1) << (4 + kCS)) | code_q;
    const bool odd = nt & 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool ok = row0 + (r & 3) + 8 * (r >> 2) < n_rows;
      const float s0 = odd ? accA0[r] : accB0[r], s1 = odd ? accA1[r] : accB1[r];
      ins_key<LL>(w.ls0, ok ? make_key(s0, mask, code0 + (r << kCS)) : T2L_NEG_INF);
      ins_key<LL>(w.ls1, ok ? make_key(s1, mask, code0 + (r << kCS)) : T2L_NEG_INF);
    }
  }
  const int part = 2 * vs + half, parts = 2 * vn;
  const int qrow0 = qb * kWideQPerBlock + wq * kWideQPerWave + col, qrow1 = qrow0 + 32;
  auto put_list = [&](int qrow, const float* ls) {  // (measured: non-temporal stores here cost +2 us per step)
    if (qrow >= Q) return;
#ifdef T2L_EXP_THIRDLISTS  // dev experiment (make exp_thirdlists, tools/listwrite_probe.py): what would a 3x smaller list write-back buy?
    if (part % 3) return;
#endif
    float* out = cand + ((size_t)qrow * parts + part) * LL;
    if constexpr (LL % 2 == 0) {  // 24-byte lists: three 8-byte stores
#pragma unroll
      for (int i = 0; i < LL / 2; ++i) {
        reinterpret_cast<float2*>(out)[i] = make_float2(ls[2 * i], ls[2 * i + 1]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < LL; ++i) out[i] = ls[i];
    }
  };
  if constexpr (MERGE) {
    // ---- the four lists of a query (this lane, its other half, the same two lanes of wave uwave ^ 4) -> top 7 of 24 + bound.
    // The keys carry code << 2 | quad << 1 already; the lane half goes into bit 0 here (the same bit into every key of a list: it stays
    // sorted). Two sorted lists meet in a bitonic network: max(A_i, B_{7-i}) are the 8 largest, three compare-exchange stages sort them.
    auto with_half = [&](float k) { return k == T2L_NEG_INF ? k : __int_as_float(__float_as_int(k) | half); };
    const float ninf = -pinf;
    auto cx = [&](float& hi, float& lo) {  // compare-exchange, one op per side (med3 against +-inf: no canonicalising extra)
      const float a = hi, b = lo;
      hi = __builtin_amdgcn_fmed3f(a, b, pinf);
      lo = __builtin_amdgcn_fmed3f(a, b, ninf);
    };
    auto sort_bitonic8 = [&](float (&c)[kMergedLL]) {
#pragma unroll
      for (int d = 4; d >= 1; d >>= 1)
#pragma unroll
        for (int i = 0; i < kMergedLL; ++i)
          if ((i & d) == 0) cx(c[i], c[i + d]);
    };
    float A0[kMergedLL], A1[kMergedLL], O0[LL], O1[LL];
#pragma unroll
    for (int i = 0; i < LL; ++i) {
      O0[i] = with_half(w.ls0[i]);
      O1[i] = with_half(w.ls1[i]);
    }
    // a lane's FLOOR (its 6th key: every row the lane saw and dropped lies at or below it) bounds what the record cannot list
    float f0 = O0[LL - 1], f1 = O1[LL - 1];
    f0 = fmaxf(f0, xor32_f32(f0, half));
    f1 = fmaxf(f1, xor32_f32(f1, half));
    // SEL: (dA >= dB) = the two largest third-best keys of the lane's tile-local groups — real keys whose code names their group. The four
    // lanes of a query meet in (DA >= DB), the two largest of their eight: two sorted pairs -> top 2 = (max heads, max(min heads, max seconds))
    auto top2 = [&](float& a, float& b, float oa, float ob) {
      const float lo = __builtin_amdgcn_fmed3f(a, oa, ninf);
      a = __builtin_amdgcn_fmed3f(a, oa, pinf);
      b = __builtin_amdgcn_fmed3f(__builtin_amdgcn_fmed3f(b, ob, pinf), lo, pinf);
    };
    float DA0 = T2L_NEG_INF, DB0 = T2L_NEG_INF, DA1 = T2L_NEG_INF, DB1 = T2L_NEG_INF;
    if constexpr (SEL) {
      DA0 = with_half(w.dA0); DB0 = with_half(w.dB0); DA1 = with_half(w.dA1); DB1 = with_half(w.dB1);
      top2(DA0, DB0, xor32_f32(DA0, half), xor32_f32(DB0, half));
      top2(DA1, DB1, xor32_f32(DA1, half), xor32_f32(DB1, half));
    }
    {
      float B0[LL], B1[LL];
#pragma unroll
      for (int i = 0; i < LL; ++i) {
        B0[i] = xor32_f32(O0[i], half);
        B1[i] = xor32_f32(O1[i], half);
      }
      static_assert(LL == 6 && kMergedLL == 8, "the half merge below is written out for 6 + 6 -> 8");
      A0[0] = O0[0]; A0[1] = O0[1]; A0[6] = B0[1]; A0[7] = B0[0];
      A1[0] = O1[0]; A1[1] = O1[1]; A1[6] = B1[1]; A1[7] = B1[0];
#pragma unroll
      for (int i = 2; i < 6; ++i) {
        A0[i] = __builtin_amdgcn_fmed3f(O0[i], B0[7 - i], pinf);
        A1[i] = __builtin_amdgcn_fmed3f(O1[i], B1[7 - i], pinf);
      }
      sort_bitonic8(A0);
      sort_bitonic8(A1);
    }
    // both halves now hold the same two merged lists; half h carries query col + 32 h from here on
    float M[kMergedLL];
#pragma unroll
    for (int i = 0; i < kMergedLL; ++i) M[i] = half ? A1[i] : A0[i];
    float fl = half ? f1 : f0;
    float DA = half ? DA1 : DA0, DB = half ? DB1 : DB0;
    // 11-float records (conflict-free) in their own 11 KB BEHIND the tile ring: no barrier between the last tile and the exchange
    float* xch = smem + NS * kSlotBytes / sizeof(float) + ((size_t)wq * 64 + lane) * (kMergedLL + 3);
    if (quad == 1) {
#pragma unroll
      for (int i = 0; i < kMergedLL; ++i) xch[i] = M[i];
      xch[kMergedLL] = fl;
      if constexpr (SEL) {
        xch[kMergedLL + 1] = DA;
        xch[kMergedLL + 2] = DB;
      }
    }
    __syncthreads();
    if (quad == 0) {
#pragma unroll
      for (int i = 0; i < kMergedLL; ++i) M[i] = __builtin_amdgcn_fmed3f(M[i], xch[kMergedLL - 1 - i], pinf);
      sort_bitonic8(M);
      fl = fmaxf(fl, xch[kMergedLL]);
      // everything evicted on the way lies at or below the 8th merged key; everything a lane dropped at or below its floor
      M[kMergedLL - 1] = fmaxf(M[kMergedLL - 1], fl);
      if constexpr (SEL) {
        // record of the tile-local selection: 6 keys, B1 = the largest third-best key of the four lanes (a KEY: the re-rank decodes
        // its tile-local group and re-scores those 8 rows when B1 alone stands between a query and its certificate), B2 = the bound on
        // everything else that is not listed: the 7th / 8th merged keys, the list floors, the other seven third-best keys
        top2(DA, DB, xch[kMergedLL + 1], xch[kMergedLL + 2]);
        M[kMergedLL - 1] = fmaxf(fmaxf(M[kMergedLL - 1], M[kMergedLL - 2]), DB);
        M[kMergedLL - 2] = DA;
      }
      const int qrow = qb * kWideQPerBlock + wq * kWideQPerWave + lane;
      if (qrow < Q) {
        float4* out = reinterpret_cast<float4*>(cand + ((size_t)qrow * nsplit + sp) * kMergedLL);
        out[0] = make_float4(M[0], M[1], M[2], M[3]);
        out[1] = make_float4(M[4], M[5], M[6], M[7]);
      }
    }
  } else {
    put_list(qrow0, w.ls0);
    put_list(qrow1, w.ls1);
  }
  if (span && threadIdx.x == 0) {
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime(), tag = (unsigned long long)span_seq << 40, tm = (1ull << 40) - 1;
    *reinterpret_cast<ulonglong2*>(span + 2 * blockIdx.x) = make_ulonglong2(tag | (wg_t0 & tm), tag | (t1 & tm));
  }
  T2L_STAMP(3);
}

// ------------------------------------------------------------------------------------------------
// The exact float64 ranking of ONE query by the re-rank's own workgroup (4 waves) — what a query gets when neither the
// certificate nor the in-wave re-score settles it (about 1 query in 800,000 on unit-Gaussian data; when it becomes common,
// search_impl's report card moves the database to the float64 MFMA stage, search_exact.hip). Lean on purpose: it lives
// inside rerank_kernel (no separate launch: an empty launch costs ~5 us of a ~50 us step), so it must fit the re-rank's
// register budget. Wave w takes row groups w, w+4, ... of 4 rows (one per 16-lane row of the wave, the re-rank's re-score
// layout and arithmetic); the wave's top-32 is a list spread over lanes (lane i = i-th best, ordered by score desc, row asc);
// the 4 lists meet in LDS and the first wave ranks the 128 entries.
// ------------------------------------------------------------------------------------------------
template <int NW>
struct WgExactSharedN {
  double score[NW][32];
  int row[NW][32];
};
typedef WgExactSharedN<4> WgExactShared;

// Rows one wave re-scores in a WIDE repair (rerank_kernel) before the query is handed to an exact scan of the whole shard.
constexpr int kWideCap = 1024;

// A wave's running top-32 as a list spread over lanes 0..31 (lane i = i-th best by (score desc, row asc); empty = -inf / INT_MAX):
// insert (cd, cr) unless that row is already in the list. Wave-uniform control flow.
__device__ __forceinline__ void top32_insert(double& ts, int& tr, double cd, int cr, int lane) {
  {  // most candidates of a long run lose to the 32nd entry: one readlane pair and a scalar compare, no cross-lane traffic
    const unsigned long long b31 = __double_as_longlong(ts);
    const double s31 = __longlong_as_double(((unsigned long long)__builtin_amdgcn_readlane((unsigned)(b31 >> 32), 31) << 32) |
                                            (unsigned)__builtin_amdgcn_readlane((unsigned)b31, 31));
    const int r31 = __builtin_amdgcn_readlane(tr, 31);
    if (s31 > cd || (s31 == cd && r31 <= cr)) return;
  }
  if (__ballot(lane < 32 && tr == cr) != 0ull) return;
  const unsigned long long ahead = __ballot(lane < 32 && (ts > cd || (ts == cd && tr < cr)));
  const int pos = __popcll(ahead);
  if (pos >= 32) return;
  const double up_s = __shfl_up(ts, 1);
  const int up_r = __shfl_up(tr, 1);
  if (lane == pos) {
    ts = cd;
    tr = cr;
  } else if (lane > pos && lane < 32) {
    ts = up_s;
    tr = up_r;
  }
}

template <int NW = 4>  // waves of the workgroup (4: rerank_kernel)
__device__ __forceinline__ void wg_exact_scan(const float* __restrict__ db, int n_rows, const float* __restrict__ qrow, int K,
                                              int row_offset, int32_t* __restrict__ out_idx, double* __restrict__ out_score,
                                              WgExactSharedN<NW>& sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seg = lane & 15, grp = lane >> 4;
  double qd[16];
  {
    const float4* qp = reinterpret_cast<const float4*>(qrow) + seg;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = qp[16 * i];
      qd[4 * i] = (double)v.x;
      qd[4 * i + 1] = (double)v.y;
      qd[4 * i + 2] = (double)v.z;
      qd[4 * i + 3] = (double)v.w;
    }
  }
  double my_s = -__builtin_inf();  // lane i (< 32): the i-th best (score, row) this wave has seen
  int my_r = INT_MAX;
  const int n_groups = (n_rows + 3) / 4;
  for (int g = wave; g < n_groups; g += NW) {
    const int row = 4 * g + grp;
    const float4* rp = reinterpret_cast<const float4*>(db + (size_t)min(row, n_rows - 1) * kD) + seg;
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 rv = rp[16 * i];
      d0 += (double)rv.x * qd[4 * i];
      d1 += (double)rv.y * qd[4 * i + 1];
      d0 += (double)rv.z * qd[4 * i + 2];
      d1 += (double)rv.w * qd[4 * i + 3];
    }
    const double d = row16_sum_f64(d0 + d1);  // every lane of 16-lane row `grp` holds the score of DB row `row`
#pragma unroll
    for (int c = 0; c < 4; ++c) {  // the 4 candidates of this pass, one after the other (rows ascend with c)
      const double cd = __shfl(d, 16 * c);
      const int cr = 4 * g + c;
      if (cr >= n_rows) break;  // wave-uniform
      // position = how many list entries stay ahead of (cd, cr); nothing to do when all 32 do
      const unsigned long long ahead = __ballot(lane < 32 && (my_s > cd || (my_s == cd && my_r < cr)));
      const int pos = __popcll(ahead);
      if (pos >= 32) continue;
      const double up_s = __shfl_up(my_s, 1);
      const int up_r = __shfl_up(my_r, 1);
      if (lane == pos) {
        my_s = cd;
        my_r = cr;
      } else if (lane > pos && lane < 32) {
        my_s = up_s;
        my_r = up_r;
      }
    }
  }
  if (lane < 32) {
    sh.score[wave][lane] = my_s;
    sh.row[wave][lane] = my_r;
  }
  __syncthreads();
  if (wave == 0) {  // 32 NW entries, NW / 2 per lane: rank by (score desc, row asc), the first K go out
    constexpr int PER = NW / 2, TOT = 32 * NW;
    double sv[PER];
    int rv[PER], kv[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      sv[e] = (&sh.score[0][0])[64 * e + lane];
      rv[e] = (&sh.row[0][0])[64 * e + lane];
      kv[e] = 0;
    }
    for (int o = 0; o < TOT; ++o) {
      const double os = (&sh.score[0][0])[o];
      const int orow = (&sh.row[0][0])[o];
#pragma unroll
      for (int e = 0; e < PER; ++e) kv[e] += (os > sv[e] || (os == sv[e] && orow < rv[e])) ? 1 : 0;
    }
    if (lane < K) {
      out_idx[lane] = -1;
      if (out_score) out_score[lane] = -__builtin_inf();
    }
#pragma unroll
    for (int e = 0; e < PER; ++e)
      if (rv[e] != INT_MAX && kv[e] < K) {
        out_idx[kv[e]] = rv[e] + row_offset;
        if (out_score) out_score[kv[e]] = sv[e];
      }
  }
  __syncthreads();
}


// The report card of the PREVIOUS call (its counters were parked at [64..] by reset_counts): mapped host memory the host reads at a
// later call — no stream operation, no synchronisation (it only steers heuristics); the sequence number is published LAST with
// system-scope release ordering, the host reads it first with an acquire load, so a new sequence number is never paired with the
// previous call's counts. One thread of one workgroup per call.
__device__ __forceinline__ void publish_report(const RerankArgs& a) {
  int32_t* fb_count = a.fb_count;
  if (a.host_stat && a.seq > 0) {
    a.host_stat[1] = fb_count[64 + 10] == 2 ? fb_count[64 + 3] : fb_count[64 + 2];  // f16-certificate failures (or the probe's)
    a.host_stat[2] = fb_count[64 + 9];
    a.host_stat[3] = fb_count[64 + 0] + fb_count[64 + 4] + fb_count[64 + 12];  // exact-stage queries
    a.host_stat[4] = fb_count[64 + 10];  // 0: not an f16-certificate count, 1: the f16 scan's own, 2: the split-bf16 stand-in's probe
    a.host_stat[5] = fb_count[64 + 10] == 2 ? fb_count[64 + 3] : fb_count[64 + 1];  // first-certificate failures (settled in the wave or not)
    __hip_atomic_store(&a.host_stat[0], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  fb_count[9] = a.Q;
  fb_count[10] = a.stat_mode;  // 0: not an f16-certificate count, 1: f16 scan, 2: split-bf16 stand-in probing for it
}

// One query, one wave. `my_wg_flag`: this wave's slot of the workgroup's "rank exactly" flags; `wr_buf`: kWideCap ints of LDS.
// MG: the lists are the MERGED records of scanp_kernel<..., MERGE> — `parts` = physical splits, one record of kMergedLL floats per
// (query, split): 7 keys whose low bits are code << 2 | source list, then the bound on every key the record does not list (what
// `lane_floor` is for a plain list: the floor of a full list).
template <int LL, int L, bool MG = false>
__device__ __forceinline__ void rerank_query(const RerankArgs& a, const int qid, const int lane, int* my_wg_flag, int* wr_buf) {
  static_assert(!MG || LL == kMergedLL, "merged records are 8 floats");
  const float* __restrict__ db = a.db;
  const float* __restrict__ q = a.q;
  const int Q = a.Q, K = a.K, parts = a.parts, code_bits = a.code_bits, row_offset = a.row_offset, half_mode = a.half_mode;
  const int slack_bits = MG ? code_bits + 2 : code_bits;  // score bits a key gave up for its code
  const int strided_rows = half_mode ? a.n_rows : 0;       // keys of the f16 scans index the strided plane (plane_row)
  // row of a key held by list `part`
  auto krow = [&](float key, int part) {
    if constexpr (MG) {
      const int b = __float_as_int(key), src = b & 3;
      return key_row(__int_as_float(b >> 2), 2 * (part + (src >> 1) * parts) + (src & 1), 2 * parts, code_bits, strided_rows);
    } else {
      return key_row(key, part, parts >> 1, code_bits, strided_rows);
    }
  };
  const float* __restrict__ cand = a.cand;
  const float eps_rel = a.eps_rel, eps_rel_probe = a.eps_rel_probe, pinf = a.pinf;
  const float* __restrict__ db_norm_max = a.db_norm_max;
  int32_t* __restrict__ out_idx = a.out_idx;
  double* __restrict__ out_score = a.out_score;
  int32_t* __restrict__ flags = a.flags;
  int32_t* __restrict__ fb_count = a.fb_count;
  const int n_rows = a.n_rows, defer = a.defer, wide_cap = a.wide_cap;

  // ---- every lane pulls its whole sorted key list into registers (one memory latency for the merge)
  float lst[LL];
  {
    const float* mine = cand + ((size_t)qid * parts + min(lane, parts - 1)) * LL;
    if constexpr (MG) {
      const float4 v0 = reinterpret_cast<const float4*>(mine)[0], v1 = reinterpret_cast<const float4*>(mine)[1];
      lst[0] = v0.x; lst[1] = v0.y; lst[2] = v0.z; lst[3] = v0.w;
      lst[4] = v1.x; lst[5] = v1.y; lst[6] = v1.z; lst[7] = v1.w;
    } else if constexpr (LL % 2 == 0) {  // (LL = 6 lists are 24 bytes: 8-byte loads keep every list aligned)
#pragma unroll
      for (int i = 0; i < LL / 2; ++i) {
        const float2 v = reinterpret_cast<const float2*>(mine)[i];
        lst[2 * i] = v.x;
        lst[2 * i + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int i = 0; i < LL; ++i) lst[i] = mine[i];
    }
    if (lane >= parts) {
#pragma unroll
      for (int i = 0; i < LL; ++i) lst[i] = T2L_NEG_INF;
    }
  }
  // a FULL list dropped rows inside its lane: all of them have key <= its floor (-inf when nothing was dropped)
  const float lane_floor = lst[LL - 1];
  if constexpr (MG) lst[LL - 1] = T2L_NEG_INF;  // (the record's last slot is its bound, not a key)
  // records of the tile-local selection (scanp_kernel<..., SEL = 1>): slot 6 is B1, the largest THIRD-best key of a tile-local group
  // (8 rows) of the record's four lanes — everything those lanes dropped past their lists lies at or below it, and all of it except
  // the other 5 rows of B1's own group at or below the record's bound B2 (slot 7, `lane_floor`)
  float b1 = T2L_NEG_INF;
  if constexpr (MG) {
    if (a.rec6) {
      b1 = lst[LL - 2];
      lst[LL - 2] = T2L_NEG_INF;
    }
  }
  const float floor_max = wave_max_f32(fmaxf(lane_floor, b1), pinf);
  // the query as float64, 16 elements per lane (dims 64 i + 4 seg + e: each load instruction reads 256 contiguous bytes
  // per 16-lane row); the lanes of one row cover the 256 dims, the 4 rows hold copies
  const int seg = lane & 15;
  double qd[16];
  float q_absmax = 0.f;
  {
    const float4* qp = reinterpret_cast<const float4*>(q + (size_t)qid * kD) + seg;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = qp[16 * i];
      qd[4 * i] = (double)v.x;
      qd[4 * i + 1] = (double)v.y;
      qd[4 * i + 2] = (double)v.z;
      qd[4 * i + 3] = (double)v.w;
      q_absmax = fmaxf(fmaxf(fmaxf(q_absmax, fabsf(v.x)), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
  }

  // ---- merge the `parts` lists into the top-L keys: L rounds of wave arg-max over the list heads. Round r leaves (key, part)
  // in lane r; the row is decoded from them ONCE after the loop (it was 12 instructions inside every round), and the loop is
  // unrolled so that popping the winner's head is 6 selects with no register copies behind them.
  // Round 5: only the first L - 4 rounds run up front — they are all the early certificate below needs (9 queries in 10 stop there);
  // the last four rounds run only on the way into the full path, in front of the four rows they select.
  float my_key = T2L_NEG_INF;
  int my_part = 0;
  auto merge_round = [&](int r) {
    const float bk = wave_max_f32(lst[0], pinf);
    const unsigned long long who = __ballot(lst[0] == bk);
    const int bl = __ffsll((long long)who) - 1;  // equal keys: lowest part first
    if (lane == r) {
      my_key = bk;
      my_part = bl;
    }
    const bool pop = lane == bl;  // the winner pops its head (register shift, no memory)
#pragma unroll
    for (int i = 0; i < LL - 1; ++i) lst[i] = pop ? lst[i + 1] : lst[i];
    lst[LL - 1] = pop ? T2L_NEG_INF : lst[LL - 1];
  };
#pragma unroll
  for (int r = 0; r < L - 4; ++r) merge_round(r);
  int my_row = (lane < L - 4 && my_key != T2L_NEG_INF) ? krow(my_key, my_part) : INT_MAX;

  // ---- certificate scale (keys of the f16 scan are true scores times 2^(shift_db + shift_q): undo that exactly)
  double kscale = 1.0;
  bool representable = true;
  {
    const float m = row16_max_f32(q_absmax);
    int sq, sd;
    representable = half_shift_of(m, sq);
    representable = half_shift_of(db_norm_max[1], sd) && representable;
    if (half_mode) {
      kscale = ldexp(1.0, -(sq + sd));
    } else {
      // the f32 / split-bf16 scans multiply the raw values: their relative error bound needs every product that matters
      // to stay a normal f32, which magnitudes within 2^-40 .. 2^40 guarantee with a wide margin
      representable = representable && abs(sq - 14) <= 40 && abs(sd - 14) <= 40;
    }
  }

  // ---- float64 re-score of the selected rows (products of f32 values are exact in f64). Four rows at a time, one per
  // 16-lane row of the wave: a lane multiplies 16 elements and the row sum is 4 DPP steps (a whole-wave reduction per
  // row costs 3x the VALU). The re-rank is bound by these gathers (16 KB per query from L2 / Infinity Cache: skipping four
  // of the sixteen rows was measured at -1.3 us per step), so they come in two stages: the best L - 4 candidates first, and
  // when the certificate already holds against the (L-4)-th merged key — 9 queries in 10 — the last four rows are never
  // fetched: everything not re-scored, those four included, has a key at or below that bound.
  double qn = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) qn += qd[i] * qd[i];
  qn = row16_sum_f64(qn);
  const double eps32 = (double)eps_rel * sqrt(qn) * (double)(*db_norm_max);
  constexpr int LA = L - 4;
  float4 rows[L / 4][4];
  auto gather = [&](int p) {
    const int row = __shfl(my_row, 4 * p + (lane >> 4));  // candidate 4p + g is re-scored by 16-lane row g
#ifdef T2L_EXP_RERANK_TWICE  // dev experiment: every gather also fetches a second, unrelated row (results unchanged): is the re-rank
    {                         // bound by its gather traffic? (tools/listwrite_probe.py; measured: see DESIGN 3.2)
      const int other = row == INT_MAX ? 0 : (row * 7 + 13) % n_rows;
      const float4* op = reinterpret_cast<const float4*>(db + (size_t)other * kD) + seg;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float4 v = op[16 * i];
        asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
      }
    }
#endif
    const float4* rp = reinterpret_cast<const float4*>(db + (size_t)(row == INT_MAX ? 0 : row) * kD) + seg;
#pragma unroll
    for (int i = 0; i < 4; ++i) rows[p][i] = rp[16 * i];
  };
  double my_d = -__builtin_inf();
  auto score = [&](int p) {
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      d0 += (double)rows[p][i].x * qd[4 * i];
      d1 += (double)rows[p][i].y * qd[4 * i + 1];
      d0 += (double)rows[p][i].z * qd[4 * i + 2];
      d1 += (double)rows[p][i].w * qd[4 * i + 3];
    }
    const double d = row16_sum_f64(d0 + d1);
    const double mine = __shfl(d, 16 * (lane & 3));  // lane c = 4p + g takes the sum of row g
    if ((lane >> 2) == p && lane < L && my_row != INT_MAX) my_d = mine;
  };
  // order by (float64 score desc, row asc) among the first N candidates: lane c < N computes its rank
  auto rank_among = [&](auto n_tag) {
    constexpr int N = decltype(n_tag)::value;
    int r = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const unsigned long long bj = __double_as_longlong(my_d);
      const double dj = __longlong_as_double(((unsigned long long)__builtin_amdgcn_readlane((unsigned)(bj >> 32), j) << 32) |
                                             (unsigned)__builtin_amdgcn_readlane((unsigned)bj, j));
      const int ij = __builtin_amdgcn_readlane(my_row, j);
      r += (dj > my_d || (dj == my_d && ij < my_row)) ? 1 : 0;
    }
    return r;
  };
  auto emit = [&](bool ok, int r) {
    if (lane < K) {
      out_idx[(size_t)qid * K + lane] = -1;
      if (out_score) out_score[(size_t)qid * K + lane] = -__builtin_inf();
    }
    if (ok && r < K) {
      out_idx[(size_t)qid * K + r] = my_row + row_offset;
      if (out_score) out_score[(size_t)qid * K + r] = my_d;
    }
  };
#pragma unroll
  for (int p = 0; p < L / 4 - 1; ++p) gather(p);
#pragma unroll
  for (int p = 0; p < L / 4 - 1; ++p) score(p);
  bool early = false;
  if (K <= LA && representable && eps_rel_probe == 0.f) {  // (the stand-in probe counts against the full certificate below)
    const float gA = fmaxf(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_key), LA - 1)), floor_max);
    if (gA != T2L_NEG_INF) {  // (fewer than L - 4 candidates: the full path certifies trivially)
      const int rA = rank_among(std::integral_constant<int, LA>{});
      const bool validA = lane < LA && my_row != INT_MAX;
      const unsigned long long kthA = __ballot(validA && rA == K - 1);
      if (kthA != 0ull) {
        const double dK = __shfl(my_d, __ffsll((long long)kthA) - 1);
        if (dK > (double)gA * kscale + key_slack(gA, slack_bits, eps32, kscale)) {  // wave-uniform
          emit(validA, rA);
          early = true;
        }
      }
    }
  }
  if (early) {
    if (lane == 0) flags[qid] = 0;
  } else {
#pragma unroll
  for (int r = L - 4; r < L; ++r) merge_round(r);
  if (lane >= L - 4 && lane < L && my_key != T2L_NEG_INF) my_row = krow(my_key, my_part);
  // every row that is NOT re-scored has key <= g: kept rows lie at or below the L-th merged key, rows dropped inside a
  // lane at or below that lane's floor (with LL == L a floor above the L-th key cannot happen; with LL < L it can)
  const float key_L = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_key), L - 1));
  const float g = fmaxf(key_L, floor_max);
  gather(L / 4 - 1);
  // ---- the group repair, anticipated (records of the tile-local selection). Three of a query's best rows in the 8 rows of ONE
  // tile-local group (1e-4 per query at N = 11 k: a query in every other batch of 4,096) leave the third as some record's B1 above
  // everything the query can certify against. The first L - 4 scores already tell: with the K-th best of them as a provisional
  // K-th score (the final one is no smaller) exactly one record's B1 reaches the threshold and no B2 does. Then the 8 rows of B1's
  // group travel TOGETHER with the last four candidates (three row groups in flight, as in the first stage) — the query costs the
  // round trips of any query that missed the early certificate (1 in 10) — and the certificate is taken against B2 in that
  // record's place. Anything else (two such records, a B2 in reach, the certificate still failing) resets the extra lanes and
  // takes the general path below.
  bool fast_done = false;
  if constexpr (MG && L == 16) {
    int d_lane = -1;
    // (cheap gate first: a B1 can only be what stands in the way when it lies above every B2 and above the L-th merged key)
    const float b1_max = a.rec6 ? wave_max_f32(b1, pinf) : T2L_NEG_INF;
    if (b1_max != T2L_NEG_INF && b1_max > fmaxf(key_L, wave_max_f32(lane_floor, pinf)) && representable && K <= LA && eps_rel_probe == 0.f) {
      const int rA = rank_among(std::integral_constant<int, LA>{});
      const unsigned long long kthA = __ballot(lane < LA && my_row != INT_MAX && rA == K - 1);
      if (kthA != 0ull) {
        const double dKA = __shfl(my_d, __ffsll((long long)kthA) - 1);
        const double tA = (dKA - 2.0 * (fabs(dKA) * ldexp(1.0, slack_bits - 22) + eps32)) / kscale;
        float thrA = (float)tA;
        if ((double)thrA > tA) thrA = __uint_as_float(__float_as_uint(thrA) + (thrA > 0.f ? -1 : 1));  // round down
        if (fabs(tA) < 3.0e38) {
          const unsigned long long bm = __ballot(lane_floor != T2L_NEG_INF && lane_floor >= thrA);
          const unsigned long long dm = __ballot(b1 != T2L_NEG_INF && b1 >= thrA);
          if (bm == 0ull && __popcll(dm) == 1) d_lane = __ffsll((long long)dm) - 1;
        }
      }
    }
    if (d_lane >= 0) {  // wave-uniform
      const int kb = __shfl(__float_as_int(b1), d_lane);
      int cand_row = INT_MAX;
      if (lane >= L && lane < L + kTileSelGroup) {  // key code: tile ordinal << 4 | accumulator register (its bit 3 = the group), << 2 | source
        const int r = krow(__int_as_float((kb & ~((kTileSelGroup - 1) << 2)) | ((lane - L) << 2)), d_lane);
        cand_row = r < n_rows ? r : INT_MAX;
      }
#pragma unroll
      for (int j = 0; j < L; ++j) cand_row = __builtin_amdgcn_readlane(my_row, j) == cand_row ? INT_MAX : cand_row;  // a row enters once
      if (lane >= L && lane < L + kTileSelGroup) my_row = cand_row;
      auto gather_d = [&](int p, float4 (&rw)[4]) {
        const int row = __shfl(my_row, L + 4 * p + (lane >> 4));
        const float4* rp = reinterpret_cast<const float4*>(db + (size_t)(row == INT_MAX ? 0 : row) * kD) + seg;
#pragma unroll
        for (int i = 0; i < 4; ++i) rw[i] = rp[16 * i];
      };
      auto score_d = [&](int p, const float4 (&rw)[4]) {
        double d0 = 0.0, d1 = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          d0 += (double)rw[i].x * qd[4 * i];
          d1 += (double)rw[i].y * qd[4 * i + 1];
          d0 += (double)rw[i].z * qd[4 * i + 2];
          d1 += (double)rw[i].w * qd[4 * i + 3];
        }
        const double d = row16_sum_f64(d0 + d1);
        const double mine = __shfl(d, 16 * (lane & 3));
        if ((lane >> 2) == L / 4 + p && my_row != INT_MAX) my_d = mine;
      };
      static_assert(kTileSelGroup == 8, "two row groups of four");
      gather_d(0, rows[0]);
      gather_d(1, rows[1]);
      score(L / 4 - 1);
      score_d(0, rows[0]);
      score_d(1, rows[1]);
      const int rank2 = rank_among(std::integral_constant<int, L + kTileSelGroup>{});
      const bool valid2 = lane < L + kTileSelGroup && my_row != INT_MAX;
      const unsigned long long kth2 = __ballot(valid2 && rank2 == K - 1);
      // what is not re-scored now: kept keys at or below the L-th merged key, every record's B2, every OTHER record's B1
      const float g2 = fmaxf(key_L, wave_max_f32(lane == d_lane ? lane_floor : fmaxf(lane_floor, b1), pinf));
      if (kth2 != 0ull) {
        const double dK2 = __shfl(my_d, __ffsll((long long)kth2) - 1);
        if (dK2 > (double)g2 * kscale + key_slack(g2, slack_bits, eps32, kscale)) {
          emit(valid2, rank2);
          if (lane == 0) {
            flags[qid] = 0;
            atomicAdd(&fb_count[1], 1);  // re-scored beyond the first L candidates (the group repair)
          }
          fast_done = true;
        }
      }
      if (!fast_done && lane >= L) {  // the general path starts from the L candidates
        my_row = INT_MAX;
        my_d = -__builtin_inf();
      }
    } else {
      score(L / 4 - 1);
    }
  } else {
    score(L / 4 - 1);
  }
  if (!fast_done) {
  const int rank = rank_among(std::integral_constant<int, L>{});
  const bool valid = lane < L && my_row != INT_MAX;
  emit(valid, rank);

  bool certified = true;
  float thr = T2L_NEG_INF;  // fallback: only rows with key >= thr can still reach the top-K
  if (g != T2L_NEG_INF) {  // an L-th candidate exists, so something may not have been re-scored
    certified = false;
    const unsigned long long kth = __ballot(valid && rank == K - 1);
    if (K <= L && kth != 0ull) {
      const double dK = __shfl(my_d, __ffsll((long long)kth) - 1);
      certified = dK > (double)g * kscale + key_slack(g, slack_bits, eps32, kscale);
      // while the split-bf16 scan stands in for the f16 scan on a DB that overwhelmed its certificate, count the queries
      // the f16 error band would still flag: the host goes back to the f16 scan when they become rare
      if (eps_rel_probe > 0.f && lane == 0 &&
          !(dK > (double)g * kscale + key_slack(g, slack_bits, (double)eps_rel_probe * sqrt(qn) * (double)(*db_norm_max), kscale)))
        atomicAdd(&fb_count[3], 1);
      // a row with key k has score <= k*kscale + |k|*kscale*2^(cb-22) + eps32; with S = 2*(|dK|*2^(cb-22) + eps32) every
      // key below (dK - S)/kscale is therefore below the current K-th best score dK (and the final K-th is >= dK)
      const double S = 2.0 * (fabs(dK) * ldexp(1.0, slack_bits - 22) + eps32);
      const double t = (dK - S) / kscale;
      thr = (float)t;
      if ((double)thr > t) thr = __uint_as_float(__float_as_uint(thr) + (thr > 0.f ? -1 : 1));  // round down
      if (!(fabs(t) < 3.0e38)) thr = T2L_NEG_INF;
    }
  }
  // ---- the usual way out of a failed certificate, without leaving the wave: the lists (what the merge left of them) are
  // still in registers. If at most 16 more kept keys reach the threshold and no full list's floor does, re-scoring exactly
  // those — same arithmetic as above, so equal rows keep bit-equal scores — makes the top-K exact by construction (the
  // argument of the fallback kernel's stage 2a, which remains for the larger cases).
  bool settled = false;
  if (!certified && representable && thr != T2L_NEG_INF && L + 16 <= 64) {
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < LL; ++i) cnt += (lst[i] >= thr) ? 1 : 0;
    int total = cnt;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) total += __shfl_xor(total, off);
    bool bad = lane_floor != T2L_NEG_INF && lane_floor >= thr;  // this list dropped rows that could still matter
    // ... or only B1 says so: then the rows that could are the 8 of B1's tile-local group (B2 bounds the rest of what the record dropped)
    const bool dbad = !bad && b1 != T2L_NEG_INF && b1 >= thr;
    unsigned long long bad_mask = __ballot(bad);
    const unsigned long long d_mask = __ballot(dbad);
    const int n_d = __popcll(d_mask);
    if (bad_mask == 0ull && total <= 16 && total + kTileSelGroup * n_d <= 64 - L) {
      for (int e = 0; e < total; ++e) {  // the next best kept keys, in key order, into lanes L, L+1, ...
        const float bk = wave_max_f32(lst[0], pinf);
        const int bl = __ffsll((long long)__ballot(lst[0] == bk)) - 1;
        if (lane == L + e) {
          my_key = bk;
          my_row = krow(bk, bl);
        }
        if (lane == bl) {
#pragma unroll
          for (int i = 0; i < LL - 1; ++i) lst[i] = lst[i + 1];
          lst[LL - 1] = T2L_NEG_INF;
        }
      }
      int n_ent = L + total;
      if constexpr (MG) {
        // the tile-local groups behind the B1 keys that reach the threshold: all 8 rows of each (its two best are usually candidates
        // already: a row enters once), compacted through this wave's LDS scratch into the lanes behind the kept keys
        for (unsigned long long m = d_mask; m != 0ull; m &= m - 1ull) {
          const int b = __ffsll((long long)m) - 1;
          const int kb = __shfl(__float_as_int(b1), b);
          int cand_row = INT_MAX;
          if (lane < kTileSelGroup) {  // key code = tile ordinal << 4 | accumulator register (bit 3 = the group), << 2 | source
            const int r = krow(__int_as_float((kb & ~((kTileSelGroup - 1) << 2)) | (lane << 2)), b);
            cand_row = r < n_rows ? r : INT_MAX;
          }
          bool dup = cand_row == INT_MAX;
          for (int j = 0; j < n_ent; ++j) dup = dup || __builtin_amdgcn_readlane(my_row, j) == cand_row;
          const unsigned long long keep = __ballot(!dup);
          if (!dup) wr_buf[__popcll(keep & ((1ull << lane) - 1ull))] = cand_row;
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (this wave's LDS writes are read back by its other lanes)
          const int nk = __popcll(keep);
          if (lane >= n_ent && lane < n_ent + nk) my_row = wr_buf[lane - n_ent];
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          n_ent += nk;
        }
      }
      for (int p = L / 4; p < (n_ent + 3) / 4; ++p) {
        const int row = __shfl(my_row, 4 * p + (lane >> 4));
        const float4* rp = reinterpret_cast<const float4*>(db + (size_t)(row == INT_MAX ? 0 : row) * kD) + seg;
        double d0 = 0.0, d1 = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 rv = rp[16 * i];
          d0 += (double)rv.x * qd[4 * i];
          d1 += (double)rv.y * qd[4 * i + 1];
          d0 += (double)rv.z * qd[4 * i + 2];
          d1 += (double)rv.w * qd[4 * i + 3];
        }
        const double d = row16_sum_f64(d0 + d1);
        const double mine = __shfl(d, 16 * (lane & 3));
        if ((lane >> 2) == p && lane < n_ent && my_row != INT_MAX) my_d = mine;
      }
      int rank2 = 0;
      for (int j = 0; j < n_ent; ++j) {
        const unsigned long long bj = __double_as_longlong(my_d);
        const double dj = __longlong_as_double(((unsigned long long)__builtin_amdgcn_readlane((unsigned)(bj >> 32), j) << 32) |
                                               (unsigned)__builtin_amdgcn_readlane((unsigned)bj, j));
        const int ij = __builtin_amdgcn_readlane(my_row, j);
        rank2 += (dj > my_d || (dj == my_d && ij < my_row)) ? 1 : 0;
      }
      if (lane < n_ent && my_row != INT_MAX && rank2 < K) {
        out_idx[(size_t)qid * K + rank2] = my_row + row_offset;
        if (out_score) out_score[(size_t)qid * K + rank2] = my_d;
      }
      settled = true;
    } else {
      // ---- the WIDE repair: more than 16 kept keys reach the threshold, or full lists dropped rows that could (their floor
      // reaches it) — what a database of tight clusters does to a few queries long before it defeats the certificates
      // wholesale. Every row whose key reaches the threshold is either a kept key of a list that dropped nothing relevant, or
      // a row of one of the `bad` lists: re-score the former and ALL rows of the latter (a list covers per * 16 rows of the
      // shard) — same arithmetic as above — into a running top-32 spread over the lanes; rows below the threshold cannot reach
      // the top-K (the argument of the in-wave re-score above). Only when that is more than kWideCap rows does the query go
      // to an exact scan of the whole shard (one workgroup: ~300 us per query at N = 11 k — two such queries were 0.6 ms of
      // a 0.7 ms step on clustered data, bench_distribution.py).
      // rows behind a list: a plain list covers the tiles of ONE virtual split at one lane half (16 rows per tile); a merged record the
      // two virtual splits of its workgroup at both halves (64 per tile ordinal; the second split may be one tile shorter: filtered below)
      // (a record whose B1 alone reaches the threshold contributes its kept keys, like any good list, and the 8 rows of B1's group)
      const int vn_ = MG ? 2 * parts : parts >> 1;
      constexpr int kRowsPerTile = MG ? 64 : 16;
      const int n_tiles_ = (n_rows + kTileRows - 1) / kTileRows;
      const int my_vs = MG ? lane : lane >> 1;
      const int my_nt = (lane < parts && my_vs < n_tiles_) ? (n_tiles_ - my_vs + vn_ - 1) / vn_ : 0;
      int work = bad ? my_nt * kRowsPerTile : cnt + (dbad ? kTileSelGroup : 0);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) work += __shfl_xor(work, off);
      if (work <= wide_cap) {
        int* wr = wr_buf;
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
        int base = 0;
#pragma unroll
        for (int i = 0; i < LL; ++i) {  // kept keys of the lists that dropped nothing relevant
          const bool pred = !bad && lst[i] >= thr;
          const unsigned long long m = __ballot(pred);
          if (pred) wr[base + __popcll(m & lt_mask)] = krow(lst[i], lane);
          base += __popcll(m);
        }
        if constexpr (MG) {
          for (unsigned long long m = d_mask; m != 0ull; m &= m - 1ull) {  // the tile-local groups behind the B1 keys in reach
            const int b = __ffsll((long long)m) - 1;
            const int kb = __shfl(__float_as_int(b1), b);
            if (lane < kTileSelGroup) {
              const int row = krow(__int_as_float((kb & ~((kTileSelGroup - 1) << 2)) | (lane << 2)), b);
              wr[base + lane] = row < n_rows ? row : INT_MAX;
            }
            base += kTileSelGroup;
          }
        }
        for (unsigned long long m = bad_mask; m != 0ull; m &= m - 1ull) {  // every row of the lists that did
          const int b = __ffsll((long long)m) - 1;
          const int nb = __shfl(my_nt, b) * kRowsPerTile;
          for (int idx = lane; idx < nb; idx += 64) {  // (a list position IS its key code: tile ordinal << 4 | accumulator register [<< 2 | source])
            const int row = krow(__int_as_float(idx), b);
            wr[base + idx] = row < n_rows ? row : INT_MAX;
          }
          base += nb;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (this wave's LDS writes are read back by its other lanes)
        double ts = -__builtin_inf();
        int tr = INT_MAX;
        for (int j = 0; j < L; ++j) {  // what is already re-scored
          const double cd = __shfl(my_d, j);
          const int cr = __shfl(my_row, j);
          if (cr != INT_MAX) top32_insert(ts, tr, cd, cr, lane);
        }
        const int n_round = (base + 3) >> 2;
        auto row_of = [&](int p) {
          const int i = 4 * p + (lane >> 4);
          return i < base ? wr[i] : INT_MAX;
        };
        float4 cur[4], nxt[4];
        int row_c = n_round > 0 ? row_of(0) : INT_MAX, row_n = INT_MAX;
        {
          const float4* rp = reinterpret_cast<const float4*>(db + (size_t)(row_c == INT_MAX ? 0 : row_c) * kD) + seg;
#pragma unroll
          for (int i = 0; i < 4; ++i) cur[i] = rp[16 * i];
        }
        for (int p = 0; p < n_round; ++p) {
          if (p + 1 < n_round) {  // the next round's rows travel while this round's are multiplied (wave-uniform)
            row_n = row_of(p + 1);
            const float4* rp = reinterpret_cast<const float4*>(db + (size_t)(row_n == INT_MAX ? 0 : row_n) * kD) + seg;
#pragma unroll
            for (int i = 0; i < 4; ++i) nxt[i] = rp[16 * i];
          }
          double d0 = 0.0, d1 = 0.0;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            d0 += (double)cur[i].x * qd[4 * i];
            d1 += (double)cur[i].y * qd[4 * i + 1];
            d0 += (double)cur[i].z * qd[4 * i + 2];
            d1 += (double)cur[i].w * qd[4 * i + 3];
          }
          const double d = row16_sum_f64(d0 + d1);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const double cd = __shfl(d, 16 * c);
            const int cr = __shfl(row_c, 16 * c);
            if (cr != INT_MAX) top32_insert(ts, tr, cd, cr, lane);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
          row_c = row_n;
        }
        if (lane < K) {
          out_idx[(size_t)qid * K + lane] = tr == INT_MAX ? -1 : tr + row_offset;
          if (out_score) out_score[(size_t)qid * K + lane] = ts;
        }
        if (lane == 0) atomicAdd(&fb_count[5], 1);
        settled = true;
      }
    }
  }
  // 2 = the f16 operands of this query (or of the DB) were not representable: its keys mean nothing, scan exactly
  if (lane == 0) {
    const int flag = !representable ? 2 : ((certified || settled) ? 0 : 1);
    flags[qid] = flag;
    if (!certified || !representable) atomicAdd(&fb_count[1], 1);  // first certificate failed (settled in the wave or not)
    if (flag) {
      atomicAdd(&fb_count[2], 1);
      if (!defer) *my_wg_flag = flag;  // settled below, by this workgroup
      else if (flag == 2) flags[4 * Q + atomicAdd(&fb_count[6], 1)] = qid;  // heavy mode: -> exact_list_kernel
      else flags[3 * Q + atomicAdd(&fb_count[4], 1)] = qid;                  //             -> exactd_kernel
    }
  }
  }  // !fast_done
  }  // !early
}

// ------------------------------------------------------------------------------------------------
// rerank (stage 1): one wave per query, 4 queries per 256-thread block.
// ------------------------------------------------------------------------------------------------
// LL = length of the per-lane lists the scan wrote, L = rows re-scored per query (LL <= L).
template <int LL, int L, bool MG = false>
// (the merged-record form holds its group-repair path inside 128 VGPRs: four waves per SIMD = all 4,096 query waves of a batch resident at once;
//  the other forms keep the occupancy the compiler finds — L = 32 takes 139 VGPRs and would spill under the bound)
__global__ __launch_bounds__(256, (MG ? 4 : 1)) void rerank_kernel(const float* __restrict__ db, const float* __restrict__ q, int Q,
                                                     int K, int parts, int code_bits,
                                                     const float* __restrict__ cand, int row_offset, float eps_rel,
                                                     const float* __restrict__ db_norm_max, int half_mode,
                                                     int32_t* __restrict__ out_idx, double* __restrict__ out_score,
                                                     int32_t* __restrict__ flags, int32_t* __restrict__ fb_count,
                                                     float eps_rel_probe, float pinf, int n_rows, int defer, int stat_mode,
                                                     int32_t* __restrict__ host_stat, int seq, int wide_cap, int rec6) {
  __shared__ WgExactShared exact_sh;
  __shared__ int wg_flag[4];
  __shared__ int wide_rows[4][kWideCap];  // per wave: the rows a wide repair re-scores
  const RerankArgs a{db, q, Q, K, parts, code_bits, cand, row_offset, eps_rel, db_norm_max, half_mode, out_idx, out_score, flags, fb_count,
                     eps_rel_probe, pinf, n_rows, defer, stat_mode, host_stat, seq, wide_cap, rec6};
  const int lane = threadIdx.x & 63;
  const int qid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (threadIdx.x < 4) wg_flag[threadIdx.x] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) publish_report(a);
  __syncthreads();
  if (qid < Q) rerank_query<LL, L, MG>(a, qid, lane, &wg_flag[threadIdx.x >> 6], wide_rows[threadIdx.x >> 6]);
  // ---- unsettled queries of this workgroup: the exact float64 ranking, all 4 waves on one query at a time
  __syncthreads();
#pragma unroll 1
  for (int w = 0; w < 4; ++w) {
    if (!wg_flag[w]) continue;  // workgroup-uniform
    const int fq = blockIdx.x * 4 + w;
    if (threadIdx.x == 0) atomicAdd(&fb_count[0], 1);
    wg_exact_scan(db, n_rows, q + (size_t)fq * kD, K, row_offset, out_idx + (size_t)fq * K,
                  out_score ? out_score + (size_t)fq * K : nullptr, exact_sh);
  }
}

// max row 2-norm of the shard (bounds the f32 dot-product error in the certificate); one atomic per workgroup
// out[0] = max row 2-norm (x 1.0001), out[1] = max |element| as raw bits (inf / NaN payloads order above every finite
// value, so a non-finite DB is seen as such by half_shift_of)
__global__ __launch_bounds__(256) void db_norm_kernel(const float* __restrict__ db, int n_rows, float* out) {
  __shared__ float red[4];
  __shared__ int red_a[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float m = 0.f;
  int am = 0;
  for (int row = blockIdx.x * 4 + wave; row < n_rows; row += gridDim.x * 4) {
    const float4 v = reinterpret_cast<const float4*>(db + (size_t)row * kD)[lane];
    float s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    int a = max(max(__float_as_int(v.x) & 0x7fffffff, __float_as_int(v.y) & 0x7fffffff),
                max(__float_as_int(v.z) & 0x7fffffff, __float_as_int(v.w) & 0x7fffffff));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      s += __shfl_xor(s, off);
      a = max(a, __shfl_xor(a, off));
    }
    m = fmaxf(m, s);
    am = max(am, a);
  }
  if (lane == 0) {
    red[wave] = m;
    red_a[wave] = am;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    atomicMax(reinterpret_cast<int*>(out), __float_as_int(sqrtf(m) * 1.0001f));  // non-negative floats order as ints
    atomicMax(reinterpret_cast<int*>(out) + 1, max(max(red_a[0], red_a[1]), max(red_a[2], red_a[3])));
  }
}

// out[0 .. 255] += sum over up to 1,024 sampled rows (every stride-th) of row / |row|: t2l_db_set turns |sum|^2 into the sample's mean
// pairwise cosine (the prior for the first searches on a new database, capi.hip). 16 workgroups x 64 rows, one wave per row.
__global__ __launch_bounds__(256) void db_cluster_kernel(const float* __restrict__ db, int n_rows, int stride, int n_sample, float* out) {
  __shared__ float4 acc[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = blockIdx.x * 64 + wave; i < min(n_sample, (int)(blockIdx.x + 1) * 64); i += 4) {
    const float4 v = reinterpret_cast<const float4*>(db + (size_t)min(i * stride, n_rows - 1) * kD)[lane];
    float s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    const float inv = s > 0.f ? 1.0f / sqrtf(s) : 0.f;
    a.x += v.x * inv; a.y += v.y * inv; a.z += v.z * inv; a.w += v.w * inv;
  }
  acc[wave][lane] = a;
  __syncthreads();
  if (wave == 0) {
    const float4 b = acc[1][lane], c = acc[2][lane], d = acc[3][lane];
    atomicAdd(out + 4 * lane + 0, a.x + b.x + c.x + d.x);
    atomicAdd(out + 4 * lane + 1, a.y + b.y + c.y + d.y);
    atomicAdd(out + 4 * lane + 2, a.z + b.z + c.z + d.z);
    atomicAdd(out + 4 * lane + 3, a.w + b.w + c.w + d.w);
  }
}

static int grow(t2l_ctx* ctx, void** p, size_t* cap, size_t need_bytes) {
  if (need_bytes <= *cap) return T2L_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  T2L_HIP(ctx, hipMalloc(p, need_bytes));
  *cap = need_bytes;
  return T2L_OK;
}

int db_norm_impl(t2l_ctx* ctx, hipStream_t s) {
  const int threads = (int)ctx->db_pad * 32;
  if (ctx->db_pad > 0) {  // the bf16 hi/lo planes (split-bf16 scan, streaming scan)
    hipLaunchKernelGGL(split_db_kernel, dim3((threads + 255) / 256), dim3(256), 0, s, ctx->db, ctx->db_split,
                       (int)ctx->db_pad);
    T2L_HIP(ctx, hipGetLastError());
  }
  T2L_HIP(ctx, hipMemsetAsync(ctx->db_norm_max, 0, (2 + 256) * sizeof(float), s));
  if (ctx->db_rows > 0) {
    const int blocks = (int)min((int64_t)1024, (ctx->db_rows + 3) / 4);
    hipLaunchKernelGGL(db_norm_kernel, dim3(blocks), dim3(256), 0, s, ctx->db, (int)ctx->db_rows, ctx->db_norm_max);
    const int n_sample = (int)min((int64_t)1024, ctx->db_rows), stride = (int)(ctx->db_rows / n_sample);
    hipLaunchKernelGGL(db_cluster_kernel, dim3((n_sample + 63) / 64), dim3(256), 0, s, ctx->db, (int)ctx->db_rows, stride, n_sample,
                       ctx->db_norm_max + 2);
    T2L_HIP(ctx, hipGetLastError());
  }
  if (ctx->db_pad > 0) {  // the scaled f16 plane of the default scan (needs the max |element| from above)
    hipLaunchKernelGGL(half_db_kernel, dim3((threads + 255) / 256), dim3(256), 0, s, ctx->db, ctx->db_norm_max, ctx->db_half,
                       (int)ctx->db_pad, (int)ctx->db_rows);
    T2L_HIP(ctx, hipGetLastError());
  }
  return T2L_OK;
}


template <typename Kern>
static void allow_lds(Kern* kern, size_t lds) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

// scan (by search_mode) -> re-rank -> fallback on one stream. LL = per-lane list length the scan keeps, L = rows the
// re-rank re-scores per query.
template <int LL, int L>
static int launch_search(t2l_ctx* ctx, const float* db, const uint4* dbs, const uint4* dbh, int n_rows, int row_offset,
                         const float* q, int Q, int K, int nsplit, int code_bits, int32_t* out_idx, double* out_score,
                         bool first, bool pair, hipStream_t s) {
  const int n_tiles = (n_rows + kTileRows - 1) / kTileRows;
  const int parts = 2 * nsplit;
  const int zero = first;
  const int half_mode = ctx->eff_mode == 0;
  const bool probing = ctx->search_mode == 0 && ctx->eff_mode == 2;  // standing in for the f16 scan (search_impl)
  // f32 dot-product error bound: gamma_n * |a||b| with n = 256 terms (+ slack for the MFMA's k order), plus the operand
  // rounding of the scan that produced the keys: f16 (RNE, both operands) 2^-10 + 2^-21 and 2^-20 for denormal
  // elements, rounded up to 9.85e-4; split-bf16 2^-16 + 2^-18, rounded up to 2e-5; f32: none
  const double operand_eps = ctx->eff_mode == 0 ? 9.85e-4 : (ctx->eff_mode == 2 ? 2.0e-5 : 0.0);
  const float eps_rel = (float)(ctx->eps_scale * ((kD + 8) * 5.9604644775390625e-08 + operand_eps));
  const float eps_probe = probing ? (float)(ctx->eps_scale * ((kD + 8) * 5.9604644775390625e-08 + 9.85e-4)) : 0.f;
  const int stat_mode = probing ? 2 : (half_mode ? 1 : 0);
  const int seq = first ? ++ctx->stat_seq : 0;  // the report card goes out once per call
  const int defer = ctx->heavy ? 1 : 0;
  bool merged = false;  // the candidate lists are merged records (scanp_kernel<..., MERGE>)
  if constexpr (LL <= 6) {  // paired f16 MFMA scan (default): one 512-thread workgroup per CU, 256 queries each; `nsplit`
    // counts VIRTUAL splits here: the kernel takes physical ones (2 virtual splits per workgroup)
    const dim3 grid((Q + kWideQPerBlock - 1) / kWideQPerBlock * (nsplit / 2));
    // (the tile ring + the merged records' exchange area: 256 records of 11 floats)
    const size_t lds = (size_t)4 * 2 * kHalfTileBytes + (size_t)256 * (kMergedLL + 3) * sizeof(float);
    static PerDeviceOnce once;
    if (once.need(ctx->device)) {
      allow_lds(&scanp_kernel<LL, 4>, lds);
      if constexpr (LL == 6) allow_lds(&scanp_kernel<LL, 4, true>, lds);
      if constexpr (LL == 6) allow_lds(&scanp_kernel<LL, 4, true, 1>, lds);
      once.mark(ctx->device);
    }
    const unsigned span_seq = ++ctx->span_seq;
    unsigned long long* span = ctx->scan_span && grid.x <= (unsigned)kSpanWgs ? ctx->scan_span + (size_t)2 * kSpanWgs * (span_seq % kSpanRing) : nullptr;
    if (ctx->span_grid) ctx->span_grid[span_seq % kSpanRing] = span ? grid.x : 0;
    // the XCD rectangle needs whole query-block groups and split groups on every XCD (else: splits only, as before)
    int xq = ctx->xcd_qgroups;
    {
      const int nqb = (Q + kWideQPerBlock - 1) / kWideQPerBlock, ns = nsplit / 2;
      if (xq < 2 || 8 % xq || nqb % xq || ns % (8 / xq) || (nqb * ns) % 8) xq = 1;
    }
    hipEvent_t ea, eb;
    const bool ev = event_pair(ctx, "search_scan", &ea, &eb);  // sampled launch: the dispatch carries its own start / stop events
    auto launch = [&](auto kern) {
      if (ev)
        hipExtLaunchKernelGGL(kern, grid, dim3(512), (uint32_t)lds, s, ea, eb, 0u, dbh, n_rows, n_tiles, code_bits, q, Q, nsplit / 2, ctx->cand_score,
                              ctx->fb_count, ctx->fb_prev, zero, __builtin_inff(), span, span_seq, xq);
      else
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, dbh, n_rows, n_tiles, code_bits, q, Q, nsplit / 2, ctx->cand_score, ctx->fb_count,
                           ctx->fb_prev, zero, __builtin_inff(), span, span_seq, xq);
    };
    if constexpr (LL == 6 && L == 16) {
      // merged records (option "search_merge_lists"): the workgroup's four lists per query leave as one 32-byte record
      // (two more code bits come out of the key's score: kept to shards whose keys still hold 12 score bits below the exponent)
      merged = (ctx->search_merge == 1 || (ctx->search_merge == 2 && ctx->merge_live && !ctx->heavy)) && code_bits <= 9;
    }
    if constexpr (LL == 6) {
      if (merged && ctx->search_tile_sel) launch(scanp_kernel<LL, 4, true, 1>);
      else if (merged) launch(scanp_kernel<LL, 4, true>);
    }
    if (!merged) launch(scanp_kernel<LL, 4>);
  } else {
  event_begin(ctx, "search_scan", s);
  if (ctx->eff_mode == 0) {  // f16 MFMA scan, one wave per SIMD (tiny shards, k > 10): 256 queries per workgroup
    const dim3 grid((Q + kWideQPerBlock - 1) / kWideQPerBlock * nsplit);
    const size_t lds = (size_t)4 * kHalfTileBytes;
    static PerDeviceOnce once;
    if (once.need(ctx->device)) {
      allow_lds(&scanh_kernel<LL>, (size_t)4 * kHalfTileBytes);
      once.mark(ctx->device);
    }
    hipLaunchKernelGGL((scanh_kernel<LL>), grid, dim3(256), lds, s, dbh, n_rows, n_tiles, code_bits, q, Q, nsplit,
                       ctx->cand_score, ctx->fb_count, ctx->fb_prev, zero, __builtin_inff());
  } else if (ctx->eff_mode == 2) {  // split-bf16 MFMA scan, same structure, three MFMAs per product
    const dim3 grid((Q + kWideQPerBlock - 1) / kWideQPerBlock * nsplit);
    const size_t lds = (size_t)4 * kTileFloats * sizeof(float);
    static PerDeviceOnce once;
    if (once.need(ctx->device)) {
      allow_lds(&scanw_kernel<LL, 4>, (size_t)4 * kTileFloats * sizeof(float));
      once.mark(ctx->device);
    }
    hipLaunchKernelGGL((scanw_kernel<LL, 4>), grid, dim3(256), lds, s, dbs, n_rows, n_tiles, code_bits, q, Q, nsplit,
                       ctx->cand_score, ctx->fb_count, ctx->fb_prev, zero, __builtin_inff());
  }
  event_end(ctx, "search_scan", s);
  }
  T2L_HIP(ctx, hipGetLastError());
  // Two launches per search. Queries the certificate and the in-wave re-score leave unsettled are ranked exactly by
  // their own re-rank workgroup (wg_exact_scan) — there is no separate fallback launch to pay for when, as usual, there
  // are none. (Heavy mode: they are deferred to the float64 MFMA stage instead, search_exact.hip.)
  hipEvent_t ea, eb;
  if constexpr (LL == 6 && L == 16) {
    if (merged) {
      const bool evr = ctx->profile_rerank && event_pair(ctx, "search_rerank", &ea, &eb);
      if (evr)
        hipExtLaunchKernelGGL((rerank_kernel<kMergedLL, L, true>), dim3((Q + 3) / 4), dim3(256), 0u, s, ea, eb, 0u, db, q, Q, K, nsplit / 2,
                              code_bits, (const float*)ctx->cand_score, row_offset, eps_rel, (const float*)ctx->db_norm_max, half_mode, out_idx,
                              out_score, ctx->flags, ctx->fb_count, eps_probe, __builtin_inff(), n_rows, defer, stat_mode,
                              ctx->host_stat_dev, seq, min(ctx->wide_repair, kWideCap), ctx->search_tile_sel ? 1 : 0);
      else
        hipLaunchKernelGGL((rerank_kernel<kMergedLL, L, true>), dim3((Q + 3) / 4), dim3(256), 0, s, db, q, Q, K, nsplit / 2, code_bits,
                           ctx->cand_score, row_offset, eps_rel, ctx->db_norm_max, half_mode, out_idx, out_score, ctx->flags,
                           ctx->fb_count, eps_probe, __builtin_inff(), n_rows, defer, stat_mode, ctx->host_stat_dev, seq,
                           min(ctx->wide_repair, kWideCap), ctx->search_tile_sel ? 1 : 0);
      T2L_HIP(ctx, hipGetLastError());
      if (ctx->heavy) return exact_stage_impl(ctx, db, n_rows, row_offset, q, Q, K, out_idx, out_score, s);
      return T2L_OK;
    }
  }
  if (ctx->profile_rerank && event_pair(ctx, "search_rerank", &ea, &eb))
    hipExtLaunchKernelGGL((rerank_kernel<LL, L>), dim3((Q + 3) / 4), dim3(256), 0u, s, ea, eb, 0u, db, q, Q, K, parts, code_bits,
                          (const float*)ctx->cand_score, row_offset, eps_rel, (const float*)ctx->db_norm_max, half_mode, out_idx,
                          out_score, ctx->flags, ctx->fb_count, eps_probe, __builtin_inff(), n_rows, defer, stat_mode,
                          ctx->host_stat_dev, seq, min(ctx->wide_repair, kWideCap), 0);
  else
    hipLaunchKernelGGL((rerank_kernel<LL, L>), dim3((Q + 3) / 4), dim3(256), 0, s, db, q, Q, K, parts, code_bits,
                       ctx->cand_score, row_offset, eps_rel, ctx->db_norm_max, half_mode, out_idx, out_score, ctx->flags,
                       ctx->fb_count, eps_probe, __builtin_inff(), n_rows, defer, stat_mode, ctx->host_stat_dev, seq,
                       min(ctx->wide_repair, kWideCap), 0);
  T2L_HIP(ctx, hipGetLastError());
  if (ctx->heavy) return exact_stage_impl(ctx, db, n_rows, row_offset, q, Q, K, out_idx, out_score, s);
  return T2L_OK;
}

// (rows one scan launch can cover — kSegmentRows = 32 splits x 512 tiles (13 code bits) x 32 rows — t2l_internal.h)

// an empty shard (legal under ragged row-sharding: shard_bounds hands the tail ranks nothing) answers -1 / -inf everywhere
__global__ __launch_bounds__(256) void empty_result_kernel(int n, int32_t* __restrict__ out_idx, double* __restrict__ out_score) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    out_idx[i] = -1;
    if (out_score) out_score[i] = -__builtin_inf();
  }
}

// all-exact mode (search_impl): no candidate scan, no re-rank — this launch does their bookkeeping (counters parked and cleared,
// the previous call's report card published, as reset_counts + rerank_kernel's first thread do) and puts every query on the exact
// stage's list.
__global__ __launch_bounds__(256) void all_exact_prep_kernel(int Q, int32_t* __restrict__ list, int32_t* __restrict__ fb_count,
                                                             int32_t* __restrict__ fb_prev, int32_t* __restrict__ host_stat, int seq) {
  if (blockIdx.x == 0) {
    reset_counts(fb_count, fb_prev, 1, threadIdx.x);
    __syncthreads();
    if (threadIdx.x == 0) {
      if (host_stat && seq > 0) {
        host_stat[1] = fb_count[64 + 10] == 2 ? fb_count[64 + 3] : fb_count[64 + 2];
        host_stat[2] = fb_count[64 + 9];
        host_stat[3] = fb_count[64 + 0] + fb_count[64 + 4] + fb_count[64 + 12];
        host_stat[4] = fb_count[64 + 10];
        host_stat[5] = fb_count[64 + 10] == 2 ? fb_count[64 + 3] : fb_count[64 + 1];
        __hip_atomic_store(&host_stat[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      fb_count[9] = Q;
      fb_count[10] = 0;
      fb_count[1] = fb_count[2] = fb_count[4] = Q;  // every query counts as flagged and deferred
    }
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < Q; i += gridDim.x * 256) list[i] = i;
}

int search_impl(t2l_ctx* ctx, const float* q, int Q, int K, int32_t* out_idx, double* out_score, hipStream_t s) {
  if (Q == 0) return T2L_OK;
  const int n_rows = (int)ctx->db_rows;
  if (n_rows <= 0) {  // no scan: the kernels below assume at least one tile per split
    hipLaunchKernelGGL(empty_result_kernel, dim3((Q * K + 255) / 256), dim3(256), 0, s, Q * K, out_idx, out_score);
    T2L_HIP(ctx, hipGetLastError());
    return T2L_OK;
  }
  ctx->last_search_small = false;
  // few queries against a large shard: stream the DB once through every CU (search_stream.hip)
  if (Q <= 64 && n_rows >= ctx->stream_min_rows && n_rows > 0 && ctx->search_mode == 0 && ctx->nsplit_override == 0)
    return search_stream_impl(ctx, q, Q, K, out_idx, out_score, s);
  // a handful of queries against a shard that fits the caches: ONE launch, exact float64 from the start (search_small.hip)
  if (search_small_applies(ctx, Q, K)) return search_small_impl(ctx, q, Q, K, out_idx, out_score, s);
  // the f16 scan's report card of an earlier call on this DB (see t2l_internal.h): more than 1 in 8 queries flagged ->
  // the split-bf16 scan from now on
  // ... and back when fewer than 1 in 16 would be (the stand-in counts them, rerank_kernel)
  // ... and its exact-stage count: when more than 1 in 64 queries of a call ended in the float64 scan, the database
  // defeats the certificates wholesale ("heavy"): later calls defer those queries to the float64 MFMA stage
  // (search_exact.hip) instead of the fallback kernel's VALU scan, until fewer than 1 in 256 need it
  if (ctx->host_stat) {
    const volatile int32_t* hs = ctx->host_stat;
    const int done = __atomic_load_n(ctx->host_stat, __ATOMIC_ACQUIRE);  // pairs with the kernel's system-scope release store
    if (done > ctx->stat_seen) {
      ctx->stat_seen = done;
      const int64_t flagged = hs[1], total = hs[2], exact_prev = hs[3], rescored = hs[5];
      if (ctx->search_mode == 0 && ctx->search_auto && hs[4]) {
        // ... or more than 1 in 2 failed the first certificate: the in-wave repairs settle them, but a wide repair re-scores
        // dozens to hundreds of rows per query — measured on a clustered database with 92 % of the queries repaired: 185 us per
        // step on the f16 scan against 133 us on the split-bf16 scan, whose 50x tighter band certifies them outright
        // (a report is two calls old: only a report of the f16 scan escalates, only one of the stand-in's probe releases —
        // an f16 report that arrives after the switch must not undo it)
        if (hs[4] == 1 && !ctx->escalated && (flagged * 8 > total || rescored * 2 > total)) ctx->escalated = true;
        else if (hs[4] == 2 && ctx->escalated && flagged * 16 < total) ctx->escalated = false;
      }
      // merged candidate records pay while repairs are rare: more than 1 query in 64 failing its first certificate -> plain lists
      // (their repairs re-score a quarter of the rows), back below 1 in 256
      if (hs[4] == 1) {
        if (ctx->merge_live && rescored * 64 > total) ctx->merge_live = false;
        else if (!ctx->merge_live && rescored * 256 <= total) ctx->merge_live = true;
      }
      if (ctx->search_auto) {
        if (!ctx->heavy && exact_prev * 64 > total) ctx->heavy = true;
        else if (ctx->heavy && exact_prev * 256 < total) ctx->heavy = false;
        // 7 in 8 queries end in the exact stage whatever the candidate scan says: stop paying for the scan and the re-rank
        // (0.26 ms of a 1.0 ms step on such a database) and hand EVERY query to the float64 MFMA stage; one call in 8 still
        // takes the long way and its report decides whether that remains true
        // (only on the word of the split-bf16 stand-in, whose band is the tightest a candidate scan has: a database that defeats
        // the f16 scan alone gets the stand-in first; reports of all-exact calls themselves carry no scan and change nothing)
        if (hs[4] == 2) ctx->all_exact = ctx->heavy && exact_prev * 8 >= total * 7;
        else if (hs[4] == 1) ctx->all_exact = false;
      }
    }
  }
  ctx->eff_mode = (ctx->search_mode == 0 && ctx->escalated) ? 2 : ctx->search_mode;
  const int qpb = kWideQPerBlock;
  const int n_qblocks = (Q + qpb - 1) / qpb;
  // rows re-scored per query: K + margin (the margin only has to absorb key-truncation ties; the certificate catches
  // the rest). (L = 12 was measured: second-stage re-scores multiply.)
  const int L = (K <= 10) ? 16 : 32;
  const int n_seg = max(1, (n_rows + kSegmentRows - 1) / kSegmentRows);
  if (n_seg * K > 256)
    return fail(ctx, T2L_EINVAL, "t2l_search: shard too large (more than 256/k segments of 524,288 rows); shard the "
                                 "database over more ranks");
  int rc;
  // flags[Q] + f32 thresholds[Q] + flagged list[Q] + deferred list[Q] + uncertified list[Q] (search_exact.hip)
  if ((rc = grow(ctx, (void**)&ctx->flags, &ctx->flag_cap, (size_t)5 * Q * sizeof(int32_t))) != T2L_OK) return rc;
  std::swap(ctx->fb_count, ctx->fb_prev);  // this call counts in the bank the previous call cleared (reset_counts)
  ctx->last_search_small = false;
  if (ctx->search_auto && ctx->heavy && ctx->all_exact && n_seg == 1 && (ctx->all_exact_calls++ & 7) != 7) {
    const int seq = ++ctx->stat_seq;
    // (block 0 parks the counters before any block's exactd successor reads them: the launches are stream-ordered)
    hipLaunchKernelGGL(all_exact_prep_kernel, dim3(min((Q + 255) / 256, 64)), dim3(256), 0, s, Q, ctx->flags + (size_t)3 * Q, ctx->fb_count,
                       ctx->fb_prev, ctx->host_stat_dev, seq);
    T2L_HIP(ctx, hipGetLastError());
    return exact_stage_impl(ctx, ctx->db, n_rows, (int)ctx->row_offset, q, Q, K, out_idx, out_score, s);
  }
  int32_t* seg_idx = out_idx;
  double* seg_score = out_score;
  if (n_seg > 1) {
    if ((rc = grow(ctx, (void**)&ctx->seg_idx, &ctx->seg_idx_cap, (size_t)n_seg * Q * K * sizeof(int32_t))) != T2L_OK ||
        (rc = grow(ctx, (void**)&ctx->seg_score, &ctx->seg_score_cap, (size_t)n_seg * Q * K * sizeof(double))) != T2L_OK)
      return rc;
  }
  for (int seg = 0; seg < n_seg; ++seg) {
    const int row0 = seg * kSegmentRows;
    const int rows = min(kSegmentRows, n_rows - row0);
    const int n_tiles = (max(rows, 0) + kTileRows - 1) / kTileRows;
    // the paired scan (two waves per SIMD, scanp_kernel) serves the f16 mode whenever the shard gives every query at least
    // 32 per-lane lists (>= 8 physical splits: 256+ rows); its splits below are VIRTUAL ones (two per workgroup)
    // (small batches keep the one-wave-per-SIMD kernel: twice the workgroups, and its prologue is the shorter one)
    const bool pair_ok = ctx->eff_mode == 0 && L == 16 && n_tiles >= 16 && Q >= 256;
    int nsplit = ctx->nsplit_override;
    if (nsplit <= 0) {
      // fill 256 CUs with one (wide scan) or two workgroups each; multiples of 8 keep a split on one XCD's L2
      nsplit = (256 + n_qblocks - 1) / n_qblocks;
      nsplit = ((nsplit + 7) / 8) * 8;
    }
    nsplit = max(1, min(nsplit, kMaxParts / 2));
    nsplit = max(1, min(nsplit, max(1, n_tiles)));
    nsplit = max(nsplit, (n_tiles + kMaxPerTiles - 1) / kMaxPerTiles);  // keep the key code within 13 bits
    bool pair = pair_ok;
    if (pair) {  // physical splits = workgroups per query block (<= 16), virtual = twice that
      int phys = ctx->nsplit_override > 0 ? max(1, ctx->nsplit_override / 2) : min(16, nsplit);
      phys = max(phys, (n_tiles + 2 * kMaxPerTiles - 1) / (2 * kMaxPerTiles));
      phys = min(phys, 16);
      if (2 * phys > n_tiles || 4 * phys < 32) pair = false;
      else nsplit = 2 * phys;
    }
    const int per = max(1, (n_tiles + nsplit - 1) / nsplit);
    int code_bits = 4;
    while ((1 << code_bits) < per * 16) ++code_bits;
    // per-lane list length of the wide scan: the global top-L spreads over 2*nsplit lists (tiles are dealt round-robin
    // to the splits, 4-row groups alternate between the lane halves), so 8 per list hold it unless more than 8 of a
    // query's best 16 fall into ONE list — with >= 16 lists a ~1e-8 event on unstructured data; the certificate
    // (floors of full lists) catches it and the fallback re-scores. Few lists (tiny shards): keep 16.
    const int LL = pair ? ctx->pair_ll : (L == 32 ? 32 : (2 * nsplit >= 16 ? 8 : 16));
    const size_t need = (size_t)n_qblocks * qpb * 2 * nsplit * LL * sizeof(float);
    if ((rc = grow(ctx, (void**)&ctx->cand_score, &ctx->cand_cap, need)) != T2L_OK) return rc;
    if (n_seg > 1) {
      seg_idx = ctx->seg_idx + (size_t)seg * Q * K;
      seg_score = ctx->seg_score + (size_t)seg * Q * K;
    }
    const float* db = ctx->db + (size_t)row0 * kD;
    const uint4* dbs = ctx->db_split ? ctx->db_split + (size_t)row0 * 64 : nullptr;
    const uint4* dbh = ctx->db_half ? ctx->db_half + (size_t)row0 * 32 : nullptr;
    const int off = (int)ctx->row_offset + row0;
#define T2L_SEARCH(LLv, Lv) \
  launch_search<LLv, Lv>(ctx, db, dbs, dbh, max(rows, 0), off, q, Q, K, nsplit, code_bits, seg_idx, seg_score, seg == 0, pair, s)
    rc = LL == 5 ? T2L_SEARCH(5, 16) : LL == 6 ? T2L_SEARCH(6, 16) : (LL == 8 ? T2L_SEARCH(8, 16) : (L == 16 ? T2L_SEARCH(16, 16) : T2L_SEARCH(32, 32)));
#undef T2L_SEARCH
    if (rc != T2L_OK) return rc;
  }
  if (n_seg > 1) return merge_impl(ctx, ctx->seg_idx, ctx->seg_score, n_seg, Q, K, out_idx, out_score, s);
  return T2L_OK;
}

// ------------------------------------------------------------------------------------------------
// lanes: pipelined independent searches (t2l_internal.h). Measured alternatives: all scans on one stream and the re-ranks on a
// second one needs two cross-stream events per call on the scan stream — 59 us per call instead of 42 (a dependent event
// hop costs more than the re-rank it hides); self-contained chains need none.
// ------------------------------------------------------------------------------------------------
static void swap_lane(t2l_ctx* ctx, t2l_ctx::SearchLane& L) {
  std::swap(ctx->cand_score, L.cand_score);
  std::swap(ctx->cand_cap, L.cand_cap);
  std::swap(ctx->flags, L.flags);
  std::swap(ctx->flag_cap, L.flag_cap);
  std::swap(ctx->fb_count, L.fb_count);
  std::swap(ctx->fb_prev, L.fb_prev);
  std::swap(ctx->host_stat, L.host_stat);
  std::swap(ctx->host_stat_dev, L.host_stat_dev);
  std::swap(ctx->stat_seen, L.stat_seen);
}

int search_join_impl(t2l_ctx* ctx, hipStream_t s) {
  for (auto& L : ctx->lanes) {
    if (!L.pending) continue;
    T2L_HIP(ctx, hipEventRecord(L.done, L.stream));
    T2L_HIP(ctx, hipStreamWaitEvent(s, L.done, 0));
    L.pending = false;
  }
  return T2L_OK;
}

void free_lanes(t2l_ctx* ctx) {
  for (auto& L : ctx->lanes) {
    for (void* p : {(void*)L.cand_score, (void*)L.flags, (void*)(L.fb_count < L.fb_prev ? L.fb_count : L.fb_prev)})
      if (p) (void)hipFree(p);
    if (L.host_stat) (void)hipHostFree(L.host_stat);
    if (L.stream) (void)hipStreamDestroy(L.stream);
    if (L.done) (void)hipEventDestroy(L.done);
    L = t2l_ctx::SearchLane();
  }
  if (ctx->lane_fork) (void)hipEventDestroy(ctx->lane_fork);
  ctx->lane_fork = nullptr;
}

int search_lanes_impl(t2l_ctx* ctx, const float* q, int Q, int K, int32_t* out_idx, double* out_score, hipStream_t s) {
  // one launch pair per call is what pipelines; everything else (small batches on the streaming scan, multi-segment shards,
  // databases in heavy mode with their extra stages, empty shards) runs in the caller's stream after a join
  const bool plain = ctx->n_lanes <= 1 || Q < 256 || ctx->heavy || ctx->db_rows <= 0 || (int)ctx->db_rows > kSegmentRows;
  if (plain) {
    int rc = search_join_impl(ctx, s);
    return rc != T2L_OK ? rc : search_impl(ctx, q, Q, K, out_idx, out_score, s);
  }
  t2l_ctx::SearchLane& L = ctx->lanes[ctx->lane_next++ % (unsigned)ctx->n_lanes];
  if (!L.ready) {
    T2L_HIP(ctx, hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
    T2L_HIP(ctx, hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
    if (!ctx->lane_fork) T2L_HIP(ctx, hipEventCreateWithFlags(&ctx->lane_fork, hipEventDisableTiming));
    T2L_HIP(ctx, hipMalloc(&L.fb_count, 256 * sizeof(int32_t)));  // two banks (reset_counts)
    T2L_HIP(ctx, hipMemset(L.fb_count, 0, 256 * sizeof(int32_t)));
    L.fb_prev = L.fb_count + 128;
    if (hipHostMalloc((void**)&L.host_stat, 8 * sizeof(int32_t), hipHostMallocMapped) == hipSuccess) {
      for (int i = 0; i < 8; ++i) L.host_stat[i] = 0;
      if (hipHostGetDevicePointer((void**)&L.host_stat_dev, L.host_stat, 0) != hipSuccess) L.host_stat_dev = nullptr;
    }
    L.ready = true;
  }
  // the queries (and whatever else the caller queued) are ordered before the lane's chain
  T2L_HIP(ctx, hipEventRecord(ctx->lane_fork, s));
  T2L_HIP(ctx, hipStreamWaitEvent(L.stream, ctx->lane_fork, 0));
  swap_lane(ctx, L);
  const int rc = search_impl(ctx, q, Q, K, out_idx, out_score, L.stream);
  swap_lane(ctx, L);
  L.pending = true;
  return rc;
}

}  // namespace t2l
