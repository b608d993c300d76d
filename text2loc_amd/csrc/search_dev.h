// Device-side building blocks shared by the search kernels (search.hip, search_stream.hip): key lists, split-bf16
// helpers, the MFMA+selection tile body, wave reductions, the float64 exact scan. gfx950 only.
#pragma once
#include <float.h>
#include <limits.h>

#include "t2l_internal.h"

namespace t2l {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define T2L_NEG_INF (-__builtin_inff())

// Sorted (descending) register list of keys. Element I of the NEW list depends only on OLD values:
// s'[I] = med3(s[I-1], s[I], x) for a descending s, s'[0] = max(s[0], x). Updating I = L-1 .. 0 in place
// therefore needs no temporaries and no compares.
template <int L, int I = L - 1>
__device__ __forceinline__ void ins_key(float (&s)[L], float x) {
  if constexpr (I == 0) {
    s[0] = __builtin_amdgcn_fmed3f(s[0], x, __builtin_inff());  // max without a canonicalising extra op
  } else {
    s[I] = __builtin_amdgcn_fmed3f(s[I - 1], s[I], x);
    ins_key<L, I - 1>(s, x);
  }
}

__device__ __forceinline__ float make_key(float v, int mask, int code) {
  return __int_as_float((__float_as_int(v) & mask) | code);
}


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// x -> (hi, lo) bf16 pair with hi = RNE_bf16(x), lo = RNE_bf16(x - hi), two values per call packed low|high:
// v_cvt_pk_bf16_f32 rounds both lanes in one instruction (5 VALU per pair instead of ~26 for integer rounding — the
// query split in a scan's prologue is 64 x 8 values per lane and was 15 % of the whole kernel).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
  const float h0 = __uint_as_float(hi << 16), h1 = __uint_as_float(hi & 0xFFFF0000u);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0 - h0, x1 - h1}, bf16x2));
}
__device__ __forceinline__ void split8(const float4 a, const float4 b, uint4& hi, uint4& lo) {
  split_pair(a.x, a.y, hi.x, lo.x);
  split_pair(a.z, a.w, hi.y, lo.y);
  split_pair(b.x, b.y, hi.z, lo.z);
  split_pair(b.z, b.w, hi.w, lo.w);
}


template <int L, int I = L - 1>
__device__ __forceinline__ void ins_key_sat(float (&s)[L], float x, float pinf) {
  if constexpr (I == 0) {
    s[0] = __builtin_amdgcn_fmed3f(s[0], x, pinf);  // max(s0, x) as ONE op (pinf is a run-time +inf: no folding to
  } else {                                          // the canonicalising v_max pair)
    s[I] = __builtin_amdgcn_fmed3f(s[I - 1], s[I], x);
    ins_key_sat<L, I - 1>(s, x, pinf);
  }
}

// ---- LDS-DMA: global_load_lds_dwordx4, one wave-instruction moves 64 x 16 B = one 1 KiB DB row into LDS (M0 = the
// row's LDS address, the hardware adds lane*16). Issued from inline asm ON PURPOSE: for the compiler's builtin,
// SIInsertWaitcnts makes every later LDS read wait for vmcnt(0) ("may alias the DMA's LDS write"), i.e. the wave sits out
// the whole L2 round trip of the NEXT tile before it touches the CURRENT one and the double buffer buys nothing
// (visible in the ISA as `s_waitcnt vmcnt(0)` between the DMA burst and the first ds_read). The kernels order DMA
// against LDS reads themselves: explicit s_waitcnt vmcnt(n) + s_barrier before a buffer is read.
// `row_base` must be wave-uniform (SGPR pair).
__device__ __forceinline__ void lds_dma_row(unsigned lds_row_addr, unsigned lane16, const void* row_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_row_addr), "v"(lane16), "s"(row_base)
               : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ int uniform_wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// ---- the wide tile body (scanw_kernel): a wave holds TWO 32-query fragments, so every DB fragment read from LDS
// feeds 6 MFMAs (hi*hi, hi*lo, lo*hi for each query group) on two independent accumulator chains, and per k-step
// one score of the previous tile of EACH group enters that group's list: 2*(1 + LL) VALU ops in the shadows of 6 MFMAs.
//
// Register plan (one wave per SIMD = the unified 512-entry file): the 64 query fragments fill ALL 256 AGPRs and are
// read by the MFMAs in place; accumulators, key lists and the LDS fragment ring live in VGPRs where the VALU can
// reach them. The compiler does not find that split by itself (it parks fragments in AGPRs and copies them back with
// v_accvgpr_read before every use, and its sched_group_barrier solver then gives up and emits all MFMAs followed by
// all insertions), so the MFMAs are inline asm with explicit register classes and the instruction order is written
// out by hand: asm volatile keeps MFMAs / DMA / waits in program order, sched_barrier(0) pins the VALU and LDS
// instructions between them. Inline-asm MFMAs are invisible to the hazard recogniser; by construction a score is
// read by the VALU no earlier than two MFMA issues (64+ cycles) after the MFMA that wrote it, and accumulators
// start from the constant-0 srcC form instead of being zeroed by the VALU.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mfma_bf16_first(f32x16& acc, const u32x4& a, const u32x4& b_agpr) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b_agpr));
}
__device__ __forceinline__ void mfma_bf16_acc(f32x16& acc, const u32x4& a, const u32x4& b_agpr) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b_agpr));
}
__device__ __forceinline__ u32x4 pin_agpr(u32x4 v) {  // from here on the value lives in an AGPR tuple
  u32x4 r;
  asm volatile("" : "=a"(r) : "0"(v));
  return r;
}
__device__ __forceinline__ u32x4 as_u32x4(const uint4 v) { return u32x4{v.x, v.y, v.z, v.w}; }

struct WideDma {
  const char* src;  // wave-uniform global address of row 8*wave of the tile
  unsigned dst;     // LDS byte address of row 8*wave of the target buffer
  unsigned lane16;  // lane * 16
  __device__ __forceinline__ void row(int i) const { lds_dma_row(dst + i * (kRowStrideF * 4), lane16, src + i * 1024); }
};

template <int LL>
struct WideLists {
  float ls0[LL], ls1[LL];  // sorted key lists of the two query groups
  float key0, key1;        // keys in flight
};

// VALU op O (of 2*LL + 2) of inserting score S of both groups: 0 = key0, 1..LL = list 0 from its tail to its head,
// LL+1 = key1, LL+2..2LL+1 = list 1. Every list element reads OLD neighbours only, so the order tail -> head is in place.
template <int LL, int S, int O>
__device__ __forceinline__ void wide_sel_op(WideLists<LL>& w, const f32x16& prev0, const f32x16& prev1, int vmask, int code,
                                            float pinf) {
  if constexpr (O == 0) {
    w.key0 = __int_as_float((__float_as_int(prev0[S]) & vmask) | code);
  } else if constexpr (O <= LL) {
    constexpr int I = LL - O;
    if constexpr (I == 0) w.ls0[0] = __builtin_amdgcn_fmed3f(w.ls0[0], w.key0, pinf);
    else w.ls0[I] = __builtin_amdgcn_fmed3f(w.ls0[I - 1], w.ls0[I], w.key0);
  } else if constexpr (O == LL + 1) {
    w.key1 = __int_as_float((__float_as_int(prev1[S]) & vmask) | code);
  } else if constexpr (O <= 2 * LL + 1) {
    constexpr int I = 2 * LL + 1 - O;
    if constexpr (I == 0) w.ls1[0] = __builtin_amdgcn_fmed3f(w.ls1[0], w.key1, pinf);
    else w.ls1[I] = __builtin_amdgcn_fmed3f(w.ls1[I - 1], w.ls1[I], w.key1);
  }
}
template <int LL, int S, int O, int O_END>
__device__ __forceinline__ void wide_sel_ops(WideLists<LL>& w, const f32x16& prev0, const f32x16& prev1, int vmask, int code,
                                             float pinf) {
  if constexpr (O < O_END && O < 2 * LL + 2) {
    wide_sel_op<LL, S, O>(w, prev0, prev1, vmask, code, pinf);
    wide_sel_ops<LL, S, O + 1, O_END>(w, prev0, prev1, vmask, code, pinf);
  }
}

// k-steps [S, S_END) of one tile. Ring slot S&3 holds fragment S; the prefetch runs 4 k-steps ahead and crosses into the
// NEXT tile's buffer (tbn) at S >= 12, where the wave also issues its 8 LDS-DMA rows of the tile NBUF-1 ahead.
template <int LL, int S, int S_END>
__device__ __forceinline__ void tilew_steps(const char* tb, const char* tbn, const u32x4 (&qh0)[16], const u32x4 (&ql0)[16],
                                            const u32x4 (&qh1)[16], const u32x4 (&ql1)[16], f32x16& cur0, f32x16& cur1,
                                            const f32x16& prev0, const f32x16& prev1, int vmask, int code0, float pinf,
                                            WideLists<LL>& w, u32x4 (&ah)[4], u32x4 (&al)[4], const WideDma& dma) {
  if constexpr (S < S_END) {
    constexpr int VPM = (2 * LL + 2 + 5) / 6;
    const u32x4 a_hi = ah[S & 3], a_lo = al[S & 3];
    const char* nx = (S + 4 < 16) ? tb + 16 * (S + 4) : tbn + 16 * (S + 4 - 16);
    const int code = __builtin_amdgcn_readfirstlane(code0 + S);
#define T2L_GAP(e)                                                                         \
  __builtin_amdgcn_sched_barrier(0);                                                   \
  wide_sel_ops<LL, S, (e) * VPM, ((e) + 1) * VPM>(w, prev0, prev1, vmask, code, pinf); \
  __builtin_amdgcn_sched_barrier(0)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S == 0) mfma_bf16_first(cur0, a_hi, qh0[S]); else mfma_bf16_acc(cur0, a_hi, qh0[S]);
    ah[S & 3] = *reinterpret_cast<const u32x4*>(nx);
    T2L_GAP(0);
    if constexpr (S == 0) mfma_bf16_first(cur1, a_hi, qh1[S]); else mfma_bf16_acc(cur1, a_hi, qh1[S]);
    al[S & 3] = *reinterpret_cast<const u32x4*>(nx + 512);
    T2L_GAP(1);
    mfma_bf16_acc(cur0, a_hi, ql0[S]);
    if constexpr (S >= 12) dma.row(2 * (S - 12));
    T2L_GAP(2);
    mfma_bf16_acc(cur1, a_hi, ql1[S]);
    T2L_GAP(3);
    mfma_bf16_acc(cur0, a_lo, qh0[S]);
    if constexpr (S >= 12) dma.row(2 * (S - 12) + 1);
    T2L_GAP(4);
    mfma_bf16_acc(cur1, a_lo, qh1[S]);
    T2L_GAP(5);
#undef T2L_GAP
    tilew_steps<LL, S + 1, S_END>(tb, tbn, qh0, ql0, qh1, ql1, cur0, cur1, prev0, prev1, vmask, code0, pinf, w, ah, al,
                                  dma);
  }
}


// ---- f16 scan (scanh_kernel): ONE f16 MFMA per product. Inputs are scaled by exact powers of two so that the largest
// |element| of the DB (resp. of each query) lands in [2^14, 2^15): no f16 overflow, and an element is f16-denormal only
// below 2^-28 of the largest one, so whether the MFMA flushes denormals is irrelevant. shift = 14 - floor(log2(absmax));
// a key is the true score times 2^(shift_db + shift_q). Products of f16 values are exact in f32, so the only error on top
// of the f32 accumulation is the RNE rounding of the operands: |a.b - a'.b'| <= (2^-10 + 2^-21) |a||b|.
// Inputs whose largest magnitude is 0 / denormal / inf / NaN or astronomically far from 1 (|shift| > kHalfShiftMax) get
// shift 0 here and are sent to the exact float64 scan by the re-rank, which evaluates the same predicate.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr int kHalfShiftMax = 100;
constexpr int kHalfTileBytes = kTileRows * 512;  // 32 rows x 256 f16 = 16 KiB, tile-chunk-major (below)

__device__ __forceinline__ bool half_shift_of(float absmax, int& shift) {  // false = not representable (see above)
  const int e = (int)((__float_as_uint(absmax) >> 23) & 0xffu) - 127;
  shift = 14 - e;
  const bool ok = shift >= -kHalfShiftMax && shift <= kHalfShiftMax;
  if (!ok) shift = 0;
  return ok;
}
__device__ __forceinline__ unsigned pack_f16x2(float x0, float x1, int shift) {  // RNE (v_cvt_pk_f16_f32) of x * 2^shift
  return __builtin_bit_cast(unsigned,
                            __builtin_convertvector(f32x2{__builtin_ldexpf(x0, shift), __builtin_ldexpf(x1, shift)}, f16x2));
}

__device__ __forceinline__ void mfma_f16_first(f32x16& acc, const u32x4& a, const u32x4& b_agpr) {
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b_agpr));
}
__device__ __forceinline__ void mfma_f16_acc(f32x16& acc, const u32x4& a, const u32x4& b_agpr) {
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b_agpr));
}

// f16 DB plane, TILE-CHUNK-MAJOR: tile t (32 rows) is 16 KiB = [chunk c = 0..31][row r = 0..31][8 f16 = 16 B], chunk c
// holding k in [8c, 8c+8). The LDS image of a tile IS its global image, so an LDS-DMA piece (1 KiB per wave-instruction)
// is a straight contiguous copy (chunks 2p, 2p+1 of all 32 rows), and lane (col, half) reads its MFMA A fragment of
// k-step S — chunk half*16 + S of row col — at  half*8192 + col*16 + S*512: one per-lane base register plus an
// immediate, the 32 lanes of a half reading 512 contiguous bytes (conflict-free). (The row-major plane of round 1 needed
// an XOR swizzle against bank conflicts: 16 per-lane LDS offsets + 4 per-lane DMA offsets in registers.)
struct HalfDma {        // this wave's 4 LDS-DMA instructions of one tile (pieces 4w .. 4w+3, 1 KiB each)
  const char* src;      // wave-uniform global address of the wave's first piece of the tile
  unsigned dst;         // LDS byte address of that piece in the target buffer
  unsigned lane16;      // lane * 16
  __device__ __forceinline__ void piece(int i) const { lds_dma_row(dst + i * 1024, lane16, src + i * 1024); }
};

// k-steps [S, S_END) of one f16 tile: per k-step one fragment read (ring of 4, crossing into the next tile's buffer at
// S >= 12), two MFMAs (one per query group), the insertion of score S of the previous tile into both lists, and at
// S >= 12 one LDS-DMA piece of the tile NBUF-1 ahead. BUF / NBUFS are compile-time: every LDS offset is an immediate.
// lb = LDS address of the lane's fragment of chunk-row 0 in buffer 0 (half*8192 + col*16 from the buffers' base).
template <int LL, int S, int S_END, int BUF, int NBUFS>
__device__ __forceinline__ void tileh_steps(const char* lb, const u32x4 (&q0)[16],
                                            const u32x4 (&q1)[16], f32x16& cur0, f32x16& cur1, const f32x16& prev0,
                                            const f32x16& prev1, int vmask, int code0, float pinf, WideLists<LL>& w,
                                            u32x4 (&ring)[4], const HalfDma& dma) {
  if constexpr (S < S_END) {
    constexpr int VPM = LL + 1;
    constexpr int NXT = (BUF + 1) % NBUFS;
    const u32x4 a = ring[S & 3];
    const int code = __builtin_amdgcn_readfirstlane(code0 + S);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S == 0) mfma_f16_first(cur0, a, q0[S]); else mfma_f16_acc(cur0, a, q0[S]);
    if constexpr (S + 4 < 16) ring[S & 3] = *reinterpret_cast<const u32x4*>(lb + (S + 4) * 512 + BUF * kHalfTileBytes);
    else ring[S & 3] = *reinterpret_cast<const u32x4*>(lb + (S + 4 - 16) * 512 + NXT * kHalfTileBytes);
    __builtin_amdgcn_sched_barrier(0);
    wide_sel_ops<LL, S, 0, VPM>(w, prev0, prev1, vmask, code, pinf);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S == 0) mfma_f16_first(cur1, a, q1[S]); else mfma_f16_acc(cur1, a, q1[S]);
    if constexpr (S >= 12) dma.piece(S - 12);
    __builtin_amdgcn_sched_barrier(0);
    wide_sel_ops<LL, S, VPM, 2 * VPM>(w, prev0, prev1, vmask, code, pinf);
    __builtin_amdgcn_sched_barrier(0);
    tileh_steps<LL, S + 1, S_END, BUF, NBUFS>(lb, q0, q1, cur0, cur1, prev0, prev1, vmask, code0, pinf, w, ring, dma);
  }
}

// ---- the paired scan (scanp_kernel): TWO waves per SIMD. The f16 scan is bound by the wave's own instruction issue (an
// MFMA costs ~10 issue cycles and every VALU filler beyond 5 per MFMA gap 4 more: tools/mfma_agpr_probe.hip), not by
// the matrix pipe; a second wave on the same SIMD issues its VALU / LDS / DMA work into exactly those gaps
// (tools/pair_probe.hip: +20 % MFMA+9-filler throughput per SIMD, +12..21 % at 6-7 fillers). Waves w and w+4 of the
// 512-thread workgroup hold the SAME 64 queries and take alternate tiles of the workgroup's DB split ("virtual splits"
// sp and sp + nsplit of 2*nsplit), sharing one LDS ring of NS step-slots x 2 tiles.
struct PairDma {        // this wave's 4 LDS-DMA pieces of one STEP (2 tiles x pieces 2w, 2w+1)
  const char* src0;     // wave-uniform global addresses of the wave's first piece in the step's two tiles
  const char* src1;
  unsigned dst;         // LDS byte address of the wave's first piece in the slot's first tile buffer
  unsigned lane16;
  __device__ __forceinline__ void piece(int e) const {
    lds_dma_row(dst + (e >> 1) * kHalfTileBytes + (e & 1) * 1024, lane16, ((e >> 1) ? src1 : src0) + (e & 1) * 1024);
  }
};

// lb0 / lb1: the lane's fragment address (quad*16384 + half*8192 + col*16 from the ring base) for slots whose offset fits
// the 16-bit immediate of ds_read (slots 0, 1) and lb0 + 65536 for the others
typedef const __attribute__((address_space(3))) char* lds_cptr;  // explicit LDS pointer (32-bit): survives an opaque asm
template <int SLOT, int CHUNK>
__device__ __forceinline__ u32x4 pair_frag(lds_cptr lb0, lds_cptr lb1) {
  constexpr int OFF = SLOT * 2 * kHalfTileBytes + CHUNK * 512;
  typedef const __attribute__((address_space(3))) u32x4* frag_ptr;
  if constexpr (OFF < 65536) return *reinterpret_cast<frag_ptr>(lb0 + OFF);
  else return *reinterpret_cast<frag_ptr>(lb1 + (OFF - 65536));
}

// CS: the key code of accumulator register S is code0 + (S << CS) (2: the merged-record form leaves the two low code bits to the source list)
template <int LL, int S, int S_END, int SLOT, int NS, int CS = 0>
__device__ __forceinline__ void tilep_steps(lds_cptr lb0, lds_cptr lb1, const u32x4 (&q0)[16], const u32x4 (&q1)[16],
                                            f32x16& cur0, f32x16& cur1, const f32x16& prev0, const f32x16& prev1, int vmask,
                                            int code0, float pinf, WideLists<LL>& w, u32x4 (&ring)[4], const PairDma& dma) {
  if constexpr (S < S_END) {
    constexpr int VPM = LL + 1;
    constexpr int NXT = (SLOT + 1) % NS;
    const u32x4 a = ring[S & 3];
    const int code = __builtin_amdgcn_readfirstlane(code0 + (S << CS));
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S == 0) mfma_f16_first(cur0, a, q0[S]); else mfma_f16_acc(cur0, a, q0[S]);
    if constexpr (S + 4 < 16) ring[S & 3] = pair_frag<SLOT, S + 4>(lb0, lb1);
    else ring[S & 3] = pair_frag<NXT, S + 4 - 16>(lb0, lb1);
    __builtin_amdgcn_sched_barrier(0);
    wide_sel_ops<LL, S, 0, VPM>(w, prev0, prev1, vmask, code, pinf);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S == 0) mfma_f16_first(cur1, a, q1[S]); else mfma_f16_acc(cur1, a, q1[S]);
    if constexpr (S >= 12) dma.piece(S - 12);
    __builtin_amdgcn_sched_barrier(0);
    wide_sel_ops<LL, S, VPM, 2 * VPM>(w, prev0, prev1, vmask, code, pinf);
    __builtin_amdgcn_sched_barrier(0);
    tilep_steps<LL, S + 1, S_END, SLOT, NS, CS>(lb0, lb1, q0, q1, cur0, cur1, prev0, prev1, vmask, code0, pinf, w, ring, dma);
  }
}


// ---- tile-local selection for the paired scan (round 6; scanp_kernel<..., SEL = 1>). Inserting every score into the lane's list
// of LL = 6 costs 1 + LL = 7 VALU per score (112 per lane-tile and query group) beside MFMAs the wave can issue in 2 x 32 cycles per
// k-step: the selection, not the matrix pipe, sets the loop's pace. A score enters a list of 6 that has seen 16 t scores with
// probability 6 / (16 t): almost every one of those 6 med3 ops copies its own input. So the scores of a lane-tile first meet in
// a LOCAL top 3 per GROUP of 8 accumulator registers (t0 >= t1 >= t2; key + 3 med3 = 4 VALU per score), the best two are inserted
// into the list once per group (12 VALU) and the third goes into (dA >= dB), the two largest THIRD keys of the lane's groups
// (2 VALU): 86 instead of 112 VALU per lane-tile and query group. Exactness is untouched: whatever the lane dropped without it
// having passed through the list lies at or below dA, and dA is a real key — its code names the group it came from. The merged
// record carries the largest dA of its four lanes as B1 and everything else (list floors, evicted keys, the other seven d values)
// as B2 (search.hip): a certificate that fails on B1 alone is repaired by re-scoring the 8 rows of that ONE group, after which
// B2 bounds all that is unseen. Groups of 8, not 16: three of a query's ~12 relevant rows in the same 16 of 11,259 rows is a 4e-4
// event per query — more than one query per batch of 4,096, each 16 more rows for its re-rank wave (one more memory round trip
// than any other wave of the launch: measured +1.7 us on the re-rank kernel, which ends with its slowest wave); the same in 8 rows
// is a quarter as likely, and 8 rows travel in the registers of the round trip that wave makes anyway.
// Neighbouring rows (overlapping cells share objects: a query's best rows come in runs) are kept out of one tile by the plane
// itself (plane_row: row -> tile is strided, not blocked).
// The two query groups take turns (group 0 in k-steps 0..7 of a tile, group 1 in 8..15), so ONE set of t0 / t1 / t2 / key
// registers serves both. NaN keys (the -inf accumulators of the first step, OR-ed with a code) must stay no-ops: v_med3_f32 with a
// NaN operand returns min3 of the others, so every op below keeps a finite-or-minus-infinity value when its key is NaN.
template <int LL>
struct TileSelLists {
  float ls0[LL], ls1[LL];  // sorted key lists of the two query groups
  float dA0, dB0, dA1, dB1;  // per group: the two largest third-best keys of the lane's tiles (dA >= dB): what it dropped past its list
  float t0, t1, t2, k;     // tile-local top 3 of the group whose turn it is, key in flight
};
constexpr int kTileSelGroup = 8;                     // scores per tile-local group: the 16 accumulator registers of a lane-tile are two groups
constexpr int kTileSelGroupOps = 43;                 // VALU ops per group: 29 selection + 12 list insertion + 2 for (dA, dB)
constexpr int kTileSelOps = 2 * kTileSelGroupOps;    // per lane-tile and query group
template <int LL, int O, int CS>
__device__ __forceinline__ void tile_sel_op(float (&ls)[LL], float& dA, float& dB, TileSelLists<LL>& w, const f32x16& prev, int vmask,
                                            int code0, float pinf) {
  static_assert(LL == 6, "op table written for lists of 6");
  constexpr int B = (O / kTileSelGroupOps) * kTileSelGroup, o = O % kTileSelGroupOps;  // first accumulator register of the group, op within it
  auto key = [&](int i) { return __int_as_float((__float_as_int(prev[B + i]) & vmask) | __builtin_amdgcn_readfirstlane(code0 + ((B + i) << CS))); };
  if constexpr (o == 0) w.k = key(0);
  else if constexpr (o == 1) w.t0 = __builtin_amdgcn_fmed3f(w.k, -pinf, pinf);
  else if constexpr (o == 2) w.k = key(1);
  else if constexpr (o == 3) w.t1 = __builtin_amdgcn_fmed3f(w.t0, w.k, -pinf);
  else if constexpr (o == 4) w.t0 = __builtin_amdgcn_fmed3f(w.t0, w.k, pinf);
  else if constexpr (o == 5) w.k = key(2);
  else if constexpr (o == 6) w.t2 = __builtin_amdgcn_fmed3f(w.t1, w.k, -pinf);
  else if constexpr (o == 7) w.t1 = __builtin_amdgcn_fmed3f(w.t0, w.t1, w.k);
  else if constexpr (o == 8) w.t0 = __builtin_amdgcn_fmed3f(w.t0, w.k, pinf);
  else if constexpr (o < 29) {
    constexpr int J = (o - 9) >> 2, P = (o - 9) & 3;
    if constexpr (P == 0) w.k = key(3 + J);
    else if constexpr (P == 1) w.t2 = __builtin_amdgcn_fmed3f(w.t1, w.t2, w.k);
    else if constexpr (P == 2) w.t1 = __builtin_amdgcn_fmed3f(w.t0, w.t1, w.k);
    else w.t0 = __builtin_amdgcn_fmed3f(w.t0, w.k, pinf);
  } else if constexpr (o < 41) {  // t0, then t1, into the list: tail -> head, every element reads OLD neighbours only
    constexpr int I = LL - 1 - (o - 29) % LL;
    const float x = (o - 29) < LL ? w.t0 : w.t1;
    if constexpr (I == 0) ls[0] = __builtin_amdgcn_fmed3f(ls[0], x, pinf);
    else ls[I] = __builtin_amdgcn_fmed3f(ls[I - 1], ls[I], x);
  } else if constexpr (o == 41) {
    dB = __builtin_amdgcn_fmed3f(dA, dB, w.t2);
  } else {
    dA = __builtin_amdgcn_fmed3f(dA, w.t2, pinf);
  }
}
template <int LL, int O, int O_END, int CS>
__device__ __forceinline__ void tile_sel_ops(float (&ls)[LL], float& dA, float& dB, TileSelLists<LL>& w, const f32x16& prev, int vmask,
                                             int code0, float pinf) {
  if constexpr (O < O_END) {
    tile_sel_op<LL, O, CS>(ls, dA, dB, w, prev, vmask, code0, pinf);
    tile_sel_ops<LL, O + 1, O_END, CS>(ls, dA, dB, w, prev, vmask, code0, pinf);
  }
}
// k-steps [S, S_END) of one step of the paired scan with the tile-local selection. RD: depth of the fragment ring (k-steps of
// prefetch); the ring crosses into the next slot RD k-steps before the tile ends — behind the step's barrier at k-step 12.
// NOSEL: the launch's first step — the accumulators of "the tile before" hold nothing yet: MFMAs, ring and DMA only.
template <int LL, int S, int S_END, int SLOT, int NS, int CS, int RD, bool NOSEL = false>
__device__ __forceinline__ void tilep3_steps(lds_cptr lb0, lds_cptr lb1, const u32x4 (&q0)[16], const u32x4 (&q1)[16],
                                             f32x16& cur0, f32x16& cur1, const f32x16& prev0, const f32x16& prev1, int vmask,
                                             int code0, float pinf, TileSelLists<LL>& w, u32x4 (&ring)[RD], const PairDma& dma) {
  static_assert(RD == 2 || RD == 4, "ring depth");
  static_assert(RD <= 4, "the ring may cross into the next slot only behind the barrier at k-step 12");
  if constexpr (S < S_END) {
    constexpr int NXT = (SLOT + 1) % NS;
    constexpr int s8 = S & 7;
    constexpr int O0 = kTileSelOps * s8 / 8, O2 = kTileSelOps * (s8 + 1) / 8, O1 = (O0 + O2) / 2;
    const u32x4 a = ring[S & (RD - 1)];
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S == 0) mfma_f16_first(cur0, a, q0[S]); else mfma_f16_acc(cur0, a, q0[S]);
    if constexpr (S + RD < 16) ring[S & (RD - 1)] = pair_frag<SLOT, S + RD>(lb0, lb1);
    else ring[S & (RD - 1)] = pair_frag<NXT, S + RD - 16>(lb0, lb1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NOSEL) {
    } else if constexpr (S < 8) tile_sel_ops<LL, O0, O1, CS>(w.ls0, w.dA0, w.dB0, w, prev0, vmask, code0, pinf);
    else tile_sel_ops<LL, O0, O1, CS>(w.ls1, w.dA1, w.dB1, w, prev1, vmask, code0, pinf);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S == 0) mfma_f16_first(cur1, a, q1[S]); else mfma_f16_acc(cur1, a, q1[S]);
    if constexpr (S >= 12) dma.piece(S - 12);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NOSEL) {
    } else if constexpr (S < 8) tile_sel_ops<LL, O1, O2, CS>(w.ls0, w.dA0, w.dB0, w, prev0, vmask, code0, pinf);
    else tile_sel_ops<LL, O1, O2, CS>(w.ls1, w.dA1, w.dB1, w, prev1, vmask, code0, pinf);
    __builtin_amdgcn_sched_barrier(0);
    tilep3_steps<LL, S + 1, S_END, SLOT, NS, CS, RD, NOSEL>(lb0, lb1, q0, q1, cur0, cur1, prev0, prev1, vmask, code0, pinf, w, ring, dma);
  }
}


// ---- wave-wide all-reduces on the VALU (DPP + v_permlane{16,32}_swap), no LDS traffic: __shfl_xor lowers to
// ds_bpermute_b32, and the re-rank is shuffle-bound (hundreds of shuffles per query).
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;

// one DPP step of a maximum as ONE instruction (v_max_f32 with the DPP modifier on src0; the compiler's form is a
// v_mov_b32_dpp + a VOP3 max). Inline asm is invisible to the hazard recogniser: the 2 wait states a DPP read needs after
// the VALU write of its source are spelled out.
#define T2L_MAX_DPP(v, ctrl) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf" : "+v"(v))
__device__ __forceinline__ float wave_max_f32(float v, float pinf) {  // every lane gets the maximum
  T2L_MAX_DPP(v, "quad_perm:[1,0,3,2]");
  T2L_MAX_DPP(v, "quad_perm:[2,3,0,1]");
  T2L_MAX_DPP(v, "row_half_mirror");
  T2L_MAX_DPP(v, "row_mirror");
  {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __builtin_amdgcn_fmed3f(__uint_as_float(r[0]), __uint_as_float(r[1]), pinf);
  }
  {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __builtin_amdgcn_fmed3f(__uint_as_float(r[0]), __uint_as_float(r[1]), pinf);
  }
  return v;
}

// the value the lane 32 away holds (lane ^ 32) — v_permlane32_swap + a select instead of a ds_bpermute round trip through the LDS pipe
__device__ __forceinline__ float xor32_f32(float v, int half) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(half ? r[0] : r[1]);
}

template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  const unsigned long long b = __double_as_longlong(v);
  const unsigned lo = dpp_u<CTRL>((unsigned)b), hi = dpp_u<CTRL>((unsigned)(b >> 32));
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {  // every lane gets the sum (fixed tree order)
  v += dpp_d<kDppXor1>(v);
  v += dpp_d<kDppXor2>(v);
  v += dpp_d<kDppHalfMirror>(v);
  v += dpp_d<kDppMirror>(v);
  {
    const unsigned long long b = __double_as_longlong(v);
    const auto l = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    v = __longlong_as_double(((unsigned long long)h[0] << 32) | l[0]) +
        __longlong_as_double(((unsigned long long)h[1] << 32) | l[1]);
  }
  {
    const unsigned long long b = __double_as_longlong(v);
    const auto l = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    v = __longlong_as_double(((unsigned long long)h[0] << 32) | l[0]) +
        __longlong_as_double(((unsigned long long)h[1] << 32) | l[1]);
  }
  return v;
}

// all-reduce within each row of 16 lanes (the DPP row): every lane of the row gets the row's sum / maximum
__device__ __forceinline__ double row16_sum_f64(double v) {
  v += dpp_d<kDppXor1>(v);
  v += dpp_d<kDppXor2>(v);
  v += dpp_d<kDppHalfMirror>(v);
  v += dpp_d<kDppMirror>(v);
  return v;
}
__device__ __forceinline__ float row16_max_f32(float v) {
  v = fmaxf(v, __uint_as_float(dpp_u<kDppXor1>(__float_as_uint(v))));
  v = fmaxf(v, __uint_as_float(dpp_u<kDppXor2>(__float_as_uint(v))));
  v = fmaxf(v, __uint_as_float(dpp_u<kDppHalfMirror>(__float_as_uint(v))));
  v = fmaxf(v, __uint_as_float(dpp_u<kDppMirror>(__float_as_uint(v))));
  return v;
}

// The f16 plane deals rows to tiles STRIDED (round 6): inside a segment of the database (kSegmentRows rows; the whole shard when
// it is smaller) with F = rows / 32 full tiles, slot j of tile t < F holds row j F + t; the partial last tile (rows % 32 of them)
// stays blocked. Neighbouring rows — overlapping KITTI360Pose cells that share most of their objects, i.e. a query's best rows come
// in runs — therefore never meet in one tile, let alone in the 16 rows of one lane-tile: the tile-local selection of the paired scan
// (TileSelLists) keeps two rows per lane-tile, and every other list is spared runs too. A position is valid (< n_rows) exactly when
// its row is: full tiles map onto [0, 32 F), the partial tile onto itself. `pos`: position in the plane; `n_rows`: rows of the
// plane (all segments); returns the row.
__device__ __forceinline__ int plane_row(int pos, int n_rows) {
  const int seg0 = pos & ~(kSegmentRows - 1), p = pos - seg0;
  const int full = min(kSegmentRows, n_rows - seg0) >> 5;
  const int t = p >> 5, j = p & 31;
  return seg0 + (t < full ? j * full + t : p);
}

// key -> local DB row. part = 2*split + half; split sp owns tiles sp, sp + nsplit, sp + 2*nsplit, ... (interleaved, so a
// run of similar neighbouring rows spreads over all the per-lane lists instead of filling one). n_rows > 0: the key comes from the
// f16 plane, whose tiles hold strided rows (plane_row); 0: from a blocked plane (the split-bf16 scan's).
__device__ __forceinline__ int key_row(float key, int part, int nsplit, int code_bits, int n_rows_strided = 0) {
  const int code = __float_as_int(key) & ((1 << code_bits) - 1);
  const int r = code & 15;
  const int pos = (((code >> 4) * nsplit + (part >> 1)) << 5) + (r & 3) + 8 * (r >> 2) + 4 * (part & 1);
  return n_rows_strided > 0 ? plane_row(pos, n_rows_strided) : pos;
}

// float64 dot of DB row `row` with the query fragment held by the wave (lane owns dims 4*lane..4*lane+3)
__device__ __forceinline__ double wave_dot64(const float* __restrict__ db, int row, const float4 qv, int lane) {
  const float4 dv = reinterpret_cast<const float4*>(db + (size_t)row * kD)[lane];
  double d = (double)dv.x * qv.x + (double)dv.y * qv.y + (double)dv.z * qv.z + (double)dv.w * qv.w;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) d += __shfl_xor(d, off);
  return d;
}

// bound on (float64 score - key * kscale) for any row whose key is <= g: truncation of `code_bits` mantissa bits
// (relative 2^(code_bits-23), doubled for slack) plus the scan's arithmetic error eps32 (in true score units).
// kscale = 1 except for the f16 scan, whose keys are scores times a power of two.
__device__ __forceinline__ double key_slack(float g, int code_bits, double eps32, double kscale = 1.0) {
  return fabs((double)g) * kscale * ldexp(1.0, code_bits - 22) + eps32;
}


template <int KMAX>
__device__ void exact_scan(const float* __restrict__ db, int n_rows, const double* qs, int K, int row_offset,
                           int32_t* __restrict__ out_idx, double* __restrict__ out_score, double* red_s, int* red_i,
                           int* red_t) {
  const int tid = threadIdx.x;
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
    if (d > ls[KMAX - 1]) {  // rows ascend per thread: strict > keeps the lower row ahead among equals
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
      out_idx[r] = ok ? red_i[0] + row_offset : -1;
      if (out_score) out_score[r] = ok ? red_s[0] : -__builtin_inf();
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


}  // namespace t2l
