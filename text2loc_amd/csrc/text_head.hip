// The text head after the frozen T5 encoder on the f16 matrix pipe.
//
// Replaces: LanguageEncoder.forward downstream of `out.last_hidden_state` up to and including `inter_mlp`
// (models/language_encoder.py:127-135): hidden [n_sentences, L, 1024] -> TransformerEncoderLayer(d_model = 1024, 4 heads,
// dim_feedforward = 4096, post-norm, ReLU, NO padding mask) over the L token positions of every sentence -> max over the
// tokens -> Linear(1024 -> D) + BatchNorm1d (eval, folded). T5 itself and the 256-wide inter-sentence layer behind it stay
// on PyTorch-ROCm (north star); this is the part that costs: 25.2 MFLOP per token, 9.9 TFLOP for one search step's worth of
// queries (4,096 descriptions x 6 hints x 16 tokens), 82 ms on rocBLAS f32.
//
// Arithmetic: split-f16 (mfma_h3.h): every f32 operand a = hi + lo, hi = f16(a), lo = f16(a - hi); a product is
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with f32 accumulation (~5e-7 relative) — or ONE product per pair with
// option encoder_f16 (~1e-4, north-star bar 1e-3). Everything that enters a product is watched against the f16 range
// (3e4): a call that leaves it raises the overflow flag and the caller (text2loc_amd.cell_retrieval.LanguageEncoder.head)
// re-runs that batch on the PyTorch path.
//
// Data layout in HBM. Rows = tokens (sentence-major: row = sentence * L + token), padded to a multiple of 256.
//   T16 (operands of the matrix pipe, as separate hi and lo planes): f16 [rows/32][cols/8][32][8] — for every 32-row tile
//       and every 8-wide k chunk, 32 rows x 16 B = 512 B contiguous. That block IS the LDS image of a 32x8 operand
//       fragment half (lane (row, kh) of a 32x32x16 MFMA reads 16 B at row * 16 of chunk 2 s + kh), so an LDS-DMA piece is a
//       contiguous 1 KiB copy (two chunks) and every fragment read is one conflict-free ds_read_b128 at base + immediate.
//       Weights [N][K] are packed the same way at load time.
//   T32 (f32 intermediates that the matrix pipe does not read): f32 [rows/32][cols/4][32][4]: the 4 consecutive output
//       columns a lane owns in the MFMA's C/D layout are one 16 B store and a wave's store is 1 KiB contiguous.
//
// Kernels:
//   th_split_kernel   row-major f32 -> T16 hi/lo planes (the T5 hidden states; the pooled sentence vectors)
//   th_gemm_kernel    C^T = W * X^T on 256 x 256 tiles, 8 waves (2 per SIMD), k in steps of 16 through a 4-slot LDS ring
//                     (32 KiB per slot: W_hi, W_lo, X_hi, X_lo) filled by LDS-DMA three steps ahead; ONE s_waitcnt vmcnt +
//                     s_barrier per step. The transposed product puts 4 consecutive output COLUMNS into one lane, so every
//                     epilogue (bias, residual from T16 planes, ReLU, re-split) stores 8/16 B pieces that tile full lines.
//   th_attn_kernel    softmax(q k^T / 16) v per (group of sentences filling a 32-row tile, head), L <= 32 tokens, f32 MFMA from
//                     registers (0.3 % of the FLOPs)
//   th_ln_kernel      LayerNorm over 1024 columns from T32, two-pass in registers; writes T16 planes, or (POOL) the max over
//                     each sentence's tokens
#include "mfma_h3.h"
#include "search_dev.h"
#include "t2l_internal.h"

namespace t2l {
namespace th {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int kDM = 1024;      // d_model of the T5 encoder the head is built for (t5-large)
constexpr int kFF = 4096;      // dim_feedforward = 4 * d_model (language_encoder.py:95)
constexpr int kHeads = 4, kHD = 256;
constexpr int kMaxL = 32;      // token positions per sentence the attention kernel holds
constexpr int kTile = 256;     // GEMM tile (rows of X and rows of W per workgroup)
constexpr int kSlotBytes = 32 * 1024, kSlots = 4, kPlaneBytes = 8 * 1024;

__device__ __forceinline__ size_t t16_index(int m, int k, int cols) {  // element index into a T16 plane
  return ((size_t)(m >> 5) * (cols >> 3) + (k >> 3)) * 256 + (m & 31) * 8 + (k & 7);
}
__device__ __forceinline__ size_t t32_index(int m, int n, int cols) {  // element index into a T32 array
  return ((size_t)(m >> 5) * (cols >> 2) + (n >> 2)) * 128 + (m & 31) * 4 + (n & 3);
}
__device__ __forceinline__ void split4(const f32x4 v, f16x4& hi, f16x4& lo) {
  hi = __builtin_convertvector(v, f16x4);
  lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), f16x4);
}
__device__ __forceinline__ bool out_of_f16_range(const f32x4 v) {  // true for |v| >= 3e4, inf and NaN
  return !(fabsf(v[0]) < kSplitF16Safe && fabsf(v[1]) < kSplitF16Safe && fabsf(v[2]) < kSplitF16Safe && fabsf(v[3]) < kSplitF16Safe);
}

// ---------------------------------------------------------------------------------------------------------------
// row-major f32 [M][C] -> T16 hi / lo planes of M_pad rows (rows >= M are written as zeros). One workgroup per 32 rows x 256
// columns, through LDS: whole-row reads (1 KiB per wave-load), then thread (row, chunk group) so that consecutive threads
// write one 512-byte block.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void th_split_kernel(const float* __restrict__ x, int M, int C, _Float16* __restrict__ hi,
                                                       _Float16* __restrict__ lo, int* __restrict__ flag) {
  __shared__ __attribute__((aligned(16))) float tile[32][260];
  const int m0 = blockIdx.x * 32, col0 = blockIdx.y * 256;
  {  // a wave reads whole 1 KiB row pieces (64 lanes x 16 B contiguous), 8 rows per wave
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = wv * 8 + i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m0 + row < M) v = *reinterpret_cast<const f32x4*>(x + (size_t)(m0 + row) * C + col0 + lane * 4);
      *reinterpret_cast<f32x4*>(&tile[row][lane * 4]) = v;
    }
  }
  __syncthreads();
  const int r = threadIdx.x & 31, cg = threadIdx.x >> 5;
  const int m = m0 + r;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int kc = (cg * 4 + j) * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(&tile[r][kc]), b = *reinterpret_cast<const f32x4*>(&tile[r][kc + 4]);
    bad = bad || out_of_f16_range(a) || out_of_f16_range(b);
    f16x4 ah, al, bh, bl;
    split4(a, ah, al);
    split4(b, bh, bl);
    const size_t o = t16_index(m, col0 + kc, C);
    *reinterpret_cast<f16x8*>(hi + o) = f16x8{ah[0], ah[1], ah[2], ah[3], bh[0], bh[1], bh[2], bh[3]};
    *reinterpret_cast<f16x8*>(lo + o) = f16x8{al[0], al[1], al[2], al[3], bl[0], bl[1], bl[2], bl[3]};
  }
  if (bad) atomicOr(flag, 1);
}

// ---------------------------------------------------------------------------------------------------------------
// The GEMM. Workgroup (m tile, n tile): OUT[m][n] = sum_k X[m][k] W[n][k] for 256 rows x 256 columns, computed transposed
// (MFMA A operand = 32 rows of W, B operand = 32 rows of X) so that lane (c, kh) of accumulator tile (nt, mt) holds row
// m = 32 mt + c and columns n = 32 nt + 8 g + 4 kh + {0,1,2,3} in registers 4 g .. 4 g + 3.
// Wave w: wm = w & 1 -> rows [128 wm, +128) (4 tiles), wn = w >> 1 -> columns [64 wn, +64) (2 tiles): 8 accumulators,
// per k-step 12 fragment reads (SINGLE: 6) feed 24 MFMAs (SINGLE: 8).
// ---------------------------------------------------------------------------------------------------------------
enum Epi { kEpiT32 = 0, kEpiResidT32 = 1, kEpiReluT16 = 2, kEpiRowMajor = 3 };

struct GemmArgs {
  const char *wh, *wl;   // T16 planes of W [n_pad][K]
  const char *xh, *xl;   // T16 planes of X [m_pad][K]
  const float* bias;     // [n_pad]
  const char *rh, *rl;   // kEpiResidT32: T16 planes of the residual [m_pad][N]
  void *out0, *out1;     // T32 f32 | T16 hi, lo | row-major f32
  int* flag;             // overflow flag (kEpiReluT16)
  int m_tiles, n_tiles, K, N, M, n_real;  // N = output columns of the tiled layouts (= 256 n_tiles), M / n_real: row-major bounds
  int relu = 0, accumulate = 0;           // kEpiRowMajor only: out = max(.., 0); accumulate 1: out += (dW accumulates, dX adds to a residual
                                          // path); 2: the same with float atomics (several workgroups per output tile: ksplit > 1)
  int ksplit = 1;                         // kEpiRowMajor only: the contraction of every output tile is cut into ksplit jobs (few-tile products —
                                          // dW [N][K] over thousands of rows — would otherwise leave most CUs idle); needs accumulate = 2
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N > 63 ? 63 : N) : "memory"); }

// PERSISTENT: the grid is one workgroup per CU; workgroup b walks the tiles vb = b, b + grid, ... as ONE continuous stream of
// k-steps — the LDS-DMA of a tile's first three steps is issued during the last three steps of the tile before it, so the
// matrix pipe restarts right behind an epilogue instead of behind a cold pipeline fill, and the epilogues' store bursts of
// different CUs drift apart instead of hitting HBM together (measured before: 85 us per 64-step tile against 51 us of MFMA time).
// vmcnt counts the epilogue's stores too (in-order retirement): for the three steps behind an epilogue the wait allows them
// to stay in flight (they are younger than the awaited pieces), from the fourth step on they must have retired.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// BF16 (the training GEMMs, train.hip -> fast_gemm): the planes hold bf16 hi / lo halves (f32's exponent range: gradients need no
// scaling) and the products run on v_mfma_f32_32x32x16_bf16 — same shape, same rate, same plane layout as the f16 form.
template <int EPI, bool SINGLE, bool BF16 = false>
__global__ __launch_bounds__(512, 1) void th_gemm_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PPW = SINGLE ? 2 : 4;  // LDS-DMA pieces (1 KiB) per wave and k-step
  // stores per wave and epilogue that are CERTAIN to be issued (an over-estimate would under-wait): the row-major epilogue
  // stores conditionally -> 0 (its stores must retire before the next tile's first step; it has one n tile per m tile)
  constexpr int NST = EPI == kEpiRowMajor ? 0 : (EPI == kEpiReluT16 && !SINGLE) ? 64 : 32;
  const int lane = threadIdx.x & 63, w = uniform_wave_id();
  // virtual block vb -> tile: the n tiles of one m tile run back to back on ONE XCD (vb % 8 = blockIdx % 8: the grid is a
  // multiple of 8), so the X rows of an m tile are fetched from HBM once and every XCD's L2 keeps the weights
  const int xcd = blockIdx.x & 7, grid = gridDim.x;
  const int ks = EPI == kEpiRowMajor ? max(1, p.ksplit) : 1;
  const int K = p.K, KT = (K >> 4) / ks;  // k-steps of one job (a tile, or a ksplit-th of one)
  const int n_vb = (p.m_tiles + 7) / 8 * 8 * p.n_tiles * ks;
  int n_my = 0;  // this workgroup's jobs: a prefix of vb = blockIdx + i * grid (an m tile beyond m_tiles ends the list)
  for (int vb = blockIdx.x; vb < n_vb; vb += grid) {
    if (((vb >> 3) / (p.n_tiles * ks)) * 8 + xcd >= p.m_tiles) break;
    ++n_my;
  }
  if (n_my == 0) return;
  const size_t rt_stride = (size_t)K * 64;  // bytes between consecutive 32-row tiles of a T16 plane

  // ---- this wave's DMA pieces: piece ids [w * PPW, +PPW) of the slot; plane = id / 8, row tile = id % 8
  const int pid0 = w * PPW;
  const int plane = pid0 >> 3, rt0 = pid0 & 7;
  const char* plane_base;
  unsigned dst_off;
  if constexpr (SINGLE) {
    plane_base = plane == 0 ? p.wh : p.xh;
    dst_off = (plane == 0 ? 0u : 2u * kPlaneBytes) + rt0 * 1024;
  } else {
    plane_base = plane == 0 ? p.wh : plane == 1 ? p.wl : plane == 2 ? p.xh : p.xl;
    dst_off = plane * kPlaneBytes + rt0 * 1024;
  }
  const bool is_w = SINGLE ? plane == 0 : plane < 2;
  auto tile_src = [&](int i) {
    const int seq = (blockIdx.x + i * grid) >> 3;
    const int kp = seq % ks, t2 = seq / ks;
    const int t = is_w ? t2 % p.n_tiles : (t2 / p.n_tiles) * 8 + xcd;
    return plane_base + ((size_t)t * 8 + rt0) * rt_stride + (size_t)kp * KT * 1024;
  };
  unsigned voff[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) voff[j] = (unsigned)(lane * 16) + (unsigned)(j * rt_stride);
  const unsigned lds0 = lds_addr_of(smem);
  int it_tile = 0, it_kt = 0;  // the issue cursor runs three k-steps ahead of the compute cursor, across tile boundaries
  unsigned gi = 0;
  const char* it_src = tile_src(0);
  auto issue = [&]() {
    if (it_tile < n_my) {
      const unsigned dst = lds0 + (gi & 3) * kSlotBytes + dst_off;
      const char* s = it_src + (size_t)it_kt * 1024;
#pragma unroll
      for (int j = 0; j < PPW; ++j) lds_dma_row(dst + j * 1024, voff[j], s);
      ++gi;
      if (++it_kt == KT) {
        it_kt = 0;
        if (++it_tile < n_my) it_src = tile_src(it_tile);
      }
    }
  };

  const int wm = w & 1, wn = w >> 1;
  const char* fr = smem + lane * 16;
  const int w_off = wn * 2 * 1024, x_off = 2 * kPlaneBytes + wm * 4 * 1024;
  const int c = lane & 31, kh = lane >> 5;
  bool bad = false;

  // Fragments are double-buffered in registers: step g multiplies the set read during step g - 1 while the set of step g + 1
  // is being read, so the LDS latency sits behind 24 MFMAs instead of in front of them. Step g's barrier therefore certifies
  // the pieces of step g + 1 (and that everybody finished READING slot g, which step g + 4's pieces then refill).
  struct Frags {
    f16x8 xh[4], xl[4], wh[2], wl[2];
  };
  auto read_frags = [&](Frags& f, unsigned slot) {
    const char* sb = fr + slot * kSlotBytes;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f.wh[i] = *reinterpret_cast<const f16x8*>(sb + w_off + i * 1024);
      if constexpr (!SINGLE) f.wl[i] = *reinterpret_cast<const f16x8*>(sb + kPlaneBytes + w_off + i * 1024);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f.xh[i] = *reinterpret_cast<const f16x8*>(sb + x_off + i * 1024);
      if constexpr (!SINGLE) f.xl[i] = *reinterpret_cast<const f16x8*>(sb + kPlaneBytes + x_off + i * 1024);
    }
  };
  const unsigned total = (unsigned)n_my * KT;
  issue();
  issue();
  issue();
  issue();
  // (total >= 64: every wait below has its three younger groups except at the very end)
  wait_vmcnt<3 * PPW>();
  asm volatile("s_barrier" ::: "memory");
  Frags fa, fb;
  read_frags(fa, 0);
  unsigned g = 0;
  for (int ti = 0; ti < n_my; ++ti) {
    const int seq = (blockIdx.x + ti * grid) >> 3;
    const int kpart = seq % ks, seq2 = seq / ks;
    const int mt_idx = (seq2 / p.n_tiles) * 8 + xcd, nt_idx = seq2 % p.n_tiles;
    h3_f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    auto step = [&](const Frags& cur, Frags& nxt, int kt) {
      // pieces of step g + 1 must have landed: groups g + 2, g + 3 (and, right behind an epilogue, its stores) may stay in flight
      const unsigned ahead = min(total - 1 - g, 3u);  // k-steps issued behind step g
      if (ahead >= 1) {
        if (ti > 0 && kt < 3) {  // the previous tile's stores are younger than group g + 1 (issued before that epilogue)
          if (ahead == 3) wait_vmcnt<2 * PPW + NST>();
          else if (ahead == 2) wait_vmcnt<PPW + NST>();
          else wait_vmcnt<NST>();
        } else {
          if (ahead == 3) wait_vmcnt<2 * PPW>();
          else if (ahead == 2) wait_vmcnt<PPW>();
          else wait_vmcnt<0>();
        }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this wave's reads of slot g % 4 returned long ago (and the compiler knows)
      asm volatile("s_barrier" ::: "memory");
      issue();  // group g + 4 into slot g % 4 (everybody read it during step g - 1)
      if (ahead >= 1) read_frags(nxt, (g + 1) & 3);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if constexpr (BF16) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.wh[a]), __builtin_bit_cast(bf16x8, cur.xh[b]), acc[a][b], 0, 0, 0);
            if constexpr (!SINGLE) {
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.wh[a]), __builtin_bit_cast(bf16x8, cur.xl[b]), acc[a][b], 0, 0, 0);
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.wl[a]), __builtin_bit_cast(bf16x8, cur.xh[b]), acc[a][b], 0, 0, 0);
            }
          } else {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wh[a], cur.xh[b], acc[a][b], 0, 0, 0);
          if constexpr (!SINGLE) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wh[a], cur.xl[b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wl[a], cur.xh[b], acc[a][b], 0, 0, 0);
          }
          }
        }
      ++g;
    };
    for (int kt = 0; kt < KT; kt += 2) {  // (K is a multiple of 32)
      step(fa, fb, kt);
      step(fb, fa, kt + 1);
    }

    // ---- epilogue. The bias comes through scalar loads (wave-uniform address; lgkmcnt, so the DMA queue keeps running);
    // a tile's 8 residual pieces are loaded together (one memory latency per batch, not one per piece)
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int n_base = nt_idx * kTile + wn * 64 + a * 32;
    f32x4 bias_v[4];
    {
      f32x8 s0, s1, s2, s3;  // this half's 32 bias values: 4 scalar loads in flight, one wait
      const float* bp = p.bias + n_base;
      asm volatile("s_load_dwordx8 %0, %4, 0x0\n\ts_load_dwordx8 %1, %4, 0x20\n\ts_load_dwordx8 %2, %4, 0x40\n\ts_load_dwordx8 %3, %4, 0x60\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3)
                   : "s"(bp)
                   : "memory");
      bias_v[0] = kh ? f32x4{s0[4], s0[5], s0[6], s0[7]} : f32x4{s0[0], s0[1], s0[2], s0[3]};
      bias_v[1] = kh ? f32x4{s1[4], s1[5], s1[6], s1[7]} : f32x4{s1[0], s1[1], s1[2], s1[3]};
      bias_v[2] = kh ? f32x4{s2[4], s2[5], s2[6], s2[7]} : f32x4{s2[0], s2[1], s2[2], s2[3]};
      bias_v[3] = kh ? f32x4{s3[4], s3[5], s3[6], s3[7]} : f32x4{s3[0], s3[1], s3[2], s3[3]};
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int m = mt_idx * kTile + wm * 128 + b * 32 + c;
      f16x4 rh[4], rl[4];
      if constexpr (EPI == kEpiResidT32) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const size_t ro = t16_index(m, n_base + 8 * g + 4 * kh, p.N);
          rh[g] = *reinterpret_cast<const f16x4*>((const _Float16*)p.rh + ro);
          rl[g] = *reinterpret_cast<const f16x4*>((const _Float16*)p.rl + ro);  // (SINGLE too: the residual stays exact)
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n0 = n_base + 8 * g + 4 * kh;
        const f32x4 bv = (kpart == 0 && p.accumulate != 3) ? bias_v[g] : f32x4{0.f, 0.f, 0.f, 0.f};  // (first k-part only; slabs: the reduction adds it)
        f32x4 v = {acc[a][b][4 * g] + bv[0], acc[a][b][4 * g + 1] + bv[1], acc[a][b][4 * g + 2] + bv[2], acc[a][b][4 * g + 3] + bv[3]};
        if constexpr (EPI == kEpiT32) {
          *reinterpret_cast<f32x4*>((float*)p.out0 + t32_index(m, n0, p.N)) = v;
        } else if constexpr (EPI == kEpiResidT32) {
          const f32x4 res = __builtin_convertvector(rh[g], f32x4) + __builtin_convertvector(rl[g], f32x4);
          *reinterpret_cast<f32x4*>((float*)p.out0 + t32_index(m, n0, p.N)) = v + res;
        } else if constexpr (EPI == kEpiReluT16) {
          v = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
          bad = bad || (m < p.M && out_of_f16_range(v));  // (padding rows carry whatever the buffers held)
          f16x4 hi, lo;
          split4(v, hi, lo);
          const size_t o = t16_index(m, n0, p.N);
          *reinterpret_cast<f16x4*>((_Float16*)p.out0 + o) = hi;
          if constexpr (!SINGLE) *reinterpret_cast<f16x4*>((_Float16*)p.out1 + o) = lo;
        } else {  // row-major f32 [M][n_real]
          if (m < p.M && n0 < p.n_real) {
            f32x4* dst = reinterpret_cast<f32x4*>((float*)p.out0 + (size_t)m * p.n_real + n0);
            if (p.relu) v = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
            if (p.accumulate == 3) {  // this k-part's slab of the partial-sum scratch (th_reduce_parts_kernel adds the slabs up)
              dst[(size_t)kpart * ((size_t)p.M * p.n_real / 4)] = v;
            } else if (p.accumulate == 2) {
              float* d = reinterpret_cast<float*>(dst);
              unsafeAtomicAdd(d, v[0]); unsafeAtomicAdd(d + 1, v[1]); unsafeAtomicAdd(d + 2, v[2]); unsafeAtomicAdd(d + 3, v[3]);
            } else {
              if (p.accumulate) v += *dst;
              *dst = v;
            }
          }
        }
      }
    }
  }
  }  // tiles
  if constexpr (EPI == kEpiReluT16) {
    if (bad) atomicOr(p.flag, 1);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Self-attention over the L token positions of every sentence, no mask (nn.TransformerEncoderLayer as
// language_encoder.py:130-131 calls it), on the f32 MFMA, from registers — no LDS. qkv: T32 [m_pad][3072] (q | k | v, head h at
// columns 256 h). One wave per (group of G = floor(32 / L) consecutive sentences = up to 32 consecutive token rows, head):
//   S^T = K Q^T   A operand = key rows, B operand = query rows; the k-sum is order-free, so lane (c, kh) takes
//                 k in [128 kh, 128 kh + 128): its operands are 32 contiguous float4 of row c. D layout: lane (i = c, kh)
//                 holds keys j = (r & 3) + 8 (r >> 2) + 4 kh, r = 0..15 — the softmax over keys is in-lane + one lane ^ 32
//                 exchange; keys of another sentence of the group (and rows beyond the group) are masked out.
//   O^T = V^T P^T the probability registers ARE the B operand (k-step s <-> register s); the A operand gathers
//                 V[key j(s, kh)][32 dt + c] straight from the T32 array (4-byte loads, 16 B segments, L1/L2 hits);
//                 lane (i, kh) ends up with 4 consecutive output columns per register quad -> 8-byte T16 stores.
// ---------------------------------------------------------------------------------------------------------------
template <int DM>  // d_model: 1024 (the token layer, head_dim 256) or 256 (the inter-sentence layer, head_dim 64)
__global__ __launch_bounds__(256) void th_attn_kernel(const float* __restrict__ qkv, int n_sent, int L, int n_groups,
                                                      _Float16* __restrict__ ohi, _Float16* __restrict__ olo, int* __restrict__ flag) {
  constexpr int kDM = DM, kHD = DM / kHeads;
  constexpr int kQuads = kHD / 8;          // float4 quads of a lane's half row (k in [kHD/2 * kh, +kHD/2))
  constexpr float kScale = kHD == 256 ? 0.0625f : 0.125f;  // 1 / sqrt(head_dim)
  static_assert(kHD == 256 || kHD == 64, "head_dim 256 or 64");
  const int lane = threadIdx.x & 63, c = lane & 31, kh = lane >> 5;
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);  // (group, head)
  if (unit >= n_groups * kHeads) return;
  const int grp = unit >> 2, h = unit & 3;
  const int G = 32 / L;
  const int s_first = grp * G, rows = min(G, n_sent - s_first) * L;  // valid token rows of this group
  const int m0 = s_first * L;
  const int N3 = 3 * kDM;
  const int mc = m0 + min(c, rows - 1);  // this lane's row (clamped: rows beyond the group are masked)
  // ---- S^T = K Q^T over head_dim = 256
  h3_f32x16 st;
#pragma unroll
  for (int r = 0; r < 16; ++r) st[r] = 0.f;
  const float* qrow = qkv + t32_index(mc, h * kHD + (kHD / 2) * kh, N3);    // quad t of the lane's half row: + t * 128 floats
  const float* krow = qkv + t32_index(mc, kDM + h * kHD + (kHD / 2) * kh, N3);
#pragma unroll 2
  for (int t0 = 0; t0 < kQuads; t0 += 4) {
    f32x4 qv[4], kv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      qv[t] = *reinterpret_cast<const f32x4*>(qrow + (size_t)(t0 + t) * 128);
      kv[t] = *reinterpret_cast<const f32x4*>(krow + (size_t)(t0 + t) * 128);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[t][e], qv[t][e], st, 0, 0, 0);
  }
  // ---- softmax over the keys of this lane's query row i = c (its own sentence only)
  const int my_sent = c / L;
  float mx = -__builtin_inff();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = (r & 3) + 8 * (r >> 2) + 4 * kh;
    const bool ok = j < rows && j / L == my_sent;
    st[r] = ok ? st[r] * kScale : -__builtin_inff();  // / sqrt(head_dim)
    mx = fmaxf(mx, st[r]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    st[r] = c < rows ? __expf(st[r] - mx) : 0.f;  // (a row beyond the group has no valid key: mx = -inf)
    sum += st[r];
  }
  sum += __shfl_xor(sum, 32);
  const float inv = c < rows ? 1.f / sum : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) st[r] *= inv;
  // ---- O^T = V^T P^T, 8 column tiles of 32
  int vrow[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int j = min((s & 3) + 8 * (s >> 2) + 4 * kh, rows - 1);  // (masked keys carry probability 0: any finite row will do)
    const int m = m0 + j;
    vrow[s] = (m >> 5) * (N3 >> 2) * 128 + (m & 31) * 4;
  }
  const float* vbase = qkv + (size_t)((2 * kDM + h * kHD) >> 2) * 128 + ((c >> 2) * 128 + (c & 3));
  bool bad = false;
  float va[16], vb[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) va[s] = vbase[vrow[s]];
#pragma unroll
  for (int dt = 0; dt < kHD / 32; ++dt) {
    if (dt + 1 < kHD / 32) {
#pragma unroll
      for (int s = 0; s < 16; ++s) vb[s] = vbase[vrow[s] + (dt + 1) * 8 * 128];
    }
    h3_f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) o = __builtin_amdgcn_mfma_f32_32x32x2f32(va[s], st[s], o, 0, 0, 0);
    if (c < rows) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = {o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
        bad = bad || out_of_f16_range(v);
        f16x4 hi, lo;
        split4(v, hi, lo);
        const size_t off = t16_index(m0 + c, h * kHD + dt * 32 + 8 * g + 4 * kh, kDM);
        *reinterpret_cast<f16x4*>(ohi + off) = hi;
        *reinterpret_cast<f16x4*>(olo + off) = lo;
      }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) va[s] = vb[s];
  }
  if (bad) atomicOr(flag, 1);
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm over the 1024 columns of every row (eps 1e-5, biased variance, two passes over registers), input T32.
// One workgroup per 32-row tile; thread (r, g) holds quads g, g + 8, ..., g + 248 of row r (128 floats).
// POOL = false: writes the T16 hi / lo planes (the next GEMM's operand and residual).
// POOL = true : max over the L token rows of each sentence -> pooled [n_sentences][1024] row-major (pre-filled with -inf;
//               sentences that lie inside this tile are stored, straddlers meet through float atomic max).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int m0_of(int row_tile) { return row_tile * 32; }
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));
}

// DM = 256 (the inter-sentence layer): the same with 8 quads per thread; POOL then adds `resid` (row-major f32 [M][DM], the
// layer's own input: `x += layer(x)`, language_encoder.py:143-144) to every normalised row before the max over each
// description's L sentence rows.
template <bool POOL, int DM = 1024>
__global__ __launch_bounds__(256) void th_ln_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, int M, int L, _Float16* __restrict__ hi,
                                                    _Float16* __restrict__ lo, float* __restrict__ pooled, int* __restrict__ flag,
                                                    const float* __restrict__ resid = nullptr) {
  constexpr int kDM = DM, NQ = DM / 32;  // quads per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);  // [8][32]
  const int r = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int rt = blockIdx.x, m = rt * 32 + r;
  const float* base = y + (size_t)rt * (kDM / 4) * 128 + r * 4;
  f32x4 v[NQ];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    v[j] = *reinterpret_cast<const f32x4*>(base + (size_t)(j * 8 + g) * 128);
    s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
  }
  red[g * 32 + r] = s;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) mean += red[k * 32 + r];
  mean *= 1.f / kDM;
  __syncthreads();
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    v[j] -= mean;
    q += (v[j][0] * v[j][0] + v[j][1] * v[j][1]) + (v[j][2] * v[j][2] + v[j][3] * v[j][3]);
  }
  red[g * 32 + r] = q;
  __syncthreads();
  float var = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) var += red[k * 32 + r];
  const float rstd = 1.f / sqrtf(var * (1.f / kDM) + 1e-5f);
  bool bad = false;
  if constexpr (!POOL) {
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int n0 = 4 * (j * 8 + g);
      const f32x4 o = v[j] * rstd * *reinterpret_cast<const f32x4*>(gamma + n0) + *reinterpret_cast<const f32x4*>(beta + n0);
      bad = bad || (m < M && out_of_f16_range(o));
      f16x4 h, l;
      split4(o, h, l);
      const size_t off = t16_index(m, n0, kDM);
      *reinterpret_cast<f16x4*>(hi + off) = h;
      *reinterpret_cast<f16x4*>(lo + off) = l;
    }
    if (bad) atomicOr(flag, 1);
  } else {
    __syncthreads();
    constexpr int RS = kDM + 4;
    float* tile = reinterpret_cast<float*>(smem);  // [32][1028]
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int n0 = 4 * (j * 8 + g);
      *reinterpret_cast<f32x4*>(tile + r * RS + n0) =
          v[j] * rstd * *reinterpret_cast<const f32x4*>(gamma + n0) + *reinterpret_cast<const f32x4*>(beta + n0);
    }
    __syncthreads();
    if (resid) {  // x + layer(x): whole-row reads of the layer's input
      for (int i = threadIdx.x; i < 32 * (kDM / 4); i += 256) {
        const int rr = i / (kDM / 4), cq = i % (kDM / 4);
        if (m0_of(rt) + rr < M)
          *reinterpret_cast<f32x4*>(tile + rr * RS + 4 * cq) += *reinterpret_cast<const f32x4*>(resid + (size_t)(m0_of(rt) + rr) * kDM + 4 * cq);
      }
      __syncthreads();
    }
    const int n0 = threadIdx.x * 4, m0 = rt * 32;
    int row = 0;
    while (n0 < kDM && row < 32 && m0 + row < M) {
      const int sent = (m0 + row) / L;
      const int first = sent * L, last = first + L - 1;  // the sentence's rows
      const int end = min(32, last - m0 + 1);
      f32x4 mx = *reinterpret_cast<const f32x4*>(tile + row * RS + n0);
      for (int k = row + 1; k < end; ++k) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(tile + k * RS + n0);
        mx = {fmaxf(mx[0], t[0]), fmaxf(mx[1], t[1]), fmaxf(mx[2], t[2]), fmaxf(mx[3], t[3])};
      }
      float* dstp = pooled + (size_t)sent * kDM + n0;
      if (first >= m0 && last < m0 + 32) {
        *reinterpret_cast<f32x4*>(dstp) = mx;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomic_max_f32(dstp + e, mx[e]);
      }
      row = end;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Operand planes of the TRAINING GEMMs (fast_gemm below): row-major f32 -> T16 planes of bf16 hi / lo halves.
//   TRANS = false: A [R][C] -> planes of A      (rows R -> padded to a multiple of 256, contraction C a multiple of 32)
//   TRANS = true : A [R][C] -> planes of A^T    (rows C a multiple of 256, contraction R -> padded to a multiple of 32)
// Through LDS, whole-line reads and 512-byte plane blocks on both sides; rows / columns beyond the matrix are zeros.
// ---------------------------------------------------------------------------------------------------------------
template <bool TRANS>
__global__ __launch_bounds__(256) void th_split_bf16_kernel(const float* __restrict__ a, int R, int C, int k_pad, __bf16* __restrict__ hi,
                                                            __bf16* __restrict__ lo, float* __restrict__ colsum = nullptr) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  auto emit = [&](int orow, int kcol, const f32x4 x, const f32x4 y) {
    const bf16x4 xh = __builtin_convertvector(x, bf16x4), yh = __builtin_convertvector(y, bf16x4);
    const bf16x4 xl = __builtin_convertvector(x - __builtin_convertvector(xh, f32x4), bf16x4);
    const bf16x4 yl = __builtin_convertvector(y - __builtin_convertvector(yh, f32x4), bf16x4);
    const size_t o = t16_index(orow, kcol, k_pad);
    *reinterpret_cast<bf16x8*>(hi + o) = bf16x8{xh[0], xh[1], xh[2], xh[3], yh[0], yh[1], yh[2], yh[3]};
    *reinterpret_cast<bf16x8*>(lo + o) = bf16x8{xl[0], xl[1], xl[2], xl[3], yl[0], yl[1], yl[2], yl[3]};
  };
  if constexpr (!TRANS) {  // grid (rows_pad / 32, ceil(k_pad / 256)): 32 output rows x 256 contraction columns
    __shared__ __attribute__((aligned(16))) float tile[32][260];
    const int o0 = blockIdx.x * 32, k0 = blockIdx.y * 256;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // whole 1 KiB row pieces
      const int row = wv * 8 + i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (o0 + row < R && k0 + lane * 4 < C) v = *reinterpret_cast<const f32x4*>(a + (size_t)(o0 + row) * C + k0 + lane * 4);
      *reinterpret_cast<f32x4*>(&tile[row][lane * 4]) = v;
    }
    __syncthreads();
    const int r = threadIdx.x & 31, cg = threadIdx.x >> 5;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kc = (cg * 4 + j) * 8;
      if (k0 + kc >= k_pad) continue;
      emit(o0 + r, k0 + kc, *reinterpret_cast<const f32x4*>(&tile[r][kc]), *reinterpret_cast<const f32x4*>(&tile[r][kc + 4]));
    }
  } else {  // grid (rows_pad / 128, k_pad / 64): 128 output rows (= input columns: 512-byte row pieces) x 64 contraction (= input rows)
    __shared__ __attribute__((aligned(16))) float tile[128][68];
    const int o0 = blockIdx.x * 128, k0 = blockIdx.y * 64;
    const int cq = threadIdx.x & 31, rg = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = i * 8 + rg;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (k0 + r < R && o0 + cq * 4 < C) v = *reinterpret_cast<const f32x4*>(a + (size_t)(k0 + r) * C + o0 + cq * 4);
      tile[cq * 4 + 0][r] = v[0];
      tile[cq * 4 + 1][r] = v[1];
      tile[cq * 4 + 2][r] = v[2];
      tile[cq * 4 + 3][r] = v[3];
    }
    __syncthreads();
    if (colsum && threadIdx.x < 128 && o0 + threadIdx.x < C) {  // TRANS only: colsum[c] += sum over the rows of a[.][c] — the bias gradient of
      float sacc = 0.f;                                          // the dY this pass reads anyway (one atomic per column and 64-row block)
#pragma unroll 8
      for (int r2 = 0; r2 < 64; ++r2) sacc += tile[threadIdx.x][r2];
      unsafeAtomicAdd(colsum + o0 + threadIdx.x, sacc);
    }
    const int r = threadIdx.x & 31, ch = threadIdx.x >> 5;  // output row within a 32-row tile, 8-wide contraction chunk
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int orow = rt * 32 + r;
      emit(o0 + orow, k0 + ch * 8, *reinterpret_cast<const f32x4*>(&tile[orow][ch * 8]), *reinterpret_cast<const f32x4*>(&tile[orow][ch * 8 + 4]));
    }
  }
}

// column sums of a row-major [M][N] matrix added to out[N] (bias gradients): grid (N / 256, row chunks), float atomics
__global__ __launch_bounds__(256) void th_colsum_kernel(const float* __restrict__ a, int M, int N, int rows_per, float* __restrict__ out) {
  const int n = blockIdx.x * 256 + threadIdx.x, m0 = blockIdx.y * rows_per, m1 = min(M, m0 + rows_per);
  if (n >= N) return;
  float s0 = 0.f, s1 = 0.f;
  int m = m0;
  for (; m + 1 < m1; m += 2) {
    s0 += a[(size_t)m * N + n];
    s1 += a[(size_t)(m + 1) * N + n];
  }
  if (m < m1) s0 += a[(size_t)m * N + n];
  unsafeAtomicAdd(out + n, s0 + s1);
}

// out[m][n] (+)= sum over the ks partial slabs (+ bias[n]) (relu): the second half of a split-K product (fast_gemm)
__global__ __launch_bounds__(256) void th_reduce_parts_kernel(const float* __restrict__ parts, int ks, size_t mn, int N, const float* __restrict__ bias,
                                                              int relu, int accumulate, float* __restrict__ out) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= mn) return;
  f32x4 v = *reinterpret_cast<const f32x4*>(parts + i);
  for (int k = 1; k < ks; ++k) v += *reinterpret_cast<const f32x4*>(parts + (size_t)k * mn + i);
  if (bias) v += *reinterpret_cast<const f32x4*>(bias + (i % (size_t)N));
  if (relu) v = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
  if (accumulate) v += *reinterpret_cast<const f32x4*>(out + i);
  *reinterpret_cast<f32x4*>(out + i) = v;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct Weights {
  // T16 planes (device) + biases; one encoder layer + inter_mlp
  char *qkv_h = nullptr, *qkv_l = nullptr, *out_h = nullptr, *out_l = nullptr, *ff1_h = nullptr, *ff1_l = nullptr, *ff2_h = nullptr,
       *ff2_l = nullptr, *mlp_h = nullptr, *mlp_l = nullptr;
  float *qkv_b = nullptr, *out_b = nullptr, *ff1_b = nullptr, *ff2_b = nullptr, *mlp_b = nullptr, *ln1_g = nullptr, *ln1_b = nullptr,
        *ln2_g = nullptr, *ln2_b = nullptr;
  int out_dim = 0;  // D of inter_mlp (256 coarse, 128 fine)
  // the inter-sentence layer (inter_module.0: d_model 256, 4 heads, dim_feedforward 1024) — coarse model only
  bool has_inter = false;
  InterFusedW fused;  // the inter layer in the encoder's fragment packing (text_inter_fused2_kernel, encode.hip)
  float *i_qkv_b = nullptr, *i_out_b = nullptr, *i_ff1_b = nullptr, *i_ff2_b = nullptr, *i_ln1_g = nullptr, *i_ln1_b = nullptr, *i_ln2_g = nullptr,
        *i_ln2_b = nullptr;
  // workspace
  char* ws = nullptr;
  size_t ws_cap = 0;
  int* flag = nullptr;        // device overflow flag
  int* flag_host = nullptr;   // pinned copy target
  std::vector<void*> owned;
};

static void pack_t16(const float* W, int rows, int rows_pad, int K, std::vector<uint16_t>& hi, std::vector<uint16_t>& lo) {
  hi.assign((size_t)rows_pad * K, 0);
  lo.assign((size_t)rows_pad * K, 0);
  for (int n = 0; n < rows; ++n)
    for (int k = 0; k < K; ++k) {
      const float w = W[(size_t)n * K + k];
      const _Float16 h = (_Float16)w;
      const _Float16 l = (_Float16)(w - (float)h);
      const size_t o = ((size_t)(n >> 5) * (K >> 3) + (k >> 3)) * 256 + (n & 31) * 8 + (k & 7);
      memcpy(&hi[o], &h, 2);
      memcpy(&lo[o], &l, 2);
    }
}

}  // namespace th

using th::Weights;

static const t2l_weight_desc* th_find(const t2l_weight_desc* w, int n, const std::string& name, int64_t numel) {
  for (int i = 0; i < n; ++i)
    if (w[i].name && name == w[i].name) return w[i].numel == numel ? &w[i] : nullptr;
  return nullptr;
}

void free_text_head(t2l_ctx* ctx) {
  Weights* W = (Weights*)ctx->text_head;
  if (!W) return;
  for (void* p : W->owned) (void)hipFree(p);
  if (W->ws) (void)hipFree(W->ws);
  if (W->flag) (void)hipFree(W->flag);
  if (W->flag_host) (void)hipHostFree(W->flag_host);
  delete W;
  ctx->text_head = nullptr;
}

int text_head_load_impl(t2l_ctx* ctx, const t2l_weight_desc* w, int n, const char* prefix) {
  using namespace th;
  const std::string P = prefix ? prefix : "language_encoder.";
  const std::string L0 = P + "intra_module.0.";
  if (th_find(w, n, P + "intra_module.1.linear1.weight", (int64_t)kFF * kDM))
    return fail(ctx, T2L_EINVAL, "t2l_text_head_load_weights: the engine's text head is built for intra_module_num_layers = 1");
  struct Req { const char* key; int64_t numel; };
  const Req req[] = {{"self_attn.in_proj_weight", 3ll * kDM * kDM}, {"self_attn.in_proj_bias", 3 * kDM},
                     {"self_attn.out_proj.weight", (int64_t)kDM * kDM}, {"self_attn.out_proj.bias", kDM},
                     {"linear1.weight", (int64_t)kFF * kDM}, {"linear1.bias", kFF}, {"linear2.weight", (int64_t)kDM * kFF},
                     {"linear2.bias", kDM}, {"norm1.weight", kDM}, {"norm1.bias", kDM}, {"norm2.weight", kDM}, {"norm2.bias", kDM}};
  const t2l_weight_desc* t[12];
  for (int i = 0; i < 12; ++i) {
    t[i] = th_find(w, n, L0 + req[i].key, req[i].numel);
    if (!t[i]) return fail(ctx, T2L_EINVAL, "t2l_text_head_load_weights: missing or mis-sized tensor " + L0 + req[i].key +
                                                " (the engine's text head is built for d_model 1024, dim_feedforward 4096)");
  }
  // inter_mlp = Linear(1024 -> D) + BatchNorm1d, eval mode: folded
  const std::string M0 = P + "inter_mlp.0.";
  int D = 0;
  const t2l_weight_desc* mw = nullptr;
  for (int i = 0; i < n; ++i)
    if (w[i].name && M0 + "0.weight" == w[i].name) { mw = &w[i]; D = (int)(w[i].numel / kDM); }
  if (!mw || D <= 0 || D > 256 || (int64_t)D * kDM != mw->numel || D % 4)
    return fail(ctx, T2L_EINVAL, "t2l_text_head_load_weights: inter_mlp.0.0.weight missing or not [D <= 256][1024]");
  const t2l_weight_desc *mb = th_find(w, n, M0 + "0.bias", D), *bg = th_find(w, n, M0 + "1.weight", D), *bb = th_find(w, n, M0 + "1.bias", D),
                        *bm = th_find(w, n, M0 + "1.running_mean", D), *bv = th_find(w, n, M0 + "1.running_var", D);
  if (!mb || !bg || !bb || !bm || !bv) return fail(ctx, T2L_EINVAL, "t2l_text_head_load_weights: inter_mlp.0 Linear/BatchNorm tensors missing");
  free_text_head(ctx);
  Weights* W = new Weights();
  ctx->text_head = W;
  W->out_dim = D;
  std::vector<uint16_t> hi, lo;
  auto upload = [&](const void* host, size_t bytes, void** dev) -> int {
    T2L_HIP(ctx, hipMalloc(dev, bytes));
    W->owned.push_back(*dev);
    T2L_HIP(ctx, hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
    return T2L_OK;
  };
  auto planes = [&](const float* src, int rows, int K, char** dh, char** dl) -> int {
    const int rows_pad = (rows + kTile - 1) / kTile * kTile;
    pack_t16(src, rows, rows_pad, K, hi, lo);
    int rc = upload(hi.data(), hi.size() * 2, (void**)dh);
    return rc ? rc : upload(lo.data(), lo.size() * 2, (void**)dl);
  };
  auto vec = [&](const float* src, int nn, int pad, float** d) -> int {
    std::vector<float> v(pad, 0.f);
    memcpy(v.data(), src, sizeof(float) * nn);
    return upload(v.data(), sizeof(float) * pad, (void**)d);
  };
  int rc;
  if ((rc = planes(t[0]->data, 3 * kDM, kDM, &W->qkv_h, &W->qkv_l)) || (rc = vec(t[1]->data, 3 * kDM, 3 * kDM, &W->qkv_b)) ||
      (rc = planes(t[2]->data, kDM, kDM, &W->out_h, &W->out_l)) || (rc = vec(t[3]->data, kDM, kDM, &W->out_b)) ||
      (rc = planes(t[4]->data, kFF, kDM, &W->ff1_h, &W->ff1_l)) || (rc = vec(t[5]->data, kFF, kFF, &W->ff1_b)) ||
      (rc = planes(t[6]->data, kDM, kFF, &W->ff2_h, &W->ff2_l)) || (rc = vec(t[7]->data, kDM, kDM, &W->ff2_b)) ||
      (rc = vec(t[8]->data, kDM, kDM, &W->ln1_g)) || (rc = vec(t[9]->data, kDM, kDM, &W->ln1_b)) ||
      (rc = vec(t[10]->data, kDM, kDM, &W->ln2_g)) || (rc = vec(t[11]->data, kDM, kDM, &W->ln2_b)))
    return rc;
  {  // y = ((x W^T + b) - mean) * gamma / sqrt(var + eps) + beta
    std::vector<float> wf((size_t)D * kDM), bf(kTile, 0.f);
    for (int o = 0; o < D; ++o) {
      const float sc = bg->data[o] / sqrtf(bv->data[o] + 1e-5f);
      for (int k = 0; k < kDM; ++k) wf[(size_t)o * kDM + k] = mw->data[(size_t)o * kDM + k] * sc;
      bf[o] = (mb->data[o] - bm->data[o]) * sc + bb->data[o];
    }
    if ((rc = planes(wf.data(), D, kDM, &W->mlp_h, &W->mlp_l)) || (rc = upload(bf.data(), sizeof(float) * kTile, (void**)&W->mlp_b))) return rc;
  }
  {  // inter_module.0 (language_encoder.py:101,143-144), when present and of the published shape: D = 256, 4 heads, ff 1024
    constexpr int ID = 256, IF = 1024;
    const std::string I0 = P + "inter_module.0.";
    const Req ireq[] = {{"self_attn.in_proj_weight", 3ll * ID * ID}, {"self_attn.in_proj_bias", 3 * ID}, {"self_attn.out_proj.weight", (int64_t)ID * ID},
                        {"self_attn.out_proj.bias", ID}, {"linear1.weight", (int64_t)IF * ID}, {"linear1.bias", IF}, {"linear2.weight", (int64_t)ID * IF},
                        {"linear2.bias", ID}, {"norm1.weight", ID}, {"norm1.bias", ID}, {"norm2.weight", ID}, {"norm2.bias", ID}};
    const t2l_weight_desc* it[12];
    bool all = D == ID && !th_find(w, n, P + "inter_module.1.linear1.weight", (int64_t)IF * ID);
    for (int i = 0; i < 12 && all; ++i) all = (it[i] = th_find(w, n, I0 + ireq[i].key, ireq[i].numel)) != nullptr;
    if (all) {
      if ((rc = vec(it[1]->data, 3 * ID, 3 * ID, &W->i_qkv_b)) || (rc = vec(it[3]->data, ID, ID, &W->i_out_b)) ||
          (rc = vec(it[5]->data, IF, IF, &W->i_ff1_b)) || (rc = vec(it[7]->data, ID, ID, &W->i_ff2_b)) ||
          (rc = vec(it[8]->data, ID, ID, &W->i_ln1_g)) || (rc = vec(it[9]->data, ID, ID, &W->i_ln1_b)) ||
          (rc = vec(it[10]->data, ID, ID, &W->i_ln2_g)) || (rc = vec(it[11]->data, ID, ID, &W->i_ln2_b)))
        return rc;
      {  // the four matrices as split-f16 MFMA fragments (mfma_h3.h packing; 1.5 MB)
        auto frag = [&](const float* Wm, int rows, int cols, const uint4** dst) -> int {
          const std::vector<float> pk = pack_split_f16(Wm, nullptr, rows, cols, cols);
          void* d = nullptr;
          const int r = upload(pk.data(), pk.size() * sizeof(float), &d);
          *dst = reinterpret_cast<const uint4*>(d);
          return r;
        };
        if ((rc = frag(it[0]->data, 3 * ID, ID, &W->fused.in_hp)) || (rc = frag(it[2]->data, ID, ID, &W->fused.out_hp)) ||
            (rc = frag(it[4]->data, IF, ID, &W->fused.ff1_hp)) || (rc = frag(it[6]->data, ID, IF, &W->fused.ff2_hp)))
          return rc;
        W->fused.in_b = W->i_qkv_b; W->fused.out_b = W->i_out_b; W->fused.ff1_b = W->i_ff1_b; W->fused.ff2_b = W->i_ff2_b;
        W->fused.ln1_w = W->i_ln1_g; W->fused.ln1_b = W->i_ln1_b; W->fused.ln2_w = W->i_ln2_g; W->fused.ln2_b = W->i_ln2_b;
      }
      W->has_inter = true;
    }
  }
  T2L_HIP(ctx, hipMalloc(&W->flag, sizeof(int)));
  T2L_HIP(ctx, hipMemset(W->flag, 0, sizeof(int)));
  T2L_HIP(ctx, hipHostMalloc((void**)&W->flag_host, sizeof(int), hipHostMallocDefault));
  *W->flag_host = 0;
  static PerDeviceOnce attr_done;
  if (attr_done.need(ctx->device)) {
    const int lds = kSlots * kSlotBytes;
#define T2L_TH_ATTR(E, S) T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&th_gemm_kernel<E, S>), hipFuncAttributeMaxDynamicSharedMemorySize, lds))
    T2L_TH_ATTR(kEpiT32, false); T2L_TH_ATTR(kEpiT32, true); T2L_TH_ATTR(kEpiResidT32, false); T2L_TH_ATTR(kEpiResidT32, true);
    T2L_TH_ATTR(kEpiReluT16, false); T2L_TH_ATTR(kEpiReluT16, true); T2L_TH_ATTR(kEpiRowMajor, false); T2L_TH_ATTR(kEpiRowMajor, true);
#undef T2L_TH_ATTR
    T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&th_ln_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 1028 * 4));
    T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&th_ln_kernel<true, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 260 * 4));
    attr_done.mark(ctx->device);
  }
  return T2L_OK;
}

template <int EPI>
static void th_launch_gemm(bool single, const th::GemmArgs& a, hipStream_t s) {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu < 8) n_cu = 256;
  }
  const int n_vb = (a.m_tiles + 7) / 8 * 8 * a.n_tiles;
  const int grid = min(n_cu / 8 * 8, n_vb);  // persistent: one workgroup per CU, a multiple of the 8 XCDs
  const int lds = th::kSlots * th::kSlotBytes;
  if (single) hipLaunchKernelGGL((th::th_gemm_kernel<EPI, true>), dim3(grid), dim3(512), lds, s, a);
  else hipLaunchKernelGGL((th::th_gemm_kernel<EPI, false>), dim3(grid), dim3(512), lds, s, a);
}

int text_head_impl(t2l_ctx* ctx, const float* hidden, int n_sent, int L, float* out, int32_t* overflow, hipStream_t s) {
  using namespace th;
  Weights* W = (Weights*)ctx->text_head;
  if (!W) return fail(ctx, T2L_ESTATE, "t2l_text_head: call t2l_text_head_load_weights first");
  if (!hidden || !out) return fail(ctx, T2L_EINVAL, "t2l_text_head: null buffer");
  if (n_sent <= 0) return n_sent == 0 ? T2L_OK : fail(ctx, T2L_EINVAL, "t2l_text_head: n_sentences < 0");
  if (L < 1 || L > kMaxL) return fail(ctx, T2L_EINVAL, "t2l_text_head: need 1 <= n_tokens <= 32");
  const bool single = ctx->encoder_f16 != 0;
  // sentences per pass: ~16 k token rows keep every intermediate of a pass inside the 256 MB Infinity Cache and make every
  // GEMM grid a multiple of the 256 CUs (64 m tiles x 12 / 4 / 16 / 4 n tiles)
  const int rows_target = ctx->text_head_rows > 0 ? ctx->text_head_rows : 16384;
  const int spc = max(1, min(n_sent, rows_target / L));
  const int m_cap = (spc * L + kTile - 1) / kTile * kTile;
  const int n_chunks = (n_sent + spc - 1) / spc;
  const int sp_cap = (n_sent + kTile - 1) / kTile * kTile;  // pooled rows (all sentences), padded
  // workspace: X planes, QKV (T32), O planes, Y (T32), X1 planes, H planes; pooled f32 + its planes
  const size_t plane = (size_t)m_cap * kDM * 2;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t o_xh = take(plane), o_xl = take(plane), o_qkv = take((size_t)m_cap * 3 * kDM * 4), o_oh = take(plane), o_ol = take(plane),
               o_y = take((size_t)m_cap * kDM * 4), o_1h = take(plane), o_1l = take(plane), o_hh = take(plane * 4), o_hl = take(plane * 4),
               o_pool = take((size_t)sp_cap * kDM * 4), o_ph = take((size_t)sp_cap * kDM * 2), o_pl = take((size_t)sp_cap * kDM * 2);
  if (W->ws_cap < off) {
    if (W->ws) T2L_HIP(ctx, hipFree(W->ws));
    W->ws = nullptr;
    W->ws_cap = 0;
    T2L_HIP(ctx, hipMalloc(&W->ws, off));
    W->ws_cap = off;
  }
  char* ws = W->ws;
  T2L_HIP(ctx, hipMemsetAsync(W->flag, 0, sizeof(int), s));
  T2L_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)(ws + o_pool), (int)0xFF800000u, (size_t)sp_cap * kDM, s));  // -inf
  event_begin(ctx, "text_head", s);
  for (int ch = 0; ch < n_chunks; ++ch) {
    const int s0 = ch * spc, ns = min(spc, n_sent - s0);
    const int M = ns * L, m_tiles = (M + kTile - 1) / kTile, m_pad = m_tiles * kTile;
    const float* x = hidden + (size_t)s0 * L * kDM;
    hipLaunchKernelGGL(th_split_kernel, dim3(m_pad / 32, kDM / 256), dim3(256), 0, s, x, M, kDM, (_Float16*)(ws + o_xh), (_Float16*)(ws + o_xl), W->flag);
    GemmArgs g{};
    g.flag = W->flag;
    g.m_tiles = m_tiles;
    g.M = M;
    // q | k | v = x W_in^T + b_in  -> T32
    g.wh = W->qkv_h; g.wl = W->qkv_l; g.xh = ws + o_xh; g.xl = ws + o_xl; g.bias = W->qkv_b; g.out0 = ws + o_qkv;
    g.n_tiles = 3 * kDM / kTile; g.K = kDM; g.N = 3 * kDM; g.n_real = 3 * kDM;
    th_launch_gemm<kEpiT32>(single, g, s);
    {
      const int G = 32 / L, n_groups = (ns + G - 1) / G;
      hipLaunchKernelGGL(th_attn_kernel<1024>, dim3(n_groups), dim3(256), 0, s, (const float*)(ws + o_qkv), ns, L, n_groups, (_Float16*)(ws + o_oh),
                         (_Float16*)(ws + o_ol), W->flag);
    }
    // y = x + o W_out^T + b_out -> T32; x1 = LayerNorm1(y) -> planes
    g.wh = W->out_h; g.wl = W->out_l; g.xh = ws + o_oh; g.xl = ws + o_ol; g.bias = W->out_b; g.rh = ws + o_xh; g.rl = ws + o_xl; g.out0 = ws + o_y;
    g.n_tiles = kDM / kTile; g.K = kDM; g.N = kDM; g.n_real = kDM;
    th_launch_gemm<kEpiResidT32>(single, g, s);
    hipLaunchKernelGGL(th_ln_kernel<false>, dim3(m_pad / 32), dim3(256), 8 * 32 * 4, s, (const float*)(ws + o_y), W->ln1_g, W->ln1_b, M, L,
                       (_Float16*)(ws + o_1h), (_Float16*)(ws + o_1l), (float*)nullptr, W->flag);
    // h = relu(x1 W_1^T + b_1) -> planes; y2 = x1 + h W_2^T + b_2 -> T32
    g.wh = W->ff1_h; g.wl = W->ff1_l; g.xh = ws + o_1h; g.xl = ws + o_1l; g.bias = W->ff1_b; g.out0 = ws + o_hh; g.out1 = ws + o_hl;
    g.n_tiles = kFF / kTile; g.K = kDM; g.N = kFF; g.n_real = kFF;
    th_launch_gemm<kEpiReluT16>(single, g, s);
    g.wh = W->ff2_h; g.wl = W->ff2_l; g.xh = ws + o_hh; g.xl = ws + o_hl; g.bias = W->ff2_b; g.rh = ws + o_1h; g.rl = ws + o_1l; g.out0 = ws + o_y;
    g.n_tiles = kDM / kTile; g.K = kFF; g.N = kDM; g.n_real = kDM;
    th_launch_gemm<kEpiResidT32>(single, g, s);
    // LayerNorm2 + max over the sentence's tokens -> pooled rows [s0, s0 + ns)
    hipLaunchKernelGGL(th_ln_kernel<true>, dim3(m_pad / 32), dim3(256), 32 * 1028 * 4, s, (const float*)(ws + o_y), W->ln2_g, W->ln2_b, M, L,
                       (_Float16*)nullptr, (_Float16*)nullptr, (float*)(ws + o_pool) + (size_t)s0 * kDM, W->flag);
  }
  // inter_mlp (Linear + folded BatchNorm) on the pooled sentence vectors -> out [n_sent][D] row-major
  hipLaunchKernelGGL(th_split_kernel, dim3(sp_cap / 32, kDM / 256), dim3(256), 0, s, (const float*)(ws + o_pool), n_sent, kDM, (_Float16*)(ws + o_ph),
                     (_Float16*)(ws + o_pl), W->flag);
  {
    GemmArgs g{};
    g.flag = W->flag;
    g.m_tiles = sp_cap / kTile; g.M = n_sent;
    g.wh = W->mlp_h; g.wl = W->mlp_l; g.xh = ws + o_ph; g.xl = ws + o_pl; g.bias = W->mlp_b; g.out0 = out;
    g.n_tiles = 1; g.K = kDM; g.N = kTile; g.n_real = W->out_dim;
    th_launch_gemm<kEpiRowMajor>(single, g, s);
  }
  event_end(ctx, "text_head", s);
  if (overflow) T2L_HIP(ctx, hipMemcpyAsync(overflow, W->flag, sizeof(int), hipMemcpyDeviceToDevice, s));
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

// The inter-sentence half of the head (models/language_encoder.py:137-147): sent [n_desc * S][256] row-major (what t2l_text_head
// returns, description-major) -> view [n_desc, S, 256] -> x += TransformerEncoderLayer(256, 4 heads, ff 1024)(x) over the S
// sentences of every description -> max over the S sentences -> out [n_desc][256]. The same kernels as the token layer with
// d_model 256: rows = sentences, "tokens per sentence" = S.
int text_inter_impl(t2l_ctx* ctx, const float* sent, int n_desc, int S, float* out, int32_t* overflow, hipStream_t s) {
  using namespace th;
  constexpr int ID = 256, IF = 1024;
  Weights* W = (Weights*)ctx->text_head;
  if (!W) return fail(ctx, T2L_ESTATE, "t2l_text_inter: call t2l_text_head_load_weights first");
  if (!W->has_inter) return fail(ctx, T2L_ESTATE, "t2l_text_inter: the loaded head has no inter_module.0 of the published shape (256 / 4 heads / 1024)");
  if (!sent || !out) return fail(ctx, T2L_EINVAL, "t2l_text_inter: null buffer");
  if (n_desc <= 0) return n_desc == 0 ? T2L_OK : fail(ctx, T2L_EINVAL, "t2l_text_inter: n_descriptions < 0");
  if (S < 1 || S > kMaxL) return fail(ctx, T2L_EINVAL, "t2l_text_inter: need 1 <= sentences per description <= 32");
  const bool single = ctx->encoder_f16 != 0;
  // ONE launch: two tiles of floor(32 / S) descriptions per eight-wave workgroup, everything in LDS (encode.hip:
  // text_inter_fused2_kernel). Rounds 3-4 also shipped a chain of tiled GEMM / attention / LayerNorm launches (0.208 ms for 4,096
  // descriptions x 6 sentences) and a one-tile form of this launch; both measured slower than this one (0.179 ms) and were removed
  // in round 5.
  T2L_HIP(ctx, hipMemsetAsync(W->flag, 0, sizeof(int), s));
  event_begin(ctx, "text_inter", s);
  const int rc_ = text_inter_fused_launch(ctx, W->fused, single, sent, n_desc, S, out, W->flag, s);
  event_end(ctx, "text_inter", s);
  if (rc_ != T2L_OK) return rc_;
  if (overflow) T2L_HIP(ctx, hipMemcpyAsync(overflow, W->flag, sizeof(int), hipMemcpyDeviceToDevice, s));
  return T2L_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// fast_gemm: OUT[Mo][No] (+)= A B^T (+ bias) (relu) for the TRAINING step of the text head (train.hip), on the tiled LDS-ring GEMM
// above with bf16 planes. A: row-major [Mo][Kc], or (a_trans) [Kc][Mo]; B: row-major [No][Kc], or (b_trans) [Kc][No]; f32
// row-major output. Covers the three products of a Linear: Y = X W^T + b (plain), dX (+)= dY W (b_trans), dW += dY^T X (a_trans,
// b_trans, accumulate). single: one bf16 product per operand pair (text_train_bf16 = 1); else split-bf16 (three). No must be a
// multiple of 256, the contraction is zero-padded to a multiple of 32. Workspace: the operand planes, grown on demand.
// ---------------------------------------------------------------------------------------------------------------
int fast_gemm(t2l_ctx* ctx, const float* A, bool a_trans, const float* B, bool b_trans, const float* bias, float* out, int Mo, int No, int Kc,
              int relu, int accumulate, bool single, hipStream_t s, float* a_colsum) {
  using namespace th;
  if (No % kTile || Mo < 1 || Kc < 1) return fail(ctx, T2L_EINVAL, "fast_gemm: unsupported shape");
  const int k_pad = (Kc + 63) / 64 * 64, m_pad = (Mo + kTile - 1) / kTile * kTile;  // (64: the transposed split's tile, and an even k-step count)
  const size_t pa = (size_t)m_pad * k_pad * 2, pb = (size_t)No * k_pad * 2;
  // few output tiles against a long contraction (dW over thousands of rows: 16 .. 64 tiles for 256 CUs; 1024-wide outputs: 96): the
  // contraction is cut into ks jobs per tile, every job stores its partial tile into its own slab (plain stores — float atomics
  // into the output were measured 2x SLOWER than no split at all) and one pass adds the slabs up (+ bias, ReLU, accumulate)
  const int m_tiles = m_pad / kTile, n_tiles = No / kTile, kt = k_pad >> 4;
  int ks = 1;
  while (ks < 8 && m_tiles * n_tiles * ks < 192 && kt % (ks * 4) == 0 && kt / (ks * 2) >= 8) ks *= 2;
  const size_t part_bytes = ks > 1 ? sizeof(float) * (size_t)ks * Mo * No : 0;
  const size_t need = 2 * pa + 2 * pb + part_bytes + 256;
  if (ctx->fast_ws_cap < need) {  // (the context's own scratch: calls of one context are serialised by its caller)
    if (ctx->fast_ws) {
      T2L_HIP(ctx, hipStreamSynchronize(s));
      T2L_HIP(ctx, hipFree(ctx->fast_ws));
    }
    ctx->fast_ws = nullptr;
    ctx->fast_ws_cap = 0;
    T2L_HIP(ctx, hipMalloc(&ctx->fast_ws, need + need / 4));
    ctx->fast_ws_cap = need + need / 4;
  }
  if (!ctx->fast_zero) {  // a zero bias vector for products without one (allocated once)
    T2L_HIP(ctx, hipMalloc(&ctx->fast_zero, sizeof(float) * 16384));
    T2L_HIP(ctx, hipMemset(ctx->fast_zero, 0, sizeof(float) * 16384));
  }
  if (No > 16384) return fail(ctx, T2L_EINVAL, "fast_gemm: more than 16384 output columns");
  char *ah = (char*)ctx->fast_ws, *al = ah + pa, *bh = al + pa, *bl = bh + pb;
  float* parts = reinterpret_cast<float*>(bl + pb);
  static PerDeviceOnce once;
  if (once.need(ctx->device)) {
    const int lds = kSlots * kSlotBytes;
    T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&th_gemm_kernel<kEpiRowMajor, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&th_gemm_kernel<kEpiRowMajor, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    once.mark(ctx->device);
  }
  const dim3 ga(m_pad / 32, (k_pad + 255) / 256), gb(No / 32, (k_pad + 255) / 256), gat(m_pad / 128, k_pad / 64), gbt(No / 128, k_pad / 64);
  if (a_trans) hipLaunchKernelGGL(th_split_bf16_kernel<true>, gat, dim3(256), 0, s, A, Kc, Mo, k_pad, (__bf16*)ah, (__bf16*)al, a_colsum);
  else hipLaunchKernelGGL(th_split_bf16_kernel<false>, ga, dim3(256), 0, s, A, Mo, Kc, k_pad, (__bf16*)ah, (__bf16*)al, (float*)nullptr);
  if (b_trans) hipLaunchKernelGGL(th_split_bf16_kernel<true>, gbt, dim3(256), 0, s, B, Kc, No, k_pad, (__bf16*)bh, (__bf16*)bl, (float*)nullptr);
  else hipLaunchKernelGGL(th_split_bf16_kernel<false>, gb, dim3(256), 0, s, B, No, Kc, k_pad, (__bf16*)bh, (__bf16*)bl, (float*)nullptr);
  GemmArgs g{};
  g.wh = bh; g.wl = bl; g.xh = ah; g.xl = al; g.bias = bias ? bias : ctx->fast_zero; g.out0 = ks > 1 ? (void*)parts : (void*)out;
  g.m_tiles = m_tiles; g.n_tiles = n_tiles; g.K = k_pad; g.N = No; g.M = Mo; g.n_real = No;
  g.relu = ks > 1 ? 0 : relu; g.accumulate = ks > 1 ? 3 : accumulate; g.ksplit = ks;
  static int n_cu = 0;
  if (!n_cu) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu < 8) n_cu = 256;
  }
  const int n_vb = (g.m_tiles + 7) / 8 * 8 * g.n_tiles * ks;
  const int grid = min(n_cu / 8 * 8, n_vb);
  const int lds = kSlots * kSlotBytes;
  if (single) hipLaunchKernelGGL((th_gemm_kernel<kEpiRowMajor, true, true>), dim3(grid), dim3(512), lds, s, g);
  else hipLaunchKernelGGL((th_gemm_kernel<kEpiRowMajor, false, true>), dim3(grid), dim3(512), lds, s, g);
  if (ks > 1) {
    const size_t mn = (size_t)Mo * No;
    hipLaunchKernelGGL(th_reduce_parts_kernel, dim3((unsigned)((mn / 4 + 255) / 256)), dim3(256), 0, s, (const float*)parts, ks, mn, No, bias, relu, accumulate, out);
  }
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

void fast_colsum(const float* a, int M, int N, float* out, hipStream_t s) {
  const int rows_per = 256;
  hipLaunchKernelGGL(th::th_colsum_kernel, dim3((N + 255) / 256, (M + rows_per - 1) / rows_per), dim3(256), 0, s, a, M, N, rows_per, out);
}

}  // namespace t2l
