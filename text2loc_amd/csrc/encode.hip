// Fused per-cell encoder: ObjectEncoder.forward (models/object_encoder.py:66-153, eval mode) +
// CellRetrievalNetwork.encode_objects (models/cell_retrieval.py:65-110) as ONE kernel, one workgroup
// (4 waves) per cell, every intermediate resident in LDS.
//
//   per object (<= 28 kept, cell_retrieval.py:94-98):
//     class  = normalize(class_embedding[idx])           | normalize(mlp_pointnet(features2))
//     color  = normalize(color_embedding[idx])           | normalize(color_encoder(mean rgb))
//     pos    = normalize(pos_encoder(center)),  num = normalize(num_encoder((n-mean)/std))
//     merged = relu(BN(Linear(cat)))  -> normalize                       (trailing ReLU, language_encoder.py:15)
//   x[28,256] zero padded -> 2 x TransformerEncoderLayer (post-norm, ReLU, NO padding mask: zero slots are
//   attended to and attend) -> max over the 28 slots -> normalize.
//
// The big contractions (feature merge, q/k/v, out_proj, feed-forward: 97 % of the FLOPs) run as SPLIT-f16 on
// v_mfma_f32_32x32x16_f16: every f32 value a is used as hi + lo with hi = f16(a), lo = f16(a - hi) (22 significand bits;
// gfx950's MFMA honours f16 denormals, tools/ probe) and a product is hi*hi + hi*lo + lo*hi with f32 accumulation —
// ~5e-7 relative, well inside the 1e-3 parity budget (measured 2e-6 on the goldens) at 1/5 of the matrix-pipe time of
// v_mfma_f32_32x32x2_f32. Activations are split on the fly (20 VALU per 8 values), weights at load time. f16 overflows at
// 65504: t2l_load_weights bounds every activation that enters a split GEMM from the weights (LayerNorm gain/bias,
// row norms) and keeps the all-f32 kernel for models that could exceed it (option encoder_f32 forces it).
// The attention core (S = K Q^T, P V) and the small MLPs stay on the f32 MFMA.
// M = 32 rows = the 28 slots + 4 dead rows (masked out of the softmax keys and the max-pool).
// Weights are BN-folded and re-laid out on the host into MFMA B-fragment order
// [n_tile][k_step][lane][4] so every wave-level weight load is one coalesced 1 KiB line.
// The MFMA's k-sum is order-free, so lane (col, half) owns k in [half*K/2, (half+1)*K/2): both operands
// are read as contiguous float4 (LDS rows padded by 4 floats -> conflict-free ds_read_b128).
#include <math.h>
#include <string.h>

#include "t2l_internal.h"
#include "mfma_h3.h"

#ifndef T2L_ENC_UNROLL
#define T2L_ENC_UNROLL 4
#endif
// dev experiment (make exp_enc EXPFLAG=-DT2L_EXP_HOTW, tools/hotw_probe.py; WRONG results, timing only): every weight fragment of a tile
// comes from its first two k-steps — the packed-weight stream out of the L2 disappears, the instruction stream stays. How much of the
// fused kernels' time is that stream?
#ifdef T2L_EXP_HOTW
#define T2L_WSTEP(s) ((s) & 1)
#else
#define T2L_WSTEP(s) (s)
#endif

namespace t2l {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kLdX = kD + 4;        // 260: row stride of every 256-wide LDS buffer
constexpr int kLdH = 64 + 4;        // 68   (hidden layer of the small MLPs)
constexpr int kXFloats = kSP * kLdX;
constexpr float kNumMean = 1826.6844940968194f;  // models/object_encoder.py:43
constexpr float kNumStd = 2516.8905096993817f;   // models/object_encoder.py:44

struct SmallMlp {  // get_mlp([in, 64, 256]) with BN folded (language_encoder.py:16-41)
  const float* w1;   // [64][in]
  const float* b1;   // [64]
  const float4* w2p; // packed [8 tiles][8][64] float4   (N=256, K=64)
  const float* b2;   // [256]
};

struct LayerW {
  const float4 *in_wp, *out_wp, *ff1_wp, *ff2_wp;
  const uint4 *in_hp, *out_hp, *ff1_hp, *ff2_hp;  // the same matrices as split-f16 fragments (pack_h)
  const float *in_b, *out_b, *ff1_b, *ff2_b, *ln1_w, *ln1_b, *ln2_w, *ln2_b;
};

struct EncParams {
  const float* class_tab;  // [n_class][256] rows already L2-normalised
  const float* color_tab;  // [n_color][256]
  int n_class, n_color;
  SmallMlp pos, color, num;
  const float4* pn_wp;     // mlp_pointnet packed (N=256,K=256)
  const uint4* pn_hp;
  const float* pn_b;
  const float4* merge_wp;  // nfeat consecutive packings (N=256, K=256), one per 256-wide feature slot
  const uint4* merge_hp;
  const float* merge_b;
  LayerW layer[4];
  int num_layers;
  int class_embed, color_embed, use_class, use_color, use_pos, use_num, nfeat;
  int split_ok;  // every activation entering a split-f16 GEMM is provably below the f16 range for these weights
};

struct EncoderWeights {
  EncParams p;
  float* blob = nullptr;
};

// ---- device helpers ----------------------------------------------------------------------------
// all-reduce sum over the 64 lanes on the VALU (DPP + v_permlane swaps): __shfl_xor lowers to ds_bpermute_b32 — six dependent
// LDS round trips per sum
template <int CTRL>
__device__ __forceinline__ float wave_sum_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += wave_sum_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
  v += wave_sum_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
  v += wave_sum_dpp<0x141>(v);  // row_half_mirror
  v += wave_sum_dpp<0x140>(v);  // row_mirror
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

// out[32][N] = A[32][K] @ W^T ; the 4 waves split N in 32-column tiles, two tiles at a time per wave
// (tiles w+8p and w+8p+4) sharing the A fragments. epi(t, r, row, col, value) is called for every element
// (t = which tile of the pair, r = accumulator register; both compile-time after unrolling).
template <typename Epi>
__device__ __forceinline__ void gemm32(const float* __restrict__ A, int lda, int K, const float4* __restrict__ Wp,
                                       int N, int wave, int lane, Epi epi) {
  const int col = lane & 31, half = lane >> 5;
  const int qn = K >> 3;
  const float* arow = A + col * lda + half * (K >> 1);
  for (int p = 0; p < (N >> 8); ++p) {
    const int nt0 = wave + 8 * p, nt1 = nt0 + 4;
    const float4* w0 = Wp + (size_t)nt0 * qn * 64 + lane;
    const float4* w1 = Wp + (size_t)nt1 * qn * 64 + lane;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc0[r] = 0.f;
      acc1[r] = 0.f;
    }
#pragma unroll 4
    for (int q = 0; q < qn; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(arow + 4 * q);
      const float4 b0 = w0[q * 64];
      const float4 b1 = w1[q * 64];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      epi(0, r, row, nt0 * 32 + col, acc0[r]);
      epi(1, r, row, nt1 * 32 + col, acc1[r]);
    }
  }
}

// acc0 += A * W0^T, acc1 += A * W1^T over qn packed k-steps (8 k each): A = this lane's LDS row half (arow), W0 / W1 =
// packed weight tiles already offset to their first step and to this lane. The caller owns initialisation and epilogue.
__device__ __forceinline__ void mm_pair(const float* __restrict__ arow, int qn, const float4* __restrict__ w0,
                                        const float4* __restrict__ w1, f32x16& acc0, f32x16& acc1) {
#pragma unroll 4
  for (int q = 0; q < qn; ++q) {
    const float4 a = *reinterpret_cast<const float4*>(arow + 4 * q);
    const float4 b0 = w0[q * 64];
    const float4 b1 = w1[q * 64];
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
  }
}

// ---- split-f16 forms of gemm32 / mm_pair (fragments and packing: mfma_h3.h)
// acc0 += A * W0^T, acc1 += A * W1^T over `steps` k-steps of 16: arow = this lane's LDS row half, w0 / w1 = the two weight
// tiles already offset to their first step and to this lane (2 uint4 per lane and step, 128 uint4 per step)
template <bool SG = false>
__device__ __forceinline__ void mm_pair_h(const float* __restrict__ arow, int steps, const uint4* __restrict__ w0,
                                          const uint4* __restrict__ w1, f32x16& acc0, f32x16& acc1) {
#pragma unroll T2L_ENC_UNROLL
  for (int s = 0; s < steps; ++s) {
    const HFrag a = split_h<SG>(arow + 8 * s);
    const HFrag b0 = load_h1<SG>(w0 + T2L_WSTEP(s) * 128), b1 = load_h1<SG>(w1 + T2L_WSTEP(s) * 128);
    mfma_h3<SG>(acc0, a, b0);
    mfma_h3<SG>(acc1, a, b1);
  }
}
template <bool SG = false, typename Epi>
__device__ __forceinline__ void gemm32_h(const float* __restrict__ A, int lda, int K, const uint4* __restrict__ Wp, int N,
                                         int wave, int lane, Epi epi) {
  const int col = lane & 31, half = lane >> 5;
  const int steps = K >> 4;
  const float* arow = A + col * lda + half * (K >> 1);
  for (int p = 0; p < (N >> 8); ++p) {
    const int nt0 = wave + 8 * p, nt1 = nt0 + 4;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc0[r] = 0.f;
      acc1[r] = 0.f;
    }
    mm_pair_h<SG>(arow, steps, Wp + ((size_t)nt0 * steps * 64 + lane) * 2, Wp + ((size_t)nt1 * steps * 64 + lane) * 2, acc0, acc1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      epi(0, r, row, nt0 * 32 + col, acc0[r]);
      epi(1, r, row, nt1 * 32 + col, acc1[r]);
    }
  }
}

// F.normalize over 256 columns of `rows` rows starting at buf (row stride ld); rows >= nvalid are zeroed.
__device__ __forceinline__ void normalize_rows(float* buf, int ld, int nvalid, int wave, int lane) {
  for (int i = wave; i < kSP; i += 4) {
    float4* p = reinterpret_cast<float4*>(buf + i * ld) + lane;
    float4 v = *p;
    if (i < nvalid) {
      const float ss = wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
      const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
      v.x *= inv;
      v.y *= inv;
      v.z *= inv;
      v.w *= inv;
    } else {
      v = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    *p = v;
  }
}

// torch.nn.LayerNorm(256, eps=1e-5) in place over the 32 rows of x
__device__ __forceinline__ void layer_norm_rows(float* x, const float* __restrict__ w, const float* __restrict__ b,
                                                int wave, int lane) {
  const float4 wv = reinterpret_cast<const float4*>(w)[lane];
  const float4 bv = reinterpret_cast<const float4*>(b)[lane];
  for (int i = wave; i < kSP; i += 4) {
    float4* p = reinterpret_cast<float4*>(x + i * kLdX) + lane;
    float4 v = *p;
    const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.f / kD);
    v.x -= mean;
    v.y -= mean;
    v.z -= mean;
    v.w -= mean;
    const float var = wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w) * (1.f / kD);
    const float inv = 1.f / sqrtf(var + 1e-5f);
    v.x = v.x * inv * wv.x + bv.x;
    v.y = v.y * inv * wv.y + bv.y;
    v.z = v.z * inv * wv.z + bv.z;
    v.w = v.w * inv * wv.w + bv.w;
    *p = v;
  }
}

// One feature branch through get_mlp([in,64,256]): hidden layer on the VALU (K = 1 or 3), 64->256 on MFMA.
template <int IN>
__device__ __forceinline__ void small_mlp(const SmallMlp& m, const float* __restrict__ in /* [nobj][IN] global */,
                                          bool is_num, int nobj, float* hbuf, float* dst /* [32][kLdX] */, int tid,
                                          int wave, int lane) {
  // hidden: 32 objects x 64 units, 8 per thread
  for (int e = tid; e < kSP * 64; e += 256) {
    const int o = e >> 6, u = e & 63;
    float acc = 0.f;
    if (o < nobj) {
      acc = m.b1[u];
#pragma unroll
      for (int k = 0; k < IN; ++k) {
        float v = in[o * IN + k];
        if (is_num) v = (v - kNumMean) / kNumStd;  // object_encoder.py:143
        acc += m.w1[u * IN + k] * v;
      }
      acc = fmaxf(acc, 0.f);
    }
    hbuf[o * kLdH + u] = acc;
  }
  __syncthreads();
  const float* b2 = m.b2;
  gemm32(hbuf, kLdH, 64, m.w2p, kD, wave, lane,
         [&](int, int, int row, int col, float v) { dst[row * kLdX + col] = fmaxf(v + b2[col], 0.f); });
  __syncthreads();
  normalize_rows(dst, kLdX, nobj, wave, lane);
}

// LDS: x [32][260] + buf [32][260] = 66.6 KB per cell, one cell per workgroup, so TWO workgroups (cells) are in flight per CU:
// while one sits in a barrier, a LayerNorm or a softmax, the other keeps the MFMA pipe busy.
// (Measured and rejected in round 2: TWO cells per workgroup sharing every weight fragment in registers — 133 KB, one
// workgroup per CU, bit-identical output — 5.35 ms against 3.70 ms for 11,259 cells: what it saves on the weight stream it loses
// twice over by leaving each SIMD a single wave to hide the L2 latency of that stream.)
// What makes a cell fit 66.6 KB:
//  * the concatenated features never exist: every 256-wide slot is produced in `buf` and immediately contracted with its
//    256-column slice of the merge weight into register accumulators;
//  * q, k, v never touch LDS: head h = wave h computes q_h^T and k_h^T TRANSPOSED (A = packed weights, B = the x rows) and
//    v_h straight (A = x rows, B = packed weights); in those MFMA output layouts k_h^T / q_h^T registers ARE the A / B
//    operands of S^T = K Q^T and the v_h registers ARE the B operand of P V, so the whole head runs from registers;
//  * the feed-forward hidden layer goes through `buf` in two halves of 256 units (chosen as the units that one half of the
//    half-split weight packing covers), the second Linear accumulating over both halves in registers.
template <int H, int NC = 1>  // H = 1: split-f16 MFMAs for the big contractions (see the file header); 2: plain f16 (one product,
                            // the high halves of the same packing); 0: everything on the f32 MFMA
__global__ __launch_bounds__(256, NC == 1 ? 2 : 1) void encode_cells_kernel(EncParams P, t2l_packed_cells in,
                                                                           float* __restrict__ out) {
  static_assert(NC == 1, "one cell per workgroup (the two-cell form was measured and rejected, see above)");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* x[NC];    // [32][260] token buffer; scratch (small-MLP hidden / features2 staging) before it is live
  float* buf[NC];  // [32][260] feature slot -> attention output -> feed-forward hidden half
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    x[c] = smem + c * 2 * kXFloats;
    buf[c] = x[c] + kXFloats;
  }
  float* red = smem + NC * 2 * kXFloats;  // [8 * NC]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  int cell[NC], obj0[NC], nobj[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    cell[c] = min((int)blockIdx.x * NC + c, in.n_cells - 1);  // an odd tail workgroup computes its last cell twice
    obj0[c] = in.offsets[cell[c]];
    nobj[c] = min(in.offsets[cell[c] + 1] - obj0[c], kS);  // objects beyond 28 are dropped (cell_retrieval.py:94-98)
  }

  // ------------------------------------------------------------------ per-object features, merged slot by slot
  f32x16 keep0[NC], keep1[NC];  // merge output tiles (wave, wave + 4)
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) keep0[c][r] = keep1[c][r] = 0.f;
  int slot = 0;
  auto merge_slot = [&]() {  // buf holds slot `slot` (normalised rows): keep += buf @ Wmerge[:, 256*slot : 256*slot+256]^T
    __syncthreads();
    if (P.nfeat > 1) {
      if constexpr (H != 0) {
        const uint4* hp = P.merge_hp + (size_t)slot * (kD * kD / 4);
        const uint4* w0 = hp + ((size_t)wave * (kD / 16) * 64 + lane) * 2;
        const uint4* w1 = hp + ((size_t)(wave + 4) * (kD / 16) * 64 + lane) * 2;
        mm_pair_h(buf[0] + col * kLdX + half * 128, kD / 16, w0, w1, keep0[0], keep1[0]);
      } else {
        const float4* wp = P.merge_wp + (size_t)slot * (kD * kD / 4);
        mm_pair(buf[0] + col * kLdX + half * 128, kD / 8, wp + (size_t)wave * (kD / 8) * 64 + lane,
                wp + (size_t)(wave + 4) * (kD / 8) * 64 + lane, keep0[0], keep1[0]);
      }
    } else {
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
          keep0[c][r] = buf[c][row * kLdX + wave * 32 + col];
          keep1[c][r] = buf[c][row * kLdX + (wave + 4) * 32 + col];
        }
    }
    ++slot;
    __syncthreads();  // every wave is done reading buf (and the x-region scratch) before the next slot rewrites them
  };
  if (P.use_class) {
    if (P.class_embed) {  // object_encoder.py:103-110 (table rows pre-normalised on the host)
#pragma unroll
      for (int c = 0; c < NC; ++c)
        for (int o = 0; o < kSP; ++o) {
          float v = 0.f;
          if (o < nobj[c]) {
            const int ci = min(max(in.class_idx[obj0[c] + o], 0), P.n_class - 1);
            v = P.class_tab[ci * kD + tid];
          }
          buf[c][o * kLdX + tid] = v;
        }
    } else {  // object_encoder.py:86-99,112: features2 -> mlp_pointnet -> normalize
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float* stage = x[c];  // park features2 in the (not yet live) token buffer
        for (int o = wave; o < kSP; o += 4) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (o < nobj[c]) v = reinterpret_cast<const float4*>(in.pn_feat + (size_t)(obj0[c] + o) * kD)[lane];
          reinterpret_cast<float4*>(stage + o * kLdX)[lane] = v;
        }
      }
      __syncthreads();
      const float* pb = P.pn_b;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float* dst = buf[c];
        auto pn_epi = [&](int, int, int row, int cc, float v) { dst[row * kLdX + cc] = fmaxf(v + pb[cc], 0.f); };
        // (features2 is an input: its magnitude is not bounded by the weights, so this GEMM stays f32)
        gemm32(x[c], kLdX, kD, P.pn_wp, kD, wave, lane, pn_epi);
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < NC; ++c) normalize_rows(buf[c], kLdX, nobj[c], wave, lane);
    }
    merge_slot();
  }
  if (P.use_color) {
    if (P.color_embed) {  // object_encoder.py:116-120
#pragma unroll
      for (int c = 0; c < NC; ++c)
        for (int o = 0; o < kSP; ++o) {
          float v = 0.f;
          if (o < nobj[c]) {
            const int ci = min(max(in.color_idx[obj0[c] + o], 0), P.n_color - 1);
            v = P.color_tab[ci * kD + tid];
          }
          buf[c][o * kLdX + tid] = v;
        }
    } else {  // object_encoder.py:121-128
#pragma unroll
      for (int c = 0; c < NC; ++c)
        small_mlp<3>(P.color, in.rgb + (size_t)obj0[c] * 3, false, nobj[c], x[c], buf[c], tid, wave, lane);
    }
    merge_slot();
  }
  if (P.use_pos) {  // object_encoder.py:130-136
#pragma unroll
    for (int c = 0; c < NC; ++c)
      small_mlp<3>(P.pos, in.center + (size_t)obj0[c] * 3, false, nobj[c], x[c], buf[c], tid, wave, lane);
    merge_slot();
  }
  if (P.use_num) {  // object_encoder.py:138-145
#pragma unroll
    for (int c = 0; c < NC; ++c) small_mlp<1>(P.num, in.n_pts + obj0[c], true, nobj[c], x[c], buf[c], tid, wave, lane);
    merge_slot();
  }
  // merge epilogue (object_encoder.py:148-149: Linear+BN folded, ReLU) + normalize (cell_retrieval.py:92)
  {
    const float* mb = P.merge_b;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int c0 = wave * 32 + col, c1 = (wave + 4) * 32 + col;
        x[c][row * kLdX + c0] = P.nfeat > 1 ? fmaxf(keep0[c][r] + mb[c0], 0.f) : keep0[c][r];
        x[c][row * kLdX + c1] = P.nfeat > 1 ? fmaxf(keep1[c][r] + mb[c1], 0.f) : keep1[c][r];
      }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NC; ++c) normalize_rows(x[c], kLdX, nobj[c], wave, lane);  // rows >= nobj: the zero pad slots (cell_retrieval.py:85)
  __syncthreads();

  // ------------------------------------------------------------------ set transformer (cell_retrieval.py:101-103)
  for (int l = 0; l < P.num_layers; ++l) {
    const LayerW& W = P.layer[l];
    {  // head h = wave, registers only
      const int h = wave;
      constexpr int QN = kD / 8;  // 32 packed k-steps
      const float* ib = W.in_b;
      // ---- pass 1: q_h^T and k_h^T of every cell (the packed weights are the A operand, the token rows the B operand); one
      // load of a weight fragment feeds every cell. v_h follows in its own pass: 12 live accumulators would not fit.
      f32x16 st[NC];  // becomes S^T, then the unnormalised probabilities
      float inv[NC];
      {
        f32x16 qT0[NC], qT1[NC], kT0[NC], kT1[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) qT0[c][r] = qT1[c][r] = kT0[c][r] = kT1[c][r] = 0.f;
        if constexpr (H != 0) {
          constexpr int HS = kD / 16;
          const uint4* hq0 = W.in_hp + ((size_t)(2 * h) * HS * 64 + lane) * 2;
          const uint4* hq1 = W.in_hp + ((size_t)(2 * h + 1) * HS * 64 + lane) * 2;
          const uint4* hk0 = W.in_hp + ((size_t)(8 + 2 * h) * HS * 64 + lane) * 2;
          const uint4* hk1 = W.in_hp + ((size_t)(9 + 2 * h) * HS * 64 + lane) * 2;
#pragma unroll 2
          for (int s = 0; s < HS; ++s) {
            HFrag xf[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) xf[c] = split_h<H == 2>(x[c] + col * kLdX + half * 128 + 8 * s);
            {
              const HFrag f = load_h1<H == 2>(hq0 + T2L_WSTEP(s) * 128);
#pragma unroll
              for (int c = 0; c < NC; ++c) mfma_h3<H == 2>(qT0[c], f, xf[c]);
            }
            {
              const HFrag f = load_h1<H == 2>(hq1 + T2L_WSTEP(s) * 128);
#pragma unroll
              for (int c = 0; c < NC; ++c) mfma_h3<H == 2>(qT1[c], f, xf[c]);
            }
            {
              const HFrag f = load_h1<H == 2>(hk0 + T2L_WSTEP(s) * 128);
#pragma unroll
              for (int c = 0; c < NC; ++c) mfma_h3<H == 2>(kT0[c], f, xf[c]);
            }
            {
              const HFrag f = load_h1<H == 2>(hk1 + T2L_WSTEP(s) * 128);
#pragma unroll
              for (int c = 0; c < NC; ++c) mfma_h3<H == 2>(kT1[c], f, xf[c]);
            }
          }
        } else {
          const float* xr = x[0] + col * kLdX + half * 128;
          const float4* wq0 = W.in_wp + (size_t)(2 * h) * QN * 64 + lane;
          const float4* wq1 = W.in_wp + (size_t)(2 * h + 1) * QN * 64 + lane;
          const float4* wk0 = W.in_wp + (size_t)(8 + 2 * h) * QN * 64 + lane;
          const float4* wk1 = W.in_wp + (size_t)(9 + 2 * h) * QN * 64 + lane;
#pragma unroll 2
          for (int q = 0; q < QN; ++q) {
            const float4 xv = *reinterpret_cast<const float4*>(xr + 4 * q);
            const float4 a0 = wq0[q * 64], a1 = wq1[q * 64], c0 = wk0[q * 64], c1 = wk1[q * 64];
#define T2L_QK(C)                                                                   \
  qT0[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.C, xv.C, qT0[0], 0, 0, 0);       \
  qT1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.C, xv.C, qT1[0], 0, 0, 0);       \
  kT0[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0.C, xv.C, kT0[0], 0, 0, 0);       \
  kT1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1.C, xv.C, kT1[0], 0, 0, 0);
            T2L_QK(x) T2L_QK(y) T2L_QK(z) T2L_QK(w)
#undef T2L_QK
          }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          // in_proj bias: q^T / k^T rows are features (register index)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int f = (r & 3) + 8 * (r >> 2) + 4 * half;
            qT0[c][r] += ib[h * 64 + f];
            qT1[c][r] += ib[h * 64 + 32 + f];
            kT0[c][r] += ib[kD + h * 64 + f];
            kT1[c][r] += ib[kD + h * 64 + 32 + f];
          }
          // S^T[j][i] = k_j . q_i: k_h^T (token j = lane col, feature pair (f, f+4) = the two lane halves) is the A operand,
          // q_h^T the B operand, one MFMA per register
#pragma unroll
          for (int r = 0; r < 16; ++r) st[c][r] = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) st[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(kT0[c][r], qT0[c][r], st[c], 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 16; ++r) st[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(kT1[c][r], qT1[c][r], st[c], 0, 0, 0);
          // lane: query i = col, keys j = (r&3) + 8*(r>>2) + 4*half ; keys >= 28 are the dead rows
          float m = -__builtin_inff();
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int j = (r & 3) + 8 * (r >> 2) + 4 * half;
            st[c][r] = (j < kS) ? st[c][r] * 0.125f : -__builtin_inff();  // 1/sqrt(head_dim = 64)
            m = fmaxf(m, st[c][r]);
          }
          m = fmaxf(m, __shfl_xor(m, 32));
          float sum = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            st[c][r] = __expf(st[c][r] - m);
            sum += st[c][r];
          }
          sum += __shfl_xor(sum, 32);
          inv[c] = 1.f / sum;
        }
      }
      // ---- pass 2: v_h straight (A = token rows, B = packed weights), then o = P V from registers
      {
        f32x16 v0[NC], v1[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) v0[c][r] = v1[c][r] = 0.f;
        if constexpr (H != 0) {
          constexpr int HS = kD / 16;
          const uint4* hv0 = W.in_hp + ((size_t)(16 + 2 * h) * HS * 64 + lane) * 2;
          const uint4* hv1 = W.in_hp + ((size_t)(17 + 2 * h) * HS * 64 + lane) * 2;
#pragma unroll 4
          for (int s = 0; s < HS; ++s) {
            const HFrag f0 = load_h1<H == 2>(hv0 + T2L_WSTEP(s) * 128), f1 = load_h1<H == 2>(hv1 + T2L_WSTEP(s) * 128);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              const HFrag xf = split_h<H == 2>(x[c] + col * kLdX + half * 128 + 8 * s);
              mfma_h3<H == 2>(v0[c], xf, f0);
              mfma_h3<H == 2>(v1[c], xf, f1);
            }
          }
        } else {
          const float* xr = x[0] + col * kLdX + half * 128;
          const float4* wv0 = W.in_wp + (size_t)(16 + 2 * h) * QN * 64 + lane;
          const float4* wv1 = W.in_wp + (size_t)(17 + 2 * h) * QN * 64 + lane;
          mm_pair(xr, QN, wv0, wv1, v0[0], v1[0]);
        }
        const float bv0 = ib[2 * kD + h * 64 + col], bv1 = ib[2 * kD + h * 64 + 32 + col];  // v columns are features (lane)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          // o[i][n] = sum_j P[i][j] v[j][n]: P (lane = query i, register = key j) is the A operand, v_h registers (lane =
          // column n, register = key j) the B operand
          f32x16 o0, o1;
#pragma unroll
          for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float p = st[c][r] * inv[c];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(p, v0[c][r] + bv0, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(p, v1[c][r] + bv1, o1, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
            buf[c][i * kLdX + h * 64 + col] = o0[r];
            buf[c][i * kLdX + h * 64 + 32 + col] = o1[r];
          }
        }
      }
    }
    __syncthreads();
    {  // x = LN1(x + o @ out_proj^T + b)
      const float* b = W.out_b;
      {
        float* xd = x[0];
        auto out_epi = [&](int, int, int row, int c, float v) { xd[row * kLdX + c] += v + b[c]; };
        if constexpr (H != 0) gemm32_h<H == 2>(buf[0], kLdX, kD, W.out_hp, kD, wave, lane, out_epi);
        else gemm32(buf[0], kLdX, kD, W.out_wp, kD, wave, lane, out_epi);
      }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; ++c) layer_norm_rows(x[c], W.ln1_w, W.ln1_b, wave, lane);
    __syncthreads();
    {  // x = LN2(x + relu(x W1^T + b1) W2^T + b2), hidden units in two halves through buf
      f32x16 acc0[NC], acc1[NC];  // output tiles (wave, wave + 4)
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[c][r] = acc1[c][r] = 0.f;
      const float* b1 = W.ff1_b;
      for (int hf = 0; hf < 2; ++hf) {
        // half hf = hidden units [128 hf, 128 hf + 128) and [256 + 128 hf, 256 + 128 hf + 128): exactly what k-steps
        // [32 hf, 32 hf + 32) of the half-split packing of W2 (K = 512) cover
        const int tA = 4 * hf + wave, tB = 8 + 4 * hf + wave;
        f32x16 h0[NC], h1[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) h0[c][r] = h1[c][r] = 0.f;
        if constexpr (H != 0) {
          const uint4* w0 = W.ff1_hp + ((size_t)tA * (kD / 16) * 64 + lane) * 2;
          const uint4* w1 = W.ff1_hp + ((size_t)tB * (kD / 16) * 64 + lane) * 2;
          mm_pair_h<H == 2>(x[0] + col * kLdX + half * 128, kD / 16, w0, w1, h0[0], h1[0]);
        } else {
          mm_pair(x[0] + col * kLdX + half * 128, kD / 8, W.ff1_wp + (size_t)tA * (kD / 8) * 64 + lane,
                  W.ff1_wp + (size_t)tB * (kD / 8) * 64 + lane, h0[0], h1[0]);
        }
        if (hf) __syncthreads();  // every wave has consumed the first half from buf
        const float bA = b1[tA * 32 + col], bB = b1[tB * 32 + col];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            buf[c][row * kLdX + 32 * wave + col] = fmaxf(h0[c][r] + bA, 0.f);
            buf[c][row * kLdX + 128 + 32 * wave + col] = fmaxf(h1[c][r] + bB, 0.f);
          }
        __syncthreads();
        if constexpr (H != 0) {  // K = 512: 32 steps per tile, half hf = steps [16 hf, 16 hf + 16)
          const uint4* w0 = W.ff2_hp + (((size_t)wave * (2 * kD / 16) + 16 * hf) * 64 + lane) * 2;
          const uint4* w1 = W.ff2_hp + (((size_t)(wave + 4) * (2 * kD / 16) + 16 * hf) * 64 + lane) * 2;
          mm_pair_h<H == 2>(buf[0] + col * kLdX + half * 128, kD / 16, w0, w1, acc0[0], acc1[0]);
        } else {
          mm_pair(buf[0] + col * kLdX + half * 128, kD / 8, W.ff2_wp + ((size_t)wave * (2 * kD / 8) + 32 * hf) * 64 + lane,
                  W.ff2_wp + ((size_t)(wave + 4) * (2 * kD / 8) + 32 * hf) * 64 + lane, acc0[0], acc1[0]);
        }
      }
      const float* b2 = W.ff2_b;
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
          const int c0 = wave * 32 + col, c1 = (wave + 4) * 32 + col;
          x[c][row * kLdX + c0] += acc0[c][r] + b2[c0];
          x[c][row * kLdX + c1] += acc1[c][r] + b2[c1];
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; ++c) layer_norm_rows(x[c], W.ln2_w, W.ln2_b, wave, lane);
    __syncthreads();
  }

  // ------------------------------------------------------------------ max over ALL 28 slots, pads included (cell_retrieval.py:107-108)
  float mx[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    mx[c] = x[c][tid];
    for (int i = 1; i < kS; ++i) mx[c] = fmaxf(mx[c], x[c][i * kLdX + tid]);
    const float ss = wave_sum(mx[c] * mx[c]);
    if (lane == 0) red[c * 4 + wave] = ss;
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float nrm = sqrtf(red[c * 4 + 0] + red[c * 4 + 1] + red[c * 4 + 2] + red[c * 4 + 3]);
    if ((int)blockIdx.x * NC + c < in.n_cells) out[(size_t)cell[c] * kD + tid] = mx[c] / fmaxf(nrm, 1e-12f);
  }
}

// ------------------------------------------------------------------------------------------------
// t2l_text_inter as ONE launch (models/language_encoder.py:137-147): x = sent.view(B, S, 256); x += TransformerEncoderLayer(256, 4 heads,
// ff 1024, post-norm, ReLU)(x) over the S sentences of a description; max over the sentences. The layer is the cell encoder's
// (above) with three differences: a 32-row tile holds floor(32 / S) whole DESCRIPTIONS and a query attends to the keys of its own
// description only (block-diagonal mask; rows past the tile's last description are zero rows that form groups of their own: every
// softmax has its own row as a key, nothing is NaN); the feed-forward hidden layer is 1,024 wide = four passes through `buf`
// (pass c = hidden units [128c, 128c+128) and [512+128c, 512+128c+128): what k-steps [16c, 16c+16) of W2's half-split packing at
// K = 1,024 cover); and the inputs are not bounded by the weights (they come out of inter_mlp), so every value that enters a
// split-f16 product is watched against the f16 range at run time: a tile that leaves it raises *flag and the caller redoes the batch
// on the PyTorch modules (the protocol of t2l_text_head).
// (The kernel: text_inter_fused2_kernel below — two tiles per eight-wave workgroup on LDS planes. The one-tile form on f32 tiles that
// this comment used to head was removed in round 5: 0.208 vs 0.179 ms for 4,096 descriptions x 6 sentences.)

// ------------------------------------------------------------------------------------------------
// The launch — the testbed for the lever the encoder family is left with (the packed-weight
// stream out of the L2 is worth 23-32 % of these kernels, the per-MFMA VALU work another third): TWO row tiles per workgroup of EIGHT
// waves, every activation resident in LDS as split-f16 PLANES (hi | lo, rows of 264 halves), so that
//  * out_proj, linear1 and linear2 (3/4 of the FLOPs) load each weight fragment ONCE for both tiles — wave w owns one 32-wide tile of
//    output features and multiplies it into both token tiles (products computed transposed: A = weight fragment, B = token fragment);
//  * no operand is split in a GEMM loop: a token fragment is two ds_read_b128; the splitting happens once per produced element in the
//    epilogues, which hold 4 consecutive features per register quad (transposed C layout) and store 8-byte plane pieces;
//  * the attention runs as in the first form, one wave per (tile, head), q/k/v projected per tile (their fragments are not shared), with
//    O^T = V^T P^T so that its output has the same store-friendly layout.
// 135 KB of LDS: one workgroup (two waves per SIMD) per CU. The residual is rebuilt from hi + lo (22 significand bits).
#ifndef T2L_RING_DEPTH
#define T2L_RING_DEPTH 3
#endif
#ifndef T2L_QK_RING
#define T2L_QK_RING 2
#endif
constexpr int kQkRing = T2L_QK_RING;  // the same for the q / k projection (four weight tiles = 32 VGPRs per step; v: twice as deep)
constexpr int kRingDepth = T2L_RING_DEPTH;  // k-steps of weight fragments in flight per wave in the row-wise products (8 VGPRs per step)
constexpr int kLdP = 264;                 // halves per plane row (528 B: rows 4 banks apart, as the f32 tiles)
constexpr int kPlane = kSP * kLdP;        // halves per plane
typedef _Float16 ti_f16x4 __attribute__((ext_vector_type(4)));
typedef float ti_f32x4 __attribute__((ext_vector_type(4)));

template <bool SG>
__device__ __forceinline__ HFrag plane_frag(const _Float16* __restrict__ hi, const _Float16* __restrict__ lo, int off) {
  HFrag f;
  f.hi = *reinterpret_cast<const h3_f16x8*>(hi + off);
  if constexpr (SG) f.lo = f.hi;
  else f.lo = *reinterpret_cast<const h3_f16x8*>(lo + off);
  return f;
}
template <bool SG>
__device__ __forceinline__ void plane_put4(_Float16* __restrict__ hi, _Float16* __restrict__ lo, int off, ti_f32x4 v) {
  h3_f16x4 h, l;
  h3_split4(h3_f32x4{v[0], v[1], v[2], v[3]}, h, l);
  *reinterpret_cast<h3_f16x4*>(hi + off) = h;
  // (the plain-f16 option drops the low halves from the PRODUCTS only: the stored activations — the residual stream — keep both)
  *reinterpret_cast<h3_f16x4*>(lo + off) = l;
}
template <bool SG>
__device__ __forceinline__ ti_f32x4 plane_get4(const _Float16* __restrict__ hi, const _Float16* __restrict__ lo, int off) {
  const h3_f32x4 v = h3_join4(*reinterpret_cast<const h3_f16x4*>(hi + off), *reinterpret_cast<const h3_f16x4*>(lo + off));
  return ti_f32x4{v[0], v[1], v[2], v[3]};
}

// The weight fragments of one tile pass, STEPS k-steps, through a register ring D steps deep: the fragment of step s + D is requested
// when step s is consumed, sched_barriers keep the requests where they are written (the compiler otherwise sinks every load to its use:
// one L2 round trip of ~1 us in front of every 0.1 us of MFMAs). body(s, fragment). Measured on t2l_text_inter (4,096 x 6): no ring
// 0.218 ms; D = 2 / 3 / 4 / 5 / 6 / 8 / 12 / 16: 0.169 / 0.168 / 0.169 / 0.170 / 0.173 / 0.174 / 0.181 / 0.186 ms — what matters is that the
// next requests are out before the MFMAs start, not how many (deeper rings cost registers and scalar spills).
template <bool SG, int STEPS, int D, typename F>
__device__ __forceinline__ void stream_weights(const uint4* __restrict__ wp, F&& body) {
  HFrag ring[D];
#pragma unroll
  for (int i = 0; i < D; ++i) ring[i] = load_h1<SG>(wp + T2L_WSTEP(i) * 128);
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const HFrag wf = ring[s % D];
    if (s + D < STEPS) ring[s % D] = load_h1<SG>(wp + T2L_WSTEP(s + D) * 128);
    __builtin_amdgcn_sched_barrier(0);
    body(s, wf);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ _Float16* pl_xh(_Float16* base, int t) { return base + (size_t)(4 * t + 0) * kPlane; }
__device__ __forceinline__ _Float16* pl_xl(_Float16* base, int t) { return base + (size_t)(4 * t + 1) * kPlane; }
__device__ __forceinline__ _Float16* pl_bh(_Float16* base, int t) { return base + (size_t)(4 * t + 2) * kPlane; }
__device__ __forceinline__ _Float16* pl_bl(_Float16* base, int t) { return base + (size_t)(4 * t + 3) * kPlane; }

// One post-norm TransformerEncoderLayer (d_model 256, 4 heads, ReLU, feed-forward of 256 FFP units) over the TWO 32-row token tiles of an
// eight-wave workgroup, in place on the tiles' X planes (B planes: scratch). mask(i, j): may query row i see key row j (tile-local)?
// WATCH: run-time f16-range watch on everything that enters a split product (`bad`). last_to_f32: the final LayerNorm leaves f32
// [32][260] tiles in the B-plane regions instead of planes (for an epilogue that needs full precision). Ends behind a barrier.
// lnred: 1,024 floats of LDS scratch (the row sums of the fused residual + LayerNorm epilogues).
template <bool SG, int FFP, bool WATCH, typename Mask>
__device__ __forceinline__ void planes_layer(_Float16* base, const InterFusedW& W, Mask mask, bool& bad, bool last_to_f32, float* lnred) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  auto watch = [&](float v) { bad = bad || !(fabsf(v) < kSplitF16Safe); };
  auto watch4 = [&](ti_f32x4 v) { watch(v[0]); watch(v[1]); watch(v[2]); watch(v[3]); };
  (void)watch4;
  {  // ---- self-attention: wave -> (tile t, head h); q^T, k^T, v from the tile's planes, scores / softmax / O^T from registers
    const int t = wave >> 2, h = wave & 3;
    constexpr int HS = kD / 16;
    const float* ib = W.in_b;
    const _Float16 *xh = pl_xh(base, t) + col * kLdP + half * 128, *xl = pl_xl(base, t) + col * kLdP + half * 128;
    f32x16 st;
    float inv;
    {
      f32x16 qT0, qT1, kT0, kT1;
#pragma unroll
      for (int r = 0; r < 16; ++r) qT0[r] = qT1[r] = kT0[r] = kT1[r] = 0.f;
      const uint4* hq0 = W.in_hp + ((size_t)(2 * h) * HS * 64 + lane) * 2;
      const uint4* hq1 = W.in_hp + ((size_t)(2 * h + 1) * HS * 64 + lane) * 2;
      const uint4* hk0 = W.in_hp + ((size_t)(8 + 2 * h) * HS * 64 + lane) * 2;
      const uint4* hk1 = W.in_hp + ((size_t)(9 + 2 * h) * HS * 64 + lane) * 2;
      {  // four weight tiles per step through a ring kQkRing steps deep (32 VGPRs per step)
        constexpr int D = kQkRing;
        HFrag ring[D][4];
        auto load4 = [&](int s, HFrag (&f)[4]) {
          f[0] = load_h1<SG>(hq0 + T2L_WSTEP(s) * 128);
          f[1] = load_h1<SG>(hq1 + T2L_WSTEP(s) * 128);
          f[2] = load_h1<SG>(hk0 + T2L_WSTEP(s) * 128);
          f[3] = load_h1<SG>(hk1 + T2L_WSTEP(s) * 128);
        };
#pragma unroll
        for (int i = 0; i < D; ++i) load4(i, ring[i]);
#pragma unroll
        for (int s = 0; s < HS; ++s) {
          HFrag f[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) f[e] = ring[s % D][e];
          if (s + D < HS) load4(s + D, ring[s % D]);
          __builtin_amdgcn_sched_barrier(0);
          const HFrag xf = plane_frag<SG>(xh, xl, 8 * s);
          mfma_h3<SG>(qT0, f[0], xf);
          mfma_h3<SG>(qT1, f[1], xf);
          mfma_h3<SG>(kT0, f[2], xf);
          mfma_h3<SG>(kT1, f[3], xf);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = (r & 3) + 8 * (r >> 2) + 4 * half;
        qT0[r] += ib[h * 64 + f];
        qT1[r] += ib[h * 64 + 32 + f];
        kT0[r] += ib[kD + h * 64 + f];
        kT1[r] += ib[kD + h * 64 + 32 + f];
      }
      // S^T = K Q^T (and O^T = V^T P^T below) as split-f16 products on the accumulator registers: kT / qT (v / P) sit in the same
      // MFMA output layout, so registers 0..7 and 8..15 of the two are matching k-halves of A and B — 12 MFMAs of 32 cycles here
      // instead of 32 f32 MFMAs of 64 (the attention core was a third of a wave's matrix-pipe time in this layer). Always the
      // three-product form (the logits carry the softmax); q, k, v are inside the load-time bound of the in_proj output (cell
      // encoder) or watched here (WATCH).
      if constexpr (WATCH) {
        float wm = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) wm = fmaxf(wm, fmaxf(fmaxf(fabsf(qT0[r]), fabsf(qT1[r])), fmaxf(fabsf(kT0[r]), fabsf(kT1[r]))));
        watch(wm);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2) {  // (fenced: the compiler otherwise splits all eight fragments first — 64 registers of temporaries)
        mfma_h3<false>(st, split_acc8<false>(kT0, m2), split_acc8<false>(qT0, m2));
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2) {
        mfma_h3<false>(st, split_acc8<false>(kT1, m2), split_acc8<false>(qT1, m2));
        __builtin_amdgcn_sched_barrier(0);
      }
      float m = -__builtin_inff();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = (r & 3) + 8 * (r >> 2) + 4 * half;
        st[r] = mask(col, j) ? st[r] * 0.125f : -__builtin_inff();
        m = fmaxf(m, st[r]);
      }
      m = fmaxf(m, __shfl_xor(m, 32));
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[r] = __expf(st[r] - m);
        sum += st[r];
      }
      sum += __shfl_xor(sum, 32);
      inv = 1.f / sum;
    }
    {
      f32x16 v0, v1;  // v straight: lane = feature column, register = token row
#pragma unroll
      for (int r = 0; r < 16; ++r) v0[r] = v1[r] = 0.f;
      const uint4* hv0 = W.in_hp + ((size_t)(16 + 2 * h) * HS * 64 + lane) * 2;
      const uint4* hv1 = W.in_hp + ((size_t)(17 + 2 * h) * HS * 64 + lane) * 2;
      {
        constexpr int D = 2 * kQkRing;
        HFrag ring[D][2];
#pragma unroll
        for (int i = 0; i < D; ++i) {
          ring[i][0] = load_h1<SG>(hv0 + T2L_WSTEP(i) * 128);
          ring[i][1] = load_h1<SG>(hv1 + T2L_WSTEP(i) * 128);
        }
#pragma unroll
        for (int s = 0; s < HS; ++s) {
          const HFrag f0 = ring[s % D][0], f1 = ring[s % D][1];
          if (s + D < HS) {
            ring[s % D][0] = load_h1<SG>(hv0 + T2L_WSTEP(s + D) * 128);
            ring[s % D][1] = load_h1<SG>(hv1 + T2L_WSTEP(s + D) * 128);
          }
          __builtin_amdgcn_sched_barrier(0);
          const HFrag xf = plane_frag<SG>(xh, xl, 8 * s);
          mfma_h3<SG>(v0, xf, f0);
          mfma_h3<SG>(v1, xf, f1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      const float bv0 = ib[2 * kD + h * 64 + col], bv1 = ib[2 * kD + h * 64 + 32 + col];
      // O^T[f][i] = sum_j v[j][f] P[i][j]: A = v registers (lane = feature, lane half = the key of register r), B = P registers (lane =
      // query i, lane half = the same key) -> lane = query (token row), register quad = 4 consecutive features
      f32x16 o0, o1;
#pragma unroll
      for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[r] *= inv;
        v0[r] += bv0;
        v1[r] += bv1;
      }
      if constexpr (WATCH) {
        float wm = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) wm = fmaxf(wm, fmaxf(fabsf(v0[r]), fabsf(v1[r])));
        watch(wm);
      }
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2) {
        const HFrag pf = split_acc8<false>(st, m2);
        mfma_h3<false>(o0, split_acc8<false>(v0, m2), pf);
        mfma_h3<false>(o1, split_acc8<false>(v1, m2), pf);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const ti_f32x4 a = {o0[4 * q], o0[4 * q + 1], o0[4 * q + 2], o0[4 * q + 3]};
        const ti_f32x4 b = {o1[4 * q], o1[4 * q + 1], o1[4 * q + 2], o1[4 * q + 3]};
        if constexpr (WATCH) watch4(a);
        if constexpr (WATCH) watch4(b);
        plane_put4<SG>(pl_bh(base, t), pl_bl(base, t), col * kLdP + h * 64 + 8 * q + 4 * half, a);
        plane_put4<SG>(pl_bh(base, t), pl_bl(base, t), col * kLdP + h * 64 + 32 + 8 * q + 4 * half, b);
      }
    }
  }
  __syncthreads();
  // operand fragments of the row-wise products: token fragment of tile t at k-step s (this lane's row `col`, k half `half`)
  const int frow = col * kLdP + half * 128;
  // x = LayerNorm(x + acc^T + bias) * g + be for both tiles — the residual epilogue and the LayerNorm behind it in one go (round 5; proven
  // on fine.hip first). A lane holds 16 of a token's 256 features per tile (its partner lane ^ 32 another 16, the other seven waves 32
  // each): the row sums meet in `lnred` behind two light barriers (mean, then centred squares: the two-pass form), the values stay in
  // registers in between and the normalised rows are written once. As a separate pass (8 rows per wave, two full-wave reductions per
  // row) the two LayerNorms of a layer were ~700 VALU instructions per wave and a plane round trip each.
  auto resid_ln = [&](const f32x16 (&acc)[2], const float* __restrict__ bias, const float* __restrict__ g, const float* __restrict__ be,
                      bool to_f32) {
    ti_f32x4 v[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float sm = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f0 = 32 * wave + 8 * q + 4 * half;
        const float4 bb = *reinterpret_cast<const float4*>(bias + f0);
        v[t][q] = plane_get4<SG>(pl_xh(base, t), pl_xl(base, t), col * kLdP + f0) +
                  ti_f32x4{acc[t][4 * q] + bb.x, acc[t][4 * q + 1] + bb.y, acc[t][4 * q + 2] + bb.z, acc[t][4 * q + 3] + bb.w};
        sm += (v[t][q][0] + v[t][q][1]) + (v[t][q][2] + v[t][q][3]);
      }
      sm += __shfl_xor(sm, 32);
      if (half == 0) lnred[(t * 8 + wave) * 32 + col] = sm;
    }
    __syncthreads();
    float mean[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float m = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) m += lnred[(t * 8 + w) * 32 + col];
      mean[t] = m * (1.f / kD);
      float qs = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[t][q] -= mean[t];
        qs += (v[t][q][0] * v[t][q][0] + v[t][q][1] * v[t][q][1]) + (v[t][q][2] * v[t][q][2] + v[t][q][3] * v[t][q][3]);
      }
      qs += __shfl_xor(qs, 32);
      if (half == 0) lnred[512 + (t * 8 + wave) * 32 + col] = qs;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float var = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) var += lnred[512 + (t * 8 + w) * 32 + col];
      const float inv = 1.f / sqrtf(var * (1.f / kD) + 1e-5f);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f0 = 32 * wave + 8 * q + 4 * half;
        const float4 gg = *reinterpret_cast<const float4*>(g + f0), bb = *reinterpret_cast<const float4*>(be + f0);
        const ti_f32x4 o = {v[t][q][0] * inv * gg.x + bb.x, v[t][q][1] * inv * gg.y + bb.y, v[t][q][2] * inv * gg.z + bb.z,
                            v[t][q][3] * inv * gg.w + bb.w};
        if (to_f32) {  // (the last LayerNorm: the tile's B planes become one f32 [32][260] tile for the epilogue)
          *reinterpret_cast<ti_f32x4*>(reinterpret_cast<float*>(pl_bh(base, t)) + col * kLdX + f0) = o;
        } else {
          if constexpr (WATCH) watch4(o);
          plane_put4<SG>(pl_xh(base, t), pl_xl(base, t), col * kLdP + f0, o);
        }
      }
    }
  };
  {  // ---- x = x + o @ out_proj^T + b, transposed: wave w owns output features [32 w, 32 w + 32) for BOTH tiles
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const uint4* wp = W.out_hp + ((size_t)wave * (kD / 16) * 64 + lane) * 2;
    stream_weights<SG, kD / 16, kRingDepth>(wp, [&](int s, const HFrag& wf) {
#pragma unroll
      for (int t = 0; t < 2; ++t) mfma_h3<SG>(acc[t], wf, plane_frag<SG>(pl_bh(base, t) + frow, pl_bl(base, t) + frow, 8 * s));
    });
    resid_ln(acc, W.out_b, W.ln1_w, W.ln1_b, false);
  }
  __syncthreads();
  {  // ---- feed-forward, four passes of 256 hidden units; wave w: one hidden tile per pass and one output tile, both token tiles
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const float* b1 = W.ff1_b;
    constexpr int FS = FFP * kD / 16;  // k-steps of one W2 tile
    const int cbase = wave < 4 ? 32 * wave : 128 + 32 * (wave - 4);  // where this wave's hidden tile lives in the B planes
    for (int c = 0; c < FFP; ++c) {
      const int tf = wave < 4 ? 4 * c + wave : 4 * FFP + 4 * c + (wave - 4);
      f32x16 hT[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) hT[t][r] = 0.f;
      const uint4* w1 = W.ff1_hp + ((size_t)tf * (kD / 16) * 64 + lane) * 2;
      stream_weights<SG, kD / 16, kRingDepth>(w1, [&](int s, const HFrag& wf) {
#pragma unroll
        for (int t = 0; t < 2; ++t) mfma_h3<SG>(hT[t], wf, plane_frag<SG>(pl_xh(base, t) + frow, pl_xl(base, t) + frow, 8 * s));
      });
      if (c) __syncthreads();  // every wave has consumed the previous pass from the B planes
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int u0 = 8 * q + 4 * half;
          const float4 bb = *reinterpret_cast<const float4*>(b1 + tf * 32 + u0);
          const ti_f32x4 v = {fmaxf(hT[t][4 * q] + bb.x, 0.f), fmaxf(hT[t][4 * q + 1] + bb.y, 0.f), fmaxf(hT[t][4 * q + 2] + bb.z, 0.f),
                              fmaxf(hT[t][4 * q + 3] + bb.w, 0.f)};
          if constexpr (WATCH) watch4(v);
          plane_put4<SG>(pl_bh(base, t), pl_bl(base, t), col * kLdP + cbase + u0, v);
        }
      __syncthreads();
      const uint4* w2 = W.ff2_hp + (((size_t)wave * FS + 16 * c) * 64 + lane) * 2;
      stream_weights<SG, kD / 16, kRingDepth>(w2, [&](int s, const HFrag& wf) {
#pragma unroll
        for (int t = 0; t < 2; ++t) mfma_h3<SG>(acc[t], wf, plane_frag<SG>(pl_bh(base, t) + frow, pl_bl(base, t) + frow, 8 * s));
      });
    }
    resid_ln(acc, W.ff2_b, W.ln2_w, W.ln2_b, last_to_f32);  // (every wave is past its reads of the B planes and of x at the first barrier inside)
  }
  __syncthreads();
}

template <int H>
__global__ __launch_bounds__(512, 1) void text_inter_fused2_kernel(InterFusedW W, const float* __restrict__ sent, int n_desc, int S, int dpt,
                                                                   float* __restrict__ out, int* __restrict__ flag) {
  static_assert(H == 1 || H == 2, "split-f16 or plain f16");
  constexpr bool SG = H == 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  _Float16* base = reinterpret_cast<_Float16*>(smem);
  // per tile t: X planes (token tile), B planes (attention output -> hidden pass; at the very end an f32 [32][260] tile)
  auto XH = [&](int t) { return base + (size_t)(4 * t + 0) * kPlane; };
  auto XL = [&](int t) { return base + (size_t)(4 * t + 1) * kPlane; };
  auto BH = [&](int t) { return base + (size_t)(4 * t + 2) * kPlane; };
  auto BL = [&](int t) { return base + (size_t)(4 * t + 3) * kPlane; };
  int* grp = reinterpret_cast<int*>(base + (size_t)8 * kPlane);  // [32]: description of a tile-local row
  float* lnred = reinterpret_cast<float*>(grp + kSP);              // [1024]: row sums of the fused residual + LayerNorm epilogues
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int d0 = blockIdx.x * 2 * dpt;
  bool bad = false;
  auto watch = [&](float v) { bad = bad || !(fabsf(v) < kSplitF16Safe); };
  auto watch4 = [&](ti_f32x4 v) { watch(v[0]); watch(v[1]); watch(v[2]); watch(v[3]); };
  int nd[2], rows[2];
  nd[0] = min(dpt, n_desc - d0);
  nd[1] = max(0, min(dpt, n_desc - d0 - dpt));
  rows[0] = nd[0] * S;
  rows[1] = nd[1] * S;
  // ---- the 64 rows -> planes (8 rows per wave, 4 columns per lane)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = wave * 8 + i, t = r >> 5, lr = r & 31;
    ti_f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (lr < rows[t]) {
      const float4 g = reinterpret_cast<const float4*>(sent + ((size_t)(d0 + t * dpt) * S + lr) * kD)[lane];
      v = ti_f32x4{g.x, g.y, g.z, g.w};
    }
    watch4(v);
    plane_put4<SG>(XH(t), XL(t), lr * kLdP + 4 * lane, v);
  }
  if (tid < kSP) grp[tid] = tid / S;
  __syncthreads();

  planes_layer<SG, 4, true>(base, W, [&](int i, int j) { return grp[i] == grp[j]; }, bad, true, lnred);
  {  // x_in + layer(x_in), max over the description's S sentences: thread = (tile, column)
    const int t = tid >> 8, c = tid & 255;
    const float* y = reinterpret_cast<const float*>(BH(t));
    const float* src = sent + (size_t)(d0 + t * dpt) * S * kD;
    for (int d = 0; d < nd[t]; ++d) {
      float m = -__builtin_inff();
      for (int s_ = 0; s_ < S; ++s_) {
        const int r = d * S + s_;
        m = fmaxf(m, y[r * kLdX + c] + src[(size_t)r * kD + c]);
      }
      out[(size_t)(d0 + t * dpt + d) * kD + c] = m;
    }
  }
  if (__syncthreads_or(bad ? 1 : 0) && tid == 0) atomicOr(flag, 1);
}

// ------------------------------------------------------------------------------------------------
// encode_cells, second form (option encoder_two_cells, split-f16 / plain-f16 arithmetic, at least two feature slots): the recipe that
// t2l_text_inter's second form proved — TWO cells per workgroup of EIGHT waves, activations as split-f16 planes in LDS,
// the weight fragments of the feature merge, out_proj and both feed-forward Linears loaded once for both cells and requested ahead of
// the MFMAs, nothing split inside a GEMM loop. The feature stage keeps the first form's code: waves 4t .. 4t+3 build cell t's feature
// slot as a normalised f32 tile (in the cell's X-plane region, not live before the merge epilogue), convert it to the cell's B planes,
// and all eight waves contract both cells' slot with its slice of the merge weight.
template <int H>
__global__ __launch_bounds__(512, 1) void encode_cells2_kernel(EncParams P, t2l_packed_cells in, float* __restrict__ out) {
  static_assert(H == 1 || H == 2, "split-f16 or plain f16 (the all-f32 encoder keeps the first form)");
  constexpr bool SG = H == 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  _Float16* base = reinterpret_cast<_Float16*>(smem);
  float* hb_all = reinterpret_cast<float*>(base + (size_t)8 * kPlane);  // [2][32][68]: hidden layer of the small MLPs
  float* red = hb_all + 2 * kSP * kLdH;                                 // [8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int tc = wave >> 2, lw = wave & 3, ltid = tid & 255;  // the cell this wave builds features for, its wave / thread index there
  int cell[2], obj0[2], nobj[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    cell[t] = min((int)blockIdx.x * 2 + t, in.n_cells - 1);  // an odd tail workgroup computes its last cell twice
    obj0[t] = in.offsets[cell[t]];
    nobj[t] = min(in.offsets[cell[t] + 1] - obj0[t], kS);
  }
  float* xf32 = reinterpret_cast<float*>(pl_xh(base, tc));  // this wave group's f32 [32][260] scratch tile (X-plane region of its cell)
  float* bf32 = reinterpret_cast<float*>(pl_bh(base, tc));  // ... and the B-plane region as one (features2 staging only)
  float* hb = hb_all + tc * kSP * kLdH;
  const int my_nobj = nobj[tc], my_obj0 = obj0[tc];
  const int frow = col * kLdP + half * 128;

  f32x16 keep[2];  // merge accumulators, transposed: wave w = output features [32 w, 32 w + 32), lane = object slot
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) keep[t][r] = 0.f;
  int slot = 0;
  auto tile_to_planes = [&]() {  // the group's normalised f32 slot (X region) -> its cell's B planes
    for (int i = lw; i < kSP; i += 4) {
      const float4 g = *(reinterpret_cast<const float4*>(xf32 + i * kLdX) + lane);
      plane_put4<SG>(pl_bh(base, tc), pl_bl(base, tc), i * kLdP + 4 * lane, ti_f32x4{g.x, g.y, g.z, g.w});
    }
  };
  auto table_to_planes = [&](const float* __restrict__ tab, const int32_t* __restrict__ idx, int n_tab) {
    for (int o = lw; o < kSP; o += 4) {
      ti_f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (o < my_nobj) {
        const int ci = min(max(idx[my_obj0 + o], 0), n_tab - 1);
        const float4 g = reinterpret_cast<const float4*>(tab + (size_t)ci * kD)[lane];
        v = ti_f32x4{g.x, g.y, g.z, g.w};
      }
      plane_put4<SG>(pl_bh(base, tc), pl_bl(base, tc), o * kLdP + 4 * lane, v);
    }
  };
  auto merge_slot = [&]() {  // both cells' B planes hold slot `slot`: keep += Wmerge[:, 256 slot : 256 slot + 256] @ slot^T
    __syncthreads();
    const uint4* wp = P.merge_hp + (size_t)slot * (kD * kD / 4) + ((size_t)wave * (kD / 16) * 64 + lane) * 2;
    stream_weights<SG, kD / 16, kRingDepth>(wp, [&](int s, const HFrag& wf) {
#pragma unroll
      for (int t = 0; t < 2; ++t) mfma_h3<SG>(keep[t], wf, plane_frag<SG>(pl_bh(base, t) + frow, pl_bl(base, t) + frow, 8 * s));
    });
    ++slot;
    __syncthreads();
  };
  if (P.use_class) {
    if (P.class_embed) {
      table_to_planes(P.class_tab, in.class_idx, P.n_class);
    } else {  // features2 -> mlp_pointnet (f32: an input, not bounded by the weights) -> normalize
      for (int o = lw; o < kSP; o += 4) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (o < my_nobj) v = reinterpret_cast<const float4*>(in.pn_feat + (size_t)(my_obj0 + o) * kD)[lane];
        reinterpret_cast<float4*>(bf32 + o * kLdX)[lane] = v;
      }
      __syncthreads();
      const float* pb = P.pn_b;
      gemm32(bf32, kLdX, kD, P.pn_wp, kD, lw, lane, [&](int, int, int row, int cc, float v) { xf32[row * kLdX + cc] = fmaxf(v + pb[cc], 0.f); });
      __syncthreads();
      normalize_rows(xf32, kLdX, my_nobj, lw, lane);
      __syncthreads();
      tile_to_planes();
    }
    merge_slot();
  }
  if (P.use_color) {
    if (P.color_embed) {
      table_to_planes(P.color_tab, in.color_idx, P.n_color);
    } else {
      small_mlp<3>(P.color, in.rgb + (size_t)my_obj0 * 3, false, my_nobj, hb, xf32, ltid, lw, lane);
      __syncthreads();
      tile_to_planes();
    }
    merge_slot();
  }
  if (P.use_pos) {
    small_mlp<3>(P.pos, in.center + (size_t)my_obj0 * 3, false, my_nobj, hb, xf32, ltid, lw, lane);
    __syncthreads();
    tile_to_planes();
    merge_slot();
  }
  if (P.use_num) {
    small_mlp<1>(P.num, in.n_pts + my_obj0, true, my_nobj, hb, xf32, ltid, lw, lane);
    __syncthreads();
    tile_to_planes();
    merge_slot();
  }
  {  // merge epilogue: relu(keep + b) -> X planes (lane = object slot, register quad = 4 consecutive features)
    const float* mb = P.merge_b;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f0 = 32 * wave + 8 * q + 4 * half;
        const float4 bb = *reinterpret_cast<const float4*>(mb + f0);
        const ti_f32x4 v = {fmaxf(keep[t][4 * q] + bb.x, 0.f), fmaxf(keep[t][4 * q + 1] + bb.y, 0.f), fmaxf(keep[t][4 * q + 2] + bb.z, 0.f),
                            fmaxf(keep[t][4 * q + 3] + bb.w, 0.f)};
        plane_put4<SG>(pl_xh(base, t), pl_xl(base, t), col * kLdP + f0, v);
      }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {  // F.normalize per object row; rows >= nobj are the zero pad slots (cell_retrieval.py:85,92)
    const int r = wave * 8 + i, t = r >> 5, lr = r & 31, off = lr * kLdP + 4 * lane;
    ti_f32x4 v = plane_get4<SG>(pl_xh(base, t), pl_xl(base, t), off);
    const float ss = wave_sum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
    const float inv = lr < nobj[t] ? 1.f / fmaxf(sqrtf(ss), 1e-12f) : 0.f;
    v *= inv;
    plane_put4<SG>(pl_xh(base, t), pl_xl(base, t), off, v);
  }
  __syncthreads();
  bool bad = false;
  for (int l = 0; l < P.num_layers; ++l) {
    const LayerW& L = P.layer[l];
    InterFusedW W;
    W.in_hp = L.in_hp; W.out_hp = L.out_hp; W.ff1_hp = L.ff1_hp; W.ff2_hp = L.ff2_hp;
    W.in_b = L.in_b; W.out_b = L.out_b; W.ff1_b = L.ff1_b; W.ff2_b = L.ff2_b;
    W.ln1_w = L.ln1_w; W.ln1_b = L.ln1_b; W.ln2_w = L.ln2_w; W.ln2_b = L.ln2_b;
    // no padding mask: the 28 slots, zero pads included, attend and are attended to; the 4 dead rows of the tile are no keys
    planes_layer<SG, 2, false>(base, W, [](int, int j) { return j < kS; }, bad, l == P.num_layers - 1, hb_all);  // (hb_all: dead behind the feature MLPs)
  }
  {  // max over ALL 28 slots, then normalize: thread = (cell, column)
    const int t = tid >> 8, c = tid & 255;
    const float* y = reinterpret_cast<const float*>(pl_bh(base, t));
    float mx = y[c];
    for (int i = 1; i < kS; ++i) mx = fmaxf(mx, y[i * kLdX + c]);
    const float ss = wave_sum(mx * mx);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    const float nrm = sqrtf(red[4 * t] + red[4 * t + 1] + red[4 * t + 2] + red[4 * t + 3]);
    if ((int)blockIdx.x * 2 + t < in.n_cells) out[(size_t)cell[t] * kD + c] = mx / fmaxf(nrm, 1e-12f);
  }
}

int text_inter_fused_launch(t2l_ctx* ctx, const InterFusedW& W, bool single, const float* sent, int n_desc, int S, float* out, int* flag,
                            hipStream_t s) {
  {
    const int dpt = kSP / S, tiles = (n_desc + 2 * dpt - 1) / (2 * dpt);
    const size_t lds = (size_t)8 * kPlane * sizeof(_Float16) + kSP * sizeof(int) + 1024 * sizeof(float);
    static PerDeviceOnce once2;
    if (once2.need(ctx->device)) {
      T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&text_inter_fused2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&text_inter_fused2_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      once2.mark(ctx->device);
    }
    if (single)
      hipLaunchKernelGGL(text_inter_fused2_kernel<2>, dim3(tiles), dim3(512), lds, s, W, sent, n_desc, S, dpt, out, flag);
    else
      hipLaunchKernelGGL(text_inter_fused2_kernel<1>, dim3(tiles), dim3(512), lds, s, W, sent, n_desc, S, dpt, out, flag);
    T2L_HIP(ctx, hipGetLastError());
    return T2L_OK;
  }
}

// ---- host side: BN folding, fragment packing, upload -------------------------------------------
namespace {

struct Blob {
  std::vector<float> h;
  size_t add(const std::vector<float>& v) {
    const size_t off = (h.size() + 3) / 4 * 4;  // 16-byte aligned
    h.resize(off);
    h.insert(h.end(), v.begin(), v.end());
    return off;
  }
};

using WMap = std::unordered_map<std::string, const t2l_weight_desc*>;

const float* need(t2l_ctx* ctx, const WMap& m, const std::string& name, int64_t numel, int* rc) {
  auto it = m.find(name);
  if (it == m.end()) {
    *rc = fail(ctx, T2L_EINVAL, "t2l_load_weights: missing required key " + name);
    return nullptr;
  }
  if (numel > 0 && it->second->numel != numel) {
    *rc = fail(ctx, T2L_EINVAL, "t2l_load_weights: wrong size for " + name);
    return nullptr;
  }
  return it->second->data;
}

// Linear(+BatchNorm1d eval) -> W' [out][in], b' [out]
bool fold(t2l_ctx* ctx, const WMap& m, const std::string& lin, const std::string& bn, int out, int in,
          std::vector<float>* W, std::vector<float>* b, int* rc) {
  const float* w = need(ctx, m, lin + ".weight", (int64_t)out * in, rc);
  const float* bb = need(ctx, m, lin + ".bias", out, rc);
  if (!w || !bb) return false;
  W->assign(w, w + (size_t)out * in);
  b->assign(bb, bb + out);
  if (!bn.empty()) {
    const float* g = need(ctx, m, bn + ".weight", out, rc);
    const float* be = need(ctx, m, bn + ".bias", out, rc);
    const float* rm = need(ctx, m, bn + ".running_mean", out, rc);
    const float* rv = need(ctx, m, bn + ".running_var", out, rc);
    if (!g || !be || !rm || !rv) return false;
    for (int o = 0; o < out; ++o) {
      // float32 arithmetic in the order torch evaluates eval-mode BN: (x - mean) / sqrt(var + eps) * w + b
      const float s = g[o] / sqrtf(rv[o] + 1e-5f);
      for (int k = 0; k < in; ++k) (*W)[(size_t)o * in + k] *= s;
      (*b)[o] = ((*b)[o] - rm[o]) * s + be[o];
    }
  }
  return true;
}

// W [N][K] row-major -> [N/32][K/8][64 lanes][4]: lane (col, half) of step q holds W[nt*32+col][half*K/2 + 4q .. +3]
std::vector<float> pack(const std::vector<float>& W, int N, int K) {
  std::vector<float> p((size_t)N * K);
  const int qn = K / 8;
  for (int nt = 0; nt < N / 32; ++nt)
    for (int q = 0; q < qn; ++q)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e)
          p[(((size_t)nt * qn + q) * 64 + lane) * 4 + e] =
              W[(size_t)(nt * 32 + (lane & 31)) * K + (lane >> 5) * (K / 2) + 4 * q + e];
  return p;
}

std::vector<float> pack_h(const std::vector<float>& W, int N, int K) { return pack_split_f16(W.data(), nullptr, N, K, K); }
float max_abs(const float* v, size_t n) { return h3_max_abs(v, n); }
float max_row_norm(const float* W, int rows, int cols) { return h3_max_row_norm(W, rows, cols); }

std::vector<float> normalized_rows(const float* t, int rows) {
  std::vector<float> o((size_t)rows * kD);
  for (int r = 0; r < rows; ++r) {
    float ss = 0.f;
    for (int c = 0; c < kD; ++c) ss += t[r * kD + c] * t[r * kD + c];
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    for (int c = 0; c < kD; ++c) o[(size_t)r * kD + c] = t[r * kD + c] * inv;
  }
  return o;
}

struct SmallOff {
  size_t w1, b1, w2p, b2;
};

bool add_small(t2l_ctx* ctx, const WMap& m, const std::string& pre, int in, Blob* blob, SmallOff* off, int* rc) {
  std::vector<float> W1, b1, W2, b2;
  if (!fold(ctx, m, pre + ".0.0", pre + ".0.1", 64, in, &W1, &b1, rc)) return false;
  if (!fold(ctx, m, pre + ".1.0", pre + ".1.1", kD, 64, &W2, &b2, rc)) return false;
  off->w1 = blob->add(W1);
  off->b1 = blob->add(b1);
  off->w2p = blob->add(pack(W2, kD, 64));
  off->b2 = blob->add(b2);
  return true;
}

}  // namespace

void free_weights(t2l_ctx* ctx) {
  if (!ctx->enc) return;
  if (ctx->enc->blob) (void)hipFree(ctx->enc->blob);
  delete ctx->enc;
  ctx->enc = nullptr;
}

int load_weights_impl(t2l_ctx* ctx, const t2l_weight_desc* w, int n, const t2l_model_config* cfg) {
  if (cfg->num_heads != 4) return fail(ctx, T2L_EINVAL, "t2l_load_weights: only num_heads == 4 (head_dim 64) is built");
  if (cfg->num_layers < 1 || cfg->num_layers > 4) return fail(ctx, T2L_EINVAL, "t2l_load_weights: num_layers must be 1..4");
  const int nfeat = (cfg->use_class != 0) + (cfg->use_color != 0) + (cfg->use_position != 0) + (cfg->use_num != 0);
  if (nfeat < 1) return fail(ctx, T2L_EINVAL, "t2l_load_weights: use_features is empty");
  WMap m;
  for (int i = 0; i < n; ++i) {
    if (!w[i].name || !w[i].data) return fail(ctx, T2L_EINVAL, "t2l_load_weights: null name/data");
    m[w[i].name] = &w[i];
  }
  int rc = T2L_OK;
  Blob blob;
  const std::string oe = "object_encoder.";
  size_t class_tab = 0, color_tab = 0, pn_wp = 0, pn_b = 0, merge_wp = 0, merge_hp = 0, merge_b = 0;
  // split-f16 safety (file header): largest weight, and an upper bound on |activation| entering a split GEMM
  float w_absmax = 0.f, act_bound = 1.f;  // merge inputs are unit rows
  int n_class = 0, n_color = 0;
  SmallOff pos{}, color{}, num{};
  if (cfg->use_class) {
    if (cfg->class_embed) {
      auto it = m.find(oe + "class_embedding.weight");
      if (it == m.end() || it->second->numel % kD) return fail(ctx, T2L_EINVAL, "missing/odd class_embedding.weight");
      n_class = (int)(it->second->numel / kD);
      class_tab = blob.add(normalized_rows(it->second->data, n_class));
    } else {
      std::vector<float> W, b;
      if (!fold(ctx, m, oe + "mlp_pointnet.0.0", oe + "mlp_pointnet.0.1", kD, kD, &W, &b, &rc)) return rc;
      pn_wp = blob.add(pack(W, kD, kD));
      pn_b = blob.add(b);
      w_absmax = fmaxf(w_absmax, max_abs(W.data(), W.size()));
    }
  }
  if (cfg->use_color) {
    if (cfg->color_embed) {
      auto it = m.find(oe + "color_embedding.weight");
      if (it == m.end() || it->second->numel % kD) return fail(ctx, T2L_EINVAL, "missing/odd color_embedding.weight");
      n_color = (int)(it->second->numel / kD);
      color_tab = blob.add(normalized_rows(it->second->data, n_color));
    } else if (!add_small(ctx, m, oe + "color_encoder", 3, &blob, &color, &rc)) {
      return rc;
    }
  }
  if (cfg->use_position && !add_small(ctx, m, oe + "pos_encoder", 3, &blob, &pos, &rc)) return rc;
  if (cfg->use_num && !add_small(ctx, m, oe + "num_encoder", 1, &blob, &num, &rc)) return rc;
  if (nfeat > 1) {
    std::vector<float> W, b;
    if (!fold(ctx, m, oe + "mlp_merge.0.0", oe + "mlp_merge.0.1", kD, nfeat * kD, &W, &b, &rc)) return rc;
    {  // one (N=256, K=256) packing per feature slot, consecutive
      std::vector<float> all, all_h;
      for (int sl = 0; sl < nfeat; ++sl) {
        std::vector<float> Ws((size_t)kD * kD);
        for (int n = 0; n < kD; ++n)
          for (int k = 0; k < kD; ++k) Ws[(size_t)n * kD + k] = W[(size_t)n * nfeat * kD + sl * kD + k];
        const std::vector<float> ps = pack(Ws, kD, kD);
        all.insert(all.end(), ps.begin(), ps.end());
        const std::vector<float> ph = pack_h(Ws, kD, kD);
        all_h.insert(all_h.end(), ph.begin(), ph.end());
      }
      merge_wp = blob.add(all);
      merge_hp = blob.add(all_h);
      w_absmax = fmaxf(w_absmax, max_abs(W.data(), W.size()));
    }
    merge_b = blob.add(b);
  }
  struct LOff {
    size_t in_wp, in_b, out_wp, out_b, ff1_wp, ff1_b, ff2_wp, ff2_b, ln1_w, ln1_b, ln2_w, ln2_b;
    size_t in_hp, out_hp, ff1_hp, ff2_hp;
  } lo[4];
  float x_norm = 1.f;  // bound on the 2-norm of a token row entering the layer (layer 0: unit rows or zero pads)
  for (int l = 0; l < cfg->num_layers; ++l) {
    const std::string p = "obj_inter_module." + std::to_string(l) + ".";
    auto lin = [&](const std::string& wn, const std::string& bn, int out, int in, size_t* wp, size_t* hp, size_t* bo) -> bool {
      const float* W = need(ctx, m, p + wn, (int64_t)out * in, &rc);
      const float* b = need(ctx, m, p + bn, out, &rc);
      if (!W || !b) return false;
      const std::vector<float> Wv(W, W + (size_t)out * in);
      *wp = blob.add(pack(Wv, out, in));
      *hp = blob.add(pack_h(Wv, out, in));
      *bo = blob.add(std::vector<float>(b, b + out));
      w_absmax = fmaxf(w_absmax, max_abs(W, Wv.size()));
      return true;
    };
    auto vec = [&](const std::string& name, size_t* o) -> bool {
      const float* v = need(ctx, m, p + name, kD, &rc);
      if (!v) return false;
      *o = blob.add(std::vector<float>(v, v + kD));
      return true;
    };
    if (!lin("self_attn.in_proj_weight", "self_attn.in_proj_bias", 3 * kD, kD, &lo[l].in_wp, &lo[l].in_hp, &lo[l].in_b)) return rc;
    if (!lin("self_attn.out_proj.weight", "self_attn.out_proj.bias", kD, kD, &lo[l].out_wp, &lo[l].out_hp, &lo[l].out_b)) return rc;
    if (!lin("linear1.weight", "linear1.bias", 2 * kD, kD, &lo[l].ff1_wp, &lo[l].ff1_hp, &lo[l].ff1_b)) return rc;
    if (!lin("linear2.weight", "linear2.bias", kD, 2 * kD, &lo[l].ff2_wp, &lo[l].ff2_hp, &lo[l].ff2_b)) return rc;
    if (!vec("norm1.weight", &lo[l].ln1_w) || !vec("norm1.bias", &lo[l].ln1_b) || !vec("norm2.weight", &lo[l].ln2_w) ||
        !vec("norm2.bias", &lo[l].ln2_b))
      return rc;
    {  // bounds on what enters this layer's split GEMMs (|LayerNorm(.)| <= sqrt(255) |gain| + |bias| per element)
      const float* Wi = m.at(p + "self_attn.in_proj_weight")->data;
      const float* bi = m.at(p + "self_attn.in_proj_bias")->data;
      const float* W1 = m.at(p + "linear1.weight")->data;
      const float* b1 = m.at(p + "linear1.bias")->data;
      auto ln_elem = [&](const char* wn, const char* bn) {
        return 16.f * max_abs(m.at(p + wn)->data, kD) + max_abs(m.at(p + bn)->data, kD);
      };
      act_bound = fmaxf(act_bound, x_norm);                                                                   // q/k/v input
      act_bound = fmaxf(act_bound, x_norm * max_row_norm(Wi, 2 * kD, kD) + max_abs(bi, 2 * kD));              // q, k: operands of the split S = K Q^T
      act_bound = fmaxf(act_bound, x_norm * max_row_norm(Wi + (size_t)2 * kD * kD, kD, kD) + max_abs(bi + 2 * kD, kD));  // v (operand of P V) and out_proj input: convex combinations of v
      const float ln1 = ln_elem("norm1.weight", "norm1.bias");
      act_bound = fmaxf(act_bound, ln1);                                                                      // linear1 input (element bound)
      act_bound = fmaxf(act_bound, 16.f * ln1 * max_row_norm(W1, 2 * kD, kD) + max_abs(b1, 2 * kD));          // linear2 input
      x_norm = 16.f * ln_elem("norm2.weight", "norm2.bias");                                                 // next layer's rows
    }
  }

  free_weights(ctx);
  EncoderWeights* ew = new EncoderWeights();
  if (hipMalloc(&ew->blob, blob.h.size() * sizeof(float)) != hipSuccess) {
    delete ew;
    return fail(ctx, T2L_ENOMEM, "t2l_load_weights: hipMalloc failed");
  }
  if (hipMemcpy(ew->blob, blob.h.data(), blob.h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(ew->blob);
    delete ew;
    return fail(ctx, T2L_EHIP, "t2l_load_weights: upload failed");
  }
  const float* B = ew->blob;
  auto f4 = [&](size_t off) { return reinterpret_cast<const float4*>(B + off); };
  EncParams& P = ew->p;
  memset(&P, 0, sizeof(P));
  P.class_tab = B + class_tab;
  P.color_tab = B + color_tab;
  P.n_class = n_class;
  P.n_color = n_color;
  auto sm = [&](const SmallOff& o) { return SmallMlp{B + o.w1, B + o.b1, f4(o.w2p), B + o.b2}; };
  P.pos = sm(pos);
  P.color = sm(color);
  P.num = sm(num);
  auto u4 = [&](size_t off) { return reinterpret_cast<const uint4*>(B + off); };
  P.pn_wp = f4(pn_wp);
  P.pn_hp = nullptr;
  P.pn_b = B + pn_b;
  P.merge_wp = f4(merge_wp);
  P.merge_hp = u4(merge_hp);
  P.merge_b = B + merge_b;
  for (int l = 0; l < cfg->num_layers; ++l)
    P.layer[l] = LayerW{f4(lo[l].in_wp),  f4(lo[l].out_wp), f4(lo[l].ff1_wp), f4(lo[l].ff2_wp),
                        u4(lo[l].in_hp),  u4(lo[l].out_hp), u4(lo[l].ff1_hp), u4(lo[l].ff2_hp),
                        B + lo[l].in_b,   B + lo[l].out_b,  B + lo[l].ff1_b,  B + lo[l].ff2_b,
                        B + lo[l].ln1_w,  B + lo[l].ln1_b,  B + lo[l].ln2_w,  B + lo[l].ln2_b};
  P.split_ok = (w_absmax < kSplitF16Safe && act_bound < kSplitF16Safe) ? 1 : 0;
  P.num_layers = cfg->num_layers;
  P.class_embed = cfg->class_embed;
  P.color_embed = cfg->color_embed;
  P.use_class = cfg->use_class;
  P.use_color = cfg->use_color;
  P.use_pos = cfg->use_position;
  P.use_num = cfg->use_num;
  P.nfeat = nfeat;
  ctx->enc = ew;
  return T2L_OK;
}

int encode_impl(t2l_ctx* ctx, const t2l_packed_cells* in, float* out, hipStream_t s) {
  const EncParams& P = ctx->enc->p;
  if (P.use_class && !P.class_embed && !in->pn_feat)
    return fail(ctx, T2L_EINVAL, "t2l_encode_cells: pn_feat required when class_embed == 0");
  if ((P.use_class && P.class_embed && !in->class_idx) || (P.use_color && P.color_embed && !in->color_idx) ||
      (P.use_color && !P.color_embed && !in->rgb) || (P.use_pos && !in->center) || (P.use_num && !in->n_pts))
    return fail(ctx, T2L_EINVAL, "t2l_encode_cells: a per-object input required by the loaded config is NULL");
  const size_t lds = (size_t)(2 * kXFloats + 8) * sizeof(float);  // 66.6 KB: two cells per CU
  static PerDeviceOnce attr_done;
  if (attr_done.need(ctx->device)) {
    T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&encode_cells_kernel<1>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&encode_cells_kernel<2>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&encode_cells_kernel<0>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done.mark(ctx->device);
  }
  event_begin(ctx, "encode_cells", s);
  if (P.split_ok && !ctx->encoder_f32 && ctx->encoder_two_cells && P.nfeat > 1) {  // second form: two cells per eight-wave workgroup on planes
    const size_t lds2 = (size_t)8 * kPlane * sizeof(_Float16) + (size_t)(2 * kSP * kLdH + 8) * sizeof(float);
    static PerDeviceOnce attr2;
    if (attr2.need(ctx->device)) {
      T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&encode_cells2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&encode_cells2_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      attr2.mark(ctx->device);
    }
    if (ctx->encoder_f16)
      hipLaunchKernelGGL(encode_cells2_kernel<2>, dim3((in->n_cells + 1) / 2), dim3(512), lds2, s, P, *in, out);
    else
      hipLaunchKernelGGL(encode_cells2_kernel<1>, dim3((in->n_cells + 1) / 2), dim3(512), lds2, s, P, *in, out);
    event_end(ctx, "encode_cells", s);
    T2L_HIP(ctx, hipGetLastError());
    return T2L_OK;
  }
  // encoder_f16 (option, off by default): ONE f16 product per operand pair instead of the three of the split form — embeddings
  // within ~1e-4 of the reference's (the north star asks for 1e-3) instead of 2e-7, 28 % less time
  if (P.split_ok && !ctx->encoder_f32 && ctx->encoder_f16)
    hipLaunchKernelGGL(encode_cells_kernel<2>, dim3(in->n_cells), dim3(256), lds, s, P, *in, out);
  else if (P.split_ok && !ctx->encoder_f32)
    hipLaunchKernelGGL(encode_cells_kernel<1>, dim3(in->n_cells), dim3(256), lds, s, P, *in, out);
  else
    hipLaunchKernelGGL(encode_cells_kernel<0>, dim3(in->n_cells), dim3(256), lds, s, P, *in, out);
  event_end(ctx, "encode_cells", s);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

}  // namespace t2l
