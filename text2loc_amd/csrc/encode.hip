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
// All contractions are f32 MFMA (v_mfma_f32_32x32x2_f32: exact f32 FMA chains; 1e-3 parity needs f32).
// M = 32 rows = the 28 slots + 4 dead rows (masked out of the softmax keys and the max-pool).
// Weights are BN-folded and re-laid out on the host into MFMA B-fragment order
// [n_tile][k_step][lane][4] so every wave-level weight load is one coalesced 1 KiB line.
// The MFMA's k-sum is order-free, so lane (col, half) owns k in [half*K/2, (half+1)*K/2): both operands
// are read as contiguous float4 (LDS rows padded by 4 floats -> conflict-free ds_read_b128).
#include <math.h>
#include <string.h>

#include "t2l_internal.h"

namespace t2l {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kLdX = kD + 4;        // 260
constexpr int kLdQ = 3 * kD + 4;    // 772  (qkv / ff / o buffer)
constexpr int kLdF = 4 * kD + 4;    // 1028 (concatenated features)
constexpr int kLdH = 64 + 4;        // 68   (hidden layer of the small MLPs)
constexpr int kXFloats = kSP * kLdX;
constexpr int kQFloats = kSP * kLdQ;
constexpr int kHFloats = kSP * kLdH;
// feats [32][1028] is overlaid on (qkv, x): 32*1028 <= 32*772 + 32*260
static_assert(kSP * kLdF <= kQFloats + kXFloats, "feature overlay must fit");
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
  const float *in_b, *out_b, *ff1_b, *ff2_b, *ln1_w, *ln1_b, *ln2_w, *ln2_b;
};

struct EncParams {
  const float* class_tab;  // [n_class][256] rows already L2-normalised
  const float* color_tab;  // [n_color][256]
  int n_class, n_color;
  SmallMlp pos, color, num;
  const float4* pn_wp;     // mlp_pointnet packed (N=256,K=256)
  const float* pn_b;
  const float4* merge_wp;  // packed (N=256, K=256*nfeat)
  const float* merge_b;
  LayerW layer[4];
  int num_layers;
  int class_embed, color_embed, use_class, use_color, use_pos, use_num, nfeat;
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
                                          bool is_num, int nobj, float* hbuf, float* dst /* feats slot */, int tid,
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
         [&](int, int, int row, int col, float v) { dst[row * kLdF + col] = fmaxf(v + b2[col], 0.f); });
  __syncthreads();
  normalize_rows(dst, kLdF, nobj, wave, lane);
}

__global__ __launch_bounds__(256, 1) void encode_cells_kernel(EncParams P, t2l_packed_cells in,
                                                              float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* qkv = smem;                 // [32][772]
  float* x = smem + kQFloats;        // [32][260]
  float* hbuf = x + kXFloats;        // [32][68]
  float* red = hbuf + kHFloats;      // [8]
  float* feats = smem;               // [32][1028] overlay on (qkv, x), only before x exists

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int cell = blockIdx.x;
  const int obj0 = in.offsets[cell];
  const int nobj = min(in.offsets[cell + 1] - obj0, kS);  // objects beyond 28 are dropped (cell_retrieval.py:94-98)

  // ------------------------------------------------------------------ per-object features
  int slot = 0;
  if (P.use_class) {
    float* dst = feats + slot * kD;
    if (P.class_embed) {  // object_encoder.py:103-110 (table rows pre-normalised on the host)
      for (int o = 0; o < kSP; ++o) {
        float v = 0.f;
        if (o < nobj) {
          const int ci = min(max(in.class_idx[obj0 + o], 0), P.n_class - 1);
          v = P.class_tab[ci * kD + tid];
        }
        dst[o * kLdF + tid] = v;
      }
    } else {  // object_encoder.py:86-99,112: features2 -> mlp_pointnet -> normalize
      float* stage = feats + 3 * kD;  // park features2 in the last feature slot (rewritten later)
      for (int o = wave; o < kSP; o += 4) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (o < nobj) v = reinterpret_cast<const float4*>(in.pn_feat + (size_t)(obj0 + o) * kD)[lane];
        reinterpret_cast<float4*>(stage + o * kLdF)[lane] = v;
      }
      __syncthreads();
      const float* pb = P.pn_b;
      gemm32(stage, kLdF, kD, P.pn_wp, kD, wave, lane,
             [&](int, int, int row, int c, float v) { dst[row * kLdF + c] = fmaxf(v + pb[c], 0.f); });
      __syncthreads();
      normalize_rows(dst, kLdF, nobj, wave, lane);
    }
    ++slot;
    __syncthreads();
  }
  if (P.use_color) {
    float* dst = feats + slot * kD;
    if (P.color_embed) {  // object_encoder.py:116-120
      for (int o = 0; o < kSP; ++o) {
        float v = 0.f;
        if (o < nobj) {
          const int ci = min(max(in.color_idx[obj0 + o], 0), P.n_color - 1);
          v = P.color_tab[ci * kD + tid];
        }
        dst[o * kLdF + tid] = v;
      }
    } else {  // object_encoder.py:121-128
      small_mlp<3>(P.color, in.rgb + (size_t)obj0 * 3, false, nobj, hbuf, dst, tid, wave, lane);
    }
    ++slot;
    __syncthreads();
  }
  if (P.use_pos) {  // object_encoder.py:130-136
    small_mlp<3>(P.pos, in.center + (size_t)obj0 * 3, false, nobj, hbuf, feats + slot * kD, tid, wave, lane);
    ++slot;
    __syncthreads();
  }
  if (P.use_num) {  // object_encoder.py:138-145
    small_mlp<1>(P.num, in.n_pts + obj0, true, nobj, hbuf, feats + slot * kD, tid, wave, lane);
    ++slot;
    __syncthreads();
  }

  // ------------------------------------------------------------------ merge (object_encoder.py:148-149) + normalize (cell_retrieval.py:92)
  {
    float keep[2][16];
    if (P.nfeat > 1) {
      const float* mb = P.merge_b;
      gemm32(feats, kLdF, P.nfeat * kD, P.merge_wp, kD, wave, lane,
             [&](int t, int r, int, int c, float v) { keep[t][r] = fmaxf(v + mb[c], 0.f); });
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        keep[0][r] = feats[row * kLdF + wave * 32 + col];
        keep[1][r] = feats[row * kLdF + (wave + 4) * 32 + col];
      }
    }
    __syncthreads();  // every wave is done reading feats; x may now be written over its tail
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      x[row * kLdX + wave * 32 + col] = keep[0][r];
      x[row * kLdX + (wave + 4) * 32 + col] = keep[1][r];
    }
  }
  __syncthreads();
  normalize_rows(x, kLdX, nobj, wave, lane);  // rows >= nobj become the zero pad slots (cell_retrieval.py:85)
  __syncthreads();

  // ------------------------------------------------------------------ set transformer (cell_retrieval.py:101-103)
  for (int l = 0; l < P.num_layers; ++l) {
    const LayerW& W = P.layer[l];
    {  // qkv = x @ in_proj^T + b
      const float* b = W.in_b;
      gemm32(x, kLdX, kD, W.in_wp, 3 * kD, wave, lane,
             [&](int, int, int row, int c, float v) { qkv[row * kLdQ + c] = v + b[c]; });
    }
    __syncthreads();
    {  // head h = wave: S^T[j][i] = k_j . q_i ; softmax over keys j in-lane ; o = P v
      const int h = wave;
      const float* kr = qkv + col * kLdQ + kD + h * 64 + half * 32;
      const float* qr = qkv + col * kLdQ + h * 64 + half * 32;
      f32x16 st;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(kr + 4 * q);
        const float4 b = *reinterpret_cast<const float4*>(qr + 4 * q);
        st = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, st, 0, 0, 0);
        st = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, st, 0, 0, 0);
        st = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, st, 0, 0, 0);
        st = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, st, 0, 0, 0);
      }
      // lane: query i = col, keys j = (r&3) + 8*(r>>2) + 4*half ; keys >= 28 are the dead rows
      float m = -__builtin_inff();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = (r & 3) + 8 * (r >> 2) + 4 * half;
        st[r] = (j < kS) ? st[r] * 0.125f : -__builtin_inff();  // 1/sqrt(head_dim = 64)
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
      const float inv = 1.f / sum;
      // o[i][n] = sum_j P[i][j] v[j][n]: MFMA step r pairs keys jA(r) (half 0) and jA(r)+4 (half 1)
      f32x16 o0, o1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o0[r] = 0.f;
        o1[r] = 0.f;
      }
      const float* vb = qkv + 2 * kD + h * 64 + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = (r & 3) + 8 * (r >> 2) + 4 * half;
        const float p = st[r] * inv;
        o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(p, vb[j * kLdQ], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(p, vb[j * kLdQ + 32], o1, 0, 0, 0);
      }
      // o_h overwrites this head's own q columns (q_h is dead once S is formed)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
        qkv[i * kLdQ + h * 64 + col] = o0[r];
        qkv[i * kLdQ + h * 64 + 32 + col] = o1[r];
      }
    }
    __syncthreads();
    {  // x = LN1(x + o @ out_proj^T + b)
      const float* b = W.out_b;
      gemm32(qkv, kLdQ, kD, W.out_wp, kD, wave, lane,
             [&](int, int, int row, int c, float v) { x[row * kLdX + c] += v + b[c]; });
    }
    __syncthreads();
    layer_norm_rows(x, W.ln1_w, W.ln1_b, wave, lane);
    __syncthreads();
    {  // ff = relu(x @ W1^T + b1)  -> qkv buffer
      const float* b = W.ff1_b;
      gemm32(x, kLdX, kD, W.ff1_wp, 2 * kD, wave, lane,
             [&](int, int, int row, int c, float v) { qkv[row * kLdQ + c] = fmaxf(v + b[c], 0.f); });
    }
    __syncthreads();
    {  // x = LN2(x + ff @ W2^T + b2)
      const float* b = W.ff2_b;
      gemm32(qkv, kLdQ, 2 * kD, W.ff2_wp, kD, wave, lane,
             [&](int, int, int row, int c, float v) { x[row * kLdX + c] += v + b[c]; });
    }
    __syncthreads();
    layer_norm_rows(x, W.ln2_w, W.ln2_b, wave, lane);
    __syncthreads();
  }

  // ------------------------------------------------------------------ max over ALL 28 slots, pads included (cell_retrieval.py:107-108)
  float mx = x[tid];
  for (int i = 1; i < kS; ++i) mx = fmaxf(mx, x[i * kLdX + tid]);
  const float ss = wave_sum(mx * mx);
  if (lane == 0) red[wave] = ss;
  __syncthreads();
  const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
  out[(size_t)cell * kD + tid] = mx / fmaxf(nrm, 1e-12f);
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
  size_t class_tab = 0, color_tab = 0, pn_wp = 0, pn_b = 0, merge_wp = 0, merge_b = 0;
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
    merge_wp = blob.add(pack(W, kD, nfeat * kD));
    merge_b = blob.add(b);
  }
  struct LOff {
    size_t in_wp, in_b, out_wp, out_b, ff1_wp, ff1_b, ff2_wp, ff2_b, ln1_w, ln1_b, ln2_w, ln2_b;
  } lo[4];
  for (int l = 0; l < cfg->num_layers; ++l) {
    const std::string p = "obj_inter_module." + std::to_string(l) + ".";
    auto lin = [&](const std::string& wn, const std::string& bn, int out, int in, size_t* wp, size_t* bo) -> bool {
      const float* W = need(ctx, m, p + wn, (int64_t)out * in, &rc);
      const float* b = need(ctx, m, p + bn, out, &rc);
      if (!W || !b) return false;
      *wp = blob.add(pack(std::vector<float>(W, W + (size_t)out * in), out, in));
      *bo = blob.add(std::vector<float>(b, b + out));
      return true;
    };
    auto vec = [&](const std::string& name, size_t* o) -> bool {
      const float* v = need(ctx, m, p + name, kD, &rc);
      if (!v) return false;
      *o = blob.add(std::vector<float>(v, v + kD));
      return true;
    };
    if (!lin("self_attn.in_proj_weight", "self_attn.in_proj_bias", 3 * kD, kD, &lo[l].in_wp, &lo[l].in_b)) return rc;
    if (!lin("self_attn.out_proj.weight", "self_attn.out_proj.bias", kD, kD, &lo[l].out_wp, &lo[l].out_b)) return rc;
    if (!lin("linear1.weight", "linear1.bias", 2 * kD, kD, &lo[l].ff1_wp, &lo[l].ff1_b)) return rc;
    if (!lin("linear2.weight", "linear2.bias", kD, 2 * kD, &lo[l].ff2_wp, &lo[l].ff2_b)) return rc;
    if (!vec("norm1.weight", &lo[l].ln1_w) || !vec("norm1.bias", &lo[l].ln1_b) || !vec("norm2.weight", &lo[l].ln2_w) ||
        !vec("norm2.bias", &lo[l].ln2_b))
      return rc;
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
  P.pn_wp = f4(pn_wp);
  P.pn_b = B + pn_b;
  P.merge_wp = f4(merge_wp);
  P.merge_b = B + merge_b;
  for (int l = 0; l < cfg->num_layers; ++l)
    P.layer[l] = LayerW{f4(lo[l].in_wp),  f4(lo[l].out_wp), f4(lo[l].ff1_wp), f4(lo[l].ff2_wp),
                        B + lo[l].in_b,   B + lo[l].out_b,  B + lo[l].ff1_b,  B + lo[l].ff2_b,
                        B + lo[l].ln1_w,  B + lo[l].ln1_b,  B + lo[l].ln2_w,  B + lo[l].ln2_b};
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
  const size_t lds = (size_t)(kQFloats + kXFloats + kHFloats + 8) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&encode_cells_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  event_begin(ctx, "encode_cells", s);
  hipLaunchKernelGGL(encode_cells_kernel, dim3(in->n_cells), dim3(256), lds, s, P, *in, out);
  event_end(ctx, "encode_cells", s);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

}  // namespace t2l
