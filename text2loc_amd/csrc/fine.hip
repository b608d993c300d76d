// Fine stage, eval mode: CrossMatch.forward downstream of the text branch
// (models/cross_matcher.py:86-135) —
//   t2l_fine_encode_objects : ObjectEncoder at fine_embed_dim (=128) + F.normalize per padded cell (16 objects)
//   t2l_fine_match          : per (query, cell) pair the cascaded cross-attention decoder layers
//                             (cross_objects[i](obj, hints); cross_hints[i](hints, obj)), max over hints, mlp_offsets
// The reference runs one Python-level forward per query over its top-k cells (evaluation/pipeline.py:113-116) and
// re-encodes the same cells for every query that retrieved them; here the per-cell object descriptors are computed
// once per database cell and the pairs are ONE launch (one workgroup per pair, everything LDS-resident).
// The match kernel runs every token-wise Linear on MFMA tiles — split-f16 (mfma_h3.h: hi*hi + hi*lo + lo*hi on the f16
// MFMA, ~5e-7 relative, a fifth of the f32 MFMA's pipe time) when the weights bound the activations below the f16 range and
// the workgroup's raw input rows pass a run-time norm guard, the f32 MFMA otherwise — and each attention block from
// registers (see f_attention_regs); the per-cell object encoder (1.1 ms for the whole database, once) contracts on the vector ALU with
// weights stored transposed [K][N]. BatchNorm is folded on the host.
#include <math.h>
#include <string.h>

#include "t2l_internal.h"
#include "mfma32.h"
#include "mfma_h3.h"

namespace t2l {

constexpr int kFD = 128;        // args.fine_embed_dim
constexpr int kFObj = 16;       // args.pad_size
constexpr int kFHintMax = 8;    // >= args.num_mentioned (6); rows >= n_hints are zero and never attended
constexpr int kFHeads = 4, kFHd = kFD / kFHeads;
constexpr int kFS = kFD + 4;    // LDS row stride of 128-wide token buffers

struct FLinear {
  const float* wt;  // [K][N] (transposed, BatchNorm folded where one follows)
  const float* b;   // [N]
};
struct FPacked {
  const float4* w;  // half-split MFMA packing of W[N][K] (mfma32.h)
  const uint4* h;   // the same matrix as split-f16 fragments (mfma_h3.h)
  const float* b;   // [N]
};
struct FDecoder {
  FPacked sa_in, sa_out, ca_in, ca_out, l1, l2;
  const float *g1, *b1, *g2, *b2, *g3, *b3;
};
struct FineParams {
  // object encoder
  const float* class_emb;  // [rows][128] or null
  const float* color_emb;
  FLinear pn, col1, col2, pos1, pos2, num1, num2, merge;
  int class_embed, color_embed, use_class, use_color, use_pos, use_num, n_feat;
  int n_layers;
  FDecoder obj[4], hint[4];
  FLinear off0, off2;
  int split_ok;  // with input rows of 2-norm <= kFGuardNorm every activation entering a split-f16 product stays < 3e4
};
constexpr float kFGuardNorm = 64.f;  // run-time guard on the raw descriptor rows a workgroup loads (unit rows in the reference's pipeline)
struct FineWeights {
  FineParams p{};
  std::vector<void*> blobs;
  int32_t* wg_flags = nullptr;  // per-workgroup guard verdicts of the split-f16 match launch
  size_t flag_cap = 0;
};

void free_fine(t2l_ctx* ctx) {
  FineWeights* W = reinterpret_cast<FineWeights*>(ctx->fine);
  if (!W) return;
  for (void* b : W->blobs) (void)hipFree(b);
  if (W->wg_flags) (void)hipFree(W->wg_flags);
  delete W;
  ctx->fine = nullptr;
}

using WMap = std::unordered_map<std::string, const t2l_weight_desc*>;

static const float* fget(const WMap& m, const std::string& k, int64_t n) {
  auto it = m.find(k);
  return (it == m.end() || it->second->numel != n) ? nullptr : it->second->data;
}
static int fupload(t2l_ctx* ctx, FineWeights* W, const std::vector<float>& v, const float** dst) {
  float* d = nullptr;
  T2L_HIP(ctx, hipMalloc(&d, v.size() * sizeof(float)));
  W->blobs.push_back(d);
  T2L_HIP(ctx, hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  *dst = d;
  return T2L_OK;
}
// Linear [N,K] (+ optional eval BatchNorm `bn` folded in) -> transposed [K][N] + bias
static int flinear(t2l_ctx* ctx, FineWeights* W, const WMap& m, const std::string& lin, const std::string& bn, int K, int N, FLinear* out) {
  const float *w = fget(m, lin + ".weight", (int64_t)N * K), *b = fget(m, lin + ".bias", N);
  if (!w || !b) return fail(ctx, T2L_EINVAL, "t2l_fine_load_weights: missing or mis-shaped '" + lin + "'");
  std::vector<float> wt((size_t)K * N), bb(N);
  std::vector<float> s(N, 1.f), sh(N, 0.f);
  if (!bn.empty()) {
    const float *g = fget(m, bn + ".weight", N), *be = fget(m, bn + ".bias", N), *rm = fget(m, bn + ".running_mean", N), *rv = fget(m, bn + ".running_var", N);
    if (!g || !be || !rm || !rv) return fail(ctx, T2L_EINVAL, "t2l_fine_load_weights: missing or mis-shaped '" + bn + "'");
    for (int n = 0; n < N; ++n) {
      s[n] = g[n] / sqrtf(rv[n] + 1e-5f);
      sh[n] = be[n] - rm[n] * s[n];
    }
  }
  for (int n = 0; n < N; ++n) {
    for (int k = 0; k < K; ++k) wt[(size_t)k * N + n] = w[(size_t)n * K + k] * s[n];
    bb[n] = b[n] * s[n] + sh[n];
  }
  int rc;
  if ((rc = fupload(ctx, W, wt, &out->wt))) return rc;
  return fupload(ctx, W, bb, &out->b);
}
// Linear [N,K] as stored by torch -> half-split MFMA packing + bias (decoder layers: no BatchNorm)
static int fpacked(t2l_ctx* ctx, FineWeights* W, const WMap& m, const std::string& wname, const std::string& bname, int K, int N, FPacked* out) {
  const float *w = fget(m, wname, (int64_t)N * K), *b = fget(m, bname, N);
  if (!w || !b) return fail(ctx, T2L_EINVAL, "t2l_fine_load_weights: missing or mis-shaped '" + wname + "'");
  const std::vector<float> packed = pack_half_split(std::vector<float>(w, w + (size_t)N * K), nullptr, N, K, K);
  const float* d = nullptr;
  int rc;
  if ((rc = fupload(ctx, W, packed, &d))) return rc;
  out->w = reinterpret_cast<const float4*>(d);
  const float* dh = nullptr;
  if ((rc = fupload(ctx, W, pack_split_f16(w, nullptr, N, K, K), &dh))) return rc;
  out->h = reinterpret_cast<const uint4*>(dh);
  return fupload(ctx, W, std::vector<float>(b, b + N), &out->b);
}
static int fraw(t2l_ctx* ctx, FineWeights* W, const WMap& m, const std::string& k, int64_t n, const float** dst) {
  const float* p = fget(m, k, n);
  if (!p) return fail(ctx, T2L_EINVAL, "t2l_fine_load_weights: missing or mis-shaped '" + k + "'");
  return fupload(ctx, W, std::vector<float>(p, p + n), dst);
}

int fine_load_impl(t2l_ctx* ctx, const t2l_weight_desc* w, int n, const t2l_model_config* cfg) {
  if (!w || n <= 0 || !cfg) return fail(ctx, T2L_EINVAL, "t2l_fine_load_weights: null argument");
  if (cfg->num_heads != kFHeads || cfg->num_layers < 0 || cfg->num_layers > 4)
    return fail(ctx, T2L_EINVAL, "t2l_fine_load_weights: built for 4 decoder heads and 0..4 decoder layers (0 = the single cross_hints layer)");
  free_fine(ctx);
  FineWeights* W = new FineWeights();
  ctx->fine = W;
  FineParams& P = W->p;
  WMap m;
  for (int i = 0; i < n; ++i)
    if (w[i].name) m[w[i].name] = &w[i];
  P.class_embed = cfg->class_embed; P.color_embed = cfg->color_embed;
  P.use_class = cfg->use_class; P.use_color = cfg->use_color; P.use_pos = cfg->use_position; P.use_num = cfg->use_num;
  P.n_feat = (P.use_class != 0) + (P.use_color != 0) + (P.use_pos != 0) + (P.use_num != 0);
  P.n_layers = cfg->num_layers;
  if (P.n_feat < 2) return fail(ctx, T2L_EINVAL, "t2l_fine_load_weights: needs at least two of the class/color/position/num features");
  const std::string oe = "object_encoder.";
  int rc;
  auto emb = [&](const std::string& k, const float** dst) -> int {
    auto it = m.find(k);
    if (it == m.end() || it->second->numel % kFD) return fail(ctx, T2L_EINVAL, "t2l_fine_load_weights: missing '" + k + "'");
    return fupload(ctx, W, std::vector<float>(it->second->data, it->second->data + it->second->numel), dst);
  };
  if (P.use_class) {
    if (P.class_embed) { if ((rc = emb(oe + "class_embedding.weight", &P.class_emb))) return rc; }
    else if ((rc = flinear(ctx, W, m, oe + "mlp_pointnet.0.0", oe + "mlp_pointnet.0.1", 256, kFD, &P.pn))) return rc;
  }
  if (P.use_color) {
    if (P.color_embed) { if ((rc = emb(oe + "color_embedding.weight", &P.color_emb))) return rc; }
    else if ((rc = flinear(ctx, W, m, oe + "color_encoder.0.0", oe + "color_encoder.0.1", 3, 64, &P.col1)) ||
             (rc = flinear(ctx, W, m, oe + "color_encoder.1.0", oe + "color_encoder.1.1", 64, kFD, &P.col2))) return rc;
  }
  if (P.use_pos && ((rc = flinear(ctx, W, m, oe + "pos_encoder.0.0", oe + "pos_encoder.0.1", 3, 64, &P.pos1)) ||
                    (rc = flinear(ctx, W, m, oe + "pos_encoder.1.0", oe + "pos_encoder.1.1", 64, kFD, &P.pos2)))) return rc;
  if (P.use_num && ((rc = flinear(ctx, W, m, oe + "num_encoder.0.0", oe + "num_encoder.0.1", 1, 64, &P.num1)) ||
                    (rc = flinear(ctx, W, m, oe + "num_encoder.1.0", oe + "num_encoder.1.1", 64, kFD, &P.num2)))) return rc;
  if ((rc = flinear(ctx, W, m, oe + "mlp_merge.0.0", oe + "mlp_merge.0.1", P.n_feat * kFD, kFD, &P.merge))) return rc;
  // fine_num_decoder_layers == 0 (cross_matcher.py:75-79, 119-120): ONE decoder layer, keys "cross_hints.*" without an index, no
  // cross_objects — the hints attend the raw object descriptors once. Loaded as hint[0]; the kernel branches on n_layers == 0.
  for (int l = 0; l < std::max(1, P.n_layers); ++l)
    for (int which = (P.n_layers == 0 ? 1 : 0); which < 2; ++which) {
      const std::string p = P.n_layers == 0 ? std::string("cross_hints") : std::string(which ? "cross_hints." : "cross_objects.") + std::to_string(l);
      FDecoder& D = which ? P.hint[l] : P.obj[l];
      if ((rc = fpacked(ctx, W, m, p + ".self_attn.in_proj_weight", p + ".self_attn.in_proj_bias", kFD, 3 * kFD, &D.sa_in)) ||
          (rc = fpacked(ctx, W, m, p + ".self_attn.out_proj.weight", p + ".self_attn.out_proj.bias", kFD, kFD, &D.sa_out)) ||
          (rc = fpacked(ctx, W, m, p + ".multihead_attn.in_proj_weight", p + ".multihead_attn.in_proj_bias", kFD, 3 * kFD, &D.ca_in)) ||
          (rc = fpacked(ctx, W, m, p + ".multihead_attn.out_proj.weight", p + ".multihead_attn.out_proj.bias", kFD, kFD, &D.ca_out)) ||
          (rc = fpacked(ctx, W, m, p + ".linear1.weight", p + ".linear1.bias", kFD, 4 * kFD, &D.l1)) ||
          (rc = fpacked(ctx, W, m, p + ".linear2.weight", p + ".linear2.bias", 4 * kFD, kFD, &D.l2)) ||
          (rc = fraw(ctx, W, m, p + ".norm1.weight", kFD, &D.g1)) || (rc = fraw(ctx, W, m, p + ".norm1.bias", kFD, &D.b1)) ||
          (rc = fraw(ctx, W, m, p + ".norm2.weight", kFD, &D.g2)) || (rc = fraw(ctx, W, m, p + ".norm2.bias", kFD, &D.b2)) ||
          (rc = fraw(ctx, W, m, p + ".norm3.weight", kFD, &D.g3)) || (rc = fraw(ctx, W, m, p + ".norm3.bias", kFD, &D.b3)))
        return rc;
    }
  if ((rc = flinear(ctx, W, m, "mlp_offsets.0", "", kFD, kFD / 2, &P.off0)) || (rc = flinear(ctx, W, m, "mlp_offsets.2", "", kFD / 2, 2, &P.off2)))
    return rc;
  {  // split-f16 safety: bound every activation that enters a split product, given input rows of norm <= kFGuardNorm
    const float sq = 11.32f;  // > sqrt(128): |LayerNorm(.)| <= sqrt(127) |gain| + |bias| per element, row norm <= sqrt(128) x that
    float wmax = 0.f, amax = kFGuardNorm, n_obj = kFGuardNorm, n_hint = kFGuardNorm;
    auto dec = [&](const std::string& p, float xn, float mn) -> float {  // returns the norm bound of the rows it leaves in x
      auto W_ = [&](const char* k, int64_t n) { return fget(m, p + k, n); };
      const float *sa = W_(".self_attn.in_proj_weight", 3 * kFD * kFD), *sab = W_(".self_attn.in_proj_bias", 3 * kFD);
      const float *ca = W_(".multihead_attn.in_proj_weight", 3 * kFD * kFD), *cab = W_(".multihead_attn.in_proj_bias", 3 * kFD);
      const float *l1 = W_(".linear1.weight", 4 * kFD * kFD), *l1b = W_(".linear1.bias", 4 * kFD);
      for (const char* k : {".self_attn.in_proj_weight", ".self_attn.out_proj.weight", ".multihead_attn.in_proj_weight",
                            ".multihead_attn.out_proj.weight", ".linear1.weight", ".linear2.weight"}) {
        auto it = m.find(p + k);
        wmax = fmaxf(wmax, h3_max_abs(it->second->data, (size_t)it->second->numel));
      }
      auto ln = [&](const char* g, const char* b) { return sq * h3_max_abs(W_(g, kFD), kFD) + h3_max_abs(W_(b, kFD), kFD); };
      amax = fmaxf(amax, fmaxf(xn, mn));                                                                          // q/k/v inputs
      amax = fmaxf(amax, xn * h3_max_row_norm(sa, 2 * kFD, kFD) + h3_max_abs(sab, 2 * kFD));                     // self q, k (S = K Q^T is a split product)
      amax = fmaxf(amax, xn * h3_max_row_norm(sa + (size_t)2 * kFD * kFD, kFD, kFD) + h3_max_abs(sab + 2 * kFD, kFD));  // self v = P V operand, out_proj input
      const float e1 = ln(".norm1.weight", ".norm1.bias");
      amax = fmaxf(amax, e1);                                                                                     // cross-attention q input
      amax = fmaxf(amax, sq * e1 * h3_max_row_norm(ca, kFD, kFD) + h3_max_abs(cab, kFD));                        // cross q
      amax = fmaxf(amax, mn * h3_max_row_norm(ca + (size_t)kFD * kFD, kFD, kFD) + h3_max_abs(cab + kFD, kFD));   // cross k
      amax = fmaxf(amax, mn * h3_max_row_norm(ca + (size_t)2 * kFD * kFD, kFD, kFD) + h3_max_abs(cab + 2 * kFD, kFD));  // cross v, out_proj input
      const float e2 = ln(".norm2.weight", ".norm2.bias");
      amax = fmaxf(amax, e2);                                                                                     // linear1 input
      amax = fmaxf(amax, sq * e2 * h3_max_row_norm(l1, 4 * kFD, kFD) + h3_max_abs(l1b, 4 * kFD));                // linear2 input
      return sq * ln(".norm3.weight", ".norm3.bias");
    };
    for (int l = 0; l < P.n_layers; ++l) {
      n_obj = dec("cross_objects." + std::to_string(l), n_obj, n_hint);
      n_hint = dec("cross_hints." + std::to_string(l), n_hint, n_obj);
    }
    if (P.n_layers == 0) n_hint = dec("cross_hints", n_hint, n_obj);
    P.split_ok = (wmax < kSplitF16Safe && amax < kSplitF16Safe) ? 1 : 0;
  }
  return T2L_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// device building blocks (one workgroup of 256 threads; token buffers in LDS)
// ---------------------------------------------------------------------------------------------------------------
// out[t][n] = act(b[n] + sum_k x[t][k] * Wt[k][n]) for t < T8*8 rows (T8 groups of 8 rows), n < N. K % 4 == 0 unless K < 4.
__device__ void f_linear(const float* __restrict__ x, int ldx, int T8, const FLinear L, int K, int N, float* __restrict__ out, int ldo,
                         int ocol, bool relu) {
  for (int item = threadIdx.x; item < N * T8; item += 256) {
    const int n = item % N, tg = item / N;
    float acc[8];
    const float bv = L.b[n];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bv;
    const float* xr = x + (size_t)tg * 8 * ldx;
    if ((K & 3) == 0) {
      for (int k = 0; k < K; k += 4) {
        const float w0 = L.wt[(size_t)(k + 0) * N + n], w1 = L.wt[(size_t)(k + 1) * N + n], w2 = L.wt[(size_t)(k + 2) * N + n],
                    w3 = L.wt[(size_t)(k + 3) * N + n];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float4 v = *reinterpret_cast<const float4*>(xr + j * ldx + k);
          acc[j] += v.x * w0 + v.y * w1 + v.z * w2 + v.w * w3;
        }
      }
    } else {
      for (int k = 0; k < K; ++k) {
        const float w0 = L.wt[(size_t)k * N + n];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += xr[j * ldx + k] * w0;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) out[(size_t)(tg * 8 + j) * ldo + ocol + n] = relu ? fmaxf(acc[j], 0.f) : acc[j];
  }
}

// all-reduce sum over the 64 lanes on the VALU (DPP + v_permlane swaps): __shfl_xor lowers to ds_bpermute_b32, six dependent
// LDS round trips per sum, and LayerNorm needs two sums per token row
template <int CTRL>
__device__ __forceinline__ float f_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float f_wsum(float v) {
  v += f_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
  v += f_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
  v += f_dpp<0x141>(v);  // row_half_mirror
  v += f_dpp<0x140>(v);  // row_mirror
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
// rows t < T of x (128 wide): x = normalize(x) (F.normalize, eps 1e-12); one wave per row
__device__ void f_normalize_rows(float* x, int ld, int T, int width = kFD) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int t = w; t < T; t += 4) {
    float s = 0.f;
    for (int c = lane; c < width; c += 64) s += x[t * ld + c] * x[t * ld + c];
    const float n = fmaxf(sqrtf(f_wsum(s)), 1e-12f);
    for (int c = lane; c < width; c += 64) x[t * ld + c] /= n;
  }
}
constexpr int kPairs = 4;  // pairs per workgroup: their 4 x 16 object tokens fill TWO 32-row MFMA tiles, their 4 x 8 hint slots ONE

// x[t] = LayerNorm(x[t]) * g + b for t < T (in place; one wave per row)
__device__ void f_ln_rows(float* x, int ld, int T, const float* __restrict__ g, const float* __restrict__ b) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int t = w; t < T; t += 4) {
    const float v0 = x[t * ld + lane], v1 = x[t * ld + lane + 64];
    const float mu = f_wsum(v0 + v1) * (1.f / kFD);
    const float d0 = v0 - mu, d1 = v1 - mu;
    const float rstd = 1.0f / sqrtf(f_wsum(d0 * d0 + d1 * d1) * (1.f / kFD) + 1e-5f);
    x[t * ld + lane] = d0 * rstd * g[lane] + b[lane];
    x[t * ld + lane + 64] = d1 * rstd * g[lane + 64] + b[lane + 64];
  }
}

// Which rows of a 32-row token tile belong together: the tile's g-th group (rows (g << shift) .. + count - 1 of the first `rows`
// rows) belongs to pair base + g of the workgroup.
struct TileGroups {
  int shift, count, rows, base;
};

// One attention block of nn.TransformerDecoderLayer for the kPairs pairs of the tile, head h = wave, REGISTERS ONLY:
// q_h^T (from the x tile) and k_h^T (from the mem tile) are computed transposed (A = packed in_proj rows, B = token rows),
// v_h straight (A = mem token rows, B = packed rows); in those MFMA output layouts k_h^T / q_h^T are the A / B operands of
// S^T = K Q^T and v_h is the B operand of P V (the trick of encode.hip). Keys of another pair (or padding rows) are masked.
// Writes o_h (32 rows x 32 columns) into obuf[:, 32 h ..]. x == mem for self-attention.
// ACC: the output is ADDED to obuf (a query tile whose groups find their keys in different mem tiles attends each mem tile in turn;
// a query row sees keys in exactly one of them and contributes zeros to the other pass).
template <int H, bool ACC = false>
__device__ __forceinline__ void f_attention_regs(const float* __restrict__ x, TileGroups gx, const float* __restrict__ mem, TileGroups gm,
                                                 const FPacked in_proj, float* __restrict__ obuf) {
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6, col = lane & 31, half = lane >> 5;
  constexpr int QN = kFD / 8;  // 16 packed k-steps
  const float* xr = x + col * kFS + half * (kFD / 2);
  const float* mr = mem + col * kFS + half * (kFD / 2);
  const float4* wq = in_proj.w + (size_t)h * QN * 64 + lane;
  const float4* wk = in_proj.w + (size_t)(4 + h) * QN * 64 + lane;
  const float4* wv = in_proj.w + (size_t)(8 + h) * QN * 64 + lane;
  f32x16 qT, kT, v;
#pragma unroll
  for (int r = 0; r < 16; ++r) qT[r] = kT[r] = v[r] = 0.f;
  if constexpr (H != 0) {
    constexpr int HS = kFD / 16;  // 8 steps of 16
    const uint4* hq = in_proj.h + ((size_t)h * HS * 64 + lane) * 2;
    const uint4* hk = in_proj.h + ((size_t)(4 + h) * HS * 64 + lane) * 2;
    const uint4* hv = in_proj.h + ((size_t)(8 + h) * HS * 64 + lane) * 2;
    // (a pinned ring for these three fragments — as in mm32_dot_h — was measured SLOWER here: 5.98 ms against 5.66 ms, the kernel
    // runs three workgroups per CU at <= 168 VGPRs)
#pragma unroll 4
    for (int st = 0; st < HS; ++st) {
      const HFrag xf = split_h<H == 2>(xr + 8 * st), mf = split_h<H == 2>(mr + 8 * st);
      mfma_h3<H == 2>(qT, load_h1<H == 2>(hq + T2L_HOT(st) * 128), xf);
      mfma_h3<H == 2>(kT, load_h1<H == 2>(hk + T2L_HOT(st) * 128), mf);
      mfma_h3<H == 2>(v, mf, load_h1<H == 2>(hv + T2L_HOT(st) * 128));
    }
  } else {
#pragma unroll 4
  for (int q = 0; q < QN; ++q) {
    const float4 xv = *reinterpret_cast<const float4*>(xr + 4 * q);
    const float4 mv = *reinterpret_cast<const float4*>(mr + 4 * q);
    const float4 a = wq[q * 64], c = wk[q * 64], e = wv[q * 64];
#define T2L_F_QKV(C)                                                        \
  qT = __builtin_amdgcn_mfma_f32_32x32x2f32(a.C, xv.C, qT, 0, 0, 0);        \
  kT = __builtin_amdgcn_mfma_f32_32x32x2f32(c.C, mv.C, kT, 0, 0, 0);        \
  v = __builtin_amdgcn_mfma_f32_32x32x2f32(mv.C, e.C, v, 0, 0, 0);
    T2L_F_QKV(x) T2L_F_QKV(y) T2L_F_QKV(z) T2L_F_QKV(w)
#undef T2L_F_QKV
  }
  }
  {
    const float* ib = in_proj.b;
    const float bv = ib[2 * kFD + h * kFHd + col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = (r & 3) + 8 * (r >> 2) + 4 * half;
      qT[r] += ib[h * kFHd + f];
      kT[r] += ib[kFD + h * kFHd + f];
      v[r] += bv;
    }
  }
  f32x16 st;
#pragma unroll
  for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kT[r], qT[r], st, 0, 0, 0);
  // lane: query i = col, keys j = (r&3) + 8*(r>>2) + 4*half
  const int gi = gx.base + (col >> gx.shift) - gm.base;  // the mem tile's group that holds this query's pair (may be out of the tile)
  float m = -__builtin_inff();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = (r & 3) + 8 * (r >> 2) + 4 * half;
    const bool ok = (j >> gm.shift) == gi && (j & ((1 << gm.shift) - 1)) < gm.count && j < gm.rows;
    st[r] = ok ? st[r] * 0.17677669529663687f : -__builtin_inff();  // 1/sqrt(32)
    m = fmaxf(m, st[r]);
  }
  m = fmaxf(m, __shfl_xor(m, 32));
  const bool any = m > -__builtin_inff();  // rows that are nobody's query (tile padding) see no key at all
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    st[r] = any ? __expf(st[r] - m) : 0.f;
    sum += st[r];
  }
  sum += __shfl_xor(sum, 32);
  const float inv = any ? 1.f / sum : 0.f;
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) o = __builtin_amdgcn_mfma_f32_32x32x2f32(st[r] * inv, v[r], o, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float* dst = obuf + ((r & 3) + 8 * (r >> 2) + 4 * half) * kFS + h * kFHd + col;
    if constexpr (ACC) *dst += o[r]; else *dst = o[r];
  }
}

// x += A @ W^T + b for a 128 -> 128 Linear (out_proj): one 32-column tile per wave
template <int H>
__device__ __forceinline__ void f_proj_add(const float* __restrict__ A, const FPacked L, float* __restrict__ x) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = lane & 31, half = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if constexpr (H != 0) mm32_dot_h<kFD / 16, H == 2>(A + col * kFS + half * (kFD / 2), L.h + ((size_t)w * (kFD / 16) * 64 + lane) * 2, acc);
  else mm32_dot<kFD / 8>(A + col * kFS + half * (kFD / 2), L.w + (size_t)w * (kFD / 8) * 64 + lane, acc);
  const float bv = L.b[w * 32 + col];
#pragma unroll
  for (int r = 0; r < 16; ++r) x[((r & 3) + 8 * (r >> 2) + 4 * half) * kFS + w * 32 + col] += acc[r] + bv;
}

// nn.TransformerDecoderLayer (post-norm, ReLU, eval, no masks) for the kPairs pairs of a workgroup. x / mem: 32-row token
// tiles (LDS, stride kFS); buf: one more 32 x 128 tile (attention output, then the feed-forward hidden in four quarters).
// mem2 != nullptr: the memory is TWO tiles (the hint tile of four pairs attends the pairs' two object tiles).
template <int H>
__device__ void f_decoder(float* x, TileGroups gx, const float* mem, TileGroups gm, const FDecoder D, float* buf,
                          const float* mem2 = nullptr, TileGroups gm2 = TileGroups{0, 0, 0, 0}) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = lane & 31, half = lane >> 5;
  f_attention_regs<H>(x, gx, x, gx, D.sa_in, buf);
  __syncthreads();
  f_proj_add<H>(buf, D.sa_out, x);
  __syncthreads();
  f_ln_rows(x, kFS, gx.rows, D.g1, D.b1);
  __syncthreads();
  f_attention_regs<H>(x, gx, mem, gm, D.ca_in, buf);
  if (mem2) f_attention_regs<H, true>(x, gx, mem2, gm2, D.ca_in, buf);  // (same wave, same obuf columns: ordered without a barrier)
  __syncthreads();
  f_proj_add<H>(buf, D.ca_out, x);
  __syncthreads();
  f_ln_rows(x, kFS, gx.rows, D.g2, D.b2);
  __syncthreads();
  {  // feed-forward 128 -> 512 -> 128: the hidden layer passes through buf in four quarters of 128 units — the units that
     // k-steps [16 q, 16 q + 16) of the half-split packing of W2 (K = 512) cover — W2's product accumulates in registers
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int qtr = 0; qtr < 4; ++qtr) {
      // quarter = hidden tiles {2 qtr, 2 qtr + 1} (buf columns 0..63) and {8 + 2 qtr, 9 + 2 qtr} (columns 64..127)
      const int tile = (w < 2 ? 2 * qtr + w : 8 + 2 * qtr + (w - 2));
      f32x16 hh;
#pragma unroll
      for (int r = 0; r < 16; ++r) hh[r] = 0.f;
      if constexpr (H != 0) mm32_dot_h<kFD / 16, H == 2>(x + col * kFS + half * (kFD / 2), D.l1.h + ((size_t)tile * (kFD / 16) * 64 + lane) * 2, hh);
      else mm32_dot<kFD / 8>(x + col * kFS + half * (kFD / 2), D.l1.w + (size_t)tile * (kFD / 8) * 64 + lane, hh);
      if (qtr) __syncthreads();  // every wave has consumed the previous quarter
      const float bv = D.l1.b[tile * 32 + col];
#pragma unroll
      for (int r = 0; r < 16; ++r) buf[((r & 3) + 8 * (r >> 2) + 4 * half) * kFS + w * 32 + col] = fmaxf(hh[r] + bv, 0.f);
      __syncthreads();
      if constexpr (H != 0)  // K = 512: 32 steps per tile, quarter qtr = steps [8 qtr, 8 qtr + 8)
        mm32_dot_h<kFD / 16, H == 2>(buf + col * kFS + half * (kFD / 2), D.l2.h + (((size_t)w * (4 * kFD / 16) + 8 * qtr) * 64 + lane) * 2, acc);
      else
        mm32_dot<kFD / 8>(buf + col * kFS + half * (kFD / 2), D.l2.w + ((size_t)w * (4 * kFD / 8) + 16 * qtr) * 64 + lane, acc);
    }
    const float bv = D.l2.b[w * 32 + col];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[((r & 3) + 8 * (r >> 2) + 4 * half) * kFS + w * 32 + col] += acc[r] + bv;
  }
  __syncthreads();
  f_ln_rows(x, kFS, gx.rows, D.g3, D.b3);
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// The same decoder layer on split-f16 PLANES (round 5; the recipe of encode.hip's second form). In the f32-tile form above every
// product splits its activation operand into hi + lo f16 on the fly — 20 VALU per 3 MFMAs, repeated by each of the four waves
// (they multiply the same token rows into different output columns) and again for every use of the same rows (q, k, v, four
// feed-forward quarters): ~2,200 of the ~2,800 VALU instructions a wave issued per tile pass were operand conversions (PMC, round 4:
// 11.4 VALU per MFMA, MFMA pipe busy 0.19 by instruction count). Here every token tile lives in LDS as two f16 planes (hi | lo,
// rows of 136 halves = 272 B: the 32 lanes of a fragment read are conflict-free 16-byte reads) and
//  * a product's activation operand is two ds_read_b128, no conversion;
//  * every product is computed TRANSPOSED (A = weight fragment, B = token fragment: D[feature][token]), so a lane ends up with 16
//    features of ONE token, 4 consecutive ones per register quad: the epilogue splits each produced value ONCE and stores 8-byte
//    plane pieces; the attention's P V product likewise as O^T = V^T P^T (the same registers with the operands swapped);
//  * the residual stream is rebuilt from hi + lo (22 significand bits; the offsets stay within 5e-5 of the reference's).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kLdF = kFD + 8;                    // halves per plane row
constexpr int kPlaneHalves = 32 * kLdF;          // one plane of a 32-row tile
constexpr int kTileBytes = 2 * kPlaneHalves * 2; // hi + lo: 17,408 B (the f32 tile of the other form: 16,896 B)
typedef _Float16 ff_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 ff_f16x2 __attribute__((ext_vector_type(2)));
typedef float ff_f32x4 __attribute__((ext_vector_type(4)));
typedef float ff_f32x2 __attribute__((ext_vector_type(2)));
struct FP {
  _Float16* hi;
  _Float16* lo;
};
__device__ __forceinline__ FP fp_at(void* base) {
  FP t;
  t.hi = reinterpret_cast<_Float16*>(base);
  t.lo = t.hi + kPlaneHalves;
  return t;
}
template <bool SG>
__device__ __forceinline__ HFrag fp_frag(const FP t, int off) {
  HFrag f;
  f.hi = *reinterpret_cast<const h3_f16x8*>(t.hi + off);
  if constexpr (SG) f.lo = f.hi;
  else f.lo = *reinterpret_cast<const h3_f16x8*>(t.lo + off);
  return f;
}
__device__ __forceinline__ void fp_put4(const FP t, int off, ff_f32x4 v) {
  h3_f16x4 h, l;
  h3_split4(h3_f32x4{v[0], v[1], v[2], v[3]}, h, l);
  *reinterpret_cast<h3_f16x4*>(t.hi + off) = h;
  *reinterpret_cast<h3_f16x4*>(t.lo + off) = l;
}
__device__ __forceinline__ ff_f32x4 fp_get4(const FP t, int off) {
  const h3_f32x4 v = h3_join4(*reinterpret_cast<const h3_f16x4*>(t.hi + off), *reinterpret_cast<const h3_f16x4*>(t.lo + off));
  return ff_f32x4{v[0], v[1], v[2], v[3]};
}
// acc (D[feature][token]) += W tile (STEPS k-steps of packed fragments at wp) x tokens (plane rows, this lane's half row at aoff)
template <int STEPS, bool SG>
__device__ __forceinline__ void fp_dot(const FP a, int aoff, const uint4* __restrict__ wp, f32x16& acc) {
  // ring depth 3 (same-box A/B against 2 and 4, round 5: 4.35 / 4.28 / 4.31 ms, plain f16 3.44 / 3.31 / 3.36): the fragments' L2
  // latency is this kernel's largest single cost now
  constexpr int kRing = 3, D = STEPS < kRing ? STEPS : kRing;
  HFrag ring[D];
#pragma unroll
  for (int i = 0; i < D; ++i) ring[i] = load_h1<SG>(wp + T2L_HOT(i) * 128);
#pragma unroll
  for (int st = 0; st < STEPS; ++st) {
    const HFrag wf = ring[st % D];
    if (st + D < STEPS) ring[st % D] = load_h1<SG>(wp + T2L_HOT(st + D) * 128);
    __builtin_amdgcn_sched_barrier(0);
    mfma_h3<SG>(acc, wf, fp_frag<SG>(a, aoff + 8 * st));
    __builtin_amdgcn_sched_barrier(0);
  }
}
// x[t][32 w + f] (+)= acc[f][t] + bias[32 w + f] for the wave's feature tile; RELU: buf = relu(acc + bias) (no accumulation)
template <bool ADD, bool RELU>
__device__ __forceinline__ void fp_epilogue(const FP dst, int dcol0, const f32x16& acc, const float* __restrict__ bias) {
  const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int f = 8 * g + 4 * half, off = col * kLdF + dcol0 + f;
    const float4 b = *reinterpret_cast<const float4*>(bias + f);
    ff_f32x4 v = {acc[4 * g] + b.x, acc[4 * g + 1] + b.y, acc[4 * g + 2] + b.z, acc[4 * g + 3] + b.w};
    if constexpr (RELU) v = ff_f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
    if constexpr (ADD) v += fp_get4(dst, off);
    fp_put4(dst, off, v);
  }
}
// x[t][32 w + f] = LayerNorm_t(x[t] + acc[.][t] + bias) * g + b — the residual epilogue and the LayerNorm behind it in one go. A lane
// holds 16 of its token's 128 features (its partner lane ^ 32 another 16, the other three waves 32 each): the row sums meet in
// 1 KB of LDS (two light barriers: mean, then the centred squares — the two-pass form, as nn.LayerNorm), the values stay in
// registers in between, and the normalised row is written ONCE. (As a separate pass — one wave per row, two full-wave reductions per
// row — the three LayerNorms were ~960 of a wave's VALU instructions per tile pass and a plane round trip each.)
__device__ __forceinline__ void fp_epilogue_ln(const FP x, const f32x16& acc, const float* __restrict__ bias, const float* __restrict__ g,
                                               const float* __restrict__ b, float* __restrict__ red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = lane & 31, half = lane >> 5;
  ff_f32x4 v[4];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int f = 32 * w + 8 * q + 4 * half;
    const float4 bb = *reinterpret_cast<const float4*>(bias + f);
    v[q] = fp_get4(x, col * kLdF + f) + ff_f32x4{acc[4 * q] + bb.x, acc[4 * q + 1] + bb.y, acc[4 * q + 2] + bb.z, acc[4 * q + 3] + bb.w};
    s += (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
  }
  s += __shfl_xor(s, 32);
  if (half == 0) red[w * 32 + col] = s;
  __syncthreads();
  const float mu = ((red[col] + red[32 + col]) + (red[64 + col] + red[96 + col])) * (1.f / kFD);
  float qs = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[q] -= ff_f32x4{mu, mu, mu, mu};
    qs += (v[q][0] * v[q][0] + v[q][1] * v[q][1]) + (v[q][2] * v[q][2] + v[q][3] * v[q][3]);
  }
  qs += __shfl_xor(qs, 32);
  if (half == 0) red[128 + w * 32 + col] = qs;
  __syncthreads();
  const float rstd = 1.0f / sqrtf(((red[128 + col] + red[160 + col]) + (red[192 + col] + red[224 + col])) * (1.f / kFD) + 1e-5f);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int f = 32 * w + 8 * q + 4 * half;
    const float4 gg = *reinterpret_cast<const float4*>(g + f), bb = *reinterpret_cast<const float4*>(b + f);
    fp_put4(x, col * kLdF + f, ff_f32x4{v[q][0] * rstd * gg.x + bb.x, v[q][1] * rstd * gg.y + bb.y, v[q][2] * rstd * gg.z + bb.z,
                                         v[q][3] * rstd * gg.w + bb.w});
  }
}
// x[t] = LayerNorm(x[t]) * g + b for t < T (in place; one wave per row, two consecutive features per lane)
__device__ void fp_ln_rows(const FP x, int T, const float* __restrict__ g, const float* __restrict__ b) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float2 gv = *reinterpret_cast<const float2*>(g + 2 * lane), bv = *reinterpret_cast<const float2*>(b + 2 * lane);
  for (int t = w; t < T; t += 4) {
    const int off = t * kLdF + 2 * lane;
    const h3_f32x2 v = h3_join2(*reinterpret_cast<const h3_f16x2*>(x.hi + off), *reinterpret_cast<const h3_f16x2*>(x.lo + off));
    const float mu = f_wsum(v[0] + v[1]) * (1.f / kFD);
    const float d0 = v[0] - mu, d1 = v[1] - mu;
    const float rstd = 1.0f / sqrtf(f_wsum(d0 * d0 + d1 * d1) * (1.f / kFD) + 1e-5f);
    const ff_f32x2 o = {d0 * rstd * gv.x + bv.x, d1 * rstd * gv.y + bv.y};
    h3_f16x2 h, l;
    h3_split2(h3_f32x2{o[0], o[1]}, h, l);
    *reinterpret_cast<h3_f16x2*>(x.hi + off) = h;
    *reinterpret_cast<h3_f16x2*>(x.lo + off) = l;
  }
}
// One attention block on planes, head h = wave (f_attention_regs with plane operands and the output transposed). SHARED: the query tile's
// groups find their keys in different mem tiles and attend each in turn — a query row sees keys in exactly ONE of the passes, and only
// that pass writes its output row. (A first version added the passes up: the row's other pass read hi + lo back, added zero and split
// the sum again — the same value, but where |lo| had been rounded up to exactly half an ulp of hi the re-split picks the other
// neighbour as hi; the products drop lo * W_lo, so the offsets then depended, at 2e-7, on which tile of the workgroup a pair sat in —
// and a sharded run_fine, which deals the pairs differently, was no longer bit-identical to the single-process run.)
template <int H, bool SHARED>
__device__ __forceinline__ void fp_attention(const FP x, TileGroups gx, const FP mem, TileGroups gm, bool self, const FPacked in_proj, const FP obuf) {
  constexpr bool SG = H == 2;
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6, col = lane & 31, half = lane >> 5;
  constexpr int HS = kFD / 16;  // 8 steps of 16
  const uint4* hq = in_proj.h + ((size_t)h * HS * 64 + lane) * 2;
  const uint4* hk = in_proj.h + ((size_t)(4 + h) * HS * 64 + lane) * 2;
  const uint4* hv = in_proj.h + ((size_t)(8 + h) * HS * 64 + lane) * 2;
  const int aoff = col * kLdF + half * (kFD / 2);
  f32x16 qT, kT, v;
#pragma unroll
  for (int r = 0; r < 16; ++r) qT[r] = kT[r] = v[r] = 0.f;
  {  // the three weight tiles of a k-step through a register ring, requested D steps ahead and pinned there (as fp_dot). The pointers
     // RUN through opaque increments: with `base + constant` addressing the compiler materialises and hoists one 64-bit address per
     // load (24 of them: 120+ spilled registers); as a partly unrolled loop without a ring (rounds 4-5) every body of four steps
     // started with the L2 round trip of its twelve loads exposed. Same-box A/B: no ring 4.28 ms, D = 1 / 2 / 3: 3.75 / 3.83 / 4.04
     // (251 registers and no spill at D = 1; 20 / 44 spilled registers at 2 / 3); plain f16 3.45 -> 2.94.
#ifndef T2L_ATT_RING
#define T2L_ATT_RING 1
#endif
    constexpr int D = T2L_ATT_RING;
    const uint4 *pq = hq, *pk = hk, *pv = hv;
    asm volatile("" : "+v"(pq), "+v"(pk), "+v"(pv));
    HFrag ring[D][3];
    auto load3 = [&](HFrag (&f)[3]) {
      f[0] = load_h1<SG>(pq);
      f[1] = load_h1<SG>(pk);
      f[2] = load_h1<SG>(pv);
#ifndef T2L_EXP_HOTW
      pq += 128; pk += 128; pv += 128;
#endif
      asm volatile("" : "+v"(pq), "+v"(pk), "+v"(pv));
    };
#pragma unroll
    for (int i = 0; i < D; ++i) load3(ring[i]);
#pragma unroll
    for (int st = 0; st < HS; ++st) {
      HFrag f[3];
#pragma unroll
      for (int e = 0; e < 3; ++e) f[e] = ring[st % D][e];
      if (st + D < HS) load3(ring[st % D]);
      __builtin_amdgcn_sched_barrier(0);
      const HFrag xf = fp_frag<SG>(x, aoff + 8 * st);
      const HFrag mf = self ? xf : fp_frag<SG>(mem, aoff + 8 * st);
      mfma_h3<SG>(qT, f[0], xf);
      mfma_h3<SG>(kT, f[1], mf);
      mfma_h3<SG>(v, mf, f[2]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  {
    const float* ib = in_proj.b;
    const float bv = ib[2 * kFD + h * kFHd + col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = (r & 3) + 8 * (r >> 2) + 4 * half;
      qT[r] += ib[h * kFHd + f];
      kT[r] += ib[kFD + h * kFHd + f];
      v[r] += bv;
    }
  }
  // S^T = K Q^T and (below) O^T = V^T P^T as split-f16 products too: kT / qT (and v / P) sit in the same accumulator layout, so
  // registers 0..7 and 8..15 of the two are matching k-halves of A and B — 6 MFMAs of 32 cycles per product instead of the 16
  // f32 MFMAs of 64 (the attention core was 2,048 of a wave's ~4,350 matrix-pipe cycles per attention block). Always the
  // three-product form, also under option encoder_f16: the logits carry the softmax. fine_load_weights bounds q, k, v.
  f32x16 st;
#pragma unroll
  for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
  for (int m2 = 0; m2 < 2; ++m2) mfma_h3<false>(st, split_acc8<false>(kT, m2), split_acc8<false>(qT, m2));
  // lane: query i = col, keys j = (r&3) + 8*(r>>2) + 4*half
  const int gi = gx.base + (col >> gx.shift) - gm.base;
  float m = -__builtin_inff();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = (r & 3) + 8 * (r >> 2) + 4 * half;
    const bool ok = (j >> gm.shift) == gi && (j & ((1 << gm.shift) - 1)) < gm.count && j < gm.rows;
    st[r] = ok ? st[r] * 0.17677669529663687f : -__builtin_inff();  // 1/sqrt(32)
    m = fmaxf(m, st[r]);
  }
  m = fmaxf(m, __shfl_xor(m, 32));
  const bool any = m > -__builtin_inff();
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    st[r] = any ? __expf(st[r] - m) : 0.f;
    sum += st[r];
  }
  sum += __shfl_xor(sum, 32);
  const float inv = any ? 1.f / sum : 0.f;
  // O^T = V^T P^T: A[feature c][key] = v (as it sits: lane (c, half) holds V[key(r, half)][c]), B[key][query] = P (lane (query, half))
  f32x16 oT;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    oT[r] = 0.f;
    st[r] *= inv;
  }
#pragma unroll
  for (int m2 = 0; m2 < 2; ++m2) mfma_h3<false>(oT, split_acc8<false>(v, m2), split_acc8<false>(st, m2));
#pragma unroll
  for (int g = 0; g < 4; ++g) {  // lane (query = col, half): features 8 g + 4 half + 0..3 of head h
    const int off = col * kLdF + h * kFHd + 8 * g + 4 * half;
    const ff_f32x4 o4 = {oT[4 * g], oT[4 * g + 1], oT[4 * g + 2], oT[4 * g + 3]};
    if (!SHARED || any) fp_put4(obuf, off, o4);
  }
}
template <int H>
__device__ void fp_decoder(const FP x, TileGroups gx, const FP mem, TileGroups gm, const FDecoder D, const FP buf, float* red,
                           const FP* mem2 = nullptr, TileGroups gm2 = TileGroups{0, 0, 0, 0}) {
  constexpr bool SG = H == 2;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = lane & 31, half = lane >> 5;
  const int aoff = col * kLdF + half * (kFD / 2);
  auto proj_add_ln = [&](const FPacked& L, const float* g, const float* b) {  // x = LN(x + buf @ W^T + bias): the wave's 32 features
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    fp_dot<kFD / 16, SG>(buf, aoff, L.h + ((size_t)w * (kFD / 16) * 64 + lane) * 2, acc);
    fp_epilogue_ln(x, acc, L.b, g, b, red);  // (nobody reads x here: the attention that did is behind a barrier)
  };
  fp_attention<H, false>(x, gx, x, gx, true, D.sa_in, buf);
  __syncthreads();
  proj_add_ln(D.sa_out, D.g1, D.b1);
  __syncthreads();
  if (!mem2) {
    fp_attention<H, false>(x, gx, mem, gm, false, D.ca_in, buf);
  } else {  // (every query row has its keys in exactly one of the two tiles)
    fp_attention<H, true>(x, gx, mem, gm, false, D.ca_in, buf);
    fp_attention<H, true>(x, gx, *mem2, gm2, false, D.ca_in, buf);
  }
  __syncthreads();
  proj_add_ln(D.ca_out, D.g2, D.b2);
  __syncthreads();
  {  // feed-forward 128 -> 512 -> 128, the hidden layer through buf in four quarters (see f_decoder)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int qtr = 0; qtr < 4; ++qtr) {
      const int tile = (w < 2 ? 2 * qtr + w : 8 + 2 * qtr + (w - 2));
      f32x16 hh;
#pragma unroll
      for (int r = 0; r < 16; ++r) hh[r] = 0.f;
      fp_dot<kFD / 16, SG>(x, aoff, D.l1.h + ((size_t)tile * (kFD / 16) * 64 + lane) * 2, hh);
      if (qtr) __syncthreads();  // every wave has consumed the previous quarter
      fp_epilogue<false, true>(buf, w * 32, hh, D.l1.b + tile * 32);
      __syncthreads();
      fp_dot<kFD / 16, SG>(buf, aoff, D.l2.h + (((size_t)w * (4 * kFD / 16) + 8 * qtr) * 64 + lane) * 2, acc);
    }
    fp_epilogue_ln(x, acc, D.l2.b, D.g3, D.b3, red);  // (the last read of x, linear1's, is behind the quarter's barriers)
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------
// One workgroup per padded cell (16 objects): ObjectEncoder.forward at D=128 (object_encoder.py:66-153) + F.normalize.
__global__ __launch_bounds__(256) void fine_objects_kernel(FineParams P, t2l_packed_cells in, float* __restrict__ out) {
  __shared__ float xin[kFObj * 260];    // branch inputs (pn_feat 256 wide at most)
  __shared__ float h64[kFObj * 68];
  __shared__ float cat[kFObj * 516];
  __shared__ float res[kFObj * kFS];
  const int cell = blockIdx.x, tid = threadIdx.x, o0 = cell * kFObj;
  int slot = 0;
  auto small_branch = [&](const float* src, int K, bool standardize, FLinear l1, FLinear l2, int sl) {
    for (int i = tid; i < kFObj * 4; i += 256) {
      const int t = i >> 2, k = i & 3;
      float v = 0.f;
      if (k < K) {
        v = src[(size_t)(o0 + t) * K + k];
        if (standardize) v = (v - 1826.6844940968194f) / 2516.8905096993817f;
      }
      xin[t * 260 + k] = v;
    }
    __syncthreads();
    f_linear(xin, 260, 2, l1, K, 64, h64, 68, 0, true);
    __syncthreads();
    f_linear(h64, 68, 2, l2, 64, kFD, cat, 516, sl * kFD, true);
    __syncthreads();
  };
  if (P.use_class) {
    if (P.class_embed) {
      for (int i = tid; i < kFObj * kFD; i += 256) cat[(i / kFD) * 516 + slot * kFD + (i % kFD)] = P.class_emb[(size_t)in.class_idx[o0 + i / kFD] * kFD + (i % kFD)];
    } else {
      for (int i = tid; i < kFObj * 256; i += 256) xin[(i >> 8) * 260 + (i & 255)] = in.pn_feat[(size_t)(o0 + (i >> 8)) * 256 + (i & 255)];
      __syncthreads();
      f_linear(xin, 260, 2, P.pn, 256, kFD, cat, 516, slot * kFD, true);
    }
    __syncthreads();
    ++slot;
  }
  if (P.use_color) {
    if (P.color_embed) {
      for (int i = tid; i < kFObj * kFD; i += 256) cat[(i / kFD) * 516 + slot * kFD + (i % kFD)] = P.color_emb[(size_t)in.color_idx[o0 + i / kFD] * kFD + (i % kFD)];
      __syncthreads();
    } else {
      small_branch(in.rgb, 3, false, P.col1, P.col2, slot);
    }
    ++slot;
  }
  if (P.use_pos) small_branch(in.center, 3, false, P.pos1, P.pos2, slot++);
  if (P.use_num) small_branch(in.n_pts, 1, true, P.num1, P.num2, slot++);
  // F.normalize each feature slot (object_encoder.py:110-145), merge, F.normalize (cross_matcher.py:104)
  for (int sl = 0; sl < P.n_feat; ++sl) f_normalize_rows(cat + sl * kFD, 516, kFObj);
  __syncthreads();
  f_linear(cat, 516, 2, P.merge, P.n_feat * kFD, kFD, res, kFS, 0, true);
  __syncthreads();
  f_normalize_rows(res, kFS, kFObj);
  __syncthreads();
  for (int i = tid; i < kFObj * kFD; i += 256) out[(size_t)cell * kFObj * kFD + i] = res[(i / kFD) * kFS + (i % kFD)];
}

// One workgroup per kPairs = 4 (query, cell) pairs: the 4 x 16 object tokens fill two 32-row MFMA tiles (pairs 0-1, pairs 2-3), the
// 4 x <= 8 hint tokens ONE (pair p at rows 8p .. 8p + n_hints - 1, other rows zero / ignored). Until round 5 a workgroup held two
// pairs: one object tile and a hint tile with only 16 of its 32 rows in use — every decoder layer over the hints streamed its
// megabyte of weights and ran its ~450 MFMAs per wave for a tile that was 5/8 empty. With four pairs the hint layers are paid
// once per four pairs instead of once per two: 3 tile passes per 4 pairs instead of 4 (+ the hint tile's second cross-attention
// pass, its pairs' objects sitting in two tiles). Five 32 x 128 LDS tiles' worth (69 KB): two workgroups per CU.
// H = split-f16 MFMAs: a workgroup whose raw descriptor rows exceed kFGuardNorm writes wg_flags[block] = 1 and leaves; the
// !H launch that follows (all-f32 MFMA) serves exactly those workgroups (wg_flags == nullptr: every workgroup).
template <int H>
__global__ __launch_bounds__(256, 2) void fine_match_kernel(FineParams P, const float* __restrict__ cell_desc, const int32_t* __restrict__ cell_index,
                                                            const float* __restrict__ hint_desc, const int32_t* __restrict__ hint_index,
                                                            int n_pairs, int n_hints, float* __restrict__ out, int32_t* __restrict__ wg_flags) {
  if constexpr (!H) {
    if (wg_flags && !wg_flags[blockIdx.x]) return;
  }
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* pooled = sm + 4 * kTileBytes / 4;  // [kPairs][128] (behind the four tiles in either form)
  float* h64 = pooled + kPairs * kFD;       // [kPairs][64]
  float* red = h64 + kPairs * 64;           // [2][4][32]: row sums of the fused residual + LayerNorm epilogues
  const int tid = threadIdx.x, pair0 = blockIdx.x * kPairs;
  if constexpr (H != 0) {
    // ---- split-f16 form: the four token tiles as f16 planes (objects A, objects B, hints, buf)
    char* base = reinterpret_cast<char*>(sm);
    const FP pa = fp_at(base), pb = fp_at(base + kTileBytes), ph = fp_at(base + 2 * kTileBytes), pbuf = fp_at(base + 3 * kTileBytes);
    float worst = 0.f;  // guard: largest 2-norm of the 96 raw rows (NaN fails the comparison too); 32 threads cover a row
    for (int i = tid; i < 96 * (kFD / 4); i += 256) {  // a float4 of a row per item: 32 items per row
      const int row = i >> 5, c = 4 * (i & 31);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < 64) {
        const int pair = min(pair0 + (row >> 4), n_pairs - 1);  // tail pairs are duplicated (their copies are not written back)
        v = *reinterpret_cast<const float4*>(cell_desc + (size_t)(cell_index ? cell_index[pair] : pair) * kFObj * kFD + (row & 15) * kFD + c);
      } else {
        const int hrow = row - 64, hp = hrow >> 3, hr = hrow & 7;
        if (hr < n_hints) {
          const int hpair = min(pair0 + hp, n_pairs - 1);
          v = *reinterpret_cast<const float4*>(hint_desc + ((size_t)(hint_index ? hint_index[hpair] : hpair) * n_hints + hr) * kFD + c);
        }
      }
      const FP t = row < 32 ? pa : (row < 64 ? pb : ph);
      fp_put4(t, (row & 31) * kLdF + c, ff_f32x4{v.x, v.y, v.z, v.w});
      float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;  // the row's 32 quads sit in 32 consecutive lanes (half a wave)
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
      worst = (ss <= kFGuardNorm * kFGuardNorm) ? worst : 1.f;  // (NaN: the comparison fails)
    }
    const int unsafe = __syncthreads_or(worst != 0.f ? 1 : 0);
    if (tid == 0) wg_flags[blockIdx.x] = unsafe ? 1 : 0;
    if (unsafe) return;  // block-uniform: the f32 launch that follows serves this workgroup
    const TileGroups gobjA{4, kFObj, 32, 0}, gobjB{4, kFObj, 32, 2}, ghint{3, n_hints, 32, 0};
    for (int l = 0; l < P.n_layers; ++l) {  // cross_matcher.py:114-118
      fp_decoder<H>(pa, gobjA, ph, ghint, P.obj[l], pbuf, red);
      fp_decoder<H>(pb, gobjB, ph, ghint, P.obj[l], pbuf, red);
      fp_decoder<H>(ph, ghint, pa, gobjA, P.hint[l], pbuf, red, &pb, gobjB);
    }
    if (P.n_layers == 0) fp_decoder<H>(ph, ghint, pa, gobjA, P.hint[0], pbuf, red, &pb, gobjB);  // cross_matcher.py:119-120
    for (int i = tid; i < kPairs * kFD; i += 256) {  // desc1.max(dim=0) over the hints (cross_matcher.py:128)
      const int p = i >> 7, c = i & 127;
      auto at = [&](int t) { return (float)ph.hi[(8 * p + t) * kLdF + c] + (float)ph.lo[(8 * p + t) * kLdF + c]; };
      float m = at(0);
      for (int t = 1; t < n_hints; ++t) m = fmaxf(m, at(t));
      pooled[p * kFD + c] = m;
    }
  } else {
  float* d0 = sm;                  // [64][kFS] objects: pair p at rows 16p.. (tile A = pairs 0-1, tile B = pairs 2-3)
  float* d1 = d0 + 64 * kFS;       // [32][kFS] hints:   pair p at rows 8p..8p+n_hints-1
  float* buf = d1 + 32 * kFS;      // [32][kFS]
  for (int i = tid; i < 64 * kFD; i += 256) {
    const int row = i / kFD, c = i % kFD, p = row >> 4;
    const int pair = min(pair0 + p, n_pairs - 1);  // tail pairs are duplicated (their copies are not written back)
    d0[row * kFS + c] = cell_desc[(size_t)(cell_index ? cell_index[pair] : pair) * kFObj * kFD + (row & 15) * kFD + c];
  }
  for (int i = tid; i < 32 * kFD; i += 256) {
    const int row = i / kFD, c = i % kFD, hp = row >> 3, hr = row & 7;
    float v = 0.f;
    if (hr < n_hints) {
      const int hpair = min(pair0 + hp, n_pairs - 1);
      v = hint_desc[((size_t)(hint_index ? hint_index[hpair] : hpair) * n_hints + hr) * kFD + c];
    }
    d1[row * kFS + c] = v;
  }
  __syncthreads();
  float* d0b = d0 + 32 * kFS;
  const TileGroups gobjA{4, kFObj, 32, 0}, gobjB{4, kFObj, 32, 2}, ghint{3, n_hints, 32, 0};
  for (int l = 0; l < P.n_layers; ++l) {  // cross_matcher.py:114-118
    f_decoder<H>(d0, gobjA, d1, ghint, P.obj[l], buf);
    f_decoder<H>(d0b, gobjB, d1, ghint, P.obj[l], buf);
    f_decoder<H>(d1, ghint, d0, gobjA, P.hint[l], buf, d0b, gobjB);
  }
  if (P.n_layers == 0) f_decoder<H>(d1, ghint, d0, gobjA, P.hint[0], buf, d0b, gobjB);  // cross_matcher.py:119-120
  for (int i = tid; i < kPairs * kFD; i += 256) {  // desc1.max(dim=0) over the hints (cross_matcher.py:128)
    const int p = i >> 7, c = i & 127;
    float m = d1[(8 * p) * kFS + c];
    for (int t = 1; t < n_hints; ++t) m = fmaxf(m, d1[(8 * p + t) * kFS + c]);
    pooled[p * kFD + c] = m;
  }
  }  // f32 form
  __syncthreads();  // mlp_offsets (cross_matcher.py:129-131)
  {
    const int p = tid >> 6, n = tid & 63;  // 4 pairs x 64 hidden units = the 256 threads
    float s = P.off0.b[n];
    for (int k = 0; k < kFD; ++k) s += pooled[p * kFD + k] * P.off0.wt[k * 64 + n];
    h64[p * 64 + n] = fmaxf(s, 0.f);
  }
  __syncthreads();
  if (tid < kPairs * 2) {
    const int p = tid >> 1, n = tid & 1;
    float s = P.off2.b[n];
    for (int k = 0; k < 64; ++k) s += h64[p * 64 + k] * P.off2.wt[k * 2 + n];
    if (pair0 + p < n_pairs) out[(size_t)(pair0 + p) * 2 + n] = s;
  }
}

int fine_encode_impl(t2l_ctx* ctx, const t2l_packed_cells* in, float* out, hipStream_t s) {
  FineWeights* W = reinterpret_cast<FineWeights*>(ctx->fine);
  if (!W) return fail(ctx, T2L_ESTATE, "t2l_fine_encode_objects: fine weights not loaded (call t2l_fine_load_weights)");
  if (!in || !out || in->n_cells <= 0 || in->n_objects != in->n_cells * kFObj)
    return fail(ctx, T2L_EINVAL, "t2l_fine_encode_objects: every cell must hold exactly pad_size = 16 objects (pad or cut on the host, "
                                 "as Kitti360TopKDataset.load_pose_and_cell does)");
  const FineParams& P = W->p;
  if ((P.use_class && (P.class_embed ? !in->class_idx : !in->pn_feat)) || (P.use_color && (P.color_embed ? !in->color_idx : !in->rgb)) ||
      (P.use_pos && !in->center) || (P.use_num && !in->n_pts))
    return fail(ctx, T2L_EINVAL, "t2l_fine_encode_objects: a packed input the configuration needs is NULL");
  event_begin(ctx, "fine_objects", s);
  hipLaunchKernelGGL(fine_objects_kernel, dim3(in->n_cells), dim3(256), 0, s, P, *in, out);
  event_end(ctx, "fine_objects", s);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

int fine_match_impl(t2l_ctx* ctx, const float* cell_desc, const int32_t* cell_index, const float* hint_desc, const int32_t* hint_index,
                    int n_pairs, int n_hints, float* out, hipStream_t s) {
  FineWeights* W = reinterpret_cast<FineWeights*>(ctx->fine);
  if (!W) return fail(ctx, T2L_ESTATE, "t2l_fine_match: fine weights not loaded (call t2l_fine_load_weights)");
  if (!cell_desc || !hint_desc || !out || n_pairs < 0) return fail(ctx, T2L_EINVAL, "t2l_fine_match: null argument");
  if (n_hints < 1 || n_hints > kFHintMax) return fail(ctx, T2L_EINVAL, "t2l_fine_match: 1 <= n_hints <= 8");
  if (n_pairs == 0) return T2L_OK;
  const size_t lds = (size_t)4 * kTileBytes + sizeof(float) * (kPairs * kFD + kPairs * 64 + 256);  // 73.7 KB: two workgroups per CU
  static PerDeviceOnce attr;
  if (attr.need(ctx->device)) {
    T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&fine_match_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&fine_match_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&fine_match_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr.mark(ctx->device);
  }
  const int n_wg = (n_pairs + kPairs - 1) / kPairs;
  const bool split = W->p.split_ok && !ctx->encoder_f32;
  if (split && (size_t)n_wg * sizeof(int32_t) > W->flag_cap) {
    if (W->wg_flags) (void)hipFree(W->wg_flags);
    W->wg_flags = nullptr;
    W->flag_cap = 0;
    T2L_HIP(ctx, hipMalloc(&W->wg_flags, (size_t)n_wg * sizeof(int32_t)));
    W->flag_cap = (size_t)n_wg * sizeof(int32_t);
  }
  event_begin(ctx, "fine_match", s);
  if (split) {
    // option encoder_f16: one f16 product per operand pair instead of three (offsets within ~1e-4 instead of 3e-7; same guard)
    if (ctx->encoder_f16)
      hipLaunchKernelGGL(fine_match_kernel<2>, dim3(n_wg), dim3(256), lds, s, W->p, cell_desc, cell_index, hint_desc, hint_index, n_pairs,
                         n_hints, out, W->wg_flags);
    else
      hipLaunchKernelGGL(fine_match_kernel<1>, dim3(n_wg), dim3(256), lds, s, W->p, cell_desc, cell_index, hint_desc, hint_index, n_pairs,
                         n_hints, out, W->wg_flags);
    hipLaunchKernelGGL(fine_match_kernel<0>, dim3(n_wg), dim3(256), lds, s, W->p, cell_desc, cell_index, hint_desc, hint_index, n_pairs,
                       n_hints, out, W->wg_flags);  // only the workgroups the guard turned away
  } else {
    hipLaunchKernelGGL(fine_match_kernel<0>, dim3(n_wg), dim3(256), lds, s, W->p, cell_desc, cell_index, hint_desc, hint_index, n_pairs,
                       n_hints, out, (int32_t*)nullptr);
  }
  event_end(ctx, "fine_match", s);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

}  // namespace t2l
