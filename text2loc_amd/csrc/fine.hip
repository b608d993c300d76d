// Fine stage, eval mode: CrossMatch.forward downstream of the text branch
// (models/cross_matcher.py:86-135) —
//   t2l_fine_encode_objects : ObjectEncoder at fine_embed_dim (=128) + F.normalize per padded cell (16 objects)
//   t2l_fine_match          : per (query, cell) pair the cascaded cross-attention decoder layers
//                             (cross_objects[i](obj, hints); cross_hints[i](hints, obj)), max over hints, mlp_offsets
// The reference runs one Python-level forward per query over its top-k cells (evaluation/pipeline.py:113-116) and
// re-encodes the same cells for every query that retrieved them; here the per-cell object descriptors are computed
// once per database cell and the pairs are ONE launch (one workgroup per pair, everything LDS-resident).
// First version: f32 VALU contractions (tokens per pair: 16 objects / 6 hints, d = 128 — 23 MFLOP per pair, the
// problem is tiny and latency-bound); BatchNorm folded on the host, weights stored transposed [K][N] so that the
// threads of a wave read consecutive output columns.
#include <math.h>
#include <string.h>

#include "t2l_internal.h"

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
struct FDecoder {
  FLinear sa_in, sa_out, ca_in, ca_out, l1, l2;
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
};
struct FineWeights {
  FineParams p{};
  std::vector<void*> blobs;
};

void free_fine(t2l_ctx* ctx) {
  FineWeights* W = reinterpret_cast<FineWeights*>(ctx->fine);
  if (!W) return;
  for (void* b : W->blobs) (void)hipFree(b);
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
static int fraw(t2l_ctx* ctx, FineWeights* W, const WMap& m, const std::string& k, int64_t n, const float** dst) {
  const float* p = fget(m, k, n);
  if (!p) return fail(ctx, T2L_EINVAL, "t2l_fine_load_weights: missing or mis-shaped '" + k + "'");
  return fupload(ctx, W, std::vector<float>(p, p + n), dst);
}

int fine_load_impl(t2l_ctx* ctx, const t2l_weight_desc* w, int n, const t2l_model_config* cfg) {
  if (!w || n <= 0 || !cfg) return fail(ctx, T2L_EINVAL, "t2l_fine_load_weights: null argument");
  if (cfg->num_heads != kFHeads || cfg->num_layers < 1 || cfg->num_layers > 4)
    return fail(ctx, T2L_EINVAL, "t2l_fine_load_weights: built for 4 decoder heads and 1..4 decoder layers");
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
  // nn.MultiheadAttention stores in_proj_weight / in_proj_bias without a sub-module: adapt the names
  auto inproj = [&](const std::string& p, FLinear* out) -> int {
    const float *wq = fget(m, p + ".in_proj_weight", 3 * kFD * kFD), *bq = fget(m, p + ".in_proj_bias", 3 * kFD);
    if (!wq || !bq) return fail(ctx, T2L_EINVAL, "t2l_fine_load_weights: missing or mis-shaped '" + p + ".in_proj_weight'");
    std::vector<float> wt((size_t)kFD * 3 * kFD);
    for (int nn = 0; nn < 3 * kFD; ++nn)
      for (int k = 0; k < kFD; ++k) wt[(size_t)k * 3 * kFD + nn] = wq[(size_t)nn * kFD + k];
    int r;
    if ((r = fupload(ctx, W, wt, &out->wt))) return r;
    return fupload(ctx, W, std::vector<float>(bq, bq + 3 * kFD), &out->b);
  };
  for (int l = 0; l < P.n_layers; ++l)
    for (int which = 0; which < 2; ++which) {
      const std::string p = std::string(which ? "cross_hints." : "cross_objects.") + std::to_string(l);
      FDecoder& D = which ? P.hint[l] : P.obj[l];
      if ((rc = inproj(p + ".self_attn", &D.sa_in)) || (rc = flinear(ctx, W, m, p + ".self_attn.out_proj", "", kFD, kFD, &D.sa_out)) ||
          (rc = inproj(p + ".multihead_attn", &D.ca_in)) || (rc = flinear(ctx, W, m, p + ".multihead_attn.out_proj", "", kFD, kFD, &D.ca_out)) ||
          (rc = flinear(ctx, W, m, p + ".linear1", "", kFD, 4 * kFD, &D.l1)) || (rc = flinear(ctx, W, m, p + ".linear2", "", 4 * kFD, kFD, &D.l2)) ||
          (rc = fraw(ctx, W, m, p + ".norm1.weight", kFD, &D.g1)) || (rc = fraw(ctx, W, m, p + ".norm1.bias", kFD, &D.b1)) ||
          (rc = fraw(ctx, W, m, p + ".norm2.weight", kFD, &D.g2)) || (rc = fraw(ctx, W, m, p + ".norm2.bias", kFD, &D.b2)) ||
          (rc = fraw(ctx, W, m, p + ".norm3.weight", kFD, &D.g3)) || (rc = fraw(ctx, W, m, p + ".norm3.bias", kFD, &D.b3)))
        return rc;
    }
  if ((rc = flinear(ctx, W, m, "mlp_offsets.0", "", kFD, kFD / 2, &P.off0)) || (rc = flinear(ctx, W, m, "mlp_offsets.2", "", kFD / 2, 2, &P.off2)))
    return rc;
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

__device__ __forceinline__ float f_wsum(float v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
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
// x[t] = LayerNorm(x[t] + a[t]) * g + b  for t < T
__device__ void f_add_ln(float* x, const float* a, int ld, int T, const float* __restrict__ g, const float* __restrict__ b) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int t = w; t < T; t += 4) {
    const float v0 = x[t * ld + lane] + a[t * ld + lane], v1 = x[t * ld + lane + 64] + a[t * ld + lane + 64];
    const float mu = f_wsum(v0 + v1) * (1.f / kFD);
    const float d0 = v0 - mu, d1 = v1 - mu;
    const float rstd = 1.0f / sqrtf(f_wsum(d0 * d0 + d1 * d1) * (1.f / kFD) + 1e-5f);
    x[t * ld + lane] = d0 * rstd * g[lane] + b[lane];
    x[t * ld + lane + 64] = d1 * rstd * g[lane + 64] + b[lane + 64];
  }
}
// multi-head attention: q [T][ldq] (cols qc..), k/v [S][ldk] (cols kc.. / vc..), 4 heads x 32 -> o [T][ldo]. prob: [4][16][16] scratch
__device__ void f_attention(const float* q, int ldq, int qc, const float* kv, int ldk, int kc, int vc, int T, int S, float* prob, float* o, int ldo) {
  const float scale = 0.17677669529663687f;  // 1/sqrt(32)
  for (int e = threadIdx.x; e < kFHeads * T * S; e += 256) {
    const int h = e / (T * S), r = e % (T * S), i = r / S, j = r % S;
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < kFHd; ++d) s += q[i * ldq + qc + h * kFHd + d] * kv[j * ldk + kc + h * kFHd + d];
    prob[(h * 16 + i) * 16 + j] = s * scale;
  }
  __syncthreads();
  if (threadIdx.x < kFHeads * T) {
    float* row = prob + ((threadIdx.x / T) * 16 + (threadIdx.x % T)) * 16;
    float mx = row[0];
    for (int j = 1; j < S; ++j) mx = fmaxf(mx, row[j]);
    float sum = 0.f;
    for (int j = 0; j < S; ++j) { row[j] = expf(row[j] - mx); sum += row[j]; }
    for (int j = 0; j < S; ++j) row[j] /= sum;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < T * kFD; e += 256) {
    const int i = e / kFD, c = e % kFD, h = c / kFHd;
    float s = 0.f;
    for (int j = 0; j < S; ++j) s += prob[(h * 16 + i) * 16 + j] * kv[j * ldk + vc + c];
    o[i * ldo + c] = s;
  }
  __syncthreads();
}

// nn.TransformerDecoderLayer (post-norm, ReLU, eval, no masks): x [T] (LDS, stride kFS) attends itself, then mem [S].
// big: [16][516] floats, holds q|k|v (stride 388) during the attentions and the feed-forward hidden (stride 516) afterwards;
// att / tmp: [16][kFS] temporaries.
__device__ void f_decoder(float* x, int T, const float* mem, int S, const FDecoder D, float* big, float* att, float* prob, float* tmp) {
  float* qkv = big;
  float* hid = tmp;
  const int T8 = (T + 7) / 8;
  f_linear(x, kFS, T8, D.sa_in, kFD, 3 * kFD, qkv, 388, 0, false);
  __syncthreads();
  f_attention(qkv, 388, 0, qkv, 388, kFD, 2 * kFD, T, T, prob, att, kFS);
  f_linear(att, kFS, T8, D.sa_out, kFD, kFD, hid, kFS, 0, false);
  __syncthreads();
  f_add_ln(x, hid, kFS, T, D.g1, D.b1);
  __syncthreads();
  // cross attention: q from x (first 128 columns of in_proj), k/v from mem (columns 128..383)
  FLinear qproj{D.ca_in.wt, D.ca_in.b};
  {  // q = x Wq: the [K][3D] transposed layout makes the first D output columns the query projection
    for (int item = threadIdx.x; item < kFD * T8; item += 256) {
      const int n = item % kFD, tg = item / kFD;
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = qproj.b[n];
      for (int k = 0; k < kFD; k += 4) {
        const float w0 = qproj.wt[(size_t)(k + 0) * 3 * kFD + n], w1 = qproj.wt[(size_t)(k + 1) * 3 * kFD + n],
                    w2 = qproj.wt[(size_t)(k + 2) * 3 * kFD + n], w3 = qproj.wt[(size_t)(k + 3) * 3 * kFD + n];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float4 v = *reinterpret_cast<const float4*>(x + (tg * 8 + j) * kFS + k);
          acc[j] += v.x * w0 + v.y * w1 + v.z * w2 + v.w * w3;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) att[(tg * 8 + j) * kFS + n] = acc[j];
    }
    const int S8 = (S + 7) / 8;
    for (int item = threadIdx.x; item < 2 * kFD * S8; item += 256) {
      const int n = item % (2 * kFD), tg = item / (2 * kFD);
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = qproj.b[kFD + n];
      for (int k = 0; k < kFD; k += 4) {
        const float w0 = qproj.wt[(size_t)(k + 0) * 3 * kFD + kFD + n], w1 = qproj.wt[(size_t)(k + 1) * 3 * kFD + kFD + n],
                    w2 = qproj.wt[(size_t)(k + 2) * 3 * kFD + kFD + n], w3 = qproj.wt[(size_t)(k + 3) * 3 * kFD + kFD + n];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float4 v = *reinterpret_cast<const float4*>(mem + (tg * 8 + j) * kFS + k);
          acc[j] += v.x * w0 + v.y * w1 + v.z * w2 + v.w * w3;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) qkv[(tg * 8 + j) * 388 + n] = acc[j];
    }
  }
  __syncthreads();
  f_attention(att, kFS, 0, qkv, 388, 0, kFD, T, S, prob, hid, kFS);
  f_linear(hid, kFS, T8, D.ca_out, kFD, kFD, att, kFS, 0, false);
  __syncthreads();
  f_add_ln(x, att, kFS, T, D.g2, D.b2);
  __syncthreads();
  f_linear(x, kFS, T8, D.l1, kFD, 4 * kFD, big, 516, 0, true);
  __syncthreads();
  f_linear(big, 516, T8, D.l2, 4 * kFD, kFD, att, kFS, 0, false);
  __syncthreads();
  f_add_ln(x, att, kFS, T, D.g3, D.b3);
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

// One workgroup per (query, cell) pair.
__global__ __launch_bounds__(256, 2) void fine_match_kernel(FineParams P, const float* __restrict__ cell_desc, const int32_t* __restrict__ cell_index,
                                                         const float* __restrict__ hint_desc, const int32_t* __restrict__ hint_index,
                                                         int n_hints, float* __restrict__ out) {
  extern __shared__ float sm[];
  float* d0 = sm;                          // [16][kFS]
  float* d1 = d0 + kFObj * kFS;            // [8][kFS]
  float* big = d1 + kFHintMax * kFS;       // [16][516]: q|k|v during the attentions, feed-forward hidden afterwards
  float* att = big + kFObj * 516;          // [16][kFS]
  float* tmp = att + kFObj * kFS;          // [16][kFS]
  float* prob = tmp + kFObj * kFS;         // [4][16][16]
  float* pooled = prob + kFHeads * 16 * 16;  // [128]
  float* h64 = pooled + kFD;               // [64]
  const int pair = blockIdx.x, tid = threadIdx.x;
  const float* cd = cell_desc + (size_t)(cell_index ? cell_index[pair] : pair) * kFObj * kFD;
  const float* hd = hint_desc + (size_t)(hint_index ? hint_index[pair] : pair) * n_hints * kFD;
  for (int i = tid; i < kFObj * kFD; i += 256) d0[(i / kFD) * kFS + (i % kFD)] = cd[i];
  for (int i = tid; i < kFHintMax * kFD; i += 256) d1[(i / kFD) * kFS + (i % kFD)] = (i / kFD) < n_hints ? hd[i] : 0.f;
  __syncthreads();
  for (int l = 0; l < P.n_layers; ++l) {  // cross_matcher.py:114-118
    f_decoder(d0, kFObj, d1, n_hints, P.obj[l], big, att, prob, tmp);
    f_decoder(d1, n_hints, d0, kFObj, P.hint[l], big, att, prob, tmp);
  }
  if (tid < kFD) {  // desc1.max(dim=0) over the hints
    float m = d1[tid];
    for (int t = 1; t < n_hints; ++t) m = fmaxf(m, d1[t * kFS + tid]);
    pooled[tid] = m;
  }
  __syncthreads();
  if (tid < 64) {
    float s = P.off0.b[tid];
    for (int k = 0; k < kFD; ++k) s += pooled[k] * P.off0.wt[k * 64 + tid];
    h64[tid] = fmaxf(s, 0.f);
  }
  __syncthreads();
  if (tid < 2) {
    float s = P.off2.b[tid];
    for (int k = 0; k < 64; ++k) s += h64[k] * P.off2.wt[k * 2 + tid];
    out[(size_t)pair * 2 + tid] = s;
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
  const size_t lds = sizeof(float) * (kFObj * kFS + kFHintMax * kFS + kFObj * 516 + 2 * kFObj * kFS + kFHeads * 16 * 16 + kFD + 64);  // 67.5 KB: two pairs per CU
  static bool attr = false;
  if (!attr) {
    T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&fine_match_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  event_begin(ctx, "fine_match", s);
  hipLaunchKernelGGL(fine_match_kernel, dim3(n_pairs), dim3(256), lds, s, W->p, cell_desc, cell_index, hint_desc, hint_index, n_hints, out);
  event_end(ctx, "fine_match", s);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

}  // namespace t2l
