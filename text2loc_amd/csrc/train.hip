// Training step of the object branch (training/coarse.py:31-58): host orchestration.
//   t2l_train_bind              live parameter / gradient / BatchNorm-buffer pointers (no copies: the optimizer's tensors)
//   t2l_encode_cells_train      CellRetrievalNetwork.encode_objects under model.train() (batch-statistics BatchNorm,
//                               the four dropout sites of each TransformerEncoderLayer), activations kept for backward
//   t2l_encode_cells_backward   what autograd does for that graph: gradients accumulated (+=) into the bound buffers
//   t2l_adam_step / t2l_zero_grad   torch.optim.Adam defaults / optimizer.zero_grad()
// Kernels: train_kernels.h.
#include <math.h>
#include <string.h>

#include "t2l_internal.h"
#include "train_kernels.h"

namespace t2l {

using namespace train;

struct TTensor {
  float* data = nullptr;
  float* grad = nullptr;
  int64_t numel = 0;
};

// One [Linear, BatchNorm1d, ReLU] block of get_mlp (models/language_encoder.py:16-41) with its saved activations.
struct MlpLayer {
  std::string prefix;  // e.g. "object_encoder.pos_encoder.1"
  int cin = 0, cout = 0;
  float *y = nullptr, *a = nullptr, *mean = nullptr, *rstd = nullptr;  // pre-BN, post-ReLU, batch statistics
};
struct Branch {
  int kind = 0;  // 0 embedding lookup, 1 small MLP (K<=3 -> 64 -> 256), 2 PointNet-feature MLP (256 -> 256)
  int slot = 0;  // 256-wide slot of the concatenated feature row
  std::string table;          // kind 0
  const int32_t* idx = nullptr;
  const float* x = nullptr;   // kind 1/2 input
  int k_in = 0, standardize = 0;
  std::vector<MlpLayer> layers;
  float* save_n = nullptr;
};
struct LayerSave {
  std::string prefix;
  const float* x_in = nullptr;
  float *qkv = nullptr, *P = nullptr, *O = nullptr, *xhat1 = nullptr, *rstd1 = nullptr, *x1 = nullptr, *h = nullptr,
        *hd = nullptr, *xhat2 = nullptr, *rstd2 = nullptr, *x2 = nullptr;
};

constexpr int kBnSlots = 8;  // BatchNorm layers per pass: <= 2*3 small branches + pointnet mlp + merge

struct TrainState {
  t2l_ctx* ctx = nullptr;
  std::unordered_map<std::string, TTensor> t;
  t2l_model_config cfg{};
  std::vector<std::string> adam_names;
  AdamTensor* d_tensors = nullptr;
  AdamChunk* d_chunks = nullptr;
  float* mv = nullptr;       // [2][mv_total]: first moments of all Adam tensors (adam_names order), then second moments
  int64_t mv_total = 0;
  double* bn_acc = nullptr;  // [2][kBnSlots][kBnStride] float64 partial sums of the BatchNorm stages (one slot per BN pass; forward | backward)
  int bn_slot = 0;
  bool fwd_acc_clean = false;  // the previous forward's last launch zeroed all accumulators (no memset launch in a step)
  bool bwd_acc_clean = false;  // the forward zeroed the backward half too; false after a backward used it (a second backward memsets)
  int n_chunks = 0;
  int pn_chunk0 = 0;         // chunks [pn_chunk0, n_chunks) belong to the PointNet++ backbone (bound last)
  int64_t step = 0;          // Adam step of the object branch
  int64_t step_pn = 0;       // ... of the backbone: it only steps when its backward ran since the last zero_grad (torch.optim.Adam
  bool pn_touched = false;   // skips parameters whose .grad is None and keeps a step count per parameter)
  // workspace (bump-allocated per forward)
  char* ws = nullptr;
  size_t ws_cap = 0, ws_off = 0;
  // the last forward
  bool have_forward = false;
  int M = 0, B = 0, T = 0, n_feat = 0;
  const int32_t* offsets = nullptr;
  std::vector<Branch> branches;
  MlpLayer merge;
  float *cat = nullptr, *X0 = nullptr, *save_nf = nullptr, *out = nullptr, *pool_n = nullptr;
  int32_t* pool_arg = nullptr;
  std::vector<LayerSave> layers;
  uint32_t seed = 0;
  float p = 0.f;
  void* pn = nullptr;  // PnTrain (pointnet_train.h): the PointNet++ backbone's training state, when its tensors are bound
};

static void pn_train_free(void* p);
static int pn_train_bind(t2l_ctx* ctx, TrainState* st, std::vector<std::string>& adam);

static TrainState* state(t2l_ctx* ctx) { return reinterpret_cast<TrainState*>(ctx->train); }

// ---- cross-rank BatchNorm statistics (t2l_train_sync_bn): slots [0, 2 kBnSlots) are the object branch's (forward | backward), the two
// behind them the text head's inter_mlp (forward, backward)
int64_t train_sync_bn_doubles() { return (int64_t)(2 * kBnSlots + 2) * kBnStride; }
void train_sync_changed(t2l_ctx* ctx) {
  if (TrainState* st = state(ctx)) st->fwd_acc_clean = st->bwd_acc_clean = false;
}
static double* acc_base(t2l_ctx* ctx, TrainState* st) { return ctx->sync_fn ? ctx->sync_buf : st->bn_acc; }
// sum `slots` consecutive accumulator slots over the ranks (between a statistics launch and its apply launch)
static void sync_slots(t2l_ctx* ctx, double* first, int slots, hipStream_t s) {
  if (!ctx->sync_fn) return;
  if (ctx->sync_fn(ctx->sync_user, first, (int64_t)slots * kBnStride, (void*)s) != 0) ctx->sync_failed = true;
}

void free_train(t2l_ctx* ctx) {
  TrainState* st = state(ctx);
  if (!st) return;
  for (void* p : {(void*)st->d_tensors, (void*)st->d_chunks, (void*)st->mv, (void*)st->ws, (void*)st->bn_acc})
    if (p) (void)hipFree(p);
  pn_train_free(st->pn);
  delete st;
  ctx->train = nullptr;
}

static Drop make_drop(uint32_t seed, int site, float p) {
  Drop d;
  d.key = seed ^ (uint32_t)((uint64_t)site * 0x85EBCA77ull);
  d.thr = p > 0.f ? (uint32_t)((double)p * 16777216.0) : 0u;
  d.scale = d.thr ? 1.0f / (1.0f - p) : 1.0f;
  return d;
}

template <typename T>
static T* bump(TrainState* st, size_t count) {
  size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
  T* p = reinterpret_cast<T*>(st->ws + st->ws_off);
  st->ws_off += bytes;
  return p;
}

// ---- GEMM launchers -----------------------------------------------------------------------------------------
static thread_local int tl_gemm_bf16 = 0;  // set from the context option "train_bf16" at the top of every forward / backward
static thread_local int tl_gemm_block64 = 0;  // option "train_gemm_block": 64 x 64 output blocks where the shape allows (default: with bf16 operands)
static inline bool blk64(int rows_out_mult, int cols_out) { return tl_gemm_block64 && rows_out_mult % 64 == 0 && cols_out % 64 == 0; }
// Y[M,N] = X[M,K] W[N,K]^T + b (relu)
static void gemm_nt_args(GemmArgs g, hipStream_t s) {  // g.M rows (ragged allowed), g.N columns
  if (blk64(64, g.N))
    hipLaunchKernelGGL((gemm4_kernel<true, true>), dim3(g.N / 64, (g.M + 63) / 64, 1), dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL((gemm_kernel<true, true>), dim3(g.N / 32, (g.M + 31) / 32, 1), dim3(256), 0, s, g);
}
static void gemm_nt(const float* X, const float* W, const float* b, float* Y, int M, int N, int K, int relu, hipStream_t s) {
  gemm_nt_args(GemmArgs{X, W, Y, b, M, N, K, K, K, N, relu, 0, K, nullptr, tl_gemm_bf16}, s);
}
// dX[M,Kp] (+)= dY[M,N] W[N,Kp]
static void gemm_nn(const float* dY, const float* W, float* dX, int M, int N, int Kp, int accumulate, hipStream_t s) {
  GemmArgs g{dY, W, dX, nullptr, M, Kp, N, N, Kp, Kp, 0, accumulate, N, nullptr, tl_gemm_bf16};
  if (blk64(64, Kp))
    hipLaunchKernelGGL((gemm4_kernel<true, false>), dim3(Kp / 64, (M + 63) / 64, 1), dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL((gemm_kernel<true, false>), dim3(Kp / 32, (M + 31) / 32, 1), dim3(256), 0, s, g);
}
// reduction split of dW[N,Kp] += dY[M,N]^T X[M,Kp] over the M rows: >= 64 rows per wave, ~1k workgroups
static void tn_split(int M, int N, int Kp, int blk, int& ksplit, int& kchunk) {
  const int tiles = (N / blk) * (Kp / blk);
  ksplit = std::max(1, std::min((M + 255) / 256, (1024 + tiles - 1) / tiles));
  kchunk = (((M + ksplit - 1) / ksplit) + 63) & ~63;
  ksplit = (M + kchunk - 1) / kchunk;
}
// dW[N,Kp] += dY[M,N]^T X[M,Kp]   (reduction over the M rows, split over grid.z, float atomics);  db[N] += column sums of dY
static void gemm_tn(const float* dY, const float* X, float* dW, float* db, int M, int N, int Kp, hipStream_t s) {
  const bool b64 = blk64(N, Kp);
  int ksplit, kchunk;
  tn_split(M, N, Kp, b64 ? 64 : 32, ksplit, kchunk);
  GemmArgs g{dY, X, dW, nullptr, N, Kp, M, N, Kp, Kp, 0, 1, kchunk, db, tl_gemm_bf16};
  if (b64)
    hipLaunchKernelGGL((gemm4_kernel<false, false>), dim3(Kp / 64, N / 64, ksplit), dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL((gemm_kernel<false, false>), dim3(Kp / 32, N / 32, ksplit), dim3(256), 0, s, g);
}
// dW[N,Kp] += dY^T X (+ db) and dX[M,Kp] (+)= dY W in ONE launch (both read dY only); mask_src / drop: the ReLU + dropout backward
// of the layer that produced X's pre-image, applied in dX's epilogue (epi 2) instead of by a launch of its own
static void gemm_tn_nn(const float* dY, const float* X, float* dW, float* db, const float* W, float* dX, int M, int N, int Kp, int accumulate,
                       const float* mask_src, const Drop* drop, hipStream_t s) {
  const bool b64 = blk64(N, Kp);
  const int blk = b64 ? 64 : 32;
  int ksplit, kchunk;
  tn_split(M, N, Kp, blk, ksplit, kchunk);
  GemmPair p{};
  p.tn = GemmArgs{dY, X, dW, nullptr, N, Kp, M, N, Kp, Kp, 0, 1, kchunk, db, tl_gemm_bf16};
  p.nn = GemmArgs{dY, W, dX, nullptr, M, Kp, N, N, Kp, Kp, 0, accumulate, N, nullptr, tl_gemm_bf16};
  if (mask_src) {
    p.nn.epi = 2;
    p.nn.mask_src = mask_src;
    p.nn.drop_key = drop->key;
    p.nn.drop_thr = drop->thr;
    p.nn.drop_scale = drop->scale;
  }
  p.tn_gx = Kp / blk;
  p.tn_gy = N / blk;
  p.tn_blocks = p.tn_gx * p.tn_gy * ksplit;
  p.nn_gx = Kp / blk;
  const int nn_blocks = p.nn_gx * ((M + blk - 1) / blk);
  if (b64)
    hipLaunchKernelGGL(gemm4_pair_kernel, dim3(p.tn_blocks + nn_blocks), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL(gemm_pair_kernel, dim3(p.tn_blocks + nn_blocks), dim3(256), 0, s, p);
}
static int need(t2l_ctx* ctx, TrainState* st, const std::string& name, int64_t numel, bool with_grad, TTensor** out) {
  auto it = st->t.find(name);
  if (it == st->t.end()) return fail(ctx, T2L_EINVAL, "t2l_train_bind: missing tensor '" + name + "'");
  if (it->second.numel != numel)
    return fail(ctx, T2L_EINVAL, "t2l_train_bind: '" + name + "' has " + std::to_string(it->second.numel) + " elements, expected " +
                                     std::to_string(numel));
  if (with_grad && !it->second.grad) return fail(ctx, T2L_EINVAL, "t2l_train_bind: '" + name + "' needs a gradient buffer");
  if (out) *out = &it->second;
  return T2L_OK;
}

static const TTensor& T_(TrainState* st, const std::string& n) { return st->t.at(n); }

// names + shapes of one get_mlp block
static int check_mlp_layer(t2l_ctx* ctx, TrainState* st, const std::string& p, int cin, int cout, std::vector<std::string>& params) {
  int rc;
  if ((rc = need(ctx, st, p + ".0.weight", (int64_t)cin * cout, true, nullptr))) return rc;
  if ((rc = need(ctx, st, p + ".0.bias", cout, true, nullptr))) return rc;
  if ((rc = need(ctx, st, p + ".1.weight", cout, true, nullptr))) return rc;
  if ((rc = need(ctx, st, p + ".1.bias", cout, true, nullptr))) return rc;
  if ((rc = need(ctx, st, p + ".1.running_mean", cout, false, nullptr))) return rc;
  if ((rc = need(ctx, st, p + ".1.running_var", cout, false, nullptr))) return rc;
  for (const char* sfx : {".0.weight", ".0.bias", ".1.weight", ".1.bias"}) params.push_back(p + sfx);
  return T2L_OK;
}

int train_bind_impl(t2l_ctx* ctx, const t2l_train_tensor* tensors, int n, const t2l_model_config* cfg) {
  if (!tensors || n <= 0 || !cfg) return fail(ctx, T2L_EINVAL, "t2l_train_bind: null argument");
  if (cfg->num_heads != kTH) return fail(ctx, T2L_EINVAL, "t2l_train_bind: the engine is built for 4 attention heads");
  const int n_feat = (cfg->use_class != 0) + (cfg->use_color != 0) + (cfg->use_position != 0) + (cfg->use_num != 0);
  if (n_feat < 2) return fail(ctx, T2L_EINVAL, "t2l_train_bind: training needs at least two of the class/color/position/num features");
  // a re-bind of the SAME model (model.to(), an externally assigned .grad, ...) only moves POINTERS: with option
  // "train_keep_adam_state" set (the Python seam sets it for exactly that case) the Adam moments and the bias-correction
  // step of the previous binding carry over when the parameter list (names and sizes) is unchanged
  TrainState* old = state(ctx);
  std::vector<std::string> old_names;
  std::vector<int64_t> old_numel;
  float* old_mv = nullptr;
  int64_t old_step = 0, old_step_pn = 0;
  if (old && ctx->train_keep_adam) {
    old_names = old->adam_names;
    for (auto& nme : old_names) old_numel.push_back(old->t[nme].numel);
    old_mv = old->mv;
    old->mv = nullptr;  // survives free_train below
    old_step = old->step;
    old_step_pn = old->step_pn;
  }
  struct MvGuard {
    float* p;
    ~MvGuard() { if (p) (void)hipFree(p); }
  } guard{old_mv};
  free_train(ctx);
  TrainState* st = new TrainState();
  ctx->train = st;
  st->cfg = *cfg;
  st->n_feat = n_feat;
  for (int i = 0; i < n; ++i) {
    if (!tensors[i].name || !tensors[i].data) return fail(ctx, T2L_EINVAL, "t2l_train_bind: null name/data");
    st->t[tensors[i].name] = TTensor{tensors[i].data, tensors[i].grad, tensors[i].numel};
  }
  std::vector<std::string>& P = st->adam_names;
  const std::string oe = "object_encoder.";
  int rc;
  if (cfg->use_class) {
    if (cfg->class_embed) {
      auto it = st->t.find(oe + "class_embedding.weight");
      if (it == st->t.end() || !it->second.grad || it->second.numel % kTD) return fail(ctx, T2L_EINVAL, "t2l_train_bind: class_embedding.weight");
      P.push_back(oe + "class_embedding.weight");
    } else if ((rc = check_mlp_layer(ctx, st, oe + "mlp_pointnet.0", 256, kTD, P))) return rc;
  }
  if (cfg->use_color) {
    if (cfg->color_embed) {
      auto it = st->t.find(oe + "color_embedding.weight");
      if (it == st->t.end() || !it->second.grad || it->second.numel % kTD) return fail(ctx, T2L_EINVAL, "t2l_train_bind: color_embedding.weight");
      P.push_back(oe + "color_embedding.weight");
    } else {
      if ((rc = check_mlp_layer(ctx, st, oe + "color_encoder.0", 3, 64, P))) return rc;
      if ((rc = check_mlp_layer(ctx, st, oe + "color_encoder.1", 64, kTD, P))) return rc;
    }
  }
  if (cfg->use_position) {
    if ((rc = check_mlp_layer(ctx, st, oe + "pos_encoder.0", 3, 64, P))) return rc;
    if ((rc = check_mlp_layer(ctx, st, oe + "pos_encoder.1", 64, kTD, P))) return rc;
  }
  if (cfg->use_num) {
    if ((rc = check_mlp_layer(ctx, st, oe + "num_encoder.0", 1, 64, P))) return rc;
    if ((rc = check_mlp_layer(ctx, st, oe + "num_encoder.1", 64, kTD, P))) return rc;
  }
  if ((rc = check_mlp_layer(ctx, st, oe + "mlp_merge.0", n_feat * kTD, kTD, P))) return rc;
  for (int l = 0; l < cfg->num_layers; ++l) {
    const std::string p = "obj_inter_module." + std::to_string(l);
    const std::pair<const char*, int64_t> req[] = {
        {".self_attn.in_proj_weight", 3 * kTD * kTD}, {".self_attn.in_proj_bias", 3 * kTD},
        {".self_attn.out_proj.weight", kTD * kTD},    {".self_attn.out_proj.bias", kTD},
        {".linear1.weight", 2 * kTD * kTD},           {".linear1.bias", 2 * kTD},
        {".linear2.weight", 2 * kTD * kTD},           {".linear2.bias", kTD},
        {".norm1.weight", kTD},                       {".norm1.bias", kTD},
        {".norm2.weight", kTD},                       {".norm2.bias", kTD}};
    for (auto& r : req) {
      if ((rc = need(ctx, st, p + r.first, r.second, true, nullptr))) return rc;
      P.push_back(p + r.first);
    }
  }
  const size_t n_obj_tensors = P.size();
  if ((rc = pn_train_bind(ctx, st, P))) return rc;  // the PointNet++ backbone, when bound with gradients
  // Adam tables: moments zero-initialised, one chunk per 1,024 elements
  std::vector<AdamTensor> ts;
  std::vector<AdamChunk> cs;
  int64_t total = 0;
  for (auto& nme : P) total += st->t[nme].numel;
  T2L_HIP(ctx, hipMalloc(&st->bn_acc, sizeof(double) * kBnStride * kBnSlots * 2));  // slots of the forward, then of the backward
  st->ctx = ctx;
  T2L_HIP(ctx, hipMalloc(&st->mv, sizeof(float) * 2 * (size_t)total));
  T2L_HIP(ctx, hipMemset(st->mv, 0, sizeof(float) * 2 * (size_t)total));
  st->mv_total = total;
  {
    bool same = old_mv != nullptr && old_names == P;
    for (size_t i = 0; same && i < P.size(); ++i) same = st->t[P[i]].numel == old_numel[i];
    if (same) {
      T2L_HIP(ctx, hipMemcpy(st->mv, old_mv, sizeof(float) * 2 * (size_t)total, hipMemcpyDeviceToDevice));
      st->step = old_step;
      st->step_pn = old_step_pn;
    }
  }
  int64_t off = 0;
  st->pn_chunk0 = -1;
  for (auto& nme : P) {
    const TTensor& t = st->t[nme];
    if (ts.size() == n_obj_tensors) st->pn_chunk0 = (int)cs.size();
    ts.push_back(AdamTensor{t.data, t.grad, st->mv + off, st->mv + total + off, t.numel});
    for (int64_t c = 0; c * 1024 < t.numel; ++c) cs.push_back(AdamChunk{(int32_t)ts.size() - 1, (int32_t)c});
    off += t.numel;
  }
  st->n_chunks = (int)cs.size();
  if (st->pn_chunk0 < 0) st->pn_chunk0 = st->n_chunks;
  T2L_HIP(ctx, hipMalloc(&st->d_tensors, sizeof(AdamTensor) * ts.size()));
  T2L_HIP(ctx, hipMalloc(&st->d_chunks, sizeof(AdamChunk) * cs.size()));
  T2L_HIP(ctx, hipMemcpy(st->d_tensors, ts.data(), sizeof(AdamTensor) * ts.size(), hipMemcpyHostToDevice));
  T2L_HIP(ctx, hipMemcpy(st->d_chunks, cs.data(), sizeof(AdamChunk) * cs.size(), hipMemcpyHostToDevice));
  return T2L_OK;
}

// ---- forward ---------------------------------------------------------------------------------------------------
static void mlp_layer_fwd(TrainState* st, MlpLayer& L, const float* x, int M, int small_k, int standardize, hipStream_t s) {
  const TTensor& W = T_(st, L.prefix + ".0.weight");
  const TTensor& b = T_(st, L.prefix + ".0.bias");
  if (small_k)
    hipLaunchKernelGGL(smallk_fwd_kernel, dim3((M * 64 + 255) / 256), dim3(256), 0, s, x, M, small_k, W.data, b.data, standardize,
                       1826.6844940968194f, 2516.8905096993817f, L.y);
  else
    gemm_nt(x, W.data, b.data, L.y, M, L.cout, L.cin, 0, s);
  const int sync = st->ctx->sync_fn ? 1 : 0;
  double* acc = acc_base(st->ctx, st) + (size_t)(st->bn_slot++ % (2 * kBnSlots)) * kBnStride;
  hipLaunchKernelGGL((bn_stats_kernel<0>), dim3(L.cout / 64, (M + kBnRows - 1) / kBnRows), dim3(256), 0, s, L.y, (const float*)nullptr,
                     (const float*)nullptr, M, L.cout, (const float*)nullptr, (const float*)nullptr, acc, sync, (float*)nullptr, (float*)nullptr);
  sync_slots(st->ctx, acc, 1, s);
  hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3((unsigned)(((size_t)M * L.cout + 255) / 256)), dim3(256), 0, s, L.y, M, L.cout, acc,
                     T_(st, L.prefix + ".1.weight").data, T_(st, L.prefix + ".1.bias").data, T_(st, L.prefix + ".1.running_mean").data,
                     T_(st, L.prefix + ".1.running_var").data, 0.1f, L.a, L.mean, L.rstd, sync);
}

static MlpLayer make_layer(TrainState* st, const std::string& prefix, int cin, int cout, int M) {
  MlpLayer L;
  L.prefix = prefix;
  L.cin = cin;
  L.cout = cout;
  L.y = bump<float>(st, (size_t)M * cout);
  L.a = bump<float>(st, (size_t)M * cout);
  L.mean = bump<float>(st, cout);
  L.rstd = bump<float>(st, cout);
  return L;
}

// ---- the small feature branches (position, point count, colour: [K -> 64 -> 256] + normalize, identical shapes) stage by stage,
// every stage ONE launch over all of them (kMaxJobs = 3): 7 launches instead of 7 per branch, forward and backward
// slot: the launch's jobs take CONSECUTIVE accumulator slots (one cross-rank sum covers them)
static BnJob bn_job(TrainState* st, const MlpLayer& L, float* d, int slot) {
  BnJob j{};
  j.y = L.y;
  j.out = L.a;
  j.d = d;
  j.acc = acc_base(st->ctx, st) + (size_t)(slot % (2 * kBnSlots)) * kBnStride;
  j.gamma = T_(st, L.prefix + ".1.weight").data;
  j.beta = T_(st, L.prefix + ".1.bias").data;
  j.run_mean = T_(st, L.prefix + ".1.running_mean").data;
  j.run_var = T_(st, L.prefix + ".1.running_var").data;
  j.save_mean = L.mean;
  j.save_rstd = L.rstd;
  j.dgamma = T_(st, L.prefix + ".1.weight").grad;
  j.dbeta = T_(st, L.prefix + ".1.bias").grad;
  return j;
}
static void small_branches_fwd(TrainState* st, const std::vector<int>& which, int M, int Kc, hipStream_t s) {
  const int n = (int)which.size();
  SmallkMulti sk{};
  sk.M = M;
  sk.mean = 1826.6844940968194f;
  sk.stdv = 2516.8905096993817f;
  BnMulti b0{}, b1{};
  b0.M = b1.M = M;
  b0.C = 64;
  b1.C = kTD;
  b0.momentum = b1.momentum = 0.1f;
  b0.sync = b1.sync = st->ctx->sync_fn ? 1 : 0;
  const int slot0 = st->bn_slot, slot1 = st->bn_slot + n;
  st->bn_slot += 2 * n;
  GemmMulti gm{};
  RownormMulti rn{};
  rn.M = M;
  rn.ld = Kc;
  for (int q = 0; q < n; ++q) {
    Branch& br = st->branches[which[q]];
    const MlpLayer &L0 = br.layers[0], &L1 = br.layers[1];
    sk.j[q].x = br.x;
    sk.j[q].K = br.k_in;
    sk.j[q].standardize = br.standardize;
    sk.j[q].w = T_(st, L0.prefix + ".0.weight").data;
    sk.j[q].b = T_(st, L0.prefix + ".0.bias").data;
    sk.j[q].y = L0.y;
    b0.j[q] = bn_job(st, L0, nullptr, slot0 + q);
    gm.j[q] = GemmArgs{L0.a, T_(st, L1.prefix + ".0.weight").data, L1.y, T_(st, L1.prefix + ".0.bias").data, M, kTD, 64, 64, 64, kTD, 0, 0, 64,
                       nullptr, tl_gemm_bf16};
    b1.j[q] = bn_job(st, L1, nullptr, slot1 + q);
    rn.j[q] = RownormJob{L1.a, nullptr, st->cat + br.slot * kTD, nullptr, br.save_n};
  }
  hipLaunchKernelGGL(smallk_fwd_multi_kernel, dim3((M * 64 + 255) / 256, n), dim3(256), 0, s, sk);
  hipLaunchKernelGGL((bn_stats_multi_kernel<0>), dim3(1, (M + kBnRows - 1) / kBnRows, n), dim3(256), 0, s, b0);
  sync_slots(st->ctx, b0.j[0].acc, n, s);
  hipLaunchKernelGGL(bn_apply_fwd_multi_kernel, dim3((unsigned)(((size_t)M * 64 + 255) / 256), n), dim3(256), 0, s, b0);
  hipLaunchKernelGGL((gemm_multi_kernel<true, true>), dim3(kTD / 32, (M + 31) / 32, n), dim3(256), 0, s, gm);
  hipLaunchKernelGGL((bn_stats_multi_kernel<0>), dim3(kTD / 64, (M + kBnRows - 1) / kBnRows, n), dim3(256), 0, s, b1);
  sync_slots(st->ctx, b1.j[0].acc, n, s);
  hipLaunchKernelGGL(bn_apply_fwd_multi_kernel, dim3((unsigned)(((size_t)M * kTD + 255) / 256), n), dim3(256), 0, s, b1);
  hipLaunchKernelGGL(rownorm_fwd_multi_kernel, dim3((M + 3) / 4, n), dim3(256), 0, s, rn);
}
// d2 / d1: [n][M][256] / [n][M][64] scratch (every branch needs its own now)
static void small_branches_bwd(TrainState* st, const std::vector<int>& which, int M, int Kc, const float* dcat, float* d2, float* d1,
                               hipStream_t s) {
  const int n = (int)which.size();
  RownormMulti rn{};
  rn.M = M;
  rn.ld = Kc;
  BnMulti b1{}, b0{};
  b0.M = b1.M = M;
  b0.C = 64;
  b1.C = kTD;
  b0.sync = b1.sync = st->ctx->sync_fn ? 1 : 0;
  const int slot1 = st->bn_slot, slot0 = st->bn_slot + n;
  st->bn_slot += 2 * n;
  GemmPairMulti gp{};
  SmallkMulti sk{};
  sk.M = M;
  sk.mean = 1826.6844940968194f;
  sk.stdv = 2516.8905096993817f;
  sk.rows_per_block = 32;  // (64 rows per workgroup halve the end-of-workgroup atomics but double the serial row loop: 7.8 -> 10.4 us)
  const int N = kTD, Kp = 64;
  const int tiles = (N / 32) * (Kp / 32);
  int ksplit = std::max(1, std::min((M + 255) / 256, (1024 + tiles - 1) / tiles));
  const int kchunk = (((M + ksplit - 1) / ksplit) + 63) & ~63;
  ksplit = (M + kchunk - 1) / kchunk;
  int pair_blocks = 0;
  for (int q = 0; q < n; ++q) {
    const Branch& br = st->branches[which[q]];
    const MlpLayer &L0 = br.layers[0], &L1 = br.layers[1];
    float* dq2 = d2 + (size_t)q * M * kTD;
    float* dq1 = d1 + (size_t)q * M * 64;
    rn.j[q] = RownormJob{dcat + br.slot * kTD, nullptr, dq2, st->cat + br.slot * kTD, br.save_n};
    b1.j[q] = bn_job(st, L1, dq2, slot1 + q);
    GemmPair& p = gp.p[q];
    p.tn = GemmArgs{dq2, L0.a, T_(st, L1.prefix + ".0.weight").grad, nullptr, N, Kp, M, N, Kp, Kp, 0, 1, kchunk,
                    T_(st, L1.prefix + ".0.bias").grad, tl_gemm_bf16};
    p.nn = GemmArgs{dq2, T_(st, L1.prefix + ".0.weight").data, dq1, nullptr, M, Kp, N, N, Kp, Kp, 0, 0, N, nullptr, tl_gemm_bf16};
    p.tn_gx = Kp / 32;
    p.tn_gy = N / 32;
    p.tn_blocks = p.tn_gx * p.tn_gy * ksplit;
    p.nn_gx = Kp / 32;
    pair_blocks = p.tn_blocks + p.nn_gx * ((M + 31) / 32);
    b0.j[q] = bn_job(st, L0, dq1, slot0 + q);
    sk.j[q].x = br.x;
    sk.j[q].K = br.k_in;
    sk.j[q].standardize = br.standardize;
    sk.j[q].dy = dq1;
    sk.j[q].dW = T_(st, L0.prefix + ".0.weight").grad;
    sk.j[q].db = T_(st, L0.prefix + ".0.bias").grad;
  }
  hipLaunchKernelGGL(rownorm_bwd_multi_kernel, dim3((M + 3) / 4, n), dim3(256), 0, s, rn);
  hipLaunchKernelGGL((bn_stats_multi_kernel<1>), dim3(kTD / 64, (M + kBnRows - 1) / kBnRows, n), dim3(256), 0, s, b1);
  sync_slots(st->ctx, b1.j[0].acc, n, s);
  hipLaunchKernelGGL(bn_apply_bwd_multi_kernel, dim3((unsigned)(((size_t)M * kTD + 255) / 256), n), dim3(256), 0, s, b1);
  hipLaunchKernelGGL(gemm_pair_multi_kernel, dim3(pair_blocks, n), dim3(256), 0, s, gp);
  hipLaunchKernelGGL((bn_stats_multi_kernel<1>), dim3(1, (M + kBnRows - 1) / kBnRows, n), dim3(256), 0, s, b0);
  sync_slots(st->ctx, b0.j[0].acc, n, s);
  hipLaunchKernelGGL(bn_apply_bwd_multi_kernel, dim3((unsigned)(((size_t)M * 64 + 255) / 256), n), dim3(256), 0, s, b0);
  hipLaunchKernelGGL(smallk_bwd_multi_kernel, dim3((M + 31) / 32, n), dim3(256), 0, s, sk);
}

int train_forward_impl(t2l_ctx* ctx, const t2l_packed_cells* in, float p, uint32_t seed, float* out_emb, hipStream_t s) {
  TrainState* st = state(ctx);
  if (!st) return fail(ctx, T2L_ESTATE, "t2l_encode_cells_train: call t2l_train_bind first");
  if (!in || !out_emb || in->n_cells <= 0 || in->n_objects <= 1 || !in->offsets)
    return fail(ctx, T2L_EINVAL, "t2l_encode_cells_train: bad arguments (BatchNorm batch statistics need >= 2 objects)");
  if (!(p >= 0.f && p < 1.f)) return fail(ctx, T2L_EINVAL, "t2l_encode_cells_train: dropout_p must be in [0,1)");
  const t2l_model_config& c = st->cfg;
  if ((c.use_class && (c.class_embed ? !in->class_idx : !in->pn_feat)) || (c.use_color && (c.color_embed ? !in->color_idx : !in->rgb)) ||
      (c.use_position && !in->center) || (c.use_num && !in->n_pts))
    return fail(ctx, T2L_EINVAL, "t2l_encode_cells_train: a packed input the configuration needs is NULL");
  const int M = in->n_objects, B = in->n_cells, T = B * kTS, Kc = st->n_feat * kTD;
  if ((uint64_t)T * 2 * kTD >= (1ull << 32)) return fail(ctx, T2L_EINVAL, "t2l_encode_cells_train: batch too large for the dropout counters");
  // workspace: generous closed-form bound, grown on demand
  const size_t need_bytes =
      sizeof(float) * ((size_t)M * (2 * Kc + 30 * kTD) + (size_t)T * kTD * 18 +
                       (size_t)c.num_layers * ((size_t)T * (13 * kTD + 8) + (size_t)B * kTH * kTS * kTS) + (size_t)B * kTD * 4) +
      (1 << 20);
  if (need_bytes > st->ws_cap) {
    if (st->ws) (void)hipFree(st->ws);
    st->ws = nullptr;
    st->ws_cap = 0;
    T2L_HIP(ctx, hipMalloc(&st->ws, need_bytes));
    st->ws_cap = need_bytes;
  }
  tl_gemm_bf16 = ctx->train_bf16;
  tl_gemm_block64 = ctx->train_gemm_block == 64 || (ctx->train_gemm_block == 0 && ctx->train_bf16 != 0);
  st->ws_off = 0;
  st->have_forward = false;
  st->M = M; st->B = B; st->T = T;
  st->offsets = in->offsets;
  st->seed = seed;
  st->p = p;
  st->branches.clear();
  st->layers.clear();
  event_begin(ctx, "train_forward", s);
  st->bn_slot = 0;
  // the accumulators are cleared by the previous forward's last launch (pool_norm_fwd_kernel); a memset only the first time or after
  // a forward that did not get that far
  ctx->sync_failed = false;
  if (!st->fwd_acc_clean) T2L_HIP(ctx, hipMemsetAsync(acc_base(ctx, st), 0, sizeof(double) * kBnStride * kBnSlots, s));
  st->fwd_acc_clean = false;

  st->cat = bump<float>(st, (size_t)M * Kc);
  const std::string oe = "object_encoder.";
  int slot = 0;
  std::vector<int> smalls;  // indices of the small branches: launched together below, one launch per stage
  auto small_branch = [&](const std::string& name, const float* x, int k, int standardize) {
    Branch br;
    br.kind = 1; br.slot = slot++; br.x = x; br.k_in = k; br.standardize = standardize;
    br.layers.push_back(make_layer(st, oe + name + ".0", k, 64, M));
    br.layers.push_back(make_layer(st, oe + name + ".1", 64, kTD, M));
    br.save_n = bump<float>(st, M);
    smalls.push_back((int)st->branches.size());
    st->branches.push_back(br);
  };
  std::vector<int> embeds;  // the embedding lookups: one launch for both tables
  auto embed_branch = [&](const std::string& table, const int32_t* idx) {
    Branch br;
    br.kind = 0; br.slot = slot++; br.table = oe + table; br.idx = idx;
    br.save_n = bump<float>(st, M);
    embeds.push_back((int)st->branches.size());
    st->branches.push_back(br);
  };
  if (c.use_class) {
    if (c.class_embed) {
      embed_branch("class_embedding.weight", in->class_idx);
    } else {  // PointNet++ features2 -> mlp_pointnet (object_encoder.py:86-99,112)
      Branch br;
      br.kind = 2; br.slot = slot++; br.x = in->pn_feat; br.k_in = 256;
      br.layers.push_back(make_layer(st, oe + "mlp_pointnet.0", 256, kTD, M));
      br.save_n = bump<float>(st, M);
      mlp_layer_fwd(st, br.layers[0], in->pn_feat, M, 0, 0, s);
      hipLaunchKernelGGL(rownorm_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, s, br.layers[0].a, (const int32_t*)nullptr, M,
                         st->cat + br.slot * kTD, Kc, br.save_n);
      st->branches.push_back(br);
    }
  }
  if (c.use_color) {
    if (c.color_embed) embed_branch("color_embedding.weight", in->color_idx);
    else small_branch("color_encoder", in->rgb, 3, 0);
  }
  if (c.use_position) small_branch("pos_encoder", in->center, 3, 0);
  if (c.use_num) small_branch("num_encoder", in->n_pts, 1, 1);
  if (!embeds.empty()) {
    RownormMulti rn{};
    rn.M = M;
    rn.ld = Kc;
    for (size_t q = 0; q < embeds.size(); ++q) {
      const Branch& br = st->branches[embeds[q]];
      rn.j[q] = RownormJob{T_(st, br.table).data, br.idx, st->cat + br.slot * kTD, nullptr, br.save_n};
    }
    hipLaunchKernelGGL(rownorm_fwd_multi_kernel, dim3((M + 3) / 4, (unsigned)embeds.size()), dim3(256), 0, s, rn);
  }
  if (!smalls.empty()) small_branches_fwd(st, smalls, M, Kc, s);

  st->merge = make_layer(st, oe + "mlp_merge.0", Kc, kTD, M);
  mlp_layer_fwd(st, st->merge, st->cat, M, 0, 0, s);
  st->X0 = bump<float>(st, (size_t)T * kTD);
  st->save_nf = bump<float>(st, M);
  hipLaunchKernelGGL(scatter_norm_fwd_kernel, dim3((T + 3) / 4), dim3(256), 0, s, st->merge.a, in->offsets, B, st->X0, st->save_nf);

  float* tmp = bump<float>(st, (size_t)T * kTD);
  const float* x = st->X0;
  for (int l = 0; l < c.num_layers; ++l) {
    LayerSave L;
    L.prefix = "obj_inter_module." + std::to_string(l);
    L.x_in = x;
    L.qkv = bump<float>(st, (size_t)T * 3 * kTD);
    L.P = bump<float>(st, (size_t)B * kTH * kTS * kTS);
    L.O = bump<float>(st, (size_t)T * kTD);
    L.xhat1 = bump<float>(st, (size_t)T * kTD);
    L.rstd1 = bump<float>(st, T);
    L.x1 = bump<float>(st, (size_t)T * kTD);
    L.h = bump<float>(st, (size_t)T * 2 * kTD);
    L.hd = p > 0.f ? bump<float>(st, (size_t)T * 2 * kTD) : L.h;
    L.xhat2 = bump<float>(st, (size_t)T * kTD);
    L.rstd2 = bump<float>(st, T);
    L.x2 = bump<float>(st, (size_t)T * kTD);
    gemm_nt(x, T_(st, L.prefix + ".self_attn.in_proj_weight").data, T_(st, L.prefix + ".self_attn.in_proj_bias").data, L.qkv, T, 3 * kTD, kTD, 0, s);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(B * kTH), dim3(256), 0, s, L.qkv, L.P, L.O, make_drop(seed, l * 4 + 0, p));
    gemm_nt(L.O, T_(st, L.prefix + ".self_attn.out_proj.weight").data, T_(st, L.prefix + ".self_attn.out_proj.bias").data, tmp, T, kTD, kTD, 0, s);
    hipLaunchKernelGGL(ln_fwd_kernel, dim3((T + 3) / 4), dim3(256), 0, s, x, tmp, T, T_(st, L.prefix + ".norm1.weight").data,
                       T_(st, L.prefix + ".norm1.bias").data, make_drop(seed, l * 4 + 1, p), L.x1, L.xhat1, L.rstd1);
    {  // linear1 + ReLU, and the dropout behind it in the same epilogue (h is kept for backward, hd feeds linear2)
      GemmArgs g{L.x1, T_(st, L.prefix + ".linear1.weight").data, L.h, T_(st, L.prefix + ".linear1.bias").data, T, 2 * kTD, kTD, kTD, kTD, 2 * kTD,
                 1, 0, kTD, nullptr, tl_gemm_bf16};
      if (p > 0.f) {
        const Drop dr = make_drop(seed, l * 4 + 2, p);
        g.epi = 1;
        g.C2 = L.hd;
        g.drop_key = dr.key;
        g.drop_thr = dr.thr;
        g.drop_scale = dr.scale;
      }
      gemm_nt_args(g, s);
    }
    gemm_nt(L.hd, T_(st, L.prefix + ".linear2.weight").data, T_(st, L.prefix + ".linear2.bias").data, tmp, T, kTD, 2 * kTD, 0, s);
    hipLaunchKernelGGL(ln_fwd_kernel, dim3((T + 3) / 4), dim3(256), 0, s, L.x1, tmp, T, T_(st, L.prefix + ".norm2.weight").data,
                       T_(st, L.prefix + ".norm2.bias").data, make_drop(seed, l * 4 + 3, p), L.x2, L.xhat2, L.rstd2);
    x = L.x2;
    st->layers.push_back(L);
  }
  st->out = bump<float>(st, (size_t)B * kTD);
  st->pool_n = bump<float>(st, B);
  st->pool_arg = bump<int32_t>(st, (size_t)B * kTD);
  hipLaunchKernelGGL(pool_norm_fwd_kernel, dim3(B), dim3(256), 0, s, x, st->out, st->pool_arg, st->pool_n, out_emb, acc_base(ctx, st),
                     kBnStride * kBnSlots * 2);
  event_end(ctx, "train_forward", s);
  T2L_HIP(ctx, hipGetLastError());
  st->fwd_acc_clean = true;
  st->bwd_acc_clean = true;
  if (st->ws_off > st->ws_cap) return fail(ctx, T2L_ENOMEM, "t2l_encode_cells_train: workspace bound exceeded (internal error)");
  if (ctx->sync_failed) return fail(ctx, T2L_ESTATE, "t2l_encode_cells_train: the cross-rank sum callback (t2l_train_sync_bn) failed");
  st->have_forward = true;
  return T2L_OK;
}

// ---- backward --------------------------------------------------------------------------------------------------
// d: gradient w.r.t. the block's ReLU output [M,cout] (overwritten); x: the block's input; dx (optional) receives d W.
static void mlp_layer_bwd(TrainState* st, const MlpLayer& L, float* d, const float* x, int M, int small_k, int standardize, float* dx,
                          hipStream_t s) {
  const int sync = st->ctx->sync_fn ? 1 : 0;
  double* acc = acc_base(st->ctx, st) + (size_t)(st->bn_slot++ % (2 * kBnSlots)) * kBnStride;
  hipLaunchKernelGGL((bn_stats_kernel<1>), dim3(L.cout / 64, (M + kBnRows - 1) / kBnRows), dim3(256), 0, s, L.y, (const float*)d, (const float*)L.a,
                     M, L.cout, (const float*)L.mean, (const float*)L.rstd, acc, sync, T_(st, L.prefix + ".1.weight").grad,
                     T_(st, L.prefix + ".1.bias").grad);
  sync_slots(st->ctx, acc, 1, s);
  hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3((unsigned)(((size_t)M * L.cout + 255) / 256)), dim3(256), 0, s, d, L.a, L.y, M, L.cout, acc,
                     T_(st, L.prefix + ".1.weight").data, L.mean, L.rstd, T_(st, L.prefix + ".1.weight").grad, T_(st, L.prefix + ".1.bias").grad,
                     sync);
  if (small_k) {
    const int rows = 32;
    hipLaunchKernelGGL(smallk_bwd_kernel, dim3((M + rows - 1) / rows), dim3(256), 0, s, x, M, small_k, d, standardize, 1826.6844940968194f,
                       2516.8905096993817f, rows, T_(st, L.prefix + ".0.weight").grad, T_(st, L.prefix + ".0.bias").grad);
  } else {
    if (dx)
      gemm_tn_nn(d, x, T_(st, L.prefix + ".0.weight").grad, T_(st, L.prefix + ".0.bias").grad, T_(st, L.prefix + ".0.weight").data, dx, M, L.cout,
                 L.cin, 0, nullptr, nullptr, s);
    else
      gemm_tn(d, x, T_(st, L.prefix + ".0.weight").grad, T_(st, L.prefix + ".0.bias").grad, M, L.cout, L.cin, s);
  }
}

int train_backward_impl(t2l_ctx* ctx, const float* grad_emb, float* grad_pn_feat, hipStream_t s) {
  TrainState* st = state(ctx);
  if (!st || !st->have_forward) return fail(ctx, T2L_ESTATE, "t2l_encode_cells_backward: no forward pass to differentiate");
  if (!grad_emb) return fail(ctx, T2L_EINVAL, "t2l_encode_cells_backward: null gradient");
  const int M = st->M, B = st->B, T = st->T, Kc = st->n_feat * kTD;
  tl_gemm_bf16 = ctx->train_bf16;
  tl_gemm_block64 = ctx->train_gemm_block == 64 || (ctx->train_gemm_block == 0 && ctx->train_bf16 != 0);
  const size_t mark = st->ws_off;
  event_begin(ctx, "train_backward", s);
  st->bn_slot = kBnSlots;  // the backward's half of the accumulators: zeroed by the forward's memset, unless this is a second backward
  ctx->sync_failed = false;
  if (!st->bwd_acc_clean)
    T2L_HIP(ctx, hipMemsetAsync(acc_base(ctx, st) + (size_t)kBnSlots * kBnStride, 0, sizeof(double) * kBnStride * kBnSlots, s));
  st->bwd_acc_clean = false;
  float* dcur = bump<float>(st, (size_t)T * kTD);
  float* dA = bump<float>(st, (size_t)T * kTD);
  float* dB = bump<float>(st, (size_t)T * kTD);
  float* dC = bump<float>(st, (size_t)T * kTD);
  float* dB2 = bump<float>(st, (size_t)T * kTD);
  float* dO = bump<float>(st, (size_t)T * kTD);
  float* dH = bump<float>(st, (size_t)T * 2 * kTD);
  float* dqkv = bump<float>(st, (size_t)T * 3 * kTD);
  float* dfeat = bump<float>(st, (size_t)M * kTD);
  float* dcat = bump<float>(st, (size_t)M * Kc);
  float* d2 = bump<float>(st, (size_t)kMaxJobs * M * kTD);  // scratch of the feature branches: one region per small branch (they run
  float* d1 = bump<float>(st, (size_t)kMaxJobs * M * 64);   // together, stage by stage); the others use region 0 one after the other
  if (st->ws_off > st->ws_cap) {
    st->ws_off = mark;
    return fail(ctx, T2L_ENOMEM, "t2l_encode_cells_backward: workspace bound exceeded (internal error)");
  }
  const float* xl = st->layers.empty() ? st->X0 : st->layers.back().x2;
  (void)xl;
  hipLaunchKernelGGL(pool_norm_bwd_kernel, dim3(B), dim3(256), 0, s, grad_emb, st->out, st->pool_arg, st->pool_n, dcur);
  const int ln_grid = std::min(32, (T + 15) / 16);  // few workgroups (each ends with 512 float atomics on the same addresses) of 16 waves
  for (int l = (int)st->layers.size() - 1; l >= 0; --l) {
    const LayerSave& L = st->layers[l];
    auto W = [&](const char* n) -> const TTensor& { return T_(st, L.prefix + n); };
    // norm2 + dropout2
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(ln_grid), dim3(1024), 0, s, dcur, L.xhat2, L.rstd2, T, W(".norm2.weight").data,
                       make_drop(st->seed, l * 4 + 3, st->p), dA, dB, W(".norm2.weight").grad, W(".norm2.bias").grad);
    // linear2
    {  // dW2 += dB^T hd, and dH = (dB W2) through the ReLU + dropout backward — one launch
      const Drop dr = make_drop(st->seed, l * 4 + 2, st->p);
      gemm_tn_nn(dB, L.hd, W(".linear2.weight").grad, W(".linear2.bias").grad, W(".linear2.weight").data, dH, T, kTD, 2 * kTD, 0, L.h, &dr, s);
    }
    // linear1; dA (= dz2, the residual path) += dH W1
    gemm_tn_nn(dH, L.x1, W(".linear1.weight").grad, W(".linear1.bias").grad, W(".linear1.weight").data, dA, T, 2 * kTD, kTD, 1, nullptr, nullptr, s);
    // norm1 + dropout1
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(ln_grid), dim3(1024), 0, s, dA, L.xhat1, L.rstd1, T, W(".norm1.weight").data,
                       make_drop(st->seed, l * 4 + 1, st->p), dC, dB2, W(".norm1.weight").grad, W(".norm1.bias").grad);
    // out_proj
    gemm_tn_nn(dB2, L.O, W(".self_attn.out_proj.weight").grad, W(".self_attn.out_proj.bias").grad, W(".self_attn.out_proj.weight").data, dO, T, kTD,
               kTD, 0, nullptr, nullptr, s);
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(B * kTH), dim3(256), 0, s, L.qkv, L.P, dO, dqkv, make_drop(st->seed, l * 4 + 0, st->p));
    // in_proj; dC (= dz1, the residual path) += dqkv Win
    gemm_tn_nn(dqkv, L.x_in, W(".self_attn.in_proj_weight").grad, W(".self_attn.in_proj_bias").grad, W(".self_attn.in_proj_weight").data, dC, T,
               3 * kTD, kTD, 1, nullptr, nullptr, s);
    std::swap(dcur, dC);
  }
  // tokens -> objects, merge MLP
  hipLaunchKernelGGL(scatter_norm_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, s, dcur, st->X0, st->save_nf, st->offsets, B, M, dfeat);
  mlp_layer_bwd(st, st->merge, dfeat, st->cat, M, 0, 0, dcat, s);
  {  // embedding tables: normalize-backward of both slots in one launch, then the per-row sums of both tables in one
    RownormMulti rn{};
    rn.M = M;
    rn.ld = Kc;
    EmbedSumMulti es{};
    es.M = M;
    int ne = 0, max_rows = 0;
    for (const Branch& br : st->branches) {
      if (br.kind != 0) continue;
      float* dq = d2 + (size_t)ne * M * kTD;
      rn.j[ne] = RownormJob{dcat + br.slot * kTD, nullptr, dq, st->cat + br.slot * kTD, br.save_n};
      es.g[ne] = dq;
      es.idx[ne] = br.idx;
      es.dtable[ne] = T_(st, br.table).grad;
      es.rows[ne] = (int)(T_(st, br.table).numel / kTD);
      max_rows = std::max(max_rows, es.rows[ne]);
      ++ne;
    }
    if (ne) {
      hipLaunchKernelGGL(rownorm_bwd_multi_kernel, dim3((M + 3) / 4, ne), dim3(256), 0, s, rn);
      if (max_rows > 1) hipLaunchKernelGGL(embed_sum_multi_kernel, dim3(max_rows - 1, kEmbSplit, ne), dim3(256), 0, s, es);
    }
  }
  for (const Branch& br : st->branches) {
    if (br.kind != 2) continue;  // the PointNet++-feature branch: one [256 -> 256] block
    hipLaunchKernelGGL(rownorm_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, s, dcat + br.slot * kTD, st->cat + br.slot * kTD, Kc, br.save_n, M,
                       d2);
    mlp_layer_bwd(st, br.layers[0], d2, br.x, M, 0, 0, grad_pn_feat, s);
  }
  {
    std::vector<int> smalls;
    for (size_t i = 0; i < st->branches.size(); ++i)
      if (st->branches[i].kind == 1) smalls.push_back((int)i);
    if (!smalls.empty()) small_branches_bwd(st, smalls, M, Kc, dcat, d2, d1, s);
  }
  event_end(ctx, "train_backward", s);
  st->ws_off = mark;
  T2L_HIP(ctx, hipGetLastError());
  if (ctx->sync_failed) return fail(ctx, T2L_ESTATE, "t2l_encode_cells_backward: the cross-rank sum callback (t2l_train_sync_bn) failed");
  return T2L_OK;
}

static void adam_launch(TrainState* st, int chunk0, int chunk1, int64_t step, float lr, float b1, float b2, float eps, hipStream_t s) {
  if (chunk1 <= chunk0) return;
  const float bc1 = (float)(1.0 - pow((double)b1, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow((double)b2, (double)step));
  hipLaunchKernelGGL(adam_kernel, dim3(chunk1 - chunk0), dim3(256), 0, s, st->d_tensors, st->d_chunks + chunk0, lr, b1, b2, eps, bc1, bc2s);
}

int adam_step_impl(t2l_ctx* ctx, float lr, float b1, float b2, float eps, hipStream_t s) {
  TrainState* st = state(ctx);
  if (!st) return fail(ctx, T2L_ESTATE, "t2l_adam_step: call t2l_train_bind first");
  st->step += 1;
  // the backbone's tensors step only when its backward added into their gradients since the last zero_grad (a batch that
  // passed precomputed features2 leaves them untouched: torch.optim.Adam would skip .grad = None parameters, not decay them)
  const bool pn = st->pn_chunk0 < st->n_chunks && st->pn_touched;
  if (pn) st->step_pn += 1;
  event_begin(ctx, "adam_step", s);
  if (pn && st->step_pn == st->step) {
    adam_launch(st, 0, st->n_chunks, st->step, lr, b1, b2, eps, s);
  } else {
    adam_launch(st, 0, st->pn_chunk0, st->step, lr, b1, b2, eps, s);
    if (pn) adam_launch(st, st->pn_chunk0, st->n_chunks, st->step_pn, lr, b1, b2, eps, s);
  }
  event_end(ctx, "adam_step", s);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

int adam_state_impl(t2l_ctx* ctx, int set, float* m, float* v, int64_t* step, int64_t* numel, hipStream_t s) {
  TrainState* st = state(ctx);
  if (!st) return fail(ctx, T2L_ESTATE, "t2l_adam_state: call t2l_train_bind first");
  if (numel) *numel = st->mv_total;
  if (!m && !v) {  // size / step query
    if (step && !set) *step = st->step | (st->step_pn << 32);
    return T2L_OK;
  }
  if (!m || !v || !step) return fail(ctx, T2L_EINVAL, "t2l_adam_state: pass m, v and step together");
  const size_t bytes = sizeof(float) * (size_t)st->mv_total;
  if (set) {
    T2L_HIP(ctx, hipMemcpyAsync(st->mv, m, bytes, hipMemcpyDeviceToDevice, s));
    T2L_HIP(ctx, hipMemcpyAsync(st->mv + st->mv_total, v, bytes, hipMemcpyDeviceToDevice, s));
    st->step = *step & 0xFFFFFFFFll;
    st->step_pn = st->pn_chunk0 < st->n_chunks ? (*step >> 32) : 0;
  } else {
    T2L_HIP(ctx, hipMemcpyAsync(m, st->mv, bytes, hipMemcpyDeviceToDevice, s));
    T2L_HIP(ctx, hipMemcpyAsync(v, st->mv + st->mv_total, bytes, hipMemcpyDeviceToDevice, s));
    *step = st->step | (st->step_pn << 32);
  }
  return T2L_OK;
}

int zero_grad_impl(t2l_ctx* ctx, hipStream_t s) {
  TrainState* st = state(ctx);
  if (!st) return fail(ctx, T2L_ESTATE, "t2l_zero_grad: call t2l_train_bind first");
  hipLaunchKernelGGL(zero_kernel, dim3(st->n_chunks), dim3(256), 0, s, st->d_tensors, st->d_chunks);
  st->pn_touched = false;
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

// =================================================================================================================
// The TEXT head in training mode (SURVEY.md 8 f-4 / a9's other half): LanguageEncoder.forward downstream of the frozen T5's
// hidden states under model.train() (models/language_encoder.py:127-147 as run by training/coarse.py:44,55-56 — the published
// command trains this head: --fixed_embedding freezes T5 only, README.md:87-99):
//   hidden [n_sent, L, 1024] -> TransformerEncoderLayer(1024, 4 heads, ff 4096; the four dropout sites live) over the L tokens ->
//   max over tokens -> Linear(1024 -> 256) + BatchNorm1d with the statistics of THIS batch of sentences (running buffers
//   updated) -> view [n_desc, S, 256] -> x += TransformerEncoderLayer(256, 4 heads, ff 1024)(x) over the S sentences -> max
//   over the sentences -> out [n_desc, 256]  (F.normalize stays with the caller: cell_retrieval.py:57-63).
// Forward keeps the activations; backward accumulates (+=) into the bound .grad buffers — the parameters stay torch's and are
// stepped by t2l_text_adam_step (one launch over the 13.6 M head parameters; torch.optim.Adam when the caller prefers). The same
// modular f32 kernels as the object branch (gemm_f32.h products, option train_bf16 for bf16 / split-bf16 operands), with the
// attention / LayerNorm kernels in their generic forms (train_kernels.h).
// =================================================================================================================
struct TextLayer {
  std::string prefix;
  int T = 0, B = 0, S = 0, site0 = 0;
  const float* x_in = nullptr;
  float *qkv = nullptr, *P = nullptr, *O = nullptr, *xhat1 = nullptr, *rstd1 = nullptr, *x1 = nullptr, *h = nullptr, *hd = nullptr,
        *xhat2 = nullptr, *rstd2 = nullptr, *x2 = nullptr;
};
struct TextTrain {
  t2l_ctx* ctx = nullptr;
  std::unordered_map<std::string, TTensor> t;
  std::string prefix;
  char* ws = nullptr;
  size_t ws_cap = 0, ws_off = 0;
  bool have_forward = false;
  int n_sent = 0, L = 0, n_desc = 0, S = 0;
  float p = 0.f;
  uint32_t seed = 0;
  TextLayer intra, inter;
  float *pooled = nullptr, *mlp_y = nullptr, *mlp_out = nullptr, *bn_mean = nullptr, *bn_rstd = nullptr, *out = nullptr;
  int32_t *tok_arg = nullptr, *sent_arg = nullptr;
  // Adam over the head's parameters (t2l_text_adam_step): moments inside the library, one launch through the chunk table
  std::vector<std::string> adam_names;  // bind order
  std::vector<int64_t> adam_numel;
  AdamTensor* d_tensors = nullptr;
  AdamChunk* d_chunks = nullptr;
  float* mv = nullptr;  // [2][mv_total]
  int64_t mv_total = 0, step = 0;
  int n_chunks = 0;
};
static TextTrain* tstate(t2l_ctx* ctx) { return reinterpret_cast<TextTrain*>(ctx->text_train); }
void free_text_train(t2l_ctx* ctx) {
  TextTrain* st = tstate(ctx);
  if (!st) return;
  for (void* p : {(void*)st->ws, (void*)st->d_tensors, (void*)st->d_chunks, (void*)st->mv})
    if (p) (void)hipFree(p);
  delete st;
  ctx->text_train = nullptr;
}
template <typename T>
static T* tbump(TextTrain* st, size_t count) {
  const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
  T* p = reinterpret_cast<T*>(st->ws + st->ws_off);
  st->ws_off += bytes;
  return p;
}
static const TTensor& TT(TextTrain* st, const std::string& n) { return st->t.at(st->prefix + n); }

__global__ void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] += b[i];
}

int text_train_bind_impl(t2l_ctx* ctx, const t2l_train_tensor* tensors, int n, const char* prefix) {
  if (!tensors || n <= 0) return fail(ctx, T2L_EINVAL, "t2l_text_train_bind: null argument");
  // a re-bind of the same parameter list (moved storage: model.to(), re-assigned .grad) keeps the optimizer state, as t2l_train_bind does
  std::vector<std::string> old_names;
  std::vector<int64_t> old_numel;
  float* old_mv = nullptr;
  int64_t old_total = 0, old_step = 0;
  if (TextTrain* o = tstate(ctx)) {
    if (ctx->train_keep_adam && o->mv) {
      old_names = o->adam_names;
      old_numel = o->adam_numel;
      old_mv = o->mv;
      old_total = o->mv_total;
      old_step = o->step;
      o->mv = nullptr;  // (free_text_train below must not free it)
    }
  }
  free_text_train(ctx);
  TextTrain* st = new TextTrain();
  ctx->text_train = st;
  st->ctx = ctx;
  st->prefix = prefix ? prefix : "language_encoder.";
  for (int i = 0; i < n; ++i) {
    if (!tensors[i].name || !tensors[i].data) return fail(ctx, T2L_EINVAL, "t2l_text_train_bind: null name/data");
    st->t[tensors[i].name] = TTensor{tensors[i].data, tensors[i].grad, tensors[i].numel};
  }
  auto need_t = [&](const std::string& name, int64_t numel, bool grad) -> int {
    auto it = st->t.find(st->prefix + name);
    if (it == st->t.end() || it->second.numel != numel || (grad && !it->second.grad))
      return fail(ctx, T2L_EINVAL, "t2l_text_train_bind: tensor '" + st->prefix + name + "' missing, mis-sized or without a gradient buffer "
                                   "(the engine trains the published head: intra_module 1 x (1024, 4 heads, 4096), inter_mlp 1024 -> 256, "
                                   "inter_module 1 x (256, 4 heads, 1024))");
    return T2L_OK;
  };
  int rc;
  auto layer = [&](const std::string& lp, int64_t D, int64_t FF) -> int {
    const std::pair<const char*, int64_t> req[] = {{".self_attn.in_proj_weight", 3 * D * D}, {".self_attn.in_proj_bias", 3 * D},
                                                   {".self_attn.out_proj.weight", D * D},    {".self_attn.out_proj.bias", D},
                                                   {".linear1.weight", FF * D},              {".linear1.bias", FF},
                                                   {".linear2.weight", D * FF},              {".linear2.bias", D},
                                                   {".norm1.weight", D},                     {".norm1.bias", D},
                                                   {".norm2.weight", D},                     {".norm2.bias", D}};
    for (auto& r : req)
      if ((rc = need_t(lp + r.first, r.second, true))) return rc;
    return T2L_OK;
  };
  if ((rc = layer("intra_module.0", 1024, 4096)) || (rc = layer("inter_module.0", 256, 1024))) return rc;
  if ((rc = need_t("inter_mlp.0.0.weight", 256 * 1024, true)) || (rc = need_t("inter_mlp.0.0.bias", 256, true)) ||
      (rc = need_t("inter_mlp.0.1.weight", 256, true)) || (rc = need_t("inter_mlp.0.1.bias", 256, true)) ||
      (rc = need_t("inter_mlp.0.1.running_mean", 256, false)) || (rc = need_t("inter_mlp.0.1.running_var", 256, false)))
    return rc;
  {  // Adam tables over every bound tensor that has a gradient buffer, in bind order; moments zero-initialised
    std::vector<AdamTensor> ts;
    std::vector<AdamChunk> cs;
    for (int i = 0; i < n; ++i)
      if (tensors[i].grad && st->t.count(tensors[i].name)) {
        st->adam_names.push_back(tensors[i].name);
        st->adam_numel.push_back(tensors[i].numel);
        st->mv_total += tensors[i].numel;
      }
    const bool same = old_mv && old_names == st->adam_names && old_numel == st->adam_numel && old_total == st->mv_total;
    if (same) {
      st->mv = old_mv;
      st->step = old_step;
    } else {
      if (old_mv) (void)hipFree(old_mv);
      T2L_HIP(ctx, hipMalloc(&st->mv, sizeof(float) * 2 * (size_t)st->mv_total));
      T2L_HIP(ctx, hipMemset(st->mv, 0, sizeof(float) * 2 * (size_t)st->mv_total));
    }
    int64_t off = 0;
    for (size_t i = 0; i < st->adam_names.size(); ++i) {
      const TTensor& t = st->t[st->adam_names[i]];
      ts.push_back(AdamTensor{t.data, t.grad, st->mv + off, st->mv + st->mv_total + off, t.numel});
      for (int64_t c = 0; c * 1024 < t.numel; ++c) cs.push_back(AdamChunk{(int32_t)i, (int32_t)c});
      off += t.numel;
    }
    st->n_chunks = (int)cs.size();
    T2L_HIP(ctx, hipMalloc(&st->d_tensors, sizeof(AdamTensor) * ts.size()));
    T2L_HIP(ctx, hipMalloc(&st->d_chunks, sizeof(AdamChunk) * cs.size()));
    T2L_HIP(ctx, hipMemcpy(st->d_tensors, ts.data(), sizeof(AdamTensor) * ts.size(), hipMemcpyHostToDevice));
    T2L_HIP(ctx, hipMemcpy(st->d_chunks, cs.data(), sizeof(AdamChunk) * cs.size(), hipMemcpyHostToDevice));
  }
  static PerDeviceOnce once;
  if (once.need(ctx->device)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_g_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_g_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    once.mark(ctx->device);
  }
  return T2L_OK;
}

// The head's Linear products run on the tiled LDS-ring GEMM of text_head.hip (fast_gemm: 256 x 256 tiles, bf16 planes, split-bf16 by
// default — 466 GFLOP per step at B = 64 are GEMM-shaped work that the object branch's tile-per-workgroup products, built for
// 1,792-row operands, serve at a third of its rate); shapes it does not take (fewer than 64 rows) keep the products of gemm_f32.h
// with the same operand arithmetic. (An f32-MFMA operand option existed until round 5: 7.6 ms per step against PyTorch's 5.6; removed.)
static bool t_fast(TextTrain* st, int M, int N, int K) {
  return N % 256 == 0 && K % 32 == 0 && K % 4 == 0 && M >= 64;
}
static void t_gemm_nt(TextTrain* st, const float* X, const float* W, const float* b, float* Y, int M, int N, int K, int relu, hipStream_t s) {
  if (t_fast(st, M, N, K)) (void)fast_gemm(st->ctx, X, false, W, false, b, Y, M, N, K, relu, 0, st->ctx->text_train_bf16 == 1, s);
  else gemm_nt(X, W, b, Y, M, N, K, relu, s);
}
// dW[N,Kp] += dY^T X, db[N] += column sums of dY, and (dX != nullptr) dX[M,Kp] (+)= dY W, through the ReLU + dropout backward of the
// layer that produced X's pre-image when mask_src is given
static void t_gemm_tn_nn(TextTrain* st, const float* dY, const float* X, float* dW, float* db, const float* W, float* dX, int M, int N, int Kp,
                         int accumulate, const float* mask_src, const Drop* drop, hipStream_t s) {
  if (t_fast(st, N, Kp, M) && (!dX || t_fast(st, M, Kp, N)) && N % 4 == 0) {
    const bool single = st->ctx->text_train_bf16 == 1;
    (void)fast_gemm(st->ctx, dY, true, X, true, nullptr, dW, N, Kp, M, 0, 1, single, s, db);  // (db: column sums of dY, in its split pass)
    if (dX) {
      (void)fast_gemm(st->ctx, dY, false, W, true, nullptr, dX, M, Kp, N, 0, accumulate, single, s);
      if (mask_src) {
        const size_t n = (size_t)M * Kp;
        hipLaunchKernelGGL(relu_drop_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dX, mask_src, n, *drop);
      }
    }
    return;
  }
  if (dX) gemm_tn_nn(dY, X, dW, db, W, dX, M, N, Kp, accumulate, mask_src, drop, s);
  else gemm_tn(dY, X, dW, db, M, N, Kp, s);
}

static size_t attn_lds(int S, int HD, bool bwd) { return sizeof(float) * ((size_t)(bwd ? 4 : 3) * S * (HD + 1) + (size_t)(bwd ? 3 : 1) * S * (S + 1)); }

template <int D>
static void text_layer_alloc(TextTrain* st, TextLayer& L, float p) {
  const size_t T = (size_t)L.T;
  L.qkv = tbump<float>(st, T * 3 * D);
  L.P = tbump<float>(st, (size_t)L.B * 4 * L.S * L.S);
  L.O = tbump<float>(st, T * D);
  L.xhat1 = tbump<float>(st, T * D);
  L.rstd1 = tbump<float>(st, T);
  L.x1 = tbump<float>(st, T * D);
  L.h = tbump<float>(st, T * 4 * D);
  L.hd = p > 0.f ? tbump<float>(st, T * 4 * D) : L.h;
  L.xhat2 = tbump<float>(st, T * D);
  L.rstd2 = tbump<float>(st, T);
  L.x2 = tbump<float>(st, T * D);
}
template <int D>
static void text_layer_fwd(TextTrain* st, TextLayer& L, float* tmp, hipStream_t s) {
  constexpr int HD = D / 4, FF = 4 * D;
  const int T = L.T;
  auto W = [&](const char* n) -> const TTensor& { return TT(st, L.prefix + n); };
  t_gemm_nt(st, L.x_in, W(".self_attn.in_proj_weight").data, W(".self_attn.in_proj_bias").data, L.qkv, T, 3 * D, D, 0, s);
  hipLaunchKernelGGL((attn_fwd_g_kernel<HD>), dim3(L.B * 4), dim3(256), attn_lds(L.S, HD, false), s, L.qkv, L.P, L.O, L.S,
                     make_drop(st->seed, L.site0 + 0, st->p));
  t_gemm_nt(st, L.O, W(".self_attn.out_proj.weight").data, W(".self_attn.out_proj.bias").data, tmp, T, D, D, 0, s);
  hipLaunchKernelGGL((ln_fwd_g_kernel<D>), dim3((T + 3) / 4), dim3(256), 0, s, L.x_in, (const float*)tmp, T, W(".norm1.weight").data,
                     W(".norm1.bias").data, make_drop(st->seed, L.site0 + 1, st->p), L.x1, L.xhat1, L.rstd1);
  if (t_fast(st, T, FF, D)) {  // linear1 + ReLU in the GEMM's epilogue (h is kept for backward); the dropout behind it as one pass
    (void)fast_gemm(st->ctx, L.x1, false, W(".linear1.weight").data, false, W(".linear1.bias").data, L.h, T, FF, D, 1, 0, st->ctx->text_train_bf16 == 1, s);
    if (st->p > 0.f) {
      const size_t n = (size_t)T * FF;
      hipLaunchKernelGGL(drop_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)L.h, n, make_drop(st->seed, L.site0 + 2, st->p), L.hd);
    }
  } else
  {
    GemmArgs g{L.x1, W(".linear1.weight").data, L.h, W(".linear1.bias").data, T, FF, D, D, D, FF, 1, 0, D, nullptr, tl_gemm_bf16};
    if (st->p > 0.f) {
      const Drop dr = make_drop(st->seed, L.site0 + 2, st->p);
      g.epi = 1;
      g.C2 = L.hd;
      g.drop_key = dr.key;
      g.drop_thr = dr.thr;
      g.drop_scale = dr.scale;
    }
    gemm_nt_args(g, s);
  }
  t_gemm_nt(st, L.hd, W(".linear2.weight").data, W(".linear2.bias").data, tmp, T, D, FF, 0, s);
  hipLaunchKernelGGL((ln_fwd_g_kernel<D>), dim3((T + 3) / 4), dim3(256), 0, s, (const float*)L.x1, (const float*)tmp, T, W(".norm2.weight").data,
                     W(".norm2.bias").data, make_drop(st->seed, L.site0 + 3, st->p), L.x2, L.xhat2, L.rstd2);
}
// dcur: gradient w.r.t. the layer's output x2 (read). Returns the gradient w.r.t. the layer's input (nullptr when need_dx is false).
template <int D>
static float* text_layer_bwd(TextTrain* st, const TextLayer& L, const float* dcur, bool need_dx, hipStream_t s) {
  constexpr int HD = D / 4, FF = 4 * D;
  const int T = L.T;
  const size_t n = (size_t)T * D;
  float *dA = tbump<float>(st, n), *dB = tbump<float>(st, n), *dC = tbump<float>(st, n), *dB2 = tbump<float>(st, n), *dO = tbump<float>(st, n),
        *dH = tbump<float>(st, n * 4), *dqkv = tbump<float>(st, n * 3);
  auto W = [&](const char* nme) -> const TTensor& { return TT(st, L.prefix + nme); };
  const int ln_grid = std::min(64, (T + 15) / 16);
  hipLaunchKernelGGL((ln_bwd_g_kernel<D>), dim3(ln_grid), dim3(256), 0, s, dcur, (const float*)L.xhat2, (const float*)L.rstd2, T,
                     W(".norm2.weight").data, make_drop(st->seed, L.site0 + 3, st->p), dA, dB, W(".norm2.weight").grad, W(".norm2.bias").grad);
  {
    const Drop dr = make_drop(st->seed, L.site0 + 2, st->p);
    t_gemm_tn_nn(st, dB, L.hd, W(".linear2.weight").grad, W(".linear2.bias").grad, W(".linear2.weight").data, dH, T, D, FF, 0, L.h, &dr, s);
  }
  t_gemm_tn_nn(st, dH, L.x1, W(".linear1.weight").grad, W(".linear1.bias").grad, W(".linear1.weight").data, dA, T, FF, D, 1, nullptr, nullptr, s);
  hipLaunchKernelGGL((ln_bwd_g_kernel<D>), dim3(ln_grid), dim3(256), 0, s, (const float*)dA, (const float*)L.xhat1, (const float*)L.rstd1, T,
                     W(".norm1.weight").data, make_drop(st->seed, L.site0 + 1, st->p), dC, dB2, W(".norm1.weight").grad, W(".norm1.bias").grad);
  t_gemm_tn_nn(st, dB2, L.O, W(".self_attn.out_proj.weight").grad, W(".self_attn.out_proj.bias").grad, W(".self_attn.out_proj.weight").data, dO, T, D,
               D, 0, nullptr, nullptr, s);
  hipLaunchKernelGGL((attn_bwd_g_kernel<HD>), dim3(L.B * 4), dim3(256), attn_lds(L.S, HD, true), s, (const float*)L.qkv, (const float*)L.P,
                     (const float*)dO, dqkv, L.S, make_drop(st->seed, L.site0 + 0, st->p));
  if (need_dx) {
    t_gemm_tn_nn(st, dqkv, L.x_in, W(".self_attn.in_proj_weight").grad, W(".self_attn.in_proj_bias").grad, W(".self_attn.in_proj_weight").data, dC,
                 T, 3 * D, D, 1, nullptr, nullptr, s);
    return dC;
  }
  t_gemm_tn_nn(st, dqkv, L.x_in, W(".self_attn.in_proj_weight").grad, W(".self_attn.in_proj_bias").grad, nullptr, nullptr, T, 3 * D, D, 0, nullptr,
               nullptr, s);
  return nullptr;
}

int text_train_forward_impl(t2l_ctx* ctx, const float* hidden, int n_sent, int L, int n_desc, float p, uint32_t seed, float* out, hipStream_t s) {
  TextTrain* st = tstate(ctx);
  if (!st) return fail(ctx, T2L_ESTATE, "t2l_text_head_train: call t2l_text_train_bind first");
  if (!hidden || !out) return fail(ctx, T2L_EINVAL, "t2l_text_head_train: null buffer");
  if (n_desc < 1 || n_sent < n_desc || n_sent % n_desc) return fail(ctx, T2L_EINVAL, "t2l_text_head_train: the sentences must split evenly over the descriptions");
  const int S = n_sent / n_desc;
  if (L < 1 || L > 32 || S > 32) return fail(ctx, T2L_EINVAL, "t2l_text_head_train: need 1 <= n_tokens <= 32 and <= 32 sentences per description");
  if (!(p >= 0.f && p < 1.f)) return fail(ctx, T2L_EINVAL, "t2l_text_head_train: dropout_p must be in [0, 1)");
  tl_gemm_bf16 = ctx->text_train_bf16;
  tl_gemm_block64 = ctx->train_gemm_block == 64 || (ctx->train_gemm_block == 0 && ctx->text_train_bf16 != 0);
  const size_t T1 = (size_t)n_sent * L;
  // saved activations + the backward's scratch: ~31 floats per (row, column) of each layer, see text_layer_alloc / text_layer_bwd
  const size_t need = sizeof(float) * (32 * (T1 * 1024 + (size_t)n_sent * 256) + 2 * (size_t)n_sent * 4 * L * L + 2 * (size_t)n_desc * 4 * S * S +
                                       8 * (size_t)n_sent * 1024 + 16 * (size_t)n_desc * 256) + (1 << 20);
  if (st->ws_cap < need) {
    if (st->ws) T2L_HIP(ctx, hipFree(st->ws));
    st->ws = nullptr;
    st->ws_cap = 0;
    T2L_HIP(ctx, hipMalloc(&st->ws, need));
    st->ws_cap = need;
  }
  st->ws_off = 0;
  st->have_forward = false;
  st->n_sent = n_sent; st->L = L; st->n_desc = n_desc; st->S = S; st->p = p; st->seed = seed;
  event_begin(ctx, "text_train_forward", s);
  float* tmp = tbump<float>(st, T1 * 1024);
  TextLayer& A = st->intra;
  A = TextLayer{};
  A.prefix = "intra_module.0"; A.T = (int)T1; A.B = n_sent; A.S = L; A.site0 = 0; A.x_in = hidden;
  text_layer_alloc<1024>(st, A, p);
  text_layer_fwd<1024>(st, A, tmp, s);
  st->pooled = tbump<float>(st, (size_t)n_sent * 1024);
  st->tok_arg = tbump<int32_t>(st, (size_t)n_sent * 1024);
  hipLaunchKernelGGL(seq_max_fwd_kernel, dim3((unsigned)(((size_t)n_sent * 1024 + 255) / 256)), dim3(256), 0, s, (const float*)A.x2, (const float*)nullptr,
                     n_sent, L, 1024, st->pooled, st->tok_arg);
  st->mlp_y = tbump<float>(st, (size_t)n_sent * 256);
  st->mlp_out = tbump<float>(st, (size_t)n_sent * 256);
  st->bn_mean = tbump<float>(st, 256);
  st->bn_rstd = tbump<float>(st, 256);
  t_gemm_nt(st, st->pooled, TT(st, "inter_mlp.0.0.weight").data, TT(st, "inter_mlp.0.0.bias").data, st->mlp_y, n_sent, 256, 1024, 0, s);
#define T2L_TEXT_BN_FWD(PHASE, ACC)                                                                                                       \
  hipLaunchKernelGGL((bn_plain_fwd_kernel<PHASE>), dim3(64), dim3(256), 0, s, (const float*)st->mlp_y, n_sent, 256,                       \
                     TT(st, "inter_mlp.0.1.weight").data, TT(st, "inter_mlp.0.1.bias").data, TT(st, "inter_mlp.0.1.running_mean").data,   \
                     TT(st, "inter_mlp.0.1.running_var").data, 0.1f, st->mlp_out, st->bn_mean, st->bn_rstd, ACC)
  if (ctx->sync_fn) {  // statistics | sum over the ranks | apply (t2l_train_sync_bn)
    double* acc = ctx->sync_buf + (size_t)(2 * kBnSlots) * kBnStride;
    ctx->sync_failed = false;
    T2L_TEXT_BN_FWD(1, acc);
    sync_slots(ctx, acc, 1, s);
    T2L_TEXT_BN_FWD(2, acc);
  } else {
    T2L_TEXT_BN_FWD(0, (double*)nullptr);
  }
#undef T2L_TEXT_BN_FWD
  TextLayer& I = st->inter;
  I = TextLayer{};
  I.prefix = "inter_module.0"; I.T = n_sent; I.B = n_desc; I.S = S; I.site0 = 4; I.x_in = st->mlp_out;
  text_layer_alloc<256>(st, I, p);
  text_layer_fwd<256>(st, I, tmp, s);
  st->out = tbump<float>(st, (size_t)n_desc * 256);
  st->sent_arg = tbump<int32_t>(st, (size_t)n_desc * 256);
  hipLaunchKernelGGL(seq_max_fwd_kernel, dim3((unsigned)(((size_t)n_desc * 256 + 255) / 256)), dim3(256), 0, s, (const float*)I.x2,
                     (const float*)st->mlp_out, n_desc, S, 256, st->out, st->sent_arg);
  T2L_HIP(ctx, hipMemcpyAsync(out, st->out, sizeof(float) * (size_t)n_desc * 256, hipMemcpyDeviceToDevice, s));
  event_end(ctx, "text_train_forward", s);
  T2L_HIP(ctx, hipGetLastError());
  if (st->ws_off > st->ws_cap) return fail(ctx, T2L_ENOMEM, "t2l_text_head_train: workspace bound exceeded (internal error)");
  if (ctx->sync_failed) return fail(ctx, T2L_ESTATE, "t2l_text_head_train: the cross-rank sum callback (t2l_train_sync_bn) failed");
  st->have_forward = true;
  return T2L_OK;
}

int text_train_backward_impl(t2l_ctx* ctx, const float* grad_out, hipStream_t s) {
  TextTrain* st = tstate(ctx);
  if (!st || !st->have_forward) return fail(ctx, T2L_ESTATE, "t2l_text_head_backward: no forward pass to differentiate");
  if (!grad_out) return fail(ctx, T2L_EINVAL, "t2l_text_head_backward: null gradient");
  tl_gemm_bf16 = ctx->text_train_bf16;
  tl_gemm_block64 = ctx->train_gemm_block == 64 || (ctx->train_gemm_block == 0 && ctx->text_train_bf16 != 0);
  const size_t mark = st->ws_off;
  const int n_sent = st->n_sent, n_desc = st->n_desc, S = st->S, L = st->L;
  event_begin(ctx, "text_train_backward", s);
  // max over the sentences: the gradient goes to the arg-max row of (x + layer(x)) — to the layer's output AND to the residual x
  float* dY2 = tbump<float>(st, (size_t)n_sent * 256);
  hipLaunchKernelGGL(seq_max_bwd_kernel, dim3((unsigned)(((size_t)n_sent * 256 + 255) / 256)), dim3(256), 0, s, grad_out, (const int32_t*)st->sent_arg,
                     n_desc, S, 256, dY2);
  float* dX = text_layer_bwd<256>(st, st->inter, dY2, true, s);
  hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)(((size_t)n_sent * 256 + 255) / 256)), dim3(256), 0, s, dX, (const float*)dY2, (size_t)n_sent * 256);
  // inter_mlp: BatchNorm (batch statistics), Linear
#define T2L_TEXT_BN_BWD(PHASE, ACC)                                                                                                    \
  hipLaunchKernelGGL((bn_plain_bwd_kernel<PHASE>), dim3(64), dim3(256), 0, s, dX, (const float*)st->mlp_y, n_sent, 256,                 \
                     TT(st, "inter_mlp.0.1.weight").data, (const float*)st->bn_mean, (const float*)st->bn_rstd,                         \
                     TT(st, "inter_mlp.0.1.weight").grad, TT(st, "inter_mlp.0.1.bias").grad, ACC)
  if (ctx->sync_fn) {
    double* acc = ctx->sync_buf + (size_t)(2 * kBnSlots + 1) * kBnStride;
    ctx->sync_failed = false;
    T2L_TEXT_BN_BWD(1, acc);
    sync_slots(ctx, acc, 1, s);
    T2L_TEXT_BN_BWD(2, acc);
  } else {
    T2L_TEXT_BN_BWD(0, (double*)nullptr);
  }
#undef T2L_TEXT_BN_BWD
  float* dpool = tbump<float>(st, (size_t)n_sent * 1024);
  t_gemm_tn_nn(st, dX, st->pooled, TT(st, "inter_mlp.0.0.weight").grad, TT(st, "inter_mlp.0.0.bias").grad, TT(st, "inter_mlp.0.0.weight").data, dpool,
               n_sent, 256, 1024, 0, nullptr, nullptr, s);
  // max over the tokens, then the d = 1024 layer (its input, T5's hidden states, is a constant: no dX)
  float* dX2 = tbump<float>(st, (size_t)n_sent * L * 1024);
  hipLaunchKernelGGL(seq_max_bwd_kernel, dim3((unsigned)(((size_t)n_sent * L * 1024 + 255) / 256)), dim3(256), 0, s, (const float*)dpool,
                     (const int32_t*)st->tok_arg, n_sent, L, 1024, dX2);
  text_layer_bwd<1024>(st, st->intra, dX2, false, s);
  event_end(ctx, "text_train_backward", s);
  const bool over = st->ws_off > st->ws_cap;
  st->ws_off = mark;
  T2L_HIP(ctx, hipGetLastError());
  if (over) return fail(ctx, T2L_ENOMEM, "t2l_text_head_backward: workspace bound exceeded (internal error)");
  if (ctx->sync_failed) return fail(ctx, T2L_ESTATE, "t2l_text_head_backward: the cross-rank sum callback (t2l_train_sync_bn) failed");
  return T2L_OK;
}

// ---- the head's optimizer: torch.optim.Adam's arithmetic (adam_kernel) over every bound parameter in ONE launch
int text_adam_step_impl(t2l_ctx* ctx, float lr, float b1, float b2, float eps, hipStream_t s) {
  TextTrain* st = tstate(ctx);
  if (!st) return fail(ctx, T2L_ESTATE, "t2l_text_adam_step: call t2l_text_train_bind first");
  st->step += 1;
  const float bc1 = (float)(1.0 - pow((double)b1, (double)st->step));
  const float bc2s = (float)sqrt(1.0 - pow((double)b2, (double)st->step));
  event_begin(ctx, "text_adam_step", s);
  hipLaunchKernelGGL(adam_kernel, dim3(st->n_chunks), dim3(256), 0, s, st->d_tensors, st->d_chunks, lr, b1, b2, eps, bc1, bc2s);
  event_end(ctx, "text_adam_step", s);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

int text_zero_grad_impl(t2l_ctx* ctx, hipStream_t s) {
  TextTrain* st = tstate(ctx);
  if (!st) return fail(ctx, T2L_ESTATE, "t2l_text_zero_grad: call t2l_text_train_bind first");
  hipLaunchKernelGGL(zero_kernel, dim3(st->n_chunks), dim3(256), 0, s, st->d_tensors, st->d_chunks);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

int text_adam_state_impl(t2l_ctx* ctx, int set, float* m, float* v, int64_t* step, int64_t* numel, hipStream_t s) {
  TextTrain* st = tstate(ctx);
  if (!st) return fail(ctx, T2L_ESTATE, "t2l_text_adam_state: call t2l_text_train_bind first");
  if (numel) *numel = st->mv_total;
  if (!m && !v) {
    if (step && !set) *step = st->step;
    return T2L_OK;
  }
  if (!m || !v || !step) return fail(ctx, T2L_EINVAL, "t2l_text_adam_state: pass m, v and step together");
  const size_t bytes = sizeof(float) * (size_t)st->mv_total;
  if (set) {
    T2L_HIP(ctx, hipMemcpyAsync(st->mv, m, bytes, hipMemcpyDeviceToDevice, s));
    T2L_HIP(ctx, hipMemcpyAsync(st->mv + st->mv_total, v, bytes, hipMemcpyDeviceToDevice, s));
    st->step = *step;
  } else {
    T2L_HIP(ctx, hipMemcpyAsync(m, st->mv, bytes, hipMemcpyDeviceToDevice, s));
    T2L_HIP(ctx, hipMemcpyAsync(v, st->mv + st->mv_total, bytes, hipMemcpyDeviceToDevice, s));
    *step = st->step;
  }
  return T2L_OK;
}

}  // namespace t2l

#include "pointnet_train.h"
