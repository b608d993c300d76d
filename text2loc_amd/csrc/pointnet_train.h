// PointNet++ object backbone in TRAINING mode (SURVEY.md §8 rows a3 + a9; models/pointcloud/pointnet2.py:18-100 under
// model.train(), trained jointly in the published configuration, README.md:87-99). Included by train.hip (same translation
// unit: it works on the TrainState's live tensors and its GEMM launchers). PARITY UNPINNED like the eval kernels
// (pointnet.hip): the index structure follows the build's deterministic restatement (oracle/t2l_oracle_pointnet.py), the
// training arithmetic oracle/t2l_oracle_pointnet_train.py.
//
// What the reference's code fixes: the backbone is called once PER CELL (models/object_encoder.py:92-95), so every
// BatchNorm1d normalises with the statistics of that cell's rows and updates its running statistics once per cell, in cell
// order. Here the whole batch runs at once: the edge rows of a level are materialised in HBM for all objects
// ([n_obj][centres][33 slots], slot 32 = PyG's bipartite self-loop edge, empty slots masked), the two Linear layers of the
// edge MLP are plain GEMMs over all rows (gemm_kernel), and BatchNorm works per (cell, channel) SEGMENT: float64 partial
// sums per cell, a finalize kernel (statistics + the sequential running-statistics updates), an apply kernel.
// Saved for backward per level: neighbour table, edge inputs, pre-BN and post-ReLU activations of both layers, per-cell
// statistics, arg-max rows. ~14 KB of activations per edge row at SA3: ≈18 GB at B = 64 cells (what 288 GB of HBM are for).
#pragma once

namespace t2l {

struct PnLevel {
  std::string prefix;
  int cin = 0, kin = 0, kp = 0, h1 = 0, h2 = 0;  // source features, real layer-1 inputs (cin + 3), padded to 32, widths
  int ns = 0, nd = 0, R = 0;                      // source points per object, groups per object, rows per group
  float radius = 0.f;
  bool sa = true;
  size_t E = 0;
  int32_t *nbr = nullptr, *arg = nullptr, *cnt = nullptr;
  float *X = nullptr, *y1 = nullptr, *a1 = nullptr, *y2 = nullptr, *a2 = nullptr;
  float *mean1 = nullptr, *rstd1 = nullptr, *mean2 = nullptr, *rstd2 = nullptr;
  float *xout = nullptr, *pos_out = nullptr, *w1p = nullptr, *dw1p = nullptr;
};

struct PnTrain {
  bool bound = false, trainable = false, have_forward = false;
  char* ws = nullptr;
  size_t ws_cap = 0, ws_off = 0;
  int n_obj = 0, n_cells = 0;
  int32_t *cell_of_obj = nullptr, *cell_base = nullptr, *cell_lo = nullptr;  // device
  const float *pos0 = nullptr, *rgb0 = nullptr;
  PnLevel lv[4];
  float *f0 = nullptr, *f1 = nullptr, *f2 = nullptr;
  double* acc = nullptr;  // [n_cells][2][1024]
};

static const char* kPnBlocks[4] = {"sa1.point_conv.local_nn", "sa2.point_conv.local_nn", "sa3.point_conv.local_nn", "ga.mlp"};

namespace train {

__device__ __forceinline__ float pt_d2(float ax, float ay, float az, float bx, float by, float bz) {  // as pointnet.hip: no FMA
  const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// farthest point sampling, one wave per object: selection order from point 0, lowest index on ties (pointnet.hip's rule)
template <int PPL>
__global__ __launch_bounds__(256) void pt_fps_kernel(const float* __restrict__ pos, int n_obj, int nd, float* __restrict__ pos_out) {
  constexpr int NS = 64 * PPL;
  const int lane = threadIdx.x & 63, o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= n_obj) return;
  const float* gp = pos + (size_t)o * NS * 3;
  float mind[PPL], px[PPL], py[PPL], pz[PPL];
#pragma unroll
  for (int q = 0; q < PPL; ++q) {
    const int p = lane + 64 * q;
    px[q] = gp[p * 3];
    py[q] = gp[p * 3 + 1];
    pz[q] = gp[p * 3 + 2];
    mind[q] = 3.0e38f;
  }
  int last = 0;
  for (int t = 0; t < nd; ++t) {
    const float cx = gp[last * 3], cy = gp[last * 3 + 1], cz = gp[last * 3 + 2];
    if (lane < 3) pos_out[((size_t)o * nd + t) * 3 + lane] = gp[last * 3 + lane];
    float best = -1.f;
    int bi = 0;
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
      mind[q] = fminf(mind[q], pt_d2(px[q], py[q], pz[q], cx, cy, cz));
      if (mind[q] > best) {
        best = mind[q];
        bi = lane + 64 * q;
      }
    }
    float m = best;  // distances are >= 0
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    int cand = best == m ? bi : 0x7fffffff;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cand = min(cand, __shfl_xor(cand, off));
    last = cand;
  }
}

// ball query, one wave per centre: the first 32 source points of the centre's object in index order with d^2 < r^2; slot 32 =
// the extra (k -> k) source of PyG's add_self_loops on the bipartite cell batch (or -1); cnt[cell] += valid rows
template <int PPL>
__global__ __launch_bounds__(256) void pt_ball_kernel(const float* __restrict__ pos_src, const float* __restrict__ pos_ctr, int n_obj,
                                                      int nd, float r2, const int32_t* __restrict__ cell_base,
                                                      const int32_t* __restrict__ cell_of_obj, int self_loops,
                                                      int32_t* __restrict__ nbr, int32_t* __restrict__ cnt) {
  constexpr int NS = 64 * PPL;
  const int lane = threadIdx.x & 63;
  const size_t g = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= (size_t)n_obj * nd) return;
  const int o = (int)(g / nd), t = (int)(g % nd);
  const float cx = pos_ctr[g * 3], cy = pos_ctr[g * 3 + 1], cz = pos_ctr[g * 3 + 2];
  const float* gp = pos_src + (size_t)o * NS * 3;
  int32_t* out = nbr + g * 33;
  int count = 0;
#pragma unroll
  for (int q = 0; q < PPL; ++q) {
    const int p = lane + 64 * q;
    const bool in = pt_d2(gp[p * 3], gp[p * 3 + 1], gp[p * 3 + 2], cx, cy, cz) < r2;
    const unsigned long long mask = __ballot(in);
    const int rank = count + __popcll(mask & ((1ull << lane) - 1ull));
    if (in && rank < 32) out[rank] = o * NS + p;
    count += __popcll(mask);
  }
  count = min(count, 32);
  if (lane >= count && lane < 32) out[lane] = -1;
  if (lane == 32) {
    const int cb = cell_base[o];
    out[32] = self_loops ? cb * NS + (o - cb) * nd + t : -1;
    atomicAdd(cnt + cell_of_obj[o], count + (self_loops ? 1 : 0));
  }
}

// edge inputs: X[row] = [x_src | pos_src - pos_centre | 0 pad] (SA), [x | pos | 0 pad] (global MLP: nbr == nullptr)
__global__ __launch_bounds__(256) void pt_gather_kernel(const float* __restrict__ x_src, const float* __restrict__ pos_src,
                                                        const float* __restrict__ pos_ctr, const int32_t* __restrict__ nbr, size_t E,
                                                        int R, int cin, int kp, float* __restrict__ X) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= E * kp) return;
  const size_t row = i / kp;
  const int col = (int)(i % kp);
  const long long src = nbr ? nbr[row] : (long long)row;
  float v = 0.f;
  if (src >= 0) {
    if (col < cin) v = x_src[(size_t)src * cin + col];
    else if (col < cin + 3) v = pos_src[(size_t)src * 3 + col - cin] - (nbr ? pos_ctr[(row / R) * 3 + col - cin] : 0.f);
  }
  X[i] = v;
}

__global__ void pt_pad_kernel(const float* __restrict__ W, int rows, int kin, int kp, float* __restrict__ Wp) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < rows * kp) Wp[i] = (i % kp) < kin ? W[(i / kp) * kin + (i % kp)] : 0.f;
}
__global__ void pt_unpad_add_kernel(const float* __restrict__ dWp, int rows, int kin, int kp, float* __restrict__ dW) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < rows * kin) dW[i] += dWp[(i / kin) * kp + (i % kin)];
}

// per-(cell, channel) sums over the valid rows of one object's row block; grid (C/64, n_obj, chunks)
// MODE 0: acc[cell][0][c] += sum y, acc[cell][1][c] += sum y^2. MODE 1: dv = a > 0 ? d : 0: sum dv, sum dv * xhat
template <int MODE>
__global__ __launch_bounds__(256) void pt_bn_stats_kernel(const float* __restrict__ y, const float* __restrict__ d,
                                                          const float* __restrict__ a, const int32_t* __restrict__ nbr, int C,
                                                          int rows_per_obj, const int32_t* __restrict__ cell_of_obj,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          double* __restrict__ acc) {
  __shared__ float r1[256], r2[256];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  const bool cok = c < C;  // (C = 32: half of the lanes idle)
  const int o = blockIdx.y, cell = cell_of_obj[o];
  const int per = (rows_per_obj + gridDim.z - 1) / gridDim.z;
  const int lo = blockIdx.z * per, hi = cok ? min(rows_per_obj, lo + per) : 0;
  const size_t base = (size_t)o * rows_per_obj;
  float mu = 0.f, rs = 0.f;
  if (MODE == 1 && cok) {
    mu = mean[(size_t)cell * C + c];
    rs = rstd[(size_t)cell * C + c];
  }
  float s1 = 0.f, s2 = 0.f;
  for (int r = lo + g; r < hi; r += 4) {
    const size_t row = base + r;
    if (nbr && nbr[row] < 0) continue;
    const size_t i = row * C + c;
    if (MODE == 0) {
      const float v = y[i];
      s1 += v;
      s2 += v * v;
    } else {
      const float dv = a[i] > 0.f ? d[i] : 0.f;
      s1 += dv;
      s2 += dv * (y[i] - mu) * rs;
    }
  }
  r1[threadIdx.x] = s1;
  r2[threadIdx.x] = s2;
  __syncthreads();
  if (g == 0 && cok) {
    const int t = threadIdx.x;
    atomicAdd(acc + ((size_t)cell * 2) * 1024 + c, (double)r1[t] + (double)r1[t + 64] + (double)r1[t + 128] + (double)r1[t + 192]);
    atomicAdd(acc + ((size_t)cell * 2 + 1) * 1024 + c, (double)r2[t] + (double)r2[t + 64] + (double)r2[t + 128] + (double)r2[t + 192]);
  }
}

// statistics of every cell + the running-statistics updates the reference performs once per cell, in cell order
__global__ void pt_bn_finalize_kernel(const double* __restrict__ acc, const int32_t* __restrict__ cnt, int n_cells, int C,
                                      float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ run_mean,
                                      float* __restrict__ run_var, float momentum) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float rm = run_mean[c], rv = run_var[c];
  for (int cell = 0; cell < n_cells; ++cell) {
    const double n = (double)max(cnt[cell], 1);
    const double m = acc[((size_t)cell * 2) * 1024 + c] / n;
    const double var = fmax(acc[((size_t)cell * 2 + 1) * 1024 + c] / n - m * m, 0.0);
    mean[(size_t)cell * C + c] = (float)m;
    rstd[(size_t)cell * C + c] = 1.0f / sqrtf((float)var + kBnEps);
    rm = (1.f - momentum) * rm + momentum * (float)m;
    rv = (1.f - momentum) * rv + momentum * (float)(var * (n / fmax(n - 1.0, 1.0)));
  }
  run_mean[c] = rm;
  run_var[c] = rv;
}

__global__ __launch_bounds__(256) void pt_bn_apply_fwd_kernel(const float* __restrict__ y, size_t E, int C, int rows_per_obj,
                                                              const int32_t* __restrict__ cell_of_obj, const int32_t* __restrict__ nbr,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ a) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= E * C) return;
  const size_t row = i / C;
  const int c = (int)(i % C);
  float v = 0.f;  // empty slots: 0 (a valid row of the same centre always exists — the centre itself — and ReLU outputs are >= 0)
  if (!nbr || nbr[row] >= 0) {
    const size_t sc = (size_t)cell_of_obj[row / rows_per_obj] * C + c;
    v = fmaxf((y[i] - mean[sc]) * rstd[sc] * gamma[c] + beta[c], 0.f);
  }
  a[i] = v;
}

// d (gradient w.r.t. the ReLU output) -> gradient w.r.t. the Linear output, in place
__global__ __launch_bounds__(256) void pt_bn_apply_bwd_kernel(float* __restrict__ d, const float* __restrict__ a, const float* __restrict__ y,
                                                              size_t E, int C, int rows_per_obj, const int32_t* __restrict__ cell_of_obj,
                                                              const int32_t* __restrict__ nbr, const int32_t* __restrict__ cnt,
                                                              const double* __restrict__ acc, const float* __restrict__ gamma,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= E * C) return;
  const size_t row = i / C;
  const int c = (int)(i % C);
  float v = 0.f;
  if (!nbr || nbr[row] >= 0) {
    const int cell = cell_of_obj[row / rows_per_obj];
    const size_t sc = (size_t)cell * C + c;
    const float n = (float)max(cnt[cell], 1), rs = rstd[sc];
    const float s1 = (float)acc[((size_t)cell * 2) * 1024 + c], s2 = (float)acc[((size_t)cell * 2 + 1) * 1024 + c];
    const float dv = a[i] > 0.f ? d[i] : 0.f;
    v = gamma[c] * rs / n * (n * dv - s1 - (y[i] - mean[sc]) * rs * s2);
  }
  d[i] = v;
}
__global__ void pt_bn_param_grad_kernel(const double* __restrict__ acc, int n_cells, int C, float* __restrict__ dgamma,
                                        float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int cell = 0; cell < n_cells; ++cell) {
    s1 += acc[((size_t)cell * 2) * 1024 + c];
    s2 += acc[((size_t)cell * 2 + 1) * 1024 + c];
  }
  dbeta[c] += (float)s1;
  dgamma[c] += (float)s2;
}

// max over the R rows of every group (valid rows only; first maximum wins) + the winning row
__global__ __launch_bounds__(256) void pt_segmax_kernel(const float* __restrict__ a, const int32_t* __restrict__ nbr, size_t n_groups, int R,
                                                        int C, float* __restrict__ xout, int32_t* __restrict__ arg) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_groups * C) return;
  const size_t g = i / C;
  const int c = (int)(i % C);
  float best = -1.f;
  long long br = -1;
  for (int r = 0; r < R; ++r) {
    const size_t row = g * R + r;
    if (nbr && nbr[row] < 0) continue;
    const float v = a[row * C + c];
    if (v > best) {
      best = v;
      br = (long long)row;
    }
  }
  xout[i] = best;
  arg[i] = (int32_t)br;
}
__global__ __launch_bounds__(256) void pt_maxbwd_kernel(const float* __restrict__ dxout, const int32_t* __restrict__ arg, size_t E, int R, int C,
                                                        float* __restrict__ dA) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= E * C) return;
  const size_t row = i / C, g = row / R;
  const int c = (int)(i % C);
  dA[i] = arg[g * C + c] == (int32_t)row ? dxout[g * C + c] : 0.f;
}
// dx_src[src][0:cin] += dX[row][0:cin]
__global__ __launch_bounds__(256) void pt_scatter_kernel(const float* __restrict__ dX, const int32_t* __restrict__ nbr, size_t E, int cin, int kp,
                                                         float* __restrict__ dx_src) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= E * cin) return;
  const size_t row = i / cin;
  const int col = (int)(i % cin);
  const int src = nbr[row];
  if (src >= 0) unsafeAtomicAdd(dx_src + (size_t)src * cin + col, dX[row * kp + col]);
}
__global__ void pt_slice_kernel(const float* __restrict__ dX, size_t rows, int cin, int kp, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < rows * cin) out[i] = dX[(i / cin) * kp + (i % cin)];
}
__global__ void pt_relu_mask_kernel(float* __restrict__ d, const float* __restrict__ f, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n && !(f[i] > 0.f)) d[i] = 0.f;
}

}  // namespace train

static PnTrain* pn_state(TrainState* st) { return reinterpret_cast<PnTrain*>(st->pn); }

static void pn_train_free(void* p) {
  PnTrain* pt = reinterpret_cast<PnTrain*>(p);
  if (!pt) return;
  if (pt->ws) (void)hipFree(pt->ws);
  delete pt;
}

// grid.y carries the row tiles of gemm_kernel: at most 65,535 of them per launch — the edge matrices have millions of rows
constexpr int kGemmRowSlice = 65535 * 32;
static void gemm_nt_rows(const float* X, const float* W, const float* b, float* Y, size_t M, int N, int K, int relu, hipStream_t s) {
  for (size_t r0 = 0; r0 < M; r0 += kGemmRowSlice)
    gemm_nt(X + r0 * K, W, b, Y + r0 * N, (int)std::min<size_t>(kGemmRowSlice, M - r0), N, K, relu, s);
}
static void gemm_nn_rows(const float* dY, const float* W, float* dX, size_t M, int N, int Kp, hipStream_t s) {
  for (size_t r0 = 0; r0 < M; r0 += kGemmRowSlice)
    gemm_nn(dY + r0 * N, W, dX + r0 * Kp, (int)std::min<size_t>(kGemmRowSlice, M - r0), N, Kp, 0, s);
}

// object_encoder.pointnet.* tensors of the binding: all of them with gradient buffers -> the backbone trains in the engine
// (their names join the Adam list); all without -> frozen (models/object_encoder.py:53-55: requires_grad_(False), but still
// under model.train(): batch statistics + running-statistics updates in the forward); absent -> no backbone on the path
static int pn_train_bind(t2l_ctx* ctx, TrainState* st, std::vector<std::string>& adam) {
  const std::string P = "object_encoder.pointnet.";
  const int cin3[4] = {6, 67, 131, 259}, h1[4] = {32, 128, 256, 512}, h2[4] = {64, 128, 256, 1024};
  std::vector<std::pair<std::string, int64_t>> req, bufs;
  for (int l = 0; l < 4; ++l) {
    const std::string b = P + kPnBlocks[l];
    const int64_t k[2] = {cin3[l], h1[l]}, c[2] = {h1[l], h2[l]};
    for (int i = 0; i < 2; ++i) {
      const std::string q = b + "." + std::to_string(i);
      req.push_back({q + ".0.weight", k[i] * c[i]});
      req.push_back({q + ".0.bias", c[i]});
      req.push_back({q + ".1.weight", c[i]});
      req.push_back({q + ".1.bias", c[i]});
      bufs.push_back({q + ".1.running_mean", c[i]});
      bufs.push_back({q + ".1.running_var", c[i]});
    }
  }
  req.push_back({P + "lin1.weight", 512 * 1024});
  req.push_back({P + "lin1.bias", 512});
  req.push_back({P + "lin2.weight", 256 * 512});
  req.push_back({P + "lin2.bias", 256});
  int with_grad = 0, present = 0;
  for (auto& r : req) {
    auto it = st->t.find(r.first);
    if (it == st->t.end()) continue;
    ++present;
    if (it->second.grad) ++with_grad;
  }
  if (present == 0) return T2L_OK;  // no backbone in this binding (precomputed features2 / class embedding)
  const bool trainable = with_grad > 0;
  if (present != (int)req.size() || (trainable && with_grad != (int)req.size()))
    return fail(ctx, T2L_EINVAL, "t2l_train_bind: object_encoder.pointnet.* must be bound completely — every sa*/ga/lin1/lin2 tensor, "
                                 "all of them with gradient buffers (trained jointly) or none (--pointnet_freeze: batch statistics in "
                                 "the forward, no backward)");
  int rc;
  for (auto& r : req)
    if ((rc = need(ctx, st, r.first, r.second, trainable, nullptr))) return rc;
  for (auto& r : bufs)
    if ((rc = need(ctx, st, r.first, r.second, false, nullptr))) return rc;
  if (trainable)
    for (auto& r : req) adam.push_back(r.first);
  PnTrain* pt = new PnTrain();
  pt->bound = true;
  pt->trainable = trainable;
  st->pn = pt;
  return T2L_OK;
}

template <typename T>
static T* pn_bump(PnTrain* pt, size_t count) {
  const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
  T* p = reinterpret_cast<T*>(pt->ws + pt->ws_off);
  pt->ws_off += bytes;
  return p;
}
static inline unsigned pn_blocks(size_t n) { return (unsigned)((n + 255) / 256); }

static size_t pn_workspace_bytes(int n_obj, int n_cells) {
  const int ns[3] = {256, 128, 64}, cin[3] = {3, 64, 128}, h1[3] = {32, 128, 256}, h2[3] = {64, 128, 256};
  size_t fl = 0, big = 0;
  for (int l = 0; l < 3; ++l) {
    const size_t nd = ns[l] / 2, E = (size_t)n_obj * nd * 33, kp = ((cin[l] + 3 + 31) / 32) * 32;
    fl += E * (kp + 2 * h1[l] + 2 * h2[l] + 1) + (size_t)n_obj * nd * (2 * h2[l] + 3) + (size_t)n_cells * (2 * h1[l] + 2 * h2[l] + 1) + 2 * h1[l] * kp;
    big = std::max(big, E * (size_t)(h1[l] + h2[l] + kp));
  }
  const size_t Eg = (size_t)n_obj * 32;
  fl += Eg * (288 + 2 * 512 + 2 * 1024) + (size_t)n_obj * (2 * 1024 + 512 + 256 + 1024) + (size_t)n_cells * (2 * 512 + 2 * 1024 + 1) + 2 * 512 * 288;
  big = std::max(big, Eg * (size_t)(512 + 1024 + 288));
  fl += big;                                                    // backward scratch (dA2, dA1, dX of the largest level)
  fl += (size_t)n_obj * (256 * 3 + 128 * 64 + 64 * 128 + 32 * 256) * 2;  // dx buffers (+ slack)
  return fl * sizeof(float) + (size_t)n_cells * 2 * 1024 * sizeof(double) + (size_t)(3 * n_obj + 64) * sizeof(int32_t) + (8u << 20);
}

// one get_mlp block in training mode over segmented rows: y = X W^T + b; per-cell BatchNorm; ReLU
static void pn_block_fwd(TrainState* st, PnTrain* pt, const PnLevel& L, int layer, const float* X, const float* W, int K, int C, float* y,
                         float* a, float* mean, float* rstd, hipStream_t s) {
  using namespace train;
  const std::string p = L.prefix + "." + std::to_string(layer);
  const int rows_per_obj = L.nd * L.R;
  gemm_nt_rows(X, W, T_(st, p + ".0.bias").data, y, L.E, C, K, 0, s);
  (void)hipMemsetAsync(pt->acc, 0, sizeof(double) * 2 * 1024 * pt->n_cells, s);
  const int chunks = std::max(1, std::min(8, rows_per_obj / 256));
  hipLaunchKernelGGL((pt_bn_stats_kernel<0>), dim3((C + 63) / 64, pt->n_obj, chunks), dim3(256), 0, s, y, (const float*)nullptr, (const float*)nullptr,
                     (const int32_t*)L.nbr, C, rows_per_obj, (const int32_t*)pt->cell_of_obj, (const float*)nullptr, (const float*)nullptr, pt->acc);
  hipLaunchKernelGGL(pt_bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const double*)pt->acc, (const int32_t*)L.cnt, pt->n_cells, C,
                     mean, rstd, T_(st, p + ".1.running_mean").data, T_(st, p + ".1.running_var").data, 0.1f);
  hipLaunchKernelGGL(pt_bn_apply_fwd_kernel, dim3(pn_blocks(L.E * C)), dim3(256), 0, s, (const float*)y, L.E, C, rows_per_obj,
                     (const int32_t*)pt->cell_of_obj, (const int32_t*)L.nbr, (const float*)mean, (const float*)rstd,
                     (const float*)T_(st, p + ".1.weight").data, (const float*)T_(st, p + ".1.bias").data, a);
}

// d: gradient w.r.t. the block's ReLU output [E, C] (overwritten with the gradient w.r.t. the Linear output)
static void pn_block_bwd(TrainState* st, PnTrain* pt, const PnLevel& L, int layer, float* d, const float* y, const float* a, int C,
                         const float* mean, const float* rstd, hipStream_t s) {
  using namespace train;
  const std::string p = L.prefix + "." + std::to_string(layer);
  const int rows_per_obj = L.nd * L.R;
  (void)hipMemsetAsync(pt->acc, 0, sizeof(double) * 2 * 1024 * pt->n_cells, s);
  const int chunks = std::max(1, std::min(8, rows_per_obj / 256));
  hipLaunchKernelGGL((pt_bn_stats_kernel<1>), dim3((C + 63) / 64, pt->n_obj, chunks), dim3(256), 0, s, y, (const float*)d, a, (const int32_t*)L.nbr, C,
                     rows_per_obj, (const int32_t*)pt->cell_of_obj, mean, rstd, pt->acc);
  hipLaunchKernelGGL(pt_bn_apply_bwd_kernel, dim3(pn_blocks(L.E * C)), dim3(256), 0, s, d, a, y, L.E, C, rows_per_obj,
                     (const int32_t*)pt->cell_of_obj, (const int32_t*)L.nbr, (const int32_t*)L.cnt, (const double*)pt->acc,
                     (const float*)T_(st, p + ".1.weight").data, mean, rstd);
  hipLaunchKernelGGL(pt_bn_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const double*)pt->acc, pt->n_cells, C,
                     T_(st, p + ".1.weight").grad, T_(st, p + ".1.bias").grad);
}

int pn_train_forward_impl(t2l_ctx* ctx, const float* pos, const float* rgb, const int32_t* cell_offsets, int n_cells, float* out_f2,
                          hipStream_t s) {
  using namespace train;
  TrainState* st = state(ctx);
  PnTrain* pt = st ? pn_state(st) : nullptr;
  if (!pt || !pt->bound)
    return fail(ctx, T2L_ESTATE, "t2l_pointnet_features_train: bind the object_encoder.pointnet.* tensors (with gradients) first (t2l_train_bind)");
  if (!pos || !rgb || !cell_offsets || n_cells <= 0 || !out_f2) return fail(ctx, T2L_EINVAL, "t2l_pointnet_features_train: bad arguments");
  const int n_obj = cell_offsets[n_cells];
  if (n_obj <= 0) return fail(ctx, T2L_EINVAL, "t2l_pointnet_features_train: no objects");
  const size_t need = pn_workspace_bytes(n_obj, n_cells);
  if (need > pt->ws_cap) {
    T2L_HIP(ctx, hipStreamSynchronize(s));
    if (pt->ws) (void)hipFree(pt->ws);
    pt->ws = nullptr;
    pt->ws_cap = 0;
    T2L_HIP(ctx, hipMalloc(&pt->ws, need));
    pt->ws_cap = need;
  }
  pt->ws_off = 0;
  pt->have_forward = false;
  pt->n_obj = n_obj;
  pt->n_cells = n_cells;
  pt->pos0 = pos;
  pt->rgb0 = rgb;
  tl_gemm_bf16 = ctx->train_bf16;
  {  // object -> cell tables
    std::vector<int32_t> h(3 * (size_t)n_obj);
    for (int c = 0; c < n_cells; ++c)
      for (int o = cell_offsets[c]; o < cell_offsets[c + 1]; ++o) {
        h[o] = c;
        h[n_obj + o] = cell_offsets[c];
      }
    pt->cell_of_obj = pn_bump<int32_t>(pt, n_obj);
    pt->cell_base = pn_bump<int32_t>(pt, n_obj);
    T2L_HIP(ctx, hipMemcpyAsync(pt->cell_of_obj, h.data(), sizeof(int32_t) * n_obj, hipMemcpyHostToDevice, s));
    T2L_HIP(ctx, hipMemcpyAsync(pt->cell_base, h.data() + n_obj, sizeof(int32_t) * n_obj, hipMemcpyHostToDevice, s));
    T2L_HIP(ctx, hipStreamSynchronize(s));  // h goes out of scope
  }
  pt->acc = pn_bump<double>(pt, (size_t)n_cells * 2 * 1024);
  event_begin(ctx, "pointnet_train_forward", s);
  const std::string P = "object_encoder.pointnet.";
  const int ns[3] = {256, 128, 64}, cin[3] = {3, 64, 128}, h1[4] = {32, 128, 256, 512}, h2[4] = {64, 128, 256, 1024};
  const float radius[3] = {0.2f, 0.3f, 0.4f};
  const float* cur_pos = pos;
  const float* cur_x = rgb;
  for (int l = 0; l < 4; ++l) {
    PnLevel& L = pt->lv[l];
    L.prefix = P + kPnBlocks[l];
    L.sa = l < 3;
    L.cin = l < 3 ? cin[l] : 256;
    L.kin = L.cin + 3;
    L.kp = (L.kin + 31) / 32 * 32;  // gemm_kernel tiles are 32 wide on every side
    L.h1 = h1[l];
    L.h2 = h2[l];
    L.ns = l < 3 ? ns[l] : 32;
    L.nd = l < 3 ? ns[l] / 2 : 1;
    L.R = l < 3 ? 33 : 32;
    L.E = (size_t)n_obj * L.nd * L.R;
    L.cnt = pn_bump<int32_t>(pt, n_cells);
    if (L.sa) {
      L.radius = radius[l];
      L.pos_out = pn_bump<float>(pt, (size_t)n_obj * L.nd * 3);
      L.nbr = pn_bump<int32_t>(pt, L.E);
      if (L.ns == 256) hipLaunchKernelGGL((pt_fps_kernel<4>), dim3((n_obj + 3) / 4), dim3(256), 0, s, cur_pos, n_obj, L.nd, L.pos_out);
      else if (L.ns == 128) hipLaunchKernelGGL((pt_fps_kernel<2>), dim3((n_obj + 3) / 4), dim3(256), 0, s, cur_pos, n_obj, L.nd, L.pos_out);
      else hipLaunchKernelGGL((pt_fps_kernel<1>), dim3((n_obj + 3) / 4), dim3(256), 0, s, cur_pos, n_obj, L.nd, L.pos_out);
      T2L_HIP(ctx, hipMemsetAsync(L.cnt, 0, sizeof(int32_t) * n_cells, s));
      const float r2 = radius[l] * radius[l];  // float32 product, as the restatement
      const unsigned bg = (unsigned)(((size_t)n_obj * L.nd + 3) / 4);
      if (L.ns == 256)
        hipLaunchKernelGGL((pt_ball_kernel<4>), dim3(bg), dim3(256), 0, s, cur_pos, (const float*)L.pos_out, n_obj, L.nd, r2,
                           (const int32_t*)pt->cell_base, (const int32_t*)pt->cell_of_obj, ctx->pn_self_loops, L.nbr, L.cnt);
      else if (L.ns == 128)
        hipLaunchKernelGGL((pt_ball_kernel<2>), dim3(bg), dim3(256), 0, s, cur_pos, (const float*)L.pos_out, n_obj, L.nd, r2,
                           (const int32_t*)pt->cell_base, (const int32_t*)pt->cell_of_obj, ctx->pn_self_loops, L.nbr, L.cnt);
      else
        hipLaunchKernelGGL((pt_ball_kernel<1>), dim3(bg), dim3(256), 0, s, cur_pos, (const float*)L.pos_out, n_obj, L.nd, r2,
                           (const int32_t*)pt->cell_base, (const int32_t*)pt->cell_of_obj, ctx->pn_self_loops, L.nbr, L.cnt);
    } else {
      L.nbr = nullptr;
      std::vector<int32_t> hc(n_cells);
      for (int c = 0; c < n_cells; ++c) hc[c] = (cell_offsets[c + 1] - cell_offsets[c]) * 32;
      T2L_HIP(ctx, hipMemcpyAsync(L.cnt, hc.data(), sizeof(int32_t) * n_cells, hipMemcpyHostToDevice, s));
      T2L_HIP(ctx, hipStreamSynchronize(s));
    }
    L.X = pn_bump<float>(pt, L.E * L.kp);
    hipLaunchKernelGGL(pt_gather_kernel, dim3(pn_blocks(L.E * L.kp)), dim3(256), 0, s, cur_x, cur_pos, (const float*)L.pos_out,
                       (const int32_t*)L.nbr, L.E, L.R, L.cin, L.kp, L.X);
    L.w1p = pn_bump<float>(pt, (size_t)L.h1 * L.kp);
    L.dw1p = pn_bump<float>(pt, (size_t)L.h1 * L.kp);
    hipLaunchKernelGGL(pt_pad_kernel, dim3(pn_blocks((size_t)L.h1 * L.kp)), dim3(256), 0, s, (const float*)T_(st, L.prefix + ".0.0.weight").data,
                       L.h1, L.kin, L.kp, L.w1p);
    L.y1 = pn_bump<float>(pt, L.E * L.h1);
    L.a1 = pn_bump<float>(pt, L.E * L.h1);
    L.y2 = pn_bump<float>(pt, L.E * L.h2);
    L.a2 = pn_bump<float>(pt, L.E * L.h2);
    L.mean1 = pn_bump<float>(pt, (size_t)n_cells * L.h1);
    L.rstd1 = pn_bump<float>(pt, (size_t)n_cells * L.h1);
    L.mean2 = pn_bump<float>(pt, (size_t)n_cells * L.h2);
    L.rstd2 = pn_bump<float>(pt, (size_t)n_cells * L.h2);
    pn_block_fwd(st, pt, L, 0, L.X, L.w1p, L.kp, L.h1, L.y1, L.a1, L.mean1, L.rstd1, s);
    pn_block_fwd(st, pt, L, 1, L.a1, T_(st, L.prefix + ".1.0.weight").data, L.h1, L.h2, L.y2, L.a2, L.mean2, L.rstd2, s);
    const size_t n_groups = (size_t)n_obj * L.nd;
    L.xout = pn_bump<float>(pt, n_groups * L.h2);
    L.arg = pn_bump<int32_t>(pt, n_groups * L.h2);
    hipLaunchKernelGGL(pt_segmax_kernel, dim3(pn_blocks(n_groups * L.h2)), dim3(256), 0, s, (const float*)L.a2, (const int32_t*)L.nbr, n_groups, L.R,
                       L.h2, L.xout, L.arg);
    if (L.sa) {
      cur_pos = L.pos_out;
      cur_x = L.xout;
    }
  }
  pt->f0 = pt->lv[3].xout;
  pt->f1 = pn_bump<float>(pt, (size_t)n_obj * 512);
  pt->f2 = pn_bump<float>(pt, (size_t)n_obj * 256);
  gemm_nt(pt->f0, T_(st, P + "lin1.weight").data, T_(st, P + "lin1.bias").data, pt->f1, n_obj, 512, 1024, 1, s);
  gemm_nt(pt->f1, T_(st, P + "lin2.weight").data, T_(st, P + "lin2.bias").data, pt->f2, n_obj, 256, 512, 1, s);
  T2L_HIP(ctx, hipMemcpyAsync(out_f2, pt->f2, sizeof(float) * (size_t)n_obj * 256, hipMemcpyDeviceToDevice, s));
  event_end(ctx, "pointnet_train_forward", s);
  T2L_HIP(ctx, hipGetLastError());
  if (pt->ws_off > pt->ws_cap) return fail(ctx, T2L_ENOMEM, "t2l_pointnet_features_train: workspace bound exceeded (internal error)");
  pt->have_forward = true;
  return T2L_OK;
}

int pn_train_backward_impl(t2l_ctx* ctx, const float* grad_f2, hipStream_t s) {
  using namespace train;
  TrainState* st = state(ctx);
  PnTrain* pt = st ? pn_state(st) : nullptr;
  if (!pt || !pt->have_forward) return fail(ctx, T2L_ESTATE, "t2l_pointnet_backward: no training-mode forward to differentiate");
  if (!pt->trainable)
    return fail(ctx, T2L_ESTATE, "t2l_pointnet_backward: the backbone was bound without gradient buffers (frozen)");
  if (!grad_f2) return fail(ctx, T2L_EINVAL, "t2l_pointnet_backward: null gradient");
  const size_t mark = pt->ws_off;
  const int n_obj = pt->n_obj;
  tl_gemm_bf16 = ctx->train_bf16;
  const std::string P = "object_encoder.pointnet.";
  event_begin(ctx, "pointnet_train_backward", s);
  float* d2 = pn_bump<float>(pt, (size_t)n_obj * 256);
  float* d1 = pn_bump<float>(pt, (size_t)n_obj * 512);
  float* dx = pn_bump<float>(pt, (size_t)n_obj * 1024);  // gradient w.r.t. the current level's output
  T2L_HIP(ctx, hipMemcpyAsync(d2, grad_f2, sizeof(float) * (size_t)n_obj * 256, hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(pt_relu_mask_kernel, dim3(pn_blocks((size_t)n_obj * 256)), dim3(256), 0, s, d2, (const float*)pt->f2, (size_t)n_obj * 256);
  gemm_tn(d2, pt->f1, T_(st, P + "lin2.weight").grad, T_(st, P + "lin2.bias").grad, n_obj, 256, 512, s);
  gemm_nn(d2, T_(st, P + "lin2.weight").data, d1, n_obj, 256, 512, 0, s);
  hipLaunchKernelGGL(pt_relu_mask_kernel, dim3(pn_blocks((size_t)n_obj * 512)), dim3(256), 0, s, d1, (const float*)pt->f1, (size_t)n_obj * 512);
  gemm_tn(d1, pt->f0, T_(st, P + "lin1.weight").grad, T_(st, P + "lin1.bias").grad, n_obj, 512, 1024, s);
  gemm_nn(d1, T_(st, P + "lin1.weight").data, dx, n_obj, 512, 1024, 0, s);
  for (int l = 3; l >= 0; --l) {
    const PnLevel& L = pt->lv[l];
    const size_t lmark = pt->ws_off;
    float* dA2 = pn_bump<float>(pt, L.E * L.h2);
    float* dA1 = pn_bump<float>(pt, L.E * L.h1);
    hipLaunchKernelGGL(pt_maxbwd_kernel, dim3(pn_blocks(L.E * L.h2)), dim3(256), 0, s, (const float*)dx, (const int32_t*)L.arg, L.E, L.R, L.h2, dA2);
    pn_block_bwd(st, pt, L, 1, dA2, L.y2, L.a2, L.h2, L.mean2, L.rstd2, s);
    gemm_tn(dA2, L.a1, T_(st, L.prefix + ".1.0.weight").grad, T_(st, L.prefix + ".1.0.bias").grad, (int)L.E, L.h2, L.h1, s);
    gemm_nn_rows(dA2, T_(st, L.prefix + ".1.0.weight").data, dA1, L.E, L.h2, L.h1, s);
    pn_block_bwd(st, pt, L, 0, dA1, L.y1, L.a1, L.h1, L.mean1, L.rstd1, s);
    T2L_HIP(ctx, hipMemsetAsync(L.dw1p, 0, sizeof(float) * (size_t)L.h1 * L.kp, s));
    gemm_tn(dA1, L.X, L.dw1p, T_(st, L.prefix + ".0.0.bias").grad, (int)L.E, L.h1, L.kp, s);
    hipLaunchKernelGGL(pt_unpad_add_kernel, dim3(pn_blocks((size_t)L.h1 * L.kin)), dim3(256), 0, s, (const float*)L.dw1p, L.h1, L.kin, L.kp,
                       T_(st, L.prefix + ".0.0.weight").grad);
    if (l > 0) {  // the input gradient: features of the level below (positions are data)
      float* dX = pn_bump<float>(pt, L.E * L.kp);
      gemm_nn_rows(dA1, L.w1p, dX, L.E, L.h1, L.kp, s);
      const PnLevel& Lb = pt->lv[l - 1];
      const size_t nprev = (size_t)n_obj * Lb.nd * Lb.h2;
      // dx lives below the per-level scratch: write the new one after it, then move it down
      float* dnew = pn_bump<float>(pt, nprev);
      if (L.sa) {
        T2L_HIP(ctx, hipMemsetAsync(dnew, 0, sizeof(float) * nprev, s));
        hipLaunchKernelGGL(pt_scatter_kernel, dim3(pn_blocks(L.E * L.cin)), dim3(256), 0, s, (const float*)dX, (const int32_t*)L.nbr, L.E, L.cin, L.kp, dnew);
      } else {
        hipLaunchKernelGGL(pt_slice_kernel, dim3(pn_blocks(L.E * L.cin)), dim3(256), 0, s, (const float*)dX, L.E, L.cin, L.kp, dnew);
      }
      pt->ws_off = lmark;
      dx = pn_bump<float>(pt, nprev);  // == the old dA2 region start: move the result there
      T2L_HIP(ctx, hipMemcpyAsync(dx, dnew, sizeof(float) * nprev, hipMemcpyDeviceToDevice, s));
    } else {
      pt->ws_off = lmark;
    }
  }
  event_end(ctx, "pointnet_train_backward", s);
  pt->ws_off = mark;
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

}  // namespace t2l
