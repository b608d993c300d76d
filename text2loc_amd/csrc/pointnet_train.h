// PointNet++ object backbone in TRAINING mode (SURVEY.md §8 rows a3 + a9; models/pointcloud/pointnet2.py:18-100 under
// model.train(), trained jointly in the published configuration, README.md:87-99). Included by train.hip (same translation
// unit: it works on the TrainState's live tensors and its GEMM launchers). PARITY UNPINNED like the eval kernels
// (pointnet.hip): the index structure follows the build's deterministic restatement (oracle/t2l_oracle_pointnet.py), the
// training arithmetic oracle/t2l_oracle_pointnet_train.py.
//
// What the reference's code fixes: the backbone is called once PER CELL (models/object_encoder.py:92-95), so every
// BatchNorm1d normalises with the statistics of that cell's rows and updates its running statistics once per cell, in cell
// order. Here the whole batch runs at once. The index structure of all three levels depends on positions only, so it is
// built first (FPS, ball query into a [group][33 slots] table, slot 32 = PyG's bipartite self-loop edge); the host
// prefix-sums the rows per group and the edge rows of a level are then COMPACTED — only real edges exist as rows (ball
// queries fill 37-47 % of their 32 slots on object-like point sets), addressed through src[row] / row_group[row] /
// row_cell[row] and goff[group]. The two Linear layers of the edge MLP are plain GEMMs over all rows (gemm_kernel), and
// BatchNorm works per (cell, channel) SEGMENT: float64 partial sums per cell, a finalize kernel (statistics + the
// sequential running-statistics updates), an apply kernel. Saved for backward per level: the row tables, edge inputs,
// pre-BN and post-ReLU activations of both layers, per-cell statistics, arg-max rows.
#pragma once
#include "gemm_rows2.h"

namespace t2l {

struct PnLevel {
  std::string prefix;
  int cin = 0, kin = 0, kp = 0, h1 = 0, h2 = 0;  // source features, real layer-1 inputs (cin + 3), padded to 32, widths
  int ns = 0, nd = 0;                             // source points per object, groups (centres) per object
  float radius = 0.f;
  bool sa = true;
  size_t E = 0, G = 0;                            // edge rows (valid ones only: compacted), groups
  int32_t *nbr33 = nullptr, *cnt_g = nullptr;     // index workspace: ball-query table [G][33] (-1 = empty), rows per group
  int32_t *goff = nullptr, *src = nullptr, *row_group = nullptr, *row_cell = nullptr, *arg = nullptr, *cnt = nullptr;
  float *X = nullptr, *y1 = nullptr, *a1 = nullptr, *y2 = nullptr;  // (the second layer's post-ReLU output is never stored)
  float *mean1 = nullptr, *rstd1 = nullptr, *mean2 = nullptr, *rstd2 = nullptr;
  float* rg1 = nullptr;  // rstd1 * gamma1 per (cell, channel): the fused BatchNorm + ReLU operand loads of the second version
  float* ysel = nullptr;  // pre-BatchNorm value of the second layer at the arg-max row, per (group, channel)
  float *xout = nullptr, *pos_out = nullptr, *w1p = nullptr;
};

struct PnTrain {
  bool bound = false, trainable = false, have_forward = false;
  char *ws = nullptr, *iws = nullptr;  // activations + scratch (sized exactly per call) / index tables (worst case, small)
  size_t ws_cap = 0, ws_off = 0, iws_cap = 0, scratch_off = 0, scratch_bytes = 0;
  int n_obj = 0, n_cells = 0;
  bool half = false;  // the edge-row tensors of the forward in memory are bf16 (train_bf16 == 1 at the forward; the backward must match)
  int32_t *cell_of_obj = nullptr, *cell_base = nullptr;  // device
  const float *pos0 = nullptr, *rgb0 = nullptr;
  PnLevel lv[4];
  float *f0 = nullptr, *f1 = nullptr, *f2 = nullptr;
  double* acc = nullptr;  // [n_cells][2][1024]
  float *bk1 = nullptr, *bk2 = nullptr;  // [n_cells][1024]: the BatchNorm backward's per-(cell, channel) coefficients (pt_bn_bwd_finalize_kernel)
  float* wt = nullptr;    // [1024 x 512]: a weight matrix transposed for the input-gradient GEMM
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
// the extra (k -> k) source of PyG's add_self_loops on the bipartite cell batch (or -1); cnt_g[group] = valid slots
template <int PPL>
__global__ __launch_bounds__(256) void pt_ball_kernel(const float* __restrict__ pos_src, const float* __restrict__ pos_ctr, int n_obj,
                                                      int nd, float r2, const int32_t* __restrict__ cell_base, int self_loops,
                                                      int32_t* __restrict__ nbr, int32_t* __restrict__ cnt_g) {
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
    cnt_g[g] = count + (self_loops ? 1 : 0);
  }
}

// [group][33 slots] -> compacted rows: src (source row of the level below), row_group, row_cell
__global__ __launch_bounds__(256) void pt_compact_kernel(const int32_t* __restrict__ nbr33, const int32_t* __restrict__ goff, size_t G, int nd,
                                                         const int32_t* __restrict__ cell_of_obj, int32_t* __restrict__ src,
                                                         int32_t* __restrict__ row_group, int32_t* __restrict__ row_cell) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= G * 33) return;
  const size_t g = i / 33;
  const int slot = (int)(i % 33);
  const int v = nbr33[i];
  if (v < 0) return;
  const int row = slot < 32 ? goff[g] + slot : goff[g + 1] - 1;  // the valid slots 0..count-1 come first, the self-loop edge last
  src[row] = v;
  row_group[row] = (int32_t)g;
  row_cell[row] = cell_of_obj[g / nd];
}
// global MLP: 32 rows per object, all valid
__global__ __launch_bounds__(256) void pt_iota_rows_kernel(size_t E, int per, const int32_t* __restrict__ cell_of_obj, int32_t* __restrict__ src,
                                                           int32_t* __restrict__ row_group, int32_t* __restrict__ row_cell,
                                                           int32_t* __restrict__ goff) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i > E) return;
  if (i % per == 0) goff[i / per] = (int32_t)i;  // (incl. goff[G] = E)
  if (i == E) return;
  src[i] = (int32_t)i;
  row_group[i] = (int32_t)(i / per);
  row_cell[i] = cell_of_obj[i / per];
}

// edge inputs: X[row] = [x_src | pos_src - pos_centre | 0 pad] (SA), [x | pos | 0 pad] (global MLP: pos_ctr == nullptr).
// One thread per 4 consecutive columns (kp is a multiple of 32): whole float4 loads of the source row where the 4 columns are
// features and cin is a multiple of 4 (every level but the first, whose 3 + 3 real columns fit the scalar path).
// ST: storage type of the edge-row tensors (float, or pn_bf16 with bf16 GEMM operands: gemm_rows2.h)
template <typename ST>
__global__ __launch_bounds__(256) void pt_gather_kernel(const float* __restrict__ x_src, const float* __restrict__ pos_src,
                                                        const float* __restrict__ pos_ctr, const int32_t* __restrict__ src,
                                                        const int32_t* __restrict__ row_group, size_t E, int cin, int kp,
                                                        ST* __restrict__ X) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int q = kp >> 2;
  if (i >= E * q) return;
  size_t row;
  int col;
  if (E * q < 0x7fffffffull) {  // (a 64-bit division costs more than the rest of this thread's work)
    const unsigned iu = (unsigned)i, ru = iu / (unsigned)q;
    row = ru;
    col = (int)(iu - ru * (unsigned)q) * 4;
  } else {
    row = i / q;
    col = (int)(i % q) * 4;
  }
  const size_t sr = (size_t)src[row];
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col + 4 <= cin && (cin & 3) == 0) {
    v = *reinterpret_cast<const float4*>(x_src + sr * cin + col);
  } else if (col < cin + 3) {
    float e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = col + j;
      float x = 0.f;
      if (c < cin) x = x_src[sr * cin + c];
      else if (c < cin + 3) x = pos_src[sr * 3 + c - cin] - (pos_ctr ? pos_ctr[(size_t)row_group[row] * 3 + c - cin] : 0.f);
      e[j] = x;
    }
    v = make_float4(e[0], e[1], e[2], e[3]);
  }
  pn_st4(X + row * kp + col, v);
}

__global__ void pt_pad_kernel(const float* __restrict__ W, int rows, int kin, int kp, float* __restrict__ Wp) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < rows * kp) Wp[i] = (i % kp) < kin ? W[(i / kp) * kin + (i % kp)] : 0.f;
}

// per-(cell, channel) sums over a block of kStatRows rows; grid (C/64, ceil(E/kStatRows)) (C = 32: grid.x = 1, half idle).
// Thread = 4 channels (float4) x one of 16 row lanes: a wave covers 4 rows x 64 channels per load, 1 KiB. Rows are sorted by
// cell, so a thread keeps one running segment and flushes it (float64 atomics) when the cell changes; a block inside ONE cell —
// nearly all of them — reduces its 16 row lanes through LDS first.
// MODE 0: acc[cell][0][c] += sum y, acc[cell][1][c] += sum y^2. MODE 1: dv = a > 0 ? d : 0: sum dv, sum dv * xhat
constexpr int kStatRows = 1024;
// The post-ReLU activation of a block's first layer is not stored by the second version: a == nullptr -> its sign is recomputed
// from y exactly as the fused operand loads compute it (pt_bn_relu: fma(y - mean, rg, beta)).
__device__ __forceinline__ float pt_bn_relu(float y, float m, float rg, float be) { return fmaxf(__fmaf_rn(__fsub_rn(y, m), rg, be), 0.f); }
template <int MODE, typename ST>
__global__ __launch_bounds__(256) void pt_bn_stats_kernel(const ST* __restrict__ y, const ST* __restrict__ d,
                                                          const ST* __restrict__ a, int C, size_t E,
                                                          const int32_t* __restrict__ row_cell, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, double* __restrict__ acc,
                                                          const float* __restrict__ rg, const float* __restrict__ beta) {
  __shared__ float4 r1[256], r2[256];
  const int c = blockIdx.x * 64 + 4 * (threadIdx.x & 15), g = threadIdx.x >> 4;
  const bool cok = c < C;
  const size_t lo = (size_t)blockIdx.y * kStatRows, hi = min(E, lo + kStatRows);
  const int cell_first = row_cell[lo], cell_last = row_cell[hi - 1];
  const bool one_cell = cell_first == cell_last;  // block-uniform
  int cur = cell_first;
  float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), rs = mu, s1 = mu, s2 = mu, rgv = mu, bev = mu;
  if (MODE == 1 && !a && cok) bev = *reinterpret_cast<const float4*>(beta + c);
  auto stat_of = [&](int cell) {
    if (MODE == 1 && cok) {
      mu = *reinterpret_cast<const float4*>(mean + (size_t)cell * C + c);
      rs = *reinterpret_cast<const float4*>(rstd + (size_t)cell * C + c);
      if (!a) rgv = *reinterpret_cast<const float4*>(rg + (size_t)cell * C + c);
    }
  };
  auto flush = [&](int cell) {
    double* p1 = acc + ((size_t)cell * 2) * 1024 + c;
    double* p2 = acc + ((size_t)cell * 2 + 1) * 1024 + c;
    atomicAdd(p1, (double)s1.x); atomicAdd(p1 + 1, (double)s1.y); atomicAdd(p1 + 2, (double)s1.z); atomicAdd(p1 + 3, (double)s1.w);
    atomicAdd(p2, (double)s2.x); atomicAdd(p2 + 1, (double)s2.y); atomicAdd(p2 + 2, (double)s2.z); atomicAdd(p2 + 3, (double)s2.w);
    s1 = s2 = make_float4(0.f, 0.f, 0.f, 0.f);
  };
  stat_of(cur);
  if (cok) {
#pragma unroll 2
    for (size_t row = lo + g; row < hi; row += 16) {
      if (!one_cell) {
        const int cell = row_cell[row];
        if (cell != cur) {
          flush(cur);
          cur = cell;
          stat_of(cur);
        }
      }
      const size_t i = row * C + c;
      const float4 v = pn_ld4(y + i);
      if (MODE == 0) {
        s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
        s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
      } else {
        const float4 dv = pn_ld4(d + i);
        const float4 av = a ? pn_ld4(a + i)
                            : make_float4(pt_bn_relu(v.x, mu.x, rgv.x, bev.x), pt_bn_relu(v.y, mu.y, rgv.y, bev.y),
                                          pt_bn_relu(v.z, mu.z, rgv.z, bev.z), pt_bn_relu(v.w, mu.w, rgv.w, bev.w));
        const float e0 = av.x > 0.f ? dv.x : 0.f, e1 = av.y > 0.f ? dv.y : 0.f, e2 = av.z > 0.f ? dv.z : 0.f, e3 = av.w > 0.f ? dv.w : 0.f;
        s1.x += e0; s1.y += e1; s1.z += e2; s1.w += e3;
        s2.x += e0 * (v.x - mu.x) * rs.x; s2.y += e1 * (v.y - mu.y) * rs.y; s2.z += e2 * (v.z - mu.z) * rs.z; s2.w += e3 * (v.w - mu.w) * rs.w;
      }
    }
  }
  if (!one_cell) {
    if (cok) flush(cur);
    return;
  }
  r1[threadIdx.x] = s1;
  r2[threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.x < 64) {  // thread t: channel blockIdx.x * 64 + t, summed over the 16 row lanes
    const int t = threadIdx.x, q = t >> 2, e = t & 3;
    if (blockIdx.x * 64 + t < C) {
      double t1 = 0.0, t2 = 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        t1 += (double)reinterpret_cast<const float*>(&r1[16 * j + q])[e];
        t2 += (double)reinterpret_cast<const float*>(&r2[16 * j + q])[e];
      }
      atomicAdd(acc + ((size_t)cur * 2) * 1024 + blockIdx.x * 64 + t, t1);
      atomicAdd(acc + ((size_t)cur * 2 + 1) * 1024 + blockIdx.x * 64 + t, t2);
    }
  }
}

// statistics of every cell + the running-statistics updates the reference performs once per cell, in cell order
// (rg != nullptr: also rg[cell][c] = rstd * gamma[c], the scale of the fused BatchNorm + ReLU operand loads — pt_bn_relu).
// Block = 32 channels x 8 cell lanes: the per-cell statistics are computed in parallel (chunks of 256 cells through LDS), the
// momentum recurrence then runs over the chunk from LDS (the first version walked the cells with two dependent float64 loads
// each: 45-65 us per launch, eight launches per forward).
__global__ __launch_bounds__(256) void pt_bn_finalize_kernel(const double* __restrict__ acc, const int32_t* __restrict__ cnt, int n_cells, int C,
                                                             float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ run_mean,
                                                             float* __restrict__ run_var, float momentum, const float* __restrict__ gamma,
                                                             float* __restrict__ rg) {
  __shared__ float sm[256][33], sv[256][33];
  const int cl = threadIdx.x & 31, lane8 = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
  const bool cok = c < C;
  const float ga = rg && cok ? gamma[c] : 0.f;
  float rm = 0.f, rv = 0.f;
  if (lane8 == 0 && cok) {
    rm = run_mean[c];
    rv = run_var[c];
  }
  for (int base = 0; base < n_cells; base += 256) {
    const int nc = min(256, n_cells - base);
    for (int j = lane8; j < nc; j += 8) {
      const int cell = base + j;
      float mf = 0.f, vu = 0.f;
      if (cok) {
        const double n = (double)max(cnt[cell], 1);
        const double m = acc[((size_t)cell * 2) * 1024 + c] / n;
        const double var = fmax(acc[((size_t)cell * 2 + 1) * 1024 + c] / n - m * m, 0.0);
        mf = (float)m;
        vu = (float)(var * (n / fmax(n - 1.0, 1.0)));
        mean[(size_t)cell * C + c] = mf;
        const float rs = 1.0f / sqrtf((float)var + kBnEps);
        rstd[(size_t)cell * C + c] = rs;
        if (rg) rg[(size_t)cell * C + c] = rs * ga;
      }
      sm[j][cl] = mf;
      sv[j][cl] = vu;
    }
    __syncthreads();
    if (lane8 == 0 && cok)
      for (int j = 0; j < nc; ++j) {
        rm = (1.f - momentum) * rm + momentum * sm[j][cl];
        rv = (1.f - momentum) * rv + momentum * sv[j][cl];
      }
    __syncthreads();
  }
  if (lane8 == 0 && cok) {
    run_mean[c] = rm;
    run_var[c] = rv;
  }
}

template <typename ST>
__global__ __launch_bounds__(256) void pt_bn_apply_fwd_kernel(const ST* __restrict__ y, size_t E, int C,
                                                              const int32_t* __restrict__ row_cell, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, ST* __restrict__ a) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;  // C is a power of two >= 32: four channels of one row
  if (i >= E * C) return;
  const int lg = __ffs(C) - 1;
  const size_t row = i >> lg;
  const int c = (int)(i & (size_t)(C - 1));
  const size_t sc = (size_t)row_cell[row] * C + c;
  const float4 v = pn_ld4(y + i), m = *reinterpret_cast<const float4*>(mean + sc),
               r = *reinterpret_cast<const float4*>(rstd + sc), ga = *reinterpret_cast<const float4*>(gamma + c),
               be = *reinterpret_cast<const float4*>(beta + c);
  float4 o;
  o.x = fmaxf((v.x - m.x) * r.x * ga.x + be.x, 0.f);
  o.y = fmaxf((v.y - m.y) * r.y * ga.y + be.y, 0.f);
  o.z = fmaxf((v.z - m.z) * r.z * ga.z + be.z, 0.f);
  o.w = fmaxf((v.w - m.w) * r.w * ga.w + be.w, 0.f);
  pn_st4(a + i, o);
}

// d (gradient w.r.t. the ReLU output) -> gradient w.r.t. the Linear output, in place. FROM_MAX (second layer of a block): the
// incoming gradient is not read from d but rebuilt from the max aggregation — row `arg[group][c]` receives dxout[group][c],
// every other row 0 — and the result is written to d.
// The BatchNorm backward's closed form per element is  dx = k0 dv - k1 - (y - mean) k2  with per-(cell, channel) coefficients
//   k0 = gamma rstd,  k1 = gamma rstd S1 / n,  k2 = gamma rstd^2 S2 / n   (S1 = sum dv, S2 = sum dv xhat over the cell's rows):
// pt_bn_bwd_finalize_kernel turns the float64 sums into the k1 / k2 tables ONCE per (cell, channel) — the element-wise pass used to
// read eight doubles and divide by n per thread — and adds the layer's gamma / beta gradients (sum over cells of S2 / S1) in the same
// launch (it replaces pt_bn_param_grad_kernel: a serial walk over the cells per channel, 25 us x 8 launches per step).
// Block = 32 channels x 8 cell lanes.
__global__ __launch_bounds__(256) void pt_bn_bwd_finalize_kernel(const double* __restrict__ acc, const int32_t* __restrict__ cnt, int n_cells, int C,
                                                                 const float* __restrict__ gamma, const float* __restrict__ rstd,
                                                                 float* __restrict__ k1, float* __restrict__ k2, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta) {
  __shared__ double r1[8][33], r2[8][33];
  const int cl = threadIdx.x & 31, lane8 = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
  const bool cok = c < C;
  const float ga = cok ? gamma[c] : 0.f;
  double t1 = 0.0, t2 = 0.0;
  if (cok)
    for (int cell = lane8; cell < n_cells; cell += 8) {
      const double s1 = acc[((size_t)cell * 2) * 1024 + c], s2 = acc[((size_t)cell * 2 + 1) * 1024 + c];
      const float n = (float)max(cnt[cell], 1), r = rstd[(size_t)cell * C + c];
      k1[(size_t)cell * C + c] = ga * r / n * (float)s1;
      k2[(size_t)cell * C + c] = ga * r / n * r * (float)s2;
      t1 += s1;
      t2 += s2;
    }
  r1[lane8][cl] = t1;
  r2[lane8][cl] = t2;
  __syncthreads();
  if (lane8 == 0 && cok) {
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      t1 += r1[j][cl];
      t2 += r2[j][cl];
    }
    dbeta[c] += (float)t1;
    dgamma[c] += (float)t2;
  }
}
// d (gradient w.r.t. the ReLU output) -> gradient w.r.t. the Linear output, in place: two forms. (The element-wise form of rounds 2-5 —
// one thread per 4 elements, every table and the group's arg / dxout loaded beside each of them — took 214 / 168 us per launch with bf16
// rows; the two below 110 / 158 us.)
// First layers, strip-wise: the iteration of pt_bn_stats_kernel — thread = 4 channels x one of 16 row lanes of a block
// of kStatRows rows; rows are sorted by cell, so the per-(cell, channel) tables are reloaded only when the cell changes (nearly all
// blocks lie inside one cell) instead of beside every element. grid (C/64, ceil(E/kStatRows)).
template <typename ST>
__global__ __launch_bounds__(256) void pt_bn_apply_bwd_rows_kernel(ST* __restrict__ d, const ST* __restrict__ a, const ST* __restrict__ y, int C,
                                                                   size_t E, const int32_t* __restrict__ row_cell, const float* __restrict__ k1,
                                                                   const float* __restrict__ k2, const float* __restrict__ gamma,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   const float* __restrict__ beta, const float* __restrict__ rg) {
  const int c = blockIdx.x * 64 + 4 * (threadIdx.x & 15), g = threadIdx.x >> 4;
  if (c >= C) return;
  const size_t lo = (size_t)blockIdx.y * kStatRows, hi = min(E, lo + kStatRows);
  const bool one_cell = row_cell[lo] == row_cell[hi - 1];  // block-uniform
  int cur = -1;
  float4 m, k0, q1, q2, g4;
  const float4 ga = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
  auto tables = [&](int cell) {
    const size_t sc = (size_t)cell * C + c;
    m = *reinterpret_cast<const float4*>(mean + sc);
    const float4 r = *reinterpret_cast<const float4*>(rstd + sc);
    k0 = make_float4(ga.x * r.x, ga.y * r.y, ga.z * r.z, ga.w * r.w);
    q1 = *reinterpret_cast<const float4*>(k1 + sc);
    q2 = *reinterpret_cast<const float4*>(k2 + sc);
    g4 = a ? k0 : *reinterpret_cast<const float4*>(rg + sc);
    cur = cell;
  };
  tables(row_cell[lo]);
#pragma unroll 2
  for (size_t row = lo + g; row < hi; row += 16) {
    if (!one_cell) {
      const int cell = row_cell[row];
      if (cell != cur) tables(cell);
    }
    const size_t i = row * C + c;
    const float4 yv = pn_ld4(y + i), dv4 = pn_ld4(d + i);
    const float4 av = a ? pn_ld4(a + i)
                        : make_float4(pt_bn_relu(yv.x, m.x, g4.x, be.x), pt_bn_relu(yv.y, m.y, g4.y, be.y), pt_bn_relu(yv.z, m.z, g4.z, be.z),
                                      pt_bn_relu(yv.w, m.w, g4.w, be.w));
    float4 o;
    o.x = k0.x * (av.x > 0.f ? dv4.x : 0.f) - q1.x - (yv.x - m.x) * q2.x;
    o.y = k0.y * (av.y > 0.f ? dv4.y : 0.f) - q1.y - (yv.y - m.y) * q2.y;
    o.z = k0.z * (av.z > 0.f ? dv4.z : 0.f) - q1.z - (yv.z - m.z) * q2.z;
    o.w = k0.w * (av.w > 0.f ? dv4.w : 0.f) - q1.w - (yv.w - m.w) * q2.w;
    pn_st4(d + i, o);
  }
}
// Second layers (FROM the max aggregation: the incoming gradient is not read from d but rebuilt — row `arg[group][c]` receives
// dxout[group][c], every other row 0 — and the result is written to d), group-wise: one thread per (group, 4 channels) walks the group's rows, as
// pt_segmax_kernel does in the forward. Everything that is per GROUP — the arg-max rows, the gradient arriving at them — and per
// (cell, channel) — a group lies inside one object, hence one cell — is loaded ONCE per thread instead of once per element (the
// element-wise form above issued 32 bytes of arg / dxout loads and six table loads beside every 8 bytes of y): per row it reads y and
// writes d. grid = ceil(G * C / 4 / 256).
template <typename ST>
__global__ __launch_bounds__(256) void pt_bn_apply_bwd_groups_kernel(ST* __restrict__ d, const ST* __restrict__ y, const int32_t* __restrict__ goff,
                                                                     size_t n_groups, int C, int nd, const int32_t* __restrict__ cell_of_obj,
                                                                     const float* __restrict__ k1, const float* __restrict__ k2,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                     const int32_t* __restrict__ arg, const float* __restrict__ dxout) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n_groups * C) return;
  const int lg = __ffs(C) - 1;  // (C is a power of two)
  const size_t g = i >> lg;
  const int c = (int)(i & (size_t)(C - 1));
  const size_t sc = (size_t)cell_of_obj[(unsigned)g / (unsigned)nd] * C + c;
  const float4 m = *reinterpret_cast<const float4*>(mean + sc), r = *reinterpret_cast<const float4*>(rstd + sc),
               ga = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c),
               q1 = *reinterpret_cast<const float4*>(k1 + sc), q2 = *reinterpret_cast<const float4*>(k2 + sc);
  const int4 ar = *reinterpret_cast<const int4*>(arg + i);
  const float4 dx = *reinterpret_cast<const float4*>(dxout + i);
  const float4 k0 = make_float4(ga.x * r.x, ga.y * r.y, ga.z * r.z, ga.w * r.w);
  const int lo = goff[g], hi = goff[g + 1];
#pragma unroll 4
  for (int row = lo; row < hi; ++row) {
    const size_t e = (size_t)row * C + c;
    const float4 yv = pn_ld4(y + e);
    float4 o;
#define T2L_PT_BWDG(X)                                                                         \
  {                                                                                            \
    const float av = (yv.X - m.X) * r.X * ga.X + be.X; /* the sign of the ReLU input, as the forward's max saw it */ \
    const float dv = (ar.X == row && av > 0.f) ? dx.X : 0.f;                                   \
    o.X = k0.X * dv - q1.X - (yv.X - m.X) * q2.X;                                              \
  }
    T2L_PT_BWDG(x) T2L_PT_BWDG(y) T2L_PT_BWDG(z) T2L_PT_BWDG(w)
#undef T2L_PT_BWDG
    pn_st4(d + e, o);
  }
}
// BatchNorm-backward sums of a block's SECOND layer, straight from the max aggregation: only the arg-max row of every
// (group, channel) carries a gradient, so sum dv and sum dv * xhat are sums over GROUPS (dv = dxout where the maximum is > 0;
// xhat from one gathered y per (group, channel)) — no pass over the edge rows. grid (ceil(C/64), ceil(G/256)).
__global__ __launch_bounds__(256) void pt_bn_stats_max_kernel(const float* __restrict__ ysel, const float* __restrict__ xout,
                                                              const float* __restrict__ dxout, size_t G, int C,
                                                              int nd, const int32_t* __restrict__ cell_of_obj,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              double* __restrict__ acc) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), gl = threadIdx.x >> 6;
  if (c >= C) return;
  const size_t lo = (size_t)blockIdx.y * 256, hi = min(G, lo + 256);
  int cur = -1;
  float mu = 0.f, rs = 0.f, s1 = 0.f, s2 = 0.f;
  for (size_t g = lo + gl; g < hi; g += 4) {
    const int cell = cell_of_obj[g / nd];
    if (cell != cur) {
      if (cur >= 0) {
        atomicAdd(acc + ((size_t)cur * 2) * 1024 + c, (double)s1);
        atomicAdd(acc + ((size_t)cur * 2 + 1) * 1024 + c, (double)s2);
      }
      s1 = s2 = 0.f;
      cur = cell;
      mu = mean[(size_t)cur * C + c];
      rs = rstd[(size_t)cur * C + c];
    }
    const size_t i = g * C + c;
    if (xout[i] > 0.f) {
      const float dv = dxout[i];
      s1 += dv;
      s2 += dv * (ysel[i] - mu) * rs;
    }
  }
  if (cur >= 0) {
    atomicAdd(acc + ((size_t)cur * 2) * 1024 + c, (double)s1);
    atomicAdd(acc + ((size_t)cur * 2 + 1) * 1024 + c, (double)s2);
  }
}
// second layer of a block: BatchNorm + ReLU applied on the fly, max over the rows of every group (first maximum wins, as
// argmax) + the winning row and its pre-BatchNorm value (ysel: the backward's statistics need y at the arg-max row and read it
// from here instead of gathering it) — the post-ReLU activations of the widest layer never exist in memory.
// One thread per (group, 4 channels): float4 loads, a wave covers 1 KiB of a row.
template <typename ST>
__global__ __launch_bounds__(256) void pt_segmax_kernel(const ST* __restrict__ y, const int32_t* __restrict__ goff, size_t n_groups, int C,
                                                        int nd, const int32_t* __restrict__ cell_of_obj, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ xout, int32_t* __restrict__ arg,
                                                        float* __restrict__ ysel) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n_groups * C) return;
  const int lg = __ffs(C) - 1;  // (C is a power of two)
  const size_t g = i >> lg;
  const int c = (int)(i & (size_t)(C - 1));
  const size_t sc = (size_t)cell_of_obj[(unsigned)g / (unsigned)nd] * C + c;
  const float4 m = *reinterpret_cast<const float4*>(mean + sc), r = *reinterpret_cast<const float4*>(rstd + sc),
               ga = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
  const int lo = goff[g], hi = goff[g + 1];
  float best[4] = {-1.f, -1.f, -1.f, -1.f}, ys[4] = {0.f, 0.f, 0.f, 0.f};  // ReLU outputs are >= 0 and every group has at least one row
  int br[4] = {-1, -1, -1, -1};
#pragma unroll 4
  for (int row = lo; row < hi; ++row) {
    const float4 yv = pn_ld4(y + (size_t)row * C + c);
    const float v[4] = {fmaxf((yv.x - m.x) * r.x * ga.x + be.x, 0.f), fmaxf((yv.y - m.y) * r.y * ga.y + be.y, 0.f),
                        fmaxf((yv.z - m.z) * r.z * ga.z + be.z, 0.f), fmaxf((yv.w - m.w) * r.w * ga.w + be.w, 0.f)};
    const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (v[j] > best[j]) {
        best[j] = v[j];
        br[j] = row;
        ys[j] = yy[j];
      }
  }
  *reinterpret_cast<float4*>(xout + i) = make_float4(fmaxf(best[0], 0.f), fmaxf(best[1], 0.f), fmaxf(best[2], 0.f), fmaxf(best[3], 0.f));
  *reinterpret_cast<int4*>(arg + i) = make_int4(br[0], br[1], br[2], br[3]);
  *reinterpret_cast<float4*>(ysel + i) = make_float4(ys[0], ys[1], ys[2], ys[3]);
}
template <typename ST>
__global__ void pt_slice_kernel(const ST* __restrict__ dX, size_t rows, int cin, int kp, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < rows * cin) out[i] = pn_ld1(dX + (i / cin) * kp + (i % cin));
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
  if (pt->iws) (void)hipFree(pt->iws);
  delete pt;
}

// dX[M,Kp] = dY[M,N] W[N,Kp]: W is transposed once (<= 2 MB) so that the weight fragments are k-contiguous float4 loads too
// (eight strided scalar loads per tile and step made this form 3 ms slower than the 32x32-tile kernel it replaces)
__global__ void pt_transpose_kernel(const float* __restrict__ W, int N, int Kp, float* __restrict__ Wt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < N * Kp) Wt[(size_t)(i % Kp) * N + i / Kp] = W[i];
}
// ---- second version: LDS-resident weights (gemm_rows2.h) ----------------------------------------------------------------
static int pn_cu_count() {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu < 8) n_cu = 256;
  }
  return n_cu;
}
template <int MODE, bool AF, bool ST, int TP>
static void rows2_launch_t(const train::Rows2Args& a, int grid, size_t lds, hipStream_t s) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(train::rows2_kernel<MODE, AF, ST, TP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            train::kRows2Lds);
  hipLaunchKernelGGL((train::rows2_kernel<MODE, AF, ST, TP>), dim3(grid), dim3(train::kRows2Threads), lds, s, a);
}
template <int MODE, bool AF, bool ST>
static void rows2_launch_tp(const train::Rows2Args& a, int grid, size_t lds, hipStream_t s) {
  if constexpr (MODE == 1) {  // bf16 operands: half the LDS bytes per tile and no low parts in registers — up to 8 tiles per pass
    switch (a.tp) {
      case 5: rows2_launch_t<MODE, AF, ST, 5>(a, grid, lds, s); return;
      case 6: rows2_launch_t<MODE, AF, ST, 6>(a, grid, lds, s); return;
      case 7: rows2_launch_t<MODE, AF, ST, 7>(a, grid, lds, s); return;
      case 8: rows2_launch_t<MODE, AF, ST, 8>(a, grid, lds, s); return;
      default: break;
    }
  }
  switch (a.tp) {
    case 1: rows2_launch_t<MODE, AF, ST, 1>(a, grid, lds, s); break;
    case 2: rows2_launch_t<MODE, AF, ST, 2>(a, grid, lds, s); break;
    case 3: rows2_launch_t<MODE, AF, ST, 3>(a, grid, lds, s); break;
    default: rows2_launch_t<MODE, AF, ST, 4>(a, grid, lds, s); break;
  }
}
template <int MODE>
static void rows2_launch_m(const train::Rows2Args& a, int grid, size_t lds, hipStream_t s) {
  const bool af = a.a_mean != nullptr, st = a.acc != nullptr;
  if (af && st) rows2_launch_tp<MODE, true, true>(a, grid, lds, s);
  else if (st) rows2_launch_tp<MODE, false, true>(a, grid, lds, s);
  else rows2_launch_tp<MODE, false, false>(a, grid, lds, s);  // (a fused operand without statistics does not occur)
}
// C[M,N] = f(A)[M,K] W[N,K]^T + bias; a_mean != nullptr: f = BatchNorm + ReLU of the layer below (tables [cell][K]); acc != nullptr:
// the BatchNorm partial sums of C per (cell, column). K is a multiple of 16, N of 32.
static void gemm_rows2(const float* A, const float* W, const float* bias, float* C, size_t M, int N, int K, const float* a_mean,
                       const float* a_rg, const float* a_beta, const int32_t* row_cell, double* acc, hipStream_t s,
                       const int32_t* scat_src = nullptr, float* scat_dst = nullptr, int scat_cols = 0) {
  const int mode = tl_gemm_bf16, bpe = mode == 1 ? 2 : 4;
  // (bf16: the fused operand transform's tables spill from 7 tiles on)
  const int maxt = mode == 1 ? (a_mean ? 6 : 8) : train::kRows2MaxT;
  int tp = std::max(1, std::min(std::min(maxt, N / 32), train::kRows2Lds / (32 * K * bpe)));
  tp = (N / 32 + (N / 32 + tp - 1) / tp - 1) / ((N / 32 + tp - 1) / tp);  // the same number of passes, balanced
  const size_t lds = (size_t)tp * 32 * K * bpe;
  const int grid = (int)std::min<size_t>((size_t)pn_cu_count(), (M + 255) / 256);
  const int rpw = (int)(((M + (size_t)grid * 8 - 1) / ((size_t)grid * 8) + 31) / 32 * 32);
  const train::Rows2Args a{A, W, bias, C, (int)M, N, K, K, K, N, tp, rpw, a_mean, a_rg, a_beta, row_cell, acc, scat_src, scat_dst, scat_cols};
  if (mode == 2) rows2_launch_m<2>(a, grid, lds, s);
  else if (mode == 1) rows2_launch_m<1>(a, grid, lds, s);
  else rows2_launch_m<0>(a, grid, lds, s);
}
template <int MODE, bool XF, int NT, int KT>
static void tn2_launch_t(const train::Tn2Args& a, dim3 grid, hipStream_t s) {
  constexpr size_t lds = 2 * (size_t)(8 / NT * 32) * (32 * NT + 4 + 32 * KT + 4) * sizeof(float);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(train::tn2_kernel<MODE, XF, NT, KT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  hipLaunchKernelGGL((train::tn2_kernel<MODE, XF, NT, KT>), grid, dim3(train::kRows2Threads), lds, s, a);
}
// the (n tiles, k tiles) block shapes the backbone's layers produce: dW2 [h2, h1] with the fused operand, dW1 [h1, kp] without
template <int MODE>
static bool tn2_launch_m(const train::Tn2Args& a, dim3 grid, int nt, int kt, bool xf, hipStream_t s) {
  if (xf) {
    if (nt == 2 && kt == 1) tn2_launch_t<MODE, true, 2, 1>(a, grid, s);
    else if (nt == 4 && kt == 4) tn2_launch_t<MODE, true, 4, 4>(a, grid, s);
    else if (nt == 8 && kt == 4) tn2_launch_t<MODE, true, 8, 4>(a, grid, s);
    else return false;
  } else {
    if (nt == 1 && kt == 1) tn2_launch_t<MODE, false, 1, 1>(a, grid, s);
    else if (nt == 4 && kt == 3) tn2_launch_t<MODE, false, 4, 3>(a, grid, s);
    else if (nt == 8 && kt == 3) tn2_launch_t<MODE, false, 8, 3>(a, grid, s);
    else if (nt == 8 && kt == 2) tn2_launch_t<MODE, false, 8, 2>(a, grid, s);
    else if (nt == 8 && kt == 4) tn2_launch_t<MODE, false, 8, 4>(a, grid, s);
    else return false;
  }
  return true;
}
// dW[N][ldw] += dY[M,N]^T f(X)[M,K] (columns k < k_real), db[n] += sum_m dY; x_mean != nullptr: f = BatchNorm + ReLU (tables [cell][K]).
// Blocks along K are launched per distinct block width (K = 160 -> one block of 3 tiles and one of 2).
static bool gemm_tn2(const float* dY, const float* X, float* dW, float* db, size_t M, int N, int K, int k_real, int ldw, const float* x_mean,
                     const float* x_rg, const float* x_beta, const int32_t* row_cell, hipStream_t s) {
  const int mode = tl_gemm_bf16;
  const int KT = K / 32, kblocks = (KT + train::kTn2MaxT - 1) / train::kTn2MaxT, kb_tiles = (KT + kblocks - 1) / kblocks;
  const int gy = (N + 255) / 256, nt = std::min(256, N) / 32, SR = 8 / nt * 32;
  const int last_tiles = KT - kb_tiles * (kblocks - 1);  // tiles of the last block along K (<= kb_tiles)
  bool ok = true;
  for (int part = 0; part < 2 && ok; ++part) {
    // part 0: the blocks [0, nfull) of kb_tiles tiles; part 1: a narrower last block, as its own launch
    const int nfull = last_tiles == kb_tiles ? kblocks : kblocks - 1;
    const int gz = part == 0 ? nfull : (last_tiles == kb_tiles ? 0 : 1);
    if (gz == 0) continue;
    const int kt = part == 0 ? kb_tiles : last_tiles;
    int gx = std::max(1, pn_cu_count() / (gy * gz));
    gx = (int)std::min<size_t>((size_t)gx, (M + SR - 1) / SR);
    const int rpw = (int)(((M + gx - 1) / gx + SR - 1) / SR * SR);
    // a narrower last block starts at column 32 * kb_tiles * nfull: shift the operand pointers, keep kb_tiles for the block stride
    const int k0 = part == 0 ? 0 : 32 * kb_tiles * nfull;
    // (X is an edge-row tensor: bf16 elements with bf16 operands — the column shift is in ITS elements)
    const float* Xs = mode == 1 ? reinterpret_cast<const float*>(reinterpret_cast<const train::pn_bf16*>(X) + k0) : X + k0;
    const train::Tn2Args a{dY, Xs, dW + k0, part == 0 ? db : nullptr, (int)M, N, K, N, K, ldw, rpw, kb_tiles, k_real - k0,
                           x_mean ? x_mean + k0 : nullptr, x_rg ? x_rg + k0 : nullptr, x_beta ? x_beta + k0 : nullptr, row_cell};
    const dim3 grid(gx, gy, gz);
    const bool xf = x_mean != nullptr;
    if (mode == 2) ok = tn2_launch_m<2>(a, grid, nt, kt, xf, s);
    else if (mode == 1) ok = tn2_launch_m<1>(a, grid, nt, kt, xf, s);
    else ok = tn2_launch_m<0>(a, grid, nt, kt, xf, s);
  }
  return ok;
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

// Every activation / table of the forward pass and the backward's scratch, laid out from pt->ws (E and G of every level are
// known: the index phase ran). Called twice: with pt->ws == nullptr to measure, then for real.
static size_t pn_layout(PnTrain* pt) {
  const int n_obj = pt->n_obj, n_cells = pt->n_cells;
  pt->ws_off = 0;
  pt->cell_of_obj = pn_bump<int32_t>(pt, n_obj);
  pt->cell_base = pn_bump<int32_t>(pt, n_obj);
  pt->acc = pn_bump<double>(pt, (size_t)n_cells * 2 * 1024);
  pt->bk1 = pn_bump<float>(pt, (size_t)n_cells * 1024);
  pt->bk2 = pn_bump<float>(pt, (size_t)n_cells * 1024);
  pt->wt = pn_bump<float>(pt, (size_t)1024 * 512);
  size_t scratch = 0;
  const size_t es = pt->half ? 2 : 4;  // bytes per element of an edge-row tensor (gemm_rows2.h: pn_store_t)
  auto edge = [&](size_t count) { return reinterpret_cast<float*>(pn_bump<char>(pt, count * es)); };
  for (int l = 0; l < 4; ++l) {
    PnLevel& L = pt->lv[l];
    L.cnt = pn_bump<int32_t>(pt, n_cells);
    L.goff = pn_bump<int32_t>(pt, L.G + 1);
    L.src = pn_bump<int32_t>(pt, L.E);
    L.row_group = pn_bump<int32_t>(pt, L.E);
    L.row_cell = pn_bump<int32_t>(pt, L.E);
    L.X = edge(L.E * L.kp);
    L.w1p = pn_bump<float>(pt, (size_t)L.h1 * L.kp);
    L.y1 = edge(L.E * L.h1);
    // second version: a1 is recomputed from y1 wherever it is consumed — except in the global MLP (45 k rows x 512: 92 MB), whose
    // wide layers would pay the fused operand transform once per column pass (K = 512 leaves two column tiles per pass)
    L.a1 = l == 3 ? edge(L.E * L.h1) : nullptr;
    L.rg1 = pn_bump<float>(pt, (size_t)n_cells * L.h1);
    L.y2 = edge(L.E * L.h2);
    L.mean1 = pn_bump<float>(pt, (size_t)n_cells * L.h1);
    L.rstd1 = pn_bump<float>(pt, (size_t)n_cells * L.h1);
    L.mean2 = pn_bump<float>(pt, (size_t)n_cells * L.h2);
    L.rstd2 = pn_bump<float>(pt, (size_t)n_cells * L.h2);
    L.xout = pn_bump<float>(pt, L.G * L.h2);
    L.arg = pn_bump<int32_t>(pt, L.G * L.h2);
    L.ysel = pn_bump<float>(pt, L.G * L.h2);
    // backward scratch of this level: dA2, dA1, dX (+ 256-byte roundings)
    scratch = std::max(scratch, L.E * (size_t)(L.h2 + L.h1 + L.kp) * es + 3 * 256);
  }
  pt->f1 = pn_bump<float>(pt, (size_t)n_obj * 512);
  pt->f2 = pn_bump<float>(pt, (size_t)n_obj * 256);
  // backward: d2, d1, two ping-pong buffers for the gradient w.r.t. a level's output (n_obj x 8192 floats covers all of
  // them: 1024 at the top, 32 x 256, 64 x 128, 128 x 64 below), then the level scratch
  pt->scratch_off = pt->ws_off;
  pt->scratch_bytes = (size_t)n_obj * (256 + 512 + 2 * 8192) * sizeof(float) + 4 * 256 + scratch;
  return pt->ws_off + pt->scratch_bytes;
}

// one get_mlp block in training mode over segmented rows: y = X W^T + b; per-cell BatchNorm; ReLU.
// Second version (default): the GEMM's epilogue forms the BatchNorm partial sums, and a block's first layer leaves only y1 — its
// BatchNorm + ReLU is applied by whoever loads it (x_fuse: the layer below's tables for THIS layer's left operand).
static void pn_block_fwd(TrainState* st, PnTrain* pt, const PnLevel& L, int layer, const float* X, const float* W, int K, int C, float* y,
                         float* a, float* mean, float* rstd, float* rg, bool x_fuse, hipStream_t s) {
  using namespace train;
  const std::string p = L.prefix + "." + std::to_string(layer);
  (void)hipMemsetAsync(pt->acc, 0, sizeof(double) * 2 * 1024 * pt->n_cells, s);
  gemm_rows2(X, W, T_(st, p + ".0.bias").data, y, L.E, C, K, x_fuse ? L.mean1 : nullptr, x_fuse ? L.rg1 : nullptr,
             x_fuse ? T_(st, L.prefix + ".0.1.bias").data : nullptr, L.row_cell, pt->acc, s);
  hipLaunchKernelGGL(pt_bn_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, s, (const double*)pt->acc, (const int32_t*)L.cnt, pt->n_cells, C,
                     mean, rstd, T_(st, p + ".1.running_mean").data, T_(st, p + ".1.running_var").data, 0.1f,
                     (const float*)T_(st, p + ".1.weight").data, rg);
  if (a) {
    if (pt->half)
      hipLaunchKernelGGL((pt_bn_apply_fwd_kernel<pn_bf16>), dim3(pn_blocks(L.E * C / 4)), dim3(256), 0, s, reinterpret_cast<const pn_bf16*>(y), L.E, C,
                         (const int32_t*)L.row_cell, (const float*)mean, (const float*)rstd, (const float*)T_(st, p + ".1.weight").data,
                         (const float*)T_(st, p + ".1.bias").data, reinterpret_cast<pn_bf16*>(a));
    else
      hipLaunchKernelGGL((pt_bn_apply_fwd_kernel<float>), dim3(pn_blocks(L.E * C / 4)), dim3(256), 0, s, (const float*)y, L.E, C,
                         (const int32_t*)L.row_cell, (const float*)mean, (const float*)rstd, (const float*)T_(st, p + ".1.weight").data,
                         (const float*)T_(st, p + ".1.bias").data, a);
  }
}

// d: gradient w.r.t. the block's ReLU output [E, C] (overwritten with the gradient w.r.t. the Linear output). dxout != nullptr
// (second layer): that gradient is implied by the max aggregation (dxout [G, C] at the arg-max rows) and d is only written.
static void pn_block_bwd(TrainState* st, PnTrain* pt, const PnLevel& L, int layer, float* d, const float* y, const float* a, int C,
                         const float* mean, const float* rstd, const float* rg, const float* dxout, hipStream_t s) {
  using namespace train;
  const std::string p = L.prefix + "." + std::to_string(layer);
  (void)hipMemsetAsync(pt->acc, 0, sizeof(double) * 2 * 1024 * pt->n_cells, s);
  int log2c = 0;
  while ((1 << log2c) < C) ++log2c;  // (C is 32 .. 1024, a power of two)
  // sums -> coefficient tables + gamma / beta gradients -> element-wise closed form
  if (dxout) {
    hipLaunchKernelGGL(pt_bn_stats_max_kernel, dim3((C + 63) / 64, (unsigned)((L.G + 255) / 256)), dim3(256), 0, s, (const float*)L.ysel,
                       (const float*)L.xout, dxout, L.G, C, L.nd, (const int32_t*)pt->cell_of_obj, mean, rstd, pt->acc);
  } else {
    const dim3 sgrid((C + 63) / 64, (unsigned)((L.E + kStatRows - 1) / kStatRows));
    if (pt->half)
      hipLaunchKernelGGL((pt_bn_stats_kernel<1, pn_bf16>), sgrid, dim3(256), 0, s, reinterpret_cast<const pn_bf16*>(y), reinterpret_cast<const pn_bf16*>(d),
                         reinterpret_cast<const pn_bf16*>(a), C, L.E, (const int32_t*)L.row_cell, mean, rstd, pt->acc, rg,
                         (const float*)T_(st, p + ".1.bias").data);
    else
      hipLaunchKernelGGL((pt_bn_stats_kernel<1, float>), sgrid, dim3(256), 0, s, y, (const float*)d, a, C, L.E, (const int32_t*)L.row_cell, mean, rstd,
                         pt->acc, rg, (const float*)T_(st, p + ".1.bias").data);
  }
  hipLaunchKernelGGL(pt_bn_bwd_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, s, (const double*)pt->acc, (const int32_t*)L.cnt, pt->n_cells, C,
                     (const float*)T_(st, p + ".1.weight").data, rstd, pt->bk1, pt->bk2, T_(st, p + ".1.weight").grad, T_(st, p + ".1.bias").grad);
  if (dxout) {
#define T2L_PN_BWD_GROUPS(ST_)                                                                                                                          \
  hipLaunchKernelGGL((pt_bn_apply_bwd_groups_kernel<ST_>), dim3(pn_blocks(L.G * C / 4)), dim3(256), 0, s, reinterpret_cast<ST_*>(d),                    \
                     reinterpret_cast<const ST_*>(y), (const int32_t*)L.goff, L.G, C, L.nd, (const int32_t*)pt->cell_of_obj, (const float*)pt->bk1,     \
                     (const float*)pt->bk2, (const float*)T_(st, p + ".1.weight").data, (const float*)T_(st, p + ".1.bias").data, mean, rstd,         \
                     (const int32_t*)L.arg, dxout)
    if (pt->half) T2L_PN_BWD_GROUPS(pn_bf16);
    else T2L_PN_BWD_GROUPS(float);
#undef T2L_PN_BWD_GROUPS
  } else {
    const dim3 agrid((C + 63) / 64, (unsigned)((L.E + kStatRows - 1) / kStatRows));
#define T2L_PN_BWD_ROWS(ST_)                                                                                                                          \
  hipLaunchKernelGGL((pt_bn_apply_bwd_rows_kernel<ST_>), agrid, dim3(256), 0, s, reinterpret_cast<ST_*>(d), reinterpret_cast<const ST_*>(a),        \
                     reinterpret_cast<const ST_*>(y), C, L.E, (const int32_t*)L.row_cell, (const float*)pt->bk1, (const float*)pt->bk2,              \
                     (const float*)T_(st, p + ".1.weight").data, mean, rstd, (const float*)T_(st, p + ".1.bias").data, rg)
    if (pt->half) T2L_PN_BWD_ROWS(pn_bf16);
    else T2L_PN_BWD_ROWS(float);
#undef T2L_PN_BWD_ROWS
  }
}

int pn_train_forward_impl(t2l_ctx* ctx, const float* pos, const float* rgb, const int32_t* cell_offsets, int n_cells, float* out_f2,
                          hipStream_t s) {
  using namespace train;
  TrainState* st = state(ctx);
  PnTrain* pt = st ? pn_state(st) : nullptr;
  if (!pt || !pt->bound)
    return fail(ctx, T2L_ESTATE, "t2l_pointnet_features_train: bind the object_encoder.pointnet.* tensors first (t2l_train_bind)");
  if (!pos || !rgb || !cell_offsets || n_cells <= 0 || !out_f2) return fail(ctx, T2L_EINVAL, "t2l_pointnet_features_train: bad arguments");
  const int n_obj = cell_offsets[n_cells];
  if (n_obj <= 0) return fail(ctx, T2L_EINVAL, "t2l_pointnet_features_train: no objects");
  for (int c = 0; c < n_cells; ++c)
    if (cell_offsets[c + 1] <= cell_offsets[c]) return fail(ctx, T2L_EINVAL, "t2l_pointnet_features_train: every cell needs at least one object");
  pt->have_forward = false;
  pt->n_obj = n_obj;
  pt->n_cells = n_cells;
  pt->pos0 = pos;
  pt->rgb0 = rgb;
  tl_gemm_bf16 = ctx->train_bf16;
  pt->half = ctx->train_bf16 == 1;  // bf16 operands -> the edge-row tensors live in memory as bf16 (gemm_rows2.h: pn_store_t)
  const std::string P = "object_encoder.pointnet.";
  const int ns[3] = {256, 128, 64}, cin[3] = {3, 64, 128}, h1[4] = {32, 128, 256, 512}, h2[4] = {64, 128, 256, 1024};
  const float radius[3] = {0.2f, 0.3f, 0.4f};
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
    L.G = (size_t)n_obj * L.nd;
    L.radius = l < 3 ? radius[l] : 0.f;
  }
  std::vector<int32_t> h_tab(2 * (size_t)n_obj);  // object -> cell, object -> first object of its cell
  for (int c = 0; c < n_cells; ++c)
    for (int o = cell_offsets[c]; o < cell_offsets[c + 1]; ++o) {
      h_tab[o] = c;
      h_tab[n_obj + o] = cell_offsets[c];
    }

  // ---- index phase: FPS + ball query of all three levels (positions only), rows per group back to the host
  size_t ineed = 256 * 8 + 2 * (size_t)n_obj * sizeof(int32_t);
  for (int l = 0; l < 3; ++l) ineed += pt->lv[l].G * (3 * sizeof(float) + 34 * sizeof(int32_t)) + 3 * 256;
  if (ineed > pt->iws_cap) {
    T2L_HIP(ctx, hipStreamSynchronize(s));
    if (pt->iws) (void)hipFree(pt->iws);
    pt->iws = nullptr;
    pt->iws_cap = 0;
    T2L_HIP(ctx, hipMalloc(&pt->iws, ineed));
    pt->iws_cap = ineed;
  }
  int32_t* i_cell_base;
  {
    size_t off = 0;
    auto take = [&](size_t bytes) {
      char* p = pt->iws + off;
      off += (bytes + 255) & ~(size_t)255;
      return p;
    };
    i_cell_base = reinterpret_cast<int32_t*>(take((size_t)n_obj * sizeof(int32_t)));
    for (int l = 0; l < 3; ++l) {
      PnLevel& L = pt->lv[l];
      L.pos_out = reinterpret_cast<float*>(take(L.G * 3 * sizeof(float)));
      L.nbr33 = reinterpret_cast<int32_t*>(take(L.G * 33 * sizeof(int32_t)));
      L.cnt_g = reinterpret_cast<int32_t*>(take(L.G * sizeof(int32_t)));
    }
  }
  T2L_HIP(ctx, hipMemcpyAsync(i_cell_base, h_tab.data() + n_obj, sizeof(int32_t) * n_obj, hipMemcpyHostToDevice, s));
  event_begin(ctx, "pointnet_train_index", s);
  {
    const float* cur_pos = pos;
    for (int l = 0; l < 3; ++l) {
      PnLevel& L = pt->lv[l];
      const float r2 = radius[l] * radius[l];  // float32 product, as the restatement
      const unsigned fg = (unsigned)((n_obj + 3) / 4), bg = (unsigned)((L.G + 3) / 4);
      if (L.ns == 256) {
        hipLaunchKernelGGL((pt_fps_kernel<4>), dim3(fg), dim3(256), 0, s, cur_pos, n_obj, L.nd, L.pos_out);
        hipLaunchKernelGGL((pt_ball_kernel<4>), dim3(bg), dim3(256), 0, s, cur_pos, (const float*)L.pos_out, n_obj, L.nd, r2,
                           (const int32_t*)i_cell_base, ctx->pn_self_loops, L.nbr33, L.cnt_g);
      } else if (L.ns == 128) {
        hipLaunchKernelGGL((pt_fps_kernel<2>), dim3(fg), dim3(256), 0, s, cur_pos, n_obj, L.nd, L.pos_out);
        hipLaunchKernelGGL((pt_ball_kernel<2>), dim3(bg), dim3(256), 0, s, cur_pos, (const float*)L.pos_out, n_obj, L.nd, r2,
                           (const int32_t*)i_cell_base, ctx->pn_self_loops, L.nbr33, L.cnt_g);
      } else {
        hipLaunchKernelGGL((pt_fps_kernel<1>), dim3(fg), dim3(256), 0, s, cur_pos, n_obj, L.nd, L.pos_out);
        hipLaunchKernelGGL((pt_ball_kernel<1>), dim3(bg), dim3(256), 0, s, cur_pos, (const float*)L.pos_out, n_obj, L.nd, r2,
                           (const int32_t*)i_cell_base, ctx->pn_self_loops, L.nbr33, L.cnt_g);
      }
      cur_pos = L.pos_out;
    }
  }
  event_end(ctx, "pointnet_train_index", s);
  std::vector<std::vector<int32_t>> h_goff(3), h_cnt(4, std::vector<int32_t>(n_cells, 0));
  for (int l = 0; l < 3; ++l) {
    h_goff[l].resize(pt->lv[l].G + 1);
    T2L_HIP(ctx, hipMemcpyAsync(h_goff[l].data() + 1, pt->lv[l].cnt_g, sizeof(int32_t) * pt->lv[l].G, hipMemcpyDeviceToHost, s));
  }
  T2L_HIP(ctx, hipStreamSynchronize(s));
  for (int l = 0; l < 3; ++l) {
    PnLevel& L = pt->lv[l];
    std::vector<int32_t>& g = h_goff[l];
    g[0] = 0;
    long long total = 0;
    for (size_t i = 0; i < L.G; ++i) {
      const int c = g[i + 1];
      if (c < 1 || c > 33) return fail(ctx, T2L_EHIP, "t2l_pointnet_features_train: ball query returned an impossible row count");
      h_cnt[l][h_tab[i / L.nd]] += c;
      total += c;
      g[i + 1] = (int32_t)total;
    }
    if (total > 0x3fffffffll) return fail(ctx, T2L_EINVAL, "t2l_pointnet_features_train: batch too large (more than 2^30 edge rows in one level)");
    L.E = (size_t)total;
  }
  pt->lv[3].E = (size_t)n_obj * 32;
  for (int c = 0; c < n_cells; ++c) h_cnt[3][c] = (cell_offsets[c + 1] - cell_offsets[c]) * 32;

  // ---- activations: exact sizes
  char* keep = pt->ws;
  pt->ws = nullptr;
  const size_t need = pn_layout(pt);
  pt->ws = keep;
  if (need > pt->ws_cap) {
    if (pt->ws) (void)hipFree(pt->ws);
    pt->ws = nullptr;
    pt->ws_cap = 0;
    T2L_HIP(ctx, hipMalloc(&pt->ws, need));
    pt->ws_cap = need;
  }
  (void)pn_layout(pt);
  T2L_HIP(ctx, hipMemcpyAsync(pt->cell_of_obj, h_tab.data(), sizeof(int32_t) * n_obj, hipMemcpyHostToDevice, s));
  T2L_HIP(ctx, hipMemcpyAsync(pt->cell_base, h_tab.data() + n_obj, sizeof(int32_t) * n_obj, hipMemcpyHostToDevice, s));
  for (int l = 0; l < 4; ++l) {
    T2L_HIP(ctx, hipMemcpyAsync(pt->lv[l].cnt, h_cnt[l].data(), sizeof(int32_t) * n_cells, hipMemcpyHostToDevice, s));
    if (l < 3)
      T2L_HIP(ctx, hipMemcpyAsync(pt->lv[l].goff, h_goff[l].data(), sizeof(int32_t) * (pt->lv[l].G + 1), hipMemcpyHostToDevice, s));
  }

  event_begin(ctx, "pointnet_train_forward", s);
  const float* cur_pos = pos;
  const float* cur_x = rgb;
  for (int l = 0; l < 4; ++l) {
    PnLevel& L = pt->lv[l];
    if (L.sa)
      hipLaunchKernelGGL(pt_compact_kernel, dim3(pn_blocks(L.G * 33)), dim3(256), 0, s, (const int32_t*)L.nbr33, (const int32_t*)L.goff, L.G,
                         L.nd, (const int32_t*)pt->cell_of_obj, L.src, L.row_group, L.row_cell);
    else
      hipLaunchKernelGGL(pt_iota_rows_kernel, dim3(pn_blocks(L.E + 1)), dim3(256), 0, s, L.E, 32, (const int32_t*)pt->cell_of_obj, L.src,
                         L.row_group, L.row_cell, L.goff);
    if (pt->half)
      hipLaunchKernelGGL((pt_gather_kernel<pn_bf16>), dim3(pn_blocks(L.E * (L.kp / 4))), dim3(256), 0, s, cur_x, cur_pos,
                         L.sa ? (const float*)L.pos_out : nullptr, (const int32_t*)L.src, (const int32_t*)L.row_group, L.E, L.cin, L.kp,
                         reinterpret_cast<pn_bf16*>(L.X));
    else
      hipLaunchKernelGGL((pt_gather_kernel<float>), dim3(pn_blocks(L.E * (L.kp / 4))), dim3(256), 0, s, cur_x, cur_pos,
                         L.sa ? (const float*)L.pos_out : nullptr, (const int32_t*)L.src, (const int32_t*)L.row_group, L.E, L.cin, L.kp, L.X);
    hipLaunchKernelGGL(pt_pad_kernel, dim3(pn_blocks((size_t)L.h1 * L.kp)), dim3(256), 0, s, (const float*)T_(st, L.prefix + ".0.0.weight").data,
                       L.h1, L.kin, L.kp, L.w1p);
    pn_block_fwd(st, pt, L, 0, L.X, L.w1p, L.kp, L.h1, L.y1, L.a1, L.mean1, L.rstd1, L.rg1, false, s);
    pn_block_fwd(st, pt, L, 1, L.a1 ? L.a1 : L.y1, T_(st, L.prefix + ".1.0.weight").data, L.h1, L.h2, L.y2, nullptr, L.mean2, L.rstd2, nullptr,
                 L.a1 == nullptr, s);
#define T2L_PN_SEGMAX(ST_)                                                                                                                        \
  hipLaunchKernelGGL((pt_segmax_kernel<ST_>), dim3(pn_blocks(L.G * L.h2 / 4)), dim3(256), 0, s, reinterpret_cast<const ST_*>(L.y2),               \
                     (const int32_t*)L.goff, L.G, L.h2, L.nd, (const int32_t*)pt->cell_of_obj, (const float*)L.mean2, (const float*)L.rstd2,       \
                     (const float*)T_(st, L.prefix + ".1.1.weight").data, (const float*)T_(st, L.prefix + ".1.1.bias").data, L.xout, L.arg, L.ysel)
    if (pt->half) T2L_PN_SEGMAX(pn_bf16);
    else T2L_PN_SEGMAX(float);
#undef T2L_PN_SEGMAX
    if (L.sa) {
      cur_pos = L.pos_out;
      cur_x = L.xout;
    }
  }
  pt->f0 = pt->lv[3].xout;
  gemm_nt(pt->f0, T_(st, P + "lin1.weight").data, T_(st, P + "lin1.bias").data, pt->f1, n_obj, 512, 1024, 1, s);
  gemm_nt(pt->f1, T_(st, P + "lin2.weight").data, T_(st, P + "lin2.bias").data, pt->f2, n_obj, 256, 512, 1, s);
  T2L_HIP(ctx, hipMemcpyAsync(out_f2, pt->f2, sizeof(float) * (size_t)n_obj * 256, hipMemcpyDeviceToDevice, s));
  event_end(ctx, "pointnet_train_forward", s);
  T2L_HIP(ctx, hipGetLastError());
  T2L_HIP(ctx, hipStreamSynchronize(s));  // the host tables above go out of scope
  pt->have_forward = true;
  return T2L_OK;
}

int pn_train_backward_impl(t2l_ctx* ctx, const float* grad_f2, hipStream_t s) {
  using namespace train;
  TrainState* st = state(ctx);
  PnTrain* pt = st ? pn_state(st) : nullptr;
  if (!pt || !pt->have_forward) return fail(ctx, T2L_ESTATE, "t2l_pointnet_backward: no training-mode forward to differentiate");
  st->pn_touched = true;  // the backbone's gradients carry this batch: its tensors take part in the next t2l_adam_step
  if (!pt->trainable)
    return fail(ctx, T2L_ESTATE, "t2l_pointnet_backward: the backbone was bound without gradient buffers (frozen)");
  if (!grad_f2) return fail(ctx, T2L_EINVAL, "t2l_pointnet_backward: null gradient");
  const int n_obj = pt->n_obj;
  if ((ctx->train_bf16 == 1) != pt->half)
    return fail(ctx, T2L_ESTATE, "t2l_pointnet_backward: option train_bf16 changed between the forward and the backward (the saved edge rows are "
                                 "bf16 exactly when the forward ran with train_bf16 = 1)");
  tl_gemm_bf16 = ctx->train_bf16;
  const size_t es = pt->half ? 2 : 4;
  const std::string P = "object_encoder.pointnet.";
  event_begin(ctx, "pointnet_train_backward", s);
  pt->ws_off = pt->scratch_off;
  float* d2 = pn_bump<float>(pt, (size_t)n_obj * 256);
  float* d1 = pn_bump<float>(pt, (size_t)n_obj * 512);
  float* dx = pn_bump<float>(pt, (size_t)n_obj * 8192);  // gradient w.r.t. the current level's output ...
  float* dx_next = pn_bump<float>(pt, (size_t)n_obj * 8192);  // ... and w.r.t. the output of the level below
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
    float* dA2 = reinterpret_cast<float*>(pn_bump<char>(pt, L.E * L.h2 * es));  // (edge-row tensors: pn_store_t)
    float* dA1 = reinterpret_cast<float*>(pn_bump<char>(pt, L.E * L.h1 * es));
    pn_block_bwd(st, pt, L, 1, dA2, L.y2, nullptr, L.h2, L.mean2, L.rstd2, nullptr, dx, s);
    const float* be1 = T_(st, L.prefix + ".0.1.bias").data;
    {  // a1 = relu(bn(y1)) is rebuilt while y1 is staged
      if (!gemm_tn2(dA2, L.a1 ? L.a1 : L.y1, T_(st, L.prefix + ".1.0.weight").grad, T_(st, L.prefix + ".1.0.bias").grad, L.E, L.h2, L.h1, L.h1,
                    L.h1, L.a1 ? nullptr : L.mean1, L.a1 ? nullptr : L.rg1, L.a1 ? nullptr : be1, L.row_cell, s))
        return fail(ctx, T2L_EHIP, "t2l_pointnet_backward: no tn2_kernel instance for this layer shape (internal error)");
      hipLaunchKernelGGL(pt_transpose_kernel, dim3((unsigned)((L.h2 * L.h1 + 255) / 256)), dim3(256), 0, s,
                         (const float*)T_(st, L.prefix + ".1.0.weight").data, L.h2, L.h1, pt->wt);
      gemm_rows2(dA2, pt->wt, nullptr, dA1, L.E, L.h1, L.h2, nullptr, nullptr, nullptr, nullptr, nullptr, s);
    }
    pn_block_bwd(st, pt, L, 0, dA1, L.y1, L.a1, L.h1, L.mean1, L.rstd1, L.rg1, nullptr, s);
    {  // straight into the unpadded gradient: the padding columns of X are not written
      if (!gemm_tn2(dA1, L.X, T_(st, L.prefix + ".0.0.weight").grad, T_(st, L.prefix + ".0.0.bias").grad, L.E, L.h1, L.kp, L.kin, L.kin,
                    nullptr, nullptr, nullptr, nullptr, s))
        return fail(ctx, T2L_EHIP, "t2l_pointnet_backward: no tn2_kernel instance for this layer shape (internal error)");
    }
    if (l > 0) {  // the input gradient: features of the level below (positions are data)
      const PnLevel& Lb = pt->lv[l - 1];
      const size_t nprev = Lb.G * Lb.h2;
      if (L.sa) {  // the product's epilogue scatters (atomics through src): no [E, kp] gradient, no scatter launch
        T2L_HIP(ctx, hipMemsetAsync(dx_next, 0, sizeof(float) * nprev, s));
        hipLaunchKernelGGL(pt_transpose_kernel, dim3((unsigned)((L.h1 * L.kp + 255) / 256)), dim3(256), 0, s, (const float*)L.w1p, L.h1, L.kp,
                           pt->wt);
        // (only the tiles that hold feature columns: N = cin rounded up to 32)
        gemm_rows2(dA1, pt->wt, nullptr, nullptr, L.E, (L.cin + 31) / 32 * 32, L.h1, nullptr, nullptr, nullptr, nullptr, nullptr, s,
                   (const int32_t*)L.src, dx_next, L.cin);
      } else {
        float* dX = reinterpret_cast<float*>(pn_bump<char>(pt, L.E * L.kp * es));  // (the global level: no sampling, the rows ARE the level below's)
        hipLaunchKernelGGL(pt_transpose_kernel, dim3((unsigned)((L.h1 * L.kp + 255) / 256)), dim3(256), 0, s, (const float*)L.w1p, L.h1, L.kp,
                           pt->wt);
        gemm_rows2(dA1, pt->wt, nullptr, dX, L.E, L.kp, L.h1, nullptr, nullptr, nullptr, nullptr, nullptr, s);
        if (pt->half)
          hipLaunchKernelGGL((pt_slice_kernel<pn_bf16>), dim3(pn_blocks(L.E * L.cin)), dim3(256), 0, s, reinterpret_cast<const pn_bf16*>(dX), L.E, L.cin,
                             L.kp, dx_next);
        else
          hipLaunchKernelGGL((pt_slice_kernel<float>), dim3(pn_blocks(L.E * L.cin)), dim3(256), 0, s, (const float*)dX, L.E, L.cin, L.kp, dx_next);
      }
      std::swap(dx, dx_next);
    }
    if (pt->ws_off > pt->scratch_off + pt->scratch_bytes || pt->ws_off > pt->ws_cap)
      return fail(ctx, T2L_ENOMEM, "t2l_pointnet_backward: scratch bound exceeded (internal error)");
    pt->ws_off = lmark;
  }
  event_end(ctx, "pointnet_train_backward", s);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

}  // namespace t2l
