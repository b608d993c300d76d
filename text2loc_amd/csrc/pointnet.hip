// PointNet++ object backbone, eval mode: models/pointcloud/pointnet2.py:18-100 —
//   3 x SetAbstraction (FPS 1/2, ball query r = .2/.3/.4 <= 32 neighbours, PointConv = max over get_mlp(cat[x_j, pos_j-pos_i]))
//   -> GlobalAbstraction (get_mlp([259,512,1024]) + max over the 32 remaining points) -> lin1/lin2 + ReLU -> features2.
// PARITY UNPINNED (third-party torch_geometric / torch-cluster arithmetic is absent from the reference tree): the
// semantics are the deterministic ones spelled out in oracle/t2l_oracle_pointnet.py, which these kernels are tested
// against. gfx950 only. The edge MLPs and the global MLP run as split-f16 MFMAs (mfma_h3.h: hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_f16, ~5e-7 relative, a fifth of the matrix-pipe time of v_mfma_f32_32x32x2_f32). There is no
// normalisation between the levels, so no static bound on the activations is useful: the split kernels watch the
// magnitude of everything that enters a split product and flag the object when it reaches 3e4 (or is not finite); the
// all-f32-MFMA launch that follows every split launch recomputes exactly the flagged objects (flags are sticky down the
// levels). Option encoder_f32 forces the f32 kernels.
//
// Level 1 (6 -> 32 -> 64 over 256 points) and the all-f32 kernels: one workgroup per object and level (pn_sa_kernel). FPS runs
// on one wave (DPP all-reduces, no LDS traffic in the loop). Every centre's ball query is a wave-level ballot compaction
// ("first 32 in index order"). The two-layer edge MLP of a centre is one 32-row tile (its <= 32 neighbours; empty slots repeat
// neighbour 0, harmless under max): layer 1 is computed TRANSPOSED (features x neighbours) so that its MFMA output registers
// ARE the A operand of layer 2 — no LDS round trip between the layers; BatchNorm and the Linear bias are folded on the host
// (bias rides on a constant-1 input column), weights are pre-packed in operand order so every wave-level weight load is one
// coalesced 1 KiB line out of L2.
// Levels 2 and 3 and the global MLP on split / plain f16 (92 % of the arithmetic): pn_fps_kernel + pn_sa_ws_kernel +
// pn_self_ws_kernel + pn_ga_ws_kernel — workgroups whose waves run their tiles in lock step over ONE stream of packed weights
// in LDS (WStream, below).
#include <math.h>
#include <string.h>

#include "t2l_internal.h"
#include "gemm_f32.h"
#include "mfma32.h"
#include "mfma_h3.h"

namespace t2l {

using train::f32x16;


struct PointNetWeights {
  uint4 *w1h[3] = {}, *w2h[3] = {};  // split-f16 packings of the same matrices (level 1; the other levels read streams)
  uint4* gash = nullptr;  // the global MLP's stream (same form)
  uint4* wsh[3] = {};  // levels 2 and 3: the fragments of both layers in the order a tile consumes them (WStream, pack_sa_stream)
  float4 *w1[3] = {}, *w2[3] = {};  // SA levels, packed
  float* b2[3] = {};
  float4 *ga1 = nullptr, *ga2 = nullptr;
  float* gab2 = nullptr;
  float *lin1w = nullptr, *lin1b = nullptr, *lin2w = nullptr, *lin2b = nullptr;
  // workspace
  char* ws = nullptr;
  size_t ws_cap = 0;
};

void free_pointnet(t2l_ctx* ctx) {
  PointNetWeights* P = reinterpret_cast<PointNetWeights*>(ctx->pn);
  if (!P) return;
  for (int l = 0; l < 3; ++l)
    for (void* p : {(void*)P->w1[l], (void*)P->w2[l], (void*)P->b2[l], (void*)P->w1h[l], (void*)P->w2h[l], (void*)P->wsh[l]})
      if (p) (void)hipFree(p);
  for (void* p : {(void*)P->gash, (void*)P->ga1, (void*)P->ga2, (void*)P->gab2, (void*)P->lin1w, (void*)P->lin1b, (void*)P->lin2w, (void*)P->lin2b, (void*)P->ws})
    if (p) (void)hipFree(p);
  delete P;
  ctx->pn = nullptr;
}

// ---------------------------------------------------------------------------------------------------------------
// host: fold BatchNorm, pack
// ---------------------------------------------------------------------------------------------------------------
using WMap = std::unordered_map<std::string, const t2l_weight_desc*>;

// get_mlp block i of `prefix`: returns W' [cout][cin] and b' [cout] with eval-mode BatchNorm folded in
static bool fold_block(const WMap& m, const std::string& prefix, int i, int cin, int cout, std::vector<float>& W, std::vector<float>& b) {
  auto get = [&](const std::string& k, int64_t n) -> const float* {
    auto it = m.find(prefix + "." + std::to_string(i) + k);
    return (it == m.end() || it->second->numel != n) ? nullptr : it->second->data;
  };
  const float *w = get(".0.weight", (int64_t)cin * cout), *bb = get(".0.bias", cout), *g = get(".1.weight", cout), *be = get(".1.bias", cout),
              *rm = get(".1.running_mean", cout), *rv = get(".1.running_var", cout);
  if (!w || !bb || !g || !be || !rm || !rv) return false;
  W.assign((size_t)cin * cout, 0.f);
  b.assign(cout, 0.f);
  for (int o = 0; o < cout; ++o) {
    const float s = g[o] / sqrtf(rv[o] + 1e-5f);
    for (int k = 0; k < cin; ++k) W[(size_t)o * cin + k] = w[(size_t)o * cin + k] * s;
    b[o] = (bb[o] - rm[o]) * s + be[o];
  }
  return true;
}

// layer 2 of an SA block, B operand: [h2/32][h1/32][4][64] float4: lane (n, kh): W[nt*32+n][ft*32 + 8*rq + 4*kh + 0..3]
static std::vector<float> pack_sa_l2(const std::vector<float>& W, int h2, int h1) {
  std::vector<float> out((size_t)h2 * h1);
  for (int nt = 0; nt < h2 / 32; ++nt)
    for (int ft = 0; ft < h1 / 32; ++ft)
      for (int rq = 0; rq < 4; ++rq)
        for (int lane = 0; lane < 64; ++lane)
          for (int c = 0; c < 4; ++c)
            out[((((size_t)nt * (h1 / 32) + ft) * 4 + rq) * 64 + lane) * 4 + c] =
                W[(size_t)(nt * 32 + (lane & 31)) * h1 + ft * 32 + 8 * rq + 4 * (lane >> 5) + c];
  return out;
}

// layer 2 of an SA block for the split-f16 path, B operand: [h2/32][h1/32][2][64 lanes][hi 16 B | lo 16 B]. The A operand of
// MFMA m of feature tile ft is registers 8m..8m+7 of the layer-1 accumulator, i.e. features (r&3) + 8(r>>2) + 4 kh of the tile
// for lane half kh: lane (n, kh) holds W[nt*32+n][ft*32 + that feature], r = 8m + e.
static std::vector<float> pack_sa_l2_h(const std::vector<float>& W, int h2, int h1) {
  std::vector<float> p((size_t)h2 * h1);
  uint16_t* out = reinterpret_cast<uint16_t*>(p.data());
  for (int nt = 0; nt < h2 / 32; ++nt)
    for (int ft = 0; ft < h1 / 32; ++ft)
      for (int mm = 0; mm < 2; ++mm)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const int r = 8 * mm + e, f = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float w = W[(size_t)(nt * 32 + (lane & 31)) * h1 + ft * 32 + f];
            const _Float16 hi = (_Float16)w;
            const _Float16 lo = (_Float16)(w - (float)hi);
            const size_t base = ((((size_t)nt * (h1 / 32) + ft) * 2 + mm) * 64 + lane) * 16;
            memcpy(out + base + e, &hi, 2);
            memcpy(out + base + 8 + e, &lo, 2);
          }
  return p;
}

template <typename T>
static int upload(t2l_ctx* ctx, T** dst, const std::vector<float>& v) {
  T2L_HIP(ctx, hipMalloc(dst, v.size() * sizeof(float)));
  T2L_HIP(ctx, hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  return T2L_OK;
}

constexpr int kCin[3] = {3, 64, 128}, kH1[3] = {32, 128, 256}, kH2[3] = {64, 128, 256};
constexpr int kWsChunk = 8;  // WStream: packed-weight steps (64 lanes x [hi 16 B | lo 16 B] = 2 KiB) per chunk of the LDS ring

// The stream of an SA level for WStream: layer 1 k-step major — step (st, ft) = fragment st of feature tile ft, so that one
// input fragment feeds the FT feature tiles back to back — padded with zero steps to a whole chunk, then layer 2 as pack_sa_l2_h
// has it ([output tile][feature tile][2]).
static std::vector<float> pack_sa_stream(const std::vector<float>& l1h, const std::vector<float>& l2h, int ft_tiles, int steps) {
  std::vector<float> out;
  out.reserve(l1h.size() + l2h.size() + kWsChunk * 512);
  for (int st = 0; st < steps; ++st)
    for (int ft = 0; ft < ft_tiles; ++ft) {
      const float* src = l1h.data() + ((size_t)ft * steps + st) * 512;  // one step = 512 floats
      out.insert(out.end(), src, src + 512);
    }
  out.resize((out.size() / 512 + kWsChunk - 1) / kWsChunk * kWsChunk * 512, 0.f);
  out.insert(out.end(), l2h.begin(), l2h.end());
  return out;
}
constexpr int k1p(int cin) { return ((cin + 4 + 7) / 8) * 8; }  // [x | pos_j - pos_i | 1 | 0...] padded to 8
constexpr int k1ph(int cin) { return ((cin + 4 + 15) / 16) * 16; }  // the same row padded to 16 (split-f16 steps of 16)

// Returns T2L_OK with ctx->pn == nullptr when the state_dict carries no PointNet++ tensors at all.
int pointnet_load_impl(t2l_ctx* ctx, const t2l_weight_desc* w, int n) {
  free_pointnet(ctx);
  const std::string p = "object_encoder.pointnet.";
  WMap m;
  for (int i = 0; i < n; ++i)
    if (w[i].name && !strncmp(w[i].name, p.c_str(), p.size())) m[w[i].name] = &w[i];
  if (m.empty()) return T2L_OK;
  PointNetWeights* P = new PointNetWeights();
  ctx->pn = P;
  int rc;
  std::vector<float> W, b;
  for (int l = 0; l < 3; ++l) {
    const std::string pre = p + "sa" + std::to_string(l + 1) + ".point_conv.local_nn";
    if (!fold_block(m, pre, 0, kCin[l] + 3, kH1[l], W, b)) return fail(ctx, T2L_EINVAL, "t2l_load_weights: incomplete " + pre + ".0");
    if ((rc = upload(ctx, &P->w1[l], pack_half_split(W, &b, kH1[l], kCin[l] + 3, k1p(kCin[l]))))) return rc;
    const std::vector<float> l1h = pack_split_f16(W.data(), b.data(), kH1[l], kCin[l] + 3, k1ph(kCin[l]));
    if (kCin[l] < 64 && (rc = upload(ctx, &P->w1h[l], l1h))) return rc;
    if (!fold_block(m, pre, 1, kH1[l], kH2[l], W, b)) return fail(ctx, T2L_EINVAL, "t2l_load_weights: incomplete " + pre + ".1");
    if ((rc = upload(ctx, &P->w2[l], pack_sa_l2(W, kH2[l], kH1[l])))) return rc;
    const std::vector<float> l2h = pack_sa_l2_h(W, kH2[l], kH1[l]);
    if (kCin[l] < 64 && (rc = upload(ctx, &P->w2h[l], l2h))) return rc;
    if (kCin[l] >= 64 && (rc = upload(ctx, &P->wsh[l], pack_sa_stream(l1h, l2h, kH1[l] / 32, k1ph(kCin[l]) / 16)))) return rc;
    if ((rc = upload(ctx, &P->b2[l], b))) return rc;
  }
  if (!fold_block(m, p + "ga.mlp", 0, 259, 512, W, b)) return fail(ctx, T2L_EINVAL, "t2l_load_weights: incomplete " + p + "ga.mlp.0");
  if ((rc = upload(ctx, &P->ga1, pack_half_split(W, &b, 512, 259, 264)))) return rc;
  const std::vector<float> ga1h = pack_split_f16(W.data(), b.data(), 512, 259, 272);
  if (!fold_block(m, p + "ga.mlp", 1, 512, 1024, W, b)) return fail(ctx, T2L_EINVAL, "t2l_load_weights: incomplete " + p + "ga.mlp.1");
  if ((rc = upload(ctx, &P->gash, pack_sa_stream(ga1h, pack_sa_l2_h(W, 1024, 512), 512 / 32, 272 / 16)))) return rc;
  if ((rc = upload(ctx, &P->ga2, pack_half_split(W, nullptr, 1024, 512, 512)))) return rc;
  if ((rc = upload(ctx, &P->gab2, b))) return rc;
  auto raw = [&](const char* name, int64_t numel, float** dst) -> int {
    auto it = m.find(p + name);
    if (it == m.end() || it->second->numel != numel) return fail(ctx, T2L_EINVAL, std::string("t2l_load_weights: missing ") + p + name);
    return upload(ctx, dst, std::vector<float>(it->second->data, it->second->data + numel));
  };
  if ((rc = raw("lin1.weight", 512 * 1024, &P->lin1w)) || (rc = raw("lin1.bias", 512, &P->lin1b)) ||
      (rc = raw("lin2.weight", 256 * 512, &P->lin2w)) || (rc = raw("lin2.bias", 256, &P->lin2b)))
    return rc;
  return T2L_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float d2_noFMA(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}
template <int CTRL>
__device__ __forceinline__ unsigned pn_dpp(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
// all-reduce over the 64 lanes; OP(a,b) on unsigned
#define PN_WAVE_REDUCE(NAME, OP)                                                                  \
  __device__ __forceinline__ unsigned NAME(unsigned v) {                                          \
    v = OP(v, pn_dpp<0xB1>(v));                                                                   \
    v = OP(v, pn_dpp<0x4E>(v));                                                                   \
    v = OP(v, pn_dpp<0x141>(v));                                                                  \
    v = OP(v, pn_dpp<0x140>(v));                                                                  \
    {                                                                                             \
      const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);                        \
      v = OP(r[0], r[1]);                                                                         \
    }                                                                                             \
    {                                                                                             \
      const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);                        \
      v = OP(r[0], r[1]);                                                                         \
    }                                                                                             \
    return v;                                                                                     \
  }
__device__ __forceinline__ unsigned pn_umax(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned pn_umin(unsigned a, unsigned b) { return a < b ? a : b; }
PN_WAVE_REDUCE(wave_umax, pn_umax)
PN_WAVE_REDUCE(wave_umin, pn_umin)

// Edge MLP of one 32-row tile. xrow[e] = v[kh*K1P/2 + e] of this lane's row (lane&31), v = [x | dpos | 1 | 0].
// Calls emit(nt, acc) for each 32-column tile of the layer-2 output (pre-bias): acc[r] = row (r&3)+8(r>>2)+4(lane>>5), col lane&31.
template <int CIN, int H1, int H2, typename Emit>
__device__ __forceinline__ void sa_mlp_tile(const float (&xrow)[k1p(CIN) / 2], const float4* __restrict__ w1, const float4* __restrict__ w2,
                                            int lane, Emit&& emit) {
  constexpr int K1P = k1p(CIN), S4 = K1P / 8, FT = H1 / 32, NT = H2 / 32;
  // One wave per SIMD (the register file holds h1 + the input row), so the L2 latency of the weight stream is hidden by an
  // explicit ring of PF float4 loads in flight (PF x 4 MFMAs = PF x 256 cycles of lookahead), not by other waves.
  constexpr int PF = 4;
  f32x16 h1[FT];
  {
    constexpr int STEPS = FT * S4;
    float4 ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = w1[(i < STEPS ? i : 0) * 64 + lane];
    f32x16 acc;
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
      const int ft = step / S4, s4 = step % S4;
      if (s4 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      }
      const float4 w = ring[step % PF];
      if (step + PF < STEPS) ring[step % PF] = w1[(step + PF) * 64 + lane];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, xrow[4 * s4 + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, xrow[4 * s4 + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, xrow[4 * s4 + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, xrow[4 * s4 + 3], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);  // keep the ring's distance: no further hoisting of loads (register pressure)
      if (s4 == S4 - 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) h1[ft][r] = fmaxf(acc[r], 0.f);  // BatchNorm + bias folded; ReLU
      }
    }
  }
#pragma unroll 1
  for (int nt = 0; nt < NT; ++nt) {  // rolled: h1 (up to 128 VGPRs) + the input row already fill most of the register file
    constexpr int STEPS = FT * 4;
    const float4* wp = w2 + (size_t)nt * STEPS * 64 + lane;
    float4 ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = wp[(i < STEPS ? i : 0) * 64];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
      const int ft = step / 4, rq = step % 4;
      const float4 w = ring[step % PF];
      if (step + PF < STEPS) ring[step % PF] = wp[(step + PF) * 64];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h1[ft][4 * rq + 0], w.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h1[ft][4 * rq + 1], w.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h1[ft][4 * rq + 2], w.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h1[ft][4 * rq + 3], w.w, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    emit(nt, acc);
  }
}

// 8 register values -> split fragment; amax tracks the largest magnitude that entered a split product
template <bool SG = false>  // SG: plain f16 (option encoder_f16): the high part only
__device__ __forceinline__ HFrag split_vals(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7,
                                            float& amax) {
  // max(amax, |a|, |b|) as ONE v_max3 (fmaxf(fabsf()) chains quiet every operand first: 12 ops per 8 values instead of 4). The
  // operands are VALU or load results at every call site, never an MFMA's destination (the compiler does not pad asm reads).
  auto amax3 = [](float m, float a, float b) {
    float d;
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(d) : "v"(m), "v"(a), "v"(b));
    return d;
  };
  amax = amax3(amax3(amax3(amax3(amax, v0, v1), v2, v3), v4, v5), v6, v7);
  const h3_f32x8 v = {v0, v1, v2, v3, v4, v5, v6, v7};
  HFrag f;
  f.hi = __builtin_convertvector(v, h3_f16x8);
  if constexpr (SG) {
    f.lo = f.hi;  // (never read)
  } else {  // v - hi (exact in f32) as ONE v_fma_mix_f32 per value, reading the packed half in place: no unpacking conversions
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    const u32x4v p = __builtin_bit_cast(u32x4v, f.hi);
    h3_f32x8 d;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d[2 * e]) : "v"(p[e]), "v"(v[2 * e]));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d[2 * e + 1]) : "v"(p[e]), "v"(v[2 * e + 1]));
    }
    f.lo = __builtin_convertvector(d, h3_f16x8);
  }
  return f;
}
// max of the 16 accumulator registers of a lane in 12 VALU ops (fmaxf quiets every operand first: 31). The first level is
// v_med3(a, b, +inf) from the builtin with an opaque +inf — the compiler sees these reads of the MFMA's result and pads the
// hazard; instructions from inline asm it does not pad (tools/mfma_hazard_scan.py found v_max3 one wait state short) — the
// rest is v_max3 from asm on those results.
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float max16(const f32x16& a) {
  float pinf = __builtin_inff();
  asm volatile("" : "+v"(pinf));
  float t[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = __builtin_amdgcn_fmed3f(a[2 * i], a[2 * i + 1], pinf);
  return vmax3(vmax3(t[0], t[1], t[2]), vmax3(t[3], t[4], t[5]), vmax3(t[6], t[7], t[7]));
}
// The steps of the two layers as template recursions (the compiler does not unroll a 72-step loop with this body, and
// run-time indices would put the fragment arrays into scratch memory).
template <int STEP, int STEPS, int S, int PF, int FT, bool SG>
__device__ __forceinline__ void sa_l1_steps(HFrag (&ring)[PF], const uint4*& wp, f32x16& acc, const HFrag (&xf)[S], HFrag (&hf)[FT][2],
                                            float& amax) {
  if constexpr (STEP < STEPS) {
    constexpr int ft = STEP / S, st = STEP % S;
    if constexpr (st == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    }
    const HFrag wf = ring[STEP % PF];
    if constexpr (STEP + PF < STEPS) {
      ring[STEP % PF] = load_h1<SG>(wp);
      wp += 128;
      asm volatile("" : "+v"(wp));
    }
    mfma_h3<SG>(acc, wf, xf[st]);
    __builtin_amdgcn_sched_barrier(0);  // keep the ring's distance: no further hoisting of loads (register pressure)
    if constexpr (st == S - 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);  // BatchNorm + bias folded; ReLU
      hf[ft][0] = split_vals<SG>(acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7], amax);
      hf[ft][1] = split_vals<SG>(acc[8], acc[9], acc[10], acc[11], acc[12], acc[13], acc[14], acc[15], amax);
    }
    sa_l1_steps<STEP + 1, STEPS, S, PF, FT, SG>(ring, wp, acc, xf, hf, amax);
  }
}
template <int STEP, int STEPS, int PF, int FT, bool SG>
__device__ __forceinline__ void sa_l2_steps(HFrag (&ring)[PF], const uint4*& wp, f32x16& acc, const HFrag (&hf)[FT][2], bool not_last_tile) {
  if constexpr (STEP < STEPS) {
    const HFrag wf = ring[STEP % PF];
    if (not_last_tile || STEP + PF < STEPS) {  // the stream continues into the next output tile
      ring[STEP % PF] = load_h1<SG>(wp);
      wp += 128;
      asm volatile("" : "+v"(wp));
    }
    mfma_h3<SG>(acc, hf[STEP / 2][STEP % 2], wf);
    __builtin_amdgcn_sched_barrier(0);
    sa_l2_steps<STEP + 1, STEPS, PF, FT, SG>(ring, wp, acc, hf, not_last_tile);
  }
}

// sa_mlp_tile on split-f16 MFMAs. xrow: this lane's half of the row padded to k1ph(CIN). Layer 1 transposed (A = packed
// weights, B = the row fragments, split once per tile); its accumulator registers 0..7 / 8..15 are the two A fragments of
// layer 2 for that feature tile (pack_sa_l2_h orders the weights to match), split once and reused by all output tiles.
template <int CIN, int H1, int H2, bool SG, typename Emit>
__device__ __forceinline__ void sa_mlp_tile_h(const float (&xrow)[k1ph(CIN) / 2], const uint4* __restrict__ w1, const uint4* __restrict__ w2,
                                              int lane, float& amax, Emit&& emit) {
  constexpr int S = k1ph(CIN) / 16, FT = H1 / 32, NT = H2 / 32;
  HFrag xf[S];
#pragma unroll
  for (int s = 0; s < S; ++s)
    xf[s] = split_vals<SG>(xrow[8 * s], xrow[8 * s + 1], xrow[8 * s + 2], xrow[8 * s + 3], xrow[8 * s + 4], xrow[8 * s + 5], xrow[8 * s + 6],
                       xrow[8 * s + 7], amax);
  // One wave per SIMD: the L2 latency of the weight stream is hidden by an explicit ring of PF fragment pairs in flight
  // (PF x 3 MFMAs = PF x 96 cycles of look-ahead). The pointers run (opaque increments): with `base + constant` addressing
  // the compiler materialises one 64-bit address per load (2 KiB apart, beyond the immediate offset) and hoists hundreds.
  HFrag hf[FT][2];
  {
    constexpr int STEPS = FT * S, PF = STEPS < 4 ? STEPS : 4;
    const uint4* wp = w1 + (size_t)lane * 2;
    asm volatile("" : "+v"(wp));
    HFrag ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      ring[i] = load_h1<SG>(wp);
      wp += 128;
      asm volatile("" : "+v"(wp));
    }
    f32x16 acc;
    sa_l1_steps<0, STEPS, S, PF, FT, SG>(ring, wp, acc, xf, hf, amax);
  }
  {
    constexpr int STEPS = FT * 2, PF = STEPS < 4 ? STEPS : 4;  // per output tile; STEPS is a multiple of PF
    const uint4* wp = w2 + (size_t)lane * 2;
    asm volatile("" : "+v"(wp));
    HFrag ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      ring[i] = load_h1<SG>(wp);
      wp += 128;
      asm volatile("" : "+v"(wp));
    }
#pragma unroll 1
    for (int nt = 0; nt < NT; ++nt) {  // rolled: the split h1 (up to 128 VGPRs) + the row fragments already fill most of the file
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      sa_l2_steps<0, STEPS, PF, FT, SG>(ring, wp, acc, hf, nt + 1 < NT);
      emit(nt, acc);
    }
  }
}

// this lane's half of the input row [x(CIN) | dpos(3) | 1 | 0...]
template <int CIN, int HALF>
__device__ __forceinline__ void build_xrow(const float* __restrict__ x, float dx, float dy, float dz, int kh, float (&xrow)[HALF]) {
#pragma unroll
  for (int e = 0; e < HALF; ++e) {
    const int k = kh * HALF + e;
    float v = 0.f;
    if (k < CIN) v = x[k];
    else if (k == CIN) v = dx;
    else if (k == CIN + 1) v = dy;
    else if (k == CIN + 2) v = dz;
    else if (k == CIN + 3) v = 1.f;
    xrow[e] = v;
  }
}

// farthest point sampling of one object on the calling wave: selection order from point 0, lowest index on ties 
template <int NS>
__device__ __forceinline__ void fps_wave(const float* spos, int* sel, int lane) {
  constexpr int ND = NS / 2, PPL = NS / 64;
  float mind[PPL], px[PPL], py[PPL], pz[PPL];
#pragma unroll
  for (int q = 0; q < PPL; ++q) {
    const int p = lane + 64 * q;
    px[q] = spos[p * 3]; py[q] = spos[p * 3 + 1]; pz[q] = spos[p * 3 + 2];
    mind[q] = 3.0e38f;
  }
  int last = 0;
  if (lane == 0) sel[0] = 0;
  for (int t = 1; t < ND; ++t) {
    const float cx = spos[last * 3], cy = spos[last * 3 + 1], cz = spos[last * 3 + 2];
    float best = -1.f;
    int bi = 0;
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
      mind[q] = fminf(mind[q], d2_noFMA(px[q], py[q], pz[q], cx, cy, cz));
      if (mind[q] > best) { best = mind[q]; bi = lane + 64 * q; }
    }
    const unsigned m = wave_umax(__float_as_uint(best));  // distances are >= 0: the bit pattern orders like the value
    last = (int)wave_umin(__float_as_uint(best) == m ? (unsigned)bi : 0x7fffffffu);
    if (lane == 0) sel[t] = last;
  }
}

// The centres of a level for all (unflagged) objects, one wave per object. Inside the per-object kernels the sampling loop (ND
// dependent steps of two wave reductions) runs on ONE wave: at level 1 (127 steps over 4 points per lane) that wave's SIMD carried
// 25 us of sampling per object on top of its 18 us share of the tiles (three workgroups per CU, all sampling on their wave 0); at
// level 2 the CU's only workgroup waited 8 of 84 us per object.
template <int NS>
__global__ __launch_bounds__(256) void pn_fps_kernel(const float* __restrict__ src_pos, float* __restrict__ dst_pos,
                                                     const int32_t* __restrict__ obj_flags, int n_obj) {
  constexpr int ND = NS / 2;
  __shared__ float spos_all[4][NS * 3];
  __shared__ int sel_all[4][ND];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, o = blockIdx.x * 4 + w;
  if (o >= n_obj || (obj_flags && obj_flags[o])) return;  // (no workgroup barrier below: the waves are independent)
  float* spos = spos_all[w];
  int* sel = sel_all[w];
  const float* gp = src_pos + (size_t)o * NS * 3;
  for (int i = lane; i < NS * 3; i += 64) spos[i] = gp[i];
  fps_wave<NS>(spos, sel, lane);
  for (int i = lane; i < ND * 3; i += 64) dst_pos[(size_t)o * ND * 3 + i] = spos[sel[i / 3] * 3 + (i % 3)];
}

struct SaParams {
  const float* src_pos;  // [n_obj][NS][3]
  const float* src_x;    // [n_obj][NS][CIN]
  float* dst_pos;        // [n_obj][ND][3]
  float* dst_x;          // [n_obj][ND][H2]
  const int32_t* cell_base;  // [n_obj]: first object of the object's cell (PyG batch)
  const float4* w1;
  const float4* w2;
  const float* b2;
  float r2;
  int self_loops;
  const uint4* w1h;
  const uint4* w2h;
  const uint4* wsh;  // levels 2 and 3: the tile's stream for WStream
  int32_t* obj_flags;
  int centres_given = 0;  // pn_sa_kernel: dst_pos already holds this level's centres (pn_fps_kernel)  // [n_obj]: 1 = this object's magnitudes left the split-f16 range (or were not finite): f32 launches only
};

template <int CIN, int H1, int H2, int NS, int H>
__global__ __launch_bounds__(256, 1) void pn_sa_kernel(SaParams P) {
  constexpr int ND = NS / 2, XS = CIN + 4, PPL = NS / 64;
  constexpr int HALF = (H ? k1ph(CIN) : k1p(CIN)) / 2;
  if (P.obj_flags) {  // split launch: skip flagged objects; f32 launch: only flagged objects
    const int flagged = P.obj_flags[blockIdx.x];
    if (H ? flagged : !flagged) return;
  }
  float amax = 0.f;
  extern __shared__ float smem[];
  float* spos = smem;                 // [NS][3]
  float* sx = spos + NS * 3;          // [NS][XS]
  float* dpos = sx + NS * XS;         // [ND][3]
  float* selfm = dpos + ND * 3;       // [ND][H2]
  int* sel = reinterpret_cast<int*>(selfm + ND * H2);  // [ND]
  int* nbr = sel + ND;                // [4 waves][32]
  const int o = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 31, kh = lane >> 5;
  const float* gp = P.src_pos + (size_t)o * NS * 3;
  const float* gx = P.src_x + (size_t)o * NS * CIN;
  for (int i = tid; i < NS * 3; i += 256) spos[i] = gp[i];
  for (int i = tid; i < NS * CIN; i += 256) sx[(i / CIN) * XS + (i % CIN)] = gx[i];
  __syncthreads();

  // ---- farthest point sampling (wave 0) unless pn_fps_kernel ran
  if (P.centres_given) {
    for (int i = tid; i < ND * 3; i += 256) dpos[i] = P.dst_pos[(size_t)o * ND * 3 + i];
  } else {
    if (w == 0) fps_wave<NS>(spos, sel, lane);
    __syncthreads();
    for (int i = tid; i < ND * 3; i += 256) {
      const float v = spos[sel[i / 3] * 3 + (i % 3)];
      dpos[i] = v;
      P.dst_pos[(size_t)o * ND * 3 + i] = v;
    }
  }
  __syncthreads();

  // ---- the extra (k -> k) messages of PyG's add_self_loops on the bipartite batch: centre k of the CELL's batch also
  // hears source node k of the cell's batch (oracle/t2l_oracle_pointnet.py)
  if (P.self_loops) {
    const int cb = P.cell_base[o];
    for (int tt = w; tt < ND / 32; tt += 4) {
      const int t = tt * 32 + j;
      const size_t k = (size_t)(o - cb) * ND + t;  // node index inside the cell's batch
      const float* sp = P.src_pos + ((size_t)cb * NS + k) * 3;
      float xrow[HALF];
      build_xrow<CIN, HALF>(P.src_x + ((size_t)cb * NS + k) * CIN, sp[0] - dpos[t * 3], sp[1] - dpos[t * 3 + 1], sp[2] - dpos[t * 3 + 2], kh, xrow);
      auto keep = [&](int nt, const f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) selfm[(tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * H2 + nt * 32 + j] = acc[r];
      };
      if constexpr (H != 0) sa_mlp_tile_h<CIN, H1, H2, H == 2>(xrow, P.w1h, P.w2h, lane, amax, keep);
      else sa_mlp_tile<CIN, H1, H2>(xrow, P.w1, P.w2, lane, keep);
    }
  } else {
    for (int i = tid; i < ND * H2; i += 256) selfm[i] = -3.0e38f;
  }
  __syncthreads();

  // ---- one 32-row tile per centre: ball query (first 32 in index order), edge MLP, max
  for (int t = w; t < ND; t += 4) {
    const float cx = dpos[t * 3], cy = dpos[t * 3 + 1], cz = dpos[t * 3 + 2];
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
      const int p = lane + 64 * q;
      const bool in = d2_noFMA(spos[p * 3], spos[p * 3 + 1], spos[p * 3 + 2], cx, cy, cz) < P.r2;
      const unsigned long long mask = __ballot(in);
      const int rank = cnt + __popcll(mask & ((1ull << lane) - 1ull));
      if (in && rank < 32) nbr[w * 32 + rank] = p;
      cnt += __popcll(mask);
    }
    cnt = min(cnt, 32);  // >= 1: the centre is one of the source points
    const int nb = nbr[w * 32 + (j < cnt ? j : 0)];
    float xrow[HALF];
    build_xrow<CIN, HALF>(sx + nb * XS, spos[nb * 3] - cx, spos[nb * 3 + 1] - cy, spos[nb * 3 + 2] - cz, kh, xrow);
    auto pool = [&](int nt, const f32x16& acc) {
      float m = max16(acc);
      m = fmaxf(m, __shfl_xor(m, 32));
      const int c = nt * 32 + j;
      if (kh == 0) P.dst_x[((size_t)o * ND + t) * H2 + c] = fmaxf(fmaxf(m, selfm[t * H2 + c]) + P.b2[c], 0.f);
    };
    if constexpr (H != 0) sa_mlp_tile_h<CIN, H1, H2, H == 2>(xrow, P.w1h, P.w2h, lane, amax, pool);
    else sa_mlp_tile<CIN, H1, H2>(xrow, P.w1, P.w2, lane, pool);
  }
  if constexpr (H != 0) {  // anything at or beyond 3e4 (or NaN: the comparison fails) entered a split product: hand the object over
    if (!(amax < kSplitF16Safe)) P.obj_flags[o] = 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Levels 2 and 3 on split / plain f16: pn_sa_ws_kernel (the centres' tiles) + pn_self_ws_kernel (the self-loop tiles).
//
// What limited the one-wave-per-SIMD form above on these levels (r05, 10,698 objects, dev builds and per-workgroup stamps,
// tools/pn_exp.sh / pn_stamps.py): (1) every wave pulled a tile's packed weights (104 KB / 400 KB) out of the L2 through a
// 4-deep register ring — 384 cycles of look-ahead against an L2 round trip of 500+; (2) with one wave per SIMD the VALU work
// of a tile (the compiler re-split the input row's fragments at every k-step to save registers: 24 VALU per 3 MFMAs, then the
// ReLU / split of layer 1 and the max pooling of layer 2) and every stall sat in line with the MFMAs; (3) a ninth round of
// four tiles per object for ONE self-loop tile. The chip is power-limited on dense split-f16 MFMAs (~21 ns per 32x32x16 MFMA
// and SIMD, 0.6 of the nominal rate: tools/pair_probe.hip), so 19,800 MFMAs per SIMD-object of level 3 cannot take less than
// 4.3 ms; they took 7.2.
//
// Here a workgroup has EIGHT waves, two per SIMD (<= 256 registers each), one tile each, and runs them in lock step over ONE
// stream of packed weights in LDS: a ring of kWsBufs chunks of kWsChunk steps filled by LDS-DMA (each wave issues its
// rows of a chunk in one burst behind the chunk's barrier — spreading them over the steps measured slower — no registers,
// look-ahead two chunks), fragments read with immediate offsets through a kWsDepth-deep register ring. Layer 1 runs k-step
// major: ONE input fragment (8 values of the lane's row half, read from LDS and split when its step comes: 9 / 5 splits per
// tile instead of 72 / 20) feeds the FT feature tiles' accumulators back to back (FT independent MFMA chains); their
// registers turn into the split A fragments of layer 2 in place. The self-loop messages are ordinary tiles of their own
// launch (eight per workgroup round, any object), which also finishes the level's output: the centres' kernel leaves the raw
// maxima, the self kernel applies max(self) + bias + ReLU. No ninth round, no 32 KB of self messages in LDS. (Measured and
// dropped: the other order — self messages first, picked up by the centres' kernel through registers or through 32 KB of LDS —
// saves 0.1 ms in the self kernels and costs the centres' kernels 0.15-0.6 ms: vector loads in flight behind a DMA burst make the
// stream's `s_waitcnt vmcnt(rows)` wait for that burst too, and the LDS copy lengthens every object's prologue.)
// ---------------------------------------------------------------------------------------------------------------
#ifndef T2L_WS_DEPTH
#define T2L_WS_DEPTH 2
#endif
constexpr int kWsBufs = 4, kWsDepth = T2L_WS_DEPTH, kWsWaves = 8;
typedef const __attribute__((address_space(3))) uint4* lds_u4;
__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p; }
__device__ __forceinline__ int uniform_wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
// global_load_lds_dwordx4: 64 lanes x 16 B to M0 + 16 lane (search_dev.h: lds_dma_row — from inline asm, so that later LDS
// reads do not wait for vmcnt(0); the stream orders DMA against its reads itself)
__device__ __forceinline__ void ws_dma_row(unsigned lds_row_addr, unsigned lane_off, const void* row_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_row_addr), "v"(lane_off), "s"(row_base)
               : "memory");
}

template <bool SG, int NW = kWsWaves>  // NW waves consume (and fetch) the stream
struct WStream {
  static constexpr int kStepB = SG ? 1024 : 2048;  // LDS bytes per step: [64 x hi 16 B][64 x lo 16 B] (plain f16: hi only)
  static constexpr int kChunkB = kWsChunk * kStepB, kRpw = kWsChunk * (SG ? 1 : 2) / NW;  // DMA rows per wave and chunk
  // this wave's rows of a chunk: rows wave + NW i = (step, half) pairs in the split form, steps in the plain form
  static constexpr int kRowDst = SG ? NW * 1024 : (NW / 2) * 2048, kRowSrc = SG ? NW * 2048 : (NW / 2) * 2048;
  const char* src;     // wave-uniform: the level's packed stream (2 KiB per step: lane l at 32 l = [hi | lo])
  unsigned ring_addr;  // LDS byte address of the ring
  unsigned lane32;
  int wave, cpr, total;   // chunks per round, chunks this workgroup consumes
  int g, sc;              // chunk being consumed; source chunk (inside the round) of the next one to issue
  lds_u4 ring, cur, nxt;  // lane-offset pointers: ring start, chunk g, chunk g + 1

  __device__ __forceinline__ void issue(int c, int srcchunk) const {
    const unsigned dst = ring_addr + (unsigned)(c & (kWsBufs - 1)) * kChunkB + (SG ? wave * 1024 : (wave >> 1) * 2048 + (wave & 1) * 1024);
    const char* sp = src + (size_t)srcchunk * (kWsChunk * 2048) + (SG ? wave * 2048 : (wave >> 1) * 2048 + (wave & 1) * 16);
#pragma unroll
    for (int i = 0; i < kRpw; ++i) ws_dma_row(dst + i * kRowDst, lane32, sp + i * kRowSrc);
  }
  __device__ __forceinline__ lds_u4 buf(int c) const { return ring + (c & (kWsBufs - 1)) * (kChunkB / 16); }
  // the first three chunks: as early in the kernel as possible
  __device__ __forceinline__ void open(const void* stream, float* ring_lds, int lane, int rounds, int chunks_per_round) {
    src = reinterpret_cast<const char*>(stream);
    ring_addr = lds_addr_of(ring_lds);
    ring = (lds_u4)(const __attribute__((address_space(3))) char*)reinterpret_cast<const char*>(ring_lds) + lane;
    lane32 = lane * 32;
    wave = uniform_wave_id();
    cpr = chunks_per_round;
    total = rounds * chunks_per_round;
    g = -1;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (c < total) issue(c, c);
    sc = 3;  // (cpr > 3)
    nxt = buf(0);
    cur = nxt;
  }
  // chunk g is consumed (or nothing yet): chunk g + 1 becomes current. After it, chunks <= g + 1 (new g) have landed for EVERY
  // wave (loads retire in order: all but the newest burst — anything else this wave has in flight is newer still and only
  // makes the wait longer), and every wave is done with chunk g - 1, whose buffer takes chunk g + 3.
  __device__ __forceinline__ void boundary() {
    g += 1;
    if (g + 2 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kRpw) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (s_nop: the epilogue of a tile pass reads the last MFMA's accumulator right behind this block; on the short path
    // through these branches the compiler's hazard padding came out one or two wait states short — tools/mfma_hazard_scan.py)
    asm volatile("s_nop 3\n\ts_barrier" ::: "memory");
    if (g + 3 < total) {
      issue(g + 3, sc);
      sc = sc + 1 == cpr ? 0 : sc + 1;
    }
    cur = nxt;
    nxt = buf(g + 1);
  }
  template <bool NEXT, int OFF>
  __device__ __forceinline__ HFrag read() const {
    const lds_u4 p = NEXT ? nxt : cur;
    HFrag f;
    f.hi = __builtin_bit_cast(h3_f16x8, p[OFF * (kStepB / 16)]);
    if constexpr (SG) f.lo = f.hi;
    else f.lo = __builtin_bit_cast(h3_f16x8, p[OFF * (kStepB / 16) + 64]);
    return f;
  }
  __device__ __forceinline__ void start(HFrag (&wr)[kWsDepth]) {  // chunks 0 and 1 in LDS, the register ring filled
    // the kernels' prologues store their points / features to LDS right before this: the first barrier publishes them, so this
    // wave's LDS stores have to be done when it arrives (the later boundaries are bare s_barriers: nothing is published there)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    boundary();
    wr[0] = read<false, 0>();
    wr[1] = read<false, 1>();
    if constexpr (kWsDepth > 2) {
      wr[2 % kWsDepth] = read<false, 2>();
      wr[3 % kWsDepth] = read<false, 3>();
    }
    static_assert(kWsDepth == 2 || kWsDepth == 4, "cold start of the register ring");
  }
};

template <int CIN, int H1, int H2>
struct SaWs {  // the geometry of a tile's stream (pack_sa_stream)
  static constexpr int S = k1ph(CIN) / 16, FT = H1 / 32, NT = H2 / 32, STEPS1 = FT * S;
  static constexpr int L1P = (STEPS1 + kWsChunk - 1) / kWsChunk * kWsChunk, BODY = FT * 2, TOT = L1P + NT * BODY, CPR = TOT / kWsChunk;
  static_assert(BODY % kWsChunk == 0 && BODY % kWsDepth == 0 && STEPS1 % kWsDepth == 0 && kWsDepth + (L1P - STEPS1) <= kWsChunk && CPR > 3,
                "WStream: chunk boundaries and register-ring slots must sit at compile-time positions");
  static constexpr int pos(int k) { return k < STEPS1 ? k : L1P + (k - STEPS1); }  // stream position of the k-th consumed step
};

// A tile's 32 input rows, one per lane & 31: the lane's half kh of [x(CIN) | pos_j - pos_i (3) | 1 | 0..] in pieces of 8.
// load<ST> reads piece ST (LDS or global memory), split<ST> makes the fragment of k-step ST from it. The pieces of the kh = 1
// half behind x are the tail [dx dy dz 1 0 0 0 0] and zeros: those lanes read piece 0 instead and drop it.
template <int CIN>
struct TileRows {
  static constexpr int HALF = k1ph(CIN) / 2;
  const float* row;  // the lane's row: x[0] (LDS: sx + nb * XS; global: src_x + node * CIN)
  int kh;
  float dx, dy, dz;
  template <int ST>
  __device__ __forceinline__ h3_f32x8 load() const {
    constexpr bool kTailOrZero = HALF + 8 * ST >= CIN;  // for the kh = 1 half
    const float* p = row + ((kTailOrZero && kh) ? 0 : kh * HALF + 8 * ST);
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    return h3_f32x8{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  }
  template <int ST, bool SG>
  __device__ __forceinline__ HFrag split(h3_f32x8 v, float& amax) const {
    constexpr int k1 = HALF + 8 * ST;
    static_assert(k1 < CIN || (k1 - CIN) % 8 == 0, "the tail starts a piece");
    if constexpr (k1 >= CIN) {
      const h3_f32x8 alt = k1 == CIN ? h3_f32x8{dx, dy, dz, 1.f, 0.f, 0.f, 0.f, 0.f} : h3_f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = kh ? alt[e] : v[e];
    }
    const HFrag f = split_vals<SG>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], amax);
    asm volatile("" : "+v"(amax));  // (evaluated here: relu_split_in_place)
    return f;
  }
};

// TileRows over rows in GLOBAL memory whose pieces are requested BATCH at a time and kept in registers: a vector load behind one of
// the stream's DMA bursts is waited for with s_waitcnt vmcnt(0) by the compiler — which waits for the burst as well. One such wait
// per batch instead of one per k-step (the global MLP: all 17 pieces of an object at once; the self-loop tiles: 5 / 3 x 3).
template <int CIN, int BATCH>
struct TileRowsHeld : TileRows<CIN> {
  static constexpr int S = k1ph(CIN) / 16;
  mutable h3_f32x8 held[BATCH];
  template <int ST, int E = 0>
  __device__ __forceinline__ void fill() const {
    if constexpr (E < BATCH && ST + E < S) {
      held[E] = TileRows<CIN>::template load<ST + E>();
      fill<ST, E + 1>();
    }
  }
  template <int ST>
  __device__ __forceinline__ h3_f32x8 load() const {
    if constexpr (ST % BATCH == 0) fill<ST>();
    return held[ST % BATCH];
  }
};

template <int K, typename G, bool SG, int NW, typename X>
__device__ __forceinline__ void ws_l1_steps(WStream<SG, NW>& ws, HFrag (&wr)[kWsDepth], f32x16 (&acc)[G::FT], const X& x, h3_f32x8& xraw, HFrag& xf,
                                            float& amax) {
  if constexpr (K < G::STEPS1) {
    constexpr int st = K / G::FT, ft = K % G::FT, pk = G::pos(K), pn = G::pos(K + kWsDepth);
    if constexpr (ft == 0) xf = x.template split<st, SG>(xraw, amax);
    if constexpr (ft == (G::FT > 1 ? 1 : 0) && st + 1 < G::S) xraw = x.template load<st + 1>();  // the next k-step's piece, FT - 1 steps ahead
    if constexpr (st == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ft][r] = 0.f;
    }
    const HFrag wf = wr[K % kWsDepth];
    wr[K % kWsDepth] = ws.template read<(pn / kWsChunk != pk / kWsChunk), pn % kWsChunk>();
    __builtin_amdgcn_sched_barrier(0);
    mfma_h3<SG>(acc[ft], wf, xf);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G::pos(K + 1) / kWsChunk != pk / kWsChunk) ws.boundary();
    ws_l1_steps<K + 1, G, SG, NW>(ws, wr, acc, x, xraw, xf, amax);
  }
}
// layer 1's accumulator tile ft after ReLU + split, IN ITS OWN REGISTERS: elements 8 m .. 8 m + 3 = the hi halves, 8 m + 4 .. 8 m + 7
// the lo halves of the A fragment m of feature tile ft (pack_sa_l2_h orders layer 2's weights to match)
__device__ __forceinline__ HFrag frag_of(const f32x16& t, int m) {
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  const f32x4v h = {t[8 * m], t[8 * m + 1], t[8 * m + 2], t[8 * m + 3]}, l = {t[8 * m + 4], t[8 * m + 5], t[8 * m + 6], t[8 * m + 7]};
  HFrag f;
  f.hi = __builtin_bit_cast(h3_f16x8, h);
  f.lo = __builtin_bit_cast(h3_f16x8, l);
  return f;
}
template <bool SG>
__device__ __forceinline__ void relu_split_in_place(f32x16& t, float& amax) {
  typedef float f32x4v __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const HFrag f = split_vals<SG>(fmaxf(t[8 * m], 0.f), fmaxf(t[8 * m + 1], 0.f), fmaxf(t[8 * m + 2], 0.f), fmaxf(t[8 * m + 3], 0.f),
                                   fmaxf(t[8 * m + 4], 0.f), fmaxf(t[8 * m + 5], 0.f), fmaxf(t[8 * m + 6], 0.f), fmaxf(t[8 * m + 7], 0.f), amax);
    const f32x4v h = __builtin_bit_cast(f32x4v, f.hi), l = __builtin_bit_cast(f32x4v, f.lo);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      t[8 * m + e] = h[e];
      t[8 * m + 4 + e] = l[e];
    }
  }
  asm volatile("" : "+v"(amax));  // here, not where amax is read: the compiler otherwise keeps the 16 f32 values alive through layer 2
}
template <int SI, typename G, bool SG, int NW>
__device__ __forceinline__ void ws_l2_steps(WStream<SG, NW>& ws, HFrag (&wr)[kWsDepth], f32x16& acc, const f32x16 (&hf)[G::FT], bool stream_goes_on) {
  if constexpr (SI < G::BODY) {
    constexpr int slot = (G::STEPS1 + SI) % kWsDepth, sn = SI + kWsDepth;  // sn: the step fetched now (past BODY: the next tile pass / round)
    const HFrag wf = wr[slot];
    if (sn < G::BODY || stream_goes_on) wr[slot] = ws.template read<(sn / kWsChunk != SI / kWsChunk), sn % kWsChunk>();
    __builtin_amdgcn_sched_barrier(0);
    mfma_h3<SG>(acc, frag_of(hf[SI / 2], SI % 2), wf);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr ((SI + 1) % kWsChunk == 0) ws.boundary();
    ws_l2_steps<SI + 1, G, SG, NW>(ws, wr, acc, hf, stream_goes_on);
  }
}

// The edge MLP of one tile with the weights out of the workgroup's LDS stream. ALL EIGHT waves call it the same number of
// times (a wave without a tile runs a copy of another and drops the result). wr: the register ring, holding the fragments of
// the first kWsDepth steps on entry and, when `more` rounds follow, of the next round's on exit. emit(nt, acc) as sa_mlp_tile.
// before(nt) runs in front of the products of output tile nt (a place to request what emit(nt) will need from global memory).
template <int CIN, int H1, int H2, bool SG, int NW, typename X, typename Emit, typename Before>
__device__ __forceinline__ void sa_mlp_tile_ws(const X& x, WStream<SG, NW>& ws, HFrag (&wr)[kWsDepth], bool more, float& amax, Emit&& emit,
                                               Before&& before) {
  using G = SaWs<CIN, H1, H2>;
  f32x16 acc[G::FT];  // layer 1's accumulators, then layer 2's A fragments
  {
    h3_f32x8 xraw = x.template load<0>();
    HFrag xf;
    ws_l1_steps<0, G, SG, NW>(ws, wr, acc, x, xraw, xf, amax);
  }
  // BatchNorm + bias are folded; ReLU + split turn the accumulators into layer 2's fragments. (Measured and dropped: the second
  // wave of each SIMD splitting a feature tile only when the first pass over layer 2 reaches it, so that the two waves' VALU
  // bursts do not coincide — level 3 4.83 -> 4.94 ms, level 2 3.00 -> 3.10.)
#pragma unroll
  for (int ft = 0; ft < G::FT; ++ft) {
    relu_split_in_place<SG>(acc[ft], amax);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll 1
  for (int nt = 0; nt < G::NT; ++nt) {
    f32x16 out;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[r] = 0.f;
    before(nt);
    ws_l2_steps<0, G, SG, NW>(ws, wr, out, acc, more || nt + 1 < G::NT);
    emit(nt, out);
  }
}
template <int CIN, int H1, int H2, bool SG, int NW, typename X, typename Emit>
__device__ __forceinline__ void sa_mlp_tile_ws(const X& x, WStream<SG, NW>& ws, HFrag (&wr)[kWsDepth], bool more, float& amax, Emit&& emit) {
  sa_mlp_tile_ws<CIN, H1, H2, SG, NW>(x, ws, wr, more, amax, emit, [](int) {});
}

template <int CIN, int H1, int H2, int NS, bool SG>
__global__ __launch_bounds__(64 * kWsWaves, 1) void pn_sa_ws_kernel(SaParams P) {
  constexpr int ND = NS / 2, XS = CIN + 4, PPL = NS / 64, NTH = 64 * kWsWaves;
  static_assert(ND % kWsWaves == 0, "whole rounds of tiles");
  if (P.obj_flags[blockIdx.x]) return;  // left to the f32 launch
  float amax = 0.f;
  extern __shared__ float smem[];
  float* spos = smem + kWsBufs * WStream<SG>::kChunkB / 4;  // [NS][3]   (the ring comes first)
  float* sx = spos + NS * 3;                                 // [NS][XS]
  float* dpos = sx + NS * XS;                                // [ND][3]
  float* sb2 = dpos + ND * 3;                                // [H2] the output bias (pn_ga_ws_kernel: why not a load in the pooling)
  int* nbr = reinterpret_cast<int*>(sb2 + H2);               // [8 waves][32]
  const int o = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 31, kh = lane >> 5;
  WStream<SG> ws;
  ws.open(P.wsh, smem, lane, ND / kWsWaves, SaWs<CIN, H1, H2>::CPR);
  const float* gp = P.src_pos + (size_t)o * NS * 3;
  const float* gx = P.src_x + (size_t)o * NS * CIN;
  for (int i = tid; i < NS * 3; i += NTH) spos[i] = gp[i];
  for (int i = tid; i < NS * CIN / 4; i += NTH) {  // rows of CIN floats -> rows of XS
    const float4 v = reinterpret_cast<const float4*>(gx)[i];
    *reinterpret_cast<float4*>(sx + (i / (CIN / 4)) * XS + (i % (CIN / 4)) * 4) = v;
  }
  for (int i = tid; i < ND * 3; i += NTH) dpos[i] = P.dst_pos[(size_t)o * ND * 3 + i];  // pn_fps_kernel
  for (int i = tid; i < H2; i += NTH) sb2[i] = P.b2[i];
  HFrag wr[kWsDepth];
  ws.start(wr);  // (its barrier publishes the loads above)
  // ---- one 32-row tile per centre: ball query (first 32 in index order), edge MLP, max
  for (int t = w; t < ND; t += kWsWaves) {
    const float cx = dpos[t * 3], cy = dpos[t * 3 + 1], cz = dpos[t * 3 + 2];
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
      const int p = lane + 64 * q;
      const bool in = d2_noFMA(spos[p * 3], spos[p * 3 + 1], spos[p * 3 + 2], cx, cy, cz) < P.r2;
      const unsigned long long mask = __ballot(in);
      const int rank = cnt + __popcll(mask & ((1ull << lane) - 1ull));
      if (in && rank < 32) nbr[w * 32 + rank] = p;
      cnt += __popcll(mask);
    }
    cnt = min(cnt, 32);  // >= 1: the centre is one of the source points
    const int nb = nbr[w * 32 + (j < cnt ? j : 0)];
    const TileRows<CIN> rows{sx + nb * XS, kh, spos[nb * 3] - cx, spos[nb * 3 + 1] - cy, spos[nb * 3 + 2] - cz};
    auto pool = [&](int nt, const f32x16& acc) {
      float m = max16(acc);
      m = fmaxf(m, __shfl_xor(m, 32));
      const int c = nt * 32 + j;
      // with self-loop messages the level's output is finished by pn_self_ws_kernel: max(self) + bias, ReLU
      if (kh == 0) P.dst_x[((size_t)o * ND + t) * H2 + c] = P.self_loops ? m : fmaxf(m + sb2[c], 0.f);
    };
    sa_mlp_tile_ws<CIN, H1, H2, SG, kWsWaves>(rows, ws, wr, t + kWsWaves < ND, amax, pool);
  }
  if (!(amax < kSplitF16Safe)) P.obj_flags[o] = 1;  // anything at or beyond 3e4 (or NaN) entered a split product: hand the object over
}

// The extra (k -> k) messages of PyG's add_self_loops on the bipartite batch (centre k of the CELL's batch also hears source
// node k of the cell's batch: oracle/t2l_oracle_pointnet.py) for all objects: tile (o, tt) = centres 32 tt.. of object o. A
// workgroup takes eight tiles per round; a tile's pass over output tile nt turns the raw maxima pn_sa_ws_kernel left into the
// level's output.
template <int CIN, int H1, int H2, int NS, bool SG>
__global__ __launch_bounds__(64 * kWsWaves, 1) void pn_self_ws_kernel(SaParams P, int n_obj) {
  constexpr int ND = NS / 2, TPO = ND / 32;
  extern __shared__ float smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 31, kh = lane >> 5;
  const int n_tiles = n_obj * TPO, per_round = kWsWaves * (int)gridDim.x, rounds = (n_tiles + per_round - 1) / per_round;
  WStream<SG> ws;
  ws.open(P.wsh, smem, lane, rounds, SaWs<CIN, H1, H2>::CPR);
  HFrag wr[kWsDepth];
  ws.start(wr);
  for (int r = 0; r < rounds; ++r) {
    const int id = (r * (int)gridDim.x + (int)blockIdx.x) * kWsWaves + w;
    bool live = id < n_tiles;
    const int o = live ? id / TPO : 0, tt = live ? id % TPO : 0;
    live = live && !P.obj_flags[o];
    const int cb = P.cell_base[o], t = tt * 32 + j;
    const size_t node = (size_t)cb * NS + (size_t)(o - cb) * ND + t;  // source node k = (o - cb) ND + t of the cell's batch
    const float* sp = P.src_pos + node * 3;
    const float* dp = P.dst_pos + ((size_t)o * ND + t) * 3;
    TileRowsHeld<CIN, (CIN > 64 ? 3 : 5)> rows;  // (level 3: 9 pieces would not fit the 256 registers of two waves per SIMD)
    rows.row = P.src_x + node * CIN;
    rows.kh = kh;
    rows.dx = sp[0] - dp[0]; rows.dy = sp[1] - dp[1]; rows.dz = sp[2] - dp[2];
    float amax = 0.f;
    // the raw maxima (and the bias) of an output tile are requested BEFORE its products: the compiler waits for them with vmcnt(0),
    // and behind the products that wait no longer sits on the stream's latest DMA burst as well
    float raw[16], bias = 0.f;
    float* const out0 = P.dst_x + ((size_t)o * ND + tt * 32 + 4 * kh) * H2 + j;
    auto request = [&](int nt) {
      if (live) {
        bias = P.b2[nt * 32 + j];
#pragma unroll
        for (int q = 0; q < 16; ++q) raw[q] = out0[(size_t)((q & 3) + 8 * (q >> 2)) * H2 + nt * 32];
      }
    };
    auto finish = [&](int nt, const f32x16& acc) {
      if (live) {
#pragma unroll
        for (int q = 0; q < 16; ++q) out0[(size_t)((q & 3) + 8 * (q >> 2)) * H2 + nt * 32] = fmaxf(fmaxf(raw[q], acc[q]) + bias, 0.f);
      }
    };
    sa_mlp_tile_ws<CIN, H1, H2, SG, kWsWaves>(rows, ws, wr, r + 1 < rounds, amax, finish, request);
    if (live && !(amax < kSplitF16Safe)) P.obj_flags[o] = 1;
  }
}

// GlobalAbstraction on split / plain f16 through the same machinery: an object's 32 points are one tile of get_mlp([259,512,1024])
// ([x(256) | pos(3) | 1] rows straight from global memory, 16 feature tiles = 256 accumulator registers, so ONE wave per SIMD), four
// objects per workgroup round share the 2.6 MB stream of packed weights that pn_ga_kernel below pulls out of the L2 once per object.
constexpr int kGaWaves = 4;
template <bool SG>
__global__ __launch_bounds__(64 * kGaWaves, 1) void pn_ga_ws_kernel(const float* __restrict__ pos3, const float* __restrict__ x3,
                                                                    const uint4* __restrict__ stream, const float* __restrict__ b2,
                                                                    float* __restrict__ f0, int32_t* __restrict__ obj_flags, int n_obj) {
  extern __shared__ float smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 31, kh = lane >> 5;
  const int per_round = kGaWaves * (int)gridDim.x, rounds = (n_obj + per_round - 1) / per_round;
  WStream<SG, kGaWaves> ws;
  ws.open(stream, smem, lane, rounds, SaWs<256, 512, 1024>::CPR);
  // the output bias through LDS: a vector load inside the pooling makes the compiler wait for vmcnt(0) there — i.e. for the DMA burst
  // the stream issued a moment earlier, one L2 round trip per output tile (32 per object)
  float* sb2 = smem + kWsBufs * WStream<SG, kGaWaves>::kChunkB / 4;  // [1024]
  for (int i = tid; i < 1024; i += 64 * kGaWaves) sb2[i] = b2[i];
  HFrag wr[kWsDepth];
  ws.start(wr);
  for (int r = 0; r < rounds; ++r) {
    const int id = (r * (int)gridDim.x + (int)blockIdx.x) * kGaWaves + w;
    const bool live = id < n_obj && !obj_flags[id < n_obj ? id : 0];
    const int o = live ? id : 0;
    const float* pp = pos3 + ((size_t)o * 32 + j) * 3;
    TileRowsHeld<256, 17> rows;
    rows.row = x3 + ((size_t)o * 32 + j) * 256;
    rows.kh = kh;
    rows.dx = pp[0]; rows.dy = pp[1]; rows.dz = pp[2];
    float amax = 0.f;
    auto pool = [&](int nt, const f32x16& acc) {
      float m = max16(acc);
      m = fmaxf(m, __shfl_xor(m, 32));
      const int c = nt * 32 + j;
      if (live && kh == 0) f0[(size_t)o * 1024 + c] = fmaxf(m + sb2[c], 0.f);
    };
    sa_mlp_tile_ws<256, 512, 1024, SG, kGaWaves>(rows, ws, wr, r + 1 < rounds, amax, pool);
    if (live && !(amax < kSplitF16Safe)) obj_flags[o] = 1;
  }
}

// GlobalAbstraction on the f32 MFMA (option encoder_f32, and the objects the split launch flagged): get_mlp([259,512,1024]) over
// the 32 points of an object, max. One workgroup per object. The 512-wide hidden layer passes through LDS in two halves of 256
// units (the units k-steps [32 hf, 32 hf + 32) of the half-split packing of the second Linear cover) while the 1024 outputs
// (8 column tiles per wave) accumulate in registers: 68 KB of LDS instead of 100 KB, two objects per CU.
constexpr int kGaH1 = 512, kGaHS = 256 + 4, kGaH2 = 1024, kGaK = 264, kGaXS = kGaK + 4;  // [x(256) | pos(3) | 1 | 0..] padded to 8
__global__ __launch_bounds__(256, 2) void pn_ga_kernel(const float* __restrict__ pos3, const float* __restrict__ x3,
                                                       const float4* __restrict__ w1, const float4* __restrict__ w2,
                                                       const float* __restrict__ b2, float* __restrict__ f0,
                                                       const int32_t* __restrict__ obj_flags) {
  extern __shared__ float smem[];
  float* X = smem;                 // [32][kGaXS]  rows = [x(256) | pos(3) | 1 | 0..]
  float* Hd = X + 32 * kGaXS;      // [32][kGaHS]  one half of the hidden layer
  const int o = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 31, kh = lane >> 5;
  if (obj_flags && !obj_flags[o]) return;  // behind a split launch: only the objects it flagged
  for (int i = tid; i < 32 * kGaK; i += 256) {
    const int r = i / kGaK, k = i % kGaK;
    float v = 0.f;
    if (k < 256) v = x3[((size_t)o * 32 + r) * 256 + k];
    else if (k < 259) v = pos3[((size_t)o * 32 + r) * 3 + (k - 256)];
    else if (k == 259) v = 1.f;
    X[r * kGaXS + k] = v;
  }
  __syncthreads();
  f32x16 acc[8];  // output column tiles w, w + 4, ..., w + 28
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int hf = 0; hf < 2; ++hf) {
    // layer 1, this half: hidden tiles 4 hf + w -> Hd columns 32 w.., and 8 + 4 hf + w -> Hd columns 128 + 32 w..
    f32x16 h[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) h[0][r] = h[1][r] = 0.f;
    {  // the two tiles share every A fragment
      const float* xr = X + j * kGaXS + kh * (kGaK / 2);
      const float4* wa = w1 + (size_t)(4 * hf + w) * (kGaK / 8) * 64 + lane;
      const float4* wb = w1 + (size_t)(8 + 4 * hf + w) * (kGaK / 8) * 64 + lane;
#pragma unroll 3
      for (int q = 0; q < kGaK / 8; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(xr + 4 * q);
        const float4 b0 = wa[q * 64], b1 = wb[q * 64];
        h[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, h[0], 0, 0, 0);
        h[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, h[1], 0, 0, 0);
        h[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, h[0], 0, 0, 0);
        h[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, h[1], 0, 0, 0);
        h[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, h[0], 0, 0, 0);
        h[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, h[1], 0, 0, 0);
        h[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, h[0], 0, 0, 0);
        h[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, h[1], 0, 0, 0);
      }
    }
    if (hf) __syncthreads();  // every wave has consumed the first half
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) Hd[((r & 3) + 8 * (r >> 2) + 4 * kh) * kGaHS + 128 * u + 32 * w + j] = fmaxf(h[u][r], 0.f);
    __syncthreads();
    {  // layer 2 partial: the 8 column tiles of this wave share every A fragment (one LDS read, 8 weight loads, 32 MFMAs)
      const float* hr = Hd + j * kGaHS + kh * 128;
      const float4* wp = w2 + ((size_t)w * (kGaH1 / 8) + 32 * hf) * 64 + lane;
#pragma unroll 2
      for (int q = 0; q < 32; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(hr + 4 * q);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float4 b = wp[((size_t)4 * t * (kGaH1 / 8) + q) * 64];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[t], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    float m = max16(acc[t]);
    m = fmaxf(m, __shfl_xor(m, 32));
    const int c = (w + 4 * t) * 32 + j;
    if (kh == 0) f0[(size_t)o * kGaH2 + c] = fmaxf(m + b2[c], 0.f);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The point batches PointNet++ eats, built on the GPU from the raw object points, for the three transforms the reference
// composes (torch_geometric.transforms, applied per object on the host: dataloading/kitti360pose/utils.py:138-143):
//   flags 0                      FixedPoints(256) only — `--no_pc_augment`, what EVERY published command passes
//                                (README.md:89,107,125-126,139-140; evaluation/pipeline.py:215-216, training/coarse.py:182-184):
//                                the backbone sees the cell-normalised coordinates as they are, radii 0.2/0.3/0.4 absolute;
//   T2L_SAMPLE_NORMALIZE         + NormalizeScale (centre on the mean of the sampled points, scale by 0.999999 / max |coordinate|)
//                                — evaluation without the flag (evaluation/pipeline.py:217-218, training/coarse.py:193);
//   T2L_SAMPLE_ROTATE|NORMALIZE  + RandomRotate(rotate_deg, axis=2) in between — training without the flag
//                                (training/coarse.py:185-192): one angle per object, uniform in [-deg, +deg].
// FixedPoints = 256 indices drawn with replacement. The reference draws from numpy's / python's global RNGs; here index j of
// object o is floor(u * n) with u = the top 24 bits of lowbias32(j * 0x9E3779B1 + (seed ^ o * 0x85EBCA77)) / 2^24 and the
// angle comes from lowbias32(0xA5A5A5A5 + (seed ^ o * 0x85EBCA77)) the same way. One workgroup per object, one thread per point.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256) void sample_points_kernel(const float* __restrict__ xyz, const float* __restrict__ rgb,
                                                            const int64_t* __restrict__ offsets, uint32_t seed, int flags,
                                                            float rotate_rad, float* __restrict__ out_pos, float* __restrict__ out_rgb) {
  __shared__ float red[4][4];
  const int o = blockIdx.x, j = threadIdx.x, lane = j & 63, w = j >> 6;
  const int64_t p0 = offsets[o];
  const uint32_t n = (uint32_t)(offsets[o + 1] - p0);
  const uint32_t okey = seed ^ ((uint32_t)o * 0x85EBCA77u);
  const uint32_t x = lowbias32((uint32_t)j * 0x9E3779B1u + okey);
  const uint32_t idx = (uint32_t)(((uint64_t)(x >> 8) * n) >> 24);
  const float* src = xyz + (size_t)(p0 + idx) * 3;
  float px = src[0], py = src[1], pz = src[2];
  const float* col = rgb + (size_t)(p0 + idx) * 3;
  const size_t ob = ((size_t)o * 256 + j) * 3;
  out_rgb[ob] = col[0]; out_rgb[ob + 1] = col[1]; out_rgb[ob + 2] = col[2];
  if (flags & T2L_SAMPLE_ROTATE) {  // pos @ [[c, s, 0], [-s, c, 0], [0, 0, 1]]
    const float u = (float)(lowbias32(0xA5A5A5A5u + okey) >> 8) * (1.f / 16777216.f);
    const float ang = rotate_rad * (2.f * u - 1.f);
    const float c = cosf(ang), sn = sinf(ang);
    const float rx = px * c - py * sn, ry = px * sn + py * c;
    px = rx; py = ry;
  }
  if (!(flags & T2L_SAMPLE_NORMALIZE)) {  // uniform branch
    out_pos[ob] = px; out_pos[ob + 1] = py; out_pos[ob + 2] = pz;
    return;
  }
  float sx = px, sy = py, sz = pz;
#pragma unroll
  for (int off = 32; off; off >>= 1) { sx += __shfl_xor(sx, off); sy += __shfl_xor(sy, off); sz += __shfl_xor(sz, off); }
  if (lane == 0) { red[w][0] = sx; red[w][1] = sy; red[w][2] = sz; }
  __syncthreads();
  const float mx = (red[0][0] + red[1][0] + red[2][0] + red[3][0]) * (1.f / 256.f);
  const float my = (red[0][1] + red[1][1] + red[2][1] + red[3][1]) * (1.f / 256.f);
  const float mz = (red[0][2] + red[1][2] + red[2][2] + red[3][2]) * (1.f / 256.f);
  px -= mx; py -= my; pz -= mz;
  float m = fmaxf(fmaxf(fabsf(px), fabsf(py)), fabsf(pz));
#pragma unroll
  for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  __syncthreads();
  if (lane == 0) red[w][3] = m;
  __syncthreads();
  const float scale = (1.0f / fmaxf(fmaxf(red[0][3], red[1][3]), fmaxf(red[2][3], red[3][3]))) * 0.999999f;
  out_pos[ob] = px * scale; out_pos[ob + 1] = py * scale; out_pos[ob + 2] = pz * scale;
}

int sample_points_impl(t2l_ctx* ctx, const float* xyz, const float* rgb, const int64_t* point_offsets_dev, int n_objects, uint32_t seed,
                       int flags, float rotate_deg, float* out_pos, float* out_rgb, hipStream_t s) {
  if (!xyz || !rgb || !point_offsets_dev || !out_pos || !out_rgb) return fail(ctx, T2L_EINVAL, "t2l_sample_object_points: null argument");
  if (n_objects <= 0) return n_objects == 0 ? T2L_OK : fail(ctx, T2L_EINVAL, "t2l_sample_object_points: n_objects < 0");
  if (flags & ~(T2L_SAMPLE_NORMALIZE | T2L_SAMPLE_ROTATE)) return fail(ctx, T2L_EINVAL, "t2l_sample_object_points: unknown transform flag");
  hipLaunchKernelGGL(sample_points_kernel, dim3(n_objects), dim3(256), 0, s, xyz, rgb, point_offsets_dev, seed, flags,
                     rotate_deg * 0.017453292519943295f, out_pos, out_rgb);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

template <int CIN, int H1, int H2, int NS>
static size_t sa_lds_bytes() {
  constexpr int ND = NS / 2, XS = CIN + 4;
  return sizeof(float) * (NS * 3 + NS * XS + ND * 3 + ND * H2) + sizeof(int) * (ND + 4 * 32);
}
template <int CIN, int NS, bool SG>
static size_t sa_ws_lds_bytes() {
  constexpr int ND = NS / 2, XS = CIN + 4;
  return (size_t)kWsBufs * WStream<SG>::kChunkB + sizeof(float) * (NS * 3 + NS * XS + ND * 3 + 256) + sizeof(int) * (kWsWaves * 32);  // (+ H2 <= 256 bias values)
}

// levels 2 and 3, split or plain f16: the centres' tiles, then the self-loop tiles (which finish the output)
template <int CIN, int H1, int H2, int NS, bool SG>
static hipError_t launch_sa_ws(const SaParams& P, int n_obj, hipStream_t s) {
  const size_t lds = sa_ws_lds_bytes<CIN, NS, SG>(), lds_self = (size_t)kWsBufs * WStream<SG>::kChunkB;
  int dev = 0;
  (void)hipGetDevice(&dev);
  static PerDeviceOnce attr;
  if (attr.need(dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pn_sa_ws_kernel<CIN, H1, H2, NS, SG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr.mark(dev);
  }
  hipLaunchKernelGGL((pn_fps_kernel<NS>), dim3((n_obj + 3) / 4), dim3(256), 0, s, P.src_pos, P.dst_pos, P.obj_flags, n_obj);
  hipLaunchKernelGGL((pn_sa_ws_kernel<CIN, H1, H2, NS, SG>), dim3(n_obj), dim3(64 * kWsWaves), lds, s, P);
  if (P.self_loops) {
    const int n_tiles = n_obj * (NS / 64), grid = std::min(256, (n_tiles + kWsWaves - 1) / kWsWaves);
    hipLaunchKernelGGL((pn_self_ws_kernel<CIN, H1, H2, NS, SG>), dim3(grid), dim3(64 * kWsWaves), lds_self, s, P, n_obj);
  }
  return hipGetLastError();
}

// split = true: the split-f16 launch over all objects (it skips flagged ones and flags new ones) followed by the f32 launch
// that serves exactly the flagged objects; split = false: one f32 launch over everything (P.obj_flags must be null)
template <int CIN, int H1, int H2, int NS>
static hipError_t launch_sa(const SaParams& P, int n_obj, bool split, bool single, hipStream_t s) {
  const size_t lds = sa_lds_bytes<CIN, H1, H2, NS>();
  int dev = 0;
  (void)hipGetDevice(&dev);
  static PerDeviceOnce attr;
  if (attr.need(dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pn_sa_kernel<CIN, H1, H2, NS, 0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if constexpr (CIN < 64) {
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pn_sa_kernel<CIN, H1, H2, NS, 1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pn_sa_kernel<CIN, H1, H2, NS, 2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (e != hipSuccess) return e;
    attr.mark(dev);
  }
  if constexpr (CIN >= 64) {
    if (split) {
      const hipError_t e = single ? launch_sa_ws<CIN, H1, H2, NS, true>(P, n_obj, s) : launch_sa_ws<CIN, H1, H2, NS, false>(P, n_obj, s);
      if (e != hipSuccess) return e;
    }
  } else {
    if (split) {
      hipLaunchKernelGGL((pn_fps_kernel<NS>), dim3((n_obj + 3) / 4), dim3(256), 0, s, P.src_pos, P.dst_pos, P.obj_flags, n_obj);
      SaParams Q = P;
      Q.centres_given = 1;
      if (single) hipLaunchKernelGGL((pn_sa_kernel<CIN, H1, H2, NS, 2>), dim3(n_obj), dim3(256), lds, s, Q);  // option encoder_f16
      else hipLaunchKernelGGL((pn_sa_kernel<CIN, H1, H2, NS, 1>), dim3(n_obj), dim3(256), lds, s, Q);
    }
  }
  hipLaunchKernelGGL((pn_sa_kernel<CIN, H1, H2, NS, 0>), dim3(n_obj), dim3(256), lds, s, P);
  return hipGetLastError();
}

int pointnet_features_impl(t2l_ctx* ctx, const float* pos, const float* rgb, const int32_t* cell_offsets, int n_cells, float* out,
                           hipStream_t s) {
  PointNetWeights* W = reinterpret_cast<PointNetWeights*>(ctx->pn);
  if (!W) return fail(ctx, T2L_ESTATE, "t2l_pointnet_features: the loaded state_dict carried no object_encoder.pointnet.* tensors");
  if (!pos || !rgb || !cell_offsets || !out || n_cells <= 0) return fail(ctx, T2L_EINVAL, "t2l_pointnet_features: null argument");
  const int n_obj = cell_offsets[n_cells] - cell_offsets[0];
  if (cell_offsets[0] != 0 || n_obj <= 0) return fail(ctx, T2L_EINVAL, "t2l_pointnet_features: cell_offsets must start at 0 and end at n_objects > 0");
  std::vector<int32_t> base(n_obj);
  for (int c = 0; c < n_cells; ++c) {
    if (cell_offsets[c + 1] < cell_offsets[c]) return fail(ctx, T2L_EINVAL, "t2l_pointnet_features: cell_offsets must be non-decreasing");
    for (int o = cell_offsets[c]; o < cell_offsets[c + 1]; ++o) base[o] = cell_offsets[c];
  }
  // level buffers for ALL objects (105 KB per object; 288 GB of HBM hold the whole KITTI360Pose DB at once)
  const size_t per_obj = sizeof(float) * (128 * 3 + 128 * 64 + 64 * 3 + 64 * 128 + 32 * 3 + 32 * 256 + 1024 + 512) + 2 * sizeof(int32_t);
  const size_t need = per_obj * (size_t)n_obj + 4096;
  if (need > W->ws_cap) {
    if (W->ws) (void)hipFree(W->ws);
    W->ws = nullptr;
    W->ws_cap = 0;
    T2L_HIP(ctx, hipMalloc(&W->ws, need));
    W->ws_cap = need;
  }
  float* p1 = reinterpret_cast<float*>(W->ws);
  float* x1 = p1 + (size_t)n_obj * 128 * 3;
  float* p2 = x1 + (size_t)n_obj * 128 * 64;
  float* x2 = p2 + (size_t)n_obj * 64 * 3;
  float* p3 = x2 + (size_t)n_obj * 64 * 128;
  float* x3 = p3 + (size_t)n_obj * 32 * 3;
  float* f0 = x3 + (size_t)n_obj * 32 * 256;
  float* f1 = f0 + (size_t)n_obj * 1024;
  int32_t* d_base = reinterpret_cast<int32_t*>(f1 + (size_t)n_obj * 512);
  const bool split = !ctx->encoder_f32;
  int32_t* d_flags = split ? d_base + n_obj : nullptr;
  if (split) T2L_HIP(ctx, hipMemsetAsync(d_flags, 0, sizeof(int32_t) * n_obj, s));
  T2L_HIP(ctx, hipMemcpyAsync(d_base, base.data(), sizeof(int32_t) * n_obj, hipMemcpyHostToDevice, s));
  T2L_HIP(ctx, hipStreamSynchronize(s));  // `base` is a host temporary
  event_begin(ctx, "pointnet", s);
  const float radii[3] = {0.2f, 0.3f, 0.4f};
  SaParams P{pos, rgb, p1, x1, d_base, W->w1[0], W->w2[0], W->b2[0], radii[0] * radii[0], ctx->pn_self_loops, W->w1h[0], W->w2h[0], W->wsh[0], d_flags};
  T2L_HIP(ctx, (launch_sa<3, 32, 64, 256>(P, n_obj, split, ctx->encoder_f16 != 0, s)));
  P = SaParams{p1, x1, p2, x2, d_base, W->w1[1], W->w2[1], W->b2[1], radii[1] * radii[1], ctx->pn_self_loops, W->w1h[1], W->w2h[1], W->wsh[1], d_flags};
  T2L_HIP(ctx, (launch_sa<64, 128, 128, 128>(P, n_obj, split, ctx->encoder_f16 != 0, s)));
  P = SaParams{p2, x2, p3, x3, d_base, W->w1[2], W->w2[2], W->b2[2], radii[2] * radii[2], ctx->pn_self_loops, W->w1h[2], W->w2h[2], W->wsh[2], d_flags};
  T2L_HIP(ctx, (launch_sa<128, 256, 256, 64>(P, n_obj, split, ctx->encoder_f16 != 0, s)));
  {
    const size_t lds_f = sizeof(float) * (32 * kGaXS + 32 * kGaHS);
    static PerDeviceOnce attr;
    if (attr.need(ctx->device)) {
      T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&pn_ga_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
      attr.mark(ctx->device);
    }
    if (split) {
      const int grid = std::min(256, (n_obj + kGaWaves - 1) / kGaWaves);
      static PerDeviceOnce ga_attr;
      const size_t lds_sg = (size_t)kWsBufs * WStream<true, kGaWaves>::kChunkB + 4096, lds_sp = (size_t)kWsBufs * WStream<false, kGaWaves>::kChunkB + 4096;
      if (ga_attr.need(ctx->device)) {
        T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&pn_ga_ws_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sg));
        T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&pn_ga_ws_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sp));
        ga_attr.mark(ctx->device);
      }
      if (ctx->encoder_f16)
        hipLaunchKernelGGL(pn_ga_ws_kernel<true>, dim3(grid), dim3(64 * kGaWaves), lds_sg, s, p3, x3, W->gash, W->gab2, f0, d_flags, n_obj);
      else
        hipLaunchKernelGGL(pn_ga_ws_kernel<false>, dim3(grid), dim3(64 * kGaWaves), lds_sp, s, p3, x3, W->gash, W->gab2, f0, d_flags, n_obj);
    }
    hipLaunchKernelGGL(pn_ga_kernel, dim3(n_obj), dim3(256), lds_f, s, p3, x3, W->ga1, W->ga2, W->gab2, f0, d_flags);
  }
  {  // lin1 / lin2 + ReLU over all objects (pointnet2.py:86-89): plain GEMMs on the row-major torch weights
    train::GemmArgs g{f0, W->lin1w, f1, W->lin1b, n_obj, 512, 1024, 1024, 1024, 512, 1, 0, 1024, nullptr};
    hipLaunchKernelGGL((train::gemm_kernel<true, true>), dim3(512 / 32, (n_obj + 31) / 32, 1), dim3(256), 0, s, g);
    train::GemmArgs h{f1, W->lin2w, out, W->lin2b, n_obj, 256, 512, 512, 512, 256, 1, 0, 512, nullptr};
    hipLaunchKernelGGL((train::gemm_kernel<true, true>), dim3(256 / 32, (n_obj + 31) / 32, 1), dim3(256), 0, s, h);
  }
  event_end(ctx, "pointnet", s);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

}  // namespace t2l
