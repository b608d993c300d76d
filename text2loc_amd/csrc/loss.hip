// Symmetric InfoNCE of training/losses.py:255-283 (ContrastiveLoss), forward + analytic backward, as ONE
// launch of 4 * ceil(B / 32) workgroups (the op is latency bound: 2*B^2*D = 2.1 MFLOP at B = 64).
//
//   ia_i = 1/|im_i|, ip_j = 1/|s_j|                                   losses.py:271-272
//   sim  = (im @ s^T) * ia_i * ip_j            f32 MFMA 32x32x2      :274
//   E    = exp(sim / T); R_i = sum_j E_ij; C_j = sum_i E_ij           :277-278 (no max-subtraction, as the reference)
//   loss = mean_i( log C_i + log R_i - 2 sim_ii / T )                 == :280-281
//   G    = dloss/dsim = (E_ij/C_j + E_ij/R_i - 2 delta_ij) / (T B)
//   d im = ((G  @ s^) - im^ * rowdot) * ia ;  d s = ((G^T @ im^) - s^ * rowdot) * ip     (x^ = x/|x|)
//
// E/G lives in LDS ([Bp][Bp+1] f32, Bp = B rounded up to 32, B <= 128); operands stream from L2.
#include "t2l_internal.h"

namespace t2l {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CTRL>
__device__ __forceinline__ float loss_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_f32(float v) {  // all-reduce over the 64 lanes on the VALU
  v += loss_dpp<0xB1>(v);
  v += loss_dpp<0x4E>(v);
  v += loss_dpp<0x141>(v);
  v += loss_dpp<0x140>(v);
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
__device__ __forceinline__ float half_sum(float v) {  // sum over the 32 lanes that share lane>>5 (DPP, no LDS traffic)
  v += loss_dpp<0xB1>(v);
  v += loss_dpp<0x4E>(v);
  v += loss_dpp<0x141>(v);
  v += loss_dpp<0x140>(v);
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// One 32-row block of  out = ((M @ X^) - Y^ * rowdot) * iy   where M is G (transpose=false) or G^T.
// X^ = x * ix (rows k), Y^ = y * iy (rows i). Result rows i0..i0+31, all 256 columns, one wave.
__device__ __forceinline__ void grad_rowblock(const float* __restrict__ G, int ldg, bool transpose, int B, int i0,
                                              const float* __restrict__ x, const float* __restrict__ ix,
                                              const float* __restrict__ y, const float* __restrict__ iy,
                                              float* __restrict__ out, int lane) {
  const int col = lane & 31, half = lane >> 5;
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int gi = i0 + col;  // A-operand row of this lane
#pragma unroll 8  // the operands of 8 steps in flight: every step's loads come from L2 and nothing else hides their latency
  for (int k0 = 0; k0 < B; k0 += 2) {
    const int k = k0 + half;
    float a = 0.f, scale = 0.f;
    if (k < B) {
      a = transpose ? G[k * ldg + gi] : G[gi * ldg + k];
      scale = ix[k];
    }
    const float* xr = x + (size_t)min(k, B - 1) * kD + col;
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xr[32 * t] * scale, acc[t], 0, 0, 0);
  }
  // lane holds column 32*t + col, rows i0 + (r&3) + 8*(r>>2) + 4*half.
  // The y rows of the NEXT four accumulator rows are loaded before the current four are stored (double-buffered registers, full
  // blocks store without a branch): written row by row — 8 loads, a reduction, 8 guarded stores — every row's loads queued behind
  // the previous row's stores (s_waitcnt vmcnt(7) x 8 per row: 16 store-drain + load round trips, ~20 of the kernel's 44 us).
  const bool full = i0 + 32 <= B;
  float yb[2][4][8], ib[2][4];
  auto load_rows = [&](int b4, float (&yy)[4][8], float (&ii)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 4 * b4 + q;
      const int row = min(i0 + (r & 3) + 8 * (r >> 2) + 4 * half, B - 1);
      ii[q] = iy[row];
      const float* yr = y + (size_t)row * kD + col;
#pragma unroll
      for (int t = 0; t < 8; ++t) yy[q][t] = yr[32 * t];
    }
  };
  load_rows(0, yb[0], ib[0]);
#pragma unroll
  for (int b4 = 0; b4 < 4; ++b4) {
    if (b4 + 1 < 4) load_rows(b4 + 1, yb[(b4 + 1) & 1], ib[(b4 + 1) & 1]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 4 * b4 + q;
      const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const float inv = row < B ? ib[b4 & 1][q] : 0.f;
      float yv[8];
      float part = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        yv[t] = yb[b4 & 1][q][t] * inv;
        part += acc[t][r] * yv[t];
      }
      const float dot = half_sum(part);
      float* orow = out + (size_t)min(row, B - 1) * kD + col;
      if (full) {
#pragma unroll
        for (int t = 0; t < 8; ++t) orow[32 * t] = (acc[t][r] - yv[t] * dot) * inv;
      } else if (row < B) {
#pragma unroll
        for (int t = 0; t < 8; ++t) orow[32 * t] = (acc[t][r] - yv[t] * dot) * inv;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The loss as 4*nb workgroups. (Rounds 1-2 ran ONE workgroup — 38.6 us per call at batch 64 against 22.4 us; removed in round 5.) The single workgroup spent half its time
// in the two gradient products — 2*nb row blocks of 8 column tiles on 4 waves — and a quarter in load round trips ahead of the
// similarity tiles. Here EVERY workgroup repeats the cheap part (similarity tiles, E, row / column sums, G: B^2 work, in its own LDS;
// the inverse norms fall out of the operand loads of the similarity tiles, there is no norm pass), and the gradients' 2*nb*8 output
// tiles are dealt one per WAVE over the whole grid: the correction term of the normalisation's backward,
//   rowdot_i = sum_c (G @ s^)_ic * im^_ic = sum_k G_ik * cos_ik        (and sum_k G_ki * cos_ki for d s),
// comes from the B x B matrices in LDS, so a tile needs nothing from its row's other tiles. Workgroup 0 writes the loss.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void contrastive_tiles_kernel(const float* __restrict__ im, const float* __restrict__ s, int B,
                                                                   float inv_t, float* __restrict__ loss, float* __restrict__ g_im,
                                                                   float* __restrict__ g_s) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Bp = (B + 31) / 32 * 32;
  const int ldg = Bp + 1;
  float* E = smem;               // [Bp][ldg]  E, then G
  float* S = E + Bp * ldg;       // [Bp][ldg]  raw dot products, then cosines
  float* ia = S + Bp * ldg;      // [Bp]
  float* ip = ia + Bp;           // [Bp]
  float* R = ip + Bp;            // [Bp]
  float* C = R + Bp;             // [Bp]
  float* rdA = C + Bp;           // [Bp]  rowdot of d im
  float* rdP = rdA + Bp;         // [Bp]  rowdot of d s
  float* part = rdP + Bp;        // [4][Bp] per-wave column partials
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int nb = Bp / 32;

  // 1. raw similarity tiles; a lane streams half a row of each operand, so the squared norms come with them
  for (int tile = wave; tile < nb * nb; tile += 4) {
    const int i0 = (tile / nb) * 32, j0 = (tile % nb) * 32;
    const float4* ap = reinterpret_cast<const float4*>(im + (size_t)min(i0 + col, B - 1) * kD + half * 128);
    const float4* pp = reinterpret_cast<const float4*>(s + (size_t)min(j0 + col, B - 1) * kD + half * 128);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float na = 0.f, np = 0.f;
#pragma unroll 16
    for (int q = 0; q < 32; ++q) {
      const float4 a = ap[q], p = pp[q];
      na += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
      np += p.x * p.x + p.y * p.y + p.z * p.z + p.w * p.w;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, p.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, p.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, p.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, p.w, acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);  // (MFMA -> VALU read behind a loop exit: see gemm_rows2.h)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    na += __shfl_xor(na, 32);
    np += __shfl_xor(np, 32);
    if (half == 0) {
      if (j0 == 0) ia[i0 + col] = i0 + col < B ? 1.f / sqrtf(na) : 0.f;
      if (i0 == 0) ip[j0 + col] = j0 + col < B ? 1.f / sqrtf(np) : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) S[(i0 + (r & 3) + 8 * (r >> 2) + 4 * half) * ldg + j0 + col] = acc[r];
  }
  __syncthreads();

  // 2. cosines, E, row sums (wave reduction) and per-wave column partials; columns j = lane, lane + 64
  {
    float c0 = 0.f, c1 = 0.f;
    const float p0 = ip[min(lane, Bp - 1)], p1 = lane + 64 < Bp ? ip[lane + 64] : 0.f;
    for (int i = wave; i < Bp; i += 4) {
      const float a = ia[i];
      float e0 = 0.f, e1 = 0.f;
      if (lane < Bp) {
        const float c = S[i * ldg + lane] * a * p0;
        S[i * ldg + lane] = c;
        e0 = (i < B && lane < B) ? __expf(c * inv_t) : 0.f;
        E[i * ldg + lane] = e0;
      }
      if (lane + 64 < Bp) {
        const float c = S[i * ldg + lane + 64] * a * p1;
        S[i * ldg + lane + 64] = c;
        e1 = (i < B && lane + 64 < B) ? __expf(c * inv_t) : 0.f;
        E[i * ldg + lane + 64] = e1;
      }
      c0 += e0;
      c1 += e1;
      const float r = wave_sum_f32(e0 + e1);
      if (lane == 0) R[i] = r;
    }
    if (lane < Bp) part[wave * Bp + lane] = c0;
    if (lane + 64 < Bp) part[wave * Bp + lane + 64] = c1;
  }
  __syncthreads();
  if (tid < Bp) C[tid] = part[tid] + part[Bp + tid] + part[2 * Bp + tid] + part[3 * Bp + tid];
  __syncthreads();
  if (blockIdx.x == 0 && wave == 0) {
    float v = 0.f;
    for (int i = lane; i < B; i += 64) v += __logf(C[i]) + __logf(R[i]) - 2.f * S[i * ldg + i] * inv_t;
    v = wave_sum_f32(v);
    if (lane == 0) loss[0] = v / (float)B;
  }
  if (!g_im) return;

  // 3. G in place of E, and the two rowdot vectors
  {
    const float sc = inv_t / (float)B;
    float d0 = 0.f, d1 = 0.f;
    const float ic0 = lane < B ? 1.f / C[lane] : 0.f, ic1 = lane + 64 < B ? 1.f / C[lane + 64] : 0.f;
    for (int i = wave; i < Bp; i += 4) {
      const float ir = i < B ? 1.f / R[i] : 0.f;
      float t0 = 0.f, t1 = 0.f;
      if (lane < Bp) {
        const float v = E[i * ldg + lane];
        const float g = (v * ic0 + v * ir - ((i == lane && i < B) ? 2.f : 0.f)) * sc;
        E[i * ldg + lane] = g;
        t0 = g * S[i * ldg + lane];
      }
      if (lane + 64 < Bp) {
        const float v = E[i * ldg + lane + 64];
        const float g = (v * ic1 + v * ir - ((i == lane + 64 && i < B) ? 2.f : 0.f)) * sc;
        E[i * ldg + lane + 64] = g;
        t1 = g * S[i * ldg + lane + 64];
      }
      d0 += t0;
      d1 += t1;
      const float r = wave_sum_f32(t0 + t1);
      if (lane == 0) rdA[i] = r;
    }
    if (lane < Bp) part[wave * Bp + lane] = d0;
    if (lane + 64 < Bp) part[wave * Bp + lane + 64] = d1;
  }
  __syncthreads();
  if (tid < Bp) rdP[tid] = part[tid] + part[Bp + tid] + part[2 * Bp + tid] + part[3 * Bp + tid];
  __syncthreads();

  // 4. one 32 x 32 tile of d im (mat 0) or d s (mat 1) per wave:  out = ((M @ X^) - Y^ * rowdot) * iy
  const int task = blockIdx.x * 4 + wave;
  if (task >= 2 * nb * 8) return;
  const int mat = task / (nb * 8), i0 = (task % (nb * 8)) / 8 * 32, c0 = (task % 8) * 32;
  const float* x = mat ? im : s;
  const float* ix = mat ? ia : ip;
  const float* y = mat ? s : im;
  const float* iy = mat ? ip : ia;
  const float* rd = mat ? rdP : rdA;
  float* out = mat ? g_s : g_im;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int gi = i0 + col;
  for (int k0 = 0; k0 < Bp; k0 += 32) {  // 16 steps per round, their operand loads all in flight
    float xv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) xv[q] = x[(size_t)min(k0 + 2 * q + half, B - 1) * kD + c0 + col];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int k = k0 + 2 * q + half;
      const float a = mat ? E[k * ldg + gi] : E[gi * ldg + k];  // zero beyond B (E is), so the clamped x rows do not count
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xv[q] * ix[k], acc, 0, 0, 0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  float yv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) yv[r] = y[(size_t)min(i0 + (r & 3) + 8 * (r >> 2) + 4 * half, B - 1) * kD + c0 + col];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    const float inv = iy[row];
    const float v = (acc[r] - yv[r] * inv * rd[row]) * inv;
    if (row < B) out[(size_t)row * kD + c0 + col] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Batches beyond the fused kernel's 128 rows — the GLOBAL contrastive matrix of data-parallel training (SURVEY.md §8e:
// every rank all-gathers the [B,256] embeddings of all ranks and evaluates the loss of the whole W*B batch; 8 x 64 = 512,
// up to 1,024 rows): the same arithmetic as a short chain of launches over a [B][B] matrix in HBM (4 MB at B = 1,024).
// Deterministic (no atomics): row / column sums are one wave per index.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cl_norm_kernel(const float* __restrict__ im, const float* __restrict__ s, int B,
                                                      float* __restrict__ ia, float* __restrict__ ip) {
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B) return;
  const float4 a = reinterpret_cast<const float4*>(im + (size_t)i * kD)[lane];
  const float4 p = reinterpret_cast<const float4*>(s + (size_t)i * kD)[lane];
  const float sa = wave_sum_f32(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w);
  const float sp = wave_sum_f32(p.x * p.x + p.y * p.y + p.z * p.z + p.w * p.w);
  if (lane == 0) { ia[i] = 1.f / sqrtf(sa); ip[i] = 1.f / sqrtf(sp); }
}

__global__ __launch_bounds__(64) void cl_sim_kernel(const float* __restrict__ im, const float* __restrict__ s, int B, int nb,
                                                    float inv_t, const float* __restrict__ ia, const float* __restrict__ ip,
                                                    float* __restrict__ E, float* __restrict__ diag) {
  const int lane = threadIdx.x, col = lane & 31, half = lane >> 5;
  const int i0 = (blockIdx.x / nb) * 32, j0 = (blockIdx.x % nb) * 32;
  const float4* ap = reinterpret_cast<const float4*>(im + (size_t)min(i0 + col, B - 1) * kD + half * 128);
  const float4* pp = reinterpret_cast<const float4*>(s + (size_t)min(j0 + col, B - 1) * kD + half * 128);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 8
  for (int q = 0; q < 32; ++q) {
    const float4 a = ap[q], p = pp[q];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, p.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, p.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, p.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, p.w, acc, 0, 0, 0);
  }
  const int j = j0 + col;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (i < B && j < B) {
      const float sim = acc[r] * ia[i] * ip[j];
      E[(size_t)i * B + j] = __expf(sim * inv_t);
      if (i == j) diag[i] = sim;
    }
  }
}

__global__ __launch_bounds__(256) void cl_sums_kernel(const float* __restrict__ E, int B, float* __restrict__ R, float* __restrict__ C) {
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B) return;
  float r = 0.f, c = 0.f;
  for (int j = lane; j < B; j += 64) {
    r += E[(size_t)i * B + j];
    c += E[(size_t)j * B + i];
  }
  r = wave_sum_f32(r);
  c = wave_sum_f32(c);
  if (lane == 0) { R[i] = r; C[i] = c; }
}

__global__ __launch_bounds__(256) void cl_loss_kernel(const float* __restrict__ R, const float* __restrict__ C,
                                                      const float* __restrict__ diag, int B, float inv_t, float* __restrict__ loss) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float v = 0.f;
  for (int i = tid; i < B; i += 256) v += __logf(C[i]) + __logf(R[i]) - 2.f * diag[i] * inv_t;
  v = wave_sum_f32(v);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  if (tid == 0) loss[0] = (red[0] + red[1] + red[2] + red[3]) / (float)B;
}

__global__ __launch_bounds__(256) void cl_g_kernel(float* __restrict__ E, const float* __restrict__ R, const float* __restrict__ C,
                                                   int B, float sc) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)B * B) return;
  const int i = (int)(e / B), j = (int)(e % B);
  const float v = E[e];
  E[e] = (v / C[j] + v / R[i] - (i == j ? 2.f : 0.f)) * sc;
}

__global__ __launch_bounds__(64) void cl_grad_kernel(const float* __restrict__ G, int B, int nb, const float* __restrict__ im,
                                                     const float* __restrict__ s, const float* __restrict__ ia,
                                                     const float* __restrict__ ip, float* __restrict__ g_im, float* __restrict__ g_s) {
  const int task = blockIdx.x, i0 = (task % nb) * 32;
  if (task < nb)
    grad_rowblock(G, B, false, B, i0, s, ip, im, ia, g_im, threadIdx.x);
  else
    grad_rowblock(G, B, true, B, i0, im, ia, s, ip, g_s, threadIdx.x);
}

static int loss_big_impl(t2l_ctx* ctx, const float* a, const float* p, int B, float temp, float* loss, float* ga, float* gp,
                         hipStream_t s) {
  const size_t need = ((size_t)B * B + 5 * (size_t)B) * sizeof(float);
  if (ctx->loss_ws_cap < need) {
    if (ctx->loss_ws) T2L_HIP(ctx, hipFree(ctx->loss_ws));
    ctx->loss_ws = nullptr;
    ctx->loss_ws_cap = 0;
    T2L_HIP(ctx, hipMalloc(&ctx->loss_ws, need));
    ctx->loss_ws_cap = need;
  }
  float* E = (float*)ctx->loss_ws;
  float *ia = E + (size_t)B * B, *ip = ia + B, *diag = ip + B, *R = diag + B, *C = R + B;
  const int nb = (B + 31) / 32, rows4 = (B + 3) / 4;
  event_begin(ctx, "contrastive_loss", s);
  hipLaunchKernelGGL(cl_norm_kernel, dim3(rows4), dim3(256), 0, s, a, p, B, ia, ip);
  hipLaunchKernelGGL(cl_sim_kernel, dim3(nb * nb), dim3(64), 0, s, a, p, B, nb, 1.0f / temp, ia, ip, E, diag);
  hipLaunchKernelGGL(cl_sums_kernel, dim3(rows4), dim3(256), 0, s, E, B, R, C);
  hipLaunchKernelGGL(cl_loss_kernel, dim3(1), dim3(256), 0, s, R, C, diag, B, 1.0f / temp, loss);
  if (ga) {
    hipLaunchKernelGGL(cl_g_kernel, dim3((unsigned)(((size_t)B * B + 255) / 256)), dim3(256), 0, s, E, R, C, B, 1.0f / (temp * (float)B));
    hipLaunchKernelGGL(cl_grad_kernel, dim3(2 * nb), dim3(64), 0, s, E, B, nb, a, p, ia, ip, ga, gp);
  }
  event_end(ctx, "contrastive_loss", s);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

int loss_impl(t2l_ctx* ctx, const float* a, const float* p, int B, float temp, float* loss, float* ga, float* gp,
              hipStream_t s) {
  if (B > T2L_MAX_LOSS_BATCH) return fail(ctx, T2L_EINVAL, "t2l_contrastive_loss: batch > 1024 not supported");
  if (B > 128) return loss_big_impl(ctx, a, p, B, temp, loss, ga, gp, s);
  const int Bp = (B + 31) / 32 * 32;
  static PerDeviceOnce attr_done;
  if (attr_done.need(ctx->device)) {
    T2L_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&contrastive_tiles_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    attr_done.mark(ctx->device);
  }
  event_begin(ctx, "contrastive_loss", s);
  const size_t lds = (2 * (size_t)Bp * (Bp + 1) + 10 * Bp) * sizeof(float);
  hipLaunchKernelGGL(contrastive_tiles_kernel, dim3(ga ? 4 * (Bp / 32) : 1), dim3(256), lds, s, a, p, B, 1.0f / temp, loss, ga, gp);
  event_end(ctx, "contrastive_loss", s);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

}  // namespace t2l
