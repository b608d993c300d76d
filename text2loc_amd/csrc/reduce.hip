// a1 — per-object reductions over raw points, on the GPU.
//
// The reference recomputes these on the HOST inside every ObjectEncoder.forward call
// (datapreparation/kitti360pose/imports.py:28-41 called from models/object_encoder.py:79-84,121-141):
//   mean rgb (get_color_rgb), nearest of the 8 fitted colour centres (get_color_text -> known_colors index),
//   mean xyz (get_center), point count (len(obj.xyz)) — 62 % of its encode_objects wall time.
// Here: one wave per run of <= 4096 points streams them once (24 B per point: the kernel is HBM-bound), accumulates in
// float64 (the reference's float32/float64 numpy sums differ from this by ~1e-6 relative; parity tolerance in the test),
// reduces across the wave on DPP and writes the packed per-object features the encoder consumes.
#include <algorithm>
#include <vector>

#include "search_dev.h"

namespace t2l {

struct ColorTable {
  float c[16][3];
  int row[16];  // COLORS index -> row of color_embedding (the reference's {name: i} dict, object_encoder.py:35)
  int n;
};

struct WorkItem {  // a contiguous run of <= kChunk points of one object
  int64_t begin;
  int32_t count;
  int32_t obj;
};
constexpr int kChunk = 4096;

// one wave per work item: 6 partial float64 sums -> atomicAdd into acc[obj][6] (objects larger than kChunk points are
// split so that the long tail of the size distribution does not serialise on one wave)
__global__ __launch_bounds__(256) void reduce_partial_kernel(const float* __restrict__ xyz, const float* __restrict__ rgb,
                                                             const WorkItem* __restrict__ items, int n_items,
                                                             double* __restrict__ acc) {
  const int lane = threadIdx.x & 63;
  const int it = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (it >= n_items) return;
  const WorkItem w = items[it];
  const int64_t p0 = w.begin, p1 = w.begin + w.count;
  double sx = 0, sy = 0, sz = 0, sr = 0, sg = 0, sb = 0;
  int64_t p = p0 + lane;
  // four points per lane and array in flight (8 x 12-byte loads before the first add: a lone wave otherwise waits out every
  // HBM round trip with 24 bytes per lane outstanding)
  for (; p + 192 < p1; p += 256) {
    float a[4][3], c[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* ap = xyz + (p + 64 * u) * 3;
      const float* cp = rgb + (p + 64 * u) * 3;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        a[u][j] = ap[j];
        c[u][j] = cp[j];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      sx += (double)a[u][0];
      sy += (double)a[u][1];
      sz += (double)a[u][2];
      sr += (double)c[u][0];
      sg += (double)c[u][1];
      sb += (double)c[u][2];
    }
  }
  for (; p < p1; p += 64) {
    const float* a = xyz + p * 3;
    const float* c = rgb + p * 3;
    sx += (double)a[0];
    sy += (double)a[1];
    sz += (double)a[2];
    sr += (double)c[0];
    sg += (double)c[1];
    sb += (double)c[2];
  }
  sx = wave_sum_f64(sx);
  sy = wave_sum_f64(sy);
  sz = wave_sum_f64(sz);
  sr = wave_sum_f64(sr);
  sg = wave_sum_f64(sg);
  sb = wave_sum_f64(sb);
  if (lane < 6) {
    const double v = lane == 0 ? sx : lane == 1 ? sy : lane == 2 ? sz : lane == 3 ? sr : lane == 4 ? sg : sb;
    atomicAdd(acc + (size_t)w.obj * 6 + lane, v);
  }
}

__global__ __launch_bounds__(256) void reduce_finalize_kernel(const double* __restrict__ acc,
                                                              const int64_t* __restrict__ offsets, int n_objects,
                                                              ColorTable ct, float* __restrict__ out_rgb,
                                                              float* __restrict__ out_center, float* __restrict__ out_npts,
                                                              int32_t* __restrict__ out_color) {
  const int obj = blockIdx.x * 256 + threadIdx.x;
  if (obj >= n_objects) return;
  const double n = (double)(offsets[obj + 1] - offsets[obj]), inv = n > 0 ? 1.0 / n : 0.0;
  const double* a = acc + (size_t)obj * 6;
  const double mr = a[3] * inv, mg = a[4] * inv, mb = a[5] * inv;
  out_center[obj * 3 + 0] = (float)(a[0] * inv);
  out_center[obj * 3 + 1] = (float)(a[1] * inv);
  out_center[obj * 3 + 2] = (float)(a[2] * inv);
  out_rgb[obj * 3 + 0] = (float)mr;
  out_rgb[obj * 3 + 1] = (float)mg;
  out_rgb[obj * 3 + 2] = (float)mb;
  out_npts[obj] = (float)n;
  int best = 0;
  double bd = 1e300;
  for (int k = 0; k < ct.n; ++k) {  // np.argmin(np.linalg.norm(mean - COLORS, axis=1)): first minimum wins
    const double dr = mr - ct.c[k][0], dg = mg - ct.c[k][1], db = mb - ct.c[k][2];
    const double d = dr * dr + dg * dg + db * db;
    if (d < bd) {
      bd = d;
      best = k;
    }
  }
  out_color[obj] = ct.row[best];
}

int reduce_impl(t2l_ctx* ctx, const float* xyz, const float* rgb, const int64_t* offsets_host, int n_objects,
                const float* centers, const int32_t* rows, int n_colors, float* out_rgb, float* out_center,
                float* out_npts, int32_t* out_color, hipStream_t s) {
  if (n_colors < 1 || n_colors > 16) return fail(ctx, T2L_EINVAL, "t2l_reduce_objects: 1..16 colour centres");
  ColorTable ct;
  ct.n = n_colors;
  for (int k = 0; k < n_colors; ++k) {
    for (int j = 0; j < 3; ++j) ct.c[k][j] = centers[k * 3 + j];
    ct.row[k] = rows[k];
  }
  std::vector<WorkItem> items;
  items.reserve((size_t)n_objects + 64);
  for (int o = 0; o < n_objects; ++o) {
    const int64_t b = offsets_host[o], e = offsets_host[o + 1];
    if (e < b) return fail(ctx, T2L_EINVAL, "t2l_reduce_objects: point_offsets must be non-decreasing");
    for (int64_t p = b; p < e; p += kChunk) items.push_back(WorkItem{p, (int32_t)std::min<int64_t>(kChunk, e - p), o});
  }
  const size_t need = items.size() * sizeof(WorkItem) + (size_t)(n_objects + 1) * sizeof(int64_t) +
                      (size_t)n_objects * 6 * sizeof(double) + 64;
  if (need > ctx->reduce_ws_cap) {
    if (ctx->reduce_ws) (void)hipFree(ctx->reduce_ws);
    ctx->reduce_ws = nullptr;
    ctx->reduce_ws_cap = 0;
    T2L_HIP(ctx, hipMalloc(&ctx->reduce_ws, need));
    ctx->reduce_ws_cap = need;
  }
  char* ws = reinterpret_cast<char*>(ctx->reduce_ws);
  double* acc = reinterpret_cast<double*>(ws);
  int64_t* d_off = reinterpret_cast<int64_t*>(ws + (size_t)n_objects * 6 * sizeof(double));
  WorkItem* d_items = reinterpret_cast<WorkItem*>(reinterpret_cast<char*>(d_off) + (size_t)(n_objects + 1) * sizeof(int64_t));
  T2L_HIP(ctx, hipMemsetAsync(acc, 0, (size_t)n_objects * 6 * sizeof(double), s));
  T2L_HIP(ctx, hipMemcpyAsync(d_off, offsets_host, (size_t)(n_objects + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
  T2L_HIP(ctx, hipMemcpyAsync(d_items, items.data(), items.size() * sizeof(WorkItem), hipMemcpyHostToDevice, s));
  T2L_HIP(ctx, hipStreamSynchronize(s));  // `items` is a local: the copy must be done before it goes away
  const int n_items = (int)items.size();
  event_begin(ctx, "reduce_objects", s);
  if (n_items > 0)
    hipLaunchKernelGGL(reduce_partial_kernel, dim3((n_items + 3) / 4), dim3(256), 0, s, xyz, rgb, d_items, n_items, acc);
  hipLaunchKernelGGL(reduce_finalize_kernel, dim3((n_objects + 255) / 256), dim3(256), 0, s, acc, d_off, n_objects, ct,
                     out_rgb, out_center, out_npts, out_color);
  event_end(ctx, "reduce_objects", s);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

}  // namespace t2l
