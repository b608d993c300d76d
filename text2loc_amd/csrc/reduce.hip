// a1 — per-object reductions over raw points, on the GPU.
//
// The reference recomputes these on the HOST inside every ObjectEncoder.forward call
// (datapreparation/kitti360pose/imports.py:28-41 called from models/object_encoder.py:79-84,121-141):
//   mean rgb (get_color_rgb), nearest of the 8 fitted colour centres (get_color_text -> known_colors index),
//   mean xyz (get_center), point count (len(obj.xyz)) — 62 % of its encode_objects wall time.
// Here: one wave per object streams the object's points once (24 B per point: the kernel is HBM-bound), accumulates in
// float64 (the reference's float32/float64 numpy sums differ from this by ~1e-6 relative; parity tolerance in the test),
// reduces across the wave on DPP and writes the packed per-object features the encoder consumes.
#include "search_dev.h"

namespace t2l {

struct ColorTable {
  float c[16][3];
  int row[16];  // COLORS index -> row of color_embedding (the reference's {name: i} dict, object_encoder.py:35)
  int n;
};

__global__ __launch_bounds__(256) void reduce_objects_kernel(const float* __restrict__ xyz, const float* __restrict__ rgb,
                                                             const int64_t* __restrict__ offsets, int n_objects,
                                                             ColorTable ct, float* __restrict__ out_rgb,
                                                             float* __restrict__ out_center, float* __restrict__ out_npts,
                                                             int32_t* __restrict__ out_color) {
  const int lane = threadIdx.x & 63;
  const int obj = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (obj >= n_objects) return;
  const int64_t p0 = offsets[obj], p1 = offsets[obj + 1];
  double sx = 0, sy = 0, sz = 0, sr = 0, sg = 0, sb = 0;
  for (int64_t p = p0 + lane; p < p1; p += 64) {
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
  if (lane == 0) {
    const double n = (double)(p1 - p0), inv = n > 0 ? 1.0 / n : 0.0;
    const double mr = sr * inv, mg = sg * inv, mb = sb * inv;
    out_center[obj * 3 + 0] = (float)(sx * inv);
    out_center[obj * 3 + 1] = (float)(sy * inv);
    out_center[obj * 3 + 2] = (float)(sz * inv);
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
}

int reduce_impl(t2l_ctx* ctx, const float* xyz, const float* rgb, const int64_t* offsets, int n_objects,
                const float* centers, const int32_t* rows, int n_colors, float* out_rgb, float* out_center,
                float* out_npts, int32_t* out_color, hipStream_t s) {
  if (n_colors < 1 || n_colors > 16) return fail(ctx, T2L_EINVAL, "t2l_reduce_objects: 1..16 colour centres");
  ColorTable ct;
  ct.n = n_colors;
  for (int k = 0; k < n_colors; ++k) {
    for (int j = 0; j < 3; ++j) ct.c[k][j] = centers[k * 3 + j];
    ct.row[k] = rows[k];
  }
  event_begin(ctx, "reduce_objects", s);
  hipLaunchKernelGGL(reduce_objects_kernel, dim3((n_objects + 3) / 4), dim3(256), 0, s, xyz, rgb, offsets, n_objects, ct,
                     out_rgb, out_center, out_npts, out_color);
  event_end(ctx, "reduce_objects", s);
  T2L_HIP(ctx, hipGetLastError());
  return T2L_OK;
}

}  // namespace t2l
