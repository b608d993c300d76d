// Micro-probe (dev tool): what a TRIVIAL kernel costs at the paired scan's launch geometry (256 workgroups x 512 threads, 128 KiB of
// dynamic LDS, ~256 registers per lane) and at lighter ones — back-to-back launches on one stream, microseconds per launch.
// Build: hipcc --offload-arch=gfx950 -O3 tools/geom_probe.hip -o /tmp/geom_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>

template <int REGS>
__global__ __launch_bounds__(512, 1) void k(float* out, int n) {
  extern __shared__ float sm[];
  float acc[REGS];
#pragma unroll
  for (int i = 0; i < REGS; ++i) acc[i] = (float)(threadIdx.x + i);
  if (n > 1000000) {  // never true: keeps the registers alive
#pragma unroll
    for (int i = 0; i < REGS; ++i) acc[i] = acc[i] * sm[(threadIdx.x + i) & 255] + out[i];
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < REGS; ++i) s += acc[i];
  if (s == -12345.f) out[0] = s;
}

template <int REGS>
static void run(const char* name, int grid, int block, int lds, float* d, hipStream_t st) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<REGS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int N = 400;
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k<REGS>, dim3(grid), dim3(block), lds, st, d, 0);
  hipStreamSynchronize(st);
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k<REGS>, dim3(grid), dim3(block), lds, st, d, 0);
  hipStreamSynchronize(st);
  const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / N;
  printf("%-44s grid %4d x %3d, lds %6d B, %3d regs: %.2f us per launch\n", name, grid, block, lds, REGS, us);
}

int main() {
  float* d;
  hipMalloc(&d, 1 << 20);
  hipStream_t st;
  hipStreamCreate(&st);
  run<8>("one wave", 1, 64, 0, d, st);
  run<8>("256 x 512, no LDS, few registers", 256, 512, 0, d, st);
  run<8>("256 x 512, 128 KiB LDS, few registers", 256, 512, 128 * 1024, d, st);
  run<120>("256 x 512, 128 KiB LDS, ~128 registers", 256, 512, 128 * 1024, d, st);
  run<240>("256 x 512, 128 KiB LDS, ~256 registers", 256, 512, 128 * 1024, d, st);
  run<240>("256 x 256, 64 KiB LDS, ~256 registers", 256, 256, 64 * 1024, d, st);
  run<240>("512 x 256, 64 KiB LDS, ~256 registers", 512, 256, 64 * 1024, d, st);
  run<120>("1024 x 256, 32 KiB LDS, ~128 registers", 1024, 256, 32 * 1024, d, st);
  return 0;
}
