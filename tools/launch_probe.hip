// Micro-probe (dev tool): GPU-side cost of a dependent kernel launch on this box — plain stream launches vs a captured
// hipGraph replay, for an empty kernel and for a tiny kernel with a memory dependence on its predecessor.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void empty_kernel() {}
__global__ void chain_kernel(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
int main() {
  float* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
  hipStream_t s; hipStreamCreate(&s);
  const int N = 200, REPS = 20;
  for (int mode = 0; mode < 2; ++mode) {
    auto body = [&](hipStream_t st) {
      for (int i = 0; i < N; ++i) {
        if (mode == 0) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st);
        else hipLaunchKernelGGL(chain_kernel, dim3(256), dim3(256), 0, st, d);
      }
    };
    body(s); hipStreamSynchronize(s);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < REPS; ++r) body(s);
    hipStreamSynchronize(s);
    double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
    printf("mode %d stream launches: %.2f us per kernel\n", mode, us / (N * REPS));
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    body(s);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < REPS; ++r) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
    printf("mode %d graph replay:    %.2f us per kernel\n", mode, us / (N * REPS));
  }
  return 0;
}
