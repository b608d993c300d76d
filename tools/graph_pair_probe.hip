// dev: would a captured hipGraph of the two search launches (scan -> re-rank) shorten the step? Two stand-in kernels that spin for the real
// kernels' in-kernel times (26 us on 256 x 512 threads, 9 us on 1024 x 256 threads; the second reads what the first wrote), 400 steps:
// (a) two stream launches per step, (b) one hipGraphLaunch per step with both nodes' parameters updated every step (what t2l_search
// would have to do: queries, outputs, counters and sequence numbers change per call).
// hipcc --offload-arch=gfx950 -O3 tools/graph_pair_probe.hip -o /tmp/graph_pair_probe && /tmp/graph_pair_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin_a(int* buf, int ticks, int seq) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) {}
  if (threadIdx.x == 0) buf[blockIdx.x] = seq;
}
__global__ void spin_b(const int* buf, int* out, int ticks, int seq) {
  const int v = buf[blockIdx.x & 255];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) {}
  if (threadIdx.x == 0) out[blockIdx.x] = v + seq;
}
int main() {
  int *buf, *out;
  CK(hipMalloc(&buf, 4096)); CK(hipMalloc(&out, 8192));
  hipStream_t s; CK(hipStreamCreate(&s));
  const int ta = 2600, tb = 900, N = 400;
  auto now = [] { return std::chrono::high_resolution_clock::now(); };
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 50; ++i) { hipLaunchKernelGGL(spin_a, dim3(256), dim3(512), 0, s, buf, ta, i); hipLaunchKernelGGL(spin_b, dim3(1024), dim3(256), 0, s, buf, out, tb, i); }
    CK(hipStreamSynchronize(s));
    auto t0 = now();
    for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(spin_a, dim3(256), dim3(512), 0, s, buf, ta, i); hipLaunchKernelGGL(spin_b, dim3(1024), dim3(256), 0, s, buf, out, tb, i); }
    CK(hipStreamSynchronize(s));
    const double us_stream = std::chrono::duration<double, std::micro>(now() - t0).count() / N;
    // graph with two kernel nodes
    hipGraph_t g; CK(hipGraphCreate(&g, 0));
    int seq = 0, tav = ta, tbv = tb;
    void* aa[] = {&buf, &tav, &seq};
    void* ab[] = {&buf, &out, &tbv, &seq};
    hipKernelNodeParams pa{}, pb{};
    pa.func = (void*)spin_a; pa.gridDim = dim3(256); pa.blockDim = dim3(512); pa.kernelParams = aa;
    pb.func = (void*)spin_b; pb.gridDim = dim3(1024); pb.blockDim = dim3(256); pb.kernelParams = ab;
    hipGraphNode_t na, nb;
    CK(hipGraphAddKernelNode(&na, g, nullptr, 0, &pa));
    CK(hipGraphAddKernelNode(&nb, g, &na, 1, &pb));
    hipGraphExec_t ex; CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    for (int i = 0; i < 50; ++i) { seq = i; CK(hipGraphExecKernelNodeSetParams(ex, na, &pa)); CK(hipGraphExecKernelNodeSetParams(ex, nb, &pb)); CK(hipGraphLaunch(ex, s)); }
    CK(hipStreamSynchronize(s));
    t0 = now();
    double host = 0;
    for (int i = 0; i < N; ++i) {
      auto h0 = now();
      seq = i; CK(hipGraphExecKernelNodeSetParams(ex, na, &pa)); CK(hipGraphExecKernelNodeSetParams(ex, nb, &pb)); CK(hipGraphLaunch(ex, s));
      host += std::chrono::duration<double, std::micro>(now() - h0).count();
    }
    CK(hipStreamSynchronize(s));
    const double us_graph = std::chrono::duration<double, std::micro>(now() - t0).count() / N;
    printf("stream launches %.2f us/step   graph (params updated) %.2f us/step   host per graph step %.2f us\n", us_stream, us_graph, host / N);
    CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
  }
  return 0;
}
