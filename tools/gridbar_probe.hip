// Micro-probe (dev tool): what one DEPENDENT phase costs on this box as (a) a kernel launch on a stream and (b) a phase of
// ONE persistent kernel separated from its predecessor by a grid-wide barrier (agent-scope release -> atomic counter ->
// agent-scope acquire). Every phase reads what OTHER workgroups (other XCDs) wrote in the phase before, so a stale L2 line
// shows up as a wrong checksum. Decides whether the training step's ~78 dependent launches are worth folding into a
// persistent "phase program" kernel.   hipcc --offload-arch=gfx950 -O3 tools/gridbar_probe.hip -o /tmp/gridbar_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>

constexpr int kThreads = 256;

__device__ __forceinline__ void phase_body(const float* __restrict__ in, float* __restrict__ out, int n, int vb, int nvb, int work) {
  // virtual block vb reads the slice of block (vb + 37) % nvb — produced on another CU/XCD — and writes its own slice
  const int per = n / nvb;
  const int src = ((vb + 37) % nvb) * per, dst = vb * per;
  for (int i = threadIdx.x; i < per; i += kThreads) {
    float v = in[src + i];
    for (int w = 0; w < work; ++w) v = v * 1.0000001f + 1e-7f;
    out[dst + i] = v + 1.f;
  }
}

__global__ __launch_bounds__(kThreads) void phase_kernel(const float* in, float* out, int n, int work) {
  phase_body(in, out, n, blockIdx.x, gridDim.x, work);
}

// grid barrier: all threads of the workgroup have finished their stores -> one thread releases at agent scope, arrives, spins,
// acquires. `bar` is a monotonically increasing counter (target = phase * gridDim.x).
__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__global__ __launch_bounds__(kThreads) void persistent_kernel(float* a, float* b, int n, int nvb, int phases, int work, unsigned* bar,
                                                              unsigned base) {
  float* in = a;
  float* out = b;
  for (int p = 0; p < phases; ++p) {
    for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x) phase_body(in, out, n, vb, nvb, work);
    grid_barrier(bar, base + (unsigned)(p + 1) * gridDim.x);
    float* t = in; in = out; out = t;
  }
}

int main() {
  const int n = 1 << 20, phases = 200, reps = 10;
  float *a, *b;
  unsigned* bar;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&bar, 256);
  hipStream_t s; hipStreamCreate(&s);
  std::vector<float> h(n);
  for (int nvb : {256, 512, 1024}) {
    for (int work : {0, 64}) {
      // (a) launches
      hipMemsetAsync(a, 0, n * 4, s);
      auto run_launches = [&]() {
        float *in = a, *out = b;
        for (int p = 0; p < phases; ++p) {
          hipLaunchKernelGGL(phase_kernel, dim3(nvb), dim3(kThreads), 0, s, in, out, n, work);
          std::swap(in, out);
        }
      };
      run_launches(); hipStreamSynchronize(s);
      auto t0 = std::chrono::high_resolution_clock::now();
      for (int r = 0; r < reps; ++r) run_launches();
      hipStreamSynchronize(s);
      double us_l = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / (phases * reps);
      // (b) persistent, grids of 256 / 512 workgroups
      for (int grid : {256, 512}) {
        if (grid > nvb) continue;
        hipMemsetAsync(a, 0, n * 4, s);
        hipMemsetAsync(bar, 0, 256, s);
        unsigned base = 0;
        hipLaunchKernelGGL(persistent_kernel, dim3(grid), dim3(kThreads), 0, s, a, b, n, nvb, phases, work, bar, base);
        base += (unsigned)phases * grid;
        hipStreamSynchronize(s);
        t0 = std::chrono::high_resolution_clock::now();
        for (int r = 0; r < reps; ++r) {
          hipLaunchKernelGGL(persistent_kernel, dim3(grid), dim3(kThreads), 0, s, a, b, n, nvb, phases, work, bar, base);
          base += (unsigned)phases * grid;
        }
        hipStreamSynchronize(s);
        double us_p = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / (phases * reps);
        // correctness: after (1 + reps) * phases phases from zeros every element must be (1 + reps) * phases (work == 0)
        hipMemcpy(h.data(), a, n * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        const float want = (float)((1 + reps) * phases);
        if (work == 0)
          for (int i = 0; i < n; ++i) bad += h[i] != want;
        printf("virtual blocks %4d work %2d: launch chain %.2f us/phase | persistent grid %d: %.2f us/phase (stale elements: %d)\n", nvb, work,
               us_l, grid, us_p, bad);
      }
    }
  }
  return 0;
}
