// Microbenchmarks that bound the design of a latency-bound step: shader clock under light load, dependent-load
// round trip (L2 / MALL), cost of a kernel boundary inside a graph.   hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o /tmp/mb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_clock(unsigned long long* out, int iters) {
  unsigned long long c0 = __builtin_readcyclecounter();  // s_memtime: shader clock
  unsigned long long r0 = wall_clock64();              // s_memrealtime: 100 MHz
  float a = threadIdx.x;
  for (int i = 0; i < iters; ++i) a = fmaf(a, 1.0001f, 0.5f);
  unsigned long long c1 = __builtin_readcyclecounter();
  unsigned long long r1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (unsigned long long)a; }
}
__global__ void k_chase(const int* next, int start, int hops, unsigned long long* out) {
  unsigned long long r0 = wall_clock64();
  int p = start;
  for (int i = 0; i < hops; ++i) p = next[p];
  unsigned long long r1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = r1 - r0; out[1] = p; }
}
__global__ void k_touch(float* p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] += 1.f; }

int main() {
  unsigned long long* d; CK(hipMalloc(&d, 64)); unsigned long long h[4];
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, s, d, 200000);
    CK(hipMemcpyAsync(h, d, 32, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
    printf("shader clock during a 1-wave spin: %.0f MHz (cycles %llu, 100MHz ticks %llu)\n", h[0] / (h[1] / 100.0), h[0], h[1]);
  }
  // pointer chase over 1 MB (L2-resident) and 64 MB (MALL) with stride permutations
  for (size_t bytes : {size_t(1) << 20, size_t(64) << 20}) {
    int n = bytes / 4; std::vector<int> nx(n);
    int stride = 4099 * 16;  // odd multiple of a cache line in ints
    for (int i = 0; i < n; ++i) nx[i] = (int)(((long long)i + stride) % n);
    int* dn; CK(hipMalloc(&dn, bytes)); CK(hipMemcpy(dn, nx.data(), bytes, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, s, dn, 0, 2000, d);
      CK(hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
      printf("dependent load chain over %zu MB: %.1f ns per hop\n", bytes >> 20, h[0] * 10.0 / 2000);
    }
    CK(hipFree(dn));
  }
  // kernel boundary cost inside a graph: N dependent launches (empty, and a 1024-WG touch kernel)
  float* buf; CK(hipMalloc(&buf, 1024 * 256 * 4)); CK(hipMemset(buf, 0, 1024 * 256 * 4));
  for (int variant = 0; variant < 2; ++variant) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 200; ++i) {
      if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
      else hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, s, buf, 1024 * 256);
    }
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("graph of 200 dependent %s kernels: %.2f us per kernel\n", variant == 0 ? "empty" : "1024-WG touch(1MB)", ms * 1e3 / 2000);
  }
  return 0;
}
