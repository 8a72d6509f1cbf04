// Dev microbenchmark: what does COLD straight-line code cost on gfx950?  A fully unrolled dependent FMA chain of N steps
// (>= 8 bytes of code per step) is executed TWICE inside one kernel (outer loop not unrolled): the first pass fetches every
// instruction line for the first time in this launch, the second pass finds them in the instruction cache.  Each pass is
// timed with wall_clock64() (100 MHz) by lane 0 of the first and the last workgroup.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_icache.hip -o tools/microbench_icache_bin && tools/microbench_icache_bin
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ unsigned long long g_t[16];

template <int N>
__global__ void chain(float* out, float a, float b) {
  float x = threadIdx.x * 1e-3f;
#pragma unroll 1
  for (int rep = 0; rep < 3; ++rep) {
    asm volatile("" : "+v"(x));
    const unsigned long long t0 = wall_clock64();
    asm volatile("" : "+v"(x));
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fmaf(x, a, b);
    asm volatile("" : "+v"(x));
    const unsigned long long t1 = wall_clock64();
    asm volatile("" : "+v"(x));
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) g_t[rep * 2 + (blockIdx.x ? 1 : 0)] = t1 - t0;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

__global__ void flusher(float* out) {  // unrelated code + data between the probes
  float x = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 512; ++i) x = __builtin_fmaf(x, 1.0001f, 3.0f);
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

template <int N>
void run(float* d, int grid, int threads) {
  unsigned long long t[16];
  double acc[6] = {0, 0, 0, 0, 0, 0};
  const int R = 50;
  for (int r = 0; r < R; ++r) {
    hipLaunchKernelGGL(flusher, dim3(512), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((chain<N>), dim3(grid), dim3(threads), 0, 0, d, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_t), sizeof(t));
    for (int i = 0; i < 6; ++i) acc[i] += (double)t[i] * 10.0;  // ns
  }
  printf("N=%5d (%5.1f KB) grid=%4d x %3d thr | first wg: pass0 %7.0f pass1 %7.0f pass2 %7.0f ns | last wg: pass0 %7.0f pass1 %7.0f ns | cold-warm %6.0f ns = %5.0f ns/KB\n",
         N, N * 8 / 1024.0, grid, threads, acc[0] / R, acc[2] / R, acc[4] / R, acc[1] / R, acc[3] / R,
         (acc[0] - acc[2]) / R, (acc[0] - acc[2]) / R / (N * 8 / 1024.0));
}

int main() {
  float* d;
  (void)hipMalloc(&d, 1 << 24);
  run<128>(d, 1, 64);
  run<512>(d, 1, 64);
  run<2048>(d, 1, 64);
  run<128>(d, 256, 64);
  run<512>(d, 256, 64);
  run<2048>(d, 256, 64);
  run<512>(d, 256, 512);
  run<512>(d, 512, 512);
  return 0;
}
