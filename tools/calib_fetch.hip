// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the step's kernels
// (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern before trusting an absolute").
//   hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o gpurun_out/calib_fetch
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d ... -- gpurun_out/calib_fetch      (and WRITE_SIZE)
// Every kernel moves exactly N = 64 MiB (reads) or 64 MiB (writes) of a 256 MiB buffer, once.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// (a) 16 B per lane, fully coalesced (the optimizer-state reads: p, m, v)
__global__ void read16(const f32x4* p, float* sink, size_t n4) {
  f32x4 a = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) a += p[i];
  if (a[0] + a[1] + a[2] + a[3] == 12345.f) *sink = 1.f;
}
// (b) 4 B per lane, a wave reads 4 row segments of 64 B (rows `ld` floats apart): the operand fetch of the dW tiles
__global__ void read4seg(const float* p, float* sink, size_t rows, int ld) {
  const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  float a = 0.f;
  const int segs = ld / 16;  // 64-byte segments per row
  for (size_t t = wave; t < (rows / 4) * segs; t += nw) {
    const size_t r4 = t / segs, s = t % segs;
    a += p[(r4 * 4 + q) * ld + s * 16 + i];
  }
  if (a == 12345.f) *sink = 1.f;
}
// (c) plain 16-byte stores   (d) write-through (sc1) 16-byte stores, as store16_wt in the tile epilogues
__global__ void write16(f32x4* p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    p[i] = f32x4{1.f, 2.f, 3.f, 4.f};
}
__global__ void write16_wt(float* base, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    float* ptr = base + i * 4;
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(v) : "memory");
  }
}
// (e) 4-byte stores, 64 B segments (activation tiles written by the MFMA epilogues)
__global__ void write4(float* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.f;
}

int main() {
  const size_t bytes = (size_t)256 << 20, n = (size_t)64 << 20;  // touch 64 MiB of a 256 MiB buffer
  float *buf, *sink;
  hipMalloc(&buf, bytes);
  hipMalloc(&sink, 64);
  hipMemset(buf, 0, bytes);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(read16, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const f32x4*>(buf), sink, n / 16);
    hipLaunchKernelGGL(read4seg, dim3(2048), dim3(256), 0, 0, buf + (n / 4), sink, (n / 4) / 400, 400);
    hipLaunchKernelGGL(write16, dim3(2048), dim3(256), 0, 0, reinterpret_cast<f32x4*>(buf + 2 * (n / 4)), n / 16);
    hipLaunchKernelGGL(write16_wt, dim3(2048), dim3(256), 0, 0, buf + 3 * (n / 4), n / 16);
    hipLaunchKernelGGL(write4, dim3(2048), dim3(256), 0, 0, buf + 2 * (n / 4), n / 4);
  }
  hipDeviceSynchronize();
  printf("bytes per kernel: %zu (read4seg: %zu)\n", n, ((n / 4) / 400) / 4 * 4 * (size_t)(400 / 16) * 64);
  return 0;
}
