// Lone-wave instruction issue rate: dependent vs independent FMA chains, hot loop vs cold straight-line code.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
#define R4096(x) R16(R256(x))
__global__ void k_cold_dep(float* o, unsigned long long* t, float a0) {
  float a = a0; unsigned long long t0 = wall_clock64();
  R4096(a = fmaf(a, 1.0001f, 0.5f);)
  unsigned long long t1 = wall_clock64(); o[threadIdx.x] = a; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_hot_dep(float* o, unsigned long long* t, float a0) {
  float a = a0; unsigned long long t0 = wall_clock64();
  for (int i = 0; i < 16; ++i) { R256(a = fmaf(a, 1.0001f, 0.5f);) }
  unsigned long long t1 = wall_clock64(); o[threadIdx.x] = a; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_hot_indep(float* o, unsigned long long* t, float a0) {
  float a = a0, b = a0 + 1, c = a0 + 2, d = a0 + 3; unsigned long long t0 = wall_clock64();
  for (int i = 0; i < 16; ++i) { R256(a = fmaf(a, 1.0001f, 0.5f); b = fmaf(b, 1.0001f, 0.5f); c = fmaf(c, 1.0001f, 0.5f); d = fmaf(d, 1.0001f, 0.5f);) }
  unsigned long long t1 = wall_clock64(); o[threadIdx.x] = a + b + c + d; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_hot_trans(float* o, unsigned long long* t, float a0) {
  float a = a0; unsigned long long t0 = wall_clock64();
  for (int i = 0; i < 16; ++i) { R256(a = __builtin_amdgcn_rcpf(a) + 0.5f;) }
  unsigned long long t1 = wall_clock64(); o[threadIdx.x] = a; if (threadIdx.x == 0) t[0] = t1 - t0;
}
int main() {
  float* o; unsigned long long* t; CK(hipMalloc(&o, 1024)); CK(hipMalloc(&t, 64));
  unsigned long long h;
#define RUN(K, NAME, N) for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(K, dim3(1), dim3(64), 0, 0, o, t, 1.0f); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost)); printf("%-34s run %d: %6.2f ns / instruction\n", NAME, rep, h * 10.0 / (N)); }
  RUN(k_cold_dep, "4096 dependent fma, straight-line", 4096)
  RUN(k_hot_dep, "4096 dependent fma, 16x256 loop", 4096)
  RUN(k_hot_indep, "16384 fma in 4 chains, loop", 16384)
  RUN(k_hot_trans, "4096 dependent rcp+add pairs", 8192)
  return 0;
}
