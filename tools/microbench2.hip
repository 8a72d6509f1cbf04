// What does the first dependent global load of a kernel cost, (a) for data the PREVIOUS kernel wrote from other CUs,
// (b) for read-only data, (c) with a 2.3 KB by-value kernel argument?   Prints ns (wall_clock64 is 100 MHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct Big { int v[580]; };
__global__ void k_write(float* x, int n, float val) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] = val + i;
}
template <bool BIG>
__global__ void k_read(const float* x, const float* w, float* out, unsigned long long* stamps, Big big, int H) {
  unsigned long long t0 = wall_clock64();
  __shared__ float s[512];
  const int tid = threadIdx.x; const size_t row = blockIdx.x;
  float a = x[row * H + tid] ;                 // produced by the previous kernel
  s[tid] = a; __syncthreads();
  unsigned long long t1 = wall_clock64();
  float b = w[tid];                              // read-only weights
  s[tid] += b; __syncthreads();
  unsigned long long t2 = wall_clock64();
  float c = BIG ? (float)big.v[(int)s[0] & 511] : 0.f;
  out[row * 256 + tid] = s[(tid + 1) & 255] + c;
  unsigned long long t3 = wall_clock64();
  if (tid == 0) { stamps[row * 4 + 0] = t1 - t0; stamps[row * 4 + 1] = t2 - t1; stamps[row * 4 + 2] = t3 - t2; stamps[row*4+3] = t0; }
}
int main() {
  const int B = 128, H = 400;
  float *x, *w, *out; unsigned long long* st;
  CK(hipMalloc(&x, B * H * 4)); CK(hipMalloc(&w, 4096)); CK(hipMalloc(&out, B * 256 * 4)); CK(hipMalloc(&st, B * 4 * 8));
  CK(hipMemset(w, 0, 4096));
  Big big; for (int i = 0; i < 580; ++i) big.v[i] = i;
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int variant = 0; variant < 3; ++variant) {
    double acc[3] = {0, 0, 0}; double span = 0;
    const int N = 100;
    for (int it = 0; it < N + 10; ++it) {
      if (variant != 1) hipLaunchKernelGGL(k_write, dim3((B * H + 255) / 256), dim3(256), 0, s, x, B * H, (float)it);
      if (variant == 2) hipLaunchKernelGGL(k_read<true>, dim3(B), dim3(256), 0, s, x, w, out, st, big, H);
      else hipLaunchKernelGGL(k_read<false>, dim3(B), dim3(256), 0, s, x, w, out, st, big, H);
      unsigned long long h[B * 4];
      CK(hipMemcpyAsync(h, st, sizeof(h), hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
      if (it >= 10) {
        unsigned long long mn = ~0ull, mx = 0;
        for (int r = 0; r < B; ++r) { for (int k = 0; k < 3; ++k) acc[k] += h[r * 4 + k] * 10.0 / B; mn = h[r*4+3] < mn ? h[r*4+3] : mn; mx = h[r*4+3] > mx ? h[r*4+3] : mx; }
        span += (mx - mn) * 10.0;
      }
    }
    const char* names[] = {"x written by previous kernel", "x static (no writer kernel)", "x written + 2.3KB by-value arg"};
    printf("%-34s first load+sync %7.0f ns | second (weights) %6.0f ns | store phase %6.0f ns | WG start skew %6.0f ns\n",
           names[variant], acc[0] / N, acc[1] / N, acc[2] / N, span / N);
  }
  return 0;
}
