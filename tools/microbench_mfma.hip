// Sustained rate of the matrix pipe under nothing but MFMAs (no memory traffic): what a contraction kernel can reach at best.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_mfma.hip -o /tmp/mb_mfma && /tmp/mb_mfma
// Per workgroup WAVES waves, each issuing ITERS x 16 independent v_mfma_f32_16x16x32_bf16 (or 16x16x4_f32); 256 x k workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// RND: four different pseudo-random operand registers per side instead of one smooth one (the data a real contraction sees:
// the pipe's power, and with it the clock, depends on how many operand bits toggle between consecutive instructions)
template <bool BF16, bool RND>
__global__ void k_mfma(float* out, int iters) {
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a[4], b[4];
  float fa[4], fb[4];
  unsigned h = 0x9e3779b9u * (threadIdx.x + 1) + blockIdx.x;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u;
      a[q][i] = RND ? (__bf16)(((int)(h >> 8) & 0xffff) * (1.0f / 32768.f) - 1.0f) : (__bf16)(1.0f + threadIdx.x * 0.001f);
      h = h * 1664525u + 1013904223u;
      b[q][i] = RND ? (__bf16)(((int)(h >> 8) & 0xffff) * (1.0f / 32768.f) - 1.0f) : (__bf16)(0.5f + i * 0.01f);
    }
    h = h * 1664525u + 1013904223u;
    fa[q] = RND ? ((int)(h >> 8) & 0xffff) * (1.0f / 32768.f) - 1.0f : 1.0f + threadIdx.x * 0.001f;
    h = h * 1664525u + 1013904223u;
    fb[q] = RND ? ((int)(h >> 8) & 0xffff) * (1.0f / 32768.f) - 1.0f : 0.5f;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (BF16) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i & 3], fb[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) s += acc[i];
  if (s[0] == 12345.678f) out[threadIdx.x] = s[1];  // (keeps the MFMAs alive)
}

template <bool BF16, bool RND>
static void run(const char* name, int wgs, int waves, int iters, double flop_per_mfma) {
  float* out;
  hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_mfma<BF16, RND>), dim3(wgs), dim3(64 * waves), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wgs * waves * iters * 16 * flop_per_mfma;
    printf("%-6s wgs %5d waves/wg %2d iters %6d: %8.3f ms  %8.1f TFLOP/s\n", name, wgs, waves, iters, ms, flops / ms / 1e9);
  }
  hipFree(out);
}

int main() {
  const double fb = 2.0 * 16 * 16 * 32, ff = 2.0 * 16 * 16 * 4;
  for (int waves : {4, 8}) {
    run<true, false>("bf16", 256, waves, 4000, fb);     // ~ms-long: the clock the kernel sees in a short burst
    run<true, false>("bf16", 256, waves, 100000, fb);   // ~25 ms+: the sustained clock
    run<true, true>("bf16r", 256, waves, 4000, fb);
    run<true, true>("bf16r", 256, waves, 100000, fb);
  }
  for (int waves : {4, 8}) {
    run<false, false>("f32", 256, waves, 50000, ff);
    run<false, true>("f32r", 256, waves, 4000, ff);
    run<false, true>("f32r", 256, waves, 50000, ff);
  }
  return 0;
}
