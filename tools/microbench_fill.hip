// How fast can a CU fill LDS from L2-resident data?  8 waves per workgroup, one workgroup per CU, every wave moves `per_step`
// 1-KiB pieces per step (lane = 16 bytes) from a hot 48 KB region into LDS, 2000 steps:
//   mode 0: global_load_lds_dwordx4 (LDS-DMA)          mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
//   mode 2: as 0, plus 18 ds_read_b128 per wave and step (the fragment reads of k_gemm_p3)     mode 3: as 1, plus the reads
//   mode 4: only the 18 reads
// hipcc --offload-arch=gfx950 -O3 tools/microbench_fill.hip -o tools/_bin/microbench_fill && tools/_bin/microbench_fill
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int PER>
__global__ __launch_bounds__(512) void k_fill(const f32x4* __restrict__ src, float* out, int steps) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[147456];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const f32x4* p = src + (size_t)blockIdx.x * 0 + wave * PER * 64 + lane;  // every CU reads the same 48 KB: L2 / L1 hot
  for (int s = 0; s < steps; ++s) {
    const int buf = (s % 3) * 49152;
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int u = 0; u < PER; ++u)
        __builtin_amdgcn_global_load_lds(p + u * 64, (__attribute__((address_space(3))) void*)(lds + buf + (wave * PER + u) * 1024), 16, 0, 0);
    }
    if (MODE == 1 || MODE == 3) {
      f32x4 v[PER];
#pragma unroll
      for (int u = 0; u < PER; ++u) v[u] = p[u * 64];
#pragma unroll
      for (int u = 0; u < PER; ++u) *reinterpret_cast<f32x4*>(lds + buf + (wave * PER + u) * 1024 + lane * 16) = v[u];
    }
    if (MODE >= 2) {
      const int rb = ((s + 1) % 3) * 49152;
#pragma unroll
      for (int u = 0; u < 18; ++u) acc += *reinterpret_cast<const f32x4*>(lds + rb + ((wave * 7 + u) % 48) * 1024 + lane * 16);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (acc[0] == 12345.f) out[threadIdx.x] = acc[1] + acc[2] + acc[3];
}
template <int MODE, int PER>
void run(const f32x4* src, float* out, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int steps = 2000;
  hipLaunchKernelGGL((k_fill<MODE, PER>), dim3(256), dim3(512), 0, 0, src, out, 200);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_fill<MODE, PER>), dim3(256), dim3(512), 0, 0, src, out, steps);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ns_step = ms * 1e6 / steps;
  printf("%-58s %d KiB / step / CU: %7.1f ns per step, %6.1f ns per 1-KiB piece, %5.1f GB/s per CU\n", name, 8 * PER, ns_step,
         ns_step / (8 * PER), MODE == 4 ? 0.0 : 8.0 * PER * 1024 / ns_step);
}
int main() {
  f32x4* src; float* out; hipMalloc(&src, 1 << 20); hipMalloc(&out, 4096); hipMemset(src, 0, 1 << 20);
  run<0, 6>(src, out, "LDS-DMA (global_load_lds_dwordx4)");
  run<1, 6>(src, out, "global_load_dwordx4 -> ds_write_b128");
  run<0, 3>(src, out, "LDS-DMA, half the pieces");
  run<1, 3>(src, out, "registers, half the pieces");
  run<2, 6>(src, out, "LDS-DMA + 18 ds_read_b128 per wave");
  run<3, 6>(src, out, "registers + 18 ds_read_b128 per wave");
  run<4, 6>(src, out, "only the 18 ds_read_b128 per wave");
  return 0;
}
