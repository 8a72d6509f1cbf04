// Probe of two gfx950 primitives the plane-operand contraction (k_gemm_p3) relies on; prints what the hardware does.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_tr_dma.hip -o tools/_bin/probe_tr_dma && tools/_bin/probe_tr_dma
// 1. ds_read_b64_tr_b16: every lane reads 8 bytes at its own address; which (source lane, element) ends up where?
// 2. global_load_lds_dwordx4 (LDS-DMA): per-lane GLOBAL address, LDS destination = uniform base + lane * 16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k_tr(unsigned short* out, int stride_bytes) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = i;  // value = element index (2-byte units)
  __syncthreads();
  const int lane = threadIdx.x;
  // stride_bytes == 0: linear (lane * 8 bytes).  else: 16-lane block g = lane >> 4 reads a [4 k][16 col] block at k rows
  // 4 g .. 4 g + 3, row stride `stride_bytes`; lane j of the block -> row (j >> 2), col quad (j & 3)
  int off;
  if (stride_bytes == 0) off = lane * 8;
  else { const int j = lane & 15, g = lane >> 4; off = (4 * g + (j >> 2)) * stride_bytes + (j & 3) * 8; }
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)((char*)lds + off));
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)v[e];
}
__global__ void k_dma(const unsigned int* src, unsigned int* out) {
  __shared__ __attribute__((aligned(16))) unsigned int lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const int lane = threadIdx.x;
  // lane L fetches the 16 bytes of source chunk (L ^ 5) -- a permuted source -- into LDS chunk L of the second KiB
  const unsigned int* g = src + ((lane ^ 5) * 4);
  __builtin_amdgcn_global_load_lds(g, (unsigned int __attribute__((address_space(3)))*)(lds + 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
  unsigned short* d; hipMalloc(&d, 256 * 2);
  std::vector<unsigned short> h(256);
  for (int stride : {0, 32, 64, 256}) {
    hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, d, stride);
    hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b16, %s (values = 2-byte element index in LDS)\n", stride ? "[4k][16col] blocks" : "linear lane*8");
    if (stride) printf(" row stride %d bytes = %d elements\n", stride, stride / 2);
    for (int l = 0; l < 64; ++l) {
      printf(" lane %2d:", l);
      for (int e = 0; e < 4; ++e) printf(" %5d", h[l * 4 + e]);
      if (stride) { printf("   (row,col):"); for (int e = 0; e < 4; ++e) printf(" (%d,%d)", h[l*4+e] / (stride/2), h[l*4+e] % (stride/2)); }
      printf("\n");
    }
  }
  unsigned int *s, *o; hipMalloc(&s, 4096); hipMalloc(&o, 4096);
  std::vector<unsigned int> hs(1024), ho(1024);
  for (int i = 0; i < 1024; ++i) hs[i] = i;
  hipMemcpy(s, hs.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_dma, dim3(1), dim3(64), 0, 0, s, o);
  hipMemcpy(ho.data(), o, 4096, hipMemcpyDeviceToHost);
  int ok = 1;
  for (int L = 0; L < 64; ++L) for (int w = 0; w < 4; ++w) if (ho[256 + L * 4 + w] != (unsigned)((L ^ 5) * 4 + w)) ok = 0;
  for (int i = 0; i < 256; ++i) if (ho[i] != 0xdeadbeefu) ok = 0;
  printf("global_load_lds_dwordx4: LDS chunk L <- source chunk (L ^ 5): %s; first words of chunks 0..3: %u %u %u %u\n", ok ? "AS EXPECTED" : "MISMATCH",
         ho[256], ho[260], ho[264], ho[268]);
  return 0;
}
