// Dev tool: can the NEXT launch of a dependent chain be in flight while the current one runs?
// A chain of N "phases" (256 workgroups x 256 threads, one per CU).  Each phase: prologue = load 32 KB of L2-resident
// weights into LDS (independent of the previous phase) | dependency on ALL workgroups of the previous phase | read 4 KB
// of its output, ~1 us of dependent FMAs, write 4 KB.
//   mode 0: one stream, the dependency is the kernel boundary (what the step does today)
//   mode 1: two streams, phases alternate between them (phase k is stream-ordered after k-2); the dependency on phase
//           k-1 is an in-memory counter: producers store write-through (sc1), drain, add to their XCD's arrival word;
//           consumers poll the 8 words (relaxed sc1 loads, bounded spin) and read the payload with sc1 loads
// Both eager and as a HIP graph (mode 1: fork/join capture).  Prints us per phase.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kWG = 256, kThr = 256, kWeights = 8192 /* floats per WG = 32 KB */, kPayload = 1024 /* floats per WG */;

__device__ __forceinline__ f32x4 load_sc1(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void store_sc1(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256) void k_phase(const float* weights, const float* in, float* out, unsigned* arrive,
                                               unsigned expect, unsigned* my_arrive, int work, unsigned* err) {
  __shared__ f32x4 wsm[kWeights / 4];
  const int tid = threadIdx.x;
  // prologue: weights -> LDS
  const f32x4* w4 = reinterpret_cast<const f32x4*>(weights + (size_t)blockIdx.x * kWeights);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < kWeights / 4 / kThr; ++u) wsm[tid + u * kThr] = w4[tid + u * kThr];
  __syncthreads();
  if (MODE == 1 && arrive) {
    if (tid < 8) {  // 8 per-XCD arrival words, 32 producers each
      const unsigned long long t0 = __builtin_readcyclecounter();
      while (__hip_atomic_load(&arrive[tid * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect) {
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_readcyclecounter() - t0 > 50000000ull) { atomicAdd(err, 1u); break; }  // ~20 ms: give up
      }
    }
    __syncthreads();
  }
  // payload of the previous phase: a DIFFERENT workgroup's output (cross-CU, cross-XCD)
  const int src = (blockIdx.x * 37 + 11) % kWG;
  const float* ip = in + (size_t)src * kPayload + tid * 4;
  f32x4 v = MODE == 1 ? load_sc1(ip) : *reinterpret_cast<const f32x4*>(ip);
  for (int i = 0; i < work; ++i) {  // dependent chain
    const f32x4 w = wsm[(tid + i) & (kWeights / 4 - 1)];
    acc = acc * 0.999f + v * w;
    v = v + acc * 1e-3f;
  }
  float* op = out + (size_t)blockIdx.x * kPayload + tid * 4;
  if (MODE == 1) {
    store_sc1(op, v);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(&my_arrive[(blockIdx.x & 7) * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    *reinterpret_cast<f32x4*>(op) = v;
  }
}

int main(int argc, char** argv) {
  const int N = 100, work = argc > 1 ? atoi(argv[1]) : 300;
  float *weights, *buf[2];
  unsigned *arrive, *err;
  CK(hipMalloc(&weights, sizeof(float) * kWG * kWeights));
  CK(hipMemset(weights, 0, sizeof(float) * kWG * kWeights));
  for (auto& b : buf) { CK(hipMalloc(&b, sizeof(float) * kWG * kPayload)); CK(hipMemset(b, 0, sizeof(float) * kWG * kPayload)); }
  // one set of 8 arrival words (64-byte apart) per phase parity; monotonic counts (expect = 32 * epoch)
  CK(hipMalloc(&arrive, sizeof(unsigned) * 2 * 128));
  CK(hipMalloc(&err, sizeof(unsigned)));
  hipStream_t s[2];
  CK(hipStreamCreate(&s[0])); CK(hipStreamCreate(&s[1]));
  hipEvent_t e0, e1, fork, join;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&fork)); CK(hipEventCreate(&join));

  auto issue = [&](int mode, unsigned epoch0) {
    for (int k = 0; k < N; ++k) {
      const float* in = buf[k & 1];
      float* out = buf[(k + 1) & 1];
      if (mode == 0) {
        hipLaunchKernelGGL(k_phase<0>, dim3(kWG), dim3(kThr), 0, s[0], weights, in, out, (unsigned*)nullptr, 0u,
                           (unsigned*)nullptr, work, err);
      } else {
        // phase k waits for phase k-1's arrivals (set (k-1)&1), announces on set k&1; per set one more epoch every 2 phases
        unsigned* wait_set = arrive + ((k + 1) & 1) * 128;
        unsigned* my_set = arrive + (k & 1) * 128;
        const unsigned expect = 32u * (epoch0 + (unsigned)((k - 1) / 2 + 1));
        hipLaunchKernelGGL(k_phase<1>, dim3(kWG), dim3(kThr), 0, s[k & 1], weights, in, out,
                           k == 0 ? (unsigned*)nullptr : wait_set, expect, my_set, work, err);
      }
    }
  };
  // note: the out buffer of phase k is the in buffer of phase k+1 and is overwritten by phase k+2, which is stream-ordered
  // after phase k ... but NOT after phase k+1 (its reader).  Phase k+2 waits for ALL of k+1's arrivals before it writes,
  // and a workgroup of k+1 arrives only after it has read: safe.
  for (int mode = 0; mode < 2; ++mode) {
    CK(hipMemset(arrive, 0, sizeof(unsigned) * 2 * 128));
    CK(hipMemset(err, 0, sizeof(unsigned)));
    unsigned epoch = 0;
    // eager
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s[0]));
      if (mode == 1) { CK(hipEventRecord(fork, s[0])); CK(hipStreamWaitEvent(s[1], fork, 0)); }
      issue(mode, epoch);
      epoch += N / 2;
      if (mode == 1) { CK(hipEventRecord(join, s[1])); CK(hipStreamWaitEvent(s[0], join, 0)); }
      CK(hipEventRecord(e1, s[0]));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) printf("mode %d eager: %.2f us per phase\n", mode, ms * 1e3 / N);
    }
    // graph: the arrival epochs are baked into the kernel arguments, so every replay is preceded by a reset of the words
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeGlobal));
    CK(hipMemsetAsync(arrive, 0, sizeof(unsigned) * 2 * 128, s[0]));
    if (mode == 1) { CK(hipEventRecord(fork, s[0])); CK(hipStreamWaitEvent(s[1], fork, 0)); }
    issue(mode, 0);
    if (mode == 1) { CK(hipEventRecord(join, s[1])); CK(hipStreamWaitEvent(s[0], join, 0)); }
    CK(hipStreamEndCapture(s[0], &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s[0]));
      CK(hipGraphLaunch(ge, s[0]));
      CK(hipEventRecord(e1, s[0]));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 3) printf("mode %d graph: %.2f us per phase\n", mode, ms * 1e3 / N);
    }
    unsigned herr; CK(hipMemcpy(&herr, err, sizeof(unsigned), hipMemcpyDeviceToHost));
    printf("mode %d: %u spin time-outs\n", mode, herr);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
