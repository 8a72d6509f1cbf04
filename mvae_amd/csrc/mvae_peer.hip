// mvae_peer.hip -- one-shot peer-read exchange of the flat gradient buffer between the ranks of ONE node
// (C ABI: include/mvae_hip.h, "Peer-read gradient exchange").  New functionality: the reference is single-device
// (SURVEY.md section 8e); this is the intra-node alternative to the RCCL all-reduce of mvae_amd/distributed.py.
//
// Every rank owns a pair of gradient slots [2][n] in its own HBM (ONE hipMalloc, exported with hipIpcGetMemHandle and
// mapped by every peer with hipIpcOpenMemHandle: on a node the mapping goes over xGMI) and all ranks share one host page
// of flags (POSIX shared memory registered with hipHostRegister: host memory is fine-grained coherent, so flag traffic
// needs no assumption about device caches).  Per step, on the step's stream, no host work (graph-capturable):
//
//   k_peer_copy     local gradients -> own slot[(seq + 1) & 1]
//   -- kernel boundary: the copy has left this device's caches --
//   k_peer_signal   seq += 1 ; flags[rank] = seq (system-scope release) ; wait until flags[r] >= seq for every r,
//                   bounded by a time-out that is COUNTED (flags[32 + rank]) instead of hanging the device
//   -- kernel boundary: the next launch starts with invalidated caches --
//   k_optim<PEER>   (mvae_step.hip) g = slot_0 + slot_1 + ... in RANK ORDER on every rank -> bit-identical sums ->
//                   Adam / SGD exactly as after an all-reduce
//
// Correctness rests on kernel boundaries (release at the end of the producer's copy, acquire at the start of the
// consumer's optimizer) and on the host-coherent flags only.  A slot is overwritten two publishes later; a rank cannot
// get there before every peer has finished READING it: publish s + 2 follows the rank's own wait for flags >= s + 1,
// and a peer raises its flag to s + 1 only after its optimizer launch of step s (stream order).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "mvae_common.hpp"

__global__ __launch_bounds__(256) void k_peer_copy(const float4* g, float4* slots, long long n4, const int* seq) {
  const int par = (seq[0] + 1) & 1;
  float4* dst = slots + (size_t)par * n4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) dst[i] = g[i];
}

// phase 0: after the copy (bumps the sequence number); phase 1 (two-shot only): after the rank's slice is reduced.
// Flags of phase ph live at flags[64 ph + r]; time-outs are counted at flags[32 + r].
__global__ __launch_bounds__(64) void k_peer_signal(int* seq, unsigned int* flags, int world, int rank,
                                                    unsigned long long timeout_ticks, int phase) {
  const int tid = threadIdx.x;
  const unsigned int s = (unsigned int)seq[0] + (phase == 0 ? 1u : 0u);
  unsigned int* fl = flags + 64 * phase;
  if (tid == 0) {
    if (phase == 0) seq[0] = (int)s;  // read by the launches that follow (parity of the slot to sum)
    __hip_atomic_store(&fl[rank], s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (tid < world && tid != rank) {
    const unsigned long long t0 = wall_clock64();
    // sequence numbers only grow; the signed difference survives the wrap of the 32-bit counter
    while ((int)(__hip_atomic_load(&fl[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - s) < 0) {
      if (wall_clock64() - t0 > timeout_ticks) {
        atomicAdd_system(&flags[32 + rank], 1u);  // reported by mvae_peer_timeouts; the step goes on with stale data
        break;
      }
      __builtin_amdgcn_s_sleep(32);
    }
  }
}

// Two-shot, first shot: rank r adds slice r of EVERY rank's slot (rank order) and leaves the sum in slice r of its own slot
// -- the only reader of that region in this phase is rank r itself; the optimizer launch of every rank then reads slice j
// from rank j.  Per link and step that is 2 n / W floats instead of the one-shot's n.
__global__ __launch_bounds__(256) void k_peer_reduce_slice(PeerSrc ps, float* own, int rank) {
  const size_t off4 = (size_t)(ps.seq[0] & 1) * (size_t)(ps.n / 4);
  const long long n4 = ps.n / 4;
  const long long lo = (long long)rank * ps.slice4, hi = lo + ps.slice4 < n4 ? lo + ps.slice4 : n4;
  for (long long i = lo + (long long)blockIdx.x * 256 + threadIdx.x; i < hi; i += (long long)gridDim.x * 256) {
    float4 a = reinterpret_cast<const float4*>(ps.slot[0])[off4 + i];
    for (int r = 1; r < ps.world; ++r) {
      const float4 o = reinterpret_cast<const float4*>(ps.slot[r])[off4 + i];
      a.x += o.x;
      a.y += o.y;
      a.z += o.z;
      a.w += o.w;
    }
    reinterpret_cast<float4*>(own)[off4 + i] = a;
  }
}

extern "C" int mvae_peer_create(int64_t n_floats, int world, int rank, const char* shm_name, double timeout_seconds,
                                mvae_peer** out) {
  if (!out || !shm_name || n_floats < 4 || (n_floats & 3)) return fail(MVAE_E_BADARG, "bad peer arguments%s", "");
  if (world < 1 || world > kPeerMaxWorld || rank < 0 || rank >= world)
    return fail(MVAE_E_BADARG, "world must be in [1, MVAE_PEER_MAX_WORLD], rank in [0, world)%s", "");
  if (strlen(shm_name) >= sizeof(mvae_peer::shm_name)) return fail(MVAE_E_BADARG, "shm name too long%s", "");
  mvae_peer* p = new mvae_peer();
  p->world = world;
  p->rank = rank;
  p->n = n_floats;
  strcpy(p->shm_name, shm_name);
  p->timeout_ticks = (unsigned long long)((timeout_seconds > 0 ? timeout_seconds : 2.0) * 1e8);
  auto bail = [&](int rc) {
    mvae_peer_destroy(p);
    return rc;
  };
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&p->slots), sizeof(float) * 2 * (size_t)n_floats);
  if (e != hipSuccess) return bail(hip_fail(e, "hipMalloc(peer slots)"));
  if ((e = hipMemset(p->slots, 0, sizeof(float) * 2 * (size_t)n_floats)) != hipSuccess)
    return bail(hip_fail(e, "hipMemset(peer slots)"));
  if ((e = hipMalloc(reinterpret_cast<void**>(&p->seq), 64)) != hipSuccess) return bail(hip_fail(e, "hipMalloc(seq)"));
  if ((e = hipMemset(p->seq, 0, 64)) != hipSuccess) return bail(hip_fail(e, "hipMemset(seq)"));
  p->peer_slots[rank] = p->slots;
  p->imported[rank] = true;
  // the flag page: every rank opens the same name; the file is created zero-filled by whoever comes first
  p->shm_fd = shm_open(shm_name, O_CREAT | O_RDWR, 0600);
  if (p->shm_fd < 0) return bail(fail(MVAE_E_SYSTEM, "shm_open failed for %s", shm_name));
  if (ftruncate(p->shm_fd, 4096) != 0) return bail(fail(MVAE_E_SYSTEM, "ftruncate failed for %s", shm_name));
  void* m = mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, p->shm_fd, 0);
  if (m == MAP_FAILED) return bail(fail(MVAE_E_SYSTEM, "mmap failed for %s", shm_name));
  p->flags_host = static_cast<unsigned int*>(m);
  if ((e = hipHostRegister(m, 4096, hipHostRegisterMapped)) != hipSuccess) {
    munmap(m, 4096);
    p->flags_host = nullptr;
    return bail(hip_fail(e, "hipHostRegister(flag page)"));
  }
  if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&p->flags_dev), m, 0)) != hipSuccess)
    return bail(hip_fail(e, "hipHostGetDevicePointer(flag page)"));
  if ((e = hipDeviceSynchronize()) != hipSuccess) return bail(hip_fail(e, "hipDeviceSynchronize"));
  *out = p;
  return 0;
}

extern "C" void mvae_peer_destroy(mvae_peer* p) {
  if (!p) return;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < p->world; ++r)
    if (r != p->rank && p->imported[r] && p->peer_slots[r]) (void)hipIpcCloseMemHandle(p->peer_slots[r]);
  if (p->flags_host) {
    (void)hipHostUnregister(p->flags_host);
    munmap(p->flags_host, 4096);
  }
  if (p->shm_fd >= 0) {
    close(p->shm_fd);
    if (p->rank == 0) shm_unlink(p->shm_name);  // the name disappears; mappings of the other ranks stay valid
  }
  if (p->slots) (void)hipFree(p->slots);
  if (p->seq) (void)hipFree(p->seq);
  delete p;
}

extern "C" int mvae_peer_export(mvae_peer* p, uint8_t handle[MVAE_IPC_HANDLE_BYTES]) {
  if (!p || !handle) return fail(MVAE_E_BADARG, "null pointer%s", "");
  static_assert(sizeof(hipIpcMemHandle_t) == MVAE_IPC_HANDLE_BYTES, "hipIpcMemHandle_t size");
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, p->slots);
  if (e != hipSuccess) return hip_fail(e, "hipIpcGetMemHandle");
  memcpy(handle, &h, sizeof(h));
  return 0;
}

extern "C" int mvae_peer_import(mvae_peer* p, int peer_rank, const uint8_t handle[MVAE_IPC_HANDLE_BYTES]) {
  if (!p || !handle) return fail(MVAE_E_BADARG, "null pointer%s", "");
  if (peer_rank < 0 || peer_rank >= p->world) return fail(MVAE_E_BADARG, "peer rank out of range%s", "");
  if (peer_rank == p->rank) return 0;
  if (p->imported[peer_rank]) return fail(MVAE_E_BADARG, "peer already imported%s", "");
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  void* ptr = nullptr;
  hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) return hip_fail(e, "hipIpcOpenMemHandle");
  p->peer_slots[peer_rank] = static_cast<float*>(ptr);
  p->imported[peer_rank] = true;
  return 0;
}

extern "C" int mvae_peer_publish(mvae_peer* p, const float* grads, void* stream) {
  if (!p || !grads) return fail(MVAE_E_BADARG, "null pointer%s", "");
  for (int r = 0; r < p->world; ++r)
    if (!p->imported[r]) return fail(MVAE_E_BADARG, "peer %s%lld has not been imported", "", r);
  if (!aligned16(grads)) return fail(MVAE_E_ALIGN, "gradient buffer must be 16-byte aligned%s", "");
  hipStream_t s = (hipStream_t)stream;
  const long long n4 = p->n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
  hipLaunchKernelGGL(k_peer_copy, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(grads),
                     reinterpret_cast<float4*>(p->slots), n4, p->seq);
  hipLaunchKernelGGL(k_peer_signal, dim3(1), dim3(64), 0, s, p->seq, p->flags_dev, p->world, p->rank,
                     p->timeout_ticks, 0);
  if (p->mode == 1) {
    PeerSrc ps{};
    for (int r = 0; r < p->world; ++r) ps.slot[r] = p->peer_slots[r];
    ps.seq = p->seq;
    ps.n = p->n;
    ps.world = p->world;
    ps.slice4 = peer_slice4(p);
    const int rb = (int)((ps.slice4 + 255) / 256 < 256 ? (ps.slice4 + 255) / 256 : 256);
    hipLaunchKernelGGL(k_peer_reduce_slice, dim3(rb), dim3(256), 0, s, ps, p->slots, p->rank);
    hipLaunchKernelGGL(k_peer_signal, dim3(1), dim3(64), 0, s, p->seq, p->flags_dev, p->world, p->rank,
                       p->timeout_ticks, 1);
  }
  LAUNCH_CHECK("peer publish");
  return 0;
}

// Sharded form, all-gather of PARAMETERS: slice j of rank j's slot holds the parameters rank j's optimizer launch has just
// written there (in place of the gradients it summed); every rank copies the slices it does not own into its own buffer.
__global__ __launch_bounds__(256) void k_peer_gather(PeerSrc ps, float* params, int rank) {
  const size_t off4 = (size_t)(ps.seq[0] & 1) * (size_t)(ps.n / 4);
  const long long n4 = ps.n / 4;
  const long long lo = (long long)rank * ps.slice4, hi = lo + ps.slice4;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    if (i >= lo && i < hi) continue;  // the owned slice: written by this rank's optimizer launch
    int owner = (int)(i / ps.slice4);
    owner = owner < ps.world ? owner : ps.world - 1;
    reinterpret_cast<float4*>(params)[i] = reinterpret_cast<const float4*>(ps.slot[owner])[off4 + i];
  }
}

int peer_gather_params(mvae_peer* p, float* params, hipStream_t s) {
  hipLaunchKernelGGL(k_peer_signal, dim3(1), dim3(64), 0, s, p->seq, p->flags_dev, p->world, p->rank, p->timeout_ticks, 1);
  if (p->world > 1) {
    PeerSrc ps{};
    for (int r = 0; r < p->world; ++r) ps.slot[r] = p->peer_slots[r];
    ps.seq = p->seq;
    ps.n = p->n;
    ps.world = p->world;
    ps.slice4 = peer_slice4(p);
    const long long n4 = p->n / 4;
    const int gb = (int)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_peer_gather, dim3(gb), dim3(256), 0, s, ps, params, p->rank);
  }
  LAUNCH_CHECK("peer parameter gather");
  return 0;
}

extern "C" int mvae_peer_set_two_shot(mvae_peer* p, int on) {
  if (!p) return fail(MVAE_E_BADARG, "null pointer%s", "");
  if (on < 0 || on > 2) return fail(MVAE_E_BADARG, "exchange form must be 0 (one-shot), 1 (two-shot) or 2 (sharded optimizer)%s", "");
  p->mode = on;
  return 0;
}

extern "C" int mvae_peer_timeouts(mvae_peer* p) {
  if (!p || !p->flags_host) return fail(MVAE_E_BADARG, "null pointer%s", "");
  return (int)__atomic_load_n(&p->flags_host[32 + p->rank], __ATOMIC_ACQUIRE);
}
