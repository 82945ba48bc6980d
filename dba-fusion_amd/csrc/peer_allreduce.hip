// One-shot all-reduce (sum, float64) of the reduced camera system between the GPUs of one node, by direct peer reads.
//
// the only exchange step of the sharded dense bundle adjustment is the sum of [H | b] over the ranks,
// 83 KB at 24 poses and 1.15 MB at 64 - far below the sizes at which a ring over xGMI is bandwidth-bound, so what a
// step pays is RCCL's launch + ring latency (7 hops at 8 ranks).  xGMI is point to point: every GPU can read every
// peer directly.  So each rank publishes its partial system in a buffer the peers have mapped (hipIpc), raises an epoch
// flag, waits for the peers' flags and then sums the world's buffers itself, in rank order - one kernel, no ring, and
// every rank adds the same numbers in the same order, so the replicas of the summed system are bit-identical (the
// sharded driver's redundant solves rely on that, as they do with RCCL).
//
// Region of a rank (dba_peer_exchange_bytes):  [0] epoch flag  [64] arrival counter of the local workgroups
//                                              [256 ...] two slots of max_doubles float64, used alternately by epoch.
// Epoch protocol: a rank writes slot e & 1 and then raises its flag to e; a peer that has raised e has finished reading
// every slot of epoch e - 1 (program order on its stream), so slot (e + 1) & 1 is free to be rewritten once all flags
// show e.  Flags and peer data are read with system-scope atomics (no stale lines out of the reader's L2); a workgroup
// that waits longer than the time-out for a peer (20 s unless DBA_PEER_TIMEOUT_MS says otherwise: a peer may be busy on
// the host between two calls) gives up and reports DBA_PEER_TIMEOUT through the sticky `status` word.  `buf` is then
// UNDEFINED (other workgroups may already have summed their part): the caller must not use it -- PeerDist.check(),
// which the sharded driver calls once per ba(), raises.
//
// Opt-in (dbaf_amd/peer.py, DBA_PEER_ALLREDUCE=1): this box has one GPU, so the path is tested with two PROCESSES that
// map each other's regions on the same device (tests/test_gpu_peer.py) - handles, epochs, ordering, determinism - but
// it has never run across xGMI; the default exchange stays RCCL.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <cstring>

#include "common.h"

namespace dba {

constexpr int PEER_MAX_WORLD = 16;
constexpr size_t PEER_HEADER = 256;
constexpr int PEER_MIN_BLOCKS = 16, PEER_MAX_BLOCKS = 128, PEER_THREADS = 256;

struct PeerRegions {
  unsigned char *r[PEER_MAX_WORLD];
};

__device__ __forceinline__ unsigned sys_load_u32(const unsigned *p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double sys_load_f64(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(PEER_THREADS) void peer_allreduce_kernel(double *__restrict__ buf, size_t n, PeerRegions R,
                                                                      int rank, int world, unsigned epoch,
                                                                      size_t max_doubles, long long timeout_ticks,
                                                                      int *__restrict__ status) {
  unsigned char *mine = R.r[rank];
  unsigned *flag = reinterpret_cast<unsigned *>(mine);
  unsigned *arrived = reinterpret_cast<unsigned *>(mine + 64);
  const size_t slot_off = PEER_HEADER + (size_t)(epoch & 1u) * max_doubles * sizeof(double);
  double *my_slot = reinterpret_cast<double *>(mine + slot_off);
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (size_t)gridDim.x * blockDim.x;

  // ---- publish: this rank's contribution -> its slot; the last workgroup to finish raises the flag
  for (size_t i = tid; i < n; i += nthreads) my_slot[i] = buf[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned before = atomicAdd(arrived, 1u);
    if (before == gridDim.x - 1) {
      *arrived = 0;
      __threadfence_system();
      __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }

  // ---- wait for every rank's flag (own included: all local workgroups have published)
  __shared__ int s_fail;
  if (threadIdx.x == 0) s_fail = 0;
  __syncthreads();
  if (threadIdx.x < (unsigned)world) {
    const unsigned *pf = reinterpret_cast<const unsigned *>(R.r[threadIdx.x]);
    const long long t0 = wall_clock64();
    // (epochs only grow; the signed difference keeps the comparison right across a wrap of the counter)
    while ((int)(sys_load_u32(pf) - epoch) < 0) {
      if (wall_clock64() - t0 > timeout_ticks) {  // (100 MHz clock)
        s_fail = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  if (s_fail) {
    if (threadIdx.x == 0) atomicExch(status, DBA_PEER_TIMEOUT);
    return;
  }

  // ---- sum in rank order (every rank: the same additions in the same order)
  for (size_t i = tid; i < n; i += nthreads) {
    double acc = 0.0;
    for (int r = 0; r < world; r++) {
      const double *slot = reinterpret_cast<const double *>(R.r[r] + slot_off);
      acc += (r == rank) ? my_slot[i] : sys_load_f64(slot + i);
    }
    buf[i] = acc;
  }
}

}  // namespace dba

using namespace dba;

extern "C" {

size_t dba_peer_exchange_bytes(size_t max_doubles) { return PEER_HEADER + 2 * max_doubles * sizeof(double); }

int dba_peer_exchange_create(size_t bytes, void **region, unsigned char *handle64) {
  if (!region || !handle64 || bytes < PEER_HEADER) return DBA_ERR_ARG;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C ABI passes IPC handles as 64 bytes");
  // The region is polled and read by OTHER devices while the producing kernel is still running.  HIP only promises
  // cross-device visibility of ordinary (coarse-grained) device memory at kernel boundaries, so the region is asked for
  // uncached first (DBA_PEER_COARSE=1 skips that); if the runtime cannot export such an allocation through hipIpc the
  // plain allocation is the fall-back, which relies on __threadfence_system writing the L2 back on gfx950.
  void *p = nullptr;
  hipIpcMemHandle_t h;
  hipError_t e = hipErrorUnknown;
  static const bool coarse_only = [] { const char *v = getenv("DBA_PEER_COARSE"); return v && v[0] == '1'; }();
  if (!coarse_only && hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) == hipSuccess) {
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
      (void)hipFree(p);
      p = nullptr;
      (void)hipGetLastError();
    }
  }
  if (!p) {
    DBA_HIP_CHECK(hipMalloc(&p, bytes));
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
  }
  if (e != hipSuccess) {
    (void)hipFree(p);
    set_last_error("dba_peer_exchange_create", e);
    return DBA_ERR_HIP;
  }
  memcpy(handle64, &h, 64);
  *region = p;
  return DBA_OK;
}

int dba_peer_exchange_open(const unsigned char *handle64, void **region) {
  if (!handle64 || !region) return DBA_ERR_ARG;
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  DBA_HIP_CHECK(hipIpcOpenMemHandle(region, h, hipIpcMemLazyEnablePeerAccess));
  return DBA_OK;
}

int dba_peer_exchange_close(void *region, int opened) {
  if (!region) return DBA_OK;
  if (opened) DBA_HIP_CHECK(hipIpcCloseMemHandle(region));
  else DBA_HIP_CHECK(hipFree(region));
  return DBA_OK;
}

int dba_peer_allreduce_f64(double *buf, size_t n, void *const *regions, int rank, int world, unsigned epoch,
                           size_t max_doubles, int *status, dba_stream_t stream) {
  if (world < 1 || world > PEER_MAX_WORLD || rank < 0 || rank >= world || !regions || !status || epoch == 0)
    return DBA_ERR_ARG;
  if (n == 0) return DBA_OK;
  if (!buf || n > max_doubles) return DBA_ERR_ARG;
  PeerRegions R;
  for (int r = 0; r < PEER_MAX_WORLD; r++) R.r[r] = (r < world) ? static_cast<unsigned char *>(regions[r]) : nullptr;
  for (int r = 0; r < world; r++)
    if (!R.r[r]) return DBA_ERR_ARG;
  // enough workgroups to keep ~8 peer reads per thread in flight (all of them must be resident at once: they meet at
  // the flags; 128 single-wave-group workgroups are a fraction of the 256 CUs)
  size_t blocks = n / (8 * PEER_THREADS);
  blocks = blocks < PEER_MIN_BLOCKS ? PEER_MIN_BLOCKS : (blocks > PEER_MAX_BLOCKS ? PEER_MAX_BLOCKS : blocks);
  static const long long timeout_ticks = [] {
    const char *v = getenv("DBA_PEER_TIMEOUT_MS");
    const long long ms = v ? atoll(v) : 20000;
    return (ms > 0 ? ms : 20000) * 100000ll;  // 100 MHz wall clock
  }();
  hipLaunchKernelGGL(peer_allreduce_kernel, dim3((unsigned)blocks), dim3(PEER_THREADS), 0, (hipStream_t)stream, buf, n, R,
                     rank, world, epoch, max_doubles, timeout_ticks, status);
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

}  // extern "C"
