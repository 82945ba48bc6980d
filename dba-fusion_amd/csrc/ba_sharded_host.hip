// The edge-sharded BA of one rank as ONE enqueued sequence (C ABI of include/dba_hip.h): stage 0, then per Gauss-Newton
// iteration { front half on the rank's edges -> sum of the partial [H | b] over the ranks -> back half on the summed
// system }, then the all-gather of the depth maps each rank owns -- every launch and every collective on the caller's stream,
// no host code between linearisation and solve (north_star: "an RCCL all-reduce over xGMI of the per-pose Hessian blocks
// before the solve").  The exchange is either RCCL called from here (a communicator of this library's own, created from a
// unique id the caller distributes: dbaf_amd/sharded.py does it through torch.distributed once per process group) or the
// one-shot peer-read kernel of peer_allreduce.hip.  Round 4 issued the collectives from Python between stage calls.
//
// librccl is loaded with dlopen at the first dba_comm_* call: the single-GPU product has no link-time dependency on it.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>

#include "ba_kernels.h"

namespace dba {

namespace {

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;      // (optional: dba_comm_info)
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  bool ok = false;
};

Rccl &rccl() {
  static Rccl R = [] {
    Rccl r;
    // the copy the process already has (PyTorch ships one), else the ROCm installation's
    const char *names[] = {"librccl.so", "librccl.so.1"};
    for (const char *nm : names)
      if (!r.lib) r.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
    for (const char *nm : names)
      if (!r.lib) r.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) r.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) return r;
#define RCCL_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name))
    RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    RCCL_SYM(CommInitRank, "ncclCommInitRank");
    RCCL_SYM(CommDestroy, "ncclCommDestroy");
    RCCL_SYM(AllReduce, "ncclAllReduce");
    RCCL_SYM(AllGather, "ncclAllGather");
    RCCL_SYM(GetErrorString, "ncclGetErrorString");
    RCCL_SYM(CommCount, "ncclCommCount");
    RCCL_SYM(CommUserRank, "ncclCommUserRank");
#undef RCCL_SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.AllGather;
    return r;
  }();
  return R;
}

int rccl_fail(const char *what, ncclResult_t e) {
  char msg[256];
  snprintf(msg, sizeof(msg), "%s -> rccl: %s", what, rccl().GetErrorString ? rccl().GetErrorString(e) : "error");
  set_last_error(msg, hipErrorUnknown);
  return DBA_ERR_HIP;
}

// rows of a [*, HW] float map by index list: dst[k] = src[rows[k]] (pack) / dst[rows[k]] = src[slots[k]] (unpack)
__global__ __launch_bounds__(256) void rows_pack_kernel(const float *__restrict__ src, const int64_t *__restrict__ rows, int HW,
                                                        float *__restrict__ dst) {
  const int k = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < HW) dst[(size_t)k * HW + i] = src[(size_t)rows[k] * HW + i];
}
__global__ __launch_bounds__(256) void rows_unpack_kernel(const float *__restrict__ src, const int64_t *__restrict__ slots,
                                                          const int64_t *__restrict__ rows, int HW, float *__restrict__ dst) {
  const int k = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < HW) dst[(size_t)rows[k] * HW + i] = src[(size_t)slots[k] * HW + i];
}
// the skyline band of [H | b]: packed[k] = hb[idx[k]] / hb[idx[k]] = packed[k]
__global__ __launch_bounds__(256) void band_take_kernel(const double *__restrict__ hb, const int64_t *__restrict__ idx, size_t n,
                                                        double *__restrict__ packed) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) packed[k] = hb[idx[k]];
}
__global__ __launch_bounds__(256) void band_put_kernel(double *__restrict__ hb, const int64_t *__restrict__ idx, size_t n,
                                                       const double *__restrict__ packed) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) hb[idx[k]] = packed[k];
}

}  // namespace

}  // namespace dba

using namespace dba;

struct dba_comm {
  ncclComm_t comm;
  int world, rank;
};

extern "C" {

int dba_comm_unique_id(void *id128) {
  if (!id128) return DBA_ERR_ARG;
  if (!rccl().ok) return DBA_ERR_UNSUPPORTED;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  const ncclResult_t e = rccl().GetUniqueId(&id);
  if (e != ncclSuccess) return rccl_fail("ncclGetUniqueId", e);
  memcpy(id128, &id, 128);
  return DBA_OK;
}

// what RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank): the number of ranks it spans and this
// process's rank in it -- bench.py puts the former on its line ("ranks_seen") so that RCCL's own view of N is on record
int dba_comm_info(dba_comm *c, int *world, int *rank) {
  if (!c) return DBA_ERR_ARG;
  int w = c->world, r = c->rank;
  if (rccl().CommCount) {
    const ncclResult_t e = rccl().CommCount(c->comm, &w);
    if (e != ncclSuccess) return rccl_fail("ncclCommCount", e);
  }
  if (rccl().CommUserRank) {
    const ncclResult_t e = rccl().CommUserRank(c->comm, &r);
    if (e != ncclSuccess) return rccl_fail("ncclCommUserRank", e);
  }
  if (world) *world = w;
  if (rank) *rank = r;
  return DBA_OK;
}

int dba_comm_create(const void *id128, int world, int rank, dba_comm **out) {
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return DBA_ERR_ARG;
  if (!rccl().ok) return DBA_ERR_UNSUPPORTED;
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  ncclComm_t c;
  const ncclResult_t e = rccl().CommInitRank(&c, world, id, rank);   // (binds the calling thread's current device)
  if (e != ncclSuccess) return rccl_fail("ncclCommInitRank", e);
  *out = new dba_comm{c, world, rank};
  return DBA_OK;
}

int dba_comm_destroy(dba_comm *c) {
  if (!c) return DBA_OK;
  const ncclResult_t e = rccl().CommDestroy(c->comm);
  delete c;
  return e == ncclSuccess ? DBA_OK : rccl_fail("ncclCommDestroy", e);
}

int dba_comm_allreduce_f64(dba_comm *c, double *buf, size_t count, dba_stream_t stream) {
  if (!c || (!buf && count)) return DBA_ERR_ARG;
  if (count == 0) return DBA_OK;
  const ncclResult_t e = rccl().AllReduce(buf, buf, count, ncclFloat64, ncclSum, c->comm, (hipStream_t)stream);
  return e == ncclSuccess ? DBA_OK : rccl_fail("ncclAllReduce", e);
}

int dba_ba_sharded_run(float *poses, float *disps, const float *intrinsics, const float *disps_sens, const float *targets,
                       const float *weights, const float *eta, int eta_rows, const int64_t *ii, const int64_t *jj,
                       const uint8_t *frame_owned, int N, int B, int ht, int wd, int t0, int t1, int iterations, float lm,
                       float ep, float alpha, int motion_only, const int32_t *window_fpose, int solver_hint, int prepared,
                       const dba_shard_exchange *x, void *ws, size_t ws_bytes, dba_stream_t stream) {
  if (!x || x->world < 1 || x->rank < 0 || x->rank >= x->world) return DBA_ERR_ARG;
  if (x->world > 1 && !x->comm && !x->peer_regions) return DBA_ERR_ARG;   // somebody has to carry the sums
  if (x->peer_regions && (!x->peer_status || !x->peer_epoch)) return DBA_ERR_ARG;
  if (x->band_len && (!x->band_idx || !x->band_buf)) return DBA_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  // The Gauss-Newton loop is dba_ba's own (ba_run_loop: the back-substitution + retraction of iteration k ride in the
  // linearisation of iteration k + 1 here too -- a rank only moves the depths of the frames it owns, the update of the poses
  // is redundant on every rank), with the sum over the ranks between the reduction and the solve.  One rank sums nothing.
  struct Ctx {
    const dba_shard_exchange *x;
  } ctx{x};
  BaExchange ex;
  ex.ctx = &ctx;
  ex.fn = [](void *c, double *hb, size_t hb_len, hipStream_t st) -> int {
    const dba_shard_exchange *x = static_cast<Ctx *>(c)->x;
    double *buf = hb;
    size_t cnt = hb_len;
    if (x->band_len) {   // large windows: only the skyline band travels (gather -> sum -> scatter)
      cnt = x->band_len, buf = x->band_buf;
      hipLaunchKernelGGL(band_take_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, hb, x->band_idx, cnt, buf);
      DBA_LAUNCH_CHECK();
    }
    int rc;
    if (x->peer_regions) {
      *x->peer_epoch += 1;
      rc = dba_peer_allreduce_f64(buf, cnt, x->peer_regions, x->rank, x->world, *x->peer_epoch, x->peer_max_doubles,
                                  x->peer_status, (dba_stream_t)st);
    } else {
      rc = dba_comm_allreduce_f64(x->comm, buf, cnt, (dba_stream_t)st);
    }
    if (rc != DBA_OK) return rc;
    if (x->band_len) {
      hipLaunchKernelGGL(band_put_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, hb, x->band_idx, cnt, buf);
      DBA_LAUNCH_CHECK();
    }
    return DBA_OK;
  };
  const bool sums = x->world > 1 && (x->comm || x->peer_regions);
  int rc = ba_run_loop(poses, disps, intrinsics, disps_sens, targets, weights, eta, eta_rows, ii, jj, frame_owned, N, B, ht, wd,
                       t0, t1, iterations, lm, ep, alpha, motion_only, nullptr, nullptr, ws, ws_bytes, stream,
                       (prepared == 1 || prepared == 2) ? prepared : 0, solver_hint, 0.f, window_fpose, sums ? &ex : nullptr);
  if (rc != DBA_OK) return rc;
  // the replicas of the depth maps are made coherent ONCE per call: every rank sends the rows it owns
  // (only with a communicator: the peer-read exchange carries the reduced system alone, its caller gathers the depths)
  if (!motion_only && iterations > 0 && x->kmax > 0 && x->comm) {
    if (!x->send || !x->recv || (x->n_mine && !x->my_rows) || (x->n_all && (!x->all_rows || !x->all_slots))) return DBA_ERR_ARG;
    const int HW = ht * wd;
    if (x->n_mine) {
      hipLaunchKernelGGL(rows_pack_kernel, dim3((HW + 255) / 256, x->n_mine), dim3(256), 0, s, disps, x->my_rows, HW, x->send);
      DBA_LAUNCH_CHECK();
    }
    const ncclResult_t e = rccl().AllGather(x->send, x->recv, (size_t)x->kmax * HW, ncclFloat32, x->comm->comm, s);
    if (e != ncclSuccess) return rccl_fail("ncclAllGather", e);
    if (x->n_all) {
      hipLaunchKernelGGL(rows_unpack_kernel, dim3((HW + 255) / 256, x->n_all), dim3(256), 0, s, x->recv, x->all_slots,
                         x->all_rows, HW, disps);
      DBA_LAUNCH_CHECK();
    }
  }
  return DBA_OK;
}

}  // extern "C"
