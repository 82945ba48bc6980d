// Damped dense Cholesky solve of the reduced camera system, float64, one workgroup.
//
// Replaces SparseBlock::solve / solveDenseD of the reference
// (/root/reference/src/droid_kernels.cu:200-218, :1248-1269: Eigen SimplicialLLT / LLT on the host,
// preceded by D2H copies of Hs, vs, S, v and followed by an H2D copy of dx): here the system never
// leaves the device.  n = 6P is 144 for a 25-keyframe window; the packed lower triangle of the system
// AUGMENTED with the right-hand side as an extra row ((n+1)(n+2)/2 doubles = 84.7 KB at n = 144) lives in
// LDS for n <= 199 and in an L2-resident global scratch otherwise.
//
// Right-looking blocked LL^T, block size NB = 12 (two pose blocks), three barriers per block step:
//   D  the 12x12 diagonal block is factored by ONE WAVE with its rows spread over lanes: the pivot and the
//      column entries travel by v_readlane, the square root is v_rsq_f64 + Newton steps (no f64 divide /
//      sqrt sequences on the critical path);
//   P  panel: one row per lane solves X L11^T = A21; the augmented row (the rhs) rides along, which is the
//      forward substitution L y = b for free;
//   U  trailing update A22 -= X X^T on the f64 matrix cores (v_mfma_f64_16x16x4_f64, one 16x16 tile per wave
//      at a time), skipping tiles whose panel rows are exactly zero (the reduced camera matrix of a sliding
//      window is block-banded, so most tiles are skipped).
// The backward substitution L^T x = y runs block-wise with the 12x12 triangle solved across lanes.
#include "ba_kernels.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace dba {

#ifdef PROFILE_SOLVE
#define PROF_DECL long long t_prev_ = wall_clock64()
#define PROF(slot)                                       \
  do {                                                   \
    if (threadIdx.x == 0 && prof) {                      \
      long long t_ = wall_clock64();                     \
      prof[slot] += t_ - t_prev_;                        \
      t_prev_ = t_;                                      \
    }                                                    \
  } while (0)
#else
#define PROF_DECL
#define PROF(slot)
#endif

constexpr int SOLVE_THREADS = 512;  // 2 waves/SIMD: 256 VGPRs, enough to hoist a whole 4x4 tile's operands

}  // namespace dba

#include "ba_solve_general.inc"

namespace dba {

template <bool USE_LDS>
__global__ __launch_bounds__(SOLVE_THREADS) void ba_solve_kernel(const double *__restrict__ H,
                                                                 const double *__restrict__ bvec, int n,
                                                                 double lm, double ep,
                                                                 float *__restrict__ dx,
                                                                 int *__restrict__ meta,
                                                                 double *__restrict__ Lglobal,
                                                                 long long *__restrict__ prof, int skip_if_solved) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // queued behind the skyline kernel (ba_solve_band.hip), which leaves meta[3] = 1 when it handled the system
  if (skip_if_solved && meta[3] != 0) return;
  ba_solve_general_body<USE_LDS>(H, bvec, n, lm, ep, dx, meta, Lglobal, prof, smem);
}


bool ba_solve_fits_lds(int n) {
  return solve_packed_bytes(n) + solve_small_bytes(n) <= (size_t)SOLVE_MAX_LDS_BYTES;
}

size_t ba_solve_scratch_doubles(int n) { return (size_t)(n + 1) * (n + 2) / 2; }

// hint 1: the one-tile skyline variant is known to take this graph's structure (it solved it in an earlier call on the
// same edge list, and whether it fits depends on the skyline alone): neither the several-tiles-per-thread variant, which only
// exists for skylines the first one cannot hold, nor the general kernel is queued behind it (4.6-4.8 us each per solve even
// when they return at once).  The one thing they were still a net for -- a partner workgroup that does not show up within
// a second -- then fails the solve (zero update, meta[1] = 1) instead of leaving it to the queue
// (Round 5, later: stage 0 of dba_ba runs the window kernel's admission test itself whenever it rebuilds a graph's tables and
// writes the same word -- ba_kernels.hip::ba_prepare_kernel --, so a changed graph is judged before its first solve is over;
// the window kernel's own report and the rare probes below remain for callers that solve without stage 0.)
// Which kernel took a workspace's last system: the window kernel reports "banded, taken" (1) or "not banded, solved by the
// general code in my launch" (2) through a word of pinned host memory per workspace (keyed by its meta pointer), read here
// without any synchronisation.  It steers the NEXT solve of that workspace -- 2: straight to the register-tile / skyline
// kernels, which are faster than the in-launch fall-back on such systems; anything else: the window kernel -- and is only
// ever a performance hint: whatever is launched solves whatever it is given.
// The pinned words of a workspace (16 ints, keyed by its meta pointer): [0] solver verdict, [1] solves since the verdict said "not
// banded", [4..6] stage 0's eta report (set, the call's eta rows, |kx|).  1024 slots; when they are all taken the OLDEST entry
// gives its slot up (one entry, not the table: kernels in flight for the other workspaces keep valid addresses, and the
// evicted workspace merely starts over with "nothing known" if it is ever used again).
namespace {
constexpr int WSW_SLOTS = 1024, WSW_STRIDE = 16;
struct WsWords {
  std::mutex mu;
  std::unordered_map<const void *, int> index;
  const void *owner[WSW_SLOTS] = {};
  int *pool = nullptr;
  int next = 0;
};
WsWords &ws_words() {
  static WsWords *w = new WsWords;   // (never destroyed: kernels may still write into the pool at exit)
  return *w;
}
}  // namespace

static int *ws_words_slot(const int *meta, bool reset) {
  WsWords &w = ws_words();
  std::lock_guard<std::mutex> lock(w.mu);
  if (!w.pool) {
    if (hipHostMalloc(reinterpret_cast<void **>(&w.pool), sizeof(int) * WSW_SLOTS * WSW_STRIDE,
                      hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
      w.pool = nullptr;
      (void)hipGetLastError();
      return nullptr;
    }
    memset(w.pool, 0, sizeof(int) * WSW_SLOTS * WSW_STRIDE);
  }
  auto it = w.index.find(meta);
  bool fresh = false;
  if (it == w.index.end()) {
    const int slot = w.next;
    w.next = (w.next + 1) % WSW_SLOTS;
    if (w.owner[slot]) w.index.erase(w.owner[slot]);
    w.owner[slot] = meta;
    it = w.index.emplace(meta, slot).first;
    fresh = true;
  }
  int *p = w.pool + it->second * WSW_STRIDE;
  if (fresh || reset)
    for (int i = 0; i < WSW_STRIDE; i++) __atomic_store_n(p + i, 0, __ATOMIC_RELAXED);
  return p;
}

int *solver_verdict_slot(const int *meta) { return ws_words_slot(meta, false); }
// a workspace that is (re)initialised forgets what an earlier workspace at the same address was told
void ws_words_reset(const int *meta) { (void)ws_words_slot(meta, true); }
int *ws_eta_status(const int *meta) {
  int *p = ws_words_slot(meta, false);
  return p ? p + 4 : nullptr;
}
// the first pending eta report of ANY workspace (meta == nullptr) or of this one; clears it
int ws_poll_eta(const int *meta, int *eta_rows, int *num_kx) {
  WsWords &w = ws_words();
  std::lock_guard<std::mutex> lock(w.mu);
  if (!w.pool) return 0;
  for (auto &kv : w.index) {
    if (meta && kv.first != meta) continue;
    int *st = w.pool + kv.second * WSW_STRIDE + 4;
    if (__atomic_load_n(st, __ATOMIC_ACQUIRE) == 0) continue;
    if (eta_rows) *eta_rows = st[1];
    if (num_kx) *num_kx = st[2];
    __atomic_store_n(st, 0, __ATOMIC_RELEASE);
    return 1;
  }
  return 0;
}

int launch_ba_solve(const double *H, const double *b, const int *fpose, int n, double lm, double ep, float *dx, int *meta,
                    double *Lscratch, hipStream_t stream, long long *prof, int hint, int *splan) {
  if (n <= 0) return DBA_OK;
  // n <= 174 (29 poses): the register-tile kernel (ba_solve_tile.hip).  Above that, up to n = 384, the skyline kernel
  // (ba_solve_band.hip) tries first; a system whose skyline does not fit one workgroup is left to this file's
  // kernel, which is queued behind it and returns at once when meta[3] says the system is solved.
  // DBA_SOLVE_KERNEL = tile | general | band forces a path (tests, comparisons).
  static const int forced = [] {
    const char *e = getenv("DBA_SOLVE_KERNEL");
    if (e && e[0] == 't') return 1;
    if (e && e[0] == 'g') return 2;
    if (e && e[0] == 'b') return 3;
    if (e && e[0] == 'w') return 4;
    const char *g = getenv("DBA_SOLVE_GENERAL");
    return (g && g[0] == '1') ? 2 : 0;
  }();
  // banded systems with a skyline table (every call of dba_ba): the five-wave window kernel, unless this workspace's last system
  // was not banded (then every 1024th solve still goes there, in case the graph has changed back).  DBA_SOLVE_KERNEL=wave forces it.
  if (!prof && fpose && (forced == 0 || forced == 4) && ba_solve_wave_supported(n)) {
    int *slot = solver_verdict_slot(meta);
    // (a workspace nobody has a verdict for yet -- the window shape changes with every keyframe -- starts from the verdict of
    // the last workspace that had one: a tracker whose graphs are not banded does not pay the in-launch fall-back per keyframe)
    static std::atomic<int> last_known{0};
    int verdict = slot ? __atomic_load_n(slot, __ATOMIC_RELAXED) : 0;
    if (verdict) last_known.store(verdict, std::memory_order_relaxed);
    else verdict = last_known.load(std::memory_order_relaxed);
    const bool probe = slot && verdict == 2 && ((__atomic_add_fetch(slot + 1, 1, __ATOMIC_RELAXED)) & 1023) == 0;   // has the graph become banded again?
    if (forced == 4 || verdict != 2 || probe) return launch_ba_solve_wave(H, b, fpose, n, lm, ep, dx, meta, Lscratch, slot, stream, nullptr, splan);
  }
  int chained = 0;
  if (!prof && forced <= 1 && ba_solve_tile_supported(n)) return launch_ba_solve_tile(H, b, n, lm, ep, dx, meta, stream);
  if (!prof && (forced == 0 || forced == 3) && ba_solve_band_supported(n)) {
    // n <= 174 (only reached with DBA_SOLVE_KERNEL=band): one workgroup, which cannot bail out -- with the scratch the
    // kernel would run on two workgroups, whose hand-shake can give the system up, and nothing is queued behind it here
    const bool single = ba_solve_tile_supported(n);
    const bool last = (hint == 1) && fpose != nullptr && !single;
    int rc = launch_ba_solve_band(H, b, fpose, n, lm, ep, dx, meta, single ? nullptr : Lscratch,
                                  (Lscratch && !single) ? ba_solve_scratch_doubles(n) : 0, false, stream, last);
    if (rc != DBA_OK || single || last) return rc;
    if (Lscratch && !ba_solve_fits_lds(n) && hint != 1) {  // wider skylines: several tiles per thread, panels in the global scratch
      rc = launch_ba_solve_band(H, b, fpose, n, lm, ep, dx, meta, Lscratch, ba_solve_scratch_doubles(n), true, stream);
      if (rc != DBA_OK) return rc;
    }
    chained = 1;
  }
  const size_t small = solve_small_bytes(n), packed = solve_packed_bytes(n);
  if (packed + small <= (size_t)SOLVE_MAX_LDS_BYTES) {
    static DeviceOnce attr_once;
    if (attr_once.needed()) {
      DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_solve_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, SOLVE_MAX_LDS_BYTES));
      attr_once.done();
    }
    hipLaunchKernelGGL(ba_solve_kernel<true>, dim3(1), dim3(SOLVE_THREADS), packed + small, stream, H, b, n,
                       lm, ep, dx, meta, Lscratch, prof, chained);
  } else {
    if (!Lscratch) return DBA_ERR_WORKSPACE;
    hipLaunchKernelGGL(ba_solve_kernel<false>, dim3(1), dim3(SOLVE_THREADS), small, stream, H, b, n, lm, ep,
                       dx, meta, Lscratch, prof, chained);
  }
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

}  // namespace dba
