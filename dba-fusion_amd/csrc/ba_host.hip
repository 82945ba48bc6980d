// Host side of the BA stages: workspace planning and launches (C ABI of include/dba_hip.h).
// Nothing here synchronises the stream except the two BACore calls that must hand float64 host
// buffers to the caller (the GTSAM side of DBA-Fusion, dbaf/depth_video.py:524-558).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ba_kernels.h"

#include <algorithm>
#include <atomic>
#include <mutex>
#include <unordered_map>

#ifndef GRAM_F32_DEFAULT
#define GRAM_F32_DEFAULT true   // per-frame Schur products: 16-term float chains flushed into float64 (DBA_SCHUR_MFMA=f64:
                                // every product on the float64 matrix pipe, 48 instead of 40 us at 64 KF / 512 edges)
#endif

namespace dba {

static thread_local char g_last_error[512] = "";

void set_last_error(const char *what, hipError_t e) {
  snprintf(g_last_error, sizeof(g_last_error), "%s -> %s", what, hipGetErrorString(e));
}

// Schur kernel of a window: the per-source-frame form (float64 Gram tiles, every row of E read once) where frames couple
// many rows -- 64 KF / 512 edges: 47 us against 86 us for the (row, partner) grid --, the (row, partner) grid on sparse
// windows, whose pairs are few and whose kernel is a chain of latencies either way (25 KF / 96 edges: 12.9 against 16.6 us).
// DBA_SCHUR_KERNEL = rows | frame or dba_ba_schur_select() force one (the tests run both).
// deterministic (fixed-point) accumulation of H, b: dba_ba_set_deterministic / DBA_DETERMINISTIC=1 (ba_kernels.hip: acc_add)
static std::atomic<int> g_deterministic{[] { const char *e = getenv("DBA_DETERMINISTIC"); return (e && e[0] == '1') ? 1 : 0; }()};
// residual check behind every solve of the stage functions and of dba_ba (opt-in: one more launch of ~5 us per solve)
static std::atomic<int> g_solve_check{[] { const char *e = getenv("DBA_SOLVE_CHECK"); return (e && e[0] == '1') ? 1 : 0; }()};

// || (H + diag(ep + lm H_ii)) x - b ||_inf <= 1e-5 (||b||_inf + max_i sum_j |H_ij x_j|)  (x is the float solution: its rounding alone is
// ~6e-8 of the row sums) -- else dx := 0, meta[1] := 1.  H: lower triangle.  One workgroup, a row per thread at a time.
__global__ __launch_bounds__(256) void ba_solve_check_kernel(const double *__restrict__ H, const double *__restrict__ b, int n, double lm,
                                                             double ep, float *__restrict__ dx, int *__restrict__ meta) {
  __shared__ double s_res[256], s_mag[256];
  __shared__ int s_bad;
  if (meta[1] != 0) return;   // already a failed solve: dx is zero
  double res = 0.0, mag = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    double acc = -b[i], m = fabs(b[i]);
    for (int j = 0; j < n; j++) {
      double h = H[(size_t)max(i, j) * n + min(i, j)];
      if (i == j) h = fma(lm, h, h) + ep;
      const double t = h * (double)dx[j];
      acc += t, m += fabs(t);
    }
    res = fmax(res, fabs(acc)), mag = fmax(mag, m);
  }
  s_res[threadIdx.x] = res, s_mag[threadIdx.x] = mag;
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = 0.0, m = 0.0;
    for (int k = 0; k < (int)blockDim.x; k++) r = fmax(r, s_res[k]), m = fmax(m, s_mag[k]);
    s_bad = !(r <= 1e-5 * m) ? 1 : 0;   // (NaN counts as bad)
    if (s_bad) meta[1] = 1;
  }
  __syncthreads();
  if (s_bad)
    for (int i = threadIdx.x; i < n; i += blockDim.x) dx[i] = 0.f;
}
static std::atomic<int> g_schur_form{[] {
  const char *e = getenv("DBA_SCHUR_KERNEL");
  return !e ? 0 : (e[0] == 'r' ? 1 : (e[0] == 'f' || e[0] == 'g') ? 2 : 0);
}()};  // 0 = automatic, 1 = (row, partner) grid, 2 = per-source-frame form; dba_ba_schur_select() changes it

// rows a source frame couples, estimated from the GRAPH (N edges over the window's P optimised poses + the fixed frame in
// front of them), not from the size of the video buffer: min(B, P + N) grows with the buffer (the reference's DepthVideo
// holds 1024 frames) and made the per-frame form unreachable in a real integration
static bool schur_auto_frame_form(int N, int P) {
  const int frames = P + 1;
  const int rows_est = 1 + (N + frames - 1) / frames;
  return rows_est > 6;
}

static std::atomic<int> g_schur_generation{0};
static thread_local int t_schur_form = 0;  // per-thread pin (dba_ba_schur_select_thread): the sharded driver's ranks

bool ba_schur_frame_form(int N, int P) {
  const int forced = g_schur_form.load(std::memory_order_relaxed);
  if (N + 1 > GRAM_LIST_CAP) return false;
  if (forced) return forced == 2;
  if (t_schur_form) return t_schur_form == 2;
  return schur_auto_frame_form(N, P);
}

int ba_plan(int N, int B, int ht, int wd, int t0, int t1, void *ws, size_t ws_bytes, BaPlan *plan) {
  if (N < 0 || B <= 0 || ht <= 0 || wd <= 0 || t1 < t0 || t0 < 0 || t1 > B) return DBA_ERR_ARG;
  const int P = t1 - t0;
  const int HW = ht * wd;
  const int Mmax = (P + N < B) ? (P + N) : B;
  // pixels per lane of the linearisation: enough waves to cover the 1024 SIMDs about once; fewer, fatter
  // waves beyond that (each wave-level reduction of the J^T W J sums is amortised over PPL pixels)
  int ppl = 1;
  {
    // read once per process: every stage call re-plans, and linearise / reduce must agree on nparts
    static const int env_ppl = [] { const char *e = getenv("DBA_LINEARIZE_PPL"); return e ? atoi(e) : 0; }();
    const long waves1 = (long)Mmax * ((HW + 63) / 64);
    if (env_ppl == 1 || env_ppl == 2 || env_ppl == 4) ppl = env_ppl;
    else if (waves1 >= 16 * 1024) ppl = 4;
    else if (waves1 >= 8 * 1024) ppl = 2;  // measured: 25 KF/64x64 (1664 waves) and 64 KF (4096) best at 1 (matrix-core sums)
  }
  const int nchunks = (HW + 256 * ppl - 1) / (256 * ppl);
  const int nparts_max = ((HW + 255) / 256) * 4;  // workspace is sized for ppl = 1
  const int nparts = nchunks * 4;
  const int n6 = 6 * P;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  dba_ba_layout L;
  memset(&L, 0, sizeof(L));
  L.meta = take(sizeof(int) * 32);  // [8..15]: handshake flags of the solver's two workgroups (nothing else writes there);
                                    // [16..23]: the window solver's plan for this graph (launch_ba_solve's splan);
                                    // [24..27]: launch counters / generation numbers of its two-workgroup form, [28..31]: the skyline kernel's
  const size_t o_gkey = take(sizeof(int) * (8 + 2 * (size_t)N));   // (right behind meta: dba_ba_workspace_init clears both)
  L.kx = take(sizeof(int) * (size_t)(Mmax > 0 ? Mmax : 1));
  const size_t o_fslot = take(sizeof(int) * (size_t)B);
  const size_t o_eoff = take(sizeof(int) * (size_t)(Mmax + 1));
  const size_t o_elist = take(sizeof(int) * (size_t)(N > 0 ? N : 1));
  const size_t o_erank = take(sizeof(int) * (size_t)(N > 0 ? N : 1));
  const size_t o_fpose = take(sizeof(int) * (size_t)(P > 0 ? P : 1));
  const size_t o_einfo = take(sizeof(int) * 2 * (size_t)(N > 0 ? N : 1));
  const size_t o_rowinfo = take(sizeof(int) * 8 * (size_t)(P + N > 0 ? P + N : 1));
  const size_t o_fhead = take(sizeof(int) * 4 * (size_t)(Mmax > 0 ? Mmax : 1));
  const size_t o_frow = take(sizeof(int) * 2 * (size_t)(P + N > 0 ? P + N : 1));
  L.E = take(sizeof(float) * (size_t)(P + N) * 6 * HW);
  L.Q = take(sizeof(float) * (size_t)Mmax * HW);
  L.w = take(sizeof(float) * (size_t)Mmax * HW);
  const size_t o_hparte = take(sizeof(float) * (size_t)(N > 0 ? N : 1) * nparts_max * HPE_STRIDE);
  const size_t o_aedge = take(sizeof(double) * 36 * (size_t)(N > 0 ? N : 1));
  L.dx = take(sizeof(float) * (size_t)(n6 > 0 ? n6 : 1));
  const size_t o_ptmp = take(sizeof(float) * 2 * 7 * (size_t)B);
  L.H = take(sizeof(double) * (size_t)(n6 > 0 ? n6 * (size_t)n6 : 1));
  L.b = take(sizeof(double) * (size_t)(n6 > 0 ? n6 : 1));
  size_t o_lscratch = 0;
  const bool lds_fits = ba_solve_fits_lds(n6);
  // systems the skyline solver may split (16 - 64 poses) always get the scratch: its two workgroups exchange their
  // contributions to the separator block through it (ba_solve_band.hip)
  const bool want_scratch = !lds_fits || (ba_solve_band_supported(n6) && n6 >= 96);
  if (want_scratch) o_lscratch = take(sizeof(double) * ba_solve_scratch_doubles(n6));
  L.P = P;
  L.Mmax = Mmax;
  L.nchunks = nchunks;
  plan->layout = L;
  plan->bytes = off;
  plan->P = P;
  plan->N = N;
  plan->B = B;
  plan->HW = HW;
  plan->nchunks = nchunks;
  if (ws) {
    if (ws_bytes < off) return DBA_ERR_WORKSPACE;
    char *base = static_cast<char *>(ws);
    plan->T.meta = reinterpret_cast<int *>(base + L.meta);
    plan->T.kx = reinterpret_cast<int *>(base + L.kx);
    plan->T.frame_slot = reinterpret_cast<int *>(base + o_fslot);
    plan->T.eoff = reinterpret_cast<int *>(base + o_eoff);
    plan->T.elist = reinterpret_cast<int *>(base + o_elist);
    plan->T.elist_rank = reinterpret_cast<int *>(base + o_erank);
    plan->T.fpose = reinterpret_cast<int *>(base + o_fpose);
    plan->T.einfo = reinterpret_cast<int *>(base + o_einfo);
    plan->T.rowinfo = reinterpret_cast<int *>(base + o_rowinfo);
    plan->T.fhead = reinterpret_cast<int *>(base + o_fhead);
    plan->T.frow = reinterpret_cast<int *>(base + o_frow);
    plan->T.gkey = reinterpret_cast<int *>(base + o_gkey);
    plan->T.Mmax = Mmax;
    plan->T.B = B;
    plan->W.E = reinterpret_cast<float *>(base + L.E);
    plan->W.Q = reinterpret_cast<float *>(base + L.Q);
    plan->W.w = reinterpret_cast<float *>(base + L.w);
    plan->W.HpartE = reinterpret_cast<float *>(base + o_hparte);
    plan->W.Aedge = reinterpret_cast<double *>(base + o_aedge);
    plan->W.dx = reinterpret_cast<float *>(base + L.dx);
    plan->W.poses_tmp = reinterpret_cast<float *>(base + o_ptmp);
    plan->W.H = reinterpret_cast<double *>(base + L.H);
    plan->W.b = reinterpret_cast<double *>(base + L.b);
    plan->W.Lscratch = want_scratch ? reinterpret_cast<double *>(base + o_lscratch) : nullptr;
    plan->W.nparts = nparts;
    plan->W.ppl = ppl;
  }
  return DBA_OK;
}

}  // namespace dba

using namespace dba;

extern "C" {

const char *dba_version(void) { return "dba_hip 0.1 (gfx950)"; }
const char *dba_last_error(void) { return g_last_error; }

size_t dba_ba_workspace_bytes(int N, int B, int ht, int wd, int t0, int t1) {
  BaPlan plan;
  if (ba_plan(N, B, ht, wd, t0, t1, nullptr, 0, &plan) != DBA_OK) return 0;
  return plan.bytes;
}

int dba_ba_get_layout(int N, int B, int ht, int wd, int t0, int t1, dba_ba_layout *out) {
  BaPlan plan;
  const int rc = ba_plan(N, B, ht, wd, t0, t1, nullptr, 0, &plan);
  if (rc != DBA_OK) return rc;
  *out = plan.layout;
  return DBA_OK;
}

// The prepare kernel reports an eta / |kx| mismatch through pinned, host-coherent words of the WORKSPACE it ran on (ws_eta_status,
// ba_solve.hip: sticky until polled; [0] = 1 when set, [1] = the eta rows the call was given, [2] = |kx| of its graph) -- and the
// call itself has changed nothing (ba_prepare_kernel::check_eta).
int dba_ba_poll_eta_error(int *eta_rows, int *num_kx) { return ws_poll_eta(nullptr, eta_rows, num_kx); }

int dba_ba_poll_eta_error_ws(int N, int B, int ht, int wd, int t0, int t1, void *ws, size_t ws_bytes, int *eta_rows, int *num_kx) {
  BaPlan plan;
  const int rc = ba_plan(N, B, ht, wd, t0, t1, ws, ws_bytes, &plan);
  if (rc != DBA_OK) return rc;
  return ws_poll_eta(plan.T.meta, eta_rows, num_kx);
}

int dba_ba_gather_edges(const float *target_inac, const float *weight_inac, const int64_t *ii_inac, const int64_t *jj_inac,
                        int n_inac, const int64_t *sel, int n_sel, const float *target_act, const float *weight_act,
                        const int64_t *ii_act, const int64_t *jj_act, int n_act, int ht, int wd, float *targets_out,
                        float *weights_out, int64_t *ii_out, int64_t *jj_out, dba_stream_t stream) {
  if (n_sel < 0 || n_act < 0 || n_inac < 0 || ht <= 0 || wd <= 0) return DBA_ERR_ARG;
  if (!sel && n_sel > n_inac) return DBA_ERR_ARG;
  const int n = n_sel + n_act;
  if (n == 0) return DBA_OK;
  if (!targets_out || !weights_out || !ii_out || !jj_out) return DBA_ERR_ARG;
  if (n_sel > 0 && (!target_inac || !weight_inac || !ii_inac || !jj_inac || n_inac == 0)) return DBA_ERR_ARG;
  if (n_act > 0 && (!target_act || !weight_act || !ii_act || !jj_act)) return DBA_ERR_ARG;
  const int HW = ht * wd;
  hipLaunchKernelGGL(ba_gather_edges_kernel, dim3((HW + 255) / 256, n, 2), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float2 *>(target_inac), reinterpret_cast<const float2 *>(weight_inac), ii_inac, jj_inac,
                     sel, n_sel, n_inac, reinterpret_cast<const float2 *>(target_act), reinterpret_cast<const float2 *>(weight_act),
                     ii_act, jj_act, HW, targets_out, weights_out, ii_out, jj_out);
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

int dba_ba_solver_verdict(int N, int B, int ht, int wd, int t0, int t1, void *ws, size_t ws_bytes) {
  BaPlan plan;
  const int rc = ba_plan(N, B, ht, wd, t0, t1, ws, ws_bytes, &plan);
  if (rc != DBA_OK) return rc;
  if (!ba_solve_wave_supported(6 * plan.P)) return 2;
  const int *slot = solver_verdict_slot(plan.T.meta);
  return slot ? __atomic_load_n(slot, __ATOMIC_RELAXED) : 0;
}

int dba_ba_workspace_init(int N, int B, int ht, int wd, int t0, int t1, void *ws, size_t ws_bytes, dba_stream_t stream) {
  BaPlan plan;
  const int rc = ba_plan(N, B, ht, wd, t0, t1, ws, ws_bytes, &plan);
  if (rc != DBA_OK) return rc;
  // meta and the graph key are adjacent: no graph is recorded, nothing was solved
  DBA_HIP_CHECK(hipMemsetAsync(plan.T.meta, 0, (size_t)((char *)(plan.T.gkey + 8) - (char *)plan.T.meta), (hipStream_t)stream));
  ws_words_reset(plan.T.meta);   // (the allocator may have handed out the address of a workspace of another shape or graph)
  return DBA_OK;
}

int dba_ba_prepare(const int64_t *ii, const int64_t *jj, int N, int B, int ht, int wd, int t0, int t1,
                   void *ws, size_t ws_bytes, dba_stream_t stream) {
  return dba_ba_prepare_keyed(ii, jj, N, B, ht, wd, t0, t1, 0, 0, ws, ws_bytes, stream);
}

// judge_band: the solves that follow use THIS graph's skyline (dba_ba; not the sharded sequence, whose ranks see their own edges
// only, and not BACore, whose system is solved on the host): stage 0 then also tells the host whether the window solver takes it
static int ba_prepare_keyed(const int64_t *ii, const int64_t *jj, int N, int B, int ht, int wd, int t0, int t1, int eta_rows,
                            int check, void *ws, size_t ws_bytes, dba_stream_t stream, bool judge_band);

int dba_ba_prepare_keyed(const int64_t *ii, const int64_t *jj, int N, int B, int ht, int wd, int t0, int t1, int eta_rows,
                         int check, void *ws, size_t ws_bytes, dba_stream_t stream) {
  return ba_prepare_keyed(ii, jj, N, B, ht, wd, t0, t1, eta_rows, check, ws, ws_bytes, stream, false);
}

static int ba_prepare_keyed(const int64_t *ii, const int64_t *jj, int N, int B, int ht, int wd, int t0, int t1, int eta_rows,
                            int check, void *ws, size_t ws_bytes, dba_stream_t stream, bool judge_band) {
  BaPlan plan;
  const int rc = ba_plan(N, B, ht, wd, t0, t1, ws, ws_bytes, &plan);
  if (rc != DBA_OK) return rc;
  if (N > 0 && (!ii || !jj)) return DBA_ERR_ARG;
  // threads: enough for one edge / frame / pose each, at most 1024 (barriers among 2 waves cost a fraction of 16)
  const int want = std::max(std::max(N, B), t1 - t0);
  const int threads = std::min(1024, std::max(64, (want + 63) / 64 * 64));
  const size_t scan_ints = std::max<size_t>(std::max(threads, t1 - t0), N > threads ? 1024 : 0) + 32;
  // (+ per-slot row counts for the frame row table: Mmax + 1 ints)
  const size_t lds = sizeof(int) * ((size_t)B + 2 * (size_t)plan.T.Mmax + 1 + scan_ints + plan.T.Mmax + 1);
  if (lds > 160 * 1024 || t1 - t0 > 16384) return DBA_ERR_UNSUPPORTED;
  if (lds > 64 * 1024) {
    DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_prepare_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  const int max_nt = judge_band ? ba_solve_wave_max_nt(6 * plan.P) : 0;
  int *band_verdict = max_nt ? solver_verdict_slot(plan.T.meta) : nullptr;
  hipLaunchKernelGGL(ba_prepare_kernel, dim3(1), dim3(threads), lds, (hipStream_t)stream, ii, jj, N, B, t0, t1,
                     (int)scan_ints, ba_schur_frame_form(N, plan.P) ? 1 : 0, check ? 1 : 0, eta_rows,
                     eta_rows > 1 ? ws_eta_status(plan.T.meta) : nullptr, plan.T, band_verdict, max_nt);
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

// upd: 0 = linearise the state as stored; bit 0 = poses still need Exp(W.dx) (retracted on the fly, the retracted window is
// stored in poses_out), bit 1 = the depths still need the previous iteration's dz (applied in place in disps_w)
static int ba_linearize_stage(const float *poses, const float *disps, const float *intrinsics,
                              const float *disps_sens, const float *targets, const float *weights,
                              const float *eta, int eta_rows, const int64_t *jj,
                              const uint8_t *frame_owned, int N, int B, int ht, int wd, int t0, int t1,
                              float alpha, int upd, float *poses_out, float *disps_w, void *ws, size_t ws_bytes,
                              dba_stream_t stream) {
  BaPlan plan;
  const int rc = ba_plan(N, B, ht, wd, t0, t1, ws, ws_bytes, &plan);
  if (rc != DBA_OK) return rc;
  if (!poses || !disps || !intrinsics || !disps_sens || !eta || eta_rows < 1) return DBA_ERR_ARG;
  // eta has one row per entry of kx, or one row that is broadcast (eta.view(-1, HW), droid_kernels.cu:1476); more rows
  // than kx can have entries cannot be right (the exact |kx| is only known on the device: see droid_backends._ba_args)
  if (eta_rows > 1 && eta_rows > plan.T.Mmax) return DBA_ERR_ARG;
  if (N > 0 && (!targets || !weights || !jj)) return DBA_ERR_ARG;
  // (EW waves share a pixel slice and split the frame's edges: EW times as many workgroups of four waves)
#define LAUNCH_LIN(PPL, MF, EW)                                                                                \
  hipLaunchKernelGGL((ba_linearize_kernel<PPL, MF, EW>), dim3(plan.nchunks * EW, plan.T.Mmax + 1), dim3(256), 0, \
                     (hipStream_t)stream, poses, disps, intrinsics, disps_sens, targets, weights, eta, eta_rows, jj, \
                     frame_owned, N, plan.HW, wd, t0, plan.P, alpha, upd, poses_out, disps_w, plan.T, plan.W)
  // one pixel per lane: the per-edge sums run on the matrix cores (DBA_LINEARIZE_MFMA=0 keeps the LDS transpose-reduce)
  static const bool no_mfma = [] { const char *e = getenv("DBA_LINEARIZE_MFMA"); return e && e[0] == '0'; }();
  if (plan.W.ppl == 4) LAUNCH_LIN(4, false, 1);
  else if (plan.W.ppl == 2) LAUNCH_LIN(2, false, 1);
  else if (no_mfma) LAUNCH_LIN(1, false, 1);
  else {
    // Two waves per pixel slice halve a wave's life but pay the prologue / epilogue (5.9 of ~19 us at 64 KF / 512 edges) twice:
    // worth it while every wave of the launch is resident at once (25 KF: 3200 waves on 4096 places), not when the launch
    // already comes in rounds (64 KF: 8192 waves).  DBA_LIN_EW=1|2 forces one.
    static const int env_ew = [] { const char *e = getenv("DBA_LIN_EW"); return e ? atoi(e) : 0; }();
    const long waves2 = (long)plan.nchunks * 2 * 4 * plan.T.Mmax;
    const bool one = env_ew ? env_ew == 1 : waves2 > 4096;
    if (one) LAUNCH_LIN(1, true, 1);
    else LAUNCH_LIN(1, true, 2);
  }
#undef LAUNCH_LIN
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

int dba_ba_linearize(const float *poses, const float *disps, const float *intrinsics,
                     const float *disps_sens, const float *targets, const float *weights,
                     const float *eta, int eta_rows, const int64_t *ii, const int64_t *jj,
                     const uint8_t *frame_owned, int N, int B, int ht, int wd, int t0, int t1,
                     float alpha, void *ws, size_t ws_bytes, dba_stream_t stream) {
  (void)ii;
  return ba_linearize_stage(poses, disps, intrinsics, disps_sens, targets, weights, eta, eta_rows, jj, frame_owned, N, B,
                            ht, wd, t0, t1, alpha, 0, nullptr, nullptr, ws, ws_bytes, stream);
}

// lower != 0: only the lower triangle of H is kept up (half the float64 atomics; the solvers read nothing else)
static int ba_reduce_stage(const int64_t *ii, const int64_t *jj, const uint8_t *frame_owned, int N, int B, int ht,
                           int wd, int t0, int t1, int motion_only, int lower, void *ws, size_t ws_bytes,
                           dba_stream_t stream) {
  BaPlan plan;
  const int rc = ba_plan(N, B, ht, wd, t0, t1, ws, ws_bytes, &plan);
  if (rc != DBA_OK) return rc;
  const int ablocks = (N + 3) / 4;   // pose-block assembly: one wave per edge (round 6: the per-frame blocks are products of the per-edge ones)
  if (plan.P <= 0) return DBA_OK;
  static const bool force_full = [] { const char *e = getenv("DBA_H_FULL"); return e && e[0] == '1'; }();
  if (force_full) lower = 0;
  const bool fixed = g_deterministic.load(std::memory_order_relaxed) != 0;
  if (fixed) lower |= 2;
  if (!motion_only) {  // Schur products and the pose-block assembly share one launch (both only add into H, b)
    // per-source-frame form (every row of E read once, Gram tiles on the matrix cores); DBA_SCHUR_KERNEL=rows keeps
    // the (row, partner) grid, which also takes graphs with more edges than the prepare kernel lists per frame
    static const int env_nch = [] { const char *e = getenv("DBA_SCHUR_NCH"); return e ? atoi(e) : 0; }();
    if (!ba_schur_frame_form(N, plan.P)) {
      hipLaunchKernelGGL(ba_schur_kernel, dim3(plan.P + N + ablocks, SCHUR_KP, SCHUR_CH), dim3(256), 0,
                         (hipStream_t)stream, ii, jj, frame_owned, N, plan.HW, t0, plan.P, lower, plan.T, plan.W);
    } else {
      // pixel chunks per frame: ~1024 pixels per workgroup (256 per wave).  A function of the map size alone: a rank of
      // the sharded driver must cut a frame exactly as a single GPU does (same partial sums, bit for bit)
      const int px = 1024;
      int nch = env_nch > 0 ? env_nch : (plan.HW + px - 1) / px;
      nch = std::max(1, std::min(nch, (plan.HW + 15) / 16));
      const dim3 grid((unsigned)(plan.T.Mmax * nch + ablocks));
      // eight waves per workgroup: two per SIMD, whose matrix products and operand loads interleave (with four, one per
      // SIMD, a wave waited 2.5 us for every 1.6 us of products: 47 us at 64 KF / 512 edges)
      static const int gram_threads = [] { const char *e = getenv("DBA_SCHUR_WAVES"); return (e && atoi(e) == 4) ? 256 : 512; }();
      // products on the float64 matrix pipe (exact) or as 16-term float chains flushed into float64 (the row-pair kernel's
      // precision class, half the pipe time): DBA_SCHUR_MFMA=f64|f32
      static const bool f32 = [] { const char *e = getenv("DBA_SCHUR_MFMA"); return e ? (e[0] == 'f' && e[1] == '3') : GRAM_F32_DEFAULT; }();
#define GRAM_LAUNCH(V, F)                                                                                              \
  hipLaunchKernelGGL((ba_schur_gram_kernel<V, F>), grid, dim3(gram_threads), 0, (hipStream_t)stream, ii, jj, frame_owned, \
                     N, plan.HW, t0, plan.P, nch, lower, plan.T, plan.W)
      if (plan.HW % 4 == 0) { if (f32) GRAM_LAUNCH(true, true); else GRAM_LAUNCH(true, false); }
      else { if (f32) GRAM_LAUNCH(false, true); else GRAM_LAUNCH(false, false); }
#undef GRAM_LAUNCH
    }
    DBA_LAUNCH_CHECK();
  } else if (ablocks > 0) {
    hipLaunchKernelGGL(ba_assemble_kernel, dim3(ablocks), dim3(256), 0, (hipStream_t)stream, ii, jj, frame_owned,
                       N, t0, plan.P, lower, plan.T, plan.W);
    DBA_LAUNCH_CHECK();
  }
  if (fixed) {  // the fixed-point sums back to float64 (one more launch: the price of the opt-in mode)
    const int n = 6 * plan.P;
    hipLaunchKernelGGL(ba_fixed_to_f64_kernel, dim3((n * n + n + 255) / 256), dim3(256), 0, (hipStream_t)stream, plan.W.H,
                       plan.W.b, n);
    DBA_LAUNCH_CHECK();
  }
  return DBA_OK;
}

// the stage as the C ABI exposes it (BACore.hessian, ba_extend, tests): the FULL matrix, mirrored from the lower triangle,
// so that H is symmetric to the last bit and identical to what the sharded BACore hands over
int dba_ba_reduce(const int64_t *ii, const int64_t *jj, const uint8_t *frame_owned, int N, int B, int ht,
                  int wd, int t0, int t1, int motion_only, void *ws, size_t ws_bytes, dba_stream_t stream) {
  const int rc = ba_reduce_stage(ii, jj, frame_owned, N, B, ht, wd, t0, t1, motion_only, 1, ws, ws_bytes, stream);
  if (rc != DBA_OK) return rc;
  return dba_ba_symmetrize(N, B, ht, wd, t0, t1, ws, ws_bytes, stream);
}

int dba_ba_schur_select(int form) {
  if (form < 0 || form > 2) return DBA_ERR_ARG;
  g_schur_form.store(form, std::memory_order_relaxed);
  g_schur_generation.fetch_add(1, std::memory_order_relaxed);
  return DBA_OK;
}

int dba_ba_schur_select_thread(int form) {
  if (form < 0 || form > 2) return DBA_ERR_ARG;
  // (which tables stage 0 builds depends on the form in force -- the frame row table exists only for the per-frame form --,
  // so whoever skips stage 0 on a prepared workspace must key it on dba_ba_schur_generation() AND dba_ba_schur_thread_form())
  t_schur_form = form;
  return DBA_OK;
}

int dba_ba_schur_thread_form(void) { return t_schur_form; }

int dba_ba_schur_auto_form(int N, int P) { return schur_auto_frame_form(N, P) ? 2 : 1; }

int dba_ba_set_deterministic(int on) {
  g_deterministic.store(on ? 1 : 0, std::memory_order_relaxed);
  return DBA_OK;
}

int dba_ba_solve_check(int N, int B, int ht, int wd, int t0, int t1, float lm, float ep, void *ws, size_t ws_bytes,
                       dba_stream_t stream) {
  BaPlan plan;
  const int rc = ba_plan(N, B, ht, wd, t0, t1, ws, ws_bytes, &plan);
  if (rc != DBA_OK) return rc;
  if (plan.P <= 0) return DBA_OK;
  hipLaunchKernelGGL(ba_solve_check_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, plan.W.H, plan.W.b, 6 * plan.P, (double)lm,
                     (double)ep, plan.W.dx, plan.T.meta);
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

int dba_ba_set_solve_check(int on) {
  g_solve_check.store(on ? 1 : 0, std::memory_order_relaxed);
  return DBA_OK;
}

int dba_ba_schur_generation(void) { return g_schur_generation.load(std::memory_order_relaxed); }

int dba_ba_symmetrize(int N, int B, int ht, int wd, int t0, int t1, void *ws, size_t ws_bytes, dba_stream_t stream) {
  BaPlan plan;
  const int rc = ba_plan(N, B, ht, wd, t0, t1, ws, ws_bytes, &plan);
  if (rc != DBA_OK) return rc;
  const int n = 6 * plan.P;
  if (n <= 0) return DBA_OK;
  hipLaunchKernelGGL(ba_symmetrize_kernel, dim3((n * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, plan.W.H, n);
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

// graph_skyline: the caller guarantees that H only has the structure of THIS workspace's graph (edges + Schur fill),
// so the solver may take the skyline from the prepare kernel's table instead of measuring it.  False for systems
// that were summed over ranks or that came from the host (BACore.optimize: GTSAM priors can couple any two poses).
// graph_fpose: a caller-supplied pose-level skyline (device, P ints) for a system summed over ranks whose combined graph
// the caller knows (the sharded driver); overrides the workspace's table.
static int ba_solve_stage(int N, int B, int ht, int wd, int t0, int t1, float lm, float ep, void *ws, size_t ws_bytes,
                          dba_stream_t stream, bool graph_skyline, const int *graph_fpose = nullptr, int hint = 0) {
  BaPlan plan;
  const int rc = ba_plan(N, B, ht, wd, t0, t1, ws, ws_bytes, &plan);
  if (rc != DBA_OK) return rc;
  const int *fpose = graph_fpose ? graph_fpose : (graph_skyline ? plan.T.fpose : nullptr);
  // (the plan kept in the workspace belongs to the workspace's own skyline table: a caller-supplied one is planned every time)
  const int rc2 = launch_ba_solve(plan.W.H, plan.W.b, fpose, 6 * plan.P, (double)lm, (double)ep, plan.W.dx, plan.T.meta,
                                  plan.W.Lscratch, (hipStream_t)stream, nullptr, hint,
                                  (fpose && fpose == plan.T.fpose) ? plan.T.meta + 16 : nullptr);
  if (rc2 != DBA_OK) return rc2;
  // opt-in guard (dba_ba_set_solve_check / DBA_SOLVE_CHECK=1): the residual of the damped system at the solution, by a kernel of
  // its own behind the solver -- a solve that went wrong silently (the window solver's waves meet through flags, not barriers)
  // becomes a zero update with meta[1] = 1, which is what a failed factorisation gives (droid_kernels.cu:1263-1266)
  if (g_solve_check.load(std::memory_order_relaxed) && plan.P > 0) {
    hipLaunchKernelGGL(ba_solve_check_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, plan.W.H, plan.W.b, 6 * plan.P, (double)lm,
                       (double)ep, plan.W.dx, plan.T.meta);
    DBA_LAUNCH_CHECK();
  }
  return DBA_OK;
}

int dba_ba_solve(int N, int B, int ht, int wd, int t0, int t1, float lm, float ep, void *ws, size_t ws_bytes,
                 dba_stream_t stream) {
  return ba_solve_stage(N, B, ht, wd, t0, t1, lm, ep, ws, ws_bytes, stream, false);
}

int dba_ba_solve_skyline(int N, int B, int ht, int wd, int t0, int t1, float lm, float ep, const int32_t *fpose, void *ws,
                         size_t ws_bytes, dba_stream_t stream) {
  return ba_solve_stage(N, B, ht, wd, t0, t1, lm, ep, ws, ws_bytes, stream, false, fpose);
}

static int ba_update_launch(float *poses, float *disps, const int64_t *jj, const uint8_t *frame_owned, int N, int B,
                            int ht, int wd, int t0, int t1, int update_poses, int update_disps, float *dz_out,
                            float *dx_out, void *ws, size_t ws_bytes, dba_stream_t stream,
                            const float *poses_src = nullptr, float disp_floor = 0.f) {
  BaPlan plan;
  const int rc = ba_plan(N, B, ht, wd, t0, t1, ws, ws_bytes, &plan);
  if (rc != DBA_OK) return rc;
  // (disp_floor > 0: one more block row per frame of the buffer, which clamp the frames this launch does not update)
  dim3 grid((plan.HW + 255) / 256, plan.T.Mmax + 1 + (disp_floor > 0.f ? B : 0));
  hipLaunchKernelGGL(ba_update_kernel, grid, dim3(256), 0, (hipStream_t)stream, poses, poses_src, disps, jj, frame_owned,
                     plan.HW, t0, plan.P, update_poses, update_disps, dz_out, dx_out, plan.T, plan.W, disp_floor);
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

int dba_ba_update(float *poses, float *disps, const int64_t *ii, const int64_t *jj,
                  const uint8_t *frame_owned, int N, int B, int ht, int wd, int t0, int t1, int update_poses,
                  int update_disps, float *dz_out, void *ws, size_t ws_bytes, dba_stream_t stream) {
  (void)ii;
  return ba_update_launch(poses, disps, jj, frame_owned, N, B, ht, wd, t0, t1, update_poses, update_disps, dz_out,
                          nullptr, ws, ws_bytes, stream);
}

int dba_ba_shard_front(const float *poses, const float *disps, const float *intrinsics, const float *disps_sens,
                       const float *targets, const float *weights, const float *eta, int eta_rows, const int64_t *ii,
                       const int64_t *jj, const uint8_t *frame_owned, int N, int B, int ht, int wd, int t0, int t1,
                       float alpha, int motion_only, void *ws, size_t ws_bytes, dba_stream_t stream) {
  const int rc = dba_ba_linearize(poses, disps, intrinsics, disps_sens, targets, weights, eta, eta_rows, ii, jj,
                                  frame_owned, N, B, ht, wd, t0, t1, alpha, ws, ws_bytes, stream);
  if (rc != DBA_OK) return rc;
  // (lower triangle only: what the redundant solves read; ShardedBACore mirrors it before handing H to the host)
  return ba_reduce_stage(ii, jj, frame_owned, N, B, ht, wd, t0, t1, motion_only, 1, ws, ws_bytes, stream);
}

int dba_ba_shard_back(float *poses, float *disps, const int64_t *ii, const int64_t *jj, const uint8_t *frame_owned,
                      int N, int B, int ht, int wd, int t0, int t1, float lm, float ep, int update_disps,
                      const int32_t *window_fpose, int solver_hint, void *ws, size_t ws_bytes, dba_stream_t stream) {
  const int rc = ba_solve_stage(N, B, ht, wd, t0, t1, lm, ep, ws, ws_bytes, stream, false, window_fpose, solver_hint);
  if (rc != DBA_OK) return rc;
  return dba_ba_update(poses, disps, ii, jj, frame_owned, N, B, ht, wd, t0, t1, 1, update_disps, nullptr, ws, ws_bytes,
                       stream);
}

// prepared = 1: the index tables in `ws` are those of this graph already (a previous dba_ba / dba_ba_prepare with the
// same ii, jj, sizes, t0, t1 and Schur form on this workspace): stage 0 is skipped.  prepared = 2: stage 0 decides that
// itself, on the device, by comparing the edge list with the key it left in the workspace (dba_ba_prepare_keyed).
extern "C++" {
int dba::ba_run_loop(float *poses, float *disps, const float *intrinsics, const float *disps_sens, const float *targets,
                const float *weights, const float *eta, int eta_rows, const int64_t *ii, const int64_t *jj,
                const uint8_t *frame_owned, int N, int B, int ht, int wd, int t0, int t1, int iterations, float lm, float ep,
                float alpha, int motion_only, float *dx_out, float *dz_out, void *ws, size_t ws_bytes, dba_stream_t stream,
                int prepared, int solver_hint, float disp_floor, const int32_t *window_fpose, const BaExchange *exchange) {
  BaPlan plan;
  int rc = ba_plan(N, B, ht, wd, t0, t1, ws, ws_bytes, &plan);
  if (rc != DBA_OK) return rc;
  if (prepared != 1) {
    rc = ba_prepare_keyed(ii, jj, N, B, ht, wd, t0, t1, eta_rows, prepared == 2, ws, ws_bytes, stream,
                          window_fpose == nullptr);
    if (rc != DBA_OK) return rc;
  }
  // Back-substitution + retraction of iteration k are folded into the linearisation of iteration k + 1 (one launch less
  // per iteration; DBA_BA_FUSE_UPDATE=0 keeps them apart): the poses the next launch reads stay untouched, the retracted
  // window travels through two workspace copies, and only the last iteration's update is a launch of its own, which
  // writes the caller's poses.
  static const bool fuse = [] { const char *e = getenv("DBA_BA_FUSE_UPDATE"); return !(e && e[0] == '0'); }();
  const float *pose_src = poses;   // where the current poses are (before the pending retraction, if any)
  bool pending = false;            // W.dx of the previous iteration has not been applied yet
  for (int itr = 0; itr < iterations; itr++) {
    float *pose_dst = plan.W.poses_tmp + (size_t)(itr & 1) * 7 * B;
    const int upd = pending ? (motion_only ? 1 : 3) : 0;
    rc = ba_linearize_stage(pose_src, disps, intrinsics, disps_sens, targets, weights, eta, eta_rows, jj, frame_owned, N, B,
                            ht, wd, t0, t1, alpha, upd, pending ? pose_dst : nullptr, disps, ws, ws_bytes, stream);
    if (rc != DBA_OK) return rc;
    if (pending) pose_src = pose_dst;
    rc = ba_reduce_stage(ii, jj, frame_owned, N, B, ht, wd, t0, t1, motion_only, 1, ws, ws_bytes, stream);
    if (rc != DBA_OK) return rc;
    if (exchange && exchange->fn && plan.P > 0) {   // the ranks' partial [H | gap | b] become the window's
      const size_t n6 = (size_t)6 * plan.P;
      rc = exchange->fn(exchange->ctx, plan.W.H, (size_t)(plan.W.b - plan.W.H) + n6, (hipStream_t)stream);
      if (rc != DBA_OK) return rc;
    }
    rc = ba_solve_stage(N, B, ht, wd, t0, t1, lm, ep, ws, ws_bytes, stream, window_fpose == nullptr, window_fpose, solver_hint);
    if (rc != DBA_OK) return rc;
    const bool last = (itr == iterations - 1);
    if (fuse && !last) {
      pending = true;
      continue;
    }
    rc = ba_update_launch(poses, disps, jj, frame_owned, N, B, ht, wd, t0, t1, 1, motion_only ? 0 : 1,
                          last ? dz_out : nullptr, last ? dx_out : nullptr, ws, ws_bytes, stream,
                          pose_src == poses ? nullptr : pose_src, last ? disp_floor : 0.f);
    if (rc != DBA_OK) return rc;
    pose_src = poses;
    pending = false;
  }
  return DBA_OK;
}
}  // extern "C++"

static int ba_run(float *poses, float *disps, const float *intrinsics, const float *disps_sens,
                  const float *targets, const float *weights, const float *eta, int eta_rows, const int64_t *ii,
                  const int64_t *jj, int N, int B, int ht, int wd, int t0, int t1, int iterations, float lm,
                  float ep, int motion_only, float *dx_out, float *dz_out, void *ws, size_t ws_bytes,
                  dba_stream_t stream, int prepared, int solver_hint, float disp_floor = 0.f) {
  return ba_run_loop(poses, disps, intrinsics, disps_sens, targets, weights, eta, eta_rows, ii, jj, nullptr, N, B, ht, wd, t0,
                     t1, iterations, lm, ep, 0.05f /* droid_kernels.cu:1474 */, motion_only, dx_out, dz_out, ws, ws_bytes, stream,
                     prepared, solver_hint, disp_floor, nullptr, nullptr);
}

int dba_ba(float *poses, float *disps, const float *intrinsics, const float *disps_sens,
           const float *targets, const float *weights, const float *eta, int eta_rows, const int64_t *ii,
           const int64_t *jj, int N, int B, int ht, int wd, int t0, int t1, int iterations, float lm,
           float ep, int motion_only, float *dx_out, float *dz_out, void *ws, size_t ws_bytes,
           dba_stream_t stream) {
  return ba_run(poses, disps, intrinsics, disps_sens, targets, weights, eta, eta_rows, ii, jj, N, B, ht, wd, t0, t1,
                iterations, lm, ep, motion_only, dx_out, dz_out, ws, ws_bytes, stream, 0, 0);
}

int dba_ba_prepared(float *poses, float *disps, const float *intrinsics, const float *disps_sens,
                    const float *targets, const float *weights, const float *eta, int eta_rows, const int64_t *ii,
                    const int64_t *jj, int N, int B, int ht, int wd, int t0, int t1, int iterations, float lm,
                    float ep, int motion_only, float *dx_out, float *dz_out, void *ws, size_t ws_bytes,
                    dba_stream_t stream, int solver_hint) {
  return ba_run(poses, disps, intrinsics, disps_sens, targets, weights, eta, eta_rows, ii, jj, N, B, ht, wd, t0, t1,
                iterations, lm, ep, motion_only, dx_out, dz_out, ws, ws_bytes, stream, 1, solver_hint);
}

int dba_ba_run(float *poses, float *disps, const float *intrinsics, const float *disps_sens, const float *targets,
               const float *weights, const float *eta, int eta_rows, const int64_t *ii, const int64_t *jj, int N, int B,
               int ht, int wd, int t0, int t1, int iterations, float lm, float ep, int motion_only, float *dx_out,
               float *dz_out, void *ws, size_t ws_bytes, dba_stream_t stream, int prepared, int solver_hint,
               float disp_floor) {
  if (!(disp_floor >= 0.f)) return DBA_ERR_ARG;
  return ba_run(poses, disps, intrinsics, disps_sens, targets, weights, eta, eta_rows, ii, jj, N, B, ht, wd, t0, t1,
                iterations, lm, ep, motion_only, dx_out, dz_out, ws, ws_bytes, stream, (prepared == 1 || prepared == 2) ? prepared : 0,
                prepared == 1 ? solver_hint : 0, disp_floor);
}

int dba_bacore_hessian(const float *poses, const float *disps, const float *intrinsics,
                       const float *disps_sens, const float *targets, const float *weights,
                       const float *eta, int eta_rows, const int64_t *ii, const int64_t *jj, int N, int B,
                       int ht, int wd, int t0, int t1, double *H_host, double *v_host, void *ws,
                       size_t ws_bytes, dba_stream_t stream) {
  return dba_bacore_hessian_run(poses, disps, intrinsics, disps_sens, targets, weights, eta, eta_rows, ii, jj, N, B, ht, wd,
                                t0, t1, H_host, v_host, ws, ws_bytes, stream, 0);
}

// ---- BACore.hessian's hand-over (round 6).  Per workspace (keyed by its meta pointer, like the pinned words of ba_solve.hip): a
// block of pinned, host-coherent, device-mapped memory for the exported system, a completion word in it and a device counter.
// The export kernel writes the system there itself and sets the word when its last workgroup is through; the host spins on the
// word.  Never freed (a process goes through a handful of window shapes; 1.2 MB at 64 poses).
namespace {
struct BacoreStage {
  double *host = nullptr;     // [doubles] payload, then the flag (one int, 64-byte aligned)
  size_t doubles = 0;
  unsigned *counter = nullptr;
  int seq = 0;
};
std::mutex g_stage_mu;
typedef std::unordered_map<const void *, BacoreStage> BacoreStageMap;
BacoreStageMap *stage_map_ptr() {   // (never destroyed: calls may still arrive at exit)
  static BacoreStageMap *m = new BacoreStageMap;
  return m;
}
#define stage_map() (*stage_map_ptr())
int *stage_flag(const BacoreStage &st) { return reinterpret_cast<int *>(st.host + st.doubles + 8); }
}  // namespace

static int bacore_stage_for(const int *meta, size_t need, hipStream_t stream, BacoreStage **out) {
  std::lock_guard<std::mutex> lock(g_stage_mu);
  BacoreStage &st = stage_map()[meta];
  if (st.doubles < need) {
    if (st.host) {   // a larger window at the same address: nothing of the old block may still be in flight
      DBA_HIP_CHECK(hipStreamSynchronize(stream));
      (void)hipHostFree(st.host);
      st.host = nullptr, st.doubles = 0;
    }
    void *p = nullptr;
    DBA_HIP_CHECK(hipHostMalloc(&p, sizeof(double) * (need + 16), hipHostMallocCoherent | hipHostMallocMapped));
    st.host = static_cast<double *>(p), st.doubles = need;
    memset(p, 0, sizeof(double) * (need + 16));
    if (!st.counter) {
      DBA_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&st.counter), 64));
      DBA_HIP_CHECK(hipMemset(st.counter, 0, 64));
    }
  }
  *out = &st;
  return DBA_OK;
}

// spins until the export kernel's word shows `seq`; every few thousand looks asks the runtime whether the stream has drained
// (a launch that failed never sets the word)
static int bacore_wait(int *flag, int seq, hipStream_t stream) {
  for (unsigned long spins = 1;; spins++) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return DBA_OK;
    if ((spins & 0x3fff) == 0) {
      const hipError_t q = hipStreamQuery(stream);
      if (q == hipSuccess) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return DBA_OK;
        set_last_error("BACore.hessian: the stream drained without the export kernel's completion word", hipErrorUnknown);
        return DBA_ERR_HIP;
      }
      if (q != hipErrorNotReady) {
        set_last_error("hipStreamQuery", q);
        return DBA_ERR_HIP;
      }
    }
    __builtin_ia32_pause();
  }
}

int dba_bacore_hessian_host(const float *poses, const float *disps, const float *intrinsics,
                            const float *disps_sens, const float *targets, const float *weights,
                            const float *eta, int eta_rows, const int64_t *ii, const int64_t *jj, int N, int B,
                            int ht, int wd, int t0, int t1, void *ws, size_t ws_bytes, dba_stream_t stream, int prepared,
                            int layout, const double *A36, double stabilizer, double **out_host) {
  BaPlan plan;
  int rc = ba_plan(N, B, ht, wd, t0, t1, ws, ws_bytes, &plan);
  if (rc != DBA_OK) return rc;
  if (!out_host || (layout != 0 && layout != 1) || (layout == 1 && !A36)) return DBA_ERR_ARG;
  if (prepared != 1) {
    rc = dba_ba_prepare_keyed(ii, jj, N, B, ht, wd, t0, t1, eta_rows, prepared == 2, ws, ws_bytes, stream);
    if (rc != DBA_OK) return rc;
  }
  rc = dba_ba_linearize(poses, disps, intrinsics, disps_sens, targets, weights, eta, eta_rows, ii, jj, nullptr,
                        N, B, ht, wd, t0, t1, 0.001f /* :1872 */, ws, ws_bytes, stream);
  if (rc != DBA_OK) return rc;
  // (the lower triangle is all the export reads: no mirroring launch)
  rc = ba_reduce_stage(ii, jj, nullptr, N, B, ht, wd, t0, t1, 0, 1, ws, ws_bytes, stream);
  if (rc != DBA_OK) return rc;
  return dba_bacore_export_host(N, B, ht, wd, t0, t1, ws, ws_bytes, stream, layout, A36, stabilizer, out_host);
}

int dba_bacore_export_host(int N, int B, int ht, int wd, int t0, int t1, void *ws, size_t ws_bytes, dba_stream_t stream, int layout,
                           const double *A36, double stabilizer, double **out_host) {
  BaPlan plan;
  int rc = ba_plan(N, B, ht, wd, t0, t1, ws, ws_bytes, &plan);
  if (rc != DBA_OK) return rc;
  if (!out_host || (layout != 0 && layout != 1) || (layout == 1 && !A36)) return DBA_ERR_ARG;
  const int n = 6 * plan.P;
  BacoreStage *st = nullptr;
  rc = bacore_stage_for(plan.T.meta, (size_t)n * (n + 1), (hipStream_t)stream, &st);
  if (rc != DBA_OK) return rc;
  *out_host = st->host;
  if (n == 0) return DBA_OK;
  ExportArg arg;
  for (int i = 0; i < 36; i++) arg.A[i] = (layout == 1) ? A36[i] : 0.0;
  const int seq = ++st->seq;
  const int total = n * (n + 1);
  hipLaunchKernelGGL(ba_export_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, plan.W.H, plan.W.b, n,
                     st->host, layout, arg, stabilizer, st->counter, stage_flag(*st), seq);
  DBA_LAUNCH_CHECK();
  return bacore_wait(stage_flag(*st), seq, (hipStream_t)stream);
}

int dba_bacore_staging(int N, int B, int ht, int wd, int t0, int t1, void *ws, size_t ws_bytes, double **out_host) {
  BaPlan plan;
  const int rc = ba_plan(N, B, ht, wd, t0, t1, ws, ws_bytes, &plan);
  if (rc != DBA_OK) return rc;
  if (!out_host) return DBA_ERR_ARG;
  std::lock_guard<std::mutex> lock(g_stage_mu);
  auto it = stage_map().find(plan.T.meta);
  *out_host = (it == stage_map().end()) ? nullptr : it->second.host;
  return *out_host ? DBA_OK : DBA_ERR_ARG;
}

int dba_bacore_hessian_run(const float *poses, const float *disps, const float *intrinsics,
                           const float *disps_sens, const float *targets, const float *weights,
                           const float *eta, int eta_rows, const int64_t *ii, const int64_t *jj, int N, int B,
                           int ht, int wd, int t0, int t1, double *H_host, double *v_host, void *ws,
                           size_t ws_bytes, dba_stream_t stream, int prepared) {
  double *stage = nullptr;
  const int rc = dba_bacore_hessian_host(poses, disps, intrinsics, disps_sens, targets, weights, eta, eta_rows, ii, jj, N, B, ht, wd,
                                         t0, t1, ws, ws_bytes, stream, prepared, 0, nullptr, 0.0, &stage);
  if (rc != DBA_OK) return rc;
  const size_t n = (size_t)6 * std::max(t1 - t0, 0);
  if (n && H_host && H_host != stage) memcpy(H_host, stage, sizeof(double) * n * n);
  if (n && v_host && v_host != stage + n * n) memcpy(v_host, stage + n * n, sizeof(double) * n);
  return DBA_OK;
}


// the externally solved update as a kernel ARGUMENT (up to 64 poses: 1.5 KB of the 4 KB an argument block may have): it reaches
// the device with the launch itself -- no staging copy, no stream synchronisation before the host buffer may go away
struct DxArg {
  float v[384];
};
__global__ void ba_dx_from_arg_kernel(DxArg a, float *__restrict__ dx, float *__restrict__ dx_out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    dx[i] = a.v[i];
    if (dx_out) dx_out[i] = a.v[i];   // (the caller's copy rides along: one launch less)
  }
}


int dba_bacore_retract(float *poses, float *disps, const int64_t *ii, const int64_t *jj, int N, int B, int ht,
                       int wd, int t0, int t1, const double *dx_host, float *dx_out, float *dz_out, void *ws,
                       size_t ws_bytes, dba_stream_t stream) {
  BaPlan plan;
  int rc = ba_plan(N, B, ht, wd, t0, t1, ws, ws_bytes, &plan);
  if (rc != DBA_OK) return rc;
  const int n = 6 * plan.P;
  if (n > 0 && n <= 384) {
    if (!dx_host) return DBA_ERR_ARG;
    DxArg a;
    for (int i = 0; i < n; i++) a.v[i] = (float)dx_host[i];  // f64 -> f32 (:1929-1930)
    hipLaunchKernelGGL(ba_dx_from_arg_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, a, plan.W.dx, dx_out, n);
    DBA_LAUNCH_CHECK();
  } else if (n > 0) {
    if (!dx_host) return DBA_ERR_ARG;
    std::vector<float> dxf((size_t)n);
    for (int i = 0; i < n; i++) dxf[i] = (float)dx_host[i];  // f64 -> f32 (:1929-1930)
    DBA_HIP_CHECK(hipMemcpyAsync(plan.W.dx, dxf.data(), sizeof(float) * n, hipMemcpyHostToDevice,
                                 (hipStream_t)stream));
    DBA_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));  // dxf goes out of scope
  }
  rc = dba_ba_update(poses, disps, ii, jj, nullptr, N, B, ht, wd, t0, t1, 1, 1, dz_out, ws, ws_bytes, stream);
  if (rc != DBA_OK) return rc;
  if (dx_out && n > 384) {
    hipLaunchKernelGGL(ba_copy_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, plan.W.dx,
                       dx_out, n);
    DBA_LAUNCH_CHECK();
  }
  return DBA_OK;
}

int dba_bacore_optimize(const double *H_host, const double *v_host, int N, int B, int ht, int wd, int t0,
                        int t1, float lm, float ep, float *dx_out, void *ws, size_t ws_bytes,
                        dba_stream_t stream) {
  BaPlan plan;
  int rc = ba_plan(N, B, ht, wd, t0, t1, ws, ws_bytes, &plan);
  if (rc != DBA_OK) return rc;
  const size_t n = (size_t)6 * plan.P;
  if (n == 0) return DBA_OK;
  if (!H_host || !v_host) return DBA_ERR_ARG;
  DBA_HIP_CHECK(hipMemcpyAsync(plan.W.H, H_host, sizeof(double) * n * n, hipMemcpyHostToDevice,
                               (hipStream_t)stream));
  DBA_HIP_CHECK(hipMemcpyAsync(plan.W.b, v_host, sizeof(double) * n, hipMemcpyHostToDevice,
                               (hipStream_t)stream));
  rc = dba_ba_solve(N, B, ht, wd, t0, t1, lm, ep, ws, ws_bytes, stream);
  if (rc != DBA_OK) return rc;
  if (dx_out) {
    hipLaunchKernelGGL(ba_copy_f32_kernel, dim3(((int)n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       plan.W.dx, dx_out, (int)n);
    DBA_LAUNCH_CHECK();
  }
  DBA_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));  // host buffers may be freed on return
  return DBA_OK;
}

}  // extern "C"
