// Internal declarations shared by the BA kernels and their host launcher.
#pragma once
#include "common.h"

namespace dba {

constexpr int HPE_STRIDE = 32;  // floats per per-wave, per-edge partial (27 used: Hjj lower triangle, vj)
#ifndef SCHUR_KP_CFG
#define SCHUR_KP_CFG 8
#endif
#ifndef SCHUR_CH_CFG
#define SCHUR_CH_CFG 2
#endif
constexpr int SCHUR_KP = SCHUR_KP_CFG;     // partner slots of the Schur grid
constexpr int SCHUR_CH = SCHUR_CH_CFG;     // pixel chunks of the Schur grid

// device index tables (all int32, inside the workspace)
struct BaTables {
  int *meta;        // [0] = |kx|, [1] = last solve failed, [2] = kx overflowed Mmax (cannot happen), [3] = solved by the
                    // skyline kernel, [4..6] = its split, [7] = which of its variants solved (1 / 2), [8..15] = handshake flags of its two workgroups (16 ints)
  int *kx;          // [Mmax]   frame id of slot m (sorted unique of arange(t0,t1) U ii)
  int *frame_slot;  // [B]      slot of frame f, -1 if absent
  int *eoff;        // [Mmax+1] CSR offsets of the out-edges of slot m
  int *elist;       // [N]      edge ids, ascending within a slot
  int *elist_rank;  // [N]      scratch of the prepare kernel
  int *fpose;       // [P]      first pose (index - t0) the reduced system couples pose p with (edges + Schur fill)
  // flat copies of the above for the per-iteration kernels: one table row per workgroup instead of a chain of
  // dependent lookups (ii -> frame_slot -> eoff -> elist -> jj is five round trips of ~1 us each before any work)
  int *einfo;       // [N][2]     per list position: edge id, target frame jj
  int *rowinfo;     // [P+N][8]   per row of E (pose p | P + edge n): slot (-1: nothing to do), target pose - t0,
                    //            first partner position, end of the slot's list, source frame
  // per source frame, for the per-source-frame Schur kernel: the rows of E it couples (its own pose row, then its
  // out-edges with a pose inside the window, in list order)
  int *fhead;       // [Mmax][4]  frame id (-1: slot unused), first entry in frow, number of rows, -
  int *frow;        // [P+N][2]   row of E, pose index (- t0) it belongs to
  int *gkey;        // [8 + 2N]   the graph these tables were built for: magic, N, B, t0, t1, Schur form, Mmax, -, then ii, jj
                    //            as int32 (stage 0 compares a call's edge list with it and skips itself: dba_ba_prepare_keyed)
  int Mmax, B;
};

struct BaBuffers {
  float *E;      // [(P+N), 6, HW]  rows 0..P-1 = Ei (pose i of frame t0+p), rows P+n = Eij of edge n
  float *Q;      // [Mmax, HW]      1 / C
  float *w;      // [Mmax, HW]
  float *HpartE; // [N, nparts, HPE_STRIDE]
  double *Aedge; // [N, 36]  per edge: A with Ji = Jj A (row-major), written by the linearisation, read by the assembly
  double *H;     // [6P, 6P]
  double *b;     // [6P]
  float *dx;     // [P, 6]
  double *Lscratch;  // packed lower triangle for systems too large for LDS
  float *poses_tmp;  // [2][B, 7] the window as retracted by the iterations folded into the next linearisation (dba_ba)
  int nparts;    // pixel slices (waves) per frame for the chosen pixels-per-lane
  int ppl;       // pixels per lane of the linearisation kernel (1, 2 or 4)
};

struct BaPlan {  // host-side view of the workspace
  BaTables T;
  BaBuffers W;
  dba_ba_layout layout;
  size_t bytes;
  int P, N, B, HW, nchunks;
};

int ba_plan(int N, int B, int ht, int wd, int t0, int t1, void *ws, size_t ws_bytes, BaPlan *plan);

// kernels (ba_kernels.hip / ba_solve.hip)
constexpr int GKEY_MAGIC = 0x6b657935;   // first word of a valid graph key
// check != 0: leave at once when the workspace's key says the tables are those of this very graph
// eta_rows > 1: must equal |kx| (droid_kernels.cu:1476 broadcasts eta over the rows of C); a mismatch is reported through
// `status` (pinned host memory, may be null): [0] = 1, [1] = eta_rows, [2] = |kx|
// band_verdict (pinned host memory, may be null): when the tables are rebuilt, 1 / 2 = the window solver will / will not admit
// the reduced system of this graph (its admission test on the new skyline; max_nt as launch_ba_solve_wave passes it)
__global__ void ba_prepare_kernel(const int64_t *ii, const int64_t *jj, int N, int B, int t0, int t1, int scan_ints,
                                  int ftable, int check, int eta_rows, int *status, BaTables T, int *band_verdict, int max_nt);
// which Schur kernel a window gets (ba_host.hip): the per-source-frame form on windows whose frames couple many rows
bool ba_schur_frame_form(int N, int P);
template <int PPL, bool MF, int EW>
__global__ void ba_linearize_kernel(const float *poses, const float *disps, const float *intrinsics,
                                    const float *disps_sens, const float *targets, const float *weights,
                                    const float *eta, int eta_rows, const int64_t *jj,
                                    const uint8_t *frame_owned, int N, int HW, int wd, int t0, int P,
                                    float alpha, int upd, float *poses_out, float *disps_w, BaTables T, BaBuffers W);
// lower (here and below): bit 0 = only the lower triangle of H is kept up (what the solvers read), bit 1 = deterministic
// accumulation (64-bit fixed point, see acc_add in ba_kernels.hip)
__global__ void ba_assemble_kernel(const int64_t *ii, const int64_t *jj, const uint8_t *frame_owned, int N,
                                   int t0, int P, int lower, BaTables T, BaBuffers W);
__global__ void ba_schur_kernel(const int64_t *ii, const int64_t *jj, const uint8_t *frame_owned, int N, int HW,
                                int t0, int P, int lower, BaTables T, BaBuffers W);
__global__ void ba_symmetrize_kernel(double *H, int n);
struct ExportArg {
  double A[36];   // row-major 6 x 6 (BA2GTSAM's block, depth_video.py:21-23)
};
__global__ void ba_export_kernel(const double *H, const double *b, int n, double *out, int gtsam, ExportArg arg, double stab,
                                 unsigned *counter, int *host_flag, int seq);
__global__ void ba_fixed_to_f64_kernel(double *H, double *b, int n);
constexpr int GRAM_LIST_CAP = 1024;  // edges (+ 1) up to which the prepare kernel builds the frame row table (one thread per edge)
template <bool VEC, bool F32>
__global__ void ba_schur_gram_kernel(const int64_t *ii, const int64_t *jj, const uint8_t *frame_owned, int N, int HW,
                                     int t0, int P, int nch, int lower, BaTables T, BaBuffers W);
__global__ void ba_update_kernel(float *poses, const float *poses_src, float *disps, const int64_t *jj, const uint8_t *frame_owned,
                                 int HW, int t0, int P, int update_poses, int update_disps, float *dz_out,
                                 float *dx_out, BaTables T, BaBuffers W, float disp_floor);
__global__ void ba_copy_dx_kernel(const double *src, float *dst, int n);
__global__ void ba_copy_f32_kernel(const float *src, float *dst, int n);
__global__ void ba_gather_edges_kernel(const float2 *tgt_inac, const float2 *wgt_inac, const int64_t *ii_inac, const int64_t *jj_inac,
                                       const int64_t *sel, int n_sel, int n_inac, const float2 *tgt_act, const float2 *wgt_act,
                                       const int64_t *ii_act, const int64_t *jj_act, int HW, float *tgt_out, float *wgt_out,
                                       int64_t *ii_out, int64_t *jj_out);

// The Gauss-Newton loop of dba_ba (ba_host.hip), shared with the sharded sequence (ba_sharded_host.hip): frame_owned restricts
// stages 1 / 2 / 4 to the rank's source frames, window_fpose is the COMPLETE graph's pose-level skyline for the redundant solve,
// and `exchange` (may be null) runs between the reduction and the solve of every iteration on the summed-to-be [H | gap | b]
// range of the workspace.  The back-substitution + retraction of iteration k ride in the linearisation of iteration k + 1 in
// both uses (a rank only moves the depths of frames it owns; the poses' update is redundant on every rank).
struct BaExchange {
  int (*fn)(void *ctx, double *hb, size_t hb_len, hipStream_t stream);
  void *ctx;
};
int ba_run_loop(float *poses, float *disps, const float *intrinsics, const float *disps_sens, const float *targets,
                const float *weights, const float *eta, int eta_rows, const int64_t *ii, const int64_t *jj,
                const uint8_t *frame_owned, int N, int B, int ht, int wd, int t0, int t1, int iterations, float lm, float ep,
                float alpha, int motion_only, float *dx_out, float *dz_out, void *ws, size_t ws_bytes, dba_stream_t stream,
                int prepared, int solver_hint, float disp_floor, const int32_t *window_fpose, const BaExchange *exchange);

// damped float64 Cholesky solve of H x = b, one workgroup
// fpose: optional [n/6] skyline of the system at pose granularity (see BaTables); null = measure it from H
// splan: optional 8 ints of the workspace (meta + 16) where the window solver keeps what it derived from fpose -- window height, the
// cut for two fronts -- between the solves of one graph; stage 0 clears it when it rebuilds the tables; null = derive every time
int launch_ba_solve(const double *H, const double *b, const int *fpose, int n, double lm, double ep, float *dx, int *meta,
                    double *Lscratch, hipStream_t stream, long long *prof = nullptr, int hint = 0, int *splan = nullptr);
bool ba_solve_fits_lds(int n);
bool ba_solve_tile_supported(int n);
bool ba_solve_band_supported(int n);
int launch_ba_solve_band(const double *H, const double *b, const int *fpose, int n, double lm, double ep, float *dx,
                         int *meta, double *scratch, size_t scratch_doubles, bool big, hipStream_t stream,
                         bool last = false);  // last: nothing is queued behind this launch (a hand-shake that times out fails the solve)
int launch_ba_solve_tile(const double *H, const double *b, int n, double lm, double ep, float *dx, int *meta,
                         hipStream_t stream);
// five-wave window solver for banded systems (ba_solve_wave.hip); needs the pose-level skyline table; a system it does not
// admit is solved by the general kernel's code in the same launch, *verdict (pinned host memory, may be null) = 1 admitted / 2 not
bool ba_solve_wave_supported(int n);
int ba_solve_wave_max_nt(int n);   // the tallest window (3 | 4 tile rows) whose panel store fits LDS for n unknowns, 0: none
// the word of pinned host memory that carries "banded (1) / not banded (2)" for the workspace whose meta block this is
// (launch_ba_solve reads it without synchronising; the window kernel and, on a graph change, stage 0 write it); null: no pool
int *solver_verdict_slot(const int *meta);
// (the workspace's pinned words, ba_solve.hip: [0] verdict, [1] probe counter, [4..6] stage 0's eta report)
void ws_words_reset(const int *meta);
int *ws_eta_status(const int *meta);         // ... [4..6]: stage 0's eta report
int ws_poll_eta(const int *meta, int *eta_rows, int *num_kx);
int launch_ba_solve_wave(const double *H, const double *b, const int *fpose, int n, double lm, double ep, float *dx, int *meta,
                         double *Lscratch, int *verdict, hipStream_t stream, long long *prof = nullptr, int *splan = nullptr);
size_t ba_solve_scratch_doubles(int n);
constexpr int SOLVE_MAX_LDS_BYTES = 160 * 1024;

}  // namespace dba
