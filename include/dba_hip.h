/* dba_hip.h -- C ABI of the MI355X (gfx950) dense-bundle-adjustment / correlation backend.
 *
 * This is the drop-in boundary of the hot path: plain pointers, sizes and a HIP stream; no
 * torch types.  Every entry point names the reference interface it replaces
 * (paths relative to GREAT-WHU/DBA-Fusion, i.e. /root/reference).  The reference binds these
 * operations through the pybind11 module `droid_backends` (src/droid.cpp:297-316); the module of
 * the same name under dba-fusion_amd/droid_backends/ is a thin adapter over this ABI
 * (see INTEGRATION.md for the binding a maintainer would add).
 *
 * Conventions
 *   - all data pointers are DEVICE pointers unless the parameter name ends in _host;
 *   - tensors are dense row-major ("contiguous"), shapes given in comments;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised,
 *     unless stated otherwise; results are visible to later work on the same stream;
 *   - return value: 0 = DBA_OK, negative = error (see enum); numerical failure of the pose
 *     solve is NOT an error: like the reference it yields a zero update
 *     (src/droid_kernels.cu:193-196, :1263-1266).
 */
#ifndef DBA_HIP_H
#define DBA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  DBA_OK = 0,
  DBA_ERR_ARG = -1,       /* bad size / null pointer */
  DBA_ERR_WORKSPACE = -2, /* workspace too small */
  DBA_ERR_HIP = -3,       /* a HIP runtime call failed (see dba_last_error) */
  DBA_ERR_UNSUPPORTED = -4
};

enum { DBA_F32 = 0, DBA_F16 = 1, DBA_F64 = 2 }; /* element type selector for volume/corr buffers */

typedef void *dba_stream_t;

const char *dba_version(void);
const char *dba_last_error(void); /* text of the last DBA_ERR_HIP on this thread */

/* ------------------------------------------------------------------------------------------
 * Dense bundle adjustment: droid_backends.ba  (src/droid.cpp:109-138 -> ba_cuda,
 * src/droid_kernels.cu:1394-1512) and class BACore (src/bacore.h:4-70,
 * src/droid_kernels.cu:1786-1956).
 *
 *   poses       [B,7]   f32 (tx,ty,tz,qx,qy,qz,qw) world->camera; rows [t0,t1) updated in place
 *   disps       [B,ht,wd] f32 inverse depth; rows kx = unique(arange(t0,t1) U ii) updated in place
 *   intrinsics  [4]     f32 (fx,fy,cx,cy) at 1/8 resolution
 *   disps_sens  [B,ht,wd] f32, 0 = no depth measurement
 *   targets, weights [N,2,ht,wd] f32 (ch0 = x/u, ch1 = y/v)
 *   eta         [eta_rows,ht,wd] f32, eta_rows == |kx| or 1 (broadcast)
 *   ii, jj      [N] int64 on the device
 *   frame_owned [B] uint8 or NULL: multi-GPU edge sharding by source frame -- a rank linearises
 *               the edges it was given and updates depth only for frames it owns (NULL = all)
 *
 * The workspace (device memory, dba_ba_workspace_bytes) holds the index tables, the
 * depth/pose coupling rows E, Q = 1/C, w, the reduced camera system (float64) and dx.
 * ---------------------------------------------------------------------------------------- */

size_t dba_ba_workspace_bytes(int N, int B, int ht, int wd, int t0, int t1);

/* byte offsets inside the workspace of the pieces a host may need to touch between stages
 * (multi-GPU all-reduce of H/b; BACore handing H/v to the caller).  H is [6P,6P] float64 row-major,
 * b is [6P] float64, dx is [P,6] float32, meta[0] = |kx| (int32), meta[1] = 1 if the last solve failed. */
typedef struct {
  size_t H, b, dx, meta, E, Q, w, kx;
  int P, Mmax, nchunks;
} dba_ba_layout;
int dba_ba_get_layout(int N, int B, int ht, int wd, int t0, int t1, dba_ba_layout *out);

/* stage 0: index sets on the device (replaces the per-iteration host bookkeeping of
 * ba_cuda :1416-1424, accum_cuda :993-1043 and schur_block :1307-1347). */
int dba_ba_prepare(const int64_t *ii, const int64_t *jj, int N, int B, int ht, int wd, int t0, int t1,
                   void *ws, size_t ws_bytes, dba_stream_t stream);
/* Stage 0 keyed on the CONTENTS of the edge list.  The reference's caller hands droid_backends.ba new ii / jj tensors with
 * the same edges on every update (torch.cat with the inactive edges, dbaf/covisible_graph.py:242-247), so "same graph as
 * last time" cannot be decided on tensor identity.  Stage 0 leaves a key in the workspace (N, B, t0, t1, Schur form, ii, jj);
 * check != 0: the launch compares the call's edge list with that key on the device and returns at once when the tables
 * are those of this graph already (no host synchronisation either way).  check != 0 needs a workspace whose key area is
 * valid: one that went through dba_ba_workspace_init (fresh memory) or an earlier stage 0.
 * eta_rows > 1: the number of rows of the call's eta, which must equal |kx| (droid_kernels.cu:1476 adds eta.view(-1, HW) to
 * the |kx| rows of C; the reference raises a broadcast error otherwise).  |kx| only exists on the device, so a mismatch is
 * recorded in pinned host memory and reported by dba_ba_poll_eta_error[_ws]; the call it belongs to changes nothing. */
int dba_ba_prepare_keyed(const int64_t *ii, const int64_t *jj, int N, int B, int ht, int wd, int t0, int t1, int eta_rows,
                         int check, void *ws, size_t ws_bytes, dba_stream_t stream);
/* diagnostic: what the library currently believes about the reduced camera system of this workspace's graph -- 1: banded enough
 * for the window solver (csrc/ba_solve_wave.hip), 2: not (the register-tile / skyline / general kernels serve it), 0: nothing
 * known yet; negative: error.  Host memory only, no synchronisation: the value is written by stage 0 when it rebuilds a graph's
 * tables inside dba_ba / dba_ba_run and by the window solver itself, and steers where the NEXT solves of the workspace are sent
 * (only ever a hint: every kernel solves whatever it is given).  Replaces the host-side choice of Eigen::LLT / SimplicialLLT
 * by matrix size in /root/reference/src/droid_kernels.cu:200-218,1248-1269. */
int dba_ba_solver_verdict(int N, int B, int ht, int wd, int t0, int t1, void *ws, size_t ws_bytes);
/* marks a freshly allocated workspace as "no graph prepared" (clears meta and the key header; asynchronous on `stream`) */
int dba_ba_workspace_init(int N, int B, int ht, int wd, int t0, int t1, void *ws, size_t ws_bytes, dba_stream_t stream);
/* 1 (and the two counts) if a stage 0 that has COMPLETED since the last poll saw eta_rows != |kx| on any workspace, else 0; clears
 * that record.
 * Host memory only: no synchronisation.  The adapters poll at the top of every ba call and raise for the earlier one. */
int dba_ba_poll_eta_error(int *eta_rows, int *num_kx);
/* the same for ONE workspace (the report lives in pinned words of the workspace stage 0 ran on, so a mismatch is attributed to the
 * caller that made it whatever other devices, streams or threads do).  The offending call itself is harmless: stage 0 also leaves
 * its verdict in the workspace, and the kernels that write the caller's state honour it -- poses and inverse depths stay as they
 * were, dx and dz come back zero (the reference raises before it touches anything: /root/reference/src/droid_kernels.cu:1476). */
int dba_ba_poll_eta_error_ws(int N, int B, int ht, int wd, int t0, int t1, void *ws, size_t ws_bytes, int *eta_rows, int *num_kx);

/* The edge tensors of one BA call as the reference's caller assembles them (dbaf/covisible_graph.py:242-247: torch.cat of the
 * selected inactive edges' and the active edges' ii / jj / target / weight; :332-333: target, weight from [n, ht, wd, 2] to the
 * planar [n, 2, ht, wd] of the binding droid.cpp:301) in one launch instead of ten.  target_* / weight_*: [n_*, ht, wd, 2]
 * float32; sel: n_sel indices into the inactive list (int64, device; null: its first n_sel edges); outputs: targets_out,
 * weights_out [n_sel + n_act, 2, ht, wd], ii_out, jj_out [n_sel + n_act] -- the arguments of dba_ba / droid_backends.ba. */
int dba_ba_gather_edges(const float *target_inac, const float *weight_inac, const int64_t *ii_inac, const int64_t *jj_inac,
                        int n_inac, const int64_t *sel, int n_sel, const float *target_act, const float *weight_act,
                        const int64_t *ii_act, const int64_t *jj_act, int n_act, int ht, int wd, float *targets_out,
                        float *weights_out, int64_t *ii_out, int64_t *jj_out, dba_stream_t stream);

/* stage 1: fused per-source-frame linearisation (projective_transform_kernel :220-468 +
 * accum_kernel :899-919 + C/w/Q assembly :1474-1478); also clears H, b.  alpha = 0.05 for
 * droid_backends.ba (:1474), 0.001 for BACore::hessian (:1872). */
int dba_ba_linearize(const float *poses, const float *disps, const float *intrinsics,
                     const float *disps_sens, const float *targets, const float *weights,
                     const float *eta, int eta_rows, const int64_t *ii, const int64_t *jj,
                     const uint8_t *frame_owned, int N, int B, int ht, int wd, int t0, int t1,
                     float alpha, void *ws, size_t ws_bytes, dba_stream_t stream);

/* stage 2: reduced camera system  H = A - E Q E^T,  b = v - E Q w  in float64
 * (SparseBlock::update_lhs/rhs :1176-1218, schur_block + EEt6x6/Ev6x1 :1046-1138, :1297-1391).
 * motion_only != 0 assembles A, v alone (:1464-1471). */
int dba_ba_reduce(const int64_t *ii, const int64_t *jj, const uint8_t *frame_owned, int N, int B,
                  int ht, int wd, int t0, int t1, int motion_only, void *ws, size_t ws_bytes,
                  dba_stream_t stream);

/* Which kernel forms the Schur products of stage 2 (must be set before dba_ba_prepare of the call it applies to: the
 * prepare stage builds the tables of the form in force): 0 = automatic (the per-source-frame form -- float64 Gram tiles on
 * the matrix cores, every row of E read once -- on windows whose frames couple many rows, the (row, partner) grid on
 * sparse ones), 1 = (row, partner) grid, 2 = per-source-frame form.  Initialised from DBA_SCHUR_KERNEL = rows | frame. */
int dba_ba_schur_select(int form);
int dba_ba_schur_select_thread(int form); /* the same choice for the calling host thread only (0 = none); the process-wide
                                           * dba_ba_schur_select wins when both are set */
int dba_ba_schur_thread_form(void);        /* the calling thread's pin (0 = none): part of the key of a prepared workspace */
int dba_ba_schur_auto_form(int N, int P); /* the form (1 or 2) the automatic choice gives a graph of N edges over a window of
                                           * P optimised poses (rows per source frame ~ 1 + N / (P + 1); a function of the
                                           * graph, not of the video buffer's size): the sharded driver asks with the
                                           * COMPLETE graph's numbers and pins that form on every rank for the duration of
                                           * the call (a rank's share has the same rows per frame, but few edges) */
int dba_ba_schur_generation(void); /* number of dba_ba_schur_select calls so far: tables prepared under another generation
                                    * may lack what the form in force needs (callers of dba_ba_prepared compare it) */

/* Deterministic accumulation of H, b (off by default; DBA_DETERMINISTIC=1 turns it on at load).  The reference adds the
 * blocks of the camera system on the host in a fixed order (SparseBlock::update_lhs / update_rhs, droid_kernels.cu:1176-1218);
 * this path adds them with float64 atomics from many workgroups, which is order-dependent in the last bits.  With the mode
 * on every addend is rounded once to a multiple of 2^-30 and summed with 64-bit integer atomics (associative): H, b, and
 * with them dx and the retracted state, are identical bit for bit from run to run and between a single GPU and any number
 * of ranks.  Costs one small launch per Gauss-Newton iteration. */
int dba_ba_set_deterministic(int on);

/* opt-in guard (also DBA_SOLVE_CHECK=1): behind every solve a kernel checks the residual of the damped system at the float
 * solution; a solve that is wrong beyond rounding becomes a zero update with the failure flag set, as a failed factorisation
 * does (/root/reference/src/droid_kernels.cu:1263-1266).  ~5 us per solve; off by default: the solvers' flag protocols are covered
 * by the cold-start stress of the GPU suite (tests/test_gpu_solve_cold.py). */
int dba_ba_set_solve_check(int on);
/* the check alone, on what lies in the workspace (H lower triangle, b, dx) */
int dba_ba_solve_check(int N, int B, int ht, int wd, int t0, int t1, float lm, float ep, void *ws, size_t ws_bytes,
                       dba_stream_t stream);

/* H <- its lower triangle mirrored.  dba_ba and the sharded front stage keep only the lower triangle of H up (what the
 * solvers read: half the float64 atomics); a caller that hands the full matrix on (ShardedBACore.hessian -> GTSAM) mirrors it
 * first.  dba_ba_reduce and dba_bacore_hessian always produce the full matrix (mirrored from the lower triangle: symmetric
 * to the last bit). */
int dba_ba_symmetrize(int N, int B, int ht, int wd, int t0, int t1, void *ws, size_t ws_bytes, dba_stream_t stream);

/* stage 3: damped dense solve in float64 on the device
 * (SparseBlock::solve :1248-1269: diag += ep + lm*diag; LL^T; zeros on failure) -> dx.  With a skyline table: the five-wave
 * window kernel for banded systems up to 64 poses (csrc/ba_solve_wave.hip).  Otherwise / for other structures: register-tile
 * block LDL^T up to 29 poses, skyline variants up to 64 poses, blocked Cholesky beyond / for wide skylines (csrc/ba_solve*.hip).
 * This stage measures the skyline from H (the system may have been summed over ranks); dba_ba takes it from the
 * prepare stage's graph tables. */
int dba_ba_solve(int N, int B, int ht, int wd, int t0, int t1, float lm, float ep, void *ws,
                 size_t ws_bytes, dba_stream_t stream);
/* the same with the caller's pose-level skyline (device, t1 - t0 ints: for every pose the first pose it is coupled with, as
 * dba_ba_shard_back's window_fpose): what dba_ba hands the solvers from its graph tables.  With a skyline, banded systems
 * (every column inside the 48-row window of its 16-column tile: ~4 poses wide) go to the five-wave window kernel
 * (csrc/ba_solve_wave.hip); a system that turns out not to be banded is solved by the general kernel's code in the same
 * launch, and the next solve on this workspace goes to the register-tile / skyline kernels directly. */
int dba_ba_solve_skyline(int N, int B, int ht, int wd, int t0, int t1, float lm, float ep, const int32_t *fpose, void *ws,
                         size_t ws_bytes, dba_stream_t stream);

/* stage 4: back-substitution + retraction (EvT6x1_kernel :1140-1160, dz :1495,
 * pose_retr_kernel :943-976, disp_retr_kernel :978-991).  dz_out [>=|kx|, ht*wd] may be NULL.
 * update_poses / update_disps select which retractions are applied. */
int dba_ba_update(float *poses, float *disps, const int64_t *ii, const int64_t *jj,
                  const uint8_t *frame_owned, int N, int B, int ht, int wd, int t0, int t1,
                  int update_poses, int update_disps, float *dz_out, void *ws, size_t ws_bytes,
                  dba_stream_t stream);

/* The two halves of one Gauss-Newton iteration of a rank of the edge-sharded driver (dbaf_amd/sharded.py), one call
 * each: front = stages 1 + 2 on the rank's edges (its partial [H | b] is then summed over the ranks by the caller: one
 * RCCL all-reduce), back = stages 3 + 4 on the summed system (every rank solves it redundantly, retracts all poses and
 * back-substitutes the depths of the frames it owns). */
int dba_ba_shard_front(const float *poses, const float *disps, const float *intrinsics, const float *disps_sens,
                       const float *targets, const float *weights, const float *eta, int eta_rows, const int64_t *ii,
                       const int64_t *jj, const uint8_t *frame_owned, int N, int B, int ht, int wd, int t0, int t1,
                       float alpha, int motion_only, void *ws, size_t ws_bytes, dba_stream_t stream);
int dba_ba_shard_back(float *poses, float *disps, const int64_t *ii, const int64_t *jj, const uint8_t *frame_owned,
                      int N, int B, int ht, int wd, int t0, int t1, float lm, float ep, int update_disps,
                      const int32_t *window_fpose, int solver_hint, void *ws, size_t ws_bytes, dba_stream_t stream);
/* window_fpose (device, t1 - t0 ints, or NULL): for every pose of the window the first pose it is coupled to in the
 * COMPLETE graph (all ranks' edges: through an edge, or through a common source frame), which lets the solver take the
 * skyline of the summed system from the graph instead of measuring it (dbaf_amd/sharded.py computes it on the host;
 * a table that names a later pose than the true one gives a wrong solve, NULL is always safe).
 * solver_hint: as for dba_ba_prepared (0 = none; 1 = meta[7] read 1 after an earlier solve of this window's summed system:
 * the several-tiles-per-thread skyline variant need not be queued behind the first one). */

/* ---- the sharded BA of one rank as ONE enqueued sequence (csrc/ba_sharded_host.hip) --------------------------------------
 * RCCL called from this library, on the caller's stream.  librccl is loaded with dlopen at the first dba_comm_* call (the copy
 * the process already has -- PyTorch ships one -- else /opt/rocm/lib); DBA_ERR_UNSUPPORTED when there is none.  The unique id
 * (128 bytes, ncclUniqueId) is made by one rank and handed to the others by the caller (dbaf_amd/sharded.py broadcasts it
 * through torch.distributed once per process group); dba_comm_create is collective and binds the calling thread's device. */
typedef struct dba_comm dba_comm;
int dba_comm_unique_id(void *id128);
int dba_comm_create(const void *id128, int world, int rank, dba_comm **out);
int dba_comm_destroy(dba_comm *c);
/* RCCL's own view of the communicator: ranks it spans, this process's rank (ncclCommCount / ncclCommUserRank) */
int dba_comm_info(dba_comm *c, int *world, int *rank);
int dba_comm_allreduce_f64(dba_comm *c, double *buf, size_t count, dba_stream_t stream);   /* in-place sum */

/* who carries what between the ranks in dba_ba_sharded_run.  world = 1: nothing is exchanged (the other fields are ignored).
 *   the reduced system [H | b] (float64, once per iteration): `comm` (RCCL all-reduce) or `peer_regions` (one-shot peer-read
 *     kernel, dba_peer_allreduce_f64: regions of all ranks, *peer_epoch is advanced by the call, peer_status as there);
 *     band_len > 0: only the entries band_idx[0..band_len) of the range travel (the skyline band of large windows, gathered
 *     into band_buf and scattered back);
 *   the inverse depths (float32, once per call, `comm` only -- a caller on the peer-read exchange gathers them itself):
 *     my_rows[n_mine] = the frames this rank owns, kmax = the largest n_mine over the ranks, send [kmax, ht*wd] /
 *     recv [world*kmax, ht*wd] staging, and for every owned row of every rank its frame all_rows[i] and its place
 *     all_slots[i] in recv. */
typedef struct {
  int world, rank;
  dba_comm *comm;
  void *const *peer_regions;
  unsigned *peer_epoch;
  size_t peer_max_doubles;
  int *peer_status;
  const int64_t *band_idx;
  size_t band_len;
  double *band_buf;
  const int64_t *my_rows;
  int n_mine, kmax;
  const int64_t *all_rows, *all_slots;
  int n_all;
  float *send, *recv;
} dba_shard_exchange;

/* stage 0 (prepared = 0 | 1 | 2 as in dba_ba_run), then dba_ba's own Gauss-Newton loop on the rank's edges -- `iterations` x
 * { linearisation (which carries the previous iteration's back-substitution + retraction, as in dba_ba: a rank moves the
 * depths of the frames it owns, the poses' update is redundant on every rank), Schur reduction, sum of [H | b] over the
 * ranks, solve with the complete graph's skyline }, the last update --, then the all-gather of the owned depth maps:
 * everything enqueued on `stream`, nothing waits on the host.  The same states as the staged sequence
 * dba_ba_shard_front / sum / dba_ba_shard_back, bit for bit.  With world = 1 nothing is summed.  The alignment gap between H
 * and b in the workspace must be zero (it is summed along). */
int dba_ba_sharded_run(float *poses, float *disps, const float *intrinsics, const float *disps_sens, const float *targets,
                       const float *weights, const float *eta, int eta_rows, const int64_t *ii, const int64_t *jj,
                       const uint8_t *frame_owned, int N, int B, int ht, int wd, int t0, int t1, int iterations, float lm,
                       float ep, float alpha, int motion_only, const int32_t *window_fpose, int solver_hint, int prepared,
                       const dba_shard_exchange *x, void *ws, size_t ws_bytes, dba_stream_t stream);

/* droid_backends.ba: `iterations` x (stage 1..4), all enqueued on `stream` with no host sync.
 * dx_out [P,6] and dz_out [>=|kx|, ht*wd] receive the last iteration's update (either may be NULL). */
int dba_ba(float *poses, float *disps, const float *intrinsics, const float *disps_sens,
           const float *targets, const float *weights, const float *eta, int eta_rows,
           const int64_t *ii, const int64_t *jj, int N, int B, int ht, int wd, int t0, int t1,
           int iterations, float lm, float ep, int motion_only, float *dx_out, float *dz_out,
           void *ws, size_t ws_bytes, dba_stream_t stream);

/* The same with stage 0 skipped: the index tables in `ws` must be those of THIS graph already, i.e. the last dba_ba /
 * dba_ba_prepare on this workspace had the same ii, jj (contents), N, B, ht, wd, t0, t1 and Schur kernel form.  For callers
 * that run several updates on one covisibility graph (CovisibleGraph.update does, dbaf/covisible_graph.py:214-342: the
 * adapter keeps one workspace per window shape and recognises an unchanged edge list, droid_backends/__init__.py).
 * Back-substitution + retraction of every iteration but the last are folded into the next iteration's linearisation in
 * both (iterations launches fewer; DBA_BA_FUSE_UPDATE=0 keeps them apart). */
int dba_ba_prepared(float *poses, float *disps, const float *intrinsics, const float *disps_sens,
                    const float *targets, const float *weights, const float *eta, int eta_rows,
                    const int64_t *ii, const int64_t *jj, int N, int B, int ht, int wd, int t0, int t1,
                    int iterations, float lm, float ep, int motion_only, float *dx_out, float *dz_out,
                    void *ws, size_t ws_bytes, dba_stream_t stream, int solver_hint);
/* solver_hint: 0 = none; 1 = meta[7] (int32 at layout.meta + 28) read 1 after an earlier call on this graph, i.e. the
 * one-tile skyline solver took its structure: the fall-back kernels behind it are then not queued (windows of 30-64
 * poses; whether it fits depends on the graph alone, a failing pivot is not a reason to fall back.  The only other thing
 * the queue was a net for, a partner workgroup more than a second late at the hand-shake, then gives a failed solve, i.e. a
 * zero update, instead of a slower correct one). */

/* dba_ba (prepared = 0) / dba_ba_prepared (prepared = 1) / stage 0 decided on the device by the graph key (prepared = 2, see
 * dba_ba_prepare_keyed; solver_hint is ignored) with the caller's next statement taken along: DepthVideo.ba clamps the inverse
 * depths right after droid_backends.ba returns (`self.disps.clamp_(min=0.001)`, dbaf/depth_video.py:560), a launch of its
 * own over the whole buffer.  disp_floor > 0: the LAST launch of the call writes max(d, disp_floor) (torch.clamp's select: a
 * NaN stays a NaN) for the frames it updates and applies the same floor to every other frame of the buffer (also when
 * motion_only leaves the depths alone): the same state, bit for bit, as the call followed by the caller's clamp over the
 * whole buffer -- the reference's caller rescales inverse depths between BA calls (dbaf_frontend.py:570,814), so frames
 * outside kx can be below the floor as well.  Earlier iterations and the
 * returned dz are untouched, exactly as when the clamp follows the call.  disp_floor = 0: dba_ba / dba_ba_prepared. */
int dba_ba_run(float *poses, float *disps, const float *intrinsics, const float *disps_sens,
               const float *targets, const float *weights, const float *eta, int eta_rows,
               const int64_t *ii, const int64_t *jj, int N, int B, int ht, int wd, int t0, int t1,
               int iterations, float lm, float ep, int motion_only, float *dx_out, float *dz_out,
               void *ws, size_t ws_bytes, dba_stream_t stream, int prepared, int solver_hint, float disp_floor);

/* dba_bacore_hessian with stage 0 as in dba_ba_run (prepared = 0 | 1 | 2): BACore.hessian runs twice per update on one
 * graph (dbaf/depth_video.py:527), the second call finds the tables in place */
int dba_bacore_hessian_run(const float *poses, const float *disps, const float *intrinsics,
                           const float *disps_sens, const float *targets, const float *weights,
                           const float *eta, int eta_rows, const int64_t *ii, const int64_t *jj, int N,
                           int B, int ht, int wd, int t0, int t1, double *H_host, double *v_host,
                           void *ws, size_t ws_bytes, dba_stream_t stream, int prepared);

/* BACore::hessian with the reduced system handed over in PINNED HOST MEMORY OF THE LIBRARY (one block per workspace, valid until
 * the next hessian call on that workspace): *out_host.  The last kernel of the sequence writes the system there itself -- mirrored
 * from the lower triangle the reduction keeps up -- and sets a completion word the call spins on: no copy engine, no stream
 * synchronisation (round 5: two hipMemcpyAsync + hipStreamSynchronize, 94 us per call on the 25-KF window).
 * layout 0: H [6P, 6P] row-major followed by v [6P] (what src/droid_kernels.cu:1889-1897 copies out).
 * layout 1: the system in the factor-graph side's tangent coordinates, as the augmented matrix [Hg | vg] of shape [6P, 6P + 1]
 *   that the GTSAM fork's BA2GTSAM returns (/root/reference/dbaf/depth_video.py:20-29, :397-401, :527-529):
 *   Hg = J^T (H + stabilizer on the first pose's diagonal, :397) J, vg = J^T v, J = blockdiag(A36), A36 the row-major 6 x 6 block
 *   -Ad(Tbc^-1) with its row halves swapped (:21-23).  A36 is ignored for layout 0. */
int dba_bacore_hessian_host(const float *poses, const float *disps, const float *intrinsics,
                            const float *disps_sens, const float *targets, const float *weights,
                            const float *eta, int eta_rows, const int64_t *ii, const int64_t *jj, int N,
                            int B, int ht, int wd, int t0, int t1, void *ws, size_t ws_bytes, dba_stream_t stream,
                            int prepared, int layout, const double *A36, double stabilizer, double **out_host);

/* where this workspace's pinned block is (what the last hessian / export call on it filled); error if there is none yet */
int dba_bacore_staging(int N, int B, int ht, int wd, int t0, int t1, void *ws, size_t ws_bytes, double **out_host);

/* the last step of dba_bacore_hessian_host alone: the reduced system that lies in the workspace (H lower triangle, b: after
 * dba_ba_reduce, or after a sum over ranks) -> the workspace's pinned block, layout as above; waits for completion. */
int dba_bacore_export_host(int N, int B, int ht, int wd, int t0, int t1, void *ws, size_t ws_bytes, dba_stream_t stream, int layout,
                           const double *A36, double stabilizer, double **out_host);

/* BACore::hessian: stages 1-2 with alpha = 0.001, then copies H [6P,6P], v [6P] (float64) to HOST
 * memory (the caller owns CPU tensors, src/droid_kernels.cu:1889-1897): dba_bacore_hessian_host + two host memcpys; complete
 * on return. */
int dba_bacore_hessian(const float *poses, const float *disps, const float *intrinsics,
                       const float *disps_sens, const float *targets, const float *weights,
                       const float *eta, int eta_rows, const int64_t *ii, const int64_t *jj, int N,
                       int B, int ht, int wd, int t0, int t1, double *H_host, double *v_host, void *ws,
                       size_t ws_bytes, dba_stream_t stream);

/* BACore::retract: dx_host [6P] float64 (HOST) -> f32 on device, then stage 4 using the E, Q, w
 * cached by the last dba_bacore_hessian on the same workspace.  Up to 64 poses the update travels as a kernel argument:
 * dx_host is consumed before the call returns and nothing synchronises the stream. */
int dba_bacore_retract(float *poses, float *disps, const int64_t *ii, const int64_t *jj, int N, int B,
                       int ht, int wd, int t0, int t1, const double *dx_host, float *dx_out,
                       float *dz_out, void *ws, size_t ws_bytes, dba_stream_t stream);

/* BACore::optimize: damped dense solve of a caller-supplied HOST system (solveDenseD :200-218);
 * result kept in the workspace dx slot and copied to dx_out (device, may be NULL). */
int dba_bacore_optimize(const double *H_host, const double *v_host, int N, int B, int ht, int wd,
                        int t0, int t1, float lm, float ep, float *dx_out, void *ws, size_t ws_bytes,
                        dba_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Correlation volume: CorrBlock (dbaf/modules/corr.py:24-71) and its lookup
 * droid_backends.corr_index_forward (src/droid.cpp:231-239, src/correlation_kernels.cu:19-70,126-155).
 * ---------------------------------------------------------------------------------------- */

/* corr_index_forward: volume [n,h1,w1,h2,w2] (dtype), coords [n,2,h1,w1] f32 (ch0 = x, ch1 = y)
 * -> corr [n,2r+1,2r+1,h1,w1] (dtype), channel = x_offset*(2r+1) + y_offset.  Bit-exact with the
 * reference arithmetic (four c10::Half read-modify-writes per tap in a fixed order). */
int dba_corr_index_forward(const void *volume, const float *coords, void *corr, int n, int h1, int w1,
                           int h2, int w2, int radius, int dtype, dba_stream_t stream);

/* fused CorrBlock.__call__ (corr.py:40-50): all pyramid levels in one launch, reading
 * coords [n,h1,w1,2] as produced by projective_transform and writing the concatenated
 * [n, L*(2r+1)^2, h1, w1] tensor directly (no per-level tensors, no torch.cat).
 * level l volume: [n,h1,w1,h2>>l,w2>>l]; coords are divided by 2^l per level. */
int dba_corr_lookup_pyramid(const void *const *volumes /* host array of L device ptrs */,
                            const float *coords_nhw2, void *corr, int n, int h1, int w1, int h2,
                            int w2, int num_levels, int radius, int dtype, dba_stream_t stream);

/* Flow-aligned ("sheared") volume: Vs_l[n][dy][dx][pixel] = V_l[n][y1][x1][ty][tx] with pixel = y1 * w1 + x1,
 * dy = (ty - (y1 >> l)) mod h2l, dx = (tx - (x1 >> l)) mod w2l; the pixel axis is padded to
 * dba_corr_sheared_plane_elems(h1, w1) = h1 * w1 rounded up to a multiple of 64, or the size of the tile grid for tiled planes
 * (dba_corr_sheared_grid; padding is never read for a real pixel) -- the size of the reference tensor of dbaf/modules/corr.py:31-36 (+ padding), but the taps that
 * neighbouring source pixels read become contiguous, so a wave fetches full, aligned 128-byte lines for ANY map
 * size (csrc/corr_sheared.hip).  Used by the CorrBlock mirror; the lookup result is bit-identical to
 * corr_index_forward on the reference layout.  f16 only, radius 3. */
int dba_corr_sheared_plane_elems(int h1, int w1);
/* Order of the pixel axis.  Maps with h1 % 4 == 0 and w1 % 16 == 0 keep their source pixels in 4 x 16 TILES,
 * pixel = ((y1 >> 2) * (w1 >> 4) + (x1 >> 4)) * 64 + (y1 & 3) * 16 + (x1 & 15), so that the 64 pixels of a 128-byte line are
 * neighbours in BOTH directions: the union of their windows -- what a lookup wave reads -- is 79 instead of 92 lines per wave
 * and level on the bench scene, and the lookup's time is proportional to that number (profiles/r04_lookup_lines.txt).  Other
 * maps: pixel = y1 * w1 + x1.  Returns the tile width (16) for the tiled order, 0 for the row-major one (DBA_SHEAR_TILES=0
 * keeps every shape row-major). */
int dba_corr_sheared_tiled(int h1, int w1);
/* ... and the grid the tiles are counted on: (h1g, w1g) = (h1, w1) rounded up to multiples of (4, 64) for tiled planes (then
 * dba_corr_sheared_plane_elems = h1g * w1g; the pad pixels' entries are never read as taps), (h1, w1) itself for linear ones.
 * Returns the tile width like dba_corr_sheared_tiled.  With DBA_SHEAR_PAD=1 (opt-in, read once per process) maps within 25 % of
 * such a grid are tiled on it (28 x 107 on 28 x 128, 55 x 55 on 56 x 64); coordinates and the returned tensors keep the map's own
 * [h1, w1] indexing.  By default only the maps that ARE such a grid are tiled. */
int dba_corr_sheared_grid(int h1, int w1, int *h1g, int *w1g);
/* which form of the sheared lookup dba_corr_lookup_pyramid_sheared launches: 0 = automatic (by map shape), 2 = resident
 * (any shape), 5 = rows over tiles (tiled planes only, falls back to resident otherwise).  Process-wide; results are
 * bit-identical.  (1, 3, 4 were round 4's streaming / pair / band forms, which no longer ship: DBA_ERR_ARG.) */
int dba_corr_lookup_select(int kernel);
/* Measurement hook: the next dba_corr_lookup_pyramid_sheared call of this thread attaches the two hipEvent_t to its
 * kernel dispatch (hipExtLaunchKernelGGL: the dispatch's own start / end timestamps, no marker packets in the stream),
 * then disarms.  hipEventElapsedTime(start, stop) is the kernel's duration; bench.py times the roofline kernel this
 * way in every timed step.  NULL, NULL disarms. */
int dba_corr_lookup_arm_timing(void *start_event, void *stop_event);
/* Fused build of the sheared pyramid straight from the feature maps (csrc/corr_build_fused.hip): MFMA GEMM,
 * 2x2 pooling of the rounded levels and the flow-aligned store in one pass; every output byte is written once.
 * Supported when dba_corr_volume_build_sheared_supported(...) returns 1 (w2 <= 128, C % 16 == 0, 4 levels, every
 * level non-empty); sheared_levels[l] is [n, h2>>l, w2>>l, dba_corr_sheared_plane_elems(h1, w1)] f16.  scratch as for
 * dba_corr_volume_build. */
int dba_corr_volume_build_sheared_supported(int C, int h1, int w1, int h2, int w2, int num_levels);
int dba_corr_volume_build_sheared(const void *fmap1, const void *fmap2, void *const *sheared_levels, int n, int C,
                                  int h1, int w1, int h2, int w2, int num_levels, void *scratch,
                                  size_t scratch_bytes, dba_stream_t stream);
int dba_corr_shear_level(const void *ref_level, void *sheared_level, int n, int h1, int w1, int h2l, int w2l,
                         int lvl, dba_stream_t stream);
int dba_corr_lookup_pyramid_sheared(const void *const *volumes /* host array of L device ptrs */,
                                    const float *coords_nhw2, void *corr, int n, int h1, int w1, int h2,
                                    int w2, int num_levels, int radius, dba_stream_t stream);

/* Slot-addressed pyramid (the MI355X form of CorrBlock.cat / CorrBlock.__getitem__, dbaf/modules/corr.py:52-60, which the
 * reference runs as torch.cat / boolean indexing of the WHOLE pyramid on every add_factors / rm_factors,
 * dbaf/covisible_graph.py:131,166).  A level store holds `capacity` edge volumes; edge e of a lookup lives in slot
 * slots[e] (device int32 [n]; NULL = slot e).  New edges are built straight into free slots (out_slots[e]: where edge e of
 * this build goes), removing or re-ordering edges edits the table: no volume is moved. */
int dba_corr_volume_build_sheared_slots(const void *fmap1, const void *fmap2, void *const *level_stores,
                                        const int *out_slots /* device [n] or NULL */, int n, int C, int h1, int w1, int h2,
                                        int w2, int num_levels, void *scratch, size_t scratch_bytes, dba_stream_t stream);
int dba_corr_lookup_pyramid_sheared_slots(const void *const *level_stores, const int *slots /* device [n] or NULL */,
                                          const float *coords_nhw2, void *corr, int n, int h1, int w1, int h2, int w2,
                                          int num_levels, int radius, dba_stream_t stream);
int dba_corr_lookup_pyramid_slots(const void *const *level_stores, const int *slots, const float *coords_nhw2, void *corr,
                                  int n, int h1, int w1, int h2, int w2, int num_levels, int radius, int dtype,
                                  dba_stream_t stream);

/* The lookup with the reprojection in its prologue: pops.projective_transform (dbaf/geom/projective_ops.py:96-125, via
 * DepthVideo.reproject, dbaf/depth_video.py:221-229) + CorrBlock.__call__ (dbaf/modules/corr.py:40-50) in ONE launch.
 * poses [B,7], disps [B,h1,w1], intrinsics_b4 [B,4] (per frame), ii, jj [n] int64; the coordinates are computed per pixel
 * (csrc/reproj.h: the arithmetic of dba_reproject, bit for bit) and never read back; coords_out [n,h1,w1,2] and valid_out
 * [n,h1,w1,1] (either may be NULL) receive what dba_reproject would have written -- the caller needs them for the motion
 * features and the BA targets (dbaf/covisible_graph.py:220-221,237).  corr is bit-identical to dba_reproject followed by
 * dba_corr_lookup_pyramid_sheared_slots. */
int dba_corr_lookup_reproject_sheared(const void *const *level_stores, const int *slots, const float *poses,
                                      const float *disps, const float *intrinsics_b4, const int64_t *ii, const int64_t *jj,
                                      float *coords_out, float *valid_out, void *corr, int n, int h1, int w1, int h2, int w2,
                                      int num_levels, int radius, dba_stream_t stream);

/* A pyramid that is built, looked up ONCE and dropped -- MotionFilter.track, dbaf/motion_filter.py:74-76:
 * `corr = CorrBlock(self.fmap[None,[0]], gmap[None,[0]])(coords0)`, once per incoming frame -- as one call: the levels go into
 * `pyramid` (device scratch of dba_corr_once_pyramid_bytes bytes that the caller keeps per stream and reuses from frame to frame:
 * calls on one stream are ordered), then the fused lookup reads them.  = dba_corr_volume_build_sheared_slots followed by
 * dba_corr_lookup_pyramid_sheared_slots with identity slots, bit for bit; what it saves is the host's share (four level
 * allocations and one library call per frame).  DBA_ERR_UNSUPPORTED where dba_corr_volume_build_sheared_supported says no. */
size_t dba_corr_once_pyramid_bytes(int n, int h1, int w1, int h2, int w2, int num_levels);
int dba_corr_build_lookup_once_sheared(const void *fmap1, const void *fmap2, const float *coords_nhw2, void *corr, void *pyramid,
                                       size_t pyramid_bytes, void *scratch, size_t scratch_bytes, int n, int C, int h1, int w1,
                                       int h2, int w2, int num_levels, int radius, dba_stream_t stream);

/* ONE level of the flow-aligned pyramid looked up with the arguments droid_backends.corr_index_forward receives from the
 * reference's unmodified CorrBlock.__call__ (dbaf/modules/corr.py:40-50): coords [n, 2, h1, w1] ALREADY divided by 2^lvl,
 * corr [n, 7, 7, h1, w1].  h2, w2 are the LEVEL-0 target map sizes (the level's planes are (h2 >> lvl) x (w2 >> lvl)).
 * Bit-identical to dba_corr_index_forward on the reference-layout level; the adapter keeps a flow-aligned shadow of a
 * reference-layout level it is asked about repeatedly and serves the lookups from it (droid_backends/__init__.py). */
int dba_corr_lookup_level_sheared(const void *sheared_level, const float *coords_n2hw_scaled, void *corr, int n, int h1,
                                  int w1, int h2, int w2, int lvl, int radius, dba_stream_t stream);

/* the same on a slot-addressed shadow store (slots[e]: where edge e of the call lives), and the re-layout pass that fills
 * it: edge e of the pass is edge src_idx[e] of ref_level and goes to slot dst_slots[e] (either NULL: e).  Used by the
 * adapter when it matches the edges of a NEW reference-layout tensor (torch.cat / boolean index of the previous one:
 * dbaf/modules/corr.py:52-60) to shadows it already holds and re-lays only the edges it has not seen. */
int dba_corr_lookup_level_sheared_slots(const void *sheared_store, const int *slots, const float *coords_n2hw_scaled,
                                        void *corr, int n, int h1, int w1, int h2, int w2, int lvl, int radius,
                                        dba_stream_t stream);
int dba_corr_shear_level_slots(const void *ref_level, void *sheared_store, const int *src_idx, const int *dst_slots, int n,
                               int h1, int w1, int h2l, int w2l, int lvl, dba_stream_t stream);

/* corr_index_backward (src/correlation_kernels.cu:73-124,157-185): adjoint of the lookup;
 * volume_grad [n,h1,w1,h2,w2] must be zero-initialised by the caller. f32 only. */
int dba_corr_index_backward(const float *coords, const float *corr_grad, float *volume_grad, int n,
                            int h1, int w1, int h2, int w2, int radius, dba_stream_t stream);

/* CorrBlock.corr + pyramid (corr.py:24-38, :63-71): fmap1, fmap2 [n,C,h,w] f16 ->
 * level l [n,h1,w1,h2>>l,w2>>l] f16 for l < num_levels.  corr = (f1/4)^T (f2/4), fp32 accumulate on
 * MFMA, rounded once to f16; levels l>0 = 2x2 average of the rounded level l-1 (F.avg_pool2d).
 * scratch: device bytes from dba_corr_volume_scratch_bytes (channels-last staging of the fmaps). */
size_t dba_corr_volume_scratch_bytes(int n, int C, int h1, int w1, int h2, int w2);
int dba_corr_volume_build(const void *fmap1, const void *fmap2, void *const *levels /* host array */,
                          int n, int C, int h1, int w1, int h2, int w2, int num_levels, void *scratch,
                          size_t scratch_bytes, dba_stream_t stream);

/* altcorr_forward (src/droid.cpp:254-264, src/altcorr_kernel.cu:27-149,290-319): on-the-fly
 * windowed correlation. fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C] channels-last f32,
 * coords [B,S,H1,W1,2] f32 -> corr [B,S,(2r+1)^2,H1,W1] f32, channel = y_offset + (2r+1)*x_offset. */
int dba_altcorr_forward(const float *fmap1, const float *fmap2, const float *coords, float *corr, int B,
                        int S, int H1, int W1, int H2, int W2, int C, int radius, dba_stream_t stream);
/* the same with the element type of fmap1, fmap2 and corr selectable (the reference dispatches
 * AT_DISPATCH_FLOATING_TYPES_AND_HALF, src/altcorr_kernel.cu:304): dtype = DBA_F32 or DBA_F16; coords stay f32.
 * The half instantiation rounds every product and sum to half like c10::Half and is bit-identical to the reference. */
int dba_altcorr_forward_t(const void *fmap1, const void *fmap2, const float *coords, void *corr, int B, int S,
                          int H1, int W1, int H2, int W2, int C, int radius, int dtype, dba_stream_t stream);

/* AltCorrBlock.corr_fn (dbaf/modules/corr.py:107-125) in ONE launch: for every edge n (source frame ii[n], target frame
 * jj[n]; NULL = frame n), coordinate set and pyramid level l < num_levels, the on-the-fly correlation of fmap1 [F,H1,W1,C]
 * (level 0 of the pyramid) with fmap2_levels[l] [F,H1>>l,W1>>l,C] at coords / 2^l; corr [B,S,num_levels*(2r+1)^2,H1,W1].
 * Same arithmetic as dba_altcorr_forward_t per level (bit-identical), no gathered copies of the maps, no scaled copies of the
 * coordinates.  B * S * num_levels <= 65535. */
int dba_altcorr_pyramid_forward(const void *fmap1, const void *const *fmap2_levels /* host array of L device ptrs */,
                                const int64_t *ii, const int64_t *jj, const float *coords, void *corr, int B, int S, int H1,
                                int W1, int C, int num_levels, int radius, int dtype, dba_stream_t stream);

/* The same lookup for HALF feature pyramids with float output, on the matrix cores: what AltCorrBlock computes in the
 * reference, whose half pyramid (fmaps / 4 under autocast) is cast with .float() at every lookup (dbaf/modules/corr.py:120)
 * and correlated in float.  Products of halves are exact in float; the result differs from the float chain of
 * altcorr_kernel.cu:58-147 only in the order of the float additions.  fmap1 / fmap2_levels: half, channels-last
 * [F, H1 >> l, W1 >> l, C]; C % 16 == 0, C <= 128; radius 3; corr: float [B, S, L * 49, H1, W1]. */
int dba_altcorr_pyramid_forward_f16maps(const void *fmap1, const void *const *fmap2_levels /* host array of L device ptrs */,
                                        const int64_t *ii, const int64_t *jj, const float *coords, float *corr, int B, int S,
                                        int H1, int W1, int C, int num_levels, int radius, dba_stream_t stream);

/* altcorr_backward (src/droid.cpp:266-278, src/altcorr_kernel.cu:152-286,321-356; training only):
 * gradients wrt the feature maps; fmap1_grad [B,H1,W1,C], fmap2_grad [B,H2,W2,C] must be zero-initialised
 * by the caller (fmap2_grad is accumulated with float atomics like the reference); the reference's
 * coords_grad is allocated and never written (stays zero), so it has no entry point here. */
int dba_altcorr_backward(const float *fmap1, const float *fmap2, const float *coords, const float *corr_grad,
                         float *fmap1_grad, float *fmap2_grad, int B, int S, int H1, int W1, int H2, int W2,
                         int C, int radius, dba_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Geometry.
 * ---------------------------------------------------------------------------------------- */

/* pops.projective_transform without jacobians (dbaf/geom/projective_ops.py:96-125) as one kernel:
 * intrinsics [B,4] per frame; coords [N,ht,wd,2], valid [N,ht,wd,1] f32. */
int dba_reproject(const float *poses, const float *disps, const float *intrinsics_b4,
                  const int64_t *ii, const int64_t *jj, int N, int ht, int wd, float *coords,
                  float *valid, dba_stream_t stream);

/* droid_backends.frame_distance (src/droid.cpp:181-197, src/droid_kernels.cu:562-702). dist [N]. */
int dba_frame_distance(const float *poses, const float *disps, const float *intrinsics,
                       const int64_t *ii, const int64_t *jj, int N, int ht, int wd, float beta,
                       float *dist, dba_stream_t stream);

/* droid_backends.projmap (src/droid.cpp:200-215, src/droid_kernels.cu:471-560).
 * coords [N,ht,wd,3] (3rd channel left 0), valid [N,ht,wd,1]. */
int dba_projmap(const float *poses, const float *disps, const float *intrinsics, const int64_t *ii,
                const int64_t *jj, int N, int ht, int wd, float *coords, float *valid,
                dba_stream_t stream);

/* droid_backends.iproj (src/droid.cpp:218-227, src/droid_kernels.cu:824-895). points [nm,ht,wd,3]. */
int dba_iproj(const float *poses, const float *disps, const float *intrinsics, int nm, int ht, int wd,
              float *points, dba_stream_t stream);

/* droid_backends.depth_filter (src/droid.cpp:281-295, src/droid_kernels.cu:706-820).
 * counter [num,ht,wd] zero-initialised by the caller; nbuf = disps.size(0). */
int dba_depth_filter(const float *poses, const float *disps, const float *intrinsics,
                     const int64_t *inds, const float *thresh, int num, int nbuf, int ht, int wd,
                     float *counter, dba_stream_t stream);

/* ---- optional exchange step of the edge-sharded driver: one-shot all-reduce by direct peer reads ------------------
 * The reference is single-GPU (nothing to replace); SURVEY.md 8(e) asks for the float64 sum of the reduced camera
 * system [H | b] over the ranks, which dbaf_amd/sharded.py does with RCCL by default.  These entry points are the
 * latency-oriented alternative for one node (csrc/peer_allreduce.hip): every rank owns an exchange region that its
 * peers map through hipIpc and sums the world's contributions itself, in rank order (bit-identical replicas).
 *   bytes   = dba_peer_exchange_bytes(max_doubles)
 *   create  : hipMalloc + zero + IPC handle (64 bytes) of this rank's region
 *   open    : map a PEER process' region from its handle (not the creator's own: use the pointer `create` returned)
 *   close   : opened != 0 -> hipIpcCloseMemHandle, else hipFree
 *   allreduce: buf[0..n) += everybody else's, in place, enqueued on `stream`; `regions[world]` are device pointers (own
 *             region at index `rank`), `epoch` = 1, 2, 3, ... identical on all ranks and increasing by one per call,
 *             `status` a device int the kernel sets to DBA_PEER_TIMEOUT if a peer's contribution did not arrive within
 *             ~2 s (buf is left untouched then). */
#define DBA_PEER_TIMEOUT 1
size_t dba_peer_exchange_bytes(size_t max_doubles);
int dba_peer_exchange_create(size_t bytes, void **region, unsigned char *handle64);
int dba_peer_exchange_open(const unsigned char *handle64, void **region);
int dba_peer_exchange_close(void *region, int opened);
int dba_peer_allreduce_f64(double *buf, size_t n, void *const *regions, int rank, int world, unsigned epoch,
                           size_t max_doubles, int *status, dba_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DBA_HIP_H */
