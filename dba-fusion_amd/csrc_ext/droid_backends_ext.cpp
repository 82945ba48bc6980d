// Compiled `droid_backends` for MI355X: the pybind11 module a maintainer would build in place of the reference's
// src/droid.cpp (/root/reference/src/droid.cpp:1-316) -- same function names, argument order, in-place semantics and error
// behaviour (CHECK_CONTIGUOUS -> c10::Error -> RuntimeError), every function a thin adapter over the C ABI of
// include/dba_hip.h (libdba_hip.so: the hand-written gfx950 kernels).  No kernels here, no torch types below this file.
//
//   module name      _droid_backends_C   (dba-fusion_amd/droid_backends/__init__.py re-exports its stateless operators;
//                                          `import _droid_backends_C as droid_backends` is a drop-in by itself: ba and
//                                          BACore keep one workspace per window shape here too and let stage 0 recognise
//                                          an unchanged edge list on the device.  Only the flow-aligned shadows of
//                                          corr_index_forward are a policy of the Python package alone.)
//   built by         make ext            (g++ against the torch headers; links libdba_hip.so by $ORIGIN-relative rpath)
//
// Work is enqueued on the caller's CURRENT HIP stream (c10::hip::getCurrentHIPStream), like every torch op around it.
#include <torch/extension.h>

#include <c10/hip/HIPStream.h>

#include <cstdint>
#include <list>
#include <mutex>
#include <unordered_map>
#include <string>
#include <tuple>
#include <vector>

#include "dba_hip.h"

namespace {

#define CHECK_CONTIGUOUS(x) TORCH_CHECK((x).is_contiguous(), #x " must be contiguous")   // src/droid.cpp:105
#define CHECK_DEVICE(x) TORCH_CHECK((x).is_cuda(), "droid_backends (MI355X): " #x " must be a HIP device tensor; there is no CPU path")
#define CHECK_INPUT(x) \
  do {                 \
    CHECK_CONTIGUOUS(x); \
    CHECK_DEVICE(x);     \
  } while (0)

void check(int rc, const char *what) {
  if (rc == DBA_OK) return;
  const char *name = rc == DBA_ERR_ARG ? "DBA_ERR_ARG" : rc == DBA_ERR_WORKSPACE ? "DBA_ERR_WORKSPACE"
                     : rc == DBA_ERR_HIP ? "DBA_ERR_HIP" : rc == DBA_ERR_UNSUPPORTED ? "DBA_ERR_UNSUPPORTED" : "error";
  TORCH_CHECK(false, "dba_hip: ", what, " failed with ", name, " ", rc == DBA_ERR_HIP ? dba_last_error() : "");
}

dba_stream_t stream_of(const torch::Tensor &t) { return (dba_stream_t)c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

struct BaDims {
  int N, B, ht, wd, t0, t1, eta_rows;
};

BaDims ba_dims(const torch::Tensor &disps, const torch::Tensor &eta, const torch::Tensor &ii, int t0, int t1) {
  TORCH_CHECK(disps.dim() == 3, "disps must be [B, ht, wd]");
  BaDims d;
  d.N = (int)ii.size(0);
  d.B = (int)disps.size(0);
  d.ht = (int)disps.size(1);
  d.wd = (int)disps.size(2);
  d.t0 = t0;
  d.t1 = t1;
  const int64_t hw = (int64_t)d.ht * d.wd;
  TORCH_CHECK(eta.numel() % hw == 0, "eta must view as [-1, ht*wd] (droid_kernels.cu:1476)");
  d.eta_rows = (int)(eta.numel() / hw);
  return d;
}

// One workspace per (pool, device, stream, window shape), the latest few, marked "no graph prepared" when allocated
// (dba_ba_workspace_init): stage 0 of a later call compares its edge list with the key the previous one left there and skips
// itself when the graph is unchanged -- the reference's caller hands over NEW ii / jj tensors with the same edges on every
// update (torch.cat, dbaf/covisible_graph.py:242-247), so tensor identity would never hit.  pool 0: ba, pool 1: BACore (so
// that nothing ba does can fall between a BACore's hessian and retract).
torch::Tensor workspace(const BaDims &d, const torch::Tensor &like, size_t *nbytes, int pool) {
  using Key = std::tuple<int, int, void *, int, int, int, int, int, int>;
  static std::mutex mu;
  // (leaked on purpose: at static-destruction time the HIP runtime may be gone already, and freeing device memory then faults)
  static auto &cache = *new std::list<std::pair<Key, torch::Tensor>>;
  *nbytes = dba_ba_workspace_bytes(d.N, d.B, d.ht, d.wd, d.t0, d.t1);
  TORCH_CHECK(*nbytes > 0, "dba_ba_workspace_bytes: invalid sizes");
  const Key key{pool, (int)like.get_device(), (void *)stream_of(like), d.N, d.B, d.ht, d.wd, d.t0, d.t1};
  std::lock_guard<std::mutex> lock(mu);
  for (auto it = cache.begin(); it != cache.end(); ++it)
    if (it->first == key) {
      cache.splice(cache.begin(), cache, it);
      return cache.front().second;
    }
  torch::Tensor ws = torch::empty({(int64_t)*nbytes}, like.options().dtype(torch::kUInt8));
  check(dba_ba_workspace_init(d.N, d.B, d.ht, d.wd, d.t0, d.t1, ws.data_ptr(), *nbytes, stream_of(like)), "dba_ba_workspace_init");
  cache.emplace_front(key, ws);
  if (cache.size() > 8) cache.pop_back();
  return ws;
}

// eta.view(-1, HW) must have one row or |kx| rows (droid_kernels.cu:1476).  |kx| only exists on the device: stage 0 compares
// and records a mismatch in pinned memory, which the NEXT call into this module raises (the reference raises in the call
// itself, whose torch::_unique stops the host anyway; these calls never do).
void raise_pending_eta_error() {
  int rows = 0, nk = 0;
  TORCH_CHECK(!dba_ba_poll_eta_error(&rows, &nk), "an earlier droid_backends.ba / BACore.hessian call was given eta with ", rows,
              " rows; it must have 1 or |unique(arange(t0,t1) U ii)| = ", nk, " rows (droid_kernels.cu:1476)");
}

void check_eta_bounds(const BaDims &d) {   // what is known without the device
  const int P = std::max(d.t1 - d.t0, 0);
  TORCH_CHECK(d.eta_rows == 1 || (d.eta_rows <= P + d.N && d.eta_rows >= std::min(P, 1)), "eta has ", d.eta_rows,
              " rows; it must have 1 or |unique(arange(t0,t1) U ii)| rows (at most ", P + d.N, " here)");
}

// |kx| for the shape of the returned dz: eta's row count when it has one row per kx entry (as DBA-Fusion passes it,
// covisible_graph.py:330; verified by stage 0), else counted on the device like the reference's torch::_unique (a sync)
int64_t num_kx(const BaDims &d, const torch::Tensor &ii) {
  if (d.eta_rows > 1) return d.eta_rows;
  torch::Tensor ts = torch::arange(d.t0, d.t1, ii.options());
  return std::get<0>(at::_unique(torch::cat({ts, ii}, 0))).size(0);
}

std::vector<torch::Tensor> ba_impl(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics, torch::Tensor disps_sens,
                                   torch::Tensor targets, torch::Tensor weights, torch::Tensor eta, torch::Tensor ii,
                                   torch::Tensor jj, const int t0, const int t1, const int iterations, const float lm,
                                   const float ep, const bool motion_only, const float disp_floor) {
  CHECK_INPUT(targets);
  CHECK_INPUT(weights);
  CHECK_INPUT(poses);
  CHECK_INPUT(disps);
  CHECK_INPUT(intrinsics);
  CHECK_INPUT(disps_sens);
  CHECK_INPUT(ii);
  CHECK_INPUT(jj);
  eta = eta.contiguous();   // the reference takes eta.view(-1, ht*wd) of whatever it is given
  CHECK_DEVICE(eta);
  raise_pending_eta_error();
  const BaDims d = ba_dims(disps, eta, ii, t0, t1);
  check_eta_bounds(d);
  if (iterations <= 0) return {torch::Tensor(), torch::Tensor()};   // (two undefined tensors, nothing touched: :1437)
  size_t nbytes;
  torch::Tensor ws = workspace(d, poses, &nbytes, 0);
  const int P = t1 - t0;
  const int Mmax = std::min(d.B, P + d.N);
  torch::Tensor dx = torch::empty({P, 6}, poses.options());
  torch::Tensor dz = torch::empty({Mmax, (int64_t)d.ht * d.wd}, poses.options());
  check(dba_ba_run(poses.data_ptr<float>(), disps.data_ptr<float>(), intrinsics.data_ptr<float>(), disps_sens.data_ptr<float>(),
                   targets.data_ptr<float>(), weights.data_ptr<float>(), eta.data_ptr<float>(), d.eta_rows, ii.data_ptr<int64_t>(),
                   jj.data_ptr<int64_t>(), d.N, d.B, d.ht, d.wd, t0, t1, iterations, lm, ep, motion_only ? 1 : 0,
                   dx.data_ptr<float>(), dz.data_ptr<float>(), ws.data_ptr(), nbytes, stream_of(poses), /*prepared=*/2, 0,
                   disp_floor),
        "dba_ba");
  if (motion_only) return {dx, torch::Tensor()};
  return {dx, dz.narrow(0, 0, num_kx(d, ii))};   // [|kx|, ht*wd], the reference's shape
}

// ba (src/droid.cpp:109-138): in place on poses[t0:t1], disps[kx]; returns {dx, dz}
std::vector<torch::Tensor> ba(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics, torch::Tensor disps_sens,
                              torch::Tensor targets, torch::Tensor weights, torch::Tensor eta, torch::Tensor ii,
                              torch::Tensor jj, const int t0, const int t1, const int iterations, const float lm,
                              const float ep, const bool motion_only) {
  return ba_impl(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, iterations, lm, ep, motion_only, 0.f);
}

// ba followed by the caller's `disps.clamp_(min=disp_floor)` (dbaf/depth_video.py:559-560) in one call (dba_ba_run): not a
// reference binding, see droid_backends.ba_clamped
std::vector<torch::Tensor> ba_clamped(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics, torch::Tensor disps_sens,
                                      torch::Tensor targets, torch::Tensor weights, torch::Tensor eta, torch::Tensor ii,
                                      torch::Tensor jj, const int t0, const int t1, const int iterations, const float lm,
                                      const float ep, const bool motion_only, const float disp_floor) {
  TORCH_CHECK(disp_floor > 0.f, "ba_clamped: disp_floor must be positive");
  return ba_impl(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, iterations, lm, ep, motion_only,
                 disp_floor);
}

// The edge tensors of one BA call as the reference's caller assembles them (dbaf/covisible_graph.py:242-247, :332-333) in one
// launch: not a reference binding, see droid_backends.gather_edges
std::vector<torch::Tensor> gather_edges(torch::Tensor target_inac, torch::Tensor weight_inac, torch::Tensor ii_inac,
                                        torch::Tensor jj_inac, c10::optional<torch::Tensor> sel, torch::Tensor target,
                                        torch::Tensor weight, torch::Tensor ii, torch::Tensor jj) {
  CHECK_INPUT(target_inac);
  CHECK_INPUT(weight_inac);
  CHECK_INPUT(ii_inac);
  CHECK_INPUT(jj_inac);
  CHECK_INPUT(target);
  CHECK_INPUT(weight);
  CHECK_INPUT(ii);
  CHECK_INPUT(jj);
  TORCH_CHECK(target.dim() >= 4 && target.size(-1) == 2, "gather_edges: target must be [..., ht, wd, 2]");
  const int ht = (int)target.size(-3), wd = (int)target.size(-2);
  auto four = [&](const torch::Tensor &x, const char *name) {
    TORCH_CHECK(x.dim() >= 4 && x.size(-1) == 2 && x.size(-3) == ht && x.size(-2) == wd && x.scalar_type() == torch::kFloat32,
                "gather_edges: ", name, " must be float32 [..., ", ht, ", ", wd, ", 2]");
    return x.reshape({-1, ht, wd, 2});
  };
  const torch::Tensor ta = four(target, "target"), wa = four(weight, "weight"), ti = four(target_inac, "target_inac"),
                      wi = four(weight_inac, "weight_inac");
  auto idx = [&](const torch::Tensor &x, const char *name) {
    TORCH_CHECK(x.scalar_type() == torch::kInt64, "gather_edges: ", name, " must be int64");
    return x.reshape({-1});
  };
  const torch::Tensor iia = idx(ii, "ii"), jja = idx(jj, "jj"), iii = idx(ii_inac, "ii_inac"), jji = idx(jj_inac, "jj_inac");
  const int n_act = (int)ta.size(0), n_inac = (int)ti.size(0);
  TORCH_CHECK(wa.size(0) == n_act && iia.size(0) == n_act && jja.size(0) == n_act,
              "gather_edges: target, weight, ii, jj disagree on the number of active edges");
  TORCH_CHECK(wi.size(0) == n_inac && iii.size(0) == n_inac && jji.size(0) == n_inac,
              "gather_edges: target_inac, weight_inac, ii_inac, jj_inac disagree on the number of inactive edges");
  torch::Tensor s;
  int n_sel = n_inac;
  if (sel.has_value()) {
    s = sel->reshape({-1});
    CHECK_DEVICE(s);
    if (s.scalar_type() == torch::kBool) s = s.nonzero().reshape({-1});   // (synchronises the host, like `x[mask]` itself)
    TORCH_CHECK(s.scalar_type() == torch::kInt64, "gather_edges: sel must be int64 indices or a boolean mask");
    s = s.contiguous();
    n_sel = (int)s.size(0);
  }
  const int n = n_sel + n_act;
  torch::Tensor tg = torch::empty({n, 2, ht, wd}, target.options()), wt = torch::empty({n, 2, ht, wd}, target.options());
  torch::Tensor ii_n = torch::empty({n}, ii.options()), jj_n = torch::empty({n}, ii.options());
  check(dba_ba_gather_edges(ti.data_ptr<float>(), wi.data_ptr<float>(), iii.data_ptr<int64_t>(), jji.data_ptr<int64_t>(), n_inac,
                            sel.has_value() ? s.data_ptr<int64_t>() : nullptr, n_sel, ta.data_ptr<float>(), wa.data_ptr<float>(),
                            iia.data_ptr<int64_t>(), jja.data_ptr<int64_t>(), n_act, ht, wd, tg.data_ptr<float>(),
                            wt.data_ptr<float>(), ii_n.data_ptr<int64_t>(), jj_n.data_ptr<int64_t>(), stream_of(target)),
        "dba_ba_gather_edges");
  return {ii_n, jj_n, tg, wt};
}

// frame_distance (src/droid.cpp:181-197)
torch::Tensor frame_distance(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics, torch::Tensor ii,
                             torch::Tensor jj, const float beta) {
  CHECK_INPUT(poses);
  CHECK_INPUT(disps);
  CHECK_INPUT(intrinsics);
  CHECK_INPUT(ii);
  CHECK_INPUT(jj);
  const int N = (int)ii.size(0);
  torch::Tensor dist = torch::empty({N}, poses.options());   // (every entry is written by its workgroup)
  check(dba_frame_distance(poses.data_ptr<float>(), disps.data_ptr<float>(), intrinsics.data_ptr<float>(), ii.data_ptr<int64_t>(),
                           jj.data_ptr<int64_t>(), N, (int)disps.size(1), (int)disps.size(2), beta, dist.data_ptr<float>(),
                           stream_of(poses)),
        "dba_frame_distance");
  return dist;
}

// projmap (src/droid.cpp:200-215)
std::vector<torch::Tensor> projmap(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics, torch::Tensor ii,
                                   torch::Tensor jj) {
  CHECK_INPUT(poses);
  CHECK_INPUT(disps);
  CHECK_INPUT(intrinsics);
  CHECK_INPUT(ii);
  CHECK_INPUT(jj);
  const int N = (int)ii.size(0), ht = (int)disps.size(1), wd = (int)disps.size(2);
  torch::Tensor coords = torch::zeros({N, ht, wd, 3}, poses.options());
  torch::Tensor valid = torch::zeros({N, ht, wd, 1}, poses.options());
  check(dba_projmap(poses.data_ptr<float>(), disps.data_ptr<float>(), intrinsics.data_ptr<float>(), ii.data_ptr<int64_t>(),
                    jj.data_ptr<int64_t>(), N, ht, wd, coords.data_ptr<float>(), valid.data_ptr<float>(), stream_of(poses)),
        "dba_projmap");
  return {coords, valid};
}

// iproj (src/droid.cpp:218-227)
torch::Tensor iproj(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics) {
  CHECK_INPUT(poses);
  CHECK_INPUT(disps);
  CHECK_INPUT(intrinsics);
  const int nm = (int)disps.size(0), ht = (int)disps.size(1), wd = (int)disps.size(2);
  torch::Tensor points = torch::zeros({nm, ht, wd, 3}, disps.options());
  check(dba_iproj(poses.data_ptr<float>(), disps.data_ptr<float>(), intrinsics.data_ptr<float>(), nm, ht, wd,
                  points.data_ptr<float>(), stream_of(disps)),
        "dba_iproj");
  return points;
}

// depth_filter (src/droid.cpp:281-295)
torch::Tensor depth_filter(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics, torch::Tensor ix,
                           torch::Tensor thresh) {
  CHECK_INPUT(poses);
  CHECK_INPUT(disps);
  CHECK_INPUT(intrinsics);
  CHECK_INPUT(ix);
  CHECK_INPUT(thresh);
  const int num = (int)ix.size(0), nbuf = (int)disps.size(0), ht = (int)disps.size(1), wd = (int)disps.size(2);
  torch::Tensor counter = torch::zeros({num, ht, wd}, disps.options());
  check(dba_depth_filter(poses.data_ptr<float>(), disps.data_ptr<float>(), intrinsics.data_ptr<float>(), ix.data_ptr<int64_t>(),
                         thresh.data_ptr<float>(), num, nbuf, ht, wd, counter.data_ptr<float>(), stream_of(disps)),
        "dba_depth_filter");
  return counter;
}

int vol_dtype(const torch::Tensor &t) {
  if (t.scalar_type() == torch::kHalf) return DBA_F16;
  if (t.scalar_type() == torch::kFloat) return DBA_F32;
  TORCH_CHECK(false, "volume / feature-map dtype not supported on the MI355X path (half / float)");
}

// corr_index_forward (src/droid.cpp:231-239): the kernel on the reference layout (no shadows here: a policy of the Python
// adapter)
std::vector<torch::Tensor> corr_index_forward(torch::Tensor volume, torch::Tensor coords, int radius) {
  CHECK_INPUT(volume);
  CHECK_INPUT(coords);
  TORCH_CHECK(coords.scalar_type() == torch::kFloat, "coords must be float32");
  const int n = (int)volume.size(0), h1 = (int)volume.size(1), w1 = (int)volume.size(2), h2 = (int)volume.size(3),
            w2 = (int)volume.size(4);
  const int rd = 2 * radius + 1;
  torch::Tensor corr = torch::empty({n, rd, rd, h1, w1}, volume.options());
  check(dba_corr_index_forward(volume.data_ptr(), coords.data_ptr<float>(), corr.data_ptr(), n, h1, w1, h2, w2, radius,
                               vol_dtype(volume), stream_of(volume)),
        "dba_corr_index_forward");
  return {corr};
}

// corr_index_backward (src/droid.cpp:241-252; training only)
std::vector<torch::Tensor> corr_index_backward(torch::Tensor volume, torch::Tensor coords, torch::Tensor corr_grad, int radius) {
  CHECK_INPUT(volume);
  CHECK_INPUT(coords);
  CHECK_INPUT(corr_grad);
  const int n = (int)volume.size(0), h1 = (int)volume.size(1), w1 = (int)volume.size(2), h2 = (int)volume.size(3),
            w2 = (int)volume.size(4);
  torch::Tensor vg = torch::zeros(volume.sizes(), volume.options().dtype(torch::kFloat));
  torch::Tensor cg = corr_grad.to(torch::kFloat).contiguous();
  check(dba_corr_index_backward(coords.data_ptr<float>(), cg.data_ptr<float>(), vg.data_ptr<float>(), n, h1, w1, h2, w2, radius,
                                stream_of(volume)),
        "dba_corr_index_backward");
  return {vg.to(volume.scalar_type())};
}

// altcorr_forward (src/droid.cpp:254-264)
std::vector<torch::Tensor> altcorr_forward(torch::Tensor fmap1, torch::Tensor fmap2, torch::Tensor coords, int radius) {
  CHECK_INPUT(fmap1);
  CHECK_INPUT(fmap2);
  CHECK_INPUT(coords);
  TORCH_CHECK(fmap1.scalar_type() == fmap2.scalar_type(), "altcorr_forward: fmap1 / fmap2 must have one dtype");
  const int B = (int)coords.size(0), S = (int)coords.size(1), H1 = (int)coords.size(2), W1 = (int)coords.size(3);
  const int H2 = (int)fmap2.size(1), W2 = (int)fmap2.size(2), C = (int)fmap2.size(3);
  const int rd = 2 * radius + 1;
  torch::Tensor corr = torch::empty({B, S, rd * rd, H1, W1}, fmap1.options());
  check(dba_altcorr_forward_t(fmap1.data_ptr(), fmap2.data_ptr(), coords.data_ptr<float>(), corr.data_ptr(), B, S, H1, W1, H2, W2,
                              C, radius, vol_dtype(fmap1), stream_of(fmap1)),
        "dba_altcorr_forward");
  return {corr};
}

// altcorr_backward (src/droid.cpp:266-278; training only): {fmap1_grad, fmap2_grad, coords_grad (zeros, like the reference's)}
std::vector<torch::Tensor> altcorr_backward(torch::Tensor fmap1, torch::Tensor fmap2, torch::Tensor coords,
                                            torch::Tensor corr_grad, int radius) {
  CHECK_INPUT(fmap1);
  CHECK_INPUT(fmap2);
  CHECK_INPUT(coords);
  CHECK_INPUT(corr_grad);
  const auto dt = fmap1.scalar_type();
  torch::Tensor f1 = fmap1.to(torch::kFloat), f2 = fmap2.to(torch::kFloat), g = corr_grad.to(torch::kFloat).contiguous();
  const int B = (int)coords.size(0), S = (int)coords.size(1), H1 = (int)coords.size(2), W1 = (int)coords.size(3);
  const int H2 = (int)fmap2.size(1), W2 = (int)fmap2.size(2), C = (int)fmap2.size(3);
  torch::Tensor g1 = torch::zeros({B, H1, W1, C}, f1.options());
  torch::Tensor g2 = torch::zeros({B, H2, W2, C}, f1.options());
  torch::Tensor gc = torch::zeros({B, S, H1, W1, 2}, f1.options());
  check(dba_altcorr_backward(f1.data_ptr<float>(), f2.data_ptr<float>(), coords.data_ptr<float>(), g.data_ptr<float>(),
                             g1.data_ptr<float>(), g2.data_ptr<float>(), B, S, H1, W1, H2, W2, C, radius, stream_of(fmap1)),
        "dba_altcorr_backward");
  return {g1.to(dt), g2.to(dt), gc};
}

// BACore (src/bacore.h:4-70, src/droid.cpp:311-315): split-phase BA for the GTSAM fusion path
class BACore {
 public:
  void init(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics, torch::Tensor disps_sens, torch::Tensor targets,
            torch::Tensor weights, torch::Tensor eta, torch::Tensor ii, torch::Tensor jj, const int t0, const int t1,
            const int iterations, const float lm, const float ep, const bool motion_only) {
    (void)iterations;   // accepted and ignored, like the reference's init
    (void)motion_only;
    CHECK_INPUT(targets);
    CHECK_INPUT(weights);
    CHECK_INPUT(poses);
    CHECK_INPUT(disps);
    CHECK_INPUT(intrinsics);
    CHECK_INPUT(disps_sens);
    CHECK_INPUT(ii);
    CHECK_INPUT(jj);
    poses_ = poses, disps_ = disps, intrinsics_ = intrinsics, disps_sens_ = disps_sens, targets_ = targets, weights_ = weights;
    eta_ = eta.contiguous(), ii_ = ii, jj_ = jj;
    raise_pending_eta_error();
    d_ = ba_dims(disps, eta_, ii, t0, t1);
    check_eta_bounds(d_);
    lm_ = lm, ep_ = ep;
    ws_ = workspace(d_, poses, &nbytes_, 1);
    ready_ = true;
  }

  // H [6P,6P], v [6P]: caller-owned CPU float64 (depth_video.py:395-396), complete on return
  void hessian(torch::Tensor H, torch::Tensor v) {
    TORCH_CHECK(ready_, "BACore.init must be called first");
    TORCH_CHECK(!H.is_cuda() && !v.is_cuda() && H.scalar_type() == torch::kDouble && v.scalar_type() == torch::kDouble,
                "BACore.hessian: H, v must be CPU float64 tensors (droid_kernels.cu:1889-1890)");
    const int n = 6 * (d_.t1 - d_.t0);
    const bool direct = H.is_contiguous() && v.is_contiguous() && H.dim() == 2 && H.size(0) == n && H.size(1) == n && v.numel() == n;
    torch::Tensor Hh = direct ? H : torch::zeros({n, n}, H.options());
    torch::Tensor vh = direct ? v : torch::zeros({n}, v.options());
    check(dba_bacore_hessian_run(poses_.data_ptr<float>(), disps_.data_ptr<float>(), intrinsics_.data_ptr<float>(),
                             disps_sens_.data_ptr<float>(), targets_.data_ptr<float>(), weights_.data_ptr<float>(),
                             eta_.data_ptr<float>(), d_.eta_rows, ii_.data_ptr<int64_t>(), jj_.data_ptr<int64_t>(), d_.N, d_.B,
                             d_.ht, d_.wd, d_.t0, d_.t1, Hh.data_ptr<double>(), vh.data_ptr<double>(), ws_.data_ptr(), nbytes_,
                             stream_of(poses_), /*prepared=*/2),
          "dba_bacore_hessian");
    {   // the linearisation retract() uses lives in the workspace BACore objects of one window shape share: whose is it now?
      std::lock_guard<std::mutex> lock(owner_mu());
      owner()[ws_.data_ptr()] = this;
    }
    raise_pending_eta_error();   // (hessian synchronises the stream: the verdict of its own stage 0 is in)
    if (!direct) {   // the reference fills H_accessor.size(0) x size(1) entries (:1892-1897)
      H.copy_(Hh.slice(0, 0, H.size(0)).slice(1, 0, H.size(1)));
      v.copy_(vh.slice(0, 0, v.size(0)));
    }
  }

  // hessian() + the caller's stabiliser on the first pose's diagonal + gtsam.BA2GTSAM(H, v, Tbc) (dbaf/depth_video.py:394-401) in
  // one call: A = the 6 x 6 block -Ad(Tbc^-1) with swapped row halves (:21-23, CPU float64); returns the augmented [6P, 6P + 1]
  // matrix [Hg | vg] as a CPU tensor over the library's pinned block (valid until the next hessian call on this window shape)
  torch::Tensor hessian_gtsam(torch::Tensor A, double stabilizer) {
    TORCH_CHECK(ready_, "BACore.init must be called first");
    torch::Tensor Ah = A.to(torch::kCPU, torch::kDouble).contiguous();
    TORCH_CHECK(Ah.numel() == 36, "BACore.hessian_gtsam: A must be the 6 x 6 tangent block");
    const int n = 6 * (d_.t1 - d_.t0);
    double *out = nullptr;
    check(dba_bacore_hessian_host(poses_.data_ptr<float>(), disps_.data_ptr<float>(), intrinsics_.data_ptr<float>(),
                                  disps_sens_.data_ptr<float>(), targets_.data_ptr<float>(), weights_.data_ptr<float>(),
                                  eta_.data_ptr<float>(), d_.eta_rows, ii_.data_ptr<int64_t>(), jj_.data_ptr<int64_t>(), d_.N, d_.B,
                                  d_.ht, d_.wd, d_.t0, d_.t1, ws_.data_ptr(), nbytes_, stream_of(poses_), /*prepared=*/2,
                                  /*layout=*/1, Ah.data_ptr<double>(), stabilizer, &out),
          "dba_bacore_hessian");
    {
      std::lock_guard<std::mutex> lock(owner_mu());
      owner()[ws_.data_ptr()] = this;
    }
    raise_pending_eta_error();
    return torch::from_blob(out, {n, n + 1}, torch::TensorOptions().dtype(torch::kDouble));
  }

  void optimize(torch::Tensor H, torch::Tensor v) {
    TORCH_CHECK(ready_, "BACore.init must be called first");
    torch::Tensor Hh = H.to(torch::kCPU, torch::kDouble).contiguous(), vh = v.to(torch::kCPU, torch::kDouble).contiguous();
    dx_ = torch::zeros({d_.t1 - d_.t0, 6}, poses_.options());
    check(dba_bacore_optimize(Hh.data_ptr<double>(), vh.data_ptr<double>(), d_.N, d_.B, d_.ht, d_.wd, d_.t0, d_.t1, lm_, ep_,
                              dx_.data_ptr<float>(), ws_.data_ptr(), nbytes_, stream_of(poses_)),
          "dba_bacore_optimize");
  }

  std::vector<torch::Tensor> retract(torch::Tensor _dx) {
    TORCH_CHECK(ready_, "BACore.init must be called first");
    {
      std::lock_guard<std::mutex> lock(owner_mu());
      auto it = owner().find(ws_.data_ptr());
      TORCH_CHECK(it == owner().end() || it->second == this,
                  "BACore.retract: another BACore of the same window shape has linearised into the shared workspace since this "
                  "object's hessian(); call hessian() again before retract()");
    }
    torch::Tensor dxh = _dx.to(torch::kCPU, torch::kDouble).contiguous().view({-1});
    const int P = d_.t1 - d_.t0;
    TORCH_CHECK(dxh.numel() >= 6 * P, "BACore.retract: dx must have ", 6 * P, " entries");
    torch::Tensor dx = torch::zeros({P, 6}, poses_.options());
    torch::Tensor dz = torch::empty({std::min(d_.B, P + d_.N), (int64_t)d_.ht * d_.wd}, poses_.options());
    check(dba_bacore_retract(poses_.data_ptr<float>(), disps_.data_ptr<float>(), ii_.data_ptr<int64_t>(), jj_.data_ptr<int64_t>(),
                             d_.N, d_.B, d_.ht, d_.wd, d_.t0, d_.t1, dxh.data_ptr<double>(), dx.data_ptr<float>(),
                             dz.data_ptr<float>(), ws_.data_ptr(), nbytes_, stream_of(poses_)),
          "dba_bacore_retract");
    dx_ = dx;
    return {dx, dz.narrow(0, 0, num_kx(d_, ii_))};
  }

 private:
  static std::mutex &owner_mu() { static auto &m = *new std::mutex; return m; }
  static std::unordered_map<void *, const void *> &owner() { static auto &o = *new std::unordered_map<void *, const void *>; return o; }
  torch::Tensor poses_, disps_, intrinsics_, disps_sens_, targets_, weights_, eta_, ii_, jj_, ws_, dx_;
  BaDims d_{};
  size_t nbytes_ = 0;
  float lm_ = 0.f, ep_ = 0.f;
  bool ready_ = false;
};

}  // namespace

PYBIND11_MODULE(_droid_backends_C, m) {
  m.doc() = "droid_backends for MI355X (compiled adapter over include/dba_hip.h)";
  m.def("version", [] { return std::string(dba_version()); });
  // bundle adjustment kernels (src/droid.cpp:299-302)
  m.def("ba", &ba, "bundle adjustment");
  m.def("ba_clamped", &ba_clamped, "bundle adjustment + the caller's clamp of the inverse depths");
  m.def("gather_edges", &gather_edges, "the caller's edge tensors for one ba call (cat of inactive + active, planar target / weight)",
        pybind11::arg("target_inac"), pybind11::arg("weight_inac"), pybind11::arg("ii_inac"), pybind11::arg("jj_inac"),
        pybind11::arg("sel"), pybind11::arg("target"), pybind11::arg("weight"), pybind11::arg("ii"), pybind11::arg("jj"));
  m.def("check_async_errors", &raise_pending_eta_error, "raises what an earlier asynchronous call found wrong on the device");
  m.def("frame_distance", &frame_distance, "frame_distance");
  m.def("projmap", &projmap, "projmap");
  m.def("depth_filter", &depth_filter, "depth_filter");
  m.def("iproj", &iproj, "back projection");
  // correlation volume kernels (:305-309)
  m.def("altcorr_forward", &altcorr_forward, "ALTCORR forward");
  m.def("altcorr_backward", &altcorr_backward, "ALTCORR backward");
  m.def("corr_index_forward", &corr_index_forward, "INDEX forward");
  m.def("corr_index_backward", &corr_index_backward, "INDEX backward");
  py::class_<BACore>(m, "BACore")
      .def(py::init<>())
      .def("init", &BACore::init)
      .def("hessian", &BACore::hessian)
      .def("hessian_gtsam", &BACore::hessian_gtsam, pybind11::arg("A"), pybind11::arg("stabilizer") = 0.00025)
      .def("optimize", &BACore::optimize)
      .def("retract", &BACore::retract);
}
