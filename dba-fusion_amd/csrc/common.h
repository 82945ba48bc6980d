// Shared device helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/dba_hip.h"

namespace dba {

constexpr int WAVE = 64;

// ---- host-side error plumbing -------------------------------------------------------------
void set_last_error(const char *what, hipError_t e);

#define DBA_HIP_CHECK(expr)                                \
  do {                                                     \
    hipError_t e_ = (expr);                                \
    if (e_ != hipSuccess) {                                \
      ::dba::set_last_error(#expr, e_);                    \
      return DBA_ERR_HIP;                                  \
    }                                                      \
  } while (0)

#define DBA_LAUNCH_CHECK() DBA_HIP_CHECK(hipGetLastError())

// Per-device one-time set-up (hipFuncSetAttribute applies to the CURRENT device only, and several host threads may
// drive different devices): `DeviceOnce once; if (once.needed()) { ...set attributes...; once.done(); }`.  The flag of
// a device is raised only after its set-up has completed, so a concurrent caller repeats the (idempotent) set-up
// instead of launching without it.
struct DeviceOnce {
  std::atomic<unsigned long long> mask{0};
  static unsigned long long bit() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;  // unknown device: always set up
    return 1ull << dev;
  }
  bool needed() const { const unsigned long long b = bit(); return b == 0 || !(mask.load(std::memory_order_acquire) & b); }
  void done() { mask.fetch_or(bit(), std::memory_order_release); }
};

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- wave64 cross-lane reductions with DPP ------------------------------------------------
// Sum over the 64 lanes of a wavefront.  row_shr 1,2,4,8 fold each 16-lane row into its lane 15,
// row_bcast:15 / row_bcast:31 carry the row totals across rows; the total lands in lane 63.
// (DPP modifiers fuse into v_add_f32, so one value costs six VALU instructions and no LDS traffic.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float x) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, false);
  return x + __int_as_float(moved);
}

__device__ __forceinline__ float wave_sum_to_lane63(float x) {
  x = dpp_add<0x111, 0xf>(x);  // row_shr:1
  x = dpp_add<0x112, 0xf>(x);  // row_shr:2
  x = dpp_add<0x114, 0xf>(x);  // row_shr:4
  x = dpp_add<0x118, 0xf>(x);  // row_shr:8
  x = dpp_add<0x142, 0xa>(x);  // row_bcast:15 -> rows 1,3
  x = dpp_add<0x143, 0xc>(x);  // row_bcast:31 -> rows 2,3
  return x;
}

// total broadcast to every lane (via SGPR)
__device__ __forceinline__ float wave_sum(float x) {
  x = wave_sum_to_lane63(x);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

// Place the lane-63 value of `reduced` into lane `LANE` of `dst` (v_readlane + v_cndmask; clang has no
// writelane builtin and hazards inside inline asm are not padded by the compiler).
template <int LANE>
__device__ __forceinline__ float deposit_lane63(float dst, float reduced, int lane) {
  const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(reduced), 63));
  return (lane == LANE) ? s : dst;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// wave-wide / 8-lane-group integer min and max on the DPP network (fused into v_min/v_max: one VALU op per
// step, no LDS traffic, no address arithmetic)
template <int CTRL, int ROW_MASK, bool IS_MIN>
__device__ __forceinline__ int dpp_minmax(int x) {
  const int moved = __builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, 0xf, false);  // lanes without a source keep x
  return IS_MIN ? min(x, moved) : max(x, moved);
}
template <bool IS_MIN>
__device__ __forceinline__ int wave_minmax(int x) {
  x = dpp_minmax<0x111, 0xf, IS_MIN>(x);  // row_shr:1
  x = dpp_minmax<0x112, 0xf, IS_MIN>(x);  // row_shr:2
  x = dpp_minmax<0x114, 0xf, IS_MIN>(x);  // row_shr:4
  x = dpp_minmax<0x118, 0xf, IS_MIN>(x);  // row_shr:8
  x = dpp_minmax<0x142, 0xa, IS_MIN>(x);  // row_bcast:15
  x = dpp_minmax<0x143, 0xc, IS_MIN>(x);  // row_bcast:31
  return __builtin_amdgcn_readlane(x, 63);
}
template <bool IS_MIN>
__device__ __forceinline__ int group8_minmax(int x) {  // result in all 8 lanes of the group
  x = dpp_minmax<0xB1, 0xf, IS_MIN>(x);   // quad_perm [1,0,3,2]
  x = dpp_minmax<0x4E, 0xf, IS_MIN>(x);   // quad_perm [2,3,0,1]
  x = dpp_minmax<0x141, 0xf, IS_MIN>(x);  // row_half_mirror: the other quad of the group
  return x;
}


// ---- pixel order of the flow-aligned volumes' pixel axis ----------------------------------------------------------------
// A 128-byte line of a flow-aligned plane holds 64 source pixels at ONE tap offset, and a lookup reads the union of those
// pixels' windows: the fewer distinct window origins among the 64 pixels, the fewer lines.  The flow varies with DISTANCE,
// so 64 pixels should be close together: a 4 x 16 tile of the map instead of a 64 x 1 strip of a row (bench scene: 79
// instead of 92 lines per wave and level; the lookup's time is proportional to that number, profiles/r04_lookup_lines.txt).
//   tiled:   p = ((y1 >> 2) * (w1g >> 4) + (x1 >> 4)) * 64 + (y1 & 3) * 16 + (x1 & 15) on the GRID (h1g, w1g) = the map's size
//            rounded up to multiples of (4, 64): maps whose rows are whole 64-pixel segments exactly, and -- opt-in, DBA_SHEAR_PAD=1
//            (round 6, last session) -- maps that reach such a grid with at most 25 % of padding: 28 x 107 on 28 x 128, 55 x 55 on
//            56 x 64.  The pad pixels' entries of a plane are never read as taps and never returned: coordinates, inverse depths
//            and the returned tensor keep the map's own [h1, w1] indexing.  Measured: lookups 4-5 % faster, builds 20-30 % slower
//            (profiles/LOOKUP_NOTES.md), hence not the default
//   linear (any other map):  p = y1 * w1 + x1, planes padded to a multiple of 64 pixels
// Which one a map shape gets is decided here and nowhere else (build, re-layout and lookup kernels all ask shear_grid);
// DBA_SHEAR_TILES=0 (read once per process) keeps every shape linear.
bool shear_grid(int h1, int w1, int *h1g, int *w1g);   // true: tiled, on the grid (h1g, w1g); false: linear, (h1g, w1g) = (h1, w1)
inline bool shear_tiled(int h1, int w1) { int a, b; return shear_grid(h1, w1, &a, &b); }
#ifndef SH_TILE_WLOG
#define SH_TILE_WLOG 4   // tile width 16 (x 4 rows); 5: 2 x 32, 3: 8 x 8 (experiments)
#endif
constexpr int SH_TW_LOG = SH_TILE_WLOG, SH_TW = 1 << SH_TW_LOG, SH_TH_LOG = 6 - SH_TW_LOG, SH_TH = 1 << SH_TH_LOG;

__host__ __device__ __forceinline__ int sh_pixel_index(int y1, int x1, int w1, bool tiled) {
  return tiled ? (((y1 >> SH_TH_LOG) * (w1 >> SH_TW_LOG) + (x1 >> SH_TW_LOG)) << 6) + ((y1 & (SH_TH - 1)) << SH_TW_LOG) + (x1 & (SH_TW - 1))
               : y1 * w1 + x1;
}
// plane index p -> (y1, x1); inv_w1 = 1.0f / w1 (linear order: float quotient + one correction either way)
__device__ __forceinline__ void sh_pixel_yx(int p, int w1, float inv_w1, bool tiled, int &y1, int &x1) {
  if (tiled) {
    const int t = p >> 6, tiles_x = w1 >> SH_TW_LOG;
    const int tyi = (int)(((float)t + 0.5f) / (float)tiles_x);   // (tiles_x <= a few hundred: exact)
    const int txi = t - tyi * tiles_x;
    y1 = (tyi << SH_TH_LOG) + ((p >> SH_TW_LOG) & (SH_TH - 1));
    x1 = (txi << SH_TW_LOG) + (p & (SH_TW - 1));
  } else {
    y1 = (int)(((float)p + 0.5f) * inv_w1);
    x1 = p - y1 * w1;
    if (x1 < 0) { y1--; x1 += w1; }
    if (x1 >= w1) { y1++; x1 -= w1; }
  }
}

// ---- SE3 helpers on (t, q_xyzw) -----------------------------------------------------------
struct Rot3 {
  float r[9];  // row-major
};

__device__ __forceinline__ Rot3 quat_to_rot(const float *q) {
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  Rot3 R;
  // v + 2 w (q x v) + 2 q x (q x v), expanded (not assuming |q| = 1 would differ only at O(|q|^2-1))
  R.r[0] = 1.f - 2.f * (y * y + z * z);
  R.r[1] = 2.f * (x * y - w * z);
  R.r[2] = 2.f * (x * z + w * y);
  R.r[3] = 2.f * (x * y + w * z);
  R.r[4] = 1.f - 2.f * (x * x + z * z);
  R.r[5] = 2.f * (y * z - w * x);
  R.r[6] = 2.f * (x * z - w * y);
  R.r[7] = 2.f * (y * z + w * x);
  R.r[8] = 1.f - 2.f * (x * x + y * y);
  return R;
}

// rotate v by quaternion q (same formula as the reference's actSO3, droid_kernels.cu:61-71)
__device__ __forceinline__ void quat_rotate(const float *q, const float *X, float *Y) {
  const float uv0 = 2.f * (q[1] * X[2] - q[2] * X[1]);
  const float uv1 = 2.f * (q[2] * X[0] - q[0] * X[2]);
  const float uv2 = 2.f * (q[0] * X[1] - q[1] * X[0]);
  Y[0] = X[0] + q[3] * uv0 + (q[1] * uv2 - q[2] * uv1);
  Y[1] = X[1] + q[3] * uv1 + (q[2] * uv0 - q[0] * uv2);
  Y[2] = X[2] + q[3] * uv2 + (q[0] * uv1 - q[1] * uv0);
}

// Gij = Tj * Ti^-1 (droid_kernels.cu:99-110)
__device__ __forceinline__ void rel_pose(const float *Pi, const float *Pj, float *tij, float *qij) {
  const float *ti = Pi, *qi = Pi + 3, *tj = Pj, *qj = Pj + 3;
  qij[0] = -qj[3] * qi[0] + qj[0] * qi[3] - qj[1] * qi[2] + qj[2] * qi[1];
  qij[1] = -qj[3] * qi[1] + qj[1] * qi[3] - qj[2] * qi[0] + qj[0] * qi[2];
  qij[2] = -qj[3] * qi[2] + qj[2] * qi[3] - qj[0] * qi[1] + qj[1] * qi[0];
  qij[3] = qj[3] * qi[3] + qj[0] * qi[0] + qj[1] * qi[1] + qj[2] * qi[2];
  float rt[3];
  quat_rotate(qij, ti, rt);
  tij[0] = tj[0] - rt[0];
  tij[1] = tj[1] - rt[1];
  tij[2] = tj[2] - rt[2];
}

// relative pose of an edge, with the stereo special case (ix == jx) of droid_kernels.cu:263-273
__device__ __forceinline__ void edge_pose(const float *poses, int ix, int jx, float *tij, float *qij) {
  if (ix == jx) {
    tij[0] = -0.1f; tij[1] = 0.f; tij[2] = 0.f;
    qij[0] = 0.f; qij[1] = 0.f; qij[2] = 0.f; qij[3] = 1.f;
  } else {
    rel_pose(poses + 7 * ix, poses + 7 * jx, tij, qij);
  }
}

}  // namespace dba
