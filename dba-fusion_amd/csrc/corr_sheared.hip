// Flow-aligned ("sheared") correlation volume and its lookup, gfx950.
//
// Why: in the reference layout [n, y1, x1, y2, x2] (dbaf/modules/corr.py:31-36) every source pixel owns a
// private (h2, w2) plane, so the (2r+2)^2 window of one pixel is 8 separate 16-byte pieces and the 64 lanes
// of a wave touch 512 different cache lines per level: HBM moves 64-128 B for every 16 B used and the
// L1/TA path sees ~20 requests per pixel and level (measured: 1.1 TB/s algorithmic).  Neighbouring source
// pixels do look at neighbouring target pixels though (the flow is smooth), which the layout below turns
// into contiguity:
//
//     Vs_l[n][dy][dx][y1][x1] = V_l[n][y1][x1][ty][tx],  dy = (ty - (y1 >> l)) mod h2l,
//                                                        dx = (tx - (x1 >> l)) mod w2l
//
// (a bijection for fixed (y1, x1), same size as the reference tensor).  A wave that owns 64 consecutive x1
// of one row needs, for tap offset (dy, dx), 64 consecutive halves = ONE full 128-byte line; the union over
// the wave's lanes of the needed (dy, dx) is (8 + spread)^2 lines, fetched with 16-byte loads into LDS,
// from which every lane then picks its own 64 taps.  The arithmetic after that is the reference's
// (correlation_kernels.cu:55-65), bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

#include <type_traits>

namespace dba {

constexpr int SH_MAX_LEVELS = 8;

struct ShLevels {
  const _Float16 *vol[SH_MAX_LEVELS];
};

// ---- reference layout -> sheared layout, one pyramid level ------------------------------------------
// block = (x1 tile of 64, ty, n*h1 + y1): loads V[n][y1][x1 tile][ty][0..w2l) and writes, for every dx,
// 64 contiguous halves of plane (dy, dx).
__global__ __launch_bounds__(256) void corr_shear_kernel(const _Float16 *__restrict__ V,
                                                         _Float16 *__restrict__ Vs, int h1, int w1, int h2l,
                                                         int w2l, int lvl) {
  extern __shared__ _Float16 tile[];  // [64][w2l + 2]
  const int pitch = w2l + 2;
  const int x0 = blockIdx.x * 64;
  const int ty = blockIdx.y;
  const int ey = blockIdx.z;  // n * h1 + y1
  const int y1 = ey % h1, e = ey / h1;
  const int nx = min(64, w1 - x0);
  const size_t plane = (size_t)h2l * w2l;
  for (int idx = threadIdx.x; idx < nx * w2l; idx += blockDim.x) {
    const int xi = idx / w2l, tx = idx - xi * w2l;
    tile[xi * pitch + tx] = V[((size_t)ey * w1 + x0 + xi) * plane + (size_t)ty * w2l + tx];
  }
  __syncthreads();
  int dy = ty - (y1 >> lvl);
  dy %= h2l;
  if (dy < 0) dy += h2l;
  const size_t HW1 = (size_t)h1 * w1;
  for (int idx = threadIdx.x; idx < nx * w2l; idx += blockDim.x) {
    const int dx = idx / nx, xi = idx - dx * nx;
    const int tx = (((x0 + xi) >> lvl) + dx) % w2l;
    Vs[(((size_t)e * h2l + dy) * w2l + dx) * HW1 + (size_t)y1 * w1 + x0 + xi] = tile[xi * pitch + tx];
  }
}

// ---- lookup ----------------------------------------------------------------------------------------------
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

struct __attribute__((aligned(16))) Half8v {
  _Float16 v[8];
};
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

// Lookup kernel.  A workgroup is SH_WAVES independent waves; a wave = 64 consecutive x1 of one source row, one
// pyramid level.
//
// Streaming form: the wave walks the plane-rows dy = by0 .. by1+7 of the union window one at a time.  Step jy
// brings the nx (<= SH_NX) 128-byte lines (dy, bx0 .. bx0+nx-1) into a 2 KB LDS row (16-B loads issued one step
// ahead, so they are in flight while the previous row is consumed); a lane whose own window starts ry rows
// into the union reads its 8 taps of tap-row j = jy - ry, combines them with the previous tap-row it kept in
// registers, and emits the 7 outputs (a, b = j-1).  LDS per wave is 2 KB, so occupancy is bounded by registers
// (8 waves/SIMD), not by staging space.
//
// The kernel is bound by VALU issue (a wave64 op holds a 16-lane SIMD for 4 cycles), so the inner step is
// written for instruction count: the blend runs on channel PAIRS with packed f16 ops (v_pk_mul_f16 /
// v_pk_add_f16 round each half exactly like the scalar ops), taps arrive from LDS already packed
// (d16 / d16_hi loads) and the odd-aligned pairs come from one v_alignbit each, the two tap-row register sets
// alternate roles instead of being copied, stores go through a scalar base per channel column with one
// per-lane offset, staging loads use a scalar row base that advances on the scalar unit, and all cross-lane
// set-up runs on DPP.
//
// Lanes whose window origin is further than SH_BAND from the wave's reference ("outliers": flow
// discontinuities, pixels thrown far away, and every lane of a ragged tile) are not allowed to widen the
// streamed region; they are appended to a workgroup-wide list and gathered afterwards with all 64 lanes of
// a wave busy, instead of each wave paying a full gather pass for its one or two odd pixels.
constexpr int SH_NX = 16;     // plane-rows per step held in LDS (union width in x: 8 + spread <= 16)
constexpr int SH_NY = 72;     // longest union in y walked by the streaming path
// cache-policy bits of the buffer instructions (gfx950: 1 = sc0, 2 = nt, 16 = sc1); 0 = default policy
#ifndef SH_LOAD_AUX
#define SH_LOAD_AUX 2  // nt: every window is read once; -10 % (82 -> 74 us) on the 96-edge lookup
#endif
#ifndef SH_STORE_AUX
#define SH_STORE_AUX 0
#endif
#ifndef SH_BAND_CFG
#define SH_BAND_CFG 4
#endif
constexpr int SH_BAND = SH_BAND_CFG;    // |origin - reference| <= SH_BAND streams
#ifndef SH_WAVES_CFG
#define SH_WAVES_CFG 8
#endif
#ifndef SH_DEPTH_CFG
#define SH_DEPTH_CFG 2
#endif
#ifndef SH_OCC_CFG
#define SH_OCC_CFG 8
#endif
constexpr int SH_WAVES = SH_WAVES_CFG;   // waves (source rows) per workgroup
constexpr int SH_BLOCK = SH_WAVES * 64;
constexpr int SH_DEPTH = SH_DEPTH_CFG;   // plane-rows in flight per wave (registers: 8 VGPRs per row)
constexpr int SH_OCC = SH_OCC_CFG;       // waves per SIMD the register budget is set for

struct ShPixel {  // per-pixel lookup state, identical arithmetic in the streaming and the gather phase
  int ix0, iy0, ox, oy;
  bool touches;
  _Float16 h00, h01, h10, h11;
};

template <int R>
__device__ __forceinline__ ShPixel sh_pixel(float2 c, int lvl, int x1, int y1, int h2l, int w2l, bool active) {
  constexpr int WN = 2 * R + 2;
  ShPixel p;
  const float scale = 1.0f / (float)(1 << lvl);
  const float x0 = c.x * scale, y0 = c.y * scale;
  const float fx = floorf(x0), fy = floorf(y0);
  const float dx = x0 - fx, dy = y0 - fy;
  // a pixel whose whole window is out of bounds contributes zeros and must not widen the staged region
  p.ix0 = (int)fmaxf(fminf(fx, 1.0e6f), -1.0e6f) - R;
  p.iy0 = (int)fmaxf(fminf(fy, 1.0e6f), -1.0e6f) - R;
  const bool finite = (fabsf(x0) < 1.0e6f) && (fabsf(y0) < 1.0e6f);
  p.touches = active && finite && (p.ix0 + WN > 0) && (p.ix0 < w2l) && (p.iy0 + WN > 0) && (p.iy0 < h2l);
  p.ox = p.ix0 - (x1 >> lvl);
  p.oy = p.iy0 - (y1 >> lvl);
  // scalar_t(dx * dy): f32 product rounded to half (see corr_lookup.hip).
  float w00 = (1.0f - dx) * (1.0f - dy), w01 = (1.0f - dx) * dy, w10 = dx * (1.0f - dy), w11 = dx * dy;
  if (!p.touches) w00 = w01 = w10 = w11 = 0.f;  // nothing in bounds (incl. NaN / inf coords): exact zeros
  asm volatile("" : "+v"(w00), "+v"(w01), "+v"(w10), "+v"(w11));
  p.h00 = (_Float16)w00;
  p.h01 = (_Float16)w01;
  p.h10 = (_Float16)w10;
  p.h11 = (_Float16)w11;
  return p;
}

// c10::Half `a * b` / `a + b` compute in float and round to half; for two halves that is exactly the IEEE half
// operation (the float product is exact; a float sum rounded to half cannot double-round because
// 24 >= 2*11 + 2), so native v_mul_f16 / v_add_f16 (and their packed forms) are bit-identical.  No fusion:
// -ffp-contract=off.
// Order: tap(a,b)*w00, tap(a,b+1)*w01, tap(a+1,b)*w10, tap(a+1,b+1)*w11 (correlation_kernels.cu:55-65).
__device__ __forceinline__ _Float16 sh_blend(_Float16 p0, _Float16 c0, _Float16 p1, _Float16 c1, const ShPixel &p) {
  _Float16 acc = p0 * p.h00;
  acc = acc + c0 * p.h01;
  acc = acc + p1 * p.h10;
  acc = acc + c1 * p.h11;
  return acc;
}

// one tap-row of a lane, as channel pairs: e[k] = (tap 2k, tap 2k+1), o[k] = (tap 2k+1, tap 2k+2)
struct ShTaps {
  h2v e[4], o[4];
};

__device__ __forceinline__ void sh_read_taps(const _Float16 *tp, ShTaps &T) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    h2v v;
    v.x = tp[(2 * k) * 64];
    v.y = tp[(2 * k + 1) * 64];
    T.e[k] = v;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const unsigned lo = __builtin_bit_cast(unsigned, T.e[k]);
    const unsigned hi = (k < 3) ? __builtin_bit_cast(unsigned, T.e[k < 3 ? k + 1 : 3]) : 0u;
    T.o[k] = __builtin_bit_cast(h2v, __builtin_amdgcn_alignbit(hi, lo, 16));
  }
}

template <int R>
__global__ __launch_bounds__(SH_BLOCK, SH_OCC) void corr_lookup_sheared_kernel(ShLevels L,
                                                                          const float2 *__restrict__ coords,
                                                                          _Float16 *__restrict__ out, int n,
                                                                          int h1, int w1, int h2, int w2,
                                                                          int num_levels) {
  constexpr int RD = 2 * R + 1, WN = 2 * R + 2;
  static_assert(WN == 8, "the streaming lookup is written for radius 3");
  __shared__ __attribute__((aligned(16))) _Float16 stage_all[SH_WAVES][SH_NX * 64];
  __shared__ __attribute__((aligned(16))) _Float16 zero_taps[WN * 64];  // tap rows of lanes that touch nothing
  __shared__ u4v keep_all[SH_WAVES][2][64];  // per staging slot: 16-bit keep mask per half (image-border pieces)
  __shared__ int olist[SH_BLOCK];
  __shared__ int ocount;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  _Float16 *stage = stage_all[wave];
  const int xtiles = (w1 + 63) / 64;
  const int HW1 = h1 * w1;
#ifndef SH_NO_XCD_SWIZZLE
  // workgroups are dealt round-robin to the 8 XCDs: give each XCD a contiguous range of rows (whole edges), so that
  // the 64 rows of a channel plane are written (and a plane's lines read) through ONE L2
  // (XCD k receives the workgroups k, k + 8, ...: q + (k < r) of them for gridDim.x = 8 q + r; a bijection)
  const int q8 = (int)(gridDim.x >> 3), r8 = (int)(gridDim.x & 7), xk = (int)(blockIdx.x & 7);
  const int lb = xk * q8 + min(xk, r8) + (int)(blockIdx.x >> 3);
  const int lvl = blockIdx.y;  // (levels stay apart in the dispatch order: interleaving them cost 30 %)
  const int rowid = lb * SH_WAVES + wave;
#else
  const int lvl = blockIdx.y;
  const int rowid = blockIdx.x * SH_WAVES + wave;  // (e * h1 + y1) * xtiles + xt
#endif
  const int h2l = h2 >> lvl, w2l = w2 >> lvl;
  const bool rowvalid = rowid < n * h1 * xtiles;
  _Float16 *olvl = out + (size_t)lvl * RD * RD * HW1;  // + e * num_levels * RD * RD * HW1 + pixel
  const size_t estride = (size_t)num_levels * RD * RD * HW1;
  if (threadIdx.x == 0) ocount = 0;
  for (int i = threadIdx.x; i < WN * 64; i += SH_BLOCK) zero_taps[i] = (_Float16)0.f;
  __syncthreads();

  if (rowvalid) {
    const int xt = rowid % xtiles;
    const int ey = rowid / xtiles;  // e * h1 + y1
    const int y1 = ey % h1, e = ey / h1;
    const int x1 = xt * 64 + lane;
    const bool active = x1 < w1;
    const int x1c = min(x1, w1 - 1);
    const ShPixel P = sh_pixel<R>(coords[(size_t)ey * w1 + x1c], lvl, x1c, y1, h2l, w2l, active);
    const bool touches = P.touches;
    const int ox = P.ox, oy = P.oy, sy = y1 >> lvl;
    _Float16 *obase = olvl + (size_t)e * estride;     // uniform
    const unsigned pix = (unsigned)(y1 * w1 + x1);   // this lane's pixel inside a channel plane

    // (the streaming path addresses one edge's level through a buffer resource: 31-bit byte range)
    const bool can_stream = ((w1 & 7) == 0) && (xt * 64 + 64 <= w1) &&
                            ((size_t)h2l * w2l * HW1 * 2 < ((size_t)1 << 31)) && ((size_t)RD * RD * HW1 * 2 < ((size_t)1 << 31));
    const unsigned long long tmask = __ballot(touches);
    int refx = 0, refy = 0;
    if (tmask) {
      const int first = __ffsll((long long)tmask) - 1, last = 63 - __clzll((long long)tmask);
      const int fxo = __builtin_amdgcn_readlane(ox, first), fyo = __builtin_amdgcn_readlane(oy, first);
      const int lxo = __builtin_amdgcn_readlane(ox, last), lyo = __builtin_amdgcn_readlane(oy, last);
      const bool nearf = touches && (abs(ox - fxo) <= SH_BAND) && (abs(oy - fyo) <= SH_BAND);
      const bool nearl = touches && (abs(ox - lxo) <= SH_BAND) && (abs(oy - lyo) <= SH_BAND);
      const bool usef = __popcll(__ballot(nearf)) >= __popcll(__ballot(nearl));
      refx = usef ? fxo : lxo;
      refy = usef ? fyo : lyo;
    }
    const bool inlier = can_stream && touches && (abs(ox - refx) <= SH_BAND) && (abs(oy - refy) <= SH_BAND);
    const bool outlier = touches && !inlier;
    if (outlier) olist[atomicAdd(&ocount, 1)] = ey * w1 + x1;  // <= 64 per wave: the list cannot overflow

    const int big = 1 << 28;
    const bool any = __ballot(inlier) != 0ull;

    if (!any) {
      // nothing streams in this wave: lanes that touch nothing are exact zeros, outliers are written later
      if (active && !outlier) {
        _Float16 *o = obase + pix;
#pragma unroll
        for (int ch = 0; ch < RD * RD; ch++) o[(size_t)ch * HW1] = (_Float16)0.f;
      }
    } else {
      const int bx0 = wave_minmax<true>(inlier ? ox : big);
      const int by0 = wave_minmax<true>(inlier ? oy : big), by1 = wave_minmax<false>(inlier ? oy : -big);
      const int ny = min(by1 - by0 + WN, SH_NY);  // (the union is at most 16 x 16 by construction)
      // window origin of this lane inside the union (0 for lanes that stream nothing: they read row 0 with zero
      // weights and emit exact zeros -- except outliers, whose outputs are left to the gather phase)
      const int rx = inlier ? ox - bx0 : 0, ry = inlier ? oy - by0 : 0;
      // per 8-lane group (= one 16-byte piece of a line): the range of window origins inside the group, packed
      // into one word so that a single cross-lane permute hands it to the lanes that fetch the group's pieces
      const int gx0 = group8_minmax<true>(inlier ? rx : 15), gx1 = group8_minmax<false>(inlier ? rx : -1);
      const int gy0 = group8_minmax<true>(inlier ? ry : 15), gy1 = group8_minmax<false>(inlier ? ry : -1);
      const int packed = (gx0 & 0xff) | ((gx1 & 0xff) << 8) | ((gy0 & 0xff) << 16) | ((gy1 & 0xff) << 24);
      const int sub = lane & 7;  // 16-byte piece (fixed for both staging slots of a lane)
      const int pg = __builtin_amdgcn_ds_bpermute(sub * 32, packed);  // from lane 8 * sub
      const int px0 = (int)(signed char)(pg & 0xff), px1 = (int)(signed char)((pg >> 8) & 0xff);
      const int py0 = (int)(signed char)((pg >> 16) & 0xff), py1 = (int)(signed char)((pg >> 24) & 0xff);
      const bool pow2 = ((w2l & (w2l - 1)) == 0) && ((h2l & (h2l - 1)) == 0);

      // staging slots: lane + 64 t -> plane-row jx = (lane >> 3) + 8 t of the union, piece sub
      const int xs = xt * 64 + sub * 8;  // first x1 of this piece
      unsigned goff[2], jlo[2], jlen[2];
      int ldsoff[2];
      u4v *keep = &keep_all[wave][0][lane];  // [t * 64]: parked in LDS, 8 VGPRs less in the streaming loop
      bool edge_any = false;
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const int jx = (lane >> 3) + 8 * t;
        const int dxv = bx0 + jx;
        // (fetching the whole union box instead of the per-piece ranges, ~30 % more bytes, changes nothing: DESIGN 4.1)
        const bool act = (px1 >= px0) && (jx >= px0) && (jx < px1 + WN);
        jlo[t] = (unsigned)py0;                 // plane-rows [py0, py1 + WN) are read by this piece's lanes
        jlen[t] = act ? (unsigned)(py1 - py0 + WN) : 0u;
        int m;
        if (pow2) m = dxv & (w2l - 1);
        else { m = dxv % w2l; m += (m < 0) ? w2l : 0; }
        // valid q: 0 <= ((xs + q) >> lvl) + dxv < w2l  <=>  qa <= q < qb
        const int lo = (dxv < 0) ? ((-dxv) << lvl) : 0;
        const int hi = (w2l - dxv > 0) ? ((w2l - dxv) << lvl) : 0;
        const int qa = max(0, lo - xs), qb = min(8, hi - xs);
        // 16-bit keep mask per half of the piece
        u4v k;
#pragma unroll
        for (int d = 0; d < 4; d++) {
          const unsigned l_ = (2 * d >= qa && 2 * d < qb) ? 0x0000ffffu : 0u;
          const unsigned h_ = (2 * d + 1 >= qa && 2 * d + 1 < qb) ? 0xffff0000u : 0u;
          k[d] = l_ | h_;
        }
        keep[t * 64] = k;
        edge_any |= act && (qa > 0 || qb < 8);
        goff[t] = 2u * ((unsigned)m * (unsigned)HW1 + (unsigned)xs);  // bytes inside one plane-row dy
        ldsoff[t] = jx * 64 + sub * 8;
      }
      const bool masked = __ballot(edge_any) != 0ull;  // some piece of this wave straddles the image border
      int dym;
      if (pow2) dym = by0 & (h2l - 1);
      else { dym = by0 % h2l; dym += (dym < 0) ? h2l : 0; }
      dym = __builtin_amdgcn_readfirstlane(dym);
      const size_t rowstride = (size_t)w2l * HW1;  // elements between consecutive dy
      const unsigned rowbytes = (unsigned)(2 * rowstride);

      // Buffer resources (raw, bounds-checked): a lane that must not load / store presents an out-of-range offset,
      // which the memory pipeline drops (loads return zeros).  No branches and no exec changes around the memory
      // instructions, so every step issues the same instruction sequence and the waits on the loads that were
      // issued SH_DEPTH steps earlier are exact counts instead of "everything outstanding".
      constexpr unsigned OOR = 0x80000000u;
      const _Float16 *vedge = L.vol[lvl] + (size_t)e * h2l * rowstride;  // uniform: this edge, this level
      const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
          (void *)(vedge + (size_t)y1 * w1), 0, (int)((unsigned)h2l * rowbytes - 2u * (unsigned)(y1 * w1)), 0x00020000);
      const __amdgpu_buffer_rsrc_t rout =
          __builtin_amdgcn_make_buffer_rsrc((void *)obase, 0, (int)(2u * RD * RD * (unsigned)HW1), 0x00020000);

      // packed weights and per-lane emission state
      h2v W00, W01, W10, W11;
      W00.x = W00.y = P.h00;
      W01.x = W01.y = P.h01;
      W10.x = W10.y = P.h10;
      W11.x = W11.y = P.h11;
      const bool writes = active && !outlier;
      // lanes that touch nothing blend zero taps with zero weights: exact zeros without a select per channel pair
      const _Float16 *tp = touches ? stage + rx * 64 + lane : zero_taps + lane;

      auto request = [&](int row, int dy, u4v (&dst)[2]) {  // pieces of plane-row `row` of the union (dy = its plane)
        const int ty = sy + by0 + row;  // uniform
        const bool rowok = (row < ny) && (ty >= 0) && (ty < h2l);
#pragma unroll
        for (int t = 0; t < 2; t++) {
          const bool need = rowok && (((unsigned)row - jlo[t]) < jlen[t]);  // row in [jlo, jlo + jlen)
          dst[t] = __builtin_amdgcn_raw_buffer_load_b128(rin, need ? goff[t] : OOR, (unsigned)dy * rowbytes, SH_LOAD_AUX);
        }
      };

      // ring of SH_DEPTH plane-rows in flight: regs[r] is receiving the pieces of row jy with jy % D == r
      u4v regs[SH_DEPTH][2];
      int dnext = dym;  // plane (mod h2l) of the next row to request
#pragma unroll
      for (int r = 0; r < SH_DEPTH; r++) {
        request(r, dnext, regs[r]);
        dnext = (dnext + 1 == h2l) ? 0 : dnext + 1;
      }

      // one plane-row: stage it, request the row SH_DEPTH ahead, read this lane's taps into `cur`, blend with `prev`
      auto step = [&](int jy, const ShTaps &prev, ShTaps &cur, auto ring, auto emits) {
        constexpr int r = decltype(ring)::value;
        constexpr bool EMITS = decltype(emits)::value;  // false for row 0: no lane has a previous tap-row yet
#pragma unroll
        for (int t = 0; t < 2; t++) {
          u4v v = regs[r][t];
          if (masked) v &= keep[t * 64];
          *reinterpret_cast<u4v *>(&stage[ldsoff[t]]) = v;  // rows / pieces nobody needs arrive as zeros
        }
#ifndef SH_ABLATE_LOADS
        request(jy + SH_DEPTH, dnext, regs[r]);
#endif
        dnext = (dnext + 1 == h2l) ? 0 : dnext + 1;
        // LDS operations of one wave execute in program order: no wait between the row's writes and the tap reads
        __builtin_amdgcn_wave_barrier();
        sh_read_taps(tp, cur);
        if constexpr (EMITS) {
          const int j = jy - ry;
          const bool emit = writes && ((unsigned)(j - 1) < (unsigned)RD);
          const unsigned voff = emit ? 2u * (pix + (unsigned)(j - 1) * (unsigned)HW1) : OOR;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            h2v acc = prev.e[k] * W00;
            acc = acc + cur.e[k] * W01;
            acc = acc + prev.o[k] * W10;
            acc = acc + cur.o[k] * W11;
            const unsigned bits = __builtin_bit_cast(unsigned, acc);
            const unsigned col = 2u * (unsigned)(2 * k * RD) * (unsigned)HW1;  // bytes to channel column a = 2k (uniform)
#ifdef SH_ABLATE_STORES  // ablation builds only (scratch/): keep the value live, store one channel
            if (k == 0) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)bits, rout, voff, col, SH_STORE_AUX); else asm volatile("" ::"v"(bits));
#else
            // (the high half goes through an explicit shift: handing the builtin `acc.y` directly makes this compiler
            // store the LOW half of the packed register)
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)bits, rout, voff, col, SH_STORE_AUX);
            if (k < 3)
              __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(bits >> 16), rout, voff, col + 2u * RD * (unsigned)HW1, SH_STORE_AUX);
#endif
          }
        }
        __builtin_amdgcn_wave_barrier();
      };

      ShTaps A, B;
#pragma unroll
      for (int k = 0; k < 4; k++) A.e[k] = A.o[k] = B.e[k] = B.o[k] = (h2v)((_Float16)0.f);
      // unrolled over lcm(2, SH_DEPTH) rows: the two tap-row register sets alternate roles and the ring slot of a
      // row is a compile-time index.  Rows past ny are requested out of range and emit nothing.
      constexpr int UNR = (SH_DEPTH % 2 == 0) ? SH_DEPTH : 2 * SH_DEPTH;
      auto body = [&](int jy, auto uc, auto emits) {
        constexpr int u = decltype(uc)::value;
        if constexpr (u % 2 == 0) step(jy + u, B, A, std::integral_constant<int, u % SH_DEPTH>{}, emits);
        else step(jy + u, A, B, std::integral_constant<int, u % SH_DEPTH>{}, emits);
      };
      auto group = [&](int jy, auto first_emits) {
        body(jy, std::integral_constant<int, 0>{}, first_emits);
        if constexpr (UNR > 1) body(jy, std::integral_constant<int, 1>{}, std::true_type{});
        if constexpr (UNR > 2) body(jy, std::integral_constant<int, 2>{}, std::true_type{});
        if constexpr (UNR > 3) body(jy, std::integral_constant<int, 3>{}, std::true_type{});
        if constexpr (UNR > 4) body(jy, std::integral_constant<int, 4>{}, std::true_type{});
        if constexpr (UNR > 5) body(jy, std::integral_constant<int, 5>{}, std::true_type{});
        static_assert(UNR <= 6, "SH_DEPTH up to 4");
      };
      group(0, std::false_type{});  // peeled: row 0 only loads taps (no lane has a previous tap-row yet); ny >= 8 > UNR
      for (int jy = UNR; jy < ny; jy += UNR) group(jy, std::true_type{});
    }
  }

  // ---- gather phase: the workgroup's outliers, 64 per wave, straight from the sheared volume ---------------
  __syncthreads();
  const int cnt = ocount;
  for (int t = threadIdx.x; t < cnt; t += SH_BLOCK) {
    const int pix = olist[t];  // (e * h1 + y1) * w1 + x1
    const int x1 = pix % w1, ey = pix / w1;
    const int y1 = ey % h1, e = ey / h1;
    const ShPixel P = sh_pixel<R>(coords[pix], lvl, x1, y1, h2l, w2l, true);
    const _Float16 *vol = L.vol[lvl] + (size_t)e * h2l * w2l * HW1 + (size_t)y1 * w1 + x1;
    _Float16 *o = olvl + (size_t)e * estride + (size_t)y1 * w1 + x1;
    int dxm[WN];
    bool cok[WN];
#pragma unroll
    for (int i = 0; i < WN; i++) {
      int m = (P.ox + i) % w2l;
      m += (m < 0) ? w2l : 0;
      dxm[i] = m;
      const int tx = P.ix0 + i;
      cok[i] = (tx >= 0) && (tx < w2l);
    }
    int dym = P.oy % h2l;
    dym += (dym < 0) ? h2l : 0;
    _Float16 prev[WN];
#pragma unroll
    for (int i = 0; i < WN; i++) prev[i] = (_Float16)0.f;
    for (int j = 0; j < WN; j++) {
      const int ty = P.iy0 + j;
      const bool rok = (ty >= 0) && (ty < h2l);
      _Float16 cur[WN];
#pragma unroll
      for (int i = 0; i < WN; i++)
        cur[i] = (rok && cok[i]) ? vol[((size_t)dym * w2l + dxm[i]) * HW1] : (_Float16)0.f;
      if (j >= 1) {
#pragma unroll
        for (int a = 0; a < RD; a++)
          o[(size_t)(a * RD + (j - 1)) * HW1] = sh_blend(prev[a], cur[a], prev[a + 1], cur[a + 1], P);
      }
#pragma unroll
      for (int i = 0; i < WN; i++) prev[i] = cur[i];
      dym = (dym + 1 == h2l) ? 0 : dym + 1;
    }
  }
}

}  // namespace dba

using namespace dba;

extern "C" {

int dba_corr_shear_level(const void *ref_level, void *sheared_level, int n, int h1, int w1, int h2l, int w2l,
                         int lvl, dba_stream_t stream) {
  if (n < 0 || h1 <= 0 || w1 <= 0 || h2l <= 0 || w2l <= 0 || lvl < 0) return DBA_ERR_ARG;
  if (n == 0) return DBA_OK;
  if ((long)n * h1 > 2147483647L / 1) return DBA_ERR_ARG;
  const size_t lds = (size_t)64 * (w2l + 2) * sizeof(_Float16);
  if (lds > 64 * 1024) return DBA_ERR_UNSUPPORTED;
  dim3 grid((w1 + 63) / 64, h2l, n * h1);
  hipLaunchKernelGGL(corr_shear_kernel, grid, dim3(256), lds, (hipStream_t)stream,
                     static_cast<const _Float16 *>(ref_level), static_cast<_Float16 *>(sheared_level), h1, w1, h2l,
                     w2l, lvl);
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

int dba_corr_lookup_pyramid_sheared(const void *const *volumes, const float *coords_nhw2, void *corr, int n, int h1,
                                    int w1, int h2, int w2, int num_levels, int radius, dba_stream_t stream) {
  if (n < 0 || h1 <= 0 || w1 <= 0 || h2 <= 0 || w2 <= 0 || num_levels < 1 || num_levels > SH_MAX_LEVELS)
    return DBA_ERR_ARG;
  if (radius != 3) return DBA_ERR_UNSUPPORTED;
  if (n == 0) return DBA_OK;
  if (!volumes || !coords_nhw2 || !corr) return DBA_ERR_ARG;
  ShLevels L;
  for (int l = 0; l < SH_MAX_LEVELS; l++) L.vol[l] = (l < num_levels) ? static_cast<const _Float16 *>(volumes[l]) : nullptr;
  const int xtiles = (w1 + 63) / 64;
  if ((long)n * h1 * w1 >= 2147483647L) return DBA_ERR_UNSUPPORTED;
  const long rows = (long)n * h1 * xtiles;
  dim3 grid((unsigned)((rows + SH_WAVES - 1) / SH_WAVES), num_levels);
  hipLaunchKernelGGL((corr_lookup_sheared_kernel<3>), grid, dim3(SH_BLOCK), 0, (hipStream_t)stream, L,
                     reinterpret_cast<const float2 *>(coords_nhw2), static_cast<_Float16 *>(corr), n, h1, w1, h2, w2,
                     num_levels);
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

}  // extern "C"
