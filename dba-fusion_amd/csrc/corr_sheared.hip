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
#include <hip/hip_ext.h>
#include <stdint.h>

#include "common.h"
#include "reproj.h"

#include <type_traits>

#include <cstdlib>

namespace dba {

bool shear_grid(int h1, int w1, int *h1g, int *w1g) {
  static const bool enabled = [] { const char *e = getenv("DBA_SHEAR_TILES"); return !(e && e[0] == '0'); }();
  // (opt-in: measured in the round's last session, the padded grids make the lookups 4-5 % faster and the builds 20-30 % slower --
  // a lookup's cost goes with its 64-column segments, not with its valid pixels; profiles/LOOKUP_NOTES.md)
  static const bool pad = [] { const char *e = getenv("DBA_SHEAR_PAD"); return e && e[0] == '1'; }();
  *h1g = h1, *w1g = w1;
  if (!enabled || h1 <= 0 || w1 <= 0) return false;
  // (maps whose rows are whole 64-pixel segments: the shapes the rows-over-tiles lookup serves; the other forms read tiled
  // planes slower than linear ones because their stores then come in 32-byte pieces)
  const int hg = (h1 + SH_TH - 1) & ~(SH_TH - 1), wg = (w1 + 63) & ~63;
  const bool exact = (hg == h1) && (wg == w1);
  if (!exact && !(pad && (long)hg * wg * 4 <= (long)h1 * w1 * 5)) return false;
  *h1g = hg, *w1g = wg;
  return true;
}

constexpr int SH_MAX_LEVELS = 8;

struct ShLevels {
  const _Float16 *vol[SH_MAX_LEVELS];
};

// Reprojection taken along by the lookup (cflags bit 2, dba_corr_lookup_reproject_sheared): the kernel computes the
// coordinates of its pixels from the poses and the source frame's inverse depths (reproj.h, the arithmetic of
// reproject_kernel) instead of reading them; the waves of pyramid level 0 also write them (and `valid`) out, because the
// caller needs them for the motion features and the BA targets (dbaf/covisible_graph.py:220-221,237).
struct ShReproj {
  const float *poses, *disps, *intr_b4;
  const int64_t *ii, *jj;
  float2 *coords_out;  // [n, h1, w1, 2] or null
  float *valid_out;    // [n, h1, w1, 1] or null
};

// ---- reference layout -> sheared layout, one pyramid level ------------------------------------------
// block = (x1 tile of 64, ty, n*h1 + y1): loads V[n][y1][x1 tile][ty][0..w2l) and writes, for every dx,
// 64 contiguous halves of plane (dy, dx).
__global__ __launch_bounds__(256) void corr_shear_kernel(const _Float16 *__restrict__ V,
                                                         _Float16 *__restrict__ Vs, int h1, int w1, int h2l,
                                                         int w2l, int lvl, int HW1p, const int *__restrict__ src_idx,
                                                         const int *__restrict__ dst_slots, int tiled, int w1g) {
  extern __shared__ _Float16 tile[];  // [64][w2l + 2]
  const int pitch = w2l + 2;
  const int x0 = blockIdx.x * 64;
  const int ty = blockIdx.y;
  const int ey = blockIdx.z;  // n * h1 + y1
  const int y1 = ey % h1, e = ey / h1;
  // (src_idx / dst_slots: edge e of this pass is edge src_idx[e] of V and goes to slot dst_slots[e] of Vs -- the
  // slot-addressed shadows of droid_backends.corr_index_forward re-lay only the edges they have not seen)
  const int es = src_idx ? src_idx[e] : e, ed = dst_slots ? dst_slots[e] : e;
  const int nx = min(64, w1 - x0);
  const size_t plane = (size_t)h2l * w2l;
  for (int idx = threadIdx.x; idx < nx * w2l; idx += blockDim.x) {
    const int xi = idx / w2l, tx = idx - xi * w2l;
    tile[xi * pitch + tx] = V[(((size_t)es * h1 + y1) * w1 + x0 + xi) * plane + (size_t)ty * w2l + tx];
  }
  __syncthreads();
  int dy = ty - (y1 >> lvl);
  dy %= h2l;
  if (dy < 0) dy += h2l;
  const size_t HW1 = (size_t)HW1p;  // planes are padded to a multiple of 64 pixels (see the resident lookup)
  for (int idx = threadIdx.x; idx < nx * w2l; idx += blockDim.x) {
    const int dx = idx / nx, xi = idx - dx * nx;
    const int tx = (((x0 + xi) >> lvl) + dx) % w2l;
    Vs[(((size_t)ed * h2l + dy) * w2l + dx) * HW1 + (size_t)sh_pixel_index(y1, x0 + xi, tiled ? w1g : w1, tiled != 0)] = tile[xi * pitch + tx];
  }
}

// ---- lookup ----------------------------------------------------------------------------------------------
// (wave-wide / 8-lane-group integer min and max on the DPP network: common.h)
struct __attribute__((aligned(16))) Half8v {
  _Float16 v[8];
};
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

// Two lookup kernels live here (the others that were tried -- the row-streaming, pair and band forms -- are shelved as
// scratch/lookup_pruned_forms_r5.diff with their measurements in profiles/LOOKUP_NOTES.md):
//   "rows over tiles"  maps whose rows are whole 64-pixel segments (tiled planes): plane-rows of the union window are
//                      streamed through 2 KB LDS rows per tile, loaders per tile, workers per map row;
//   "resident"         every other map shape (linear planes): the union of a 64-pixel strip's windows is held in LDS.
// Both are bound by VALU issue and memory requests, so the inner steps are written for instruction count: the blend runs on
// channel PAIRS with packed f16 ops (v_pk_mul_f16 / v_pk_add_f16 round each half exactly like the scalar ops), taps arrive
// from LDS already packed (d16 / d16_hi loads) and the odd-aligned pairs come from one v_alignbit each, stores go through a
// scalar base per channel column with one per-lane offset, and all cross-lane set-up runs on DPP.
// Lanes whose window origin is further than SH_BAND from the wave's reference ("outliers": flow discontinuities, pixels
// thrown far away) are not allowed to widen the staged region; they are gathered afterwards.
constexpr int SH_NX = 16;     // plane-rows per step held in LDS (union width in x: 8 + spread <= 16)
// cache-policy bits of the buffer instructions (gfx950: 1 = sc0, 2 = nt, 16 = sc1); 0 = default policy
#ifndef SH_LOAD_AUX
// 2 (nt) is 10 % faster when the same windows are replayed out of the 256 MB Infinity Cache (76 vs 88 us on the 96-edge
// window), the default policy is 6 % faster when they come from HBM (rotating pyramid copies: 94.9 vs 101.0 us; 512
// edges: 517 vs 540 us) -- which is what an update sees, so the default policy it is
#define SH_LOAD_AUX 0
#endif
#ifndef SH_STORE_AUX
#define SH_STORE_AUX 0
#endif
#ifndef SH_BAND_CFG
#define SH_BAND_CFG 4
#endif
constexpr int SH_BAND = SH_BAND_CFG;    // |origin - reference| <= SH_BAND streams
// workgroup barrier that orders LDS traffic only: global loads and stores of the wave stay in flight across it
__device__ __forceinline__ void lds_handoff() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct ShPixel {  // per-pixel lookup state, identical arithmetic in the streaming and the gather phase
  int ix0, iy0, ox, oy;
  bool touches;
  _Float16 h00, h01, h10, h11;
};

// slvl: the level the coordinates still have to be scaled to (lvl, or 0 when the caller passes coords / 2^lvl already, as
// the reference's CorrBlock.__call__ does per level: a float divided by a power of two is the same float either way)
template <int R>
__device__ __forceinline__ ShPixel sh_pixel(float2 c, int lvl, int x1, int y1, int h2l, int w2l, bool active, int slvl) {
  constexpr int WN = 2 * R + 2;
  ShPixel p;
  const float scale = 1.0f / (float)(1 << slvl);
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

// coordinates of pixel p of edge e: [n, h, w, 2] (the reprojection's layout) or planar [n, 2, h, w] (what
// droid_backends.corr_index_forward is handed, corr.py:44-47)
__device__ __forceinline__ float2 sh_coord(const float2 *__restrict__ coords, bool planar, size_t e, int HW1, int p) {
  if (!planar) return coords[e * HW1 + p];
  const float *cf = reinterpret_cast<const float *>(coords) + e * 2 * HW1 + p;
  return make_float2(cf[0], cf[HW1]);
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


// =====================================================================================================================
// Lookup, second form ("resident"): the union of the wave's windows is brought into LDS ONCE, then every lane reads
// its own 8x8 taps from there.
//
//   * a wave = 64 CONSECUTIVE PIXELS of the flattened (y1, x1) index of one edge, one pyramid level; planes of the
//     sheared volume are padded to a multiple of 64 pixels (HW1p), so a strip is one aligned 128-byte line of every plane
//     whatever the map width is (28x107, 55x55 and 48x64 stream exactly like 64x64; a strip may span a row end: the
//     offsets (dy, dx) are per pixel anyway);
//   * staging: exec-masked `buffer_load_dwordx4 ... lds` (LDS-DMA, 16 B per lane, no VGPR round trip, no ds_write) of the
//     union rows; all of a wave's lines are in flight at once, one wait, no per-row hand-shake;
//   * because the whole union is resident, all lanes walk their OWN tap rows j = 0..7 in lock-step: 7 emitting steps
//     instead of 8 + spread, and every store instruction writes one channel for all 64 pixels = one full 128-byte line
//     (the streaming form wrote partial lines whenever lanes sat on different rows);
//     (exchanging channel pairs between neighbouring lanes so that a lane stores 4 bytes -- 4 store instructions per step
//     instead of 7 -- was built and measured 3 % SLOWER: neither form of the kernel is bound by store instructions);
//   * lanes whose window is far from the others (flow discontinuities, strips that span a row end on strongly divergent
//     flow) are taken in further passes of the same code on the remaining lanes (up to SH2_MAXPASS), the rest by a
//     per-lane gather; the union of a pass must fit SH2_LCAP lines of LDS (the band around the pass's reference lane is
//     halved until it does: a single window always fits).
// Arithmetic: identical to the streaming form and to the reference, bit for bit (same packed-f16 blend order).
#ifndef SH2_LCAP_CFG
#define SH2_LCAP_CFG 144  // 96: 138 us, 120: 122 us, 144: 104 us, 160-192: 120 us on the KITTI-shaped window (LDS per wave vs passes)
#endif
#ifndef SH2_WAVES_CFG
#define SH2_WAVES_CFG 2
#endif
#ifndef SH2_MAXPASS_CFG
#define SH2_MAXPASS_CFG 3
#endif
#ifndef SH2_MINOCC_CFG
#define SH2_MINOCC_CFG 2
#endif
constexpr int SH2_LCAP = SH2_LCAP_CFG;      // lines (128 B) of staging per wave
constexpr int SH2_WAVES = SH2_WAVES_CFG;    // independent waves per workgroup
constexpr int SH2_MAXPASS = SH2_MAXPASS_CFG;
constexpr int SH2_WAVE_BYTES = SH2_LCAP * 128 + 1024;  // + 8 zero lines (tap rows of lanes that touch nothing)

// v mod n for |v| < ~2^22, n >= 1 (inv_n = 1.0f / n): float quotient estimate + one correction either way
__device__ __forceinline__ int sh2_mod(int v, int n, float inv_n, bool pow2) {
  if (pow2) return v & (n - 1);
  int q = (int)floorf(((float)v + 0.5f) * inv_n);
  int m = v - q * n;
  m += (m < 0) ? n : 0;
  m -= (m >= n) ? n : 0;
  return m;
}

template <int R>
__global__ __launch_bounds__(SH2_WAVES * 64, SH2_MINOCC_CFG) void corr_lookup_resident_kernel(
    ShLevels L, const float2 *__restrict__ coords, _Float16 *__restrict__ out, int n, int h1, int w1, int h2, int w2,
    int num_levels, int HW1p, float inv_w1, int lvl0, int cflags, const int *__restrict__ slots, ShReproj RP, int tiled, int w1g) {
  constexpr int RD = 2 * R + 1, WN = 2 * R + 2;
  static_assert(WN == 8, "written for radius 3");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int HW1 = h1 * w1;
  const int strips = HW1p >> 6;  // per edge
  // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs; give each XCD a contiguous range of strips
#ifndef SH2_NO_XCD
  const int q8 = (int)(gridDim.x >> 3), r8 = (int)(gridDim.x & 7), xk = (int)(blockIdx.x & 7);
  const int lb = xk * q8 + min(xk, r8) + (int)(blockIdx.x >> 3);
#else
  const int lb = blockIdx.x;
#endif
  const int lvl = blockIdx.y + lvl0;   // (lvl0, cflags: see the streaming kernel)
  const int slvl = (cflags & 2) ? 0 : lvl;
  const bool cplanar = (cflags & 1) != 0;
  const int sid = lb * SH2_WAVES + wave;
  if (sid >= n * strips) return;  // (waves of a workgroup are independent: no barriers below)
  const int e = sid / strips, p0 = (sid - e * strips) << 6;
  const int h2l = h2 >> lvl, w2l = w2 >> lvl;
  const bool pow2 = ((w2l & (w2l - 1)) == 0) && ((h2l & (h2l - 1)) == 0);
  const float inv_w2l = 1.0f / (float)w2l, inv_h2l = 1.0f / (float)h2l;

  unsigned char *const stage = smem + wave * SH2_WAVE_BYTES;   // [row][nxa][64] halves
  unsigned char *const zeros = stage + SH2_LCAP * 128;         // 8 lines of zeros
  {
    u4v z = {0u, 0u, 0u, 0u};
    *reinterpret_cast<u4v *>(zeros + lane * 16) = z;
  }

  const int p = p0 + lane;            // index on the planes' pixel axis (linear or 4 x 16 tiles: common.h)
  int y1, x1;
  bool active;
  if (tiled) {   // tiles of the grid (h1g, w1g >= the map): a pixel of the padding is no pixel (p < HW1p = the grid's size)
    sh_pixel_yx(p, w1g, 0.f, true, y1, x1);
    active = (y1 < h1) && (x1 < w1);
    y1 = active ? y1 : 0;
    x1 = active ? x1 : 0;
  } else {
    active = p < HW1;
    sh_pixel_yx(min(p, HW1 - 1), w1, inv_w1, false, y1, x1);
  }
  const int plin = y1 * w1 + x1;      // the pixel in row-major order: coordinates, inverse depths, outputs
  float2 cxy;
  if (cflags & 4) {  // the reprojection taken along (see ShReproj): this wave's edge geometry, then one pixel per lane
    const int ix = (int)RP.ii[e];
    const float dsrc = RP.disps[(size_t)ix * HW1 + plin];
    const EdgeGeom G = edge_geom(RP.poses, RP.intr_b4, ix, (int)RP.jj[e]);  // uniform
    float ok;
    cxy = reproject_pixel(G, (float)x1, (float)y1, dsrc, ok);
    if (lvl == 0 && active) {
      if (RP.coords_out) RP.coords_out[(size_t)e * HW1 + plin] = cxy;
      if (RP.valid_out) RP.valid_out[(size_t)e * HW1 + plin] = ok;
    }
  } else {
    cxy = sh_coord(coords, cplanar, (size_t)e, HW1, plin);
  }
  const ShPixel P = sh_pixel<R>(cxy, lvl, x1, y1, h2l, w2l, active, slvl);
  const bool touches = P.touches;
  const int ox = P.ox, oy = P.oy;

  _Float16 *obase = out + ((size_t)e * num_levels + blockIdx.y) * RD * RD * HW1;  // this edge, this level: [49][HW1]
  const size_t rowstride = (size_t)w2l * HW1p;                              // elements between consecutive dy
  const unsigned rowbytes = (unsigned)(2 * rowstride);
  const int es = slots ? slots[e] : e;                                      // the edge's slot in the stores (uniform)
  const _Float16 *vedge = L.vol[lvl] + (size_t)es * h2l * rowstride;
  const bool can_stream = ((size_t)h2l * rowbytes < ((size_t)1 << 31)) && ((size_t)RD * RD * HW1 * 2 < ((size_t)1 << 31));
  constexpr unsigned OOR = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc((void *)vedge, 0, can_stream ? (int)((unsigned)h2l * rowbytes) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout =
      __builtin_amdgcn_make_buffer_rsrc((void *)obase, 0, can_stream ? (int)(2u * RD * RD * (unsigned)HW1) : 0, 0x00020000);

  h2v W00, W01, W10, W11;
  W00.x = W00.y = P.h00;
  W01.x = W01.y = P.h01;
  W10.x = W10.y = P.h10;
  W11.x = W11.y = P.h11;

  // validity of this lane's taps (image border): columns i in [ia, ib), rows j in [ja, jb)
  const int ia = max(0, -P.ix0), ib = min(WN, w2l - P.ix0);
  const int ja = max(0, -P.iy0), jb = min(WN, h2l - P.iy0);
  const bool clipped = touches && (ia > 0 || ib < WN || ja > 0 || jb < WN);
  unsigned cm[4];
#pragma unroll
  for (int k = 0; k < 4; k++)
    cm[k] = ((2 * k >= ia && 2 * k < ib) ? 0x0000ffffu : 0u) | ((2 * k + 1 >= ia && 2 * k + 1 < ib) ? 0xffff0000u : 0u);

  unsigned long long todo = can_stream ? __ballot(touches) : 0ull;
  const unsigned long long untouched = __ballot(active && !touches);
  bool first = true;
  const int big = 1 << 28;

  for (int pass = 0; pass < SH2_MAXPASS && (todo != 0ull || (first && untouched != 0ull)); pass++) {
    const bool mine = ((todo >> lane) & 1ull) != 0ull;
    // ---- which lanes this pass takes: all remaining ones if their union fits, else a band around the first of them
    bool in = mine;
    int bx0 = 0, by0 = 0, nxa = 8, ny = 8;
    if (todo != 0ull) {
      const int fl = __ffsll((long long)todo) - 1;
      const int refx = __builtin_amdgcn_readlane(ox, fl), refy = __builtin_amdgcn_readlane(oy, fl);
      int band = 64;  // first try: everything
      for (;;) {
        in = mine && (abs(ox - refx) <= band) && (abs(oy - refy) <= band);
        bx0 = wave_minmax<true>(in ? ox : big);
        by0 = wave_minmax<true>(in ? oy : big);
        const int bx1 = wave_minmax<false>(in ? ox : -big), by1 = wave_minmax<false>(in ? oy : -big);
        nxa = bx1 - bx0 + WN;
        ny = by1 - by0 + WN;
        if (nxa <= 16 && ny <= 16 && nxa * ny <= SH2_LCAP) break;
        band = (band > 4) ? 4 : (band >> 1);  // 64 -> 4 -> 2 -> 1 -> 0; band 0 is one window: 8 x 8 lines
      }
    } else {
      in = false;
    }
    const unsigned long long inmask = __ballot(in);
    const int rx = in ? ox - bx0 : 0, ry = in ? oy - by0 : 0;

    // ---- staging -----------------------------------------------------------------------------------------------
    if (inmask != 0ull) {
      // per 8-lane group (= the 16-byte piece `sub` of every line): range of window origins inside the group
      const int gx0 = group8_minmax<true>(in ? rx : 31), gx1 = group8_minmax<false>(in ? rx : -1);
      const int gy0 = group8_minmax<true>(in ? ry : 31), gy1 = group8_minmax<false>(in ? ry : -1);
      const int packed = (gx0 & 0xff) | ((gx1 & 0xff) << 8) | ((gy0 & 0xff) << 16) | ((gy1 & 0xff) << 24);
      const int sub = lane & 7;
      const int pg = __builtin_amdgcn_ds_bpermute(sub * 32, packed);  // from lane 8 * sub
      const int px0 = (int)(signed char)(pg & 0xff), px1 = (int)(signed char)((pg >> 8) & 0xff);
      const int py0 = (int)(signed char)((pg >> 16) & 0xff), py1 = (int)(signed char)((pg >> 24) & 0xff);
      unsigned goff[2], jlo, jlen[2];
      jlo = (unsigned)py0;
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const int jx = (lane >> 3) + 8 * t;
        const bool act = (px1 >= px0) && (jx >= px0) && (jx < px1 + WN) && (jx < nxa);
        jlen[t] = act ? (unsigned)(py1 - py0 + WN) : 0u;
        const int m = sh2_mod(bx0 + jx, w2l, inv_w2l, pow2);
        goff[t] = 2u * ((unsigned)m * (unsigned)HW1p + (unsigned)p0 + (unsigned)sub * 8u);
      }
      int dym = __builtin_amdgcn_readfirstlane(sh2_mod(by0, h2l, inv_h2l, pow2));
      const bool wide = nxa > 8;
      unsigned ldsrow = (unsigned)(stage - smem);
      const unsigned ldspitch = (unsigned)nxa * 128u;
      for (int row = 0; row < ny; row++) {
        const unsigned soff = (unsigned)dym * rowbytes;
        dym = (dym + 1 == h2l) ? 0 : dym + 1;
#ifndef SH2_ABLATE_LOADS
        if (((unsigned)row - jlo) < jlen[0])
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void *)(smem + ldsrow), 16,
                                                   goff[0], soff, 0, SH_LOAD_AUX);
        if (wide) {
          if (((unsigned)row - jlo) < jlen[1])
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void *)(smem + ldsrow + 1024u),
                                                     16, goff[1], soff, 0, SH_LOAD_AUX);
        }
#endif
        ldsrow += ldspitch;
      }
    }
    // the LDS-DMA writes are tracked by vmcnt; LDS operations of one wave are then in program order
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();

    // ---- compute: tap rows j = 0..7 of every lane in lock-step ------------------------------------------------------
    const bool zero_lane = first && active && !touches;   // exact zeros, produced by the arithmetic (zero weights)
    const bool writes = in || zero_lane;
    const unsigned long long wm = __ballot(writes);
    const bool masked = __ballot(in && clipped) != 0ull;
    const unsigned pix2 = 2u * (unsigned)plin;
    const unsigned tb = in ? (unsigned)(stage - smem) + (unsigned)((ry * nxa + rx) * 128) + 2u * (unsigned)lane
                           : (unsigned)(zeros - smem) + 2u * (unsigned)lane;
    const unsigned radv = in ? (unsigned)nxa * 128u : 0u;
    const unsigned voff = writes ? pix2 : OOR;
    const unsigned chb = 2u * (unsigned)HW1;  // bytes per channel plane

    ShTaps A, B;
    auto load_row = [&](int j, ShTaps &T) {
      sh_read_taps(reinterpret_cast<const _Float16 *>(smem + tb + (unsigned)j * radv), T);
      if (masked) {
        const bool rowok = (j >= ja) && (j < jb);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const unsigned keep = (rowok || !in) ? ((in && clipped) ? cm[k] : 0xffffffffu) : 0u;
          T.e[k] = __builtin_bit_cast(h2v, __builtin_bit_cast(unsigned, T.e[k]) & keep);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const unsigned lo = __builtin_bit_cast(unsigned, T.e[k]);
          const unsigned hi = (k < 3) ? __builtin_bit_cast(unsigned, T.e[k < 3 ? k + 1 : 3]) : 0u;
          T.o[k] = __builtin_bit_cast(h2v, __builtin_amdgcn_alignbit(hi, lo, 16));
        }
      }
    };
    auto emit = [&](int b, const ShTaps &prev, const ShTaps &cur) {  // output row b = j - 1 of all 7 columns a
#pragma unroll
      for (int k = 0; k < 4; k++) {
#ifdef SH2_ABLATE_COMPUTE  // scratch builds: memory traffic only
        const unsigned bits = __builtin_bit_cast(unsigned, W00) + (unsigned)b;
#else
        h2v acc = prev.e[k] * W00;
        acc = acc + cur.e[k] * W01;
        acc = acc + prev.o[k] * W10;
        acc = acc + cur.o[k] * W11;
        const unsigned bits = __builtin_bit_cast(unsigned, acc);
#endif
#ifdef SH2_ABLATE_STORES
        if (k != 0 || b != 0) { asm volatile("" ::"v"(bits)); continue; }
#endif
        const unsigned soff = (unsigned)((2 * k) * RD + b) * chb;  // channel (a = 2k, b), uniform
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)bits, rout, voff, soff, SH_STORE_AUX);
        if (k < 3)
          __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(bits >> 16), rout, voff, soff + (unsigned)RD * chb, SH_STORE_AUX);
      }
    };
#ifdef SH2_ABLATE_COMPUTE
#define load_row(j, T) (void)0
#endif
    if (wm != 0ull) {
      load_row(0, A);
      load_row(1, B); emit(0, A, B);
      load_row(2, A); emit(1, B, A);
      load_row(3, B); emit(2, A, B);
      load_row(4, A); emit(3, B, A);
      load_row(5, B); emit(4, A, B);
      load_row(6, A); emit(5, B, A);
      load_row(7, B); emit(6, A, B);
    }
#ifdef SH2_ABLATE_COMPUTE
#undef load_row
#endif
    __builtin_amdgcn_wave_barrier();  // the next pass overwrites the staging area
    todo &= ~inmask;
    first = false;
  }

  // ---- what no pass took (or could take): per-lane gather straight from the sheared volume ---------------------------
  const bool left = can_stream ? (((todo >> lane) & 1ull) != 0ull) : active;
  if (left) {
    const _Float16 *vol = vedge + p;
    _Float16 *o = obase + plin;
    if (!touches) {
#pragma unroll
      for (int ch = 0; ch < RD * RD; ch++) o[(size_t)ch * HW1] = (_Float16)0.f;
    } else {
      int dxm[WN];
      bool cok[WN];
#pragma unroll
      for (int i = 0; i < WN; i++) {
        dxm[i] = sh2_mod(P.ox + i, w2l, inv_w2l, pow2);
        const int tx = P.ix0 + i;
        cok[i] = (tx >= 0) && (tx < w2l);
      }
      int dym = sh2_mod(P.oy, h2l, inv_h2l, pow2);
      _Float16 prev[WN];
#pragma unroll
      for (int i = 0; i < WN; i++) prev[i] = (_Float16)0.f;
      for (int j = 0; j < WN; j++) {
        const int ty = P.iy0 + j;
        const bool rok = (ty >= 0) && (ty < h2l);
        _Float16 cur[WN];
#pragma unroll
        for (int i = 0; i < WN; i++)
          cur[i] = (rok && cok[i]) ? vol[((size_t)dym * w2l + dxm[i]) * HW1p] : (_Float16)0.f;
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
}


// =====================================================================================================================
// Lookup, fifth form ("rows over tiles"): the streaming form with the READS in tiles and the WRITES in rows.
//
// A workgroup of four waves owns a band of the map -- 4 rows x 64 columns = the four 4 x 16 tiles of one tile row of the
// planes (common.h; maps whose width is a multiple of 64, a band cut into 64-column segments).  Wave w plays two parts in every step of the plane-row walk:
//   LOADER of tile w   its lanes 8 s .. 8 s + 7 are the piece s of every 128-byte line of the tile, it requests the pieces of
//                      the plane-row two steps ahead and stages the arrived one into the tile's 2 KB LDS row;
//   WORKER of row w    lane = x1 of map row 4 band + w; a pixel's taps sit in the LDS row of ITS tile (x1 >> 4) at the slot
//                      of its position in that tile (16 w + (x1 & 15)); the blend and the stores are the streaming form's:
//                      a store instruction writes the lanes' channel (a, b) of ONE map row -- runs of whole lines.
// The tiles' staging rows are double-buffered and one workgroup barrier per step hands them from the loaders to the workers.
// What the loaders need of their tile's pixels (window origins) the workers publish through LDS in the prologue; the walk's
// first row and length are common to the band (the union over the four tiles), each loader only requests the rows and pieces
// its own tile needs.  Same arithmetic, bit for bit.
constexpr int SB_TILES = 4;
#ifndef SB_MIN_WAVES
#define SB_MIN_WAVES 1   // (measured: 6 waves per SIMD with 77 registers beat 7 and 8 with fewer)
#endif
#ifndef SB_DEPTH
#define SB_DEPTH 2
#endif

template <int R>
__global__ __launch_bounds__(256, SB_MIN_WAVES) void corr_lookup_rowtile_kernel(ShLevels L, const float2 *__restrict__ coords,
                                                                  _Float16 *__restrict__ out, int n, int h1, int w1, int h2,
                                                                  int w2, int num_levels, int lvl0, int cflags,
                                                                  const int *__restrict__ slots, ShReproj RP, int h1g, int w1g) {
  constexpr int RD = 2 * R + 1, WN = 2 * R + 2;
  static_assert(WN == 8, "written for radius 3");
  static_assert(SH_TW == 16 && SH_TH == 4, "a band is one row of 4 x 16 tiles");
  __shared__ __attribute__((aligned(16))) _Float16 stage_all[2][SB_TILES][SH_NX * 64];   // [buffer][tile][line][slot]
  __shared__ __attribute__((aligned(16))) _Float16 zero_taps[WN * 64];
  // window origins of the band's pixels, [row][x1]: only the prologue needs them, in the first staging buffer's space
  int(*const org_x)[64] = reinterpret_cast<int(*)[64]>(&stage_all[0][0][0]);
  int(*const org_y)[64] = reinterpret_cast<int(*)[64]>(&stage_all[0][2][0]);
  static_assert(sizeof(stage_all[0][0]) * 2 >= SB_TILES * 64 * sizeof(int), "origins fit in two tile rows");
  __shared__ unsigned long long touch_row[SB_TILES], inl_tile[SB_TILES];
  __shared__ int tinfo[SB_TILES][4];                            // per tile: bx0, by0, by1, any
  __shared__ int olist[256];
  __shared__ int ocount;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // (h1g, w1g): the grid the planes are tiled on -- the map's size rounded up to multiples of (4, 64), common.h; HW1g entries per
  // plane.  Coordinates, inverse depths and outputs keep the map's own [h1, w1] indexing; a pixel of the padding is no pixel.
  const int HW1 = h1 * w1, HW1g = h1g * w1g;
  const int segs = w1g >> 6, units = (h1g >> 2) * segs;   // (w1g % 64 == 0: a band is cut into segments of four tiles)
  const int q8 = (int)(gridDim.x >> 3), r8 = (int)(gridDim.x & 7), xk = (int)(blockIdx.x & 7);
  const int bid = xk * q8 + min(xk, r8) + (int)(blockIdx.x >> 3);
  const int lvl = blockIdx.y + lvl0;
  const int slvl = (cflags & 2) ? 0 : lvl;
  const bool cplanar = (cflags & 1) != 0;
  const int e = bid / units, unit = bid - e * units;
  const int band = unit / segs, seg = unit - band * segs;
  const int h2l = h2 >> lvl, w2l = w2 >> lvl;
  _Float16 *olvl = out + (size_t)blockIdx.y * RD * RD * HW1;
  const size_t estride = (size_t)num_levels * RD * RD * HW1;
  if (threadIdx.x == 0) ocount = 0;
  for (int i = threadIdx.x; i < WN * 64; i += 256) zero_taps[i] = (_Float16)0.f;
  const int xseg = seg << 6;

  // ---- worker pixel: (row 4 band + wave, x1 = lane) ------------------------------------------------------------------------
  const int y1 = (band << 2) + wave, x1 = xseg + lane;
  const bool pvalid = (y1 < h1) && (x1 < w1);
  const unsigned pix = pvalid ? (unsigned)(y1 * w1 + x1) : 0u;
  float2 cxy;
  if (cflags & 4) {
    const int ix = (int)RP.ii[e];
    const float dsrc = RP.disps[(size_t)ix * HW1 + pix];
    const EdgeGeom G = edge_geom(RP.poses, RP.intr_b4, ix, (int)RP.jj[e]);   // uniform
    float ok;
    cxy = reproject_pixel(G, (float)x1, (float)y1, dsrc, ok);
    if (lvl == 0 && pvalid) {
      if (RP.coords_out) RP.coords_out[(size_t)e * HW1 + pix] = cxy;
      if (RP.valid_out) RP.valid_out[(size_t)e * HW1 + pix] = ok;
    }
  } else {
    cxy = sh_coord(coords, cplanar, (size_t)e, HW1, (int)pix);
  }
  const ShPixel P = sh_pixel<R>(cxy, lvl, x1, y1, h2l, w2l, pvalid, slvl);
  const bool touches = P.touches;
  org_x[wave][lane] = P.ox;
  org_y[wave][lane] = P.oy;
  {
    const unsigned long long tm = __ballot(touches);
    if (lane == 0) touch_row[wave] = tm;
  }
  __syncthreads();

  // ---- loader of tile `wave`: its pixels are (row lane >> 4, x1 = 16 wave + (lane & 15)) of the band --------------------------
  const int trow = lane >> 4, tcol = (wave << 4) + (lane & 15);
  const int tox = org_x[trow][tcol], toy = org_y[trow][tcol];
  const bool ttouch = ((touch_row[trow] >> tcol) & 1ull) != 0ull;
  const bool can_stream = ((size_t)h2l * w2l * HW1g * 2 < ((size_t)1 << 31)) && ((size_t)RD * RD * HW1 * 2 < ((size_t)1 << 31));
  bool tinl;
  {
    const unsigned long long tmask = __ballot(ttouch);
    int refx = 0, refy = 0;
    if (tmask) {
      const int first = __ffsll((long long)tmask) - 1, last = 63 - __clzll((long long)tmask);
      const int fxo = __builtin_amdgcn_readlane(tox, first), fyo = __builtin_amdgcn_readlane(toy, first);
      const int lxo = __builtin_amdgcn_readlane(tox, last), lyo = __builtin_amdgcn_readlane(toy, last);
      const bool nearf = ttouch && (abs(tox - fxo) <= SH_BAND) && (abs(toy - fyo) <= SH_BAND);
      const bool nearl = ttouch && (abs(tox - lxo) <= SH_BAND) && (abs(toy - lyo) <= SH_BAND);
      const bool usef = __popcll(__ballot(nearf)) >= __popcll(__ballot(nearl));
      refx = usef ? fxo : lxo;
      refy = usef ? fyo : lyo;
    }
    tinl = can_stream && ttouch && (abs(tox - refx) <= SH_BAND) && (abs(toy - refy) <= SH_BAND);
  }
  const int big = 1 << 28;
  const unsigned long long tinl_mask = __ballot(tinl);
  const int tbx0 = wave_minmax<true>(tinl ? tox : big);
  const int tby0 = wave_minmax<true>(tinl ? toy : big), tby1 = wave_minmax<false>(tinl ? toy : -big);
  if (lane == 0) {
    tinfo[wave][0] = tbx0, tinfo[wave][1] = tby0, tinfo[wave][2] = tby1, tinfo[wave][3] = (tinl_mask != 0ull) ? 1 : 0;
    inl_tile[wave] = tinl_mask;
  }
  __syncthreads();

  // ---- the band's common walk: plane-rows BY0 .. BY0 + ny - 1 ----------------------------------------------------------------
  int BY0 = big, BY1 = -big;
#pragma unroll
  for (int t = 0; t < SB_TILES; t++) {
    if (tinfo[t][3]) {
      BY0 = min(BY0, tinfo[t][1]);
      BY1 = max(BY1, tinfo[t][2]);
    }
  }
  const bool any = BY1 >= BY0;
  const int ny = any ? BY1 - BY0 + WN : 0;   // (no LDS row depends on it: a band of far-apart tiles just walks longer)

  // worker side
  const int mytile = lane >> 4, slot = (wave << 4) | (lane & 15);
  const bool inlier = ((inl_tile[mytile] >> slot) & 1ull) != 0ull;
  const bool outlier = touches && !inlier;
  if (outlier) olist[atomicAdd(&ocount, 1)] = (int)((unsigned)e * (unsigned)HW1 + pix);
  const int rx = inlier ? P.ox - tinfo[mytile][0] : 0, ry = inlier ? P.oy - BY0 : 0;
  _Float16 *obase = olvl + (size_t)e * estride;

  if (!any) {
    if (!outlier && pvalid) {   // nothing streams in this band: untouched pixels are exact zeros
      _Float16 *o = obase + pix;
#pragma unroll
      for (int ch = 0; ch < RD * RD; ch++) o[(size_t)ch * HW1] = (_Float16)0.f;
    }
  } else {
    // ---- loader geometry (the streaming form's, per tile; rows relative to the band's BY0) ---------------------------------
    const bool pow2 = ((w2l & (w2l - 1)) == 0) && ((h2l & (h2l - 1)) == 0);
    const int lrx = tinl ? tox - tbx0 : 0, lry = tinl ? toy - BY0 : 0;
    const int gx0 = group8_minmax<true>(tinl ? lrx : 15), gx1 = group8_minmax<false>(tinl ? lrx : -1);
    const int gy0 = group8_minmax<true>(tinl ? lry : big), gy1 = group8_minmax<false>(tinl ? lry : -1);
    const int sub = lane & 7;
    const int px0 = __builtin_amdgcn_ds_bpermute(sub * 32, gx0), px1 = __builtin_amdgcn_ds_bpermute(sub * 32, gx1);
    const int py0 = __builtin_amdgcn_ds_bpermute(sub * 32, gy0), py1 = __builtin_amdgcn_ds_bpermute(sub * 32, gy1);
    // the piece's 8 pixels: row (sub >> 1) of the tile, x1 from (segment) + 16 wave + 8 (sub & 1)
    const int xs = xseg + (wave << 4) + ((sub & 1) << 3);
    const int ysl = ((band << 2) + (sub >> 1)) >> lvl;
    const int vr0 = max(0, -(ysl + BY0)), vr1 = min(ny, h2l - (ysl + BY0));
    unsigned goff[2], jlo[2], jlen[2], kbits[2];
    int ldsoff[2];
    bool edge_any = false;
    const unsigned p0 = (unsigned)((band * (w1g >> 4) + (seg << 2) + wave) << 6);   // plane index of the tile's first pixel
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int jx = (lane >> 3) + 8 * t;
      const int dxv = tbx0 + jx;
      const bool act = (px1 >= px0) && (jx >= px0) && (jx < px1 + WN);
      const int r0 = max(py0, vr0), r1 = min(py1 + WN, vr1);
      jlo[t] = (unsigned)r0;
      jlen[t] = (act && r1 > r0) ? (unsigned)(r1 - r0) : 0u;
      int m;
      if (pow2) m = dxv & (w2l - 1);
      else { m = dxv % w2l; m += (m < 0) ? w2l : 0; }
      const int lo = (dxv < 0) ? ((-dxv) << lvl) : 0;
      const int hi = (w2l - dxv > 0) ? ((w2l - dxv) << lvl) : 0;
      const int qa = max(0, lo - xs), qb = min(8, hi - xs);
      // which of the piece's 8 pixels have this tap column inside the level's map: one bit each (expanded to a 16-byte mask
      // only in the waves that have such a border at all)
      kbits[t] = (qb > qa) ? ((0xffu >> (8 - (qb - qa))) << qa) : 0u;
      edge_any |= act && (qa > 0 || qb < 8);
      goff[t] = 2u * ((unsigned)m * (unsigned)HW1g + p0 + (unsigned)sub * 8u);
      ldsoff[t] = jx * 64 + sub * 8;
    }
    const bool masked = __ballot(edge_any) != 0ull;
    int dym;
    if (pow2) dym = BY0 & (h2l - 1);
    else { dym = BY0 % h2l; dym += (dym < 0) ? h2l : 0; }
    dym = __builtin_amdgcn_readfirstlane(dym);
    const size_t rowstride = (size_t)w2l * HW1g;
    const unsigned rowbytes = (unsigned)(2 * rowstride);
    constexpr unsigned OOR = 0x80000000u;
    const int es = slots ? slots[e] : e;
    const _Float16 *vedge = L.vol[lvl] + (size_t)es * h2l * rowstride;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)vedge, 0, (int)((unsigned)h2l * rowbytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)obase, 0, (int)(2u * RD * RD * (unsigned)HW1), 0x00020000);

    h2v W00, W01, W10, W11;
    W00.x = W00.y = P.h00;
    W01.x = W01.y = P.h01;
    W10.x = W10.y = P.h10;
    W11.x = W11.y = P.h11;
    const bool writes = !outlier && pvalid;
    // worker's taps: the LDS row of its tile (buffer 0 / 1 alternate), line rx, slot
    const _Float16 *tp0 = inlier ? &stage_all[0][mytile][rx * 64 + slot] : zero_taps + lane;
    const _Float16 *tp1 = inlier ? &stage_all[1][mytile][rx * 64 + slot] : zero_taps + lane;
    _Float16 *const st0 = stage_all[0][wave], *const st1 = stage_all[1][wave];

    auto request = [&](int row, int dy, u4v (&dst)[2]) {
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const bool need = (((unsigned)row - jlo[t]) < jlen[t]);
        dst[t] = __builtin_amdgcn_raw_buffer_load_b128(rin, need ? goff[t] : OOR, (unsigned)dy * rowbytes, SH_LOAD_AUX);
      }
    };
    u4v regs[SB_DEPTH][2];   // ring of SB_DEPTH plane-rows in flight
    int dnext = dym;
#pragma unroll
    for (int r = 0; r < SB_DEPTH; r++) {
      request(r, dnext, regs[r]);
      dnext = (dnext + 1 == h2l) ? 0 : dnext + 1;
    }
    // step jy: loaders stage row jy into buffer jy & 1 and request row jy + SB_DEPTH; barrier; workers read row jy's taps and blend
    // them with row jy - 1's (kept in registers).  Buffer jy & 1 is written again at step jy + 2, after the barrier of step
    // jy + 1, which every worker passes only when it has read row jy.
    auto step = [&](int jy, const ShTaps &prev, ShTaps &cur, auto ring, auto buffer, auto emits) {
      constexpr int r = decltype(ring)::value, bf = decltype(buffer)::value;
      constexpr bool EMITS = decltype(emits)::value;
      _Float16 *const st = bf ? st1 : st0;
#pragma unroll
      for (int t = 0; t < 2; t++) {
        u4v v = regs[r][t];
        if (masked) {
#pragma unroll
          for (int d = 0; d < 4; d++) {
            const unsigned l_ = (unsigned)__builtin_amdgcn_sbfe((int)kbits[t], 2 * d, 1) & 0x0000ffffu;
            const unsigned h_ = (unsigned)__builtin_amdgcn_sbfe((int)kbits[t], 2 * d + 1, 1) & 0xffff0000u;
            v[d] &= (l_ | h_);
          }
        }
        *reinterpret_cast<u4v *>(&st[ldsoff[t]]) = v;
      }
      request(jy + SB_DEPTH, dnext, regs[r]);
      dnext = (dnext + 1 == h2l) ? 0 : dnext + 1;
      lds_handoff();
      sh_read_taps(bf ? tp1 : tp0, cur);
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
          const unsigned col = 2u * (unsigned)(2 * k * RD) * (unsigned)HW1;
          __builtin_amdgcn_raw_buffer_store_b16((unsigned short)bits, rout, voff, col, SH_STORE_AUX);
          if (k < 3)
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(bits >> 16), rout, voff, col + 2u * RD * (unsigned)HW1, SH_STORE_AUX);
        }
      }
    };
    ShTaps A, B;
#pragma unroll
    for (int k = 0; k < 4; k++) A.e[k] = A.o[k] = B.e[k] = B.o[k] = (h2v)((_Float16)0.f);
    // unrolled over lcm(2, SB_DEPTH) rows: ring slot, staging buffer and the roles of the two tap-row sets are compile-time
    constexpr int UNR = (SB_DEPTH % 2 == 0) ? SB_DEPTH : 2 * SB_DEPTH;
    auto body = [&](int jy, auto uc, auto emits) {
      constexpr int u = decltype(uc)::value;
      if constexpr (u % 2 == 0) step(jy + u, B, A, std::integral_constant<int, u % SB_DEPTH>{}, std::integral_constant<int, 0>{}, emits);
      else step(jy + u, A, B, std::integral_constant<int, u % SB_DEPTH>{}, std::integral_constant<int, 1>{}, emits);
    };
    auto group = [&](int jy, auto first_emits) {
      body(jy, std::integral_constant<int, 0>{}, first_emits);
      body(jy, std::integral_constant<int, 1>{}, std::true_type{});
      if constexpr (UNR > 2) body(jy, std::integral_constant<int, 2>{}, std::true_type{});
      if constexpr (UNR > 3) body(jy, std::integral_constant<int, 3>{}, std::true_type{});
      if constexpr (UNR > 4) body(jy, std::integral_constant<int, 4>{}, std::true_type{});
      if constexpr (UNR > 5) body(jy, std::integral_constant<int, 5>{}, std::true_type{});
      static_assert(UNR <= 6, "SB_DEPTH 2, 3 or 4");
    };
    group(0, std::false_type{});   // row 0 only loads taps; ny >= 8 > UNR.  (ny is uniform over the workgroup: every wave
    for (int jy = UNR; jy < ny; jy += UNR) group(jy, std::true_type{});   // meets every barrier)
  }

  // ---- gather phase: the band's outliers straight from the planes ---------------------------------------------------------
  __syncthreads();
  const int cnt = ocount;
  for (int t = threadIdx.x; t < cnt; t += 256) {
    const int gp = olist[t];
    const int gx = gp % w1, gey = gp / w1;
    const int gy = gey % h1, ge = gey / h1;
    float2 c;
    if (cflags & 4) {
      const int ix = (int)RP.ii[ge];
      float ok;
      c = reproject_pixel(edge_geom(RP.poses, RP.intr_b4, ix, (int)RP.jj[ge]), (float)gx, (float)gy,
                          RP.disps[(size_t)ix * HW1 + gy * w1 + gx], ok);
    } else {
      c = sh_coord(coords, cplanar, (size_t)ge, HW1, gy * w1 + gx);
    }
    const ShPixel Q = sh_pixel<R>(c, lvl, gx, gy, h2l, w2l, true, slvl);
    const int ges = slots ? slots[ge] : ge;
    const _Float16 *vol = L.vol[lvl] + (size_t)ges * h2l * w2l * HW1g + (size_t)sh_pixel_index(gy, gx, w1g, true);
    _Float16 *o = olvl + (size_t)ge * estride + (size_t)gy * w1 + gx;
    int dxm[WN];
    bool cok[WN];
#pragma unroll
    for (int i = 0; i < WN; i++) {
      int m = (Q.ox + i) % w2l;
      m += (m < 0) ? w2l : 0;
      dxm[i] = m;
      const int tx = Q.ix0 + i;
      cok[i] = (tx >= 0) && (tx < w2l);
    }
    int dym = Q.oy % h2l;
    dym += (dym < 0) ? h2l : 0;
    _Float16 prev[WN];
#pragma unroll
    for (int i = 0; i < WN; i++) prev[i] = (_Float16)0.f;
    for (int j = 0; j < WN; j++) {
      const int ty = Q.iy0 + j;
      const bool rok = (ty >= 0) && (ty < h2l);
      _Float16 cur[WN];
#pragma unroll
      for (int i = 0; i < WN; i++) cur[i] = (rok && cok[i]) ? vol[((size_t)dym * w2l + dxm[i]) * HW1g] : (_Float16)0.f;
      if (j >= 1) {
#pragma unroll
        for (int a = 0; a < RD; a++) o[(size_t)(a * RD + (j - 1)) * HW1] = sh_blend(prev[a], cur[a], prev[a + 1], cur[a + 1], Q);
      }
#pragma unroll
      for (int i = 0; i < WN; i++) prev[i] = cur[i];
      dym = (dym + 1 == h2l) ? 0 : dym + 1;
    }
  }
}

}  // namespace dba

using namespace dba;

// 0 = automatic, 2 = resident form, 5 = rows over tiles (initialised from DBA_LOOKUP_KERNEL = resident | rowtile; the
// numbers are those of round 4's C ABI, whose forms 1, 3 and 4 no longer ship)
static std::atomic<int> g_lookup_select{[] {
  const char *e = getenv("DBA_LOOKUP_KERNEL");
  return (e && e[0] == 'r' && e[1] == 'e') ? 2 : (e && e[0] == 'r' && e[1] == 'o') ? 5 : 0;
}()};

// events armed for the next lookup launch of this thread (dba_corr_lookup_arm_timing)
static thread_local hipEvent_t g_time_start = nullptr, g_time_stop = nullptr;

extern "C" {

int dba_corr_lookup_arm_timing(void *start_event, void *stop_event) {
  g_time_start = (hipEvent_t)start_event;
  g_time_stop = (hipEvent_t)stop_event;
  return DBA_OK;
}

int dba_corr_lookup_select(int kernel) {
  if (kernel != 0 && kernel != 2 && kernel != 5) return DBA_ERR_ARG;
  g_lookup_select.store(kernel, std::memory_order_relaxed);
  return DBA_OK;
}

int dba_corr_sheared_tiled(int h1, int w1) { return shear_tiled(h1, w1) ? SH_TW : 0; }

int dba_corr_sheared_grid(int h1, int w1, int *h1g, int *w1g) {
  int a = h1, b = w1;
  const bool t = shear_grid(h1, w1, &a, &b);
  if (h1g) *h1g = a;
  if (w1g) *w1g = b;
  return t ? SH_TW : 0;
}

int dba_corr_sheared_plane_elems(int h1, int w1) {
  if (h1 <= 0 || w1 <= 0) return 0;
  int hg, wg;
  if (shear_grid(h1, w1, &hg, &wg)) return hg * wg;   // (a multiple of 64)
  return (h1 * w1 + 63) / 64 * 64;
}

int dba_corr_shear_level(const void *ref_level, void *sheared_level, int n, int h1, int w1, int h2l, int w2l,
                         int lvl, dba_stream_t stream) {
  return dba_corr_shear_level_slots(ref_level, sheared_level, nullptr, nullptr, n, h1, w1, h2l, w2l, lvl, stream);
}

int dba_corr_shear_level_slots(const void *ref_level, void *sheared_store, const int *src_idx, const int *dst_slots, int n,
                               int h1, int w1, int h2l, int w2l, int lvl, dba_stream_t stream) {
  if (n < 0 || h1 <= 0 || w1 <= 0 || h2l <= 0 || w2l <= 0 || lvl < 0) return DBA_ERR_ARG;
  if (n == 0) return DBA_OK;
  if ((long)n * h1 > 2147483647L / 1) return DBA_ERR_ARG;
  const size_t lds = (size_t)64 * (w2l + 2) * sizeof(_Float16);
  if (lds > 64 * 1024) return DBA_ERR_UNSUPPORTED;
  dim3 grid((w1 + 63) / 64, h2l, n * h1);
  int h1g_, w1g_;
  const bool tiled_ = shear_grid(h1, w1, &h1g_, &w1g_);
  hipLaunchKernelGGL(corr_shear_kernel, grid, dim3(256), lds, (hipStream_t)stream,
                     static_cast<const _Float16 *>(ref_level), static_cast<_Float16 *>(sheared_store), h1, w1, h2l,
                     w2l, lvl, dba_corr_sheared_plane_elems(h1, w1), src_idx, dst_slots, tiled_ ? 1 : 0, w1g_);
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

// levels [lvl0, lvl0 + nlv) of the pyramid -> corr [n, nlv, 49, h1, w1]
static int lookup_sheared_launch(const ShLevels &L, const float *coords, void *corr, int n, int h1, int w1, int h2, int w2,
                                 int lvl0, int nlv, int cflags, dba_stream_t stream, const int *slots = nullptr,
                                 const ShReproj &RP = ShReproj{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}) {
  if ((long)n * h1 * w1 >= 2147483647L) return DBA_ERR_UNSUPPORTED;
  const int HW1p = dba_corr_sheared_plane_elems(h1, w1);
  // "rows over tiles" on tiled planes (maps whose rows are whole 64-pixel segments, common.h), "resident" on every other
  // shape and on planes too large for the row walk's 32-bit offsets; dba_corr_lookup_select() / DBA_LOOKUP_KERNEL override.
  int h1g, w1g;
  const int tiled = shear_grid(h1, w1, &h1g, &w1g) ? 1 : 0;   // the planes' pixel order (common.h): a wave owns a 4 x 16 tile of the grid
  const int sel = g_lookup_select.load(std::memory_order_relaxed);
  hipEvent_t e0 = g_time_start, e1 = g_time_stop;
  g_time_start = g_time_stop = nullptr;
  const bool rowtile_ok = tiled && SH_TW == 16 && (w1g & 63) == 0 && (size_t)(h2 >> lvl0) * (w2 >> lvl0) * HW1p * 2 < ((size_t)1 << 31);
  if ((sel == 5 || sel == 0) && rowtile_ok) {   // loaders per tile, workers per map row
    dim3 grid((unsigned)((long)n * (h1g / 4) * (w1g / 64)), nlv);
    hipExtLaunchKernelGGL((corr_lookup_rowtile_kernel<3>), grid, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, L,
                          reinterpret_cast<const float2 *>(coords), static_cast<_Float16 *>(corr), n, h1, w1, h2, w2, nlv, lvl0,
                          cflags, slots, RP, h1g, w1g);
    DBA_LAUNCH_CHECK();
    return DBA_OK;
  }
  const long strips = (long)n * (HW1p / 64);
  dim3 grid((unsigned)((strips + SH2_WAVES - 1) / SH2_WAVES), nlv);
  const size_t lds = (size_t)SH2_WAVES * SH2_WAVE_BYTES;
  if (lds > 64 * 1024) {
    static DeviceOnce attr_once;
    if (attr_once.needed()) {
      DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&corr_lookup_resident_kernel<3>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_once.done();
    }
  }
  hipExtLaunchKernelGGL((corr_lookup_resident_kernel<3>), grid, dim3(SH2_WAVES * 64), lds, (hipStream_t)stream, e0, e1, 0, L,
                        reinterpret_cast<const float2 *>(coords), static_cast<_Float16 *>(corr), n, h1, w1, h2, w2,
                        nlv, HW1p, 1.0f / (float)w1, lvl0, cflags, slots, RP, tiled, w1g);
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
  if ((h2 >> (num_levels - 1)) < 1 || (w2 >> (num_levels - 1)) < 1) return DBA_ERR_ARG;
  ShLevels L;
  for (int l = 0; l < SH_MAX_LEVELS; l++) L.vol[l] = (l < num_levels) ? static_cast<const _Float16 *>(volumes[l]) : nullptr;
  return lookup_sheared_launch(L, coords_nhw2, corr, n, h1, w1, h2, w2, 0, num_levels, 0, stream);
}

int dba_corr_lookup_pyramid_sheared_slots(const void *const *volumes, const int *slots, const float *coords_nhw2, void *corr,
                                          int n, int h1, int w1, int h2, int w2, int num_levels, int radius,
                                          dba_stream_t stream) {
  if (n < 0 || h1 <= 0 || w1 <= 0 || h2 <= 0 || w2 <= 0 || num_levels < 1 || num_levels > SH_MAX_LEVELS)
    return DBA_ERR_ARG;
  if (radius != 3) return DBA_ERR_UNSUPPORTED;
  if (n == 0) return DBA_OK;
  if (!volumes || !coords_nhw2 || !corr) return DBA_ERR_ARG;
  if ((h2 >> (num_levels - 1)) < 1 || (w2 >> (num_levels - 1)) < 1) return DBA_ERR_ARG;
  ShLevels L;
  for (int l = 0; l < SH_MAX_LEVELS; l++) L.vol[l] = (l < num_levels) ? static_cast<const _Float16 *>(volumes[l]) : nullptr;
  return lookup_sheared_launch(L, coords_nhw2, corr, n, h1, w1, h2, w2, 0, num_levels, 0, stream, slots);
}

// The motion filter's unit (dbaf/motion_filter.py:74-76): a pyramid built, looked up once and dropped, as one call
size_t dba_corr_once_pyramid_bytes(int n, int h1, int w1, int h2, int w2, int num_levels) {
  if (n <= 0 || h1 <= 0 || w1 <= 0 || h2 <= 0 || w2 <= 0 || num_levels < 1 || num_levels > SH_MAX_LEVELS) return 0;
  const size_t hw1p = (size_t)dba_corr_sheared_plane_elems(h1, w1);
  size_t total = 0;
  for (int l = 0; l < num_levels; l++) total += align_up((size_t)n * (h2 >> l) * (w2 >> l) * hw1p * sizeof(_Float16), 256);
  return total;
}

int dba_corr_build_lookup_once_sheared(const void *fmap1, const void *fmap2, const float *coords_nhw2, void *corr, void *pyramid,
                                       size_t pyramid_bytes, void *scratch, size_t scratch_bytes, int n, int C, int h1, int w1,
                                       int h2, int w2, int num_levels, int radius, dba_stream_t stream) {
  if (n < 0 || num_levels < 1 || num_levels > SH_MAX_LEVELS) return DBA_ERR_ARG;
  if (radius != 3) return DBA_ERR_UNSUPPORTED;
  if (n == 0) return DBA_OK;
  if (!pyramid || pyramid_bytes < dba_corr_once_pyramid_bytes(n, h1, w1, h2, w2, num_levels)) return DBA_ERR_WORKSPACE;
  void *levels[SH_MAX_LEVELS];
  const size_t hw1p = (size_t)dba_corr_sheared_plane_elems(h1, w1);
  char *p = static_cast<char *>(pyramid);
  for (int l = 0; l < num_levels; l++) {
    levels[l] = p;
    p += align_up((size_t)n * (h2 >> l) * (w2 >> l) * hw1p * sizeof(_Float16), 256);
  }
  const int rc = dba_corr_volume_build_sheared_slots(fmap1, fmap2, levels, nullptr, n, C, h1, w1, h2, w2, num_levels, scratch,
                                                     scratch_bytes, stream);
  if (rc != DBA_OK) return rc;
  return dba_corr_lookup_pyramid_sheared_slots(levels, nullptr, coords_nhw2, corr, n, h1, w1, h2, w2, num_levels, radius, stream);
}

int dba_corr_lookup_reproject_sheared(const void *const *volumes, const int *slots, const float *poses, const float *disps,
                                      const float *intrinsics_b4, const int64_t *ii, const int64_t *jj, float *coords_out,
                                      float *valid_out, void *corr, int n, int h1, int w1, int h2, int w2, int num_levels,
                                      int radius, dba_stream_t stream) {
  if (n < 0 || h1 <= 0 || w1 <= 0 || h2 <= 0 || w2 <= 0 || num_levels < 1 || num_levels > SH_MAX_LEVELS)
    return DBA_ERR_ARG;
  if (radius != 3) return DBA_ERR_UNSUPPORTED;
  if (n == 0) return DBA_OK;
  if (!volumes || !poses || !disps || !intrinsics_b4 || !ii || !jj || !corr) return DBA_ERR_ARG;
  if ((h2 >> (num_levels - 1)) < 1 || (w2 >> (num_levels - 1)) < 1) return DBA_ERR_ARG;
  ShLevels L;
  for (int l = 0; l < SH_MAX_LEVELS; l++) L.vol[l] = (l < num_levels) ? static_cast<const _Float16 *>(volumes[l]) : nullptr;
  const ShReproj RP{poses, disps, intrinsics_b4, ii, jj, reinterpret_cast<float2 *>(coords_out), valid_out};
  return lookup_sheared_launch(L, nullptr, corr, n, h1, w1, h2, w2, 0, num_levels, 4, stream, slots, RP);
}

int dba_corr_lookup_level_sheared(const void *sheared_level, const float *coords_n2hw_scaled, void *corr, int n, int h1,
                                  int w1, int h2, int w2, int lvl, int radius, dba_stream_t stream) {
  return dba_corr_lookup_level_sheared_slots(sheared_level, nullptr, coords_n2hw_scaled, corr, n, h1, w1, h2, w2, lvl, radius,
                                             stream);
}

int dba_corr_lookup_level_sheared_slots(const void *sheared_level, const int *slots, const float *coords_n2hw_scaled,
                                        void *corr, int n, int h1, int w1, int h2, int w2, int lvl, int radius,
                                        dba_stream_t stream) {
  if (n < 0 || h1 <= 0 || w1 <= 0 || h2 <= 0 || w2 <= 0 || lvl < 0 || lvl >= SH_MAX_LEVELS) return DBA_ERR_ARG;
  if (radius != 3) return DBA_ERR_UNSUPPORTED;
  if (n == 0) return DBA_OK;
  if (!sheared_level || !coords_n2hw_scaled || !corr || (h2 >> lvl) < 1 || (w2 >> lvl) < 1) return DBA_ERR_ARG;
  ShLevels L;
  for (int l = 0; l < SH_MAX_LEVELS; l++) L.vol[l] = (l == lvl) ? static_cast<const _Float16 *>(sheared_level) : nullptr;
  return lookup_sheared_launch(L, coords_n2hw_scaled, corr, n, h1, w1, h2, w2, lvl, 1, 3, stream, slots);
}

}  // extern "C"
