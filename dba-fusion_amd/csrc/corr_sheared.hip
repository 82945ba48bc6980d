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
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}

struct __attribute__((aligned(16))) Half8v {
  _Float16 v[8];
};

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
// Lanes whose window origin is further than SH_BAND from the wave's reference ("outliers": flow
// discontinuities, pixels thrown far away, and every lane of a ragged tile) are not allowed to widen the
// streamed region; they are appended to a workgroup-wide list and gathered afterwards with all 64 lanes of
// a wave busy, instead of each wave paying a full gather pass for its one or two odd pixels.
constexpr int SH_NX = 16;     // plane-rows per step held in LDS (union width in x: 8 + spread <= 16)
constexpr int SH_NY = 72;     // longest union in y walked by the streaming path
constexpr int SH_BAND = 4;    // |origin - reference| <= SH_BAND streams
constexpr int SH_WAVES = 8;   // waves (source rows) per workgroup
constexpr int SH_BLOCK = SH_WAVES * 64;

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
// 24 >= 2*11 + 2), so native v_mul_f16 / v_add_f16 are bit-identical.  No fusion: -ffp-contract=off.
// Order: tap(a,b)*w00, tap(a,b+1)*w01, tap(a+1,b)*w10, tap(a+1,b+1)*w11 (correlation_kernels.cu:55-65).
__device__ __forceinline__ _Float16 sh_blend(_Float16 p0, _Float16 c0, _Float16 p1, _Float16 c1, const ShPixel &p) {
  _Float16 acc = p0 * p.h00;
  acc = acc + c0 * p.h01;
  acc = acc + p1 * p.h10;
  acc = acc + c1 * p.h11;
  return acc;
}

template <int R>
__global__ __launch_bounds__(SH_BLOCK, 8) void corr_lookup_sheared_kernel(ShLevels L,
                                                                          const float2 *__restrict__ coords,
                                                                          _Float16 *__restrict__ out, int n,
                                                                          int h1, int w1, int h2, int w2,
                                                                          int num_levels) {
  constexpr int RD = 2 * R + 1, WN = 2 * R + 2;
  static_assert(WN == 8, "the streaming lookup is written for radius 3");
  __shared__ __attribute__((aligned(16))) _Float16 stage_all[SH_WAVES][SH_NX * 64];
  __shared__ int olist[SH_BLOCK];
  __shared__ int ocount;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  _Float16 *stage = stage_all[wave];
  const int xtiles = (w1 + 63) / 64;
  const int lvl = blockIdx.y;
  const int h2l = h2 >> lvl, w2l = w2 >> lvl;
  const int HW1 = h1 * w1;
  const int rowid = blockIdx.x * SH_WAVES + wave;  // (e * h1 + y1) * xtiles + xt
  const bool rowvalid = rowid < n * h1 * xtiles;
  _Float16 *olvl = out + (size_t)lvl * RD * RD * HW1;  // + e * num_levels * RD * RD * HW1 + pixel
  const size_t estride = (size_t)num_levels * RD * RD * HW1;
  if (threadIdx.x == 0) ocount = 0;
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
    _Float16 *o = olvl + (size_t)e * estride + (size_t)y1 * w1 + x1;

    const bool can_stream = ((w1 & 7) == 0) && (xt * 64 + 64 <= w1);
    const unsigned long long tmask = __ballot(touches);
    int refx = 0, refy = 0;
    if (tmask) {
      const int first = __ffsll((long long)tmask) - 1, last = 63 - __clzll((long long)tmask);
      const int fxo = __shfl(ox, first, 64), fyo = __shfl(oy, first, 64);
      const int lxo = __shfl(ox, last, 64), lyo = __shfl(oy, last, 64);
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
    const int bx0 = wave_min_i32(inlier ? ox : big), bx1 = wave_max_i32(inlier ? ox : -big);
    const int by0 = wave_min_i32(inlier ? oy : big), by1 = wave_max_i32(inlier ? oy : -big);
    const bool any = bx1 >= bx0;

    if (!any) {
      // nothing streams in this wave: lanes that touch nothing are exact zeros, outliers are written later
      if (active && !outlier) {
#pragma unroll
        for (int ch = 0; ch < RD * RD; ch++) o[(size_t)ch * HW1] = (_Float16)0.f;
      }
    } else {
      const int nx = bx1 - bx0 + WN, ny = min(by1 - by0 + WN, SH_NY);  // nx <= 16, ny <= 16 by construction
      const _Float16 *vol = L.vol[lvl] + (size_t)e * h2l * w2l * HW1 + (size_t)y1 * w1;
      // per 8-lane group (= one 16-byte piece of a line): the range of window origins inside the group; a
      // piece of line dx / plane-row dy is fetched only if some lane of its group reads it
      int gx0 = inlier ? ox : big, gx1 = inlier ? ox : -big, gy0 = inlier ? oy : big, gy1 = inlier ? oy : -big;
#pragma unroll
      for (int off = 1; off <= 4; off <<= 1) {
        gx0 = min(gx0, __shfl_xor(gx0, off, 64));
        gx1 = max(gx1, __shfl_xor(gx1, off, 64));
        gy0 = min(gy0, __shfl_xor(gy0, off, 64));
        gy1 = max(gy1, __shfl_xor(gy1, off, 64));
      }
      // staging slots: lane + 64 t -> plane-row jx = slot >> 3, 16-byte piece sub = slot & 7 (fixed for all steps)
      int qa[2], qb[2], ldsoff[2], jy_lo[2], jy_hi[2];
      unsigned goff[2];
      bool act[2];
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const int slot = lane + 64 * t;
        const int jx = slot >> 3, sub = slot & 7;
        const int dxv = bx0 + jx;
        const int px0 = __shfl(gx0, sub * 8, 64), px1 = __shfl(gx1, sub * 8, 64);
        const int py0 = __shfl(gy0, sub * 8, 64), py1 = __shfl(gy1, sub * 8, 64);
        act[t] = (jx < nx) && (dxv >= px0) && (dxv < px1 + WN);
        jy_lo[t] = py0 - by0;       // plane-rows [jy_lo, jy_hi) are read by this piece's lanes
        jy_hi[t] = py1 - by0 + WN;
        int m = dxv % w2l;
        m += (m < 0) ? w2l : 0;
        const int xs = xt * 64 + sub * 8;  // first x1 of this piece
        // valid q: 0 <= ((xs + q) >> lvl) + dxv < w2l  <=>  qa <= q < qb
        const int lo = (dxv < 0) ? ((-dxv) << lvl) : 0;
        const int hi = (w2l - dxv > 0) ? ((w2l - dxv) << lvl) : 0;
        qa[t] = max(0, lo - xs);
        qb[t] = min(8, hi - xs);
        goff[t] = (unsigned)m * (unsigned)HW1 + (unsigned)xs;
        ldsoff[t] = jx * 64 + sub * 8;
      }
      int dym = by0 % h2l;
      dym += (dym < 0) ? h2l : 0;
      const unsigned rowstride = (unsigned)w2l * (unsigned)HW1;  // elements between consecutive dy (< 2^32)

      Half8v regs[2];
#pragma unroll
      for (int t = 0; t < 2; t++)
        if (act[t] && jy_lo[t] <= 0) regs[t] = *reinterpret_cast<const Half8v *>(vol + ((unsigned)dym * rowstride + goff[t]));

      // lanes that stream nothing read row 0 with zero weights: they emit exact zeros (their window is out of
      // bounds) -- except outliers, whose outputs are left to the gather phase
      const int rx = inlier ? ox - bx0 : 0, ry = inlier ? oy - by0 : 0;
      const _Float16 *tp = stage + rx * 64 + lane;
      _Float16 prev[WN];
#pragma unroll
      for (int i = 0; i < WN; i++) prev[i] = (_Float16)0.f;

      for (int jy = 0; jy < ny; jy++) {
        const int ty = sy + by0 + jy;
        const bool rowok = (ty >= 0) && (ty < h2l);
#pragma unroll
        for (int t = 0; t < 2; t++) {
          if (act[t] && jy >= jy_lo[t] && jy < jy_hi[t]) {
            Half8v v = regs[t];
            const int a_ = rowok ? qa[t] : 8, b_ = rowok ? qb[t] : 0;
            if (a_ > 0 || b_ < 8) {
#pragma unroll
              for (int q = 0; q < 8; q++)
                if (q < a_ || q >= b_) v.v[q] = (_Float16)0.f;
            }
            *reinterpret_cast<Half8v *>(&stage[ldsoff[t]]) = v;
          }
        }
        dym = (dym + 1 == h2l) ? 0 : dym + 1;
#ifndef SH_ABLATE_LOADS
        if (jy + 1 < ny) {  // next plane-row's lines fly while this one is consumed
#pragma unroll
          for (int t = 0; t < 2; t++)
            if (act[t] && jy + 1 >= jy_lo[t] && jy + 1 < jy_hi[t])
              regs[t] = *reinterpret_cast<const Half8v *>(vol + ((unsigned)dym * rowstride + goff[t]));
        }
#endif
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS row is written before its lanes read it
        __builtin_amdgcn_wave_barrier();
        const int j = jy - ry;
        if (j >= 0 && j < WN) {
          _Float16 cur[WN];
#pragma unroll
          for (int i = 0; i < WN; i++) cur[i] = tp[i * 64];
          if (j >= 1 && active && !outlier) {
            _Float16 *ob = o + (size_t)(j - 1) * HW1;
#pragma unroll
            for (int a = 0; a < RD; a++) {
              const _Float16 acc = sh_blend(prev[a], cur[a], prev[a + 1], cur[a + 1], P);
#ifdef SH_ABLATE_STORES  // ablation builds only (scratch/): keep the value live, store one channel
              if (a == 0 && j == 1) ob[0] = acc; else asm volatile("" ::"v"(acc));
#else
              ob[(size_t)(a * RD) * HW1] = touches ? acc : (_Float16)0.f;
#endif
            }
          }
#pragma unroll
          for (int i = 0; i < WN; i++) prev[i] = cur[i];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);  // tap reads done before the next row overwrites the LDS lines
        __builtin_amdgcn_wave_barrier();
      }
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
