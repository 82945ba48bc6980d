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

// one wave per workgroup; wave = 64 consecutive x1 of one source row, one pyramid level.
//
// Streaming form: the wave walks the plane-rows dy = by0 .. by1+7 of the union window one at a time.  Step jy
// brings the nx (<= SH_NX) 128-byte lines (dy, bx0 .. bx0+nx-1) into a 2 KB LDS row (16-B loads issued one step
// ahead, so they are in flight while the previous row is consumed); a lane whose own window starts ry rows
// into the union reads its 8 taps of tap-row j = jy - ry, combines them with the previous tap-row it kept in
// registers, and emits the 7 outputs (a, b = j-1).  LDS per wave is 2 KB, so occupancy is bounded by registers
// (8 waves/SIMD), not by staging space.
constexpr int SH_NX = 16;   // plane-rows per step held in LDS (union width in x: 8 + spread <= 16)
constexpr int SH_NY = 72;   // longest union in y walked by the streaming path

template <int R>
__global__ __launch_bounds__(64) void corr_lookup_sheared_kernel(ShLevels L, const float2 *__restrict__ coords,
                                                                 _Float16 *__restrict__ out, int n, int h1,
                                                                 int w1, int h2, int w2, int num_levels) {
  constexpr int RD = 2 * R + 1, WN = 2 * R + 2;
  static_assert(WN == 8, "the streaming lookup is written for radius 3");
  __shared__ __attribute__((aligned(16))) _Float16 stage[SH_NX * 64];
  const int lane = threadIdx.x;
  const int xtiles = (w1 + 63) / 64;
  const int xt = blockIdx.x % xtiles;
  const int ey = blockIdx.x / xtiles;  // e * h1 + y1
  const int y1 = ey % h1, e = ey / h1;
  const int lvl = blockIdx.y;
  const int h2l = h2 >> lvl, w2l = w2 >> lvl;
  const int x1 = xt * 64 + lane;
  const bool active = x1 < w1;
  const int HW1 = h1 * w1;
  const size_t pix = (size_t)ey * w1 + min(x1, w1 - 1);

  const float2 c = coords[pix];
  const float scale = 1.0f / (float)(1 << lvl);
  const float x0 = c.x * scale, y0 = c.y * scale;
  const float fx = floorf(x0), fy = floorf(y0);
  const float dx = x0 - fx, dy = y0 - fy;
  // a pixel whose whole window is out of bounds contributes zeros and must not widen the staged region
  const int ix0 = (int)fmaxf(fminf(fx, 1.0e6f), -1.0e6f) - R;
  const int iy0 = (int)fmaxf(fminf(fy, 1.0e6f), -1.0e6f) - R;
  const bool finite = (fabsf(x0) < 1.0e6f) && (fabsf(y0) < 1.0e6f);
  const bool touches = active && finite && (ix0 + WN > 0) && (ix0 < w2l) && (iy0 + WN > 0) && (iy0 < h2l);
  const int sx = min(x1, w1 - 1) >> lvl, sy = y1 >> lvl;
  const int ox = ix0 - sx, oy = iy0 - sy;  // window origin relative to the shear

  // Lanes whose window origin lies within +-4 of a reference lane's stream together (union <= 16 lines
  // wide); the others ("outliers": flow discontinuities, pixels thrown far away) gather their taps one by
  // one afterwards, so a single incoherent pixel does not push the whole wave onto the slow path.
  const unsigned long long tmask = __ballot(touches);
  int refx = 0, refy = 0;
  if (tmask) {
    const int first = __ffsll((long long)tmask) - 1, last = 63 - __clzll((long long)tmask);
    const int fxo = __shfl(ox, first, 64), fyo = __shfl(oy, first, 64);
    const int lxo = __shfl(ox, last, 64), lyo = __shfl(oy, last, 64);
    const bool nearf = touches && (abs(ox - fxo) <= 4) && (abs(oy - fyo) <= 4);
    const bool nearl = touches && (abs(ox - lxo) <= 4) && (abs(oy - lyo) <= 4);
    const bool usef = __popcll(__ballot(nearf)) >= __popcll(__ballot(nearl));
    refx = usef ? fxo : lxo;
    refy = usef ? fyo : lyo;
  }
  const bool inlier = touches && (abs(ox - refx) <= 4) && (abs(oy - refy) <= 4);
  const bool outlier = touches && !inlier;
  const int big = 1 << 28;
  const int bx0 = wave_min_i32(inlier ? ox : big), bx1 = wave_max_i32(inlier ? ox : -big);
  const int by0 = wave_min_i32(inlier ? oy : big), by1 = wave_max_i32(inlier ? oy : -big);
  const bool any = bx1 >= bx0;
  const int nx = any ? (bx1 - bx0 + WN) : 0, ny = any ? (by1 - by0 + WN) : 0;
  const _Float16 *vol = L.vol[lvl] + (size_t)e * h2l * w2l * HW1 + (size_t)y1 * w1;

  // scalar_t(dx * dy): f32 product rounded to half (see corr_lookup.hip).
  float w00 = (1.0f - dx) * (1.0f - dy), w01 = (1.0f - dx) * dy, w10 = dx * (1.0f - dy), w11 = dx * dy;
  if (!touches) w00 = w01 = w10 = w11 = 0.f;  // nothing in bounds (incl. NaN / inf coords): exact zeros
  asm volatile("" : "+v"(w00), "+v"(w01), "+v"(w10), "+v"(w11));
  const _Float16 h00 = (_Float16)w00, h01 = (_Float16)w01, h10 = (_Float16)w10, h11 = (_Float16)w11;
  // c10::Half `a * b` / `a + b` compute in float and round to half; for two halves that is exactly the
  // IEEE half operation (the float product is exact; a float sum rounded to half cannot double-round because
  // 24 >= 2*11 + 2), so native v_mul_f16 / v_add_f16 are bit-identical.  No fusion: -ffp-contract=off.
  _Float16 *o = out + ((size_t)e * num_levels * RD * RD + (size_t)lvl * RD * RD) * HW1 + (size_t)y1 * w1 + x1;

  if (!tmask) {  // the whole wave is out of bounds
    if (active) {
#pragma unroll
      for (int ch = 0; ch < RD * RD; ch++) o[(size_t)ch * HW1] = (_Float16)0.f;
    }
    return;
  }

  const bool stream = (nx <= SH_NX) && (ny <= SH_NY) && ((w1 & 7) == 0) && (xt * 64 + 64 <= w1);
  if (stream) {
    // staging slots: lane + 64 t -> plane-row jx = slot >> 3, 16-byte piece sub = slot & 7 (fixed for all steps)
    int dxm[2], qa[2], qb[2], ldsoff[2];
    unsigned goff[2];
    bool act[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int slot = lane + 64 * t;
      const int jx = slot >> 3, sub = slot & 7;
      act[t] = jx < nx;
      const int dxv = bx0 + jx;
      int m = dxv % w2l;
      m += (m < 0) ? w2l : 0;
      dxm[t] = m;
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
    const unsigned rowstride = (unsigned)w2l * (unsigned)HW1;  // elements between consecutive dy (< 2^32 per edge)

    Half8v regs[2];
#pragma unroll
    for (int t = 0; t < 2; t++)
      if (act[t]) regs[t] = *reinterpret_cast<const Half8v *>(vol + ((unsigned)dym * rowstride + goff[t]));

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
        if (act[t]) {
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
          if (act[t]) regs[t] = *reinterpret_cast<const Half8v *>(vol + ((unsigned)dym * rowstride + goff[t]));
      }
#endif
      __syncthreads();  // one wave: orders the LDS row write before the tap reads
      const int j = jy - ry;
      if (j >= 0 && j < WN) {
        _Float16 cur[WN];
#pragma unroll
        for (int i = 0; i < WN; i++) cur[i] = tp[i * 64];
        if (j >= 1 && active && !outlier) {
          _Float16 *ob = o + (size_t)(j - 1) * HW1;
#pragma unroll
          for (int a = 0; a < RD; a++) {
            // tap(a,b)*w00, tap(a,b+1)*w01, tap(a+1,b)*w10, tap(a+1,b+1)*w11 (correlation_kernels.cu:55-65)
            _Float16 acc = prev[a] * h00;
            acc = acc + cur[a] * h01;
            acc = acc + prev[a + 1] * h10;
            acc = acc + cur[a + 1] * h11;
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
      __syncthreads();  // tap reads done before the next row overwrites the LDS line buffer
    }
    if (!outlier) return;
  }

  // outliers of a streaming wave, or every lane of a wave that could not stream (ragged width):
  // gather straight from the sheared volume
  _Float16 win[WN][WN];
#pragma unroll
  for (int j = 0; j < WN; j++) {
    const int ty = iy0 + j;
    const bool rok = touches && (ty >= 0) && (ty < h2l);
    int dym = (oy + j) % h2l;
    dym += (dym < 0) ? h2l : 0;
#pragma unroll
    for (int i = 0; i < WN; i++) {
      const int tx = ix0 + i;
      const bool ok = rok && (tx >= 0) && (tx < w2l);
      int dxm = (ox + i) % w2l;
      dxm += (dxm < 0) ? w2l : 0;
      win[j][i] = ok ? vol[((size_t)dym * w2l + dxm) * HW1 + x1] : (_Float16)0.f;
    }
  }
  if (!active) return;
#pragma unroll
  for (int a = 0; a < RD; a++) {
#pragma unroll
    for (int b = 0; b < RD; b++) {
      _Float16 acc = win[b][a] * h00;
      acc = acc + win[b + 1][a] * h01;
      acc = acc + win[b][a + 1] * h10;
      acc = acc + win[b + 1][a + 1] * h11;
      o[(size_t)(a * RD + b) * HW1] = touches ? acc : (_Float16)0.f;
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
  dim3 grid((unsigned)(n * h1 * xtiles), num_levels);
  hipLaunchKernelGGL((corr_lookup_sheared_kernel<3>), grid, dim3(64), 0, (hipStream_t)stream, L,
                     reinterpret_cast<const float2 *>(coords_nhw2), static_cast<_Float16 *>(corr), n, h1, w1, h2, w2,
                     num_levels);
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

}  // extern "C"
