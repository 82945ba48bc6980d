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
constexpr int SH_CAP = 112;  // plane-rows (128 B each) a wave may stage in LDS: 14 KB -> 11 waves per CU

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

// one wave per workgroup; wave = 64 consecutive x1 of one source row, one pyramid level
template <int R>
__global__ __launch_bounds__(64) void corr_lookup_sheared_kernel(ShLevels L, const float2 *__restrict__ coords,
                                                                 _Float16 *__restrict__ out, int n, int h1,
                                                                 int w1, int h2, int w2, int num_levels) {
  constexpr int RD = 2 * R + 1, WN = 2 * R + 2;
  __shared__ __attribute__((aligned(16))) _Float16 stage[SH_CAP * 64];
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

  const int big = 1 << 28;
  const int bx0 = wave_min_i32(touches ? ox : big), bx1 = wave_max_i32(touches ? ox : -big);
  const int by0 = wave_min_i32(touches ? oy : big), by1 = wave_max_i32(touches ? oy : -big);
  const bool any = bx1 >= bx0;
  const int nx = any ? (bx1 - bx0 + WN) : 0, ny = any ? (by1 - by0 + WN) : 0;
  const _Float16 *vol = L.vol[lvl] + (size_t)e * h2l * w2l * HW1 + (size_t)y1 * w1;

  _Float16 win[WN][WN];
  const bool staged = any && (nx * ny <= SH_CAP) && ((w1 & 7) == 0) && (xt * 64 + 64 <= w1);
  if (staged) {
    // 8 lanes x 16 B fetch one 128-byte plane-row; 8 plane-rows per wave-instruction.  Elements whose
    // target pixel is out of bounds are zeroed HERE (a contiguous run of x1 per plane-row), so the 64 tap
    // reads per lane below need no bounds logic at all.
    const int sub = lane & 7, rsel = lane >> 3;
    const int xs = xt * 64 + sub * 8;  // first x1 of this lane's 8 elements
    // (nx >= 8, so the 8 row selectors start in plane-row 0 and each step wraps at most once)
    int jy = 0, jx = rsel;
    int dym = by0 % h2l, dxm = (bx0 + jx) % w2l;
    dym += (dym < 0) ? h2l : 0;
    dxm += (dxm < 0) ? w2l : 0;
    const unsigned HWu = (unsigned)HW1;
    for (int r = rsel; r < nx * ny; r += 8) {
      const int dyv = by0 + jy, dxv = bx0 + jx;
      const unsigned off = ((unsigned)dym * (unsigned)w2l + (unsigned)dxm) * HWu + (unsigned)xs;  // < 2^32 per edge
      Half8v v = *reinterpret_cast<const Half8v *>(vol + off);
      const int ty = sy + dyv;
      // valid q: 0 <= ((xs + q) >> lvl) + dxv < w2l  <=>  qa <= q < qb
      const int lo = (dxv < 0) ? ((-dxv) << lvl) : 0;
      const int hi = (w2l - dxv > 0) ? ((w2l - dxv) << lvl) : 0;
      int qa = max(0, lo - xs), qb = min(8, hi - xs);
      if (ty < 0 || ty >= h2l) qb = 0;
      if (qa > 0 || qb < 8) {
#pragma unroll
        for (int q = 0; q < 8; q++)
          if (q < qa || q >= qb) v.v[q] = (_Float16)0.f;
      }
      *reinterpret_cast<Half8v *>(&stage[r * 64 + sub * 8]) = v;
      jx += 8;
      dxm += 8;
      if (jx >= nx) {  // next plane-row
        jx -= nx;
        jy++;
        dym = (dym + 1 == h2l) ? 0 : dym + 1;
        dxm = (bx0 + jx) % w2l;
        dxm += (dxm < 0) ? w2l : 0;
      } else {
        while (dxm >= w2l) dxm -= w2l;
      }
    }
    __syncthreads();  // single wave: just the LDS write -> read ordering
    // lanes that touch nothing read row 0 (their weights are zero and their outputs are forced to zero)
    const int rx = touches ? ox - bx0 : 0, ry = touches ? oy - by0 : 0;
    const _Float16 *tp = stage + (ry * nx + rx) * 64 + lane;
    const int rstride = nx * 64;
#pragma unroll
    for (int j = 0; j < WN; j++) {
#pragma unroll
      for (int i = 0; i < WN; i++) win[j][i] = tp[j * rstride + i * 64];
    }
  } else {
    // incoherent flow (or ragged width): gather straight from the sheared volume
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
  }
  if (!active) return;

  // scalar_t(dx * dy): f32 product rounded to half (see corr_lookup.hip).
  float w00 = (1.0f - dx) * (1.0f - dy), w01 = (1.0f - dx) * dy, w10 = dx * (1.0f - dy), w11 = dx * dy;
  if (!touches) w00 = w01 = w10 = w11 = 0.f;  // nothing in bounds (incl. NaN / inf coords): exact zeros
  asm volatile("" : "+v"(w00), "+v"(w01), "+v"(w10), "+v"(w11));
  const _Float16 h00 = (_Float16)w00, h01 = (_Float16)w01, h10 = (_Float16)w10, h11 = (_Float16)w11;
  // c10::Half `a * b` / `a + b` compute in float and round to half; for two halves that is exactly the
  // IEEE half operation (the float product is exact; a float sum rounded to half cannot double-round because
  // 24 >= 2*11 + 2), so native v_mul_f16 / v_add_f16 are bit-identical.  No fusion: -ffp-contract=off.
  _Float16 *o = out + ((size_t)e * num_levels * RD * RD + (size_t)lvl * RD * RD) * HW1 + (size_t)y1 * w1 + x1;
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
