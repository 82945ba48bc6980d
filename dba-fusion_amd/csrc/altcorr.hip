// On-the-fly windowed correlation (no materialised volume), gfx950.
//
// Replaces altcorr_forward_kernel (/root/reference/src/altcorr_kernel.cu:27-149, launcher :290-319: float and half),
// the op behind AltCorrBlock (/root/reference/dbaf/modules/corr.py:91-139).  The reference runs 32-thread blocks that
// rely on NVIDIA warp-synchronous execution (no barrier between the shared-memory dot product and the next tap's
// overwrite); on wave64 hardware that assumption does not hold, and its per-tap re-staging of 32 x 32 scalars is all
// latency.  Here:
//   * a wave owns a 4 x 16 tile of source pixels of one (batch, coordinate set); a lane is a pixel;
//   * the tile's windows overlap (coherent flow), so the UNION of the 64 windows of fmap2 -- (16 + 2r + 1 + spread) x
//     (4 + 2r + 1 + spread) target pixels -- is staged in LDS in 32-byte channel slices (8 floats / 16 halves) with
//     coalesced 16-byte loads: a target pixel's slice is fetched once per tile instead of once per tap and lane;
//     pixel pitch 48 B makes the lanes' 16-byte LDS reads conflict-free (neighbouring lanes read neighbouring pixels);
//   * every lane then walks its (2r+2)^2 taps: two 16-byte LDS reads + one multiply-add per channel into the tap's
//     running dot product; after each 32-channel chunk the dot products are scattered with the four bilinear weights
//     into the lane's (2r+1)^2 outputs -- the reference's chunk / tap / nw-ne-sw-se order exactly
//     (altcorr_kernel.cu:58-147), so the half instantiation (c10::Half: every product and sum rounded to half) is
//     bit-identical to it and the float one differs only by the fused multiply-add nvcc emits for `s += a * b`
//     (-fmad=true is its default; this file uses __builtin_fmaf there);
//   * tiles whose union does not fit the staging area (incoherent coordinates) read fmap2 per lane instead.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

// bit-exact parity with the reference arithmetic: no implicit mul+add fusion anywhere in this file
#pragma clang fp contract(off)

namespace dba {

typedef unsigned u4v_alt __attribute__((ext_vector_type(4)));

constexpr int ALT_TH = 4, ALT_TW = 16;   // tile of source pixels per wave
constexpr int ALT_UMAX = 448;            // union pixels that fit the staging area (21 KB per wave at 48 B per pixel)
constexpr int ALT_PITCH = 48;            // bytes per staged pixel: 32 of data + 16 of padding (bank spread)

template <typename T>
struct AltT;
template <>
struct AltT<float> {
  static constexpr int KS = 8;  // channels per 32-byte slice
  static __device__ __forceinline__ float mad(float s, float a, float b) { return __builtin_fmaf(a, b, s); }
  static __device__ __forceinline__ float mul(float a, float b) { return a * b; }
  static __device__ __forceinline__ float add(float a, float b) { return a + b; }
  static __device__ __forceinline__ float from_float(float x) { return x; }
};
template <>
struct AltT<_Float16> {
  static constexpr int KS = 16;
  // c10::Half: `s += a * b` is two roundings (product to half, sum to half); native half ops round identically
  static __device__ __forceinline__ _Float16 mad(_Float16 s, _Float16 a, _Float16 b) { return s + a * b; }
  static __device__ __forceinline__ _Float16 mul(_Float16 a, _Float16 b) { return a * b; }
  static __device__ __forceinline__ _Float16 add(_Float16 a, _Float16 b) { return a + b; }
  static __device__ __forceinline__ _Float16 from_float(float x) { return (_Float16)x; }
};

// Pyramid mode (num_levels > 0; AltCorrBlock.corr_fn, /root/reference/dbaf/modules/corr.py:107-125, in ONE launch): the grid's
// z index is (edge, coordinate set, level); level l reads fmap2 from Lv.f2[l] ([F, H1 >> l, W1 >> l, C]), the level-0
// coordinates divided by 2^l (exact), source / target frames through ii / jj (no gathered copies of the maps), and writes
// its 49 channels behind the lower levels' in corr [B, S, L * 49, H1, W1].
struct AltLevels {
  const void *f2[8];
};

template <int R, typename T>
__global__ __launch_bounds__(64) void altcorr_forward_kernel(const T *__restrict__ fmap1, const T *__restrict__ fmap2,
                                                             const float *__restrict__ coords, T *__restrict__ corr,
                                                             int B, int S, int H1, int W1, int H2, int W2, int C,
                                                             AltLevels Lv, int num_levels, const int64_t *__restrict__ ii,
                                                             const int64_t *__restrict__ jj) {
  constexpr int RD = 2 * R + 1, WN = 2 * R + 2, KS = AltT<T>::KS;
  __shared__ __attribute__((aligned(16))) unsigned char stage[ALT_UMAX * ALT_PITCH];
  const int lane = threadIdx.x;
  int bs = blockIdx.z, lvl = 0;
  if (num_levels > 0) {
    lvl = bs % num_levels;
    bs /= num_levels;
    H2 = H1 >> lvl;
    W2 = W1 >> lvl;
    fmap2 = static_cast<const T *>(Lv.f2[lvl]);
  }
  const int b = bs / S;
  const int b1 = ii ? (int)ii[b] : b, b2 = jj ? (int)jj[b] : b;   // frames of the edge's source / target maps
  const int h1 = blockIdx.y * ALT_TH + (lane >> 4), w1 = blockIdx.x * ALT_TW + (lane & 15);
  const bool inb = (h1 < H1) && (w1 < W1);
  const int HW1 = H1 * W1;
  const int pix = min(h1, H1 - 1) * W1 + min(w1, W1 - 1);

  const float *cp = coords + ((size_t)bs * HW1 + pix) * 2;
  const float cscale = 1.0f / (float)(1 << lvl);   // coords / 2**i (corr.py:116): exact
  const float x2 = cp[0] * cscale, y2 = cp[1] * cscale;
  const float fxf = floorf(x2), fyf = floorf(y2);
  const float dx = x2 - fxf, dy = y2 - fyf;
  const bool sane = (fabsf(x2) < 1.0e6f) && (fabsf(y2) < 1.0e6f);   // the reference's float -> int cast is undefined beyond
  const int wx0 = sane ? (int)fxf - R : -(1 << 20), wy0 = sane ? (int)fyf - R : -(1 << 20);
  // a window that misses the map entirely contributes exact zeros and must not stretch the staged union
  const bool hits = inb && sane && (wx0 + WN > 0) && (wx0 < W2) && (wy0 + WN > 0) && (wy0 < H2);
  // static_cast<scalar_t>(dy * dx) etc. (altcorr_kernel.cu:112-115): a FLOAT product, then one rounding to T.  The empty
  // asm keeps the compiler from folding product and conversion into one mixed-precision instruction (v_fma_mix), which
  // rounds once and differs from the reference where the float product lands on a half-way point of the half grid.
  float pnw = dy * dx, pne = dy * (1 - dx), psw = (1 - dy) * dx, pse = (1 - dy) * (1 - dx);
  asm volatile("" : "+v"(pnw), "+v"(pne), "+v"(psw), "+v"(pse));
  const T wnw = AltT<T>::from_float(pnw), wne = AltT<T>::from_float(pne);
  const T wsw = AltT<T>::from_float(psw), wse = AltT<T>::from_float(pse);

  const int big = 1 << 28;
  const bool any = __ballot(hits) != 0ull;
  const int bx0 = wave_minmax<true>(hits ? wx0 : big), bx1 = wave_minmax<false>(hits ? wx0 : -big);
  const int by0 = wave_minmax<true>(hits ? wy0 : big), by1 = wave_minmax<false>(hits ? wy0 : -big);
  const int UW = bx1 - bx0 + WN, UH = by1 - by0 + WN;
  const bool vec_ok = ((C * (int)sizeof(T)) % 16 == 0);  // 16-byte slices of a pixel's channels are aligned
  const bool staged = any && vec_ok && (UW > 0) && (UH > 0) && ((long)UW * UH <= ALT_UMAX);
  const unsigned tap0 = hits ? (unsigned)(((wy0 - by0) * UW + (wx0 - bx0)) * ALT_PITCH) : 0u;  // this lane's first tap

  T acc[RD * RD];  // channel = iy + RD * ix
#pragma unroll
  for (int i = 0; i < RD * RD; i++) acc[i] = AltT<T>::from_float(0.f);

  const T *f1 = fmap1 + ((size_t)b1 * HW1 + pix) * C;
  const T *f2b = fmap2 + (size_t)b2 * H2 * W2 * C;
  for (int c0 = 0; c0 < C; c0 += 32) {
    T sdot[WN * WN];
#pragma unroll
    for (int i = 0; i < WN * WN; i++) sdot[i] = AltT<T>::from_float(0.f);
    for (int c = c0; c < min(c0 + 32, C); c += KS) {
      const int cn = min(KS, C - c);
      // this pixel's slice of fmap1
      T a[KS];
      if (vec_ok && cn == KS) {
        u4v_alt r0 = *reinterpret_cast<const u4v_alt *>(f1 + c), r1 = *reinterpret_cast<const u4v_alt *>(f1 + c + KS / 2);
        __builtin_memcpy(&a[0], &r0, 16);
        __builtin_memcpy(&a[KS / 2], &r1, 16);
      } else {
#pragma unroll
        for (int k = 0; k < KS; k++) a[k] = (k < cn) ? f1[c + k] : AltT<T>::from_float(0.f);
      }
      if (staged) {
        // ---- union of the tile's windows, this slice: coalesced 16-byte loads, zeros outside the map ----
        const int total = UW * UH * 2;  // 16-byte pieces
        for (int u = lane; u < total; u += 64) {
          const int up = u >> 1, hf = u & 1;
          const int uy = up / UW, ux = up - uy * UW;
          const int h2 = by0 + uy, w2 = bx0 + ux;
          u4v_alt v = {0u, 0u, 0u, 0u};
          if (h2 >= 0 && h2 < H2 && w2 >= 0 && w2 < W2) {
            const T *src = f2b + ((size_t)h2 * W2 + w2) * C + c + hf * (KS / 2);
            if (cn == KS) v = *reinterpret_cast<const u4v_alt *>(src);
            else {
              T tmp[KS / 2];
#pragma unroll
              for (int k = 0; k < KS / 2; k++) tmp[k] = (hf * (KS / 2) + k < cn) ? src[k] : AltT<T>::from_float(0.f);
              __builtin_memcpy(&v, tmp, 16);
            }
          }
          *reinterpret_cast<u4v_alt *>(stage + up * ALT_PITCH + hf * 16) = v;
        }
        __builtin_amdgcn_wave_barrier();  // LDS operations of one wave execute in program order
#pragma unroll
        for (int iy = 0; iy < WN; iy++) {
#pragma unroll
          for (int ix = 0; ix < WN; ix++) {
            const unsigned char *tp = stage + tap0 + (unsigned)((iy * UW + ix) * ALT_PITCH);
            u4v_alt r0 = *reinterpret_cast<const u4v_alt *>(tp), r1 = *reinterpret_cast<const u4v_alt *>(tp + 16);
            T v[KS];
            __builtin_memcpy(&v[0], &r0, 16);
            __builtin_memcpy(&v[KS / 2], &r1, 16);
            T sd = sdot[iy * WN + ix];
#pragma unroll
            for (int k = 0; k < KS; k++) sd = AltT<T>::mad(sd, a[k], v[k]);  // ascending channels, like the reference
            sdot[iy * WN + ix] = sd;
          }
        }
        __builtin_amdgcn_wave_barrier();  // the next slice overwrites the staging area
      } else {
        // ---- incoherent tile (or odd channel count): every lane reads its own taps ----
#pragma unroll
        for (int iy = 0; iy < WN; iy++) {
#pragma unroll
          for (int ix = 0; ix < WN; ix++) {
            const int h2 = wy0 + iy, w2 = wx0 + ix;
            if (h2 >= 0 && h2 < H2 && w2 >= 0 && w2 < W2) {
              const T *f2 = f2b + ((size_t)h2 * W2 + w2) * C + c;
              T sd = sdot[iy * WN + ix];
#pragma unroll
              for (int k = 0; k < KS; k++) sd = AltT<T>::mad(sd, a[k], (k < cn) ? f2[k] : AltT<T>::from_float(0.f));
              sdot[iy * WN + ix] = sd;
            }
          }
        }
      }
    }
    // ---- scatter the chunk's dot products: corr[iy-1][ix-1] += s * w, four updates per tap in the reference's order ----
#pragma unroll
    for (int iy = 0; iy < WN; iy++) {
#pragma unroll
      for (int ix = 0; ix < WN; ix++) {
        const T sd = hits ? sdot[iy * WN + ix] : AltT<T>::from_float(0.f);
        if (iy > 0 && ix > 0) acc[(iy - 1) + RD * (ix - 1)] = AltT<T>::add(acc[(iy - 1) + RD * (ix - 1)], AltT<T>::mul(sd, wnw));
        if (iy > 0 && ix < RD) acc[(iy - 1) + RD * ix] = AltT<T>::add(acc[(iy - 1) + RD * ix], AltT<T>::mul(sd, wne));
        if (iy < RD && ix > 0) acc[iy + RD * (ix - 1)] = AltT<T>::add(acc[iy + RD * (ix - 1)], AltT<T>::mul(sd, wsw));
        if (iy < RD && ix < RD) acc[iy + RD * ix] = AltT<T>::add(acc[iy + RD * ix], AltT<T>::mul(sd, wse));
      }
    }
  }
  if (inb) {
    T *o = corr + ((size_t)(bs * max(num_levels, 1) + lvl) * RD * RD) * HW1 + pix;
#pragma unroll
    for (int i = 0; i < RD * RD; i++) o[(size_t)i * HW1] = acc[i];
  }
}

// adjoint of the above wrt the feature maps (altcorr_backward_kernel, altcorr_kernel.cu:152-286; training only).
// One lane per (batch, pixel, 32-channel chunk): the fmap1 gradient accumulates in registers, the fmap2
// gradient is scattered with float atomics like the reference's atomicAdd (:267).
template <int R>
__global__ __launch_bounds__(256) void altcorr_backward_kernel(const float *__restrict__ fmap1,
                                                               const float *__restrict__ fmap2,
                                                               const float *__restrict__ coords,
                                                               const float *__restrict__ corr_grad,
                                                               float *__restrict__ fmap1_grad,
                                                               float *__restrict__ fmap2_grad, int B, int S, int H1,
                                                               int W1, int H2, int W2, int C) {
  constexpr int RD = 2 * R + 1;
  const int HW1 = H1 * W1;
  const int chunks = (C + 31) / 32;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)B * HW1 * chunks) return;
  const int pix = (int)(gid % HW1);
  const int ck = (int)((gid / HW1) % chunks);
  const int b = (int)(gid / ((long)HW1 * chunks));
  const int c = ck * 32, cn = min(32, C - c);
  const float *f1 = fmap1 + ((size_t)b * HW1 + pix) * C + c;
  float a[32], f1g[32];
#pragma unroll
  for (int k = 0; k < 32; k++) {
    a[k] = (k < cn) ? f1[k] : 0.f;
    f1g[k] = 0.f;
  }
  for (int s = 0; s < S; s++) {
    const float *cp = coords + (((size_t)b * S + s) * HW1 + pix) * 2;
    const float x2 = cp[0], y2 = cp[1];
    if (!((fabsf(x2) < 1.0e6f) && (fabsf(y2) < 1.0e6f))) continue;
    const float fxf = floorf(x2), fyf = floorf(y2);
    const float dx = x2 - fxf, dy = y2 - fyf;
    const int w0 = (int)fxf - R, h0 = (int)fyf - R;
    const float *gp = corr_grad + (((size_t)b * S + s) * RD * RD) * HW1 + pix;
#pragma unroll
    for (int iy = 0; iy < RD + 1; iy++) {
#pragma unroll
      for (int ix = 0; ix < RD + 1; ix++) {
        const int h2 = h0 + iy, w2 = w0 + ix;
        if (h2 < 0 || h2 >= H2 || w2 < 0 || w2 >= W2) continue;
        float g = 0.f;
        if (iy > 0 && ix > 0) g += gp[(size_t)((iy - 1) + RD * (ix - 1)) * HW1] * dy * dx;
        if (iy > 0 && ix < RD) g += gp[(size_t)((iy - 1) + RD * ix) * HW1] * dy * (1 - dx);
        if (iy < RD && ix > 0) g += gp[(size_t)(iy + RD * (ix - 1)) * HW1] * (1 - dy) * dx;
        if (iy < RD && ix < RD) g += gp[(size_t)(iy + RD * ix) * HW1] * (1 - dy) * (1 - dx);
        const size_t o2 = (((size_t)b * H2 + h2) * W2 + w2) * C + c;
        const float *f2 = fmap2 + o2;
        float *f2g = fmap2_grad + o2;
#pragma unroll
        for (int k = 0; k < 32; k++) {
          if (k < cn) {
            f1g[k] += g * f2[k];
            atomicAdd(&f2g[k], g * a[k]);
          }
        }
      }
    }
  }
  float *o = fmap1_grad + ((size_t)b * HW1 + pix) * C + c;
#pragma unroll
  for (int k = 0; k < 32; k++)
    if (k < cn) o[k] += f1g[k];
}


// =====================================================================================================================
// The same op as a contraction on the matrix cores, for HALF feature pyramids and float output: what AltCorrBlock computes
// in the reference (dbaf/modules/corr.py:107-125: the half pyramid of fmaps / 4 is cast with .float() at every lookup and
// correlated in float).  Products of two halves are exact in float, so feeding the halves to v_mfma_f32_32x32x16_f16
// differs from the reference's float chain only in the ORDER of the float additions (tests: <= 2e-5 of the reference).
//
// A workgroup (8 waves) owns a 4 x 16 tile of source pixels of one (edge, coordinate set, level).  The windows of the 64
// pixels overlap (coherent flow): their union box, clipped to the level's map, is a few hundred target pixels (bench scene:
// 252 / 128 / 72 / 42 on the four levels, never above 378).  The tile's correlation with that box -- 64 sources x NB targets
// x C channels -- is ONE small GEMM instead of 64 x 64 separate dot products that each re-read their operands:
//   * wave w takes the target blocks w, w + 8 (32 targets each); their fragments (a target pixel's C halves are
//     contiguous in the channels-last pyramid: 16-byte loads, no staging) stay in registers for both source halves;
//   * per half of the tile (32 sources: two tile rows) the products targets x sources leave the accumulators as float quads
//     of four consecutive targets of one source into an LDS tile T[32][NB] (50 KB: three workgroups per CU);
//   * window phase: thread (pixel p of the half, output row g, half of the output columns) reads its 2 x 5 taps from T[p] -- taps outside
//     the box are outside the map: zeros --, blends them with the pixel's four bilinear weights in the reference's
//     se, sw, ne, nw order and stores its 4 (3) outputs of row g.
// A tile whose clipped box exceeds ALTM_NB targets (incoherent coordinates) computes its taps as per-thread dot products.
typedef _Float16 altm_half8 __attribute__((ext_vector_type(8)));
typedef float altm_f16v __attribute__((ext_vector_type(16)));
typedef float altm_f4v __attribute__((ext_vector_type(4)));
constexpr int ALTM_NB = 384;                // targets of a box the LDS tile holds (12 blocks of 32)
constexpr int ALTM_PITCH = ALTM_NB + 4;     // floats per source row: 16-byte aligned quads, rows 4 banks apart
constexpr int ALTM_KS = 8;                  // k-steps of 16 channels: C <= 128
constexpr int ALTM_ROW = 144;               // bytes of a staged operand row: 64 channels + 16 bytes (bank spread)
#ifndef ALTM_WAVES_DEF
#define ALTM_WAVES_DEF 4
#endif
#ifndef ALTM_MIN_WAVES
#define ALTM_MIN_WAVES 2
#endif
// waves per workgroup (4 or 8), target blocks per wave (waves x blocks x 32 >= ALTM_NB), output columns per window thread
// (4 waves: a thread takes an output row; 8 waves: half a row)
constexpr int ALTM_WAVES = ALTM_WAVES_DEF, ALTM_BPW = 12 / ALTM_WAVES + (ALTM_WAVES == 8 ? 1 : 0), ALTM_NOX = (ALTM_WAVES == 8) ? 4 : 7;
static_assert(ALTM_WAVES * ALTM_BPW * 32 >= ALTM_NB, "every block of the box has a wave");
constexpr int ALTM_SRC_OFF = 32 * ALTM_PITCH * 4, ALTM_TGT_OFF = ALTM_SRC_OFF + 64 * ALTM_ROW;
constexpr int ALTM_LDS_BYTES = ALTM_TGT_OFF + ALTM_WAVES * 32 * ALTM_ROW;

__global__ __launch_bounds__(64 * ALTM_WAVES, ALTM_MIN_WAVES) void altcorr_mfma_kernel(const _Float16 *__restrict__ fmap1, AltLevels Lv,
                                                           const float *__restrict__ coords, float *__restrict__ corr, int S,
                                                           int H1, int W1, int C, int num_levels,
                                                           const int64_t *__restrict__ ii, const int64_t *__restrict__ jj) {
  constexpr int R = 3, RD = 7, WN = 8;
  extern __shared__ __attribute__((aligned(16))) float altm_tile[];   // [32][ALTM_PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bs = blockIdx.z;
  const int lvl = bs % num_levels;
  bs /= num_levels;
  const int H2 = H1 >> lvl, W2 = W1 >> lvl;
  const _Float16 *fmap2 = static_cast<const _Float16 *>(Lv.f2[lvl]);
  const int b = bs / S;
  const int b1 = ii ? (int)ii[b] : b, b2 = jj ? (int)jj[b] : b;
  const int HW1 = H1 * W1;

  // ---- every wave: lane = pixel of the tile (row lane >> 4, column lane & 15) ------------------------------------------------
  const int h1 = blockIdx.y * ALT_TH + (lane >> 4), w1 = blockIdx.x * ALT_TW + (lane & 15);
  const bool inb = (h1 < H1) && (w1 < W1);
  const int pix = min(h1, H1 - 1) * W1 + min(w1, W1 - 1);
  const float *cp = coords + ((size_t)bs * HW1 + pix) * 2;
  const float cscale = 1.0f / (float)(1 << lvl);   // coords / 2**i (corr.py:116): exact
  const float x2 = cp[0] * cscale, y2 = cp[1] * cscale;
  const float fxf = floorf(x2), fyf = floorf(y2);
  const float dx = x2 - fxf, dy = y2 - fyf;
  const bool sane = (fabsf(x2) < 1.0e6f) && (fabsf(y2) < 1.0e6f);
  const int wx0 = sane ? (int)fxf - R : -(1 << 20), wy0 = sane ? (int)fyf - R : -(1 << 20);
  const bool hits = inb && sane && (wx0 + WN > 0) && (wx0 < W2) && (wy0 + WN > 0) && (wy0 < H2);
  const float wnw = dy * dx, wne = dy * (1 - dx), wsw = (1 - dy) * dx, wse = (1 - dy) * (1 - dx);   // (altcorr_kernel.cu:112-115)
  const int big = 1 << 28;
  const bool any = __ballot(hits) != 0ull;
  // the union box of the windows, clipped to the map: columns cx0 .. cx1 - 1, rows cy0 .. cy1 - 1
  const int cx0 = max(wave_minmax<true>(hits ? wx0 : big), 0), cx1 = min(wave_minmax<false>(hits ? wx0 : -big) + WN, W2);
  const int cy0 = max(wave_minmax<true>(hits ? wy0 : big), 0), cy1 = min(wave_minmax<false>(hits ? wy0 : -big) + WN, H2);
  const int CW = any ? cx1 - cx0 : 0, CH = any ? cy1 - cy0 : 0;
  const int NBX = CW * CH;
  const bool boxed = any && NBX <= ALTM_NB;
  const int nblk = boxed ? (NBX + 31) >> 5 : 0;
  const _Float16 *f2b = fmap2 + (size_t)b2 * H2 * W2 * C;
  const int l31 = lane & 31;

  // ---- operands through LDS.  A lane that loads ITS fragment straight from the channels-last map takes 16 bytes out of a line
  // of its own: a wave's load instruction then touches 32 lines for 1 KB (the texture addresser walks them one by one; the
  // kernel was bound by exactly that: 445 us).  Staged, an instruction reads whole lines -- lane = (pixel l >> 3, 16-byte
  // piece l & 7): eight pixels x 128 contiguous bytes -- into rows of 144 bytes (36 words: the fragment reads of 32
  // consecutive pixels, 16 bytes each, fall into different banks), one HALF of the channels at a time:
  //   ALTM_SRC  [64 pixels][144 B]   the tile's sources, staged by the whole workgroup
  //   ALTM_TGT  [wave][32 targets][144 B]   the block a wave is multiplying, private to the wave (LDS operations of one wave
  //             execute in order: no barrier between its writes and its fragment reads)
  // offsets (halves) of this lane's pieces: sources P = tid >> 3 and P + 32; targets 8 j + (lane >> 3) of block u
  const int piece = lane & 7;
  unsigned char *const lds_src = reinterpret_cast<unsigned char *>(altm_tile) + ALTM_SRC_OFF;
  unsigned char *const lds_tgt = reinterpret_cast<unsigned char *>(altm_tile) + ALTM_TGT_OFF + wave * (32 * ALTM_ROW);
  long soff[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int P = (tid >> 3) + 32 * j;
    soff[j] = ((long)b1 * HW1 + __builtin_amdgcn_ds_bpermute(P * 4, pix)) * C + 8 * piece;
  }
  int toff[ALTM_BPW][4];   // -1: no such target
  const float inv_cw = 1.0f / (float)max(CW, 1);
#pragma unroll
  for (int u = 0; u < ALTM_BPW; u++) {
    const int blk = wave + ALTM_WAVES * u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int t = blk * 32 + 8 * j + (lane >> 3);
      const bool tok = (blk < nblk) && (t < NBX);
      const int tyy = tok ? (int)(((float)t + 0.5f) * inv_cw) : 0, txx = tok ? t - tyy * CW : 0;   // (t < 384: exact)
      toff[u][j] = tok ? ((cy0 + tyy) * W2 + (cx0 + txx)) * C + 8 * piece : -1;
    }
  }
  const altm_half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  altm_f16v acc[2][ALTM_BPW];   // [source half][target block]: targets x sources, both halves of the tile at once
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int u = 0; u < ALTM_BPW; u++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[h][u][r] = 0.f;
  if (boxed) {
    for (int kq = 0; kq * 64 < C; kq++) {   // channels 64 kq .. 64 kq + 63
      const int kbase = 64 * kq;
      const bool pok = kbase + 8 * piece < C;   // (C % 16 == 0: a 16-byte piece is inside or outside)
      altm_half8 sreg[2], treg[ALTM_BPW][4];
#pragma unroll
      for (int j = 0; j < 2; j++) sreg[j] = pok ? *reinterpret_cast<const altm_half8 *>(fmap1 + soff[j] + kbase) : zero8;
#pragma unroll
      for (int u = 0; u < ALTM_BPW; u++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          treg[u][j] = (pok && toff[u][j] >= 0) ? *reinterpret_cast<const altm_half8 *>(f2b + toff[u][j] + kbase) : zero8;
      if (kq) __syncthreads();   // every wave has read the previous half's source fragments
#pragma unroll
      for (int j = 0; j < 2; j++)
        *reinterpret_cast<altm_half8 *>(lds_src + ((tid >> 3) + 32 * j) * ALTM_ROW + piece * 16) = sreg[j];
      __syncthreads();
      altm_half8 sf[2][4];   // source fragments: pixel 32 h + (lane & 31), channels 16 ks + 8 (lane >> 5) .. + 7
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
          sf[h][ks] = *reinterpret_cast<const altm_half8 *>(lds_src + (32 * h + l31) * ALTM_ROW + (2 * ks + (lane >> 5)) * 16);
#pragma unroll
      for (int u = 0; u < ALTM_BPW; u++) {
        if (wave + ALTM_WAVES * u < nblk) {   // (wave-uniform)
#pragma unroll
          for (int j = 0; j < 4; j++)
            *reinterpret_cast<altm_half8 *>(lds_tgt + (8 * j + (lane >> 3)) * ALTM_ROW + piece * 16) = treg[u][j];
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int ks = 0; ks < 4; ks++) {
            const altm_half8 tfr = *reinterpret_cast<const altm_half8 *>(lds_tgt + l31 * ALTM_ROW + (2 * ks + (lane >> 5)) * 16);
#pragma unroll
            for (int h = 0; h < 2; h++)
              acc[h][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tfr, sf[h][ks], acc[h][u], 0, 0, 0);  // targets x sources
          }
          __builtin_amdgcn_wave_barrier();   // the next block overwrites the staging rows
        }
      }
    }
  }

  // window-phase role of this thread: pixel p of the current half, output row g (tap rows g, g + 1), output columns
  // ox0 .. ox0 + 3 (tap columns ox0 .. ox0 + 4): the 7 x 7 outputs of a pixel are dealt to 14 threads
  const int p = tid & 31, g = (tid >> 5) & 7, ox0 = (tid >> 8) * ALTM_NOX;
  float *const orow_base = corr + ((size_t)(bs * num_levels + lvl) * RD * RD) * HW1;

#pragma unroll
  for (int half = 0; half < 2; half++) {
    const int src_lane = half * 32 + p;   // the pixel's lane in the prologue's numbering
    const int pwx0 = __builtin_amdgcn_ds_bpermute(src_lane * 4, wx0), pwy0 = __builtin_amdgcn_ds_bpermute(src_lane * 4, wy0);
    const float pnw = __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane * 4, __float_as_int(wnw)));
    const float pne = __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane * 4, __float_as_int(wne)));
    const float psw = __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane * 4, __float_as_int(wsw)));
    const float pse = __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane * 4, __float_as_int(wse)));
    const int pflags = __builtin_amdgcn_ds_bpermute(src_lane * 4, (hits ? 1 : 0) | (inb ? 2 : 0));
    const int ppix = __builtin_amdgcn_ds_bpermute(src_lane * 4, pix);
    const bool phits = (pflags & 1) != 0, pinb = (pflags & 2) != 0;
    float taps[2][ALTM_NOX + 1];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int i = 0; i < ALTM_NOX + 1; i++) taps[r][i] = 0.f;

    if (boxed) {
      if (half) __syncthreads();   // the first half's window phase has read the tile
      // D layout: column = lane & 31 (source), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (target within the block)
#pragma unroll
      for (int u = 0; u < ALTM_BPW; u++) {
        if (wave + ALTM_WAVES * u < nblk) {
#pragma unroll
          for (int rq = 0; rq < 4; rq++) {
            altm_f4v v;
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = acc[half][u][4 * rq + q];
            *reinterpret_cast<altm_f4v *>(altm_tile + l31 * ALTM_PITCH + (wave + ALTM_WAVES * u) * 32 + 8 * rq + 4 * (lane >> 5)) = v;
          }
        }
      }
      __syncthreads();
      if (phits && g < RD) {
        // taps outside the box are outside the map: zeros.  Columns i_lo .. i_hi - 1 of the thread's taps are inside.
        const int i_lo = cx0 - (pwx0 + ox0), i_hi = cx1 - (pwx0 + ox0);
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const int ty = pwy0 + g + r;
          const bool rok = (ty >= cy0) && (ty < cy1);
          const float *trow = altm_tile + p * ALTM_PITCH + (rok ? (ty - cy0) * CW - i_lo : 0);
          const int lo = rok ? i_lo : 1 << 20;
#pragma unroll
          for (int i = 0; i < ALTM_NOX + 1; i++) taps[r][i] = (i >= lo && i < i_hi) ? trow[i] : 0.f;
        }
      }
    } else if (phits && g < RD) {
      // ---- incoherent tile: this thread's 16 taps as dot products (float accumulation of exact products) -----------------------
      const _Float16 *sp = fmap1 + ((size_t)b1 * HW1 + ppix) * C;
      for (int c = 0; c < C; c += 8) {
        const altm_half8 a = *reinterpret_cast<const altm_half8 *>(sp + c);
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const int ty = pwy0 + g + r;
#pragma unroll
          for (int i = 0; i < ALTM_NOX + 1; i++) {
            const int tx = pwx0 + ox0 + i;
            if (ty >= 0 && ty < H2 && tx >= 0 && tx < W2) {
              const altm_half8 v = *reinterpret_cast<const altm_half8 *>(f2b + ((size_t)ty * W2 + tx) * C + c);
              float sd = taps[r][i];
#pragma unroll
              for (int k = 0; k < 8; k++) sd = __builtin_fmaf((float)a[k], (float)v[k], sd);
              taps[r][i] = sd;
            }
          }
        }
      }
    }
    // ---- blend (the reference's order of a tap's four updates: se, then sw, ne, nw of the later taps) and store row g ----------
    if (pinb && g < RD) {
      float *o = orow_base + ppix + (size_t)(g + RD * ox0) * HW1;   // channel = iy + RD * ix
#pragma unroll
      for (int ox = 0; ox < ALTM_NOX; ox++) {
        float v = taps[0][ox] * pse;
        v = v + taps[0][ox + 1] * psw;
        v = v + taps[1][ox] * pne;
        v = v + taps[1][ox + 1] * pnw;
        // (no hit: zero taps; NaN coordinates give NaN weights and NaN here, like the reference)
        if (ox0 + ox < RD) o[(size_t)(RD * ox) * HW1] = v;
      }
    }
  }
}

}  // namespace dba

using namespace dba;

template <typename T>
static int altcorr_forward_launch(const void *fmap1, const void *fmap2, const float *coords, void *corr, int B, int S, int H1,
                                  int W1, int H2, int W2, int C, int radius, hipStream_t stream,
                                  const AltLevels &Lv = AltLevels{}, int num_levels = 0, const int64_t *ii = nullptr,
                                  const int64_t *jj = nullptr) {
  dim3 grid((W1 + ALT_TW - 1) / ALT_TW, (H1 + ALT_TH - 1) / ALT_TH, B * S * (num_levels > 0 ? num_levels : 1));
#define LAUNCH_R(RR)                                                                                             \
  hipLaunchKernelGGL((altcorr_forward_kernel<RR, T>), grid, dim3(64), 0, stream, static_cast<const T *>(fmap1),  \
                     static_cast<const T *>(fmap2), coords, static_cast<T *>(corr), B, S, H1, W1, H2, W2, C, Lv,  \
                     num_levels, ii, jj)
  switch (radius) {
    case 1: LAUNCH_R(1); break;
    case 2: LAUNCH_R(2); break;
    case 3: LAUNCH_R(3); break;
    case 4: LAUNCH_R(4); break;
    default: return DBA_ERR_UNSUPPORTED;
  }
#undef LAUNCH_R
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

extern "C" int dba_altcorr_forward_t(const void *fmap1, const void *fmap2, const float *coords, void *corr, int B, int S,
                                     int H1, int W1, int H2, int W2, int C, int radius, int dtype, dba_stream_t stream) {
  if (B < 0 || S < 0 || H1 <= 0 || W1 <= 0 || H2 <= 0 || W2 <= 0 || C <= 0) return DBA_ERR_ARG;
  if ((long)B * S == 0) return DBA_OK;
  if ((long)B * S > 65535) return DBA_ERR_UNSUPPORTED;
  if (!fmap1 || !fmap2 || !coords || !corr) return DBA_ERR_ARG;
  if (dtype == DBA_F32)
    return altcorr_forward_launch<float>(fmap1, fmap2, coords, corr, B, S, H1, W1, H2, W2, C, radius, (hipStream_t)stream);
  if (dtype == DBA_F16)
    return altcorr_forward_launch<_Float16>(fmap1, fmap2, coords, corr, B, S, H1, W1, H2, W2, C, radius,
                                            (hipStream_t)stream);
  return DBA_ERR_UNSUPPORTED;
}

extern "C" int dba_altcorr_pyramid_forward(const void *fmap1, const void *const *fmap2_levels, const int64_t *ii,
                                           const int64_t *jj, const float *coords, void *corr, int B, int S, int H1, int W1,
                                           int C, int num_levels, int radius, int dtype, dba_stream_t stream) {
  if (B < 0 || S < 0 || H1 <= 0 || W1 <= 0 || C <= 0 || num_levels < 1 || num_levels > 8) return DBA_ERR_ARG;
  if ((long)B * S == 0) return DBA_OK;
  if ((long)B * S * num_levels > 65535) return DBA_ERR_UNSUPPORTED;
  if (!fmap1 || !fmap2_levels || !coords || !corr || !ii || !jj) return DBA_ERR_ARG;
  if ((H1 >> (num_levels - 1)) < 1 || (W1 >> (num_levels - 1)) < 1) return DBA_ERR_ARG;
  AltLevels Lv;
  for (int l = 0; l < 8; l++) Lv.f2[l] = (l < num_levels) ? fmap2_levels[l] : nullptr;
  if (dtype == DBA_F32)
    return altcorr_forward_launch<float>(fmap1, nullptr, coords, corr, B, S, H1, W1, H1, W1, C, radius, (hipStream_t)stream, Lv,
                                         num_levels, ii, jj);
  if (dtype == DBA_F16)
    return altcorr_forward_launch<_Float16>(fmap1, nullptr, coords, corr, B, S, H1, W1, H1, W1, C, radius, (hipStream_t)stream,
                                            Lv, num_levels, ii, jj);
  return DBA_ERR_UNSUPPORTED;
}

extern "C" int dba_altcorr_pyramid_forward_f16maps(const void *fmap1, const void *const *fmap2_levels, const int64_t *ii,
                                                   const int64_t *jj, const float *coords, float *corr, int B, int S, int H1,
                                                   int W1, int C, int num_levels, int radius, dba_stream_t stream) {
  if (B < 0 || S < 0 || H1 <= 0 || W1 <= 0 || C <= 0 || num_levels < 1 || num_levels > 8) return DBA_ERR_ARG;
  if (radius != 3 || (C % 16) != 0 || C > 16 * ALTM_KS) return DBA_ERR_UNSUPPORTED;
  if ((long)H1 * W1 * C >= 2147483647L) return DBA_ERR_UNSUPPORTED;   // (a level's map is addressed with 32-bit offsets)
  if ((long)B * S == 0) return DBA_OK;
  if ((long)B * S * num_levels > 65535) return DBA_ERR_UNSUPPORTED;
  if (!fmap1 || !fmap2_levels || !coords || !corr || !ii || !jj) return DBA_ERR_ARG;
  if ((H1 >> (num_levels - 1)) < 1 || (W1 >> (num_levels - 1)) < 1) return DBA_ERR_ARG;
  AltLevels Lv;
  for (int l = 0; l < 8; l++) Lv.f2[l] = (l < num_levels) ? fmap2_levels[l] : nullptr;
  dim3 grid((W1 + ALT_TW - 1) / ALT_TW, (H1 + ALT_TH - 1) / ALT_TH, B * S * num_levels);
  hipLaunchKernelGGL(altcorr_mfma_kernel, grid, dim3(64 * ALTM_WAVES), (size_t)ALTM_LDS_BYTES, (hipStream_t)stream,
                     static_cast<const _Float16 *>(fmap1), Lv, coords, corr, S, H1, W1, C, num_levels, ii, jj);
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

extern "C" int dba_altcorr_forward(const float *fmap1, const float *fmap2, const float *coords, float *corr,
                                   int B, int S, int H1, int W1, int H2, int W2, int C, int radius,
                                   dba_stream_t stream) {
  return dba_altcorr_forward_t(fmap1, fmap2, coords, corr, B, S, H1, W1, H2, W2, C, radius, DBA_F32, stream);
}

extern "C" int dba_altcorr_backward(const float *fmap1, const float *fmap2, const float *coords,
                                    const float *corr_grad, float *fmap1_grad, float *fmap2_grad, int B, int S,
                                    int H1, int W1, int H2, int W2, int C, int radius, dba_stream_t stream) {
  if (B < 0 || S < 0 || H1 <= 0 || W1 <= 0 || H2 <= 0 || W2 <= 0 || C <= 0) return DBA_ERR_ARG;
  const long total = (long)B * H1 * W1 * ((C + 31) / 32);
  if (total == 0 || S == 0) return DBA_OK;
  dim3 grid((unsigned)((total + 255) / 256));
#define LAUNCH_R(RR)                                                                                           \
  hipLaunchKernelGGL((altcorr_backward_kernel<RR>), grid, dim3(256), 0, (hipStream_t)stream, fmap1, fmap2, coords, \
                     corr_grad, fmap1_grad, fmap2_grad, B, S, H1, W1, H2, W2, C)
  switch (radius) {
    case 1: LAUNCH_R(1); break;
    case 2: LAUNCH_R(2); break;
    case 3: LAUNCH_R(3); break;
    case 4: LAUNCH_R(4); break;
    default: return DBA_ERR_UNSUPPORTED;
  }
#undef LAUNCH_R
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}
