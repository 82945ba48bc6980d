// Windowed bilinear lookup from the all-pairs correlation volume, gfx950.
//
// Replaces corr_index_forward_kernel / corr_index_backward_kernel
// (/root/reference/src/correlation_kernels.cu:19-124) and, in its fused form, the four per-level
// launches + torch.cat of CorrBlock.__call__ (/root/reference/dbaf/modules/corr.py:40-50).
//
// The reference scatters every tap into four outputs with read-modify-write `+=` on c10::Half in global
// memory (4 RMW x 64 taps per pixel).  Here one lane owns one (edge, pixel, level): it pulls the
// (2r+2)^2 window of its private (h2,w2) plane into registers with 16-byte row loads (the plane rows
// are the only HBM traffic: 8 x 16 B per pixel and level), forms each of the (2r+1)^2 outputs from its
// four taps in the reference's accumulation order with the same round-to-half after every operation
// (bit-exact), and writes every output once, coalesced across the wave.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

#include <cstdlib>
#include <type_traits>

// bit-exact parity with the reference arithmetic: no mul+add fusion anywhere in this file
#pragma clang fp contract(off)

namespace dba {

constexpr int MAX_LEVELS = 8;

struct LookupLevels {
  const void *vol[MAX_LEVELS];
};

struct __attribute__((packed, aligned(2))) Half8 {
  _Float16 v[8];
};
struct __attribute__((packed, aligned(4))) Float8 {
  float v[8];
};

template <typename T>
struct Arith;

// c10::Half semantics: every operator computes in float and rounds the result to half.
template <>
struct Arith<_Float16> {
  // scalar_t(dx * dy): the product is rounded to f32 first, then to half.  v_fma_mixlo_f16 (which the
  // compiler would pick for mul+convert) rounds the exact product once, which differs in ~1e-4 of the
  // cases, so the f32 value is pinned in a register before the conversion.
  __device__ static __forceinline__ float weight(float w) {
    asm volatile("" : "+v"(w));
    return (float)(_Float16)w;
  }
  __device__ static __forceinline__ float mul(float s, float w) { return (float)(_Float16)(s * w); }
  __device__ static __forceinline__ float add(float a, float b) { return (float)(_Float16)(a + b); }
};
template <>
struct Arith<float> {
  __device__ static __forceinline__ float weight(float w) { return w; }
  __device__ static __forceinline__ float mul(float s, float w) { return __fmul_rn(s, w); }
  __device__ static __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
};

// COORD_NHW2: coords laid out [n,h1,w1,2] (projective_transform output) instead of [n,2,h1,w1]
#ifndef CL_OCC_CFG
#define CL_OCC_CFG 1
#endif
template <typename T, int R, bool COORD_NHW2>
__global__ __launch_bounds__(256, CL_OCC_CFG) void corr_lookup_kernel(LookupLevels L, const float *__restrict__ coords,
                                                          T *__restrict__ out, int n, int h1, int w1, int h2,
                                                          int w2, int num_levels, const int *__restrict__ slots) {
  constexpr int RD = 2 * R + 1, WN = 2 * R + 2;
  const int HW1 = h1 * w1;
  const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= (long)n * HW1) return;
  const int lvl = blockIdx.y;
  const int e = (int)(pix / HW1), rem = (int)(pix - (long)e * HW1);
  const int h2l = h2 >> lvl, w2l = w2 >> lvl;

  float cx, cy;
  if constexpr (COORD_NHW2) {
    const float2 c = reinterpret_cast<const float2 *>(coords)[pix];
    cx = c.x;
    cy = c.y;
  } else {
    cx = coords[((size_t)e * 2 + 0) * HW1 + rem];
    cy = coords[((size_t)e * 2 + 1) * HW1 + rem];
  }
  // `coords / 2**i` (corr.py:47): exact power-of-two division
  const float scale = 1.0f / (float)(1 << lvl);
  const float x0 = cx * scale, y0 = cy * scale;
  const float fx = floorf(x0), fy = floorf(y0);
  const float dx = x0 - fx, dy = y0 - fy;
  // coordinates far outside any plane (or non-finite) see no in-bounds tap at all
  const bool sane = (fabsf(x0) < 1.0e6f) && (fabsf(y0) < 1.0e6f);
  const int ix0 = sane ? (int)fx - R : -(1 << 20);
  const int iy0 = sane ? (int)fy - R : -(1 << 20);

  // slots: edge e's volumes live in slot slots[e] of the level stores (null: slot e; the slot-addressed CorrBlock)
  const size_t vpix = slots ? (size_t)slots[e] * HW1 + rem : (size_t)pix;
  const T *plane = static_cast<const T *>(L.vol[lvl]) + vpix * h2l * w2l;

  float win[WN][WN];  // [row j (y)][col i (x)]
  const bool interior = (ix0 >= 0) && (iy0 >= 0) && (ix0 + WN <= w2l) && (iy0 + WN <= h2l);
  if (WN == 8 && interior) {
#pragma unroll
    for (int j = 0; j < WN; j++) {
      const T *row = plane + (size_t)(iy0 + j) * w2l + ix0;
      if constexpr (sizeof(T) == 2) {
        const Half8 v = *reinterpret_cast<const Half8 *>(row);
#pragma unroll
        for (int i = 0; i < 8; i++) win[j][i] = (float)v.v[i];
      } else {
        const Float8 v = *reinterpret_cast<const Float8 *>(row);
#pragma unroll
        for (int i = 0; i < 8; i++) win[j][i] = v.v[i];
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < WN; j++) {
      const int y1 = iy0 + j;
      const bool rok = (y1 >= 0) && (y1 < h2l);
#pragma unroll
      for (int i = 0; i < WN; i++) {
        const int x1 = ix0 + i;
        const bool ok = rok && (x1 >= 0) && (x1 < w2l);
        win[j][i] = ok ? (float)plane[(size_t)y1 * w2l + x1] : 0.f;  // out-of-bounds taps contribute +0
      }
    }
  }

  // non-finite coordinates read nothing and yield exact zeros (the reference's float->int cast is
  // undefined there)
  const float w00 = sane ? Arith<T>::weight((1.0f - dx) * (1.0f - dy)) : 0.f;
  const float w01 = sane ? Arith<T>::weight((1.0f - dx) * dy) : 0.f;
  const float w10 = sane ? Arith<T>::weight(dx * (1.0f - dy)) : 0.f;
  const float w11 = sane ? Arith<T>::weight(dx * dy) : 0.f;

  T *o = out + ((size_t)e * num_levels * RD * RD + (size_t)lvl * RD * RD) * HW1 + rem;
#pragma unroll
  for (int a = 0; a < RD; a++) {    // x offset (outer loop i of the reference)
#pragma unroll
    for (int b = 0; b < RD; b++) {  // y offset
      // accumulation order of correlation_kernels.cu:55-65 seen from output (a,b):
      //   tap(a,b)*(1-dx)(1-dy), tap(a,b+1)*(1-dx)dy, tap(a+1,b)*dx(1-dy), tap(a+1,b+1)*dx*dy
      float acc = Arith<T>::mul(win[b][a], w00);  // 0 + p == p
      acc = Arith<T>::add(acc, Arith<T>::mul(win[b + 1][a], w01));
      acc = Arith<T>::add(acc, Arith<T>::mul(win[b][a + 1], w10));
      acc = Arith<T>::add(acc, Arith<T>::mul(win[b + 1][a + 1], w11));
      o[(size_t)(a * RD + b) * HW1] = (T)acc;
    }
  }
}

// ---- half volumes, radius 3: one 16-byte load per window row ---------------------------------------------------------------
// The generic kernel above reads a border window tap by tap (64 predicated 2-byte loads per lane), and at pyramid levels 2
// and 3 (16- and 8-wide planes) nearly every window touches the border: 94-105 us per level at 96 edges for 50-100 MB of
// traffic.  Here a wave owns 64 consecutive pixels of ONE edge and addresses the edge's level volume through a raw buffer
// resource: every window row is ONE 16-byte load at the window's own (2-byte aligned) offset, rows and columns outside the
// plane are dropped by the bounds check (rows: an out-of-range offset) or masked (columns: the load then covers a neighbouring
// row of the plane), so interior and border pixels run the same branch-free code.  The blend is the packed-f16 arithmetic of
// csrc/corr_sheared.hip (channel pairs; each half rounds like the scalar operation): bit-identical to the generic kernel.
typedef _Float16 h2q __attribute__((ext_vector_type(2)));
typedef unsigned u4q __attribute__((ext_vector_type(4)));

template <bool COORD_NHW2>
__global__ __launch_bounds__(256) void corr_lookup_rows_kernel(LookupLevels L, const float *__restrict__ coords,
                                                               _Float16 *__restrict__ out, int n, int h1, int w1, int h2,
                                                               int w2, int num_levels, const int *__restrict__ slots) {
  constexpr int R = 3, RD = 7, WN = 8;
  const int HW1 = h1 * w1;
  const int strips = (HW1 + 63) >> 6;
  const int lane = threadIdx.x & 63;
  const int sid = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (edge, 64-pixel strip)
  if (sid >= n * strips) return;
  const int lvl = blockIdx.y;
  const int e = sid / strips, rem = ((sid - e * strips) << 6) + lane;
  const bool active = rem < HW1;
  const int remc = min(rem, HW1 - 1);
  const int h2l = h2 >> lvl, w2l = w2 >> lvl;
  float cx, cy;
  if constexpr (COORD_NHW2) {
    const float2 c = reinterpret_cast<const float2 *>(coords)[(size_t)e * HW1 + remc];
    cx = c.x;
    cy = c.y;
  } else {
    cx = coords[((size_t)e * 2 + 0) * HW1 + remc];
    cy = coords[((size_t)e * 2 + 1) * HW1 + remc];
  }
  const float scale = 1.0f / (float)(1 << lvl);
  const float x0 = cx * scale, y0 = cy * scale;
  const float fx = floorf(x0), fy = floorf(y0);
  const float dx = x0 - fx, dy = y0 - fy;
  const bool sane = (fabsf(x0) < 1.0e6f) && (fabsf(y0) < 1.0e6f);
  const int ix0 = sane ? (int)fx - R : -(1 << 20);
  const int iy0 = sane ? (int)fy - R : -(1 << 20);
  const bool touches = active && sane && (ix0 + WN > 0) && (ix0 < w2l) && (iy0 + WN > 0) && (iy0 < h2l);

  // this edge's level volume: [HW1][h2l][w2l] halves
  const int es = slots ? slots[e] : e;
  const size_t plane = (size_t)h2l * w2l;
  const _Float16 *vedge = static_cast<const _Float16 *>(L.vol[lvl]) + (size_t)es * HW1 * plane;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)vedge, 0, (int)(2u * (unsigned)(HW1 * plane)), 0x00020000);
  constexpr unsigned OOR = 0x80000000u;
  const int base = (int)((size_t)remc * plane) + ix0;   // halves; + ty * w2l per row (may be negative at the first pixel)

  // column keep masks per channel pair (tap 2k | tap 2k+1)
  unsigned cm[4];
#pragma unroll
  for (int k = 0; k < 4; k++)
    cm[k] = ((ix0 + 2 * k >= 0 && ix0 + 2 * k < w2l) ? 0x0000ffffu : 0u) |
            ((ix0 + 2 * k + 1 >= 0 && ix0 + 2 * k + 1 < w2l) ? 0xffff0000u : 0u);
  u4q rows[WN];
#pragma unroll
  for (int j = 0; j < WN; j++) {
    const int ty = iy0 + j;
    const int off = base + ty * w2l;
    const bool ok = touches && (ty >= 0) && (ty < h2l) && (off >= 0);
    rows[j] = __builtin_amdgcn_raw_buffer_load_b128(rin, ok ? 2u * (unsigned)off : OOR, 0, 0);
  }
  // The first and the last pixel of an edge can have a window row whose 16 bytes start before the volume (first pixel, row
  // 0, ix0 < 0) or whose last dword straddles its end (last pixel, last row, odd offset): the range check drops those dwords
  // whole, valid columns included.  Those (at most a few) lanes per edge and level re-read their window tap by tap.
  const int row_first = base + max(iy0, 0) * w2l, row_last = base + min(iy0 + WN - 1, h2l - 1) * w2l;
  const bool slow = touches && (row_first < 0 || row_last + WN > (int)(HW1 * plane));
  if (__ballot(slow) != 0ull) {
    if (slow) {
      const _Float16 *pl = vedge + (size_t)remc * plane;
#pragma unroll
      for (int j = 0; j < WN; j++) {
        const int ty = iy0 + j;
        const bool rok = (ty >= 0) && (ty < h2l);
        unsigned w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int xa = ix0 + 2 * k, xb = xa + 1;
          const _Float16 va = (rok && xa >= 0 && xa < w2l) ? pl[(size_t)ty * w2l + xa] : (_Float16)0.f;
          const _Float16 vb = (rok && xb >= 0 && xb < w2l) ? pl[(size_t)ty * w2l + xb] : (_Float16)0.f;
          h2q v;
          v.x = va;
          v.y = vb;
          w[k] = __builtin_bit_cast(unsigned, v);
        }
        rows[j][0] = w[0]; rows[j][1] = w[1]; rows[j][2] = w[2]; rows[j][3] = w[3];
      }
    }
  }

  // weights: scalar_t(f32 product), see Arith<_Float16>::weight
  float w00 = (1.0f - dx) * (1.0f - dy), w01 = (1.0f - dx) * dy, w10 = dx * (1.0f - dy), w11 = dx * dy;
  if (!touches) w00 = w01 = w10 = w11 = 0.f;
  asm volatile("" : "+v"(w00), "+v"(w01), "+v"(w10), "+v"(w11));
  h2q W00, W01, W10, W11;
  W00.x = W00.y = (_Float16)w00;
  W01.x = W01.y = (_Float16)w01;
  W10.x = W10.y = (_Float16)w10;
  W11.x = W11.y = (_Float16)w11;

  _Float16 *o = out + ((size_t)e * num_levels * RD * RD + (size_t)lvl * RD * RD) * HW1 + rem;
  auto pairs = [&](int j, h2q (&ev)[4], h2q (&od)[4]) {   // row j as channel pairs: ev[k] = (tap 2k, 2k+1), od[k] = (2k+1, 2k+2)
    const u4q r = rows[j];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const unsigned lo = r[k] & cm[k];
      ev[k] = __builtin_bit_cast(h2q, lo);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const unsigned lo = __builtin_bit_cast(unsigned, ev[k]);
      const unsigned hi = (k < 3) ? __builtin_bit_cast(unsigned, ev[k < 3 ? k + 1 : 3]) : 0u;
      od[k] = __builtin_bit_cast(h2q, __builtin_amdgcn_alignbit(hi, lo, 16));
    }
  };
  h2q pe[4], po[4], ce[4], co[4];
  pairs(0, pe, po);
#pragma unroll
  for (int b = 0; b < RD; b++) {
    pairs(b + 1, ce, co);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      h2q acc = pe[k] * W00;
      acc = acc + ce[k] * W01;
      acc = acc + po[k] * W10;
      acc = acc + co[k] * W11;
      if (active) {
        o[(size_t)((2 * k) * RD + b) * HW1] = acc.x;
        if (k < 3) o[(size_t)((2 * k + 1) * RD + b) * HW1] = acc.y;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) pe[k] = ce[k], po[k] = co[k];
  }
}

// adjoint (training only): every (pixel, tap) owns its volume_grad element, no atomics needed
template <int R>
__global__ __launch_bounds__(256) void corr_lookup_backward_kernel(const float *__restrict__ coords,
                                                                   const float *__restrict__ corr_grad,
                                                                   float *__restrict__ volume_grad, int n,
                                                                   int h1, int w1, int h2, int w2) {
  constexpr int RD = 2 * R + 1, WN = 2 * R + 2;
  const int HW1 = h1 * w1;
  const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= (long)n * HW1) return;
  const int e = (int)(pix / HW1), rem = (int)(pix - (long)e * HW1);
  const float x0 = coords[((size_t)e * 2 + 0) * HW1 + rem];
  const float y0 = coords[((size_t)e * 2 + 1) * HW1 + rem];
  const float fx = floorf(x0), fy = floorf(y0);
  const float dx = x0 - fx, dy = y0 - fy;
  const bool sane = (fabsf(x0) < 1.0e6f) && (fabsf(y0) < 1.0e6f);
  if (!sane) return;
  const int ix0 = (int)fx - R, iy0 = (int)fy - R;
  const float *g = corr_grad + (size_t)e * RD * RD * HW1 + rem;
  float *plane = volume_grad + (size_t)pix * h2 * w2;
#pragma unroll
  for (int i = 0; i < WN; i++)
#pragma unroll
    for (int j = 0; j < WN; j++) {
      const int x1 = ix0 + i, y1 = iy0 + j;
      if (x1 < 0 || x1 >= w2 || y1 < 0 || y1 >= h2) continue;
      float acc = 0.f;
      if (i > 0 && j > 0) acc = __fadd_rn(acc, __fmul_rn(g[(size_t)((i - 1) * RD + (j - 1)) * HW1], dx * dy));
      if (i > 0 && j < RD) acc = __fadd_rn(acc, __fmul_rn(g[(size_t)((i - 1) * RD + j) * HW1], dx * (1.0f - dy)));
      if (i < RD && j > 0) acc = __fadd_rn(acc, __fmul_rn(g[(size_t)(i * RD + (j - 1)) * HW1], (1.0f - dx) * dy));
      if (i < RD && j < RD) acc = __fadd_rn(acc, __fmul_rn(g[(size_t)(i * RD + j) * HW1], (1.0f - dx) * (1.0f - dy)));
      plane[(size_t)y1 * w2 + x1] += acc;
    }
}

template <typename T, bool NHW2>
static int launch_lookup(const LookupLevels &L, const float *coords, void *out, int n, int h1, int w1, int h2,
                         int w2, int num_levels, int radius, hipStream_t stream, const int *slots = nullptr) {
  const long total = (long)n * h1 * w1;
  if (total == 0) return DBA_OK;
  if constexpr (std::is_same<T, _Float16>::value) {
    static const bool generic_only = [] { const char *e = getenv("DBA_REF_LOOKUP_KERNEL"); return e && e[0] == 'g'; }();
    const size_t edge_bytes = (size_t)h1 * w1 * h2 * w2 * 2;   // level 0: the largest
    if (radius == 3 && !generic_only && edge_bytes < ((size_t)1 << 31) && (long)n * ((h1 * w1 + 63) / 64) < (1L << 31)) {
      const long strips = (long)n * ((h1 * w1 + 63) / 64);
      hipLaunchKernelGGL((corr_lookup_rows_kernel<NHW2>), dim3((unsigned)((strips + 3) / 4), num_levels), dim3(256), 0, stream, L,
                         coords, (_Float16 *)out, n, h1, w1, h2, w2, num_levels, slots);
      DBA_LAUNCH_CHECK();
      return DBA_OK;
    }
  }
  dim3 grid((unsigned)((total + 255) / 256), num_levels);
#define LAUNCH_R(RR)                                                                                     \
  hipLaunchKernelGGL((corr_lookup_kernel<T, RR, NHW2>), grid, dim3(256), 0, stream, L, coords, (T *)out, n, \
                     h1, w1, h2, w2, num_levels, slots)
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

}  // namespace dba

using namespace dba;

extern "C" {

int dba_corr_index_forward(const void *volume, const float *coords, void *corr, int n, int h1, int w1, int h2,
                           int w2, int radius, int dtype, dba_stream_t stream) {
  if (n < 0 || h1 <= 0 || w1 <= 0 || h2 <= 0 || w2 <= 0) return DBA_ERR_ARG;
  if (n > 0 && (!volume || !coords || !corr)) return DBA_ERR_ARG;
  LookupLevels L;
  for (int l = 0; l < MAX_LEVELS; l++) L.vol[l] = volume;
  if (dtype == DBA_F16)
    return launch_lookup<_Float16, false>(L, coords, corr, n, h1, w1, h2, w2, 1, radius, (hipStream_t)stream);
  if (dtype == DBA_F32)
    return launch_lookup<float, false>(L, coords, corr, n, h1, w1, h2, w2, 1, radius, (hipStream_t)stream);
  return DBA_ERR_UNSUPPORTED;
}

int dba_corr_lookup_pyramid(const void *const *volumes, const float *coords_nhw2, void *corr, int n, int h1,
                            int w1, int h2, int w2, int num_levels, int radius, int dtype,
                            dba_stream_t stream) {
  return dba_corr_lookup_pyramid_slots(volumes, nullptr, coords_nhw2, corr, n, h1, w1, h2, w2, num_levels, radius, dtype,
                                       stream);
}

int dba_corr_lookup_pyramid_slots(const void *const *volumes, const int *slots, const float *coords_nhw2, void *corr, int n,
                                  int h1, int w1, int h2, int w2, int num_levels, int radius, int dtype,
                                  dba_stream_t stream) {
  if (n < 0 || h1 <= 0 || w1 <= 0 || h2 <= 0 || w2 <= 0 || num_levels < 1 || num_levels > MAX_LEVELS)
    return DBA_ERR_ARG;
  if (n > 0 && (!volumes || !coords_nhw2 || !corr)) return DBA_ERR_ARG;
  LookupLevels L;
  for (int l = 0; l < MAX_LEVELS; l++) L.vol[l] = (l < num_levels) ? volumes[l] : nullptr;
  if (dtype == DBA_F16)
    return launch_lookup<_Float16, true>(L, coords_nhw2, corr, n, h1, w1, h2, w2, num_levels, radius,
                                         (hipStream_t)stream, slots);
  if (dtype == DBA_F32)
    return launch_lookup<float, true>(L, coords_nhw2, corr, n, h1, w1, h2, w2, num_levels, radius,
                                      (hipStream_t)stream, slots);
  return DBA_ERR_UNSUPPORTED;
}

int dba_corr_index_backward(const float *coords, const float *corr_grad, float *volume_grad, int n, int h1,
                            int w1, int h2, int w2, int radius, dba_stream_t stream) {
  if (n < 0 || h1 <= 0 || w1 <= 0 || h2 <= 0 || w2 <= 0) return DBA_ERR_ARG;
  const long total = (long)n * h1 * w1;
  if (total == 0) return DBA_OK;
  dim3 grid((unsigned)((total + 255) / 256));
#define LAUNCH_R(RR)                                                                                          \
  hipLaunchKernelGGL((corr_lookup_backward_kernel<RR>), grid, dim3(256), 0, (hipStream_t)stream, coords, corr_grad, \
                     volume_grad, n, h1, w1, h2, w2)
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

}  // extern "C"
