// On-the-fly windowed correlation (no materialised volume), gfx950.
//
// Replaces altcorr_forward_kernel (/root/reference/src/altcorr_kernel.cu:27-149, launcher :290-319),
// the op behind AltCorrBlock (/root/reference/dbaf/modules/corr.py:91-139).  The reference runs 32-thread
// blocks that rely on NVIDIA warp-synchronous execution (no barrier between the shared-memory dot
// product and the next tap's overwrite); on wave64 hardware that assumption does not hold, so the
// kernel is organised differently: one lane owns one (batch, pixel) and keeps all (2r+1)^2 outputs of
// one coordinate set in registers; channels are walked in the reference's chunks of 32 and every
// chunk's tap dot-products are scattered with the four bilinear weights before the next chunk, so the
// fp32 accumulation order matches the reference's chunk/tap order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

// bit-exact parity with the reference arithmetic: no mul+add fusion anywhere in this file
#pragma clang fp contract(off)

namespace dba {

template <int R>
__global__ __launch_bounds__(256) void altcorr_forward_kernel(const float *__restrict__ fmap1,
                                                              const float *__restrict__ fmap2,
                                                              const float *__restrict__ coords,
                                                              float *__restrict__ corr, int B, int S, int H1,
                                                              int W1, int H2, int W2, int C) {
  constexpr int RD = 2 * R + 1;
  const int HW1 = H1 * W1;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)B * S * HW1) return;
  const int pix = (int)(gid % HW1);
  const int s = (int)((gid / HW1) % S);
  const int b = (int)(gid / ((long)HW1 * S));

  const float *cp = coords + (((size_t)b * S + s) * HW1 + pix) * 2;
  const float x2 = cp[0], y2 = cp[1];
  const float fxf = floorf(x2), fyf = floorf(y2);
  const float dx = x2 - fxf, dy = y2 - fyf;
  const bool sane = (fabsf(x2) < 1.0e6f) && (fabsf(y2) < 1.0e6f);
  const int w0 = sane ? (int)fxf - R : -(1 << 20), h0 = sane ? (int)fyf - R : -(1 << 20);
  const float wnw = dy * dx, wne = dy * (1 - dx), wsw = (1 - dy) * dx, wse = (1 - dy) * (1 - dx);

  float acc[RD * RD];  // channel = iy + RD * ix
#pragma unroll
  for (int i = 0; i < RD * RD; i++) acc[i] = 0.f;

  const float *f1 = fmap1 + ((size_t)b * HW1 + pix) * C;
  for (int c = 0; c < C; c += 32) {
    const int cn = min(32, C - c);
    float a[32];
#pragma unroll
    for (int k = 0; k < 32; k++) a[k] = (k < cn) ? f1[c + k] : 0.f;
#pragma unroll
    for (int iy = 0; iy < RD + 1; iy++) {
#pragma unroll
      for (int ix = 0; ix < RD + 1; ix++) {
        const int h2 = h0 + iy, w2 = w0 + ix;
        float sdot = 0.f;
        if (h2 >= 0 && h2 < H2 && w2 >= 0 && w2 < W2) {
          const float *f2 = fmap2 + (((size_t)b * H2 + h2) * W2 + w2) * C + c;
#pragma unroll
          for (int k = 0; k < 32; k++) sdot = __fadd_rn(sdot, __fmul_rn(a[k], (k < cn) ? f2[k] : 0.f));
        }
        if (iy > 0 && ix > 0) acc[(iy - 1) + RD * (ix - 1)] = __fadd_rn(acc[(iy - 1) + RD * (ix - 1)], __fmul_rn(sdot, wnw));
        if (iy > 0 && ix < RD) acc[(iy - 1) + RD * ix] = __fadd_rn(acc[(iy - 1) + RD * ix], __fmul_rn(sdot, wne));
        if (iy < RD && ix > 0) acc[iy + RD * (ix - 1)] = __fadd_rn(acc[iy + RD * (ix - 1)], __fmul_rn(sdot, wsw));
        if (iy < RD && ix < RD) acc[iy + RD * ix] = __fadd_rn(acc[iy + RD * ix], __fmul_rn(sdot, wse));
      }
    }
  }
  float *o = corr + (((size_t)b * S + s) * RD * RD) * HW1 + pix;
#pragma unroll
  for (int i = 0; i < RD * RD; i++) o[(size_t)i * HW1] = acc[i];
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

}  // namespace dba

using namespace dba;

extern "C" int dba_altcorr_forward(const float *fmap1, const float *fmap2, const float *coords, float *corr,
                                   int B, int S, int H1, int W1, int H2, int W2, int C, int radius,
                                   dba_stream_t stream) {
  if (B < 0 || S < 0 || H1 <= 0 || W1 <= 0 || H2 <= 0 || W2 <= 0 || C <= 0) return DBA_ERR_ARG;
  const long total = (long)B * S * H1 * W1;
  if (total == 0) return DBA_OK;
  dim3 grid((unsigned)((total + 255) / 256));
#define LAUNCH_R(RR)                                                                                          \
  hipLaunchKernelGGL((altcorr_forward_kernel<RR>), grid, dim3(256), 0, (hipStream_t)stream, fmap1, fmap2, coords, \
                     corr, B, S, H1, W1, H2, W2, C)
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
