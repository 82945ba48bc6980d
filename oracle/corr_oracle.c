/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU oracle for the correlation half of the hot path:
 *   - all-pairs volume + 4-level average-pool pyramid
 *       (/root/reference/dbaf/modules/corr.py:24-38, :63-71)
 *   - windowed bilinear lookup from the volume
 *       (/root/reference/src/correlation_kernels.cu:19-70)
 *   - its adjoint (correlation_kernels.cu:73-124)
 *   - on-the-fly windowed correlation
 *       (/root/reference/src/altcorr_kernel.cu:27-149)
 * fp16 data is carried as uint16_t bit patterns; c10::Half arithmetic in the
 * reference is float arithmetic with a round-to-nearest-even conversion back to
 * half after every operator, which is what the *_f16 variants do.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load this.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "half.h"

static inline int within_bounds(int h, int w, int H, int W) { /* correlation_kernels.cu:15-17 */
  return h >= 0 && h < H && w >= 0 && w < W;
}

/* ---- all-pairs volume, corr.py:63-71 -------------------------------------------
 * fmap1,fmap2: [n][C][HW]  (the reference reshapes [batch,num,dim,ht,wd] -> [bn,dim,ht*wd],
 * divides both by 4.0 and computes fmap1^T fmap2).  out: [n][HW1][HW2]. */
void oracle_corr_volume_f32(const float *f1, const float *f2, int n, int C, int HW1, int HW2, float *out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int e = 0; e < n; e++)
    for (int p1 = 0; p1 < HW1; p1++) {
      const float *a = f1 + (size_t)e * C * HW1, *b = f2 + (size_t)e * C * HW2;
      float *o = out + ((size_t)e * HW1 + p1) * HW2;
      for (int p2 = 0; p2 < HW2; p2++) {
        double acc = 0.0; /* BLAS accumulation order is unspecified; double is the arbiter */
        for (int c = 0; c < C; c++)
          acc += (double)(a[(size_t)c * HW1 + p1] / 4.0f) * (double)(b[(size_t)c * HW2 + p2] / 4.0f);
        o[p2] = (float)acc;
      }
    }
}

/* half in / half out: inputs are scaled in half (x/4 is exact unless subnormal),
 * products accumulate in fp32 (GPU BLAS half GEMM accumulates in fp32), one rounding
 * to half at the end. */
void oracle_corr_volume_f16(const uint16_t *f1, const uint16_t *f2, int n, int C, int HW1, int HW2,
                            uint16_t *out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int e = 0; e < n; e++)
    for (int p1 = 0; p1 < HW1; p1++) {
      const uint16_t *a = f1 + (size_t)e * C * HW1, *b = f2 + (size_t)e * C * HW2;
      uint16_t *o = out + ((size_t)e * HW1 + p1) * HW2;
      float av[1024];
      for (int c = 0; c < C && c < 1024; c++)
        av[c] = half_to_float(float_to_half(half_to_float(a[(size_t)c * HW1 + p1]) / 4.0f));
      for (int p2 = 0; p2 < HW2; p2++) {
        float acc = 0.0f;
        for (int c = 0; c < C; c++) {
          const float bv = half_to_float(float_to_half(half_to_float(b[(size_t)c * HW2 + p2]) / 4.0f));
          acc += av[c] * bv;
        }
        o[p2] = float_to_half(acc);
      }
    }
}

/* F.avg_pool2d(corr, 2, stride=2) over the trailing (h2, w2) plane, corr.py:35-38.
 * in: [planes][h2][w2] -> out: [planes][h2/2][w2/2] (floor). */
void oracle_avg_pool2_f32(const float *in, size_t planes, int h2, int w2, float *out) {
  const int ho = h2 / 2, wo = w2 / 2;
#pragma omp parallel for schedule(static)
  for (size_t p = 0; p < planes; p++)
    for (int y = 0; y < ho; y++)
      for (int x = 0; x < wo; x++) {
        const float *s = in + p * h2 * w2;
        const float sum = s[(2 * y) * w2 + 2 * x] + s[(2 * y) * w2 + 2 * x + 1] +
                          s[(2 * y + 1) * w2 + 2 * x] + s[(2 * y + 1) * w2 + 2 * x + 1];
        out[(p * ho + y) * wo + x] = sum / 4.0f;
      }
}

void oracle_avg_pool2_f16(const uint16_t *in, size_t planes, int h2, int w2, uint16_t *out) {
  const int ho = h2 / 2, wo = w2 / 2;
#pragma omp parallel for schedule(static)
  for (size_t p = 0; p < planes; p++)
    for (int y = 0; y < ho; y++)
      for (int x = 0; x < wo; x++) {
        const uint16_t *s = in + p * h2 * w2;
        /* ATen's half avg_pool2d accumulates in float (acc_type) and rounds once */
        const float sum = half_to_float(s[(2 * y) * w2 + 2 * x]) + half_to_float(s[(2 * y) * w2 + 2 * x + 1]) +
                          half_to_float(s[(2 * y + 1) * w2 + 2 * x]) +
                          half_to_float(s[(2 * y + 1) * w2 + 2 * x + 1]);
        out[(p * ho + y) * wo + x] = float_to_half(sum / 4.0f);
      }
}

/* ---- corr_index_forward_kernel, correlation_kernels.cu:19-70 --------------------
 * volume [n][h1][w1][h2][w2], coords [n][2][h1][w1] (ch0 = x, ch1 = y),
 * corr [n][rd][rd][h1][w1] zero-initialised by the launcher (:142-143). */
#define LOOKUP_BODY(LOAD, MULW, ADDTO)                                                             \
  const int rd = 2 * r + 1;                                                                        \
  const size_t HW1 = (size_t)h1 * w1;                                                              \
  _Pragma("omp parallel for collapse(2) schedule(static)") for (int e = 0; e < n; e++) for (int y = 0; y < h1; y++) \
    for (int x = 0; x < w1; x++) {                                                                 \
      const float x0 = coords[((size_t)e * 2 + 0) * HW1 + (size_t)y * w1 + x];                     \
      const float y0 = coords[((size_t)e * 2 + 1) * HW1 + (size_t)y * w1 + x];                     \
      const float dx = x0 - floorf(x0), dy = y0 - floorf(y0);                                      \
      for (int i = 0; i < rd + 1; i++)                                                             \
        for (int j = 0; j < rd + 1; j++) {                                                         \
          const int x1 = (int)floorf(x0) - r + i, y1 = (int)floorf(y0) - r + j;                    \
          if (!within_bounds(y1, x1, h2, w2)) continue;                                            \
          LOAD(volume[((((size_t)e * h1 + y) * w1 + x) * h2 + y1) * w2 + x1]);                     \
          if (i > 0 && j > 0) ADDTO(i - 1, j - 1, MULW(dx * dy));                                  \
          if (i > 0 && j < rd) ADDTO(i - 1, j, MULW(dx * (1.0f - dy)));                            \
          if (i < rd && j > 0) ADDTO(i, j - 1, MULW((1.0f - dx) * dy));                            \
          if (i < rd && j < rd) ADDTO(i, j, MULW((1.0f - dx) * (1.0f - dy)));                      \
        }                                                                                          \
    }

void oracle_corr_index_forward_f32(const float *volume, const float *coords, float *corr, int n, int h1,
                                   int w1, int h2, int w2, int r) {
  memset(corr, 0, sizeof(float) * (size_t)n * (2 * r + 1) * (2 * r + 1) * h1 * w1);
#define LOAD_(v) const float s = (v)
#define MULW_(w) (s * (float)(w))
#define ADD_(a, b, val) corr[((((size_t)e * rd + (a)) * rd + (b)) * h1 + y) * w1 + x] += (val)
  LOOKUP_BODY(LOAD_, MULW_, ADD_)
#undef LOAD_
#undef MULW_
#undef ADD_
}

void oracle_corr_index_forward_f16(const uint16_t *volume, const float *coords, uint16_t *corr, int n,
                                   int h1, int w1, int h2, int w2, int r) {
  memset(corr, 0, sizeof(uint16_t) * (size_t)n * (2 * r + 1) * (2 * r + 1) * h1 * w1);
  /* s * scalar_t(w): weight rounded to half, product in float rounded to half;
   * corr += p: float add rounded to half (c10::Half operators) */
#define LOAD_(v) const float s = half_to_float(v)
#define MULW_(w) half_to_float(float_to_half(s * half_to_float(float_to_half((float)(w)))))
#define ADD_(a, b, val)                                                                            \
  do {                                                                                             \
    uint16_t *c_ = &corr[((((size_t)e * rd + (a)) * rd + (b)) * h1 + y) * w1 + x];                 \
    *c_ = float_to_half(half_to_float(*c_) + (val));                                               \
  } while (0)
  LOOKUP_BODY(LOAD_, MULW_, ADD_)
#undef LOAD_
#undef MULW_
#undef ADD_
}

/* ---- corr_index_backward_kernel, correlation_kernels.cu:73-124 (float) -------- */
void oracle_corr_index_backward_f32(const float *coords, const float *corr_grad, float *volume_grad,
                                    int n, int h1, int w1, int h2, int w2, int r) {
  const int rd = 2 * r + 1;
  const size_t HW1 = (size_t)h1 * w1;
  memset(volume_grad, 0, sizeof(float) * (size_t)n * HW1 * h2 * w2);
#pragma omp parallel for collapse(2) schedule(static)
  for (int e = 0; e < n; e++)
    for (int y = 0; y < h1; y++)
      for (int x = 0; x < w1; x++) {
        const float x0 = coords[((size_t)e * 2 + 0) * HW1 + (size_t)y * w1 + x];
        const float y0 = coords[((size_t)e * 2 + 1) * HW1 + (size_t)y * w1 + x];
        const float dx = x0 - floorf(x0), dy = y0 - floorf(y0);
        for (int i = 0; i < rd + 1; i++)
          for (int j = 0; j < rd + 1; j++) {
            const int x1 = (int)floorf(x0) - r + i, y1 = (int)floorf(y0) - r + j;
            if (!within_bounds(y1, x1, h2, w2)) continue;
            float g = 0.0f;
#define CG_(a, b) corr_grad[((((size_t)e * rd + (a)) * rd + (b)) * h1 + y) * w1 + x]
            if (i > 0 && j > 0) g += CG_(i - 1, j - 1) * (dx * dy);
            if (i > 0 && j < rd) g += CG_(i - 1, j) * (dx * (1.0f - dy));
            if (i < rd && j > 0) g += CG_(i, j - 1) * ((1.0f - dx) * dy);
            if (i < rd && j < rd) g += CG_(i, j) * ((1.0f - dx) * (1.0f - dy));
#undef CG_
            volume_grad[((((size_t)e * h1 + y) * w1 + x) * h2 + y1) * w2 + x1] += g;
          }
      }
}

/* ---- altcorr_forward_kernel, altcorr_kernel.cu:27-149 (float) -------------------
 * fmap1 [B][H1][W1][C], fmap2 [B][H2][W2][C] (channels last), coords [B][S][H1][W1][2],
 * corr [B][S][rd*rd][H1][W1], channel = iy + rd*ix (:102-105).  Channels are
 * processed in chunks of 32 (CHANNEL_STRIDE :19); each chunk's 32-long dot product
 * is scattered with the four bilinear weights before the next chunk starts. */
void oracle_altcorr_forward_f32(const float *fmap1, const float *fmap2, const float *coords, float *corr,
                                int B, int S, int H1, int W1, int H2, int W2, int C, int r) {
  const int rd = 2 * r + 1;
  const size_t HW1 = (size_t)H1 * W1;
  memset(corr, 0, sizeof(float) * (size_t)B * S * rd * rd * HW1);
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; b++)
    for (int h1 = 0; h1 < H1; h1++)
      for (int w1 = 0; w1 < W1; w1++)
        for (int c = 0; c < C; c += 32)
          for (int s = 0; s < S; s++) {
            const float *cp = coords + ((((size_t)b * S + s) * H1 + h1) * W1 + w1) * 2;
            const float x2 = cp[0], y2 = cp[1];
            const float dx = x2 - floorf(x2), dy = y2 - floorf(y2);
            const float *f1 = fmap1 + (((size_t)b * H1 + h1) * W1 + w1) * C + c;
            float *cr = corr + (((size_t)b * S + s) * rd * rd) * HW1 + (size_t)h1 * W1 + w1;
            for (int iy = 0; iy < rd + 1; iy++)
              for (int ix = 0; ix < rd + 1; ix++) {
                const int h2 = (int)floorf(y2) - r + iy, w2 = (int)floorf(x2) - r + ix;
                float sdot = 0.0f;
                if (within_bounds(h2, w2, H2, W2)) {
                  const float *f2 = fmap2 + (((size_t)b * H2 + h2) * W2 + w2) * C + c;
                  for (int k = 0; k < 32 && c + k < C; k++) sdot += f1[k] * f2[k];
                }
                const float nw = sdot * (dy * dx), ne = sdot * (dy * (1 - dx));
                const float sw = sdot * ((1 - dy) * dx), se = sdot * ((1 - dy) * (1 - dx));
                if (iy > 0 && ix > 0) cr[(size_t)((iy - 1) + rd * (ix - 1)) * HW1] += nw;
                if (iy > 0 && ix < rd) cr[(size_t)((iy - 1) + rd * ix) * HW1] += ne;
                if (iy < rd && ix > 0) cr[(size_t)(iy + rd * (ix - 1)) * HW1] += sw;
                if (iy < rd && ix < rd) cr[(size_t)(iy + rd * ix) * HW1] += se;
              }
          }
}

/* the half instantiation of the same kernel (AT_DISPATCH_FLOATING_TYPES_AND_HALF, altcorr_kernel.cu:304): scalar_t =
 * c10::Half, i.e. every product and every sum is computed in float and rounded to half (`s += f1 * f2` :102-103, the
 * four weighted terms :112-115 with static_cast<scalar_t>(dy * dx), the read-modify-writes of corr :131-141);
 * x2s, y2s, dx, dy stay float (:45-46, :70-71). */
void oracle_altcorr_forward_f16(const uint16_t *fmap1, const uint16_t *fmap2, const float *coords, uint16_t *corr,
                                int B, int S, int H1, int W1, int H2, int W2, int C, int r) {
  const int rd = 2 * r + 1;
  const size_t HW1 = (size_t)H1 * W1;
  memset(corr, 0, sizeof(uint16_t) * (size_t)B * S * rd * rd * HW1);
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; b++)
    for (int h1 = 0; h1 < H1; h1++)
      for (int w1 = 0; w1 < W1; w1++)
        for (int c = 0; c < C; c += 32)
          for (int s = 0; s < S; s++) {
            const float *cp = coords + ((((size_t)b * S + s) * H1 + h1) * W1 + w1) * 2;
            const float x2 = cp[0], y2 = cp[1];
            const float dx = x2 - floorf(x2), dy = y2 - floorf(y2);
            const uint16_t *f1 = fmap1 + (((size_t)b * H1 + h1) * W1 + w1) * C + c;
            uint16_t *cr = corr + (((size_t)b * S + s) * rd * rd) * HW1 + (size_t)h1 * W1 + w1;
            const uint16_t wnw = float_to_half(dy * dx), wne = float_to_half(dy * (1 - dx));
            const uint16_t wsw = float_to_half((1 - dy) * dx), wse = float_to_half((1 - dy) * (1 - dx));
            for (int iy = 0; iy < rd + 1; iy++)
              for (int ix = 0; ix < rd + 1; ix++) {
                const int h2 = (int)floorf(y2) - r + iy, w2 = (int)floorf(x2) - r + ix;
                uint16_t sdot = 0;
                if (within_bounds(h2, w2, H2, W2)) {
                  const uint16_t *f2 = fmap2 + (((size_t)b * H2 + h2) * W2 + w2) * C + c;
                  for (int k = 0; k < 32 && c + k < C; k++) {
                    const uint16_t prod = float_to_half(half_to_float(f1[k]) * half_to_float(f2[k]));
                    sdot = float_to_half(half_to_float(sdot) + half_to_float(prod));
                  }
                }
#define HMUL_(a, b) float_to_half(half_to_float(a) * half_to_float(b))
#define HACC_(dst, v) (dst) = float_to_half(half_to_float(dst) + half_to_float(v))
                if (iy > 0 && ix > 0) HACC_(cr[(size_t)((iy - 1) + rd * (ix - 1)) * HW1], HMUL_(sdot, wnw));
                if (iy > 0 && ix < rd) HACC_(cr[(size_t)((iy - 1) + rd * ix) * HW1], HMUL_(sdot, wne));
                if (iy < rd && ix > 0) HACC_(cr[(size_t)(iy + rd * (ix - 1)) * HW1], HMUL_(sdot, wsw));
                if (iy < rd && ix < rd) HACC_(cr[(size_t)(iy + rd * ix) * HW1], HMUL_(sdot, wse));
#undef HMUL_
#undef HACC_
              }
          }
}

/* ---- altcorr_backward_kernel, altcorr_kernel.cu:152-286 (float; launcher :321-356) ---------------
 * fmap1_grad [B][H1][W1][C], fmap2_grad [B][H2][W2][C] (zero-initialised here like the launcher does);
 * coords_grad is allocated but never written by the reference (stays zero). */
void oracle_altcorr_backward_f32(const float *fmap1, const float *fmap2, const float *coords,
                                 const float *corr_grad, float *fmap1_grad, float *fmap2_grad, int B, int S,
                                 int H1, int W1, int H2, int W2, int C, int r) {
  const int rd = 2 * r + 1;
  const size_t HW1 = (size_t)H1 * W1;
  memset(fmap1_grad, 0, sizeof(float) * (size_t)B * HW1 * C);
  memset(fmap2_grad, 0, sizeof(float) * (size_t)B * H2 * W2 * C);
  for (int b = 0; b < B; b++)
    for (int h1 = 0; h1 < H1; h1++)
      for (int w1 = 0; w1 < W1; w1++)
        for (int c = 0; c < C; c += 32) {
          const float *f1 = fmap1 + (((size_t)b * H1 + h1) * W1 + w1) * C + c;
          float f1g[32];
          for (int k = 0; k < 32; k++) f1g[k] = 0.0f;
          for (int s = 0; s < S; s++) {
            const float *cp = coords + ((((size_t)b * S + s) * H1 + h1) * W1 + w1) * 2;
            const float x2 = cp[0], y2 = cp[1];
            const float dx = x2 - floorf(x2), dy = y2 - floorf(y2);
            const float *gp = corr_grad + (((size_t)b * S + s) * rd * rd) * HW1 + (size_t)h1 * W1 + w1;
            for (int iy = 0; iy < rd + 1; iy++)
              for (int ix = 0; ix < rd + 1; ix++) {
                const int h2 = (int)floorf(y2) - r + iy, w2 = (int)floorf(x2) - r + ix;
                float g = 0.0f;
                if (iy > 0 && ix > 0) g += gp[(size_t)((iy - 1) + rd * (ix - 1)) * HW1] * dy * dx;
                if (iy > 0 && ix < rd) g += gp[(size_t)((iy - 1) + rd * ix) * HW1] * dy * (1 - dx);
                if (iy < rd && ix > 0) g += gp[(size_t)(iy + rd * (ix - 1)) * HW1] * (1 - dy) * dx;
                if (iy < rd && ix < rd) g += gp[(size_t)(iy + rd * ix) * HW1] * (1 - dy) * (1 - dx);
                if (!within_bounds(h2, w2, H2, W2)) continue; /* f2 = 0: no f1 grad; no atomicAdd */
                const float *f2 = fmap2 + (((size_t)b * H2 + h2) * W2 + w2) * C + c;
                float *f2g = fmap2_grad + (((size_t)b * H2 + h2) * W2 + w2) * C + c;
                for (int k = 0; k < 32 && c + k < C; k++) {
                  f1g[k] += g * f2[k];
                  f2g[k] += g * f1[k];
                }
              }
          }
          float *o = fmap1_grad + (((size_t)b * H1 + h1) * W1 + w1) * C + c;
          for (int k = 0; k < 32 && c + k < C; k++) o[k] += f1g[k];
        }
}
