// All-pairs correlation volume + average-pool pyramid, gfx950 MFMA.
//
// Replaces CorrBlock.corr (torch.matmul) and the avg_pool2d pyramid of CorrBlock.__init__
// (/root/reference/dbaf/modules/corr.py:24-38, :63-71):
//     corr[e, p1, p2] = sum_c (f1[e,c,p1]/4) * (f2[e,c,p2]/4)          fp16 in, fp32 accumulate, fp16 out
//     level l+1 = 2x2 average of the ROUNDED level l over the (h2, w2) plane
// This is the one genuine contraction on the hot path (4.3 GFLOP / edge at 64x64, K = 128), so it runs
// on v_mfma_f32_32x32x16_f16; it is still write-bound (46.7 MB/edge, 92 FLOP/B), which is why the
// store path matters more than the MFMA schedule.
//
// Stage A: re-lay both feature maps k-block-major [C/16][HW][16] and apply the /4 in half: the fragment of one
//          16-deep MFMA k-step for 32 consecutive pixels is then 1 KB contiguous, so a wave's 16-byte-per-lane
//          fragment load touches 8 full cache lines (pixel-major [HW][C] touched 32 lines for 32 B each, and the
//          vector L1 handles one line per cycle: fragment loads were a third of the fused kernel's time).
// Stage B: 128x128 output tile per workgroup, four waves of 64x64 (2x2 MFMA tiles), the whole K in
//          registers' reach (C = 128 -> 8 MFMA k-steps); fragments are 16-byte loads straight from the
//          re-laid maps (L2-resident: 1 MB per map).
// Stage C: pyramid levels by 2x2 averaging in fp32 of the rounded halves (== ATen's half avg_pool2d).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace dba {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

// [n][C][HW] -> [n][C/kb][HW][kb] (kb = 16 everywhere today; blocks of 8 for the target map, so that a half wave's
// fragment loads are 512 contiguous bytes, were measured and change nothing), value / 4 rounded to half
// (corr.py:67-68); C is a multiple of 16.  Reads as 16-byte pieces along the pixels, writes as 16-byte pieces along the
// channels (round 3; 2-byte accesses before: 23.4 us per 32-edge map, two launches).
// w_tiled > 0: the output pixel axis is in the 4 x 16 tile order of common.h (the source operand of the flow-aligned build,
// whose 64-pixel strips are then tiles of the map); 0: linear.
// HWo / w_grid (w_tiled > 0): the output's pixels per map and the width of the grid its tiles are counted on -- the map's own
// (HW, w_tiled), or the padded grid of common.h (the pad pixels' rows are never written: whatever they hold only ever reaches the
// pad pixels' own entries of the planes)
__device__ __forceinline__ void fmap_pixel_major_body(const _Float16 *__restrict__ in, _Float16 *__restrict__ out, int C, int HW,
                                                      int kb, int w_tiled, int e, int HWo, int w_grid) {
  __shared__ _Float16 tile[64][72];   // [channel][pixel]; pitch 144 B: the 8 lanes of a channel row write 16 B each
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const _Float16 *src = in + (size_t)e * C * HW;
  _Float16 *dst = out + (size_t)e * HWo * C;
  const int t = threadIdx.x;
  const bool vec_in = ((HW & 7) == 0) && (p0 + 64 <= HW);
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {  // 32 channel rows per pass: lane -> (row, 8-pixel piece)
    const int r = pass * 32 + (t >> 3), piece = t & 7;
    const int c = c0 + r, p = p0 + 8 * piece;
    half8 v;
    if (c < C && vec_in) {
      v = *reinterpret_cast<const half8 *>(src + (size_t)c * HW + p);
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = (c < C && p + k < HW) ? src[(size_t)c * HW + p + k] : (_Float16)0;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) tile[r][8 * piece + k] = (_Float16)((float)v[k] / 4.0f);
  }
  __syncthreads();
  // lane -> (pixel, 8 consecutive channels): kb = 16 keeps the two halves of a pixel's 16-channel block adjacent
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    const int idx = pass * 256 + t;        // 64 pixels x 8 channel octets
    const int pl = idx >> 3, oct = idx & 7;
    const int pin = p0 + pl, c = c0 + 8 * oct;
    if (pin < HW && c < C) {
      const int p = w_tiled ? sh_pixel_index(pin / w_tiled, pin % w_tiled, w_grid, true) : pin;
      half8 v;
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = tile[8 * oct + k][pl];
      if ((kb & 7) == 0 && c + 8 <= C) {
        *reinterpret_cast<half8 *>(dst + ((size_t)(c / kb) * HWo + p) * kb + (c % kb)) = v;
      } else {
#pragma unroll
        for (int k = 0; k < 8; k++)
          if (c + k < C) dst[((size_t)((c + k) / kb) * HWo + p) * kb + ((c + k) % kb)] = v[k];
      }
    }
  }
}

__global__ __launch_bounds__(256) void fmap_pixel_major_kernel(const _Float16 *__restrict__ in,
                                                               _Float16 *__restrict__ out, int C, int HW, int kb,
                                                               int w_tiled, int HWo, int w_grid) {
  fmap_pixel_major_body(in, out, C, HW, kb, w_tiled, (int)blockIdx.z, HWo, w_grid);
}

// both maps of a build in ONE launch (maps of the same size: grid.z = 2 n; the first n slices are map 1 in the pixel order
// w_tiled1 asks for, the others map 2, linear): a launch and its gap less in front of the small builds (one edge of the motion
// filter, the six of a new keyframe)
__global__ __launch_bounds__(256) void fmap_pixel_major_pair_kernel(const _Float16 *__restrict__ in1, _Float16 *__restrict__ out1,
                                                                    int w_tiled1, const _Float16 *__restrict__ in2,
                                                                    _Float16 *__restrict__ out2, int C, int HW, int kb, int n,
                                                                    int HWo1, int w_grid1) {
  const int z = (int)blockIdx.z;
  if (z < n) fmap_pixel_major_body(in1, out1, C, HW, kb, w_tiled1, z, HWo1, w_grid1);
  else fmap_pixel_major_body(in2, out2, C, HW, kb, 0, z - n, HW, 0);
}

// C must be a multiple of 16.  A = fmap1 pixel-major [HW1][C], B = fmap2 pixel-major [HW2][C].
__global__ __launch_bounds__(256) void corr_gemm_kernel(const _Float16 *__restrict__ A,
                                                        const _Float16 *__restrict__ Bm,
                                                        _Float16 *__restrict__ out, int C, int HW1, int HW2) {
  const int e = blockIdx.z;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row0 = blockIdx.y * 128 + (wave >> 1) * 64;  // p1 base of this wave
  const int col0 = blockIdx.x * 128 + (wave & 1) * 64;   // p2 base of this wave
  const _Float16 *Ae = A + (size_t)e * HW1 * C;
  const _Float16 *Be = Bm + (size_t)e * HW2 * C;
  const int l31 = lane & 31, kh = (lane >> 5) * 8;

  float16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // clamp rows so edge tiles load valid memory; the stores are masked instead
  int ar[2], br[2];
#pragma unroll
  for (int t = 0; t < 2; t++) {
    ar[t] = min(row0 + t * 32 + l31, HW1 - 1);
    br[t] = min(col0 + t * 32 + l31, HW2 - 1);
  }
  for (int k = 0; k < C; k += 16) {
    half8 a[2], b[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
      a[t] = *reinterpret_cast<const half8 *>(Ae + ((size_t)(k >> 4) * HW1 + ar[t]) * 16 + kh);
      b[t] = *reinterpret_cast<const half8 *>(Be + ((size_t)(k >> 4) * HW2 + br[t]) * 16 + kh);
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  // D layout: col = lane & 31 (p2), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (p1)
  _Float16 *oe = out + (size_t)e * HW1 * HW2;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int p2 = col0 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int p1 = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (p1 < HW1 && p2 < HW2) oe[(size_t)p1 * HW2 + p2] = (_Float16)acc[i][j][r];
      }
    }
}

// F.avg_pool2d(x, 2, stride=2) over the trailing plane: in [planes][h][w] -> out [planes][h/2][w/2]
__global__ __launch_bounds__(256) void avg_pool2_kernel(const _Float16 *__restrict__ in,
                                                        _Float16 *__restrict__ out, size_t planes, int h,
                                                        int w) {
  const int ho = h / 2, wo = w / 2;
  const size_t per = (size_t)ho * wo;
  const size_t total = planes * per;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const size_t pl = idx / per;
    const int rem = (int)(idx - pl * per);
    const int y = rem / wo, x = rem - y * wo;
    const _Float16 *s = in + pl * h * w + (size_t)(2 * y) * w + 2 * x;
    const float sum = (float)s[0] + (float)s[1] + (float)s[w] + (float)s[w + 1];
    out[idx] = (_Float16)(sum / 4.0f);
  }
}

}  // namespace dba

using namespace dba;

extern "C" {

size_t dba_corr_volume_scratch_bytes(int n, int C, int h1, int w1, int h2, int w2) {
  if (n < 0 || C <= 0 || h1 <= 0 || w1 <= 0 || h2 <= 0 || w2 <= 0) return 0;
  // (the source map's copy may live on the padded grid of its planes: dba_corr_sheared_plane_elems >= h1 * w1)
  const size_t hw1 = (size_t)dba_corr_sheared_plane_elems(h1, w1);
  return align_up((size_t)n * C * hw1 * 2, 256) + align_up((size_t)n * C * h2 * w2 * 2, 256);
}

int dba_corr_volume_build(const void *fmap1, const void *fmap2, void *const *levels, int n, int C, int h1,
                          int w1, int h2, int w2, int num_levels, void *scratch, size_t scratch_bytes,
                          dba_stream_t stream) {
  if (n < 0 || C <= 0 || (C % 16) != 0 || h1 <= 0 || w1 <= 0 || h2 <= 0 || w2 <= 0 || num_levels < 1)
    return DBA_ERR_ARG;
  if (n == 0) return DBA_OK;
  if (!fmap1 || !fmap2 || !levels || !scratch) return DBA_ERR_ARG;
  if (scratch_bytes < dba_corr_volume_scratch_bytes(n, C, h1, w1, h2, w2)) return DBA_ERR_WORKSPACE;
  const int HW1 = h1 * w1, HW2 = h2 * w2;
  hipStream_t s = (hipStream_t)stream;
  _Float16 *A = static_cast<_Float16 *>(scratch);
  _Float16 *Bm = reinterpret_cast<_Float16 *>(static_cast<char *>(scratch) + align_up((size_t)n * C * HW1 * 2, 256));
  hipLaunchKernelGGL(fmap_pixel_major_kernel, dim3((HW1 + 63) / 64, (C + 63) / 64, n), dim3(256), 0, s,
                     static_cast<const _Float16 *>(fmap1), A, C, HW1, 16, 0, HW1, 0);
  hipLaunchKernelGGL(fmap_pixel_major_kernel, dim3((HW2 + 63) / 64, (C + 63) / 64, n), dim3(256), 0, s,
                     static_cast<const _Float16 *>(fmap2), Bm, C, HW2, 16, 0, HW2, 0);
  hipLaunchKernelGGL(corr_gemm_kernel, dim3((HW2 + 127) / 128, (HW1 + 127) / 128, n), dim3(256), 0, s, A, Bm,
                     static_cast<_Float16 *>(levels[0]), C, HW1, HW2);
  DBA_LAUNCH_CHECK();
  int h = h2, w = w2;
  for (int l = 1; l < num_levels; l++) {
    const size_t planes = (size_t)n * HW1;
    const size_t total = planes * (size_t)(h / 2) * (w / 2);
    if (total > 0) {
      const unsigned blocks = (unsigned)((total + 255) / 256 < 65535u * 16u ? (total + 255) / 256 : 65535u * 16u);
      hipLaunchKernelGGL(avg_pool2_kernel, dim3(blocks), dim3(256), 0, s,
                         static_cast<const _Float16 *>(levels[l - 1]), static_cast<_Float16 *>(levels[l]), planes,
                         h, w);
      DBA_LAUNCH_CHECK();
    }
    h /= 2;
    w /= 2;
  }
  return DBA_OK;
}

}  // extern "C"
