// Fused all-pairs correlation + 4-level pyramid + flow-aligned ("sheared") store, gfx950 MFMA.
//
// One pass replaces CorrBlock.corr (torch.matmul), the three avg_pool2d passes of CorrBlock.__init__
// (/root/reference/dbaf/modules/corr.py:24-38, :63-71) and the re-layout into the sheared volume
// (corr_sheared.hip): the reference materialises level 0 (33.5 MB/edge), reads it back three times for the
// pooling, and this repo's unfused path then reads and rewrites every level once more for the shear.
// Here every output byte is written exactly once (44.6 MB/edge) and the inputs (2 x 1 MB/edge) stay in L2.
//
// Workgroup = 8 waves; tile = 64 source pixels (one source row segment x1 = 0..63 of row y1) x 512 targets
// (8 target rows ty0..ty0+7 x w2 = 64 columns), full K = C in registers' reach:
//   * wave w owns target row ty0 + w: 64 x 64 outputs = 2 x 2 v_mfma_f32_32x32x16_f16 tiles, 16-byte
//     fragment loads straight from the pixel-major feature maps;
//   * accumulators -> f16 (the single rounding of the reference's half GEMM) -> LDS tile T[x1][ty][tx];
//   * level 0: each wave re-reads its own row along diagonals and writes 128-byte segments of the sheared
//     volume Vs0[dy][dx][y1][x1]; levels 1..3: 2x2 averages of the ROUNDED level below (== F.avg_pool2d on
//     half), kept unsheared in LDS for the next level and written sheared.
// Shapes: h1 == h2, w1 == w2 == 64, h2 % 8 == 0, C % 16 == 0, 4 levels (64x64 is the 512x512 benchmark shape);
// anything else takes the unfused path of corr_build.hip + corr_shear_kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace dba {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

struct FusedLevels {
  _Float16 *vs[4];
};

struct __attribute__((aligned(16))) H8 {
  _Float16 v[8];
};

constexpr int FW = 64;         // w1 == w2
constexpr int FT_ROWS = 8;     // target rows per tile
constexpr int T_PITCH = FT_ROWS * FW + 4;  // halves per source pixel in the LDS tile (+4: the 32 lanes of an 8-byte
                                           // accumulator write land in 32 different bank pairs)

__device__ __forceinline__ _Float16 pool4(_Float16 a, _Float16 b, _Float16 c, _Float16 d) {
  // ATen avg_pool2d on half: float accumulate, one rounding
  return (_Float16)(((float)a + (float)b + (float)c + (float)d) / 4.0f);
}

__global__ __launch_bounds__(512, 4) void corr_build_fused_kernel(const _Float16 *__restrict__ A,
                                                               const _Float16 *__restrict__ Bm, FusedLevels L,
                                                               int C, int h) {
  extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
  // 66 KB per workgroup, two workgroups per CU: the pooled levels reuse the level-0 tile once it is dead
  _Float16 *T = smem;                         // [64][T_PITCH]           level 0, rounded
  _Float16 *P1 = T;                           // [64][4][32]             level 1, rounded, unsheared (after T)
  _Float16 *P2 = P1 + 64 * 4 * 32;            // [64][2][16]
  _Float16 *P3 = P2 + 64 * 2 * 16;            // [64][1][8]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int y1 = blockIdx.x;                  // source row
  const int ty0 = blockIdx.y * FT_ROWS;       // first target row of the tile
  const int e = blockIdx.z;
  const int HW = h * FW;
  const _Float16 *Ae = A + ((size_t)e * HW + (size_t)y1 * FW) * C;            // 64 source pixels
  const _Float16 *Be = Bm + ((size_t)e * HW + (size_t)(ty0 + wave) * FW) * C; // this wave's 64 targets
  const int l31 = lane & 31, kh = (lane >> 5) * 8;

  // ---- MFMA: acc[i][j] = 32x32 tile (sources 32 i .. , targets 32 j ..) ---------------------------------
  float16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  for (int k = 0; k < C; k += 16) {
    half8 a[2], b[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
      a[t] = *reinterpret_cast<const half8 *>(Ae + (size_t)(t * 32 + l31) * C + k + kh);
      b[t] = *reinterpret_cast<const half8 *>(Be + (size_t)(t * 32 + l31) * C + k + kh);
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j], a[i], acc[i][j], 0, 0, 0);  // targets x sources
  }
  // D layout: col = lane & 31 (source x1 within the 32-block), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (target tx):
  // four consecutive targets of one source per register quad -> one 8-byte LDS write
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int rq = 0; rq < 4; rq++) {
        const int x1 = i * 32 + l31;
        const int tx = j * 32 + 8 * rq + 4 * (lane >> 5);
        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
        half4 v;
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = (_Float16)acc[i][j][4 * rq + e];
        *reinterpret_cast<half4 *>(T + x1 * T_PITCH + wave * FW + tx) = v;
      }
  __syncthreads();

  const size_t eoff = (size_t)e;  // level l volume of edge e: [h>>l][64>>l][HW]
  const int dl = lane >> 3, sub = lane & 7;  // 8 lanes x 16 B = one 128-byte segment of 64 x1
  // ---- level 0 (this wave's own target row): Vs0[dy][dx][y1][x1] = T[x1][ty][(x1 + dx) & 63] -----------------
  {
    const int ty = ty0 + wave;
    int dy = ty - y1;
    dy += (dy < 0) ? h : 0;
    _Float16 *dst = L.vs[0] + (eoff * h + dy) * (size_t)FW * HW + (size_t)y1 * FW + sub * 8;
#pragma unroll 2
    for (int it = 0; it < 8; it++) {
      const int dx = it * 8 + dl;
      H8 v;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int x1 = sub * 8 + q;
        v.v[q] = T[x1 * T_PITCH + wave * FW + ((x1 + dx) & 63)];
      }
      *reinterpret_cast<H8 *>(dst + (size_t)dx * HW) = v;
    }
  }
  // ---- level 1, unsheared: P1[x1][ty1][tx1], through registers because it takes the tile's place --------------
  {
    _Float16 p1v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int idx = tid + 512 * u;
      const int tx1 = idx & 31, ty1 = (idx >> 5) & 3, x1 = idx >> 7;
      const _Float16 *s = T + x1 * T_PITCH + (2 * ty1) * FW + 2 * tx1;
      p1v[u] = pool4(s[0], s[1], s[FW], s[FW + 1]);
    }
    __syncthreads();  // every read of the level-0 tile is done (its sheared store above included)
#pragma unroll
    for (int u = 0; u < 16; u++) P1[tid + 512 * u] = p1v[u];
  }
  __syncthreads();
  // level 2 and the sheared store of level 1 only read P1
  for (int idx = tid; idx < 64 * 2 * 16; idx += 512) {
    const int tx2 = idx & 15, ty2 = (idx >> 4) & 1, x1 = idx >> 5;
    const _Float16 *s = P1 + (x1 * 4 + 2 * ty2) * 32 + 2 * tx2;
    P2[idx] = pool4(s[0], s[1], s[32], s[33]);
  }
  {
    // Vs1[dy][dx][y1][x1], dy = (ty1g - (y1 >> 1)) mod h/2, dx = (tx1 - (x1 >> 1)) mod 32: 4 x 32 segments
    const int h1l = h >> 1;
    for (int seg = wave * 8 + dl; seg < 4 * 32; seg += 64) {
      const int ty1 = seg >> 5, dx = seg & 31;
      int dy = (ty0 >> 1) + ty1 - (y1 >> 1);
      dy += (dy < 0) ? h1l : 0;
      H8 v;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int x1 = sub * 8 + q;
        v.v[q] = P1[(x1 * 4 + ty1) * 32 + (((x1 >> 1) + dx) & 31)];
      }
      *reinterpret_cast<H8 *>(L.vs[1] + ((eoff * h1l + dy) * 32 + dx) * (size_t)HW + (size_t)y1 * FW + sub * 8) = v;
    }
  }
  __syncthreads();
  if (tid < 64 * 8) {  // level 3 from P2: [64][1][8]
    const int tx3 = tid & 7, x1 = tid >> 3;
    const _Float16 *s = P2 + (x1 * 2) * 16 + 2 * tx3;
    P3[tid] = pool4(s[0], s[1], s[16], s[17]);
  }
  {
    const int h2l = h >> 2;
    for (int seg = wave * 8 + dl; seg < 2 * 16; seg += 64) {
      const int ty2 = seg >> 4, dx = seg & 15;
      int dy = (ty0 >> 2) + ty2 - (y1 >> 2);
      dy += (dy < 0) ? h2l : 0;
      H8 v;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int x1 = sub * 8 + q;
        v.v[q] = P2[(x1 * 2 + ty2) * 16 + (((x1 >> 2) + dx) & 15)];
      }
      *reinterpret_cast<H8 *>(L.vs[2] + ((eoff * h2l + dy) * 16 + dx) * (size_t)HW + (size_t)y1 * FW + sub * 8) = v;
    }
  }
  __syncthreads();
  {
    const int h3l = h >> 3;
    for (int seg = wave * 8 + dl; seg < 8; seg += 64) {
      const int dx = seg;
      int dy = (ty0 >> 3) - (y1 >> 3);
      dy += (dy < 0) ? h3l : 0;
      H8 v;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int x1 = sub * 8 + q;
        v.v[q] = P3[x1 * 8 + (((x1 >> 3) + dx) & 7)];
      }
      *reinterpret_cast<H8 *>(L.vs[3] + ((eoff * h3l + dy) * 8 + dx) * (size_t)HW + (size_t)y1 * FW + sub * 8) = v;
    }
  }
}

// defined in corr_build.hip
__global__ void fmap_pixel_major_kernel(const _Float16 *in, _Float16 *out, int C, int HW);

}  // namespace dba

using namespace dba;

extern "C" {

int dba_corr_volume_build_sheared_supported(int C, int h1, int w1, int h2, int w2, int num_levels) {
  return (h1 == h2 && w1 == 64 && w2 == 64 && (h2 % 8) == 0 && (C % 16) == 0 && num_levels == 4) ? 1 : 0;
}

int dba_corr_volume_build_sheared(const void *fmap1, const void *fmap2, void *const *sheared_levels, int n, int C,
                                  int h1, int w1, int h2, int w2, int num_levels, void *scratch,
                                  size_t scratch_bytes, dba_stream_t stream) {
  if (!dba_corr_volume_build_sheared_supported(C, h1, w1, h2, w2, num_levels)) return DBA_ERR_UNSUPPORTED;
  if (n < 0) return DBA_ERR_ARG;
  if (n == 0) return DBA_OK;
  if (!fmap1 || !fmap2 || !sheared_levels || !scratch) return DBA_ERR_ARG;
  if (scratch_bytes < dba_corr_volume_scratch_bytes(n, C, h1, w1, h2, w2)) return DBA_ERR_WORKSPACE;
  const int HW = h1 * w1;
  hipStream_t s = (hipStream_t)stream;
  _Float16 *A = static_cast<_Float16 *>(scratch);
  _Float16 *Bm = reinterpret_cast<_Float16 *>(static_cast<char *>(scratch) + align_up((size_t)n * C * HW * 2, 256));
  hipLaunchKernelGGL(fmap_pixel_major_kernel, dim3((HW + 63) / 64, (C + 63) / 64, n), dim3(256), 0, s,
                     static_cast<const _Float16 *>(fmap1), A, C, HW);
  hipLaunchKernelGGL(fmap_pixel_major_kernel, dim3((HW + 63) / 64, (C + 63) / 64, n), dim3(256), 0, s,
                     static_cast<const _Float16 *>(fmap2), Bm, C, HW);
  FusedLevels L;
  for (int l = 0; l < 4; l++) L.vs[l] = static_cast<_Float16 *>(sheared_levels[l]);
  const size_t lds = sizeof(_Float16) * ((size_t)64 * T_PITCH);  // the pooled levels live inside the dead tile
  static DeviceOnce attr_once;
  if (attr_once.needed()) {
    DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&corr_build_fused_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_once.done();
  }
  hipLaunchKernelGGL(corr_build_fused_kernel, dim3(h1, h2 / FT_ROWS, n), dim3(512), lds, s, A, Bm, L, C, h1);
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

}  // extern "C"
