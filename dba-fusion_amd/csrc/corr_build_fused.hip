// Fused all-pairs correlation + 4-level pyramid + flow-aligned ("sheared") store, gfx950 MFMA, any map size.
//
// One pass replaces CorrBlock.corr (torch.matmul), the three avg_pool2d passes of CorrBlock.__init__
// (/root/reference/dbaf/modules/corr.py:24-38, :63-71) and the re-layout into the sheared volume
// (corr_sheared.hip): the reference materialises level 0 (33.5 MB/edge at 64x64), reads it back three times for the
// pooling, and this repo's unfused path then reads and rewrites every level once more for the shear.
// Here every output byte is written exactly once (44.6 MB/edge) and the inputs (2 x 1 MB/edge) stay in L2.
//
// Workgroup = 8 waves; tile = one STRIP of 64 consecutive source pixels of the flattened (y1, x1) index (the unit the
// lookup reads, see corr_sheared.hip) x 8 target rows ty0..ty0+7 x all w2 target columns (w2 <= 128):
//   * wave w owns target row ty0 + w: 64 sources x (32 NT) targets = 2 x NT v_mfma_f32_32x32x16_f16 tiles computed as
//     targets x sources, so that a register quad is four consecutive targets of one source (one 8-byte LDS write);
//     16-byte fragment loads straight from the k-block-major feature maps [C/16][HW][16] (corr_build.hip; L2-resident,
//     8 full lines per load instruction), the next k-step's fragments are requested before the current one is
//     multiplied;
//   * accumulators -> f16 (the single rounding of the reference's half GEMM) -> LDS tile T[source][ty][tx];
//   * stores: every element goes from the tile to Vs_l[(ty_l - (y1 >> l)) mod h2l][dx][pixel] with PER-PIXEL offsets, so
//     any map width works (a strip may span a row end).  Levels 0 and 1 (94 % of the bytes): a lane owns four
//     consecutive pixels of one of four (dy, dx) lines and stores 8 bytes (a quad that spans a row end is stored by the
//     whole wave, 2 bytes per lane, sixteen offsets per instruction); levels 2, 3: a lane is one pixel.  The barriers synchronise LDS only (`s_waitcnt lgkmcnt(0); s_barrier`): a
//     `__syncthreads()` would drain every outstanding store first;
//   * levels 1..3: 2x2 averages of the ROUNDED level below (== F.avg_pool2d on half, floor sizes): each thread pools one
//     8 x 8 block of the tile down to its 4 x 4 + 2 x 2 + 1 values in registers, which then take the dead tile's place in
//     LDS (two barriers) for the sheared store loops;
//   * the strip's source operand (16 KB at C = 128) is staged ONCE per workgroup in the LDS the tile will take, and the
//     target fragments are requested two k-steps ahead.
// Where the time goes, one-strip form (in-kernel timestamps, -DFB_PROF, 64x64, 32 edges; a workgroup lives 47 k cycles,
// two per CU): source operand 9 %, MFMA phase 43 % (the 32 MFMAs per wave need a third of it; the rest is the target
// fragments, 16 KB per wave out of L2 at ~20 B/clk/CU), tile write 6 %, level-0 stores 25 %, pooled levels 13 %, drain
// 3 %: 517 us per 32 edges = 16.2 us per edge, 0.36 of the HBM peak on the kernel alone (round 1: 23.3 us; before the
// store loops below were cut from ~70 to ~10 VALU instructions per 8-byte store: 20.1 us, 1935 VALU instructions per
// wave, VALU busy 64 %).  The strip-walking form (LOOP, see the kernel's comment) takes 413-439 us = 12.9-13.7 us per
// edge, 0.43-0.45: ~21 k cycles per strip and wave, of which level-0 stores + pooling 7.1 k, pooled stores 4.5 k, 3.3-4.4 k
// at the top of a strip (the next operand's loads wait behind the previous strip's stores), products 2.4 k, tile write 1.9 k.
// What was measured and did not help (scratch/cu_rates.hip, scratch/lds_rates.hip, scratch/corr_build_pipe_experiment.hip):
//   * one CU alone stores 28 B/clk, the whole chip 5.4-5.6 TB/s = 9 B/clk/CU: in the store phases, which all
//     workgroups enter together, the chip is at HBM's write ceiling, in the MFMA phases HBM idles.  A persistent,
//     wave-specialised form (8 compute + 8 store waves, tile double-buffered in LDS, stores of tile i - 1 issued while
//     tile i is multiplied) was built and is bit-exact, but runs at 20.5-21.5 us/edge: its waves wait 63 % of their
//     cycles at the three barriers that couple the roles (VALU busy 28 %, LDS busy 17 %);
//   * target map in blocks of 8 channels (a half wave's fragment load = 512 contiguous bytes): no change;
//   * staggered start of the two co-resident workgroups, 4-deep fragment prefetch: no change.
// Halving the fragment traffic needs a 128-pixel tile (132 KB of LDS, one workgroup per CU).
// Operands (round 5): where 16-byte pieces of the maps are aligned (map width a multiple of 8) the strip-walking form reads
// the caller's [n][C][h][w] maps as they lie (NATIVE below) -- the two fmap_pixel_major_kernel launches (20 us each per 32
// edges at 64x64) and their scratch traffic are gone; the kernel itself pays ~1 % for the 32-byte pieces a tiled strip's
// channel slice consists of (16.9 against 17.6 us per edge end to end; DBA_BUILD_OPERANDS=copy|bnative for the A/B run).
// Shapes: w2 <= 128, C % 16 == 0, 4 levels, h2 >> 3 >= 1, w2 >> 3 >= 1; anything else takes the unfused path of
// corr_build.hip + corr_shear_kernel.
// Kernels in this file (round 6): corr_build_fused_kernel<NT, LOOP, NATIVE, NATIVE_B> -- the eight-wave forms described above (maps up
// to 32 wide, tiled planes that miss the sixteen-wave form's conditions, widths that are multiples of 8 on linear planes, 128-wide
// tiled planes); corr_build_fused16_kernel -- the strip walk on sixteen waves, tiled planes, 64 columns (the headline shapes);
// corr_build_fused16g_kernel<W2C> -- the same walk on linear planes, 33..63 columns whose 16-byte pieces are not aligned (55 x 55);
// corr_build_fused16w_kernel -- one strip per workgroup on sixteen waves, linear planes, 65..128 columns (28 x 107).  The host
// function at the end of the file picks one; DBA_BUILD_WAVES=8 / DBA_BUILD_KERNEL / DBA_BUILD_OPERANDS force the older forms.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <cstdlib>

#include "common.h"

namespace dba {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float16v __attribute__((ext_vector_type(16)));

struct FusedLevels {
  _Float16 *vs[4];
};

constexpr int FT_ROWS = 8;  // target rows per tile
#ifndef FB_STORE_AUX
#define FB_STORE_AUX 0   // cache policy bits of the level stores (gfx950: 1 = sc0, 2 = nt, 16 = sc1); measured, see profiles/
#endif

__device__ __forceinline__ _Float16 pool4(_Float16 a, _Float16 b, _Float16 c, _Float16 d) {
  // ATen avg_pool2d on half: float accumulate, one rounding
  return (_Float16)(((float)a + (float)b + (float)c + (float)d) / 4.0f);
}

// workgroup barrier that orders LDS traffic only (global stores stay in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LOOP (64-wide tiles, C = 128): the workgroup walks `strips_per_wg` consecutive strips of its target-row tile.  The
// wave's target fragments (16 KB: every k-step of its row) then live in 64 registers for the whole walk - no fragment
// load sits between the matrix products any more -, and the next strip's source operand is requested at the top of a
// strip and put into LDS (double-buffered) right after the strip's tile write, BEFORE the strip's stores are issued: a
// wave's loads and stores complete in order, so consumed after those stores the operand would wait for all of them to
// reach HBM; consumed before them it only waits for the previous strip's, which have had a whole multiplication to drain.
// The stores then drain while the next strip is multiplied.  One workgroup per CU (103 KB of LDS, up to 256 registers
// per lane).
// NATIVE / NATIVE_B (LOOP form only): A / Bm are the caller's feature maps as they lie, [n][C][h][w] -- no k-block-major copies
// are made first (a launch and 2 x the map's bytes of traffic less per map and build).  The source operand's 16 KB per strip
// are read as 16-byte pieces (8 consecutive pixels of one channel, which both pixel orders of the planes keep adjacent), 4
// channels per thread (threads 0..255), scaled (/ 4: corr.py:67-68) and transposed in registers into the LDS fragment layout's
// [pixel][4 channels] units; the wave's target fragments -- read once per walk -- go through 2 KB of LDS per wave
// ([pixel][16 channels], written by halves, read back as fragments).
template <int NT, bool LOOP, bool NATIVE = false, bool NATIVE_B = NATIVE>
__global__ __launch_bounds__(512, LOOP ? 2 : ((NT <= 2) ? 4 : 2)) void corr_build_fused_kernel(
    const _Float16 *__restrict__ A, const _Float16 *__restrict__ Bm, FusedLevels L, int C, int h1, int w1, int h2, int w2,
    int HW1p, float inv_w1, int strips_per_wg, const int *__restrict__ oslots, int tiled
#ifdef FB_PROF
    , unsigned long long *prof
#endif
) {
#ifdef FB_PROF
  // (LOOP: slot i accumulates the time since the previous stamp over the workgroup's strips; slot 7 counts the strips)
  unsigned long long fb_last_ = __builtin_amdgcn_s_memtime();
#define FB_STAMP(i) do { if ((threadIdx.x & 63) == 0) { unsigned long long *ps_ = prof + ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (threadIdx.x >> 6)) * 8; const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (LOOP) { ps_[i] += t_ - fb_last_; fb_last_ = t_; } else ps_[i] = t_; } } while (0)
#else
#define FB_STAMP(i) (void)0
#endif
  FB_STAMP(0);
  constexpr int W2P = 32 * NT;               // tile columns (targets beyond w2 are computed and never read)
  constexpr int RP = W2P + 4;                // tile row: the w2 columns, then columns 0..3 once more (level-0 store loop)
  constexpr int PITCH = FT_ROWS * RP + 4;    // halves per source pixel
  extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
  _Float16 *T = smem;                              // [64][PITCH]  level 0, rounded; before that: the strip's A operand
  _Float16 *Ab0 = LOOP ? smem + 64 * PITCH : smem;  // the strip's source operand (LOOP: two 16 KB buffers behind the tile)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // Workgroup -> (strip group, target-row tile, edge).  Tiled planes: the row tiles ty0 and ty0 + 8 of one source tile write
  // complementary pieces of the same lines (see the level-0 stores), so all row tiles of a (strip group, edge) go to ONE
  // XCD -- consecutive dispatch ids round-robin over the 8 XCDs, each XCD takes a contiguous eighth of the logical ids,
  // row tile fastest -- and run side by side: the pieces then meet in that XCD's L2 and leave it as whole lines.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
#ifndef FB_NO_XCD_MAP
  if (tiled) {
    const int total = (int)(gridDim.x * gridDim.y * gridDim.z);
    const int lin = bx + (int)gridDim.x * (by + (int)gridDim.y * bz);
    const int q8 = total >> 3, r8 = total & 7, xk = lin & 7;
    const int logical = xk * q8 + min(xk, r8) + (lin >> 3);
    by = logical % (int)gridDim.y;
    const int rest = logical / (int)gridDim.y;
    bx = rest % (int)gridDim.x;
    bz = rest / (int)gridDim.x;
  }
#endif
  const int ty0 = by * FT_ROWS;                    // first target row of the tile
  const int e = bz;
  const int HW1 = h1 * w1, HW2 = h2 * w2;
  const int l31 = lane & 31, kh = (lane >> 5) * 8;
  const int nstrips = HW1p >> 6;
  const int s_begin = LOOP ? bx * strips_per_wg : bx;
  const int s_end = LOOP ? min(nstrips, s_begin + strips_per_wg) : s_begin + 1;
  constexpr int KSL = 8;                           // k-steps of the LOOP form (C = 128)
  half8 bres[LOOP ? KSL : 1][NT];                  // LOOP: the wave's target fragments, resident
  half8 apre[2];                                   // LOOP: this thread's two 16-byte pieces of the next source operand
  half8 anat[4];                                   // NATIVE: 8 pixels x 4 channels of the next source operand (threads 0..255)
  auto request_a = [&](int strip) {
    const _Float16 *Ae = A + (size_t)e * HW1 * C;
    if constexpr (NATIVE) {   // thread = (channel group of 4, pixel group of 8: half a tile row, or 8 pixels of the linear order)
      if (tid < 256) {
        const int kg = tid >> 3, pg = tid & 7;
        int yy, xx;
        sh_pixel_yx(min(strip * 64 + 8 * pg, HW1 - 8), w1, inv_w1, tiled != 0, yy, xx);
        const _Float16 *src = Ae + (size_t)(4 * kg) * HW1 + yy * w1 + xx;
#pragma unroll
        for (int c = 0; c < 4; c++) anat[c] = *reinterpret_cast<const half8 *>(src + (size_t)c * HW1);
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int idx = tid + 512 * u, kbk = idx >> 7, r = idx & 127, px = r >> 1, hf = r & 1;
      apre[u] = *reinterpret_cast<const half8 *>(Ae + ((size_t)kbk * HW1 + min(strip * 64 + px, HW1 - 1)) * 16 + hf * 8);
    }
  };
  auto stage_a = [&](_Float16 *dst) {  // this thread's pieces -> LDS, fragment layout (see the non-LOOP staging below)
    if constexpr (NATIVE) {
      if (tid < 256) {
        const int kg = tid >> 3, pg = tid & 7, kbk = kg >> 2, qq = kg & 1, hf = (kg >> 1) & 1, px = 8 * pg;
        _Float16 *d = dst + (((((kbk * 2 + (px >> 5)) * 2 + qq) * 2 + hf) * 32 + (px & 31)) * 4);
#pragma unroll
        for (int j = 0; j < 8; j++) {   // pixel j: its four channels
          half4 o;
#pragma unroll
          for (int c = 0; c < 4; c++) o[c] = anat[c][j];
          *reinterpret_cast<half4 *>(d + 4 * j) = o * (_Float16)0.25f;   // (corr.py:67-68: both maps / 4, one rounding to half)
        }
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int idx = tid + 512 * u, kbk = idx >> 7, r = idx & 127, px = r >> 1, hf = r & 1;
      const int fa = ((((kbk * 2 + (px >> 5)) * 2 + 0) * 2 + hf) * 32 + (px & 31)) * 4;
      half4 lo, hi;
#pragma unroll
      for (int c = 0; c < 4; c++) lo[c] = apre[u][c], hi[c] = apre[u][4 + c];
      *reinterpret_cast<half4 *>(dst + fa) = lo;
      *reinterpret_cast<half4 *>(dst + fa + 2 * 32 * 4) = hi;
    }
  };
  if constexpr (LOOP) {
    const int ty = min(ty0 + wave, h2 - 1);
    if constexpr (NATIVE_B) {
      // k-step ks: 16 channels x 64 targets of row ty = 128 pieces of 8 targets x 1 channel, two per lane
      const _Float16 *Be = Bm + (size_t)e * HW2 * C + (size_t)ty * w2;
      const int kq = lane >> 3, tx0 = min(8 * (lane & 7), w2 - 8);
      half8 raw[KSL][2];
#pragma unroll
      for (int ks = 0; ks < KSL; ks++)
#pragma unroll
        for (int u = 0; u < 2; u++)
          raw[ks][u] = *reinterpret_cast<const half8 *>(Be + (size_t)(16 * ks + kq + 8 * u) * HW2 + tx0);
      _Float16 *scr = T + wave * (64 * 16);   // (the tile's LDS: nothing lives there before the first strip)
#pragma unroll
      for (int ks = 0; ks < KSL; ks++) {
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
          for (int j = 0; j < 8; j++) scr[(8 * (lane & 7) + j) * 16 + kq + 8 * u] = raw[ks][u][j] * (_Float16)0.25f;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < NT; t++) bres[ks][t] = *reinterpret_cast<const half8 *>(scr + (t * 32 + l31) * 16 + kh);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    } else {
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const _Float16 *bpt = Bm + (size_t)e * HW2 * C + (size_t)min(ty * w2 + t * 32 + l31, HW2 - 1) * 16 + kh;
#pragma unroll
      for (int ks = 0; ks < KSL; ks++) bres[ks][t] = *reinterpret_cast<const half8 *>(bpt + (size_t)ks * 16 * HW2);
    }
    }
    // (pinned: left to itself the compiler sinks these loads into the strip loop and re-reads the fragments per strip)
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
      for (int ks = 0; ks < KSL; ks++) asm volatile("" : "+v"(bres[ks][t]));
    request_a(s_begin);
    stage_a(Ab0);
  }
  for (int strip = s_begin; strip < s_end; strip++) {
  const int p0 = strip * 64;                       // first source pixel of the strip
  _Float16 *Ab = LOOP ? Ab0 + ((strip - s_begin) & 1) * (64 * 128) : Ab0;
  // LOOP: the next strip's operand is requested now and consumed after this strip's tile write, i.e. BEFORE this strip's
  // stores are issued: the wait for it (loads and stores of a wave complete in order) then only covers the previous
  // strip's stores, which have had this strip's whole multiplication to drain
  if constexpr (LOOP) {
    if (strip + 1 < s_end) request_a(strip + 1);
  }
#ifdef FB_PROF
  if (LOOP && (threadIdx.x & 63) == 0) prof[((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (threadIdx.x >> 6)) * 8 + 7] += 1;
#endif

  // ---- the strip's source operand, shared by the 8 waves: [C/16][64 pixels][16] halves into the (still unused) tile ----
  // (k-block-major like the global map, so a k-step's fragment is 32 lanes x 32 B contiguous: conflict-free b128 reads)
#ifndef FB_ABLATE_MFMA
  if constexpr (!LOOP) {
    const _Float16 *Ae = A + (size_t)e * HW1 * C;
    const int kblocks = C >> 4;
    for (int idx = tid; idx < kblocks * 128; idx += 512) {  // 16-byte pieces: [kblock][pixel][half of the 16 channels]
      const int kbk = idx >> 7, r = idx & 127, px = r >> 1, hf = r & 1;
      const half8 v = *reinterpret_cast<const half8 *>(Ae + ((size_t)kbk * HW1 + min(p0 + px, HW1 - 1)) * 16 + hf * 8);
      // LDS layout [k-step][32-pixel block][q][k-half][pixel][4 halves] (q inside the block since round 6: a lane's two halves
      // pair up into one two-address read whose registers are the fragment): a wave's fragment read is TWO ds_read_b64 of
      // 512 consecutive bytes each (3.4 cycles of the LDS pipe per instruction; the one ds_read_b128 at a 32-byte pitch
      // this replaces takes 32: scratch/lds_rates.hip)
      const int fa = ((((kbk * 2 + (px >> 5)) * 2 + 0) * 2 + hf) * 32 + (px & 31)) * 4;
      half4 lo, hi;
#pragma unroll
      for (int c = 0; c < 4; c++) lo[c] = v[c], hi[c] = v[4 + c];
      *reinterpret_cast<half4 *>(T + fa) = lo;
      *reinterpret_cast<half4 *>(T + fa + 2 * 32 * 4) = hi;
    }
  }
#endif
  if constexpr (LOOP) lds_barrier();  // (a __syncthreads would wait for the previous strip's stores)
  else __syncthreads();
  FB_STAMP(1);

  // ---- MFMA: acc[i][j] = 32x32 tile (sources 32 i .. , targets 32 j ..) of target row ty0 + wave ------------------
  float16v acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  auto read_a = [&](int ks, half8 (&a)[2]) {
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const _Float16 *fp = Ab + ((((ks * 2 + t) * 2 + 0) * 2 + (lane >> 5)) * 32 + l31) * 4;
      const half4 lo = *reinterpret_cast<const half4 *>(fp), hi = *reinterpret_cast<const half4 *>(fp + 2 * 32 * 4);
#pragma unroll
      for (int c = 0; c < 4; c++) a[t][c] = lo[c], a[t][4 + c] = hi[c];
    }
  };
  if constexpr (LOOP) {
#pragma unroll
    for (int ks = 0; ks < KSL; ks++) {
      half8 a[2];
      read_a(ks, a);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bres[ks][j], a[i], acc[i][j], 0, 0, 0);  // targets x sources
    }
  } else {
    const int ty = min(ty0 + wave, h2 - 1);  // rows past the map are computed on a valid row and never stored
    const _Float16 *bp[NT];  // k-block-major map: element (k, pixel) at ((k >> 4) * HW + pixel) * 16 + (k & 15)
#pragma unroll
    for (int t = 0; t < NT; t++) bp[t] = Bm + (size_t)e * HW2 * C + (size_t)min(ty * w2 + t * 32 + l31, HW2 - 1) * 16 + kh;
    const size_t kb = (size_t)HW2;  // elements between consecutive k-blocks / 16
    // target fragments two k-steps ahead of their use (L2 latency), source fragments from LDS right before it
    half8 b0[NT], b1[NT], b2[NT];
    const int ksteps = C >> 4;
#pragma unroll
    for (int t = 0; t < NT; t++) {
      b0[t] = *reinterpret_cast<const half8 *>(bp[t]);
      b1[t] = *reinterpret_cast<const half8 *>(bp[t] + (size_t)min(1, ksteps - 1) * 16 * kb);
    }
#ifdef FB_ABLATE_MFMA  // scratch builds only
    for (int ks = 0; ks < 0; ks++) {
#else
    for (int ks = 0; ks < ksteps; ks++) {
#endif
      const int kn = min(ks + 2, ksteps - 1);  // (the last steps re-request a fragment: no branch in the loop)
#pragma unroll
      for (int t = 0; t < NT; t++) b2[t] = *reinterpret_cast<const half8 *>(bp[t] + (size_t)kn * 16 * kb);
      half8 a[2];
      read_a(ks, a);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0[j], a[i], acc[i][j], 0, 0, 0);  // targets x sources
#pragma unroll
      for (int t = 0; t < NT; t++) {
        b0[t] = b1[t];
        b1[t] = b2[t];
      }
    }
  }
  FB_STAMP(2);
  lds_barrier();  // every wave is done with the source operand: the tile takes its place
  // D layout: col = lane & 31 (source within the 32-block), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (target tx):
  // four consecutive targets of one source per register quad -> one 8-byte LDS write
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int rq = 0; rq < 4; rq++) {
        const int src = i * 32 + l31;
        const int tx = j * 32 + 8 * rq + 4 * (lane >> 5);
        half4 v;
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = (_Float16)acc[i][j][4 * rq + q];
        *reinterpret_cast<half4 *>(T + src * PITCH + wave * RP + tx) = v;
      }
  // columns 0..3 once more behind column w2 - 1 (after the row's own writes: for w2 < W2P those put unused targets
  // there): a store lane's four diagonal reads then never wrap inside a quad
  if (lane < 32) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      _Float16 *wr = T + (i * 32 + l31) * PITCH + wave * RP + w2;
#pragma unroll
      for (int q = 0; q < 4; q++) wr[q] = (_Float16)acc[i][0][q];
    }
  }
  lds_barrier();  // the last barrier: everything below reads the tile only
  FB_STAMP(3);
  if constexpr (LOOP) {
    if (strip + 1 < s_end) stage_a(Ab0 + ((strip + 1 - s_begin) & 1) * (64 * 128));  // ahead of this strip's stores
    FB_STAMP(6);
  }

  // ---- this lane's source pixel (levels 2, 3: a lane is one pixel) and its QUAD of pixels (levels 0, 1) ------------------
  const int p = p0 + lane;
  const bool active = p < HW1;
  int x1, y1;
  auto pixel_xy = [&](int pix, int &x, int &y) {   // plane index -> source pixel (common.h: linear or 4 x 16 tiles)
    sh_pixel_yx(min(pix, HW1 - 1), w1, inv_w1, tiled != 0, y, x);
  };
  pixel_xy(p, x1, y1);
  constexpr unsigned OOR = 0x80000000u;
  // level l volume of edge e: [h2l][w2l][HW1p] halves, addressed through a buffer resource (an edge-level is < 2 GB);
  // oslots: edge e is written into slot oslots[e] of the level stores (the slot-addressed CorrBlock builds new edges
  // straight into the free slots of its pyramid: nothing is concatenated afterwards)
  const int eo = oslots ? oslots[e] : e;
  auto level_rsrc = [&](int lvl) {
    const size_t elems = (size_t)(h2 >> lvl) * (w2 >> lvl) * HW1p;
    return __builtin_amdgcn_make_buffer_rsrc((void *)(L.vs[lvl] + (size_t)eo * elems), 0, (int)(2 * elems), 0x00020000);
  };
  const unsigned plane_bytes = 2u * (unsigned)HW1p;

  // Levels 0 and 1 carry 94 % of the bytes.  A 2-byte-per-lane store instruction costs the vector memory pipe as much as
  // an 8-byte one (~13 cycles per wave instruction: 10 B/clk/CU, which capped the store phases), so there a lane owns
  // FOUR consecutive pixels (q = lane & 15) of one of four lines (g = lane >> 4) and stores 8 bytes: one instruction is
  // four full 128-byte lines.  A quad whose pixels do not share a source row (a strip that spans a row end, or the last
  // pixels of the map) is not stored by its lanes but by the whole wave afterwards (`irregular`).
  const int q4 = (lane & 15) * 4, g = lane >> 4;
  int qx[4], qy[4];
#pragma unroll
  for (int i = 0; i < 4; i++) pixel_xy(p0 + q4 + i, qx[i], qy[i]);
  const bool quad_regular = (p0 + q4 + 3 < HW1) && (qy[0] == qy[3]);
  // Quads that are not regular are rare (one per row end inside the strip, one at the end of the map) but their four
  // pixels sit on four different lines: the WHOLE wave takes them, lane = (pixel lane & 3 of the quad, offset lane >> 2),
  // sixteen offsets per 2-byte store instruction -- w2 / 16 instructions per such quad.  (Left to the quad's own four
  // lanes, as until round 3, they cost 4 x w2 / 4 instructions issued by a wave with 4 of 64 lanes alive: strips with a
  // row end took 2.5-3x as long, which is most of what 28x107 and 55x55 lost against 64x64.)
  const unsigned long long irregular = __ballot(!quad_regular && g == 0 && (p0 + q4 < HW1));  // bit = quad index (wave-uniform)
  const int ipx = lane & 3, idx16 = lane >> 2;
  // ---- level 0: Vs0[(ty - y1) mod h2][dx][pixel] = T[pixel][ty][(x1 + dx) mod w2], this wave's own target row.  A lane's
  // quad of pixels reads the tile along a diagonal (pixel + 1, column + 1); with columns 0..3 replicated behind the row
  // only the first column wraps, once per line, and the store offset just advances by four planes: ~10 VALU
  // instructions per 8-byte store where the general form needs ~70 ----
  // Tiled pixel order (common.h): the strip is a 4 x 16 tile of the map, its quads sit on four source rows, and a line of
  // the planes holds all four at ONE dy = ty - y1.  So that a store instruction still writes whole lines, the quads of tile
  // row rr take the target row (wave + rr) mod 8 of the workgroup's tile instead of the wave's own: dy is then the same for
  // every lane of waves 0..4 (four full lines per instruction), and two values, 8 apart, in waves 5..7.  (With the wave's
  // own row for every quad each instruction wrote sixteen 32-byte pieces: 16.7 against 14.7 us per edge.)
  const int rr = tiled ? ((lane & 15) >> 2) : 0;
  {
    const int wrow = (wave + rr) & (FT_ROWS - 1);
    const int ty = ty0 + wave, tyq = ty0 + wrow;
#ifdef FB_ABLATE_L0STORE
    if (ty < 0) {
#else
    if (tiled || ty < h2) {  // (wave-uniform)
#endif
      const __amdgpu_buffer_rsrc_t r0 = level_rsrc(0);
      if (quad_regular && tyq < h2) {
        int t = qx[0] + g;
        t -= (t >= w2) ? w2 : 0;
        int dy = tyq - qy[0];
        dy += (dy < 0) ? h2 : 0;
        unsigned voff = ((unsigned)dy * (unsigned)w2 + (unsigned)g) * plane_bytes + 2u * (unsigned)(p0 + q4);
        const _Float16 *lb = T + q4 * PITCH + wrow * RP;
        for (int dx0 = 0; dx0 < w2; dx0 += 16) {  // four lines per batch: their 16 LDS reads are in flight together
          unsigned short a[4][4];
#pragma unroll
          for (int b = 0; b < 4; b++) {
            const _Float16 *pp = lb + t;
#pragma unroll
            for (int u = 0; u < 4; u++) a[b][u] = __builtin_bit_cast(unsigned short, pp[u * (PITCH + 1)]);
            t += 4;
            t -= (t >= w2) ? w2 : 0;
          }
#pragma unroll
          for (int b = 0; b < 4; b++) {
            typedef unsigned u2v __attribute__((ext_vector_type(2)));
            u2v d;
            d.x = (unsigned)a[b][0] | ((unsigned)a[b][1] << 16);
            d.y = (unsigned)a[b][2] | ((unsigned)a[b][3] << 16);
            __builtin_amdgcn_raw_buffer_store_b64(d, r0, (dx0 + 4 * b + g < w2) ? voff : OOR, 0, FB_STORE_AUX);
            voff += 4u * plane_bytes;
          }
        }
      }
      for (unsigned long long m = (ty < h2) ? irregular : 0ull; m; m &= m - 1) {
        const int pl = 4 * (int)__builtin_ctzll(m) + ipx;  // this lane's pixel of the quad, within the strip
        int xi, yi;
        pixel_xy(p0 + pl, xi, yi);
        const bool pok = p0 + pl < HW1;
        int dy = ty - yi;
        dy += (dy < 0) ? h2 : 0;
        const _Float16 *row = T + pl * PITCH + wave * RP;
        const unsigned vbase = (unsigned)dy * (unsigned)w2 * plane_bytes + 2u * (unsigned)(p0 + pl);
        for (int dx0 = 0; dx0 < w2; dx0 += 16) {
          const int dx = dx0 + idx16;
          int tx = xi + dx;
          tx -= (tx >= w2) ? w2 : 0;
          tx = min(tx, w2 - 1);  // (dx beyond the map in the last group: read something valid, store nothing)
          const _Float16 v = row[tx];
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, v), r0,
                                                (pok && dx < w2) ? vbase + (unsigned)dx * plane_bytes : OOR, 0, FB_STORE_AUX);
        }
      }
    }
  }
  // ---- levels 1..3.  Each thread pools ONE 8 x 8 block of one source pixel's tile into its 4 x 4 level-1, 2 x 2 level-2
  // and 1 level-3 values (from the ROUNDED level below each time), all in registers; after one barrier (every read of
  // the tile, the level-0 stores included, is done) the values take the tile's place in LDS, after a second one the
  // three sheared store loops read them.  (Pooling every element in the lane that stores it needs no barrier at all
  // but 2.4x the arithmetic -- the kernel is bound by VALU issue, 1935 instructions per wave before, see the header.)
  constexpr int RP1 = W2P / 2 + 4;                 // level-1 row: w2 >> 1 columns, then columns 0..3 once more
  _Float16 *P1 = T;                                // [64][4][RP1]
  _Float16 *P2 = P1 + 64 * 4 * RP1;                // [64][2][W2P / 4]
  _Float16 *P3 = P2 + 64 * 2 * (W2P / 4);          // [64][W2P / 8]
  constexpr int NBLK = W2P / 8;                    // 8-column blocks per pixel
  constexpr int PER = 64 * NBLK / 512;             // blocks per thread (1 for 64-wide tiles, 2 for 128-wide)
  _Float16 q1[PER][4][4], q2[PER][2][2], q3[PER];
#pragma unroll
  for (int u = 0; u < PER; u++) {
    const int blk = tid + 512 * u, src = blk / NBLK, cb = blk - src * NBLK;
    const _Float16 *tb = T + src * PITCH + 8 * cb;
    _Float16 t8[8][8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const half4 lo = *reinterpret_cast<const half4 *>(tb + r * RP), hi = *reinterpret_cast<const half4 *>(tb + r * RP + 4);
#pragma unroll
      for (int c = 0; c < 4; c++) t8[r][c] = lo[c], t8[r][4 + c] = hi[c];
    }
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int c = 0; c < 4; c++) q1[u][r][c] = pool4(t8[2 * r][2 * c], t8[2 * r][2 * c + 1], t8[2 * r + 1][2 * c], t8[2 * r + 1][2 * c + 1]);
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int c = 0; c < 2; c++)
        q2[u][r][c] = pool4(q1[u][2 * r][2 * c], q1[u][2 * r][2 * c + 1], q1[u][2 * r + 1][2 * c], q1[u][2 * r + 1][2 * c + 1]);
    q3[u] = pool4(q2[u][0][0], q2[u][0][1], q2[u][1][0], q2[u][1][1]);
  }
  lds_barrier();  // every read of the level-0 tile is done (its sheared store above included)
#pragma unroll
  for (int u = 0; u < PER; u++) {
    const int blk = tid + 512 * u, src = blk / NBLK, cb = blk - src * NBLK;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      half4 v;
#pragma unroll
      for (int c = 0; c < 4; c++) v[c] = q1[u][r][c];
      _Float16 *row1 = P1 + (src * 4 + r) * RP1;
      if (4 * cb + 4 <= (w2 >> 1)) {
        *reinterpret_cast<half4 *>(row1 + 4 * cb) = v;
      } else {  // the block that holds column (w2 >> 1) - 1: the columns behind it belong to the copy of block 0
#pragma unroll
        for (int c = 0; c < 4; c++)
          if (4 * cb + c < (w2 >> 1)) row1[4 * cb + c] = q1[u][r][c];
      }
      if (cb == 0) {
#pragma unroll
        for (int c = 0; c < 4; c++) row1[(w2 >> 1) + c] = q1[u][r][c];
      }
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
      half2v v;
      v.x = q2[u][r][0], v.y = q2[u][r][1];
      *reinterpret_cast<half2v *>(P2 + (src * 2 + r) * (W2P / 4) + 2 * cb) = v;
    }
    P3[src * (W2P / 8) + cb] = q3[u];
  }
  lds_barrier();
  FB_STAMP(4);
#ifndef FB_ABLATE_POOLSTORE
  {  // level 1: 4 rows x (w2 >> 1) offsets, four lines per store instruction; wave w takes row w & 3 and every other
     // group of four offsets.  The quad's columns (x >> 1) - (x0 >> 1) are 0, 0|1, 1, 1|2: lane constants
    const int w2l = w2 >> 1, h2l = h2 >> 1;
    const __amdgpu_buffer_rsrc_t rl = level_rsrc(1);
    const int tyl = wave & 3, tyg = (ty0 >> 1) + tyl;
    // (tiled: the quads of tile rows 2, 3 -- level-1 source row + 1 -- take the next pooled row, see level 0)
    const int tylq = (tyl + (rr >> 1)) & 3, tygq = (ty0 >> 1) + tylq;
    if (tiled || tyg < h2l) {  // floor sizes of avg_pool2d: the last partial row of the level below is dropped
      if (quad_regular && tygq < h2l) {
        const int xh = qx[0] >> 1;
        const int o1 = (qx[1] >> 1) - xh, o2 = (qx[2] >> 1) - xh, o3 = (qx[3] >> 1) - xh;
        int t = xh + 4 * (wave >> 2) + g;
        t -= (t >= w2l) ? w2l : 0;
        t -= (t >= w2l) ? w2l : 0;  // (maps down to 8 columns: twice)
        int dy = tygq - (qy[0] >> 1);
        dy += (dy < 0) ? h2l : 0;
        unsigned voff = ((unsigned)dy * (unsigned)w2l + (unsigned)(4 * (wave >> 2) + g)) * plane_bytes + 2u * (unsigned)(p0 + q4);
        const _Float16 *lb = P1 + (q4 * 4 + tylq) * RP1;
        for (int dx0 = 4 * (wave >> 2); dx0 < w2l; dx0 += 8) {
          const _Float16 *pp = lb + t;
          const unsigned short a0 = __builtin_bit_cast(unsigned short, pp[0]), a1 = __builtin_bit_cast(unsigned short, pp[4 * RP1 + o1]);
          const unsigned short a2 = __builtin_bit_cast(unsigned short, pp[8 * RP1 + o2]), a3 = __builtin_bit_cast(unsigned short, pp[12 * RP1 + o3]);
          typedef unsigned u2v __attribute__((ext_vector_type(2)));
          u2v d;
          d.x = (unsigned)a0 | ((unsigned)a1 << 16);
          d.y = (unsigned)a2 | ((unsigned)a3 << 16);
          __builtin_amdgcn_raw_buffer_store_b64(d, rl, (dx0 + g < w2l) ? voff : OOR, 0, FB_STORE_AUX);
          voff += 8u * plane_bytes;
          t += 8;
          t -= (t >= w2l) ? w2l : 0;
          t -= (t >= w2l) ? w2l : 0;
        }
      }
      for (unsigned long long m = (tyg < h2l) ? irregular : 0ull; m; m &= m - 1) {  // (the two waves of a row take alternate groups of 16 offsets)
        const int pl = 4 * (int)__builtin_ctzll(m) + ipx;
        int xi, yi;
        pixel_xy(p0 + pl, xi, yi);
        const bool pok = p0 + pl < HW1;
        int dy = tyg - (yi >> 1);
        dy += (dy < 0) ? h2l : 0;
        const _Float16 *row = P1 + (pl * 4 + tyl) * RP1;
        const unsigned vbase = (unsigned)dy * (unsigned)w2l * plane_bytes + 2u * (unsigned)(p0 + pl);
        for (int dx0 = 16 * (wave >> 2); dx0 < w2l; dx0 += 32) {
          const int dx = dx0 + idx16;
          int tx = (xi >> 1) + dx;
          tx -= (tx >= w2l) ? w2l : 0;
          tx = min(tx, w2l - 1);
          const _Float16 v = row[tx];
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, v), rl,
                                                (pok && dx < w2l) ? vbase + (unsigned)dx * plane_bytes : OOR, 0, FB_STORE_AUX);
        }
      }
    }
  }
  // levels 2 and 3: a lane is one source pixel, (ty_l, dx) segments are dealt to the waves
  auto store_level = [&](int lvl, const _Float16 *Pl, int rows, int pitch_cols) {
    const int h2l = h2 >> lvl, w2l = w2 >> lvl;
    const __amdgpu_buffer_rsrc_t rl = level_rsrc(lvl);
    const int x1l = x1 >> lvl, y1l = y1 >> lvl;
    for (int seg = wave; seg < rows * w2l; seg += 8) {  // (wave-uniform)
      const int tyl = seg / w2l, dx = seg - tyl * w2l;
      const int tyg = (ty0 >> lvl) + tyl;
      if (tyg >= h2l) continue;
      int dy = tyg - y1l;
      dy += (dy < 0) ? h2l : 0;
      int tx = x1l + dx;
      tx -= (tx >= w2l) ? w2l : 0;
      const _Float16 v = Pl[(lane * rows + tyl) * pitch_cols + tx];
      const unsigned voff = active ? ((unsigned)dy * (unsigned)w2l + (unsigned)dx) * plane_bytes + 2u * (unsigned)p : OOR;
      __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, v), rl, voff, 0, FB_STORE_AUX);
    }
  };
  store_level(2, P2, 2, W2P / 4);
  store_level(3, P3, 1, W2P / 8);
#endif
  if constexpr (LOOP) FB_STAMP(5);
  }  // strips of this workgroup
  if constexpr (!LOOP) FB_STAMP(5);
#ifdef FB_PROF
  if constexpr (!LOOP) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FB_STAMP(6);
  }
#endif
}


// =====================================================================================================================
// The strip walk on SIXTEEN waves (round 6): tiled planes, 64 target columns, the caller's maps read as they lie.
//
// The eight-wave walk above is a chain of phases whose costs add up (profiles/r06_late_experiments.txt, item 5: products 2.9,
// level-0 store loop 5.0, pooled stores 4.8, pooling 2.1, the rest 5.5 us per edge): each phase is a short chain of dependent
// latencies (LDS -> VALU -> store issue), a workgroup's waves are all in the same phase, and with two waves per SIMD nothing hides
// anything -- not HBM (the stores' memory share is ~3 of 10 us), not the vector ALU (vector work in the matrix instructions' shadow
// buys nothing).  What it lacks is waves.  Here a target row is shared by TWO waves -- wave (r, hf) owns targets 32 hf .. 32 hf + 31
// of row ty0 + r: 32 fragment registers + 32 accumulators instead of 64 + 64 -- so a workgroup is 16 waves, four per SIMD, in the
// same LDS (tile + two operand buffers + the pooled region: 126 KB, one workgroup per CU as before).  Same arithmetic in the same
// order: bit-identical to the eight-wave walk.
// Restricted to what the headline shapes are: tiled planes (h1 % 4 == 0, w1 % 64 == 0), w2 == 64, h2 % 8 == 0, C == 128, maps whose
// 16-byte pieces are aligned; everything else keeps the kernel above.
__global__ __launch_bounds__(1024) void corr_build_fused16_kernel(const _Float16 *__restrict__ A, const _Float16 *__restrict__ Bm,
                                                                  FusedLevels L, int h1, int w1, int h2, int HW1p, int strips_per_wg,
                                                                  const int *__restrict__ oslots
#ifdef F16_PROF
                                                                  , unsigned long long *prof
#endif
                                                                  ) {
  constexpr int C = 128, W2 = 64, KSL = 8;
#ifdef F16_PROF   // scratch builds: time per phase of the walk, summed per wave (s_memtime ticks)
  unsigned long long f16_acc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, f16_last_ = __builtin_amdgcn_s_memtime();
#define F16_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); f16_acc_[i] += t_ - f16_last_; f16_last_ = t_; } while (0)
#else
#define F16_STAMP(i) (void)0
#endif
  constexpr int RP = W2 + 4, PITCH = FT_ROWS * RP + 4, RP1 = W2 / 2 + 4;
  extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
  _Float16 *T = smem;                          // [64][PITCH]   level 0 of the current strip, rounded
  _Float16 *Ab0 = smem + 64 * PITCH;           // two source-operand buffers of 16 KB
  _Float16 *P1 = Ab0 + 2 * 64 * 128;           // [64][4][RP1]  pooled levels of the current strip
  _Float16 *P2 = P1 + 64 * 4 * RP1;            // [64][2][16]
  _Float16 *P3 = P2 + 64 * 2 * (W2 / 4);       // [64][8]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = wave & 7, hf = wave >> 3;      // this wave's target row of the tile, its half of the row's targets
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {  // all row tiles of a (strip group, edge) on one XCD, side by side (see the kernel above)
    const int total = (int)(gridDim.x * gridDim.y * gridDim.z);
    const int lin = bx + (int)gridDim.x * (by + (int)gridDim.y * bz);
    const int q8 = total >> 3, r8 = total & 7, xk = lin & 7;
    const int logical = xk * q8 + min(xk, r8) + (lin >> 3);
    by = logical % (int)gridDim.y;
    const int rest = logical / (int)gridDim.y;
    bx = rest % (int)gridDim.x;
    bz = rest / (int)gridDim.x;
  }
  const int ty0 = by * FT_ROWS, e = bz;
  const int HW1 = h1 * w1, HW2 = h2 * W2, tiles_x = w1 >> 4;
  const int l31 = lane & 31, kh = (lane >> 5) * 8;
  const int nstrips = HW1p >> 6;
  const int s_begin = bx * strips_per_wg, s_end = min(nstrips, s_begin + strips_per_wg);
  if (s_begin >= s_end) return;   // (workgroup-uniform)
  const int eo = oslots ? oslots[e] : e;
  auto level_rsrc = [&](int lvl) {
    const size_t elems = (size_t)(h2 >> lvl) * (W2 >> lvl) * HW1p;
    return __builtin_amdgcn_make_buffer_rsrc((void *)(L.vs[lvl] + (size_t)eo * elems), 0, (int)(2 * elems), 0x00020000);
  };
  const unsigned plane_bytes = 2u * (unsigned)HW1p;

  // ---- source operand of a strip (= a 4 x 16 tile of the map): threads 0..255 read 8 pixels x 4 channels each ------------------
  half8 anat[4];
  auto request_a = [&](int strip) {
    if (tid < 256) {
      const int kg = tid >> 3, pg = tid & 7;
      const int tyi = strip / tiles_x, txi = strip - tyi * tiles_x;
      const int yy = 4 * tyi + (pg >> 1), xx = 16 * txi + 8 * (pg & 1);
      const _Float16 *src = A + (size_t)e * HW1 * C + (size_t)(4 * kg) * HW1 + yy * w1 + xx;
#pragma unroll
      for (int c = 0; c < 4; c++) anat[c] = *reinterpret_cast<const half8 *>(src + (size_t)c * HW1);
    }
  };
  auto stage_a = [&](_Float16 *dst) {   // -> LDS, fragment layout [k-step][32-pixel block][q][k-half][pixel][4 halves], scaled
    if (tid < 256) {
      const int kg = tid >> 3, pg = tid & 7, kbk = kg >> 2, qq = kg & 1, hfk = (kg >> 1) & 1, px = 8 * pg;
      // [k-step][32-pixel block][q][k-half]..: a lane's two halves of a fragment are 512 B apart and pair up into ONE two-address
      // read whose registers are the fragment (with q outside the block the pairs were (block 0, block 1): six moves per k-step)
      _Float16 *d = dst + (((((kbk * 2 + (px >> 5)) * 2 + qq) * 2 + hfk) * 32 + (px & 31)) * 4);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        half4 o;
#pragma unroll
        for (int c = 0; c < 4; c++) o[c] = anat[c][j];
        *reinterpret_cast<half4 *>(d + 4 * j) = o * (_Float16)0.25f;   // (corr.py:67-68: both maps / 4, one rounding to half)
      }
    }
  };
  request_a(s_begin);   // (in flight while the target fragments below are fetched and transposed)
  // ---- this wave's target fragments, resident for the whole walk: 32 targets x 128 channels ------------------------------------
  half8 bres[KSL];
  {
    const _Float16 *Be = Bm + (size_t)e * HW2 * C + (size_t)(ty0 + r) * W2;
    const int kq = lane >> 2, tx0 = 32 * hf + 8 * (lane & 3);
    half8 raw[KSL];
#pragma unroll
    for (int ks = 0; ks < KSL; ks++) raw[ks] = *reinterpret_cast<const half8 *>(Be + (size_t)(16 * ks + kq) * HW2 + tx0);
    _Float16 *scr = T + wave * (32 * 16);   // (the tile's LDS: nothing lives there before the first strip)
#pragma unroll
    for (int ks = 0; ks < KSL; ks++) {
#pragma unroll
      for (int j = 0; j < 8; j++) scr[(8 * (lane & 3) + j) * 16 + kq] = raw[ks][j] * (_Float16)0.25f;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      bres[ks] = *reinterpret_cast<const half8 *>(scr + l31 * 16 + kh);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int ks = 0; ks < KSL; ks++) asm volatile("" : "+v"(bres[ks]));   // (pinned: not re-read per strip)
  }
  stage_a(Ab0);
  F16_STAMP(0);

  constexpr unsigned OOR = 0x80000000u;
  const int q = lane & 15, g = lane >> 4, q4 = 4 * q, rr = q >> 2;   // the lane's quad of pixels: tile row rr, columns 4 (q & 3) ..
  // ---- level 0: Vs0[(ty - y1) mod h2][dx][pixel] = T[pixel][ty][(x1 + dx) mod 64]; the quads of tile row rr take the target
  // row (r + rr) mod 8 of the tile (whole lines per store instruction, see the kernel above); this wave: dx = 32 hf .. 32 hf + 31.
  // A lane owns EIGHT consecutive offsets D0 .. D0 + 7 (D0 = 32 hf + 8 g) of its quad's four pixels: pixel u's values for them are
  // the tile columns base + u + j, base = (x0 + D0) mod 64 a multiple of 4 -- three ALIGNED 8-byte reads per pixel (two for pixel
  // 0; blocks of four columns never straddle the row's end, so no wrap columns), and store j's quad is picked out of the 22
  // registers by two byte permutes.  (Until the round's last sessions: sixteen 2-byte reads along the diagonal per four stores --
  // 32 instead of 11 LDS instructions per lane and strip; the LDS pipe was the busiest unit of this phase.)
  // -DF16_L0_SPLIT (measured, slower: 13.8-14.4 against 12.3-13.6 us per edge on one box): the two waves of a target row split the
  // strip's level-0 stores in TIME as well -- the waves that pool (hf == 0) right behind the tile write, their partners in the NEXT
  // strip's product phase, in front of their matrix instructions (the tile stays valid until that strip's tile write): the product
  // phase grows by what the phase behind the tile write loses (profiles/r06_build16.txt, item 5).
  auto level0_stores = [&](int strip) {
    const int p0 = strip * 64;
    const int tyi = strip / tiles_x, txi = strip - tyi * tiles_x;
    const int ybase = 4 * tyi, xbase = 16 * txi;
    const int qx0 = xbase + 4 * (q & 3), qy = ybase + rr;
    {
      const int wrow = (r + rr) & (FT_ROWS - 1), tyq = ty0 + wrow;
      const __amdgpu_buffer_rsrc_t r0 = level_rsrc(0);
      const int D0 = 32 * hf + 8 * g;
      int dy = tyq - qy;
      dy += (dy < 0) ? h2 : 0;
      const unsigned voff = ((unsigned)dy * (unsigned)W2 + (unsigned)D0) * plane_bytes + 2u * (unsigned)(p0 + q4);
      const int base = (qx0 + D0) & (W2 - 1);
      const _Float16 *lb = T + q4 * PITCH + wrow * RP;
      typedef unsigned u2v __attribute__((ext_vector_type(2)));
      u2v R[4][3];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int col = (base + 4 * k) & (W2 - 1);
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (u + 7 >= 4 * k) R[u][k] = *reinterpret_cast<const u2v *>(lb + u * PITCH + col);
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        // element e = u + j of pixel u's twelve halves: dword (e >> 1) of its three reads, half e & 1
        const unsigned hA = (unsigned)(j & 1), hB = (unsigned)((j + 1) & 1);
        const unsigned sel = (2u * hA) | ((2u * hA + 1u) << 8) | ((4u + 2u * hB) << 16) | ((5u + 2u * hB) << 24);
        u2v d;
        d.x = __builtin_amdgcn_perm(R[1][(j + 1) >> 2][((j + 1) >> 1) & 1], R[0][j >> 2][(j >> 1) & 1], sel);
        d.y = __builtin_amdgcn_perm(R[3][(j + 3) >> 2][((j + 3) >> 1) & 1], R[2][(j + 2) >> 2][((j + 2) >> 1) & 1], sel);
#ifndef F16_ABLATE_L0
        __builtin_amdgcn_raw_buffer_store_b64(d, r0, voff, (unsigned)j * plane_bytes, FB_STORE_AUX);
#else
        __builtin_amdgcn_raw_buffer_store_b64(d, r0, voff | OOR, (unsigned)j * plane_bytes, FB_STORE_AUX);
#endif
      }
    }
  };
  // ---- levels 1..3 of a strip from the pooled region, behind the strip's pooling and a barrier.  -DF16_DEFER (measured, not faster:
  // 13.3-13.9 against 13.0-13.4 us per edge at 64x64, profiles/r06_build16.txt) issues the stores of strip s in strip s + 1's product
  // phase instead -- half of the waves in front of their matrix instructions, half behind them -- with the next strip's first
  // barrier in place of this one.
  auto pooled_stores = [&](int strip) {
    const int p0 = strip * 64;
    const int tyi = strip / tiles_x, txi = strip - tyi * tiles_x;
    const int ybase = 4 * tyi, xbase = 16 * txi;
    const int qx0 = xbase + 4 * (q & 3), qy = ybase + rr;
    {  // level 1: wave w takes pooled row w & 3 and the groups of four offsets (w >> 2) + 4 k; a quad's columns (x >> 1) - (x0 >> 1)
      const int w2l = W2 >> 1, h2l = h2 >> 1;
      const __amdgpu_buffer_rsrc_t rl = level_rsrc(1);
      const int tyl = wave & 3, grp = wave >> 2;
      const int tylq = (tyl + (rr >> 1)) & 3, tygq = (ty0 >> 1) + tylq;   // (the quads of tile rows 2, 3 take the next pooled row)
      const int xh = qx0 >> 1;
      const int o1 = ((qx0 + 1) >> 1) - xh, o2 = ((qx0 + 2) >> 1) - xh, o3 = ((qx0 + 3) >> 1) - xh;
      int t = (xh + 4 * grp + g) & (w2l - 1);
      int dy = tygq - (qy >> 1);
      dy += (dy < 0) ? h2l : 0;
      unsigned voff = ((unsigned)dy * (unsigned)w2l + (unsigned)(4 * grp + g)) * plane_bytes + 2u * (unsigned)(p0 + q4);
      const _Float16 *lb = P1 + (q4 * 4 + tylq) * RP1;
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const _Float16 *pp = lb + t;
        const unsigned short a0 = __builtin_bit_cast(unsigned short, pp[0]), a1 = __builtin_bit_cast(unsigned short, pp[4 * RP1 + o1]);
        const unsigned short a2 = __builtin_bit_cast(unsigned short, pp[8 * RP1 + o2]), a3 = __builtin_bit_cast(unsigned short, pp[12 * RP1 + o3]);
        typedef unsigned u2v __attribute__((ext_vector_type(2)));
        u2v d;
        d.x = (unsigned)a0 | ((unsigned)a1 << 16);
        d.y = (unsigned)a2 | ((unsigned)a3 << 16);
#ifndef F16_ABLATE_POOLST
        __builtin_amdgcn_raw_buffer_store_b64(d, rl, voff, 0, FB_STORE_AUX);
#else
        __builtin_amdgcn_raw_buffer_store_b64(d, rl, voff | OOR, 0, FB_STORE_AUX);
#endif
        voff += 16u * plane_bytes;
        t = (t + 16) & (w2l - 1);
      }
    }
    {  // levels 2 and 3: a lane is one source pixel, (ty_l, dx) segments are dealt to the waves
      const int x1 = xbase + (lane & 15), y1 = ybase + (lane >> 4);
      auto store_level = [&](int lvl, const _Float16 *Pl, int rows, int pitch_cols) {
        const int h2l = h2 >> lvl, w2l = W2 >> lvl;
        const __amdgpu_buffer_rsrc_t rl = level_rsrc(lvl);
        const int x1l = x1 >> lvl, y1l = y1 >> lvl;
        for (int seg = wave; seg < rows * w2l; seg += 16) {  // (wave-uniform)
          const int tyl = seg / w2l, dx = seg - tyl * w2l;
          int dy = (ty0 >> lvl) + tyl - y1l;
          dy += (dy < 0) ? h2l : 0;
          const int tx = (x1l + dx) & (w2l - 1);
          const _Float16 v = Pl[(lane * rows + tyl) * pitch_cols + tx];
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, v), rl,
#ifdef F16_ABLATE_POOLST
                                                OOR |
#endif
                                                (((unsigned)dy * (unsigned)w2l + (unsigned)dx) * plane_bytes + 2u * (unsigned)(p0 + lane)), 0, FB_STORE_AUX);
        }
      };
      store_level(2, P2, 2, W2 / 4);
      store_level(3, P3, 1, W2 / 8);
    }
  };
  for (int strip = s_begin; strip < s_end; strip++) {
    _Float16 *Ab = Ab0 + ((strip - s_begin) & 1) * (64 * 128);
    if (strip + 1 < s_end) request_a(strip + 1);   // consumed behind the tile write, BEFORE this strip's stores are issued
    lds_barrier();
    F16_STAMP(1);
#ifdef F16_L0_SPLIT
    if (hf == 1 && strip > s_begin) level0_stores(strip - 1);
#endif
#ifdef F16_DEFER
#ifndef F16_DEFER_SPLIT
#define F16_DEFER_SPLIT 1   // 0: every wave in front of its matrix instructions, 2: every wave behind them (A/B builds)
#endif
    const bool stores_first = F16_DEFER_SPLIT == 0 || (F16_DEFER_SPLIT == 1 && ((wave >> 2) & 1) == 0);   // (waves w, w + 4, w + 8, w + 12 share a SIMD)
    if (strip > s_begin && stores_first) pooled_stores(strip - 1);
#endif

    // ---- products: acc[i] = 32 x 32 tile (targets 32 hf .., sources 32 i ..) of target row ty0 + r -----------------------------
    float16v acc[2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int k = 0; k < 16; k++) acc[i][k] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSL; ks++) {
      half8 a[2];
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const _Float16 *fp = Ab + ((((ks * 2 + t) * 2 + 0) * 2 + (lane >> 5)) * 32 + l31) * 4;
        const half4 lo = *reinterpret_cast<const half4 *>(fp), hi = *reinterpret_cast<const half4 *>(fp + 2 * 32 * 4);
#pragma unroll
        for (int c = 0; c < 4; c++) a[t][c] = lo[c], a[t][4 + c] = hi[c];
      }
#pragma unroll
#ifndef F16_ABLATE_MFMA
      for (int i = 0; i < 2; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bres[ks], a[i], acc[i], 0, 0, 0);  // targets x sources
#else
      acc[0][0] += (float)a[0][0] + (float)a[1][1];
#endif
    }
    F16_STAMP(2);
#ifdef F16_DEFER
    if (strip > s_begin && !stores_first) pooled_stores(strip - 1);
    F16_STAMP(8);
#endif
    lds_barrier();  // every wave is done with the source operand and with the previous strip's tile
    F16_STAMP(3);
    // D layout: col = lane & 31 (source within the 32-block), row = (k & 3) + 8 (k >> 2) + 4 (lane >> 5) (target)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int rq = 0; rq < 4; rq++) {
        half4 v;
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = (_Float16)acc[i][4 * rq + k];
        *reinterpret_cast<half4 *>(T + (i * 32 + l31) * PITCH + r * RP + 32 * hf + 8 * rq + 4 * (lane >> 5)) = v;
      }
    lds_barrier();
    F16_STAMP(4);
    if (strip + 1 < s_end) stage_a(Ab0 + ((strip + 1 - s_begin) & 1) * (64 * 128));
    F16_STAMP(5);

#ifndef F16_L0_SPLIT
    level0_stores(strip);
#else
    if (hf == 0) level0_stores(strip);
#endif
    F16_STAMP(6);
    // ---- levels 1..3: threads 0..511 pool one 8 x 8 block each (from the ROUNDED level below each time) into the pooled region
    if (tid < 512) {
      const int src = tid >> 3, cb = tid & 7;
      const _Float16 *tb = T + src * PITCH + 8 * cb;
      _Float16 t8[8][8], q1[4][4], q2[2][2];
#pragma unroll
      for (int rw = 0; rw < 8; rw++) {
        const half4 lo = *reinterpret_cast<const half4 *>(tb + rw * RP), hi = *reinterpret_cast<const half4 *>(tb + rw * RP + 4);
#pragma unroll
        for (int c = 0; c < 4; c++) t8[rw][c] = lo[c], t8[rw][4 + c] = hi[c];
      }
#pragma unroll
      for (int rw = 0; rw < 4; rw++)
#pragma unroll
        for (int c = 0; c < 4; c++) q1[rw][c] = pool4(t8[2 * rw][2 * c], t8[2 * rw][2 * c + 1], t8[2 * rw + 1][2 * c], t8[2 * rw + 1][2 * c + 1]);
#pragma unroll
      for (int rw = 0; rw < 2; rw++)
#pragma unroll
        for (int c = 0; c < 2; c++) q2[rw][c] = pool4(q1[2 * rw][2 * c], q1[2 * rw][2 * c + 1], q1[2 * rw + 1][2 * c], q1[2 * rw + 1][2 * c + 1]);
      const _Float16 q3 = pool4(q2[0][0], q2[0][1], q2[1][0], q2[1][1]);
#pragma unroll
      for (int rw = 0; rw < 4; rw++) {
        half4 v;
#pragma unroll
        for (int c = 0; c < 4; c++) v[c] = q1[rw][c];
        _Float16 *row1 = P1 + (src * 4 + rw) * RP1;
        *reinterpret_cast<half4 *>(row1 + 4 * cb) = v;
        if (cb == 0) *reinterpret_cast<half4 *>(row1 + W2 / 2) = v;   // columns 0..3 once more behind column 31
      }
#pragma unroll
      for (int rw = 0; rw < 2; rw++) {
        half2v v;
        v.x = q2[rw][0], v.y = q2[rw][1];
        *reinterpret_cast<half2v *>(P2 + (src * 2 + rw) * (W2 / 4) + 2 * cb) = v;
      }
      P3[src * (W2 / 8) + cb] = q3;
    }
#ifndef F16_DEFER
    lds_barrier();
    F16_STAMP(7);
    pooled_stores(strip);
#endif
    (void)OOR;
    F16_STAMP(9);
  }
#ifdef F16_L0_SPLIT
  if (hf == 1) level0_stores(s_end - 1);   // (nothing has touched the last strip's tile)
#endif
#ifdef F16_DEFER
  lds_barrier();
  pooled_stores(s_end - 1);
#endif
#ifdef F16_PROF
  if (lane == 0) {
    unsigned long long *ps = prof + ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + wave) * 12;
    for (int i = 0; i < 10; i++) ps[i] = f16_acc_[i];
    ps[10] = (unsigned long long)(s_end - s_begin);
  }
#endif
}

// =====================================================================================================================
// The sixteen-wave strip walk for the maps whose planes keep the LINEAR pixel order (round 6, last session): 32 < w2 <= 64,
// C = 128, any h2 -- 55 x 55 is what the reference's TUM-VI demo runs on (512 x 512 resized to 440 x 440,
// demo_vio_tumvi.py:55-60).  Same work split as corr_build_fused16_kernel (a target row shared by two waves, four waves per SIMD,
// tile + two operand buffers + pooled region in LDS), same arithmetic in the same order as the eight-wave walk
// (corr_build_fused_kernel<2, true>): bit-identical to it.  What differs from the tiled form: a strip is 64 consecutive pixels of
// the flattened map and may span a row end (quads that do are stored by the whole wave, sixteen offsets per instruction), the map
// width is not a power of two (wraps by compare-and-subtract; the tile's wrap columns sit behind column w2 - 1 and are the second
// wave's own products: its fragments for those columns are targets 0..3 of the row, so no write of the tile is masked; the pooled
// region has no wrap columns, the level-1 loop wraps its offsets itself), the last row tile may be partial, and the operands are
// the k-block-major copies (16-byte pieces of the caller's maps are not aligned at these widths).  The two waves of a row split the
// offsets dx of the level-0 lines (batches of four lines; the wave that also pools takes one batch less).
// W2C > 0: the map width as a compile-time constant (w1 == w2 == W2C: the wraps, the splits of the lines between the waves and the
// divisions by the width fold into constants and the store loops unroll); 0: any width at run time.
template <int W2C>
__global__ __launch_bounds__(1024) void corr_build_fused16g_kernel(const _Float16 *__restrict__ A, const _Float16 *__restrict__ Bm,
                                                                   FusedLevels L, int h1, int w1_, int h2, int w2_, int HW1p,
                                                                   float inv_w1_, int strips_per_wg, const int *__restrict__ oslots
#ifdef F16_PROF
                                                                   , unsigned long long *prof
#endif
                                                                   ) {
  constexpr int C = 128, W2P = 64, KSL = 8;
  const int w2 = W2C > 0 ? W2C : w2_, w1 = W2C > 0 ? W2C : w1_;
  const float inv_w1 = W2C > 0 ? 1.0f / (float)W2C : inv_w1_;
#ifdef F16_PROF   // scratch builds: time per phase of the walk, summed per wave (s_memtime ticks; F16_STAMP: see the kernel above)
  unsigned long long f16_acc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, f16_last_ = __builtin_amdgcn_s_memtime();
#endif
  constexpr int RP = W2P + 4, PITCH = FT_ROWS * RP + 4, RP1 = W2P / 2 + 4;
  extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
  _Float16 *T = smem;                          // [64][PITCH]   level 0 of the current strip, rounded
  _Float16 *Ab0 = smem + 64 * PITCH;           // two source-operand buffers of 16 KB
  _Float16 *P1 = Ab0 + 2 * 64 * 128;           // [64][4][RP1]  pooled levels of the current strip
  _Float16 *P2 = P1 + 64 * 4 * RP1;            // [64][2][16]
  _Float16 *P3 = P2 + 64 * 2 * (W2P / 4);      // [64][8]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = wave & 7, hf = wave >> 3;      // this wave's target row of the tile, its half of the row's targets
  const int ty0 = blockIdx.y * FT_ROWS, e = blockIdx.z;
  const int HW1 = h1 * w1, HW2 = h2 * w2;
  const int l31 = lane & 31, kh = (lane >> 5) * 8;
  const int nstrips = HW1p >> 6;
  const int s_begin = blockIdx.x * strips_per_wg, s_end = min(nstrips, s_begin + strips_per_wg);
  if (s_begin >= s_end) return;   // (workgroup-uniform)
  const int eo = oslots ? oslots[e] : e;
  auto level_rsrc = [&](int lvl) {
    const size_t elems = (size_t)(h2 >> lvl) * (w2 >> lvl) * HW1p;
    return __builtin_amdgcn_make_buffer_rsrc((void *)(L.vs[lvl] + (size_t)eo * elems), 0, (int)(2 * elems), 0x00020000);
  };
  const unsigned plane_bytes = 2u * (unsigned)HW1p;
  constexpr unsigned OOR = 0x80000000u;

  // ---- source operand of a strip: 1024 pieces of 16 bytes ([k-block][pixel][half of the 16 channels]), one per thread ----------
  half8 apre;
  const int a_kbk = tid >> 7, a_px = (tid & 127) >> 1, a_hf = tid & 1;
  auto request_a = [&](int strip) {
    apre = *reinterpret_cast<const half8 *>(A + (size_t)e * HW1 * C + ((size_t)a_kbk * HW1 + min(strip * 64 + a_px, HW1 - 1)) * 16 + a_hf * 8);
  };
  auto stage_a = [&](_Float16 *dst) {   // -> LDS, fragment layout [k-step][32-pixel block][q][k-half][pixel][4 halves]
    const int fa = ((((a_kbk * 2 + (a_px >> 5)) * 2 + 0) * 2 + a_hf) * 32 + (a_px & 31)) * 4;
    half4 lo, hi;
#pragma unroll
    for (int c = 0; c < 4; c++) lo[c] = apre[c], hi[c] = apre[4 + c];
    *reinterpret_cast<half4 *>(dst + fa) = lo;
    *reinterpret_cast<half4 *>(dst + fa + 2 * 32 * 4) = hi;
  };
  request_a(s_begin);
  // ---- this wave's target fragments, resident for the whole walk: 32 targets x 128 channels (rows past the map: a valid row,
  // never stored; targets past the row's end: the next row's, never stored) ----------------------------------------------------
  half8 bres[KSL];
  {
    const int ty = min(ty0 + r, h2 - 1);
    // The tile wants columns 0..3 of a row once more behind column w2 - 1 (the level-0 loop's diagonal reads).  Those columns lie
    // in the second wave's half (w2 > 32): its fragments for them are targets 0..3 of the row, so the products land there by
    // themselves -- same operands, same order of accumulation, the same bits as in columns 0..3 -- and the tile write needs no mask
    int txl = 32 * hf + l31;
    txl -= (txl >= w2 && txl < w2 + 4) ? w2 : 0;
    const _Float16 *bpt = Bm + (size_t)e * HW2 * C + (size_t)min(ty * w2 + txl, HW2 - 1) * 16 + kh;
#pragma unroll
    for (int ks = 0; ks < KSL; ks++) bres[ks] = *reinterpret_cast<const half8 *>(bpt + (size_t)ks * 16 * HW2);
#pragma unroll
    for (int ks = 0; ks < KSL; ks++) asm volatile("" : "+v"(bres[ks]));   // (pinned: not re-read per strip)
  }
  stage_a(Ab0);
  F16_STAMP(0);

  const int q4 = (lane & 15) * 4, g = lane >> 4;     // the lane's quad of pixels and its line of a batch of four
  const int ipx = lane & 3, idx16 = lane >> 2;       // irregular quads: the lane's pixel of the quad, its offset of sixteen
#ifndef G16_L0_SHIFT
#define G16_L0_SHIFT 1   // batches of level-0 lines moved from the row's first wave, which also pools, to its partner (0 / 1 / 2 measured:
                         // 55x55 11.56 / 11.52 / 11.90, 44x60 8.07 / 7.73 / 8.30 us per edge, profiles/r06_build_g16.txt)
#endif
  const int nb = (w2 + 3) >> 2, nb0 = ((nb + 1) >> 1) - G16_L0_SHIFT; // batches of four level-0 lines: [0, nb0) to the row's first wave, the rest to its partner
  const int b_begin = hf ? nb0 : 0, b_end = hf ? nb : nb0;
  auto pixel_xy = [&](int pix, int &x, int &y) { sh_pixel_yx(min(pix, HW1 - 1), w1, inv_w1, false, y, x); };

  for (int strip = s_begin; strip < s_end; strip++) {
    const int p0 = strip * 64;
    _Float16 *Ab = Ab0 + ((strip - s_begin) & 1) * (64 * 128);
    if (strip + 1 < s_end) request_a(strip + 1);   // consumed behind the tile write, BEFORE this strip's stores are issued
    lds_barrier();
    F16_STAMP(1);

    // ---- products: acc[i] = 32 x 32 tile (targets 32 hf .., sources 32 i ..) of target row ty0 + r -----------------------------
    float16v acc[2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int k = 0; k < 16; k++) acc[i][k] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSL; ks++) {
      half8 a[2];
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const _Float16 *fp = Ab + ((((ks * 2 + t) * 2 + 0) * 2 + (lane >> 5)) * 32 + l31) * 4;
        const half4 lo = *reinterpret_cast<const half4 *>(fp), hi = *reinterpret_cast<const half4 *>(fp + 2 * 32 * 4);
#pragma unroll
        for (int c = 0; c < 4; c++) a[t][c] = lo[c], a[t][4 + c] = hi[c];
      }
#pragma unroll
      for (int i = 0; i < 2; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bres[ks], a[i], acc[i], 0, 0, 0);  // targets x sources
    }
    F16_STAMP(2);
    lds_barrier();  // every wave is done with the source operand and with the previous strip's tile
    F16_STAMP(3);
    // D layout: col = lane & 31 (source within the 32-block), row = (k & 3) + 8 (k >> 2) + 4 (lane >> 5) (target).  Columns
    // w2 .. w2 + 3 hold columns 0..3 once more (the second wave's products, see its fragments); the ones behind them are never read.
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int rq = 0; rq < 4; rq++) {
        half4 v;
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = (_Float16)acc[i][4 * rq + k];
        *reinterpret_cast<half4 *>(T + (i * 32 + l31) * PITCH + r * RP + 32 * hf + 8 * rq + 4 * (lane >> 5)) = v;
      }
    // (maps 61..63 wide: the wrap columns from 64 on are outside the second wave's half -- the first wave's own values go there)
    if (w2 + 3 >= W2P && hf == 0 && lane < 32) {
#pragma unroll
      for (int i = 0; i < 2; i++) {
        _Float16 *wr = T + (i * 32 + l31) * PITCH + r * RP + w2;
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (w2 + k >= W2P) wr[k] = (_Float16)acc[i][k];
      }
    }
    lds_barrier();
    F16_STAMP(4);
    if (strip + 1 < s_end) stage_a(Ab0 + ((strip + 1 - s_begin) & 1) * (64 * 128));
    F16_STAMP(5);

    // ---- the lane's quad of pixels (levels 0, 1) and the strip's irregular quads -----------------------------------------------
    int qx0, qy0, qx3, qy3;
    pixel_xy(p0 + q4, qx0, qy0);
    pixel_xy(p0 + q4 + 3, qx3, qy3);
    const bool quad_regular = (p0 + q4 + 3 < HW1) && (qy0 == qy3);
    const unsigned long long irregular = __ballot(!quad_regular && g == 0 && (p0 + q4 < HW1));   // bit = quad index (wave-uniform)
    // ---- level 0: Vs0[(ty - y1) mod h2][dx][pixel] = T[pixel][ty][(x1 + dx) mod w2], this wave's target row and half of the dx ----
    {
      const int ty = ty0 + r;
      if (ty < h2) {  // (wave-uniform)
        const __amdgpu_buffer_rsrc_t r0 = level_rsrc(0);
        if (quad_regular) {
          int t = qx0 + g + 4 * b_begin;
          t -= (t >= w2) ? w2 : 0;
          t -= (t >= w2) ? w2 : 0;
          int dy = ty - qy0;
          dy += (dy < 0) ? h2 : 0;
          unsigned voff = ((unsigned)dy * (unsigned)w2 + (unsigned)(g + 4 * b_begin)) * plane_bytes + 2u * (unsigned)(p0 + q4);
          const _Float16 *lb = T + q4 * PITCH + r * RP;
          // (a line dx = 4 b + g is this lane's to store iff b < b_end and dx < w2: ONE compare against the lane's limit; the column
          // counter wraps as an unsigned minimum: t + 4 - w2 is huge while t + 4 < w2)
          const unsigned dx_lim = (unsigned)min(w2, 4 * b_end + g);
          unsigned dxu = (unsigned)(4 * b_begin + g), tu = (unsigned)t;
          for (int b0 = b_begin; b0 < b_end; b0 += 4) {  // four batches at a time: their 16 LDS reads are in flight together
            unsigned short a[4][4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
              const _Float16 *pp = lb + tu;
#pragma unroll
              for (int u = 0; u < 4; u++) a[b][u] = __builtin_bit_cast(unsigned short, pp[u * (PITCH + 1)]);
              tu = min(tu + 4u, tu + 4u - (unsigned)w2);
            }
#pragma unroll
            for (int b = 0; b < 4; b++) {
              typedef unsigned u2v __attribute__((ext_vector_type(2)));
              u2v d;
              d.x = (unsigned)a[b][0] | ((unsigned)a[b][1] << 16);
              d.y = (unsigned)a[b][2] | ((unsigned)a[b][3] << 16);
              __builtin_amdgcn_raw_buffer_store_b64(d, r0, (dxu < dx_lim) ? voff : OOR, 0, FB_STORE_AUX);
              voff += 4u * plane_bytes;
              dxu += 4u;
            }
          }
        }
        for (unsigned long long m = irregular; m; m &= m - 1) {
          const int pl = 4 * (int)__builtin_ctzll(m) + ipx;  // this lane's pixel of the quad, within the strip
          int xi, yi;
          pixel_xy(p0 + pl, xi, yi);
          const bool pok = p0 + pl < HW1;
          int dy = ty - yi;
          dy += (dy < 0) ? h2 : 0;
          const _Float16 *row = T + pl * PITCH + r * RP;
          const unsigned vbase = (unsigned)dy * (unsigned)w2 * plane_bytes + 2u * (unsigned)(p0 + pl);
          for (int dx0 = 32 * hf; dx0 < min(w2, 32 * hf + 32); dx0 += 16) {
            const int dx = dx0 + idx16;
            int tx = xi + dx;
            tx -= (tx >= w2) ? w2 : 0;
            tx = min(tx, w2 - 1);  // (dx beyond the map in the last group: read something valid, store nothing)
            const _Float16 v = row[tx];
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, v), r0,
                                                  (pok && dx < w2) ? vbase + (unsigned)dx * plane_bytes : OOR, 0, FB_STORE_AUX);
          }
        }
      }
    }
    F16_STAMP(6);
    // ---- levels 1..3: threads 0..511 pool one 8 x 8 block each (from the ROUNDED level below each time) into the pooled region
    if (tid < 512) {
      const int src = tid >> 3, cb = tid & 7;
      const _Float16 *tb = T + src * PITCH + 8 * cb;
      _Float16 t8[8][8], q1[4][4], q2[2][2];
#pragma unroll
      for (int rw = 0; rw < 8; rw++) {
        const half4 lo = *reinterpret_cast<const half4 *>(tb + rw * RP), hi = *reinterpret_cast<const half4 *>(tb + rw * RP + 4);
#pragma unroll
        for (int c = 0; c < 4; c++) t8[rw][c] = lo[c], t8[rw][4 + c] = hi[c];
      }
#pragma unroll
      for (int rw = 0; rw < 4; rw++)
#pragma unroll
        for (int c = 0; c < 4; c++) q1[rw][c] = pool4(t8[2 * rw][2 * c], t8[2 * rw][2 * c + 1], t8[2 * rw + 1][2 * c], t8[2 * rw + 1][2 * c + 1]);
#pragma unroll
      for (int rw = 0; rw < 2; rw++)
#pragma unroll
        for (int c = 0; c < 2; c++) q2[rw][c] = pool4(q1[2 * rw][2 * c], q1[2 * rw][2 * c + 1], q1[2 * rw + 1][2 * c], q1[2 * rw + 1][2 * c + 1]);
      const _Float16 q3 = pool4(q2[0][0], q2[0][1], q2[1][0], q2[1][1]);
      // (no wrap columns in the pooled region here, unlike the eight-wave walk: with a run-time width they would be 2-byte writes
      // at an odd offset and the row's last block a masked write -- the level-1 loop wraps its three offset columns itself)
#pragma unroll
      for (int rw = 0; rw < 4; rw++) {
        half4 v;
#pragma unroll
        for (int c = 0; c < 4; c++) v[c] = q1[rw][c];
        *reinterpret_cast<half4 *>(P1 + (src * 4 + rw) * RP1 + 4 * cb) = v;
      }
#pragma unroll
      for (int rw = 0; rw < 2; rw++) {
        half2v v;
        v.x = q2[rw][0], v.y = q2[rw][1];
        *reinterpret_cast<half2v *>(P2 + (src * 2 + rw) * (W2P / 4) + 2 * cb) = v;
      }
      P3[src * (W2P / 8) + cb] = q3;
    }
    lds_barrier();
    F16_STAMP(7);
    {  // level 1: 4 rows x w2l offsets, four lines per store instruction; wave w takes pooled row w & 3 and the groups of four
       // offsets (w >> 2) + 4 k.  The quad's columns (x >> 1) - (x0 >> 1) are 0, 0|1, 1, 1|2
      const int w2l = w2 >> 1, h2l = h2 >> 1;
      const __amdgpu_buffer_rsrc_t rl = level_rsrc(1);
      const int tyl = wave & 3, grp = wave >> 2, tyg = (ty0 >> 1) + tyl;
      if (tyg < h2l) {  // floor sizes of avg_pool2d: the last partial row of the level below is dropped
        if (quad_regular) {
          const int xh = qx0 >> 1;
          const int o1 = ((qx0 + 1) >> 1) - xh, o2 = ((qx0 + 2) >> 1) - xh, o3 = ((qx0 + 3) >> 1) - xh;
          int t = xh + 4 * grp + g;
          t -= (t >= w2l) ? w2l : 0;
          t -= (t >= w2l) ? w2l : 0;
          int dy = tyg - (qy0 >> 1);
          dy += (dy < 0) ? h2l : 0;
          unsigned voff = ((unsigned)dy * (unsigned)w2l + (unsigned)(4 * grp + g)) * plane_bytes + 2u * (unsigned)(p0 + q4);
          const _Float16 *lb = P1 + (q4 * 4 + tyl) * RP1;
          for (int dx0 = 4 * grp; dx0 < w2l; dx0 += 16) {
            int c1 = t + o1, c2 = t + o2, c3 = t + o3;   // (t < w2l; the offsets are 0..2)
            c1 -= (c1 >= w2l) ? w2l : 0;
            c2 -= (c2 >= w2l) ? w2l : 0;
            c3 -= (c3 >= w2l) ? w2l : 0;
            const unsigned short a0 = __builtin_bit_cast(unsigned short, lb[t]), a1 = __builtin_bit_cast(unsigned short, lb[4 * RP1 + c1]);
            const unsigned short a2 = __builtin_bit_cast(unsigned short, lb[8 * RP1 + c2]), a3 = __builtin_bit_cast(unsigned short, lb[12 * RP1 + c3]);
            typedef unsigned u2v __attribute__((ext_vector_type(2)));
            u2v d;
            d.x = (unsigned)a0 | ((unsigned)a1 << 16);
            d.y = (unsigned)a2 | ((unsigned)a3 << 16);
            __builtin_amdgcn_raw_buffer_store_b64(d, rl, (dx0 + g < w2l) ? voff : OOR, 0, FB_STORE_AUX);
            voff += 16u * plane_bytes;
            t += 16;
            t -= (t >= w2l) ? w2l : 0;
            t -= (t >= w2l) ? w2l : 0;
          }
        }
        for (unsigned long long m = irregular; m; m &= m - 1) {  // (the four waves of a pooled row take every fourth group of 16 offsets)
          const int pl = 4 * (int)__builtin_ctzll(m) + ipx;
          int xi, yi;
          pixel_xy(p0 + pl, xi, yi);
          const bool pok = p0 + pl < HW1;
          int dy = tyg - (yi >> 1);
          dy += (dy < 0) ? h2l : 0;
          const _Float16 *row = P1 + (pl * 4 + tyl) * RP1;
          const unsigned vbase = (unsigned)dy * (unsigned)w2l * plane_bytes + 2u * (unsigned)(p0 + pl);
          for (int dx0 = 16 * grp; dx0 < w2l; dx0 += 64) {
            const int dx = dx0 + idx16;
            int tx = (xi >> 1) + dx;
            tx -= (tx >= w2l) ? w2l : 0;
            tx = min(tx, w2l - 1);
            const _Float16 v = row[tx];
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, v), rl,
                                                  (pok && dx < w2l) ? vbase + (unsigned)dx * plane_bytes : OOR, 0, FB_STORE_AUX);
          }
        }
      }
    }
    {  // levels 2 and 3: a lane is one source pixel, (ty_l, dx) segments are dealt to the waves
      const int p = p0 + lane;
      const bool active = p < HW1;
      int x1, y1;
      pixel_xy(p, x1, y1);
      auto store_level = [&](int lvl, const _Float16 *Pl, int rows, int pitch_cols) {
        const int h2l = h2 >> lvl, w2l = w2 >> lvl;
        const __amdgpu_buffer_rsrc_t rl = level_rsrc(lvl);
        const int x1l = x1 >> lvl, y1l = y1 >> lvl;
        for (int seg = wave; seg < rows * w2l; seg += 16) {  // (wave-uniform)
          const int tyl = seg / w2l, dx = seg - tyl * w2l;
          const int tyg = (ty0 >> lvl) + tyl;
          if (tyg >= h2l) continue;
          int dy = tyg - y1l;
          dy += (dy < 0) ? h2l : 0;
          int tx = x1l + dx;
          tx -= (tx >= w2l) ? w2l : 0;
          const _Float16 v = Pl[(lane * rows + tyl) * pitch_cols + tx];
          const unsigned voff = active ? ((unsigned)dy * (unsigned)w2l + (unsigned)dx) * plane_bytes + 2u * (unsigned)p : OOR;
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, v), rl, voff, 0, FB_STORE_AUX);
        }
      };
      store_level(2, P2, 2, W2P / 4);
      store_level(3, P3, 1, W2P / 8);
    }
    F16_STAMP(9);
  }
#ifdef F16_PROF
  if (lane == 0) {
    unsigned long long *ps = prof + ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + wave) * 12;
    for (int i = 0; i < 10; i++) ps[i] = f16_acc_[i];
    ps[10] = (unsigned long long)(s_end - s_begin);
  }
#endif
}

// =====================================================================================================================
// Maps 65..128 wide in the linear pixel order on SIXTEEN waves (round 6, last session): 28 x 107 is the KITTI-360 shape of
// BASELINE's configs.  One strip per workgroup like corr_build_fused_kernel<4, false> -- a 128-column tile (136 KB) leaves no LDS for
// a walk's second operand buffer and no registers for resident target fragments -- but a target row is shared by two waves: wave
// (r, hf) owns the columns 64 hf .. 64 hf + 63 of row ty0 + r (2 x 2 accumulator tiles, 64 registers, instead of 2 x 4), so four
// waves per SIMD instead of two wait for the row's target fragments, which come out of L2 in every strip (32 KB per row: the
// matrix phase of the eight-wave form is mostly that wait).  Same arithmetic in the same order: bit-identical.  The conventions of
// corr_build_fused16g_kernel: the tile's wrap columns are the second wave's own products, no masked tile writes, no wrap columns
// in the pooled region (which takes the dead tile's place, as in the eight-wave form), level-0 lines split between a row's waves.
__global__ __launch_bounds__(1024) void corr_build_fused16w_kernel(const _Float16 *__restrict__ A, const _Float16 *__restrict__ Bm,
                                                                   FusedLevels L, int C, int h1, int w1, int h2, int w2, int HW1p,
                                                                   float inv_w1, const int *__restrict__ oslots) {
  constexpr int W2P = 128;
  constexpr int RP = W2P + 4, PITCH = FT_ROWS * RP + 4, RP1 = W2P / 2 + 4;
  extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
  _Float16 *T = smem;                          // [64][PITCH]   level 0 of the strip, rounded; before that: the strip's source operand
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = wave & 7, hf = wave >> 3;      // this wave's target row of the tile, its half of the row's columns
  const int strip = blockIdx.x, ty0 = blockIdx.y * FT_ROWS, e = blockIdx.z;
  const int HW1 = h1 * w1, HW2 = h2 * w2;
  const int l31 = lane & 31, kh = (lane >> 5) * 8;
  const int p0 = strip * 64;
  const int eo = oslots ? oslots[e] : e;
  auto level_rsrc = [&](int lvl) {
    const size_t elems = (size_t)(h2 >> lvl) * (w2 >> lvl) * HW1p;
    return __builtin_amdgcn_make_buffer_rsrc((void *)(L.vs[lvl] + (size_t)eo * elems), 0, (int)(2 * elems), 0x00020000);
  };
  const unsigned plane_bytes = 2u * (unsigned)HW1p;
  constexpr unsigned OOR = 0x80000000u;
  const int ksteps = C >> 4;

  // ---- the strip's source operand into the (still unused) tile: [k-step][32-pixel block][q][k-half][pixel][4 halves] ----------
  {
    const _Float16 *Ae = A + (size_t)e * HW1 * C;
    for (int idx = tid; idx < ksteps * 128; idx += 1024) {  // 16-byte pieces: [k-block][pixel][half of the 16 channels]
      const int kbk = idx >> 7, rem = idx & 127, px = rem >> 1, hk = rem & 1;
      const half8 v = *reinterpret_cast<const half8 *>(Ae + ((size_t)kbk * HW1 + min(p0 + px, HW1 - 1)) * 16 + hk * 8);
      const int fa = ((((kbk * 2 + (px >> 5)) * 2 + 0) * 2 + hk) * 32 + (px & 31)) * 4;
      half4 lo, hi;
#pragma unroll
      for (int c = 0; c < 4; c++) lo[c] = v[c], hi[c] = v[4 + c];
      *reinterpret_cast<half4 *>(T + fa) = lo;
      *reinterpret_cast<half4 *>(T + fa + 2 * 32 * 4) = hi;
    }
  }
  // ---- this wave's target fragments: streamed per k-step, two steps ahead (rows past the map: a valid row, never stored; the
  // columns w2 .. w2 + 3: targets 0..3 of the row -- the tile's wrap columns; columns behind them: never read) -------------------
  const _Float16 *bp[2];
  {
    const int ty = min(ty0 + r, h2 - 1);
#pragma unroll
    for (int j = 0; j < 2; j++) {
      int txl = 64 * hf + 32 * j + l31;
      txl -= (txl >= w2 && txl < w2 + 4) ? w2 : 0;
      bp[j] = Bm + (size_t)e * HW2 * C + (size_t)min(ty * w2 + txl, HW2 - 1) * 16 + kh;
    }
  }
  const size_t kb = (size_t)HW2;
  half8 b0[2], b1[2], b2[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    b0[j] = *reinterpret_cast<const half8 *>(bp[j]);
    b1[j] = *reinterpret_cast<const half8 *>(bp[j] + (size_t)min(1, ksteps - 1) * 16 * kb);
  }
  __syncthreads();

  // ---- products: acc[i][j] = 32 x 32 tile (sources 32 i .., targets 64 hf + 32 j ..) of target row ty0 + r ------------------------
  float16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int k = 0; k < 16; k++) acc[i][j][k] = 0.f;
  for (int ks = 0; ks < ksteps; ks++) {
    const int kn = min(ks + 2, ksteps - 1);  // (the last steps re-request a fragment: no branch in the loop)
#pragma unroll
    for (int j = 0; j < 2; j++) b2[j] = *reinterpret_cast<const half8 *>(bp[j] + (size_t)kn * 16 * kb);
    half8 a[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const _Float16 *fp = T + ((((ks * 2 + t) * 2 + 0) * 2 + (lane >> 5)) * 32 + l31) * 4;
      const half4 lo = *reinterpret_cast<const half4 *>(fp), hi = *reinterpret_cast<const half4 *>(fp + 2 * 32 * 4);
#pragma unroll
      for (int c = 0; c < 4; c++) a[t][c] = lo[c], a[t][4 + c] = hi[c];
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0[j], a[i], acc[i][j], 0, 0, 0);  // targets x sources
#pragma unroll
    for (int j = 0; j < 2; j++) {
      b0[j] = b1[j];
      b1[j] = b2[j];
    }
  }
  lds_barrier();  // every wave is done with the source operand: the tile takes its place
  // D layout: col = lane & 31 (source within the 32-block), row = (k & 3) + 8 (k >> 2) + 4 (lane >> 5) (target)
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int rq = 0; rq < 4; rq++) {
        half4 v;
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = (_Float16)acc[i][j][4 * rq + k];
        *reinterpret_cast<half4 *>(T + (i * 32 + l31) * PITCH + r * RP + 64 * hf + 32 * j + 8 * rq + 4 * (lane >> 5)) = v;
      }
  // (maps 125..128 wide: the wrap columns from 128 on are outside the second wave's half -- the first wave's own values go there)
  if (w2 + 3 >= W2P && hf == 0 && lane < 32) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      _Float16 *wr = T + (i * 32 + l31) * PITCH + r * RP + w2;
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (w2 + k >= W2P) wr[k] = (_Float16)acc[i][0][k];
    }
  }
  lds_barrier();

  const int q4 = (lane & 15) * 4, g = lane >> 4;     // the lane's quad of pixels and its line of a batch of four
  const int ipx = lane & 3, idx16 = lane >> 2;       // irregular quads: the lane's pixel of the quad, its offset of sixteen
  const int nb = (w2 + 3) >> 2, nb0 = ((nb + 1) >> 1) - 1;   // batches of four level-0 lines: [0, nb0) to the row's first wave (it has the
  const int b_begin = hf ? nb0 : 0, b_end = hf ? nb : nb0;   // larger share of the pooling's reads in front of it), the rest to its partner
  auto pixel_xy = [&](int pix, int &x, int &y) { sh_pixel_yx(min(pix, HW1 - 1), w1, inv_w1, false, y, x); };
  int qx0, qy0, qx3, qy3;
  pixel_xy(p0 + q4, qx0, qy0);
  pixel_xy(p0 + q4 + 3, qx3, qy3);
  const bool quad_regular = (p0 + q4 + 3 < HW1) && (qy0 == qy3);
  const unsigned long long irregular = __ballot(!quad_regular && g == 0 && (p0 + q4 < HW1));   // bit = quad index (wave-uniform)
  // ---- level 0: Vs0[(ty - y1) mod h2][dx][pixel] = T[pixel][ty][(x1 + dx) mod w2], this wave's target row and share of the dx ----
  {
    const int ty = ty0 + r;
    if (ty < h2) {  // (wave-uniform)
      const __amdgpu_buffer_rsrc_t r0 = level_rsrc(0);
      if (quad_regular) {
        int t = qx0 + g + 4 * b_begin;
        t -= (t >= w2) ? w2 : 0;
        t -= (t >= w2) ? w2 : 0;
        int dy = ty - qy0;
        dy += (dy < 0) ? h2 : 0;
        unsigned voff = ((unsigned)dy * (unsigned)w2 + (unsigned)(g + 4 * b_begin)) * plane_bytes + 2u * (unsigned)(p0 + q4);
        const _Float16 *lb = T + q4 * PITCH + r * RP;
        for (int b0_ = b_begin; b0_ < b_end; b0_ += 4) {  // four batches at a time: their 16 LDS reads are in flight together
          unsigned short a[4][4];
#pragma unroll
          for (int b = 0; b < 4; b++) {
            const _Float16 *pp = lb + t;
#pragma unroll
            for (int u = 0; u < 4; u++) a[b][u] = __builtin_bit_cast(unsigned short, pp[u * (PITCH + 1)]);
            t += 4;
            t -= (t >= w2) ? w2 : 0;
          }
#pragma unroll
          for (int b = 0; b < 4; b++) {
            typedef unsigned u2v __attribute__((ext_vector_type(2)));
            u2v d;
            d.x = (unsigned)a[b][0] | ((unsigned)a[b][1] << 16);
            d.y = (unsigned)a[b][2] | ((unsigned)a[b][3] << 16);
            __builtin_amdgcn_raw_buffer_store_b64(d, r0, (b0_ + b < b_end && 4 * (b0_ + b) + g < w2) ? voff : OOR, 0, FB_STORE_AUX);
            voff += 4u * plane_bytes;
          }
        }
      }
      for (unsigned long long m = irregular; m; m &= m - 1) {
        const int pl = 4 * (int)__builtin_ctzll(m) + ipx;  // this lane's pixel of the quad, within the strip
        int xi, yi;
        pixel_xy(p0 + pl, xi, yi);
        const bool pok = p0 + pl < HW1;
        int dy = ty - yi;
        dy += (dy < 0) ? h2 : 0;
        const _Float16 *row = T + pl * PITCH + r * RP;
        const unsigned vbase = (unsigned)dy * (unsigned)w2 * plane_bytes + 2u * (unsigned)(p0 + pl);
        for (int dx0 = 64 * hf; dx0 < min(w2, 64 * hf + 64); dx0 += 16) {
          const int dx = dx0 + idx16;
          int tx = xi + dx;
          tx -= (tx >= w2) ? w2 : 0;
          tx = min(tx, w2 - 1);  // (dx beyond the map in the last group: read something valid, store nothing)
          const _Float16 v = row[tx];
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, v), r0,
                                                (pok && dx < w2) ? vbase + (unsigned)dx * plane_bytes : OOR, 0, FB_STORE_AUX);
        }
      }
    }
  }
  // ---- levels 1..3: every thread pools one 8 x 8 block (from the ROUNDED level below each time) in registers; behind a barrier (the
  // tile is dead: its level-0 stores have read it) the values take its place, behind a second one the sheared store loops read them
  _Float16 *P1 = T;                                // [64][4][RP1]
  _Float16 *P2 = P1 + 64 * 4 * RP1;                // [64][2][W2P / 4]
  _Float16 *P3 = P2 + 64 * 2 * (W2P / 4);          // [64][W2P / 8]
  {
    const int src = tid >> 4, cb = tid & 15;
    const _Float16 *tb = T + src * PITCH + 8 * cb;
    _Float16 t8[8][8], q1[4][4], q2[2][2];
#pragma unroll
    for (int rw = 0; rw < 8; rw++) {
      const half4 lo = *reinterpret_cast<const half4 *>(tb + rw * RP), hi = *reinterpret_cast<const half4 *>(tb + rw * RP + 4);
#pragma unroll
      for (int c = 0; c < 4; c++) t8[rw][c] = lo[c], t8[rw][4 + c] = hi[c];
    }
#pragma unroll
    for (int rw = 0; rw < 4; rw++)
#pragma unroll
      for (int c = 0; c < 4; c++) q1[rw][c] = pool4(t8[2 * rw][2 * c], t8[2 * rw][2 * c + 1], t8[2 * rw + 1][2 * c], t8[2 * rw + 1][2 * c + 1]);
#pragma unroll
    for (int rw = 0; rw < 2; rw++)
#pragma unroll
      for (int c = 0; c < 2; c++) q2[rw][c] = pool4(q1[2 * rw][2 * c], q1[2 * rw][2 * c + 1], q1[2 * rw + 1][2 * c], q1[2 * rw + 1][2 * c + 1]);
    const _Float16 q3 = pool4(q2[0][0], q2[0][1], q2[1][0], q2[1][1]);
    lds_barrier();  // every read of the level-0 tile is done (its sheared store above included)
#pragma unroll
    for (int rw = 0; rw < 4; rw++) {
      half4 v;
#pragma unroll
      for (int c = 0; c < 4; c++) v[c] = q1[rw][c];
      *reinterpret_cast<half4 *>(P1 + (src * 4 + rw) * RP1 + 4 * cb) = v;
    }
#pragma unroll
    for (int rw = 0; rw < 2; rw++) {
      half2v v;
      v.x = q2[rw][0], v.y = q2[rw][1];
      *reinterpret_cast<half2v *>(P2 + (src * 2 + rw) * (W2P / 4) + 2 * cb) = v;
    }
    P3[src * (W2P / 8) + cb] = q3;
  }
  lds_barrier();
  {  // level 1: 4 rows x w2l offsets, four lines per store instruction; wave w takes pooled row w & 3 and the groups of four
     // offsets (w >> 2) + 4 k.  The quad's columns (x >> 1) - (x0 >> 1) are 0, 0|1, 1, 1|2
    const int w2l = w2 >> 1, h2l = h2 >> 1;
    const __amdgpu_buffer_rsrc_t rl = level_rsrc(1);
    const int tyl = wave & 3, grp = wave >> 2, tyg = (ty0 >> 1) + tyl;
    if (tyg < h2l) {  // floor sizes of avg_pool2d: the last partial row of the level below is dropped
      if (quad_regular) {
        const int xh = qx0 >> 1;
        const int o1 = ((qx0 + 1) >> 1) - xh, o2 = ((qx0 + 2) >> 1) - xh, o3 = ((qx0 + 3) >> 1) - xh;
        int t = xh + 4 * grp + g;
        t -= (t >= w2l) ? w2l : 0;
        t -= (t >= w2l) ? w2l : 0;
        int dy = tyg - (qy0 >> 1);
        dy += (dy < 0) ? h2l : 0;
        unsigned voff = ((unsigned)dy * (unsigned)w2l + (unsigned)(4 * grp + g)) * plane_bytes + 2u * (unsigned)(p0 + q4);
        const _Float16 *lb = P1 + (q4 * 4 + tyl) * RP1;
        for (int dx0 = 4 * grp; dx0 < w2l; dx0 += 16) {
          int c1 = t + o1, c2 = t + o2, c3 = t + o3;   // (t < w2l; the offsets are 0..2)
          c1 -= (c1 >= w2l) ? w2l : 0;
          c2 -= (c2 >= w2l) ? w2l : 0;
          c3 -= (c3 >= w2l) ? w2l : 0;
          const unsigned short a0 = __builtin_bit_cast(unsigned short, lb[t]), a1 = __builtin_bit_cast(unsigned short, lb[4 * RP1 + c1]);
          const unsigned short a2 = __builtin_bit_cast(unsigned short, lb[8 * RP1 + c2]), a3 = __builtin_bit_cast(unsigned short, lb[12 * RP1 + c3]);
          typedef unsigned u2v __attribute__((ext_vector_type(2)));
          u2v d;
          d.x = (unsigned)a0 | ((unsigned)a1 << 16);
          d.y = (unsigned)a2 | ((unsigned)a3 << 16);
          __builtin_amdgcn_raw_buffer_store_b64(d, rl, (dx0 + g < w2l) ? voff : OOR, 0, FB_STORE_AUX);
          voff += 16u * plane_bytes;
          t += 16;
          t -= (t >= w2l) ? w2l : 0;
          t -= (t >= w2l) ? w2l : 0;
        }
      }
      for (unsigned long long m = irregular; m; m &= m - 1) {  // (the four waves of a pooled row take every fourth group of 16 offsets)
        const int pl = 4 * (int)__builtin_ctzll(m) + ipx;
        int xi, yi;
        pixel_xy(p0 + pl, xi, yi);
        const bool pok = p0 + pl < HW1;
        int dy = tyg - (yi >> 1);
        dy += (dy < 0) ? h2l : 0;
        const _Float16 *row = P1 + (pl * 4 + tyl) * RP1;
        const unsigned vbase = (unsigned)dy * (unsigned)w2l * plane_bytes + 2u * (unsigned)(p0 + pl);
        for (int dx0 = 16 * grp; dx0 < w2l; dx0 += 64) {
          const int dx = dx0 + idx16;
          int tx = (xi >> 1) + dx;
          tx -= (tx >= w2l) ? w2l : 0;
          tx = min(tx, w2l - 1);
          const _Float16 v = row[tx];
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, v), rl,
                                                (pok && dx < w2l) ? vbase + (unsigned)dx * plane_bytes : OOR, 0, FB_STORE_AUX);
        }
      }
    }
  }
  {  // levels 2 and 3: a lane is one source pixel, (ty_l, dx) segments are dealt to the waves
    const int p = p0 + lane;
    const bool active = p < HW1;
    int x1, y1;
    pixel_xy(p, x1, y1);
    auto store_level = [&](int lvl, const _Float16 *Pl, int rows, int pitch_cols) {
      const int h2l = h2 >> lvl, w2l = w2 >> lvl;
      const __amdgpu_buffer_rsrc_t rl = level_rsrc(lvl);
      const int x1l = x1 >> lvl, y1l = y1 >> lvl;
      for (int seg = wave; seg < rows * w2l; seg += 16) {  // (wave-uniform)
        const int tyl = seg / w2l, dx = seg - tyl * w2l;
        const int tyg = (ty0 >> lvl) + tyl;
        if (tyg >= h2l) continue;
        int dy = tyg - y1l;
        dy += (dy < 0) ? h2l : 0;
        int tx = x1l + dx;
        tx -= (tx >= w2l) ? w2l : 0;
        const _Float16 v = Pl[(lane * rows + tyl) * pitch_cols + tx];
        const unsigned voff = active ? ((unsigned)dy * (unsigned)w2l + (unsigned)dx) * plane_bytes + 2u * (unsigned)p : OOR;
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, v), rl, voff, 0, FB_STORE_AUX);
      }
    };
    store_level(2, P2, 2, W2P / 4);
    store_level(3, P3, 1, W2P / 8);
  }
}

// defined in corr_build.hip
__global__ void fmap_pixel_major_kernel(const _Float16 *in, _Float16 *out, int C, int HW, int kb, int w_tiled, int HWo, int w_grid);
__global__ void fmap_pixel_major_pair_kernel(const _Float16 *in1, _Float16 *out1, int w_tiled1, const _Float16 *in2, _Float16 *out2,
                                             int C, int HW, int kb, int n, int HWo1, int w_grid1);

}  // namespace dba

using namespace dba;

extern "C" {

int dba_corr_volume_build_sheared_supported(int C, int h1, int w1, int h2, int w2, int num_levels) {
  if (C <= 0 || (C % 16) != 0 || num_levels != 4 || h1 <= 0 || w1 <= 0) return 0;
  if (w2 > 128 || (h2 >> 3) < 1 || (w2 >> 3) < 1) return 0;
  if (C > 512) return 0;  // the strip's source operand (64 x C halves) is staged inside the level-0 tile
  const size_t elems0 = (size_t)h2 * w2 * (size_t)dba_corr_sheared_plane_elems(h1, w1);
  return (2 * elems0 < ((size_t)1 << 31)) ? 1 : 0;  // one edge-level is addressed through a 31-bit buffer range
}

int dba_corr_volume_build_sheared(const void *fmap1, const void *fmap2, void *const *sheared_levels, int n, int C,
                                  int h1, int w1, int h2, int w2, int num_levels, void *scratch,
                                  size_t scratch_bytes, dba_stream_t stream) {
  return dba_corr_volume_build_sheared_slots(fmap1, fmap2, sheared_levels, nullptr, n, C, h1, w1, h2, w2, num_levels, scratch,
                                             scratch_bytes, stream);
}

int dba_corr_volume_build_sheared_slots(const void *fmap1, const void *fmap2, void *const *sheared_levels,
                                        const int *out_slots, int n, int C, int h1, int w1, int h2, int w2, int num_levels,
                                        void *scratch, size_t scratch_bytes, dba_stream_t stream) {
  if (!dba_corr_volume_build_sheared_supported(C, h1, w1, h2, w2, num_levels)) return DBA_ERR_UNSUPPORTED;
  if (n < 0) return DBA_ERR_ARG;
  if (n == 0) return DBA_OK;
  if (!fmap1 || !fmap2 || !sheared_levels || !scratch) return DBA_ERR_ARG;
  if (scratch_bytes < dba_corr_volume_scratch_bytes(n, C, h1, w1, h2, w2)) return DBA_ERR_WORKSPACE;
  const int HW1 = h1 * w1, HW2 = h2 * w2;
  const int HW1p = dba_corr_sheared_plane_elems(h1, w1);
  hipStream_t s = (hipStream_t)stream;
  _Float16 *A = static_cast<_Float16 *>(scratch);
  _Float16 *Bm = reinterpret_cast<_Float16 *>(static_cast<char *>(scratch) + align_up((size_t)n * C * HW1p * 2, 256));
  // the source pixels in 4 x 16 tiles of the grid (h1g, w1g) >= the map: the planes' pixel order (common.h).  On a padded grid the
  // kernels below are handed the GRID as their source map -- a strip is a tile of it, the operand copy has a row per grid pixel,
  // and what the pad pixels' rows hold only ever reaches their own entries of the planes, which nobody reads
  int h1g, w1g;
  const int tiled = shear_grid(h1, w1, &h1g, &w1g) ? 1 : 0;
  const bool padded = tiled && (h1g != h1 || w1g != w1);
  const int h1k = tiled ? h1g : h1, w1k = tiled ? w1g : w1;
  // DBA_BUILD_KERNEL=classic|loop forces one form where both apply (tests, A/B runs); read once per process
  static const int force = [] {
    const char *e = getenv("DBA_BUILD_KERNEL");
    return !e ? 0 : (e[0] == 'c' ? 1 : (e[0] == 'l' ? 2 : 0));
  }();
  // the strip walk for every map up to 64 wide at C = 128 (since the row-end quads are stored by the whole wave it also
  // wins where strips span row ends: 55x55 13.3 against 14.3 us/edge)
  const bool loop_form = (w2 <= 64 && C == 128 && force != 1);
  // ... and, where 8-byte pieces of the maps are aligned, straight from the caller's [n][C][h][w] maps
  // (DBA_BUILD_OPERANDS=copy keeps the k-block-major copies: A/B runs)
  static const int native_mode = [] {   // 0: copies of both maps, 1: both maps as they lie, 2: the target map only
    const char *e = getenv("DBA_BUILD_OPERANDS");
    return !e ? 1 : (e[0] == 'c' ? 0 : (e[0] == 'b' ? 2 : 1));
  }();
  // sixteen waves per workgroup (DBA_BUILD_WAVES=8 keeps the eight-wave walks: A/B runs); on planes in the linear pixel order: the
  // general form, on the k-block-major copies
  static const bool waves16 = [] { const char *e = getenv("DBA_BUILD_WAVES"); return !(e && atoi(e) == 8); }();
  // (where 16-byte pieces of the maps are aligned -- widths that are multiples of 8 -- the eight-wave walk on the caller's own maps
  // is as fast or faster: 40 x 56 6.1 against 6.2 us per edge, 30 x 40 2.6 against 2.8; profiles/r06_build_g16.txt)
  const bool general16 = loop_form && waves16 && !tiled && w2 > 32 && h1 == h2 && w1 == w2 && ((w2 % 8) != 0 || (HW2 % 8) != 0);
  const bool native_b = loop_form && !general16 && !padded && native_mode != 0 && (w2 % 8 == 0) && (HW2 % 8 == 0);
  const bool native = native_b && native_mode == 1 && (HW1 % 8 == 0) && (w1 % 8 == 0);
  const int HW1o = tiled ? HW1p : HW1;   // pixels per map of the source operand's copy
  if (!native && !native_b && HW1 == HW2 && 2 * (long long)n <= 65535)   // both copies in one launch
    hipLaunchKernelGGL(fmap_pixel_major_pair_kernel, dim3((HW1 + 63) / 64, (C + 63) / 64, 2 * n), dim3(256), 0, s,
                       static_cast<const _Float16 *>(fmap1), A, tiled ? w1 : 0, static_cast<const _Float16 *>(fmap2), Bm, C, HW1, 16, n,
                       HW1o, w1g);
  else {
    if (!native)
      hipLaunchKernelGGL(fmap_pixel_major_kernel, dim3((HW1 + 63) / 64, (C + 63) / 64, n), dim3(256), 0, s,
                         static_cast<const _Float16 *>(fmap1), A, C, HW1, 16, tiled ? w1 : 0, HW1o, w1g);
    if (!native_b)
      hipLaunchKernelGGL(fmap_pixel_major_kernel, dim3((HW2 + 63) / 64, (C + 63) / 64, n), dim3(256), 0, s,
                         static_cast<const _Float16 *>(fmap2), Bm, C, HW2, 16, 0, HW2, 0);
  }
  FusedLevels L;
  for (int l = 0; l < 4; l++) L.vs[l] = static_cast<_Float16 *>(sheared_levels[l]);
  const dim3 grid(HW1p / 64, (h2 + FT_ROWS - 1) / FT_ROWS, n);
  static DeviceOnce attr_once;
  if (attr_once.needed()) {
    DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&corr_build_fused_kernel<2, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&corr_build_fused_kernel<4, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&corr_build_fused_kernel<2, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&corr_build_fused_kernel<2, true, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&corr_build_fused_kernel<2, true, false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_once.done();
  }
  const float inv_w1 = 1.0f / (float)w1k;
#ifdef FB_PROF
  static unsigned long long *prof = nullptr;
  if (!prof) { (void)hipMalloc(&prof, (size_t)64 << 20); }
  (void)hipMemsetAsync(prof, 0, (size_t)grid.x * grid.y * grid.z * 8 * 8 * 8, s);
#define FB_PROF_ARG , prof
#else
#define FB_PROF_ARG
#endif
  if (loop_form) {
    // strips per workgroup (at most 16).  One workgroup per CU (LDS), so the launch runs in ceil(workgroups / 256) rounds, each as
    // long as a walk: spw strips + the prologue (the waves' target fragments, 128 KB per workgroup: ~1.5 strips' worth).  Pick the
    // spw that minimises rounds x (spw + 1.5).  (Until the round's last session: as many as leave ~256 workgroups, capped -- right
    // for one edge (two strips, one round: 36 -> 28 us against 512 workgroups of one strip) and for 32 edges at 64x64 (4 rounds of
    // 16), but 32 edges at 48x64 ran 2.25 rounds of 16 strips where 3 rounds of 12 do, and a six-edge build two rounds of 12 where
    // one round of 16 does.)  DBA_BUILD_WG_TARGET=<n> keeps the old rule with n workgroups as its aim (A/B runs).
    const int nstrips = HW1p / 64;
    const long long rows = (long long)grid.y * n;
    static const int wg_target = [] { const char *e = getenv("DBA_BUILD_WG_TARGET"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
    static const int spw_cap = [] { const char *e = getenv("DBA_BUILD_SPW_CAP"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 16; }();
    int spw;
    if (wg_target > 0) {
      spw = (int)(((long long)nstrips * rows + wg_target - 1) / wg_target);
      spw = spw < 1 ? 1 : (spw > spw_cap ? spw_cap : spw);
    } else {
      spw = 1;
      double best = 1e300;
      // (the general sixteen-wave walk may take a whole row of strips: 36 edges at 55 x 55 -- the TUM-VI window -- are 252 rows, one
      // round of one walk each: 10.5 against 10.8 us per edge with the cap of 16; the other forms were measured with the cap)
      static const bool cap_given = getenv("DBA_BUILD_SPW_CAP") != nullptr;
      const int cap = (general16 && !cap_given) ? nstrips : spw_cap;
      for (int c = 1; c <= cap && c <= nstrips; c++) {
        const long long wgs = (long long)((nstrips + c - 1) / c) * rows;
        // (a last round that is partly empty is cheaper than a full one -- the walk is half chain, half memory traffic --: the
        // mean of whole and fractional rounds matches the measured picks, profiles/r06_build16.txt item 7)
        const double rounds = 0.5 * (double)((wgs + 255) / 256) + 0.5 * ((double)wgs / 256.0 < 1.0 ? 1.0 : (double)wgs / 256.0);
        const double cost = rounds * ((double)c + 1.5);
        if (cost <= best) best = cost, spw = c;   // (ties: the longer walk -- fewer prologues in total)
      }
    }
    if (force == 2 && spw < 2) spw = 2;
    static const bool spw_dbg = getenv("DBA_BUILD_DEBUG") != nullptr;
    if (spw_dbg) fprintf(stderr, "build: n=%d nstrips=%d rows=%lld spw=%d\n", n, nstrips, rows, spw);
    const size_t lds = sizeof(_Float16) * ((size_t)64 * (FT_ROWS * (64 + 4) + 4) + (size_t)2 * 64 * 128);
    const dim3 lgrid((nstrips + spw - 1) / spw, grid.y, n);
    const size_t lds16 = sizeof(_Float16) * ((size_t)64 * (FT_ROWS * (64 + 4) + 4) + (size_t)2 * 64 * 128 +
                                             (size_t)64 * (4 * (32 + 4) + 2 * 16 + 8));   // tile, two operand buffers, pooled region
#ifdef F16_PROF
    static unsigned long long *prof16 = nullptr;
    const size_t nslots = (size_t)lgrid.x * lgrid.y * lgrid.z * 16;
    if (!prof16) (void)hipMalloc(&prof16, (size_t)64 << 20);
    (void)hipMemsetAsync(prof16, 0, nslots * 12 * 8, s);
#define F16_PROF_ARG , prof16
    auto f16_prof_dump = [&]() {
      (void)hipStreamSynchronize(s);
      unsigned long long *hp = (unsigned long long *)malloc(nslots * 12 * 8);
      (void)hipMemcpy(hp, prof16, nslots * 12 * 8, hipMemcpyDeviceToHost);
      static const char *names[10] = {"prologue (per walk)", "operand wait + barrier 1", "products issued", "barrier 2", "tile write + barrier 3",
                                      "next operand into LDS", "level-0 loop", "pooling (+ barrier 4)", "pooled stores (not deferred)", "pooled stores behind the products"};
      for (int grp = 0; grp < 2; grp++) {   // waves 0..7 (they also pool; 0..3 stage the operand) and 8..15
        double ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, strips = 0, waves = 0;
        for (size_t i = 0; i < nslots; i++) {
          if ((int)((i & 15) >> 3) != grp || hp[i * 12 + 10] == 0) continue;
          for (int k = 0; k < 10; k++) ph[k] += (double)hp[i * 12 + k];
          strips += (double)hp[i * 12 + 10], waves += 1;
        }
        fprintf(stderr, "F16_PROF n=%d spw=%d waves %d..%d | ticks per strip and wave:", n, spw, 8 * grp, 8 * grp + 7);
        double tot = 0;
        for (int k = 1; k < 10; k++) fprintf(stderr, " %s %.0f,", names[k], ph[k] / strips), tot += ph[k] / strips;
        fprintf(stderr, " sum %.0f | %s %.0f per wave\n", tot, names[0], ph[0] / waves);
      }
      free(hp);
    };
#define F16_PROF_DUMP() f16_prof_dump()
#else
#define F16_PROF_ARG
#define F16_PROF_DUMP() (void)0
#endif
    if (general16) {
      static DeviceOnce attr16g_once;
      if (attr16g_once.needed()) {
        DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&corr_build_fused16g_kernel<0>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&corr_build_fused16g_kernel<55>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr16g_once.done();
      }
      // 55 x 55 (the reference's TUM-VI demo, demo_vio_tumvi.py:55-60) has its own instantiation (DBA_BUILD_G16_GENERIC=1: A/B runs)
      static const bool g16_generic = getenv("DBA_BUILD_G16_GENERIC") != nullptr;
      if (w2 == 55 && !g16_generic)
        hipLaunchKernelGGL(corr_build_fused16g_kernel<55>, lgrid, dim3(1024), lds16, s, A, Bm, L, h1, w1, h2, w2, HW1p, inv_w1, spw, out_slots F16_PROF_ARG);
      else
        hipLaunchKernelGGL(corr_build_fused16g_kernel<0>, lgrid, dim3(1024), lds16, s, A, Bm, L, h1, w1, h2, w2, HW1p, inv_w1, spw, out_slots F16_PROF_ARG);
      F16_PROF_DUMP();
    } else if (native && tiled && w2 == 64 && (h2 % 8) == 0 && waves16) {   // the shapes the headline runs on
      static DeviceOnce attr16_once;
      if (attr16_once.needed()) {
        DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&corr_build_fused16_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr16_once.done();
      }
      hipLaunchKernelGGL(corr_build_fused16_kernel, lgrid, dim3(1024), lds16, s, static_cast<const _Float16 *>(fmap1),
                         static_cast<const _Float16 *>(fmap2), L, h1, w1, h2, HW1p, spw, out_slots F16_PROF_ARG);
      F16_PROF_DUMP();
    } else if (native)
      hipLaunchKernelGGL((corr_build_fused_kernel<2, true, true>), lgrid, dim3(512), lds, s, static_cast<const _Float16 *>(fmap1),
                         static_cast<const _Float16 *>(fmap2), L, C, h1k, w1k, h2, w2, HW1p, inv_w1, spw, out_slots,
                         tiled FB_PROF_ARG);
    else if (native_b)
      hipLaunchKernelGGL((corr_build_fused_kernel<2, true, false, true>), lgrid, dim3(512), lds, s, A,
                         static_cast<const _Float16 *>(fmap2), L, C, h1k, w1k, h2, w2, HW1p, inv_w1, spw, out_slots,
                         tiled FB_PROF_ARG);
    else
      hipLaunchKernelGGL((corr_build_fused_kernel<2, true>), lgrid, dim3(512), lds, s, A, Bm, L, C, h1k, w1k, h2, w2, HW1p,
                         inv_w1, spw, out_slots, tiled FB_PROF_ARG);
  } else if (w2 <= 64) {
    const size_t lds = sizeof(_Float16) * (size_t)64 * (FT_ROWS * (64 + 4) + 4);  // the pooled levels live inside the dead tile
    hipLaunchKernelGGL((corr_build_fused_kernel<2, false>), grid, dim3(512), lds, s, A, Bm, L, C, h1k, w1k, h2, w2, HW1p, inv_w1,
                       1, out_slots, tiled FB_PROF_ARG);
  } else {
    const size_t lds = sizeof(_Float16) * (size_t)64 * (FT_ROWS * (128 + 4) + 4);
    // sixteen waves per workgroup where the planes keep the linear pixel order (DBA_BUILD_WAVES=8: the eight-wave form; A/B runs)
    if (waves16 && !tiled && h1 == h2 && w1 == w2 && (C % 16) == 0 && C <= 512) {
      static DeviceOnce attr16w_once;
      if (attr16w_once.needed()) {
        DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&corr_build_fused16w_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr16w_once.done();
      }
      hipLaunchKernelGGL(corr_build_fused16w_kernel, grid, dim3(1024), lds, s, A, Bm, L, C, h1k, w1k, h2, w2, HW1p, inv_w1, out_slots);
    } else
      hipLaunchKernelGGL((corr_build_fused_kernel<4, false>), grid, dim3(512), lds, s, A, Bm, L, C, h1k, w1k, h2, w2, HW1p, inv_w1,
                         1, out_slots, tiled FB_PROF_ARG);
  }
  DBA_LAUNCH_CHECK();
#ifdef FB_PROF
  if (w2 <= 64 && C == 128 && force != 1) {
    (void)hipStreamSynchronize(s);
    const int nstrips = HW1p / 64;
    const size_t nw = (size_t)((nstrips + 15) / 1) * grid.y * n * 8;  // upper bound of wave slots
    unsigned long long *hp = (unsigned long long *)malloc(nw * 64);
    (void)hipMemcpy(hp, prof, nw * 64, hipMemcpyDeviceToHost);
    double ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < nw; i++)
      for (int k = 0; k < 8; k++) ph[k] += (double)hp[i * 8 + k];
    const double per = ph[7] > 0 ? 1.0 / ph[7] : 0.0;
    fprintf(stderr, "FB_PROF loop n=%d | ticks per strip and wave: operand in LDS + barrier %.0f, mfma %.0f, tile write %.0f, next operand (waits for the previous strip's stores) %.0f, level-0 stores + pooling %.0f, pooled stores %.0f | strips %.0f\n",
            n, ph[1] * per, ph[2] * per, ph[3] * per, ph[6] * per, ph[4] * per, ph[5] * per, ph[7]);
    free(hp);
  } else
  {  // scratch builds: dump the phase timestamps of this launch (cycles of the 100 MHz constant clock)
    (void)hipStreamSynchronize(s);
    const size_t nw = (size_t)grid.x * grid.y * grid.z * 8;
    unsigned long long *hp = (unsigned long long *)malloc(nw * 64);
    (void)hipMemcpy(hp, prof, nw * 64, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0;
    double ph[7] = {0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < nw; i++) {
      const unsigned long long *r = hp + i * 8;
      if (r[0] < t0) t0 = r[0];
      if (r[6] > t1) t1 = r[6];
      for (int k = 1; k < 7; k++) ph[k] += (double)(r[k] - r[k - 1]);
    }
    fprintf(stderr, "FB_PROF n=%d waves=%zu span=%.2f us | per wave (us): stageA %.2f mfma %.2f wait+Twrite %.2f level0 %.2f pooled %.2f drain %.2f\n",
            n, nw, (t1 - t0) * 0.01, ph[1] / nw * 0.01, ph[2] / nw * 0.01, ph[3] / nw * 0.01, ph[4] / nw * 0.01, ph[5] / nw * 0.01,
            ph[6] / nw * 0.01);
    free(hp);
  }
#endif
  return DBA_OK;
}

}  // extern "C"
