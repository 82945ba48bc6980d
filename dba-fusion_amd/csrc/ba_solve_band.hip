// Damped solve of the reduced camera system, float64, one workgroup, register tiles allocated INSIDE THE SKYLINE.
//
// Replaces the host-side Eigen LLT / SimplicialLLT of the reference
// (/root/reference/src/droid_kernels.cu:200-218 solveDenseD, :1248-1269 SparseBlock::solve).
//
// Same elimination as ba_solve_tile.hip (block LDL^T with 2x2 pivots, raw column pairs published to LDS once, one
// barrier per pair, the owner of a diagonal tile publishes the next pivot inverse, one wave substitutes at the
// end), but threads and LDS are handed out only where the matrix can ever be non-zero:
//   * the skyline (first non-zero column tile of every 4-row tile) is measured on the device; fill-in cannot
//     leave it.  The right-hand side rides along as a dense last row tile;
//   * a thread owns one 4 x TW tile inside the skyline.  When twice the tile count fits the workgroup the tiles
//     are 4 x 2 (half the dependent work per thread and step: 25-KF windows), otherwise 4 x 4;
//   * a column pair's panel keeps only the rows that can be non-zero below it, so the panels of a 64-KF window
//     (n = 378, half-bandwidth ~36) fit in LDS (133 KB) where the dense lower triangle (575 KB) does not.
// Two workgroups (the one-tile-per-thread variant, when the band allows it): the elimination is a chain of n / 2
// dependent steps whatever the width of the band, so the chain is cut instead.  A separator S of whole tiles in the
// middle (wide enough that no row below it reaches a column above it) splits the unknowns into top | S | bottom;
// workgroup 0 eliminates the top block, workgroup 1 the bottom block in REVERSED order (its local system is the
// bottom block mirrored, then S), each on its own CU with its own panels.  When both have reached S they add each
// other's Schur-complement contribution to their S x S tiles and right-hand side (through global memory, one flag
// handshake), factor the - now identical - separator redundantly and back-substitute their own block: the same kernel
// body on a permuted local system of ~n / 2 + |S| unknowns, (n - |S|) / 4 + |S| / 2 steps instead of n / 2, and half the
// tiles per workgroup, so a 64-pose window fits the one-tile-per-thread variant with its panels in LDS.
// A sliding-window system is block-banded, so n up to 384 runs here; a system whose skyline does not fit
// (1024 tiles, 148 KB of panels) is left untouched, meta[3] stays 0 and the general kernel (ba_solve.hip) takes it.
#include "ba_kernels.h"

#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace dba {

constexpr int BD_THREADS = 1024;
constexpr int BD_MAX_N = 384;          // 6 column registers of 64 lanes in the substitution
constexpr int BD_TILES = 1536;         // tile slots: 1024 threads x 1 tile, or 512 threads x 3 tiles (BD_BIG_*)
constexpr int BD_BIG_THREADS = 512;    // the many-tile variant: 256 VGPRs per thread
constexpr int BD_BIG_SLOTS = 3;
constexpr int BD_INTS = 2048;          // first[100] pre[100] hiK[100] poff[196] tinfo[1536] flags[16]
constexpr int BD_PINV = 4 * (BD_MAX_N / 2);

typedef double bd2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double bd_rcp(double d) {
  double y = __builtin_amdgcn_rcp(d);
  double e = fma(-d, y, 1.0);
  y = fma(y, e, y);
  e = fma(-d, y, 1.0);
  y = fma(y, e, y);
  return y;
}

__device__ __forceinline__ double bd_readlane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// inclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ int bd_wave_scan(int v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(v, off, 64);
    if (lane >= off) v += o;
  }
  return v;
}

// THREADS x SLOTS tiles.  GP: the column-pair panels live in global memory (G, zero-filled here) for the
// substitution and only the pair being eliminated is kept in LDS (double-buffered), for skylines whose panels exceed
// LDS; such systems also need more than 1024 tiles, hence several tiles per thread and fewer, fatter threads.
template <int THREADS, int SLOTS, bool GP>
__global__ __launch_bounds__(THREADS) void ba_solve_band_kernel(const double *__restrict__ H,
                                                                const double *__restrict__ bvec,
                                                                const int *__restrict__ fpose, int ng,
                                                                double lm, double ep, float *__restrict__ dx,
                                                                int *__restrict__ meta, double *__restrict__ G,
                                                                int gcap, int nofb
#ifdef PROFILE_SOLVE
                                                                   , long long *__restrict__ prof
#endif
                                                                   ) {
#ifdef PROFILE_SOLVE
#define BPROF(slot) do { if (threadIdx.x == 0) { long long t_ = wall_clock64(), c_ = clock64(); prof[16 * blockIdx.x + slot] += t_ - tprev_; prof[16 * blockIdx.x + 8 + slot] += c_ - cprev_; tprev_ = t_; cprev_ = c_; } } while (0)
  long long tprev_ = wall_clock64(), cprev_ = clock64();
#else
#define BPROF(slot)
#endif
  // tile slots and what is left for the panels depend on the variant (threads x slots)
  constexpr int NTILES = (THREADS * SLOTS > BD_TILES) ? THREADS * SLOTS : BD_TILES;
  constexpr int NINTS = BD_INTS - BD_TILES + NTILES;
  constexpr int CAP = (SOLVE_MAX_LDS_BYTES - NINTS * 4 - BD_PINV * 8) / 8;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  int *first = (int *)smem;   // [T]    first non-zero column tile of a row tile (0 for the last, dense one)
  int *pre = first + 100;     // [T+1]  tiles before row tile I in the allocation order
  int *hiK = pre + 100;       // [KT]   last banded row tile below column tile K
  int *poff = hiK + 100;      // [npairs+1] doubles before the panel of a column pair
  int *tinfo = poff + 196;    // [1536] I | K << 8 | half << 16 | valid << 24
  int *flags = tinfo + NTILES;  // 0: fail, 1: tile width, 2: unsupported, 3: doubles of the largest panel
  double *pinv = smem + NINTS / 2;
  double *C = pinv + BD_PINV;

  const int tid = threadIdx.x, nt = blockDim.x;
  if (GP && meta[3] != 0) return;  // queued behind the one-tile-per-thread variant, which solved the system
  const int wave = tid >> 6, lane = tid & 63;
  constexpr bool SPLIT = (SLOTS == 1 && !GP);
  const int role = SPLIT ? (int)blockIdx.x : 0;  // 1: the workgroup of the bottom block (only launched with SPLIT)
  // The hand-shake words carry a per-launch generation number, counted ON THE DEVICE: a counter per workgroup in the workspace
  // (meta[28], meta[29]; the number itself in meta[30 + workgroup], read back behind the barriers below) -- a launch replayed from
  // a hipGraph gets a new number like any other (as the window solver's two-workgroup form does with meta[24..27])
  if (SPLIT && gridDim.x == 2 && tid == 0) {   // (one workgroup: no hand-shake)
    const unsigned g = (unsigned)atomicAdd(meta + 28 + role, 1) + 1u;
    __hip_atomic_store(meta + 30 + role, (int)((g & 0x0fffffffu) | 0x40000000u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  auto generation = [&]() { return __hip_atomic_load(meta + 30 + role, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };

  // ---- skyline of the whole system at tile level (first non-zero column tile of every row tile)
  // local index -> index in H / b / dx.  mode 0: identity; mode 1 (bottom workgroup): own block mirrored, then S
  int pmode = 0, p_own = 0, p_ua = 0;
  auto gidx = [&](int i) { return pmode == 0 ? i : (i < p_own ? ng - 1 - i : p_ua + (i - p_own)); };
  auto measure = [&](int nloc) {  // a wave reads four matrix rows of a tile row at a time, 64 columns per load
    const int Tlm = ((nloc + 1 + 3) >> 2) - 1;
    for (int I = wave; I < Tlm; I += nt >> 6) {
      const int ncol = min(4 * I + 4, nloc);
      unsigned long long seen[6];
#pragma unroll
      for (int ch = 0; ch < 6; ch++) {
        const int col = 64 * ch + lane;
        bool nz = false;
        if (64 * ch < ncol) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int i = 4 * I + r;  // (upper entries inside the diagonal tile mirror the lower ones)
            if (col < ncol && i < nloc) {
              const int gi = gidx(i), gk = gidx(col);
              nz |= (H[(size_t)max(gi, gk) * ng + min(gi, gk)] != 0.0);
            }
          }
        }
        seen[ch] = __ballot(nz);
      }
      int fc = ncol;
#pragma unroll
      for (int ch = 5; ch >= 0; ch--)
        if (seen[ch]) fc = 64 * ch + (int)__builtin_ctzll(seen[ch]);
      if (lane == 0) first[I] = min(first[I], fc >> 2);
    }
  };
  {
    const int Tg = (ng + 1 + 3) >> 2, KTg = (ng + 3) >> 2, Tlg = Tg - 1;
    if (tid < Tg) first[tid] = (tid == Tlg) ? 0 : min(tid, KTg - 1);
    if (tid < 16) flags[tid] = 0;
    __syncthreads();
    if (fpose) {  // skyline from the graph (ba_prepare_kernel): rows 4I .. 4I+3 belong to at most two poses
      if (tid < Tlg) {
        const int P = ng / 6;
        const int p0 = (4 * tid) / 6, p1 = min((4 * tid + 3) / 6, P - 1);
        const int fp = max(0, min(fpose[p0], fpose[p1]));
        first[tid] = min(first[tid], (6 * fp) >> 2);
      }
    } else {
      measure(ng);
    }
    // ng % 4 == 2: the last two matrix rows share the (dense) tile row of the right-hand side; the split below needs
    // their true first column tile
    if ((ng & 2) && wave == 1) {
      int fl = KTg - 1;
      if (fpose) {
        fl = (6 * max(0, fpose[ng / 6 - 1])) >> 2;
      } else {
        int fc = ng;
        for (int c0_ = 0; c0_ < ng - 2; c0_ += 64) {
          const int col = c0_ + lane;
          const bool nz = col < ng - 2 && (H[(size_t)(ng - 2) * ng + col] != 0.0 || H[(size_t)(ng - 1) * ng + col] != 0.0);
          const unsigned long long b = __ballot(nz);
          if (b) { fc = c0_ + (int)__builtin_ctzll(b); break; }
        }
        fl = min(fc, ng - 2) >> 2;
      }
      if (lane == 0) flags[11] = fl;
    } else if (!(ng & 2) && tid == 0) {
      flags[11] = KTg - 1;
    }
    __syncthreads();
  }
  // ---- two workgroups?  top = tiles [0, Ta), bottom = the last kb 4-blocks, S in between.  Needed: no row from the
  // bottom block on reaches a column tile < Ta (suffix minimum of `first`), |S| <= XS_MAX, blocks worth the handshake.
  constexpr int XS_MAX = 64, XROWS = XS_MAX + 8;                 // exchange: (|S| + right-hand side + tile padding) x |S|
  constexpr int XNEED = 2 * XROWS * XS_MAX;                      // doubles: the two contributions
  // the handshake flags live in meta[8..15], which no other kernel writes (the scratch is reused for panels and packed
  // triangles by the kernels queued behind: a stale double there could look like this launch's generation number)
  int *xflag = meta + 8;
  double *X = G;
  if (SPLIT) {
    if (wave == 0) {
      const int KTg = (ng + 3) >> 2, Tlg = ((ng + 1 + 3) >> 2) - 1;
      int best = 0, bTa = 0, bkb = 0;
      if (gridDim.x == 2 && G != nullptr && gcap >= XNEED && ng >= 96) {
        // sm[I] = min over I' >= I of first[I'] (and of the last two rows' when ng % 4 == 2): suffix minimum over the
        // <= 128 row tiles, two per lane (pre[] as scratch)
        const int i0 = 2 * lane, i1 = 2 * lane + 1;
        const int f1 = (i1 < Tlg) ? first[i1] : KTg, f0 = min((i0 < Tlg) ? first[i0] : KTg, f1);
        int run = min(f0, flags[11]);  // suffix minimum over the lanes >= this one
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int o = __shfl_down(run, off, 64);
          if (lane + off < 64) run = min(run, o);
        }
        const int above = __shfl_down(run, 1, 64);  // minimum over the lanes > this one
        const int tail = (lane < 63) ? min(above, flags[11]) : flags[11];
        if (i1 < Tlg) pre[i1] = min(f1, tail);
        if (i0 < Tlg) pre[i0] = min(f0, tail);
        __builtin_amdgcn_wave_barrier();
        // candidates: a bottom block of kb 4-blocks (rows from ub = ng - 4 kb on) allows top blocks up to sm[ub >> 2]
        // column tiles; the largest one leaves the smallest S
        for (int kb = 1 + lane; 4 * kb + 8 <= ng; kb += 64) {
          const int ub = ng - 4 * kb;
          const int Ta = min(pre[min(ub >> 2, Tlg - 1)], (ub - 4) >> 2);
          const int ws = ub - 4 * Ta;
          const int score = (Ta > 0 && ws >= 4 && ws <= XS_MAX) ? min(Ta, kb) : 0;
          if (score > best || (score == best && score > 0 && Ta < bTa)) best = score, bTa = Ta, bkb = kb;
        }
        // best over the lanes (ties: smallest Ta)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
          const int ob = __shfl_xor(best, off, 64), oT = __shfl_xor(bTa, off, 64), ok = __shfl_xor(bkb, off, 64);
          if (ob > best || (ob == best && ob > 0 && oT < bTa)) best = ob, bTa = oT, bkb = ok;
        }
      }
      if (lane == 0) {
        const bool go = best >= 6;  // at least 6 column tiles (12 steps) saved per workgroup
        flags[8] = go ? 1 : 0, flags[9] = bTa, flags[10] = bkb;
      }
    }
    __syncthreads();
  }
  const bool split = SPLIT && flags[8] != 0;
  // (for the tests and the profiling harness: which split was taken; meta[4..6] have no other use)
  if (SPLIT && tid == 0 && role == 0) meta[4] = flags[8], meta[5] = flags[9], meta[6] = flags[10];
  if (!split && role != 0) return;
  const int ua = split ? 4 * flags[9] : 0, ub = split ? ng - 4 * flags[10] : ng;
  const int n = split ? (role == 0 ? ub : ng - ua) : ng;         // size of this workgroup's (local) system
  const int nown = split ? (role == 0 ? ua : ng - ub) : n;       // its own block comes first, S behind it
  const int Kown = nown >> 2;                                    // (nown is a multiple of 4: the exchange sits between tiles)
  if (split && role == 1) pmode = 1, p_own = nown, p_ua = ua;
  const int T = (n + 1 + 3) >> 2, KT = (n + 3) >> 2, npairs = n >> 1, Tl = T - 1;
  if (split) {
    __syncthreads();
    if (role == 1) {
      // the mirrored system's skyline out of the global one: local row i is global row g = gidx(i); what it has left
      // of its diagonal are the entries of COLUMN g below the diagonal, which end at the last row tile whose skyline
      // reaches column tile g >> 2
      const int KTg = (ng + 3) >> 2, Tlg = ((ng + 1 + 3) >> 2) - 1;
      if (tid < KTg) hiK[tid] = tid;
      __syncthreads();
      if (tid < Tlg) {
        for (int Kq = first[tid]; Kq <= min(tid, KTg - 1); Kq++) atomicMax(&hiK[Kq], tid);
      } else if (tid == Tlg && (ng & 2)) {
        for (int Kq = flags[11]; Kq < KTg; Kq++) atomicMax(&hiK[Kq], Tlg);  // the two rows next to the right-hand side
      }
      __syncthreads();
      int fl = 0;
      if (tid < Tl) {
        int cmax = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int i = min(4 * tid + r, n - 1), g = gidx(i);
          cmax = max(cmax, min(4 * hiK[g >> 2] + 3, ng - 1));
        }
        fl = max(0, min(tid, (ng - 1 - cmax) >> 2));  // (global row c is local column ng - 1 - c of the own block)
      }
      __syncthreads();
      if (tid < T) first[tid] = min(fl, KT - 1);
      __syncthreads();
    }
    // S x S is taken as dense in both workgroups (they must factor the same tiles), the right-hand side row is dense
    if (tid < T) first[tid] = (tid == Tl) ? 0 : ((tid >= Kown) ? min(first[tid], Kown) : first[tid]);
  }
  __syncthreads();
  if (tid < KT) hiK[tid] = tid;
  __syncthreads();
  BPROF(0);
  // hiK[K] = last banded row tile whose skyline reaches column tile K
  if (tid < Tl) {
    for (int K = first[tid]; K <= min(tid, KT - 1); K++) atomicMax(&hiK[K], tid);
  }
  __syncthreads();
  if (wave == 0) {  // allocation order: banded rows first, the dense last row at the end (T <= 128: two per lane)
    int c[2], tot = 0;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int I = 2 * lane + q;
      c[q] = (I < Tl) ? min(I, KT - 1) - first[I] + 1 : 0;
      tot += c[q];
    }
    const int incl = bd_wave_scan(tot, lane);
    int run = incl - tot;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int I = 2 * lane + q;
      if (I < Tl) pre[I] = run;
      run += c[q];
    }
    const int total = __shfl(incl, 63, 64) + KT;
    if (lane == 0) {
      pre[Tl] = total - KT;
      pre[T] = total;
      flags[1] = (SLOTS == 1 && 2 * total <= nt) ? 2 : 4;
      if (total > nt * SLOTS) flags[2] = 1;
    }
  } else if (wave == 1) {  // panels: rows 4K .. 4 hiK[K] + 3 of the banded part, two zero rows (what the substitution
                           // reads for pivots below the panel), then the 4 rows of the last tile (+2: skew)
    int c[3], tot = 0;    // npairs <= 192: three per lane
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int sp = 3 * lane + q;
      const int Kq = min(sp >> 1, KT - 1);
      c[q] = (sp < npairs) ? 2 * (4 * (min(hiK[Kq], Tl - 1) - Kq + 1) + 2 + 4) + 2 : 0;
      tot += c[q];
    }
    int big = max(c[0], max(c[1], c[2]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) big = max(big, __shfl_xor(big, off, 64));
    const int incl = bd_wave_scan(tot, lane);
    int run = 2 + incl - tot;  // (C[0..1] stay zero)
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int sp = 3 * lane + q;
      if (sp < npairs) poff[sp] = run;
      run += c[q];
    }
    const int total = 2 + __shfl(incl, 63, 64);
    if (lane == 0) {
      poff[npairs] = total;
      flags[3] = big;
      if (GP ? (total > gcap || 2 * big > CAP) : (total > CAP)) flags[2] = 1;
    }
  }
  for (int e = tid; e < NTILES; e += nt) tinfo[e] = 0;
  __syncthreads();
  if (flags[2]) {  // skyline too large for one workgroup: the general kernel takes the system
    if (nofb) {  // (cannot happen when the caller's hint is true; should it, the update is zero rather than stale)
      for (int j = tid; j < ng; j += nt) dx[j] = 0.f;
    }
    if (tid == 0) {
      meta[3] = nofb ? 1 : 0;
      if (nofb) meta[1] = 1;
      if (split) {  // the partner learns it at the exchange and leaves as well
        xflag[2 + role] = 2;
        __threadfence();
        __hip_atomic_store(xflag + role, generation(), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  const int TW = flags[1], halves = 4 / TW;
  if (tid < Tl) {
    const int kend = min(tid, KT - 1);
    for (int K = first[tid]; K <= kend; K++)
      for (int hh = 0; hh < halves; hh++)
        tinfo[(pre[tid] + (K - first[tid])) * halves + hh] = tid | (K << 8) | (hh << 16) | (1 << 24);
  } else if (tid >= 128 && tid < 128 + KT) {
    const int K = tid - 128;
    for (int hh = 0; hh < halves; hh++) tinfo[(pre[Tl] + K) * halves + hh] = Tl | (K << 8) | (hh << 16) | (1 << 24);
  }
  if (GP) {
    for (int e = tid; e < poff[npairs]; e += nt) G[e] = 0.0;
    __threadfence_block();
  } else {
    for (int e = tid; e < poff[npairs]; e += nt) C[e] = 0.0;
  }
  __syncthreads();

  BPROF(1);
  // ---- this thread's tiles (slot q: allocation index tid + q * THREADS)
  bool valid[SLOTS];
  int I[SLOTS], K[SLOTS], sstart[SLOTS];
  double a[SLOTS][4][4];
  const int hh = (tinfo[tid] >> 16) & 0xff;       // half tiles exist only with one slot
  const int c0 = (TW == 2) ? 2 * hh : 0;          // first column of the tile inside its 4-column tile
  bool anyvalid = false;
  {
    const double *src[SLOTS][4][4];
    bool okm[SLOTS][4][4];
#pragma unroll
    for (int q = 0; q < SLOTS; q++) {
      const int info = tinfo[tid + q * THREADS];
      valid[q] = (info >> 24) & 1;
      I[q] = info & 0xff, K[q] = (info >> 8) & 0xff;
      anyvalid |= valid[q];
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const int i = 4 * I[q] + r, k = 4 * K[q] + c0 + c;
          okm[q][r][c] = valid[q] && (c < TW) && k < n && i <= n;
          // the bottom workgroup starts S x S and S's right-hand side from zero: the top one brings the original values
          if (split && role == 1 && k >= nown && i >= nown) okm[q][r][c] = false;
          const int gi = gidx(min(i, n - 1)), gk = gidx(min(k, n - 1));
          const double *src_ = (i == n) ? bvec + gk : H + (size_t)max(gi, gk) * ng + min(gi, gk);  // mirrored upper half on the diagonal
          src[q][r][c] = okm[q][r][c] ? src_ : H;
        }
    }
#pragma unroll
    for (int q = 0; q < SLOTS; q++)
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) a[q][r][c] = *src[q][r][c];
#pragma unroll
    for (int q = 0; q < SLOTS; q++)
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
          double v = okm[q][r][c] ? a[q][r][c] : 0.0;
          if (4 * I[q] + r == 4 * K[q] + c0 + c && 4 * I[q] + r < n && okm[q][r][c]) v += ep + lm * v;  // damping (:1252-1253)
          a[q][r][c] = v;
        }
#pragma unroll
    for (int q = 0; q < SLOTS; q++)  // (rows 4K.. of the column side live in row tile K)
      sstart[q] = valid[q] ? max(first[I[q]], first[K[q]]) : 0x7fffffff;
  }
  if (wave != 0 && __ballot(anyvalid) == 0ull) return;  // a finished wave no longer counts at the barriers
  int *fail = flags;
  const int pmax = flags[3];  // GP: doubles per LDS panel buffer

  auto publish_pinv = [&](int sp, double pa, double pb, double pc) {
    const double det = fma(-pb, pb, pa * pc);
    const bool ok = pa > 0.0 && det > 0.0;
    if (!ok) *fail = 1;
    const double idet = ok ? bd_rcp(det) : 0.0;
    bd2 lo;
    lo.x = pc * idet;
    lo.y = -pb * idet;
    *(bd2 *)(pinv + 4 * sp) = lo;
    pinv[4 * sp + 2] = pa * idet;
  };

  BPROF(2);
  // ---- factorisation (one instantiation per tile width; the choice is workgroup-uniform)
  auto factor = [&](auto twc) {
    constexpr int W = decltype(twc)::value;
#pragma unroll
    for (int q = 0; q < SLOTS; q++)
      if (valid[q] && I[q] == 0 && K[q] == 0 && (W == 4 || hh == 0)) publish_pinv(0, a[q][0][0], a[q][1][0], a[q][1][1]);
    // panel offset and the slot of the last row tile are read one step ahead (they sit on the chain otherwise)
    int pcur = poff[0], pnxt = 0;
    int last0 = 4 * (min(hiK[0], Tl - 1) - 0 + 1) + 2, lastn = 0;  // panel slot of the first row of the last row tile
    for (int Ks = 0; Ks < KT; Ks++) {
      if (SPLIT && split && Ks == Kown) {
        // ---- both blocks are eliminated: the workgroups swap their contributions to S x S and S's right-hand side
        BPROF(5);
        double *Xm = X + role * (XROWS * XS_MAX);
        const double *Xp = X + (1 - role) * (XROWS * XS_MAX);
        if (valid[0] && K[0] >= Kown) {
#pragma unroll
          for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < W; c++) {
              const int si = 4 * I[0] + r - nown, sk = 4 * K[0] + c0 + c - nown;
              if (si < XROWS && sk < XS_MAX) Xm[si * XS_MAX + sk] = a[0][r][c];
            }
        }
        __threadfence();
        __syncthreads();
        if (tid == 0) {
          xflag[2 + role] = (*fail != 0) ? 1 : 0;  // (a non-SPD pivot in either block fails the whole solve)
          __threadfence();
          __hip_atomic_store(xflag + role, generation(), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          const long long t0 = wall_clock64();
          int st = 0;
          while (__hip_atomic_load(xflag + (1 - role), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != generation()) {
            if (wall_clock64() - t0 > 100000000ll) {  // ~1 s: the partner never came; leave the system to the next kernel
              st = 2;
              break;
            }
            __builtin_amdgcn_s_sleep(4);
          }
          if (st == 0) st = __hip_atomic_load(xflag + 2 + (1 - role), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          flags[12] = st;
        }
        __syncthreads();
        const int st = flags[12];
        if (st == 2) {  // partner unsupported / absent: nothing has been written yet, the queue behind takes over
          if (nofb) {
            // nothing is queued behind (the caller knows this structure fits, so the partner is merely late by > 1 s): the
            // solve fails as a whole -- zero update, like a non-SPD system (:1263-1266) --, and the partner, should it
            // still arrive, finds this workgroup's contribution, completes, and learns the verdict at the second handshake
            for (int j = tid; j < ng; j += nt) dx[j] = 0.f;
            if (tid == 0) {
              xflag[6 + role] = 1;
              __threadfence();
              __hip_atomic_store(xflag + 4 + role, generation(), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
              meta[1] = 1;
              meta[3] = 1;
            }
          } else if (tid == 0) {
            meta[3] = 0;
          }
          __builtin_amdgcn_endpgm();
        }
        if (st == 1 && tid == 0) *fail = 1;
        // (every wave drops what its caches may hold of the partner's buffer, then the loads go out together)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (valid[0] && K[0] >= Kown) {
          double xv[4][W];
#pragma unroll
          for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < W; c++) {
              const int si = 4 * I[0] + r - nown, sk = 4 * K[0] + c0 + c - nown;
              xv[r][c] = (si < XROWS && sk < XS_MAX) ? __builtin_nontemporal_load(Xp + si * XS_MAX + sk) : 0.0;
            }
#pragma unroll
          for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < W; c++) a[0][r][c] += xv[r][c];
        }
        // the first pivot of S belongs to the summed block (its inverse was published from this workgroup's share)
        if (valid[0] && I[0] == Kown && K[0] == Kown && (W == 4 || hh == 0)) publish_pinv(2 * Kown, a[0][0][0], a[0][1][0], a[0][1][1]);
        BPROF(6);
      }
      auto step = [&](auto hc) {
        constexpr int h = decltype(hc)::value;
        const int s = 2 * Ks + h;
        double *P = GP ? C + (s & 1) * pmax : C + pcur;  // the pair being eliminated (LDS)
#pragma unroll
        for (int q = 0; q < SLOTS; q++) {
          if (valid[q] && K[q] == Ks && (W == 4 || hh == h)) {
            const int si = (I[q] == Tl) ? last0 : 4 * (I[q] - Ks);  // panel slot of this tile's first row
            constexpr int cc = (W == 4) ? 2 * h : 0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
              bd2 v;
              v.x = a[q][r][cc];
              v.y = a[q][r][cc + 1];
              *(bd2 *)(P + 2 * (si + r)) = v;
              if (GP) *(bd2 *)(G + pcur + 2 * (si + r)) = v;  // kept for the substitution
            }
          }
        }
        __syncthreads();
        pnxt = poff[min(s + 1, npairs)];
        if (h == 0) lastn = 4 * (min(hiK[min(Ks + 1, KT - 1)], Tl - 1) - (Ks + 1) + 1) + 2;
        const bool later = (W == 4) ? (h == 0) : (hh == 1 && h == 0);  // columns right of the pair inside column tile Ks
        const bd2 pv = *(const bd2 *)(pinv + 4 * s);
        const double p00 = pv.x, p01 = pv.y, p11 = pinv[4 * s + 2];
#pragma unroll
        for (int q = 0; q < SLOTS; q++) {
          const bool active = Ks >= sstart[q] && (K[q] > Ks || (K[q] == Ks && later));
          if (active) {
            const int si = (I[q] == Tl) ? last0 : 4 * (I[q] - Ks);
            const int sk = ((K[q] == Tl) ? last0 : 4 * (K[q] - Ks)) + c0;  // panel slot of the tile's first column-side row
            bd2 ri[4], rk[W];
#pragma unroll
            for (int r = 0; r < 4; r++) ri[r] = *(const bd2 *)(P + 2 * (si + r));
#pragma unroll
            for (int c = 0; c < W; c++) rk[c] = *(const bd2 *)(P + 2 * (sk + c));
#pragma unroll
            for (int c = 0; c < W; c++) {
              const double u0 = fma(p01, rk[c].y, p00 * rk[c].x);
              const double u1 = fma(p11, rk[c].y, p01 * rk[c].x);
#pragma unroll
              for (int r = 0; r < 4; r++) a[q][r][c] = fma(-ri[r].y, u1, fma(-ri[r].x, u0, a[q][r][c]));
            }
          }
        }
        // the owner of the next pivot publishes its inverse (whether or not this step touched the tile)
        constexpr int hn = 1 - h;
        const int Kn = (h == 0) ? Ks : Ks + 1;
#pragma unroll
        for (int q = 0; q < SLOTS; q++) {
          // (with two workgroups the first pivot of S is published after the exchange, from the summed block)
          if (valid[q] && I[q] == Kn && K[q] == Kn && 2 * (s + 1) < n && (W == 4 || hh == hn) &&
              !(SPLIT && split && s + 1 == 2 * Kown)) {
            constexpr int pc0 = (W == 4) ? 2 * hn : 0;
            publish_pinv(s + 1, a[q][2 * hn][pc0], a[q][2 * hn + 1][pc0], a[q][2 * hn + 1][pc0 + 1]);
          }
        }
        pcur = pnxt;
      };
      step(std::integral_constant<int, 0>{});
      if (4 * Ks + 2 < n) step(std::integral_constant<int, 1>{});
      last0 = lastn;
    }
  };
  if (SLOTS == 1 && TW == 2) factor(std::integral_constant<int, 2>{});
  else factor(std::integral_constant<int, 4>{});
  if (GP) __threadfence_block();
  __syncthreads();
  BPROF(3);

  // ---- L^T-side block substitution by wave 0: lane l holds t_j for the columns j = l + 64 r, r < 6
  if (wave != 0) return;
  const double *Cs = GP ? G : C;  // all published panels
  constexpr int RMAX = BD_MAX_N / 64;
  double t[RMAX], xo[RMAX];
  int cb[RMAX], cl[RMAX];  // C(i, j) = C[cb + 2 i] for banded rows i, C[cl + 2 i] for rows of the last tile
  int cap[RMAX];           // pair index of the two zero rows behind the banded rows of column j's panel
  const int Rn = (n + 63) >> 6;
#pragma unroll
  for (int r = 0; r < RMAX; r++) {
    const int j = lane + 64 * r;
    const int jc = (j < n) ? j : 0;
    const int Kc = jc >> 2;
    cb[r] = poff[jc >> 1] - 8 * Kc + (jc & 1);
    const int hi = min(hiK[Kc], Tl - 1);
    cl[r] = poff[jc >> 1] + 8 * (hi - Kc + 1) + 4 - 8 * Tl + (jc & 1);
    cap[r] = 2 * hi + 2;
    t[r] = (j < n) ? Cs[cl[r] + 2 * n] : 0.0;
    xo[r] = 0.0;
  }
  auto sweep = [&](auto rc) {
    constexpr int r0 = decltype(rc)::value;
    const int shi = min(npairs, 32 * (r0 + 1)) - 1, slo = 32 * r0;
    struct Ops { double l0[r0 + 1], l1[r0 + 1], p[3]; };
    // Operands of pivot pair s for the lane's columns: a pivot below the banded rows of a column's panel reads the two
    // zero rows kept behind them (slots inside the panel but outside the pivot row's own skyline were zero-filled and
    // never written).  Columns right of the pivot are solved: what they read (some other panel) lands in a t nobody
    // looks at again.  Two VALU instructions per column group: the substitution is bound by the instruction issue of
    // its single wave, and masks / a skyline look-up per step cost more than the step's arithmetic.
    auto fetch = [&](int s, Ops &o) {
      const int sc = max(s, 0);
      const bool lastrow = ((sc >> 1) == Tl);  // pivot rows inside the last row tile: every panel has them
#pragma unroll
      for (int r = 0; r <= r0; r++) {
        const int at = lastrow ? cl[r] + 4 * sc : cb[r] + 4 * min(sc, cap[r]);
        o.l0[r] = Cs[at];
        o.l1[r] = Cs[at + 2];
      }
      o.p[0] = pinv[4 * sc], o.p[1] = pinv[4 * sc + 1], o.p[2] = pinv[4 * sc + 2];
    };
    auto pin = [&](Ops &o) {  // keeps the prefetch where it was issued
#pragma unroll
      for (int r = 0; r <= r0; r++) asm volatile("" : "+v"(o.l0[r]), "+v"(o.l1[r]));
      asm volatile("" : "+v"(o.p[0]), "+v"(o.p[1]), "+v"(o.p[2]));
    };
    auto solve_step = [&](int s, const Ops &o) {
      const int l0 = (2 * s) & 63;
      const double t0 = bd_readlane(t[r0], l0), t1 = bd_readlane(t[r0], l0 + 1);
      const double x0 = fma(o.p[1], t1, o.p[0] * t0), x1 = fma(o.p[2], t1, o.p[1] * t0);
#pragma unroll
      for (int r = 0; r <= r0; r++) t[r] = fma(-o.l1[r], x1, fma(-o.l0[r], x0, t[r]));
      xo[r0] = (lane == l0) ? x0 : ((lane == l0 + 1) ? x1 : xo[r0]);
    };
    // several steps per LDS round trip (four while the register budget allows it)
    // (panels in global memory: one round trip per chunk, so the chunk is as long as the registers allow - each
    // step in flight holds 2 (r0 + 1) + 3 doubles)
    constexpr int CH = (GP && THREADS <= 512) ? (r0 == 0 ? 16 : r0 == 1 ? 12 : r0 == 2 ? 10 : r0 == 3 ? 8 : r0 == 4 ? 7 : 6)
                                             : (GP ? (r0 <= 3 ? 4 : 2) : ((r0 <= 2) ? 4 : 2));
    for (int s = shi; s >= slo; s -= CH) {
      Ops o[CH];
#pragma unroll
      for (int q = 0; q < CH; q++) fetch(s - q, o[q]);
#pragma unroll
      for (int q = 0; q < CH; q++) pin(o[q]);
#pragma unroll
      for (int q = 0; q < CH; q++)
        if (s - q >= slo) solve_step(s - q, o[q]);
    }
  };
  if (Rn > 5) sweep(std::integral_constant<int, 5>{});
  if (Rn > 4) sweep(std::integral_constant<int, 4>{});
  if (Rn > 3) sweep(std::integral_constant<int, 3>{});
  if (Rn > 2) sweep(std::integral_constant<int, 2>{});
  if (Rn > 1) sweep(std::integral_constant<int, 1>{});
  sweep(std::integral_constant<int, 0>{});

  // non-finite results count as failure too; failure => zero update (:1263-1266)
  bool bad = false;
#pragma unroll
  for (int r = 0; r < RMAX; r++)
    if (lane + 64 * r < n && !isfinite(xo[r])) bad = true;
  int failed = (*fail != 0) || (__ballot(bad) != 0ull);
  bool absent = false;
  if (SPLIT && split) {  // one verdict for both blocks
    int other = 0;
    if (lane == 0) {
      xflag[6 + role] = failed;
      __threadfence();
      __hip_atomic_store(xflag + 4 + role, generation(), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const long long t0 = wall_clock64();
      other = 2;  // (2 = the partner never reported: its block of dx is unknown)
      // (it has passed the exchange, so it is resident and running; with nothing queued behind, the wait is 30x longer)
      while (wall_clock64() - t0 < (nofb ? 3000000000ll : 100000000ll)) {
        if (__hip_atomic_load(xflag + 4 + (1 - role), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == generation()) {
          other = __hip_atomic_load(xflag + 6 + (1 - role), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
    }
    other = __builtin_amdgcn_readfirstlane(other);
    absent = (other == 2) && !nofb;  // (nofb: a partner that is still missing counts as a failed solve)
    failed |= (other != 0);
  }
#pragma unroll
  for (int r = 0; r < RMAX; r++) {
    const int j = lane + 64 * r;
    // the top workgroup stores its block and S, the bottom one its block only
    if (j < n && (role == 0 || j < nown)) dx[gidx(j)] = failed ? 0.f : (float)xo[r];
  }
  BPROF(4);
  if (lane == 0 && role == 0) {
    meta[1] = failed;
    // solved here: the general kernel queued behind this one returns at once -- unless the partner workgroup never
    // delivered its verdict (its block of dx is then stale): the system is left to the kernel queued behind
    meta[3] = absent ? 0 : 1;
    if (!absent) meta[7] = GP ? 2 : 1;  // which variant solved it (the adapter's per-graph solver plan, ba_host.hip)
  }
  BPROF(7);
}

bool ba_solve_band_supported(int n) { return n > 0 && !(n & 1) && n <= BD_MAX_N; }

#ifdef PROFILE_SOLVE
#define BD_PROF_ARG , g_band_prof
extern long long *g_band_prof;
#else
#define BD_PROF_ARG
#endif

// big = false: 1024 threads, one tile each, panels in LDS.  big = true: 512 threads, up to three tiles each, panels in
// `scratch` (scratch_doubles >= the packed lower triangle).  Either variant leaves meta[3] = 0 when the skyline
// does not fit it.
int launch_ba_solve_band(const double *H, const double *b, const int *fpose, int n, double lm, double ep, float *dx,
                         int *meta, double *scratch, size_t scratch_doubles, bool big, hipStream_t stream, bool last) {
  static DeviceOnce attr_once;
  if (attr_once.needed()) {
    DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_solve_band_kernel<BD_THREADS, 1, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SOLVE_MAX_LDS_BYTES));
    DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_solve_band_kernel<BD_BIG_THREADS, BD_BIG_SLOTS, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SOLVE_MAX_LDS_BYTES));
    DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_solve_band_kernel<BD_THREADS, 2, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SOLVE_MAX_LDS_BYTES));
    attr_once.done();
  }
  if (!big) {
    // two workgroups: the second one only works when the band lets the system be split (decided on the device).
    const int gcap = (int)(scratch_doubles > 0x7fffffff ? 0x7fffffff : scratch_doubles);
    static const bool one_wg = [] { const char *e = getenv("DBA_SOLVE_SPLIT"); return e && e[0] == '0'; }();
    hipLaunchKernelGGL((ba_solve_band_kernel<BD_THREADS, 1, false>), dim3((scratch && !one_wg) ? 2 : 1), dim3(BD_THREADS),
                       SOLVE_MAX_LDS_BYTES, stream, H, b, fpose, n, lm, ep, dx, meta, scratch, scratch ? gcap : 0,
                       last ? 1 : 0 BD_PROF_ARG);
  } else {
    if (!scratch) return DBA_ERR_WORKSPACE;
    const int gcap = (int)(scratch_doubles > 0x7fffffff ? 0x7fffffff : scratch_doubles);
    // two tiles per thread on 1024 threads (a step costs what one wave's tiles cost).  The older variant with three tiles
    // on 512 threads has fewer tile slots (1536 against 2048) and the same panel limits, so nothing reaches it that the
    // first one left: it is no longer queued (4.6 us per solve even when it returns at once), only kept for
    // DBA_SOLVE_BAND_BIG=512
    hipLaunchKernelGGL((ba_solve_band_kernel<BD_THREADS, 2, true>), dim3(1), dim3(BD_THREADS), SOLVE_MAX_LDS_BYTES, stream,
                       H, b, fpose, n, lm, ep, dx, meta, scratch, gcap, 0 BD_PROF_ARG);
    static const bool also512 = [] { const char *e = getenv("DBA_SOLVE_BAND_BIG"); return e && e[0] == '5'; }();
    if (also512)
      hipLaunchKernelGGL((ba_solve_band_kernel<BD_BIG_THREADS, BD_BIG_SLOTS, true>), dim3(1), dim3(BD_BIG_THREADS),
                         SOLVE_MAX_LDS_BYTES, stream, H, b, fpose, n, lm, ep, dx, meta, scratch, gcap, 0 BD_PROF_ARG);
  }
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

}  // namespace dba
