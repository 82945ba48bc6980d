// Damped dense solve of the reduced camera system for sliding-window sizes (n = 6P <= 174, i.e. up to 29
// optimised poses), float64, one workgroup, register-resident.
//
// Replaces the host-side Eigen LLT / SimplicialLLT of the reference
// (/root/reference/src/droid_kernels.cu:200-218 solveDenseD, :1248-1269 SparseBlock::solve) for the window
// sizes the tracker actually uses; ba_solve.hip keeps the general-size path.
//
// The solve is latency-bound (1 MFLOP): a lone wave issues roughly one instruction per 5-8 cycles (a float64 FMA
// every ~9.5), so a step of the elimination costs what the busiest wave's step body is long.  The design therefore
// minimises the number of dependent steps and the instructions per step instead of counting flops:
//   * the system, augmented with the right-hand side as an extra row, is cut into 4x4 tiles; every tile of the
//     lower triangle lives in the registers of ONE thread for the whole factorisation (703 threads at n = 144),
//     tiles in column-major order so that the tiles of a step sit in few waves;
//   * elimination is a block LDL^T with 2x2 pivots (one v_rcp_f64 + Newton per TWO columns, no square roots): per
//     step the owners of the pivot pair's panel have published their raw values to LDS (one ds_write_b128 per
//     row) and the owner of the diagonal tile the inverted pivot; ONE barrier; every tile inside the skyline reads
//     the panel entries of its rows and columns and applies the rank-2 update to its registers - first to the
//     elements the NEXT panel consists of, which are published before the rest of the tile is touched;
//   * banded systems (every sliding window) are eliminated from BOTH ends at once (a twisted factorisation): the
//     top front takes the column pairs 0, 1, ..., the bottom front the row pairs npairs-1, npairs-2, ..., one
//     pair each per barrier while their fill regions stay disjoint, then the top front finishes the middle:
//     72 -> 50 barriers for the 25-keyframe window.  Both fronts address their panel by global row / column
//     index, so one update body serves both; the step body is specialised per wave (top-only / bottom-only /
//     both) because every instruction of a body is paid for whether or not a lane needs it;
//   * the published panels are never rewritten and are exactly what the substitution needs (reverse elimination
//     order: middle, then the two fronts), which one wave runs with v_readlane broadcasts;
//   * structure: a row tile's first non-zero column tile (the skyline) is found once after the load; fill-in
//     cannot leave the (monotone) skyline, so tiles outside it never enter the update and waves without an
//     active tile only meet the barrier.
#include "ba_kernels.h"

#include <type_traits>

namespace dba {

constexpr int TILE_MAX_THREADS = 1024;

__device__ __forceinline__ double rcp_nr(double d) {
  double y = __builtin_amdgcn_rcp(d);
  double e = fma(-d, y, 1.0);
  y = fma(y, e, y);
  e = fma(-d, y, 1.0);
  y = fma(y, e, y);
  return y;
}

__device__ __forceinline__ double readlane_dyn_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

typedef double dbl2 __attribute__((ext_vector_type(2)));

// Panel store: column pair s keeps the rows 4*(s>>1) .. 4T-1 (whole tiles, so publishing a tile needs no row
// tests; rows past the right-hand-side row are zero padding), two doubles per row, plus two doubles of padding
// per pair: the back-substitution reads one matrix ROW across 64 column lanes, and without the skew the pair
// blocks sit a multiple of 16 banks apart (8-way conflicts); with it the row read is conflict-free.
// Doubles before pair s:
__device__ __host__ __forceinline__ int pair_off(int s, int T) {
  const int m = s >> 1;
  return 4 * (4 * T * m - 2 * m * (m - 1)) + ((s & 1) ? 2 * (4 * T - 4 * m) : 0) + 2 * s;
}

// Bottom-front row panels: n + 4 entries of two doubles, indexed by unknown (the right-hand side at n); the stride
// is 2 mod 32 doubles so that the substitution's reads of consecutive pairs fall into different banks.
__device__ __host__ __forceinline__ int bottom_stride(int n) {
  const int raw = 2 * (n + 4);
  return raw + ((34 - (raw & 31)) & 31);
}

// MAXT: launch bound.  Windows up to 25 optimised poses (n <= 150) need at most 768 threads, which leaves the
// compiler 168 registers per lane instead of 128 (no spills of the per-thread step constants).
template <int MAXT>
__global__ __launch_bounds__(MAXT) void ba_solve_tile_kernel(const double *__restrict__ H,
                                                                         const double *__restrict__ bvec,
                                                                         int n_in, int n, double lm, double ep,
                                                                         float *__restrict__ dx,
                                                                         int *__restrict__ meta, int cb_doubles
#ifdef PROFILE_SOLVE
                                                                         , long long *__restrict__ prof
#endif
                                                                         ) {
#ifdef PROFILE_SOLVE
#define TPROF(slot) do { if (threadIdx.x == 0) { long long t_ = wall_clock64(); prof[slot] += t_ - tprev_; tprev_ = t_; } } while (0)
  long long tprev_ = wall_clock64();
#else
#define TPROF(slot)
#endif
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // n_in = size of the system; n = n_in, or n_in + 2 when the host pads a system with n_in % 4 == 2 by two decoupled
  // unknowns (unit diagonal, zero right-hand side, solution 0) so that it qualifies for the two-front elimination
  const int n1 = n + 1;                 // rows incl. the right-hand side
  const int T = (n1 + 3) >> 2;          // row tiles
  const int KT = (n + 3) >> 2;          // column tiles
  const int npairs = n >> 1;
  const int PBS = bottom_stride(n);     // doubles per bottom-front panel
  double *C = smem;                                   // raw panels of the top front, pair-major
  double *pinv = C + pair_off(npairs, T);           // 4 doubles per pair (p00, p01, p11, -)
  int *first = (int *)(pinv + 4 * npairs);            // skyline: first non-zero column tile of a row tile
  int *colmax = first + T;                            // last row tile whose (monotone) skyline reaches a column tile
  int *fail = colmax + T;                             // [0] failure, [1] c1
  double *Cb = (double *)(fail + 4 + 2 * (T & 1));     // row panels of the bottom front (slot = step), 16-byte aligned
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;

  // ---- tile of this thread: COLUMN-major over the lower triangle of tiles, so the tiles still being updated at
  // step Ks (columns >= Ks) are a suffix of the thread ids: the waves of the eliminated columns skip the step body
  // and the live lanes stay packed
  const int ntile = T * (T + 1) / 2;
  const int rev = max(ntile - 1 - tid, 0);  // reversed, the order is row-major over a triangle of T - 1 - K
  int Kr = (int)((sqrtf(8.f * rev + 1.f) - 1.f) * 0.5f);
  while ((Kr + 1) * (Kr + 2) / 2 <= rev) Kr++;
  while (Kr * (Kr + 1) / 2 > rev) Kr--;
  const int K = T - 1 - Kr;
  const int I = (tid < ntile) ? T - 1 - (rev - Kr * (Kr + 1) / 2) : T;
  const bool valid = I < T && 4 * K < n;
  const bool rhs = (I == T - 1);  // the tile row that carries the right-hand side

  if (tid < T) first[tid] = min(tid, KT - 1), colmax[tid] = 0;
  if (tid < 4) fail[tid] = 0;

  // zero the panel stores (padding rows are read as operands) while the tile loads are in flight
  double a[4][4];
  const double *src[4][4];
  bool okm[4][4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int i = 4 * I + r, k = 4 * K + c;
      okm[r][c] = valid && k < n_in && (i < n_in || i == n);   // (padding unknowns: zero, unit diagonal below)
      const int hi = max(i, k), lo = min(i, k);   // diagonal tiles keep the mirrored upper half
      const double *q = (i == n) ? bvec + k : H + (size_t)hi * n_in + lo;
      src[r][c] = okm[r][c] ? q : H;
    }
  }
  // Off-diagonal tiles whose four columns exist read each row as two 16-byte loads (the tile order puts the lanes of
  // a wave in different cache lines, so the load phase is bound by the number of requests: 8 instead of 16 per
  // tile); diagonal tiles (mirrored upper half: a gather) and ragged ones keep the scalar loads.
  const bool vec = valid && I != K && 4 * K + 3 < n_in && ((((size_t)H | (size_t)bvec) & 15) == 0);
  if (vec) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = 4 * I + r;
      const double *rowp = (i == n) ? bvec + 4 * K : H + (size_t)min(i, n_in - 1) * n_in + 4 * K;
      const dbl2 v0 = *(const dbl2 *)rowp, v1 = *(const dbl2 *)(rowp + 2);
      a[r][0] = v0.x, a[r][1] = v0.y, a[r][2] = v1.x, a[r][3] = v1.y;  // (rows past the system are masked by okm below)
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int c = 0; c < 4; c++) a[r][c] = *src[r][c];
  }
  // two fronts need n % 4 == 0 (the right-hand side alone in its tile row) and room for the row panels
  const int c1cap = ((n & 3) == 0) ? min((KT - 1) >> 1, cb_doubles / (2 * PBS)) : 0;
  {
    dbl2 z;
    z.x = z.y = 0.0;
    for (int e = tid; 2 * e < pair_off(npairs, T); e += blockDim.x) *(dbl2 *)(C + 2 * e) = z;
    for (int e = tid; e < c1cap * PBS; e += blockDim.x) *(dbl2 *)(Cb + 2 * e) = z;
  }
  bool nz = false;
#pragma unroll
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      double v = okm[r][c] ? a[r][c] : 0.0;
      if (4 * I + r == 4 * K + c && 4 * I + r < n) v = (4 * I + r < n_in) ? v + (ep + lm * v) : 1.0;  // damping (:1252-1253)
      a[r][c] = v;
      nz |= (v != 0.0);
    }
  }
  __syncthreads();
  if (valid && nz) atomicMin(&first[I], K);
  __syncthreads();

  // ---- two fronts.  A banded system is eliminated from both ends at once: the top front takes the column pairs
  // 0, 1, ... as before, the bottom front the pairs npairs-1, npairs-2, ... (a tile ROW's two halves, bottom up),
  // one pair each per barrier, for the first c1 tile columns of either end; then the top front finishes the
  // middle.  c1 is the largest count for which the fronts never touch the same tile: with f' the skyline made
  // monotone (suffix minimum: bottom-up elimination fills a row to the left as far as any row below it reaches)
  // and colmax[K] = last row tile with f' <= K, the top front at tile column c works in rows <= colmax[c], the
  // bottom front at tile row KT-1-c in columns >= f'[KT-1-c].  Dense or arrow-shaped systems give c1 = 0.
  if (wave == 0) {
    int c1 = 0;  // (shadows nothing: the kernel-wide c1 is read back from LDS below)
    if (c1cap >= 2) {
      int fp = (lane < KT) ? first[lane] : 0x7fffffff;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_down(fp, off, 64);
        if (lane + off < 64) fp = min(fp, o);
      }
      // colmax[K] = last row with f' <= K: scatter each row to its first column, then a running maximum
      if (lane < KT) atomicMax(&colmax[fp], lane);
      int cm = (lane < KT) ? colmax[lane] : 0;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(cm, off, 64);
        if (lane >= off) cm = max(cm, o);
      }
      const int fo = __shfl(fp, max(KT - 1 - lane, 0), 64);
      const unsigned long long okb = __ballot(lane < c1cap && cm < fo);
      c1 = (int)__builtin_ctzll(~okb);
      if (c1 < 2) c1 = 0;
      if (c1 > 0) {
        if (lane < KT) first[lane] = fp, colmax[lane] = cm;
        const int fb = __shfl(fp, KT - c1, 64);  // leftmost column the bottom front writes in the right-hand side row
        if (lane == 0) first[T - 1] = min(first[T - 1], fb);
      }
    }
    if (lane == 0) fail[1] = c1;
  }
  __syncthreads();
  const int c1 = __builtin_amdgcn_readfirstlane(fail[1]);  // wave-uniform: keeps the loop control on the scalar unit
  const int sstart = valid ? max(first[I], first[min(K, T - 1)]) : 0x3fffffff;
  const int cmK = (c1 > 0 && valid) ? colmax[K] : -1;           // the bottom front reaches this tile from row cmK up
  const bool frozen = c1 > 0 && !rhs && I >= KT - c1;          // rows the bottom front eliminates
  const int nsteps = npairs - 2 * c1;                           // barriers: pairs of the top front
  // Steps at which this tile is updated, as windows of the step counter s (one subtract + one unsigned compare per
  // front and step; everything a step needs per thread is a constant from here on: the lone waves of the dependent
  // chain pay ~5 cycles per instruction, so the step body is kept branch-free and short).
  //   top front, pair s = tile column Ks = s >> 1, half h = s & 1:  Ks >= sstart and (K > Ks or (K == Ks and h == 0))
  //   bottom front, pair sb = npairs - 1 - s = tile row Kb, half hb:  Kb <= cmK, (K < Kb or (K == Kb and hb == 1)),
  //                 (I < Kb or (I == Kb and hb == 1) or right-hand side), s < 2 c1
  int t_lo = 0x40000000, t_len = 0, b_lo = 0x40000000, b_len = 0;
  if (valid && !frozen && sstart <= K) t_lo = 2 * sstart, t_len = 2 * K - 2 * sstart;
  {
    const int sb_lo = max(max(2 * K + 1, npairs - 2 * c1), rhs ? 0 : 2 * I + 1), sb_hi = 2 * cmK + 1;
    if (valid && c1 > 0 && sb_lo <= sb_hi) b_lo = npairs - 1 - sb_hi, b_len = sb_hi - sb_lo;
  }
  const int offI = 64 * I, offK = 64 * K;  // bytes of 4 panel entries (two doubles each): operands by row / column index
  int bad = 0;  // a non-positive pivot seen by this thread (collected after the loop; keeps the step branch-free)
  TPROF(0);

  // The owner of a diagonal tile inverts the NEXT 2x2 pivot right after updating it and publishes the inverse with
  // the panel: the readers get P^-1 with one LDS read instead of each redoing the reciprocal, and the
  // back-substitution finds it in place.
  auto publish_pinv = [&](int sp, double pa, double pb, double pc) {
    const double det = fma(-pb, pb, pa * pc);
    const bool ok = pa > 0.0 && det > 0.0;
    bad |= !ok;  // the verdict is collected after the substitution; keep things finite
    const double idet = ok ? rcp_nr(det) : 0.0;
    dbl2 lo;
    lo.x = pc * idet;
    lo.y = -pb * idet;
    *(dbl2 *)(pinv + 4 * sp) = lo;
    pinv[4 * sp + 2] = pa * idet;
  };
  // Panels are addressed in bytes from C: entry i of a panel sits at base + 16 i, so a tile's four row (column)
  // operands are 64 contiguous bytes at base + offI (offK).  The bases only depend on the step and are carried as
  // running scalars (the compiler does not strength-reduce them out of the specialised loops by itself):
  //   top pair s:     tb(s) = 8 pair_off(s, T) - 64 (s >> 1),  tb(s + 1) - tb(s) = 64 (T - (s >> 1)) + 16 - 64 (s & 1)
  //   bottom slot s:  bb(s) = (Cb - C) + 8 PBS s
  char *const Cc = (char *)C;
  const int bb0 = (int)((char *)Cb - Cc), pv0 = (int)((char *)pinv - Cc);
  // column panel of the top front (pair sp, half hp of its tile column) at base tbn, from the tiles of that column
  auto publish_top = [&](int tbn, int sp, auto hc) {
    constexpr int hp = decltype(hc)::value;
    dbl2 *const q = (dbl2 *)(Cc + tbn + offI);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      dbl2 v;
      v.x = a[r][2 * hp];
      v.y = a[r][2 * hp + 1];
      q[r] = v;
    }
    if (I == K) publish_pinv(sp, a[2 * hp][2 * hp], a[2 * hp + 1][2 * hp], a[2 * hp + 1][2 * hp + 1]);
  };
  // row panel of the bottom front at base bbn: the pivot rows (half hp of their tile row) over all columns, indexed
  // by column; the diagonal tile only contributes the columns left of the pivot (the rest stays zero) and the pivot
  auto publish_bottom = [&](int bbn, int sp, auto hc) {
    constexpr int hp = decltype(hc)::value;
    dbl2 *const q = (dbl2 *)(Cc + bbn + offK);
    if (I != K) {
#pragma unroll
      for (int c = 0; c < 4; c++) {
        dbl2 v;
        v.x = a[2 * hp][c];
        v.y = a[2 * hp + 1][c];
        q[c] = v;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 2 * hp; c++) {
        dbl2 v;
        v.x = a[2 * hp][c];
        v.y = a[2 * hp + 1][c];
        q[c] = v;
      }
      publish_pinv(sp, a[2 * hp][2 * hp], a[2 * hp + 1][2 * hp], a[2 * hp + 1][2 * hp + 1]);
    }
  };
  // ... and the right-hand side's two entries at index n
  auto publish_bottom_rhs = [&](int bbn, auto hc) {
    constexpr int hp = decltype(hc)::value;
    dbl2 v;
    v.x = a[0][2 * hp];
    v.y = a[0][2 * hp + 1];
    *(dbl2 *)(Cc + bbn + 16 * n) = v;
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  if (valid && K == 0 && !frozen) publish_top(0, 0, H0{});
  if (c1 > 0 && valid && I == KT - 1) publish_bottom(bb0, npairs - 1, H1{});
  if (c1 > 0 && valid && rhs && K == KT - 1) publish_bottom_rhs(bb0, H1{});

  // ---- factorisation: block LDL^T, 2x2 pivots, one barrier per step.  A step first updates the elements the NEXT
  // step's panels consist of (two columns for the top front, two rows for the bottom front) and publishes them
  // (panel + inverted pivot) before touching the rest, so the LDS round trip of the hand-off runs under the
  // remaining FMAs.  Panels are per pair: no write-after-read hazard.  Both fronts read their operands by global
  // row / column index from the pair's panel, so one update body serves both.
  // A lone wave issues about one instruction per 5-8 cycles, so a step costs what the busiest wave's step body is
  // long: the body is specialised per WAVE.  MODE 0: top front only (also the middle and one-front systems),
  // 1: bottom front only, 2: both (a wave whose tiles straddle the two regions).  (Tried: one straight-line region
  // per step in which lanes that do not publish store into a dump block and every lane inverts its own pivot, so
  // that the reciprocal chain can overlap the remaining updates - slower, every busy wave then pays for it.)
  int tb = 0, bb = bb0;  // bases of the current step's panels
  auto step = [&](int Ks, auto hc, auto mode) {
    constexpr int h = decltype(hc)::value;
    constexpr int MODE = decltype(mode)::value;
    constexpr bool TOP = (MODE != 1), BOT = (MODE != 0);
    constexpr int hn = 1 - h;
    using HN = std::integral_constant<int, hn>;
    const int s = 2 * Ks + h;
    const int Kn = (h == 0) ? Ks : Ks + 1;
    // bottom front (n % 4 == 0: npairs is even, so its half is the opposite one)
    const int sb = npairs - 1 - s;
    const int Kbn = (sb - 1) >> 1;
    const int tbn = tb + 64 * (T - Ks) + 16 - 64 * h, bbn = bb + 8 * PBS;
    __syncthreads();
    const bool topA = TOP && (unsigned)(s - t_lo) <= (unsigned)t_len;
    const bool botA = BOT && (unsigned)(s - b_lo) <= (unsigned)b_len;
    dbl2 ri[4];
    double u0[4], u1[4];
    auto upd = [&](int r, int c) { a[r][c] = fma(-ri[r].y, u1[c], fma(-ri[r].x, u0[c], a[r][c])); };
    // the tile in four quarters: the next top pivot's columns are (2 hn, 2 hn + 1), the next bottom pivot's rows
    // (2 h, 2 h + 1).  quarter(rh, ch) = rows 2 rh.., columns 2 ch..
    auto quarter = [&](int rh, int ch) {
      upd(2 * rh, 2 * ch);
      upd(2 * rh, 2 * ch + 1);
      upd(2 * rh + 1, 2 * ch);
      upd(2 * rh + 1, 2 * ch + 1);
    };
    if (topA | botA) {
      const bool useb = BOT && (!TOP || botA);
      const char *const base = Cc + (useb ? bb : tb);
      const double *const pvp = (const double *)(Cc + pv0 + 32 * (useb ? sb : s));
      const dbl2 pv = *(const dbl2 *)pvp;
      const double p00 = pv.x, p01 = pv.y, p11 = pvp[2];
      const dbl2 *const pi = (const dbl2 *)(base + offI), *const pk = (const dbl2 *)(base + offK);
      dbl2 rk[4];
#pragma unroll
      for (int r = 0; r < 4; r++) ri[r] = pi[r];
#pragma unroll
      for (int c = 0; c < 4; c++) rk[c] = pk[c];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        u0[c] = fma(p01, rk[c].y, p00 * rk[c].x);
        u1[c] = fma(p11, rk[c].y, p01 * rk[c].x);
      }
      // first what the next step's panels consist of (one static order per mode: a per-lane order would make the
      // compiler address the tile through scratch)
      if (TOP) quarter(h, hn), quarter(hn, hn);
      if (BOT) quarter(h, h);
      if (BOT && !TOP) quarter(h, hn);
    }
    // published whether or not this step touched the tile (block-diagonal systems); the bottom front's row panel
    // only inside the skyline (the row's tiles sit in every column; the rest stays zero)
    if (TOP) {
      if (valid & (K == Kn) & !frozen & (s + 1 < nsteps)) publish_top(tbn, s + 1, HN{});
    }
    if (BOT) {
      if (valid & (I == Kbn) & (Kbn <= cmK) & (s + 1 < 2 * c1)) publish_bottom(bbn, sb - 1, hc);
    }
    if (topA | botA) {
      // (the elements are pinned behind the publish: left alone, the compiler hoists these FMAs over the stores
      // and the hand-off waits for the whole tile again)
      auto pin_quarter = [&](int rh, int ch) {
        asm volatile("" : "+v"(a[2 * rh][2 * ch]), "+v"(a[2 * rh][2 * ch + 1]), "+v"(a[2 * rh + 1][2 * ch]),
                     "+v"(a[2 * rh + 1][2 * ch + 1]));
      };
      if (!BOT) pin_quarter(h, h), pin_quarter(hn, h);
      if (BOT && TOP) pin_quarter(hn, h);
      if (BOT && !TOP) pin_quarter(hn, h), pin_quarter(hn, hn);
      if (!BOT) quarter(h, h);
      if (TOP) quarter(hn, h);
      if (BOT && !TOP) quarter(hn, h), quarter(hn, hn);
    }
    // (the right-hand side's entries sit in row 0 of its tile, which is complete only now)
    if (BOT) {
      if (valid & rhs & (K == Kbn) & (s + 1 < 2 * c1)) publish_bottom_rhs(bbn, hc);
    }
    tb = tbn, bb = bbn;
  };
  {
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>;
    // which fronts this wave's tiles ever serve while both are running (columns up to c1 publish top panels)
    const bool my_top = valid && !frozen && (K <= c1 || t_lo < 2 * c1);
    const bool my_bot = b_lo != 0x40000000;
    const int wmode = (__ballot(my_bot) != 0ull) ? ((__ballot(my_top) != 0ull) ? 2 : 1) : 0;
    int Ks = 0;
    if (wmode == 2) {
      for (; Ks < c1; Ks++) step(Ks, H0{}, M2{}), step(Ks, H1{}, M2{});
    } else if (wmode == 1) {
      for (; Ks < c1; Ks++) step(Ks, H0{}, M1{}), step(Ks, H1{}, M1{});
    } else {
      for (; Ks < c1; Ks++) step(Ks, H0{}, M0{}), step(Ks, H1{}, M0{});
    }
    for (; 2 * Ks < nsteps; Ks++) {
      step(Ks, H0{}, M0{});
      if (2 * Ks + 1 < nsteps) step(Ks, H1{}, M0{});
    }
  }
  if (bad) *fail = 1;
  __syncthreads();
  TPROF(1);
  TPROF(2);

  // ---- substitution, one wave.  t_j = y_j - sum over the solved unknowns i of coef(i, j) x_i, where coef(i, j) is
  // entry i of the panel of j's pair (a column panel for the top front / middle, a row panel for the bottom
  // front): lane l keeps t_j for the columns j = l, l + 64, l + 128 and reads its panels by unknown index.  Order =
  // reverse elimination: the middle downwards, then the top front's pairs downwards and the bottom front's upwards.
  // With two fronts the substitution's two chains run on two waves (different SIMDs): both walk the middle, then
  // wave 0 takes the top front's pairs and wave 1 the bottom front's (a lone wave is issue-bound here as well, so
  // interleaving the chains in one wave buys nothing - measured).
  const bool two_waves = c1 > 0 && blockDim.x >= 128;  // (tiny systems have a single wave: it walks both chains)
  if (wave == 0 || (wave == 1 && two_waves)) {
    double t[3];
    int cbase[3];  // coef(i, j) = C[cbase + 2 i] for this lane's column j (columns past n alias column 0; masked)
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const int j = lane + 64 * r;
      const int jc = (j < n) ? j : 0;
      const int sj = jc >> 1;
      cbase[r] = (sj >= npairs - 2 * c1) ? (int)(Cb - C) + (npairs - 1 - sj) * PBS + (jc & 1)
                                         : pair_off(sj, T) - 8 * (jc >> 2) + (jc & 1);
      t[r] = (j < n) ? C[cbase[r] + 2 * n] : 0.0;
    }
    double xo[3] = {0.0, 0.0, 0.0};  // solution entries of this lane's columns (kept off the dependent chain)
    // pairs s_from, s_from + DIR, ..., s_to of column group r0; the t registers RLO..RHI receive the update
    auto sweep = [&](auto rc, auto rlo, auto rhi, auto dir, int s_from, int s_to) {
      constexpr int r0 = decltype(rc)::value, RLO = decltype(rlo)::value, RHI = decltype(rhi)::value;
      constexpr int DIR = decltype(dir)::value;
      constexpr int R = RHI - RLO + 1;
      struct Ops { double l0[R], l1[R], p[3]; };
      // raw operands of step s: entries 2s, 2s+1 of this lane's panels and the pivot inverse
      auto fetch = [&](int s, Ops &o) {
        const int sc = min(max(s, 0), npairs - 1);
#pragma unroll
        for (int r = 0; r < R; r++) {
          const double *q = C + cbase[RLO + r] + 4 * sc;
          o.l0[r] = q[0];
          o.l1[r] = q[2];
        }
        o.p[0] = pinv[4 * sc], o.p[1] = pinv[4 * sc + 1], o.p[2] = pinv[4 * sc + 2];
      };
      auto solve_step = [&](int s, const Ops &o) {
        const int j0 = 2 * s, l0 = j0 & 63;
        const double t0 = readlane_dyn_f64(t[r0], l0), t1 = readlane_dyn_f64(t[r0], l0 + 1);
        const double x0 = fma(o.p[1], t1, o.p[0] * t0), x1 = fma(o.p[2], t1, o.p[1] * t0);
        // every lane updates: finished columns receive garbage, but their solution sits in xo and they are never
        // read again, so no per-lane masking on the single issuing wave
#pragma unroll
        for (int r = 0; r < R; r++) t[RLO + r] = fma(-o.l1[r], x1, fma(-o.l0[r], x0, t[RLO + r]));
        xo[r0] = (lane == l0) ? x0 : ((lane == l0 + 1) ? x1 : xo[r0]);
      };
      // keeps a prefetch where it was issued (otherwise the loads sink to their first use)
      auto pin = [&](Ops &o) {
#pragma unroll
        for (int r = 0; r < R; r++) asm volatile("" : "+v"(o.l0[r]), "+v"(o.l1[r]));
        asm volatile("" : "+v"(o.p[0]), "+v"(o.p[1]), "+v"(o.p[2]));
      };
      // operands are fetched four steps at a time: one LDS round trip per four links of the dependent chain
      for (int s = s_from; DIR > 0 ? s <= s_to : s >= s_to; s += 4 * DIR) {
        Ops o[4];
#pragma unroll
        for (int q = 0; q < 4; q++) fetch(s + q * DIR, o[q]);
#pragma unroll
        for (int q = 0; q < 4; q++) pin(o[q]);
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (DIR > 0 ? s + q <= s_to : s - q >= s_to) solve_step(s + q * DIR, o[q]);
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using IM = std::integral_constant<int, -1>;
    // the part of [lo, hi] in column group r0 (32 pairs per group), downwards or upwards
    auto down = [&](auto rc, auto rhi, int lo, int hi) {
      constexpr int r0 = decltype(rc)::value;
      const int a_ = min(hi, 32 * r0 + 31), b_ = max(lo, 32 * r0);
      if (a_ >= b_) sweep(rc, I0{}, rhi, IM{}, a_, b_);
    };
    auto up = [&](auto rc, int lo, int hi) {
      constexpr int r0 = decltype(rc)::value;
      const int a_ = max(lo, 32 * r0), b_ = min(hi, 32 * r0 + 31);
      if (a_ <= b_) sweep(rc, rc, I2{}, I1{}, a_, b_);
    };
    const int mlo = 2 * c1, mhi = npairs - 1 - 2 * c1;
    if (c1 > 0) {  // the bottom front's columns (in the upper groups) also take the middle's updates
      down(I2{}, I2{}, mlo, mhi);
      down(I1{}, I2{}, mlo, mhi);
      down(I0{}, I2{}, mlo, mhi);
      if (wave == 0) {
        down(I2{}, I2{}, 0, mlo - 1);
        down(I1{}, I1{}, 0, mlo - 1);
        down(I0{}, I0{}, 0, mlo - 1);
      }
      if (wave == 1 || !two_waves) {
        up(I0{}, mhi + 1, npairs - 1);
        up(I1{}, mhi + 1, npairs - 1);
        up(I2{}, mhi + 1, npairs - 1);
      }
    } else {
      down(I2{}, I2{}, 0, npairs - 1);
      down(I1{}, I1{}, 0, npairs - 1);
      down(I0{}, I0{}, 0, npairs - 1);
    }

    // non-finite results count as failure too; failure => zero update (:1263-1266).  Wave 0 owns the columns of the
    // middle and the top front, wave 1 those of the bottom front
    const int jsplit = two_waves ? n - 4 * c1 : n;
    bool bad = false;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const int j = lane + 64 * r;
      if (j < n && (wave == 0) == (j < jsplit) && !isfinite(xo[r])) bad = true;
    }
    if (__ballot(bad) != 0ull && lane == 0) *fail = 1;
    if (two_waves) __syncthreads();  // (only these two waves are still alive: the others have left the kernel)
    const int failed = (*fail != 0);
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const int j = lane + 64 * r;
      if (j < n_in && (wave == 0) == (j < jsplit)) dx[j] = failed ? 0.f : (float)xo[r];
    }
    if (tid == 0) meta[1] = failed;
  }
  TPROF(3);
}

static int tile_rows(int n) { return (n + 1 + 3) / 4; }

static size_t tile_lds_bytes(int n) {
  const int np = n / 2;
  // panels + pivot inverses, then first[T] colmax[T] flags[4] (+ 2 ints so that what follows is 16-byte aligned)
  return ((size_t)pair_off(np, tile_rows(n)) + 4 * (size_t)np) * sizeof(double) + (2 * (size_t)tile_rows(n) + 4 + 2) * sizeof(int);
}

bool ba_solve_tile_supported(int n) {
  if (n <= 0 || (n & 1) || n > 192) return false;
  const int T = tile_rows(n);
  return T * (T + 1) / 2 <= TILE_MAX_THREADS && tile_lds_bytes(n) <= (size_t)SOLVE_MAX_LDS_BYTES;
}

int launch_ba_solve_tile(const double *H, const double *b, int n, double lm, double ep, float *dx, int *meta,
                         hipStream_t stream) {
  static DeviceOnce attr_once;
  if (attr_once.needed()) {
    DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_solve_tile_kernel<768>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SOLVE_MAX_LDS_BYTES));
    DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_solve_tile_kernel<TILE_MAX_THREADS>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SOLVE_MAX_LDS_BYTES));
    attr_once.done();
  }
  // what is left of the LDS holds the bottom front's row panels (DBA_SOLVE_TWIST=0: one front only)
  static const bool twist = [] { const char *e = getenv("DBA_SOLVE_TWIST"); return !(e && e[0] == '0'); }();
  // n % 4 == 2 (an odd number of poses): two padding unknowns make the system eligible for two fronts, as long as
  // the padded system still fits the 768-thread variant
  int nw = n;
  if (twist && (n & 3) == 2) {
    const int Tp = tile_rows(n + 2);
    if (Tp * (Tp + 1) / 2 <= 768 && tile_lds_bytes(n + 2) <= (size_t)SOLVE_MAX_LDS_BYTES) nw = n + 2;
  }
  const int T = tile_rows(nw);
  const int threads = ((T * (T + 1) / 2 + 63) / 64) * 64;
  const int cb_doubles = twist ? (int)((SOLVE_MAX_LDS_BYTES - tile_lds_bytes(nw)) / sizeof(double)) : 0;
#ifdef PROFILE_SOLVE
  extern long long *g_tile_prof;
#define TILE_PROF_ARG , g_tile_prof
#else
#define TILE_PROF_ARG
#endif
  if (threads <= 768)
    hipLaunchKernelGGL(ba_solve_tile_kernel<768>, dim3(1), dim3(threads), SOLVE_MAX_LDS_BYTES, stream, H, b, n, nw, lm,
                       ep, dx, meta, cb_doubles TILE_PROF_ARG);
  else
    hipLaunchKernelGGL(ba_solve_tile_kernel<TILE_MAX_THREADS>, dim3(1), dim3(threads), SOLVE_MAX_LDS_BYTES, stream, H,
                       b, n, nw, lm, ep, dx, meta, cb_doubles TILE_PROF_ARG);
#undef TILE_PROF_ARG
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

}  // namespace dba
