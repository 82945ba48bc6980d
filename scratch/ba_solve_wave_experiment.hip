// EXPERIMENT (round 5, not built into the library): correct on every case of scratch/solve_wave_test.hip (23 systems: bands
// of 0-7 poses, 1-64 poses, not positive definite, arrow / extra couplings, admission refusals) on the first run on the GPU --
// scratch/wave_solver_model.py is the kernel lane by lane in numpy and was held against dense solves before any device run --
// but SLOWER than the kernels that ship: n = 144: 49.5 us against 36.9 us (ba_solve_tile.hip); n = 378: 125 us against 101.5
// (ba_solve_band.hip).  A lone wave issues one instruction per ~5 cycles (LDS b64 ~10), and a 4-column step is ~270 of them:
// measured 2340 cycles per step = inverse of the pivot block 390 + six f64 matrix instructions 580 (on MI355X they run on the
// float64 vector pipe: 64 cycles each at best, no overlap with the VALU) + reads / extraction / operands / control 1370.
// That is 585 cycles per column -- the register-tile kernel's barrier steps cost 550.  See profiles/SOLVER_NOTES.md (round 5)
// for what would make it pay (two waves per front + a separator between two fronts: estimated 17 us / 37 us).
//
// Damped solve of the reduced camera system by ONE wavefront: a sliding window of the band in matrix-core accumulators.
//
// Replaces the host-side Eigen LLT / SimplicialLLT of the reference (/root/reference/src/droid_kernels.cu:200-218
// solveDenseD, :1248-1269 SparseBlock::solve) for the systems a sliding-window tracker produces: block-banded, 6 x 6 pose
// blocks, a handful of blocks wide.  ba_solve_tile.hip / ba_solve_band.hip / ba_solve.hip keep every other structure.
//
// The solve is ~0.1 MFLOP; what costs is the dependent chain (n pivots) and what hangs on every link of it.  The
// register-tile kernel (ba_solve_tile.hip) spreads the matrix over 11 waves and pays a workgroup barrier, an LDS hand-off
// and a poorly filled float64 pipe per two columns (0.47 us).  Here one wave owns the whole active part of the matrix:
//   * block LDL^T with 4 x 4 pivots; the trailing window -- the NT x NT lower tile triangle (16 x 16 tiles, NT = 3: 48 rows)
//     below / right of the pivot's tile column -- lives in matrix-core accumulators for the whole factorisation, and a
//     step's rank-4 update of it is one v_mfma_f64_16x16x4_f64 per tile: C -= R (W R^T) with R the raw panel, W the inverted
//     pivot block.  The instruction broadcasts its operands itself: no shuffles, no barrier, all 64 lanes busy;
//   * per step the next four columns leave the accumulators through LDS (the panel store, which is also what the
//     substitution reads later) and come back in the operand layouts;
//   * every lane inverts the pivot block, but lane group k (the k of the operand layouts) reads it with its indices XOR k,
//     so that ROW 0 of its inverse is row k of W: each lane computes only the row it needs and nothing is selected or
//     exchanged afterwards;
//   * the window slides: after the four steps of a tile column the tiles move up one place (register copies) and the
//     next tile row comes in from global memory, requested four steps (~1.5 us) before;
//   * the right-hand side rides along (z = W b1, b2 -= R z), the backward substitution runs right-looking: lane (slot,
//     k) accumulates v_s[k] = sum_i R_s[i][k] x[i] for the step s whose window still receives solved unknowns, so the
//     chain per step is four v_readlane + two short FMA chains, no reduction.
// Admission: the skyline (the prepare stage's pose-level table, made monotone) must stay inside the window in every step;
// the kernel tests that itself (ba_solve_wave_admits) and leaves the system to the other kernels otherwise.
// tests/wave_solver_model.py is this file lane by lane in numpy (pinned against a dense solve on the CPU).
#include "ba_kernels.h"

#include <type_traits>

namespace dba {
bool ba_solve_wave_supported(int n);
int launch_ba_solve_wave(const double *H, const double *b, const int *fpose, int n, double lm, double ep, float *dx, int *meta,
                         hipStream_t stream, long long *prof = nullptr);
}

namespace dba {

typedef double wv_d4 __attribute__((ext_vector_type(4)));
typedef double wv_d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double wv_rcp(double d) {  // v_rcp_f64 + one Newton step
  double y = __builtin_amdgcn_rcp(d);
  const double e = fma(-d, y, 1.0);
  return fma(y, e, y);
}

__device__ __forceinline__ double wv_readlane(double v, int l) {  // l wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// Row 0 of the inverse of the symmetric 4 x 4 block with lower triangle a b c / d e h / f g i j (rows 0..3), through
// 2 x 2 blocks: [A B^T; B C]^-1, S = C - B A^-1 B^T.  ok = positive definite (as far as the pivots of this order say).
__device__ __forceinline__ void wv_invert_row0(double a, double b, double c, double d, double e, double f, double g, double h,
                                               double i, double j, double (&w)[4], bool &ok) {
  const double detA = fma(-b, b, a * c);
  const bool okA = (a > 0.0) && (detA > 0.0);
  const double iA = okA ? wv_rcp(detA) : 0.0;
  const double a00 = c * iA, a01 = -b * iA, a11 = a * iA;
  const double x00 = fma(e, a01, d * a00), x01 = fma(e, a11, d * a01);  // X = B A^-1
  const double x10 = fma(g, a01, f * a00), x11 = fma(g, a11, f * a01);
  const double s00 = fma(-x01, e, fma(-x00, d, h));  // S = C - X B^T
  const double s01 = fma(-x01, g, fma(-x00, f, i));
  const double s11 = fma(-x11, g, fma(-x10, f, j));
  const double detS = fma(-s01, s01, s00 * s11);
  ok = okA && (s00 > 0.0) && (detS > 0.0);
  const double iS = ok ? wv_rcp(detS) : 0.0;
  const double t00 = s11 * iS, t01 = -s01 * iS, t11 = s00 * iS;  // S^-1
  const double y00 = fma(t01, x10, t00 * x00), y01 = fma(t01, x11, t00 * x01);  // Y = S^-1 X
  const double y10 = fma(t11, x10, t01 * x00), y11 = fma(t11, x11, t01 * x01);
  // a failed block contributes nothing (iA = iS = 0 makes everything below zero); the verdict is collected separately
  w[0] = fma(x10, y10, fma(x00, y00, a00));
  w[1] = fma(x10, y11, fma(x00, y01, a01));
  w[2] = -y00;
  w[3] = -y10;
}

// LDS, in doubles: panel store [S][16 NT][4] | z of every step [S][4] | right-hand side / solution [np + 64]
template <int NT>
__device__ __host__ __forceinline__ size_t wv_lds_doubles(int n) {
  const int np = (n + 15) & ~15, S = np >> 2;
  return (size_t)S * (16 * NT * 4 + 4) + np + 64;
}

// The kernel's admission test, one wave: with the pose-level skyline fpose (first pose a pose is coupled with) made
// monotone, every column's last row must lie inside the window of its step's tile column: row < 16 (s >> 2) + 16 NT.
// Returns the smallest NT in {3, 4} (<= max_nt: the panel store of NT = 4 does not fit LDS for the largest systems) that admits
// the system, or 0.
__device__ __forceinline__ int ba_solve_wave_admits(const int *__restrict__ fpose, int n, int lane, int max_nt) {
  const int P = n / 6;
  if (!fpose || P > 64 || n != 6 * P) return 0;
  int g = (lane < P) ? fpose[lane] : 0x7fffffff;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {  // suffix minimum: fill-in keeps the skyline monotone
    const int o = __shfl_down(g, off, 64);
    if (lane + off < 64) g = min(g, o);
  }
  int last = lane;  // last(q) = the last pose p with g[p] <= q (g is non-decreasing)
  for (int p = 0; p < P; p++) {
    const int gp = __builtin_amdgcn_readlane(g, p);
    if (gp <= lane) last = max(last, p);
  }
  const int np = (n + 15) & ~15, S = np >> 2;
  bool ok3 = true, ok4 = true;
  for (int base = 0; base < S; base += 64) {   // (uniform trip count: the shuffle below is executed by all lanes)
    const int s = base + lane, c = 4 * s;
    const int q3 = min(min(c + 3, n - 1) / 6, P - 1);
    const int lastrow = 6 * __shfl(last, q3, 64) + 5;
    const bool live = (s < S) && (c < n);
    ok3 = ok3 && (!live || lastrow <= 16 * (s >> 2) + 47);
    ok4 = ok4 && (!live || lastrow <= 16 * (s >> 2) + 63);
  }
  if (__ballot(!ok3) == 0ull) return 3;
  if (__ballot(!ok4) == 0ull && max_nt >= 4) return 4;
  return 0;
}

#ifdef PROFILE_SOLVE
#define WPROF(slot) do { if (lane == 0 && prof) { long long t_ = wall_clock64(); prof[slot] += t_ - tprev_; tprev_ = t_; } } while (0)
#else
#define WPROF(slot)
#endif

// compiler-only ordering of this wave's LDS traffic: the hardware executes one wave's LDS instructions in order, so a read
// issued after a write of another lane of the SAME wave sees it; what must not happen is the compiler moving one across the other
__device__ __forceinline__ void wv_order() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// hand-over between the two waves of the kernel: everything this wave wrote to LDS is complete, then the workgroup barrier
__device__ __forceinline__ void wv_handover() {
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_wave_barrier();
}

template <int NT>
struct WvLayout {   // LDS, in doubles
  static constexpr int PR = 16 * NT;   // rows of a step's panel store (the window of its tile column)
  static constexpr int PD = PR * 4;    // doubles per step
  int np, S;
  double *PAN, *ZST, *BV;
  int *flag;
  __device__ WvLayout(double *smem, int n) {
    np = (n + 15) & ~15, S = np >> 2;
    PAN = smem, ZST = PAN + (size_t)S * PD, BV = ZST + 4 * S, flag = (int *)(BV + np + 62);
  }
};

// ---- wave 0: the factorisation.  H: [n, n] float64 row-major, lower triangle read.
template <int NT>
__device__ void ba_solve_wave_factor(const double *__restrict__ H, int n, double lm, double ep, double *__restrict__ smem,
                                     int lane, long long *__restrict__ prof) {
#ifdef PROFILE_SOLVE
  long long tprev_ = wall_clock64();
#endif
  const WvLayout<NT> L(smem, n);
  constexpr int PD = WvLayout<NT>::PD;
  const int np = L.np, S = L.S;
  double *const PAN = L.PAN;
  const int li = lane & 15, lk = lane >> 4;

  // A tile of the damped, padded system in the accumulator layout: reg r <-> row 16 TI + lk + 4 r, column 16 TJ + li.
  // Two halves: the loads (always from inside the matrix, so there is no branch and no wait at the request) and, when the
  // tile is needed, the padding / damping -- the tile row a rotation brings in was requested four steps earlier.
  auto load_raw = [&](int TI, int TJ) {
    wv_d4 t;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = 16 * TI + lk + 4 * r, col = 16 * TJ + li;
      const int hi = min(max(row, col), n - 1), lo = min(min(row, col), n - 1);   // (diagonal tiles: mirrored upper half)
      t[r] = H[hi * n + lo];   // (n <= 384: the index fits 32 bits)
    }
    return t;
  };
  auto finish_tile = [&](wv_d4 t, int TI, int TJ) {
    if (16 * TI + 15 < n) {   // wave-uniform: a tile inside the system -- only a diagonal tile's diagonal changes
      if (TI == TJ) {
#pragma unroll
        for (int r = 0; r < 4; r++) t[r] = (lk + 4 * r == li) ? fma(lm, t[r], t[r]) + ep : t[r];   // damping (:1252-1253)
      }
      return t;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = 16 * TI + lk + 4 * r, col = 16 * TJ + li;
      const bool in = max(row, col) < n;
      const double hv = t[r];
      const double dv = in ? fma(lm, hv, hv) + ep : ((row < np) ? 1.0 : 0.0);   // identity padding up to np, zeros beyond
      t[r] = (row == col) ? dv : (in ? hv : 0.0);
    }
    return t;
  };

  wv_d4 acc[NT][NT], nxt[NT];
#pragma unroll
  for (int ti = 0; ti < NT; ti++)
#pragma unroll
    for (int tj = 0; tj <= ti; tj++) acc[ti][tj] = load_raw(ti, tj);
#pragma unroll
  for (int j = 0; j < NT; j++) nxt[j] = load_raw(NT, 1 + j);
#pragma unroll
  for (int ti = 0; ti < NT; ti++)
#pragma unroll
    for (int tj = 0; tj <= ti; tj++) acc[ti][tj] = finish_tile(acc[ti][tj], ti, tj);

  // tile row t of the columns of step sn out of tile column 0 of the window -> its panel store (all 16 rows: the rows above
  // the pivot are dead values nobody reads)
  auto extract = [&](int sn, int t, const wv_d4 &c) {
    if ((li >> 2) == (sn & 3)) {
      double *p = PAN + (size_t)sn * PD + (16 * t + lk) * 4 + (li & 3);
#pragma unroll
      for (int r = 0; r < 4; r++) p[16 * r] = c[r];
    }
  };
  // per-lane index of the pivot block's entry (i, j), i >= j, as lane group lk sees it (indices XOR lk, lower triangle)
  auto pidx = [&](int i, int j) {
    const int ii = i ^ lk, jj = j ^ lk;
    return max(ii, jj) * 4 + min(ii, jj);
  };
  const int px[10] = {pidx(0, 0), pidx(1, 0), pidx(1, 1), pidx(2, 0), pidx(2, 1), pidx(2, 2), pidx(3, 0), pidx(3, 1), pidx(3, 2), pidx(3, 3)};
  double pv[10];   // the pivot block of the coming step: a b c / d e h / f g i j
  auto read_pivot = [&](int sn) {
    const double *pp = PAN + (size_t)sn * PD + 16 * (sn & 3);
#pragma unroll
    for (int e = 0; e < 10; e++) pv[e] = pp[px[e]];
  };
  bool bad = false;
#pragma unroll
  for (int t = 0; t < NT; t++) extract(0, t, acc[t][0]);
  wv_order();
  read_pivot(0);
  WPROF(0);

  for (int s = 0; s < S; s++) {
    const int q = s & 3, tb = s >> 2, cl = 4 * q;
    double *const pan = PAN + (size_t)s * PD;
    // 1. this lane's panel rows (row 16 t + li of the window, columns XOR lk); the pivot block was read a step ahead
    double raw[NT][4];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
      for (int j = 0; j < 4; j++) raw[t][j] = pan[(16 * t + li) * 4 + (j ^ lk)];
    // 2. this lane group's row of the inverse: w[j] = W[lk][lk ^ j]
    double w[4];
    bool ok;
#ifdef WV_NO_INV
    w[0] = pv[0], w[1] = pv[1], w[2] = pv[3], w[3] = pv[6], ok = true;
#else
    wv_invert_row0(pv[0], pv[1], pv[2], pv[3], pv[4], pv[6], pv[7], pv[5], pv[8], pv[9], w, ok);
#endif
    bad |= !ok;
    {
      const bool live = li > cl + 3;  // rows of tile 0 at or above the pivot are eliminated: they take no part
#pragma unroll
      for (int j = 0; j < 4; j++) raw[0][j] = live ? raw[0][j] : 0.0;
    }
    // 3. operands: A = -R (row li, k = lk), B = W R^T (k = lk, column li); tile rows the panel does not reach are skipped
    double av[NT], uv[NT];
    bool need[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
      av[t] = -raw[t][0];
      uv[t] = fma(w[3], raw[t][3], fma(w[2], raw[t][2], fma(w[1], raw[t][1], w[0] * raw[t][0])));
      // (lane (li, lk) holds R[16 t + li][lk] in raw[t][0]: the ballot sees every entry of the tile row's panel)
      need[t] = (t == 0) || (__ballot(raw[t][0] != 0.0) != 0ull);
    }
    // 5. W takes the pivot block's place in the panel store (the substitution reads it there)
    wv_order();
    if (li == 0) {
#pragma unroll
      for (int j = 0; j < 4; j++) pan[(cl + lk) * 4 + (lk ^ j)] = w[j];
    }
    auto upd = [&](int ti, int tj) {
#ifdef WV_NO_MFMA
      if (need[ti] && need[tj]) acc[ti][tj][0] += av[ti] * uv[tj];
#else
      if (need[ti] && need[tj]) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ti], uv[tj], acc[ti][tj], 0, 0, 0);
#endif
    };
    const bool more = s + 1 < S;
    if (q != 3) {
      // 4. the tile column of the next panel first; its pivot block is requested as soon as tile 0 is out
#pragma unroll
      for (int t = 0; t < NT; t++) upd(t, 0);
      if (more) {
        extract(s + 1, 0, acc[0][0]);
        wv_order();
        read_pivot(s + 1);
#pragma unroll
        for (int t = 1; t < NT; t++) extract(s + 1, t, acc[t][0]);
      }
#pragma unroll
      for (int tj = 1; tj < NT; tj++)
#pragma unroll
        for (int ti = tj; ti < NT; ti++) upd(ti, tj);
    } else {
      // the window moves on by one tile column: tile column 0 is finished (its updates are skipped), the tiles move up one
      // place, the prefetched tile row comes in, the next one is requested
#pragma unroll
      for (int t = 1; t < NT; t++) upd(t, 1);
      if (more) {
        extract(s + 1, 0, acc[1][1]);
        wv_order();
        read_pivot(s + 1);
#pragma unroll
        for (int t = 2; t < NT; t++) extract(s + 1, t - 1, acc[t][1]);
      }
#pragma unroll
      for (int tj = 2; tj < NT; tj++)
#pragma unroll
        for (int ti = tj; ti < NT; ti++) upd(ti, tj);
#pragma unroll
      for (int ti = 0; ti + 1 < NT; ti++)
#pragma unroll
        for (int tj = 0; tj <= ti; tj++) acc[ti][tj] = acc[ti + 1][tj + 1];
#pragma unroll
      for (int j = 0; j < NT; j++) acc[NT - 1][j] = finish_tile(nxt[j], tb + NT, tb + 1 + j);
      if (more) extract(s + 1, NT - 1, acc[NT - 1][0]);
#pragma unroll
      for (int j = 0; j < NT; j++) nxt[j] = load_raw(tb + 1 + NT, tb + 2 + j);
    }
#ifdef WV_NO_HANDOVER
    wv_order();
#else
    wv_handover();   // step s is in the panel store (W, R): the substitution wave may take it
#endif
  }
  if (__ballot(bad) != 0ull && lane == 0) *L.flag = 1;
  wv_handover();
  WPROF(1);
}

// ---- wave 1: the right-hand side behind the factorisation (z = W b1, b2 -= R z, one step behind wave 0), then the backward
// substitution, right-looking: lane (slot, k) = (lane >> 2, lane & 3) accumulates v_s[k] = sum_i R_s[i][k] x[i] for the step
// s = slot (mod 16) that still receives solved unknowns (a window spans at most 4 NT <= 16 steps); x1 = z - W v.
template <int NT>
__device__ void ba_solve_wave_subst(const double *__restrict__ bvec, int n, float *__restrict__ dx, int *__restrict__ meta,
                                    double *__restrict__ smem, int lane, long long *__restrict__ prof) {
#ifdef PROFILE_SOLVE
  long long tprev_ = wall_clock64();
#endif
  const WvLayout<NT> L(smem, n);
  constexpr int PR = WvLayout<NT>::PR, PD = WvLayout<NT>::PD;
  const int np = L.np, S = L.S;
  double *const PAN = L.PAN, *const ZST = L.ZST, *const BV = L.BV;
  for (int i = lane; i < np + 62; i += 64) {
    const double bv = bvec[min(i, n - 1)];
    BV[i] = (i < n) ? bv : 0.0;
  }
  if (lane == 0) *L.flag = 0;
  for (int s = 0; s < S; s++) {
    const int tb = s >> 2, cl = 4 * (s & 3);
    const double *pan = PAN + (size_t)s * PD;
#ifndef WV_NO_HANDOVER
    wv_handover();
#endif
    double z[4];
    const double b0 = BV[4 * s], b1 = BV[4 * s + 1], b2 = BV[4 * s + 2], b3 = BV[4 * s + 3];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const double *wr = pan + (cl + k) * 4;
      z[k] = fma(wr[3], b3, fma(wr[2], b2, fma(wr[1], b1, wr[0] * b0)));
    }
    if (lane < PR && lane > cl + 3) {
      const double *rr = pan + lane * 4;
      double *bp = BV + 16 * tb + lane;
      *bp = fma(-rr[3], z[3], fma(-rr[2], z[2], fma(-rr[1], z[1], fma(-rr[0], z[0], *bp))));
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 4; k++) ZST[4 * s + k] = z[k];
    }
    wv_order();
  }
  wv_handover();
  const bool bad = *L.flag != 0;
  WPROF(4);
  {
    const int slot = lane >> 2, kk = lane & 3;
    double v = 0.0;
    struct Ops { double w[16], z[4], r[4]; bool valid; };
    auto fetch = [&](int sp, Ops &o) {   // everything step sp reads that does not hang on the chain
      const int spc = max(sp, 0);
      const double *pan = PAN + (size_t)spc * PD + 16 * (spc & 3);
#pragma unroll
      for (int e = 0; e < 16; e++) o.w[e] = pan[e];
#pragma unroll
      for (int k = 0; k < 4; k++) o.z[k] = ZST[4 * spc + k];
      const int dd = (spc - 1 - slot) & 15, sq = spc - 1 - dd;
      const int lrow = 4 * spc - 16 * (sq >> 2);
      o.valid = (sq >= 0) && (lrow + 3 <= PR - 1);
      const double *rp = PAN + (size_t)max(sq, 0) * PD + min(lrow, PR - 4) * 4 + kk;
#pragma unroll
      for (int m = 0; m < 4; m++) o.r[m] = rp[4 * m];
    };
    auto pin = [&](Ops &o) {   // keeps the prefetch where it was issued
#pragma unroll
      for (int e = 0; e < 16; e++) asm volatile("" : "+v"(o.w[e]));
#pragma unroll
      for (int k = 0; k < 4; k++) asm volatile("" : "+v"(o.z[k]), "+v"(o.r[k]));
    };
    auto solve_step = [&](int sp, const Ops &o) {
      const int l0 = 4 * (sp & 15);
      const double v0 = wv_readlane(v, l0), v1 = wv_readlane(v, l0 + 1), v2 = wv_readlane(v, l0 + 2), v3 = wv_readlane(v, l0 + 3);
      double x1[4];
#pragma unroll
      for (int k = 0; k < 4; k++)
        x1[k] = fma(-o.w[4 * k + 3], v3, fma(-o.w[4 * k + 2], v2, fma(-o.w[4 * k + 1], v1, fma(-o.w[4 * k], v0, o.z[k]))));
      if (lane < 4) BV[4 * sp + lane] = (lane == 0) ? x1[0] : (lane == 1) ? x1[1] : (lane == 2) ? x1[2] : x1[3];
      const double upd = fma(o.r[3], x1[3], fma(o.r[2], x1[2], fma(o.r[1], x1[1], o.r[0] * x1[0])));
      v = o.valid ? v + upd : v;
      v = (slot == (sp & 15)) ? 0.0 : v;
    };
    Ops oa, ob;
    fetch(S - 1, oa);
    for (int sp = S - 1; sp >= 0; sp -= 2) {   // two steps per trip, each one's operands requested a step ahead
      fetch(sp - 1, ob);
      pin(ob);
      solve_step(sp, oa);
      if (sp - 1 < 0) break;
      fetch(sp - 2, oa);
      pin(oa);
      solve_step(sp - 1, ob);
    }
  }
  wv_order();
  WPROF(5);
  // non-finite results count as failure too; failure => zero update (:1263-1266)
  bool nf = false;
  for (int j = lane; j < n; j += 64) nf |= !isfinite(BV[j]);
  const bool failed = bad || (__ballot(nf) != 0ull);
  for (int j = lane; j < n; j += 64) dx[j] = failed ? 0.f : (float)BV[j];
  if (lane == 0) meta[1] = failed ? 1 : 0;
  WPROF(6);
}

// the kernel: two waves.  Both run the admission test (no exchange needed to agree); meta[3] = 1 when the system was taken
// (a kernel queued behind with `skip_if_solved` then returns at once), 0 when it is left to that kernel
__global__ __launch_bounds__(128) void ba_solve_wave_kernel(const double *__restrict__ H, const double *__restrict__ bvec,
                                                            const int *__restrict__ fpose, int n, double lm, double ep,
                                                            float *__restrict__ dx, int *__restrict__ meta, int max_nt,
                                                            long long *__restrict__ prof) {
  extern __shared__ __attribute__((aligned(16))) double wv_smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nt = ba_solve_wave_admits(fpose, n, lane, max_nt);
  if (threadIdx.x == 0) meta[3] = (nt != 0) ? 1 : 0;
  if (nt == 3) {
    if (wave == 0) ba_solve_wave_factor<3>(H, n, lm, ep, wv_smem, lane, prof);
    else ba_solve_wave_subst<3>(bvec, n, dx, meta, wv_smem, lane, prof);
  } else if (nt == 4) {
    if (wave == 0) ba_solve_wave_factor<4>(H, n, lm, ep, wv_smem, lane, prof);
    else ba_solve_wave_subst<4>(bvec, n, dx, meta, wv_smem, lane, prof);
  }
}

static int wave_max_nt(int n) {
  if (n <= 0 || n % 6 != 0 || n / 6 > 64) return 0;
  if (wv_lds_doubles<4>(n) * sizeof(double) <= (size_t)SOLVE_MAX_LDS_BYTES) return 4;
  if (wv_lds_doubles<3>(n) * sizeof(double) <= (size_t)SOLVE_MAX_LDS_BYTES) return 3;
  return 0;
}

bool ba_solve_wave_supported(int n) { return wave_max_nt(n) != 0; }

int launch_ba_solve_wave(const double *H, const double *b, const int *fpose, int n, double lm, double ep, float *dx, int *meta,
                         hipStream_t stream, long long *prof) {
  static DeviceOnce attr_once;
  if (attr_once.needed()) {
    DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_solve_wave_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SOLVE_MAX_LDS_BYTES));
    attr_once.done();
  }
  const int max_nt = wave_max_nt(n);
  if (!max_nt) return DBA_ERR_UNSUPPORTED;
  const size_t lds = (max_nt == 4 ? wv_lds_doubles<4>(n) : wv_lds_doubles<3>(n)) * sizeof(double);
  hipLaunchKernelGGL(ba_solve_wave_kernel, dim3(1), dim3(128), lds, stream, H, b, fpose, n, lm, ep, dx, meta, max_nt, prof);
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

}  // namespace dba
