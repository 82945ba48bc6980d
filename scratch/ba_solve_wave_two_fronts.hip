// SHELVED EXPERIMENT (round 5) -- not part of the product; see profiles/SOLVER_NOTES.md "two fronts".
// csrc/ba_solve_wave.hip generalised to INSTANCES (WvInst): the system cut into top | separator | bottom, the top part eliminated
// forwards by waves 0..2, the bottom part backwards (the same code on the mirrored matrix) by waves 3..5, what both leave of the
// separator block added up into the separator's own system, one substitution wave serving both fronts, one loader wave.
// Correct on every case of scratch/solve_wave_test.hip (-DTWO_FRONTS) and modelled lane by lane in tests/wave_solver_model.py
// (TwoFrontSolver / split_plan), but SLOWER than one front: n = 144: 37.1 us against 29.8.  The fronts' chains do halve
// (15 + 16 steps side by side in 10.5 us against 36 steps in 17.6), but the separator costs 4.9 us for five steps (dump, flags,
// tiles summed out of LDS), the single substitution wave finishes the two forward passes 7 us behind the factorisation and its
// interleaved backward pass takes 10 us, and eight waves instead of six cost the one-front path itself 15 us at n = 378.
// A ten-wave form (own substitution and loader waves per front) has 168 VGPRs per wave: the allocator spilled in every path.
// NOTE: this copy predates two things the product file has: the instruction diet (29.5 -> 25.1 us) and the fix of the hand-over
// race on flagE[NT - 1] (the ring taker must wait for the old holder's last announcement; profiles/SOLVER_NOTES.md).
//
// Damped solve of the reduced camera system of a sliding window: FIVE (SIX) WAVES, the band's trailing window in matrix-core
// accumulators, no workgroup barrier anywhere.
//
// Replaces the host-side Eigen LLT / SimplicialLLT of the reference (/root/reference/src/droid_kernels.cu:200-218
// solveDenseD, :1248-1269 SparseBlock::solve) for the systems a sliding-window tracker produces: block-banded, 6 x 6 pose
// blocks, up to four blocks wide with three factor waves (every column ends inside the 48-row window of its tile column:
// ba_solve_wave_admits), up to ~seven with four (64 rows; systems up to 45 poses, whose taller panel store still fits LDS).
// Anything else is solved by the general blocked kernel's code inside the same launch (ba_solve_general.inc), and the host
// learns the verdict through pinned memory, so that the next solve of that workspace goes to ba_solve_tile.hip /
// ba_solve_band.hip directly (launch_ba_solve in ba_solve.hip).
//
// The solve is ~0.1 MFLOP; what costs is the dependent chain (n pivots) and the instructions that hang on every link: a lone
// wave issues one instruction per 5-8 cycles.  So the chain is made short per column and everything else is taken off it:
//   * block LDL^T with 4 x 4 pivots.  The trailing window -- the 3 x 3 lower tile triangle (16 x 16 tiles, 48 rows) under / right
//     of the pivot's tile column -- lives in v_mfma_f64_16x16x4_f64 accumulators for the whole factorisation; a step's rank-4
//     update of a tile is ONE instruction (C -= R (W R^T), R the raw panel, W the inverted pivot block) that broadcasts its
//     operands itself: no shuffles, no barrier;
//   * THREE factor waves, one per tile row of the window (role r holds the tiles (r, 0..r)).  Role 0 is the chain: invert the
//     pivot block (row 0 of the inverse by cofactors, 35 operations 13 deep), publish W, update the pivot tile, send its
//     rows of the next panel to LDS and read the next pivot block back.  Roles 1, 2 pick W up and do the same for their
//     rows, off the chain.  After the four steps of a tile column the ROLES rotate, not the tiles: role r becomes r - 1 (its
//     tile (r, r) is the next (r - 1, r - 1)), the wave whose pivot tile is finished takes the tile row that enters the window;
//   * lane group k (the k of the operand layouts) reads the pivot block with its indices XOR k, so that ROW 0 of its inverse is
//     row k of W: every lane computes only the row it needs, nothing is selected or exchanged;
//   * a LOADER wave brings the entering tile rows from global memory into one LDS slot, a tile column ahead: no register of a
//     factor wave ever waits for global memory (a register prefetch made every loop trip wait: the compiler's copies of
//     loop-carried registers cannot pass a pending load);
//   * a SUBSTITUTION wave runs the right-hand side one step behind the factorisation (z = W b1, b2 -= R z) and then the
//     backward substitution right-looking: lane (slot, k) accumulates v_s[k] = sum_i R_s[i][k] x[i] for the step s whose
//     window still receives solved unknowns, so the chain per step is v_readlane -> 4 FMA -> quad broadcast -> 4 FMA;
//   * the waves meet through monotone counters in LDS (flags written after the data, in program order: the LDS unit executes
//     a wave's DS instructions in order), never at a barrier.
// Measured (scratch/solve_wave_test.hip, profiles/r05_solver_stages.txt): n = 144: 30.3 us (register-tile kernel 36.9),
// n = 174: 36 (57), n = 378: 75 (skyline kernel 101.5).  tests/wave_solver_model.py is the arithmetic and the index logic of
// this file lane by lane in numpy, pinned against dense solves on the CPU (tests/test_wave_solver_model.py).
#include "ba_kernels.h"

#include <algorithm>
#include <type_traits>

#include "ba_solve_general.inc"

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

// LDS, in doubles: panel store [S][16 NT][4] | z of every step [S][4] | right-hand side / solution + flags [np + 64] | the loader's slot
__device__ __host__ __forceinline__ size_t wv_lds_doubles(int n, int nt) {
  const int np = (n + 15) & ~15, S = np >> 2;
  return (size_t)S * (16 * nt * 4 + 4) + np + 64 + (size_t)nt * 4 * 64;
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

// ---- flags between the waves of the kernel (LDS ints, monotone counters).  The CU's LDS unit executes the DS instructions it is
// handed one after the other, and a wave hands them over in program order: a flag written AFTER the data (no wait in between)
// is performed after the data, and a reader that has seen the flag reads the data.  What is needed is only that the compiler
// keeps that order (volatile accesses + memory clobbers); a release / acquire pair at workgroup scope would also wait for the
// wave's outstanding GLOBAL loads -- the tile row requested four steps ahead -- at every flag.
// (the flags are addressed as LDS explicitly: through a generic pointer the volatile accesses become FLAT instructions, which are
// slow, wait on vmcnt -- the prefetched tile row again -- and are not ordered with the DS instructions around them)
typedef __attribute__((address_space(3))) volatile int wv_lds_vint;
__device__ __forceinline__ void wv_publish(int *flag, int value) {
  asm volatile("" ::: "memory");
  *(wv_lds_vint *)flag = value;
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wv_await(int *flag, int need) {
  __builtin_amdgcn_wave_barrier();
  while (*(wv_lds_vint *)flag < need) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// NT = tile rows of the window = factor waves: 3 (48 rows: bands up to 4 poses wide) or 4 (64 rows: up to ~7 poses)
//
// An INSTANCE is one (sub-)system a group of NT + 2 waves works on.  The whole system on one front is one instance (mode 0,
// every step eliminated).  With two fronts (see ba_solve_wave_split) there are three: the top part with the separator behind
// it (mode 0, stops after the top part's steps), the bottom part in REVERSE order with the separator behind it (mode 1), and the
// separator's own system, assembled in LDS from what both fronts leave of it (mode 2).
template <int WNT>
struct WvInst {
  static constexpr int PR = 16 * WNT;  // rows of a step's panel store (the window of its tile column)
  static constexpr int PD = PR * 4;    // doubles per step
  int mode;               // 0: H as it lies; 1: H read backwards, local i <-> original n4 - 1 - i; 2: DA + DB (LDS)
  const double *H;        // the original system, [n, n] float64 row-major, lower triangle read
  int n, n4;              // its size; the size padded to a multiple of 4 (mode 1's mirror point)
  int zfrom;              // mode 1: the block [zfrom, nloc)^2 reads as ZERO -- the separator's own entries are the top front's
  const double *DA, *DB;  // mode 2: what the two fronts left of the separator block, [SEPLD][SEPLD] lower triangles
  int nloc, np;           // unknowns of this instance; padded to whole tiles (identity up to np, zeros beyond)
  int sel;                // steps (of four unknowns) to eliminate
  int smore;              // a finished pivot tile's wave takes a new tile row iff s + 1 < smore (sel: complete; sel + 1: a front)
  bool ring;              // the entering tile rows come through the loader's slot (false: they are known to be zero)
  // LDS: panel store [sel (+1)][16 NT][4] | z of every step [sel][4] | right-hand side / solution + flags [np + 64] | the loader's slot
  double *PAN, *ZST, *BV, *RING;   // RING: the tile row on its way into the window, [WNT tiles][4 regs][64 lanes]
  // W of step s stored: flagW >= s + 1; row j of panel s stored: flagE[j] >= s + 1; tile row WNT + k in the slot: flagL >= k + 1,
  // taken out of it: flagC >= k + 1
  int *flagW, *flagE, *fail, *flagL, *flagC;
  __device__ int rotations() const { return (smore > sel) ? (sel >> 2) : max((sel >> 2) - 1, 0); }   // tile rows the loader brings
  __device__ double *place(double *base) {   // lays the instance out from `base`; returns the first double behind it
    PAN = base, ZST = PAN + (size_t)(sel + ((smore > sel || (sel & 3)) ? 1 : 0)) * PD, BV = ZST + 4 * sel;   // (+1: the panel a stopped front / a partial last column still extracts)
    int *f = (int *)(BV + np + 60);
    flagW = f, flagE = f + 1, fail = f + 1 + WNT, flagL = f + 2 + WNT, flagC = f + 3 + WNT;   // (8 ints at most)
    RING = BV + np + 64;
    return RING + (ring ? WNT * 256 : 0);
  }
  __device__ void clear_flags(int t) const {
    if (t < 8) flagW[t] = 0;
  }
};
constexpr int WV_SEPLD = 36;   // the widest separator two fronts take (NT = 3: 48 window rows minus the 12 a front may stop at)

// one entry of an instance's damped, padded matrix (damping :1252-1253; always a load from inside the source, selected afterwards)
template <int WNT>
__device__ __forceinline__ double wv_elem(const WvInst<WNT> &I, int row, int col, double lm, double ep) {
  const int hi = max(row, col), lo = min(row, col);
  const double pad = (row == col && row < I.np) ? 1.0 : 0.0;
  if (I.mode == 2) {
    const int a = min(hi, WV_SEPLD - 1) * WV_SEPLD + min(lo, WV_SEPLD - 1);
    const double v = I.DA[a] + I.DB[a];
    return (hi < I.nloc) ? v : pad;
  }
  if (I.mode == 0) {
    const bool in = hi < I.nloc;
    const double hv = I.H[min(hi, I.n - 1) * I.n + min(lo, I.n - 1)];
    return in ? ((row == col) ? fma(lm, hv, hv) + ep : hv) : pad;
  }
  const int orow = I.n4 - 1 - lo, ocol = I.n4 - 1 - hi;   // (orow >= ocol)
  const bool zero = (lo >= I.zfrom) && (hi < I.nloc);
  const bool in = (hi < I.nloc) && (orow < I.n) && !zero;
  const double hv = I.H[min(orow, I.n - 1) * I.n + max(ocol, 0)];
  return in ? ((row == col) ? fma(lm, hv, hv) + ep : hv) : (zero ? 0.0 : pad);
}

// A tile of an instance in the accumulator layout: reg r <-> row 16 TI + lk + 4 r, column 16 TJ + li
template <int WNT>
__device__ __forceinline__ wv_d4 wv_load_tile(const WvInst<WNT> &I, double lm, double ep, int TI, int TJ, int lane) {
  const int li = lane & 15, lk = lane >> 4;
  wv_d4 t;
  if (TI != TJ && I.mode == 0 && 16 * TI + 15 < I.nloc) {   // wave-uniform: a tile inside the system, below the diagonal
    const double *p = I.H + (16 * TI + lk) * I.n + 16 * TJ + li;   // (n <= 384: the index fits 32 bits)
#pragma unroll
    for (int r = 0; r < 4; r++) t[r] = p[4 * r * I.n];
    return t;
  }
  if (TI != TJ && I.mode == 1 && 16 * TI + 15 < I.zfrom && 16 * TJ >= I.n4 - I.n) {   // the same, read backwards
    const double *p = I.H + (I.n4 - 1 - 16 * TJ - li) * I.n + (I.n4 - 1 - 16 * TI - lk);
#pragma unroll
    for (int r = 0; r < 4; r++) t[r] = p[-4 * r];
    return t;
  }
#pragma unroll
  for (int r = 0; r < 4; r++) t[r] = wv_elem(I, 16 * TI + lk + 4 * r, 16 * TJ + li, lm, ep);
  return t;
}

// Row 0 of the inverse of a symmetric positive definite 4 x 4 block (lower triangle a b c / d e h / f g i j) by cofactors:
// the six 2 x 2 minors of rows 2, 3 serve the four 3 x 3 cofactors of row 0; det = sum_j A[0][j] C[0][j].  35 operations, 13
// deep (the 2 x 2-block route: 42, 26 deep), on the chain of every step.  pmin collects the smallest leading minor seen
// (orders 1, 2, 3 and the determinant: all positive <=> positive definite, Sylvester); the verdict is drawn from it off the
// chain, once per tile column.  A block that is not positive definite gives garbage here and a failed solve there.
__device__ __forceinline__ void wv_invert_row0_cof(double a, double b, double c, double d, double e, double f, double g, double h,
                                                   double i, double j, double (&w)[4], double &pmin) {
  // rows: r0 = (a b d f), r1 = (b c e g), r2 = (d e h i), r3 = (f g i j)
  const double m01 = fma(d, g, -(e * f));   // |r2 r3| columns (0,1)
  const double m02 = fma(d, i, -(h * f));   // (0,2)
  const double m03 = fma(d, j, -(i * f));   // (0,3)
  const double m12 = fma(e, i, -(h * g));   // (1,2)
  const double m13 = fma(e, j, -(i * g));   // (1,3)
  const double m23 = fma(h, j, -(i * i));   // (2,3)
  // cofactors of row 0 (3 x 3 minors of rows 1..3 with the sign): expand along row 1 = (b c e g)
  const double C0 = fma(c, m23, fma(-e, m13, g * m12));
  const double C1 = -fma(b, m23, fma(-e, m03, g * m02));
  const double C2 = fma(b, m13, fma(-c, m03, g * m01));
  const double C3 = -fma(b, m12, fma(-c, m02, e * m01));
  const double det = fma(a, C0, fma(b, C1, fma(d, C2, f * C3)));
  const double det2 = fma(a, c, -(b * b));
  const double det3 = fma(d, fma(b, e, -(c * d)), fma(-e, fma(a, e, -(b * d)), h * det2));   // rows / columns 0..2
  pmin = fmin(fmin(pmin, a), fmin(det2, fmin(det3, det)));
  const double id = wv_rcp(det);
  w[0] = C0 * id, w[1] = C1 * id, w[2] = C2 * id, w[3] = C3 * id;
}

// ---- the factor waves of an instance.  Wave w (of the group) starts as the owner of tile row w of the window (role r = w:
// tiles (r, 0..r) in T[0..r]).  A step, per role: role 0 inverts the pivot block and publishes W; every role forms its operands
// from the panel rows 0..r, updates its tiles and sends its row of the next panel to the store.  After the four steps of a tile
// column the roles rotate instead of the tiles: role r becomes r - 1 (its tile (r, r) IS the next (r-1, r-1)), the wave whose
// pivot tile is finished takes the tile row that enters the window -- brought into LDS by the loader wave, so that no register
// of a factor wave ever waits for global memory (a prefetch into registers made every loop trip wait: the compiler's copies of
// the loop-carried registers cannot pass a pending load).
// On return T holds this wave's tile row of the window as the last eliminated step left it, `role` says which one it is
// (the window starts at tile column sel >> 2).
template <int NT>
__device__ void ba_solve_wave_factor(const WvInst<NT> &I, double lm, double ep, int lane, int wave, wv_d4 (&T)[NT], int &role,
                                     long long *__restrict__ prof) {
#ifdef PROFILE_SOLVE
  long long tprev_ = wall_clock64();
#endif
  constexpr int PD = WvInst<NT>::PD;
  double *const PAN = I.PAN;
  const int li = lane & 15, lk = lane >> 4;

  // tile row t of the columns of step sn -> its panel store (all 16 rows: the rows above the pivot are dead values)
  auto extract = [&](int sn, int t, const wv_d4 &c) {
    if ((li >> 2) == (sn & 3)) {
      double *p = PAN + (size_t)sn * PD + (16 * t + lk) * 4 + (li & 3);
#pragma unroll
      for (int r = 0; r < 4; r++) p[16 * r] = c[r];
    }
  };
  auto pidx = [&](int i, int j) {
    const int ii = i ^ lk, jj = j ^ lk;
    return max(ii, jj) * 4 + min(ii, jj);
  };
  const int px[10] = {pidx(0, 0), pidx(1, 0), pidx(1, 1), pidx(2, 0), pidx(2, 1), pidx(2, 2), pidx(3, 0), pidx(3, 1), pidx(3, 2), pidx(3, 3)};
  double pv[10];     // the pivot block of the coming step: a b c / d e h / f g i j (the role-0 wave's)
  double raw0n[4];   // ... and this lane's row of tile 0 of that step's panel (columns XOR lk), requested together with it
  auto read_pivot = [&](int sn) {
    const double *pp = PAN + (size_t)sn * PD + 16 * (sn & 3);
#pragma unroll
    for (int e = 0; e < 10; e++) pv[e] = pp[px[e]];
#pragma unroll
    for (int j = 0; j < 4; j++) raw0n[j] = PAN[(size_t)sn * PD + li * 4 + (j ^ lk)];
  };

#pragma unroll
  for (int j = 0; j < NT; j++) T[j] = wv_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int e = 0; e < 10; e++) pv[e] = 0.0;
#pragma unroll
  for (int j = 0; j < 4; j++) raw0n[j] = 0.0;
#pragma unroll
  for (int j = 0; j < NT; j++)
    if (j <= wave) T[j] = wv_load_tile(I, lm, ep, wave, j, lane);
  extract(0, wave, T[0]);
  wv_order();
  if (wave == 0) read_pivot(0);
  wv_publish(I.flagE + wave, 1);
  WPROF(0);

  double pmin = 1.0, wkeep[4] = {0.0, 0.0, 0.0, 0.0};
  // role 0, after its matrix instruction has been issued: W takes the pivot block's place in the panel store, the others may go
  auto publish_w = [&](int s, bool draw_verdict) {
    const int cl = 4 * (s & 3);
    double *const pan = PAN + (size_t)s * PD;
    if (draw_verdict) {   // the verdict on this wave's pivot blocks, before the step's W is announced
      if (__ballot(!(pmin > 0.0)) != 0ull && lane == 0) *(wv_lds_vint *)I.fail = 1;
    }
    if (li == 0) {
#pragma unroll
      for (int j = 0; j < 4; j++) pan[(cl + lk) * 4 + (lk ^ j)] = wkeep[j];
    }
    wv_publish(I.flagW, s + 1);
  };
  // the part of a step every variant shares: W (computed or fetched), the operands; returns whether this tile row is touched
  auto operands = [&](int s, auto rc, double &av, auto &uv) {
    constexpr int R = decltype(rc)::value;
    const int cl = 4 * (s & 3);
    double *const pan = PAN + (size_t)s * PD;
    double w[4];
    if constexpr (R == 0) {
      wv_invert_row0_cof(pv[0], pv[1], pv[2], pv[3], pv[4], pv[6], pv[7], pv[5], pv[8], pv[9], w, pmin);
#pragma unroll
      for (int j = 0; j < 4; j++) wkeep[j] = w[j];   // (stored and published after the matrix instruction is under way)
    } else {
      wv_await(I.flagW, s + 1);
#pragma unroll
      for (int j = 0; j < 4; j++) w[j] = pan[(cl + lk) * 4 + (lk ^ j)];
#pragma unroll
      for (int t = 0; t < R; t++) wv_await(I.flagE + t, s + 1);   // the rows above this wave's, stored by their owners
    }
    double raw[R + 1][4];   // this lane's rows 16 t + li of the panel, columns XOR lk, t = 0..R
    if constexpr (R == 0) {
#pragma unroll
      for (int j = 0; j < 4; j++) raw[0][j] = raw0n[j];   // (requested a step ahead, with the pivot block)
    } else {
#pragma unroll
      for (int t = 0; t <= R; t++)
#pragma unroll
        for (int j = 0; j < 4; j++) raw[t][j] = pan[(16 * t + li) * 4 + (j ^ lk)];
    }
    {
      const bool live = li > cl + 3;  // rows of tile 0 at or above the pivot are eliminated: they take no part
#pragma unroll
      for (int j = 0; j < 4; j++) raw[0][j] = live ? raw[0][j] : 0.0;
    }
    av = -raw[R][0];
#pragma unroll
    for (int t = 0; t <= R; t++) uv[t] = fma(w[3], raw[t][3], fma(w[2], raw[t][2], fma(w[1], raw[t][1], w[0] * raw[t][0])));
    // (lane (li, lk) holds R[16 R + li][lk] in raw[R][0]: the ballot sees every entry of this tile row's panel)
    return (R == 0) || (__ballot(raw[R][0] != 0.0) != 0ull);
  };
  // steps 4 tb .. 4 tb + cnt - 1 in role R; after a whole column (cnt = 4) the wave is role R - 1 (R >= 1) or NT - 1 (R = 0)
  auto run_column = [&](int tb, int cnt, auto rc) {
    constexpr int R = decltype(rc)::value;
    const int mid = min(cnt, 3);
    for (int q = 0; q < mid; q++) {
      const int s = 4 * tb + q;
      double av, uv[R + 1];
      const bool any = operands(s, rc, av, uv);
      if (any) {
#pragma unroll
        for (int j = 0; j <= R; j++) T[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, uv[j], T[j], 0, 0, 0);
      }
      if constexpr (R == 0) publish_w(s, s + 1 == I.sel);
      extract(s + 1, R, T[0]);
      if constexpr (R == 0) {
        wv_order();
        read_pivot(s + 1);
      }
      wv_publish(I.flagE + R, s + 2);
    }
    if (cnt < 4) return;
    const int s = 4 * tb + 3;
    const bool more = s + 1 < I.smore;
    double av, uv[R + 1];
    const bool any = operands(s, rc, av, uv);
    if constexpr (R >= 1) {   // tile column 0 is finished: its updates are skipped; this wave's tile (R, 1) is the next (R-1, 0)
      if (any) {
#pragma unroll
        for (int j = 1; j <= R; j++) T[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, uv[j], T[j], 0, 0, 0);
      }
      if (more) {
        extract(s + 1, R - 1, T[1]);
        if constexpr (R == 1) {   // the chain is this wave's now
          wv_order();
          read_pivot(s + 1);
        }
        wv_publish(I.flagE + (R - 1), s + 2);
      }
#pragma unroll
      for (int j = 1; j <= R; j++) T[j - 1] = T[j];
    } else {                  // the pivot tile is finished: take the tile row that enters the window
      publish_w(s, true);
      if (more) {
        if (I.ring) {         // ... from the loader's slot
          wv_await(I.flagL, tb + 1);
          const double *ring = I.RING;
#pragma unroll
          for (int j = 0; j < NT; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) T[j][r] = ring[(j * 4 + r) * 64 + lane];
        } else {              // ... known to be empty (the separator's system fits the first window)
#pragma unroll
          for (int j = 0; j < NT; j++) T[j] = wv_d4{0.0, 0.0, 0.0, 0.0};
        }
        extract(s + 1, NT - 1, T[0]);
        wv_publish(I.flagE + (NT - 1), s + 2);   // (orders the reads of the slot before ...)
        if (I.ring) wv_publish(I.flagC, tb + 1);   // ... the slot is free again)
      }
    }
  };
  role = wave;
  for (int tb = 0; 4 * tb < I.sel; tb++) {
    const int cnt = min(I.sel - 4 * tb, 4);
    if (role == 0) run_column(tb, cnt, std::integral_constant<int, 0>{});
    else if (role == 1) run_column(tb, cnt, std::integral_constant<int, 1>{});
    else if (role == 2 || NT == 3) run_column(tb, cnt, std::integral_constant<int, 2>{});
    else run_column(tb, cnt, std::integral_constant<int, NT - 1>{});
    if (cnt == 4) role = (role == 0) ? NT - 1 : role - 1;
  }
  WPROF(1 + wave);
}

// what a front that stopped after sel steps (a = 4 sel unknowns) leaves of the block [a, a + sep)^2: its lower triangle out of
// the window's tiles into D [WV_SEPLD][WV_SEPLD]; `mirrored`: the front ran backwards, the block's indices are reversed
template <int NT>
__device__ __forceinline__ void wv_dump(const wv_d4 (&T)[NT], int role, int sel, int sep, double *__restrict__ D, bool mirrored,
                                        int lane) {
  const int li = lane & 15, lk = lane >> 4, tbf = sel >> 2, a = 4 * sel;
#pragma unroll
  for (int j = 0; j < NT; j++) {
    if (j > role) continue;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = 16 * (tbf + role) + lk + 4 * r - a, col = 16 * (tbf + j) + li - a;
      if (col >= 0 && col <= row && row < sep)
        D[mirrored ? (sep - 1 - col) * WV_SEPLD + (sep - 1 - row) : row * WV_SEPLD + col] = T[j][r];
    }
  }
}

// ---- the loader wave.  Tile row NT + k of an instance (the tiles (NT + k, k + 1 .. k + NT): what the window gains when it
// leaves tile column k) -> the instance's one LDS slot, as soon as the previous occupant has been taken.  One wave serves both
// fronts, a tile column ahead of each.
template <int WNT>
__device__ void ba_solve_wave_loader(const WvInst<WNT> &Ia, const WvInst<WNT> &Ib, bool both, double lm, double ep, int lane) {
  const int na = Ia.rotations(), nb = both ? Ib.rotations() : 0;
  auto put = [&](const WvInst<WNT> &I, int k, const wv_d4 (&t)[WNT]) {
    wv_await(I.flagC, k);
#pragma unroll
    for (int j = 0; j < WNT; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) I.RING[(j * 4 + r) * 64 + lane] = t[j][r];
    wv_publish(I.flagL, k + 1);
  };
  for (int k = 0; k < max(na, nb); k++) {
    wv_d4 ta[WNT], tb[WNT];
    if (k < na) {
#pragma unroll
      for (int j = 0; j < WNT; j++) ta[j] = wv_load_tile(Ia, lm, ep, WNT + k, k + 1 + j, lane);
    }
    if (k < nb) {
#pragma unroll
      for (int j = 0; j < WNT; j++) tb[j] = wv_load_tile(Ib, lm, ep, WNT + k, k + 1 + j, lane);
    }
    if (k < na) put(Ia, k, ta);
    if (k < nb) put(Ib, k, tb);
  }
}

// ---- the substitution wave, first half: the right-hand side behind the factorisation (z = W b1, b2 -= R z), a step at a time
template <int WNT>
__device__ __forceinline__ bool wv_forward_ready(const WvInst<WNT> &I, int s) {
  bool ok = *(wv_lds_vint *)I.flagW >= s + 1;
#pragma unroll
  for (int t = 0; t < WNT; t++) ok = ok && (*(wv_lds_vint *)(I.flagE + t) >= s + 1);
  return ok;
}
template <int WNT>
struct WvFwd {   // a step's operands: requested together (both fronts' when both are ready), used afterwards
  static constexpr int PR = WvInst<WNT>::PR, PD = WvInst<WNT>::PD;
  double b[4], w[4][4], rr[4], bo;
  __device__ __forceinline__ void load(const WvInst<WNT> &I, int s, int lane) {
    const int tb = s >> 2, cl = 4 * (s & 3), row = min(lane, PR - 1);
    const double *pan = I.PAN + (size_t)s * PD;
#pragma unroll
    for (int k = 0; k < 4; k++) b[k] = I.BV[4 * s + k];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int m = 0; m < 4; m++) w[k][m] = pan[(cl + k) * 4 + m];
#pragma unroll
    for (int m = 0; m < 4; m++) rr[m] = pan[row * 4 + m];
    bo = I.BV[16 * tb + row];
  }
  __device__ __forceinline__ void finish(const WvInst<WNT> &I, int s, int lane) const {
    const int tb = s >> 2, cl = 4 * (s & 3);
    double z[4];
#pragma unroll
    for (int k = 0; k < 4; k++) z[k] = fma(w[k][3], b[3], fma(w[k][2], b[2], fma(w[k][1], b[1], w[k][0] * b[0])));
    if (lane < PR && lane > cl + 3) I.BV[16 * tb + lane] = fma(-rr[3], z[3], fma(-rr[2], z[2], fma(-rr[1], z[1], fma(-rr[0], z[0], bo))));
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 4; k++) I.ZST[4 * s + k] = z[k];
    }
  }
};
template <int WNT>
__device__ void ba_solve_wave_forward(const WvInst<WNT> &I, int lane) {
  for (int s = 0; s < I.sel; s++) {
    wv_await(I.flagW, s + 1);
#pragma unroll
    for (int t = 0; t < WNT; t++) wv_await(I.flagE + t, s + 1);
    WvFwd<WNT> f;
    f.load(I, s, lane);
    f.finish(I, s, lane);
    wv_order();
  }
}
// both fronts, whichever has a step ready
template <int WNT>
__device__ void ba_solve_wave_forward2(const WvInst<WNT> &Ia, const WvInst<WNT> &Ib, int lane) {
  int sa = 0, sb = 0;
  while (sa < Ia.sel || sb < Ib.sel) {
    __builtin_amdgcn_wave_barrier();
    const bool ra = sa < Ia.sel && wv_forward_ready(Ia, sa), rb = sb < Ib.sel && wv_forward_ready(Ib, sb);
    asm volatile("" ::: "memory");
    WvFwd<WNT> fa, fb;
    if (ra && rb) {
      fa.load(Ia, sa, lane), fb.load(Ib, sb, lane);
      fa.finish(Ia, sa++, lane), fb.finish(Ib, sb++, lane);
    } else if (ra) {
      fa.load(Ia, sa, lane);
      fa.finish(Ia, sa++, lane);
    } else if (rb) {
      fb.load(Ib, sb, lane);
      fb.finish(Ib, sb++, lane);
    } else {
      __builtin_amdgcn_s_sleep(1);
    }
    wv_order();
  }
}

// ... second half: the backward substitution, right-looking: lane (slot, k) = (lane >> 2, lane & 3) accumulates
// v_s[k] = sum_i R_s[i][k] x[i] for the step s = slot (mod 16) that still receives solved unknowns (a window spans at most
// 4 NT <= 16 steps); x1 = z - W v.  Steps sel .. Sx - 1 are not this instance's to solve: their unknowns lie in BV already (the
// separator's, solved elsewhere) and are only handed on.
// Lane (slot, k).  Per step: the four v of the step's slot come by v_readlane; EVERY lane forms x1[k] for its own k (row k
// of W: two 16-byte reads) -- so each quad holds x1[0..3] -- and fetches the other three from its quad neighbours (DPP);
// then its own accumulator takes the step's contribution.  ~40 instructions per step, the chain: readlane -> 4 FMA ->
// quad broadcast -> 4 FMA.
template <int WNT>
struct WvBack {
  static constexpr int PR = WvInst<WNT>::PR, PD = WvInst<WNT>::PD;
  struct Ops { wv_d2 w01, w23; double z, r[4]; bool valid; };
  const double *PAN, *ZST;
  double *BV;
  int sel, slot, kk, lane;
  double v;
  Ops oa, ob;
  __device__ WvBack(const WvInst<WNT> &I, int lane_) : PAN(I.PAN), ZST(I.ZST), BV(I.BV), sel(I.sel), slot(lane_ >> 2), kk(lane_ & 3), lane(lane_), v(0.0) {}
  __device__ __forceinline__ void fetch(int sp, Ops &o) {   // everything step sp reads that does not hang on the chain
    const int spc = max(sp, 0);
    const bool own = spc < sel;
    const int spa = min(spc, sel - 1);
    const double *wr = PAN + (size_t)spa * PD + 16 * (spa & 3) + 4 * kk;   // row k of W (in the pivot block's place)
    const wv_d2 w01 = *(const wv_d2 *)wr, w23 = *(const wv_d2 *)(wr + 2);
    const double zs = ZST[4 * spa + kk], xg = BV[4 * spc + kk];
    o.w01 = own ? w01 : wv_d2{0.0, 0.0}, o.w23 = own ? w23 : wv_d2{0.0, 0.0};
    o.z = own ? zs : xg;
    const int dd = (spc - 1 - slot) & 15, sq = spc - 1 - dd;
    const int lrow = 4 * spc - 16 * (sq >> 2);
    o.valid = (sq >= 0) && (sq < sel) && (lrow + 3 <= PR - 1);
    const double *rp = PAN + (size_t)min(max(sq, 0), sel - 1) * PD + min(lrow, PR - 4) * 4 + kk;
#pragma unroll
    for (int m = 0; m < 4; m++) o.r[m] = rp[4 * m];
    // keeps the prefetch where it was issued
    asm volatile("" : "+v"(o.w01), "+v"(o.w23), "+v"(o.z));
#pragma unroll
    for (int k = 0; k < 4; k++) asm volatile("" : "+v"(o.r[k]));
  }
  template <int M>
  static __device__ __forceinline__ double quad(double x) {   // the value of lane (quad, M) in all four lanes of the quad
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), M * 0x55, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), M * 0x55, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
  }
  // act = false: a step that does not exist (the shorter of two interleaved chains): nothing changes.  `trash` != nullptr: the
  // lanes that have nothing to store write there instead of branching around the store (one basic block for two chains)
  __device__ __forceinline__ void solve_step(int sp, const Ops &o, bool act = true, double *trash = nullptr) {
    const int l0 = 4 * (sp & 15);
    const double v0 = wv_readlane(v, l0), v1 = wv_readlane(v, l0 + 1), v2 = wv_readlane(v, l0 + 2), v3 = wv_readlane(v, l0 + 3);
    const double xk = fma(-o.w23.y, v3, fma(-o.w23.x, v2, fma(-o.w01.y, v1, fma(-o.w01.x, v0, o.z))));   // x1[kk]
    if (trash) *((lane < 4 && act) ? BV + 4 * sp + lane : trash + lane) = xk;
    else if (lane < 4) BV[4 * sp + lane] = xk;
    const double x0 = quad<0>(xk), x1 = quad<1>(xk), x2 = quad<2>(xk), x3 = quad<3>(xk);
    const double upd = fma(o.r[3], x3, fma(o.r[2], x2, fma(o.r[1], x1, o.r[0] * x0)));
    v = (o.valid && act) ? v + upd : v;
    v = (act && slot == (sp & 15)) ? 0.0 : v;
  }
};
template <int WNT>
__device__ void ba_solve_wave_backward(const WvInst<WNT> &I, int Sx, int lane) {
  WvBack<WNT> A(I, lane);
  A.fetch(Sx - 1, A.oa);
  for (int sp = Sx - 1; sp >= 0; sp -= 2) {   // two steps per trip, each one's operands requested a step ahead
    A.fetch(sp - 1, A.ob);
    A.solve_step(sp, A.oa);
    if (sp - 1 < 0) break;
    A.fetch(sp - 2, A.oa);
    A.solve_step(sp - 1, A.ob);
  }
  wv_order();
  __builtin_amdgcn_s_waitcnt(0xc07f);
}
// two instances at once: the two chains are independent, their steps alternate in the instruction stream
template <int WNT>
__device__ void ba_solve_wave_backward2(const WvInst<WNT> &Ia, int Sxa, const WvInst<WNT> &Ib, int Sxb, int lane) {
  WvBack<WNT> A(Ia, lane), B(Ib, lane);
  A.fetch(Sxa - 1, A.oa), B.fetch(Sxb - 1, B.oa);
  int pa = Sxa - 1, pb = Sxb - 1;
  for (; pa >= 0 || pb >= 0; pa -= 2, pb -= 2) {   // (the loaders are through: their slots take the idle lanes' stores)
    A.fetch(pa - 1, A.ob), B.fetch(pb - 1, B.ob);
    A.solve_step(pa, A.oa, pa >= 0, Ia.RING), B.solve_step(pb, B.oa, pb >= 0, Ib.RING);
    A.fetch(pa - 2, A.oa), B.fetch(pb - 2, B.oa);
    A.solve_step(pa - 1, A.ob, pa >= 1, Ia.RING), B.solve_step(pb - 1, B.ob, pb >= 1, Ib.RING);
  }
  wv_order();
  __builtin_amdgcn_s_waitcnt(0xc07f);
}

// the verdict and the store: x [n] (LDS) -> dx; non-finite results count as failure too; failure => zero update (:1263-1266)
__device__ __forceinline__ void wv_store(const double *x, int n, bool bad, float *__restrict__ dx, int *__restrict__ meta, int lane) {
  bool nf = false;
  for (int j = lane; j < n; j += 64) nf |= !isfinite(x[j]);
  const bool failed = bad || (__ballot(nf) != 0ull);
  for (int j = lane; j < n; j += 64) dx[j] = failed ? 0.f : (float)x[j];
  if (lane == 0) meta[1] = failed ? 1 : 0;
}

// ---- one front, window of NT tile rows: NT + 2 waves (NT for the factorisation, one for the substitution, one that brings tile
// rows in).  The kernel uses it for NT = 4; the 48-row window goes through ba_solve_wave_run_fronts.
template <int NT>
__device__ __forceinline__ void ba_solve_wave_run(const double *__restrict__ H, const double *__restrict__ bvec, int n, double lm,
                                                  double ep, float *__restrict__ dx, int *__restrict__ meta,
                                                  double *__restrict__ smem, int lane, int wave, long long *__restrict__ prof) {
  WvInst<NT> I;
  I.mode = 0, I.H = H, I.n = n, I.n4 = n, I.zfrom = 0, I.DA = I.DB = nullptr, I.nloc = n, I.np = (n + 15) & ~15;
  I.sel = I.smore = I.np >> 2, I.ring = true;
  I.place(smem);
  I.clear_flags(threadIdx.x);
  __syncthreads();
  if (wave < NT) {
    wv_d4 T[NT];
    int role;
    ba_solve_wave_factor<NT>(I, lm, ep, lane, wave, T, role, prof);
  } else if (wave == NT) {
#ifdef PROFILE_SOLVE
    long long tprev_ = wall_clock64();
#endif
    for (int i = lane; i < I.np + 60; i += 64) {
      const double bv = bvec[min(i, n - 1)];
      I.BV[i] = (i < n) ? bv : 0.0;
    }
    wv_order();
    ba_solve_wave_forward<NT>(I, lane);
    const bool bad = *(wv_lds_vint *)I.fail != 0;
    WPROF(4);
    ba_solve_wave_backward<NT>(I, I.sel, lane);
    WPROF(7);
    wv_store(I.BV, n, bad, dx, meta, lane);
    WPROF(6);
  } else if (wave == NT + 1) {
    ba_solve_wave_loader<NT>(I, I, false, lm, ep, lane);
  }
}

// ---- the 48-row window (NT = 3), eight waves: ONE front (plan.sep = 0: waves 0..2 factorise, 6 substitutes, 7 loads) or TWO
// FRONTS AROUND A SEPARATOR.  The system, padded to n4 = n + (n & 2) unknowns, is cut into top [0, a_t) | separator
// [a_t, a_t + sep) | bottom [a_t + sep, n4) such that no top column reaches the bottom part.  Waves 0..2 eliminate the top part
// (forwards), waves 3..5 the bottom part (BACKWARDS: the same code on the mirrored matrix), each with the separator's rows
// behind it; what the two leave of the separator block is added up into the separator's own system, which the top front's
// waves factorise; the substitution wave serves both fronts (forward: whichever has a step ready; backward: the two chains
// interleaved), with the separator's solve in between.  The chain is half as long, the rest (the separator) is a few steps.
struct WvSplit { int a_t, sep, a_b; };

__device__ __forceinline__ size_t wv_split_doubles(int n4, const WvSplit &p) {
  constexpr int PD = WvInst<3>::PD;
  auto inst = [&](int nloc, int sel, bool front) {
    return (size_t)(sel + 1) * PD + 4 * sel + ((nloc + 15) & ~15) + 64 + (front ? 3 * 256 : 0);
  };
  return inst(p.a_t + p.sep, p.a_t >> 2, true) + inst(p.a_b + p.sep, p.a_b >> 2, true) + inst(p.sep, (p.sep + 3) >> 2, false) +
         2 * WV_SEPLD * WV_SEPLD + n4 + 16;
}

__device__ __forceinline__ void ba_solve_wave_run_fronts(const double *__restrict__ H, const double *__restrict__ bvec, int n, double lm,
                                                         double ep, float *__restrict__ dx, int *__restrict__ meta,
                                                         double *__restrict__ smem, int lane, int wave, const WvSplit plan,
                                                         long long *__restrict__ prof) {
  constexpr int NT = 3;
  const int n4 = n + (n & 2), sep = plan.sep, a_t = sep ? plan.a_t : n, a_b = sep ? plan.a_b : 0, ssep = (sep + 3) >> 2;
  WvInst<NT> It, Ib, Is;
  It.mode = 0, It.H = H, It.n = n, It.n4 = n4, It.zfrom = 0, It.DA = It.DB = nullptr;
  It.nloc = a_t + sep, It.np = (It.nloc + 15) & ~15, It.ring = true;
  It.sel = sep ? (a_t >> 2) : (It.np >> 2), It.smore = It.sel + (sep ? 1 : 0);   // (sep = 0: the whole system on this front)
  Ib = It;
  Ib.mode = 1, Ib.zfrom = a_b, Ib.nloc = a_b + sep, Ib.np = (Ib.nloc + 15) & ~15, Ib.sel = a_b >> 2, Ib.smore = Ib.sel + 1;
  double *p = It.place(smem);
  if (sep) p = Ib.place(p);
  else Ib.place(p);   // (never touched)
  Is = It;
  Is.mode = 2, Is.nloc = sep, Is.np = 16 * NT, Is.sel = Is.smore = ssep, Is.ring = false;
  if (sep) p = Is.place(p);
  else Is.place(p);
  // (one front: nothing behind its instance, the solution is stored from its BV)
  double *const DT = p, *const DBm = DT + WV_SEPLD * WV_SEPLD, *const X = sep ? DBm + WV_SEPLD * WV_SEPLD : It.BV;
  Is.DA = DT, Is.DB = DBm;
  int *const dumped = (int *)(X + n4);   // dumped[front * 3 + wave]: that wave's part of the separator block is in DT / DBm
  It.clear_flags(threadIdx.x);
  if (sep) {
    Ib.clear_flags(threadIdx.x), Is.clear_flags(threadIdx.x);
    if (threadIdx.x < 8) dumped[threadIdx.x] = 0;
  }
  __syncthreads();
  if (wave < 2 * NT) {
    const bool bottom = wave >= NT;
    if (bottom && !sep) return;
    const int w = bottom ? wave - NT : wave;
    wv_d4 T[NT];
    int role;
    ba_solve_wave_factor<NT>(bottom ? Ib : It, lm, ep, lane, w, T, role, (prof && bottom) ? prof + 8 : prof);
    if (!sep) return;
    wv_dump<NT>(T, role, bottom ? Ib.sel : It.sel, sep, bottom ? DBm : DT, bottom, lane);
    wv_publish(dumped + wave, 1);
    if (!bottom) {   // the separator's system: its tiles are sums of the two dumps
      for (int k = 0; k < 2 * NT; k++) wv_await(dumped + k, 1);
      ba_solve_wave_factor<NT>(Is, lm, ep, lane, w, T, role, prof ? prof + 12 : nullptr);
    }
  } else if (wave == 2 * NT) {
#ifdef PROFILE_SOLVE
    long long tprev_ = wall_clock64();
#endif
    // right-hand sides: the separator's own entries are the top front's, the bottom front starts from zero there
    for (int i = lane; i < It.np + 60; i += 64) {
      const double bv = bvec[min(i, n - 1)];
      It.BV[i] = (i < It.nloc) ? bv : 0.0;
    }
    if (sep) {
      for (int i = lane; i < Ib.np + 60; i += 64) {
        const int o = n4 - 1 - i;
        const double bv = bvec[min(max(o, 0), n - 1)];
        Ib.BV[i] = (i < a_b && o < n) ? bv : 0.0;
      }
    }
    wv_order();
    if (sep) ba_solve_wave_forward2<NT>(It, Ib, lane);
    else ba_solve_wave_forward<NT>(It, lane);
    WPROF(4);
    bool bad = *(wv_lds_vint *)It.fail != 0;
    if (sep) {
      for (int i = lane; i < Is.np + 60; i += 64)
        Is.BV[i] = (i < sep) ? It.BV[a_t + i] + Ib.BV[a_b + sep - 1 - i] : 0.0;
      wv_order();
      ba_solve_wave_forward<NT>(Is, lane);
      ba_solve_wave_backward<NT>(Is, ssep, lane);
      for (int i = lane; i < sep; i += 64) {
        const double x = Is.BV[i];
        It.BV[a_t + i] = x, Ib.BV[a_b + sep - 1 - i] = x;
      }
      wv_order();
      bad = bad || (*(wv_lds_vint *)Ib.fail != 0) || (*(wv_lds_vint *)Is.fail != 0);
      WPROF(5);
      ba_solve_wave_backward2<NT>(It, It.sel + ssep, Ib, Ib.sel + ssep, lane);
      for (int i = lane; i < a_b; i += 64) X[n4 - 1 - i] = Ib.BV[i];
    } else {
      ba_solve_wave_backward<NT>(It, It.sel, lane);
    }
    if (sep) {
      for (int i = lane; i < It.nloc; i += 64) X[i] = It.BV[i];
    }
    wv_order();
    WPROF(7);
    wv_store(X, n, bad, dx, meta, lane);
    WPROF(6);
  } else {
    ba_solve_wave_loader<NT>(It, Ib, sep != 0, lm, ep, lane);
  }
}

// The admission test, by every wave for itself: with the pose-level skyline fpose (first pose a pose is coupled with) made
// monotone, every column's last row must lie inside the window of its step's tile column: row < 16 (s >> 2) + 16 NT.
// Returns the smallest NT in {3, 4} (<= max_nt: the panel store of NT = 4 does not fit LDS for the largest systems) that admits
// the system, or 0.  With `split` (and NT = 3) also looks for the cut of two fronts: the smallest separator (a multiple of 4)
// such that no column of the top part reaches the bottom part, the block [a, a + sep) lies inside the window a front stops in
// on both sides, the bottom part passes the window test in ITS order of elimination, and everything fits lds_doubles.
__device__ __forceinline__ int ba_solve_wave_admits(const int *__restrict__ fpose, int n, int lane, int max_nt, WvSplit *split,
                                                    size_t lds_doubles) {
  const int P = n / 6;
  if (split) split->sep = 0;
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
  if (max_nt >= 3 && __ballot(!ok3) == 0ull) {
    if (split) {
      const int n4 = n + (n & 2);
      for (int sep = 4; sep <= WV_SEPLD; sep += 4) {
        WvSplit c;
        c.sep = sep, c.a_t = ((n4 - sep) / 2) & ~3, c.a_b = n4 - sep - c.a_t;
        if (c.a_t < 16 || c.a_b < 16 || c.a_b > 256) break;
        if (6 * __builtin_amdgcn_readlane(last, (c.a_t - 1) / 6) + 5 >= c.a_t + sep) continue;
        if (4 * ((c.a_t >> 2) & 3) + sep > 48 || 4 * ((c.a_b >> 2) & 3) + sep > 48) continue;
        const int omin = n4 - 1 - (4 * lane + 3);    // lane = a step of the bottom front: the smallest original column it eliminates
        const int oc = min(max(omin, 0), n - 1);
        const int first = (omin < n) ? 6 * __shfl(g, oc / 6, 64) : omin;
        const bool live = 4 * lane < c.a_b;
        if (__ballot(live && (n4 - 1 - first > 16 * (lane >> 2) + 47)) != 0ull) continue;
        if (wv_split_doubles(n4, c) > lds_doubles) break;
        *split = c;
        break;
      }
    }
    return 3;
  }
  if (max_nt >= 4 && __ballot(!ok4) == 0ull) return 4;
  return 0;
}

// the kernel: eight waves.  Every wave runs the admission test (no exchange needed to agree).  A system that is not admitted is
// solved by the general blocked kernel's code with the same waves; `verdict` (pinned host memory) tells the host which it
// was: 1 taken, 2 not.  fronts = 1: never two fronts.
template <bool GENERAL_IN_LDS>
__global__ __launch_bounds__(512) void ba_solve_wave_kernel(const double *__restrict__ H, const double *__restrict__ bvec,
                                                            const int *__restrict__ fpose, int n, double lm, double ep,
                                                            float *__restrict__ dx, int *__restrict__ meta,
                                                            double *__restrict__ Lglobal, int *__restrict__ verdict, int max_nt,
                                                            int fronts, unsigned lds_doubles, long long *__restrict__ prof) {
  extern __shared__ __attribute__((aligned(16))) double wv_smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  WvSplit split;
  const int nt = ba_solve_wave_admits(fpose, n, lane, max_nt, fronts >= 2 ? &split : nullptr, lds_doubles);
  if (fronts < 2) split.sep = 0;
  if (threadIdx.x == 0) {
    meta[3] = 1;   // (solved either way: a kernel queued behind with `skip_if_solved` returns at once)
    if (verdict) __hip_atomic_store(verdict, nt ? 1 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (nt == 3) ba_solve_wave_run_fronts(H, bvec, n, lm, ep, dx, meta, wv_smem, lane, wave, split, prof);
  else if (nt == 4) ba_solve_wave_run<4>(H, bvec, n, lm, ep, dx, meta, wv_smem, lane, wave, prof);
  else ba_solve_general_body<GENERAL_IN_LDS>(H, bvec, n, lm, ep, dx, meta, Lglobal, nullptr, wv_smem);
}

static int wave_max_nt(int n) {   // the tallest window whose panel store fits LDS for a system of n unknowns (0: none)
  if (n <= 0 || n % 6 != 0 || n / 6 > 64) return 0;
  if (wv_lds_doubles(n, 4) * sizeof(double) <= (size_t)SOLVE_MAX_LDS_BYTES) return 4;
  if (wv_lds_doubles(n, 3) * sizeof(double) <= (size_t)SOLVE_MAX_LDS_BYTES) return 3;
  return 0;
}

bool ba_solve_wave_supported(int n) { return wave_max_nt(n) != 0; }

// two fronts are tried for the sizes whose three instances can fit LDS at all and that are long enough to gain
// (DBA_SOLVE_FRONTS=1 keeps one front)
static int wave_fronts(int n) {
  static const int fronts = [] {
    const char *e = getenv("DBA_SOLVE_FRONTS");
    return e ? atoi(e) : 2;
  }();
  return (fronts >= 2 && n >= 72 && n <= 252) ? 2 : 1;
}

template <bool GENERAL_IN_LDS>
static int wave_launch(const double *H, const double *b, const int *fpose, int n, double lm, double ep, float *dx, int *meta,
                       double *Lscratch, int *verdict, int max_nt, int fronts, size_t lds, hipStream_t stream, long long *prof) {
  static DeviceOnce attr_once;
  if (attr_once.needed()) {
    DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_solve_wave_kernel<GENERAL_IN_LDS>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SOLVE_MAX_LDS_BYTES));
    attr_once.done();
  }
  hipLaunchKernelGGL((ba_solve_wave_kernel<GENERAL_IN_LDS>), dim3(1), dim3(512), lds, stream, H, b, fpose, n, lm, ep, dx, meta,
                     Lscratch, verdict, max_nt, fronts, (unsigned)(lds / sizeof(double)), prof);
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

// Lscratch: the workspace's packed-triangle scratch (needed by the fall-back when the system does not fit LDS: n > 199)
int launch_ba_solve_wave(const double *H, const double *b, const int *fpose, int n, double lm, double ep, float *dx, int *meta,
                         double *Lscratch, int *verdict, hipStream_t stream, long long *prof) {
  const int max_nt = wave_max_nt(n);
  if (!max_nt || !fpose) return DBA_ERR_UNSUPPORTED;
  const size_t wave_lds = wv_lds_doubles(n, max_nt) * sizeof(double);
  const size_t gen_lds = solve_packed_bytes(n) + solve_small_bytes(n);
  const bool in_lds = gen_lds <= (size_t)SOLVE_MAX_LDS_BYTES;
  if (!in_lds && !Lscratch) return DBA_ERR_WORKSPACE;
  const int fronts = wave_fronts(n);
  const size_t lds = fronts >= 2 ? (size_t)SOLVE_MAX_LDS_BYTES
                                 : std::min((size_t)SOLVE_MAX_LDS_BYTES, std::max(wave_lds, in_lds ? gen_lds : solve_small_bytes(n)));
  return in_lds ? wave_launch<true>(H, b, fpose, n, lm, ep, dx, meta, Lscratch, verdict, max_nt, fronts, lds, stream, prof)
                : wave_launch<false>(H, b, fpose, n, lm, ep, dx, meta, Lscratch, verdict, max_nt, fronts, lds, stream, prof);
}

}  // namespace dba
