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
//   * instruction issue is the budget of every wave here, so what the chain wave and the substitution wave do per step was cut by
//     hand: per-lane addresses are formed once per tile column and the step inside the column sits in the instructions' offset
//     fields (19 address instructions per step gone from the chain, ~10 from the substitution); a wave that waits for several
//     flags reads them with ONE 16-byte LDS access instead of one round trip after the other, and polls without sleeping (the
//     chain passes through such a wake-up once per tile column); the positive-definiteness minors are computed while the next
//     pivot block's LDS reads are in flight; the substitution's readlane values feed the FMAs from scalar registers (inline asm:
//     the compiler copied each into a vector register first).  n = 144: 29.5 -> 25.2 us, n = 378: 72.9 -> 62.1 us.
// Measured (scratch/solve_wave_test.hip, profiles/r05_solver_stages.txt): n = 144: 25.2 us (register-tile kernel 37.0),
// n = 174: 30.0 (59.5), n = 240: 39.9, n = 378: 62.1 (skyline kernel 101.5-106), 64-row window: 26.7.
// tests/wave_solver_model.py is the arithmetic and the index logic of this file lane by lane in numpy, pinned against dense
// solves on the CPU (tests/test_wave_solver_model.py).
#include "ba_kernels.h"

#include <algorithm>
#include <type_traits>

#include "ba_solve_admit.h"
#include "ba_solve_general.inc"

namespace dba {

constexpr int WV_THREADS = 448;   // up to five factor waves, the substitution wave, the loader

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

// LDS, in doubles: panel store [K][16 NT][4] | z of every step [S][4] | right-hand side / solution + flags [np + 72] | the loader's slot
// K = S: the panel of every step stays in LDS.  K < S (round 6: the 64-row window on systems of more than 45 poses, whose panel
// store is 192 KB at 63 poses): the store is a RING of K steps -- the forward pass only ever needs the current panel, the
// substitution wave copies each of the first E = S - K panels to global scratch as it passes (it reads the whole panel anyway),
// panel s + K then takes panel s's slot, and during the backward pass the loader wave, which has nothing left to do, brings
// the early panels back into the slots the backward pass has finished with, several tile columns ahead of their use.  E is a
// multiple of 4 (whole tile columns) and E <= K, so that both [0, E) and [E, S) are contiguous in the ring: the per-column
// pointer arithmetic of the factor waves and the backward pass's address recurrence stay as they are, only the base changes at
// the seam (a lane of the backward pass changes its pending step in strides of 16: it crosses the seam at a step of its own).
__device__ __host__ __forceinline__ size_t wv_lds_doubles_k(int n, int nt, int K) {
  const int np = (n + 15) & ~15, S = np >> 2;
  return (size_t)K * (16 * nt * 4) + (size_t)S * 4 + np + 88 + (size_t)nt * 4 * 64;
}
// the number of early panels that leave LDS (0: the whole store fits; -1: no admissible ring either)
__device__ __host__ __forceinline__ int wv_ring_early(int n, int nt) {
  const int np = (n + 15) & ~15, S = np >> 2;
  for (int E = 0; E == 0 || (2 * E <= S && S - E >= 32); E += 4)   // (K >= 32: the slot hand-over distances, see the loader)
    if (wv_lds_doubles_k(n, nt, S - E) * sizeof(double) <= (size_t)SOLVE_MAX_LDS_BYTES) return E;
  return -1;
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
// N consecutive flags (flag is 16-byte aligned) all >= need: ONE LDS read per look instead of N round trips one after the other
typedef int wv_i4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) volatile wv_i4 wv_lds_vint4;
template <int N>
__device__ __forceinline__ void wv_await_all(int *flag, int need) {
  static_assert(N >= 1 && N <= 6, "flagW and up to five flagE");
  __builtin_amdgcn_wave_barrier();
  for (;;) {
    const wv_i4 v = *(wv_lds_vint4 *)flag;
    int m = v.x;
    if (N > 1) m = min(m, v.y);
    if (N > 2) m = min(m, v.z);
    if (N > 3) m = min(m, v.w);
    if (N > 4) m = min(m, *(wv_lds_vint *)(flag + 4));
    if (N > 5) m = min(m, *(wv_lds_vint *)(flag + 5));
    if (m >= need) break;   // (polling without a sleep: these waves have their SIMD to themselves, and the chain passes through
  }                         // the wake-up of role 1 once per tile column and of the substitution wave at the end)
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// NT = tile rows of the window = factor waves: 3 (48 rows: bands up to 4 poses wide) or 4 (64 rows: up to ~7 poses)
template <int WNT, bool WRING = false>
struct WvLayout {   // LDS, in doubles
  static constexpr int PR = 16 * WNT;  // rows of a step's panel store (the window of its tile column)
  static constexpr int PD = PR * 4;    // doubles per step
  int np, S, K, E;                     // K panels in LDS, the first E = S - K of the S leave it for global scratch (WRING)
  double *PAN, *ZST, *BV, *RING;       // RING: the tile row on its way into the window, [WNT tiles][4 regs][64 lanes]
  // W of step s stored: flagW >= s + 1; row j of panel s stored: flagE[j] >= s + 1; tile row WNT + k in the slot: flagL >= k + 1,
  // taken out of it: flagC >= k + 1.  Ring of panels: the substitution wave has left step s behind (forward): flagF >= s + 1
  // (announced per tile column); tile columns it has finished on the way back: flagB; early tile columns back in LDS: flagR
  int *flagW, *flagE, *fail, *flagL, *flagC, *flagF, *flagB, *flagR;
  __device__ WvLayout(double *smem, int n, int ring_e) {
    np = (n + 15) & ~15, S = np >> 2;
    E = WRING ? ring_e : 0, K = S - E;
    PAN = smem, ZST = PAN + (size_t)K * PD, BV = ZST + 4 * S;
    int *f = (int *)(BV + np + 80);   // (the right-hand side reaches 16 nt rows past the last tile column's first: np + 64 at most)
    flagW = f, flagE = f + 1;                                   // (flagW and up to five flagE: adjacent, 16-byte aligned)
    fail = f + 8, flagL = f + 9, flagC = f + 10, flagF = f + 11, flagB = f + 12, flagR = f + 13;
    RING = BV + np + 88;
  }
  // the panel of step s: steps [E, S) lie at the ring's slots [0, K), steps [0, E) at [K - E, K) (until step s + K takes the slot)
  __device__ __forceinline__ double *pan(int s) const {
    if constexpr (!WRING) return PAN + (size_t)s * PD;
    else return PAN + (size_t)(s - E + (s < E ? K : 0)) * PD;
  }
};

// A tile of the damped, padded system in the accumulator layout: reg r <-> row 16 TI + lk + 4 r, column 16 TJ + li
// (H: [n, n] float64 row-major, lower triangle read; damping :1252-1253; identity padding up to np, zeros beyond)
__device__ __forceinline__ wv_d4 wv_load_tile(const double *__restrict__ H, int n, int np, double lm, double ep, int TI, int TJ,
                                              int lane) {
  const int li = lane & 15, lk = lane >> 4;
  wv_d4 t;
  if (16 * TI + 15 < n) {   // wave-uniform: a tile inside the system
    if (TI != TJ) {
      const double *p = H + (16 * TI + lk) * n + 16 * TJ + li;   // (n <= 384: the index fits 32 bits)
#pragma unroll
      for (int r = 0; r < 4; r++) t[r] = p[4 * r * n];
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 16 * TI + lk + 4 * r, col = 16 * TJ + li;
        const double hv = H[max(row, col) * n + min(row, col)];   // (the mirrored upper half)
        t[r] = (row == col) ? fma(lm, hv, hv) + ep : hv;
      }
    }
    return t;
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = 16 * TI + lk + 4 * r, col = 16 * TJ + li;
    const int hi = max(row, col), lo = min(row, col);
    const bool in = hi < n;
    const double hv = H[min(hi, n - 1) * n + min(lo, n - 1)];   // (always a load from inside the matrix, selected afterwards)
    const double dv = in ? fma(lm, hv, hv) + ep : ((row < np) ? 1.0 : 0.0);
    t[r] = (row == col) ? dv : (in ? hv : 0.0);
  }
  return t;
}

// Row 0 of the inverse of a symmetric positive definite 4 x 4 block (lower triangle a b c / d e h / f g i j) by cofactors:
// the six 2 x 2 minors of rows 2, 3 serve the four 3 x 3 cofactors of row 0; det = sum_j A[0][j] C[0][j].  35 operations, 13
// deep (the 2 x 2-block route: 42, 26 deep), on the chain of every step.  A block that is not positive definite gives garbage
// here and a failed solve there: wv_pd_minors.
__device__ __forceinline__ void wv_invert_row0_cof(double a, double b, double c, double d, double e, double f, double g, double h,
                                                   double i, double j, double (&w)[4], double &det) {
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
  det = fma(a, C0, fma(b, C1, fma(d, C2, f * C3)));
  const double id = wv_rcp(det);
  w[0] = C0 * id, w[1] = C1 * id, w[2] = C2 * id, w[3] = C3 * id;
}
// pmin collects the smallest leading minor seen (orders 1, 2, 3 and the determinant: all positive <=> positive definite,
// Sylvester); the verdict is drawn from it once per tile column.  Not on the chain: the role-0 wave computes it while the next
// pivot block's LDS reads are in flight.
__device__ __forceinline__ void wv_pd_minors(double a, double b, double c, double d, double e, double h, double det, double &pmin) {
  const double det2 = fma(a, c, -(b * b));
  const double det3 = fma(d, fma(b, e, -(c * d)), fma(-e, fma(a, e, -(b * d)), h * det2));   // rows / columns 0..2
  pmin = fmin(fmin(pmin, a), fmin(det2, fmin(det3, det)));
}

// ---- waves 0..2: the factorisation.  Wave w starts as the owner of tile row w of the window (role r = w: tiles (r, 0..r)
// in T[0..r]).  A step, per role: role 0 inverts the pivot block and publishes W; every role forms its operands from the
// panel rows 0..r, updates its tiles and sends its row of the next panel to the store.  After the four steps of a tile column
// the roles rotate instead of the tiles: role r becomes r - 1 (its tile (r, r) IS the next (r-1, r-1)), the wave whose pivot
// tile is finished takes the tile row that enters the window -- brought into LDS by the loader wave, so that no register of a
// factor wave ever waits for global memory (a prefetch into registers made every loop trip wait: the compiler's copies of the
// loop-carried registers cannot pass a pending load).
template <int NT, bool RING>
__device__ void ba_solve_wave_factor(int n, double *__restrict__ smem, const double *__restrict__ H, double lm, double ep, int lane,
                                     int wave, int ring_e, long long *__restrict__ prof) {
#ifdef PROFILE_SOLVE
  long long tprev_ = wall_clock64();
#endif
  const WvLayout<NT, RING> L(smem, n, ring_e);
  constexpr int PD = WvLayout<NT, RING>::PD;
  const int S = L.S, TB = S >> 2;
  const int li = lane & 15, lk = lane >> 4;

  // tile row t of the columns of step sn -> its panel store (all 16 rows: the rows above the pivot are dead values)
  auto extract = [&](int sn, int t, const wv_d4 &c) {
    if ((li >> 2) == (sn & 3)) {
      double *p = L.pan(sn) + (16 * t + lk) * 4 + (li & 3);
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
    const double *pn = L.pan(sn), *pp = pn + 16 * (sn & 3);
#pragma unroll
    for (int e = 0; e < 10; e++) pv[e] = pp[px[e]];
#pragma unroll
    for (int j = 0; j < 4; j++) raw0n[j] = pn[li * 4 + (j ^ lk)];
  };

  // The same accesses inside a tile column whose first step's panel lies at `col`: the per-lane part of every address is
  // formed ONCE per column, the step inside the column (a compile-time q) goes into the instructions' offset fields -- a lone
  // wave pays 5-8 cycles for every address it computes, and the chain wave formed ~19 per step.
  struct ColPtr {
    const double *pe[10], *pr[4];
    double *xw[4];
  };
  auto col_ptrs = [&](double *col, ColPtr &C) {
#pragma unroll
    for (int e = 0; e < 10; e++) C.pe[e] = col + px[e];
#pragma unroll
    for (int j = 0; j < 4; j++) C.pr[j] = col + li * 4 + (j ^ lk), C.xw[j] = col + lk * 4 + (lk ^ j);
  };
  auto read_pivot_c = [&](const ColPtr &C, auto qc) {
    constexpr int q = decltype(qc)::value;
#pragma unroll
    for (int e = 0; e < 10; e++) pv[e] = C.pe[e][q * (PD + 16)];
#pragma unroll
    for (int j = 0; j < 4; j++) raw0n[j] = C.pr[j][q * PD];
  };
  auto extract_c = [&](double *xe, auto qc, const wv_d4 &c) {   // xe = col + (16 t + lk) * 4 + (li & 3)
    constexpr int q = decltype(qc)::value;
    if ((li >> 2) == q) {
#pragma unroll
      for (int r = 0; r < 4; r++) xe[q * PD + 16 * r] = c[r];
    }
  };

  wv_d4 T[NT];
#pragma unroll
  for (int j = 0; j < NT; j++) T[j] = wv_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int e = 0; e < 10; e++) pv[e] = 0.0;
#pragma unroll
  for (int j = 0; j < 4; j++) raw0n[j] = 0.0;
#pragma unroll
  for (int j = 0; j < NT; j++)
    if (j <= wave) T[j] = wv_load_tile(H, n, L.np, lm, ep, wave, j, lane);
  extract(0, wave, T[0]);
  wv_order();
  if (wave == 0) read_pivot(0);
  wv_publish(L.flagE + wave, 1);
  WPROF(0);

  double pmin = 1.0, wkeep[4] = {0.0, 0.0, 0.0, 0.0}, detk = 1.0;
  // role 0, after its matrix instruction has been issued: W takes the pivot block's place in the panel store, the others may go
  auto publish_w = [&](int s, bool last_of_column) {
    const int cl = 4 * (s & 3);
    double *const pan = L.pan(s);
    if (last_of_column) {   // the verdict on this wave's four pivot blocks, before the step's W is announced
      if (__ballot(!(pmin > 0.0)) != 0ull && lane == 0) *(wv_lds_vint *)L.fail = 1;
    }
    if (li == 0) {
#pragma unroll
      for (int j = 0; j < 4; j++) pan[(cl + lk) * 4 + (lk ^ j)] = wkeep[j];
    }
    wv_publish(L.flagW, s + 1);
  };
  auto publish_w_c = [&](const ColPtr &C, int s, auto qc) {
    constexpr int q = decltype(qc)::value;
    if (li == 0) {
#pragma unroll
      for (int j = 0; j < 4; j++) C.xw[j][q * (PD + 16)] = wkeep[j];
    }
    wv_publish(L.flagW, s + 1);
  };
  // the part of a step every variant shares: W (computed or fetched), the operands; returns whether this tile row is touched
  auto operands = [&](int s, const double *pan, auto rc, double &av, auto &uv) {   // pan: the panel of step s
    constexpr int R = decltype(rc)::value;
    const int cl = 4 * (s & 3);
    double w[4];
    if constexpr (R == 0) {
      wv_invert_row0_cof(pv[0], pv[1], pv[2], pv[3], pv[4], pv[6], pv[7], pv[5], pv[8], pv[9], w, detk);
#pragma unroll
      for (int j = 0; j < 4; j++) wkeep[j] = w[j];   // (stored and published after the matrix instruction is under way)
    } else {
      wv_await_all<R + 1>(L.flagW, s + 1);   // W, and the rows above this wave's, stored by their owners (flagW, flagE[0 .. R-1])
#pragma unroll
      for (int j = 0; j < 4; j++) w[j] = pan[(cl + lk) * 4 + (lk ^ j)];
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
    av = -raw[R][0];
#pragma unroll
    for (int t = 0; t <= R; t++) uv[t] = fma(w[3], raw[t][3], fma(w[2], raw[t][2], fma(w[1], raw[t][1], w[0] * raw[t][0])));
    {
      const bool live = li > cl + 3;  // rows of tile 0 at or above the pivot are eliminated: they take no part
      uv[0] = live ? uv[0] : 0.0;
      if constexpr (R == 0) av = live ? av : 0.0;
    }
    // (lane (li, lk) holds R[16 R + li][lk] in raw[R][0]: the ballot sees every entry of this tile row's panel)
    return (R == 0) || (__ballot(raw[R][0] != 0.0) != 0ull);
  };
  // steps 4 tb .. 4 tb + 3 in role R; afterwards the wave is role R - 1 (R >= 1) or NT - 1 (R = 0)
  auto run_column = [&](int tb, auto rc) {
    constexpr int R = decltype(rc)::value;
    double *const col = L.pan(4 * tb);
    double *const xe = col + (16 * R + lk) * 4 + (li & 3);
    ColPtr C;
    if constexpr (R == 0) col_ptrs(col, C);
    auto mid = [&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const int s = 4 * tb + q;
      double av, uv[R + 1];
      double ma = pv[0], mb = pv[1], mc = pv[2], md = pv[3], me = pv[4], mh = pv[5];   // (role 0: for the minors, below)
      const bool any = operands(s, col + q * PD, rc, av, uv);
      if (any) {
#pragma unroll
        for (int j = 0; j <= R; j++) T[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, uv[j], T[j], 0, 0, 0);
      }
      if constexpr (R == 0) publish_w_c(C, s, qc);
      extract_c(xe, std::integral_constant<int, q + 1>{}, T[0]);
      if constexpr (R == 0) {
        wv_order();
        read_pivot_c(C, std::integral_constant<int, q + 1>{});
        // while those reads are under way: was this step's pivot block positive definite?
        asm volatile("" : "+v"(ma), "+v"(mb), "+v"(mc), "+v"(md), "+v"(me), "+v"(mh));
        wv_pd_minors(ma, mb, mc, md, me, mh, detk, pmin);
      }
      wv_publish(L.flagE + R, s + 2);
    };
    mid(std::integral_constant<int, 0>{});
    mid(std::integral_constant<int, 1>{});
    mid(std::integral_constant<int, 2>{});
    const int s = 4 * tb + 3;
    const bool more = s + 1 < S;
    if constexpr (R >= 1) {
      if (!more) return;   // the very last step only concerns the pivot tile (and nobody reads a panel the way back may be replacing)
    }
    double av, uv[R + 1];
    const bool any = operands(s, col + 3 * PD, rc, av, uv);
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
        wv_publish(L.flagE + (R - 1), s + 2);
      }
#pragma unroll
      for (int j = 1; j <= R; j++) T[j - 1] = T[j];
    } else {                  // the pivot tile is finished: take the tile row that enters the window from the loader's slot
      wv_pd_minors(pv[0], pv[1], pv[2], pv[3], pv[4], pv[5], detk, pmin);
      publish_w(s, true);
      if (more) {
        wv_await(L.flagL, tb + 1);
        const double *ring = L.RING;
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
          for (int r = 0; r < 4; r++) T[j][r] = ring[(j * 4 + r) * 64 + lane];
        extract(s + 1, NT - 1, T[0]);
        // flagE[NT - 1] changes hands here, from the wave that held the last tile row through this column to this one.  That
        // wave runs off the chain, behind: its own last announcement (row NT - 1 of panel s, value s + 1) must be out before
        // this one -- otherwise the readers take panel s's row for stored when it is not, and the counter steps back when the
        // late store lands.  (This wave is off the chain from here on: the wait costs nothing.)  Found in round 5 when the
        // chain got faster: one wrong solve in ~40 cold starts, scratch/solve_stress.py / solve_cold.py.
#ifndef WV_TEST_UNFIXED_HANDOVER   // (only ever defined by scratch/build_unfixed_lib.sh: does the cold-start stress test see the race?)
        wv_await(L.flagE + (NT - 1), s + 1);
#endif
        wv_publish(L.flagE + (NT - 1), s + 2);   // (orders the reads of the slot before ...)
        wv_publish(L.flagC, tb + 1);             // ... the slot is free again)
      }
    }
  };
  int role = wave;
  for (int tb = 0; tb < TB; tb++) {
    if (role == 0) __builtin_amdgcn_s_setprio(3);
    else if (role == 1) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(0);
    if (role == 0) run_column(tb, std::integral_constant<int, 0>{});
    else if (role == 1) run_column(tb, std::integral_constant<int, 1>{});
    else if (role == 2 || NT == 3) run_column(tb, std::integral_constant<int, 2>{});
    else if (role == 3 || NT == 4) run_column(tb, std::integral_constant<int, (NT > 3 ? 3 : 2)>{});
    else run_column(tb, std::integral_constant<int, NT - 1>{});
    role = (role == 0) ? NT - 1 : role - 1;
  }
  WPROF(wave < 3 ? 1 + wave : 5 + wave);   // (slots 1-3, then 8, 9; the substitution wave's are 4-6)
}

// ---- wave 4: the loader.  Tile row NT + k of the system (the tiles (NT + k, k + 1 .. k + NT): what the window gains when it
// leaves tile column k) -> the one LDS slot, as soon as the previous occupant has been taken.
// Ring of panels (WRING): this wave is also the one that keeps a panel slot from being written before its occupant has been
// copied out, and the one that brings the early panels back.
//   * forward: panel s + K takes the slot of panel s.  The chain wave runs through a tile column without waiting for anybody,
//     so a wave working in tile column c is only known to have seen flagL >= c - (WNT - 1) (at the rotation in which it was
//     handed the entering tile row); in column c it writes panels up to 4 c + 4.  flagL = k + 1 is therefore only announced once
//     the substitution wave has left step 4 (k + WNT) + 4 - K behind (flagF).  The factor waves can complete everything up to
//     step 4 k + 3 without that announcement, the substitution wave follows them: no deadlock as long as K >= 4 WNT + 1; in
//     practice the substitution wave is two or three steps behind the chain and the wait never spins.
//   * backward: tile column c of the early panels goes back into the slots of tile column c + K / 4, which the substitution wave
//     has finished with when it announces flagB >= TB - (c + K / 4); it reads column c from iteration c + WNT on (its lanes'
//     pending steps reach 4 WNT - 1 steps below the one it solves) and waits for flagR there: K / 4 - WNT >= 4 columns of
//     slack, of which a column's round trip to L2 takes about two.
template <int WNT, bool WRING>
__device__ void ba_solve_wave_loader(int n, double *__restrict__ smem, const double *__restrict__ H, double lm, double ep, int lane,
                                     int ring_e, const double *__restrict__ spill) {
  const WvLayout<WNT, WRING> L(smem, n, ring_e);
  constexpr int PD = WvLayout<WNT, WRING>::PD;
  const int TB = L.S >> 2;
  for (int k = 0; k + 1 < TB; k++) {   // (the rotation behind the last tile column brings nothing in)
    wv_d4 t[WNT];
#pragma unroll
    for (int j = 0; j < WNT; j++) t[j] = wv_load_tile(H, n, L.np, lm, ep, WNT + k, k + 1 + j, lane);
    wv_await(L.flagC, k);
    if constexpr (WRING) {
      const int need = 4 * (k + WNT) + 5 - L.K;
      if (need > 0) wv_await(L.flagF, need);
    }
#pragma unroll
    for (int j = 0; j < WNT; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) L.RING[(j * 4 + r) * 64 + lane] = t[j][r];
    wv_publish(L.flagL, k + 1);
  }
  if constexpr (WRING) {
    const int E4 = L.E >> 2, K4 = L.K >> 2;
    constexpr int NL = PD / 32;          // 16-byte pieces per lane of a tile column's four panels
    wv_await(L.flagB, 1);                // the forward pass is over and the substitution wave has waited for its stores
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (nothing of the scratch may be served from this CU's L1)
    for (int c = E4 - 1; c >= 0; c--) {
      const double *g = spill + (size_t)(4 * c) * PD + 2 * lane;
      wv_d2 t[NL];
#pragma unroll
      for (int i = 0; i < NL; i++) t[i] = *(const wv_d2 *)(g + 128 * i);
      wv_await(L.flagB, TB - (c + K4));
      double *p = L.pan(4 * c) + 2 * lane;
#pragma unroll
      for (int i = 0; i < NL; i++) *(wv_d2 *)(p + 128 * i) = t[i];
      wv_publish(L.flagR, E4 - c);
    }
  }
}

// ---- wave 3: the right-hand side behind the factorisation (z = W b1, b2 -= R z), then the backward substitution,
// right-looking: lane (slot, k) = (lane >> 2, lane & 3) accumulates v_s[k] = sum_i R_s[i][k] x[i] for the step s = slot
// (mod 16) that still receives solved unknowns (a window spans at most 4 NT <= 16 steps); x1 = z - W v.
template <int WNT, bool WRING>
__device__ void ba_solve_wave_subst(const double *__restrict__ bvec, int n, float *__restrict__ dx, int *__restrict__ meta,
                                    double *__restrict__ smem, int lane, int ring_e, double *__restrict__ spill,
                                    long long *__restrict__ prof) {
#ifdef PROFILE_SOLVE
  long long tprev_ = wall_clock64();
#endif
  const WvLayout<WNT, WRING> L(smem, n, ring_e);
  constexpr int PR = WvLayout<WNT, WRING>::PR, PD = WvLayout<WNT, WRING>::PD;
  constexpr int XR = PR > 64 ? PR - 64 : 0;   // rows of the window beyond one per lane (16 with five tile rows)
  static_assert(!WRING || PR >= 64, "the copy to scratch takes a panel row per lane");
  const int np = L.np, S = L.S;
  double *const PAN = L.PAN, *const ZST = L.ZST, *const BV = L.BV;
  for (int i = lane; i < np + 80; i += 64) {
    const double bv = bvec[min(i, n - 1)];
    BV[i] = (i < n) ? bv : 0.0;
  }
  wv_order();
  for (int s = 0; s < S; s++) {
    const int tb = s >> 2, cl = 4 * (s & 3);
    const double *pan = L.pan(s);
    wv_await_all<WNT + 1>(L.flagW, s + 1);   // (flagW, flagE[0 .. WNT-1]: adjacent)
    double z[4];
    const double b0 = BV[4 * s], b1 = BV[4 * s + 1], b2 = BV[4 * s + 2], b3 = BV[4 * s + 3];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const double *wr = pan + (cl + k) * 4;
      z[k] = fma(wr[3], b3, fma(wr[2], b2, fma(wr[1], b1, wr[0] * b0)));
    }
    auto sub_row = [&](int row, const wv_d2 &r01, const wv_d2 &r23) {   // b2[row] -= R[row][:] z
      double *bp = BV + 16 * tb + row;
      *bp = fma(-r23.y, z[3], fma(-r23.x, z[2], fma(-r01.y, z[1], fma(-r01.x, z[0], *bp))));
    };
    if constexpr (WRING) {
      // every lane reads its panel row (the dead rows above the pivot and W's rows too): the early panels leave for scratch
      const wv_d2 r01 = *(const wv_d2 *)(pan + lane * 4), r23 = *(const wv_d2 *)(pan + lane * 4 + 2);
      wv_d2 x01 = {0.0, 0.0}, x23 = {0.0, 0.0};
      if constexpr (XR > 0) {
        if (lane < XR) x01 = *(const wv_d2 *)(pan + (64 + lane) * 4), x23 = *(const wv_d2 *)(pan + (64 + lane) * 4 + 2);
      }
      if (s < L.E) {
        double *g = spill + (size_t)s * PD + lane * 4;
        *(wv_d2 *)g = r01, *(wv_d2 *)(g + 2) = r23;
        if constexpr (XR > 0) {
          if (lane < XR) *(wv_d2 *)(g + 256) = x01, *(wv_d2 *)(g + 258) = x23;
        }
      }
      if (lane > cl + 3) sub_row(lane, r01, r23);
      if constexpr (XR > 0) {
        if (lane < XR) sub_row(64 + lane, x01, x23);
      }
    } else {
      if (lane < PR && lane > cl + 3) {
        const double *rr = pan + lane * 4;
        double *bp = BV + 16 * tb + lane;
        *bp = fma(-rr[3], z[3], fma(-rr[2], z[2], fma(-rr[1], z[1], fma(-rr[0], z[0], *bp))));
      }
      if constexpr (XR > 0) {
        if (lane < XR) sub_row(64 + lane, *(const wv_d2 *)(pan + (64 + lane) * 4), *(const wv_d2 *)(pan + (64 + lane) * 4 + 2));
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 4; k++) ZST[4 * s + k] = z[k];
    }
    wv_order();
    if constexpr (WRING) {
      if ((s & 3) == 3) wv_publish(L.flagF, s + 1);   // panels up to s have been read (and sent off): their slots may be rewritten
    }
  }
  if constexpr (WRING) {   // the copies have arrived before anybody is told that the way back has begun (flagB)
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  }
  const bool bad = *(wv_lds_vint *)L.fail != 0;
  WPROF(4);
  {
    // Lane (slot, k).  Per step: the four v of the step's slot come by v_readlane; EVERY lane forms x1[k] for its own k (row k
    // of W: two 16-byte reads) -- so each quad holds x1[0..3] -- and fetches the other three from its quad neighbours (DPP);
    // then its own accumulator takes the step's contribution.  ~40 instructions per step, the chain: readlane -> 4 FMA ->
    // quad broadcast -> 4 FMA.
    // Five tile rows (80-row window): a panel spans up to 19 steps, the 16 slots do not cover them -- a lane's PREVIOUS pending
    // step (16 steps further down) already receives its first contributions (distances 16 .. 18 - (slot & 3)) while the
    // lane's accumulator still belongs to the current one; they go into a second accumulator, which becomes the first when the
    // current step is solved.
    constexpr bool FAR = PR > 64;
    const int slot = lane >> 2, kk = lane & 3;
    double v = 0.0, vfar = 0.0;
    struct Ops { wv_d2 w01, w23; double z, r[4], r2[4]; bool valid, valid2; };
    // Addresses.  The steps are walked a tile column (four steps, q = 3..0) at a time, so that what depends on the step inside
    // the column sits in the instructions' offset fields and the per-lane part is formed once per column (row k of W:
    // wc + q (PD + 16); z: zc + 4 q; the solved unknowns: xc + 4 q).  The panel rows a lane's accumulator takes
    // (R_s[lrow .. lrow + 3][k] of its pending step s) move by a fixed stride from step to step: with u = sp - 1 - slot,
    //   index = slot PD + 16 (slot & 3) + k + 16 + 16 u + 16 (PD - 16) (u >> 4),  valid <=> u >= 0 and (u & 15) <= dmax(slot)
    // (tests/test_wave_solver_model.py holds the closed form against the direct one), so rp just steps down by 16 doubles, by
    // 16 (PD - 15) when the lane's slot is the step's own (u & 15 wraps) -- ~7 instead of ~17 address instructions per step.
    const int dmax = PR / 4 - 2 - (slot & 3);
    int u = S - 2 - slot;                                  // for the step fetched next (S - 1 first)
    const double *rp = PAN + (slot * PD + 16 * (slot & 3) + kk + 16) + 16 * u + 16 * (PD - 16) * (u >> 4);
    if constexpr (WRING) {   // (that was the address in a store of all S panels: the ring holds step s at s - E, or s - E + K below the seam)
      const int pending = slot + 16 * (u >> 4);
      rp += (ptrdiff_t)((u >= 0 && pending < L.E) ? L.K - L.E : -L.E) * PD;
    }
    // a lane's pending step slot + 16 m crosses the seam when m drops below seam_m, i.e. when its u becomes 16 seam_m - 1
    const int seam_m = (L.E - slot + 15) >> 4, seam_u = 16 * seam_m - 1, seam_jump = L.K * PD;
    const double *const rsafe = PAN + kk;
    auto fetch = [&](const double *wc, const double *zc, int sp, auto qc, Ops &o) {   // everything step sp = 4 c + q reads off the chain
      constexpr int q = decltype(qc)::value;
      o.w01 = *(const wv_d2 *)(wc + q * (PD + 16)), o.w23 = *(const wv_d2 *)(wc + q * (PD + 16) + 2);
      o.z = zc[4 * q];
      o.valid = (u >= 0) && ((u & 15) <= dmax);
      const double *ra = o.valid ? rp : rsafe;
#pragma unroll
      for (int m = 0; m < 4; m++) o.r[m] = ra[4 * m];
      if constexpr (FAR) {   // the same rows of x in the panel 16 steps down: 64 rows further into its window
        o.valid2 = (u >= 16) && ((u & 15) <= dmax - 16);
        const double *rb = rp - 16 * PD + 256;
        if constexpr (WRING) rb += ((u >> 4) == seam_m) ? seam_jump : 0;   // (that panel lies below the seam, this one above)
        rb = o.valid2 ? rb : rsafe;
#pragma unroll
        for (int m = 0; m < 4; m++) o.r2[m] = rb[4 * m];
      }
      // ... and the lane's state for the step below this one
      u -= 1;
      rp -= (slot == ((sp - 1) & 15)) ? 16 * (PD - 15) : 16;
      if constexpr (WRING) rp += (u == seam_u) ? seam_jump : 0;
      // (keeps the prefetch where it was issued)
      asm volatile("" : "+v"(o.w01), "+v"(o.w23), "+v"(o.z));
#pragma unroll
      for (int k = 0; k < 4; k++) asm volatile("" : "+v"(o.r[k]));
      if constexpr (FAR) {
#pragma unroll
        for (int k = 0; k < 4; k++) asm volatile("" : "+v"(o.r2[k]));
      }
    };
    auto quad = [&](double x, auto mc) {   // the value of lane (quad, m) in all four lanes of the quad
      constexpr int m = decltype(mc)::value;
      const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), m * 0x55, 0xf, 0xf, false);
      const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), m * 0x55, 0xf, 0xf, false);
      return __hiloint2double(hi, lo);
    };
    auto solve_step = [&](double *xc, int sp, auto qc, const Ops &o) {
      constexpr int q = decltype(qc)::value;
      const int l0 = 4 * (sp & 15);
      const double v0 = wv_readlane(v, l0), v1 = wv_readlane(v, l0 + 1), v2 = wv_readlane(v, l0 + 2), v3 = wv_readlane(v, l0 + 3);
      // x1[kk] = z - W[k][:] v: the four v stay in scalar registers (left to itself the compiler copies each into a vector
      // register first: 8 moves per step on a chain that is bound by instruction issue)
      auto fnma_vs = [](double w, double sv, double acc) {
        double r;
        asm("v_fma_f64 %0, -%1, %2, %3" : "=v"(r) : "v"(w), "s"(sv), "v"(acc));
        return r;
      };
      const double xk = fnma_vs(o.w23.y, v3, fnma_vs(o.w23.x, v2, fnma_vs(o.w01.y, v1, fnma_vs(o.w01.x, v0, o.z))));
      if (lane < 4) xc[4 * q] = xk;
      const double x0 = quad(xk, std::integral_constant<int, 0>{}), x1 = quad(xk, std::integral_constant<int, 1>{}),
                   x2 = quad(xk, std::integral_constant<int, 2>{}), x3 = quad(xk, std::integral_constant<int, 3>{});
      const double upd = fma(o.r[3], x3, fma(o.r[2], x2, fma(o.r[1], x1, o.r[0] * x0)));
      if constexpr (FAR) {
        const double upd2 = fma(o.r2[3], x3, fma(o.r2[2], x2, fma(o.r2[1], x1, o.r2[0] * x0)));
        const bool own = slot == (sp & 15);        // this lane's step has just been solved: its next one takes over
        const double base = own ? vfar : v;
        v = o.valid ? base + upd : base;
        vfar = own ? 0.0 : (o.valid2 ? vfar + upd2 : vfar);
      } else {
        v = o.valid ? v + upd : v;
        v = (slot == (sp & 15)) ? 0.0 : v;
      }
    };
    using Q0 = std::integral_constant<int, 0>;
    using Q1 = std::integral_constant<int, 1>;
    using Q2 = std::integral_constant<int, 2>;
    using Q3 = std::integral_constant<int, 3>;
    auto wcol = [&](int c) { return L.pan(4 * c) + 4 * kk; };   // row k of W of the column's first step
    Ops oa, ob;
    const int TBs = S >> 2;   // (S is a multiple of 4: np is a multiple of 16)
    const double *wc = wcol(TBs - 1), *zc = ZST + 16 * (TBs - 1) + kk;
    fetch(wc, zc, S - 1, Q3{}, oa);
    for (int c = TBs - 1; c >= 0; c--) {   // each step's operands requested a step ahead
      const int s0 = 4 * c, cn = max(c - 1, 0);
      double *const xc = BV + 16 * c + lane;
      if constexpr (WRING) {   // this iteration reads tile columns c - WNT .. c: are the early ones back?
        const int E4 = L.E >> 2;
        if (c - WNT < E4) wv_await(L.flagR, min(E4, E4 - (c - WNT)));
      }
      fetch(wc, zc, s0 + 2, Q2{}, ob);
      solve_step(xc, s0 + 3, Q3{}, oa);
      fetch(wc, zc, s0 + 1, Q1{}, oa);
      solve_step(xc, s0 + 2, Q2{}, ob);
      fetch(wc, zc, s0, Q0{}, ob);
      solve_step(xc, s0 + 1, Q1{}, oa);
      wc = wcol(cn), zc = ZST + 16 * cn + kk;
      fetch(wc, zc, s0 - 1, Q3{}, oa);     // (column c - 1; below the first column: a harmless re-read of column 0)
      solve_step(xc, s0, Q0{}, ob);
      if constexpr (WRING) wv_publish(L.flagB, TBs - c);   // (every read of tile column c and above has been issued)
    }
  }
  wv_order();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  WPROF(5);
  // non-finite results count as failure too; failure => zero update (:1263-1266)
  bool nf = false;
  for (int j = lane; j < n; j += 64) nf |= !isfinite(BV[j]);
  const bool failed = bad || (__ballot(nf) != 0ull);
  for (int j = lane; j < n; j += 64) dx[j] = failed ? 0.f : (float)BV[j];
  if (lane == 0) meta[1] = failed ? 1 : 0;
  WPROF(6);
}

// the kernel: NT + 2 waves (NT for the factorisation, one for the substitution, one that brings tile rows in; launched with six,
// the sixth leaves at once when the system fits the 48-row window).  Every wave runs the admission test (no exchange needed
// to agree).  A system that is not admitted is solved by the general blocked kernel's code with the same waves; `verdict`
// (pinned host memory) tells the host which it was: 1 taken, 2 not.
template <int NT, bool RING>
__device__ __forceinline__ void ba_solve_wave_run(const double *__restrict__ H, const double *__restrict__ bvec, int n, double lm,
                                                  double ep, float *__restrict__ dx, int *__restrict__ meta,
                                                  double *__restrict__ smem, int lane, int wave, int ring_e,
                                                  double *__restrict__ spill, long long *__restrict__ prof) {
  {
    const WvLayout<NT, RING> L(smem, n, ring_e);
    if (threadIdx.x < 16) L.flagW[threadIdx.x] = 0;   // flagW, flagE[NT] | fail, flagL, flagC, flagF, flagB, flagR
  }
  __syncthreads();
  // Which hardware wave plays which part: the CU deals a workgroup's waves out to its four SIMDs in turn (wave i -> SIMD i & 3),
  // and a wave that shares its SIMD with another busy one issues at half the rate -- with five factor waves the chain slowed from
  // 0.40 to 0.55 us per step.  So: the factor waves that share a SIMD are two roles apart (never chain and next-in-line), the
  // loader (asleep most of the time) and the substitution wave take the other shared places, and the instruction arbiter is told
  // who matters (s_setprio: chain 3, next in line 2, substitution 1).
  int part = wave;                    // 0 .. NT - 1: factor wave, NT: substitution, NT + 1: loader
  if constexpr (NT == 5) {
    // SIMD 0: factor 0 + 2, SIMD 1: factor 1 + loader, SIMD 2: factor 3 + substitution, SIMD 3: factor 4
    part = wave == 2 ? 3 : wave == 3 ? 4 : wave == 4 ? 2 : wave == 5 ? 6 : wave == 6 ? 5 : wave;
  }
  if (part < NT) ba_solve_wave_factor<NT, RING>(n, smem, H, lm, ep, lane, part, ring_e, prof);
  else if (part == NT) {
    __builtin_amdgcn_s_setprio(1);
    ba_solve_wave_subst<NT, RING>(bvec, n, dx, meta, smem, lane, ring_e, spill, prof);
  } else if (part == NT + 1) ba_solve_wave_loader<NT, RING>(n, smem, H, lm, ep, lane, ring_e, spill);
}

template <bool GENERAL_IN_LDS>
__global__ __launch_bounds__(WV_THREADS) void ba_solve_wave_kernel(const double *__restrict__ H, const double *__restrict__ bvec,
                                                            const int *__restrict__ fpose, int n, double lm, double ep,
                                                            float *__restrict__ dx, int *__restrict__ meta,
                                                            double *__restrict__ Lglobal, int *__restrict__ verdict, int max_nt,
                                                            int ring_e4, int ring_e5, long long *__restrict__ prof) {
  extern __shared__ __attribute__((aligned(16))) double wv_smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nt = ba_solve_wave_admits(fpose, n, lane, max_nt);
  if (threadIdx.x == 0) {
    meta[3] = 1;   // (solved either way: a kernel queued behind with `skip_if_solved` returns at once)
    if (verdict) __hip_atomic_store(verdict, nt ? 1 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // (ring_e4: how many of the 64-row window's early panels leave LDS for Lglobal -- 0 up to 45 poses)
  if (nt == 3) ba_solve_wave_run<3, false>(H, bvec, n, lm, ep, dx, meta, wv_smem, lane, wave, 0, nullptr, prof);
  else if (nt == 4 && ring_e4 == 0) ba_solve_wave_run<4, false>(H, bvec, n, lm, ep, dx, meta, wv_smem, lane, wave, 0, nullptr, prof);
  else if (nt == 4) ba_solve_wave_run<4, true>(H, bvec, n, lm, ep, dx, meta, wv_smem, lane, wave, ring_e4, Lglobal, prof);
  else if (nt == 5 && ring_e5 == 0) ba_solve_wave_run<5, false>(H, bvec, n, lm, ep, dx, meta, wv_smem, lane, wave, 0, nullptr, prof);
  else if (nt == 5) ba_solve_wave_run<5, true>(H, bvec, n, lm, ep, dx, meta, wv_smem, lane, wave, ring_e5, Lglobal, prof);
  else ba_solve_general_body<GENERAL_IN_LDS>(H, bvec, n, lm, ep, dx, meta, Lglobal, nullptr, wv_smem);
}

int ba_solve_wave_max_nt(int n) {   // the tallest window whose panel store (or a ring of it) fits LDS for n unknowns (0: none)
  if (n <= 0 || n % 6 != 0 || n / 6 > 64) return 0;
  if (wv_ring_early(n, 5) >= 0) return 5;
  if (wv_ring_early(n, 4) >= 0) return 4;
  if (wv_ring_early(n, 3) == 0) return 3;
  return 0;
}

bool ba_solve_wave_supported(int n) { return ba_solve_wave_max_nt(n) != 0; }

// Lscratch: the workspace's packed-triangle scratch (needed by the fall-back when the system does not fit LDS: n > 199)
int launch_ba_solve_wave(const double *H, const double *b, const int *fpose, int n, double lm, double ep, float *dx, int *meta,
                         double *Lscratch, int *verdict, hipStream_t stream, long long *prof) {
  int max_nt = ba_solve_wave_max_nt(n);
  if (!max_nt || !fpose) return DBA_ERR_UNSUPPORTED;
  const int S = ((n + 15) & ~15) >> 2;
  int ring_e[6] = {0, 0, 0, 0, 0, 0};
  const size_t scratch = Lscratch ? ba_solve_scratch_doubles(n) : 0;
  for (int nt = max_nt; nt >= 3; nt--) {   // the tallest window whose early panels (if any leave LDS) have room in the scratch
    ring_e[nt] = wv_ring_early(n, nt);
    if (ring_e[nt] < 0 || (size_t)ring_e[nt] * (64 * nt) > scratch) max_nt = nt - 1;
  }
  if (max_nt < 3) return DBA_ERR_UNSUPPORTED;
  static DeviceOnce attr_once;
  if (attr_once.needed()) {
    DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_solve_wave_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SOLVE_MAX_LDS_BYTES));
    DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_solve_wave_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SOLVE_MAX_LDS_BYTES));
    attr_once.done();
  }
  size_t wave_lds = 0;
  for (int nt = 3; nt <= max_nt; nt++) wave_lds = std::max(wave_lds, wv_lds_doubles_k(n, nt, S - ring_e[nt]) * sizeof(double));
  const size_t gen_lds = solve_packed_bytes(n) + solve_small_bytes(n);
  if (gen_lds <= (size_t)SOLVE_MAX_LDS_BYTES) {
    hipLaunchKernelGGL(ba_solve_wave_kernel<true>, dim3(1), dim3(WV_THREADS), std::max(wave_lds, gen_lds), stream, H, b, fpose, n, lm,
                       ep, dx, meta, Lscratch, verdict, max_nt, ring_e[4], ring_e[5], prof);
  } else {
    if (!Lscratch) return DBA_ERR_WORKSPACE;
    hipLaunchKernelGGL(ba_solve_wave_kernel<false>, dim3(1), dim3(WV_THREADS), std::max(wave_lds, solve_small_bytes(n)), stream, H, b,
                       fpose, n, lm, ep, dx, meta, Lscratch, verdict, max_nt, ring_e[4], ring_e[5], prof);
  }
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

}  // namespace dba
