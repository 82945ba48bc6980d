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
#include <atomic>
#include <type_traits>

#include "ba_solve_admit.h"
#include "ba_solve_general.inc"

namespace dba {

constexpr int WV_PLAN_TAG = 0x5a10;   // splan[0] = tag | window height (0: not admitted)
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
  // kfr > 0 (a front of the two-workgroup solve: WvFront): only the panels of the steps it eliminates are kept
  __device__ WvLayout(double *smem, int n, int ring_e, int kfr = 0) {
    np = (n + 15) & ~15, S = np >> 2;
    E = WRING ? ring_e : 0, K = kfr > 0 ? kfr : S - E;
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

// ---- Two fronts on two workgroups (round 6, systems of 50+ poses).  The chain is the cost of this kernel, and it is as long as the
// system: cut the system into top | separator | bottom such that no top column reaches the bottom part (the separator is one band
// wide), let workgroup 0 eliminate the top part and workgroup 1 the bottom part IN REVERSE ORDER (the same code on the mirrored
// matrix) at the same time, add up what both leave of the separator block, solve that small dense system (redundantly, by the
// same code), and back-substitute each part with the separator's unknowns given.  Each chain is (n - sep) / 8 steps instead of
// n / 4.  tests/wave_solver_model.py::TwoFrontSolver is this arithmetic lane by lane; round 5 built it on ONE workgroup (eight /
// ten waves sharing four SIMDs: slower than one front, scratch/ba_solve_wave_two_fronts.hip) -- two workgroups share nothing but
// a hand-shake through global memory.
struct WvFront {
  int nloc;     // unknowns of the local system: the front's own part, then the separator
  int sel;      // steps it eliminates (own part / 4)
  int sep, a;   // separator unknowns, first local index of the separator (= 4 sel)
  int mirror;   // 0: local index = original index; 1: local index i <-> original n4 - 1 - i
  int n4;       // the system padded to a multiple of 4 (identity)
};

// a tile of the damped, padded LOCAL system of a front (accumulator layout, as wv_load_tile)
__device__ __forceinline__ wv_d4 wv_load_tile_front(const double *__restrict__ H, int n, const WvFront F, int np, double lm, double ep,
                                                    int TI, int TJ, int lane) {
  const int li = lane & 15, lk = lane >> 4;
  wv_d4 t;
  // a tile below the diagonal, inside the local system and inside the original one: the local element (row, col) is
  // H[row n + col], or, mirrored, H[(n4 - 1 - col) n + (n4 - 1 - row)] -- affine either way: four loads off one address (the
  // loader wave has a tile column's time for its tiles: with the general path below it set the fronts' pace)
  if (TI != TJ && 16 * TI + 15 < F.nloc && (!F.mirror || 16 * TJ >= F.n4 - n)) {   // (wave-uniform)
    const ptrdiff_t sr = F.mirror ? -1 : n, sc = F.mirror ? -(ptrdiff_t)n : 1;
    const double *p = H + (F.mirror ? (ptrdiff_t)(F.n4 - 1) * (n + 1) : 0) + (16 * TI + lk) * sr + (16 * TJ + li) * sc;
#pragma unroll
    for (int r = 0; r < 4; r++) t[r] = p[4 * r * sr];
    return t;
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = 16 * TI + lk + 4 * r, col = 16 * TJ + li;
    const int hi = max(row, col), lo = min(row, col);
    // (always a load from inside the matrix, selected afterwards: loads inside branches would be waited for one by one, and the
    // loader wave has a tile column's time for five tiles)
    const int hic = min(hi, F.nloc - 1), loc = min(lo, F.nloc - 1);
    const int oi = F.mirror ? F.n4 - 1 - loc : hic, oj = F.mirror ? F.n4 - 1 - hic : loc;   // original (row >= column)
    const double hv = H[(size_t)min(oi, n - 1) * n + min(oj, n - 1)];
    const double vin = (oi >= n) ? ((oi == oj) ? 1.0 : 0.0) : ((oi == oj) ? fma(lm, hv, hv) + ep : hv);
    t[r] = (hi >= np) ? 0.0 : (hi >= F.nloc) ? ((row == col) ? 1.0 : 0.0) : vin;
  }
  return t;
}
// element (i >= j) of the damped, padded local system (the separator block's own entries, for its assembly)
__device__ __forceinline__ double wv_front_elem(const double *__restrict__ H, int n, const WvFront F, double lm, double ep, int i, int j) {
  const int oi = F.mirror ? F.n4 - 1 - j : i, oj = F.mirror ? F.n4 - 1 - i : j;
  if (oi >= n) return (oi == oj) ? 1.0 : 0.0;
  const double hv = H[(size_t)oi * n + oj];
  return (oi == oj) ? fma(lm, hv, hv) + ep : hv;
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
// FR: a front -- the local system of F, its first F.sel steps only; what is left of the separator block goes to `dump`
// ([sep][sep], lower triangle, local orientation)
template <int NT, bool RING, bool FR = false>
__device__ void ba_solve_wave_factor(int n, double *__restrict__ smem, const double *__restrict__ H, double lm, double ep, int lane,
                                     int wave, int ring_e, long long *__restrict__ prof, const WvFront F = WvFront{},
                                     double *__restrict__ dump = nullptr) {
#ifdef PROFILE_SOLVE
  long long tprev_ = wall_clock64();
#endif
  static_assert(!(FR && RING), "a front keeps its panels in LDS");
  const WvLayout<NT, RING> L(smem, FR ? F.nloc : n, ring_e, FR ? ((F.sel + 4) & ~3) : 0);
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
    if (j <= wave) {
      if constexpr (FR) T[j] = wv_load_tile_front(H, n, F, L.np, lm, ep, wave, j, lane);
      else T[j] = wv_load_tile(H, n, L.np, lm, ep, wave, j, lane);
    }
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
    // (a front stops behind its step F.sel - 1, wherever in the tile column that is; the chain wave's verdict on the pivot blocks
    // it has seen is due then)
    auto front_done = [&](int sdone) {
      if constexpr (FR) {
        if (sdone + 1 == F.sel) {
          if constexpr (R == 0) {
            if (__ballot(!(pmin > 0.0)) != 0ull && lane == 0) *(wv_lds_vint *)L.fail = 1;
          }
          return true;
        }
      }
      return false;
    };
    mid(std::integral_constant<int, 0>{});
    if (front_done(4 * tb)) return true;
    mid(std::integral_constant<int, 1>{});
    if (front_done(4 * tb + 1)) return true;
    mid(std::integral_constant<int, 2>{});
    if (front_done(4 * tb + 2)) return true;
    const int s = 4 * tb + 3;
    const bool more = s + 1 < S;
    if constexpr (R >= 1) {
      if (!more) return true;   // the very last step only concerns the pivot tile (and nobody reads a panel the way back may be replacing)
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
    return FR && (s + 1 == F.sel);   // (the rotation is complete: the window stands at the tile column of the front's first kept step)
  };
  int role = wave;
  for (int tb = 0; tb < TB; tb++) {
    if (role == 0) __builtin_amdgcn_s_setprio(3);
    else if (role == 1) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(0);
    bool done;
    if (role == 0) done = run_column(tb, std::integral_constant<int, 0>{});
    else if (role == 1) done = run_column(tb, std::integral_constant<int, 1>{});
    else if (role == 2 || NT == 3) done = run_column(tb, std::integral_constant<int, 2>{});
    else if (role == 3 || NT == 4) done = run_column(tb, std::integral_constant<int, (NT > 3 ? 3 : 2)>{});
    else done = run_column(tb, std::integral_constant<int, NT - 1>{});
    const bool column_over = !FR || !done || ((F.sel & 3) == 0);
    if (column_over) role = (role == 0) ? NT - 1 : role - 1;
    if (FR && done) break;
  }
  if constexpr (FR) {
    // what is left of the separator block: this wave holds the tiles (role, 0 .. role) of the window, which stands at tile column
    // F.sel >> 2; register r of tile j <-> local row 16 (tbw + role) + lk + 4 r, column 16 (tbw + j) + li
    const int tbw = F.sel >> 2;
#pragma unroll
    for (int j = 0; j < NT; j++) {
      if (j <= role) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int i = 16 * (tbw + role) + lk + 4 * r - F.a, c = 16 * (tbw + j) + li - F.a;
          if (c >= 0 && c <= i && i < F.sep) dump[(size_t)i * F.sep + c] = T[j][r];
        }
      }
    }
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
template <int WNT, bool WRING, bool FR = false>
__device__ void ba_solve_wave_loader(int n, double *__restrict__ smem, const double *__restrict__ H, double lm, double ep, int lane,
                                     int ring_e, const double *__restrict__ spill, const WvFront F = WvFront{}) {
  const WvLayout<WNT, WRING> L(smem, FR ? F.nloc : n, ring_e, FR ? ((F.sel + 4) & ~3) : 0);
  constexpr int PD = WvLayout<WNT, WRING>::PD;
  const int TB = L.S >> 2;
  for (int k = 0; k + 1 < TB; k++) {   // (the rotation behind the last tile column brings nothing in)
    if constexpr (FR) {
      if (4 * k + 4 > F.sel) break;    // (a front's last rotation is the one behind its step F.sel - 1, if that ends a tile column)
    }
    wv_d4 t[WNT];
#pragma unroll
    for (int j = 0; j < WNT; j++) {
      if constexpr (FR) t[j] = wv_load_tile_front(H, n, F, L.np, lm, ep, WNT + k, k + 1 + j, lane);
      else t[j] = wv_load_tile(H, n, L.np, lm, ep, WNT + k, k + 1 + j, lane);
    }
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
// PH 0: the whole solve.  A front (WvFront): PH 1 = the right-hand side behind its F.sel steps, then the separator rows' reduced
// right-hand side -> rhs_out (global) and the verdict so far -> returned; PH 2 = the way back with the separator's unknowns given
// (in BV[F.a ..], put there by the caller), then this front's own part of dx (mirrored for the bottom front)
template <int WNT, bool WRING, int PH = 0>
__device__ bool ba_solve_wave_subst(const double *__restrict__ bvec, int n, float *__restrict__ dx, int *__restrict__ meta,
                                    double *__restrict__ smem, int lane, int ring_e, double *__restrict__ spill,
                                    long long *__restrict__ prof, const WvFront F = WvFront{}, double *__restrict__ rhs_out = nullptr,
                                    bool bad_in = false) {
#ifdef PROFILE_SOLVE
  long long tprev_ = wall_clock64();
#endif
  constexpr bool FR = PH != 0;
  const WvLayout<WNT, WRING> L(smem, FR ? F.nloc : n, ring_e, FR ? ((F.sel + 4) & ~3) : 0);
  constexpr int PR = WvLayout<WNT, WRING>::PR, PD = WvLayout<WNT, WRING>::PD;
  constexpr int XR = PR > 64 ? PR - 64 : 0;   // rows of the window beyond one per lane (16 with five tile rows)
  static_assert(!WRING || PR >= 64, "the copy to scratch takes a panel row per lane");
  const int np = L.np, S = L.S;
  double *const PAN = L.PAN, *const ZST = L.ZST, *const BV = L.BV;
  if constexpr (PH != 2) {
  if constexpr (FR) {
    for (int i = lane; i < np + 80; i += 64) {
      const int oi = F.mirror ? F.n4 - 1 - i : i;   // (negative beyond the local system when mirrored: zero)
      const bool in = i < F.nloc && oi >= 0 && oi < n;
      const double bv = bvec[in ? oi : 0];
      BV[i] = in ? bv : 0.0;
    }
  } else {
    for (int i = lane; i < np + 80; i += 64) {
      const double bv = bvec[min(i, n - 1)];
      BV[i] = (i < n) ? bv : 0.0;
    }
  }
  wv_order();
  const int Sfwd = FR ? F.sel : S;
  for (int s = 0; s < Sfwd; s++) {
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
  }   // PH != 2
  if constexpr (PH == 1) {
    wv_order();
    for (int i = lane; i < F.sep; i += 64) rhs_out[i] = BV[F.a + i];
    return *(wv_lds_vint *)L.fail != 0;
  }
  const bool bad = FR ? bad_in : (*(wv_lds_vint *)L.fail != 0);
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
      if constexpr (FR) {   // steps the front did not eliminate: their unknowns are given (x = z - 0 v), their panels do not exist
        const bool given = sp >= F.sel;
        const double xg = BV[4 * max(sp, 0) + kk];
        o.w01 = given ? wv_d2{0.0, 0.0} : o.w01, o.w23 = given ? wv_d2{0.0, 0.0} : o.w23;
        o.z = given ? xg : o.z;
        o.valid = o.valid && (slot + 16 * (u >> 4) < F.sel);
      }
      const double *ra = o.valid ? rp : rsafe;
#pragma unroll
      for (int m = 0; m < 4; m++) o.r[m] = ra[4 * m];
      if constexpr (FAR) {   // the same rows of x in the panel 16 steps down: 64 rows further into its window
        o.valid2 = (u >= 16) && ((u & 15) <= dmax - 16);
        if constexpr (FR) o.valid2 = o.valid2 && (slot + 16 * (u >> 4) - 16 < F.sel);
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
    auto wcol = [&](int c) {   // row k of W of the column's first step (a front: only panels it has; the others' W reads are discarded)
      if constexpr (FR) return L.pan(min(4 * c, (F.sel - 1) & ~3)) + 4 * kk;
      else return L.pan(4 * c) + 4 * kk;
    };
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
  if constexpr (FR) {
    // this front's part of dx: its own unknowns (the bottom front's mirrored back) and, from the top front, the separator's
    bool nf = false;
    const int nout = F.mirror ? F.a : F.nloc;
    for (int j = lane; j < nout; j += 64) nf |= !isfinite(BV[j]);
    const bool failed = bad || (__ballot(nf) != 0ull);
    for (int j = lane; j < nout; j += 64) {
      const int oj = F.mirror ? F.n4 - 1 - j : j;
      if (oj >= 0 && oj < n) dx[oj] = failed ? 0.f : (float)BV[j];
    }
    return failed;
  } else {
  bool nf = false;
  for (int j = lane; j < n; j += 64) nf |= !isfinite(BV[j]);
  const bool failed = bad || (__ballot(nf) != 0ull);
  for (int j = lane; j < n; j += 64) dx[j] = failed ? 0.f : (float)BV[j];
  if (lane == 0) meta[1] = failed ? 1 : 0;
  WPROF(6);
  return failed;
  }
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

// ---- the two-front solve: plan, hand-shake, separator, way back ---------------------------------------------------------------
constexpr int WV_XCH_STRIDE = 5120;   // doubles per front in the exchange area: [0, 68^2) its separator block, [4700, 4768) its
                                      // separator right-hand side, [4800] its verdict so far; then, per front, the assembled system

// The cut (tests/wave_solver_model.py::split_plan, by every wave for itself): the system padded to n4 = n + (n & 2) unknowns; the
// smallest separator (a multiple of 4) such that no column of the top part reaches the bottom part, the block [a, a + sep) lies
// inside the window a front stops in, and the bottom part -- eliminated in REVERSE order -- passes the window test too.
// Returns 0 or the window height NT in 3..5 both fronts can use; a_t, sep, a_b through the pointers.
__device__ __forceinline__ int wv_front_plan(const int *__restrict__ fpose, int n, int lane, int nt_min, int *a_t, int *sep_o, int *a_b) {
  const int P = n / 6;
  if (!fpose || P > 64 || n != 6 * P || nt_min < 3) return 0;
  int g = (lane < P) ? fpose[lane] : 0x7fffffff;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_down(g, off, 64);
    if (lane + off < 64) g = min(g, o);
  }
  int last = lane;
  for (int p = 0; p < P; p++) {
    const int gp = __builtin_amdgcn_readlane(g, p);
    if (gp <= lane) last = max(last, p);
  }
  const int n4 = n + (n & 2);
  for (int nt = nt_min; nt <= 5; nt++) {
    const int rows = 16 * nt;
    for (int sep = 4; sep <= min(rows - 12, 64); sep += 4) {   // (64: the separator's own dense solve runs in the 64-row window)
      const int at = ((n4 - sep) / 2) & ~3, ab = n4 - sep - at;
      if (at < 16 || ab < 16) break;
      const int qt = (at - 1) / 6;
      if (6 * __shfl(last, qt, 64) + 5 >= at + sep) continue;               // a top column reaches past the separator
      if (4 * ((at / 4) % 4) + sep > rows || 4 * ((ab / 4) % 4) + sep > rows) continue;
      bool ok = true;
      for (int base = 0; base < ab / 4; base += 64) {                        // the bottom part's own window test, reversed order
        const int sb = base + lane;
        const int omin = n4 - 1 - (4 * sb + 3);
        const int gq = __shfl(g, min(max(omin, 0) / 6, P - 1), 64);
        const int first = (omin < n) ? 6 * gq : omin;
        ok = ok && (sb >= ab / 4 || (n4 - 1 - first) <= 16 * (sb >> 2) + rows - 1);
      }
      // ... and the top part's (the whole system need not have passed at this height)
      for (int base = 0; base < at / 4; base += 64) {
        const int st = base + lane, c = 4 * st;
        const int q3 = min(min(c + 3, n - 1) / 6, P - 1);
        const int lastrow = min(6 * __shfl(last, q3, 64) + 5, at + sep - 1);
        ok = ok && (st >= at / 4 || lastrow <= 16 * (st >> 2) + rows - 1);
      }
      if (__ballot(!ok) == 0ull) {
        *a_t = at, *sep_o = sep, *a_b = ab;
        return nt;
      }
    }
  }
  return 0;
}

// LDS of a front, in doubles: its own layout, the separator's solve from its (then idle) ring slot on, the separator's dx / verdict
__device__ __host__ __forceinline__ size_t wv_front_lds_doubles(int nloc, int sel, int nt) {
  const int np = (nloc + 15) & ~15, S = np >> 2, K = (sel + 4) & ~3;
  const size_t front = (size_t)K * (16 * nt * 4) + (size_t)S * 4 + np + 88;   // (without its ring slot)
  return front + wv_lds_doubles_k(64, 4, 16) + 64;
}

template <int NT>
__device__ void ba_solve_wave_run_front(const double *__restrict__ H, const double *__restrict__ bvec, int n, double lm, double ep,
                                        float *__restrict__ dx, int *__restrict__ meta, double *__restrict__ smem, int lane, int wave,
                                        const WvFront F, int front, double *__restrict__ xch) {
  const WvLayout<NT, false> L(smem, F.nloc, 0, (F.sel + 4) & ~3);
  double *const own = xch + (size_t)front * WV_XCH_STRIDE, *const oth = xch + (size_t)(1 - front) * WV_XCH_STRIDE;
  double *const ssep = xch + (size_t)(2 + front) * WV_XCH_STRIDE, *const bsep = ssep + 68 * 68;
  int *const xflag = meta + 8;
  double *const smem2 = L.RING;                                   // the separator's solve: from the front's ring slot on
  double *const tail = smem2 + wv_lds_doubles_k(64, 4, 16);       // [0, 32): its dx (float), [32]: its meta (ints), [40]: this front's words
  float *const dx_sep = reinterpret_cast<float *>(tail);
  int *const meta_sep = reinterpret_cast<int *>(tail + 32);
  int *const words = reinterpret_cast<int *>(tail + 40);         // [0] this front's verdict after its steps, [1] the partner did not show up
#ifdef WV_FRONT_PROF   // scratch builds: wall-clock stamps of the phases, per front (10 ns ticks), behind the exchange area
  long long *const stamps = reinterpret_cast<long long *>(xch + 4 * (size_t)WV_XCH_STRIDE) + 16 * front;
#define FSTAMP(k) do { if (threadIdx.x == 0) stamps[k] = wall_clock64(); } while (0)
#else
#define FSTAMP(k)
#endif
  FSTAMP(0);
  if (threadIdx.x < 16) L.flagW[threadIdx.x] = 0;
  if (threadIdx.x < 8) meta_sep[threadIdx.x] = 0, words[threadIdx.x] = 0;
  __syncthreads();
  // (this launch's generation number was counted at the kernel's entry, ba_solve_wave_fronts_kernel: read where it is needed)
  auto generation = [&]() { return (unsigned)__hip_atomic_load(meta + 26 + front, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  int part = wave;
  if constexpr (NT == 5) part = wave == 2 ? 3 : wave == 3 ? 4 : wave == 4 ? 2 : wave == 5 ? 6 : wave == 6 ? 5 : wave;
  if (part < NT) ba_solve_wave_factor<NT, false, true>(n, smem, H, lm, ep, lane, part, 0, nullptr, F, own);
  else if (part == NT) {
    const bool bad = ba_solve_wave_subst<NT, false, 1>(bvec, n, dx, meta, smem, lane, 0, nullptr, nullptr, F, own + 4700);
    if (lane == 0) words[0] = bad ? 1 : 0, own[4800] = bad ? 1.0 : 0.0;
  } else if (part == NT + 1) ba_solve_wave_loader<NT, false, true>(n, smem, H, lm, ep, lane, 0, nullptr, F);
  __builtin_amdgcn_s_setprio(0);
  // ---- both fronts have left their share of the separator block in the exchange area: meet
  FSTAMP(1);
  __threadfence();
  __syncthreads();
  FSTAMP(2);
  if (threadIdx.x == 0) {
    const unsigned gen = generation();
    __hip_atomic_store(xflag + front, (int)gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(xflag + (1 - front), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != (int)gen) {
      if (wall_clock64() - t0 > 100000000ll) {   // 1 s: the partner workgroup never came (both then give the solve up: zero update)
        words[1] = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  FSTAMP(3);
  const bool lost = words[1] != 0;
  // ---- the separator's system in this front's orientation: own + partner (mirrored) - the block's own entries, which both carry
  {
    const int sep = F.sep;
    // (every thread's entries at once: all their loads in flight together -- one entry after the other this was 5 us)
    constexpr int PER = (64 * 64 + WV_THREADS - 1) / WV_THREADS;
    double va[PER], vb[PER], vc[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const int idx = min((int)threadIdx.x + k * WV_THREADS, sep * sep - 1);
      const int i = idx / sep, j = idx - i * sep;
      const int il = max(i, j), jl = min(i, j);   // (the upper triangle's threads load the mirrored entry: discarded)
      va[k] = own[il * sep + jl];
      vb[k] = oth[(sep - 1 - jl) * sep + (sep - 1 - il)];
      vc[k] = wv_front_elem(H, n, F, lm, ep, F.a + il, F.a + jl);
    }
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const int idx = (int)threadIdx.x + k * WV_THREADS;
      const int i = idx / sep, j = idx - i * sep;
      if (idx < sep * sep && j <= i) ssep[idx] = lost ? ((i == j) ? 1.0 : 0.0) : (va[k] + vb[k]) - vc[k];
    }
    for (int i = threadIdx.x; i < sep; i += blockDim.x) {
      const int oi = F.mirror ? F.n4 - 1 - (F.a + i) : F.a + i;
      const double b0 = (oi >= 0 && oi < n) ? bvec[oi] : 0.0;
      bsep[i] = lost ? 0.0 : (own + 4700)[i] + (oth + 4700)[sep - 1 - i] - b0;
    }
  }
  __threadfence_block();
  __syncthreads();
  FSTAMP(4);
  // ... solved by the one-front code in the 64-row window (dense, up to 64 unknowns), by both workgroups alike
  ba_solve_wave_run<4, false>(ssep, bsep, F.sep, 0.0, 0.0, dx_sep, meta_sep, smem2, lane, wave, 0, nullptr, nullptr);
  __syncthreads();
  FSTAMP(5);
  if (part != NT) return;
  // ---- the way back, the separator's unknowns given
  {
    const WvLayout<4, false> Ls(smem2, F.sep, 0);
    for (int i = lane; i < F.sep; i += 64) L.BV[F.a + i] = Ls.BV[i];
    wv_order();
  }
  const bool bad = lost || words[0] != 0 || oth[4800] != 0.0 || meta_sep[1] != 0;
  const bool failed = ba_solve_wave_subst<NT, false, 2>(bvec, n, dx, meta, smem, lane, 0, nullptr, nullptr, F, nullptr, bad);
#ifdef WV_FRONT_PROF
  if (lane == 0) stamps[6] = wall_clock64();
#endif
  // one verdict for both parts of dx
  bool all_failed = failed;
  if (lane == 0) {
    __threadfence();
    const unsigned gen = generation();
    const int mine = (int)gen | (failed ? 0x10000000 : 0);
    __hip_atomic_store(xflag + 4 + front, mine, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0 = wall_clock64();
    int theirs;
    for (;;) {
      theirs = __hip_atomic_load(xflag + 4 + (1 - front), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      if ((theirs & ~0x10000000) == (int)gen) break;
      if (lost || wall_clock64() - t0 > 100000000ll) {
        theirs = (int)gen | 0x10000000;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    all_failed = failed || (theirs & 0x10000000) != 0;
  }
  all_failed = __builtin_amdgcn_readfirstlane(all_failed ? 1 : 0) != 0;
  if (all_failed && !failed) {   // the partner's part is zero: so is this one
    const int nout = F.mirror ? F.a : F.nloc;
    for (int j = lane; j < nout; j += 64) {
      const int oj = F.mirror ? F.n4 - 1 - j : j;
      if (oj >= 0 && oj < n) dx[oj] = 0.f;
    }
  }
  if (lane == 0) meta[1] = all_failed ? 1 : 0;
#ifdef WV_FRONT_PROF
  if (lane == 0) stamps[7] = wall_clock64();
#endif
}

template <bool GENERAL_IN_LDS>
__global__ __launch_bounds__(WV_THREADS) void ba_solve_wave_kernel(const double *__restrict__ H, const double *__restrict__ bvec,
                                                            const int *__restrict__ fpose, int n, double lm, double ep,
                                                            float *__restrict__ dx, int *__restrict__ meta,
                                                            double *__restrict__ Lglobal, int *__restrict__ verdict, int max_nt,
                                                            int ring_e4, int ring_e5, long long *__restrict__ prof,
                                                            int *__restrict__ splan) {
  extern __shared__ __attribute__((aligned(16))) double wv_smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // the window height this graph's skyline admits: derived once per graph (stage 0 clears the record when the graph changes)
  const int tag = splan ? __builtin_amdgcn_readfirstlane(splan[0]) : 0;
  int nt;
  if ((tag & ~7) == WV_PLAN_TAG) nt = tag & 7;
  else {
    nt = ba_solve_wave_admits(fpose, n, lane, max_nt);
    if (splan && threadIdx.x == 0) splan[1] = -1, splan[0] = WV_PLAN_TAG | nt;   // ([1] = -1: the cut for two fronts is not known)
  }
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

// Two workgroups: workgroup f eliminates front f when the cut exists (wv_front_plan) and a front fits LDS; otherwise workgroup 0
// solves the whole system as the one-front kernel does and workgroup 1 leaves.  Only systems the general fall-back keeps in global
// scratch get here (n > 199), so one instantiation serves.
__global__ __launch_bounds__(WV_THREADS) void ba_solve_wave_fronts_kernel(const double *__restrict__ H, const double *__restrict__ bvec,
                                                                   const int *__restrict__ fpose, int n, double lm, double ep,
                                                                   float *__restrict__ dx, int *__restrict__ meta,
                                                                   double *__restrict__ Lglobal, int *__restrict__ verdict, int max_nt,
                                                                   int ring_e4, int ring_e5, double *__restrict__ xch,
                                                                   int *__restrict__ splan) {
  extern __shared__ __attribute__((aligned(16))) double wv_smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // The hand-shake words carry a per-launch generation number.  It is counted ON THE DEVICE -- a counter per workgroup in the
  // workspace (meta[24], meta[25]: both advance by one per launch, whatever path the launch takes) -- so that a launch replayed
  // from a hipGraph gets a new number too (the skyline kernel's comes as a kernel argument and must not be captured).
  // (one thread counts and leaves the number in meta[26 + workgroup]; the threads that need it read it back behind a barrier)
  if (threadIdx.x == WV_THREADS - 64) {   // (the last wave: off the chain at the start, whatever the window height)
    const unsigned g = (unsigned)atomicAdd(meta + 24 + blockIdx.x, 1) + 1u;
    __hip_atomic_store(meta + 26 + blockIdx.x, (int)((g & 0x0fffffffu) | 0x20000000u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#ifdef WV_FRONT_PROF
  if (threadIdx.x == 0) (reinterpret_cast<long long *>(xch + 4 * (size_t)WV_XCH_STRIDE) + 16 * blockIdx.x)[8] = wall_clock64();
#endif
  int nt, ntf, at = 0, sep = 0, ab = 0;
  const int tag = splan ? __builtin_amdgcn_readfirstlane(splan[0]) : 0;
  const int tag1 = splan ? __builtin_amdgcn_readfirstlane(splan[1]) : -1;
  if ((tag & ~7) == WV_PLAN_TAG && tag1 >= 0) {   // what an earlier solve of this graph found
    nt = tag & 7, ntf = tag1;
    at = __builtin_amdgcn_readfirstlane(splan[2]), sep = __builtin_amdgcn_readfirstlane(splan[3]);
    ab = __builtin_amdgcn_readfirstlane(splan[4]);
  } else {
    nt = ba_solve_wave_admits(fpose, n, lane, max_nt);
    ntf = nt ? wv_front_plan(fpose, n, lane, nt, &at, &sep, &ab) : 0;
    if (ntf && wv_front_lds_doubles(max(at, ab) + sep, max(at, ab) / 4, ntf) * sizeof(double) > (size_t)SOLVE_MAX_LDS_BYTES) ntf = 0;
    if (splan && blockIdx.x == 0 && threadIdx.x == 0) {
      splan[2] = at, splan[3] = sep, splan[4] = ab, splan[1] = ntf;
      __threadfence();
      splan[0] = WV_PLAN_TAG | nt;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    meta[3] = 1;
    meta[4] = ntf ? 1 : 0, meta[5] = at / 4, meta[6] = ab / 4;   // (as the skyline kernel reports its split: taken?, top / bottom steps)
    if (verdict) __hip_atomic_store(verdict, nt ? 1 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (ntf) {
    const int front = (int)blockIdx.x;
    WvFront F;
    F.sep = sep, F.n4 = n + (n & 2), F.mirror = front;
    F.a = front ? ab : at, F.sel = F.a / 4, F.nloc = F.a + sep;
    if (ntf == 3) ba_solve_wave_run_front<3>(H, bvec, n, lm, ep, dx, meta, wv_smem, lane, wave, F, front, xch);
    else if (ntf == 4) ba_solve_wave_run_front<4>(H, bvec, n, lm, ep, dx, meta, wv_smem, lane, wave, F, front, xch);
    else ba_solve_wave_run_front<5>(H, bvec, n, lm, ep, dx, meta, wv_smem, lane, wave, F, front, xch);
    return;
  }
  if (blockIdx.x != 0) return;
  if (nt == 3) ba_solve_wave_run<3, false>(H, bvec, n, lm, ep, dx, meta, wv_smem, lane, wave, 0, nullptr, nullptr);
  else if (nt == 4 && ring_e4 == 0) ba_solve_wave_run<4, false>(H, bvec, n, lm, ep, dx, meta, wv_smem, lane, wave, 0, nullptr, nullptr);
  else if (nt == 4) ba_solve_wave_run<4, true>(H, bvec, n, lm, ep, dx, meta, wv_smem, lane, wave, ring_e4, Lglobal, nullptr);
  else if (nt == 5 && ring_e5 == 0) ba_solve_wave_run<5, false>(H, bvec, n, lm, ep, dx, meta, wv_smem, lane, wave, 0, nullptr, nullptr);
  else if (nt == 5) ba_solve_wave_run<5, true>(H, bvec, n, lm, ep, dx, meta, wv_smem, lane, wave, ring_e5, Lglobal, nullptr);
  else ba_solve_general_body<false>(H, bvec, n, lm, ep, dx, meta, Lglobal, nullptr, wv_smem);
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
                         double *Lscratch, int *verdict, hipStream_t stream, long long *prof, int *splan) {
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
  // two fronts on two workgroups: systems of 50 poses and more (DBA_SOLVE_FRONTS=0 keeps one front, =1 also tries it from 34 poses;
  // the exchange area lies behind the early panels' room in the scratch)
  static const int fronts_mode = [] { const char *e = getenv("DBA_SOLVE_FRONTS"); return e ? atoi(e) : -1; }();
  const int fronts_min_n = fronts_mode == 1 ? 204 : 270;
  const size_t xch_off = 16384;   // doubles: past the largest ring's early panels (40 x 320)
  if (fronts_mode != 0 && !prof && n >= fronts_min_n && gen_lds > (size_t)SOLVE_MAX_LDS_BYTES && Lscratch &&
      scratch >= xch_off + 4 * (size_t)WV_XCH_STRIDE) {
    static DeviceOnce attr2_once;
    if (attr2_once.needed()) {
      DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_solve_wave_fronts_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, SOLVE_MAX_LDS_BYTES));
      attr2_once.done();
    }
    hipLaunchKernelGGL(ba_solve_wave_fronts_kernel, dim3(2), dim3(WV_THREADS), (size_t)SOLVE_MAX_LDS_BYTES, stream, H, b, fpose, n, lm, ep,
                       dx, meta, Lscratch, verdict, max_nt, ring_e[4], ring_e[5], Lscratch + xch_off, splan);
    DBA_LAUNCH_CHECK();
    return DBA_OK;
  }
  if (gen_lds <= (size_t)SOLVE_MAX_LDS_BYTES) {
    hipLaunchKernelGGL(ba_solve_wave_kernel<true>, dim3(1), dim3(WV_THREADS), std::max(wave_lds, gen_lds), stream, H, b, fpose, n, lm,
                       ep, dx, meta, Lscratch, verdict, max_nt, ring_e[4], ring_e[5], prof, splan);
  } else {
    if (!Lscratch) return DBA_ERR_WORKSPACE;
    hipLaunchKernelGGL(ba_solve_wave_kernel<false>, dim3(1), dim3(WV_THREADS), std::max(wave_lds, solve_small_bytes(n)), stream, H, b,
                       fpose, n, lm, ep, dx, meta, Lscratch, verdict, max_nt, ring_e[4], ring_e[5], prof, splan);
  }
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

}  // namespace dba
