// EXPERIMENT (not built into the library): f64-MFMA block-LDL^T solver with 4x4 pivots, LDS-resident tiles, four waves.
// Correct on every case of scratch/solve_tile_test.hip but 82 us at n = 144 against 46 us for ba_solve_tile.hip: with one
// wave per SIMD every instruction on the chain (scalar bookkeeping included) costs ~5 cycles, so the ~600 instructions
// a wave issues per 4-column step dominate; see DESIGN.md section 4.2.
// Damped dense solve of the reduced camera system for sliding-window sizes (n = 6P <= 168, i.e. up to 28 optimised
// poses), float64, one workgroup of FOUR waves (one per SIMD), matrix-core trailing updates, LDS-resident.
//
// Replaces the host-side Eigen LLT / SimplicialLLT of the reference
// (/root/reference/src/droid_kernels.cu:200-218 solveDenseD, :1248-1269 SparseBlock::solve).
//
// The solve is ~1 MFLOP: what costs is the dependent chain per eliminated column and the number of instructions
// ONE wave has to issue along it (measured on MI355X, scratch/lat_bench.hip: 12 cycles per dependent f64 FMA,
// ~9.5 cycles between two f64 ops of a lone wave, 82 cycles for v_rcp_f64 + two Newton steps, ~200 cycles for
// LDS write -> barrier -> read across four waves).  Design:
//   * the system augmented with the right-hand side as an extra row lives in LDS as 16x16 tiles of its lower
//     triangle (55 tiles at n = 144);
//   * elimination is a block LDL^T with 4x4 pivots, in place: per step ONE barrier, every wave inverts the 4x4
//     pivot redundantly (the four waves sit on four SIMDs, so the redundancy costs no wall time) and applies the
//     rank-4 update to its share of the tiles inside the skyline with a single v_mfma_f64_16x16x4_f64 each
//     (A = -R_I, the 16x4 panel rows; B = P^-1 R_K^T); the tile operands are requested before the inversion and
//     arrive while it runs;
//   * eliminated columns are never written again (stores into the pivot's own column tile are masked), so the
//     raw panels are still in place for the block back-substitution (D L^T x = y with y the eliminated
//     right-hand-side row), which one wave runs with v_readlane broadcasts;
//   * a row tile's first non-zero column tile (the skyline) is found once after the load; fill-in cannot leave
//     it, so the set of tiles each column tile's elimination touches is a bit mask computed once.  A
//     sliding-window system is block-banded: ~6 of the 55 tiles are touched per step.
// n is padded to a multiple of 4 with identity rows (n = 6P is 2 mod 4 for odd P).
#include "ba_kernels.h"

#include <type_traits>

namespace dba {

constexpr int MF_WAVES = 4;
constexpr int MF_THREADS = 64 * MF_WAVES;
constexpr int MF_MAX_TILES = 128;  // two 64-bit masks
constexpr int MF_PITCH = 18;       // doubles per tile row (16 + 2: spreads the operand reads over the banks)

typedef double mf_d4 __attribute__((ext_vector_type(4)));
typedef double mf_d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double mf_rcp(double d) {
  double y = __builtin_amdgcn_rcp(d);
  double e = fma(-d, y, 1.0);
  y = fma(y, e, y);
  e = fma(-d, y, 1.0);
  y = fma(y, e, y);
  return y;
}

__device__ __forceinline__ double mf_readlane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// symmetric 4x4 inverse through 2x2 blocks; ok = the block is positive definite
struct MfPinv {
  double p[4][4];
  bool ok;
};

__device__ __forceinline__ MfPinv mf_invert(const double (&P)[4][4]) {  // lower triangle of P is read
  MfPinv R;
  const double a = P[0][0], b = P[1][0], c = P[1][1];
  const double d = P[2][0], e = P[2][1], f = P[3][0], g = P[3][1];
  const double h = P[2][2], i = P[3][2], j = P[3][3];
  const double detA = fma(-b, b, a * c);
  const bool okA = (a > 0.0) && (detA > 0.0);
  const double iA = okA ? mf_rcp(detA) : 0.0;
  const double a00 = c * iA, a01 = -b * iA, a11 = a * iA;
  // X = B A^-1
  const double x00 = fma(e, a01, d * a00), x01 = fma(e, a11, d * a01);
  const double x10 = fma(g, a01, f * a00), x11 = fma(g, a11, f * a01);
  // S = C - X B^T
  const double s00 = h - fma(x01, e, x00 * d);
  const double s01 = i - fma(x01, g, x00 * f);
  const double s11 = j - fma(x11, g, x10 * f);
  const double detS = fma(-s01, s01, s00 * s11);
  const bool okS = (s00 > 0.0) && (detS > 0.0);
  R.ok = okA && okS;
  const double iS = R.ok ? mf_rcp(detS) : 0.0;
  const double t00 = s11 * iS, t01 = -s01 * iS, t11 = s00 * iS;  // S^-1
  // Y = S^-1 X
  const double y00 = fma(t01, x10, t00 * x00), y01 = fma(t01, x11, t00 * x01);
  const double y10 = fma(t11, x10, t01 * x00), y11 = fma(t11, x11, t01 * x01);
  const double k = R.ok ? 1.0 : 0.0;  // a failed block contributes nothing (the verdict is reported separately)
  R.p[0][0] = k * (a00 + fma(x10, y10, x00 * y00));
  R.p[0][1] = R.p[1][0] = k * (a01 + fma(x10, y11, x00 * y01));
  R.p[1][1] = k * (a11 + fma(x11, y11, x01 * y01));
  R.p[2][0] = R.p[0][2] = -y00;
  R.p[2][1] = R.p[1][2] = -y01;
  R.p[3][0] = R.p[0][3] = -y10;
  R.p[3][1] = R.p[1][3] = -y11;
  R.p[2][2] = t00;
  R.p[2][3] = R.p[3][2] = t01;
  R.p[3][3] = t11;
  return R;
}

// tile (TI, TJ), TJ <= TI: index in the row-major enumeration of the lower triangle
__device__ __host__ __forceinline__ int mf_tile(int TI, int TJ) { return TI * (TI + 1) / 2 + TJ; }

struct MfShape {
  int np, nsteps, TR, TC, ntiles, ts;  // ts: doubles per tile
};

__device__ __host__ __forceinline__ MfShape mf_shape(int n, int ts) {
  MfShape S;
  S.np = (n + 3) & ~3;  // padded with identity rows
  S.nsteps = S.np >> 2;
  S.TR = (S.np + 1 + 15) >> 4;  // row tiles (incl. the right-hand-side row np)
  S.TC = (S.np + 15) >> 4;      // column tiles
  S.ntiles = S.TC * (S.TC + 1) / 2 + ((S.TR > S.TC) ? S.TC : 0);
  S.ts = ts;
  return S;
}

__global__ __launch_bounds__(MF_THREADS, 1) void ba_solve_mfma_kernel(const double *__restrict__ H,
                                                                      const double *__restrict__ bvec, int n,
                                                                      int ts, double lm, double ep,
                                                                      float *__restrict__ dx, int *__restrict__ meta
#ifdef PROFILE_SOLVE
                                                                      , long long *__restrict__ prof
#endif
                                                                      ) {
#ifdef PROFILE_SOLVE
#define MPROF(slot) do { if (threadIdx.x == 0) { long long t_ = wall_clock64(); prof[slot] += t_ - tprev_; tprev_ = t_; } } while (0)
  long long tprev_ = wall_clock64();
#else
#define MPROF(slot)
#endif
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const MfShape S = mf_shape(n, ts);
  const int np = S.np, nsteps = S.nsteps, TR = S.TR, TC = S.TC, ntiles = S.ntiles;
  double *T = smem;                                  // tiles, [ntiles][16][MF_PITCH] (+ padding up to ts)
  double *pinvs = T + (size_t)ntiles * ts;           // 16 doubles per step
  unsigned long long *masks = (unsigned long long *)(pinvs + 16 * nsteps);  // [TC][4]: gt lo/hi, eq lo/hi
  int *first = (int *)(masks + 4 * TC);
  int *fail = first + TR;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;

  if (tid < TR) first[tid] = min(tid, TC - 1);
  if (tid == 0) *fail = 0;
  __syncthreads();

  // ---- load: tile t -> wave t % 4; element (row (lane >> 4) + 4 r, column lane & 15), the C/D layout of the MFMA
  for (int t0 = wave; t0 < ntiles; t0 += 4 * MF_WAVES) {
    double v[4][4];
    int kind[4][4];  // 0: zero, 1: load, 2: load + damping, 3: one
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int t = t0 + q * MF_WAVES;
      int TI = (int)((sqrtf(8.f * t + 1.f) - 1.f) * 0.5f);
      while ((TI + 1) * (TI + 2) / 2 <= t) TI++;
      while (TI * (TI + 1) / 2 > t) TI--;
      const int TJ = t - TI * (TI + 1) / 2;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 16 * TI + lk + 4 * r, col = 16 * TJ + li;
        int kd = 0;
        const double *src = H;
        if (t < ntiles && col < np && row <= np) {
          if (row == np) { if (col < n) { kd = 1; src = bvec + col; } }
          else if (row >= n || col >= n) kd = (row == col) ? 3 : 0;  // identity padding
          else { kd = (row == col) ? 2 : 1; src = H + (size_t)max(row, col) * n + min(row, col); }  // mirrored upper half
        }
        kind[q][r] = kd;
        v[q][r] = *src;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int t = t0 + q * MF_WAVES;
      if (t >= ntiles) continue;
      int TI = (int)((sqrtf(8.f * t + 1.f) - 1.f) * 0.5f);
      while ((TI + 1) * (TI + 2) / 2 <= t) TI++;
      while (TI * (TI + 1) / 2 > t) TI--;
      const int TJ = t - TI * (TI + 1) / 2;
      bool nz = false;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        double x = v[q][r];
        if (kind[q][r] == 2) x += ep + lm * x;  // damping (:1252-1253)
        x = (kind[q][r] == 0) ? 0.0 : (kind[q][r] == 3) ? 1.0 : x;
        T[(size_t)t * ts + (lk + 4 * r) * MF_PITCH + li] = x;
        nz |= (x != 0.0);
      }
      if (__ballot(nz) != 0ull && lane == 0) atomicMin(&first[TI], TJ);
    }
  }
  __syncthreads();
  // per column tile: the tiles its elimination touches (skyline), as bit masks over the tile index
  if (tid < 64) {
    for (int TJc = 0; TJc < TC; TJc++) {
      unsigned long long gt[2], eq[2];
#pragma unroll
      for (int w = 0; w < 2; w++) {
        const int t = lane + 64 * w;
        int TI = (int)((sqrtf(8.f * t + 1.f) - 1.f) * 0.5f);
        while ((TI + 1) * (TI + 2) / 2 <= t) TI++;
        while (TI * (TI + 1) / 2 > t) TI--;
        const int TJ = t - TI * (TI + 1) / 2;
        const bool in = (t < ntiles) && (TJc >= max(first[min(TI, TR - 1)], first[min(TJ, TR - 1)]));
        gt[w] = __ballot(in && TJ > TJc);
        eq[w] = __ballot(in && TJ == TJc);
      }
      if (lane == 0) {
        masks[4 * TJc + 0] = gt[0], masks[4 * TJc + 1] = gt[1];
        masks[4 * TJc + 2] = eq[0], masks[4 * TJc + 3] = eq[1];
      }
    }
  }
  MPROF(0);

  // lane l keeps (TI | TJ << 8) of the tiles l and l + 64: a tile's coordinates are one v_readlane away
  int tinfo[2];
#pragma unroll
  for (int w = 0; w < 2; w++) {
    const int t = lane + 64 * w;
    int TI = (int)((sqrtf(8.f * t + 1.f) - 1.f) * 0.5f);
    while ((TI + 1) * (TI + 2) / 2 <= t) TI++;
    while (TI * (TI + 1) / 2 > t) TI--;
    tinfo[w] = TI | ((t - TI * (TI + 1) / 2) << 8);
  }
  __syncthreads();  // masks
  unsigned long long mg0 = 0, mg1 = 0, me0 = 0, me1 = 0;

  // ---- factorisation: block LDL^T, 4x4 pivots, in place, one barrier per four columns
  for (int s = 0; s < nsteps; s++) {
    const int c0 = 4 * s, TJc = c0 >> 4, lc0 = c0 & 15;
    if (lc0 == 0) {  // tile sets of this column tile (wave-uniform)
      mg0 = masks[4 * TJc], mg1 = masks[4 * TJc + 1], me0 = masks[4 * TJc + 2], me1 = masks[4 * TJc + 3];
    }
    __syncthreads();
    unsigned long long m0 = mg0, m1 = mg1;
    if (lc0 < 12) m0 |= me0, m1 |= me1;
    // this wave takes every fourth active tile
    struct Job { int t, TJ; double a; mf_d2 r0, r1; mf_d4 c; double *cp; };
    int seen = 0;
    auto next_job = [&](Job &j) {  // uniform; false when the wave has no further tile
      while (m0 | m1) {
        int t;
        if (m0) { t = __builtin_ctzll(m0); m0 &= m0 - 1; }
        else { t = 64 + __builtin_ctzll(m1); m1 &= m1 - 1; }
        if (((seen++) & (MF_WAVES - 1)) != wave) continue;
        const int info = (t < 64) ? __builtin_amdgcn_readlane(tinfo[0], t) : __builtin_amdgcn_readlane(tinfo[1], t - 64);
        const int TI = info & 0xff, TJ = info >> 8;
        const double *pi = T + (size_t)mf_tile(TI, TJc) * ts + lc0;   // panel rows of the tile's rows
        const double *pj = T + (size_t)mf_tile(TJ, TJc) * ts + lc0;   // panel rows of the tile's columns
        j.t = t, j.TJ = TJ;
        j.a = pi[li * MF_PITCH + lk];
        j.r0 = *(const mf_d2 *)(pj + li * MF_PITCH);
        j.r1 = *(const mf_d2 *)(pj + li * MF_PITCH + 2);
        j.cp = T + (size_t)t * ts + lk * MF_PITCH + li;
#pragma unroll
        for (int r = 0; r < 4; r++) j.c[r] = j.cp[4 * r * MF_PITCH];
        return true;
      }
      return false;
    };
    Job j0, j1;
#ifdef MF_SKIP_JOBS
    const bool h0 = false, h1 = false;
#else
    const bool h0 = next_job(j0);
    const bool h1 = h0 && next_job(j1);
#endif
    // pivot block (the diagonal tile keeps a mirrored upper half; the lower one is read)
    const double *pp = T + (size_t)mf_tile(TJc, TJc) * ts + lc0 * MF_PITCH + lc0;
    double Pv[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++) {
      const mf_d2 lo = *(const mf_d2 *)(pp + a * MF_PITCH), hi = *(const mf_d2 *)(pp + a * MF_PITCH + 2);
      Pv[a][0] = lo.x, Pv[a][1] = lo.y, Pv[a][2] = hi.x, Pv[a][3] = hi.y;
    }
#ifdef MF_SKIP_PINV
    MfPinv PI; for (int a_ = 0; a_ < 4; a_++) for (int b_ = 0; b_ < 4; b_++) PI.p[a_][b_] = Pv[a_][b_]; PI.ok = true;
#else
    const MfPinv PI = mf_invert(Pv);
#endif
    if (!PI.ok && tid == 0) *fail = 1;
    // this lane's row of P^-1 (k = lane >> 4)
    double pk[4];
#pragma unroll
    for (int q = 0; q < 4; q++) pk[q] = (lk == 0) ? PI.p[0][q] : (lk == 1) ? PI.p[1][q] : (lk == 2) ? PI.p[2][q] : PI.p[3][q];
    if (wave == MF_WAVES - 1 && li < 4)  // kept for the back-substitution
      pinvs[16 * s + 4 * lk + li] = (li == 0) ? pk[0] : (li == 1) ? pk[1] : (li == 2) ? pk[2] : pk[3];
    auto apply = [&](Job &j) {
      const double u = fma(pk[3], j.r1.y, fma(pk[2], j.r1.x, fma(pk[1], j.r0.y, pk[0] * j.r0.x)));
      const mf_d4 c = __builtin_amdgcn_mfma_f64_16x16x4f64(-j.a, u, j.c, 0, 0, 0);
      // eliminated columns keep their raw values: in the pivot's own column tile only columns right of it move
      if (j.TJ != TJc || li > lc0 + 3) {
#pragma unroll
        for (int r = 0; r < 4; r++) j.cp[4 * r * MF_PITCH] = c[r];
      }
    };
#ifndef MF_SKIP_APPLY
    if (h0) apply(j0);
    if (h1) apply(j1);
#endif
    if (h1) {
      while (next_job(j0)) {  // further tiles (dense systems)
        const bool more = next_job(j1);
        apply(j0);
        if (more) apply(j1); else break;
      }
    }
  }
  __syncthreads();
  MPROF(1);

  // ---- L^T-side block substitution by wave 0: lane l holds t_j for the columns j = l, l + 64, l + 128
  if (wave == 0) {
    double t[3], xo[3] = {0.0, 0.0, 0.0};
    int cb[3];  // C(i, j) = T[tile(i >> 4, j >> 4) * ts + (i & 15) * MF_PITCH + (j & 15)]
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const int j = lane + 64 * r;
      const int jc = (j < np) ? j : 0;
      cb[r] = (jc >> 4) * ts + (jc & 15);
      t[r] = (j < np) ? T[(size_t)mf_tile(np >> 4, 0) * ts + cb[r] + (np & 15) * MF_PITCH] : 0.0;
    }
    auto sweep = [&](auto rc) {
      constexpr int r0 = decltype(rc)::value;
      constexpr int R = r0 + 1;
      const int shi = min(nsteps, 16 * (r0 + 1)) - 1, slo = 16 * r0;
      struct Ops { double l[R][4], p[16]; };
      auto fetch = [&](int s, Ops &o) {
        const int sc = max(s, 0);
        const double *rowp = T + (size_t)mf_tile((4 * sc) >> 4, 0) * ts + ((4 * sc) & 15) * MF_PITCH;
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
          for (int k = 0; k < 4; k++) o.l[r][k] = rowp[cb[r] + k * MF_PITCH];  // rows 4 sc + k of this lane's column
#pragma unroll
        for (int e = 0; e < 16; e++) o.p[e] = pinvs[16 * sc + e];
      };
      auto pin = [&](Ops &o) {  // keeps the prefetch where it was issued
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
          for (int k = 0; k < 4; k++) asm volatile("" : "+v"(o.l[r][k]));
#pragma unroll
        for (int e = 0; e < 16; e++) asm volatile("" : "+v"(o.p[e]));
      };
      auto solve_step = [&](int s, const Ops &o) {
        const int l0 = (4 * s) & 63;
        double tb[4], x[4];
#pragma unroll
        for (int k = 0; k < 4; k++) tb[k] = mf_readlane(t[r0], l0 + k);
#pragma unroll
        for (int k = 0; k < 4; k++)
          x[k] = fma(o.p[4 * k + 3], tb[3], fma(o.p[4 * k + 2], tb[2], fma(o.p[4 * k + 1], tb[1], o.p[4 * k] * tb[0])));
        // every lane updates: columns at or right of the pivot block receive garbage, but they are finished
#pragma unroll
        for (int r = 0; r < R; r++)
          t[r] = fma(-o.l[r][3], x[3], fma(-o.l[r][2], x[2], fma(-o.l[r][1], x[1], fma(-o.l[r][0], x[0], t[r]))));
        const int d = lane - l0;
        if ((unsigned)d < 4u) xo[r0] = (d == 0) ? x[0] : (d == 1) ? x[1] : (d == 2) ? x[2] : x[3];
      };
      for (int s = shi; s >= slo; s -= 2) {  // two steps per LDS round trip
        Ops o[2];
        fetch(s, o[0]);
        fetch(s - 1, o[1]);
        pin(o[0]);
        pin(o[1]);
        solve_step(s, o[0]);
        if (s - 1 >= slo) solve_step(s - 1, o[1]);
      }
    };
    if (nsteps > 32) sweep(std::integral_constant<int, 2>{});
    if (nsteps > 16) sweep(std::integral_constant<int, 1>{});
    sweep(std::integral_constant<int, 0>{});

    // non-finite results count as failure too; failure => zero update (:1263-1266)
    bool bad = false;
#pragma unroll
    for (int r = 0; r < 3; r++)
      if (lane + 64 * r < n && !isfinite(xo[r])) bad = true;
    const int failed = (*fail != 0) || (__ballot(bad) != 0ull);
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const int j = lane + 64 * r;
      if (j < n) dx[j] = failed ? 0.f : (float)xo[r];
    }
    if (lane == 0) meta[1] = failed;
  }
  MPROF(2);
}

static size_t mfma_lds_bytes(int n, int ts) {
  const MfShape S = mf_shape(n, ts);
  return ((size_t)S.ntiles * ts + 16 * (size_t)S.nsteps) * sizeof(double) + 4 * (size_t)S.TC * sizeof(unsigned long long) +
         ((size_t)S.TR + 4) * sizeof(int);
}

// doubles per tile: 16 more than the 16 x 18 block when LDS allows (the back-substitution's row reads across four
// column tiles then fall into different banks)
static int mfma_tile_doubles(int n) {
  return (mfma_lds_bytes(n, 16 * MF_PITCH + 16) <= (size_t)SOLVE_MAX_LDS_BYTES) ? 16 * MF_PITCH + 16 : 16 * MF_PITCH;
}

bool ba_solve_mfma_supported(int n) {
  if (n <= 0 || n > 188) return false;
  const MfShape S = mf_shape(n, 16 * MF_PITCH);
  return S.ntiles <= MF_MAX_TILES && mfma_lds_bytes(n, 16 * MF_PITCH) <= (size_t)SOLVE_MAX_LDS_BYTES;
}

#ifdef PROFILE_SOLVE
long long *g_mfma_prof;
#endif

int launch_ba_solve_mfma(const double *H, const double *b, int n, double lm, double ep, float *dx, int *meta,
                         hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_solve_mfma_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SOLVE_MAX_LDS_BYTES));
    attr_set = true;
  }
  const int ts = mfma_tile_doubles(n);
#ifdef PROFILE_SOLVE
  hipLaunchKernelGGL(ba_solve_mfma_kernel, dim3(1), dim3(MF_THREADS), mfma_lds_bytes(n, ts), stream, H, b, n, ts, lm, ep,
                     dx, meta, g_mfma_prof);
#else
  hipLaunchKernelGGL(ba_solve_mfma_kernel, dim3(1), dim3(MF_THREADS), mfma_lds_bytes(n, ts), stream, H, b, n, ts, lm, ep,
                     dx, meta);
#endif
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

}  // namespace dba
