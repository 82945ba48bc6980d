// Damped dense solve of the reduced camera system for sliding-window sizes (n = 6P <= 174, i.e. up to 29
// optimised poses), float64, one workgroup, register-resident.
//
// Replaces the host-side Eigen LLT / SimplicialLLT of the reference
// (/root/reference/src/droid_kernels.cu:200-218 solveDenseD, :1248-1269 SparseBlock::solve) for the window
// sizes the tracker actually uses; ba_solve.hip keeps the general-size path.
//
// The solve is latency-bound (1 MFLOP), so the design minimises the length of the dependent chain per
// eliminated column instead of counting flops:
//   * the system, augmented with the right-hand side as an extra row, is cut into 4x4 tiles; every tile of the
//     lower triangle lives in the registers of ONE thread for the whole factorisation (703 threads at n = 144);
//   * elimination is a block LDL^T with 2x2 pivots: per step the owners of a column pair publish their raw
//     values to LDS (one ds_write_b128 per row), ONE barrier, then every thread whose tile is inside the
//     skyline reads the two panel rows it needs, inverts the 2x2 pivot redundantly (one v_rcp_f64 + Newton per
//     TWO columns, no square roots) and applies the rank-2 update to its registers;
//   * the published panels are never rewritten and are exactly what the block back-substitution needs
//     (D L^T x = y with y the eliminated right-hand-side row), which one wave runs with v_readlane broadcasts;
//   * structure: a row tile's first non-zero column tile (the skyline) is found once after the load; fill-in
//     cannot leave the skyline, so tiles outside it never enter the update and waves without an active tile
//     only meet the barrier.  A sliding-window system is block-banded, so 2-3 of the 11 waves work per step.
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

__global__ __launch_bounds__(TILE_MAX_THREADS) void ba_solve_tile_kernel(const double *__restrict__ H,
                                                                         const double *__restrict__ bvec,
                                                                         int n, double lm, double ep,
                                                                         float *__restrict__ dx,
                                                                         int *__restrict__ meta
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
  const int n1 = n + 1;                 // rows incl. the right-hand side
  const int T = (n1 + 3) >> 2;          // row tiles
  const int KT = (n + 3) >> 2;          // column tiles
  const int npairs = n >> 1;
  double *C = smem;                                   // raw panels, pair-major
  double *pinv = C + pair_off(npairs, T);           // 4 doubles per pair (p00, p01, p11, -)
  int *first = (int *)(pinv + 4 * npairs);            // skyline: first non-zero column tile of a row tile
  int *fail = first + T;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;

  // ---- tile of this thread (row-major over the lower triangle of tiles)
  int I = (int)((sqrtf(8.f * tid + 1.f) - 1.f) * 0.5f);
  while ((I + 1) * (I + 2) / 2 <= tid) I++;
  while (I * (I + 1) / 2 > tid) I--;
  const int K = tid - I * (I + 1) / 2;
  const bool valid = I < T && 4 * K < n;

  if (tid < T) first[tid] = min(tid, KT - 1);
  if (tid == 0) *fail = 0;

  // zero the panel store (padding rows are read as operands) while the tile loads are in flight
  double a[4][4];
  const double *src[4][4];
  bool okm[4][4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int i = 4 * I + r, k = 4 * K + c;
      okm[r][c] = valid && k < n && i <= n;
      const int hi = max(i, k), lo = min(i, k);   // diagonal tiles keep the mirrored upper half
      const double *q = (i == n) ? bvec + k : H + (size_t)hi * n + lo;
      src[r][c] = okm[r][c] ? q : H;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; r++)
#pragma unroll
    for (int c = 0; c < 4; c++) a[r][c] = *src[r][c];
  for (int e = tid; e < pair_off(npairs, T); e += blockDim.x) C[e] = 0.0;
  bool nz = false;
#pragma unroll
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      double v = okm[r][c] ? a[r][c] : 0.0;
      if (4 * I + r == 4 * K + c && 4 * I + r < n) v += ep + lm * v;  // damping (:1252-1253)
      a[r][c] = v;
      nz |= (v != 0.0);
    }
  }
  __syncthreads();
  if (valid && nz) atomicMin(&first[I], K);
  __syncthreads();
  const int sstart = valid ? max(first[I], first[min(K, T - 1)]) : 0x7fffffff;
  TPROF(0);

  // The owner of a diagonal tile inverts the NEXT 2x2 pivot right after updating it (its dependent chain runs
  // under the rest of the thread's FMAs) and publishes the inverse with the panel: the readers get P^-1 with one
  // LDS read instead of each redoing the reciprocal, and the back-substitution finds it in place.
  auto publish_pinv = [&](int sp, double pa, double pb, double pc) {
    const double det = fma(-pb, pb, pa * pc);
    const bool ok = pa > 0.0 && det > 0.0;
    if (!ok) *fail = 1;  // a non-positive pivot: the verdict is collected after the substitution; keep things finite
    const double idet = ok ? rcp_nr(det) : 0.0;
    dbl2 lo;
    lo.x = pc * idet;
    lo.y = -pb * idet;
    *(dbl2 *)(pinv + 4 * sp) = lo;
    pinv[4 * sp + 2] = pa * idet;
  };
  if (valid && I == 0 && K == 0) publish_pinv(0, a[0][0], a[1][0], a[1][1]);

  // ---- factorisation: block LDL^T, 2x2 pivots, one barrier per column pair
  for (int Ks = 0; Ks < KT; Ks++) {
    auto step = [&](auto hc) {
      constexpr int h = decltype(hc)::value;
      const int j0 = 4 * Ks + 2 * h;
      const int s = j0 >> 1;
      double *P = C + pair_off(s, T);  // row 4 Ks first
      if (valid && K == Ks) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          dbl2 v;
          v.x = a[r][2 * h];
          v.y = a[r][2 * h + 1];
          *(dbl2 *)(P + 2 * (4 * (I - Ks) + r)) = v;
        }
      }
      __syncthreads();
      const bool active = Ks >= sstart && (K > Ks || (K == Ks && h == 0));
      if (active) {
        const dbl2 pv = *(const dbl2 *)(pinv + 4 * s);
        const double p00 = pv.x, p01 = pv.y, p11 = pinv[4 * s + 2];
        dbl2 ri[4], rk[4];
#pragma unroll
        for (int r = 0; r < 4; r++) ri[r] = *(const dbl2 *)(P + 2 * (4 * (I - Ks) + r));
#pragma unroll
        for (int c = 0; c < 4; c++) rk[c] = *(const dbl2 *)(P + 2 * (4 * (K - Ks) + c));
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const double u0 = fma(p01, rk[c].y, p00 * rk[c].x);
          const double u1 = fma(p11, rk[c].y, p01 * rk[c].x);
#pragma unroll
          for (int r = 0; r < 4; r++) a[r][c] = fma(-ri[r].y, u1, fma(-ri[r].x, u0, a[r][c]));
        }
      }
      // next pivot: columns j0 + 2, j0 + 3 = the other half of this tile column (h == 0) or the first half of the
      // next one (h == 1); its owner publishes whether or not this step touched the tile (block-diagonal systems)
      constexpr int hn = 1 - h;
      const int Kn = (h == 0) ? Ks : Ks + 1;
      if (valid && I == Kn && K == Kn && 2 * (s + 1) < n)
        publish_pinv(s + 1, a[2 * hn][2 * hn], a[2 * hn + 1][2 * hn], a[2 * hn + 1][2 * hn + 1]);
    };
    step(std::integral_constant<int, 0>{});
    if (4 * Ks + 2 < n) step(std::integral_constant<int, 1>{});
  }
  __syncthreads();
  TPROF(1);
  TPROF(2);

  if (wave == 0) {
    // lane l holds t_j for the columns j = l, l + 64, l + 128; C(i, j) = raw panel value of row i in column j
    auto Cij = [&](int i, int j) { return C[pair_off(j >> 1, T) + 2 * (i - 4 * (j >> 2)) + (j & 1)]; };
    double t[3];
    int cbase[3];  // C(i, j) = C[cbase + 2 i] for this lane's column j (columns past n alias column 0; masked)
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const int j = lane + 64 * r;
      const int jc = (j < n) ? j : 0;
      cbase[r] = pair_off(jc >> 1, T) - 8 * (jc >> 2) + (jc & 1);
      t[r] = (j < n) ? Cij(n, j) : 0.0;
    }
    double xo[3] = {0.0, 0.0, 0.0};  // solution entries of this lane's columns (kept off the dependent chain)
    auto sweep = [&](auto rc) {
      constexpr int r0 = decltype(rc)::value;
      constexpr int R = r0 + 1;
      const int shi = min(npairs, 32 * (r0 + 1)) - 1, slo = 32 * r0;
      struct Ops { double l0[R], l1[R], p[3]; };
      // raw operands of step s: rows 2s, 2s+1 of this lane's columns and the pivot inverse
      auto fetch = [&](int s, Ops &o) {
        const int sc = max(s, 0);
#pragma unroll
        for (int r = 0; r < R; r++) {
          const double *q = C + cbase[r] + 4 * sc;
          o.l0[r] = q[0];
          o.l1[r] = q[2];
        }
        o.p[0] = pinv[4 * sc], o.p[1] = pinv[4 * sc + 1], o.p[2] = pinv[4 * sc + 2];
      };
      auto solve_step = [&](int s, const Ops &o) {
        const int j0 = 2 * s, l0 = j0 & 63;
        const double t0 = readlane_dyn_f64(t[r0], l0), t1 = readlane_dyn_f64(t[r0], l0 + 1);
        const double x0 = fma(o.p[1], t1, o.p[0] * t0), x1 = fma(o.p[2], t1, o.p[1] * t0);
        // every lane updates: columns at or right of the pivot pair receive garbage, but they are finished (their
        // solution sits in xo) and are never read again, so no per-lane masking on the single issuing wave
#pragma unroll
        for (int r = 0; r < R; r++) t[r] = fma(-o.l1[r], x1, fma(-o.l0[r], x0, t[r]));
        xo[r0] = (lane == l0) ? x0 : ((lane == l0 + 1) ? x1 : xo[r0]);
      };
      // keeps a prefetch where it was issued (otherwise the loads sink to their first use)
      auto pin = [&](Ops &o) {
#pragma unroll
        for (int r = 0; r < R; r++) asm volatile("" : "+v"(o.l0[r]), "+v"(o.l1[r]));
        asm volatile("" : "+v"(o.p[0]), "+v"(o.p[1]), "+v"(o.p[2]));
      };
      // operands are fetched four steps at a time: one LDS round trip per four links of the dependent chain
      for (int s = shi; s >= slo; s -= 4) {
        Ops o[4];
#pragma unroll
        for (int q = 0; q < 4; q++) fetch(s - q, o[q]);
#pragma unroll
        for (int q = 0; q < 4; q++) pin(o[q]);
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (s - q >= slo) solve_step(s - q, o[q]);
      }
    };
    if (npairs > 64) sweep(std::integral_constant<int, 2>{});
    if (npairs > 32) sweep(std::integral_constant<int, 1>{});
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
  TPROF(3);
}

static int tile_rows(int n) { return (n + 1 + 3) / 4; }

static size_t tile_lds_bytes(int n) {
  const int np = n / 2;
  return ((size_t)pair_off(np, tile_rows(n)) + 4 * (size_t)np) * sizeof(double) + ((size_t)tile_rows(n) + 4) * sizeof(int);
}

bool ba_solve_tile_supported(int n) {
  if (n <= 0 || (n & 1) || n > 192) return false;
  const int T = tile_rows(n);
  return T * (T + 1) / 2 <= TILE_MAX_THREADS && tile_lds_bytes(n) <= (size_t)SOLVE_MAX_LDS_BYTES;
}

int launch_ba_solve_tile(const double *H, const double *b, int n, double lm, double ep, float *dx, int *meta,
                         hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_solve_tile_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SOLVE_MAX_LDS_BYTES));
    attr_set = true;
  }
  const int T = tile_rows(n);
  const int threads = ((T * (T + 1) / 2 + 63) / 64) * 64;
#ifdef PROFILE_SOLVE
  extern long long *g_tile_prof;
  hipLaunchKernelGGL(ba_solve_tile_kernel, dim3(1), dim3(threads), tile_lds_bytes(n), stream, H, b, n, lm, ep, dx,
                     meta, g_tile_prof);
#else
  hipLaunchKernelGGL(ba_solve_tile_kernel, dim3(1), dim3(threads), tile_lds_bytes(n), stream, H, b, n, lm, ep, dx,
                     meta);
#endif
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

}  // namespace dba
