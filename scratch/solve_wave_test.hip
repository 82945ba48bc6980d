// scratch: correctness + timing of the one-wave window solver (csrc/ba_solve_wave.hip) vs host Cholesky and the tile kernel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstdlib>
#define PROFILE_SOLVE 1
namespace dba { long long *g_tile_prof; }
#include "../dba-fusion_amd/csrc/ba_solve_tile.hip"
#ifdef TWO_FRONTS   // the shelved eight-wave / two-front variant (profiles/SOLVER_NOTES.md): hipcc -DTWO_FRONTS -I../dba-fusion_amd/csrc
#include "ba_solve_wave_two_fronts.hip"
#else
#include "../dba-fusion_amd/csrc/ba_solve_wave.hip"
#endif
namespace dba { void set_last_error(const char*, hipError_t) {} size_t ba_solve_scratch_doubles(int n) { return (size_t)(n + 1) * (n + 2) / 2; } }
static bool host_solve(std::vector<double> A, std::vector<double> b, int n, std::vector<double>& x) {
  for (int j = 0; j < n; j++) {
    double d = A[j*n+j]; for (int k = 0; k < j; k++) d -= A[j*n+k]*A[j*n+k];
    if (!(d > 0)) return false; d = std::sqrt(d); A[j*n+j] = d;
    for (int i = j+1; i < n; i++) { double s = A[i*n+j]; for (int k = 0; k < j; k++) s -= A[i*n+k]*A[j*n+k]; A[i*n+j] = s/d; }
  }
  for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= A[i*n+k]*b[k]; b[i] = s/A[i*n+i]; }
  for (int i = n-1; i >= 0; i--) { double s = b[i]; for (int k = i+1; k < n; k++) s -= A[k*n+i]*b[k]; b[i] = s/A[i*n+i]; }
  x = b; return true;
}
static long long *g_wprof;
static double *gscr;
// P poses, pose p coupled with p-w..p (+ one extra pair), spd: make one diagonal entry negative otherwise
int run(int P, int w, bool spd, bool timeit, int ex_p = -1, int ex_q = -1) {
  const int n = 6 * P;
  std::vector<double> H(n*n, 0.0), b(n); std::vector<int> fpose(P);
  srand(P*131 + w*7 + 1);
  auto rnd = [] { return ((rand() % 2001) - 1000) / 1000.0; };
  for (int p = 0; p < P; p++) {
    fpose[p] = p;
    for (int q = 0; q <= p; q++) {
      // ex_p == -2: the literal skyline of the reduced system of BASELINE's 64-KF / 512-edge graph (|i - j| <= 4 plus (i, i + 5) for
      // i < 10, frame 0 fixed; two poses are coupled when they see the same source frame: 8 poses wide, 9-10 among the first 18)
      const int lit = p <= 10 ? 0 : p <= 13 ? p - 10 : p <= 17 ? p - 9 : p - 8;
      const bool on = (p - q <= w) || (p == ex_p && q == ex_q) || (ex_p == -2 && q >= lit);
      if (!on) continue;
      if (q < fpose[p]) fpose[p] = q;
      for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) {
        if (p == q && c > a) continue;
        const double v = 0.3 * rnd();
        H[(6*p+a)*n + 6*q+c] = v; H[(6*q+c)*n + 6*p+a] = v;
      }
    }
  }
  for (int i = 0; i < n; i++) { double s = 0; for (int j = 0; j < n; j++) s += std::fabs(H[i*n+j]); H[i*n+i] = spd ? s + 1.0 + (rand()%100)/50.0 : ((i == n/2) ? -1.0 : s + 1.0); b[i] = std::sin(i*1.3); }
  const double lm = 1e-4, ep = 0.1;
  std::vector<double> Hd = H; for (int i = 0; i < n; i++) Hd[i*n+i] += ep + lm*Hd[i*n+i];
  std::vector<double> xr; bool ok = host_solve(Hd, b, n, xr);
  // the device reads the lower triangle only: poison the upper one
  std::vector<double> Hl = H; for (int i = 0; i < n; i++) for (int j = i+1; j < n; j++) Hl[i*n+j] = 1e300;
  double *dH, *db; float* dx; int* meta; int *dfp;
  hipMalloc(&dH, n*n*8); hipMalloc(&db, n*8); hipMalloc(&dx, n*4); hipMalloc(&meta, 256); hipMalloc(&dfp, P*4);
  hipMemcpy(dH, Hl.data(), n*n*8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n*8, hipMemcpyHostToDevice);
  hipMemcpy(dfp, fpose.data(), P*4, hipMemcpyHostToDevice); hipMemset(meta, 0, 256); hipMemset(dx, 0xff, n*4);
  int rc = dba::launch_ba_solve_wave(dH, db, dfp, n, lm, ep, dx, meta, gscr, nullptr, 0, nullptr, getenv("HARNESS_NO_PLAN_CACHE") ? nullptr : meta + 16);
  hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess || rc) { printf("P=%d launch error %s rc=%d\n", P, hipGetErrorString(e), rc); return 1; }
  std::vector<float> x(n); int hm[8]; hipMemcpy(x.data(), dx, n*4, hipMemcpyDeviceToHost); hipMemcpy(hm, meta, 32, hipMemcpyDeviceToHost);

  double maxe = 0, maxx = 0; for (int i = 0; i < n; i++) { double r = ok ? xr[i] : 0.0; maxe = fmax(maxe, fabs(x[i]-r)); maxx = fmax(maxx, fabs(r)); }
  printf("wave P=%2d n=%3d w=%d extra=(%d,%d) spd=%d host_ok=%d dev_failed=%d max|x|=%.3e max err=%.3e fronts=(%d,%d,%d) %s\n", P, n, w, ex_p, ex_q, spd, ok, hm[1], maxx, maxe, hm[4], hm[5], hm[6], (maxe <= 2e-7*fmax(maxx,1e-30)+1e-30 && hm[1] == !ok) ? "OK" : "MISMATCH");
  if (getenv("HARNESS_DUMP")) { for (int i = 0; i < n; i++) if (fabs(x[i] - (ok ? xr[i] : 0.0)) > 1e-6) printf("    x[%3d] dev % .6e ref % .6e\n", i, x[i], ok ? xr[i] : 0.0); }
  if (timeit) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // full matrix for the tile kernel (it reads the lower triangle too, mirrors the diagonal tiles)
    hipMemset(g_wprof, 0, 256);
    for (int mode = 0; mode < 2; mode++) {
      if (mode == 1 && !dba::ba_solve_tile_supported(n)) continue;
      hipEventRecord(e0);
      for (int it = 0; it < 200; it++) { if (mode == 0) dba::launch_ba_solve_wave(dH, db, dfp, n, lm, ep, dx, meta, gscr, nullptr, 0, nullptr, getenv("HARNESS_NO_PLAN_CACHE") ? nullptr : meta + 16); else dba::launch_ba_solve_tile(dH, db, n, lm, ep, dx, meta, 0); }
      hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("   %s: %.2f us per solve\n", mode == 0 ? "wave" : "tile", ms*1000/200);
      if (mode == 0) {
#ifdef WV_FRONT_PROF
  if (hm[4]) { long long st[32]; hipMemcpy(st, gscr + 16384 + 4 * 5120, sizeof(st), hipMemcpyDeviceToHost);
    for (int f = 0; f < 2; f++) { long long *t = st + 16 * f; printf("   front %d (kernel entry -> plan done %.2f us; us from there; start offset %.2f): steps done %.2f | fence+barrier %.2f | met %.2f | assembled %.2f | separator solved %.2f | back %.2f | verdict %.2f\n", f, (t[0] - t[8]) * 0.01, (t[0] - st[0]) * 0.01, (t[1]-t[0])*0.01, (t[2]-t[0])*0.01, (t[3]-t[0])*0.01, (t[4]-t[0])*0.01, (t[5]-t[0])*0.01, (t[6]-t[0])*0.01, (t[7]-t[0])*0.01); } }
#endif
      }

      if (mode == 0 && getenv("HARNESS_STEPS_OLD")) { long long hp[16]; hipMemcpy(hp, g_wprof, 128, hipMemcpyDeviceToHost); const double S_ = ((n + 15) / 16 * 4) * 200.0; printf("   wave step phases, shader cycles per step (drained at every mark): sync %.0f reads %.0f inverse %.0f operands+mfma %.0f W+rhs %.0f rotate %.0f extract %.0f\n", hp[8]/S_, hp[9]/S_, hp[10]/S_, hp[11]/S_, hp[12]/S_, hp[13]/S_, hp[14]/S_); }
      if (mode == 0) { for (int it = 0; it < 200; it++) dba::launch_ba_solve_wave(dH, db, dfp, n, lm, ep, dx, meta, gscr, nullptr, 0, g_wprof); (void)hipDeviceSynchronize(); long long hp[32]; hipMemcpy(hp, g_wprof, 256, hipMemcpyDeviceToHost); auto u = [&](int i) { return hp[i]/200.0/100; };
#ifdef TWO_FRONTS
        printf("   stages us (wall clock per wave; one front: top only): top factor load %.2f w0 %.2f w1 %.2f w2 %.2f | bottom factor load %.2f w0 %.2f w1 %.2f w2 %.2f | separator factor load %.2f w0 %.2f w1 %.2f w2 %.2f\n", u(0)/3, u(1), u(2), u(3), u(8)/3, u(9), u(10), u(11), u(12)/3, u(13), u(14), u(15));
        printf("          top subst: init+forward %.2f separator (wait, forward, backward) %.2f backward %.2f wait for bottom + store %.2f | bottom subst: init+forward %.2f wait for the separator %.2f backward %.2f\n", u(4), u(5), u(7), u(6), u(20), u(21), u(22)); }
#else
        printf("   stages us (wall clock per wave): [factor waves] load + first panel (sum) %.2f factor w0 %.2f w1 %.2f w2 %.2f w3 %.2f w4 %.2f | [substitution wave] init + forward (behind the factorisation) %.2f backward %.2f verdict + store %.2f\n", u(0), u(1), u(2), u(3), u(8), u(9), u(4), u(5), u(6)); }
#endif
    }
  }
  hipFree(dH); hipFree(db); hipFree(dx); hipFree(meta); hipFree(dfp);
  return 0;
}
int main() {
  hipMalloc(&dba::g_tile_prof, 2048 + (1 << 20)); hipMemset(dba::g_tile_prof, 0, 2048);
  (void)hipMalloc(&g_wprof, 256); (void)hipMemset(g_wprof, 0, 256); hipMalloc(&gscr, 8 << 20);
  if (getenv("HARNESS_ONLY")) { int P = 40, w = 4; sscanf(getenv("HARNESS_ONLY"), "%d %d", &P, &w); return run(P, w, true, false); }   // one cold solve and out
  run(24, 4, true, true); run(24, 3, true, true); run(25, 4, true, true); run(24, 2, true, false); run(24, 1, true, false); run(24, 0, true, false);
  run(8, 4, true, false); run(3, 2, true, false); run(2, 1, true, false); run(1, 0, true, false); run(29, 4, true, true); run(16, 3, true, false);
  run(63, 4, true, true); run(40, 4, true, true); run(64, 3, true, false);
  run(24, 5, true, true); run(24, 6, true, true); run(24, 7, true, false); run(24, 23, true, false);   // wider bands: NT = 4, then not admitted
  // round 6: the ring of panels (64-row window beyond 48 poses)
  run(63, 4, true, true, -2); run(63, 8, true, true); run(63, 9, true, true); run(63, 10, true, true); run(64, 10, true, true);
  run(24, 8, true, true); run(24, 10, true, true); run(37, 9, true, true); run(38, 10, true, true); run(45, 8, true, false); run(52, 10, true, false);
  run(60, 9, true, true); run(59, 10, true, false); run(63, 10, false, false); run(30, 8, false, false); run(63, 11, true, false);
  run(63, 5, true, true); run(63, 6, true, true); run(63, 7, true, true); run(64, 7, true, true);
  run(49, 5, true, true); run(61, 6, true, false); run(50, 7, true, false); run(56, 5, true, false); run(48, 7, true, true);
  run(63, 5, false, false); run(64, 7, false, false);
  run(24, 4, false, false); run(63, 4, false, false); run(8, 2, false, false);                          // not positive definite
  run(24, 2, true, false, 20, 3); run(24, 2, true, false, 9, 5); run(24, 3, true, false, 23, 17);       // an extra coupling: arrow / inside the window
  return 0;
}
