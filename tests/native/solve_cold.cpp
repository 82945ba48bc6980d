// Test infrastructure: ONE process = one cold start of the reduced-system solvers of libdba_hip.so.
//
// The window solver (dba-fusion_amd/csrc/ba_solve_wave.hip) synchronises its waves through LDS counters, the skyline kernel's two
// workgroups (ba_solve_band.hip) through a global-memory handshake.  A race in such a protocol shows when the waves start out
// of step -- the first launch of a process: code not yet in the instruction cache, clocks down, the waves dealt out one by one --
// and hides in warm back-to-back solves (round 5: one wrong solve in ~40 cold starts, none in 300 warm ones).  So this
// program initialises the device, builds ONE system, solves it ONCE through the C ABI (dba_ba_solve_skyline: replaces the host
// Eigen solve of /root/reference/src/droid_kernels.cu:1248-1269) as the first kernel of the process, checks it against a host
// Cholesky in float64, and exits; tests/test_gpu_solve_cold.py starts a few hundred of them.
//
//   solve_cold <libdba_hip.so> <P> <w> <seed> [extra warm solves of other systems, each checked]
//   w >= 0: pose p is coupled with p - w .. p;  w == -2: the skyline of the reduced system of BASELINE's 64-KF / 512-edge graph
//   exit code 0: all solves right, 1: a wrong / failed solve, 2: set-up error
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef struct {
  size_t H, b, dx, meta, E, Q, w, kx;
  int P, Mmax, nchunks;
} layout_t;   // dba_ba_layout of include/dba_hip.h
typedef size_t (*ws_bytes_fn)(int, int, int, int, int, int);
typedef int (*get_layout_fn)(int, int, int, int, int, int, layout_t *);
typedef int (*solve_fn)(int, int, int, int, int, int, float, float, const int32_t *, void *, size_t, void *);

static bool host_solve(std::vector<double> A, std::vector<double> b, int n, std::vector<double> &x) {
  for (int j = 0; j < n; j++) {
    double d = A[j * n + j];
    for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0)) return false;
    d = std::sqrt(d), A[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[i * n + j];
      for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= A[i * n + k] * b[k];
    b[i] = s / A[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = b[i];
    for (int k = i + 1; k < n; k++) s -= A[k * n + i] * b[k];
    b[i] = s / A[i * n + i];
  }
  x = b;
  return true;
}

#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("SETUP %s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char **argv) {
  if (argc < 5) return 2;
  const int P = atoi(argv[2]), w = atoi(argv[3]), seed = atoi(argv[4]), extra = argc > 5 ? atoi(argv[5]) : 0;
  void *lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) { printf("SETUP dlopen: %s\n", dlerror()); return 2; }
  ws_bytes_fn ws_bytes = (ws_bytes_fn)dlsym(lib, "dba_ba_workspace_bytes");
  get_layout_fn get_layout = (get_layout_fn)dlsym(lib, "dba_ba_get_layout");
  solve_fn solve = (solve_fn)dlsym(lib, "dba_ba_solve_skyline");
  if (!ws_bytes || !get_layout || !solve) { printf("SETUP dlsym\n"); return 2; }
  const int n = 6 * P, N = 1, B = P + 2, ht = 8, wd = 8, t0 = 1, t1 = 1 + P;
  const size_t nbytes = ws_bytes(N, B, ht, wd, t0, t1);
  layout_t lay;
  if (get_layout(N, B, ht, wd, t0, t1, &lay) != 0) { printf("SETUP layout\n"); return 2; }
  char *ws;
  int32_t *dfp;
  HIPOK(hipMalloc((void **)&ws, nbytes));
  HIPOK(hipMalloc((void **)&dfp, sizeof(int32_t) * P));
  HIPOK(hipMemset(ws, 0, nbytes));
  int bad = 0;
  for (int rep = 0; rep <= extra; rep++) {
    std::mt19937_64 rng(1000003ull * seed + 7919ull * rep + 131ull * P + w);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::vector<double> H((size_t)n * n, 0.0), b(n);
    std::vector<int32_t> fpose(P);
    for (int p = 0; p < P; p++) {
      fpose[p] = p;
      const int lit = p <= 10 ? 0 : p <= 13 ? p - 10 : p <= 17 ? p - 9 : p - 8;
      for (int q = 0; q <= p; q++) {
        if (!(w >= 0 ? p - q <= w : q >= lit)) continue;
        if (q < fpose[p]) fpose[p] = q;
        for (int a = 0; a < 6; a++)
          for (int c = 0; c < 6; c++) {
            if (p == q && c > a) continue;
            const double v = 0.3 * U(rng);
            H[(size_t)(6 * p + a) * n + 6 * q + c] = v, H[(size_t)(6 * q + c) * n + 6 * p + a] = v;
          }
      }
    }
    const double scale = 0.5 + 1.5 * std::fabs(U(rng));
    for (int i = 0; i < n; i++) {
      double s = 0;
      for (int j = 0; j < n; j++) s += std::fabs(H[(size_t)i * n + j]);
      H[(size_t)i * n + i] = s + 1.0 + std::fabs(U(rng));
      b[i] = scale * std::sin(1.3 * i + seed);
    }
    const double lm = 1e-4, ep = 0.1;
    std::vector<double> Hd = H, xr;
    for (int i = 0; i < n; i++) Hd[(size_t)i * n + i] += ep + lm * H[(size_t)i * n + i];   // droid_kernels.cu:1252-1253
    if (!host_solve(Hd, b, n, xr)) { printf("SETUP the host system is not positive definite\n"); return 2; }
    std::vector<double> Hl = H;   // the device reads the lower triangle only: poison the upper one
    for (int i = 0; i < n; i++)
      for (int j = i + 1; j < n; j++) Hl[(size_t)i * n + j] = 1e300;
    std::vector<float> x7(n, 7.0f);
    HIPOK(hipMemcpy(ws + lay.H, Hl.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
    HIPOK(hipMemcpy(ws + lay.b, b.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    HIPOK(hipMemcpy(ws + lay.dx, x7.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    HIPOK(hipMemcpy(dfp, fpose.data(), sizeof(int32_t) * P, hipMemcpyHostToDevice));
    HIPOK(hipDeviceSynchronize());
    if (solve(N, B, ht, wd, t0, t1, (float)lm, (float)ep, dfp, ws, nbytes, nullptr) != 0) { printf("SETUP solve rc\n"); return 2; }
    HIPOK(hipDeviceSynchronize());
    std::vector<float> x(n);
    int meta[8];
    HIPOK(hipMemcpy(x.data(), ws + lay.dx, sizeof(float) * n, hipMemcpyDeviceToHost));
    HIPOK(hipMemcpy(meta, ws + lay.meta, sizeof(meta), hipMemcpyDeviceToHost));
    double maxe = 0, maxx = 0;
    int first_bad = -1;
    for (int i = 0; i < n; i++) {
      const double e = std::fabs((double)x[i] - xr[i]);
      maxx = std::fmax(maxx, std::fabs(xr[i]));
      if (!(e <= maxe)) maxe = e;
    }
    const double tol = 3e-7 * std::fmax(1.0, maxx);
    for (int i = 0; i < n && first_bad < 0; i++)
      if (!(std::fabs((double)x[i] - xr[i]) <= tol)) first_bad = i;
    if (meta[1] != 0 || first_bad >= 0) {
      printf("BAD P=%d w=%d seed=%d solve %d: failed=%d max err %.3e (tol %.1e) first wrong unknown %d\n", P, w, seed, rep, meta[1], maxe, tol,
             first_bad);
      bad = 1;
    }
  }
  if (!bad) printf("ok P=%d w=%d seed=%d solves=%d\n", P, w, seed, extra + 1);
  return bad;
}
