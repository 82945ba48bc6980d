// scratch: phase timing of ba_solve_kernel with s_memtime
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define PROFILE_SOLVE 1
#include "../dba-fusion_amd/csrc/ba_solve.hip"
namespace dba { void set_last_error(const char*, hipError_t) {} }
int main() {
  const int n = 144;
  std::vector<double> H(n*n, 0.0), b(n);
  for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) { double v = (std::abs(i-j) < 30) ? 1.0/(1+std::abs(i-j)) : 0.0; H[i*n+j] = v; } H[i*n+i] += 5.0; b[i] = std::sin(i); }
  double *dH, *db; float* dx; int* meta; long long* prof;
  hipMalloc(&dH, n*n*8); hipMalloc(&db, n*8); hipMalloc(&dx, n*4); hipMalloc(&meta, 64); hipMalloc(&prof, 8*16);
  hipMemcpy(dH, H.data(), n*n*8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n*8, hipMemcpyHostToDevice);
  hipMemset(prof, 0, 8*16);
    for (int it = 0; it < 3; it++) { hipMemset(prof, 0, 8*16); dba::launch_ba_solve(dH, db, n, 1e-4, 0.1, dx, meta, nullptr, 0, prof); hipDeviceSynchronize(); }
  long long hp[16]; hipMemcpy(hp, prof, 8*16, hipMemcpyDeviceToHost);
  const char* names[] = {"load","diag","panel","upd_barrier","upd_mask","upd_tiles","bwd_tri","bwd_upd","final"};
  long long tot = 0; for (int i = 0; i < 9; i++) tot += hp[i];
  for (int i = 0; i < 9; i++) printf("%-8s %8lld ticks (%.1f%%)\n", names[i], hp[i], 100.0*hp[i]/tot);
  printf("total %lld ticks (100MHz const clock => %.1f us)\n", tot, tot/100.0);
  std::vector<float> x(n); hipMemcpy(x.data(), dx, n*4, hipMemcpyDeviceToHost);
  // residual check
  double maxr = 0; for (int i = 0; i < n; i++) { double s = 0; for (int j = 0; j < n; j++) { double a = H[i*n+j]; if (i==j) a += 0.1 + 1e-4*a; s += a*x[j]; } maxr = fmax(maxr, fabs(s-b[i])); }
  printf("max residual %.3e\n", maxr);
}
