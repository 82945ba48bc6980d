// scratch: dependent-chain latencies of the f64 VALU ops the reduced-system solvers are made of (one wave, gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rl(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <int MODE>
__global__ void k(double *out, long long *cyc, double a, double b, int iters) {
  double x = out[threadIdx.x], y = x + 1.0;
  const long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (MODE == 0) x = fma(x, a, b);                                   // dependent fma
      if (MODE == 1) { x = fma(x, a, b); y = fma(y, a, b); }             // two independent chains
      if (MODE == 2) x = fma(rl(x, (u * 2) & 63), a, x);                 // readlane -> fma
      if (MODE == 3) x = __builtin_amdgcn_rcp(x) + a;                    // rcp -> add
      if (MODE == 4) { float f = (float)x; f = __builtin_fmaf(f, (float)a, (float)b); x = f; }  // cvt, f32 fma, cvt
      if (MODE == 5) {                                                   // one substitution step
        const double t0_ = rl(x, (2 * u) & 63), t1_ = rl(x, (2 * u + 1) & 63);
        const double x0 = fma(b, t1_, a * t0_), x1 = fma(a, t1_, b * t0_);
        x = fma(-y, x1, fma(-a, x0, x));
      }
    }
  }
  const long long t1 = clock64();
  out[threadIdx.x] = x + y;
  if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
  double *o; long long *c; hipMalloc(&o, 64 * 8); hipMalloc(&c, 64); hipMemset(o, 0, 512);
  const int it = 1000;
#define R(M, name) { k<M><<<1, 64>>>(o, c, 0.999, 1e-3, it); k<M><<<1, 64>>>(o, c, 0.999, 1e-3, it); hipDeviceSynchronize(); long long h[8]; hipMemcpy(h, c, 64, hipMemcpyDeviceToHost); printf("%-28s %.1f cycles per link\n", name, (double)h[M] / (16.0 * it)); }
  R(0, "fma_f64 dependent"); R(1, "2 independent fma_f64"); R(2, "readlane(2) -> fma_f64"); R(3, "rcp_f64 -> add_f64"); R(4, "cvt f64->f32, fma_f32, cvt"); R(5, "substitution step");
}
