#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rl(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__global__ void k(long long* out, double* sink, int reps) {
  __shared__ double buf[1024];
  const int tid = threadIdx.x;
  double x = sink[0] + tid * 1e-9, y = sink[1];
  long long c0 = clock64();
  for (int i = 0; i < reps; i++) { x = fma(rl(x, (i * 2) & 63), y, x); }            // readlane -> fma chain
  long long c1 = clock64();
  for (int i = 0; i < reps; i++) { double a = rl(x, (i*2)&63), b = rl(x, ((i*2)&63) + 1); double x0 = fma(y, b, y * a), x1 = fma(y, a, y * b); x = fma(-y, x1, fma(-y, x0, x)); }  // like the backsub
  long long c2 = clock64();
  for (int i = 0; i < reps; i++) { x = fma(x, y, 1.0); x = fma(x, y, 1.0); }   // 2 dep fma
  long long c3 = clock64();
  for (int i = 0; i < reps; i++) { int lo = __builtin_amdgcn_ds_bpermute(((i*2)&63) << 2, __double2loint(x)); int hi = __builtin_amdgcn_ds_bpermute(((i*2)&63) << 2, __double2hiint(x)); x = fma(__hiloint2double(hi, lo), y, x); }
  long long c4 = clock64();
  // readlane with fixed lane
  for (int i = 0; i < reps; i++) { x = fma(rl(x, 5), y, x); }
  long long c5 = clock64();
  // f32 chain for comparison
  float f = (float)x; for (int i = 0; i < reps; i++) { f = fmaf(f, 0.5f, 1.0f); f = fmaf(f, 0.5f, 1.0f); }
  long long c6 = clock64();
  sink[2 + tid] = x + f;
  if (tid == 0) { long long v[] = {c1-c0, c2-c1, c3-c2, c4-c3, c5-c4, c6-c5}; for (int i = 0; i < 6; i++) out[i] = v[i]; }
}
int main() {
  long long* out; double* sink; hipMalloc(&out, 256); hipMalloc(&sink, 8 * 2048);
  double h[2] = {1.0000001, 0.5}; hipMemcpy(sink, h, 16, hipMemcpyHostToDevice);
  const int reps = 1000;
  for (int it = 0; it < 2; it++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, sink, reps); hipDeviceSynchronize(); }
  long long v[6]; hipMemcpy(v, out, sizeof(v), hipMemcpyDeviceToHost);
  const char* nm[] = {"readlane(dyn)->fma", "backsub-like step", "2 dep fma f64", "bpermute->fma", "readlane(const)->fma", "2 dep fma f32"};
  for (int i = 0; i < 6; i++) printf("  %-24s %7.1f clk/iter\n", nm[i], (double)v[i]/reps);
}
