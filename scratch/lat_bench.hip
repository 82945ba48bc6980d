// scratch: latency microbenchmarks on one workgroup (704 threads)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(long long* out, double* sink, int reps) {
  __shared__ double buf[4096];
  const int tid = threadIdx.x;
  long long w0 = wall_clock64(), c0 = clock64();
  // (a) barriers only
  for (int i = 0; i < reps; i++) __syncthreads();
  long long w1 = wall_clock64(), c1 = clock64();
  // (b) dependent f64 fma chain
  double x = sink[0], y = sink[1];
  for (int i = 0; i < reps; i++) { x = fma(x, y, 1.0); x = fma(x, y, 1.0); x = fma(x, y, 1.0); x = fma(x, y, 1.0); }
  long long w2 = wall_clock64(), c2 = clock64();
  // (c) write -> barrier -> read -> dependent fma
  for (int i = 0; i < reps; i++) { buf[(tid + i) & 4095] = x; __syncthreads(); x = fma(buf[(tid * 7 + i + 1) & 4095], y, x); }
  long long w3 = wall_clock64(), c3 = clock64();
  // (d) rcp + newton chain
  for (int i = 0; i < reps; i++) { double r = __builtin_amdgcn_rcp(x); double e = fma(-x, r, 1.0); r = fma(r, e, r); e = fma(-x, r, 1.0); r = fma(r, e, r); x = r + 1.5; }
  long long w4 = wall_clock64(), c4 = clock64();
  // (e) independent f64 fma throughput (8 chains)
  double z[8]; for (int j = 0; j < 8; j++) z[j] = x + j;
  for (int i = 0; i < reps; i++) { for (int j = 0; j < 8; j++) z[j] = fma(z[j], y, 1.0); }
  long long w5 = wall_clock64(), c5 = clock64();
  // (f) LDS read latency chain (pointer chase)
  int p = tid & 63; int* ib = (int*)buf; __syncthreads(); ib[tid & 1023] = (tid * 13 + 5) & 1023; __syncthreads();
  long long w6 = wall_clock64(), c6 = clock64();
  for (int i = 0; i < reps; i++) p = ib[p];
  long long w7 = wall_clock64(), c7 = clock64();
  for (int j = 0; j < 8; j++) x += z[j];
  sink[2 + tid] = x + p;
  if (tid == 0) { long long v[] = {w1-w0,c1-c0,w2-w1,c2-c1,w3-w2,c3-c2,w4-w3,c4-c3,w5-w4,c5-c4,w7-w6,c7-c6}; for (int i = 0; i < 12; i++) out[i] = v[i]; }
}
int main() {
  long long* out; double* sink; hipMalloc(&out, 256); hipMalloc(&sink, 8 * 2048);
  double h[2] = {1.0000001, 0.5}; hipMemcpy(sink, h, 16, hipMemcpyHostToDevice);
  const int reps = 1000;
  for (int threads : {64, 704, 1024}) {
    for (int it = 0; it < 2; it++) { hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, out, sink, reps); hipDeviceSynchronize(); }
    long long v[12]; hipMemcpy(v, out, sizeof(v), hipMemcpyDeviceToHost);
    const char* nm[] = {"barrier", "4 dep fma", "write-barrier-read-fma", "rcp+2newton+add", "8 indep fma", "lds chase"};
    printf("threads=%d\n", threads);
    for (int i = 0; i < 6; i++) printf("  %-24s %7.1f ns/iter  %7.1f clk/iter (clock64)  => %.0f MHz\n", nm[i], v[2*i]*10.0/reps, (double)v[2*i+1]/reps, v[2*i+1]/(v[2*i]*10.0)*1000);
  }
}
