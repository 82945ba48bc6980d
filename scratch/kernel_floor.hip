// scratch: what a kernel costs before it does anything (rocprofv3 --kernel-trace --stats on this binary): an empty kernel,
// one that reads a flag and returns (the queued fall-back solvers), with and without a 160 KB LDS reservation, and a
// kernel that dirties 12 MB (the linearisation's E) so that the end-of-kernel write-back shows.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_flag(const int *f, int *o) { if (f[0] == 12345) o[0] = 1; }
__global__ void k_flag_lds(const int *f, int *o) { extern __shared__ int s[]; if (f[0] == 12345) { s[threadIdx.x] = 1; o[0] = s[0]; } }
__global__ void k_dirty(float *p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.f; }
int main() {
  int *f, *o; float *p; (void)hipMalloc(&f, 64); (void)hipMalloc(&o, 64); (void)hipMalloc(&p, 12 << 20); (void)hipMemset(f, 0, 64);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_flag_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int it = 0; it < 200; it++) {
    hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0);
    hipLaunchKernelGGL(k_flag, dim3(1), dim3(512), 0, 0, f, o);
    hipLaunchKernelGGL(k_flag_lds, dim3(1), dim3(1024), 160 * 1024, 0, f, o);
    hipLaunchKernelGGL(k_dirty, dim3(768), dim3(256), 0, 0, p, (size_t)3 << 20);
    hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, 0);
  }
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  for (int it = 0; it < 1000; it++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("1000 empty kernels back to back: %.2f us each (stream throughput, no profiler)\n", ms);
  return 0;
}
