// scratch: read-bandwidth vs bytes-per-lane and access pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
template <typename T> struct Z { static __device__ float s(T v); };
template <typename T, int NLOADS>
__global__ __launch_bounds__(256) void seq(const T* __restrict__ p, float* out, size_t nelem, size_t stride_elems) {
  // each thread issues NLOADS loads; lane-contiguous within a load; consecutive loads `stride_elems` apart
  size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) / 64, lane = threadIdx.x & 63;
  size_t base = wave * 64 /*lanes*/ ;
  T acc[NLOADS];
#pragma unroll
  for (int i = 0; i < NLOADS; i++) {
    size_t idx = (base + lane + (size_t)i * stride_elems) % nelem;
    acc[i] = p[idx];
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NLOADS; i++) { const unsigned char* b = (const unsigned char*)&acc[i]; s += b[0]; }
  if (s == 12345.f) out[0] = s;
}
int main() {
  size_t bytes = (size_t)4 << 30;
  void* buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto kern, size_t esz, int nloads, size_t stride_bytes, size_t total_bytes) {
    size_t nelem = bytes / esz;
    size_t per_thread = (size_t)nloads * esz;
    size_t threads = total_bytes / per_thread;
    dim3 grid((threads + 255) / 256);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, (decltype(nullptr))nullptr, out, nelem, stride_bytes / esz);
  };
  (void)run;
#define RUN(T, NL, STRIDE_BYTES, NAME)                                                                \
  {                                                                                                   \
    size_t esz = sizeof(T), nelem = bytes / esz, total = (size_t)1 << 30;                             \
    size_t threads = total / (NL * esz);                                                              \
    dim3 grid((threads + 255) / 256);                                                                 \
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((seq<T, NL>), grid, dim3(256), 0, 0, (const T*)buf, out, nelem, (size_t)(STRIDE_BYTES) / esz); \
    hipEventRecord(e0);                                                                               \
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((seq<T, NL>), grid, dim3(256), 0, 0, (const T*)buf, out, nelem, (size_t)(STRIDE_BYTES) / esz); \
    hipEventRecord(e1); hipEventSynchronize(e1);                                                      \
    float ms; hipEventElapsedTime(&ms, e0, e1);                                                       \
    printf("%-44s %7.1f us  %6.0f GB/s\n", NAME, ms / 5 * 1e3, total / (ms / 5 * 1e-3) / 1e9);        \
  }
  // stride = distance between consecutive loads of the same thread
  RUN(uint16_t, 64, 8192, "2B/lane x64 loads, 8KB stride (sheared-like)");
  RUN(uint16_t, 64, 128 * 1024 * 1024 / 64, "2B/lane x64 loads, 2MB stride");
  RUN(uint16_t, 64, 128, "2B/lane x64 loads, contiguous per wave (8KB)");
  RUN(uint32_t, 32, 8192, "4B/lane x32 loads, 8KB stride");
  RUN(uint32_t, 32, 256, "4B/lane x32 loads, contiguous per wave");
  RUN(uint2, 16, 8192, "8B/lane x16 loads, 8KB stride");
  RUN(uint4, 8, 8192, "16B/lane x8 loads, 8KB stride");
  RUN(uint4, 8, 1024, "16B/lane x8 loads, contiguous per wave");
  RUN(uint4, 1, 1024, "16B/lane x1 load (plain stream)");
  return 0;
}
