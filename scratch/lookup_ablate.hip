// scratch: ablation of the streaming sheared lookup (full / no stores / no loads), timing only
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstdlib>
#include "../dba-fusion_amd/csrc/corr_sheared.hip"
namespace dba { void set_last_error(const char*, hipError_t) {} }
__global__ void fill_rand(unsigned short* p, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; size_t st = (size_t)gridDim.x * 256; for (; i < n; i += st) { unsigned x = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 13); x ^= x << 13; x ^= x >> 17; x ^= x << 5; p[i] = (unsigned short)((x & 0x83ff) | 0x3800); } }
int main() {
  const int n = 96, h = 64, w = 64, HW = h * w;
  const void* vols[4];
  for (int l = 0; l < 4; l++) { size_t bytes = (size_t)n * HW * (h >> l) * (w >> l) * 2; void* p; hipMalloc(&p, bytes); hipMemset(p, 0, bytes); if (getenv("RANDFILL")) hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, 0, (unsigned short*)p, bytes / 2); vols[l] = p; }
  std::vector<float> c((size_t)n * HW * 2);
  srand(1);
  for (int e = 0; e < n; e++) {
    float fx = (rand() % 2000) / 100.f - 10.f, fy = (rand() % 2000) / 100.f - 10.f;
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
      float jit = 0.3f * sinf(0.2f * x + 0.13f * y + e);
      c[((size_t)e * HW + y * w + x) * 2] = x + fx + jit + 0.02f * x; c[((size_t)e * HW + y * w + x) * 2 + 1] = y + fy + 0.5f * jit + 0.01f * y;
    }
  }
  float* dc; hipMalloc(&dc, c.size() * 4); hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice);
  void* out; hipMalloc(&out, (size_t)n * 196 * HW * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; i++) dba_corr_lookup_pyramid_sheared(vols, dc, out, n, h, w, h, w, 4, 3, 0);
  hipEventRecord(e0);
  for (int i = 0; i < 20; i++) dba_corr_lookup_pyramid_sheared(vols, dc, out, n, h, w, h, w, 4, 3, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%s: %.1f us -> %.0f GB/s algorithmic\n", VARIANT, ms / 20 * 1e3, 358.6e6 / (ms / 20 * 1e-3) / 1e9);
}
