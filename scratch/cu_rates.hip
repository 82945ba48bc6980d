// scratch: per-CU rates of the vector memory pipe, as a function of how many CUs are busy (gfx950).
//   stores: 8 B/lane, every wave its own lines (HBM-bound when the whole chip stores?)
//   loads : 16 B/lane out of an L2-resident 1 MB region shared by all workgroups (the build kernel's target fragments)
//   both  : the same wave alternates 2 loads : 1 store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void k(const char *__restrict__ src, char *__restrict__ dst, int reps, unsigned *sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t wid = (size_t)blockIdx.x * 8 + wave;
  unsigned acc = 0;
  char *d = dst + wid * (size_t)reps * 16 * 512;
  for (int r = 0; r < reps; r++) {
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        u2v v; v.x = acc + i; v.y = r;
        *(u2v *)(d + ((size_t)r * 16 + i) * 512 + lane * 8) = v;
      }
    }
    if (MODE == 1 || MODE == 2) {
      u4v v[16];
#pragma unroll
      for (int i = 0; i < 16; i++) v[i] = *(const u4v *)(src + ((((size_t)r * 16 + i) * 8 + wave) & 1023) * 1024 + lane * 16);
#pragma unroll
      for (int i = 0; i < 16; i++) acc += v[i].x ^ v[i].w;
    }
  }
  if (acc == 0x1234567u) sink[0] = acc;
}
int main() {
  char *src, *dst; unsigned *sink;
  hipMalloc(&src, 1 << 20); hipMemset(src, 1, 1 << 20); hipMalloc(&sink, 4);
  const int reps = 64;  // per wave: 64 x 16 x 512 B = 512 KB stored, 64 x 16 KB loaded
  hipMalloc(&dst, (size_t)2048 * 8 * reps * 16 * 512);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[3] = {"stores 8 B/lane", "loads 16 B/lane (L2 hits)", "both"};
  for (int mode = 0; mode < 3; mode++)
    for (int G : {32, 64, 128, 256, 512, 1024, 2048}) {
      float best = 1e9f;
      for (int it = 0; it < 4; it++) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(G), dim3(512), 0, 0, src, dst, reps, sink);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(G), dim3(512), 0, 0, src, dst, reps, sink);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(G), dim3(512), 0, 0, src, dst, reps, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms;
      }
      const double sb = (mode != 1) ? (double)G * 8 * reps * 16 * 512 : 0, lb = (mode != 0) ? (double)G * 8 * reps * 16 * 1024 : 0;
      const int cus = G < 256 ? G : 256;
      printf("%-28s G=%4d  %8.1f us  store %6.2f TB/s %5.1f B/clk/CU   load %6.2f TB/s %5.1f B/clk/CU\n", names[mode], G, best * 1e3,
             sb / best / 1e9, sb / (best * 1e-3) / 2.4e9 / cus, lb / best / 1e9, lb / (best * 1e-3) / 2.4e9 / cus);
    }
}
