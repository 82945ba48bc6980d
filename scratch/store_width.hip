// scratch: the vector memory pipe's store rate per CU by bytes per lane, in the volume build's shape: a store instruction writes
// whole 128-byte lines that lie one plane (8 KB) apart -- 2 B/lane: one line per instruction, 4 B: two, 8 B: four, 16 B: eight --,
// 16 waves per workgroup, G workgroups (= busy CUs up to 256).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
template <int BYTES>
__global__ __launch_bounds__(1024) void k(char *__restrict__ dst, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPL = 128 / BYTES;          // lanes per line
  constexpr int LINES = 64 / LPL;           // lines per instruction
  const size_t wid = (size_t)blockIdx.x * 16 + wave;
  // a wave's region: reps x 16 instructions x LINES lines, lines 8 KB apart: [plane][wave's 128-byte column]
  char *d = dst + (wid & 63) * 128 + (wid >> 6) * ((size_t)reps * 16 * LINES * 8192) + (size_t)(lane / LPL) * 8192 + (lane % LPL) * BYTES;
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      char *p = d + ((size_t)r * 16 + i) * LINES * 8192;
      if (BYTES == 2) *(unsigned short *)p = (unsigned short)(r + i);
      if (BYTES == 4) *(unsigned *)p = r + i;
      if (BYTES == 8) { u2v v; v.x = r; v.y = i; *(u2v *)p = v; }
      if (BYTES == 16) { u4v v; v.x = r; v.y = i; v.z = lane; v.w = wave; *(u4v *)p = v; }
    }
  }
}
int main() {
  char *dst;
  const size_t total = (size_t)6 << 30;
  if (hipMalloc(&dst, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int bytes : {2, 4, 8, 16})
    for (int G : {32, 64, 96, 128, 192, 256, 512}) {
      // same BYTES per wave in every mode: 256 KB
      const int reps = 256 * 1024 / (16 * 64 * bytes);
      const double sb = (double)G * 16 * reps * 16 * 64 * bytes;
      if (sb > (double)total) continue;
      float best = 1e9f;
      for (int it = 0; it < 4; it++) {
        hipEventRecord(e0);
        if (bytes == 2) hipLaunchKernelGGL(k<2>, dim3(G), dim3(1024), 0, 0, dst, reps);
        if (bytes == 4) hipLaunchKernelGGL(k<4>, dim3(G), dim3(1024), 0, 0, dst, reps);
        if (bytes == 8) hipLaunchKernelGGL(k<8>, dim3(G), dim3(1024), 0, 0, dst, reps);
        if (bytes == 16) hipLaunchKernelGGL(k<16>, dim3(G), dim3(1024), 0, 0, dst, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms;
      }
      const int cus = G < 256 ? G : 256;
      printf("stores %2d B/lane (%d lines/instr)  G=%4d  %8.1f us  %6.2f TB/s  %5.1f GB/s per CU\n", bytes, 64 / (128 / bytes), G, best * 1e3,
             sb / best / 1e9, sb / (best * 1e-3) / 1e9 / cus);
    }
}
