// scratch: HBM read rate for the lookup's staging pattern under two volume layouts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
// wave = one (edge, y1) row: walks `steps` rows; per row loads nlines lines of 128 B (16 B / lane, 2 slots)
__global__ __launch_bounds__(512, 8) void k(const uint4* __restrict__ vol, float* out, int nwaves, int steps, int nlines,
                                           size_t wave_stride16, size_t row_stride16, size_t line_stride16, int wr,
                                           uint16_t* __restrict__ wout) {
  const int wave = blockIdx.x * 8 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wave >= nwaves) return;
  const uint4* base = vol + (wave_stride16 ? (size_t)wave * wave_stride16 : (size_t)(wave >> 6) * (32u << 20) / 16 + (size_t)(wave & 63) * 8);
  unsigned acc = 0;
  for (int s = 0; s < steps; s++) {
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int slot = lane + 64 * t, jx = slot >> 3, sub = slot & 7;
      if (jx < nlines) { uint4 v = base[(size_t)s * row_stride16 + (size_t)jx * line_stride16 + sub]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (wr && s >= 1 && s < 8) {  // 7 channel stores per step, 49 total, 2 B per lane, planes 8 KB apart
#pragma unroll
      for (int a = 0; a < 7; a++) wout[((size_t)(wave >> 6) * 196 + (size_t)(a * 7 + s - 1)) * 4096 + (wave & 63) * 64 + lane] = (uint16_t)(acc + a);
    }
  }
  if (acc == 0x12345u) out[0] = acc;
}
int main() {
  const int nwaves = 96 * 64;  // level 0 only
  const size_t volbytes = (size_t)96 * 64 * 64 * 64 * 64 * 2;  // 3.2 GB: level 0 of 96 edges
  void* buf; hipMalloc(&buf, volbytes); hipMemset(buf, 1, volbytes);
  uint16_t* wout; hipMalloc(&wout, (size_t)96 * 196 * 4096 * 2);
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int steps = 12, nlines = 11;
  auto run = [&](const char* name, size_t wave_stride, size_t row_stride, size_t line_stride, int wr) {
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k, dim3(nwaves / 8), dim3(512), 0, 0, (const uint4*)buf, out, nwaves, steps, nlines, wave_stride / 16, row_stride / 16, line_stride / 16, wr, wout);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k, dim3(nwaves / 8), dim3(512), 0, 0, (const uint4*)buf, out, nwaves, steps, nlines, wave_stride / 16, row_stride / 16, line_stride / 16, wr, wout);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    const double rd = (double)nwaves * steps * nlines * 128, w = wr ? (double)nwaves * 49 * 128 : 0;
    printf("%-58s %7.1f us  read %6.0f GB/s  total %6.0f GB/s\n", name, ms * 1e3, rd / ms / 1e6, (rd + w) / ms / 1e6);
  };
  // current layout [e][dy][dx][y1][x1]: wave (e,y1) -> +128 B per y1, 32 MB per e; line stride 8 KB; row stride 512 KB
  // emulate wave index = e*64 + y1: wave_stride can't express both -> use y1-major approx: waves of one edge 128 B apart
  run("old: line 8KB, row 512KB, y1 128B, edge 32MB (read only)", 0, 512 * 1024, 8192, 0);
  run("new: line 128B, row 8KB, wave 512KB (read only)", 512 * 1024, 8192, 128, 0);
  run("old + writes", 0, 512 * 1024, 8192, 1);
  run("new + writes", 512 * 1024, 8192, 128, 1);
  run("writes only-ish (1 line)", 512 * 1024, 8192, 128, 1);
}
