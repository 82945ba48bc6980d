// scratch: lookup address-pattern experiment (timing only; contents are garbage)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstdlib>
struct __attribute__((packed, aligned(2))) Half8 { _Float16 v[8]; };
struct Levels { const _Float16* vol[4]; };

// MODE 0: reference layout [n][h1][w1][h2][w2], 8 x 16B row loads per pixel
// MODE 1: sheared layout [n][dy][dx][h1][w1], 64 x 2B loads per pixel, coalesced across lanes
template <int MODE, bool COMPUTE>
__global__ __launch_bounds__(256) void k(Levels L, const float2* coords, _Float16* out, int n, int h, int w) {
  const int HW = h * w;
  const long pix = (long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= (long)n * HW) return;
  const int lvl = blockIdx.y;
  const int e = pix / HW, rem = pix - (long)e * HW;
  const int y1 = rem / w, x1 = rem - y1 * w;
  const int hl = h >> lvl, wl = w >> lvl;
  const float2 c = coords[pix];
  const float sc = 1.f / (1 << lvl);
  const float x0 = c.x * sc, y0 = c.y * sc;
  const int ix0 = (int)floorf(x0) - 3, iy0 = (int)floorf(y0) - 3;
  const float dx = x0 - floorf(x0), dy = y0 - floorf(y0);
  float win[8][8];
  if (MODE == 0) {
    const _Float16* plane = L.vol[lvl] + (size_t)pix * hl * wl;
    const bool interior = ix0 >= 0 && iy0 >= 0 && ix0 + 8 <= wl && iy0 + 8 <= hl;
    if (interior) {
      Half8 raw[8];
#pragma unroll
      for (int j = 0; j < 8; j++) raw[j] = *reinterpret_cast<const Half8*>(plane + (size_t)(iy0 + j) * wl + ix0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; j++)
#pragma unroll
        for (int i = 0; i < 8; i++) win[j][i] = (float)raw[j].v[i];
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
          int yy = iy0 + j, xx = ix0 + i;
          bool ok = yy >= 0 && yy < hl && xx >= 0 && xx < wl;
          win[j][i] = ok ? (float)plane[(size_t)yy * wl + xx] : 0.f;
        }
    }
  } else {
    const _Float16* base = L.vol[lvl] + (size_t)e * hl * wl * HW + rem;
    const int sy = y1 >> lvl, sx = x1 >> lvl;
    _Float16 raw[8][8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int ty = iy0 + j;
      int dyi = ty - sy; dyi += (dyi < 0) ? hl : 0;
      const bool rok = ty >= 0 && ty < hl;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int tx = ix0 + i;
        int dxi = tx - sx; dxi += (dxi < 0) ? wl : 0;
        const bool ok = rok && tx >= 0 && tx < wl;
        raw[j][i] = ok ? base[(size_t)(dyi * wl + dxi) * HW] : (_Float16)0.f;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
      for (int i = 0; i < 8; i++) win[j][i] = (float)raw[j][i];
  }
  _Float16* o = out + ((size_t)e * 196 + lvl * 49) * HW + rem;
  if (COMPUTE) {
    const float w00 = (float)(_Float16)((1 - dx) * (1 - dy)), w01 = (float)(_Float16)((1 - dx) * dy);
    const float w10 = (float)(_Float16)(dx * (1 - dy)), w11 = (float)(_Float16)(dx * dy);
#pragma unroll
    for (int a = 0; a < 7; a++)
#pragma unroll
      for (int b = 0; b < 7; b++) {
        float acc = (float)(_Float16)(win[b][a] * w00);
        acc = (float)(_Float16)(acc + (float)(_Float16)(win[b + 1][a] * w01));
        acc = (float)(_Float16)(acc + (float)(_Float16)(win[b][a + 1] * w10));
        acc = (float)(_Float16)(acc + (float)(_Float16)(win[b + 1][a + 1] * w11));
        o[(size_t)(a * 7 + b) * HW] = (_Float16)acc;
      }
  } else {
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
      for (int i = 0; i < 8; i++) s += win[j][i];
    o[0] = (_Float16)s;
  }
}

int main() {
  const int n = 96, h = 64, w = 64, HW = h * w;
  Levels L;
  for (int l = 0; l < 4; l++) {
    size_t bytes = (size_t)n * HW * (h >> l) * (w >> l) * 2;
    hipMalloc((void**)&L.vol[l], bytes);
    hipMemset((void*)L.vol[l], 0, bytes);
  }
  std::vector<float2> c((size_t)n * HW);
  srand(1);
  for (int e = 0; e < n; e++) {
    float fx = (rand() % 2000) / 100.f - 10.f, fy = (rand() % 2000) / 100.f - 10.f;  // per-edge flow
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        float jit = 0.3f * sinf(0.2f * x + 0.13f * y + e);
        c[(size_t)e * HW + y * w + x] = make_float2(x + fx + jit + 0.02f * x, y + fy + 0.5f * jit + 0.01f * y);
      }
  }
  float2* dc; hipMalloc(&dc, c.size() * 8); hipMemcpy(dc, c.data(), c.size() * 8, hipMemcpyHostToDevice);
  _Float16* out; hipMalloc(&out, (size_t)n * 196 * HW * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid((n * HW + 255) / 256, 4);
  auto run = [&](const char* name, auto kern) {
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, L, dc, out, n, h, w);
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(kern, grid, dim3(256), 0, 0, L, dc, out, n, h, w);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.1f us  -> %.0f GB/s algorithmic\n", name, ms / 20 * 1e3, 358.6e6 / (ms / 20 * 1e-3) / 1e9);
  };
  run("ref layout, full", k<0, true>);
  run("ref layout, loads only", k<0, false>);
  run("sheared layout, full", k<1, true>);
  run("sheared layout, loads only", k<1, false>);
  return 0;
}
