// scratch (round 6): a no-arithmetic replica of the lookup's memory traffic on the RAGGED map shapes (55x55: HW = 3025, 28x107:
// HW = 2996), to see what the output rows' misalignment costs and what aligned line ownership would buy BEFORE building it.
// Geometry as the resident lookup: per (edge, strip of 64 pixels, level) a wave reads NY x NX lines of 128 B from the flow-aligned
// planes (pixel axis padded to a multiple of 64: reads are line-aligned) and writes 49 channel rows of 64 halfs into
// out[e][lvl*49 + ch][HW] -- channel rows HW*2 bytes apart, i.e. starting anywhere inside a 128-byte line when HW % 64 != 0.
//   V0  as the kernel stores today: wave = 64 consecutive pixels, one 2 B/lane store per channel (touches 2 lines when misaligned)
//   V1  the same with the channel pitch rounded up to a multiple of 128 B (what a padded output would give: NOT what the caller can take)
//   V2  aligned ownership: a workgroup (4 waves) owns 256 consecutive pixels; per channel its byte span [A, A + 512) is cut at the
//       128-byte boundaries: every wave stores whole lines, the partial head / tail lines are stored by the first / last wave
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scratch/mem_pattern3 scratch/mem_pattern3.hip && ./scratch/mem_pattern3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
typedef unsigned u4v __attribute__((ext_vector_type(4)));
constexpr int NY = 10, NX = 10;
__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
struct Geo {
  const char *vol; char *out;
  int E, HW, strips;       // strips = ceil(HW / 64)
  size_t chb;              // bytes between channel rows of the output
  size_t plane;            // bytes between (dy, dx) lines groups: strips * 128
  int h2;                  // plane grid (h2 x h2 offsets), >= 60
};
template <int MODE, bool READS>
__global__ __launch_bounds__(256) void replica(Geo g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wg = blockIdx.x;                  // (edge, 4 strips)
  const int groups = (g.strips + 3) / 4;
  const int lvl = blockIdx.y;
  const int e = wg / groups, s0 = (wg - e * groups) * 4, s = s0 + wave;
  unsigned acc = lane;
  if (READS && s < g.strips) {
    const unsigned h = hash32((e * 64 + s) * 4 + lvl);
    const int dy0 = h % (g.h2 - NY), dx0 = (h >> 8) % (g.h2 - NX - 8);
    const char *vb = g.vol + ((size_t)lvl * g.E + e) * (size_t)g.h2 * g.h2 * g.plane;
#pragma unroll
    for (int r = 0; r < NY; r++)
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const int jx = (lane >> 3) + 8 * t, sub = lane & 7;
        if (jx < NX) {
          const u4v v = *(const u4v *)(vb + ((size_t)(dy0 + r) * g.h2 + dx0 + jx) * g.plane + (size_t)s * 128 + sub * 16);
          acc += v.x ^ v.y ^ v.z ^ v.w;
        }
      }
  }
  char *ob = g.out + ((size_t)e * 196 + (size_t)lvl * 49) * g.chb;
  if (MODE == 0 || MODE == 1) {
    if (s < g.strips) {
      const int p = s * 64 + lane;
      if (p < g.HW) {
#pragma unroll 7
        for (int ch = 0; ch < 49; ch++) *(uint16_t *)(ob + (size_t)ch * g.chb + (size_t)p * 2) = (uint16_t)(acc + ch);
      }
    }
  } else {
    const int P0 = s0 * 64, P1 = min(P0 + 256, g.HW);      // the workgroup's pixels
    for (int ch = 0; ch < 49; ch++) {
      const size_t A = (size_t)(e * 196 + lvl * 49 + ch) * g.chb + (size_t)P0 * 2, Bz = A + (size_t)(P1 - P0) * 2;
      const size_t a0 = (A + 127) & ~(size_t)127;          // first whole line
      // whole lines a0 + 128 k: wave k % 4
      size_t la = a0 + (size_t)wave * 128;
      for (; la + 128 <= Bz; la += 512) *(uint16_t *)(g.out + la + lane * 2) = (uint16_t)(acc + ch);
      if (wave == 0 && A < a0) {                            // head piece [A, a0)
        const size_t q = A + (size_t)lane * 2;
        if (q < a0 && q < Bz) *(uint16_t *)(g.out + q) = (uint16_t)(acc + ch);
      }
      if (wave == 3) {                                       // tail piece [last whole line end, Bz)
        const size_t t0 = (a0 <= Bz) ? a0 + ((Bz - a0) / 128) * 128 : Bz;
        const size_t q = t0 + (size_t)lane * 2;
        if (q < Bz) *(uint16_t *)(g.out + q) = (uint16_t)(acc + ch);
      }
    }
  }
}
template <int MODE, bool READS>
static float run(const Geo &g, int reps) {
  const int groups = (g.strips + 3) / 4;
  dim3 grid(g.E * groups, 4);
  hipLaunchKernelGGL((replica<MODE, READS>), grid, dim3(256), 0, 0, g);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((replica<MODE, READS>), grid, dim3(256), 0, 0, g);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1000.f / reps;
}
int main() {
  struct Shape { const char *name; int HW, E; } shapes[] = {{"55x55, 36 edges", 3025, 36}, {"55x55, 96 edges", 3025, 96}, {"28x107, 122 edges", 2996, 122}, {"64x64, 96 edges", 4096, 96}};
  for (auto &sh : shapes) {
    Geo g; g.E = sh.E; g.HW = sh.HW; g.strips = (sh.HW + 63) / 64; g.plane = (size_t)g.strips * 128; g.h2 = 64;
    const size_t vol_bytes = (size_t)4 * g.E * g.h2 * g.h2 * g.plane;
    const size_t chb_pad = ((size_t)sh.HW * 2 + 127) & ~(size_t)127;
    const size_t out_bytes = (size_t)g.E * 196 * chb_pad + 4096;
    char *vol, *out;
    // rotate over 3 copies so that nothing is served from the MALL
    hipMalloc(&vol, vol_bytes * 3); hipMalloc(&out, out_bytes * 3);
    hipMemset(vol, 1, vol_bytes * 3); hipMemset(out, 0, out_bytes * 3);
    const double rd = (double)g.E * g.strips * 4 * NY * NX * 128, wr = (double)g.E * 196 * sh.HW * 2;
    printf("%s: read %.0f MB, write %.0f MB per launch (algorithmic: taps %.0f + out %.0f MB)\n", sh.name, rd / 1e6, wr / 1e6,
           (double)g.E * sh.HW * 4 * 128 / 1e6, wr / 1e6);
    auto go = [&](const char *label, int mode, bool reads, size_t chb) {
      float best = 1e9f;
      for (int c = 0; c < 3; c++) {
        g.vol = vol + (size_t)c * vol_bytes; g.out = out + (size_t)c * out_bytes; g.chb = chb;
        float us = 0;
        if (mode == 0) us = reads ? run<0, true>(g, 3) : run<0, false>(g, 3);
        if (mode == 1) us = reads ? run<1, true>(g, 3) : run<1, false>(g, 3);
        if (mode == 2) us = reads ? run<2, true>(g, 3) : run<2, false>(g, 3);
        best = us < best ? us : best;
      }
      printf("  %-78s %7.1f us   %5.2f TB/s\n", label, best, ((reads ? rd : 0) + wr) / best / 1e6);
    };
    go("V0 writes only, rows at their real offsets (2 B/lane, 64 consecutive pixels)", 0, false, (size_t)sh.HW * 2);
    go("V1 writes only, channel pitch padded to 128 B", 1, false, chb_pad);
    go("V2 writes only, real offsets, whole lines per wave + head / tail pieces", 2, false, (size_t)sh.HW * 2);
    go("V0 reads + writes, real offsets", 0, true, (size_t)sh.HW * 2);
    go("V1 reads + writes, padded pitch", 1, true, chb_pad);
    go("V2 reads + writes, whole lines per wave + head / tail pieces", 2, true, (size_t)sh.HW * 2);
    hipFree(vol); hipFree(out);
  }
  return 0;
}
