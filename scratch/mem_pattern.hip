// scratch: what the HBM system gives for the lookup's read and write patterns in isolation (no arithmetic), cold.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/bin/mem_pattern scratch/mem_pattern.hip
// geometry: E edges x 64 strips x 4 "levels" (all level-0 sized: 64 x 64 planes of 64 x 64 pixels, 2 B)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
typedef unsigned u4v __attribute__((ext_vector_type(4)));
constexpr int NY = 10, NX = 11;  // union of the bench scene at level 0 ~ 110 lines; average over levels ~ 90

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// RMODE 0: none, 1: [dy][dx][strip] (line stride 8 KB), 2: [dy][strip][dx] (lines of a row contiguous)
// WMODE 0: none, 1: 49 lines x 2 B/lane, 2: 4 B/lane to two planes, 3: wave = 2 strips, 4 B/lane (256 B per plane), 4: linear
// aux bit 0: nt loads; bit 1: the window origin depends on a value loaded first (the lookup's coords -> address chain);
// bit 2: per-lane predicates trim ~25 % of the 16-byte pieces (the lookup's per-group trimming)
template <int RMODE, int WMODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const char* __restrict__ vol, char* __restrict__ out, int E, unsigned* sink, int aux) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q8 = gridDim.x >> 3, r8 = gridDim.x & 7, xk = blockIdx.x & 7;
  const int lb = xk * q8 + min(xk, r8) + (blockIdx.x >> 3);
  const int lvl = blockIdx.y;
  int sid = lb * WAVES + wave;
  const int strips_per_wave = (WMODE == 3) ? 2 : 1;
  const int nstr = E * 64 / strips_per_wave;
  if (sid >= nstr) return;
  const int e = (sid * strips_per_wave) / 64, s = (sid * strips_per_wave) % 64;
  const size_t EDGE = (size_t)64 * 64 * 8192;  // bytes per edge-level
  const char* vb = vol + ((size_t)lvl * E + e) * EDGE;
  unsigned acc = 0;
  if (RMODE) {
    unsigned h = hash32(sid * 4 + lvl);
    if (aux & 2) {  // dependent chain: 8 bytes per lane from a coords-like array, the origin comes out of it
      const unsigned long long c = ((const unsigned long long*)out)[(size_t)(sid % (E * 64)) * 64 + lane];
      h += (unsigned)__builtin_amdgcn_readfirstlane((int)(c & 3));
    }
    const int dy0 = h % 50, dx0 = (h >> 8) % 50;
    const bool trim = (aux & 4) != 0;
    for (int ss = 0; ss < strips_per_wave; ss++) {
      u4v v[NY * 2];
#pragma unroll
      for (int r = 0; r < NY; r++)
#pragma unroll
        for (int t = 0; t < 2; t++) {
          const int jx = (lane >> 3) + 8 * t, sub = lane & 7;
          size_t off;
          if (RMODE == 1) off = ((size_t)(dy0 + r) * 64 + dx0 + jx) * 8192 + (size_t)(s + ss) * 128 + sub * 16;
          else off = (((size_t)(dy0 + r) * 64 + (s + ss)) * 64 + dx0 + jx) * 128 + sub * 16;
          const bool need = (jx < NX) && !(trim && ((hash32(lane * 31 + r * 7 + t) & 3) == 0));
          v[r * 2 + t] = need ? ((aux & 1) ? __builtin_nontemporal_load((const u4v*)(vb + off)) : *(const u4v*)(vb + off)) : u4v{0, 0, 0, 0};
        }
#pragma unroll
      for (int i = 0; i < NY * 2; i++) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    }
  }
  if (WMODE) {
    char* ob = out + ((size_t)e * 196 + (size_t)lvl * 49) * 8192;
    if (WMODE == 1) {
      for (int ch = 0; ch < 49; ch++) *(uint16_t*)(ob + (size_t)ch * 8192 + s * 128 + lane * 2) = (uint16_t)(acc + ch);
    } else if (WMODE == 2) {
      for (int ch = 0; ch < 48; ch += 2) *(unsigned*)(ob + (size_t)(ch + (lane & 1)) * 8192 + s * 128 + (lane >> 1) * 4) = acc + ch;
      *(uint16_t*)(ob + (size_t)48 * 8192 + s * 128 + lane * 2) = (uint16_t)acc;
    } else if (WMODE == 3) {
      for (int ch = 0; ch < 49; ch++) *(unsigned*)(ob + (size_t)ch * 8192 + s * 128 + lane * 4) = acc + ch;
    } else if (WMODE == 4) {
      char* lin = out + ((size_t)(sid * 4 + lvl)) * 49 * 128;
      for (int ch = 0; ch < 49; ch++) *(uint16_t*)(lin + ch * 128 + lane * 2) = (uint16_t)(acc + ch);
    }
  }
  if (acc == 0x12345u && !WMODE) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int E = argc > 1 ? atoi(argv[1]) : 96;
  const size_t volbytes = (size_t)4 * E * 64 * 64 * 8192;
  const size_t outbytes = (size_t)E * 196 * 8192;
  char* vol; hipMalloc(&vol, volbytes); hipMemset(vol, 1, volbytes);
  char* out[3]; for (int i = 0; i < 3; i++) { hipMalloc(&out[i], outbytes); hipMemset(out[i], 0, outbytes); }
  unsigned* sink; hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double rbytes = (double)E * 64 * 4 * NY * NX * 128, wbytes = (double)E * 196 * 8192;
#define RUN(NAME, R, W, WV, AUX)                                                                                  \
  {                                                                                                               \
    const int spw = (W == 3) ? 2 : 1;                                                                             \
    dim3 grid((E * 64 / spw + WV - 1) / WV, 4);                                                                   \
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<R, W, WV>), grid, dim3(WV * 64), 0, 0, vol, out[i % 3], E, sink, AUX); \
    hipEventRecord(e0);                                                                                           \
    for (int i = 0; i < 9; i++) hipLaunchKernelGGL((k<R, W, WV>), grid, dim3(WV * 64), 0, 0, vol, out[i % 3], E, sink, AUX); \
    hipEventRecord(e1); hipEventSynchronize(e1);                                                                  \
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 9;                                                          \
    const double b = (R ? rbytes : 0) + (W ? wbytes : 0);                                                         \
    printf("%-64s %8.1f us  %6.2f TB/s  (read %.0f MB, write %.0f MB)\n", NAME, ms * 1e3, b / ms / 1e9,           \
           R ? rbytes / 1e6 : 0.0, W ? wbytes / 1e6 : 0.0);                                                      \
  }
  RUN("read  [dy][dx][strip] (8 KB line stride), plain loads", 1, 0, 4, 0);
  RUN("read  [dy][dx][strip], nt loads", 1, 0, 4, 1);
  RUN("read  [dy][strip][dx] (rows contiguous), plain loads", 2, 0, 4, 0);
  RUN("read  [dy][strip][dx], nt loads", 2, 0, 4, 1);
  RUN("write 49 lines x 2 B/lane (8 KB apart)", 0, 1, 4, 0);
  RUN("write 4 B/lane to two planes", 0, 2, 4, 0);
  RUN("write wave = 2 strips, 4 B/lane (256 B runs)", 0, 3, 4, 0);
  RUN("write linear (fill-like)", 0, 4, 4, 0);
  RUN("write 49 lines x 2 B/lane, 1 wave per workgroup", 0, 1, 1, 0);
  RUN("write 49 lines x 2 B/lane, 8 waves per workgroup", 0, 1, 8, 0);
  RUN("read [dy][dx][strip] + write 2 B/lane", 1, 1, 4, 0);
  RUN("read [dy][strip][dx] + write 2 B/lane", 2, 1, 4, 0);
  RUN("read [dy][strip][dx] + write 2 strips", 2, 3, 4, 0);
  RUN("read [dy][strip][dx] + write linear", 2, 4, 4, 0);
  RUN("read [dy][dx][strip] + write 2 B/lane, origin from a loaded value", 1, 1, 4, 2);
  RUN("read [dy][dx][strip] + write 2 B/lane, 25 % of the pieces trimmed", 1, 1, 4, 4);
  RUN("read [dy][dx][strip] + write 2 B/lane, both", 1, 1, 4, 6);
  RUN("read [dy][dx][strip] + write 2 B/lane, both, 8 waves / workgroup", 1, 1, 8, 6);
  RUN("read [dy][dx][strip] + write 2 B/lane, both, 1 wave / workgroup", 1, 1, 1, 6);
  return 0;
}
