// scratch: LDS read throughput per CU by access width (gfx950): 16 waves, conflict-free lane addresses
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned *out, long long *cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  for (int i = threadIdx.x; i < 65536 / 4; i += 1024) ((unsigned *)lds)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned acc = 0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int base = ((wave * 16 + u) * 256 + it * 4096) & 32767;
      if (MODE == 0) acc += *(const unsigned short *)(lds + base + lane * 2);            // u16, consecutive halves
      if (MODE == 1) acc += *(const unsigned *)(lds + base + lane * 4);                  // b32
      if (MODE == 2) { const uint2 v = *(const uint2 *)(lds + ((base + lane * 8) & 32767)); acc += v.x ^ v.y; }
      if (MODE == 3) { const uint4 v = *(const uint4 *)(lds + ((base + lane * 16) & 32767)); acc += v.x ^ v.w; }
      if (MODE == 4) acc += *(const unsigned short *)(lds + ((base + lane * 1098) & 32767 & ~1));  // u16, the tile's diagonal stride
      if (MODE == 5) acc += *(const unsigned char *)(lds + base + lane);
    }
  }
  const long long t1 = clock64();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
  unsigned *o; long long *c; (void)hipMalloc(&o, 4096); (void)hipMalloc(&c, 64);
  const int it = 256;
  const char *names[6] = {"ds_read_u16 (2 B/lane)", "ds_read_b32", "ds_read_b64", "ds_read_b128", "ds_read_u16, stride 1098 B", "ds_read_u8"};
#define R(M) { k<M><<<1, 1024>>>(o, c, it); k<M><<<1, 1024>>>(o, c, it); (void)hipDeviceSynchronize(); long long h[8]; (void)hipMemcpy(h, c, 64, hipMemcpyDeviceToHost); printf("%-30s %.1f cycles per wave instruction (16 waves on the CU -> %.1f cycles of LDS pipe each)\n", names[M], (double)h[M] / (16.0 * it), (double)h[M] / (16.0 * it) / 16.0); }
  R(0); R(1); R(2); R(3); R(4); R(5);
}
