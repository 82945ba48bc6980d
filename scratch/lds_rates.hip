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
      // the solvers' pattern: few distinct addresses per wave (a panel row is read by every lane that owns a tile in it)
      if (MODE == 6) { const uint4 v = *(const uint4 *)(lds + ((base + (lane >> 3) * 16) & 32767)); acc += v.x ^ v.w; }       // b128, 8 addresses
      if (MODE == 7) { const uint2 v = *(const uint2 *)(lds + ((base + (lane >> 3) * 16) & 32767)); acc += v.x ^ v.y; }       // b64, 8 addresses
      if (MODE == 8) { const uint4 v = *(const uint4 *)(lds + (base & 32767)); acc += v.x ^ v.w; }                             // b128, 1 address
      if (MODE == 9) { const uint4 v = *(const uint4 *)(lds + ((base + lane * 48) & 32767 & ~15)); acc += v.x ^ v.w; }          // b128, 48-byte pitch
      if (MODE == 10) { const uint4 v = *(const uint4 *)(lds + ((base + (lane & 31) * 32 + (lane >> 5) * 16) & 32767)); acc += v.x ^ v.w; }  // b128, the MFMA fragment pattern
    }
  }
  const long long t1 = clock64();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
  unsigned *o; long long *c; (void)hipMalloc(&o, 4096); (void)hipMalloc(&c, 128);
  const int it = 256;
  const char *names[11] = {"ds_read_u16 (2 B/lane)", "ds_read_b32", "ds_read_b64", "ds_read_b128", "ds_read_u16, stride 1098 B", "ds_read_u8",
                           "ds_read_b128, 8 distinct addresses", "ds_read_b64, 8 distinct addresses", "ds_read_b128, 1 address", "ds_read_b128, 48-byte pitch", "ds_read_b128, fragment pattern (32 B pitch, halves)"};
#define R(M) { k<M><<<1, 1024>>>(o, c, it); k<M><<<1, 1024>>>(o, c, it); (void)hipDeviceSynchronize(); long long h[16]; (void)hipMemcpy(h, c, 128, hipMemcpyDeviceToHost); printf("%-30s %.1f cycles per wave instruction (16 waves on the CU -> %.1f cycles of LDS pipe each)\n", names[M], (double)h[M] / (16.0 * it), (double)h[M] / (16.0 * it) / 16.0); }
  R(0); R(1); R(2); R(3); R(4); R(5); R(6); R(7); R(8); R(9); R(10);
}
