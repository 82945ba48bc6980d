// scratch: latencies on the chain of the window solver, one wave, shader cycles per link (s_memtime around 256 links)
//   (a) dependent v_mfma_f64_16x16x4 (C in = previous D)            (b) dependent v_mfma_f64_4x4x4 (4 blocks)
//   (c) ds_write_b64 -> ds_read_b64 of the same word (other lane)   (d) mfma 16x16x4 -> ds_write of a result reg -> ds_read -> operand of the next mfma
//   (e) as (d) with the 4x4x4 form                                  (f) dependent f64 fma, for scale
// + the operand / result layout of v_mfma_f64_4x4x4_4b decoded with one-hot inputs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void lat(long long *out, double *sink) {
  __shared__ double lds[256];
  const int lane = threadIdx.x;
  lds[lane] = lane * 1e-3; lds[lane + 64] = 1.0; lds[lane + 128] = 0.0; lds[lane + 192] = 0.0;
  __syncthreads();
  d4 c = {0.0, 0.0, 0.0, 0.0};
  double a = 1e-3 * lane, b = 1e-3, x = 0.5;
  long long t0, t1;
  // (a)
  t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < 256; i++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  asm volatile("" : "+v"(c));
  a += c[0] * 1e-30;
  t1 = clock64(); if (lane == 0) out[0] = t1 - t0;
  // (b)
  double e = 0.0;
  t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < 256; i++) e = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, e, 0, 0, 0);
  a += e * 1e-30;
  t1 = clock64(); if (lane == 0) out[1] = t1 - t0;
  // (c)
  t0 = clock64();
  double v = x;
  for (int i = 0; i < 256; i++) { lds[lane] = v; asm volatile("" ::: "memory"); v = lds[(lane + 1) & 63] + 1e-9; asm volatile("" ::: "memory"); }
  t1 = clock64(); if (lane == 0) out[2] = t1 - t0;
  // (d)
  t0 = clock64();
  for (int i = 0; i < 256; i++) {
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    lds[lane] = c[1]; asm volatile("" ::: "memory");
    a = lds[(lane + 17) & 63] * 1e-9; asm volatile("" ::: "memory");
  }
  t1 = clock64(); if (lane == 0) out[3] = t1 - t0;
  // (e)
  t0 = clock64();
  for (int i = 0; i < 256; i++) {
    e = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, e, 0, 0, 0);
    lds[lane] = e; asm volatile("" ::: "memory");
    a = lds[(lane + 17) & 63] * 1e-9; asm volatile("" ::: "memory");
  }
  t1 = clock64(); if (lane == 0) out[4] = t1 - t0;
  // (f)
  t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < 256; i++) x = fma(x, 0.999, v);
  t1 = clock64(); if (lane == 0) out[5] = t1 - t0;
  // (g) 10 independent ds_read_b64 + wait, repeated (the pivot read)
  t0 = clock64();
  double acc = 0;
  for (int i = 0; i < 64; i++) { double r[10];
#pragma unroll
    for (int k = 0; k < 10; k++) r[k] = lds[(lane * 4 + k * 7 + i) & 255];
#pragma unroll
    for (int k = 0; k < 10; k++) acc += r[k]; asm volatile("" ::: "memory"); }
  t1 = clock64(); if (lane == 0) out[6] = t1 - t0;
  sink[lane] = c[0] + c[1] + c[2] + c[3] + e + v + x + a + acc;
}
// layout decode of the 4x4x4 form: A one-hot at lane la (value 1), B = lane id + 1 -> D tells which B lanes met A lane la
__global__ void decode(double *out) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; la++) {
    const double a = (lane == la) ? 1.0 : 0.0, b = lane + 1.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[la * 64 + lane] = d;
  }
}
int main() {
  long long *o; double *s, *dd; hipMalloc(&o, 64); hipMalloc(&s, 512); hipMalloc(&dd, 64 * 64 * 8);
  for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(lat, dim3(1), dim3(64), 0, 0, o, s);
  long long h[8]; hipMemcpy(h, o, 56, hipMemcpyDeviceToHost);
  printf("shader cycles per link: (a) dependent mfma_f64_16x16x4 %.1f  (b) dependent mfma_f64_4x4x4 %.1f  (c) ds_write->ds_read %.1f  (d) mfma16->ds_write->ds_read->mfma %.1f  (e) same, 4x4x4 %.1f  (f) dependent fma_f64 %.1f  (g) 10 x ds_read_b64 + wait %.1f\n",
         h[0] / 256.0, h[1] / 256.0, h[2] / 256.0, h[3] / 256.0, h[4] / 256.0, h[5] / 256.0, h[6] / 64.0);
  hipLaunchKernelGGL(decode, dim3(1), dim3(64), 0, 0, dd);
  static double hd[64 * 64]; hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
  // for A lane la: list (D lane <- B lane) pairs
  for (int la = 0; la < 64; la += 1) { if (la % 16 >= 8 && la >= 16) continue; printf("A lane %2d:", la); for (int l = 0; l < 64; l++) if (hd[la * 64 + l] != 0.0) printf(" D[%d]<-B[%d]", l, (int)hd[la * 64 + l] - 1); printf("\n"); }
  return 0;
}
