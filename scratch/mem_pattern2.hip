// scratch: round-3 follow-up of mem_pattern.hip -- WHY do torch add / fill_ reach 6.0 / 6.9 TB/s while every form of the
// lookup's traffic stays at 5.0-5.5?  Same geometry (E edges x 64 strips x 4 "levels", planes of 64 x 64 pixels, 2 B), same
// bytes (NY x NX lines of 128 B read, 49 lines written per strip and level), no arithmetic; what varies is the STRUCTURE:
//   one-shot waves vs persistent software-pipelined waves, 2-byte vs 16-byte-per-lane stores, reader / writer waves split,
//   LDS-DMA staging, plane pitch (power of two or not), and an add-like streaming kernel of the same size as calibration.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scratch/bin/mem_pattern2 scratch/mem_pattern2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
typedef unsigned u4v __attribute__((ext_vector_type(4)));
constexpr int NY = 10, NX = 11;

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

struct Geo {
  const char* vol; char* out; int E; size_t plane;  // plane = bytes between consecutive (dy, dx) lines of a strip (8192 + pad)
  __host__ __device__ size_t edge_bytes() const { return (size_t)64 * 64 * plane; }
};

__device__ __forceinline__ void item_decode(const Geo& g, int item, int& e, int& s, int& lvl) {
  const int nstr = g.E * 64;
  lvl = item / nstr;
  const int sid = item - lvl * nstr;
  e = sid >> 6, s = sid & 63;
}

// the wave's 20 staging loads of one strip-level (8 lanes = one 128-byte line, 16 B each)
__device__ __forceinline__ void issue_loads(const Geo& g, int e, int s, int lvl, int lane, u4v (&v)[NY * 2]) {
  const unsigned h = hash32((e * 64 + s) * 4 + lvl);
  const int dy0 = h % 50, dx0 = (h >> 8) % 50;
  const char* vb = g.vol + ((size_t)lvl * g.E + e) * g.edge_bytes();
#pragma unroll
  for (int r = 0; r < NY; r++)
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int jx = (lane >> 3) + 8 * t, sub = lane & 7;
      const size_t off = ((size_t)(dy0 + r) * 64 + dx0 + jx) * g.plane + (size_t)s * 128 + sub * 16;
      v[r * 2 + t] = (jx < NX) ? *(const u4v*)(vb + off) : u4v{0, 0, 0, 0};
    }
}
__device__ __forceinline__ unsigned fold(const u4v (&v)[NY * 2]) {
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < NY * 2; i++) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  return acc;
}
template <int SW>
__device__ __forceinline__ void issue_stores(const Geo& g, int e, int s, int lvl, int lane, unsigned acc) {
  char* ob = g.out + ((size_t)e * 196 + (size_t)lvl * 49) * 8192 + s * 128;
  if (SW == 2) {
#pragma unroll 7
    for (int ch = 0; ch < 49; ch++) *(uint16_t*)(ob + (size_t)ch * 8192 + lane * 2) = (uint16_t)(acc + ch);
  } else {  // 16 B per lane: 8 lanes = one line, 8 lines per instruction
#pragma unroll
    for (int i = 0; i < 7; i++) {
      const int ch = (lane >> 3) + 8 * i;
      if (ch < 49) *(u4v*)(ob + (size_t)ch * 8192 + (lane & 7) * 16) = u4v{acc, acc + 1u, acc + 2u, acc + (unsigned)ch};
    }
  }
}

// XCD-aware block order of the lookup kernels
__device__ __forceinline__ int xcd_order() {
  const int q8 = gridDim.x >> 3, r8 = gridDim.x & 7, xk = blockIdx.x & 7;
  return xk * q8 + min(xk, r8) + (blockIdx.x >> 3);
}

// A: one wave = one strip-level, then it retires (what mem_pattern.hip and both lookup forms do).  MODE bit 0: reads, bit 1: writes
template <int SW, int MODE>
__global__ __launch_bounds__(256) void k_oneshot(Geo g, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int item = xcd_order() * 4 + wave;
  if (item >= g.E * 64 * 4) return;
  int e, s, lvl;
  item_decode(g, item, e, s, lvl);
  unsigned acc = item;
  if (MODE & 1) { u4v v[NY * 2]; issue_loads(g, e, s, lvl, lane, v); acc = fold(v); }
  if (MODE & 2) issue_stores<SW>(g, e, s, lvl, lane, acc);
  else if (acc == 0x12345u) sink[0] = acc;
}

// B: persistent: every wave walks a contiguous range of items; the next item's loads are issued BEFORE the current item's
// stores (memory instructions of a wave complete in order: a load waited for right behind stores waits for them too)
template <int SW>
__global__ __launch_bounds__(256, 1) void k_persist(Geo g, unsigned* sink, int per_wave) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int w = xcd_order() * 4 + wave;
  const int total = g.E * 64 * 4;
  int it = w * per_wave;
  const int end = min(total, it + per_wave);
  if (it >= end) return;
  int e, s, lvl;
  u4v va[NY * 2], vb[NY * 2];
  item_decode(g, it, e, s, lvl);
  issue_loads(g, e, s, lvl, lane, va);
  for (; it < end; it += 2) {
    int e2 = e, s2 = s, l2 = lvl;
    if (it + 1 < end) { item_decode(g, it + 1, e2, s2, l2); issue_loads(g, e2, s2, l2, lane, vb); }
    issue_stores<SW>(g, e, s, lvl, lane, fold(va));
    if (it + 1 >= end) break;
    if (it + 2 < end) { item_decode(g, it + 2, e, s, lvl); issue_loads(g, e, s, lvl, lane, va); }
    issue_stores<SW>(g, e2, s2, l2, lane, fold(vb));
  }
  if (sink && lane == 77) sink[0] = 1;
}

// C: reader waves and writer waves: waves 0-3 of a workgroup only load, waves 4-7 only store (independent vmcnt streams)
template <int SW>
__global__ __launch_bounds__(512) void k_split(Geo g, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int item = xcd_order() * 4 + (wave & 3);
  if (item >= g.E * 64 * 4) return;
  int e, s, lvl;
  item_decode(g, item, e, s, lvl);
  if (wave < 4) { u4v v[NY * 2]; issue_loads(g, e, s, lvl, lane, v); if (fold(v) == 0x12345u) sink[0] = 1; }
  else issue_stores<SW>(g, e, s, lvl, lane, item);
}

// D: staging by LDS-DMA (buffer_load ... lds), one-shot
template <int SW>
__global__ __launch_bounds__(256) void k_ldsdma(Geo g, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int item = xcd_order() * 4 + wave;
  if (item >= g.E * 64 * 4) return;
  int e, s, lvl;
  item_decode(g, item, e, s, lvl);
  e = __builtin_amdgcn_readfirstlane(e); lvl = __builtin_amdgcn_readfirstlane(lvl);
  const unsigned h = hash32((e * 64 + s) * 4 + lvl);
  const int dy0 = h % 50, dx0 = (h >> 8) % 50;
  const char* vb = g.vol + ((size_t)lvl * g.E + e) * g.edge_bytes();
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, (int)g.edge_bytes(), 0x00020000);
  unsigned ldsrow = wave * (NY * 2 * 1024);
#pragma unroll
  for (int r = 0; r < NY; r++)
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int jx = (lane >> 3) + 8 * t, sub = lane & 7;
      const unsigned off = (unsigned)(((size_t)(dy0 + r) * 64 + dx0 + jx) * g.plane + (size_t)s * 128 + sub * 16);
      if (jx < NX)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem + ldsrow), 16, off, 0, 0, 0);
      ldsrow += 1024;
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned acc = *(const unsigned*)(smem + wave * (NY * 2 * 1024) + lane * 4);
  issue_stores<SW>(g, e, s, lvl, lane, acc);
}

// E: add-like streaming kernel of the same size: c = a + b, 16 B per lane, 4 vectors per thread
__global__ __launch_bounds__(256) void k_add(const u4v* __restrict__ a, const u4v* __restrict__ b, u4v* __restrict__ c, size_t n) {
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
  u4v x[4], y[4];
#pragma unroll
  for (int i = 0; i < 4; i++) if (base + i * 256 < n) { x[i] = a[base + i * 256]; y[i] = b[base + i * 256]; }
#pragma unroll
  for (int i = 0; i < 4; i++) if (base + i * 256 < n) c[base + i * 256] = x[i] + y[i];
}
// F: fill-like with 2 / 4 / 8 / 16 bytes per lane, linear
template <typename T>
__global__ __launch_bounds__(256) void k_fill(T* __restrict__ c, size_t n, T val) {
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; i++) if (base + i * 256 < n) c[base + i * 256] = val;
}

int main(int argc, char** argv) {
  const int E = argc > 1 ? atoi(argv[1]) : 96;
  const int pad = argc > 2 ? atoi(argv[2]) : 0;  // extra bytes per plane (0: 8 KB line stride)
  Geo g;
  g.E = E;
  g.plane = 8192 + pad;
  const size_t volbytes = (size_t)4 * E * g.edge_bytes();
  const size_t outbytes = (size_t)E * 196 * 8192;
  char* vol; if (hipMalloc(&vol, volbytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(vol, 1, volbytes);
  char* out[3]; for (int i = 0; i < 3; i++) { hipMalloc(&out[i], outbytes); hipMemset(out[i], 0, outbytes); }
  unsigned* sink; hipMalloc(&sink, 4);
  g.vol = vol;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double rbytes = (double)E * 64 * 4 * NY * NX * 128, wbytes = (double)E * 196 * 8192;
  const int items = E * 64 * 4;
  printf("E = %d, plane pitch %zu B: read %.0f MB, write %.0f MB per launch\n", E, g.plane, rbytes / 1e6, wbytes / 1e6);
#define TIME(NAME, BYTES, ...)                                                           \
  {                                                                                      \
    for (int i = 0; i < 3; i++) { g.out = out[i % 3]; __VA_ARGS__; }                         \
    hipEventRecord(e0);                                                                  \
    for (int i = 0; i < 9; i++) { g.out = out[i % 3]; __VA_ARGS__; }                         \
    hipEventRecord(e1); hipEventSynchronize(e1);                                         \
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 9;                                 \
    printf("%-72s %8.1f us  %6.2f TB/s\n", NAME, ms * 1e3, (BYTES) / ms / 1e9);          \
  }
  const int og = (items + 3) / 4;
  TIME("one-shot, reads only", rbytes, hipLaunchKernelGGL((k_oneshot<2, 1>), dim3(og), dim3(256), 0, 0, g, sink));
  TIME("one-shot, writes only, 2 B/lane", wbytes, hipLaunchKernelGGL((k_oneshot<2, 2>), dim3(og), dim3(256), 0, 0, g, sink));
  TIME("one-shot, writes only, 16 B/lane", wbytes, hipLaunchKernelGGL((k_oneshot<16, 2>), dim3(og), dim3(256), 0, 0, g, sink));
  TIME("one-shot, reads + 2 B/lane writes", rbytes + wbytes, hipLaunchKernelGGL((k_oneshot<2, 3>), dim3(og), dim3(256), 0, 0, g, sink));
  TIME("one-shot, reads + 16 B/lane writes", rbytes + wbytes, hipLaunchKernelGGL((k_oneshot<16, 3>), dim3(og), dim3(256), 0, 0, g, sink));
  TIME("reader waves + writer waves (2 B/lane)", rbytes + wbytes, hipLaunchKernelGGL((k_split<2>), dim3(og), dim3(512), 0, 0, g, sink));
  TIME("reader waves + writer waves (16 B/lane)", rbytes + wbytes, hipLaunchKernelGGL((k_split<16>), dim3(og), dim3(512), 0, 0, g, sink));
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ldsdma<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ldsdma<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  TIME("LDS-DMA staging + 2 B/lane writes", rbytes + wbytes, hipLaunchKernelGGL((k_ldsdma<2>), dim3(og), dim3(256), 4 * NY * 2 * 1024, 0, g, sink));
  TIME("LDS-DMA staging + 16 B/lane writes", rbytes + wbytes, hipLaunchKernelGGL((k_ldsdma<16>), dim3(og), dim3(256), 4 * NY * 2 * 1024, 0, g, sink));
  for (int wg_per_cu = 1; wg_per_cu <= 2; wg_per_cu++) {
    const int G = 256 * wg_per_cu, waves = G * 4, per = (items + waves - 1) / waves;
    char nm[128];
    snprintf(nm, sizeof nm, "persistent pipelined, %d workgroup(s)/CU, 2 B/lane writes", wg_per_cu);
    TIME(nm, rbytes + wbytes, hipLaunchKernelGGL((k_persist<2>), dim3(G), dim3(256), 0, 0, g, sink, per));
    snprintf(nm, sizeof nm, "persistent pipelined, %d workgroup(s)/CU, 16 B/lane writes", wg_per_cu);
    TIME(nm, rbytes + wbytes, hipLaunchKernelGGL((k_persist<16>), dim3(G), dim3(256), 0, 0, g, sink, per));
  }
  {  // calibration: the same number of bytes as an add (2 reads : 1 write) and as fills of 2..16 B per lane
    const size_t nvec = (size_t)((rbytes + wbytes) / 3 / 16);
    u4v* a = (u4v*)vol; u4v* b = a + nvec; u4v* c = (u4v*)out[0];
    const size_t cvec = outbytes / 16 < nvec ? outbytes / 16 : nvec;
    TIME("add-like (2 x 16 B loads + 16 B store per lane), same total bytes", 3.0 * cvec * 16, hipLaunchKernelGGL(k_add, dim3((cvec + 1023) / 1024), dim3(256), 0, 0, a, b, c, cvec));
    const size_t fb = outbytes;
    TIME("fill 16 B/lane (write bytes of one launch)", (double)fb, hipLaunchKernelGGL((k_fill<u4v>), dim3((fb / 16 + 1023) / 1024), dim3(256), 0, 0, (u4v*)g.out, fb / 16, u4v{1, 2, 3, 4}));
    TIME("fill 8 B/lane", (double)fb, hipLaunchKernelGGL((k_fill<unsigned long long>), dim3((fb / 8 + 1023) / 1024), dim3(256), 0, 0, (unsigned long long*)g.out, fb / 8, 5ull));
    TIME("fill 4 B/lane", (double)fb, hipLaunchKernelGGL((k_fill<unsigned>), dim3((fb / 4 + 1023) / 1024), dim3(256), 0, 0, (unsigned*)g.out, fb / 4, 5u));
    TIME("fill 2 B/lane", (double)fb, hipLaunchKernelGGL((k_fill<uint16_t>), dim3((fb / 2 + 1023) / 1024), dim3(256), 0, 0, (uint16_t*)g.out, fb / 2, (uint16_t)5));
    // a large fill (what torch fill_ of 1 GiB measures): all three output buffers are separate allocations, so use the volume
    const size_t big = volbytes < ((size_t)1 << 30) ? volbytes : ((size_t)1 << 30);
    TIME("fill 16 B/lane, 1 GiB", (double)big, hipLaunchKernelGGL((k_fill<u4v>), dim3((big / 16 + 1023) / 1024), dim3(256), 0, 0, (u4v*)vol, big / 16, u4v{1, 1, 1, 1}));
    TIME("fill 2 B/lane, 1 GiB", (double)big, hipLaunchKernelGGL((k_fill<uint16_t>), dim3((big / 2 + 1023) / 1024), dim3(256), 0, 0, (uint16_t*)vol, big / 2, (uint16_t)0x0101));
  }
  return 0;
}
