// The window solver's admission test (ba_solve_wave.hip), also run by stage 0 when it rebuilds a graph's tables
// (ba_kernels.hip::ba_prepare_kernel): whether the reduced system of this graph is banded enough is known the moment the
// pose-level skyline is, so the host's choice of solver for the workspace's next solves does not have to wait for a solve.
#pragma once
#include <hip/hip_runtime.h>

namespace dba {

// The admission test, by every wave for itself: with the pose-level skyline fpose (first pose a pose is coupled with) made
// monotone, every column's last row must lie inside the window of its step's tile column: row < 16 (s >> 2) + 16 NT.
// Returns the smallest NT in {3, 4, 5} (<= max_nt) that admits the system, or 0.  (The reduced system couples two poses whenever
// they see the same source frame's depths: a covisibility graph of radius r gives a band of 2 r poses -- 4 on the 25-KF window,
// 8-10 on BASELINE's 64-KF / 512-edge graph, which needs the 80-row window.)
__device__ __forceinline__ int ba_solve_wave_admits(const int *__restrict__ fpose, int n, int lane, int max_nt) {
  const int P = n / 6;
  if (!fpose || P > 64 || n != 6 * P) return 0;
  int g = (lane < P) ? fpose[lane] : 0x7fffffff;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {  // suffix minimum: fill-in keeps the skyline monotone
    const int o = __shfl_down(g, off, 64);
    if (lane + off < 64) g = min(g, o);
  }
  int last = lane;  // last(q) = the last pose p with g[p] <= q (g is non-decreasing)
  for (int p = 0; p < P; p++) {
    const int gp = __builtin_amdgcn_readlane(g, p);
    if (gp <= lane) last = max(last, p);
  }
  const int np = (n + 15) & ~15, S = np >> 2;
  bool ok3 = true, ok4 = true, ok5 = true;
  for (int base = 0; base < S; base += 64) {   // (uniform trip count: the shuffle below is executed by all lanes)
    const int s = base + lane, c = 4 * s;
    const int q3 = min(min(c + 3, n - 1) / 6, P - 1);
    const int lastrow = 6 * __shfl(last, q3, 64) + 5;
    const bool live = (s < S) && (c < n);
    ok3 = ok3 && (!live || lastrow <= 16 * (s >> 2) + 47);
    ok4 = ok4 && (!live || lastrow <= 16 * (s >> 2) + 63);
    ok5 = ok5 && (!live || lastrow <= 16 * (s >> 2) + 79);
  }
  if (max_nt >= 3 && __ballot(!ok3) == 0ull) return 3;
  if (max_nt >= 4 && __ballot(!ok4) == 0ull) return 4;
  if (max_nt >= 5 && __ballot(!ok5) == 0ull) return 5;
  return 0;
}

}  // namespace dba
