// Damped dense Cholesky solve of the reduced camera system, float64, one workgroup.
//
// Replaces SparseBlock::solve / solveDenseD of the reference
// (/root/reference/src/droid_kernels.cu:200-218, :1248-1269: Eigen SimplicialLLT / LLT on the host,
// preceded by D2H copies of Hs, vs, S, v and followed by an H2D copy of dx): here the system never
// leaves the device.  n = 6P is 144 for a 25-keyframe window; the packed lower triangle
// (n(n+1)/2 doubles = 83.5 KB at n = 144) lives in LDS for n <= 200 and in an L2-resident global
// scratch otherwise.  Right-looking blocked LL^T with block size NB = 12 (two pose blocks):
//   diag factor (one lane, registers) -> panel solve (one row per thread) -> trailing update.
#include "ba_kernels.h"

namespace dba {

constexpr int NB = 12;
constexpr int SOLVE_THREADS = 512;  // 2 waves/SIMD -> 256 VGPRs: the register-resident 12x12 blocks do not spill

__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }  // j <= i

template <bool USE_LDS>
__global__ __launch_bounds__(SOLVE_THREADS) void ba_solve_kernel(const double *__restrict__ H,
                                                                 const double *__restrict__ bvec, int n,
                                                                 double lm, double ep,
                                                                 float *__restrict__ dx,
                                                                 int *__restrict__ meta,
                                                                 double *__restrict__ Lglobal) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // layout: [A packed (LDS mode only)] [x: n] [rdiag: n] [D: NB*NB] [invd: NB] [fail flag]
  double *A;
  double *x;
  if constexpr (USE_LDS) {
    A = smem;
    x = smem + (size_t)n * (n + 1) / 2;
  } else {
    A = Lglobal;
    x = smem;
  }
  double *rdiag = x + n;      // 1 / L_jj for every column
  double *D = rdiag + n;      // current diagonal block, row-major NB x NB
  double *invd = D + NB * NB; // reciprocal diagonal of the current block
  int *fail = (int *)(invd + NB);
  const int tid = threadIdx.x, nt = blockDim.x;

  if (tid == 0) *fail = 0;
  // load lower triangle with damping: diag += ep + lm * diag (:1252-1253)
  for (int e = tid; e < n * n; e += nt) {
    const int i = e / n, j = e - i * n;
    if (j > i) continue;
    double v = H[e];
    if (i == j) v += ep + lm * v;
    A[tri(i, j)] = v;
  }
  for (int i = tid; i < n; i += nt) x[i] = bvec[i];
  __syncthreads();

  for (int kb = 0; kb < n; kb += NB) {
    const int nb = min(NB, n - kb);
    // (1) factor the diagonal block in registers of one lane
    if (tid == 0) {
      double a[NB][NB];
#pragma unroll
      for (int i = 0; i < NB; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) a[i][j] = (i < nb) ? A[tri(kb + i, kb + j)] : (i == j ? 1.0 : 0.0);
      bool bad = false;
#pragma unroll
      for (int j = 0; j < NB; j++) {
        double d = a[j][j];
#pragma unroll
        for (int c = 0; c < j; c++) d -= a[j][c] * a[j][c];
        if (!(d > 0.0)) { bad = true; d = 1.0; }
        const double s = sqrt(d);
        const double inv = 1.0 / s;
        a[j][j] = s;
        invd[j] = inv;
        if (j < nb) rdiag[kb + j] = inv;
#pragma unroll
        for (int i = j + 1; i < NB; i++) {
          double t = a[i][j];
#pragma unroll
          for (int c = 0; c < j; c++) t -= a[i][c] * a[j][c];
          a[i][j] = t * inv;
        }
      }
      if (bad) *fail = 1;
#pragma unroll
      for (int i = 0; i < NB; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
          D[i * NB + j] = a[i][j];
          if (i < nb) A[tri(kb + i, kb + j)] = a[i][j];
        }
    }
    __syncthreads();
    const int r0 = kb + nb;  // first trailing row
    // (2) panel: rows below solve X L11^T = A21
    for (int i = r0 + tid; i < n; i += nt) {
      double xr[NB];
      const int base = tri(i, kb);
#pragma unroll
      for (int j = 0; j < NB; j++) xr[j] = (j < nb) ? A[base + j] : 0.0;
#pragma unroll
      for (int j = 0; j < NB; j++) {
        double t = xr[j];
#pragma unroll
        for (int c = 0; c < j; c++) t -= xr[c] * D[j * NB + c];
        xr[j] = t * invd[j];
      }
#pragma unroll
      for (int j = 0; j < NB; j++)
        if (j < nb) A[base + j] = xr[j];
    }
    __syncthreads();
    // (3) trailing update A22 -= L21 L21^T : wave per row, lanes across columns
    {
      const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
      for (int i = r0 + wave; i < n; i += nw) {
        double li[NB];
        const int bi = tri(i, kb);
#pragma unroll
        for (int c = 0; c < NB; c++) li[c] = (c < nb) ? A[bi + c] : 0.0;
        for (int j = r0 + lane; j <= i; j += 64) {
          const int bj = tri(j, kb);
          double s = 0.0;
#pragma unroll
          for (int c = 0; c < NB; c++) s += li[c] * ((c < nb) ? A[bj + c] : 0.0);
          A[tri(i, j)] -= s;
        }
      }
    }
    __syncthreads();
  }

  // forward substitution L y = b (blocked; the nb x nb triangle is solved in one lane's registers,
  // multiplying by the reciprocal diagonal kept from the factorisation)
  for (int kb = 0; kb < n; kb += NB) {
    const int nb = min(NB, n - kb);
    if (tid == 0) {
      double l[NB][NB], y[NB], rd[NB];
#pragma unroll
      for (int j = 0; j < NB; j++) {
        y[j] = (j < nb) ? x[kb + j] : 0.0;
        rd[j] = (j < nb) ? rdiag[kb + j] : 1.0;
#pragma unroll
        for (int c = 0; c < j; c++) l[j][c] = (j < nb) ? A[tri(kb + j, kb + c)] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < NB; j++) {
        double t = y[j];
#pragma unroll
        for (int c = 0; c < j; c++) t -= l[j][c] * y[c];
        y[j] = t * rd[j];
      }
#pragma unroll
      for (int j = 0; j < NB; j++)
        if (j < nb) x[kb + j] = y[j];
    }
    __syncthreads();
    for (int i = kb + nb + tid; i < n; i += nt) {
      double t = x[i];
      const int bi = tri(i, kb);
      for (int c = 0; c < nb; c++) t -= A[bi + c] * x[kb + c];
      x[i] = t;
    }
    __syncthreads();
  }
  // backward substitution L^T x = y (blocked, bottom-up)
  for (int kb = ((n - 1) / NB) * NB; kb >= 0; kb -= NB) {
    const int nb = min(NB, n - kb);
    if (tid == 0) {
      double l[NB][NB], y[NB], rd[NB];
#pragma unroll
      for (int j = 0; j < NB; j++) {
        y[j] = (j < nb) ? x[kb + j] : 0.0;
        rd[j] = (j < nb) ? rdiag[kb + j] : 1.0;
#pragma unroll
        for (int c = 0; c < j; c++) l[j][c] = (j < nb) ? A[tri(kb + j, kb + c)] : 0.0;
      }
#pragma unroll
      for (int j = NB - 1; j >= 0; j--) {
        double t = y[j];
#pragma unroll
        for (int c = j + 1; c < NB; c++) t -= l[c][j] * y[c];
        y[j] = t * rd[j];
      }
#pragma unroll
      for (int j = 0; j < NB; j++)
        if (j < nb) x[kb + j] = y[j];
    }
    __syncthreads();
    for (int i = tid; i < kb; i += nt) {
      double t = x[i];
      for (int c = 0; c < nb; c++) t -= A[tri(kb + c, i)] * x[kb + c];
      x[i] = t;
    }
    __syncthreads();
  }

  // non-finite results count as failure too; failure => zero update (:1263-1266)
  int bad = 0;
  for (int i = tid; i < n; i += nt)
    if (!isfinite(x[i])) bad = 1;
  if (bad) *fail = 1;
  __syncthreads();
  const int failed = *fail;
  for (int i = tid; i < n; i += nt) dx[i] = failed ? 0.f : (float)x[i];
  if (tid == 0) meta[1] = failed;
}

bool ba_solve_fits_lds(int n) {
  const size_t small = ((size_t)2 * n + NB * NB + NB + 2) * sizeof(double);
  const size_t packed = (size_t)n * (n + 1) / 2 * sizeof(double);
  return packed + small <= (size_t)SOLVE_MAX_LDS_BYTES;
}

int launch_ba_solve(const double *H, const double *b, int n, double lm, double ep, float *dx, int *meta,
                    double *Lscratch, hipStream_t stream) {
  if (n <= 0) return DBA_OK;
  const size_t small = ((size_t)2 * n + NB * NB + NB + 2) * sizeof(double);
  const size_t packed = (size_t)n * (n + 1) / 2 * sizeof(double);
  if (packed + small <= (size_t)SOLVE_MAX_LDS_BYTES) {
    static bool attr_set = false;
    if (!attr_set) {
      DBA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&ba_solve_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, SOLVE_MAX_LDS_BYTES));
      attr_set = true;
    }
    hipLaunchKernelGGL(ba_solve_kernel<true>, dim3(1), dim3(SOLVE_THREADS), packed + small, stream, H, b, n,
                       lm, ep, dx, meta, Lscratch);
  } else {
    hipLaunchKernelGGL(ba_solve_kernel<false>, dim3(1), dim3(SOLVE_THREADS), small, stream, H, b, n, lm, ep,
                       dx, meta, Lscratch);
  }
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

}  // namespace dba
