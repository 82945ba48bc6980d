// Dense bundle adjustment for gfx950 (MI355X): device-resident Gauss-Newton iteration.
//
// Replaces, behind the C ABI of include/dba_hip.h, the reference's
//   projective_transform_kernel / accum_kernel / EEt6x6_kernel / Ev6x1_kernel / EvT6x1_kernel /
//   pose_retr_kernel / disp_retr_kernel and the host-side SparseBlock / schur_block / Eigen solve
//   (/root/reference/src/droid_kernels.cu:220-468, :899-1160, :1162-1512).
//
// Design (not a translation of the CUDA kernels):
//   * the unit of work is a SOURCE FRAME slice, not an edge: a wave owns 64 pixels of one source
//     frame and walks that frame's out-edges, so the per-frame sums C = sum Cii, w = sum bz,
//     Ei = sum Eii (the reference's three accum_cuda round trips) stay in registers and
//     Eii is never materialised;
//   * the 78+12 per-edge J^T W J sums are folded across the wave with DPP row-shift adds
//     (no LDS, no barriers) and written as per-wave partials;
//   * index sets (kx, per-frame edge lists) are built once per call on the device: no D2H;
//   * the reduced camera system is accumulated in float64 with hardware f64 atomics (the reference
//     sums the same f32 blocks in double on the host) and solved by a single-workgroup
//     LDS-resident blocked Cholesky in float64.
#include "ba_kernels.h"

namespace dba {

// ---------------------------------------------------------------------------------------------
// stage 0: index sets
// ---------------------------------------------------------------------------------------------
// One workgroup. LDS: flag[B] | cnt[Mmax+1] | scan[1024]
__global__ __launch_bounds__(1024) void ba_prepare_kernel(const int64_t *__restrict__ ii,
                                                          const int64_t *__restrict__ jj, int N, int B,
                                                          int t0, int t1, BaTables T) {
  extern __shared__ int sm[];
  int *flag = sm;
  int *cnt = sm + B;
  int *scan = cnt + T.Mmax + 1;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int P = t1 - t0;

  for (int f = tid; f < B; f += nt) flag[f] = 0;
  __syncthreads();
  for (int p = tid; p < P; p += nt) {
    const int f = t0 + p;
    if (f >= 0 && f < B) flag[f] = 1;
  }
  for (int n = tid; n < N; n += nt) {
    const int f = (int)ii[n];
    if (f >= 0 && f < B) flag[f] = 1;
  }
  __syncthreads();

  // exclusive scan of flag over frames: each thread owns a contiguous run
  const int per = (B + nt - 1) / nt;
  const int lo = min(tid * per, B), hi = min(lo + per, B);
  int c = 0;
  for (int f = lo; f < hi; f++) c += flag[f];
  scan[tid] = c;
  __syncthreads();
  for (int off = 1; off < nt; off <<= 1) {  // Hillis-Steele inclusive scan
    const int v = (tid >= off) ? scan[tid - off] : 0;
    __syncthreads();
    scan[tid] += v;
    __syncthreads();
  }
  int slot = scan[tid] - c;
  const int M = scan[nt - 1];
  for (int f = lo; f < hi; f++) {
    if (flag[f]) {
      if (slot < T.Mmax) T.kx[slot] = f;
      T.frame_slot[f] = (slot < T.Mmax) ? slot : -1;
      flag[f] = slot + 1;  // keep slot+1 in LDS for the passes below
      slot++;
    } else {
      T.frame_slot[f] = -1;
    }
  }
  for (int m = tid; m <= T.Mmax; m += nt) cnt[m] = 0;
  __syncthreads();
  if (tid == 0) {
    T.meta[0] = min(M, T.Mmax);
    T.meta[1] = 0;
    T.meta[2] = (M > T.Mmax) ? 1 : 0;
  }

  // out-edges per slot
  for (int n = tid; n < N; n += nt) {
    const int f = (int)ii[n];
    if (f >= 0 && f < B && flag[f] > 0 && flag[f] <= T.Mmax) atomicAdd(&cnt[flag[f] - 1], 1);
  }
  __syncthreads();
  if (tid == 0) {  // Mmax is small (<= P+N): a serial exclusive scan is a few hundred cycles
    int run = 0;
    for (int m = 0; m < T.Mmax; m++) {
      const int v = cnt[m];
      cnt[m] = run;
      T.eoff[m] = run;
      run += v;
    }
    cnt[T.Mmax] = run;
    T.eoff[T.Mmax] = run;
  }
  __syncthreads();
  // ascending-n fill: position = #earlier edges with the same source frame
  for (int n = tid; n < N; n += nt) {
    const int f = (int)ii[n];
    if (!(f >= 0 && f < B && flag[f] > 0 && flag[f] <= T.Mmax)) continue;
    int rank = 0;
    for (int q = 0; q < n; q++) rank += ((int)ii[q] == f);
    T.elist[cnt[flag[f] - 1] + rank] = n;
  }
}

// ---------------------------------------------------------------------------------------------
// stage 1: fused linearisation per (source frame, 64-pixel wave slice)
// ---------------------------------------------------------------------------------------------
// Per-edge partial layout (HP_STRIDE floats per wave): [0,78) lower triangle of the 12x12
// (Ji | Jj) normal matrix, row-major (a >= b, index a(a+1)/2 + b); [78,84) vi; [84,90) vj.

struct PixelLin {
  float Ju[12], Jv[12];  // rows of the 2x12 Jacobian wrt (pose i | pose j)
  float Jzu, Jzv;        // wrt inverse depth of the source pixel
  float ru, rv, wu, wv;
};

__device__ __forceinline__ void linearize_pixel(float u, float v, float disp, float tu, float tv,
                                                float wgt_u, float wgt_v, const float *intr,
                                                const float *tij, const Rot3 &R, bool stereo,
                                                PixelLin &L, float &Cii, float &bz) {
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const float X0 = (u - cx) / fx, X1 = (v - cy) / fy;
  // Xj = R [X0 X1 1]^T + disp * tij ; Xj[3] = disp   (actSE3, droid_kernels.cu:73-80)
  const float x = fmaf(disp, tij[0], fmaf(R.r[0], X0, fmaf(R.r[1], X1, R.r[2])));
  const float y = fmaf(disp, tij[1], fmaf(R.r[3], X0, fmaf(R.r[4], X1, R.r[5])));
  const float z = fmaf(disp, tij[2], fmaf(R.r[6], X0, fmaf(R.r[7], X1, R.r[8])));
  const float h = disp;
  const bool close = z < 0.25f;  // MIN_DEPTH, droid_kernels.cu:29,346-350
  const float d = close ? 0.f : 1.f / z;
  const float d2 = d * d;
  float wu = close ? 0.f : 0.001f * wgt_u;
  float wv = close ? 0.f : 0.001f * wgt_v;
  L.ru = tu - fmaf(fx * d, x, cx);
  L.rv = tv - fmaf(fy * d, y, cy);

  float *Jju = L.Ju + 6, *Jjv = L.Jv + 6;
  Jju[0] = fx * (h * d);
  Jju[1] = 0.f;
  Jju[2] = fx * (-x * h * d2);
  Jju[3] = fx * (-x * y * d2);
  Jju[4] = fx * fmaf(x * x, d2, 1.f);
  Jju[5] = fx * (-y * d);
  Jjv[0] = 0.f;
  Jjv[1] = fy * (h * d);
  Jjv[2] = fy * (-y * h * d2);
  Jjv[3] = fy * (-fmaf(y * y, d2, 1.f));
  Jjv[4] = fy * (x * y * d2);
  Jjv[5] = fy * (x * d);
  L.Jzu = fx * (tij[0] * d - tij[2] * (x * d2));
  L.Jzv = fy * (tij[1] * d - tij[2] * (y * d2));

  // depth block uses the real weight even on stereo edges (droid_kernels.cu:363-367,396-400)
  Cii = wu * L.Jzu * L.Jzu + wv * L.Jzv * L.Jzv;
  bz = wu * L.ru * L.Jzu + wv * L.rv * L.Jzv;
  if (stereo) { wu = 0.f; wv = 0.f; }
  L.wu = wu;
  L.wv = wv;

  // Ji = -Ad(Gij)^T Jj :  tau part -R^T a_tau ; phi part -(R^T a_phi + R^T (a_tau x t))
#pragma unroll
  for (int c = 0; c < 2; c++) {
    const float *a = c ? Jjv : Jju;
    float *o = c ? L.Jv : L.Ju;
    const float c0 = a[1] * tij[2] - a[2] * tij[1] + a[3];
    const float c1 = a[2] * tij[0] - a[0] * tij[2] + a[4];
    const float c2 = a[0] * tij[1] - a[1] * tij[0] + a[5];
    o[0] = -(R.r[0] * a[0] + R.r[3] * a[1] + R.r[6] * a[2]);
    o[1] = -(R.r[1] * a[0] + R.r[4] * a[1] + R.r[7] * a[2]);
    o[2] = -(R.r[2] * a[0] + R.r[5] * a[1] + R.r[8] * a[2]);
    o[3] = -(R.r[0] * c0 + R.r[3] * c1 + R.r[6] * c2);
    o[4] = -(R.r[1] * c0 + R.r[4] * c1 + R.r[7] * c2);
    o[5] = -(R.r[2] * c0 + R.r[5] * c1 + R.r[8] * c2);
  }
}

// compile-time loop helper: reduce value #L across the wave and deposit it in lane L%64 of acc[L/64]
template <int L>
__device__ __forceinline__ void reduce_deposit(float val, float &acc0, float &acc1) {
  const float red = wave_sum_to_lane63(val);
  const int lane = lane_id();
  if constexpr (L < 64)
    acc0 = deposit_lane63<L>(acc0, red, lane);
  else
    acc1 = deposit_lane63<L - 64>(acc1, red, lane);
}

template <int A, int B_>
struct HLoop {
  __device__ __forceinline__ static void run(const PixelLin &L, float &acc0, float &acc1) {
    constexpr int idx = A * (A + 1) / 2 + B_;
    const float val = L.wu * L.Ju[A] * L.Ju[B_] + L.wv * L.Jv[A] * L.Jv[B_];
    reduce_deposit<idx>(val, acc0, acc1);
    if constexpr (B_ < A)
      HLoop<A, B_ + 1>::run(L, acc0, acc1);
    else if constexpr (A < 11)
      HLoop<A + 1, 0>::run(L, acc0, acc1);
  }
};

template <int A>
struct VLoop {
  __device__ __forceinline__ static void run(const PixelLin &L, float &acc0, float &acc1) {
    const float val = L.wu * L.ru * L.Ju[A] + L.wv * L.rv * L.Jv[A];  // A<6: vi, A>=6: vj
    reduce_deposit<78 + A>(val, acc0, acc1);
    if constexpr (A < 11) VLoop<A + 1>::run(L, acc0, acc1);
  }
};

__global__ __launch_bounds__(256) void ba_linearize_kernel(
    const float *__restrict__ poses, const float *__restrict__ disps, const float *__restrict__ intrinsics,
    const float *__restrict__ disps_sens, const float *__restrict__ targets,
    const float *__restrict__ weights, const float *__restrict__ eta, int eta_rows,
    const int64_t *__restrict__ jj, const uint8_t *__restrict__ frame_owned, int N, int HW, int wd,
    int t0, int P, float alpha, BaTables T, BaBuffers W) {
  const int m = blockIdx.y;
  if (m == T.Mmax) {  // extra row of workgroups: clear the reduced camera system for stage 2
    const int n6 = 6 * P;
    const size_t total = (size_t)n6 * n6;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
      W.H[i] = 0.0;
    if (blockIdx.x == 0)
      for (int i = threadIdx.x; i < n6; i += blockDim.x) W.b[i] = 0.0;
    return;
  }
  const int M = T.meta[0];
  if (m >= M) return;
  const int frame = T.kx[m];
  if (frame_owned && !frame_owned[frame]) return;

  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = k < HW;
  const int kc = active ? k : 0;
  const int wave_global = blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x >> 6);  // pixel slice id
  const int nparts = W.nparts;
  const int lane = lane_id();

  float intr[4] = {intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]};
  const float u = (float)(kc % wd), v = (float)(kc / wd);
  const float disp = disps[(size_t)frame * HW + kc];

  float Csum = 0.f, wsum = 0.f;
  float Ei[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const int e0 = T.eoff[m], e1 = T.eoff[m + 1];
  for (int e = e0; e < e1; e++) {
    const int n = T.elist[e];
    const int jx = (int)jj[n];
    float tij[3], qij[4];
    edge_pose(poses, frame, jx, tij, qij);
    const Rot3 R = quat_to_rot(qij);

    const size_t tb = (size_t)n * 2 * HW + kc;
    const float tu = targets[tb], tv = targets[tb + HW];
    const float wgu = active ? weights[tb] : 0.f, wgv = active ? weights[tb + HW] : 0.f;

    PixelLin L;
    float Cii, bz;
    linearize_pixel(u, v, disp, tu, tv, wgu, wgv, intr, tij, R, frame == jx, L, Cii, bz);
    Csum += Cii;
    wsum += bz;

    float *Eij = W.E + ((size_t)(P + n) * 6) * HW + kc;
#pragma unroll
    for (int c = 0; c < 6; c++) {
      const float wJzu = L.wu * L.Jzu, wJzv = L.wv * L.Jzv;
      Ei[c] += wJzu * L.Ju[c] + wJzv * L.Jv[c];
      if (active) Eij[(size_t)c * HW] = wJzu * L.Ju[6 + c] + wJzv * L.Jv[6 + c];
    }

    float acc0 = 0.f, acc1 = 0.f;
    HLoop<0, 0>::run(L, acc0, acc1);
    VLoop<0>::run(L, acc0, acc1);
    float *hp = W.Hpart + ((size_t)n * nparts + wave_global) * HP_STRIDE;
    hp[lane] = acc0;
    if (lane < HP_STRIDE - 64) hp[64 + lane] = acc1;
  }

  if (active) {
    const size_t fk = (size_t)frame * HW + k;
    const float ds = disps_sens[fk];
    const float mm = (ds > 0.f) ? 1.f : 0.f;
    const float et = eta[(size_t)(eta_rows == 1 ? 0 : m) * HW + k];
    const float C = (Csum + mm * alpha) + (1.f - mm) * et;            // droid_kernels.cu:1476
    const float w = wsum - (mm * alpha) * (disp - ds);                // :1477
    W.Q[(size_t)m * HW + k] = 1.f / C;                                // :1478
    W.w[(size_t)m * HW + k] = w;
    const int p = frame - t0;
    if (p >= 0 && p < P) {
      float *Er = W.E + ((size_t)p * 6) * HW + k;
#pragma unroll
      for (int c = 0; c < 6; c++) Er[(size_t)c * HW] = Ei[c];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// stage 2: reduced camera system in float64
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_add_f64(double *p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// workgroups [0, P+N): one row r1 of E each -> its Schur products with the later rows of the same
// source frame.  workgroups [P+N, P+N+ceil(N/2)): fold the per-wave J^T W J partials of two edges
// and scatter the pose blocks.
__global__ __launch_bounds__(256) void ba_reduce_kernel(const int64_t *__restrict__ ii,
                                                        const int64_t *__restrict__ jj,
                                                        const uint8_t *__restrict__ frame_owned, int N,
                                                        int HW, int t0, int P, int motion_only,
                                                        BaTables T, BaBuffers W) {
  __shared__ float red[4][44];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n6 = 6 * P;
  const int R = P + N;

  if ((int)blockIdx.x >= R) {
    // ---- pose-block assembly (SparseBlock::update_lhs / update_rhs, :1176-1218, :1457-1462)
    const int n = 2 * ((int)blockIdx.x - R) + (tid >> 7);
    const int l = tid & 127;
    if (n >= N || l >= 90) return;
    const int src = (int)ii[n];
    if (frame_owned && !frame_owned[src]) return;
    const float *hp = W.Hpart + (size_t)n * W.nparts * HP_STRIDE + l;
    double s = 0.0;
    for (int part = 0; part < W.nparts; part++) s += (double)hp[(size_t)part * HP_STRIDE];
    const int i = src - t0, j = (int)jj[n] - t0;
    const bool iv = (i >= 0 && i < P), jv = (j >= 0 && j < P);
    if (l < 78) {
      int a = 0;
      while ((a + 1) * (a + 2) / 2 <= l) a++;
      const int b = l - a * (a + 1) / 2;
      if (a < 6) {  // Hii
        if (iv) {
          atomic_add_f64(&W.H[(size_t)(6 * i + a) * n6 + 6 * i + b], s);
          if (a != b) atomic_add_f64(&W.H[(size_t)(6 * i + b) * n6 + 6 * i + a], s);
        }
      } else if (b < 6) {  // Hji[a-6][b] and Hij[b][a-6]
        if (iv && jv) {
          atomic_add_f64(&W.H[(size_t)(6 * j + a - 6) * n6 + 6 * i + b], s);
          atomic_add_f64(&W.H[(size_t)(6 * i + b) * n6 + 6 * j + a - 6], s);
        }
      } else {  // Hjj
        if (jv) {
          atomic_add_f64(&W.H[(size_t)(6 * j + a - 6) * n6 + 6 * j + b - 6], s);
          if (a != b) atomic_add_f64(&W.H[(size_t)(6 * j + b - 6) * n6 + 6 * j + a - 6], s);
        }
      }
    } else if (l < 84) {
      if (iv) atomic_add_f64(&W.b[6 * i + (l - 78)], s);
    } else {
      if (jv) atomic_add_f64(&W.b[6 * j + (l - 84)], s);
    }
    return;
  }

  if (motion_only) return;

  // ---- Schur complement rows (schur_block + EEt6x6_kernel + Ev6x1_kernel, :1046-1138, :1297-1391)
  const int r1 = blockIdx.x;
  int frame, tgt1, first_partner;  // partners: r1 itself, then edges elist[first_partner ..)
  if (r1 < P) {
    frame = t0 + r1;
    tgt1 = r1;
  } else {
    frame = (int)ii[r1 - P];
    tgt1 = (int)jj[r1 - P] - t0;
  }
  if (tgt1 < 0 || tgt1 >= P) return;
  if (frame < 0 || frame >= T.B) return;
  const int m = T.frame_slot[frame];
  if (m < 0) return;
  if (frame_owned && !frame_owned[frame]) return;
  const int e0 = T.eoff[m], e1 = T.eoff[m + 1];
  if (r1 < P) {
    first_partner = e0;
  } else {
    first_partner = e1;
    for (int e = e0; e < e1; e++)
      if (T.elist[e] == r1 - P) { first_partner = e + 1; break; }
  }

  const float *E1 = W.E + (size_t)r1 * 6 * HW;
  const float *Qm = W.Q + (size_t)m * HW;
  const float *wm = W.w + (size_t)m * HW;

  for (int pe = first_partner - 1; pe < e1; pe++) {
    const bool self = (pe == first_partner - 1);
    const int r2 = self ? r1 : P + T.elist[pe];
    const int tgt2 = self ? tgt1 : (int)jj[r2 - P] - t0;
    if (tgt2 < 0 || tgt2 >= P) continue;
    const float *E2 = W.E + (size_t)r2 * 6 * HW;

    float acc[36];
    float sv[6];
#pragma unroll
    for (int c = 0; c < 36; c++) acc[c] = 0.f;
#pragma unroll
    for (int c = 0; c < 6; c++) sv[c] = 0.f;
    for (int k = tid; k < HW; k += 256) {
      const float q = Qm[k];
      float e1v[6], e2v[6];
#pragma unroll
      for (int c = 0; c < 6; c++) {
        e1v[c] = E1[(size_t)c * HW + k] * q;
        e2v[c] = E2[(size_t)c * HW + k];
      }
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = 0; b < 6; b++) acc[a * 6 + b] = fmaf(e1v[a], e2v[b], acc[a * 6 + b]);
      if (self) {
        const float wk = wm[k];
#pragma unroll
        for (int c = 0; c < 6; c++) sv[c] = fmaf(e1v[c], wk, sv[c]);
      }
    }
    __syncthreads();  // protect `red` from the previous partner's readers
#pragma unroll
    for (int c = 0; c < 36; c++) {
      const float r = wave_sum_to_lane63(acc[c]);
      if (lane == 63) red[wv][c] = r;
    }
    if (self) {
#pragma unroll
      for (int c = 0; c < 6; c++) {
        const float r = wave_sum_to_lane63(sv[c]);
        if (lane == 63) red[wv][36 + c] = r;
      }
    }
    __syncthreads();
    if (tid < 36) {
      const double s = (double)red[0][tid] + (double)red[1][tid] + (double)red[2][tid] + (double)red[3][tid];
      const int a = tid / 6, b = tid % 6;
      atomic_add_f64(&W.H[(size_t)(6 * tgt1 + a) * n6 + 6 * tgt2 + b], -s);
      if (!self) atomic_add_f64(&W.H[(size_t)(6 * tgt2 + b) * n6 + 6 * tgt1 + a], -s);
    } else if (self && tid < 42) {
      const double s = (double)red[0][tid] + (double)red[1][tid] + (double)red[2][tid] + (double)red[3][tid];
      atomic_add_f64(&W.b[6 * tgt1 + (tid - 36)], -s);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// stage 4: back-substitution + retraction
// ---------------------------------------------------------------------------------------------
// expSE3 / retrSE3 (droid_kernels.cu:113-178, :922-940); quaternion deliberately not renormalised.
__device__ void retract_pose(float *pose, const float *xi) {
  const float *tau = xi, *phi = xi + 3;
  const float th2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const float th4 = th2 * th2;
  const float th = sqrtf(th2);
  float imag, real;
  if (th2 < 1e-8f) {
    imag = 0.5f - (1.0f / 48.0f) * th2 + (1.0f / 3840.0f) * th4;
    real = 1.0f - (1.0f / 8.0f) * th2 + (1.0f / 384.0f) * th4;
  } else {
    imag = sinf(0.5f * th) / th;
    real = cosf(0.5f * th);
  }
  const float dq[4] = {imag * phi[0], imag * phi[1], imag * phi[2], real};
  float dt[3] = {tau[0], tau[1], tau[2]};
  if (th > 1e-4f) {
    const float a = (1.f - cosf(th)) / th2;
    const float b = (th - sinf(th)) / (th * th2);
    const float c1[3] = {phi[1] * tau[2] - phi[2] * tau[1], phi[2] * tau[0] - phi[0] * tau[2],
                         phi[0] * tau[1] - phi[1] * tau[0]};
    const float c2[3] = {phi[1] * c1[2] - phi[2] * c1[1], phi[2] * c1[0] - phi[0] * c1[2],
                         phi[0] * c1[1] - phi[1] * c1[0]};
#pragma unroll
    for (int c = 0; c < 3; c++) dt[c] += a * c1[c] + b * c2[c];
  }
  const float t[3] = {pose[0], pose[1], pose[2]};
  const float q[4] = {pose[3], pose[4], pose[5], pose[6]};
  float q1[4], t1[3];
  q1[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
  q1[1] = dq[3] * q[1] + dq[1] * q[3] + dq[2] * q[0] - dq[0] * q[2];
  q1[2] = dq[3] * q[2] + dq[2] * q[3] + dq[0] * q[1] - dq[1] * q[0];
  q1[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
  quat_rotate(dq, t, t1);
  pose[0] = t1[0] + dt[0];
  pose[1] = t1[1] + dt[1];
  pose[2] = t1[2] + dt[2];
  pose[3] = q1[0];
  pose[4] = q1[1];
  pose[5] = q1[2];
  pose[6] = q1[3];
}

__global__ __launch_bounds__(256) void ba_update_kernel(float *__restrict__ poses, float *__restrict__ disps,
                                                        const int64_t *__restrict__ jj,
                                                        const uint8_t *__restrict__ frame_owned, int HW,
                                                        int t0, int P, int update_poses, int update_disps,
                                                        float *__restrict__ dz_out, BaTables T,
                                                        BaBuffers W) {
  const int m = blockIdx.y;
  if (m == T.Mmax) {  // pose retraction: T_k <- Exp(dx_k) T_k for k in [t0, t1)
    if (!update_poses || blockIdx.x != 0) return;
    for (int p = threadIdx.x; p < P; p += blockDim.x) retract_pose(poses + 7 * (t0 + p), W.dx + 6 * p);
    return;
  }
  if (!update_disps) return;
  const int M = T.meta[0];
  if (m >= M) return;
  const int frame = T.kx[m];
  if (frame_owned && !frame_owned[frame]) return;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= HW) return;

  float acc = 0.f;
  const int p = frame - t0;
  // EvT6x1_kernel skips rows whose pose index is <= 0 or >= P (droid_kernels.cu:1150)
  if (p > 0 && p < P) {
    const float *Er = W.E + ((size_t)p * 6) * HW + k;
    const float *x = W.dx + 6 * p;
    float dw = 0.f;
#pragma unroll
    for (int c = 0; c < 6; c++) dw += Er[(size_t)c * HW] * x[c];
    acc += dw;
  }
  const int e0 = T.eoff[m], e1 = T.eoff[m + 1];
  for (int e = e0; e < e1; e++) {
    const int n = T.elist[e];
    const int tgt = (int)jj[n] - t0;
    if (tgt <= 0 || tgt >= P) continue;
    const float *Er = W.E + ((size_t)(P + n) * 6) * HW + k;
    const float *x = W.dx + 6 * tgt;
    float dw = 0.f;
#pragma unroll
    for (int c = 0; c < 6; c++) dw += Er[(size_t)c * HW] * x[c];
    acc += dw;
  }
  const size_t mk = (size_t)m * HW + k;
  const float dz = W.Q[mk] * (W.w[mk] - acc);  // :1495
  const size_t fk = (size_t)frame * HW + k;
  disps[fk] = disps[fk] + dz;                  // disp_retr_kernel :988
  if (dz_out) dz_out[mk] = dz;
}

__global__ void ba_copy_dx_kernel(const double *__restrict__ src, float *__restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)src[i];
}

__global__ void ba_copy_f32_kernel(const float *__restrict__ src, float *__restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

}  // namespace dba
