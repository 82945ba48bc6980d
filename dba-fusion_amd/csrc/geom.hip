// Geometry kernels of the hot path's neighbours, gfx950:
//   dba_reproject       <- pops.projective_transform (dbaf/geom/projective_ops.py:96-125), the ~15 small
//                          torch/lietorch kernels issued by DepthVideo.reproject on every update
//   dba_frame_distance  <- frame_distance_kernel (src/droid_kernels.cu:562-702)
//   dba_projmap         <- projmap_kernel        (src/droid_kernels.cu:471-560)
//   dba_iproj           <- iproj_kernel          (src/droid_kernels.cu:824-895)
//   dba_depth_filter    <- depth_filter_kernel   (src/droid_kernels.cu:706-820)
// All are one pass over [N or B, ht*wd] with 4-20 bytes per pixel: HBM/latency trivial, so the only
// design rule is "one launch, coalesced, enough workgroups" (grid = pixels x edges).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "reproj.h"

namespace dba {

__device__ __forceinline__ void act_point(const Rot3 &R, const float *t, float X0, float X1, float d,
                                          float &x, float &y, float &z) {
  x = fmaf(d, t[0], fmaf(R.r[0], X0, fmaf(R.r[1], X1, R.r[2])));
  y = fmaf(d, t[1], fmaf(R.r[3], X0, fmaf(R.r[4], X1, R.r[5])));
  z = fmaf(d, t[2], fmaf(R.r[6], X0, fmaf(R.r[7], X1, R.r[8])));
}

__global__ __launch_bounds__(256) void reproject_kernel(const float *__restrict__ poses,
                                                        const float *__restrict__ disps,
                                                        const float *__restrict__ intr_b4,
                                                        const int64_t *__restrict__ ii,
                                                        const int64_t *__restrict__ jj, int HW, int wd,
                                                        float2 *__restrict__ coords, float *__restrict__ valid) {
  const int n = blockIdx.y;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= HW) return;
  const int ix = (int)ii[n], jx = (int)jj[n];
  // (reproj.h: the arithmetic the lookup kernels repeat in their prologue when they take the reprojection along --
  // dba_corr_lookup_reproject_sheared -- so that the fused call equals this kernel + the lookup bit for bit)
  const EdgeGeom G = edge_geom(poses, intr_b4, ix, jx);  // stereo edges: (-0.1,0,0), identity (projective_ops.py:105)
  const float u = (float)(k % wd), v = (float)(k / wd);
  float ok;
  coords[(size_t)n * HW + k] = reproject_pixel(G, u, v, disps[(size_t)ix * HW + k], ok);
  valid[(size_t)n * HW + k] = ok;  // X0.z == 1 > MIN_DEPTH always (:112)
}

__global__ __launch_bounds__(256) void frame_distance_kernel(const float *__restrict__ poses,
                                                             const float *__restrict__ disps,
                                                             const float *__restrict__ intr,
                                                             const int64_t *__restrict__ ii,
                                                             const int64_t *__restrict__ jj, int HW, int wd,
                                                             float beta, float *__restrict__ dist) {
  __shared__ float red[3][4];
  const int n = blockIdx.x;
  const int ix = (int)ii[n], jx = (int)jj[n];
  float tij[3], qij[4];
  rel_pose(poses + 7 * ix, poses + 7 * jx, tij, qij);
  const Rot3 R = quat_to_rot(qij);
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  float accum = 0.f, valid = 0.f, total = 0.f;
  for (int k = threadIdx.x; k < HW; k += blockDim.x) {
    const float u = (float)(k % wd), v = (float)(k / wd);
    const float X0 = (u - cx) / fx, X1 = (v - cy) / fy;
    const float d = disps[(size_t)ix * HW + k];
    float x, y, z;
    act_point(R, tij, X0, X1, d, x, y, z);
    float du = fx * (x / z) + cx - u, dv = fy * (y / z) + cy - v;
    float r = sqrtf(du * du + dv * dv);
    total += beta;
    if (z > 0.25f) { accum += beta * r; valid += beta; }
    // translation-only flow (:662-680)
    x = X0 + d * tij[0];
    y = X1 + d * tij[1];
    z = 1.0f + d * tij[2];
    du = fx * (x / z) + cx - u;
    dv = fy * (y / z) + cy - v;
    r = sqrtf(du * du + dv * dv);
    total += (1.f - beta);
    if (z > 0.25f) { accum += (1.f - beta) * r; valid += (1.f - beta); }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float a = wave_sum(accum), vv = wave_sum(valid), tt = wave_sum(total);
  if (lane == 0) { red[0][wv] = a; red[1][wv] = vv; red[2][wv] = tt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float A = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const float V = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const float T = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    dist[n] = ((double)V / ((double)T + 1e-8) < 0.75) ? 1000.0f : A / V;  // :700
  }
}

__global__ __launch_bounds__(256) void projmap_kernel(const float *__restrict__ poses,
                                                      const float *__restrict__ disps,
                                                      const float *__restrict__ intr,
                                                      const int64_t *__restrict__ ii,
                                                      const int64_t *__restrict__ jj, int HW, int wd,
                                                      float *__restrict__ coords, float *__restrict__ valid) {
  const int n = blockIdx.y;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= HW) return;
  const int ix = (int)ii[n], jx = (int)jj[n];
  float tij[3], qij[4];
  rel_pose(poses + 7 * ix, poses + 7 * jx, tij, qij);
  const Rot3 R = quat_to_rot(qij);
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const float u = (float)(k % wd), v = (float)(k / wd);
  float x, y, z;
  act_point(R, tij, (u - cx) / fx, (v - cy) / fy, disps[(size_t)ix * HW + k], x, y, z);
  float *c = coords + ((size_t)n * HW + k) * 3;
  float ou = u, ov = v;
  if (z > 0.01f) { ou = fx * (x / z) + cx; ov = fy * (y / z) + cy; }
  c[0] = ou;
  c[1] = ov;
  c[2] = 0.f;  // the reference allocates 3 channels and writes 2 (:549-554, :1718)
  valid[(size_t)n * HW + k] = (z > 0.25f) ? 1.0f : 0.0f;
}

__global__ __launch_bounds__(256) void iproj_kernel(const float *__restrict__ poses,
                                                    const float *__restrict__ disps,
                                                    const float *__restrict__ intr, int HW, int wd,
                                                    float *__restrict__ points) {
  const int b = blockIdx.y;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= HW) return;
  const float *t = poses + 7 * b;
  const Rot3 R = quat_to_rot(poses + 7 * b + 3);
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const float u = (float)(k % wd), v = (float)(k / wd);
  const float d = disps[(size_t)b * HW + k];
  float x, y, z;
  act_point(R, t, (u - cx) / fx, (v - cy) / fy, d, x, y, z);
  float *p = points + ((size_t)b * HW + k) * 3;
  p[0] = x / d;
  p[1] = y / d;
  p[2] = z / d;
}

__global__ __launch_bounds__(256) void depth_filter_kernel(const float *__restrict__ poses,
                                                           const float *__restrict__ disps,
                                                           const float *__restrict__ intr,
                                                           const int64_t *__restrict__ inds,
                                                           const float *__restrict__ thresh, int nbuf, int ht,
                                                           int wd, float *__restrict__ counter) {
  // one lane per (keyframe, pixel); the six neighbours are visited in order so no atomics are needed
  const int b = blockIdx.y;
  const int HW = ht * wd;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= HW) return;
  const int ix = (int)inds[b];
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const float t = thresh[b];
  const float u = (float)(k % wd), v = (float)(k / wd);
  const float X0 = (u - cx) / fx, X1 = (v - cy) / fy;
  const float di = disps[(size_t)ix * HW + k];
  float count = 0.f;
  for (int neigh = 0; neigh < 6; neigh++) {
    const int jx = (neigh < 3) ? ix - neigh - 1 : ix + neigh;  // :740
    if (jx < 0 || jx >= nbuf) continue;
    float tij[3], qij[4];
    rel_pose(poses + 7 * ix, poses + 7 * jx, tij, qij);
    const Rot3 R = quat_to_rot(qij);
    float x, y, z;
    act_point(R, tij, X0, X1, di, x, y, z);
    const float uj = fx * (x / z) + cx, vj = fy * (y / z) + cy;
    const float dj = di / z;
    const int u0 = (int)floorf(uj), v0 = (int)floorf(vj);
    if (u0 >= 0 && v0 >= 0 && u0 < wd - 1 && v0 < ht - 1) {
      const float *dp = disps + (size_t)jx * HW + (size_t)v0 * wd + u0;
      const double idj = 1.0 / (double)dj;  // abs(1.0/dj - 1.0/d00) is double arithmetic (:813-817)
      const double tt = (double)t;
      if (fabs(idj - 1.0 / (double)dp[0]) < tt) count += 1.f;
      else if (fabs(idj - 1.0 / (double)dp[1]) < tt) count += 1.f;
      else if (fabs(idj - 1.0 / (double)dp[wd]) < tt) count += 1.f;
      else if (fabs(idj - 1.0 / (double)dp[wd + 1]) < tt) count += 1.f;
    }
  }
  counter[(size_t)b * HW + k] += count;
}

}  // namespace dba

using namespace dba;

extern "C" {

int dba_reproject(const float *poses, const float *disps, const float *intrinsics_b4, const int64_t *ii,
                  const int64_t *jj, int N, int ht, int wd, float *coords, float *valid, dba_stream_t stream) {
  if (N < 0 || ht <= 0 || wd <= 0) return DBA_ERR_ARG;
  if (N == 0) return DBA_OK;
  const int HW = ht * wd;
  hipLaunchKernelGGL(reproject_kernel, dim3((HW + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, poses, disps,
                     intrinsics_b4, ii, jj, HW, wd, reinterpret_cast<float2 *>(coords), valid);
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

int dba_frame_distance(const float *poses, const float *disps, const float *intrinsics, const int64_t *ii,
                       const int64_t *jj, int N, int ht, int wd, float beta, float *dist, dba_stream_t stream) {
  if (N < 0 || ht <= 0 || wd <= 0) return DBA_ERR_ARG;
  if (N == 0) return DBA_OK;
  hipLaunchKernelGGL(frame_distance_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, poses, disps, intrinsics,
                     ii, jj, ht * wd, wd, beta, dist);
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

int dba_projmap(const float *poses, const float *disps, const float *intrinsics, const int64_t *ii,
                const int64_t *jj, int N, int ht, int wd, float *coords, float *valid, dba_stream_t stream) {
  if (N < 0 || ht <= 0 || wd <= 0) return DBA_ERR_ARG;
  if (N == 0) return DBA_OK;
  const int HW = ht * wd;
  hipLaunchKernelGGL(projmap_kernel, dim3((HW + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, poses, disps,
                     intrinsics, ii, jj, HW, wd, coords, valid);
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

int dba_iproj(const float *poses, const float *disps, const float *intrinsics, int nm, int ht, int wd,
              float *points, dba_stream_t stream) {
  if (nm < 0 || ht <= 0 || wd <= 0) return DBA_ERR_ARG;
  if (nm == 0) return DBA_OK;
  const int HW = ht * wd;
  hipLaunchKernelGGL(iproj_kernel, dim3((HW + 255) / 256, nm), dim3(256), 0, (hipStream_t)stream, poses, disps,
                     intrinsics, HW, wd, points);
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

int dba_depth_filter(const float *poses, const float *disps, const float *intrinsics, const int64_t *inds,
                     const float *thresh, int num, int nbuf, int ht, int wd, float *counter,
                     dba_stream_t stream) {
  if (num < 0 || ht <= 0 || wd <= 0) return DBA_ERR_ARG;
  if (num == 0) return DBA_OK;
  const int HW = ht * wd;
  hipLaunchKernelGGL(depth_filter_kernel, dim3((HW + 255) / 256, num), dim3(256), 0, (hipStream_t)stream, poses,
                     disps, intrinsics, inds, thresh, nbuf, ht, wd, counter);
  DBA_LAUNCH_CHECK();
  return DBA_OK;
}

}  // extern "C"
