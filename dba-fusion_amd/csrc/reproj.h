// Reprojection of one pixel along one edge, shared by reproject_kernel (geom.hip) and the lookup kernels that take the
// reprojection in their prologue (corr_sheared.hip).  Replaces pops.projective_transform without Jacobians
// (/root/reference/dbaf/geom/projective_ops.py:96-125: iproj :18-38, Gij = Tj * Ti^-1 with the stereo special case :105,
// actp :67-71, proj :40-49, valid :112).
//
// Both translation units must produce the SAME bits (the fused lookup has to equal reprojection + lookup, bit for bit),
// and they are compiled with different contraction settings (-ffp-contract=off for the lookup's half arithmetic, the
// default elsewhere), so every operation here is spelled with an intrinsic the compiler neither fuses nor reassociates:
// __fmul_rn / __fadd_rn / __fsub_rn / fmaf / v_rcp_f32.
#pragma once
#include "common.h"

namespace dba {

struct EdgeGeom {  // 20 floats: what a pixel of the edge needs
  float R[9];      // rotation of Gij, row-major
  float t[3];      // translation of Gij
  float ifx, ify, cxi, cyi;  // source intrinsics: 1/fx, 1/fy, cx, cy
  float fxj, fyj, cxj, cyj;  // target intrinsics
};
constexpr int EDGE_GEOM_FLOATS = 20;

__device__ __forceinline__ float rp_mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float rp_add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float rp_sub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float rp_rcp(float a) { return __builtin_amdgcn_rcpf(a); }  // v_rcp_f32: 1 ulp, deterministic

// Gij = Tj * Ti^-1 on (t, q_xyzw) rows (relSE3, /root/reference/src/droid_kernels.cu:99-110), stereo edges (ix == jx):
// t = (-0.1, 0, 0), q = identity (projective_ops.py:105); intr_b4: per-frame intrinsics [B, 4]
__device__ __forceinline__ EdgeGeom edge_geom(const float *__restrict__ poses, const float *__restrict__ intr_b4, int ix, int jx) {
  EdgeGeom G;
  float q[4], t[3];
  if (ix == jx) {
    q[0] = q[1] = q[2] = 0.f;
    q[3] = 1.f;
    t[0] = -0.1f;
    t[1] = t[2] = 0.f;
  } else {
    const float *Pi = poses + 7 * ix, *Pj = poses + 7 * jx;
    const float *ti = Pi, *qi = Pi + 3, *tj = Pj, *qj = Pj + 3;
    // qij = qj * conj(qi)
    q[0] = fmaf(qj[2], qi[1], fmaf(-qj[1], qi[2], fmaf(qj[0], qi[3], rp_mul(-qj[3], qi[0]))));
    q[1] = fmaf(qj[0], qi[2], fmaf(-qj[2], qi[0], fmaf(qj[1], qi[3], rp_mul(-qj[3], qi[1]))));
    q[2] = fmaf(qj[1], qi[0], fmaf(-qj[0], qi[1], fmaf(qj[2], qi[3], rp_mul(-qj[3], qi[2]))));
    q[3] = fmaf(qj[2], qi[2], fmaf(qj[1], qi[1], fmaf(qj[0], qi[0], rp_mul(qj[3], qi[3]))));
    // tij = tj - qij (x) ti   (actSO3, droid_kernels.cu:61-71: v + w uv + q x uv, uv = 2 q x v)
    const float uv0 = rp_mul(2.f, fmaf(q[1], ti[2], rp_mul(-q[2], ti[1])));
    const float uv1 = rp_mul(2.f, fmaf(q[2], ti[0], rp_mul(-q[0], ti[2])));
    const float uv2 = rp_mul(2.f, fmaf(q[0], ti[1], rp_mul(-q[1], ti[0])));
    const float r0 = rp_add(fmaf(q[3], uv0, ti[0]), fmaf(q[1], uv2, rp_mul(-q[2], uv1)));
    const float r1 = rp_add(fmaf(q[3], uv1, ti[1]), fmaf(q[2], uv0, rp_mul(-q[0], uv2)));
    const float r2 = rp_add(fmaf(q[3], uv2, ti[2]), fmaf(q[0], uv1, rp_mul(-q[1], uv0)));
    t[0] = rp_sub(tj[0], r0);
    t[1] = rp_sub(tj[1], r1);
    t[2] = rp_sub(tj[2], r2);
  }
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  G.R[0] = fmaf(-2.f, fmaf(y, y, rp_mul(z, z)), 1.f);
  G.R[1] = rp_mul(2.f, fmaf(x, y, rp_mul(-w, z)));
  G.R[2] = rp_mul(2.f, fmaf(x, z, rp_mul(w, y)));
  G.R[3] = rp_mul(2.f, fmaf(x, y, rp_mul(w, z)));
  G.R[4] = fmaf(-2.f, fmaf(x, x, rp_mul(z, z)), 1.f);
  G.R[5] = rp_mul(2.f, fmaf(y, z, rp_mul(-w, x)));
  G.R[6] = rp_mul(2.f, fmaf(x, z, rp_mul(-w, y)));
  G.R[7] = rp_mul(2.f, fmaf(y, z, rp_mul(w, x)));
  G.R[8] = fmaf(-2.f, fmaf(x, x, rp_mul(y, y)), 1.f);
  G.t[0] = t[0];
  G.t[1] = t[1];
  G.t[2] = t[2];
  const float *Ki = intr_b4 + 4 * ix, *Kj = intr_b4 + 4 * jx;
  G.ifx = __fdiv_rn(1.f, Ki[0]);
  G.ify = __fdiv_rn(1.f, Ki[1]);
  G.cxi = Ki[2];
  G.cyi = Ki[3];
  G.fxj = Kj[0];
  G.fyj = Kj[1];
  G.cxj = Kj[2];
  G.cyj = Kj[3];
  return G;
}

// pixel (u, v) of the source frame with inverse depth d -> coordinates in the target frame; `valid` as the reference's
// ((X1.z > MIN_DEPTH) & (X0.z > MIN_DEPTH)), X0.z == 1
__device__ __forceinline__ float2 reproject_pixel(const EdgeGeom &G, float u, float v, float d, float &valid) {
  const float X0 = rp_mul(rp_sub(u, G.cxi), G.ifx), X1 = rp_mul(rp_sub(v, G.cyi), G.ify);
  const float x = fmaf(d, G.t[0], fmaf(G.R[0], X0, fmaf(G.R[1], X1, G.R[2])));
  const float y = fmaf(d, G.t[1], fmaf(G.R[3], X0, fmaf(G.R[4], X1, G.R[5])));
  const float z = fmaf(d, G.t[2], fmaf(G.R[6], X0, fmaf(G.R[7], X1, G.R[8])));
  const float Z = (z < 0.5f * 0.2f) ? 1.0f : z;  // proj(): Z < 0.5 * MIN_DEPTH -> 1 (projective_ops.py:44)
  const float iz = rp_rcp(Z);
  valid = (z > 0.2f) ? 1.0f : 0.0f;
  return make_float2(fmaf(G.fxj, rp_mul(x, iz), G.cxj), fmaf(G.fyj, rp_mul(y, iz), G.cyj));
}

}  // namespace dba
