// Shared device helpers for libdroid_hip (gfx950 / CDNA4, wave64).
// SE(3) helpers follow the semantics of the reference's device functions
// (reference src/droid_kernels.cu:67-184, 886-904); written for this library, f32.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/droid_hip.h"
#include "options.h"

#define DH_WAVE 64
#define DH_MIN_DEPTH 0.25f        // kernels' depth threshold (src/droid_kernels.cu:35)
#define DH_MIN_DEPTH_PY 0.2f      // Python reprojection threshold (geom/projective_ops.py:6)

#define DH_LAUNCH_CHECK()                                                                  \
  do {                                                                                     \
    const hipError_t dh_e_ = hipGetLastError();                                            \
    if (dh_e_ != hipSuccess) {                                                             \
      if (dh::opts().debug) fprintf(stderr, "libdroid_hip: %s at %s:%d\n", hipGetErrorString(dh_e_), __FILE__, __LINE__); \
      return DH_ERR_LAUNCH;                                                                \
    }                                                                                      \
  } while (0)

// > 64 KB of dynamic LDS must be opted into per kernel AND per device (the attribute is per device on HIP): done once
// per (call site, device), so a process that drives several GPUs gets it on each of them.
#define DH_LDS_OPTIN(fn_, bytes_)                                                                                   \
  do {                                                                                                               \
    static unsigned char dh_done_[64];                                                                               \
    int dh_dev_ = 0;                                                                                                 \
    if (hipGetDevice(&dh_dev_) != hipSuccess) return DH_ERR_LAUNCH;                                                  \
    if (dh_dev_ < 0 || dh_dev_ >= 64 || !dh_done_[dh_dev_]) {                                                        \
      const hipError_t dh_e2_ = hipFuncSetAttribute(reinterpret_cast<const void*>(fn_),                              \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (bytes_));           \
      if (dh_e2_ != hipSuccess) {                                                                                    \
        if (dh::opts().debug) fprintf(stderr, "libdroid_hip: hipFuncSetAttribute: %s at %s:%d\n", hipGetErrorString(dh_e2_), __FILE__, __LINE__); \
        return DH_ERR_LAUNCH;                                                                                        \
      }                                                                                                              \
      if (dh_dev_ >= 0 && dh_dev_ < 64) dh_done_[dh_dev_] = 1;                                                       \
    }                                                                                                                \
  } while (0)

namespace dh {

struct Vec3 { float x, y, z; };
struct Quat { float x, y, z, w; };
struct SE3f { Vec3 t; Quat q; };

__device__ __forceinline__ Vec3 cross(const Vec3& a, const Vec3& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// rotate X by unit quaternion q
__device__ __forceinline__ Vec3 rot(const Quat& q, const Vec3& X) {
  Vec3 qv{q.x, q.y, q.z};
  Vec3 uv = cross(qv, X);
  uv.x *= 2.f; uv.y *= 2.f; uv.z *= 2.f;
  Vec3 c = cross(qv, uv);
  return {X.x + q.w * uv.x + c.x, X.y + q.w * uv.y + c.y, X.z + q.w * uv.z + c.z};
}

__device__ __forceinline__ Quat qconj(const Quat& q) { return {-q.x, -q.y, -q.z, q.w}; }

__device__ __forceinline__ Quat qmul(const Quat& a, const Quat& b) {
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
          a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x,
          a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}

__device__ __forceinline__ SE3f load_pose(const float* p) {
  return {{p[0], p[1], p[2]}, {p[3], p[4], p[5], p[6]}};
}
__device__ __forceinline__ void store_pose(float* p, const SE3f& T) {
  p[0] = T.t.x; p[1] = T.t.y; p[2] = T.t.z; p[3] = T.q.x; p[4] = T.q.y; p[5] = T.q.z; p[6] = T.q.w;
}

// Tij = Tj * Ti^-1
__device__ __forceinline__ SE3f rel(const SE3f& Ti, const SE3f& Tj) {
  SE3f r;
  r.q = qmul(Tj.q, qconj(Ti.q));
  Vec3 rt = rot(r.q, Ti.t);
  r.t = {Tj.t.x - rt.x, Tj.t.y - rt.y, Tj.t.z - rt.z};
  return r;
}

__device__ __forceinline__ SE3f inv(const SE3f& T) {
  SE3f r;
  r.q = qconj(T.q);
  Vec3 rt = rot(r.q, T.t);
  r.t = {-rt.x, -rt.y, -rt.z};
  return r;
}

__device__ __forceinline__ SE3f mul(const SE3f& A, const SE3f& B) {
  SE3f r;
  r.q = qmul(A.q, B.q);
  Vec3 rt = rot(A.q, B.t);
  r.t = {A.t.x + rt.x, A.t.y + rt.y, A.t.z + rt.z};
  return r;
}

// stereo pair: fixed right-camera offset (src/droid_kernels.cu:228-238)
__device__ __forceinline__ SE3f stereo_rel() { return {{-0.1f, 0.f, 0.f}, {0.f, 0.f, 0.f, 1.f}}; }

// Y = Adj(T)^T X on 6-vectors (tau,phi)
__device__ __forceinline__ void adjT(const SE3f& T, const float* X, float* Y) {
  Quat qi = qconj(T.q);
  Vec3 a = rot(qi, {X[0], X[1], X[2]});
  Vec3 b = rot(qi, {X[3], X[4], X[5]});
  Vec3 u = cross({X[0], X[1], X[2]}, T.t);
  Vec3 v = rot(qi, u);
  Y[0] = a.x; Y[1] = a.y; Y[2] = a.z;
  Y[3] = b.x + v.x; Y[4] = b.y + v.y; Y[5] = b.z + v.z;
}

__device__ __forceinline__ Quat so3_exp(const Vec3& phi) {
  float th2 = phi.x * phi.x + phi.y * phi.y + phi.z * phi.z;
  float th4 = th2 * th2;
  float th = sqrtf(th2);
  float imag, real;
  if (th2 < 1e-8f) {
    imag = 0.5f - (1.0f / 48.0f) * th2 + (1.0f / 3840.0f) * th4;
    real = 1.0f - (1.0f / 8.0f) * th2 + (1.0f / 384.0f) * th4;
  } else {
    imag = sinf(0.5f * th) / th;
    real = cosf(0.5f * th);
  }
  return {imag * phi.x, imag * phi.y, imag * phi.z, real};
}

__device__ __forceinline__ SE3f se3_exp(const float* xi) {
  Vec3 tau{xi[0], xi[1], xi[2]}, phi{xi[3], xi[4], xi[5]};
  SE3f r;
  r.q = so3_exp(phi);
  float th2 = phi.x * phi.x + phi.y * phi.y + phi.z * phi.z;
  float th = sqrtf(th2);
  r.t = tau;
  if (th > 1e-4f) {
    float a = (1.f - cosf(th)) / th2;
    float b = (th - sinf(th)) / (th * th2);
    Vec3 c1 = cross(phi, tau);
    Vec3 c2 = cross(phi, c1);
    r.t.x += a * c1.x + b * c2.x;
    r.t.y += a * c1.y + b * c2.y;
    r.t.z += a * c1.z + b * c2.z;
  }
  return r;
}

// log: (t, q) -> (tau, phi).  phi = log of the unit quaternion (lietorch SO3::Log: 2 atan(|v| / w) / |v| * v, series near
// the identity), tau = V(phi)^-1 t with V^-1 = I - 1/2 [phi]x + c [phi]x^2, c = (1 - theta cos(theta/2) / (2 sin(theta/2))) / theta^2
// (1/12 near the identity): the inverse of se3_exp above.  Needed by the callers of the path: the frontend's motion model
// (droid_frontend.py:59-63) and the trajectory filler's pose interpolation (trajectory_filler.py:55-65).
__device__ __forceinline__ void se3_log(const SE3f& T, float* xi) {
  const float vx = T.q.x, vy = T.q.y, vz = T.q.z, w = T.q.w;
  const float n2 = vx * vx + vy * vy + vz * vz, n = sqrtf(n2);
  float k;                                              // phi = k * v
  if (n2 < 1e-10f) k = 2.f / w - (2.f / 3.f) * n2 / (w * w * w);
  else if (fabsf(w) < 1e-10f) k = (w >= 0.f ? 3.14159265358979f : -3.14159265358979f) / n;
  else k = 2.f * atanf(n / w) / n;
  const Vec3 phi{k * vx, k * vy, k * vz};
  const float th2 = phi.x * phi.x + phi.y * phi.y + phi.z * phi.z, th = sqrtf(th2);
  float c;
  if (th < 1e-3f) c = 1.f / 12.f + th2 / 720.f;
  else { const float h = 0.5f * th; c = (1.f - h * cosf(h) / sinf(h)) / th2; }
  const Vec3 c1 = cross(phi, T.t), c2 = cross(phi, c1);
  xi[0] = T.t.x - 0.5f * c1.x + c * c2.x; xi[1] = T.t.y - 0.5f * c1.y + c * c2.y; xi[2] = T.t.z - 0.5f * c1.z + c * c2.z;
  xi[3] = phi.x; xi[4] = phi.y; xi[5] = phi.z;
}

// exp(xi) * T
__device__ __forceinline__ SE3f retr(const float* xi, const SE3f& T) { return mul(se3_exp(xi), T); }

// ---- wave64 reductions -------------------------------------------------------------------------
// sum over the 16 lanes of a DPP row, left in every lane of the row (4 VALU adds with DPP operands, no LDS crossbar)
__device__ __forceinline__ float row_sum16(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, true));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true));   // row_ror:8
  return v;
}

// Four wave sums for the price of (almost) one: v_permlane32_swap / v_permlane16_swap (gfx950) pair the values up so
// that each halving step also halves the number of registers.  Returns a register whose DPP rows 0..3 hold, in every
// lane of the row, the wave totals of a, c, b, d (in that order).
__device__ __forceinline__ float wave_sum4(float a, float b, float c, float d) {
  auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  auto y = __builtin_amdgcn_permlane32_swap(__float_as_uint(c), __float_as_uint(d), false, false);
  const float xs = __uint_as_float(x[0]) + __uint_as_float(x[1]);      // lanes 0..31: a, lanes 32..63: b
  const float ys = __uint_as_float(y[0]) + __uint_as_float(y[1]);      //              c                d
  auto z = __builtin_amdgcn_permlane16_swap(__float_as_uint(xs), __float_as_uint(ys), false, false);
  return row_sum16(__uint_as_float(z[0]) + __uint_as_float(z[1]));     // rows: a, c, b, d
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float to_float(__half h) { return __half2float(h); }
__device__ __forceinline__ float to_float(float f) { return f; }
template <typename T> __device__ __forceinline__ T from_float(float f);
template <> __device__ __forceinline__ float from_float<float>(float f) { return f; }
template <> __device__ __forceinline__ __half from_float<__half>(float f) { return __float2half(f); }
// arithmetic type of a lookup on volumes of element type T: the reference computes the bilinear blend in scalar_t
// (src/correlation_kernels.cu:40-67), i.e. in double for double volumes; half / float volumes are blended in fp32 here
template <typename T> struct BlendType { using type = float; };
template <> struct BlendType<double> { using type = double; };
__device__ __forceinline__ double to_blend(double v) { return v; }
__device__ __forceinline__ float to_blend(float v) { return v; }
__device__ __forceinline__ float to_blend(__half v) { return __half2float(v); }
template <typename T, typename A> __device__ __forceinline__ T from_blend(A v) { return (T)v; }
template <> __device__ __forceinline__ __half from_blend<__half, float>(float v) { return __float2half(v); }

}  // namespace dh
