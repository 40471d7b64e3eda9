// Geometry kernels of the droid_backends API and the SE(3) ops / fused reprojection that the Python
// callers reach through lietorch.  Replaces frame_distance / projmap / iproj / depth_filter
// (reference src/droid_kernels.cu:436-859, 1447-1550) and lietorch's SE3 inv/mul/act/adjT/exp/retr
// as used by droid_slam/geom/projective_ops.py:165-198.
//
// All of these are small HBM-bound maps/reductions: lane = pixel (coalesced reads of disps, coalesced
// writes), the relative pose is computed redundantly per lane from 14 scalars (cheaper than an LDS
// broadcast + barrier), reductions use wave64 shuffles and one LDS hop across the 4 waves.
#include "common.h"

namespace {
using namespace dh;

__device__ __forceinline__ void pixel_ray(int p, int wd, float fx, float fy, float cx, float cy,
                                          float& u, float& v, float& X, float& Y) {
  const int row = p / wd, col = p - row * wd;
  u = (float)col; v = (float)row;
  X = (u - cx) / fx; Y = (v - cy) / fy;
}

// ---- projmap ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void projmap_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intr,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, float* __restrict__ coords,
    float* __restrict__ valid, int HW, int wd) {
  const int m = blockIdx.x;
  const int p = blockIdx.y * 256 + threadIdx.x;
  if (p >= HW) return;
  const int i = (int)ii[m], j = (int)jj[m];
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const SE3f T = rel(load_pose(poses + 7 * (long)i), load_pose(poses + 7 * (long)j));
  float u, v, X, Y;
  pixel_ray(p, wd, fx, fy, cx, cy, u, v, X, Y);
  const float h = disps[(long)i * HW + p];
  Vec3 R = rot(T.q, {X, Y, 1.f});
  const float x = R.x + h * T.t.x, y = R.y + h * T.t.y, z = R.z + h * T.t.z;
  float cu = u, cv = v;
  if (z > 0.01f) { cu = fx * (x / z) + cx; cv = fy * (y / z) + cy; }
  float* c = coords + ((long)m * HW + p) * 3;
  c[0] = cu; c[1] = cv; c[2] = 0.f;
  valid[(long)m * HW + p] = (z > DH_MIN_DEPTH) ? 1.f : 0.f;
}

// ---- frame_distance ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void frame_distance_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intr,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, float* __restrict__ dist,
    int HW, int wd, float beta) {
  const int m = blockIdx.x;
  const int tid = threadIdx.x;
  const int i = (int)ii[m], j = (int)jj[m];
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const SE3f T = rel(load_pose(poses + 7 * (long)i), load_pose(poses + 7 * (long)j));
  float accum = 0.f, val = 0.f, total = 0.f;
  for (int p = tid; p < HW; p += 256) {
    float u, v, X, Y;
    pixel_ray(p, wd, fx, fy, cx, cy, u, v, X, Y);
    const float h = disps[(long)i * HW + p];
    Vec3 R = rot(T.q, {X, Y, 1.f});
    {   // full motion
      const float x = R.x + h * T.t.x, y = R.y + h * T.t.y, z = R.z + h * T.t.z;
      const float du = fx * (x / z) + cx - u, dv = fy * (y / z) + cy - v;
      total += beta;
      if (z > DH_MIN_DEPTH) { accum += beta * sqrtf(du * du + dv * dv); val += beta; }
    }
    {   // translation only
      const float x = X + h * T.t.x, y = Y + h * T.t.y, z = 1.f + h * T.t.z;
      const float du = fx * (x / z) + cx - u, dv = fy * (y / z) + cy - v;
      total += 1.f - beta;
      if (z > DH_MIN_DEPTH) { accum += (1.f - beta) * sqrtf(du * du + dv * dv); val += 1.f - beta; }
    }
  }
  __shared__ float s[3][4];
  accum = wave_sum(accum); val = wave_sum(val); total = wave_sum(total);
  if ((tid & 63) == 0) { s[0][tid >> 6] = accum; s[1][tid >> 6] = val; s[2][tid >> 6] = total; }
  __syncthreads();
  if (tid == 0) {
    const float a = (s[0][0] + s[0][1]) + (s[0][2] + s[0][3]);
    const float vv = (s[1][0] + s[1][1]) + (s[1][2] + s[1][3]);
    const float t = (s[2][0] + s[2][1]) + (s[2][2] + s[2][3]);
    dist[m] = (vv / (t + 1e-8f) < 0.75f) ? 1000.f : a / vv;
  }
}

// ---- iproj ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void iproj_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intr,
    float* __restrict__ points, int HW, int wd) {
  const int n = blockIdx.x;
  const int p = blockIdx.y * 256 + threadIdx.x;
  if (p >= HW) return;
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const SE3f T = load_pose(poses + 7 * (long)n);
  float u, v, X, Y;
  pixel_ray(p, wd, fx, fy, cx, cy, u, v, X, Y);
  const float h = disps[(long)n * HW + p];
  Vec3 R = rot(T.q, {X, Y, 1.f});
  float* o = points + ((long)n * HW + p) * 3;
  o[0] = (R.x + h * T.t.x) / h; o[1] = (R.y + h * T.t.y) / h; o[2] = (R.z + h * T.t.z) / h;
}

// ---- depth_filter -----------------------------------------------------------------------------
// one lane per pixel, loops over the 6 neighbours itself -> plain store, no atomics, deterministic
__global__ __launch_bounds__(256) void depth_filter_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intr,
    const int64_t* __restrict__ inds, const float* __restrict__ thresh, float* __restrict__ counter,
    int num, int ht, int wd) {
  const int HW = ht * wd;
  const int m = blockIdx.x;
  const int p = blockIdx.y * 256 + threadIdx.x;
  if (p >= HW) return;
  const int i = (int)inds[m];
  const float t = thresh[m];
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  float u, v, X, Y;
  pixel_ray(p, wd, fx, fy, cx, cy, u, v, X, Y);
  const float di = disps[(long)i * HW + p];
  const SE3f Ti = load_pose(poses + 7 * (long)i);
  float count = 0.f;
#pragma unroll
  for (int nb = 0; nb < 6; ++nb) {
    const int j = (nb < 3) ? i - nb - 1 : i + nb;
    if (j < 0 || j >= num) continue;
    const SE3f T = rel(Ti, load_pose(poses + 7 * (long)j));
    Vec3 R = rot(T.q, {X, Y, 1.f});
    const float x = R.x + di * T.t.x, y = R.y + di * T.t.y, z = R.z + di * T.t.z;
    const float uj = fx * (x / z) + cx, vj = fy * (y / z) + cy, dj = di / z;
    // the reference converts floor() to int first and compares integers (droid_kernels.cu:760-763): a NaN coordinate
    // converts to 0 on this hardware and passes, +-inf saturates and fails -- same conversion here
    const int u0 = (int)floorf(uj), v0 = (int)floorf(vj);
    if (!(u0 >= 0 && v0 >= 0 && u0 < wd - 1 && v0 < ht - 1)) continue;
    const float* dn = disps + (long)j * HW;
    const float d00 = dn[v0 * wd + u0], d01 = dn[v0 * wd + u0 + 1];
    const float d10 = dn[(v0 + 1) * wd + u0], d11 = dn[(v0 + 1) * wd + u0 + 1];
    // the reference's test is written with double literals, `abs(1.0/dj - 1.0/d00) < t` (droid_kernels.cu:775-778):
    // reciprocals, difference and comparison are fp64; in fp32 the count differs on ~0.2 % of the pixels
    const double idj = 1.0 / (double)dj, td = (double)t;
    if (fabs(idj - 1.0 / (double)d00) < td || fabs(idj - 1.0 / (double)d01) < td || fabs(idj - 1.0 / (double)d10) < td ||
        fabs(idj - 1.0 / (double)d11) < td)
      count += 1.f;
  }
  counter[(long)m * HW + p] = count;
}

// ---- fused reprojection (Python thresholds) -----------------------------------------------------
__global__ __launch_bounds__(256) void reproject_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intr,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, float* __restrict__ coords,
    float* __restrict__ valid, int HW, int wd, int intr_stride) {
  const int e = blockIdx.x;
  const int p = blockIdx.y * 256 + threadIdx.x;
  if (p >= HW) return;
  const int i = (int)ii[e], j = (int)jj[e];
  // back-projection with the source frame's intrinsics, projection with the target frame's (projective_ops.py:180,183:
  // intrinsics[:,ii] / intrinsics[:,jj]); intr_stride = 0: one camera for every frame
  const float* ki = intr + (long)i * intr_stride;
  const float* kj = intr + (long)j * intr_stride;
  const float fx = kj[0], fy = kj[1], cx = kj[2], cy = kj[3];
  const SE3f T = (i == j) ? stereo_rel() : mul(load_pose(poses + 7 * (long)j), inv(load_pose(poses + 7 * (long)i)));
  float u, v, X, Y;
  pixel_ray(p, wd, ki[0], ki[1], ki[2], ki[3], u, v, X, Y);
  const float h = disps[(long)i * HW + p];
  Vec3 R = rot(T.q, {X, Y, 1.f});
  const float x = R.x + h * T.t.x, y = R.y + h * T.t.y, z = R.z + h * T.t.z;
  const float zc = (z < 0.5f * DH_MIN_DEPTH_PY) ? 1.f : z;
  const float d = 1.f / zc;
  float2 c = make_float2(fx * (x * d) + cx, fy * (y * d) + cy);
  reinterpret_cast<float2*>(coords)[(long)e * HW + p] = c;
  if (valid) valid[(long)e * HW + p] = (z > DH_MIN_DEPTH_PY) ? 1.f : 0.f;
}

// ---- SE3 elementwise ops ------------------------------------------------------------------------
enum { OP_INV = 0, OP_MUL = 1, OP_EXP = 2, OP_RETR = 3 };

template <int OP>
__global__ void se3_op_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  SE3f r;
  if (OP == OP_INV) r = inv(load_pose(a + 7 * (long)i));
  else if (OP == OP_MUL) r = mul(load_pose(a + 7 * (long)i), load_pose(b + 7 * (long)i));
  else if (OP == OP_EXP) r = se3_exp(a + 6 * (long)i);
  else r = retr(a + 6 * (long)i, load_pose(b + 7 * (long)i));
  store_pose(out + 7 * (long)i, r);
}

__global__ void se3_act4_kernel(const float* __restrict__ a, const float* __restrict__ X, float* __restrict__ Y,
                                int n, int npts) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * npts) return;
  const SE3f T = load_pose(a + 7 * (idx / npts));
  const float4 x = reinterpret_cast<const float4*>(X)[idx];
  Vec3 R = rot(T.q, {x.x, x.y, x.z});
  reinterpret_cast<float4*>(Y)[idx] = make_float4(R.x + x.w * T.t.x, R.y + x.w * T.t.y, R.z + x.w * T.t.z, x.w);
}

__global__ void se3_adjT_kernel(const float* __restrict__ a, const float* __restrict__ X, float* __restrict__ Y,
                                int n, int npts) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * npts) return;
  const SE3f T = load_pose(a + 7 * (idx / npts));
  float x[6], y[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) x[k] = X[idx * 6 + k];
  adjT(T, x, y);
#pragma unroll
  for (int k = 0; k < 6; ++k) Y[idx * 6 + k] = y[k];
}

}  // namespace

extern "C" int dh_projmap(const float* poses, const float* disps, const float* intrinsics,
                          const int64_t* ii, const int64_t* jj, float* coords, float* valid,
                          int M, int ht, int wd, dh_stream_t stream) {
  if (M < 0 || ht <= 0 || wd <= 0) return DH_ERR_ARG;
  if (M == 0) return DH_OK;
  if (!poses || !disps || !intrinsics || !ii || !jj || !coords || !valid) return DH_ERR_ARG;
  const int HW = ht * wd;
  hipLaunchKernelGGL(projmap_kernel, dim3(M, (HW + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     poses, disps, intrinsics, ii, jj, coords, valid, HW, wd);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_frame_distance(const float* poses, const float* disps, const float* intrinsics,
                                 const int64_t* ii, const int64_t* jj, float* dist,
                                 int M, int ht, int wd, float beta, dh_stream_t stream) {
  if (M < 0 || ht <= 0 || wd <= 0) return DH_ERR_ARG;
  if (M == 0) return DH_OK;
  if (!poses || !disps || !intrinsics || !ii || !jj || !dist) return DH_ERR_ARG;
  hipLaunchKernelGGL(frame_distance_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream,
                     poses, disps, intrinsics, ii, jj, dist, ht * wd, wd, beta);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_iproj(const float* poses, const float* disps, const float* intrinsics, float* points,
                        int N, int ht, int wd, dh_stream_t stream) {
  if (N < 0 || ht <= 0 || wd <= 0) return DH_ERR_ARG;
  if (N == 0) return DH_OK;
  if (!poses || !disps || !intrinsics || !points) return DH_ERR_ARG;
  const int HW = ht * wd;
  hipLaunchKernelGGL(iproj_kernel, dim3(N, (HW + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     poses, disps, intrinsics, points, HW, wd);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_depth_filter(const float* poses, const float* disps, const float* intrinsics,
                               const int64_t* ix, const float* thresh, float* counter,
                               int M, int num_frames, int ht, int wd, dh_stream_t stream) {
  if (M < 0 || num_frames <= 0 || ht <= 0 || wd <= 0) return DH_ERR_ARG;
  if (M == 0) return DH_OK;
  if (!poses || !disps || !intrinsics || !ix || !thresh || !counter) return DH_ERR_ARG;
  const int HW = ht * wd;
  hipLaunchKernelGGL(depth_filter_kernel, dim3(M, (HW + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     poses, disps, intrinsics, ix, thresh, counter, num_frames, ht, wd);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_reproject(const float* poses, const float* disps, const float* intrinsics,
                            const int64_t* ii, const int64_t* jj, float* coords, float* valid,
                            int E, int ht, int wd, dh_stream_t stream) {
  return dh_reproject_ex(poses, disps, intrinsics, 0, ii, jj, coords, valid, E, ht, wd, stream);
}

extern "C" int dh_reproject_ex(const float* poses, const float* disps, const float* intrinsics, int per_frame_intrinsics,
                               const int64_t* ii, const int64_t* jj, float* coords, float* valid,
                               int E, int ht, int wd, dh_stream_t stream) {
  if (E < 0 || ht <= 0 || wd <= 0) return DH_ERR_ARG;
  if (E == 0) return DH_OK;
  if (!poses || !disps || !intrinsics || !ii || !jj || !coords) return DH_ERR_ARG;
  const int HW = ht * wd;
  hipLaunchKernelGGL(reproject_kernel, dim3(E, (HW + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     poses, disps, intrinsics, ii, jj, coords, valid, HW, wd, per_frame_intrinsics ? 4 : 0);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

#define DH_SE3_LAUNCH(OP, A, B)                                                                   \
  if (n < 0) return DH_ERR_ARG;                                                                   \
  if (n == 0) return DH_OK;                                                                       \
  if (!(A) || !out) return DH_ERR_ARG;                                                            \
  hipLaunchKernelGGL(se3_op_kernel<OP>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, \
                     A, B, out, n);                                                               \
  DH_LAUNCH_CHECK();                                                                              \
  return DH_OK;

extern "C" int dh_se3_inv(const float* a, float* out, int n, dh_stream_t stream) {
  DH_SE3_LAUNCH(OP_INV, a, (const float*)nullptr)
}
extern "C" int dh_se3_mul(const float* a, const float* b, float* out, int n, dh_stream_t stream) {
  if (n > 0 && !b) return DH_ERR_ARG;
  DH_SE3_LAUNCH(OP_MUL, a, b)
}
extern "C" int dh_se3_exp(const float* xi, float* out, int n, dh_stream_t stream) {
  DH_SE3_LAUNCH(OP_EXP, xi, (const float*)nullptr)
}
extern "C" int dh_se3_retr(const float* xi, const float* a, float* out, int n, dh_stream_t stream) {
  if (n > 0 && !a) return DH_ERR_ARG;
  DH_SE3_LAUNCH(OP_RETR, xi, a)
}

__global__ __launch_bounds__(256) void se3_log_kernel(const float* __restrict__ a, float* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float xi[6];
  se3_log(load_pose(a + 7 * (long)i), xi);
#pragma unroll
  for (int k = 0; k < 6; ++k) out[6 * (long)i + k] = xi[k];
}

extern "C" int dh_se3_log(const float* a, float* out, int n, dh_stream_t stream) {
  if (n < 0) return DH_ERR_ARG;
  if (n == 0) return DH_OK;
  if (!a || !out) return DH_ERR_ARG;
  hipLaunchKernelGGL(se3_log_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, out, n);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_se3_act4(const float* a, const float* X, float* Y, int n, int npts, dh_stream_t stream) {
  if (n < 0 || npts < 0) return DH_ERR_ARG;
  const long tot = (long)n * npts;
  if (tot == 0) return DH_OK;
  if (!a || !X || !Y) return DH_ERR_ARG;
  hipLaunchKernelGGL(se3_act4_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     a, X, Y, n, npts);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_se3_adjT(const float* a, const float* X, float* Y, int n, int npts, dh_stream_t stream) {
  if (n < 0 || npts < 0) return DH_ERR_ARG;
  const long tot = (long)n * npts;
  if (tot == 0) return DH_OK;
  if (!a || !X || !Y) return DH_ERR_ARG;
  hipLaunchKernelGGL(se3_adjT_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     a, X, Y, n, npts);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" const char* dh_version(void) { return "droid_hip 0.1 (gfx950)"; }

extern "C" const char* dh_status_string(int status) {
  switch (status) {
    case DH_OK: return "ok";
    case DH_ERR_ARG: return "invalid argument";
    case DH_ERR_WORKSPACE: return "workspace too small";
    case DH_ERR_LAUNCH: return "HIP launch/runtime error";
    case DH_ERR_UNSUPPORTED: return "unsupported dtype or configuration";
    default: return "unknown status";
  }
}
