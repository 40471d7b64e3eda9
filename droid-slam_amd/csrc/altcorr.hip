// On-the-fly ("alt") correlation lookup: the (2r+1)^2 bilinear window of <f1(p)/4, f2(q)/4> without a
// materialised volume.  Replaces altcorr_forward / altcorr_backward
// (reference src/altcorr_kernel.cu:24-225; Python side droid_slam/modules/corr.py:74-117).
//
// v1 (register-tiled VALU, r = 3 fast path): one lane per source pixel, consecutive lanes = consecutive
// pixels, so (i) the per-channel f1 load is one coalesced 128-B run per wave and (ii) with a smooth
// flow field the lanes' window rows overlap and a wave touches ~2 cache lines per row load.  Each
// lane keeps the full 8x8 integer-tap window in 64 fp32 accumulators and per channel issues 1 + 8x2
// sixteen-byte loads (aligned-chunk gather, gather8.h) for 64 FMAs -- the reference issues 2 scalar
// loads per FMA and materialises four [M,8,8,H,W] temporaries for the blend.  The blend happens in
// registers and the (x-offset outer) result is stored once, coalesced.
// Products and sums are fp32 (the reference rounds every product to the feature dtype).
#include "common.h"
#include "gather8.h"

namespace {
using namespace dh;

template <typename T>
__global__ __launch_bounds__(256) void altcorr_fwd_r3_kernel(
    const T* __restrict__ fmap1, const T* __restrict__ fmap2, const float* __restrict__ coords,
    const int64_t* __restrict__ us, const int64_t* __restrict__ vs, T* __restrict__ corr,
    int B, int N1, int N2, int C, int HW, int W, int H2, int W2, int M) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int m = blockIdx.y, b = blockIdx.z;
  if (p >= HW) return;
  const int ix = (int)us[m], jx = (int)vs[m];
  const float* cbase = coords + ((long)(b * M + m) * 2) * HW;
  const float x0 = cbase[p], y0 = cbase[HW + p];
  float fxf = floorf(x0), fyf = floorf(y0);
  const float dx = x0 - fxf, dy = y0 - fyf;
  fxf = fminf(fmaxf(fxf, -65536.f), 65536.f);
  fyf = fminf(fmaxf(fyf, -65536.f), 65536.f);
  const int xs = (int)fxf - 3, ys = (int)fyf - 3;
  constexpr int EPC = ChunkTraits<T>::EPC;
  const long S2 = (long)H2 * W2;
  const long chunk_hi = ((long)B * N2 * C * S2) / EPC - 1;
  const uint4* f2c = reinterpret_cast<const uint4*>(fmap2);
  const T* f1 = fmap1 + ((long)(b * N1 + ix) * C) * HW + p;
  const long f2base = ((long)(b * N2 + jx) * C) * S2;

  float acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;

  for (int c = 0; c < C; ++c) {
    const float a = to_float(f1[(long)c * HW]);
    const long plane = f2base + (long)c * S2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int y1 = ys + i;
      if ((unsigned)y1 < (unsigned)H2) {
        float t[8];
        fetch8(f2c, plane + (long)y1 * W2 + xs, chunk_hi, t, T());
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] += a * t[j];
      }
    }
  }
  // mask out-of-range columns (rows were never accumulated), scale (1/4 * 1/4), blend, store x-outer
  bool colok[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) colok[j] = (unsigned)(xs + j) < (unsigned)W2;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = colok[j] ? acc[i][j] * 0.0625f : 0.f;
  T* out = corr + ((long)(b * M + m) * 49) * HW + p;
#pragma unroll
  for (int yo = 0; yo < 7; ++yo)
#pragma unroll
    for (int xo = 0; xo < 7; ++xo) {
      const float top = acc[yo][xo] + dx * (acc[yo][xo + 1] - acc[yo][xo]);
      const float bot = acc[yo + 1][xo] + dx * (acc[yo + 1][xo + 1] - acc[yo + 1][xo]);
      out[(long)(xo * 7 + yo) * HW] = from_float<T>(top + dy * (bot - top));
    }
}

// generic radius / unaligned planes: one lane per (pixel, output), four dot products
template <typename T>
__global__ __launch_bounds__(256) void altcorr_fwd_generic_kernel(
    const T* __restrict__ fmap1, const T* __restrict__ fmap2, const float* __restrict__ coords,
    const int64_t* __restrict__ us, const int64_t* __restrict__ vs, T* __restrict__ corr,
    int B, int N1, int N2, int C, int HW, int W, int H2, int W2, int M, int r) {
  const int rd = 2 * r + 1;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)B * M * rd * rd * HW;
  if (idx >= total) return;
  const int p = (int)(idx % HW);
  long rest = idx / HW;
  const int yo = (int)(rest % rd); rest /= rd;
  const int xo = (int)(rest % rd); rest /= rd;
  const int m = (int)(rest % M);
  const int b = (int)(rest / M);
  const int ix = (int)us[m], jx = (int)vs[m];
  const float* cbase = coords + ((long)(b * M + m) * 2) * HW;
  const float x0 = cbase[p], y0 = cbase[HW + p];
  float fxf = floorf(x0), fyf = floorf(y0);
  const float dx = x0 - fxf, dy = y0 - fyf;
  fxf = fminf(fmaxf(fxf, -65536.f), 65536.f);
  fyf = fminf(fmaxf(fyf, -65536.f), 65536.f);
  const int x1 = (int)fxf - r + xo, y1 = (int)fyf - r + yo;
  const T* f1 = fmap1 + ((long)(b * N1 + ix) * C) * HW + p;
  const T* f2 = fmap2 + ((long)(b * N2 + jx) * C) * (long)H2 * W2;
  float s[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  for (int c = 0; c < C; ++c) {
    const float a = to_float(f1[(long)c * HW]);
    const T* pl = f2 + (long)c * H2 * W2;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int yy = y1 + u, xx = x1 + v;
        if ((unsigned)yy < (unsigned)H2 && (unsigned)xx < (unsigned)W2) s[u][v] += a * to_float(pl[(long)yy * W2 + xx]);
      }
  }
  const float top = s[0][0] + dx * (s[0][1] - s[0][0]);
  const float bot = s[1][0] + dx * (s[1][1] - s[1][0]);
  corr[idx] = from_float<T>(0.0625f * (top + dy * (bot - top)));
}

// backward (training only): scatter with fp32 atomics, one lane per (pixel, integer tap)
template <typename T>
__global__ __launch_bounds__(256) void altcorr_bwd_kernel(
    const T* __restrict__ fmap1, const T* __restrict__ fmap2, const float* __restrict__ coords,
    const int64_t* __restrict__ us, const int64_t* __restrict__ vs, const float* __restrict__ corr_grad,
    float* __restrict__ g1, float* __restrict__ g2,
    int B, int N1, int N2, int C, int HW, int W, int H2, int W2, int M, int r) {
  const int D = 2 * r + 2, rd = 2 * r + 1;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)B * M * D * D * HW;
  if (idx >= total) return;
  const int p = (int)(idx % HW);
  long rest = idx / HW;
  const int tj = (int)(rest % D); rest /= D;       // x tap
  const int ti = (int)(rest % D); rest /= D;       // y tap
  const int m = (int)(rest % M);
  const int b = (int)(rest / M);
  const int ix = (int)us[m], jx = (int)vs[m];
  const float* cbase = coords + ((long)(b * M + m) * 2) * HW;
  const float x0 = cbase[p], y0 = cbase[HW + p];
  float fxf = floorf(x0), fyf = floorf(y0);
  const float dx = x0 - fxf, dy = y0 - fyf;
  fxf = fminf(fmaxf(fxf, -65536.f), 65536.f);
  fyf = fminf(fmaxf(fyf, -65536.f), 65536.f);
  const int x1 = (int)fxf - r + tj, y1 = (int)fyf - r + ti;
  if ((unsigned)y1 >= (unsigned)H2 || (unsigned)x1 >= (unsigned)W2) return;
  const float* gb = corr_grad + ((long)(b * M + m) * rd * rd) * HW + p;
  auto G = [&](int xo, int yo) -> float {
    return (xo >= 0 && xo < rd && yo >= 0 && yo < rd) ? gb[(long)(xo * rd + yo) * HW] : 0.f;
  };
  const float g = 0.0625f * (G(tj, ti) * (1.f - dx) * (1.f - dy) + G(tj - 1, ti) * dx * (1.f - dy) +
                             G(tj, ti - 1) * (1.f - dx) * dy + G(tj - 1, ti - 1) * dx * dy);
  if (g == 0.f) return;
  const T* f1 = fmap1 + ((long)(b * N1 + ix) * C) * HW + p;
  const T* f2 = fmap2 + ((long)(b * N2 + jx) * C) * (long)H2 * W2 + (long)y1 * W2 + x1;
  float* o1 = g1 + ((long)(b * N1 + ix) * C) * HW + p;
  float* o2 = g2 + ((long)(b * N2 + jx) * C) * (long)H2 * W2 + (long)y1 * W2 + x1;
  for (int c = 0; c < C; ++c) {
    atomicAdd(o1 + (long)c * HW, g * to_float(f2[(long)c * H2 * W2]));
    atomicAdd(o2 + (long)c * H2 * W2, g * to_float(f1[(long)c * HW]));
  }
}

template <typename T>
int launch_fwd(const void* fmap1, const void* fmap2, const float* coords, const int64_t* ii, const int64_t* jj,
               void* corr, int B, int N1, int N2, int C, int H, int W, int H2, int W2, int M, int radius,
               hipStream_t st) {
  const int HW = H * W;
  const bool aligned = (((long)H2 * W2 * sizeof(T)) % 16 == 0) && (((uintptr_t)fmap2) % 16 == 0);
  if (radius == 3 && aligned) {
    hipLaunchKernelGGL(altcorr_fwd_r3_kernel<T>, dim3((HW + 255) / 256, M, B), dim3(256), 0, st,
                       (const T*)fmap1, (const T*)fmap2, coords, ii, jj, (T*)corr, B, N1, N2, C, HW, W, H2, W2, M);
  } else {
    const int rd = 2 * radius + 1;
    const long total = (long)B * M * rd * rd * HW;
    hipLaunchKernelGGL(altcorr_fwd_generic_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const T*)fmap1, (const T*)fmap2, coords, ii, jj, (T*)corr, B, N1, N2, C, HW, W, H2, W2, M,
                       radius);
  }
  DH_LAUNCH_CHECK();
  return DH_OK;
}

int check(int B, int N1, int N2, int C, int H, int W, int H2, int W2, int M, int radius) {
  if (B < 0 || M < 0 || N1 <= 0 || N2 <= 0 || C <= 0 || H <= 0 || W <= 0 || H2 <= 0 || W2 <= 0) return DH_ERR_ARG;
  if (radius < 0 || radius > 16) return DH_ERR_ARG;
  return DH_OK;
}

}  // namespace

extern "C" int dh_altcorr_fwd(const void* fmap1, const void* fmap2, const float* coords,
                              const int64_t* ii, const int64_t* jj, void* corr, int dtype,
                              int B, int N1, int N2, int C, int H, int W, int H2, int W2, int M, int radius,
                              dh_stream_t stream) {
  int rc = check(B, N1, N2, C, H, W, H2, W2, M, radius);
  if (rc != DH_OK) return rc;
  if (B == 0 || M == 0) return DH_OK;
  if (!fmap1 || !fmap2 || !coords || !ii || !jj || !corr) return DH_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DH_F16) return launch_fwd<__half>(fmap1, fmap2, coords, ii, jj, corr, B, N1, N2, C, H, W, H2, W2, M, radius, st);
  if (dtype == DH_F32) return launch_fwd<float>(fmap1, fmap2, coords, ii, jj, corr, B, N1, N2, C, H, W, H2, W2, M, radius, st);
  return DH_ERR_UNSUPPORTED;
}

extern "C" int dh_altcorr_bwd(const void* fmap1, const void* fmap2, const float* coords,
                              const int64_t* ii, const int64_t* jj, const float* corr_grad,
                              float* fmap1_grad, float* fmap2_grad, int dtype,
                              int B, int N1, int N2, int C, int H, int W, int H2, int W2, int M, int radius,
                              dh_stream_t stream) {
  int rc = check(B, N1, N2, C, H, W, H2, W2, M, radius);
  if (rc != DH_OK) return rc;
  if (B == 0 || M == 0) return DH_OK;
  if (!fmap1 || !fmap2 || !coords || !ii || !jj || !corr_grad || !fmap1_grad || !fmap2_grad) return DH_ERR_ARG;
  if (dtype != DH_F16 && dtype != DH_F32) return DH_ERR_UNSUPPORTED;
  const int D = 2 * radius + 2;
  const long total = (long)B * M * D * D * H * W;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DH_F16)
    hipLaunchKernelGGL(altcorr_bwd_kernel<__half>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const __half*)fmap1, (const __half*)fmap2, coords, ii, jj, corr_grad, fmap1_grad,
                       fmap2_grad, B, N1, N2, C, H * W, W, H2, W2, M, radius);
  else
    hipLaunchKernelGGL(altcorr_bwd_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float*)fmap1, (const float*)fmap2, coords, ii, jj, corr_grad, fmap1_grad,
                       fmap2_grad, B, N1, N2, C, H * W, W, H2, W2, M, radius);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
