// Bilinear window lookup into a materialised 4-D correlation volume (reference layout).
//
// Replaces corr_index_forward / corr_index_backward of the reference extension
// (reference src/correlation_kernels.cu:20-186).  HBM-bound gather: every source pixel owns a private
// h2 x w2 slice of the volume, reads a (2r+2)^2 window of it and writes (2r+1)^2 blended values.
//
// MI355X design (r = 3 fast path)
//   * one lane per source pixel, consecutive lanes = consecutive pixels, so the 49 output planes are
//     written fully coalesced (a wave stores 128 B / 256 B runs) and exactly once -- no zero-init and
//     no read-modify-write of the output (the reference accumulates into a zeroed tensor);
//   * each window row is fetched with 16-byte aligned vector loads (2 per row for f16, 3 for f32)
//     instead of 8 scalar loads: the gather touches the same cache lines with 4x fewer
//     vector-memory instructions, which is what bounds a divergent gather on the CU's address path;
//     the 8 taps are then extracted with a two-stage register select + v_alignbit (no scratch/LDS);
//   * separable interpolation in fp32 registers: 7 lerps along x per row, 7 along y per row pair,
//     one rounding to the volume dtype at the store.
// Generic path (any radius, unaligned slices): four-tap gather per output.
#include "common.h"
#include "gather8.h"
#include <type_traits>

namespace {

using dh::to_float;
using dh::ChunkTraits;
using dh::fetch8;

// ---- fast path: radius 3, 16-byte aligned slices --------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void corr_index_fwd_r3_kernel(
    const T* __restrict__ volume, const float* __restrict__ coords, T* __restrict__ corr,
    int N, int HW1, int h2, int w2) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;          // source pixel (n, y, x) flattened
  if (p >= (long)N * HW1) return;
  const int n = (int)(p / HW1);
  const int yx = (int)(p - (long)n * HW1);
  const float x0 = coords[((long)n * 2 + 0) * HW1 + yx];
  const float y0 = coords[((long)n * 2 + 1) * HW1 + yx];
  float fxf = floorf(x0), fyf = floorf(y0);
  const float dx = x0 - fxf, dy = y0 - fyf;
  fxf = fminf(fmaxf(fxf, -65536.f), 65536.f);
  fyf = fminf(fmaxf(fyf, -65536.f), 65536.f);
  const int xs = (int)fxf - 3;                                   // x of tap 0
  const int ys = (int)fyf - 3;
  constexpr int EPC = ChunkTraits<T>::EPC;
  const long S = (long)h2 * w2;
  const long chunk_hi = ((long)N * HW1 * S) / EPC - 1;
  const uint4* vol16 = reinterpret_cast<const uint4*>(volume);
  const long base = p * S;
  T* out = corr + (long)n * 49 * HW1 + yx;

  float prev[7];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int y1 = ys + j;
    const bool rowok = (unsigned)y1 < (unsigned)h2;
    float t[8];
    if (rowok) {
      fetch8(vol16, base + (long)y1 * w2 + xs, chunk_hi, t, T());
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = ((unsigned)(xs + k) < (unsigned)w2) ? t[k] : 0.f;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = 0.f;
    }
    float cur[7];
#pragma unroll
    for (int a = 0; a < 7; ++a) cur[a] = t[a] + dx * (t[a + 1] - t[a]);
    if (j > 0) {
#pragma unroll
      for (int a = 0; a < 7; ++a) {
        float v = prev[a] + dy * (cur[a] - prev[a]);
        out[(long)(a * 7 + (j - 1)) * HW1] = dh::from_float<T>(v);
      }
    }
#pragma unroll
    for (int a = 0; a < 7; ++a) prev[a] = cur[a];
  }
}

// ---- generic path ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void corr_index_fwd_generic_kernel(
    const T* __restrict__ volume, const float* __restrict__ coords, T* __restrict__ corr,
    int N, int HW1, int h2, int w2, int r) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long)N * HW1) return;
  const int n = (int)(p / HW1);
  const int yx = (int)(p - (long)n * HW1);
  const float x0 = coords[((long)n * 2 + 0) * HW1 + yx];
  const float y0 = coords[((long)n * 2 + 1) * HW1 + yx];
  using A = typename dh::BlendType<T>::type;
  float fxf = floorf(x0), fyf = floorf(y0);
  const A dx = (A)(x0 - fxf), dy = (A)(y0 - fyf);
  fxf = fminf(fmaxf(fxf, -65536.f), 65536.f);
  fyf = fminf(fmaxf(fyf, -65536.f), 65536.f);
  const int rd = 2 * r + 1;
  const T* slice = volume + p * (long)h2 * w2;
  T* out = corr + (long)n * rd * rd * HW1 + yx;
  auto tap = [&](int x1, int y1) -> A {
    return ((unsigned)x1 < (unsigned)w2 && (unsigned)y1 < (unsigned)h2) ? dh::to_blend(slice[(long)y1 * w2 + x1]) : (A)0;
  };
  for (int a = 0; a < rd; ++a) {
    const int x1 = (int)fxf - r + a;
    for (int b = 0; b < rd; ++b) {
      const int y1 = (int)fyf - r + b;
      A t00 = tap(x1, y1), t10 = tap(x1 + 1, y1), t01 = tap(x1, y1 + 1), t11 = tap(x1 + 1, y1 + 1);
      A top = t00 + dx * (t10 - t00), bot = t01 + dx * (t11 - t01);
      out[(long)(a * rd + b) * HW1] = dh::from_blend<T, A>(top + dy * (bot - top));
    }
  }
}

// ---- backward: each tap of a pixel's window receives exactly one value ------------------------
template <typename T>
__global__ __launch_bounds__(256) void corr_index_bwd_kernel(
    const float* __restrict__ coords, const T* __restrict__ corr_grad, T* __restrict__ volume_grad,
    int N, int HW1, int h2, int w2, int r) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long)N * HW1) return;
  const int n = (int)(p / HW1);
  const int yx = (int)(p - (long)n * HW1);
  const float x0 = coords[((long)n * 2 + 0) * HW1 + yx];
  const float y0 = coords[((long)n * 2 + 1) * HW1 + yx];
  using A = typename dh::BlendType<T>::type;
  float fxf = floorf(x0), fyf = floorf(y0);
  const A dx = (A)(x0 - fxf), dy = (A)(y0 - fyf);
  fxf = fminf(fmaxf(fxf, -65536.f), 65536.f);
  fyf = fminf(fmaxf(fyf, -65536.f), 65536.f);
  const int rd = 2 * r + 1;
  T* slice = volume_grad + p * (long)h2 * w2;
  const T* g = corr_grad + (long)n * rd * rd * HW1 + yx;
  auto G = [&](int a, int b) -> A {
    return (a >= 0 && a < rd && b >= 0 && b < rd) ? dh::to_blend(g[(long)(a * rd + b) * HW1]) : (A)0;
  };
  for (int i = 0; i <= rd; ++i) {
    const int x1 = (int)fxf - r + i;
    if ((unsigned)x1 >= (unsigned)w2) continue;
    for (int j = 0; j <= rd; ++j) {
      const int y1 = (int)fyf - r + j;
      if ((unsigned)y1 >= (unsigned)h2) continue;
      A v = G(i - 1, j - 1) * (dx * dy) + G(i - 1, j) * (dx * ((A)1 - dy)) +
            G(i, j - 1) * (((A)1 - dx) * dy) + G(i, j) * (((A)1 - dx) * ((A)1 - dy));
      slice[(long)y1 * w2 + x1] = dh::from_blend<T, A>(v);
    }
  }
}

template <typename T>
int launch_fwd(const void* volume, const float* coords, void* corr, int N, int h1, int w1, int h2, int w2,
               int radius, hipStream_t st) {
  const long npix = (long)N * h1 * w1;
  if (npix == 0) return DH_OK;
  const unsigned grid = (unsigned)((npix + 255) / 256);
  const long S = (long)h2 * w2;
  const bool aligned = ((S * sizeof(T)) % 16 == 0) && (((uintptr_t)volume) % 16 == 0) && S > 0;
  if constexpr (!std::is_same<T, double>::value) {
    if (radius == 3 && aligned) {
      hipLaunchKernelGGL(corr_index_fwd_r3_kernel<T>, dim3(grid), dim3(256), 0, st,
                         (const T*)volume, coords, (T*)corr, N, h1 * w1, h2, w2);
      DH_LAUNCH_CHECK();
      return DH_OK;
    }
  }
  {
    hipLaunchKernelGGL(corr_index_fwd_generic_kernel<T>, dim3(grid), dim3(256), 0, st,
                       (const T*)volume, coords, (T*)corr, N, h1 * w1, h2, w2, radius);
  }
  DH_LAUNCH_CHECK();
  return DH_OK;
}

}  // namespace

extern "C" int dh_corr_index_fwd(const void* volume, const float* coords, void* corr, int dtype,
                                 int N, int h1, int w1, int h2, int w2, int radius, dh_stream_t stream) {
  if (N < 0 || h1 <= 0 || w1 <= 0 || h2 <= 0 || w2 <= 0 || radius < 0 || radius > 16) return DH_ERR_ARG;
  if (N > 0 && (!volume || !coords || !corr)) return DH_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DH_F16) return launch_fwd<__half>(volume, coords, corr, N, h1, w1, h2, w2, radius, st);
  if (dtype == DH_F32) return launch_fwd<float>(volume, coords, corr, N, h1, w1, h2, w2, radius, st);
  if (dtype == DH_F64) return launch_fwd<double>(volume, coords, corr, N, h1, w1, h2, w2, radius, st);   // correlation_kernels.cu:146 dispatches double too
  return DH_ERR_UNSUPPORTED;
}

extern "C" int dh_corr_index_bwd(const float* coords, const void* corr_grad, void* volume_grad, int dtype,
                                 int N, int h1, int w1, int h2, int w2, int radius, dh_stream_t stream) {
  if (N < 0 || h1 <= 0 || w1 <= 0 || h2 <= 0 || w2 <= 0 || radius < 0 || radius > 16) return DH_ERR_ARG;
  if (N > 0 && (!coords || !corr_grad || !volume_grad)) return DH_ERR_ARG;
  if (dtype != DH_F16 && dtype != DH_F32 && dtype != DH_F64) return DH_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const long npix = (long)N * h1 * w1;
  if (npix == 0) return DH_OK;
  const size_t esz = dtype == DH_F16 ? 2 : dtype == DH_F32 ? 4 : 8;
  if (hipMemsetAsync(volume_grad, 0, (size_t)npix * h2 * w2 * esz, st) != hipSuccess) return DH_ERR_LAUNCH;
  const unsigned grid = (unsigned)((npix + 255) / 256);
  if (dtype == DH_F16)
    hipLaunchKernelGGL(corr_index_bwd_kernel<__half>, dim3(grid), dim3(256), 0, st, coords,
                       (const __half*)corr_grad, (__half*)volume_grad, N, h1 * w1, h2, w2, radius);
  else if (dtype == DH_F32)
    hipLaunchKernelGGL(corr_index_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, coords,
                       (const float*)corr_grad, (float*)volume_grad, N, h1 * w1, h2, w2, radius);
  else
    hipLaunchKernelGGL(corr_index_bwd_kernel<double>, dim3(grid), dim3(256), 0, st, coords,
                       (const double*)corr_grad, (double*)volume_grad, N, h1 * w1, h2, w2, radius);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
