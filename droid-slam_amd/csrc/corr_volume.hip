// All-pairs correlation volume in the REFERENCE layout [E, h1, w1, h2, w2] and its 2x2 average pooling, for any image size.
//
// Replaces CorrBlock.corr + the three F.avg_pool2d of CorrBlock.__init__ (reference droid_slam/modules/corr.py:23-38,
// 63-71) for callers that need the reference's own tensor: droid_amd.corr.CorrBlockRef, i.e. image sizes outside the
// MI355X pyramid layout of corr_pyramid.hip (w not in {16,32,64} or h % 8 != 0: TUM's 30x40, ...) and A/B measurements.
//
//   corr[e, p, q] = sum_c (f1[e, c, p] / 4) * (f2[e, c, q] / 4)        p, q = flattened pixels, fp32 accumulation,
//                                                                      one rounding to the feature dtype
//   pooled[n, y, x] = mean of the 2x2 block of level l (floor(h2/2) x floor(w2/2), like avg_pool2d(2, stride=2))
//
// fp16 features: a true contraction over the channels on the MFMA (v_mfma_f32_32x32x16_f16).  Workgroup = one 64 x 64
// tile of (p, q) of one edge: the two [C, 64] feature slabs are staged in LDS transposed to [pixel][channel] (the features
// are channel-major in HBM: a fragment's 8 consecutive channels of one pixel are HW elements apart), four waves = four
// 32 x 32 quadrants, C / 16 MFMAs each.  fp32 features: the same tiling on the vector ALU (exact fp32 products like
// torch.matmul on fp32 tensors).  MFMA-bound in principle (2 * HW^2 * C flop per edge); as written it re-reads each slab
// HW / 64 times and is a fallback, not a roofline kernel: the product path for the supported sizes is corr_pyramid.hip.
#include "common.h"

namespace {
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int TP = 64;            // tile edge (pixels)
constexpr int CK = 128;           // channels staged per pass
constexpr int LDT = CK + 8;       // LDS row stride (halves): 272 B, 16-byte aligned, conflict-free for the 16-byte fragment reads

__global__ __launch_bounds__(256) void corr_volume_f16_kernel(const __half* __restrict__ f1, const __half* __restrict__ f2,
                                                              __half* __restrict__ out, int C, int HW) {
  __shared__ __half sA[TP * LDT];
  __shared__ __half sB[TP * LDT];
  const int e = blockIdx.z, p0 = blockIdx.y * TP, q0 = blockIdx.x * TP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const __half* a = f1 + (long)e * C * HW;
  const __half* b = f2 + (long)e * C * HW;
  const int mr = (wave >> 1) * 32, nr = (wave & 1) * 32;       // this wave's quadrant
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int c0 = 0; c0 < C; c0 += CK) {
    // stage [CK channels][64 pixels] of both maps as [pixel][channel]; consecutive threads = consecutive pixels (coalesced)
    for (int o = tid; o < CK * TP; o += 256) {
      const int c = o >> 6, px = o & 63;
      const bool cok = c0 + c < C;
      sA[px * LDT + c] = (cok && p0 + px < HW) ? a[(long)(c0 + c) * HW + p0 + px] : __float2half(0.f);
      sB[px * LDT + c] = (cok && q0 + px < HW) ? b[(long)(c0 + c) * HW + q0 + px] : __float2half(0.f);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < CK / 16; ++ks) {
      // 32x32x16: lane holds A[row = lane & 31][k = (lane >> 5) * 8 .. +8], B[k = (lane >> 5) * 8 .. +8][col = lane & 31]
      const half8 af = *reinterpret_cast<const half8*>(sA + (mr + (lane & 31)) * LDT + ks * 16 + (lane >> 5) * 8);
      const half8 bf = *reinterpret_cast<const half8*>(sB + (nr + (lane & 31)) * LDT + ks * 16 + (lane >> 5) * 8);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // D[row][col]: lane holds col = lane & 31, rows 8 * (i / 4) + (lane >> 5) * 4 + (i & 3)
  __half* o = out + (long)e * HW * HW;
  const int q = q0 + nr + (lane & 31);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int p = p0 + mr + 8 * (i >> 2) + (lane >> 5) * 4 + (i & 3);
    if (p < HW && q < HW) o[(long)p * HW + q] = __float2half(acc[i] * 0.0625f);
  }
}

// fp32 features: 64 x 64 tile, thread = 4 x 4 outputs, fp32 FMA (the arithmetic of torch.matmul on fp32 tensors)
__global__ __launch_bounds__(256) void corr_volume_f32_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                              float* __restrict__ out, int C, int HW) {
  __shared__ float sA[16][TP + 1];
  __shared__ float sB[16][TP + 1];
  const int e = blockIdx.z, p0 = blockIdx.y * TP, q0 = blockIdx.x * TP;
  const int tid = threadIdx.x, tp = (tid >> 4) * 4, tq = (tid & 15) * 4;
  const float* a = f1 + (long)e * C * HW;
  const float* b = f2 + (long)e * C * HW;
  float acc[4][4] = {};
  for (int c0 = 0; c0 < C; c0 += 16) {
    for (int o = tid; o < 16 * TP; o += 256) {
      const int c = o >> 6, px = o & 63;
      const bool cok = c0 + c < C;
      sA[c][px] = (cok && p0 + px < HW) ? a[(long)(c0 + c) * HW + p0 + px] * 0.25f : 0.f;
      sB[c][px] = (cok && q0 + px < HW) ? b[(long)(c0 + c) * HW + q0 + px] * 0.25f : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(sA[c][tp + i], sB[c][tq + j], acc[i][j]);
    __syncthreads();
  }
  float* o = out + (long)e * HW * HW;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (p0 + tp + i < HW && q0 + tq + j < HW) o[(long)(p0 + tp + i) * HW + q0 + tq + j] = acc[i][j];
}

template <typename T>
__global__ __launch_bounds__(256) void corr_volume_pool_kernel(const T* __restrict__ in, T* __restrict__ out, long n_slices,
                                                               int h2, int w2) {
  const int ho = h2 >> 1, wo = w2 >> 1;
  const long total = n_slices * ho * wo;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / (ho * wo);
    const int r = (int)(i - n * (ho * wo)), y = r / wo, x = r - y * wo;
    const T* s = in + n * (long)h2 * w2 + (long)(2 * y) * w2 + 2 * x;
    const float v = 0.25f * (dh::to_float(s[0]) + dh::to_float(s[1]) + dh::to_float(s[w2]) + dh::to_float(s[w2 + 1]));
    out[i] = dh::from_float<T>(v);
  }
}

}  // namespace

extern "C" int dh_corr_volume_build(const void* fmap1, const void* fmap2, void* volume, int dtype,
                                    int E, int C, int h, int w, dh_stream_t stream) {
  if (E < 0 || C <= 0 || h <= 0 || w <= 0) return DH_ERR_ARG;
  if (E == 0) return DH_OK;
  if (!fmap1 || !fmap2 || !volume) return DH_ERR_ARG;
  const int HW = h * w, nt = (HW + TP - 1) / TP;
  if (E > 65535) return DH_ERR_ARG;                     // grid.z
  const dim3 grid(nt, nt, E);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DH_F16)
    hipLaunchKernelGGL(corr_volume_f16_kernel, grid, dim3(256), 0, st, (const __half*)fmap1, (const __half*)fmap2, (__half*)volume, C, HW);
  else if (dtype == DH_F32)
    hipLaunchKernelGGL(corr_volume_f32_kernel, grid, dim3(256), 0, st, (const float*)fmap1, (const float*)fmap2, (float*)volume, C, HW);
  else
    return DH_ERR_UNSUPPORTED;
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_corr_volume_pool(const void* in, void* out, int dtype, long n_slices, int h2, int w2, dh_stream_t stream) {
  if (n_slices < 0 || h2 <= 0 || w2 <= 0) return DH_ERR_ARG;
  const long total = n_slices * (h2 >> 1) * (w2 >> 1);
  if (total == 0) return DH_OK;
  if (!in || !out) return DH_ERR_ARG;
  const unsigned grid = (unsigned)std::min<long>((total + 255) / 256, 1L << 20);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DH_F16)
    hipLaunchKernelGGL(corr_volume_pool_kernel<__half>, dim3(grid), dim3(256), 0, st, (const __half*)in, (__half*)out, n_slices, h2, w2);
  else if (dtype == DH_F32)
    hipLaunchKernelGGL(corr_volume_pool_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)in, (float*)out, n_slices, h2, w2);
  else
    return DH_ERR_UNSUPPORTED;
  DH_LAUNCH_CHECK();
  return DH_OK;
}
