// Normalisation / residual glue of the feature and context encoders (reference droid_slam/modules/extractor.py:6-50,
// 120-198: BasicEncoder with InstanceNorm2d for the feature network, no normalisation for the context network), on
// channel-last fp16 activations as the convolution kernel of this library (csrc/conv.hip) reads and writes them.
// The encoders run once per incoming frame (~11 GFLOP each at 384x512), not once per BA iteration; their convolutions go
// through dh_conv2d_nhwc_f16 (stride-2 layers = the stride-1 result at the even positions), these kernels are the rest:
//   instance_norm_stats_kernel  per (image, channel) sum and sum of squares over the pixels (fp32)
//   norm_act_kernel             y = relu?( (x - mean) * rsqrt(var + eps) [+ residual] )  /  y = relu(x + residual)
#include "common.h"

namespace {
using namespace dh;

// x [N, HW, C] fp16, C % 8 == 0.  grid (pixel chunks, N); thread = (8-channel group, pixel lane)
__global__ __launch_bounds__(256) void instance_norm_stats_kernel(const __half* __restrict__ x, float* __restrict__ stats, int HW, int C,
                                                                  int pix_per_block) {
  const int n = blockIdx.y, c8 = C / 8;
  const int g = threadIdx.x % c8, lane = threadIdx.x / c8, nl = 256 / c8;
  const bool active = lane < nl;                            // (C / 8 does not divide 256: the last threads only take part in the barrier)
  const int p0 = blockIdx.x * pix_per_block, p1 = min(HW, p0 + pix_per_block);
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int p = p0 + lane; active && p < p1; p += nl) {
    const uint4 v = *reinterpret_cast<const uint4*>(x + ((long)n * HW + p) * C + g * 8);
    const __half2* h2 = reinterpret_cast<const __half2*>(&v);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float2 f = __half22float2(h2[k]);
      s[2 * k] += f.x; s[2 * k + 1] += f.y; q[2 * k] += f.x * f.x; q[2 * k + 1] += f.y * f.y;
    }
  }
  // Workgroup reduction first, ONE atomic per (workgroup, channel, moment).  (Rounds 3-5 let every thread add its 16 partial sums to
  // the 2 C global accumulators of the image directly: 196 608 float atomics on 64 addresses for a 192 x 256 x 32 layer, 0.46 ms per
  // call, fifteen calls per frame = 6.9 of the feature encoder's 7.8 ms -- profiles/r06_encoder_kernel_stats.md.)
  __shared__ float red[16][257];
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[k][threadIdx.x] = s[k]; red[8 + k][threadIdx.x] = q[k]; }
  __syncthreads();
  for (int t = threadIdx.x; t < c8 * 16; t += 256) {
    const int gg = t % c8, k = (t / c8) & 7, moment = t / (c8 * 8);
    float a = 0.f;
    for (int l = 0; l < nl; ++l) a += red[moment * 8 + k][l * c8 + gg];
    atomicAdd(&stats[((long)n * C + gg * 8 + k) * 2 + moment], a);
  }
}

// mode 0: y = act((x - mean) * rstd), mode 1: y = act(x + res), mode 2: y = act((x - mean) * rstd) then act(res + y)?  (not needed)
__global__ __launch_bounds__(256) void norm_act_kernel(const __half* __restrict__ x, const float* __restrict__ stats, const __half* __restrict__ res,
                                                       __half* __restrict__ y, long n8, int HW, int C, float eps, int relu) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const int c8 = C / 8;
  const int g = (int)(i % c8);
  const long n = i / ((long)HW * c8);
  const uint4 v = reinterpret_cast<const uint4*>(x)[i];
  uint4 rv{0u, 0u, 0u, 0u};
  if (res) rv = reinterpret_cast<const uint4*>(res)[i];
  const __half2* h2 = reinterpret_cast<const __half2*>(&v); const __half2* r2 = reinterpret_cast<const __half2*>(&rv);
  uint4 o; __half2* o2 = reinterpret_cast<__half2*>(&o);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float2 f = __half22float2(h2[k]);
    if (stats) {
      const float inv_n = 1.f / (float)HW;
      const float* st = stats + ((long)n * C + g * 8 + 2 * k) * 2;
      const float m0 = st[0] * inv_n, m1 = st[2] * inv_n;
      const float v0 = fmaxf(st[1] * inv_n - m0 * m0, 0.f), v1 = fmaxf(st[3] * inv_n - m1 * m1, 0.f);
      f.x = (f.x - m0) * rsqrtf(v0 + eps); f.y = (f.y - m1) * rsqrtf(v1 + eps);
    }
    if (res) { const float2 r = __half22float2(r2[k]); f.x += r.x; f.y += r.y; }
    if (relu) { f.x = fmaxf(f.x, 0.f); f.y = fmaxf(f.y, 0.f); }
    o2[k] = __floats2half2_rn(f.x, f.y);
  }
  reinterpret_cast<uint4*>(y)[i] = o;
}

}  // namespace

// y = act( instance_norm(x) ) if normalize else act(x + residual); x, y, residual [N,H,W,C] f16 (C % 8 == 0, C <= 2048);
// stats_ws [N*C*2] f32 scratch (zeroed here).  residual may be NULL.  eps = 1e-5 (nn.InstanceNorm2d default)
extern "C" int dh_norm_act_nhwc_f16(const void* x, const void* residual, void* y, float* stats_ws, int N, int HW, int C,
                                    int normalize, int relu, dh_stream_t stream) {
  if (N < 0 || HW <= 0 || C <= 0 || C % 8 || C > 2048) return DH_ERR_ARG;
  if (N == 0) return DH_OK;
  if (!x || !y || (normalize && !stats_ws)) return DH_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (normalize) {
    if (hipMemsetAsync(stats_ws, 0, sizeof(float) * (size_t)N * C * 2, st) != hipSuccess) return DH_ERR_LAUNCH;
    const int ppb = 1024;
    hipLaunchKernelGGL(instance_norm_stats_kernel, dim3((HW + ppb - 1) / ppb, N), dim3(256), 0, st, (const __half*)x, stats_ws, HW, C, ppb);
    DH_LAUNCH_CHECK();
  }
  const long n8 = (long)N * HW * (C / 8);
  hipLaunchKernelGGL(norm_act_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, (const __half*)x,
                     normalize ? (const float*)stats_ws : (const float*)nullptr, (const __half*)residual, (__half*)y, n8, HW, C, 1e-5f, relu);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
