// Factor-graph side kernels of the update iteration: everything that sits between the big operators in the reference's
// FactorGraph.update / add_proximity_factors (droid_slam/factor_graph.py:214-263, 346-412) and DepthVideo.upsample
// (droid_slam/depth_video.py:155-159 -> droid_net.py:21-35 cvx_upsample) as elementwise torch code or as Python loops
// over a distance matrix that was first copied to the host.
//
//   motion_features_kernel   motn = cat(coords1 - coords0, target - coords1).clamp(-64, 64)  -> fp16 NHWC (8 ch, 4 real)
//   ba_inputs_kernel         target = coords1 + delta, weight; both also in the [E,2,h,w] layout that ba consumes
//   cvx_upsample_kernel      softmax-weighted 3x3 convex combination to 8x resolution (depth maps for the viewer / dump)
//   prox_mask_kernel + prox_nms_kernel   candidate masking and the GREEDY non-maximum suppression of
//                            add_proximity_factors on the device: the reference copies the t x t distance matrix to
//                            the CPU and walks it in Python, which is what a 5 ms BA ends up waiting for
#include "common.h"

namespace {
using namespace dh;

// one thread per pixel: coords1, target [E,h,w,2] f32 -> flow [E,h,w,8] f16 (channels 4..7 zero)
__global__ __launch_bounds__(256) void motion_features_kernel(const float* __restrict__ coords1, const float* __restrict__ target,
                                                              __half* __restrict__ flow, long npix, int HW, int W) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix) return;
  const int r = (int)(i % HW), y = r / W, x = r - y * W;
  const float2 c = reinterpret_cast<const float2*>(coords1)[i], t = reinterpret_cast<const float2*>(target)[i];
  const float lo = -64.f, hi = 64.f;
  const float m0 = fminf(fmaxf(c.x - (float)x, lo), hi), m1 = fminf(fmaxf(c.y - (float)y, lo), hi);
  const float m2 = fminf(fmaxf(t.x - c.x, lo), hi), m3 = fminf(fmaxf(t.y - c.y, lo), hi);
  const __half2 a = __floats2half2_rn(m0, m1), b = __floats2half2_rn(m2, m3);
  uint4 o{__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b), 0u, 0u};
  reinterpret_cast<uint4*>(flow)[i] = o;
}

// dw [E,h,w,4] f32 = (delta_x, delta_y, w_x, w_y) from the heads; target = coords1 + delta
//   -> target, weight [E,h,w,2] (what FactorGraph keeps) and target_ba, weight_ba [E,2,h,w] (what ba reads)
__global__ __launch_bounds__(256) void ba_inputs_kernel(const float* __restrict__ coords1, const float* __restrict__ dw,
                                                        float* __restrict__ target, float* __restrict__ weight,
                                                        float* __restrict__ target_ba, float* __restrict__ weight_ba, long npix, int HW) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix) return;
  const float2 c = reinterpret_cast<const float2*>(coords1)[i];
  const float4 d = reinterpret_cast<const float4*>(dw)[i];
  const float tx = c.x + d.x, ty = c.y + d.y;
  if (target) reinterpret_cast<float2*>(target)[i] = float2{tx, ty};
  if (weight) reinterpret_cast<float2*>(weight)[i] = float2{d.z, d.w};
  const long e = i / HW, r = i - e * HW;
  target_ba[(e * 2 + 0) * HW + r] = tx; target_ba[(e * 2 + 1) * HW + r] = ty;
  weight_ba[(e * 2 + 0) * HW + r] = d.z; weight_ba[(e * 2 + 1) * HW + r] = d.w;
}

// one wave per source pixel, lane = sub-pixel (sy*8 + sx).  mask [K,h,w,576] f16 channel-last, channel = k*64 + lane with
// k = dy*3 + dx (droid_net.py:25: view(batch,1,9,8,8,ht,wd)); the softmax weights are rounded to fp16 like the reference's
// softmax of the half-precision mask, the combination is fp32.
__global__ __launch_bounds__(256) void cvx_upsample_kernel(const float* __restrict__ disp, const __half* __restrict__ mask,
                                                           float* __restrict__ out, long npix, int H, int W) {
  const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= npix) return;
  const int lane = threadIdx.x & 63;
  const int HW = H * W;
  const long k = pix / HW; const int r = (int)(pix - k * HW), y = r / W, x = r - y * W;
  float m[9], mx = -1e30f;
#pragma unroll
  for (int q = 0; q < 9; ++q) { m[q] = __half2float(mask[pix * 576 + q * 64 + lane]); mx = fmaxf(mx, m[q]); }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 9; ++q) { m[q] = __expf(m[q] - mx); s += m[q]; }
  const float inv = 1.f / s;
  float acc = 0.f;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const int yy = y + q / 3 - 1, xx = x + q % 3 - 1;
    const float d = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? disp[k * HW + (long)yy * W + xx] : 0.f;
    acc += __half2float(__float2half(m[q] * inv)) * d;
  }
  out[(k * 8 * H + (long)y * 8 + (lane >> 3)) * (8 * W) + x * 8 + (lane & 7)] = acc;
}

// ---- add_proximity_factors on the device (factor_graph.py:346-412) ---------------------------------------------------
// d [n_i x n_j] over i in [t0, t), j in [t1, t).  Step 1 (parallel): d[i - rad < j] = inf, d[d > 100] = inf, suppress the
// neighbourhood of every existing edge (active, bad, inactive) and of the always-added temporal edges.
__device__ __forceinline__ int nms_reach(int i, int j, int nms) { const int a = abs(i - j) - 2; const int m = a < nms ? a : nms; return m > 0 ? m : 0; }

__global__ __launch_bounds__(256) void prox_mask_kernel(float* __restrict__ d, int t0, int t1, int t, int rad, int stereo) {
  const int nj = t - t1;
  const long n = (long)(t - t0) * nj;
  const long o = (long)blockIdx.x * 256 + threadIdx.x;
  if (o >= n) return;
  const int i = t0 + (int)(o / nj), j = t1 + (int)(o % nj);
  float v = d[o];
  if (i - rad < j || v > 100.f) v = INFINITY;                                   // :363-364
  // temporal neighbours j in [max(i - rad - 1, 0), i) are always added as edges (:381-384) and, for stereo rigs, the self
  // edge (:377-379): both leave the candidate set
  if ((j < i && j >= (i - rad - 1 > 0 ? i - rad - 1 : 0)) || (stereo && i == j)) v = INFINITY;
  d[o] = v;
}

// one thread per existing edge x neighbourhood cell
__global__ __launch_bounds__(256) void prox_suppress_kernel(float* __restrict__ d, const int64_t* __restrict__ ei, const int64_t* __restrict__ ej,
                                                            int n_edges, int t0, int t1, int t, int nms) {
  const int side = 2 * nms + 1;
  const long o = (long)blockIdx.x * 256 + threadIdx.x;
  if (o >= (long)n_edges * side * side) return;
  const int e = (int)(o / (side * side)), c = (int)(o % (side * side));
  const int di = c / side - nms, dj = c % side - nms;
  const int i = (int)ei[e], j = (int)ej[e];
  if (abs(di) + abs(dj) > nms_reach(i, j, nms)) return;
  const int i1 = i + di, j1 = j + dj;
  if (i1 >= t0 && i1 < t && j1 >= t1 && j1 < t) d[(long)(i1 - t0) * (t - t1) + (j1 - t1)] = INFINITY;
}

// Step 2 (sequential by nature): walk the candidates in ascending distance; accept (i,j) if not suppressed, emit (i,j) and
// (j,i), suppress its neighbourhood.  ONE workgroup: the walk is a dependent chain, the suppression of an accepted edge is
// done by the threads in parallel.  `order` = argsort(d); out_edges [2 * max_new][2] i64; out_count[0] = accepted pairs.
// The reference stops when len(es) > max_factors (checked before each accepted pair; es already holds n_es0 edges).
__global__ __launch_bounds__(256) void prox_nms_kernel(float* __restrict__ d, const float* __restrict__ sorted, const int64_t* __restrict__ order, long n, int t0, int t1, int t,
                                                       int nms, float thresh, int max_factors, int n_es0,
                                                       int64_t* __restrict__ out_edges, int max_new, int* __restrict__ out_count) {
  __shared__ int s_i, s_j, s_take, s_stop;
  const int tid = threadIdx.x;
  const int nj = t - t1, side = 2 * nms + 1;
  int count = 0;
  for (long q = 0; q < n; ++q) {
    if (tid == 0) {
      const long k = order[q];
      const float v = d[k];                           // plain loads: this workgroup is the only writer, barriers order them
      s_stop = 0; s_take = 0;
      if (!(sorted[q] <= thresh)) s_stop = 1;         // value at sort time: ascending, nothing further can pass (inf / nan too)
      else if (!(v <= thresh)) { }                    // suppressed since the sort: skipped, the walk goes on (:389-390 `continue`)
      else if (max_factors > 0 && n_es0 + 2 * count > max_factors) s_stop = 1;
      else if (count >= max_new) s_stop = 1;
      else { s_take = 1; s_i = t0 + (int)(k / nj); s_j = t1 + (int)(k % nj); }
    }
    __syncthreads();
    if (s_stop) break;
    if (s_take) {
      const int i = s_i, j = s_j;
      if (tid == 0) {
        out_edges[(2 * count) * 2 + 0] = i; out_edges[(2 * count) * 2 + 1] = j;
        out_edges[(2 * count + 1) * 2 + 0] = j; out_edges[(2 * count + 1) * 2 + 1] = i;
      }
      if (tid < side * side) {
        const int di = tid / side - nms, dj = tid % side - nms;
        if (abs(di) + abs(dj) <= nms_reach(i, j, nms)) {
          const int i1 = i + di, j1 = j + dj;
          if (i1 >= t0 && i1 < t && j1 >= t1 && j1 < t) d[(long)(i1 - t0) * nj + (j1 - t1)] = INFINITY;
        }
      }
      ++count;
    }
    __threadfence_block();
    __syncthreads();
  }
  if (tid == 0) out_count[0] = count;
}

// an h x w image kept on an Hc x Wc canvas (channel-last fp16): zero every pixel outside the image.  The update operator's
// production convolutions are specialised to 64-pixel rows and four-row tiles; other image sizes run on a zero-padded canvas
// and every activation that feeds a 3x3 layer has its padding re-zeroed, so that the layer sees the zero border the
// reference's padded convolution sees (droid_net.py:83-108: nn.Conv2d(..., padding=1)).
__global__ __launch_bounds__(256) void canvas_mask_kernel(__half* __restrict__ x, long npieces, int Hc, int Wc, int cpieces, int h, int w) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= npieces) return;
  const long pix = i / cpieces;
  const int col = (int)(pix % Wc), row = (int)((pix / Wc) % Hc);
  if (row >= h || col >= w) reinterpret_cast<uint4*>(x)[i] = uint4{0u, 0u, 0u, 0u};
}

}  // namespace

extern "C" int dh_canvas_mask_f16(void* x, int N, int Hc, int Wc, int C, int h, int w, dh_stream_t stream) {
  if (N < 0 || Hc <= 0 || Wc <= 0 || C <= 0 || C % 8 || h <= 0 || w <= 0 || h > Hc || w > Wc) return DH_ERR_ARG;
  if (N == 0 || (h == Hc && w == Wc)) return DH_OK;
  if (!x || ((uintptr_t)x & 15)) return DH_ERR_ARG;
  const long n = (long)N * Hc * Wc * (C / 8);
  hipLaunchKernelGGL(canvas_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (__half*)x, n, Hc, Wc, C / 8, h, w);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_motion_features(const float* coords1, const float* target, void* flow, int E, int ht, int wd, dh_stream_t stream) {
  if (E < 0 || ht <= 0 || wd <= 0) return DH_ERR_ARG;
  if (E == 0) return DH_OK;
  if (!coords1 || !target || !flow) return DH_ERR_ARG;
  const long n = (long)E * ht * wd;
  hipLaunchKernelGGL(motion_features_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, coords1, target,
                     (__half*)flow, n, ht * wd, wd);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_ba_inputs(const float* coords1, const float* dw, float* target, float* weight, float* target_ba, float* weight_ba,
                            int E, int ht, int wd, dh_stream_t stream) {
  if (E < 0 || ht <= 0 || wd <= 0) return DH_ERR_ARG;
  if (E == 0) return DH_OK;
  if (!coords1 || !dw || !target_ba || !weight_ba) return DH_ERR_ARG;
  const long n = (long)E * ht * wd;
  hipLaunchKernelGGL(ba_inputs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, coords1, dw, target, weight,
                     target_ba, weight_ba, n, ht * wd);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_cvx_upsample(const float* disp, const void* mask, float* out, int K, int ht, int wd, dh_stream_t stream) {
  if (K < 0 || ht <= 0 || wd <= 0) return DH_ERR_ARG;
  if (K == 0) return DH_OK;
  if (!disp || !mask || !out) return DH_ERR_ARG;
  const long n = (long)K * ht * wd;
  hipLaunchKernelGGL(cvx_upsample_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, disp, (const __half*)mask, out, n, ht, wd);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_proximity_nms(float* dist, const float* sorted, const int64_t* order, const int64_t* edges_i, const int64_t* edges_j, int n_edges,
                                int t0, int t1, int t, int rad, int nms, float thresh, int max_factors, int n_es0, int stereo,
                                int64_t* out_edges, int max_new, int* out_count, int stage, dh_stream_t stream) {
  if (t0 < 0 || t1 < 0 || t <= t0 || t <= t1 || nms < 0 || nms > 7 || rad < 0) return DH_ERR_ARG;
  if (!dist) return DH_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)(t - t0) * (t - t1);
  if (stage == 0) {                                     // masking + suppression around the existing edges
    hipLaunchKernelGGL(prox_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dist, t0, t1, t, rad, stereo);
    DH_LAUNCH_CHECK();
    if (n_edges > 0) {
      if (!edges_i || !edges_j) return DH_ERR_ARG;
      const long m = (long)n_edges * (2 * nms + 1) * (2 * nms + 1);
      hipLaunchKernelGGL(prox_suppress_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, dist, edges_i, edges_j, n_edges, t0, t1, t, nms);
      DH_LAUNCH_CHECK();
    }
    return DH_OK;
  }
  if (!order || !sorted || !out_edges || !out_count || max_new < 0) return DH_ERR_ARG;
  hipLaunchKernelGGL(prox_nms_kernel, dim3(1), dim3(256), 0, st, dist, sorted, order, n, t0, t1, t, nms, thresh, max_factors, n_es0, out_edges,
                     max_new, out_count);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
