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
  // NOT scaled by 1/16: the reference's corr_backward_kernel (altcorr_kernel.cu:117-125) multiplies the gradient with
  // the raw features although its forward divides both by 4 -- reproduced (found by the oracle/_ref parity test)
  const float g = G(tj, ti) * (1.f - dx) * (1.f - dy) + G(tj - 1, ti) * dx * (1.f - dy) +
                  G(tj, ti - 1) * (1.f - dx) * dy + G(tj - 1, ti - 1) * dx * dy;
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

// ---- MFMA path (channel-last fp16 features) ------------------------------------------------------------------
// The feature correlation is a true contraction over the 128 channels, so it belongs on the matrix cores; what keeps
// it from being a plain GEMM is that every source pixel wants its own 8x8 window of targets.  With a spatially
// coherent flow the windows of an 8x8 source block overlap almost completely: their union fits a 16x16 target tile.
// One wave = one source block:
//   * its 64 source features stay in registers as MFMA B operands (64 VGPRs) for the whole block;
//   * the target tile is streamed in 4 groups of 4 rows: 64 targets x 128 ch are staged through LDS with fully
//     coalesced loads (a target row of 16 pixels is 4 KB contiguous in NHWC; out-of-image targets are staged as
//     zeros, which IS the reference's zero padding), then 32 x v_mfma_f32_32x32x16_f16 give D[target][pixel];
//   * in D's layout a lane holds one pixel's column, so it writes its own pixel's row of a [pixel][target] fp16 tile
//     and then reads back the 8 taps of each window row at its own offset (3 aligned 8-byte reads + funnel shift),
//     interpolates and stores the 49 outputs as the rows arrive.
// 4x the algorithmic flops (16x16 instead of 8x8 targets per pixel) at ~100x the rate of the VALU version.
// Blocks whose windows do not fit the tile (incoherent flow) take a per-lane scalar path with the same results.
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int AC = 128;                // channels
constexpr int SF_LD = AC + 8;          // staged target row stride (halves)
constexpr int SC_LD = 64 + 4;          // [pixel][64 targets] row stride (halves)
constexpr int AWPB = 2;                // waves (source blocks) per workgroup

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}

#ifdef DH_ABLATION   // first form of the MFMA alt-correlation kernel (1.4x slower, bit-identical): A/B only
__global__ __launch_bounds__(AWPB * 64) void altcorr_mfma_kernel(
    const __half* __restrict__ f1, const __half* __restrict__ f2, const float* __restrict__ coords,
    const int64_t* __restrict__ us, const int64_t* __restrict__ vs, __half* __restrict__ corr,
    int H, int W, int H2, int W2, int nblk, float inv_scale, long corr_stride_m) {
  extern __shared__ __half s_alt[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int blk = blockIdx.x * AWPB + wv;
  if (blk >= nblk) return;                         // whole wave (no workgroup barrier is used below)
  __half* sF = s_alt + wv * (64 * SF_LD + 64 * SC_LD);
  __half* sC = sF + 64 * SF_LD;
  const int m = blockIdx.y;
  const int nbx = W / 8, by = blk / nbx, bx = blk - by * nbx;
  const int HW = H * W;
  const int ix = (int)us[m], jx = (int)vs[m];
  const int yy = lane >> 3, xx = lane & 7;
  const int pix = (by * 8 + yy) * W + bx * 8 + xx;
  const float* cb = coords + (long)m * 2 * HW;
  const float x0 = cb[pix] * inv_scale, y0 = cb[HW + pix] * inv_scale;      // (a power of two: exact, as coords / 2**level)
  float fxf = floorf(x0), fyf = floorf(y0);
  const float dx = x0 - fxf, dy = y0 - fyf;
  fxf = fminf(fmaxf(fxf, -65536.f), 65536.f);
  fyf = fminf(fmaxf(fyf, -65536.f), 65536.f);
  const int X0 = (int)fxf - 3, Y0 = (int)fyf - 3;
  const int Xmin = wave_min_i(X0), Xmax = wave_max_i(X0), Ymin = wave_min_i(Y0), Ymax = wave_max_i(Y0);
  __half* out = corr + (long)m * corr_stride_m + pix;
  const __half* f1b = f1 + (long)ix * HW * AC;
  const __half* f2b = f2 + (long)jx * H2 * W2 * AC;

  if (Xmax - Xmin > 8 || Ymax - Ymin > 8) {
    // ---- incoherent block: per-lane scalar evaluation (same arithmetic order of the blend) ----
    const __half* a = f1b + (long)pix * AC;
    float prevs[7];
    for (int j = 0; j < 8; ++j) {
      float t[8];
      for (int i = 0; i < 8; ++i) {
        const int y2 = Y0 + j, x2 = X0 + i;
        float s = 0.f;
        if ((unsigned)y2 < (unsigned)H2 && (unsigned)x2 < (unsigned)W2) {
          const __half* b = f2b + ((long)y2 * W2 + x2) * AC;
          for (int c = 0; c < AC; c += 8) {
            const uint4 av = *reinterpret_cast<const uint4*>(a + c), bv = *reinterpret_cast<const uint4*>(b + c);
            const __half2* ah = reinterpret_cast<const __half2*>(&av);
            const __half2* bh = reinterpret_cast<const __half2*>(&bv);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float2 af = __half22float2(ah[q]), bf = __half22float2(bh[q]); s += af.x * bf.x + af.y * bf.y; }
          }
        }
        t[i] = __half2float(__float2half(s * 0.0625f));
      }
      float c7[7];
      for (int q = 0; q < 7; ++q) c7[q] = t[q] + dx * (t[q + 1] - t[q]);
      if (j > 0)
        for (int q = 0; q < 7; ++q) out[(long)(q * 7 + (j - 1)) * HW] = __float2half(prevs[q] + dy * (c7[q] - prevs[q]));
      for (int q = 0; q < 7; ++q) prevs[q] = c7[q];
    }
    return;
  }

  // ---- source features of the block: B operands, resident ----
  half8 bfrag[2][8];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int pn = nt * 32 + (lane & 31);
    const __half* row = f1b + ((long)(by * 8 + (pn >> 3)) * W + bx * 8 + (pn & 7)) * AC + (lane >> 5) * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) bfrag[nt][ks] = *reinterpret_cast<const half8*>(row + ks * 16);
  }

  float prev[7];
  const int c0 = X0 - Xmin;                       // first tap column inside the 16-wide tile, 0..8
  const int sh = (c0 & 3) * 16;                   // funnel shift (bits) inside the aligned 8-byte words
#pragma unroll 1
  for (int g = 0; g < 4; ++g) {
    const int Yg = Ymin + 4 * g;
    // stage 64 targets (4 rows x 16) x 128 channels: piece id = lane + 64*i -> target id>>4, 16-byte group id&15
    uint4 st[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int id = lane + 64 * i, tg = id >> 4;
      const int y = Yg + (tg >> 4), x = Xmin + (tg & 15);
      const bool ok = (unsigned)y < (unsigned)H2 && (unsigned)x < (unsigned)W2;
      const uint4 v = *reinterpret_cast<const uint4*>(f2b + ((long)(ok ? y : 0) * W2 + (ok ? x : 0)) * AC + (id & 15) * 8);
      const uint32_t mk = ok ? 0xffffffffu : 0u;
      st[i] = uint4{v.x & mk, v.y & mk, v.z & mk, v.w & mk};
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int id = lane + 64 * i;
      *reinterpret_cast<uint4*>(sF + (id >> 4) * SF_LD + (id & 15) * 8) = st[i];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): wave-private tiles
    __builtin_amdgcn_wave_barrier();
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const half8 af = *reinterpret_cast<const half8*>(sF + (mt * 32 + (lane & 31)) * SF_LD + ks * 16 + (lane >> 5) * 8);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bfrag[nt][ks], acc[mt][nt], 0, 0, 0);
      }
    // D[target][pixel] -> sC[pixel][target] (fp16, scaled by 1/16): lane = pixel nt*32 + (lane&31), 4 consecutive targets per store
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const __half2 lo = __floats2half2_rn(acc[mt][nt][4 * qq] * 0.0625f, acc[mt][nt][4 * qq + 1] * 0.0625f);
          const __half2 hi = __floats2half2_rn(acc[mt][nt][4 * qq + 2] * 0.0625f, acc[mt][nt][4 * qq + 3] * 0.0625f);
          uint2 pk; pk.x = __builtin_bit_cast(uint32_t, lo); pk.y = __builtin_bit_cast(uint32_t, hi);
          *reinterpret_cast<uint2*>(sC + (nt * 32 + (lane & 31)) * SC_LD + mt * 32 + 8 * qq + 4 * (lane >> 5)) = pk;
        }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // lane = pixel: consume the rows of this group that belong to its window
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = Yg + r - Y0;                  // window row of this pixel
      if (j < 0 || j > 7) continue;
      const __half* rowp = sC + lane * SC_LD + r * 16 + (c0 & ~3);
      const uint64_t w0 = *reinterpret_cast<const uint64_t*>(rowp);
      const uint64_t w1 = *reinterpret_cast<const uint64_t*>(rowp + 4);
      const uint64_t w2 = *reinterpret_cast<const uint64_t*>(rowp + 8);
      const uint64_t t03 = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
      const uint64_t t47 = sh ? (w1 >> sh) | (w2 << (64 - sh)) : w1;
      float t[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        t[i] = (float)__builtin_bit_cast(_Float16, (unsigned short)(t03 >> (16 * i)));
        t[4 + i] = (float)__builtin_bit_cast(_Float16, (unsigned short)(t47 >> (16 * i)));
      }
      float c7[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) c7[q] = t[q] + dx * (t[q + 1] - t[q]);
      if (j > 0) {
#pragma unroll
        for (int q = 0; q < 7; ++q) out[(long)(q * 7 + (j - 1)) * HW] = __float2half(prev[q] + dy * (c7[q] - prev[q]));
      }
#pragma unroll
      for (int q = 0; q < 7; ++q) prev[q] = c7[q];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);           // tile reads done before the next group overwrites it
    __builtin_amdgcn_wave_barrier();
  }
}
#endif  // DH_ABLATION

// ---- second form of the MFMA kernel (default; DH_ALTCORR_V1=1 selects the first) ----------------------------------------
// The first form spends its time waiting: a wave stages 64 targets x 256 B through 64 registers, then multiplies, then
// blends, four times in a row, and 26 KB of LDS per wave leave six waves per CU to hide that chain (measured 437 us per
// 512 edges and level: 2 % of the MFMA peak).  Here
//   * targets arrive by LDS-DMA (`global_load_lds_dwordx4`: no staging registers), 32 targets (2 window rows x 16 columns)
//     per group = eight 1-KB pieces; rows are 256 B unpadded, the 16-byte slot of a row XOR-ed with the row index (the
//     DMA's lane-linear LDS image is permuted on the SOURCE side: lane (target t, slot s) fetches slot s ^ (t & 15)), so
//     the fragment reads stay conflict-free;
//   * the next group's DMA is issued as soon as the current group's MFMAs have consumed the tile, and lands while the
//     products are written to the [pixel][target] tile and blended;
//   * out-of-image targets are fetched from a clamped address and zeroed at the blend (the reference's zero padding);
//   * 12.6 KB of LDS and ~130 registers per wave: 12 waves per CU.
// Same arithmetic as the first form: fp32 accumulation over the 128 channels, product rounded to fp16 after the 1/16
// scale, separable blend in fp32.
typedef __attribute__((address_space(3))) char lds_char_t;
constexpr int A2_WPB = 4;                      // waves (source blocks) per workgroup
constexpr int A2_TG = 32;                      // targets per group
constexpr int A2_F_BYTES = A2_TG * AC * 2;     // 8192
constexpr int A2_C_LD = A2_TG + 4;             // [pixel][32 targets (+4)] row stride in halves: 72 B
constexpr int A2_C_BYTES = 64 * A2_C_LD * 2;   // 4608
constexpr int A2_WAVE_BYTES = A2_F_BYTES + A2_C_BYTES;

__global__ __launch_bounds__(A2_WPB * 64, 3) void altcorr_mfma2_kernel(
    const __half* __restrict__ f1, const __half* __restrict__ f2, const float* __restrict__ coords,
    const int64_t* __restrict__ us, const int64_t* __restrict__ vs, __half* __restrict__ corr,
    int H, int W, int H2, int W2, int nblk, float inv_scale, long corr_stride_m) {
  extern __shared__ __attribute__((aligned(16))) char s_alt2[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int blk = blockIdx.x * A2_WPB + wv;
  if (blk >= nblk) return;                         // whole wave (no workgroup barrier below)
  char* const sF = s_alt2 + wv * A2_WAVE_BYTES;
  __half* const sC = reinterpret_cast<__half*>(sF + A2_F_BYTES);
  const unsigned sF_lds = (unsigned)(uintptr_t)(lds_char_t*)sF;
  const int m = blockIdx.y;
  const int nbx = W / 8, by = blk / nbx, bx = blk - by * nbx;
  const int HW = H * W;
  const int ix = (int)us[m], jx = (int)vs[m];
  const int yy = lane >> 3, xx = lane & 7;
  const int pix = (by * 8 + yy) * W + bx * 8 + xx;
  const float* cb = coords + (long)m * 2 * HW;
  const float x0 = cb[pix] * inv_scale, y0 = cb[HW + pix] * inv_scale;
  float fxf = floorf(x0), fyf = floorf(y0);
  const float dx = x0 - fxf, dy = y0 - fyf;
  fxf = fminf(fmaxf(fxf, -65536.f), 65536.f);
  fyf = fminf(fmaxf(fyf, -65536.f), 65536.f);
  const int X0 = (int)fxf - 3, Y0 = (int)fyf - 3;
  const int Xmin = wave_min_i(X0), Xmax = wave_max_i(X0), Ymin = wave_min_i(Y0), Ymax = wave_max_i(Y0);
  __half* out = corr + (long)m * corr_stride_m + pix;
  const __half* f1b = f1 + (long)ix * HW * AC;
  const __half* f2b = f2 + (long)jx * H2 * W2 * AC;

  if (Xmax - Xmin > 8 || Ymax - Ymin > 8) {
    // ---- incoherent block: per-lane scalar evaluation (same arithmetic order of the blend) ----
    const __half* a = f1b + (long)pix * AC;
    float prevs[7];
    for (int j = 0; j < 8; ++j) {
      float t[8];
      for (int i = 0; i < 8; ++i) {
        const int y2 = Y0 + j, x2 = X0 + i;
        float sacc = 0.f;
        if ((unsigned)y2 < (unsigned)H2 && (unsigned)x2 < (unsigned)W2) {
          const __half* b = f2b + ((long)y2 * W2 + x2) * AC;
          for (int c = 0; c < AC; c += 8) {
            const uint4 av = *reinterpret_cast<const uint4*>(a + c), bv = *reinterpret_cast<const uint4*>(b + c);
            const __half2* ah = reinterpret_cast<const __half2*>(&av);
            const __half2* bh = reinterpret_cast<const __half2*>(&bv);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float2 af = __half22float2(ah[q]), bf = __half22float2(bh[q]); sacc += af.x * bf.x + af.y * bf.y; }
          }
        }
        t[i] = __half2float(__float2half(sacc * 0.0625f));
      }
      float c7[7];
      for (int q = 0; q < 7; ++q) c7[q] = t[q] + dx * (t[q + 1] - t[q]);
      if (j > 0)
        for (int q = 0; q < 7; ++q) out[(long)(q * 7 + (j - 1)) * HW] = __float2half(prevs[q] + dy * (c7[q] - prevs[q]));
      for (int q = 0; q < 7; ++q) prevs[q] = c7[q];
    }
    return;
  }

  // ---- source features of the block: B operands, resident (lane holds column = pixel nt*32 + (lane & 31), k-chunk lane >> 5)
  half8 bfrag[2][8];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int pn = nt * 32 + (lane & 31);
    const __half* row = f1b + ((long)(by * 8 + (pn >> 3)) * W + bx * 8 + (pn & 7)) * AC + (lane >> 5) * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) bfrag[nt][ks] = *reinterpret_cast<const half8*>(row + ks * 16);
  }
  // DMA role: piece i of a group covers targets 4i .. 4i+3; lane -> (target 4i + (lane >> 4), LDS slot lane & 15)
  const int d_t = lane >> 4, d_s = lane & 15;
  // one group = window rows Yg, Yg + 1; 16 columns from Xmin.  Source address per lane, clamped into the image.
#define A2_DMA_GROUP(g_)                                                                                               \
  {                                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                    \
      const int t = 4 * i + d_t;                              /* target inside the group */                            \
      const int y = min(max(Ymin + 2 * (g_) + (t >> 4), 0), H2 - 1), x = min(max(Xmin + (t & 15), 0), W2 - 1);         \
      const int voff = (y * W2 + x) * (AC * 2) + ((d_s ^ (t & 15)) << 4);                                              \
      const unsigned dst = __builtin_amdgcn_readfirstlane(sF_lds + i * 1024);                                          \
      unsigned keep_;                                                                                                  \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                   : "=&s"(keep_) : "v"(voff), "s"(f2b), "s"(dst) : "memory");                                         \
    }                                                                                                                  \
  }
  const int c0 = X0 - Xmin;                       // first tap column inside the 16-wide tile, 0..8
  const int sh = (c0 & 3) * 16;                   // funnel shift (bits) inside the aligned 8-byte words
  // column validity of this lane's 8 taps (row validity is per group row)
  uint32_t colmask = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) colmask |= ((unsigned)(X0 + i) < (unsigned)W2 ? 1u : 0u) << i;
  float prev[7];
  A2_DMA_GROUP(0)
#pragma unroll 1
  for (int g = 0; g < 8; ++g) {
    const int Yg = Ymin + 2 * g;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this group's eight pieces have landed (the bfrag loads too)
    __builtin_amdgcn_wave_barrier();
    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[b][q] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      // A fragment: row = target lane & 31, channels ks*16 + (lane >> 5)*8 .. +8 = logical slot 2*ks + (lane >> 5)
      const int t = lane & 31;
      const half8 af = *reinterpret_cast<const half8*>(sF + t * 256 + (((2 * ks + (lane >> 5)) ^ (t & 15)) << 4));
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bfrag[nt][ks], acc[nt], 0, 0, 0);
    }
    // the tile has been read (the MFMAs above hold their operands in registers once issued; wait for the LDS reads)
    __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    if (g + 1 < 8) A2_DMA_GROUP(g + 1)
    // D[target][pixel] -> sC[pixel][target] (fp16, scaled by 1/16): lane = pixel nt*32 + (lane & 31), 4 consecutive targets per store
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const __half2 lo = __floats2half2_rn(acc[nt][4 * qq] * 0.0625f, acc[nt][4 * qq + 1] * 0.0625f);
        const __half2 hi = __floats2half2_rn(acc[nt][4 * qq + 2] * 0.0625f, acc[nt][4 * qq + 3] * 0.0625f);
        uint2 pk; pk.x = __builtin_bit_cast(uint32_t, lo); pk.y = __builtin_bit_cast(uint32_t, hi);
        *reinterpret_cast<uint2*>(sC + (nt * 32 + (lane & 31)) * A2_C_LD + 8 * qq + 4 * (lane >> 5)) = pk;
      }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // lane = pixel: consume the rows of this group that belong to its window
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int j = Yg + r - Y0;                  // window row of this pixel
      if (j < 0 || j > 7) continue;
      const __half* rowp = sC + lane * A2_C_LD + r * 16 + (c0 & ~3);
      const uint64_t w0 = *reinterpret_cast<const uint64_t*>(rowp);
      const uint64_t w1 = *reinterpret_cast<const uint64_t*>(rowp + 4);
      const uint64_t w2 = *reinterpret_cast<const uint64_t*>(rowp + 8);
      const uint64_t t03 = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
      const uint64_t t47 = sh ? (w1 >> sh) | (w2 << (64 - sh)) : w1;
      const uint32_t rmask = (unsigned)(Yg + r) < (unsigned)H2 ? colmask : 0u;
      float t[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        t[i] = (rmask >> i) & 1u ? (float)__builtin_bit_cast(_Float16, (unsigned short)(t03 >> (16 * i))) : 0.f;
        t[4 + i] = (rmask >> (4 + i)) & 1u ? (float)__builtin_bit_cast(_Float16, (unsigned short)(t47 >> (16 * i))) : 0.f;
      }
      float c7[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) c7[q] = t[q] + dx * (t[q + 1] - t[q]);
      if (j > 0) {
#pragma unroll
        for (int q = 0; q < 7; ++q) out[(long)(q * 7 + (j - 1)) * HW] = __float2half(prev[q] + dy * (c7[q] - prev[q]));
      }
#pragma unroll
      for (int q = 0; q < 7; ++q) prev[q] = c7[q];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);           // sC reads done before the next group overwrites it
    __builtin_amdgcn_wave_barrier();
  }
#undef A2_DMA_GROUP
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

extern "C" int dh_altcorr_fwd_nhwc(const void* fmap1, const void* fmap2, const float* coords,
                                   const int64_t* ii, const int64_t* jj, void* corr,
                                   int N1, int N2, int C, int H, int W, int H2, int W2, int M, dh_stream_t stream) {
  return dh_altcorr_fwd_nhwc_level(fmap1, fmap2, coords, ii, jj, corr, N1, N2, C, H, W, H2, W2, M, 0, 49L * H * W, stream);
}

extern "C" int dh_altcorr_fwd_nhwc_level(const void* fmap1, const void* fmap2, const float* coords,
                                         const int64_t* ii, const int64_t* jj, void* corr,
                                         int N1, int N2, int C, int H, int W, int H2, int W2, int M,
                                         int level, long corr_stride_m, dh_stream_t stream) {
  if (N1 <= 0 || N2 <= 0 || H <= 0 || W <= 0 || H2 <= 0 || W2 <= 0 || M < 0) return DH_ERR_ARG;
  if (level < 0 || level > 15 || corr_stride_m < 49L * H * W) return DH_ERR_ARG;
  if (C != AC || H % 8 || W % 8) return DH_ERR_UNSUPPORTED;
  if (M == 0) return DH_OK;
  if (!fmap1 || !fmap2 || !coords || !ii || !jj || !corr) return DH_ERR_ARG;
  const int nblk = (H / 8) * (W / 8);
#ifdef DH_ABLATION
  if (opts().altcorr_v1) {
    const size_t lds = (size_t)AWPB * (64 * SF_LD + 64 * SC_LD) * sizeof(__half);
    hipLaunchKernelGGL(altcorr_mfma_kernel, dim3((nblk + AWPB - 1) / AWPB, M), dim3(AWPB * 64), lds, (hipStream_t)stream,
                       (const __half*)fmap1, (const __half*)fmap2, coords, ii, jj, (__half*)corr, H, W, H2, W2, nblk,
                       1.0f / (float)(1 << level), corr_stride_m);
  } else
#endif
  {
    if (((uintptr_t)fmap2) % 16 || (long)H2 * W2 * AC * 2 >= (1L << 31)) return DH_ERR_ARG;      // 16-byte DMA pieces, 32-bit lane offsets
    hipLaunchKernelGGL(altcorr_mfma2_kernel, dim3((nblk + A2_WPB - 1) / A2_WPB, M), dim3(A2_WPB * 64), (size_t)A2_WPB * A2_WAVE_BYTES,
                       (hipStream_t)stream, (const __half*)fmap1, (const __half*)fmap2, coords, ii, jj, (__half*)corr, H, W, H2, W2, nblk,
                       1.0f / (float)(1 << level), corr_stride_m);
  }
  DH_LAUNCH_CHECK();
  return DH_OK;
}
