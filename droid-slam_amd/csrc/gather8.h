// Aligned 8-element window-row gather shared by the correlation kernels.
// A window row is 8 consecutive elements at an arbitrary (element-aligned) offset; it is fetched
// with 16-byte aligned vector loads (2 for f16, 3 for f32) and the 8 taps are extracted in registers
// with a two-stage select + v_alignbit -- no scalar loads, no scratch, no LDS.
#pragma once
#include "common.h"

namespace dh {

template <typename T> struct ChunkTraits;
template <> struct ChunkTraits<__half> { static constexpr int EPC = 8; static constexpr int NCH = 2; };
template <> struct ChunkTraits<float>  { static constexpr int EPC = 4; static constexpr int NCH = 3; };

__device__ __forceinline__ uint32_t sel(bool c, uint32_t a, uint32_t b) { return c ? a : b; }

// fetch the 8 consecutive elements slice[e0 .. e0+7] (e0 may be out of the slice: callers mask)
// using aligned 16-byte loads only.  `vol16` = whole tensor as uint4 chunks, `chunk_lo/hi` = clamp
// range (inclusive) that is safe to read.
__device__ __forceinline__ void fetch8(const uint4* __restrict__ vol16, long e_abs, long chunk_hi,
                                       float (&t)[8], __half) {
  long c0 = e_abs >> 3;                     // 8 halfs per chunk (floor: e_abs may be negative)
  int s = (int)(e_abs - (c0 << 3));         // 0..7
  long ca = c0 < 0 ? 0 : (c0 > chunk_hi ? chunk_hi : c0);
  long cb = c0 + 1 < 0 ? 0 : (c0 + 1 > chunk_hi ? chunk_hi : c0 + 1);
  uint4 A = vol16[ca];
  uint4 B = vol16[cb];
  uint32_t d[8] = {A.x, A.y, A.z, A.w, B.x, B.y, B.z, B.w};
  const bool w1 = (s >> 1) & 1, w2 = (s >> 2) & 1;
  uint32_t f[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) f[i] = sel(w1, d[i + 1], d[i]);
  uint32_t e[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) e[i] = sel(w2, f[i + 2], f[i]);
  const uint32_t sh = (s & 1) * 16;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    uint32_t o = __builtin_amdgcn_alignbit(e[m + 1], e[m], sh);
    __half2 h2 = *reinterpret_cast<__half2*>(&o);
    t[2 * m] = __low2float(h2);
    t[2 * m + 1] = __high2float(h2);
  }
}

__device__ __forceinline__ void fetch8(const uint4* __restrict__ vol16, long e_abs, long chunk_hi,
                                       float (&t)[8], float) {
  long c0 = e_abs >> 2;                     // 4 floats per chunk
  int s = (int)(e_abs - (c0 << 2));         // 0..3
  uint32_t d[12];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    long c = c0 + k;
    c = c < 0 ? 0 : (c > chunk_hi ? chunk_hi : c);
    uint4 A = vol16[c];
    d[4 * k] = A.x; d[4 * k + 1] = A.y; d[4 * k + 2] = A.z; d[4 * k + 3] = A.w;
  }
  const bool b0 = s & 1, b1 = (s >> 1) & 1;
  uint32_t f[11];
#pragma unroll
  for (int i = 0; i < 11; ++i) f[i] = sel(b0, d[i + 1], d[i]);
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = __uint_as_float(sel(b1, f[i + 2], f[i]));
}


}  // namespace dh
