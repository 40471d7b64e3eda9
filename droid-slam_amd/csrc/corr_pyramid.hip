// MI355X-native correlation pyramid: own HBM layout, MFMA build, ONE fused 4-level lookup kernel.
//
// Replaces CorrBlock.__init__ (all-pairs torch.matmul + 3 avg_pool2d, reference
// droid_slam/modules/corr.py:23-38,63-71) and CorrBlock.__call__ (4 x corr_index_forward launches,
// corr.py:40-50 -> src/correlation_kernels.cu:20-71) for callers that let this library own the volume.
//
// Why a new layout.  In the reference layout [E,h1,w1,h2,w2] every source pixel owns a private slice, so
// an 8x8 window touches 8 different 128-byte lines to use 16 bytes of each (measured on MI355X: the
// per-level kernel moves ~4.3 TB/s of cache lines for ~0.55 TB/s of useful taps).  Here the volume of
// level l is stored as
//
//     V'[e][level][sb = 8x8 source block][v][u/2][p][u&1]   (fp16, p = 6-bit index inside the source block)
//     v = (y2 - (y1 >> l)) mod h2_l ,  u = (x2 - (x1 >> l)) mod w2_l        ("displacement" coordinates)
//     plus one all-zero row v = h2_l per source block (where out-of-image window rows are pointed)
//
// i.e. 256 contiguous bytes hold, for ONE pair of x-adjacent displacements, the 64 pixels of an 8x8 source
// block, and every lane finds two of its own window taps in one aligned dword.  A wave = one source block;
// when the flow is spatially coherent (it is: it comes from a reprojection) all 64 lanes want the same
// displacement cells, so a tap load of the wave is one fully used 256-byte run and the window rows of a wave
// are consecutive runs.  Why pairs: the texture-address path of a CU retires a wave-wide load in ~6-8 cycles
// per 128-byte line it touches whatever the width per lane (measured, scripts/ubench/ta_rate.hip), and at the
// coarser levels neighbouring lanes alternate between two adjacent cells, so 2-byte-per-lane loads (one cell
// per 128-byte line) cost ~14 cycles each and 64 of them per level made the lookup TA-bound at 3.3 TB/s; with
// pairs a window row is 5 dword loads instead of 8 ushort loads.  Out-of-image columns are zeroed in
// registers from the un-wrapped x2.  Size: reference pyramid + 1/h2_l per level (zero rows), 25.6 MB/edge.
//
// Build: per level a true contraction over the 128 feature channels on the fp16 MFMA
// (v_mfma_f32_16x16x32_f16), V_l = f1^T * pool_l(f2) / 16 (pooling commutes with the contraction; the
// reference's alt path pools features the same way, corr.py:89-101).  Default: the row-ring kernel
// (pyr_build_ring_kernel: a workgroup walks the target rows of a level, a ring of 8 >> l displacement rows in
// LDS completes one row per step and leaves as one contiguous run of full lines; 2.75 ms per 256 edges).  The
// first form (pyr_build_kernel, DH_PYR_BUILD=chunk: one wave per 64-target chunk, wave-private skewed LDS tile,
// 32-byte pieces to HBM; 5.3 ms) is kept for A/B runs and as the bit-exactness partner of the ring kernel.
//
// Lookup: workgroup = 8-row strip of one edge (w/8 waves, one per 8x8 block), all 4 levels in one launch
// (coords read once), separable bilinear interpolation in fp32 registers, results staged per level in
// LDS and written as full 128-byte rows of the [E,196,h,w] output.
#include "common.h"
#include <cstdlib>

namespace {
using namespace dh;

constexpr int NLEV = 4;
constexpr int CH = 128;          // feature channels (K of the contraction)
constexpr int RAD = 3;
constexpr int WIN = 2 * RAD + 2; // 8 integer taps per axis
constexpr int OUTW = 2 * RAD + 1;
constexpr int NCH_OUT = OUTW * OUTW;   // 49

struct PyrDims {
  int h, w, nblk;                // source image, number of 8x8 source blocks
  int h2[NLEV], w2[NLEV];
  long blk_elems[NLEV];          // elements of one source block of a level: (h2 + 1 zero row) * w2 * 64
  long lev_off[NLEV];            // element offset of each level inside one edge
  long edge_elems;
  int tgt_off[NLEV + 1];         // target-pixel offset of each level in the pooled f2 pyramid
};

__host__ __device__ inline PyrDims make_dims(int h, int w) {
  PyrDims d;
  d.h = h; d.w = w; d.nblk = (h / 8) * (w / 8);
  long off = 0; int t = 0;
  for (int l = 0; l < NLEV; ++l) {
    d.h2[l] = h >> l; d.w2[l] = w >> l;
    d.blk_elems[l] = (long)(d.h2[l] + 1) * d.w2[l] * 64;
    d.lev_off[l] = off; d.tgt_off[l] = t;
    off += (long)d.nblk * d.blk_elems[l];
    t += d.h2[l] * d.w2[l];
  }
  d.tgt_off[NLEV] = t;
  d.edge_elems = off;
  return d;
}

// element offset of cell (v,u), pixel p inside a source block of a level with row length w2 (even)
__host__ __device__ inline long cell_off(int v, int u, int p, int w2) {
  return (((long)v * (w2 >> 1) + (u >> 1)) * 64 + p) * 2 + (u & 1);
}

// ---------------------------------------------------------------------------------------- prep
// channel-major fp16 maps [E,C,h,w] -> channel-last rows: f1T [E,HW,C], f2T [E,T,C] with the 4 pooled
// levels stacked (T = sum_l h2_l*w2_l); pooling accumulates in fp32 from the level below (rounded to fp16
// per level, like F.avg_pool2d on an fp16 tensor).
__global__ __launch_bounds__(256) void pyr_transpose_kernel(const __half* __restrict__ src, __half* __restrict__ dst,
                                                            int HW, long dst_stride_e, int dst_row0) {
  // tile: 64 pixels x 128 channels through LDS
  __shared__ __half tile[64][CH + 2];
  const int e = blockIdx.y;
  const int p0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  const __half* s = src + (long)e * CH * HW;
  for (int o = tid; o < 64 * CH; o += 256) {
    const int c = o >> 6, p = o & 63;
    tile[p][c] = (p0 + p < HW) ? s[(long)c * HW + p0 + p] : __float2half(0.f);
  }
  __syncthreads();
  __half* d = dst + (long)e * dst_stride_e + (long)(dst_row0 + p0) * CH;
  for (int o = tid; o < 64 * CH / 2; o += 256) {
    const int p = o / (CH / 2), c2 = o % (CH / 2);
    if (p0 + p < HW) reinterpret_cast<__half2*>(d)[(long)p * (CH / 2) + c2] = __halves2half2(tile[p][2 * c2], tile[p][2 * c2 + 1]);
  }
}

// pooled level l from level l-1 of the channel-last pyramid (rows = target pixels)
// (h_valid, w_valid): cells of the output level beyond them are stored as zeros -- the image-on-a-canvas mode of
// dh_corr_pyramid_build_canvas, where avg_pool2d's floor (corr.py:36: 30x40 -> 15x20 -> 7x10 -> 3x5) must not pool the
// canvas' zero padding into a level's last row / column
__global__ __launch_bounds__(128) void pyr_pool_kernel(__half* __restrict__ f2T, long stride_e, int row_in, int row_out,
                                                       int h_in, int w_in, int h_valid, int w_valid) {
  const int e = blockIdx.y;
  const int q = blockIdx.x;                      // output pixel
  const int w_out = w_in >> 1;
  const int y = q / w_out, x = q - y * w_out;
  const int c = threadIdx.x;
  const __half* in = f2T + (long)e * stride_e + (long)row_in * CH;
  const long r00 = ((long)(2 * y) * w_in + 2 * x) * CH + c;
  const float s = __half2float(in[r00]) + __half2float(in[r00 + CH]) + __half2float(in[r00 + (long)w_in * CH]) +
                  __half2float(in[r00 + (long)w_in * CH + CH]);
  f2T[(long)e * stride_e + (long)(row_out + q) * CH + c] = (y < h_valid && x < w_valid) ? __float2half(0.25f * s) : __float2half(0.f);
}

// ---------------------------------------------------------------------------------------- build
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

__device__ __forceinline__ int wrap(int a, int n) { return a < 0 ? a + n : (a >= n ? a - n : a); }

#ifdef DH_ABLATION   // first form of the build kernel, kept for A/B measurements (scripts/check_build_ab.py)
// workgroup = 4 waves, one 8x8 source block of one edge; waves take target chunks of 64 pixels
__global__ __launch_bounds__(256) void pyr_build_kernel(const __half* __restrict__ f1T, const __half* __restrict__ f2T,
                                                        __half* __restrict__ pyr, PyrDims D, long f1_stride_e,
                                                        long f2_stride_e) {
  __shared__ __half s_tile[4][64 * 64];          // wave-private staging, skew-ordered
  const int e = blockIdx.y, sb = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nbx = D.w / 8;
  const int by = sb / nbx, bx = sb - by * nbx;
  // A fragments (source pixels): m-tile t covers source rows yy = 2t, 2t+1; lane&15 -> p = t*16 + (lane&15)
  // MFMA 16x16x32: lane holds A[i = lane&15][k = (lane>>4)*8 .. +8]
  half8 afrag[4][4];
  {
    const __half* a = f1T + (long)e * f1_stride_e;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int p = t * 16 + (lane & 15);
      const int y1 = by * 8 + (p >> 3), x1 = bx * 8 + (p & 7);
      const __half* row = a + ((long)y1 * D.w + x1) * CH;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        afrag[t][ks] = *reinterpret_cast<const half8*>(row + ks * 32 + (lane >> 4) * 8);
    }
  }
  __half* stage = s_tile[wave];
  const __half* bbase = f2T + (long)e * f2_stride_e;
  __half* obase = pyr + (long)e * D.edge_elems;

  // chunk list over all levels: chunk c of level l covers targets [c*64, c*64+64) of that level
  int nchunk[NLEV], cstart[NLEV + 1];
  cstart[0] = 0;
#pragma unroll
  for (int l = 0; l < NLEV; ++l) { nchunk[l] = (D.h2[l] * D.w2[l] + 63) / 64; cstart[l + 1] = cstart[l] + nchunk[l]; }
  for (int cc = wave; cc < cstart[NLEV]; cc += 4) {
    int l = 0;
#pragma unroll
    for (int t = 1; t < NLEV; ++t) if (cc >= cstart[t]) l = t;
    const int q0 = (cc - cstart[l]) * 64;
    const int h2 = D.h2[l], w2 = D.w2[l], T = h2 * w2;
    const int nq = min(64, T - q0);
    // B fragments straight from HBM/L2: lane holds B[k = (lane>>4)*8..+8][j = lane&15]  (target q0 + nt*16 + j)
    f32x4 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int q = q0 + nt * 16 + (lane & 15);
      const int qc = q < T ? q : T - 1;
      const __half* brow = bbase + ((long)D.tgt_off[l] + qc) * CH;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const half8 b = *reinterpret_cast<const half8*>(brow + ks * 32 + (lane >> 4) * 8);
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag[t][ks], b, acc[t][nt], 0, 0, 0);
      }
    }
    // D[row = p (source), col = q (target)]: lane holds col = lane&15, rows (lane>>4)*4 + r of each tile.
    // Staging tile [pair of cells][p][2]: a target keeps its row inside the chunk and swaps x2 for u, so
    // cell_local = q_local - x2 + u; cells (2k, 2k+1) of a row are interleaved per pixel like in HBM.
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int ql = nt * 16 + (lane & 15);
      const int q = q0 + ql;
      const int y2 = q / w2, x2 = q - y2 * w2;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int p = t * 16 + (lane >> 4) * 4 + r;
          const int x1l = (bx * 8 + (p & 7)) >> l;
          const int u = wrap(x2 - x1l, w2);
          const int cl = ql - x2 + u;
          stage[((cl >> 1) * 64 + p) * 2 + (cl & 1)] = __float2half(acc[t][nt][r] * 0.0625f);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): wave-private tile complete
    __builtin_amdgcn_wave_barrier();
    // read back 16-byte pieces: piece = (pair of cells, source row yy, half row) = 4 source pixels x 2 cells
    // global cell: v = wrap(y2 - (y1 >> l)), same u
    __half* lbase = obase + D.lev_off[l] + (long)sb * D.blk_elems[l];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int piece = it * 64 + lane;
      const int pr = piece >> 4, yy = (piece >> 1) & 7, xh = piece & 1;
      const int cl = pr * 2;
      if (cl < nq) {
        const int q = q0 + cl;
        const int y2 = q / w2, u = q - y2 * w2;
        const int v = wrap(y2 - ((by * 8 + yy) >> l), h2);
        const int p = yy * 8 + xh * 4;
        const uint4 val = *reinterpret_cast<const uint4*>(stage + (pr * 64 + p) * 2);
        *reinterpret_cast<uint4*>(lbase + cell_off(v, u, p, w2)) = val;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // the all-zero row v = h2 of every level of this source block
#pragma unroll
  for (int l = 0; l < NLEV; ++l) {
    uint4* z = reinterpret_cast<uint4*>(obase + D.lev_off[l] + (long)sb * D.blk_elems[l] + (long)D.h2[l] * D.w2[l] * 64);
    const int n16 = D.w2[l] * 64 / 8;
    for (int o = tid; o < n16; o += 256) z[o] = uint4{0u, 0u, 0u, 0u};
  }
}
#endif  // DH_ABLATION

// ---- measurement switches of the ring build (round 6; profiles/r06_pyr_build_ab.txt) ------------------------------------------------
// The ring loop prefetches a target row three steps ahead (global loads) and writes one finished displacement row per step (global
// stores); as C++ the loop gets `s_waitcnt vmcnt(0)` at its head (LLVM stops counting across the back edge), i.e. every third step a
// wave waits for all its outstanding stores.  DH_PYR_ASM issues the loads (bit 0) / stores (bit 1) as inline asm, invisible to that
// pass, with the one wait a step needs counted by hand (vmcnt returns in order; 2 x NQ loads are issued behind the row a step waits
// for, stores ignored = conservative; rows are then fetched at EVERY step and drained once per level).  MEASURED NEUTRAL (2.43 vs
// 2.48 ms per 256 edges) -- the stall is not what bounds the kernel -- and an asm load keeps landing in registers the compiler has
// reused if a wave never waits again, so the shipped build uses plain C++ (0).  DH_PYR_ABL removes one stage at a time (wrong results,
// timing only): the stages' times ADD UP (stores 0.67 + row fetches 0.54 + B reads / MFMA 0.48 + scatter 0.19 + row staging 0.09 ms
// of 2.57; bare loop 0.68): a step is a serial chain, and two thirds of it is the CU's vector-memory path (192 lines per block-step).
// Second pass (counters, V2 / V3 / pipelined / two-block forms, phase timeline): profiles/r06_v_pyr_build_pmc.txt.
#ifndef DH_PYR_ASM
#define DH_PYR_ASM 0        // measurement switch (scripts/bench_pyr_build.py, profiles/r06_pyr_build_ab.txt): bit 0 = loads as asm, bit 1 = stores as asm; 0 = plain C++ (shipped)
#endif
#ifndef DH_PYR_ABL
#define DH_PYR_ABL 0        // timing ablations of the ring build (WRONG results; never in a shipped build): 1 = no record stores, 2 = no scatter into the ring,
#endif                      // 4 = no B-fragment reads / MFMAs, 8 = no global loads of the target rows, 16 = no LDS writes of them
#ifndef DH_PYR_V2
#define DH_PYR_V2 1         // round 6, second pass (profiles/r06_v_pyr_build_pmc.txt: 124 instructions per wave and step, 56 of them scalar; trimming them to
#endif                      // 97 bought 5 % -- the kernel is not issue-bound either).  1 = (a) the ring's bank swizzle moves whole 16-byte quads only (bits 2-4 of the
                            // pixel index instead of bits 0, 1, 4), so a finished row leaves LDS with ONE ds_read_b128 per lane and no 12-instruction
                            // dword rotation; (b) the next target row's address is kept as a pointer that wraps instead of a `% h2` per fetch (22
                            // scalar instructions).  Same MFMAs, same rounding, same records.  0 = rounds 2-6a (variant builds, for A/B runs).
__device__ __forceinline__ uint32_t pack_h2_scaled(float a, float b) {      // {fp16(a / 16), fp16(b / 16)}, round to nearest even like the 2-byte scatter
  const __half2 h = __halves2half2(__float2half(a * 0.0625f), __float2half(b * 0.0625f));
  return __builtin_bit_cast(uint32_t, h);
}
#ifdef DH_PYR_TS            // scripts/ubench/pyr_ts.hip: phase timestamps (s_memtime) of two waves of ONE workgroup (blockIdx = (5, DH_PYR_TS)) at level 0,
__device__ unsigned long long g_pyr_ts[2 * 16 * 8];      // steps 16 .. 31 (never defined in the library build)
#define PYR_TS_DECL(l_)                                                                                              \
  const int ts_w = ((l_) == 0 && blockIdx.x == 5 && blockIdx.y == DH_PYR_TS) ? (tid_all == 0 ? 0 : tid_all == DH_PYR_TS_WAVE * 64 ? 1 : -1) : -1; \
  unsigned long long ts_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PYR_TS(i) ts_[i] = __builtin_readcyclecounter();
#define PYR_TS_FLUSH(k_)                                                                                             \
  if (ts_w >= 0 && (k_) >= 16 && (k_) < 32) { _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) g_pyr_ts[(ts_w * 16 + (k_) - 16) * 8 + i_] = ts_[i_]; }
#else
#define PYR_TS_DECL(l_)
#define PYR_TS(i)
#define PYR_TS_FLUSH(k_)
#endif
#ifndef DH_PYR_NT
#define DH_PYR_NT 1         // 1 (shipped: -3 % same-box, profiles/r06_v_pyr_build_pmc.txt item 8) = the record stores of the ring build carry the non-temporal hint; 0 = plain stores (variant builds)
#endif
#ifndef DH_PYR_V3
#define DH_PYR_V3 0         // (bit-identical, NOT faster: profiles/r06_v_pyr_build_pmc.txt) round 6, third pass (scripts/ubench/pyr_ts.hip: the 64 ds_write_b16 of a step take 470-570 of its ~2 000 cycles).  1 = the MFMA's
#endif                      // operands are SWAPPED (targets as rows, source pixels as columns): a lane then holds 4 consecutive targets = 4 consecutive
                            // displacements u of ONE pixel, and with a wave's 16 pixels chosen with equal parity of x1 >> l and its target tiles shifted by
                            // that parity, u(r = 0) is even: the 4 values are two whole cell pairs -> 2 ds_write_b32 instead of 4 ds_write_b16 per tile,
                            // 16 pixels of 4 rows on 16 distinct banks without any swizzle.  Same products summed in the same k order: same records.
                            // Needs DH_PYR_V2 (quad-only swizzle, here none).  0 = rounds 2-6b (variant builds).
#ifndef DH_PYR_PIPE
#define DH_PYR_PIPE 0       // (measured SLOWER: 2.44 vs 2.39 ms single, 2.43 vs 2.26 ms dual, profiles/r06_v_pyr_build_pmc.txt) 1 = the scatter of a step is issued between the MFMAs of the next one (ring_level, RING_ITER); 0 = RING_STEP (variant builds)
#endif
// dword swizzle of the ring: pixel p of cell pair `up` lives at dword up * 64 + (p ^ ring_f(up))
#ifndef DH_PYR_SWZ
#define DH_PYR_SWZ 0        // 1 (variant builds) = rounds 2-5's conflict-free dword swizzle (bits 0, 1, 4) kept for the scatter, the read-out as FOUR ds_read_b32
#endif                      // at pre-rotated addresses instead of one ds_read_b128 + 12 v_cndmask (V2's quad swizzle costs 111 M bank-conflict cycles per launch)
__device__ __forceinline__ int ring_f(int up) { return (DH_PYR_V3 && DH_PYR_V2) ? 0 : (DH_PYR_V2 && !DH_PYR_SWZ) ? ((up & 7) << 2) : ((up & 3) | ((up & 4) << 2)); }
// one 16-byte piece (pixels p0 .. p0 + 3 of cell pair up) of a finished ring row, in HBM order
__device__ __forceinline__ u32x4 ring_read_piece(const unsigned char* src, int up, int p0) {
  const int f = ring_f(up);
  if (DH_PYR_SWZ) {
    const unsigned char* q = src + ((up * 64 + (p0 ^ (f & 16))) << 2);
    const int sw = f & 3;
    u32x4 o;
    o[0] = *reinterpret_cast<const uint32_t*>(q + ((0 ^ sw) << 2));
    o[1] = *reinterpret_cast<const uint32_t*>(q + ((1 ^ sw) << 2));
    o[2] = *reinterpret_cast<const uint32_t*>(q + ((2 ^ sw) << 2));
    o[3] = *reinterpret_cast<const uint32_t*>(q + ((3 ^ sw) << 2));
    return o;
  }
  if (DH_PYR_V2) return *reinterpret_cast<const u32x4*>(src + ((up * 64 + (p0 ^ f)) << 2));
  const uint4 c = *reinterpret_cast<const uint4*>(src + ((up * 64 + (p0 ^ (f & 16))) << 2));
  const int sw = f & 3;               /* out[i] = in[i ^ sw] */
  u32x4 o;            /* a native vector: member-wise built HIP uint4 stores are split into 4 dword stores */
  o[0] = sw == 0 ? c.x : sw == 1 ? c.y : sw == 2 ? c.z : c.w;
  o[1] = sw == 0 ? c.y : sw == 1 ? c.x : sw == 2 ? c.w : c.z;
  o[2] = sw == 0 ? c.z : sw == 1 ? c.w : sw == 2 ? c.x : c.y;
  o[3] = sw == 0 ? c.w : sw == 1 ? c.z : sw == 2 ? c.y : c.x;
  return o;
}
__device__ __forceinline__ void gload16_async(u32x4& dst, const __half* p) {
#if DH_PYR_ABL & 8
  dst = u32x4{(uint32_t)(uintptr_t)p, 1u, 2u, 3u};
#elif DH_PYR_ASM & 1
  asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
#else
  dst = *reinterpret_cast<const u32x4*>(p);
#endif
}
__device__ __forceinline__ void gstore16_async(__half* p, const u32x4& v) {
#if DH_PYR_ABL & 1
  if (v[0] == 0x7fc07fc1u) *reinterpret_cast<u32x4*>(p) = v;      // (never true: keeps the read-out alive, drops the store)
#elif DH_PYR_ASM & 2
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#elif DH_PYR_NT
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));      // the records are read back much later (the lookups of the following iterations)
#else
  *reinterpret_cast<u32x4*>(p) = v;
#endif
}
template <int N>
__device__ __forceinline__ void wait_vm_for(u32x4& a) {
#if DH_PYR_ASM & 1
#ifdef DH_PYR_WAIT
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(DH_PYR_WAIT) : "memory");
#else
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory");
#endif
#endif
}

// ---- build, second form: row ring ------------------------------------------------------------------------------------
// The chunk kernel above hands HBM 32-byte pieces (a 128-byte line per piece on the store path) and every wave reloads
// its target rows in half-line fragments: measured 5.3 ms per 256 edges, bound by the vector-memory path, not by the
// MFMA or HBM.  Here a workgroup (one 8x8 source block, 4 waves = 4 source row pairs) walks the target rows of a level
// in order.  A target row y2 contributes to the displacement rows v = y2 - y1 of the block's source rows, so a ring
// of R = 8 >> l displacement rows in LDS ([cell pair][pixel][2], the HBM order, dword-swizzled so that the 2-byte
// scatter of the MFMA result is conflict-free) completes exactly one row per step, which leaves as ONE contiguous run
// (8 KB at level 0) of full lines.  The target row itself is staged once per workgroup in full 128-byte lines (LDS,
// quad-swizzled for the ds_read_b128 of the B fragments) and prefetched a step ahead.  Steps h2 .. h2+R-2 revisit the
// first target rows for the displacement rows that wrap around (15 % more MFMA work at level 0, nothing else).
// one level of the ring build; NTL = 16-target tiles per target row (compile time: everything below stays in registers)
// NT = threads of the workgroup: 256 (4 waves = 4 source row pairs, every wave all NTL target tiles) or 512 (8 waves: the
// second four take the upper half of the target tiles -- twice the waves per CU on the same 80 KB of LDS)
// DUAL (round 6, option pyr_build_dual): a workgroup of 2 x NT threads builds TWO horizontally adjacent source blocks -- each half with
// its own ring, its own records and the wave roles above -- from ONE staged copy of every target row: the row fetches (16 KB per step,
// two thirds of what a step moves through the CU's vector-memory path) are shared by the two blocks.  `tid_all` = thread of the
// workgroup (staging role), `tid` = thread of the half (all other roles).
template <int NTL, int NT, bool DUAL = false>
__device__ __forceinline__ void ring_level(const half8 (&afrag)[4], const __half* __restrict__ trow0, __half* __restrict__ lbase,
                                           unsigned char* __restrict__ ring, unsigned char* __restrict__ sB,
                                           int l, int h2, int w2, int by, int bx, int tid_all, const __half* __restrict__ f1e = nullptr) {
  constexpr int NTS = DUAL ? 2 * NT : NT;                               // threads that stage a target row
  const int tid = DUAL ? (tid_all & (NT - 1)) : tid_all;
  const int lane = tid & 63, wave = (tid >> 6) & 3, wt = tid >> 8;      // wave = source row pair, wt = target half (NT = 512)
  constexpr int NTW = (NT == 512 && NTL >= 2) ? NTL / 2 : NTL;           // target tiles of this wave: wt * NTW .. + NTW
  const bool has_tiles = NT == 256 || NTL >= 2 || wt == 0;
  const int nt0 = (NT == 512 && NTL >= 2) ? wt * NTW : 0;
  const int j = lane & 15, yy = 2 * wave + (lane >> 5), xq = ((lane >> 4) & 1) * 4;     // pixel p_r = yy*8 + xq + r
  const int R = 8 >> l, ybase = (by * 8) >> l;
  const int rowbytes = w2 * 128;                                  // one displacement row: w2/2 pairs x 64 px x 4 B
  const int nsteps = h2 + R - 1;
  PYR_TS_DECL(l)
  const int g4 = lane >> 4, prow3 = 4 * (wave >> 1) + (j >> 2), psel = j & 3, par = wave & 1;
  const int d = ((by * 8 + (DH_PYR_V3 ? prow3 : yy)) >> l) - ybase;      // this lane's source row offset inside the block
  // byte offset inside a ring row of the value (tile nt, register r) of this lane -- the same at every step -- or -1
  // for the padding columns of a row shorter than 16 targets
  int soff[NTW][4];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int x2 = (nt0 + nt) * 16 + j, x1l = (bx * 8 + xq + r) >> l;
      const int u = wrap(x2 - x1l, w2), up = u >> 1, p = yy * 8 + xq + r;
      const int f = ring_f(up);
      soff[nt][r] = x2 < w2 ? ((up * 64 + (p ^ f)) << 2) + ((u & 1) << 1) : -1;
    }
  // DH_PYR_V3: this wave's 16 pixels = 4 source rows x the 4 columns x1 whose bit l equals the wave's parity (level 3: any split),
  // lane i = lane & 15 -> row 4 (wave >> 1) + (i >> 2); products arrive transposed: lane (g = lane >> 4, i) holds targets 4 g + r of pixel i
  const int x13 = l == 1 ? (psel & 1) + 4 * (psel >> 1) + 2 * par : l == 2 ? psel + 4 * par : 2 * psel + par;
  const int x1l3 = (bx * 8 + x13) >> l, sig = x1l3 & 1, p3 = prow3 * 8 + x13;
  half8 pfrag[4];
  int soffp[NTW][2], taddr[NTW][4];
  if (DH_PYR_V3) {
    const __half* prow_ptr = f1e + ((long)(by * 8 + prow3) * (w2 << l) + bx * 8 + x13) * CH;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) pfrag[ks] = *reinterpret_cast<const half8*>(prow_ptr + ks * 32 + g4 * 8);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int T = ((nt0 + nt) * 16 + j + sig) & (w2 - 1);          // the staged target this lane feeds into row i of the tile
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) taddr[nt][ks] = T * 256 + (((ks * 4 + g4) ^ (T & 15)) << 4);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int c = (nt0 + nt) * 16 + 4 * g4 + 2 * q;               // tile row of the pair's first target (before the parity shift)
        const int x2 = (c + sig) & (w2 - 1);
        const int u = wrap(x2 - x1l3, w2);                             // even by construction
        soffp[nt][q] = c < w2 ? (((u >> 1) * 64 + p3) << 2) : -1;
      }
    }
  }
  // B staging role: piece id = tid + 256*q -> (target id>>4, quad id&15), q < NTL.  Target rows are fetched THREE steps
  // ahead into registers.  (A __syncthreads() would drain vmcnt and with it the prefetch: the LDS traffic of a step only
  // needs lgkmcnt(0) + s_barrier.)
  constexpr int NQ = (NTL * 256 + NTS - 1) / NTS;      // 16-byte pieces of a target row per thread
  const bool stages = NTL * 256 >= NTS || tid_all < NTL * 256;
  u32x4 breg[3][NQ];      // (native vectors: arrays of HIP's uint4 struct assigned under a condition end up in scratch)
  int b_src[NQ], b_dst[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int id = tid_all + NTS * q, row = id >> 4, quad = id & 15;
    b_src[q] = min(row, w2 - 1) * CH + quad * 8;
    b_dst[q] = row * 256 + ((quad ^ (row & 15)) << 4);
  }
#define RING_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  // (DH_PYR_ASM & 1, the hand-counted waits: a row is fetched for EVERY k, also the three beyond the level's last step -- the row of
  // the last step again: every step then has exactly 2 x NQ loads issued behind the row it waits for)
  // the row of fetch k is (ybase + k) % h2; fetches are issued in the order of k, so DH_PYR_V2 keeps a pointer that wraps
  const __half* fetch_p = trow0 + (long)ybase * w2 * CH;
  const __half* const fetch_end = trow0 + (long)h2 * w2 * CH;
#define RING_FETCH_B(k_, s_)                                                                                         \
  if ((DH_PYR_ASM & 1) || (k_) < nsteps) {                                                                           \
    const __half* trow = (DH_PYR_V2 && !(DH_PYR_ASM & 1)) ? fetch_p : trow0 + (long)((ybase + min((k_), nsteps - 1)) % h2) * w2 * CH; \
    fetch_p += w2 * CH;                                                                                              \
    fetch_p = fetch_p == fetch_end ? trow0 : fetch_p;                                                                \
    if (stages) { _Pragma("unroll") for (int q = 0; q < NQ; ++q) gload16_async(breg[s_][q], trow + b_src[q]); }         \
  }
#define RING_STEP(k_, s_)                                                                                            \
  if ((k_) < nsteps) {                                                                                               \
    const int k = (k_);                                                                                              \
    PYR_TS(0)                                                                                                        \
    if (DH_PYR_ABL & 16) { _Pragma("unroll") for (int q = 0; q < NQ; ++q) asm volatile("" ::"v"(breg[s_][q])); }     \
    if (stages && !(DH_PYR_ABL & 16)) {                                                                              \
      _Pragma("unroll") for (int q = 0; q < NQ; ++q) wait_vm_for<2 * NQ>(breg[s_][q]);   /* the row fetched three steps ago */ \
      _Pragma("unroll") for (int q = 0; q < NQ; ++q) *reinterpret_cast<u32x4*>(sB + b_dst[q]) = breg[s_][q];           \
    }                                                                                                                \
    PYR_TS(1)                                                                                                        \
    RING_BARRIER();                         /* target row staged; last step's completed row has been read out */    \
    PYR_TS(2)                                                                                                        \
    RING_FETCH_B(k + 3, s_)                                                                                          \
    PYR_TS(3)                                                                                                        \
    const int v = k - d;                    /* displacement row this lane's pixels contribute to */                  \
    const bool vok = v >= 0 && v < h2;                                                                               \
    unsigned char* const rslot = ring + (v & (R - 1)) * rowbytes;                                                    \
    f32x4 acc[NTW];                                                                                                  \
    _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};                          \
    if (has_tiles && !(DH_PYR_ABL & 4)) {                                                                            \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {        /* the tiles' accumulation chains interleaved */       \
      half8 b[NTW];                                                                                                  \
      _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt)                                                             \
        b[nt] = *reinterpret_cast<const half8*>(sB + (DH_PYR_V3 ? taddr[nt][ks] : ((nt0 + nt) * 16 + j) * 256 + (((ks * 4 + (lane >> 4)) ^ j) << 4))); \
      _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt)                                                             \
        acc[nt] = DH_PYR_V3 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(b[nt], pfrag[ks], acc[nt], 0, 0, 0)             \
                            : __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag[ks], b[nt], acc[nt], 0, 0, 0);            \
    }                                                                                                                \
    }                                                                                                                \
    PYR_TS(4)                                                                                                        \
    if (DH_PYR_ABL & 2) { _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt) asm volatile("" ::"v"(acc[nt])); }      \
    if (vok && has_tiles && !(DH_PYR_ABL & 2)) {                                                                     \
      if (DH_PYR_V3) {                                                                                               \
        _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt)                                                           \
          _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                              \
            if (NTL > 1 || soffp[nt][q] >= 0)                                                                        \
              *reinterpret_cast<uint32_t*>(rslot + soffp[nt][q]) = pack_h2_scaled(acc[nt][2 * q], acc[nt][2 * q + 1]); \
      } else {                                                                                                       \
      _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt)                                                             \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                                \
          if (NTL > 1 || soff[nt][r] >= 0)                                                                           \
            *reinterpret_cast<__half*>(rslot + soff[nt][r]) = __float2half(acc[nt][r] * 0.0625f);                    \
      }                                                                                                              \
    }                                                                                                                \
    PYR_TS(5)                                                                                                        \
    RING_BARRIER();                         /* scatter complete */                                                  \
    PYR_TS(6)                                                                                                        \
    const int vdone = k - (R - 1);                                                                                   \
    if (vdone >= 0 && vdone < h2) {                                                                                  \
      const unsigned char* src = ring + (vdone & (R - 1)) * rowbytes;                                                \
      __half* dst = lbase + (long)vdone * (rowbytes >> 1);                                                           \
      for (int n4 = tid; n4 < w2 * 8; n4 += NT) {                                                                    \
        const u32x4 o = ring_read_piece(src, n4 >> 4, (n4 & 15) << 2);                                               \
        gstore16_async(dst + (long)n4 * 8, o);                                                                       \
      }                                                                                                              \
    }                                                                                                                \
    PYR_TS(7)                                                                                                        \
    PYR_TS_FLUSH(k)                                                                                                  \
  }
  // DH_PYR_PIPE (round 6): the scatter of step t - 1 is issued BETWEEN the MFMAs of step t (software pipelining inside the wave: the
  // products of a step wait one iteration in 4 * NTW registers).  A workgroup's waves run the phases of a step in lockstep between
  // its two barriers, so what a step costs is the SUM of its phases (B reads + MFMA chains ~ 600 cycles, conversion + scatter ~ 450,
  // read-out, staging: profiles/r06_v_pyr_build_pmc.txt); here the 8 conversions / 8 address adds / 8 ds_write_b16 fill the issue
  // slots the two dependent MFMA chains leave empty.  The scatter is unconditional per lane: a lane whose displacement row
  // v = t - 1 - d lies outside [0, h2) writes into the slot of row v +- R at ITS OWN cells -- before that row's real values arrive
  // (v < 0) or after the row has left (v >= h2) -- so no exec-mask branch splits the block.  Iteration t = nsteps only drains.
  // Rows complete one iteration later than in RING_STEP; slot reuse keeps its one-barrier distance (row t - 1 takes the slot of
  // row t - 1 - R, read out in iteration t - 1).  Same MFMAs, same conversion: same records.
#define RING_ITER(t_, s_)                                                                                            \
  if ((t_) <= nsteps) {                                                                                              \
    const int t = (t_);                                                                                              \
    if (stages) {                                                                                                    \
      _Pragma("unroll") for (int q = 0; q < NQ; ++q) wait_vm_for<2 * NQ>(breg[s_][q]);                               \
      _Pragma("unroll") for (int q = 0; q < NQ; ++q) *reinterpret_cast<u32x4*>(sB + b_dst[q]) = breg[s_][q];           \
    }                                                                                                                \
    RING_BARRIER();                         /* target row t staged; the row completed last iteration has been read out */ \
    RING_FETCH_B(t + 3, s_)                                                                                          \
    unsigned char* const rslot = ring + ((t - 1 - d) & (R - 1)) * rowbytes;                                          \
    f32x4 acc[NTW];                                                                                                  \
    _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};                          \
    if (has_tiles) {                                                                                                 \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                             \
        half8 b[NTW];                                                                                                \
        _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt)                                                           \
          b[nt] = *reinterpret_cast<const half8*>(sB + ((nt0 + nt) * 16 + j) * 256 + (((ks * 4 + (lane >> 4)) ^ j) << 4)); \
        _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt)                                                           \
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag[ks], b[nt], acc[nt], 0, 0, 0);                      \
        _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt)        /* register ks of last step's products */          \
          if (NTL > 1 || soff[nt][ks] >= 0)                                                                          \
            *reinterpret_cast<__half*>(rslot + soff[nt][ks]) = __float2half(prev[nt][ks] * 0.0625f);                 \
      }                                                                                                              \
    }                                                                                                                \
    _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt) prev[nt] = acc[nt];                                           \
    RING_BARRIER();                         /* scatter of step t - 1 complete */                                    \
    const int vdone = t - R;                                                                                         \
    if (vdone >= 0 && vdone < h2) {                                                                                  \
      const unsigned char* src = ring + (vdone & (R - 1)) * rowbytes;                                                \
      __half* dst = lbase + (long)vdone * (rowbytes >> 1);                                                           \
      for (int n4 = tid; n4 < w2 * 8; n4 += NT) {                                                                    \
        const u32x4 o = ring_read_piece(src, n4 >> 4, (n4 & 15) << 2);                                               \
        gstore16_async(dst + (long)n4 * 8, o);                                                                       \
      }                                                                                                              \
    }                                                                                                                \
  }
  RING_FETCH_B(0, 0) RING_FETCH_B(1, 1) RING_FETCH_B(2, 2)
  if (DH_PYR_PIPE && !DH_PYR_ABL && !(DH_PYR_ASM & 1)) {
    f32x4 prev[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) prev[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t3 = 0; t3 <= nsteps; t3 += 3) {
      RING_ITER(t3, 0)
      RING_ITER(t3 + 1, 1)
      RING_ITER(t3 + 2, 2)
    }
  } else {
    for (int k3 = 0; k3 < nsteps; k3 += 3) {
      RING_STEP(k3, 0)
      RING_STEP(k3 + 1, 1)
      RING_STEP(k3 + 2, 2)
    }
  }
#undef RING_ITER
#undef RING_STEP
#undef RING_FETCH_B
  // the three rows fetched past the level's end were never waited for: their data must have landed before the registers are anything
  // else (a wave that stages nothing at the next level would never wait again) -- one full drain per level
  if (DH_PYR_ASM & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RING_BARRIER();                           // the ring is reused by the next level
#undef RING_BARRIER
  // the all-zero row v = h2 of this level
  uint4* z = reinterpret_cast<uint4*>(lbase + (long)h2 * w2 * 64);
  for (int o = tid; o < w2 * 8; o += NT) z[o] = uint4{0u, 0u, 0u, 0u};
}

// ---- level 0 of a 64-column image, TILE-MAJOR wave roles (round 6) ----------------------------------------------------------
// ring_level gives a wave 16 source pixels (one row pair) and ALL (NT = 256) or half (NT = 512) of the target tiles of the staged row:
// every target tile's B fragments are read from LDS by four waves, 64 KB of the 96 KB of LDS traffic of a level-0 step
// (profiles/r05 layout notes).  Here a wave owns ONE 16-target tile and holds the A fragments of RB = 1024 / NT row pairs
// (NT = 512: two waves per tile, 32 source pixels each, 32 A registers; NT = 256: one wave per tile with all 64 source pixels, 64 A
// registers): a tile's B fragments are read by 2 (1) waves -- 32 (16) KB per step instead of 64.  Same MFMA instruction, same k
// order, same fp32 -> fp16 rounding: the records are bit-identical to ring_level's (tests: test_native_pyramid_build_kernels_are_bit_identical).
// Ring, staging, read-out: as in ring_level.
template <int NT>
__device__ __forceinline__ void ring_level0_tm(const __half* __restrict__ f1rows, const __half* __restrict__ trow0, __half* __restrict__ lbase,
                                               unsigned char* __restrict__ ring, unsigned char* __restrict__ sB,
                                               int h2, int by, int bx, int tid) {
  constexpr int W = 64, w2 = 64, NTL = 4, R = 8;
  constexpr int RB = 1024 / NT;                         // row pairs (16 source pixels each) per wave: 4 row pairs over NT / 256 waves per tile
  const int lane = tid & 63, wave = tid >> 6;
  const int tile = NT == 512 ? wave >> 1 : wave, rb0 = NT == 512 ? (wave & 1) * 2 : 0;
  const int j = lane & 15, xq = ((lane >> 4) & 1) * 4;
  const int ybase = by * 8;
  constexpr int rowbytes = w2 * 128;
  const int nsteps = h2 + R - 1;
  // A fragments of the wave's RB row pairs: lane holds A[i = lane & 15][k = (lane >> 4) * 8 .. + 8] of pixel p = rb * 16 + i
  half8 afrag[RB][4];
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    const int p = (rb0 + b) * 16 + (lane & 15);
    const __half* row = f1rows + ((long)(by * 8 + (p >> 3)) * W + bx * 8 + (p & 7)) * CH;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) afrag[b][ks] = *reinterpret_cast<const half8*>(row + ks * 32 + (lane >> 4) * 8);
  }
  // ring-row byte offset of (row pair 0, register r) of this lane; row pair rb: ^ (rb << 6) (p = rb * 16 + ..: bits 4, 5 of p, no carry)
  int soff0[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int x2 = tile * 16 + j, x1l = bx * 8 + xq + r;
    const int u = wrap(x2 - x1l, w2), up = u >> 1, p = (lane >> 5) * 8 + xq + r;
    const int f = ring_f(up);
    soff0[r] = ((up * 64 + (p ^ f)) << 2) + ((u & 1) << 1);
  }
  constexpr int NQ = (NTL * 256 + NT - 1) / NT;
  u32x4 breg[3][NQ];
  // staging piece q of this thread: target row (tid >> 4) + q * (NT / 16), quad tid & 15 -- NT / 16 is a multiple of 16, so the source and
  // LDS offsets of piece q are those of piece 0 plus a constant (no per-piece address registers: the kernel sits at its register limit)
  const int b_src0 = (tid >> 4) * CH + (tid & 15) * 8;
  const int b_dst0 = (tid >> 4) * 256 + (((tid & 15) ^ ((tid >> 4) & 15)) << 4);
  constexpr int B_SRC_Q = (NT >> 4) * CH, B_DST_Q = (NT >> 4) * 256;
#define RING_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define RING_FETCH_B(k_, s_)                                                                                         \
  if ((DH_PYR_ASM & 1) || (k_) < nsteps) {                                                                           \
    const __half* trow = trow0 + (long)((ybase + min((k_), nsteps - 1)) % h2) * w2 * CH;                             \
    _Pragma("unroll") for (int q = 0; q < NQ; ++q) gload16_async(breg[s_][q], trow + b_src0 + q * B_SRC_Q);          \
  }
#define RING_STEP(k_, s_)                                                                                            \
  if ((k_) < nsteps) {                                                                                               \
    const int k = (k_);                                                                                              \
    _Pragma("unroll") for (int q = 0; q < NQ; ++q) wait_vm_for<2 * NQ>(breg[s_][q]);                                 \
    _Pragma("unroll") for (int q = 0; q < NQ; ++q) *reinterpret_cast<u32x4*>(sB + b_dst0 + q * B_DST_Q) = breg[s_][q];           \
    RING_BARRIER();                                                                                                  \
    RING_FETCH_B(k + 3, s_)                                                                                          \
    f32x4 acc[RB];                                                                                                   \
    _Pragma("unroll") for (int b = 0; b < RB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};                               \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                               \
      const half8 bf = *reinterpret_cast<const half8*>(sB + (tile * 16 + j) * 256 + (((ks * 4 + (lane >> 4)) ^ j) << 4)); \
      _Pragma("unroll") for (int b = 0; b < RB; ++b)                                                                 \
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag[b][ks], bf, acc[b], 0, 0, 0);                          \
    }                                                                                                                \
    _Pragma("unroll") for (int b = 0; b < RB; ++b) {                                                                 \
      const int v = k - (2 * (rb0 + b) + (lane >> 5));      /* displacement row of this lane's pixels of row pair b */ \
      if (v >= 0 && v < h2) {                                                                                        \
        unsigned char* const rslot = ring + (v & (R - 1)) * rowbytes;                                                \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                                \
          *reinterpret_cast<__half*>(rslot + (soff0[r] ^ ((rb0 + b) << 6))) = __float2half(acc[b][r] * 0.0625f);    \
      }                                                                                                              \
    }                                                                                                                \
    RING_BARRIER();                                                                                                  \
    const int vdone = k - (R - 1);                                                                                   \
    if (vdone >= 0 && vdone < h2) {                                                                                  \
      const unsigned char* src = ring + (vdone & (R - 1)) * rowbytes;                                                \
      __half* dst = lbase + (long)vdone * (rowbytes >> 1);                                                           \
      for (int n4 = tid; n4 < w2 * 8; n4 += NT) {                                                                    \
        const u32x4 o = ring_read_piece(src, n4 >> 4, (n4 & 15) << 2);                                               \
        gstore16_async(dst + (long)n4 * 8, o);                                                                       \
      }                                                                                                              \
    }                                                                                                                \
  }
  RING_FETCH_B(0, 0) RING_FETCH_B(1, 1) RING_FETCH_B(2, 2)
  for (int k3 = 0; k3 < nsteps; k3 += 3) {
    RING_STEP(k3, 0)
    RING_STEP(k3 + 1, 1)
    RING_STEP(k3 + 2, 2)
  }
#undef RING_STEP
#undef RING_FETCH_B
  if (DH_PYR_ASM & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the rows fetched past the level's end: see ring_level)
  RING_BARRIER();
#undef RING_BARRIER
  uint4* z = reinterpret_cast<uint4*>(lbase + (long)h2 * w2 * 64);
  for (int o = tid; o < w2 * 8; o += NT) z[o] = uint4{0u, 0u, 0u, 0u};
}

// idx1 / idx2 (round 5, optional): edge e reads the prepared rows of FRAME idx1[e] / idx2[e] (dh_corr_pyramid_prepare_frames +
// dh_corr_pyramid_build_indexed: features transposed and pooled once per frame instead of once per edge); nullptr = row e
template <int W, int NT = 256, bool TM = false, bool DUAL = false>
__global__ __launch_bounds__(DUAL ? 2 * NT : NT, TM ? (NT == 512 ? 4 : 2) : 1) void pyr_build_ring_kernel(const __half* __restrict__ f1T, const __half* __restrict__ f2T,
                                                             __half* __restrict__ pyr, PyrDims D, long f1_stride_e,
                                                             long f2_stride_e, const int64_t* __restrict__ idx1 = nullptr,
                                                             const int64_t* __restrict__ idx2 = nullptr, int xcd_edges = 0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_ring_raw[];
  const int half_id = DUAL ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x / NT)) : 0;      // DUAL: which of the workgroup's two source blocks (wave-uniform)
  unsigned char* const ring = s_ring_raw + (size_t)half_id * W * 1024;      // [R][w2 * 32 dwords] (DUAL: one ring per half)
  unsigned char* const sB = s_ring_raw + (size_t)(DUAL ? 2 : 1) * W * 1024; // [target][256 B], 16-byte quads XOR-ed with row & 15
  // Workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest).  In the plain order the source blocks of ONE edge
  // land on all eight XCDs and every L2 fetches that edge's target rows; with xcd_edges = 8 * (E / 8) > 0 (option pyr_build_xcd) the
  // workgroups an XCD receives walk the source blocks of one edge after the other, so an edge's target rows (1 MB with the pooled
  // levels) are read from HBM once and re-read by its other source blocks from ONE L2.  Edges >= xcd_edges keep the plain order.
  int e = blockIdx.y, sb = blockIdx.x;
  if (e < xcd_edges) {
    const int n = e * (int)gridDim.x + sb, x = n & 7, loc = n >> 3;
    e = (loc / (int)gridDim.x) * 8 + x;
    sb = loc % (int)gridDim.x;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3;
  constexpr int nbx = W / 8;
  if (DUAL) sb = (sb / (nbx / 2)) * nbx + (sb % (nbx / 2)) * 2 + half_id;      // blockIdx.x numbers PAIRS of blocks (bx, bx + 1)
  const int by = sb / nbx, bx = sb - by * nbx;
  // A fragments of this wave's 16 source pixels (rows yy = 2*wave, 2*wave+1): lane holds A[i = lane&15][k = (lane>>4)*8..+8]
  const __half* bbase = f2T + (idx2 ? (long)idx2[e] : (long)e) * f2_stride_e;
  __half* obase = pyr + (long)e * D.edge_elems;
  const int h = D.h;
  if constexpr (TM && W == 64)          // (before the row-pair fragments below are loaded: its own A fragments take 32 / 64 registers)
    ring_level0_tm<NT>(f1T + (idx1 ? (long)idx1[e] : (long)e) * f1_stride_e, bbase + (long)D.tgt_off[0] * CH,
                       obase + D.lev_off[0] + (long)sb * D.blk_elems[0], ring, sB, h, by, bx, tid);
  const __half* const f1e = f1T + (idx1 ? (long)idx1[e] : (long)e) * f1_stride_e;
  half8 afrag[4];
  {
    const int p = wave * 16 + (lane & 15);
    const __half* row = f1e + ((long)(by * 8 + (p >> 3)) * W + bx * 8 + (p & 7)) * CH;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) afrag[ks] = *reinterpret_cast<const half8*>(row + ks * 32 + (lane >> 4) * 8);
  }
#define RING_LEVEL(l_)                                                                                               \
  {                                                                                                                  \
    constexpr int w2 = W >> (l_);                                                                                    \
    constexpr int NTL = w2 >= 16 ? w2 / 16 : 1;                                                                      \
    ring_level<NTL, NT, DUAL>(afrag, bbase + (long)D.tgt_off[l_] * CH, obase + D.lev_off[l_] + (long)sb * D.blk_elems[l_], \
                    ring, sB, l_, h >> (l_), w2, by, bx, tid, f1e);                                                  \
  }
  if constexpr (!(TM && W == 64)) {
    RING_LEVEL(0)
  }
  RING_LEVEL(1) RING_LEVEL(2) RING_LEVEL(3)
#undef RING_LEVEL
}

// ---------------------------------------------------------------------------------------- lookup
// workgroup = (edge, 8-row strip), wave = 8x8 source block, lane p = yy*8 + xx.
//   * a window row (8 taps from displacement cell u0) is FIVE aligned dword loads (cell pairs floor(u0/2)..+4,
//     modulo the row); the 8 taps are extracted with v_alignbit by the lane's parity of u0.  Loads are
//     UNCONDITIONAL (wave-uniform SGPR base + 32-bit lane offset): out-of-image window rows point at the
//     block's all-zero row, out-of-image columns are zeroed afterwards by AND-ing the packed pairs.
//     (A predicated load costs a branch + s_waitcnt vmcnt(0) each and leaves ONE request in flight per wave:
//     that version ran at 1.7 TB/s.)
//   * rolling prefetch in half-level batches (4 window rows = 20 loads = 20 VGPRs): while one half is
//     interpolated the other half and/or the next level's first half are in flight, also across the staging
//     barriers and the output stores.  The loads are inline asm so that they can stay outstanding across the
//     barriers (hipcc would drain vmcnt at a __syncthreads()) and the waits are explicit counted
//     s_waitcnt vmcnt(N): vector-memory operations return in order on this ISA family, so "at most N
//     outstanding" with N younger operations issued means the older batch has landed.
//   * staging tile in LDS: [49 planes][8 rows][w] fp16, the 16-byte segments of a row XOR-swizzled by the
//     row so that the per-wave column of segments spreads over the banks; x-neighbour lanes exchange packed
//     channel pairs (DPP quad_perm + v_perm) so every lane writes one dword per channel pair;
//   * the strip of one channel is 8 full rows = 1 KB contiguous in [E,196,h,w]: one 16-byte store per lane,
//     wave k of the workgroup stores channels k, k+8, ...
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ uint32_t dpp_swap_x(uint32_t v) {      // value of lane ^ 1
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}

constexpr int HALF_ROWS = WIN / 2;                 // window rows per batch
constexpr int NPAIR = WIN / 2 + 1;                 // dword loads per window row
constexpr int BATCH = HALF_ROWS * NPAIR;           // loads per batch (20)
constexpr int MIN_STORES = (NCH_OUT * 8) / 64;     // output-store instructions every wave issues per level (>= 6)

struct LevelGeom {                                  // per-lane addressing / weights of one level
  const __half* base;                               // wave-uniform: source block 0 of this level (this edge)
  int sboff;                                        // wave-uniform byte offset of this wave's source block
  int coloff[NPAIR];                                // byte offsets of the 5 cell pairs (lane included)
  uint32_t cmask[WIN / 2];                          // column validity of the aligned tap pairs (0xffff per tap)
  int Y0, y1l, par16;                               // first window row, own row at this level, 16 * (u0 & 1)
  int v0;                                           // fused kernel: displacement row of window row 0, folded once into [0, 2 h2)
  float dx, dy;
};
struct HalfTaps { uint32_t raw[HALF_ROWS][NPAIR]; };
struct HalfTapsQ { u32x4 q[NPAIR]; };            // MODE 4 (timing ablation): one 16-byte load per cell pair and window-row group

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NHWC = false: out [E,196,h,w] (the reference's tensor, channel = level*49 + xoff*7 + yoff), staged through LDS.
// NHWC = true : out [4,E,h,w,56] for the update operator of this library (csrc/conv.hip takes the 4 levels as 4
//               channel segments): channel = yoff*7 + xoff, channels 49..55 zero.  A lane owns all channels of
//               its pixel, so they are packed in registers, transposed through a wave-private LDS tile and
//               stored as 1 KB runs: no workgroup barrier at all.
// NHWC workgroups are single waves (nothing is shared between the waves of a strip in that mode), which lets the
// register allocator use 168 VGPRs at 3 waves/SIMD: the inline-asm tap loads must never be spilled or copied
// while they are in flight (scripts/audit_asm_loads.py checks the generated ISA for exactly that at build time).
typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
// MODE (reference-layout output only; opts().lookup_mode): 0 = product; 1 = tap loads with the `nt` cache policy; ablations that
// give WRONG results and exist to attribute the kernel's time (scripts/bench_lookup.py --modes): 2 = no output stores,
// 3 = no tap loads, 4 = the tap loads as 16-byte "quad" loads (5 instead of 20 per half level, no transpose).
template <int W, bool NHWC, int MODE = 0>
__global__ __launch_bounds__(NHWC ? 64 : W * 8, NHWC ? 3 : 4) void pyr_lookup_kernel(const __half* __restrict__ pyr, const float* __restrict__ coords,
                                                               __half* __restrict__ out, PyrDims D) {
  extern __shared__ __half s_out[];               // [49][8 rows][W], swizzled
  constexpr int NBX = W / 8, NTHREADS = W * 8;
  const int e = blockIdx.y, by = NHWC ? blockIdx.x / NBX : blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int bx = NHWC ? blockIdx.x % NBX : __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = D.h, HW = h * W;
  const int yy = lane >> 3, xx = lane & 7;
  const int y1 = by * 8 + yy, x1 = bx * 8 + xx;
  const int sb = by * NBX + bx;
  const float2 c0 = reinterpret_cast<const float2*>(coords)[(long)e * HW + (long)y1 * W + x1];
  const __half* ebase = pyr + (long)e * D.edge_elems;

  auto geom = [&](int l, LevelGeom& G) {
    const int w2 = W >> l;                                   // power of two
    const float inv = 1.0f / (float)(1 << l);
    const float cx = c0.x * inv, cy = c0.y * inv;            // exact: power-of-two scaling, as coords / 2**i
    float fxf = floorf(cx), fyf = floorf(cy);
    G.dx = cx - fxf; G.dy = cy - fyf;
    fxf = fminf(fmaxf(fxf, -65536.f), 65536.f);
    fyf = fminf(fmaxf(fyf, -65536.f), 65536.f);
    const int X0 = (int)fxf - RAD;
    G.Y0 = (int)fyf - RAD;
    G.y1l = y1 >> l;
    // Out-of-image window rows read zeros.  Every source block has its own all-zero row, but pointing all blocks at
    // block 0's makes those reads L2 hits instead of HBM lines nobody else will ever touch (8 % of the lookup's reads at
    // 48x64: two of the eight window rows at level 3, one to two at level 2): addresses are formed against block 0's base.
    G.base = ebase + D.lev_off[l];
    G.sboff = (int)((long)sb * D.blk_elems[l] * 2);
    const int u0 = (X0 - (x1 >> l)) & (w2 - 1);              // two's complement: correct for any X0
    G.par16 = (u0 & 1) * 16;
    const int k0 = u0 >> 1;
#pragma unroll
    for (int m = 0; m < NPAIR; ++m) {
      // even u0: the window is exactly pairs k0..k0+3; the 5th load re-reads pair k0+3 instead of fetching a
      // line nobody needs (coherent waves share the parity, so this is a real saving in HBM lines)
      const int mm = (m == NPAIR - 1) ? NPAIR - 2 + (u0 & 1) : m;
      G.coloff[m] = (((k0 + mm) & (w2 / 2 - 1)) * 64 + lane) * 4;
    }
#pragma unroll
    for (int n = 0; n < WIN / 2; ++n) {
      const uint32_t lo = (unsigned)(X0 + 2 * n) < (unsigned)w2 ? 0xffffu : 0u;
      const uint32_t hi = (unsigned)(X0 + 2 * n + 1) < (unsigned)w2 ? 0xffff0000u : 0u;
      G.cmask[n] = lo | hi;
    }
  };
  auto request = [&](int l, const LevelGeom& G, int half, HalfTaps& T) {
    const int w2 = W >> l, h2 = D.h2[l];
#pragma unroll
    for (int jj = 0; jj < HALF_ROWS; ++jj) {
      const int y2 = G.Y0 + half * HALF_ROWS + jj;
      const bool inside = (unsigned)y2 < (unsigned)h2;
      const int rowoff = inside ? wrap(y2 - G.y1l, h2) * (w2 * 128) + G.sboff : h2 * (w2 * 128);      // else block 0's all-zero row
#pragma unroll
      for (int m = 0; m < NPAIR; ++m) {
        if (MODE == 4) continue;
        if (MODE == 1) asm volatile("global_load_dword %0, %1, %2 nt" : "=v"(T.raw[jj][m]) : "v"(rowoff + G.coloff[m]), "s"(G.base));
        else if (MODE == 3) asm volatile("v_mov_b32 %0, %1" : "=v"(T.raw[jj][m]) : "v"(rowoff + G.coloff[m]));
        else asm volatile("global_load_dword %0, %1, %2" : "=v"(T.raw[jj][m]) : "v"(rowoff + G.coloff[m]), "s"(G.base));
      }
    }
  };
  // MODE 4, timing ablation of "quad loads" (WRONG results: no transpose): lane (quad q = lane >> 2, r = lane & 3) fetches
  // for window row r of the half the 16 bytes that hold the cell pair's dwords of its quad's four pixels -- the same lines as
  // the 20 dword loads, in 5 instructions
  auto requestQ = [&](int l, const LevelGeom& G, int half, HalfTapsQ& T) {
    const int w2 = W >> l, h2 = D.h2[l];
    const int r = lane & 3;
    const int y2 = G.Y0 + half * HALF_ROWS + r;
    const bool inside = (unsigned)y2 < (unsigned)h2;
    const int rowoff = (inside ? wrap(y2 - G.y1l, h2) * (w2 * 128) + G.sboff : h2 * (w2 * 128)) - 4 * r;
#pragma unroll
    for (int m = 0; m < NPAIR; ++m)
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(T.q[m]) : "v"(rowoff + G.coloff[m]), "s"(G.base));
  };
  auto landedQ = [&](HalfTapsQ& T) {
    asm volatile("" : "+v"(T.q[0]), "+v"(T.q[1]), "+v"(T.q[2]), "+v"(T.q[3]), "+v"(T.q[4]));
  };
  // ties every tap register to this point so that no use can be scheduled above the preceding wait
  auto landed = [&](HalfTaps& T) {
#pragma unroll
    for (int jj = 0; jj < HALF_ROWS; ++jj)
      asm volatile("" : "+v"(T.raw[jj][0]), "+v"(T.raw[jj][1]), "+v"(T.raw[jj][2]), "+v"(T.raw[jj][3]), "+v"(T.raw[jj][4]));
  };

  const bool odd = xx & 1;
  const uint32_t psel = odd ? 0x03020706u : 0x05040100u;        // v_perm_b32(P, W, psel): see put_pair
  const int seg = bx ^ (yy & (NBX - 1));
  __half* srow = s_out + yy * W + seg * 8 + (xx & ~1);           // + ch * 8 * W
  float prev[OUTW], stash = 0.f;
  constexpr int NHWC_LEVEL_CH = 56, NHWC_CH = NLEV * NHWC_LEVEL_CH;   // 49 + 7 zero channels per level
  uint32_t packed[NHWC_LEVEL_CH / 2];                                 // NHWC mode: this pixel's channels of the level
  // interpolate the 4 window rows of one batch; outputs of window row pair (j-1, j) go to LDS / the packed run
  auto consume = [&](const LevelGeom& G, const HalfTaps& T, int half) {
    const float dx = G.dx, dy = G.dy;
#pragma unroll
    for (int jj = 0; jj < HALF_ROWS; ++jj) {
      const int j = half * HALF_ROWS + jj;
      float t[WIN];
#pragma unroll
      for (int n = 0; n < WIN / 2; ++n) {
        const uint32_t pr = __builtin_amdgcn_alignbit(T.raw[jj][n + 1], T.raw[jj][n], G.par16) & G.cmask[n];
        t[2 * n] = (float)__builtin_bit_cast(_Float16, (unsigned short)(pr & 0xffffu));
        t[2 * n + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(pr >> 16));
      }
      float c[OUTW];
#pragma unroll
      for (int a = 0; a < OUTW; ++a) c[a] = t[a] + dx * (t[a + 1] - t[a]);
      if (j > 0) {
        float o[OUTW];
#pragma unroll
        for (int a = 0; a < OUTW; ++a) o[a] = prev[a] + dy * (c[a] - prev[a]);
        if (NHWC) {
          // channels (j-1)*7 + a, a = 0..6, of this level: append to the lane's packed fp16 run
#pragma unroll
          for (int a = 0; a < OUTW; ++a) {
            const int ch = (j - 1) * OUTW + a;
            const uint32_t hv = (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)o[a]);
            if (ch & 1) packed[ch >> 1] |= hv << 16; else packed[ch >> 1] = hv;
          }
        } else {
        // channel ch = a*7 + (j-1).  Two channels per step (A stored by even lanes, B by odd lanes):
        // Wd = [A_self, B_self] (fp16 pair), Pd = the x-neighbour's Wd; even lane stores [A_self, A_nb] = bytes
        // (Wd.lo, Pd.lo), odd lane [B_nb, B_self] = (Pd.hi, Wd.hi): one v_perm_b32 with a per-lane selector.
        auto put_pair = [&](float vA, float vB, int chA, int chB) {
          const uint32_t Wd = __builtin_bit_cast(uint32_t, __floats2half2_rn(vA, vB));
          const uint32_t Pd = dpp_swap_x(Wd);
          const uint32_t pk = __builtin_amdgcn_perm(Pd, Wd, psel);
          *reinterpret_cast<uint32_t*>(srow + (odd ? chB : chA) * 8 * W) = pk;
        };
        put_pair(o[0], o[1], 0 * OUTW + (j - 1), 1 * OUTW + (j - 1));
        put_pair(o[2], o[3], 2 * OUTW + (j - 1), 3 * OUTW + (j - 1));
        put_pair(o[4], o[5], 4 * OUTW + (j - 1), 5 * OUTW + (j - 1));
        if ((j - 1) & 1) put_pair(stash, o[6], 6 * OUTW + (j - 2), 6 * OUTW + (j - 1));
        else stash = o[6];
        if (j == WIN - 1) {                                  // last channel: both lanes of a pair write the same dword
          const uint32_t Wd = __builtin_bit_cast(uint32_t, __floats2half2_rn(o[6], o[6]));
          const uint32_t Pd = dpp_swap_x(Wd);
          *reinterpret_cast<uint32_t*>(srow + (6 * OUTW + 6) * 8 * W) = odd ? ((Pd & 0xffffu) | (Wd & 0xffff0000u)) : ((Wd & 0xffffu) | (Pd & 0xffff0000u));
        }
        }
      }
#pragma unroll
      for (int a = 0; a < OUTW; ++a) prev[a] = c[a];
    }
  };

  LevelGeom G, Gn;
  HalfTaps A, B;
  HalfTapsQ QA, QB;
  constexpr int NB_ = MODE == 4 ? NPAIR : BATCH;          // vector-memory operations per batch
  auto unpack = [&](const HalfTapsQ& Q, HalfTaps& T) {    // (MODE 4: no transpose, register moves only)
#pragma unroll
    for (int jj = 0; jj < HALF_ROWS; ++jj)
#pragma unroll
      for (int m = 0; m < NPAIR; ++m) T.raw[jj][m] = Q.q[m][jj];
  };
  geom(0, G);
  if (MODE == 4) { requestQ(0, G, 0, QA); requestQ(0, G, 1, QB); }
  else { request(0, G, 0, A); request(0, G, 1, B); }
#pragma unroll
  for (int l = 0; l < NLEV; ++l) {
    const bool more = l + 1 < NLEV;
    // issue order so far: ... A_l(NB_) B_l(NB_) [stores of level l-1 (>= MIN_STORES)]
    if (l == 0) wait_vm<NB_>(); else wait_vm<NB_ + MIN_STORES>();
    if (MODE == 4) { landedQ(QA); unpack(QA, A); } else landed(A);
    consume(G, A, 0);
    if (more) { geom(l + 1, Gn); if (MODE == 4) requestQ(l + 1, Gn, 0, QA); else request(l + 1, Gn, 0, A); }
    // younger than B_l: [stores of level l-1] [A_{l+1}(NB_)]
    if (l == 0) wait_vm<NB_>(); else if (more) wait_vm<NB_ + MIN_STORES>(); else wait_vm<0>();
    if (MODE == 4) { landedQ(QB); unpack(QB, B); } else landed(B);
    consume(G, B, 1);
    if (more) { if (MODE == 4) requestQ(l + 1, Gn, 1, QB); else request(l + 1, Gn, 1, B); }
    if (NHWC) {
      asm volatile("" ::: "memory");       // the stores below stay younger than the B_{l+1} requests (wait counts)
#pragma unroll
      for (int q = NCH_OUT / 2 + 1; q < NHWC_LEVEL_CH / 2; ++q) packed[q] = 0u;
      packed[NCH_OUT / 2] &= 0xffffu;      // channel 49 (upper half of the dword of channel 48) is padding
      // wave-private LDS image of the block's 8 pixel rows (8 px x 112 B = 896 B each): lane p writes its 112 B
      // at p*112 (conflict-free: 8-lane groups, stride 28 dwords), then the wave reads 16-byte pieces in memory
      // order, so one store instruction covers 1 KB of (at most two) contiguous runs instead of 64 lines.
      uint4* wtile = reinterpret_cast<uint4*>(s_out);
#pragma unroll
      for (int q = 0; q < NHWC_LEVEL_CH / 8; ++q)
        wtile[lane * (NHWC_LEVEL_CH / 8) + q] = uint4{packed[4 * q], packed[4 * q + 1], packed[4 * q + 2], packed[4 * q + 3]};
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      __half* lbase = out + (((long)l * gridDim.y + e) * h + by * 8) * (long)W * NHWC_LEVEL_CH + bx * 8 * NHWC_LEVEL_CH;
      constexpr int PPR = 8 * NHWC_LEVEL_CH / 8;                 // 16-byte pieces per block row (56)
#pragma unroll
      for (int it = 0; it < NHWC_LEVEL_CH / 8; ++it) {
        const int p = it * 64 + lane;
        const int row = p / PPR, off = p - row * PPR;
        const uint4 v = wtile[p];
        // streaming store: the 5.6 GB of lookup output are next read by another kernel, long after they have left every cache;
        // kept out of L2 they leave it to the pyramid lines that neighbouring blocks share (-2 % measured)
        __builtin_nontemporal_store(__builtin_bit_cast(u32x4_nt, v), reinterpret_cast<u32x4_nt*>(lbase + (long)row * W * NHWC_LEVEL_CH + off * 8));
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);                        // the tile is reused by the next level
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
    } else {
    lds_barrier();
      // one channel strip = 8 full rows = contiguous in the output; piece = 8 pixels (16 B).
      // piece o -> sg = o % NBX, row = (o / NBX) % 8, ch = o / (NBX * 8): all shifts (NBX is a power of two)
      constexpr int NPIECES = NCH_OUT * 8 * NBX;
      __half* obase = out + ((long)e * (NLEV * NCH_OUT) + l * NCH_OUT) * HW + (long)by * 8 * W;
  #pragma unroll
      for (int it = 0; it < (NPIECES + NTHREADS - 1) / NTHREADS; ++it) {
        const int o = tid + it * NTHREADS;
        if (it * NTHREADS + NTHREADS <= NPIECES || o < NPIECES) {
          const int sg = o % NBX, row = (o / NBX) % 8, ch = o / (NBX * 8);
          const uint4 val = *reinterpret_cast<const uint4*>(s_out + (ch * 8 + row) * W + (sg ^ (row & (NBX - 1))) * 8);
          if (MODE == 2) { if (val.x == 0x12345678u && val.y == 0x9abcdef0u) out[0] = __float2half(0.f); }    // keeps the LDS reads alive
          else __builtin_nontemporal_store(__builtin_bit_cast(u32x4_nt, val), reinterpret_cast<u32x4_nt*>(obase + (long)ch * HW + row * W + sg * 8));
        }
      }
      lds_barrier();
    }
    if (more) G = Gn;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Fused lookup + first correlation-encoder layer (droid_net.py:96-100: Conv2d(196, 128, 1) + ReLU on the lookup's output):
// the 196 window samples of a pixel never leave the CU.  A lane converts them to fp16 exactly as the unfused kernel stores
// them and writes them, eight channels (one 16-byte piece) at a time, into its wave's LDS tile in the B-operand layout of
// v_mfma_f32_32x32x16_f16 ([k-step][pixel tile][k half][32 pixels][8 channels]: the writes and the wave's own ds_read_b128
// fragment reads are contiguous); the weights ([13 k-steps][128 couts][16 channels] fp16, 52 KB) sit in LDS for the life of a
// PERSISTENT workgroup, and the MFMA accumulates D[128 couts][64 pixels] (128 accumulator registers per lane).  What goes to
// HBM is the 128-channel layer output (256 B per edge-pixel) instead of the 392 B of samples that the 1x1 layer would read
// back.  (The operands went through registers at first -- 24 packed dwords + v_permlane32_swap per level -- which spilled.)
// K order: level l, k-step s < 3 holds the level's channels kk = 16 s .. 16 s + 15 (kk = yoff * 7 + xoff, i.e. the order
// in which the interpolation produces them); the four channels kk = 48 share the 13th k-step.
constexpr int FK_STEPS = NLEV * 3 + 1;
constexpr int F_COUT = 128;
constexpr int FW_BYTES = FK_STEPS * F_COUT * 32;          // 53248
constexpr int FT_ROWB = F_COUT * 2 + 16;                  // transposition tile: 32 pixel rows of 128 couts + 16 B pad
constexpr int FT_BYTES = 32 * FT_ROWB;                    // 8704 per wave
constexpr int FX_BYTES = 2048;                            // B operands of the 13th k-step
constexpr int FC_BYTES = 1024;                            // coordinates of this block and of the next one (2 x [x 64][y 64] f32)
constexpr int FWAVE_BYTES = FT_BYTES + FX_BYTES + FC_BYTES;
constexpr int F_STORES = 2 * 8;                           // output store instructions per wave and block
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using half2f = __attribute__((ext_vector_type(2))) _Float16;

// t1 - t0 of two fp16 values that sit in halves H1 / H0 of packed registers, as ONE instruction: v_fma_mix_f32 takes fp16 operands
// (either half) straight into an fp32 FMA, t1 * 1.0 + (-t0) is exact -- the same value v_cvt_f32_f16 x 2 + v_sub_f32 produce.
// (hipcc folds conversions into v_fma_mix_f32 for the multiply-add of the interpolation by itself once its SLP vectoriser is off
// for this file -- build.py -- but not for a bare subtraction.)
template <int H1, int H0>
__device__ __forceinline__ float sub_f16_halves(uint32_t p1, uint32_t p0) {
  float d;
  if constexpr (H1 == 0 && H0 == 0) asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(p1), "v"(p0));
  else if constexpr (H1 == 1 && H0 == 0) asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(p1), "v"(p0));
  else if constexpr (H1 == 0 && H0 == 1) asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(p1), "v"(p0));
  else asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(p1), "v"(p0));
  return d;
}

// MIX (option lookup_mix, default 1): the interpolation without fp16 -> fp32 / fp32 -> fp16 conversion instructions (the taps enter the
// fp32 arithmetic as fp16 operands of v_fma_mix_f32, the samples leave it through v_fma_mixlo / mixhi_f16); 0 = the conversions
// spelled out.  Same operations in the same order: bit-identical (tests/test_gpu_parity.py).
// FILL (option lookup_fill, -DDH_ABLATION builds): how the 40 tap registers of a lane are refilled.  0 = two half-level batches of 4 window rows: a batch is
// requested again (for the next level) when all of its rows are interpolated, so during an interpolation phase only the other batch is
// in flight.  1 = by window row: row r of the next level is requested as soon as row r of this level is interpolated, 35 of the 40
// registers are in flight at every wait.  Same loads, same arithmetic in the same order: bit-identical.
template <int W, int MODE = 0, bool MIX = true, int FILL = 0>
__global__ __launch_bounds__(W * 8, 1) void pyr_lookup_corr0_kernel(const __half* __restrict__ pyr, const float* __restrict__ coords,
                                                                    const __half* __restrict__ wpk, const float* __restrict__ bias,
                                                                    __half* __restrict__ out, PyrDims D, int n_strips) {
  extern __shared__ __align__(16) unsigned char f_smem[];
  constexpr int NBX = W / 8, NTHREADS = W * 8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int bx = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = D.h, HW = h * W, NBY = h / 8;
  const int yy = lane >> 3, xx = lane & 7;
  // weights + bias -> LDS, once per workgroup
  for (int i = tid; i < FW_BYTES / 16; i += NTHREADS)
    reinterpret_cast<uint4*>(f_smem)[i] = reinterpret_cast<const uint4*>(wpk)[i];
  float* s_bias = reinterpret_cast<float*>(f_smem + FW_BYTES);
  if (tid < F_COUT) s_bias[tid] = bias[tid];
  unsigned char* tile = f_smem + FW_BYTES + F_COUT * 4 + bx * FWAVE_BYTES;
  for (int i = lane; i < FX_BYTES / 16; i += 64) reinterpret_cast<uint4*>(tile + FT_BYTES)[i] = uint4{0u, 0u, 0u, 0u};
  __syncthreads();
  const unsigned char* wlane = f_smem + (lane & 31) * 32 + (lane >> 5) * 16;       // + (ks * 4 + mt) * 1024

  int strip = blockIdx.x;
  if (strip >= n_strips) return;
  int e = strip / NBY, by = strip - e * NBY;
  // the lane's coordinates live in LDS (two slots: the next block's arrive by LDS-DMA while this block is in progress)
  float* cslot = reinterpret_cast<float*>(tile + FT_BYTES + FX_BYTES);
  int cur = 0;
  {
    const float2 v = reinterpret_cast<const float2*>(coords)[(long)e * HW + (long)(by * 8 + yy) * W + bx * 8 + xx];
    cslot[lane] = v.x; cslot[64 + lane] = v.y;
  }

  // part 1: what the tap requests need (row / column offsets); part 2: what the interpolation needs (weights, masks).  Kept
  // apart so that the next level's part 2 is not alive while this level is interpolated (registers: 2 waves / SIMD).
  auto geom = [&](int l, LevelGeom& G, int part) {
    const int w2 = W >> l;
    const int y1 = by * 8 + yy, x1 = bx * 8 + xx;
    const int sb = by * NBX + bx;
    const float inv = 1.0f / (float)(1 << l);
    const float cx = cslot[cur * 128 + lane] * inv, cy = cslot[cur * 128 + 64 + lane] * inv;
    float fxf = floorf(cx), fyf = floorf(cy);
    if (part == 2) { G.dx = cx - fxf; G.dy = cy - fyf; }
    fxf = fminf(fmaxf(fxf, -65536.f), 65536.f);
    fyf = fminf(fmaxf(fyf, -65536.f), 65536.f);
    const int X0 = (int)fxf - RAD;
    const int u0 = (X0 - (x1 >> l)) & (w2 - 1);
    if (part == 1) {
      G.Y0 = (int)fyf - RAD;
      G.y1l = y1 >> l;
      {
        // displacement row of window row r is (Y0 + r - y1l) mod h2.  Folded ONCE per level: for every window row inside the image
        // v0 + r lies in [0, 2 h2), so the row needs one add and one unsigned min (request()) instead of a two-sided wrap with
        // four compares / selects -- 32 window rows per block, and the kernel is bound by the instructions a wave issues
        const int a0 = G.Y0 - G.y1l;
        G.v0 = a0 + (a0 < 0 ? D.h2[l] : 0);
      }
      G.base = pyr + (long)e * D.edge_elems + D.lev_off[l];
      G.sboff = (int)((long)sb * D.blk_elems[l] * 2);
      const int k0 = u0 >> 1;
#pragma unroll
      for (int m = 0; m < NPAIR; ++m) {
        const int mm = (m == NPAIR - 1) ? NPAIR - 2 + (u0 & 1) : m;
        G.coloff[m] = (((k0 + mm) & (w2 / 2 - 1)) * 64 + lane) * 4;
      }
    } else {
      G.par16 = (u0 & 1) * 16;
#pragma unroll
      for (int n = 0; n < WIN / 2; ++n) {
        const uint32_t lo = (unsigned)(X0 + 2 * n) < (unsigned)w2 ? 0xffffu : 0u;
        const uint32_t hi = (unsigned)(X0 + 2 * n + 1) < (unsigned)w2 ? 0xffff0000u : 0u;
        G.cmask[n] = lo | hi;
      }
    }
  };
  auto request_row = [&](int l, const LevelGeom& G, int r, uint32_t (&raw)[NPAIR]) {
    const int w2 = W >> l, h2 = D.h2[l];
    const bool inside = (unsigned)(G.Y0 + r) < (unsigned)h2;
    const unsigned vr = (unsigned)(G.v0 + r);
    const unsigned v = min(vr, vr - (unsigned)h2);          // (v0 + r) mod h2 for rows inside the image (see geom); w2 * 128 is a power of two
    int own = (int)(v * (unsigned)(w2 * 128)) + G.sboff;
    asm("" : "+v"(own));               // computed for every lane: hipcc would otherwise sink it under an exec-mask branch per window row
    const int rowoff = inside ? own : h2 * (w2 * 128);      // else block 0's all-zero row
#pragma unroll
    for (int m = 0; m < NPAIR; ++m)
      if (MODE == 3) asm volatile("v_mov_b32 %0, %1" : "=v"(raw[m]) : "v"(rowoff + G.coloff[m]));
      else asm volatile("global_load_dword %0, %1, %2" : "=v"(raw[m]) : "v"(rowoff + G.coloff[m]), "s"(G.base));
    // MODE 6 = the synchronous twin of the product kernel: every request is waited for where it is issued, so no register is ever
    // in flight across compiler-scheduled code.  Same arithmetic in the same order -> the test suite requires bit-identical
    // output from MODE 0, which is how the product binary's explicit vmcnt bookkeeping is checked on real launches.
    if (MODE == 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto landed_row = [&](uint32_t (&raw)[NPAIR]) {
    asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]));
  };

  f32x16 acc[4][2];
  uint32_t piece[4] = {0u, 0u, 0u, 0u};  // the 8 channels being packed (one 16-byte piece of a B operand)
  float prev[OUTW];
  // B operands of the level in the wave's tile: [k-step s][pixel tile nt][kh][32 pixels][8 channels] fp16; lane = pixel
  unsigned char* fwr = tile + (lane >> 5) * 1024 + (lane & 31) * 16;              // + (s * 4 + kh) * 512
  const unsigned char* frd = tile + (lane >> 5) * 512 + (lane & 31) * 16;        // + (s * 4 + nt * 2) * 512
  // four MFMAs: k-step ks (12 = the shared 13th) on pixel tile nt.  (Issuing them between the rows of the following half
  // level's interpolation, to run the matrix pipe in the shadow of the wave's own vector instructions, was measured: k-step 0
  // placed that way changes nothing, 1.852 vs 1.854 ms; all three k-steps that way need 7 registers more than exist.)
  auto unit = [&](int ks, int nt) {
    const int off = ks < NLEV * 3 ? ((ks % 3) * 4 + nt * 2) * 512 : FT_BYTES + nt * 1024;
    const half8 bf = *reinterpret_cast<const half8*>(frd + off);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (MODE == 5 && mt > 0) continue;                 // timing ablation: a quarter of the MFMAs (wrong results)
      const half8 af = *reinterpret_cast<const half8*>(wlane + (ks * 4 + mt) * 1024);
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[mt][nt], 0, 0, 0);
    }
  };
  auto consume_row = [&](int l, const LevelGeom& G, const uint32_t (&raw)[NPAIR], int j) {
    const float dx = G.dx, dy = G.dy;
    if (MODE == 7) {                       // timing ablation (wrong results): the taps are touched, nothing is interpolated
      piece[j & 3] ^= raw[0] ^ raw[1] ^ raw[2] ^ raw[3] ^ raw[4];
      if (j == WIN - 1) *reinterpret_cast<u32x4*>(fwr) = u32x4{piece[0], piece[1], piece[2], piece[3]};
      return;
    }
    {
      float c[OUTW];
      if constexpr (MIX) {
        uint32_t pr[WIN / 2];
#pragma unroll
        for (int n = 0; n < WIN / 2; ++n) pr[n] = __builtin_amdgcn_alignbit(raw[n + 1], raw[n], G.par16) & G.cmask[n];
        // c[a] = t[a] + dx * (t[a + 1] - t[a]); tap a = half (a & 1) of pr[a >> 1]
#define DH_LERP_X(a_) { const float d_ = sub_f16_halves<((a_) + 1) & 1, (a_) & 1>(pr[((a_) + 1) >> 1], pr[(a_) >> 1]);                     \
                        c[a_] = __builtin_fmaf(dx, d_, (float)__builtin_bit_cast(half2f, pr[(a_) >> 1])[(a_) & 1]); }
        DH_LERP_X(0) DH_LERP_X(1) DH_LERP_X(2) DH_LERP_X(3) DH_LERP_X(4) DH_LERP_X(5) DH_LERP_X(6)
#undef DH_LERP_X
      } else {
        float t[WIN];
#pragma unroll
        for (int n = 0; n < WIN / 2; ++n) {
          const uint32_t pr = __builtin_amdgcn_alignbit(raw[n + 1], raw[n], G.par16) & G.cmask[n];
          t[2 * n] = (float)__builtin_bit_cast(_Float16, (unsigned short)(pr & 0xffffu));
          t[2 * n + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(pr >> 16));
        }
#pragma unroll
        for (int a = 0; a < OUTW; ++a) c[a] = t[a] + dx * (t[a + 1] - t[a]);
      }
      if (j > 0) {
#pragma unroll
        for (int a = 0; a < OUTW; ++a) {
          const float o = prev[a] + dy * (c[a] - prev[a]);
          const int ch = (j - 1) * OUTW + a;
          if (ch == NCH_OUT - 1) {       // 13th k-step, k = l
            *reinterpret_cast<_Float16*>(fwr + FT_BYTES + l * 2) = (_Float16)o;
          } else {
            const int d = (ch >> 1) & 3;
            if constexpr (MIX) {
              // prev + dy * (c - prev) rounded to fp16 straight into its half of the packed pair: v_fma_mixlo / mixhi_f16 write one
              // half of the destination and keep the other (hipcc picks mixlo into a temporary + v_or_b32_sdwa for the upper halves)
              const float dd = c[a] - prev[a];
              if (ch & 1) asm("v_fma_mixhi_f16 %0, %1, %2, %3" : "+v"(piece[d]) : "v"(dy), "v"(dd), "v"(prev[a]));
              else asm("v_fma_mixlo_f16 %0, %1, %2, %3" : "+v"(piece[d]) : "v"(dy), "v"(dd), "v"(prev[a]));
            } else {
              const uint32_t hv = (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)o);
              if (ch & 1) piece[d] |= hv << 16; else piece[d] = hv;
            }
            if ((ch & 7) == 7) {
              const int q = ch >> 3, s = q >> 1, kh = q & 1;
              *reinterpret_cast<u32x4*>(fwr + (s * 4 + kh) * 512) = u32x4{piece[0], piece[1], piece[2], piece[3]};
            }
          }
        }
      }
#pragma unroll
      for (int a = 0; a < OUTW; ++a) prev[a] = c[a];
    }
  };

  LevelGeom G, R;                        // G: interpolation part of the current level; R: request part of the level in flight
  uint32_t T[WIN][NPAIR];                // the taps of one level: window row r in T[r] (FILL 0: rows 0..3 and 4..7 are the two batches)
  geom(0, R, 1);
#pragma unroll
  for (int r = 0; r < WIN; ++r) request_row(0, R, r, T[r]);
  bool first = true;
  for (;;) {
    const int next = strip + gridDim.x;
    const bool has_next = next < n_strips;
    const int en = next / NBY, byn = next - en * NBY;
    // accumulators start from the bias
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(s_bias + mt * 32 + g * 8 + (lane >> 5) * 4);
        acc[mt][0][4 * g] = b4.x; acc[mt][0][4 * g + 1] = b4.y; acc[mt][0][4 * g + 2] = b4.z; acc[mt][0][4 * g + 3] = b4.w;
        acc[mt][1][4 * g] = b4.x; acc[mt][1][4 * g + 1] = b4.y; acc[mt][1][4 * g + 2] = b4.z; acc[mt][1][4 * g + 3] = b4.w;
      }
    const int e_cur = e, by_cur = by;
#pragma unroll
    for (int l = 0; l < NLEV; ++l) {
      const bool more = l + 1 < NLEV;
      if constexpr (FILL == 0) {
        // issue order: ... A_l B_l [16 stores of the previous block, l == 0 only] [next coords, after B_1's request]
        if (l == 0 && !first && MODE != 2) wait_vm<BATCH + F_STORES>(); else wait_vm<BATCH>();
#pragma unroll
        for (int r = 0; r < HALF_ROWS; ++r) landed_row(T[r]);
        geom(l, G, 2);
#pragma unroll
        for (int r = 0; r < HALF_ROWS; ++r) consume_row(l, G, T[r], r);       // channels 0..20
        if (more) geom(l + 1, R, 1);
        else if (has_next) { cur ^= 1; e = en; by = byn; geom(0, R, 1); }
        if (more || has_next) {
#pragma unroll
          for (int r = 0; r < HALF_ROWS; ++r) request_row(more ? l + 1 : 0, R, r, T[r]);
        }
        if (l == 0 && !first && MODE != 2) wait_vm<BATCH + F_STORES>(); else if (more || has_next) wait_vm<BATCH>(); else wait_vm<0>();
#pragma unroll
        for (int r = HALF_ROWS; r < WIN; ++r) landed_row(T[r]);
#pragma unroll
        for (int r = HALF_ROWS; r < WIN; ++r) consume_row(l, G, T[r], r);     // channels 21..48
        if (more || has_next) {                                              // (before the MFMAs: the batch has their time to land)
#pragma unroll
          for (int r = HALF_ROWS; r < WIN; ++r) request_row(more ? l + 1 : 0, R, r, T[r]);
        }
      } else {
        // issue order: ... rows 0..7 of level 0 [16 stores of the previous block] | (1,0) .. (1,7) [2 coordinate DMAs] | (2,r) | (3,r) |
        // (0',r): row r of the next level is issued right after row r of this level is interpolated, so the requests younger than
        // (l, r) are rows r+1..7 of level l and rows 0..r-1 of the next: 35, plus the 16 stores while level 0 of a later block is
        // waited for.  (The two coordinate DMAs sit between (1,7) and (2,0): waiting for 35 instead of 37 at level 1 is on the
        // safe side.)  Last block of the workgroup, last level: nothing follows, rows r+1..7 only.
        geom(l, G, 2);
        if (more) geom(l + 1, R, 1);
        else if (has_next) { cur ^= 1; e = en; by = byn; geom(0, R, 1); }
#pragma unroll
        for (int r = 0; r < WIN; ++r) {
          constexpr int YOUNGER = (WIN - 1) * NPAIR;
          if (more || has_next) { if (l == 0 && !first && MODE != 2) wait_vm<YOUNGER + F_STORES>(); else wait_vm<YOUNGER>(); }
          else {
            switch (r) {                   // (r is a constant after unrolling)
              case 0: wait_vm<7 * NPAIR>(); break; case 1: wait_vm<6 * NPAIR>(); break; case 2: wait_vm<5 * NPAIR>(); break;
              case 3: wait_vm<4 * NPAIR>(); break; case 4: wait_vm<3 * NPAIR>(); break; case 5: wait_vm<2 * NPAIR>(); break;
              case 6: wait_vm<NPAIR>(); break; default: wait_vm<0>(); break;
            }
          }
          landed_row(T[r]);
          consume_row(l, G, T[r], r);
          if (more || has_next) request_row(more ? l + 1 : 0, R, r, T[r]);
        }
      }
      if (MODE != 7) {
#pragma unroll
        for (int s = 0; s < 3; ++s) { unit(l * 3 + s, 0); unit(l * 3 + s, 1); }
      }
      if (l == 0 && has_next) {           // the next block's coordinates: LDS-DMA, no registers held across the levels
        const int voff = ((byn * 8 + yy) * W + bx * 8 + xx) * 8;
        const float* cbase = coords + (long)en * HW * 2;
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(cslot + (cur ^ 1) * 128));
        unsigned keep_;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\t"
                     "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %2, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep_) : "v"(voff), "v"(voff + 4), "s"(cbase), "s"(dst) : "memory");
        if (MODE == 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    // the 13th k-step (sample 48 of the four levels: k = level, the rest of that operand region stays zero)
    unit(NLEV * 3, 0); unit(NLEV * 3, 1);
    // ReLU, fp16, transposition through the wave's tile, 1 KB runs to the channel-last output
    asm volatile("" ::: "memory");
    __half* obase = out + (((long)e_cur * h + by_cur * 8) * W + bx * 8) * F_COUT;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x16& a = acc[mt][nt];
          // ReLU after the rounding (same result: rounding is monotonic and keeps the sign), two values per instruction
          const half2f z = {(_Float16)0.f, (_Float16)0.f};
          const half2f lo = __builtin_elementwise_max(half2f{(_Float16)a[4 * g], (_Float16)a[4 * g + 1]}, z);
          const half2f hi = __builtin_elementwise_max(half2f{(_Float16)a[4 * g + 2], (_Float16)a[4 * g + 3]}, z);
          *reinterpret_cast<uint2*>(tile + (lane & 31) * FT_ROWB + (mt * 32 + g * 8 + (lane >> 5) * 4) * 2) =
              uint2{__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi)};
        }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int p = it * 64 + lane, px = p >> 4, slot = p & 15;
        const uint4 v = *reinterpret_cast<const uint4*>(tile + px * FT_ROWB + slot * 16);
        const int row = nt * 4 + (px >> 3), col = px & 7;
        if (MODE == 2) { if (v.x == 0x12345678u && v.y == 0x9abcdef0u) out[0] = __float2half(0.f); }       // timing ablation: no output stores
        else __builtin_nontemporal_store(__builtin_bit_cast(u32x4_nt, v), reinterpret_cast<u32x4_nt*>(obase + ((long)row * W + col) * F_COUT + slot * 8));
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    }
    asm volatile("" ::: "memory");
    if (!has_next) break;
    strip = next;
    first = false;
  }
}

// the skewed chunks of the build kernel need 64 % w2_l == 0 and cell pairs need w2_l even on every level:
// w in {16,32,64}; h multiple of 8
bool dims_ok(int h, int w) { return h >= 8 && h % 8 == 0 && (w == 16 || w == 32 || w == 64); }

}  // namespace

extern "C" size_t dh_corr_pyramid_bytes(int E, int h, int w) {
  if (E < 0 || !dims_ok(h, w)) return 0;
  return (size_t)E * (size_t)make_dims(h, w).edge_elems * sizeof(__half);
}

extern "C" size_t dh_corr_pyramid_workspace_bytes(int E, int h, int w) {
  if (E < 0 || !dims_ok(h, w)) return 0;
  const PyrDims D = make_dims(h, w);
  return (size_t)E * ((size_t)h * w + (size_t)D.tgt_off[NLEV]) * CH * sizeof(__half) + 256;
}

extern "C" int dh_corr_pyramid_build(const void* fmap1, const void* fmap2, void* pyramid, void* workspace,
                                     size_t workspace_bytes, int E, int C, int h, int w, dh_stream_t stream) {
  return dh_corr_pyramid_build_canvas(fmap1, fmap2, pyramid, workspace, workspace_bytes, E, C, h, w, h, w, stream);
}

// An image of h_real x w_real (any size <= the canvas) on a canvas of h x w that satisfies the layout's constraints: the
// caller passes canvas-sized features, zero outside the image.  Level l of the image's pyramid has (h_real >> l) x (w_real >> l)
// cells (avg_pool2d floors, corr.py:36); the canvas levels carry zeros beyond them, which is what a lookup outside a
// reference-layout volume reads (correlation_kernels.cu:48 within_bounds).
namespace {
// the row-ring build kernel on prepared (channel-last, pooled) features; idx1 / idx2 = nullptr: row e of f1T / f2T
int launch_ring(const __half* f1T, const __half* f2T, __half* pyramid, const PyrDims& D, long s1, long s2, int E, int w,
                const int64_t* idx1, const int64_t* idx2, hipStream_t st) {
  const size_t lds = (size_t)w * 1024 + (size_t)(w > 16 ? w : 16) * 256;       // row ring + one staged target row
  const size_t lds_pad = (size_t)opts().pyr_lds_pad;        // (measurement only, DH_PYR_LDS_PAD: extra LDS bytes = fewer workgroups per CU)
  if (w == 64 && opts().pyr_build_dual && !opts().pyr_build_tm && opts().pyr_build_waves == 8) {
    const int xe2 = opts().pyr_build_xcd && (long)D.nblk * E < (1L << 30) ? (E / 8) * 8 : 0;
    DH_LDS_OPTIN((&pyr_build_ring_kernel<64, 512, false, true>), 160 * 1024);
    hipLaunchKernelGGL((pyr_build_ring_kernel<64, 512, false, true>), dim3(D.nblk / 2, E), dim3(1024), (size_t)(2 * 64 * 1024 + 64 * 256), st,
                       f1T, f2T, pyramid, D, s1, s2, idx1, idx2, xe2);
    return DH_OK;
  }
  const dim3 grid(D.nblk, E);
  const int xe = (opts().pyr_build_xcd && (D.nblk * (long)E) % 8 == 0 && (long)D.nblk * E < (1L << 30)) ? (E / 8) * 8 : 0;
  if (w == 64 && opts().pyr_build_tm && opts().pyr_build_waves == 8) {
    DH_LDS_OPTIN((&pyr_build_ring_kernel<64, 512, true>), 80 * 1024);
    hipLaunchKernelGGL((pyr_build_ring_kernel<64, 512, true>), grid, dim3(512), lds, st, f1T, f2T, pyramid, D, s1, s2, idx1, idx2, xe);
  } else if (w == 64 && opts().pyr_build_tm) {
    DH_LDS_OPTIN((&pyr_build_ring_kernel<64, 256, true>), 80 * 1024);
    hipLaunchKernelGGL((pyr_build_ring_kernel<64, 256, true>), grid, dim3(256), lds, st, f1T, f2T, pyramid, D, s1, s2, idx1, idx2, xe);
  } else if (w == 64 && opts().pyr_build_waves == 8) {
    DH_LDS_OPTIN((&pyr_build_ring_kernel<64, 512>), 160 * 1024);
    hipLaunchKernelGGL((pyr_build_ring_kernel<64, 512>), grid, dim3(512), lds + lds_pad, st, f1T, f2T, pyramid, D, s1, s2, idx1, idx2, xe);
  } else if (w == 64) {
    DH_LDS_OPTIN((&pyr_build_ring_kernel<64>), 80 * 1024);
    hipLaunchKernelGGL(pyr_build_ring_kernel<64>, grid, dim3(256), lds, st, f1T, f2T, pyramid, D, s1, s2, idx1, idx2, xe);
  } else if (w == 32) {
    hipLaunchKernelGGL(pyr_build_ring_kernel<32>, grid, dim3(256), lds, st, f1T, f2T, pyramid, D, s1, s2, idx1, idx2, xe);
  } else {
    hipLaunchKernelGGL(pyr_build_ring_kernel<16>, grid, dim3(256), lds, st, f1T, f2T, pyramid, D, s1, s2, idx1, idx2, xe);
  }
  return DH_OK;
}
}  // namespace

// Frame-level preparation (round 5): fmaps [F,C,h,w] f16 (canvas-sized, zero outside the h_real x w_real image) -> prepared
// [F][T = sum_l h2_l w2_l][C] rows: level 0 = the channel-last transpose, levels 1..3 = 2x2 mean pooling with avg_pool2d's floor
// (what dh_corr_pyramid_build_canvas does per EDGE: a frame appears in ~8 edges as source and ~8 as target).
extern "C" size_t dh_corr_pyramid_prepared_bytes(int F, int h, int w) {
  if (F < 0 || !dims_ok(h, w)) return 0;
  return (size_t)F * (size_t)make_dims(h, w).tgt_off[NLEV] * CH * sizeof(__half);
}
extern "C" int dh_corr_pyramid_prepare_frames(const void* fmaps, void* prepared, int F, int C, int h, int w, int h_real, int w_real,
                                              dh_stream_t stream) {
  if (h_real <= 0 || w_real <= 0 || h_real > h || w_real > w) return DH_ERR_ARG;
  if (F < 0 || C != CH || !dims_ok(h, w)) return C != CH && F >= 0 && dims_ok(h, w) ? DH_ERR_UNSUPPORTED : DH_ERR_ARG;
  if (F == 0) return DH_OK;
  if (!fmaps || !prepared) return DH_ERR_ARG;
  const PyrDims D = make_dims(h, w);
  hipStream_t st = (hipStream_t)stream;
  const int HW = h * w;
  const long s2 = (long)D.tgt_off[NLEV] * CH;
  hipLaunchKernelGGL(pyr_transpose_kernel, dim3((HW + 63) / 64, F), dim3(256), 0, st, (const __half*)fmaps, (__half*)prepared, HW, s2, 0);
  for (int l = 1; l < NLEV; ++l)
    hipLaunchKernelGGL(pyr_pool_kernel, dim3(D.h2[l] * D.w2[l], F), dim3(CH), 0, st, (__half*)prepared, s2, D.tgt_off[l - 1],
                       D.tgt_off[l], D.h2[l - 1], D.w2[l - 1], h_real >> l, w_real >> l);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
// pyramid of E edges from prepared frames: edge e = (source frame idx1[e], target frame idx2[e]) of `prepared` (F frames; the
// source side reads its level-0 rows).  Bit-identical to dh_corr_pyramid_build_canvas on the gathered per-edge features.
extern "C" int dh_corr_pyramid_build_indexed(const void* prepared, const int64_t* idx1, const int64_t* idx2, void* pyramid,
                                             int F, int E, int h, int w, dh_stream_t stream) {
  if (F < 0 || E < 0 || !dims_ok(h, w)) return DH_ERR_ARG;
  if (E == 0) return DH_OK;
  if (!prepared || !idx1 || !idx2 || !pyramid || F == 0) return DH_ERR_ARG;
  const PyrDims D = make_dims(h, w);
  const long s2 = (long)D.tgt_off[NLEV] * CH;
  const int rc = launch_ring((const __half*)prepared, (const __half*)prepared, (__half*)pyramid, D, s2, s2, E, w, idx1, idx2, (hipStream_t)stream);
  if (rc != DH_OK) return rc;
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_corr_pyramid_build_canvas(const void* fmap1, const void* fmap2, void* pyramid, void* workspace,
                                            size_t workspace_bytes, int E, int C, int h, int w, int h_real, int w_real, dh_stream_t stream) {
  if (h_real <= 0 || w_real <= 0 || h_real > h || w_real > w) return DH_ERR_ARG;
  if (E < 0 || C != CH || !dims_ok(h, w)) return C != CH && E >= 0 && dims_ok(h, w) ? DH_ERR_UNSUPPORTED : DH_ERR_ARG;
  if (E == 0) return DH_OK;
  if (!fmap1 || !fmap2 || !pyramid || !workspace) return DH_ERR_ARG;
  if (workspace_bytes < dh_corr_pyramid_workspace_bytes(E, h, w)) return DH_ERR_WORKSPACE;
  const PyrDims D = make_dims(h, w);
  hipStream_t st = (hipStream_t)stream;
  const int HW = h * w, T = D.tgt_off[NLEV];
  __half* f1T = (__half*)workspace;
  __half* f2T = f1T + (size_t)E * HW * CH;
  const long s1 = (long)HW * CH, s2 = (long)T * CH;
  hipLaunchKernelGGL(pyr_transpose_kernel, dim3((HW + 63) / 64, E), dim3(256), 0, st, (const __half*)fmap1, f1T, HW, s1, 0);
  hipLaunchKernelGGL(pyr_transpose_kernel, dim3((HW + 63) / 64, E), dim3(256), 0, st, (const __half*)fmap2, f2T, HW, s2, 0);
  for (int l = 1; l < NLEV; ++l)
    hipLaunchKernelGGL(pyr_pool_kernel, dim3(D.h2[l] * D.w2[l], E), dim3(CH), 0, st, f2T, s2, D.tgt_off[l - 1],
                       D.tgt_off[l], D.h2[l - 1], D.w2[l - 1], h_real >> l, w_real >> l);
#ifdef DH_ABLATION
  if (opts().pyr_build_chunk) {                            // the first form of the build kernel (A/B measurements)
    hipLaunchKernelGGL(pyr_build_kernel, dim3(D.nblk, E), dim3(256), 0, st, (const __half*)f1T, (const __half*)f2T,
                       (__half*)pyramid, D, s1, s2);
  } else
#endif
  {
    const int rc = launch_ring((const __half*)f1T, (const __half*)f2T, (__half*)pyramid, D, s1, s2, E, w, nullptr, nullptr, st);
    if (rc != DH_OK) return rc;
  }
  DH_LAUNCH_CHECK();
  return DH_OK;
}

namespace {
template <bool NHWC>
int launch_lookup(const void* pyramid, const float* coords, void* out, int E, int h, int w, hipStream_t st) {
  const PyrDims D = make_dims(h, w);
  const size_t lds = NHWC ? (size_t)64 * 56 * sizeof(__half) : (size_t)NCH_OUT * 8 * w * sizeof(__half);
  const dim3 grid(NHWC ? (h / 8) * (w / 8) : h / 8, E);
  const dim3 block(NHWC ? 64 : w * 8);
  const int mode = NHWC ? 0 : opts().lookup_mode;
  if (w == 64 && mode == 1)
    hipLaunchKernelGGL((pyr_lookup_kernel<64, NHWC, 1>), grid, block, lds, st, (const __half*)pyramid, coords, (__half*)out, D);
#ifdef DH_ABLATION   // timing ablations with WRONG results: only in -DDH_ABLATION builds (options.hip refuses the modes otherwise)
  else if (w == 64 && mode == 2)
    hipLaunchKernelGGL((pyr_lookup_kernel<64, NHWC, 2>), grid, block, lds, st, (const __half*)pyramid, coords, (__half*)out, D);
  else if (w == 64 && mode == 3)
    hipLaunchKernelGGL((pyr_lookup_kernel<64, NHWC, 3>), grid, block, lds, st, (const __half*)pyramid, coords, (__half*)out, D);
  else if (w == 64 && mode == 4)
    hipLaunchKernelGGL((pyr_lookup_kernel<64, NHWC, 4>), grid, block, lds, st, (const __half*)pyramid, coords, (__half*)out, D);
#endif
  else if (w == 64)
    hipLaunchKernelGGL((pyr_lookup_kernel<64, NHWC>), grid, block, lds, st, (const __half*)pyramid, coords, (__half*)out, D);
  else if (w == 32)
    hipLaunchKernelGGL((pyr_lookup_kernel<32, NHWC>), grid, block, lds, st, (const __half*)pyramid, coords, (__half*)out, D);
  else
    hipLaunchKernelGGL((pyr_lookup_kernel<16, NHWC>), grid, block, lds, st, (const __half*)pyramid, coords, (__half*)out, D);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
}  // namespace

extern "C" int dh_corr_pyramid_lookup(const void* pyramid, const float* coords, void* out,
                                      int E, int h, int w, dh_stream_t stream) {
  if (E < 0 || !dims_ok(h, w)) return DH_ERR_ARG;
  if (E == 0) return DH_OK;
  if (!pyramid || !coords || !out) return DH_ERR_ARG;
  return launch_lookup<false>(pyramid, coords, out, E, h, w, (hipStream_t)stream);
}

extern "C" int dh_corr_pyramid_lookup_nhwc(const void* pyramid, const float* coords, void* out,
                                           int E, int h, int w, dh_stream_t stream) {
  if (E < 0 || !dims_ok(h, w)) return DH_ERR_ARG;
  if (E == 0) return DH_OK;
  if (!pyramid || !coords || !out) return DH_ERR_ARG;
  return launch_lookup<true>(pyramid, coords, out, E, h, w, (hipStream_t)stream);
}

extern "C" int dh_corr_pyramid_lookup_corr0(const void* pyramid, const float* coords, const void* wpk, const float* bias, void* out,
                                            int E, int h, int w, dh_stream_t stream) {
  if (E < 0 || !dims_ok(h, w)) return DH_ERR_ARG;
  if (E == 0) return DH_OK;
  if (!pyramid || !coords || !wpk || !bias || !out) return DH_ERR_ARG;
  const PyrDims D = make_dims(h, w);
  const int n_strips = E * (h / 8);
  const size_t lds = (size_t)FW_BYTES + F_COUT * 4 + (size_t)(w / 8) * FWAVE_BYTES;
  static int cu_count[64];
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
    if (!cu_count[dev]) { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cu_count[dev] = n; }
    if (cu_count[dev]) cus = cu_count[dev];
  }
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(160 * 1024 / lds, (size_t)(16 / (w / 8))));   // LDS and 2 waves / SIMD
  const dim3 grid((unsigned)std::min<long>(n_strips, (long)cus * per_cu)), block(w * 8);
  hipStream_t st = (hipStream_t)stream;
  const int mode = opts().lookup_mode;                     // 2 / 3 / 5 / 7: timing ablations (wrong results); 6: synchronous twin (same results)
  const bool mix = opts().lookup_mix != 0;                  // 0: conversions spelled out (same results; A/B and the bit-identity test)
  const bool rows = opts().lookup_fill != 0;                // 1: tap registers refilled window row by window row (same results; -DDH_ABLATION builds)
#define DH_FUSED(M_, X_, F_)                                                                                                   \
    { DH_LDS_OPTIN((&pyr_lookup_corr0_kernel<64, M_, X_, F_>), 160 * 1024);                                                    \
      hipLaunchKernelGGL((pyr_lookup_corr0_kernel<64, M_, X_, F_>), grid, block, lds, st, (const __half*)pyramid, coords, (const __half*)wpk, bias, (__half*)out, D, n_strips); }
  if (w == 64 && mode == 6 && mix) DH_FUSED(6, true, 0)   // the synchronous twin: same results (tests compare bit for bit)
  else if (w == 64 && mode == 6) DH_FUSED(6, false, 0)
#ifdef DH_ABLATION   // timing ablations with WRONG results: only in -DDH_ABLATION builds
  else if (w == 64 && mode == 2) DH_FUSED(2, true, 0)
  else if (w == 64 && mode == 3) DH_FUSED(3, true, 0)
  else if (w == 64 && mode == 5) DH_FUSED(5, true, 0)
  else if (w == 64 && mode == 7 && rows) DH_FUSED(7, true, 1)
  else if (w == 64 && mode == 7) DH_FUSED(7, true, 0)
  else if (w == 64 && rows && mix) DH_FUSED(0, true, 1)    // measured: 1.874 vs 1.865 ms (profiles/r04_j_lookup_fill_ab.txt) -> A/B builds only
#endif
  else if (w == 64 && !mix) DH_FUSED(0, false, 0)
  else if (w == 64) DH_FUSED(0, true, 0)
#undef DH_FUSED
  else if (w == 32) {
    DH_LDS_OPTIN((&pyr_lookup_corr0_kernel<32>), 160 * 1024);
    hipLaunchKernelGGL((pyr_lookup_corr0_kernel<32>), grid, block, lds, st, (const __half*)pyramid, coords, (const __half*)wpk, bias, (__half*)out, D, n_strips);
  } else {
    DH_LDS_OPTIN((&pyr_lookup_corr0_kernel<16>), 160 * 1024);
    hipLaunchKernelGGL((pyr_lookup_corr0_kernel<16>), grid, block, lds, st, (const __half*)pyramid, coords, (const __half*)wpk, bias, (__half*)out, D, n_strips);
  }
  DH_LAUNCH_CHECK();
  return DH_OK;
}
