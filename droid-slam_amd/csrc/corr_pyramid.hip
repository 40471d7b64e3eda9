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
//     V'[e][level][sb = 8x8 source block][v][u][p]      (fp16, p = 6-bit index inside the source block)
//     v = (y2 - (y1 >> l)) mod h2_l ,  u = (x2 - (x1 >> l)) mod w2_l        ("displacement" coordinates)
//
// i.e. one 128-byte line holds, for ONE displacement (v,u), the 64 pixels of an 8x8 source block.  A
// wave = one source block; when the flow is spatially coherent (it is: it comes from a reprojection)
// all 64 lanes want the same displacement cells, so every tap load of the wave is one fully used line
// and the window rows of a wave are 8 consecutive lines.  Zero padding outside the image is applied from
// the un-wrapped (x2,y2) in registers.  Same size as the reference pyramid (no padding).
//
// Build: per level a true contraction over the 128 feature channels on the fp16 MFMA
// (v_mfma_f32_16x16x32_f16), V_l = f1^T * pool_l(f2) / 16 (pooling commutes with the contraction; the
// reference's alt path pools features the same way, corr.py:89-101).  One wave owns a source block (A
// fragments stay in registers for the whole kernel) and streams target chunks of 64 pixels; the 64x64
// result goes through a wave-private LDS tile that is written in skewed order and read back as 16-byte
// pieces, so HBM only sees 16-byte aligned runs (8 per lane per chunk).
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
    d.lev_off[l] = off; d.tgt_off[l] = t;
    off += (long)h * w * d.h2[l] * d.w2[l];
    t += d.h2[l] * d.w2[l];
  }
  d.tgt_off[NLEV] = t;
  d.edge_elems = off;
  return d;
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
__global__ __launch_bounds__(128) void pyr_pool_kernel(__half* __restrict__ f2T, long stride_e, int row_in, int row_out,
                                                       int h_in, int w_in) {
  const int e = blockIdx.y;
  const int q = blockIdx.x;                      // output pixel
  const int w_out = w_in >> 1;
  const int y = q / w_out, x = q - y * w_out;
  const int c = threadIdx.x;
  const __half* in = f2T + (long)e * stride_e + (long)row_in * CH;
  const long r00 = ((long)(2 * y) * w_in + 2 * x) * CH + c;
  const float s = __half2float(in[r00]) + __half2float(in[r00 + CH]) + __half2float(in[r00 + (long)w_in * CH]) +
                  __half2float(in[r00 + (long)w_in * CH + CH]);
  f2T[(long)e * stride_e + (long)(row_out + q) * CH + c] = __float2half(0.25f * s);
}

// ---------------------------------------------------------------------------------------- build
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ int wrap(int a, int n) { return a < 0 ? a + n : (a >= n ? a - n : a); }

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
    // stage[(cell_local) * 64 + p]: cell_local = (target row inside chunk)*w2 + u  -- but rows of the chunk
    // are addressed by (q - q0) with u replacing x2: idx = (q_local - x2 + u) = q_local + (u - x2)
    const int sx = bx * 8 >> l, sy = by * 8 >> l;      // block origin at this level (for the wrap arithmetic)
    (void)sy;
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
          // position inside the chunk keeps the target row, swaps x2 for u
          stage[(ql - x2 + u) * 64 + p] = __float2half(acc[t][nt][r] * 0.0625f);
        }
    }
    (void)sx;
    __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): wave-private tile complete
    __builtin_amdgcn_wave_barrier();
    // read back 16-byte pieces: piece = (cell_local, yy) = 8 source pixels of one source row
    // global cell: v = wrap(y2 - (y1 >> l)), same u
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int piece = it * 64 + lane;
      const int cl = piece >> 3, yy = piece & 7;
      if (cl < nq) {
        const int q = q0 + cl;
        const int y2 = q / w2, u = q - y2 * w2;
        const int v = wrap(y2 - ((by * 8 + yy) >> l), h2);
        const uint4 val = *reinterpret_cast<const uint4*>(stage + cl * 64 + yy * 8);
        __half* dst = obase + D.lev_off[l] + ((long)sb * T + (long)v * w2 + u) * 64 + yy * 8;
        *reinterpret_cast<uint4*>(dst) = val;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------- lookup
// workgroup = (edge, 8-row strip), wave = 8x8 source block, lane p = yy*8 + xx.
//   * tap loads are UNCONDITIONAL, from clamped in-slice offsets (wave-uniform SGPR base + 32-bit lane
//     offset); out-of-image taps are zeroed afterwards in registers.  (A predicated load costs a branch +
//     s_waitcnt vmcnt(0) each and leaves ONE request in flight per wave: measured 1.7 TB/s.)
//   * rolling prefetch in half-level batches (4 window rows = 32 loads = 32 VGPRs): while one half is
//     interpolated the other half and/or the next level's first half are in flight, also across the staging
//     barriers and the output stores.  The loads are inline asm so that they can stay outstanding across the
//     barriers (hipcc would drain vmcnt at a __syncthreads()) and the waits are explicit counted
//     s_waitcnt vmcnt(N): vector-memory operations return in order on this ISA family, so "at most N
//     outstanding" with N younger operations issued means the older batch has landed.  (d16 / d16_hi loads
//     cannot be used to pack two taps per VGPR: with SRAM-ECC they clear the other half.)
//   * staging tile in LDS: [49 planes][8 rows][w] fp16, the 16-byte segments of a row XOR-swizzled by the
//     row so that the per-wave column of segments spreads over the banks; x-neighbour lanes exchange values
//     (DPP quad_perm) so every lane writes one packed dword per channel pair;
//   * the strip of one channel is 8 full rows = 1 KB contiguous in [E,196,h,w]: one 16-byte store per lane.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float dpp_swap_x(float v) {      // value of lane ^ 1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

constexpr int HALF_ROWS = WIN / 2;                 // window rows per batch
constexpr int MIN_STORES = (NCH_OUT * 8) / 64;     // output-store instructions every wave issues per level (>= 6)

struct LevelGeom {                                  // per-lane addressing / weights of one level
  const __half* base;                               // wave-uniform
  int coloff[WIN];                                  // byte offsets (lane included)
  uint32_t cmask[WIN / 2];                          // 0xffff / 0 per window column, two columns per register
  int Y0, y1l, h2, w2;
  float dx, dy;
};
struct HalfTaps { uint32_t raw[HALF_ROWS][WIN]; uint32_t rows; };

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(512, 4) void pyr_lookup_kernel(const __half* __restrict__ pyr, const float* __restrict__ coords,
                                                             __half* __restrict__ out, PyrDims D) {
  extern __shared__ __half s_out[];               // [49][8 rows][w], swizzled
  const int e = blockIdx.y, by = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int bx = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = D.w, h = D.h, HW = h * w;
  const int nbx = w / 8, nthreads = nbx * 64;
  const int yy = lane >> 3, xx = lane & 7;
  const int y1 = by * 8 + yy, x1 = bx * 8 + xx;
  const int sb = by * nbx + bx;
  const float2 c0 = reinterpret_cast<const float2*>(coords)[(long)e * HW + (long)y1 * w + x1];
  const __half* ebase = pyr + (long)e * D.edge_elems;

  auto geom = [&](int l, LevelGeom& G) {
    const int h2 = D.h2[l], w2 = D.w2[l];
    const float inv = 1.0f / (float)(1 << l);
    const float cx = c0.x * inv, cy = c0.y * inv;           // exact: power-of-two scaling, as coords / 2**i
    float fxf = floorf(cx), fyf = floorf(cy);
    G.dx = cx - fxf; G.dy = cy - fyf;
    fxf = fminf(fmaxf(fxf, -65536.f), 65536.f);
    fyf = fminf(fmaxf(fyf, -65536.f), 65536.f);
    const int X0 = (int)fxf - RAD;
    G.Y0 = (int)fyf - RAD;
    const int x1l = x1 >> l;
    G.y1l = y1 >> l; G.h2 = h2; G.w2 = w2;
    G.base = ebase + D.lev_off[l] + (long)sb * h2 * w2 * 64;
#pragma unroll
    for (int i = 0; i < WIN; ++i) {
      const int x2 = X0 + i;
      const bool ok = (unsigned)x2 < (unsigned)w2;
      if ((i & 1) == 0) G.cmask[i >> 1] = ok ? 0xffffu : 0u; else G.cmask[i >> 1] |= ok ? 0xffff0000u : 0u;
      G.coloff[i] = ((ok ? wrap(x2 - x1l, w2) * 64 : 0) + lane) * 2;
    }
  };
  auto request = [&](const LevelGeom& G, int half, HalfTaps& T) {
    uint32_t rows = 0;
#pragma unroll
    for (int jj = 0; jj < HALF_ROWS; ++jj) {
      const int y2 = G.Y0 + half * HALF_ROWS + jj;
      const bool ok = (unsigned)y2 < (unsigned)G.h2;
      rows |= ok ? (1u << jj) : 0u;
      const int rowoff = (ok ? wrap(y2 - G.y1l, G.h2) * G.w2 * 64 : 0) * 2;
#pragma unroll
      for (int i = 0; i < WIN; ++i)
        asm volatile("global_load_ushort %0, %1, %2" : "=v"(T.raw[jj][i]) : "v"(rowoff + G.coloff[i]), "s"(G.base));
    }
    T.rows = rows;
  };
  // ties every tap register to this point so that no use can be scheduled above the preceding wait
  auto landed = [&](HalfTaps& T) {
#pragma unroll
    for (int jj = 0; jj < HALF_ROWS; ++jj)
      asm volatile("" : "+v"(T.raw[jj][0]), "+v"(T.raw[jj][1]), "+v"(T.raw[jj][2]), "+v"(T.raw[jj][3]),
                        "+v"(T.raw[jj][4]), "+v"(T.raw[jj][5]), "+v"(T.raw[jj][6]), "+v"(T.raw[jj][7]));
  };

  const bool odd = xx & 1;
  const int seg = bx ^ (yy & (nbx - 1));
  __half* srow = s_out + yy * w + seg * 8 + (xx & ~1);           // + ch * 8 * w
  float prev[OUTW], stash = 0.f;
  // interpolate the 4 window rows of one batch; outputs of window row pair (j-1, j) go to LDS
  auto consume = [&](const LevelGeom& G, const HalfTaps& T, int half) {
    const float dx = G.dx, dy = G.dy;
#pragma unroll
    for (int jj = 0; jj < HALF_ROWS; ++jj) {
      const int j = half * HALF_ROWS + jj;
      const bool rowok = (T.rows >> jj) & 1;
      float t[WIN];
#pragma unroll
      for (int i = 0; i < WIN; ++i)
        t[i] = (float)__builtin_bit_cast(_Float16, (unsigned short)(T.raw[jj][i] & ((i & 1) ? (G.cmask[i >> 1] >> 16) : G.cmask[i >> 1])));
      float c[OUTW];
#pragma unroll
      for (int a = 0; a < OUTW; ++a) { const float v = t[a] + dx * (t[a + 1] - t[a]); c[a] = rowok ? v : 0.f; }
      if (j > 0) {
        float o[OUTW];
#pragma unroll
        for (int a = 0; a < OUTW; ++a) o[a] = prev[a] + dy * (c[a] - prev[a]);
        // channel ch = a*7 + (j-1).  Pair two channels (A for even lanes, B for odd lanes): each lane sends the
        // partner the value of the partner's channel and writes (x even, x odd) of its own channel as one dword.
        auto put_pair = [&](float vA, float vB, int chA, int chB) {
          const float mine = odd ? vB : vA, give = odd ? vA : vB;
          const float got = dpp_swap_x(give);
          const __half2 pk = odd ? __floats2half2_rn(got, mine) : __floats2half2_rn(mine, got);
          *reinterpret_cast<__half2*>(srow + (odd ? chB : chA) * 8 * w) = pk;
        };
        put_pair(o[0], o[1], 0 * OUTW + (j - 1), 1 * OUTW + (j - 1));
        put_pair(o[2], o[3], 2 * OUTW + (j - 1), 3 * OUTW + (j - 1));
        put_pair(o[4], o[5], 4 * OUTW + (j - 1), 5 * OUTW + (j - 1));
        if ((j - 1) & 1) put_pair(stash, o[6], 6 * OUTW + (j - 2), 6 * OUTW + (j - 1));
        else stash = o[6];
        if (j == WIN - 1) put_pair(o[6], o[6], 6 * OUTW + (j - 1), 6 * OUTW + (j - 1));   // both lanes: same dword
      }
#pragma unroll
      for (int a = 0; a < OUTW; ++a) prev[a] = c[a];
    }
  };

  LevelGeom G, Gn;
  HalfTaps A, B;
  geom(0, G);
  request(G, 0, A);
  request(G, 1, B);
#pragma unroll
  for (int l = 0; l < NLEV; ++l) {
    const bool more = l + 1 < NLEV;
    // issue order so far: ... A_l(32) B_l(32) [stores of level l-1 (>= MIN_STORES)]
    if (l == 0) wait_vm<32>(); else wait_vm<32 + MIN_STORES>();
    landed(A);
    consume(G, A, 0);
    if (more) { geom(l + 1, Gn); request(Gn, 0, A); }
    // younger than B_l: [stores of level l-1] [A_{l+1}(32)]
    if (l == 0) wait_vm<32>(); else if (more) wait_vm<32 + MIN_STORES>(); else wait_vm<0>();
    landed(B);
    consume(G, B, 1);
    if (more) request(Gn, 1, B);
    lds_barrier();
    // one channel strip = 8 full rows = contiguous in the output; piece = 8 pixels (16 B)
    const int npieces = NCH_OUT * 8 * nbx;
    for (int o = tid; o < npieces; o += nthreads) {
      const int sg = o % nbx, row = (o / nbx) % 8, ch = o / (nbx * 8);
      const uint4 val = *reinterpret_cast<const uint4*>(s_out + (ch * 8 + row) * w + (sg ^ (row & (nbx - 1))) * 8);
      __half* dst = out + (((long)e * (NLEV * NCH_OUT) + l * NCH_OUT + ch) * h + by * 8 + row) * w + sg * 8;
      *reinterpret_cast<uint4*>(dst) = val;
    }
    lds_barrier();
    if (more) G = Gn;
  }
}

// the skewed chunks of the build kernel need 64 % w2_l == 0 on every level: w in {8,16,32,64}; h multiple of 8
bool dims_ok(int h, int w) { return h >= 8 && h % 8 == 0 && (w == 8 || w == 16 || w == 32 || w == 64); }

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
                       D.tgt_off[l], D.h2[l - 1], D.w2[l - 1]);
  hipLaunchKernelGGL(pyr_build_kernel, dim3(D.nblk, E), dim3(256), 0, st, (const __half*)f1T, (const __half*)f2T,
                     (__half*)pyramid, D, s1, s2);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_corr_pyramid_lookup(const void* pyramid, const float* coords, void* out,
                                      int E, int h, int w, dh_stream_t stream) {
  if (E < 0 || !dims_ok(h, w)) return DH_ERR_ARG;
  if (E == 0) return DH_OK;
  if (!pyramid || !coords || !out) return DH_ERR_ARG;
  const PyrDims D = make_dims(h, w);
  const size_t lds = (size_t)NCH_OUT * 8 * w * sizeof(__half);
  hipLaunchKernelGGL(pyr_lookup_kernel, dim3(h / 8, E), dim3((w / 8) * 64), lds, (hipStream_t)stream,
                     (const __half*)pyramid, coords, (__half*)out, D);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
