// Implicit-GEMM 2-D convolution (stride 1, "same" padding) on the fp16 MFMA with fused epilogues: the
// building block of the ConvGRU update operator (reference droid_slam/droid_net.py:78-143 UpdateModule,
// droid_slam/modules/gru.py:5-33 ConvGRU, droid_net.py:44-75 GraphAgg; cuDNN/cuBLAS under autocast there).
//
// GEMM view:  D[pixel][cout] = sum_{tap, c} X[pixel + tap][c] * Wt[cout][tap * Ctot + c]
//   M = N*H*W pixels (activations NHWC fp16, up to 4 channel segments that are concatenated on the fly:
//       the 448-channel GRU input cat(net, inp, corr, flow) is never materialised),
//   N = Cout, K = KH*KW*Ctot (weights pre-packed [CoutPad][Kpad], K contiguous, zero padded).
// Workgroup = 8 waves, tile 256 pixels x BN couts, K chunks of 64; v_mfma_f32_32x32x16_f16, fp32 accumulate.
//   * operands go global -> registers (issued before the MFMAs of the current chunk) -> LDS, one LDS stage:
//     2 workgroups per CU hide each other's staging;
//   * LDS rows are 64 + 8 halves (144 B): the 16-byte fragment reads of 16 consecutive rows fall in 16
//     disjoint bank quads, so ds_read_b128 is conflict-free (row stride 36 dwords, 36*m mod 64 distinct);
//   * epilogues fuse bias, activation and the GRU algebra, so z, r*net and the new hidden state are written
//     once and never re-read by an elementwise kernel:
//       ZR   cout <  128: z = sigmoid(acc + b + g)          cout >= 128: r*net = sigmoid(acc + b + g) * net
//       Q    net' = (1 - z) * net + z * tanh(acc + b + g)   (in place on net)
//       GLO  sum over pixels of sigmoid(acc + b) * net  -> per-image fp32 accumulators (global context)
//   where g[image][cout] is the 1x1 "global context" term of the ConvGRU.
//
// Kernels in this file, in the order a convolution is dispatched (dh_conv2d_nhwc_f16 at the bottom):
//   conv3x3_dma_kernel    opt-in experiment (DH_CONV_DMA=1): everything by LDS-DMA, one workgroup per CU;
//   conv3x3_halo2_kernel  3x3, cout tiles of 128 (the GRU gates, corr/flow encoders' second layers, first head layers):
//                         weights by LDS-DMA, halo in 64-byte runs, LDS-staged epilogue -- the kernel most of an update
//                         iteration is spent in (1.1 PFLOP/s); kept free of scratch by a build-time ISA audit (build.py);
//   conv3x3_halo_kernel   3x3 halo-tile loop with register staging: cout tiles of 64 and 32 (small heads), and the first
//                         form of the 128-cout tile (DH_CONV_HALO2=0);
//   conv_igemm_kernel     the generic loop described above: 1x1 and 7x7 convolutions, images that are not 64 wide, fallback;
//   corr0_nchw_kernel     corr_encoder.0 (1x1, 196 -> 128) on the reference-layout lookup output: in-register tile transpose,
//                         persistent workgroups (dh_corr0_nchw_f16); conv7x7_c4_kernel / glo_reduce_kernel / glo_gemv_kernel /
//                         heads_gather_kernel / segment_mean_kernel: the single-purpose kernels of the update operator;
//   staged_epilogue / staged_glo_epilogue   results parked in LDS and finished in 16-byte pieces (coalesced stores and GRU
//                         operand loads); conv_epilogue is the per-element form for tiles that straddle images or write fp32.
#include "common.h"

namespace {
#ifndef DH_CONV_NT
#define DH_CONV_NT 0        // 1 = the staged epilogues' output stores carry the non-temporal hint (variant builds for A/B runs)
#endif
typedef unsigned int u32x4_cnt __attribute__((ext_vector_type(4)));
using namespace dh;

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

using half2v = __attribute__((ext_vector_type(2))) _Float16;
constexpr int BM = 256;            // pixels per workgroup
constexpr int BK = 64;             // K chunk
constexpr int LDT = BK + 8;        // LDS row stride (halves)
constexpr int MAXSEG = 4;

enum Epi { EPI_LINEAR = 0, EPI_RELU = 1, EPI_SIGMOID = 2, EPI_GRU_ZR = 3, EPI_GRU_Q = 4, EPI_GLO = 5,
           EPI_SOFTPLUS_001 = 6, EPI_HEADS = 7, EPI_HEADS0 = 8 };

struct ConvParams {
  const __half* in[MAXSEG]; int segC[MAXSEG]; int segS[MAXSEG]; int nseg; int Ctot;   // segS = pixel stride (elements)
  const __half* wt; const __half* wt_halo; const float* bias;
  int N, H, W, KH, KW, Cout, CoutPad, Kreal, Kpad, epi;
  void* out; int out_f32; int out_stride;
  const float* gterm; const __half* aux0; int aux0_stride; const __half* aux1; int aux1_stride; float* red;
  // accumulator start values: acc(image n, pixel r, cout) = cinit[(cinit_idx[n] * H*W + r) * cinit_stride + cinit_off + cout]
  // -- the contribution of input channels that are shared by all images of a group (the ConvGRU's per-source-frame
  // context features), computed once per group by another convolution and kept in fp32
  // cinit_stride < 0 (round 5): the ACCUMULATOR-TILE layout -- cinit holds, per (frame pixel tile of 256, cout tile of 128), the
  // 64 x 64 wave tiles of conv3x3_halo2_kernel as the registers hold them: [pixel tile][cout tile of -cinit_stride / 128][wave 8]
  // [a*2+b][q>>2][lane 64][q&3] fp32, written by the same kernel with out_f32 == 3 (store_acc_tile) and read back as sixteen
  // 16-byte loads per lane (1 KB contiguous per wave-load) instead of 64 dword loads
  const float* cinit; const int64_t* cinit_idx; int cinit_stride; int cinit_off;
  // stride-2 "same" convolution (the encoders' down-sampling layers, extractor.py:140,151,24): H, W above are the OUTPUT size, the input
  // is Hin x Win = 2H x 2W and output pixel (y, x) reads input (2y + dy - pad, 2x + dx - pad); conv_igemm_kernel only; 1 / H / W otherwise
  int stride, Hin, Win;
  // Round 6, EPI_GRU_Q only (glo_red != nullptr): the NEXT iteration's global-context reduction fused behind the state update --
  // glo_red[image][c] += sum over the tile's pixels of sigmoid(glo_wt . net' + glo_bias)[c] * net'[c] on the new hidden state that is
  // still in LDS (what glo_reduce_kernel computes from HBM at the start of the next iteration: 3.2 GB of reads at 4096 edges)
  const __half* glo_wt; const float* glo_bias; float* glo_red;
  int nt_out;             // 1: the staged epilogue stores its tile with the non-temporal hint (write-dominated launches: the stem, the upmask head)
  int xcd_tiles;          // > 0: 1-D grid, workgroup id -> (pixel tile, cout tile) through xcd_decode(); = pixel tiles per XCD
  int ny;                 // cout tiles
#ifdef DH_ABLATION
  unsigned long long* ts; // phase timestamps per workgroup (dh_conv_set_timestamps; scripts/conv_timeline.py), else nullptr
  unsigned long long* ts_step; unsigned ts_step_wg0;     // per-step stamps of 256 workgroups from ts_step_wg0 on (64 slots each), else nullptr
#endif
};

// Phase timestamps of the 3x3 kernels (-DDH_ABLATION builds, when a buffer is set): thread 0 of every workgroup stores the
// 100 MHz wall clock at  0 kernel entry, 1 accumulators initialised / first fetches issued, 2 first barrier passed (the first
// MFMA can issue), 3 main loop left, 4 epilogue done;  slot 5 = HW_ID, 6 = XCC_ID (which CU ran it).
#ifdef DH_ABLATION
#define DH_CTS(i) do { if (P.ts && threadIdx.x == 0) P.ts[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
#define DH_CTS_ID() do { if (P.ts && threadIdx.x == 0) { P.ts[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 5] = __builtin_amdgcn_s_getreg(63492); \
                                                         P.ts[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 6] = __builtin_amdgcn_s_getreg(63508); } } while (0)
// per-STEP stamps of the main loop for the workgroups [P.ts_step_wg0, P.ts_step_wg0 + 256): 64 slots each behind the per-workgroup block
#define DH_CTS_STEP(step_) do { if (P.ts_step && threadIdx.x == 0) { const unsigned w_ = blockIdx.y * gridDim.x + blockIdx.x - P.ts_step_wg0; \
                                 if (w_ < 256u && (step_) < 64) P.ts_step[(size_t)w_ * 64 + (step_)] = wall_clock64(); } } while (0)
#else
#define DH_CTS(i) do { } while (0)
#define DH_CTS_ID() do { } while (0)
#define DH_CTS_STEP(step_) do { } while (0)
#endif

// XCD-aware workgroup order (the dispatcher places workgroup b on XCD b % 8, each XCD has its own L2): XCD x walks the
// contiguous run of pixel tiles [x * T, (x + 1) * T) and visits the cout tiles of a pixel tile back to back, so the halo
// rows shared by vertically adjacent tiles, the second cout tile's re-read of the same activations and the accumulator
// start values of a source frame's edges are L2 hits instead of trips to the other side of the fabric.
// Round 6: the number of pixel tiles need not be a multiple of 8 (a rank's shard of an 8-rank run holds 515 or 1025 edges: 6180 /
// 12300 tiles, which fell back to the plain order and lost 13 % on the gates): T = ceil(tiles / 8), the last XCD's run is shorter and
// the workgroups beyond it return at once (-> false).
__device__ __forceinline__ bool xcd_decode(const ConvParams& P, long& m0, int& n0, int bn, int bm = BM) {
  if (P.xcd_tiles > 0) {
    const unsigned id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const unsigned y = slot % (unsigned)P.ny, t = slot / (unsigned)P.ny;
    m0 = ((long)xcd * P.xcd_tiles + t) * bm; n0 = (int)y * bn;
    return m0 < (long)P.N * P.H * P.W;
  }
  m0 = (long)blockIdx.x * bm; n0 = blockIdx.y * bn;
  return true;
}

// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (ten instructions per gate value): every result is rounded to fp16
// right after, as the reference's autocast does with its own fp32 sigmoid / tanh
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { const float e = __expf(-2.f * fabsf(x)); const float t = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e); return x < 0 ? -t : t; }
__device__ __forceinline__ float round_h(float v) { return __half2float(__float2half(v)); }

// v if ok else 0, component-wise (a select on the 128-bit vector is lowered through scratch memory by hipcc)
__device__ __forceinline__ uint4 keep_if(bool ok, const uint4& v) {
  const uint32_t m = ok ? 0xffffffffu : 0u;
  return uint4{v.x & m, v.y & m, v.z & m, v.w & m};
}

// accumulators start from zero or from P.cinit (see ConvParams); same element mapping as conv_epilogue.  one_img: the
// tile lies inside one image (always true for the halo kernels), so the image index is formed once
// start values for a tile that lies inside ONE image (the halo kernels): no per-element image index
template <int TM, int TN>
__device__ __forceinline__ void init_acc_tile(const ConvParams& P, f32x16 (&acc)[TM][TN], long m0, int n0, int wm0, int wn0,
                                              int lane, int HW) {
  const int img0 = (int)(m0 / HW);
  const float* base = P.cinit + ((long)P.cinit_idx[img0] * HW + (m0 - (long)img0 * HW) + wm0 + 4 * (lane >> 5)) * P.cinit_stride
                      + P.cinit_off + n0 + wn0 + (lane & 31);
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const bool co_ok = n0 + wn0 + b * 32 + (lane & 31) < P.Cout;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int q = 0; q < 16; ++q)            // 32 lanes read one 128-byte run of a pixel's couts
        acc[a][b][q] = co_ok ? base[(long)(a * 32 + (q & 3) + 8 * (q >> 2)) * P.cinit_stride + b * 32] : 0.f;
  }
}

// the accumulator-tile layout (ConvParams::cinit_stride < 0): register dump / restore of the 2 x 2 tiles of a 64 x 64 wave tile
using f32x4v = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ long acc_tile_base(long pixel_tile, int n_cout_tiles, int cout_tile, int wave) {
  return (((pixel_tile * n_cout_tiles + cout_tile) * 8 + wave) * 4) * 4 * 64;          // in 16-byte units
}
__device__ __forceinline__ void init_acc_tile_tiled(const ConvParams& P, f32x16 (&acc)[2][2], long m0, int n0, int wave, int lane, int HW) {
  const int img0 = (int)(m0 / HW);
  const long pt = (long)P.cinit_idx[img0] * (HW / BM) + (m0 - (long)img0 * HW) / BM;
  const f32x4v* base = reinterpret_cast<const f32x4v*>(P.cinit) + acc_tile_base(pt, -P.cinit_stride / 128, (P.cinit_off + n0) / 128, wave) + lane;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4v v = base[((a * 2 + b) * 4 + j) * 64];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[a][b][4 * j + i] = v[i];
      }
}
__device__ __forceinline__ void store_acc_tile(const ConvParams& P, const f32x16 (&acc)[2][2], long m0, int n0, int wn0, int wave, int lane) {
  f32x4v* base = reinterpret_cast<f32x4v*>(P.out) + acc_tile_base(m0 / BM, P.CoutPad / 128, n0 / 128, wave) + lane;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const float bias = P.bias[n0 + wn0 + b * 32 + (lane & 31)];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4v v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[a][b][4 * j + i] + bias;
        base[((a * 2 + b) * 4 + j) * 64] = v;
      }
  }
}

template <int TM, int TN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[TM][TN]) {
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
}

// ---- epilogue shared by both main loops: lane holds cout = n0 + wn0 + b*32 + (lane&31) and the 16 pixels
// m0 + wm0 + a*32 + (q&3) + 8*(q>>2) + 4*(lane>>5) of every accumulator tile
template <int EPI, int TM, int TN>
__device__ __forceinline__ void conv_epilogue(const ConvParams& P, f32x16 (&acc)[TM][TN], long M, long m0, int n0, int wm0,
                                              int wn0, int lane, int HW) {
  const int img0 = (int)(m0 / HW);
  const bool one_img = (m0 + BM - 1) / HW == img0;        // wave-uniform
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int co = n0 + wn0 + b * 32 + (lane & 31);
    const bool co_ok = co < P.Cout;
    const float bias = co < P.CoutPad ? P.bias[co] : 0.f;
    const float g_one = (one_img && P.gterm && co < P.CoutPad) ? P.gterm[(long)img0 * P.CoutPad + co] : 0.f;
    float glo_sum = 0.f; long glo_img = -1;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const long pix = m0 + wm0 + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (pix >= M || !co_ok) continue;
        // (a 64-bit division per element is ~100 instructions: the tile normally lies inside one image)
        const int img = one_img ? img0 : (int)(pix / HW);
        float v = acc[a][b][q] + bias + (one_img ? g_one : (P.gterm ? P.gterm[(long)img * P.CoutPad + co] : 0.f));
        switch (EPI) {
          case EPI_RELU: v = fmaxf(v, 0.f); break;
          case EPI_SIGMOID: v = sigmoidf_(v); break;
          case EPI_GRU_ZR: {
            v = sigmoidf_(v);
            if (co >= 128) v = round_h(v) * __half2float(P.aux0[pix * P.aux0_stride + co - 128]);   // r * net
          } break;
          case EPI_GRU_Q: {
            const float qv = round_h(tanhf_(v));
            const float z = __half2float(P.aux1[pix * P.aux1_stride + co]);
            const float h = __half2float(P.aux0[pix * P.aux0_stride + co]);
            v = (1.f - z) * h + z * qv;
          } break;
          case EPI_GLO: {
            v = round_h(sigmoidf_(v)) * __half2float(P.aux0[pix * P.aux0_stride + co]);
            v = round_h(v);
            if (glo_img >= 0 && glo_img != img) { atomicAdd(&P.red[glo_img * P.Cout + co], glo_sum); glo_sum = 0.f; }
            glo_img = img; glo_sum += v;
          } break;
          case EPI_SOFTPLUS_001: { const float hv = round_h(v); v = 0.01f * (hv > 20.f ? hv : log1pf(__expf(hv))); } break;   // fp32 softplus of the fp16 conv output (autocast's fp32 list)
          case EPI_HEADS: if (co >= 2) v = sigmoidf_(round_h(v)); break;          // (delta_x, delta_y, w_x, w_y)
          default: break;
        }
        if (EPI == EPI_GLO) continue;
        if (P.out_f32 == 2 || (P.out_f32 && EPI == EPI_SOFTPLUS_001)) reinterpret_cast<float*>(P.out)[pix * P.out_stride + co] = v;   // raw fp32
        else if (P.out_f32) reinterpret_cast<float*>(P.out)[pix * P.out_stride + co] = round_h(v);
        else reinterpret_cast<__half*>(P.out)[pix * P.out_stride + co] = __float2half(v);
      }
    if (EPI == EPI_GLO && glo_img >= 0) {
      // lanes l and l+32 hold the same cout: combine, one atomic per (wave, cout)
      const float other = __shfl_xor(glo_sum, 32, 64);
      const long oimg = ((long)__shfl_xor((int)glo_img, 32, 64));
      if (oimg == glo_img) { if (lane < 32) atomicAdd(&P.red[glo_img * P.Cout + co], glo_sum + other); }
      else atomicAdd(&P.red[glo_img * P.Cout + co], glo_sum);
    }
  }
}

// ---- LDS-staged epilogue of the 256 px x 128 cout workgroup tile (halo kernel, BN = 128) -----------------------------
// The per-element epilogue above stores 2 bytes per lane (64 store instructions per lane, two half-used lines each) and
// gathers the GRU operands the same way; measured, prologue + epilogue cost a workgroup 11 us next to 5.5 us per
// 16-channel chunk.  Here every wave applies bias + activation to its accumulators and parks the fp16 results in an
// LDS tile [256 px][128 cout] (row stride 272 B); then the workgroup walks the tile in 16-byte pieces (8 couts of one
// pixel): the GRU operands arrive as 16-byte loads, the gate algebra is applied, and a pixel's 128 couts leave as one
// 256-byte run.  Same operations and roundings per element as conv_epilogue.

template <int EPI>
inline bool staged_epilogue_ok(const ConvParams& P) {
  if (!opts().conv_epi_staged) return false;                // 0: per-element epilogue everywhere (A/B runs)
  if (EPI == EPI_HEADS0) return P.aux1 && P.red && P.Cout == P.CoutPad && P.CoutPad % 128 == 0 && ((uintptr_t)P.aux1) % 16 == 0 &&
                                (!P.out || (!P.out_f32 && P.out_stride % 8 == 0 && ((uintptr_t)P.out) % 16 == 0));
  if (EPI == EPI_GLO) return P.Cout % 8 == 0 && P.aux0_stride % 8 == 0 && ((uintptr_t)P.aux0) % 16 == 0;
  if (P.out_f32 || P.Cout % 8 || P.out_stride % 8 || ((uintptr_t)P.out) % 16) return false;
  if (EPI == EPI_GRU_ZR && (P.aux0_stride % 8 || ((uintptr_t)P.aux0) % 16)) return false;
  if (EPI == EPI_GRU_Q && (P.aux0_stride % 8 || P.aux1_stride % 8 || ((uintptr_t)P.aux0) % 16 || ((uintptr_t)P.aux1) % 16)) return false;
  return true;
}

// second phase of the staged epilogues: the workgroup walks the fp16 tile [256 px][BNT couts] in 16-byte pieces (8 couts of
// one pixel), applies the gate algebra with 16-byte operand loads and stores a pixel's couts as one run
// DH_EPI_EARLY (round 6, default 1; -DDH_EPI_EARLY=0 = round 5): the GRU operands (old state / z: 16-byte loads from HBM) are requested WHILE
// the accumulators are being parked -- two pieces' worth after each of the four accumulator tiles, whose sixteen registers have just
// become free -- instead of after the barrier behind the parking: their latency runs under the gate activations (128 transcendentals
// per lane) instead of in front of the store loop.  Same loads, same values.
#ifndef DH_EPI_EARLY
#define DH_EPI_EARLY 1
#endif
template <int EPI, int BNT, int NT, int NIT>
struct GruOperands { uint4 hv[(EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q) ? NIT : 1], zv[EPI == EPI_GRU_Q ? NIT : 1]; };

// operands of piece `it` of thread `tid` (piece id = tid + NT * it -> pixel row id / PPR, couts (id % PPR) * 8 ..)
template <int EPI, int BNT, int NT, int NIT>
__device__ __forceinline__ void gru_operand_load(const ConvParams& P, GruOperands<EPI, BNT, NT, NIT>& g, int it, long m0, int n0, int tid) {
  constexpr int PPR = BNT / 8;
  if constexpr (EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q) {
    const int id = tid + NT * it, row = id / PPR, co = n0 + (id % PPR) * 8;
    const long pix = m0 + row;
    g.hv[it] = uint4{0u, 0u, 0u, 0u};
    if (co < P.Cout) {
      if (EPI == EPI_GRU_Q) {
        g.zv[it] = *reinterpret_cast<const uint4*>(P.aux1 + pix * P.aux1_stride + co);
        g.hv[it] = *reinterpret_cast<const uint4*>(P.aux0 + pix * P.aux0_stride + co);
      } else if (co >= 128) {
        g.hv[it] = *reinterpret_cast<const uint4*>(P.aux0 + pix * P.aux0_stride + co - 128);
      }
    }
  }
}

template <int EPI, int BNT, int NT = 512, bool KEEP = false, int ROWS = 256, bool PRELOADED = false>      // KEEP: the finished pieces are also written back into the LDS tile
__device__ __forceinline__ void staged_tile_store(const ConvParams& P, __half* __restrict__ sT, long m0, int n0, int tid,
                                                  GruOperands<EPI, BNT, NT, ROWS * (BNT / 8) / NT>& g) {
  constexpr int ELD = BNT + 8;
  constexpr int PPR = BNT / 8;
  constexpr int NIT = ROWS * PPR / NT;
  // 256 px x PPR pieces; thread -> (pixel row, piece): PPR consecutive lanes cover the couts of one pixel (NT = threads of the workgroup)
  // Round 5: the GRU operands of ALL the thread's pieces are requested first.  The q gate updates the hidden state in place (out ==
  // aux0), so with the loads inside the store loop the compiler had to keep every load behind the previous piece's store: eight
  // exposed memory latencies per workgroup -- 16.0 us of epilogue against 4.3 us for the plain one (scripts/conv_timeline.py).
  // A thread only ever reads the piece it is about to overwrite itself, so requesting them up front changes no value.
  if constexpr ((EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q) && !PRELOADED) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) gru_operand_load<EPI, BNT, NT, NIT>(P, g, it, m0, n0, tid);
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int id = tid + NT * it, row = id / PPR, c8 = (id % PPR) * 8;
    const int co = n0 + c8;
    if (co >= P.Cout) continue;
    const long pix = m0 + row;
    uint4 v = *reinterpret_cast<const uint4*>(sT + row * ELD + c8);
    if (EPI == EPI_GRU_ZR) {
      if (co >= 128) {                                                                   // r * net
        const __half2* a2 = reinterpret_cast<const __half2*>(&v); const __half2* h2 = reinterpret_cast<const __half2*>(&g.hv[it]);
        uint4 o; __half2* o2 = reinterpret_cast<__half2*>(&o);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float2 af = __half22float2(a2[k]), hf = __half22float2(h2[k]);
          o2[k] = __floats2half2_rn(af.x * hf.x, af.y * hf.y);
        }
        v = uint4{o.x, o.y, o.z, o.w};
      }
    } else if (EPI == EPI_GRU_Q) {
      const __half2* q2 = reinterpret_cast<const __half2*>(&v); const __half2* z2 = reinterpret_cast<const __half2*>(&g.zv[it]);
      const __half2* h2 = reinterpret_cast<const __half2*>(&g.hv[it]);
      uint4 o; __half2* o2 = reinterpret_cast<__half2*>(&o);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float2 qf = __half22float2(q2[k]), zf = __half22float2(z2[k]), hf = __half22float2(h2[k]);
        o2[k] = __floats2half2_rn((1.f - zf.x) * hf.x + zf.x * qf.x, (1.f - zf.y) * hf.y + zf.y * qf.y);
      }
      v = uint4{o.x, o.y, o.z, o.w};
    }
    // (round 6: -6 % on the stem, which writes 3.2 GB and reads 0.1 GB; neutral on the MFMA-bound layers, slightly negative on small launches
    //  whose output the next kernel finds in L2 -- hence per launch; -DDH_CONV_NT=1 forces it everywhere for A/B runs)
    // (as inline asm: LLVM merges an if / else of two stores of the same value into ONE plain store and drops the hint)
    if (DH_CONV_NT) {
      __builtin_nontemporal_store(__builtin_bit_cast(u32x4_cnt, v), reinterpret_cast<u32x4_cnt*>(reinterpret_cast<__half*>(P.out) + pix * P.out_stride + co));
    } else if (P.nt_out) {
      const __half* optr = reinterpret_cast<const __half*>(P.out) + pix * P.out_stride + co;
      asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(optr), "v"(__builtin_bit_cast(u32x4_cnt, v)) : "memory");
    } else {
      *reinterpret_cast<uint4*>(reinterpret_cast<__half*>(P.out) + pix * P.out_stride + co) = v;
    }
    if constexpr (KEEP) *reinterpret_cast<uint4*>(sT + row * ELD + c8) = v;      // (a thread overwrites only the piece it has just read)
  }
}

template <int EPI, int BNT, int NT = 512, bool KEEP = false, int ROWS = 256>
__device__ __forceinline__ void staged_tile_store(const ConvParams& P, __half* __restrict__ sT, long m0, int n0, int tid) {
  GruOperands<EPI, BNT, NT, ROWS * (BNT / 8) / NT> g;
  staged_tile_store<EPI, BNT, NT, KEEP, ROWS, false>(P, sT, m0, n0, tid, g);
}

// TN = 32-cout accumulator tiles per wave, BNT = couts of the workgroup tile (128 or 64)
template <int EPI, int TN, int BNT, int NT = 512, bool KEEP = false, int ROWS = 256>
__device__ __forceinline__ void staged_epilogue(const ConvParams& P, f32x16 (&acc)[2][TN], __half* __restrict__ sT, long m0, int n0,
                                                int wm0, int wn0, int tid, int HW) {
  constexpr int ELD = BNT + 8;            // LDS row stride of the staged tile (halves)
  constexpr int PPR = BNT / 8;            // 16-byte pieces per pixel row
  constexpr int NIT = ROWS * PPR / NT;
  constexpr bool EARLY = DH_EPI_EARLY && (EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q) && NIT % (2 * TN) == 0;
  const int lane = tid & 63;
  const int img = (int)(m0 / HW);
  GruOperands<EPI, BNT, NT, NIT> g;
  __syncthreads();                        // the operand tiles of the main loop are dead
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int cl = wn0 + b * 32 + (lane & 31), co = n0 + cl;
    const float add = (co < P.CoutPad ? P.bias[co] : 0.f) + ((P.gterm && co < P.CoutPad) ? P.gterm[(long)img * P.CoutPad + co] : 0.f);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = wm0 + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        float v = acc[a][b][q] + add;
        switch (EPI) {
          case EPI_RELU: v = fmaxf(v, 0.f); break;
          case EPI_SIGMOID: case EPI_GRU_ZR: v = sigmoidf_(v); break;
          case EPI_GRU_Q: v = tanhf_(v); break;
          default: break;
        }
        sT[row * ELD + cl] = __float2half(v);
      }
      if constexpr (EARLY) {               // this accumulator tile's registers are free: request the next NIT / (2 TN) pieces' operands
        constexpr int PER = NIT / (2 * TN);
#pragma unroll
        for (int k = 0; k < PER; ++k) gru_operand_load<EPI, BNT, NT, NIT>(P, g, (b * 2 + a) * PER + k, m0, n0, tid);
      }
    }
  }
  __syncthreads();
  staged_tile_store<EPI, BNT, NT, KEEP, ROWS, EARLY>(P, sT, m0, n0, tid, g);
}

// EPI_HEADS0: first layer of the delta | weight heads (3x3, 128 -> 256, relu) FUSED with the second layer's channel
// contraction.  The second layer is a 3x3 convolution 256 -> 4: o(p) = sum_t W2_t . hd(p + t).  Its per-pixel part
// P_t(q) = W2_t . hd(q) (9 taps x 4 outputs = 36 dot products over the channels) only needs the pixel's own hidden
// vector, which this workgroup has just computed: the relu'd fp16 tile is parked in LDS as for the other staged epilogues,
// multiplied on the MFMA with the 36 (padded to 64) x 128 weight slice of this cout tile, and the fp32 partial products
// [cout tile][pixel][36] are all that leaves the kernel -- the 256-channel head activations (6.4 GB at 4096 edges) are
// never written nor re-read with a halo by a second convolution; the taps whose source pixel lies in this tile are summed here,
// heads_gather_kernel adds the rows that the tiles above and below contribute.
// P.aux1 = W2 packed [CoutPad/128][64][128] f16 (n = tap*4 + output), P.red = partials [CoutPad/128][M/256 tiles][6 rows][64][4] f32.
__device__ __forceinline__ void staged_heads0_epilogue(const ConvParams& P, f32x16 (&acc)[2][2], __half* __restrict__ sT, long m0, int n0,
                                                       int wm0, int wn0, int tid) {
  constexpr int ELD = 128 + 8;
  const int lane = tid & 63, wave = tid >> 6;
  __syncthreads();
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int cl = wn0 + b * 32 + (lane & 31), co = n0 + cl;
    const float add = co < P.CoutPad ? P.bias[co] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = wm0 + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        sT[row * ELD + cl] = __float2half(fmaxf(acc[a][b][q] + add, 0.f));
      }
  }
  __syncthreads();
  // round 6: P.out != nullptr -- the relu'd activations ALSO leave the kernel (GraphAgg's second convolution: the upmask head reads them,
  // and the 3x3 eta head is the fused second layer here; for the delta | weight heads nothing else reads them and P.out is null)
  if (P.out) staged_tile_store<EPI_RELU, 128>(P, sT, m0, n0, tid);
  const int mrow = (wave & 3) * 64, nh = (wave >> 2) * 32, p = lane & 31, kh = lane >> 5;
  const int tile = n0 >> 7;
  const __half* wsrc = P.aux1 + ((long)tile * 64 + nh + p) * 128 + kh * 8;
  f32x16 d[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int q = 0; q < 16; ++q) d[a][q] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const half8 bf = *reinterpret_cast<const half8*>(wsrc + ks * 16);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const half8 af = *reinterpret_cast<const half8*>(sT + (mrow + a * 32 + p) * ELD + ks * 16 + kh * 8);
      d[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, d[a], 0, 0, 0);
    }
  }
  // transpose through LDS so that the partials leave as [tap][pixel][4 outputs] planes in full lines (and the gather
  // reads them coalesced): plane stride 1024 + 4 floats = conflict-free scatter of the MFMA layout
  constexpr int PLD = 256 * 4 + 4;
  float* sD = reinterpret_cast<float*>(sT);
  __syncthreads();                                  // every wave has read its A fragments: the tile may be overwritten
  const int n = nh + p;
  if (n < 36) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = mrow + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
        sD[(n >> 2) * PLD + row * 4 + (n & 3)] = d[a][q];
      }
  }
  __syncthreads();
  // The tile is four full image rows (W = 64, m0 a multiple of 256, H a multiple of 4): the x-neighbours of every pixel and up to three
  // of its y-neighbours are in this tile.  Sum the taps whose source pixel is here before anything leaves the CU: six rows of
  // o-partials (output rows -1 .. 4 relative to the tile) instead of nine planes of four rows -- 6 KB instead of 36 KB per tile.
  const long ntm = ((long)P.N * P.H * P.W) >> 8;     // pixel tiles
  if (tid < 6 * 64) {
    const int ro = (tid >> 6) - 1, x = tid & 63;
    float4 acc4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      const int sr = ro + ty - 1;                    // o(y, x) += P_t(y + ty - 1, x + tx - 1)
      if ((unsigned)sr >= 4u) continue;
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const int sx = x + tx - 1;
        if ((unsigned)sx >= 64u) continue;
        const float4 v = *reinterpret_cast<const float4*>(sD + (ty * 3 + tx) * PLD + (sr * 64 + sx) * 4);
        acc4.x += v.x; acc4.y += v.y; acc4.z += v.z; acc4.w += v.w;
      }
    }
    *reinterpret_cast<float4*>(P.red + (((long)tile * ntm + (m0 >> 8)) * 384 + tid) * 4) = acc4;
  }
}

// o(p) = bias + sum over cout tiles of (own tile's row partial + the row the tile above / below contributes, same image), then the
// heads' activation (delta_x, delta_y raw; sigmoid on the two confidence outputs), values rounded to fp16 like the convolution it
// replaces.  part = [cout tiles][pixel tiles of 4 rows][6 output rows: -1 .. 4][64][4] f32 (staged_heads0_epilogue); W = 64.
// MODE 1 (round 6): GraphAgg's eta head -- output 0 only, 0.01 * softplus of the fp16-rounded sum (EPI_SOFTPLUS_001), one float per pixel
template <int MODE>
__global__ __launch_bounds__(256) void heads_gather_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ dw,
                                                           long M, int ntiles, int H) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const long pt = i >> 8, ntm = M >> 8;
  const int ry = (int)(i & 255) >> 6, x = (int)i & 63;
  const int y = (int)((i >> 6) % H);
  float o[4] = {bias[0], bias[1], bias[2], bias[3]};
  auto add = [&](long tile_, int row_) {
    const float4 v = *reinterpret_cast<const float4*>(part + ((tile_ * 6 + row_) * 64 + x) * 4);
    o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w;
  };
  for (int tl = 0; tl < ntiles; ++tl) {
    const long t0 = (long)tl * ntm + pt;
    add(t0, ry + 1);
    if (ry == 0 && y > 0) add(t0 - 1, 5);
    if (ry == 3 && y < H - 1) add(t0 + 1, 0);
  }
  if constexpr (MODE == 1) {
    const float hv = round_h(o[0]);
    dw[i] = 0.01f * (hv > 20.f ? hv : log1pf(__expf(hv)));
  } else {
    float4 res;
    res.x = round_h(o[0]); res.y = round_h(o[1]);
    res.z = round_h(sigmoidf_(round_h(o[2]))); res.w = round_h(sigmoidf_(round_h(o[3])));
    reinterpret_cast<float4*>(dw)[i] = res;
  }
}

// EPI_GLO through the same staged tile: sigmoid gate -> LDS, gate * feature with 16-byte operand loads, per-cout sum
// over the 256 pixels of the tile in registers / cross-lane / LDS, ONE atomic per (workgroup, cout) instead of one per
// (wave, cout) fed by 2-byte gathers.  Needs the whole tile inside one image.
__device__ __forceinline__ void staged_glo_epilogue(const ConvParams& P, f32x16 (&acc)[2][2], __half* __restrict__ sT, long m0, int n0,
                                                    int wm0, int wn0, int tid, int HW) {
  constexpr int ELD = 128 + 8;
  const int lane = tid & 63, wave = tid >> 6;
  const int img = (int)(m0 / HW);
  __syncthreads();
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int cl = wn0 + b * 32 + (lane & 31), co = n0 + cl;
    const float add = co < P.CoutPad ? P.bias[co] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = wm0 + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        sT[row * ELD + cl] = __float2half(sigmoidf_(acc[a][b][q] + add));
      }
  }
  __syncthreads();
  const int c8 = (tid & 15) * 8, co = n0 + c8;
  float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (co < P.Cout) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = (tid >> 4) + 32 * it;
      const uint4 g = *reinterpret_cast<const uint4*>(sT + row * ELD + c8);
      const uint4 f = *reinterpret_cast<const uint4*>(P.aux0 + (m0 + row) * P.aux0_stride + co);
      const __half2* g2 = reinterpret_cast<const __half2*>(&g); const __half2* f2 = reinterpret_cast<const __half2*>(&f);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float2 gf = __half22float2(g2[k]), ff = __half22float2(f2[k]);
        sum[2 * k] += round_h(gf.x * ff.x); sum[2 * k + 1] += round_h(gf.y * ff.y);
      }
    }
  }
  // lanes l, l^16, l^32, l^48 hold the same couts
#pragma unroll
  for (int k = 0; k < 8; ++k) { sum[k] += __shfl_xor(sum[k], 16, 64); sum[k] += __shfl_xor(sum[k], 32, 64); }
  __syncthreads();                          // the staged tile is dead: reuse its head as [8 waves][128] floats
  float* sR = reinterpret_cast<float*>(sT);
  if (lane < 16) {
#pragma unroll
    for (int k = 0; k < 8; ++k) sR[wave * 128 + c8 + k] = sum[k];
  }
  __syncthreads();
  if (tid < 128 && n0 + tid < P.Cout) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += sR[w * 128 + tid];
    atomicAdd(&P.red[(long)img * P.Cout + n0 + tid], t);
  }
}

// WM x WN = per-wave output tile, waves arranged (BM / WM) x (BN / WN)
template <int WM, int WN, int BN, int EPI, bool STAGED = false>
__global__ __launch_bounds__(512, 4) void conv_igemm_kernel(ConvParams P) {
  constexpr int WAVES_M = BM / WM, TM = WM / 32, TN = WN / 32;
  static_assert(WAVES_M * (BN / WN) == 8, "8 waves");
  extern __shared__ __half s_conv[];
  __half* sA = s_conv;                     // [BM][LDT]
  __half* sB = s_conv + BM * LDT;          // [BN][LDT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave % WAVES_M) * WM, wn0 = (wave / WAVES_M) * WN;
  const long M = (long)P.N * P.H * P.W;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int HW = P.H * P.W;
  const int padh = P.KH / 2, padw = P.KW / 2;

  // ---- loader roles: A pieces (16 B = 8 channels of one pixel), 4 per thread; B pieces, BN*8/512 per thread
  constexpr int A_PIECES = BM * 8 / 512, B_PIECES = (BN * 8 + 511) / 512;
  int a_y[A_PIECES], a_x[A_PIECES], a_row0[A_PIECES];       // a_row0 = image * H (row index of the image's first row)
#pragma unroll
  for (int i = 0; i < A_PIECES; ++i) {
    const int id = tid + 512 * i;
    const long pix = m0 + (id >> 3);
    const bool ok = pix < M;
    const long pc = ok ? pix : 0;
    const int n = (int)(pc / HW), r = (int)(pc - (long)n * HW);
    a_row0[i] = n * P.Hin; a_y[i] = ok ? (r / P.W) * P.stride : -(1 << 20); a_x[i] = (r - (r / P.W) * P.W) * P.stride;
  }
  const int kc = tid & 7;                  // 8-half piece inside the chunk (same for all of a thread's pieces)
  // running decomposition of k = chunk*64 + kc*8 into (tap -> dy, dx) and channel c
  int cur_c = kc * 8, cur_dy = 0, cur_dx = 0;
  while (cur_c >= P.Ctot) { cur_c -= P.Ctot; if (++cur_dx == P.KW) { cur_dx = 0; ++cur_dy; } }

  uint4 ra[A_PIECES], rb[B_PIECES];
  const int nchunks = P.Kpad / BK;
  auto fetch = [&](int chunk) {
    // segment of channel cur_c
    int s = 0, cs = cur_c;
#pragma unroll
    for (int q = 0; q < MAXSEG - 1; ++q) if (s == q && q + 1 < P.nseg && cs >= P.segC[q]) { cs -= P.segC[q]; s = q + 1; }
    const __half* base = P.in[0]; int segc = P.segS[0];
#pragma unroll
    for (int q = 1; q < MAXSEG; ++q) if (s == q) { base = P.in[q]; segc = P.segS[q]; }
    const bool kvalid = cur_dy < P.KH;     // beyond the last tap: zero padding of K
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
      const int yy = a_y[i] + cur_dy - padh, xx = a_x[i] + cur_dx - padw;
      const bool ok = kvalid && (unsigned)yy < (unsigned)P.Hin && (unsigned)xx < (unsigned)P.Win;
      const __half* src = base + ((long)(a_row0[i] + (ok ? yy : 0)) * P.Win + (ok ? xx : 0)) * segc + cs;
      ra[i] = ok ? *reinterpret_cast<const uint4*>(src) : uint4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i) {
      const int id = tid + 512 * i;
      const int row = id >> 3;
      const bool ok = row < BN && n0 + row < P.CoutPad;
      rb[i] = ok ? *reinterpret_cast<const uint4*>(P.wt + (long)(n0 + row) * P.Kpad + (long)chunk * BK + kc * 8) : uint4{0u, 0u, 0u, 0u};
    }
    // advance k by one chunk
    cur_c += BK;
    while (cur_c >= P.Ctot) { cur_c -= P.Ctot; if (++cur_dx == P.KW) { cur_dx = 0; ++cur_dy; } }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
      const int id = tid + 512 * i;
      *reinterpret_cast<uint4*>(sA + (id >> 3) * LDT + kc * 8) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i) {
      const int id = tid + 512 * i;
      if ((id >> 3) < BN) *reinterpret_cast<uint4*>(sB + (id >> 3) * LDT + kc * 8) = rb[i];
    }
  };

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);                   // (start values: halo2 kernel only; the dispatcher refuses them here)

  fetch(0);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    __syncthreads();                       // previous chunk's fragments have been read
    stage();
    __syncthreads();
    if (chunk + 1 < nchunks) fetch(chunk + 1);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      half8 af[TM], bf[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a)
        af[a] = *reinterpret_cast<const half8*>(sA + (wm0 + a * 32 + (lane & 31)) * LDT + ks * 16 + (lane >> 5) * 8);
#pragma unroll
      for (int b = 0; b < TN; ++b)
        bf[b] = *reinterpret_cast<const half8*>(sB + (wn0 + b * 32 + (lane & 31)) * LDT + ks * 16 + (lane >> 5) * 8);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
  }
  if constexpr (STAGED && EPI == EPI_GLO) staged_glo_epilogue(P, acc, s_conv, m0, n0, wm0, wn0, tid, HW);
  else if constexpr (STAGED) staged_epilogue<EPI, TN, BN>(P, acc, s_conv, m0, n0, wm0, wn0, tid, HW);
  else conv_epilogue<EPI, TM, TN>(P, acc, M, m0, n0, wm0, wn0, lane, HW);
}

// ---- 3x3 fast path: halo tile -----------------------------------------------------------------------------------
// For W == 64 the 256 consecutive pixels of a workgroup are 4 full image rows.  Per 16-channel chunk the 6 x 66 pixel
// halo of the tile is staged ONCE and all 9 taps read their A fragments from it at shifted pixel addresses, next to
// the 9 x 128 x 16 weight slab of the chunk (74 KB of LDS: two workgroups per CU): 9x less activation traffic from L2,
// staging and address arithmetic per MFMA than the generic loop and two barriers per 36 MFMAs instead of per 16.  Measured on MI355X the generic loop is
// bound by the texture-address cost of its gathers (removing the fetch alone: 0.73 -> 1.17 PFLOP/s), so the weight
// slab is read from a second, pre-packed copy [cout tile][chunk][tap][128][16] in which a chunk's slab is one
// contiguous 36 KB run (fully coalesced 1 KB wave loads) and the halo is read as 32-byte runs per pixel.  The generic
// loop moves ~9x the activations through the L2->L1 path (measured 8.5 TB/s on a 128->128 convolution: at its limit).
// channels per chunk: 16 for the 128-cout tile (74 KB of LDS, two workgroups per CU), 32 for the 64- and 32-cout
// tiles, whose weight slabs are small and whose time goes into fetching the halo (64-byte instead of 32-byte runs
// per pixel: half the lines touched per channel)
constexpr int halo_ck(int bn) { return bn == 128 ? 16 : 32; }
constexpr int halo_afc(int bn) { return bn == 32 ? 2 : 1; }     // chunks per halo fetch (see the kernel)
constexpr int HROWS = 6, HCOLS = 66, HPIX = HROWS * HCOLS;

// BN = 128: waves = 4 image rows x 2 cout halves, 64 px x 64 cout per wave (every 3x3 convolution with >= 128 couts).
// BN = 64 : waves = 4 image rows x 2 cout halves, 64 px x 32 cout per wave (flow encoder 128 -> 64).
// BN = 32 : waves = 4 image rows x 2 half rows, 32 px x 32 cout per wave: the 2-/1-channel heads (Cout padded to 32),
//           where the generic loop is bound by re-reading the activations 9 times from L2 (58 GB at 4096 edges).
template <int EPI, int BN, bool STAGED = false>
__global__ __launch_bounds__(512, 4) void conv3x3_halo_kernel(ConvParams P) {
  constexpr int WM = BN == 32 ? 32 : 64, WN = BN == 128 ? 64 : 32, TM = WM / 32, TN = WN / 32;
  constexpr int HCK = halo_ck(BN);        // channels per chunk
  constexpr int HLD = HCK + 8;            // LDS row stride (halves): 48 or 80 B, 16 consecutive rows hit 16 disjoint bank quads
  constexpr int HPC = HCK / 8;            // 16-byte pieces per pixel / cout row
  constexpr int HSLAB = 9 * BN * HCK;     // halves of one (cout tile, chunk) weight slab
  extern __shared__ __half s_conv[];
  __half* sA = s_conv;                    // [HPIX][HLD]
  __half* sB = s_conv + HPIX * HLD;       // [9][BN][HLD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wrow = wave & 3;
  const int wx0 = BN == 32 ? (wave >> 2) * 32 : 0;                  // first image column of the wave's pixels
  const int wn0 = BN == 32 ? 0 : (wave >> 2) * WN, wm0 = wrow * 64 + wx0;
  const long M = (long)P.N * P.H * P.W;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int HW = P.H * P.W;
  const int img = (int)(m0 / HW), y0 = (int)((m0 - (long)img * HW) / 64);

  // Measured (448->256, 1024 edges): without the halo fetch 7.2 -> 5.2 ms, without the weight fetch 5.8 ms, without
  // both 4.2 ms = 1.56 PFLOP/s -- the fetch path (cost ~ 128-byte lines touched per wave load, not bytes) is what holds
  // this kernel back.  Where registers allow (the 32-cout tile) the halo is therefore FETCHED for AFC = 2 chunks at a
  // time (full 128-byte lines per pixel) and staged one chunk at a time: a thread's pieces all belong to the same chunk
  // of the pair (512 % APP == 0), so at each chunk the threads of that chunk write theirs.  On the 128- and 64-cout
  // tiles the 8-16 extra registers spill inside the loop at the 128-VGPR budget of two workgroups per CU (9.6 vs 7.2 ms).
  constexpr int AFC = halo_afc(BN), APP = AFC * HPC;            // chunks per halo fetch, 16-byte pieces per pixel per fetch
  constexpr int A_PIECES = (HPIX * APP + 511) / 512, B_PIECES = (9 * BN * HPC + 511) / 512;
  int a_off[A_PIECES]; bool a_ok[A_PIECES];
#pragma unroll
  for (int i = 0; i < A_PIECES; ++i) {
    const int id = tid + 512 * i, hp = id / APP;
    const int hy = hp / HCOLS, hx = hp - hy * HCOLS;
    const int y = y0 - 1 + hy, x = hx - 1;
    a_ok[i] = hp < HPIX && (unsigned)y < (unsigned)P.H && (unsigned)x < 64u;
    a_off[i] = a_ok[i] ? (img * P.H + y) * 64 + x : 0;
  }
  const int aq = tid % APP;                                     // this thread's piece inside the pixel's fetched run
  const int a_phase = aq / HPC, a_c8 = (aq % HPC) * 8;          // chunk of the pair it belongs to, offset inside the LDS row
  const int c8 = (tid % HPC) * 8;
  const int nchunks = P.Ctot / HCK;
  const __half* bslab = P.wt_halo + ((long)blockIdx.y * nchunks) * HSLAB + tid * 8;     // + chunk * HSLAB + i * 4096

  uint4 ra[A_PIECES], rb[B_PIECES];
  // (written as macros: with by-reference lambdas hipcc keeps ra/rb in scratch memory)
#define HALO_FETCH_A(chunk_)                                                                                         \
  {                                                                                                                  \
    int cs = (chunk_) * HCK, sgi = 0;                                                                                \
    _Pragma("unroll") for (int q = 0; q < MAXSEG - 1; ++q)                                                           \
      if (sgi == q && q + 1 < P.nseg && cs >= P.segC[q]) { cs -= P.segC[q]; sgi = q + 1; }                          \
    const __half* base = P.in[0]; int segs = P.segS[0];                                                              \
    _Pragma("unroll") for (int q = 1; q < MAXSEG; ++q) if (sgi == q) { base = P.in[q]; segs = P.segS[q]; }           \
    _Pragma("unroll") for (int i = 0; i < A_PIECES; ++i) {                                                           \
      const uint4 v = *reinterpret_cast<const uint4*>(base + (long)a_off[i] * segs + cs + aq * 8);                   \
      ra[i] = keep_if(a_ok[i], v);                                                                                   \
    }                                                                                                                \
  }
#define HALO_FETCH_B(chunk_)                                                                                         \
  {                                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < B_PIECES; ++i)                                                             \
    {                                                                                                                \
      const bool inb = tid + 512 * i < 9 * BN * HPC;                                                                 \
      const uint4 v = *reinterpret_cast<const uint4*>(bslab + (long)(chunk_) * HSLAB + (inb ? i * 4096 : 0));        \
      rb[i] = uint4{v.x, v.y, v.z, v.w};                                                                             \
    }                                                                                                                \
  }
#define HALO_STAGE(chunk_)                                                                                           \
  {                                                                                                                  \
    if (a_phase == ((chunk_) % AFC)) {                                                                               \
      _Pragma("unroll") for (int i = 0; i < A_PIECES; ++i) {                                                         \
        const int id = tid + 512 * i;                                                                                \
        if (id < HPIX * APP) *reinterpret_cast<uint4*>(sA + (id / APP) * HLD + a_c8) = ra[i];                        \
      }                                                                                                              \
    }                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < B_PIECES; ++i) {                                                           \
      const int id = tid + 512 * i;                                                                                  \
      if (id < 9 * BN * HPC) *reinterpret_cast<uint4*>(sB + (id / HPC) * HLD + c8) = rb[i];                          \
    }                                                                                                                \
  }

  f32x16 acc[TM][TN];
  zero_acc<TM, TN>(acc);                  // (start values are not supported by this kernel: the dispatcher routes them elsewhere)

  HALO_FETCH_A(0)
  HALO_FETCH_B(0)
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    __syncthreads();
    HALO_STAGE(chunk)
    __syncthreads();
    if (chunk + 1 < nchunks) {
      HALO_FETCH_B(chunk + 1)
      if ((chunk + 1) % AFC == 0) HALO_FETCH_A(chunk + 1)      // the registers of the pair were released by this chunk's stage
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3, dx = t - dy * 3;
#pragma unroll
      for (int ks = 0; ks < HCK / 16; ++ks) {
        half8 af[TM], bf[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
          af[a] = *reinterpret_cast<const half8*>(sA + ((wrow + dy) * HCOLS + wx0 + a * 32 + (lane & 31) + dx) * HLD + ks * 16 + (lane >> 5) * 8);
#pragma unroll
        for (int b = 0; b < TN; ++b)
          bf[b] = *reinterpret_cast<const half8*>(sB + (t * BN + wn0 + b * 32 + (lane & 31)) * HLD + ks * 16 + (lane >> 5) * 8);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
      }
    }
  }
#undef HALO_FETCH_A
#undef HALO_FETCH_B
#undef HALO_STAGE
  if constexpr (STAGED) staged_epilogue<EPI, TN, BN>(P, acc, s_conv, m0, n0, wm0, wn0, tid, HW);
  else conv_epilogue<EPI, TM, TN>(P, acc, M, m0, n0, wm0, wn0, lane, HW);
}

// ---- 3x3 path with LDS-DMA staging: OPT-IN experiment (DH_CONV_DMA=1), not the default ----------------------------------
// Built to test whether the halo kernel above is held back by its staging (16-channel chunks use 32 of the 128 bytes of
// every line they touch, go through VGPRs and cost two barriers per 36 MFMAs).  Here a chunk is 64 channels = ONE full
// 128-byte line per halo pixel, brought in by `global_load_lds_dwordx4` (no VGPRs, no ds_write pass) into a double-buffered
// halo (2 x 50 KB) next to a triple-buffered (tap, chunk) weight slab (3 x 16 KB): 148 KB of LDS, one workgroup per CU,
// one barrier per 16 MFMAs per wave, fragments read one k-step ahead across the barrier (see the loop comment).
// The LDS image of a DMA is lane-linear (M0 + lane * 16), so rows are 128 B unpadded and bank conflicts are removed by
// an XOR swizzle of the 16-byte slot with bits 1..3 of the row index -- applied to the per-lane SOURCE address of the
// halo, baked into the pre-packed weights ([cout tile][chunk][tap][128 couts][8 slots][8]) and applied to the ds_read
// addresses.  Out-of-image halo pixels read a zero line.
// Measured on MI355X (1024 edges, scripts/bench_conv.py; DH_DMA_VAR selects the ablations):
//     448->256: 7.5 ms (864 TFLOP/s)   448->128: 3.6 ms (900)     halo kernel above: 7.5 / 3.7 ms
//     without the DMA inside the loop (stale operands): 6.1 / 3.1 ms (1.06-1.11 PFLOP/s) -- the ds_read + MFMA + barrier
//     skeleton alone; DMA only: 3.7 / 1.8 ms.  Skeleton ablations (no DMA): MFMA + barrier on constant operands 4.2 ms
//     (1.56 PFLOP/s: the ceiling of this tile incl. its exposed prologue/epilogue), ds_read + barrier 2.9 ms, no barrier
//     5.8 ms: LDS reads and MFMAs add up instead of overlapping, and the barrier is free.  Moving the DMA issue points
//     (after the barrier / a k-step later / split between the two waves of a SIMD), the fragment prefetch distance, or
//     running the two waves of each SIMD half a step apart (load phase | MFMA phase ping-pong, two barriers per step:
//     8.4 ms, skeleton 5.9 ms) changes nothing or loses.
// Inside the full update iteration it is SLOWER than the halo kernel (update operator 95.8 vs 92.3 ms at 4096 edges):
// with one workgroup per CU the prologue (first halo + slabs) and the epilogue are not overlapped with another
// workgroup's main loop.  Kept, tested (tests/test_gpu_parity.py runs all three main loops) and off by default; what it
// shows is that staging is not what limits these convolutions: the 64 x 64 wave tile itself tops out near 1.1 PFLOP/s.
constexpr int DCK = 64;                                   // channels per chunk
constexpr int DA_PIECES = (HPIX * 8 + 63) / 64;           // 1-KB DMA pieces per halo chunk (396 px x 128 B -> 50)
constexpr int DA_BYTES = DA_PIECES * 1024;
constexpr int DB_BYTES = 128 * DCK * 2;                   // one (tap, chunk) weight slab
constexpr int DB_BUFS = 3;                                // weight slabs are fetched two steps ahead
constexpr int DMA_LDS_BYTES = 2 * DA_BYTES + DB_BUFS * DB_BYTES;
typedef __attribute__((address_space(3))) char lds_char;
__device__ __attribute__((aligned(128))) uint4 g_zero_line[8];

// one wave-wide LDS-DMA: lane l copies the 16 bytes at gptr_ (per lane) to LDS byte address ldsaddr_ + l * 16
#define GLDS16(gptr_, ldsaddr_)                                                                                      \
  {                                                                                                                  \
    unsigned keep_;                                                                                                  \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(gptr_), "s"(ldsaddr_) : "memory");                                             \
  }

// VAR: 0 = product; timing ablations: 1 no DMA inside the loop, 2 no ds_read/MFMA inside the loop; and without DMA:
// 5 no ds_read (MFMA + barrier), 6 no barrier (ds_read + MFMA), 7 no MFMA (ds_read + barrier)
#ifdef DH_ABLATION   // opt-in LDS-DMA prototype (measured slower than conv3x3_halo2_kernel): not in release builds
template <int EPI, int VAR = 0>
__global__ __launch_bounds__(512, 2) void conv3x3_dma_kernel(ConvParams P) {
  extern __shared__ __half s_conv[];
  const char* const lds = reinterpret_cast<const char*>(s_conv);
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)s_conv;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wrow = wave & 3, wn0 = (wave >> 2) * 64, wm0 = wrow * 64;
  const long M = (long)P.N * P.H * P.W;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * 128;
  const int HW = P.H * P.W;
  const int img = (int)(m0 / HW), y0 = (int)((m0 - (long)img * HW) / 64);
  const int nchunks = P.Ctot / DCK, nsteps = nchunks * 9;

  // halo DMA roles: piece j = t * 8 + wave (t = 0..6) covers halo pixels j*8 .. j*8+7, lane -> (pixel, swizzled slot)
  int a_src[7];                                          // (image pixel index << 3) | source slot, or -1 (zero line)
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int j = t * 8 + wave, hp = j * 8 + (lane >> 3);
    const int slot = (lane & 7) ^ ((hp >> 1) & 7);
    const int hy = hp / HCOLS, hx = hp - hy * HCOLS;
    const int y = y0 - 1 + hy, x = hx - 1;
    const bool ok = hp < HPIX && (unsigned)y < (unsigned)P.H && (unsigned)x < 64u;
    a_src[t] = ok ? ((((img * P.H + y) * 64 + x) << 3) | slot) : -1;
  }
  const char* const zero_src = reinterpret_cast<const char*>(g_zero_line) + (lane & 7) * 16;
  const char* const bsrc = reinterpret_cast<const char*>(P.wt_halo) + (long)blockIdx.y * nsteps * DB_BYTES + lane * 16;

  // fragment addressing: A row = halo pixel, B row = cout; 16-byte slot s of row r sits at r*128 + ((s ^ ((r>>1)&7)) << 4)
  const int pl = wrow * HCOLS + (lane & 31), kh = lane >> 5;
  int b_row[2], b_x[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int r = wn0 + b * 32 + (lane & 31);
    b_row[b] = r * 128; b_x[b] = ((r >> 1) & 7) ^ kh;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

  // source of the halo chunk `chunk_`: segment base (+ channel offset) and pixel stride in bytes
#define DMA_SEGMENT(chunk_, base_, stride_)                                                                          \
  {                                                                                                                  \
    int cs = (chunk_) * DCK, sgi = 0;                                                                                \
    _Pragma("unroll") for (int q = 0; q < MAXSEG - 1; ++q)                                                           \
      if (sgi == q && q + 1 < P.nseg && cs >= P.segC[q]) { cs -= P.segC[q]; sgi = q + 1; }                          \
    const __half* sb = P.in[0]; int ss = P.segS[0];                                                                  \
    _Pragma("unroll") for (int q = 1; q < MAXSEG; ++q) if (sgi == q) { sb = P.in[q]; ss = P.segS[q]; }               \
    base_ = reinterpret_cast<const char*>(sb + cs); stride_ = (long)ss * 2;                                          \
  }
#define DMA_HALO_PIECE(t_, base_, stride_, buf_)                                                                     \
  if ((t_) * 8 + wave < DA_PIECES) {                                                                                 \
    const int v = a_src[t_];                                                                                         \
    const char* g = v >= 0 ? (base_) + (long)(v >> 3) * (stride_) + (v & 7) * 16 : zero_src;                         \
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (buf_) * DA_BYTES + ((t_) * 8 + wave) * 1024);        \
    GLDS16(g, dst)                                                                                                   \
  }
#define DMA_SLAB(step_, buf_)                                                                                        \
  {                                                                                                                  \
    const char* g = bsrc + (long)(step_) * DB_BYTES + wave * 1024;                                                   \
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + 2 * DA_BYTES + (buf_) * DB_BYTES + wave * 1024);      \
    GLDS16(g, dst)                                                                                                   \
    const char* g2 = g + 8 * 1024;                                                                                   \
    const unsigned dst2 = dst + 8 * 1024;                                                                            \
    GLDS16(g2, dst2)                                                                                                 \
  }
  // every DMA this wave has issued has landed (they were all issued a full step ago), then the workgroup barrier
#define DMA_WAIT_BARRIER() asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory")
  // fragment addresses of tap t_: A row = halo pixel of the lane shifted by the tap
#define SET_TAP(t_)                                                                                                  \
  {                                                                                                                  \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                                  \
      const int p = pl + ((t_) / 3) * HCOLS + ((t_) % 3) + a * 32;                                                   \
      a_row[a] = p * 128; a_x[a] = ((p >> 1) & 7) ^ kh;                                                              \
    }                                                                                                                \
  }
#define LOAD_FRAGS(af_, bf_, A_, B_, ks_)                                                                            \
  if (VAR != 5) {                                                                                                    \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                                    \
      af_[a] = *reinterpret_cast<const half8*>((A_) + a_row[a] + ((((ks_) * 2) ^ a_x[a]) << 4));                     \
    _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                                    \
      bf_[b] = *reinterpret_cast<const half8*>((B_) + b_row[b] + ((((ks_) * 2) ^ b_x[b]) << 4));                     \
    __builtin_amdgcn_sched_barrier(0);   /* keep the reads AHEAD of the previous k-step's MFMAs (hipcc sinks them) */ \
  }
#define MFMA_2X2(af_, bf_)                                                                                           \
  {                                                                                                                  \
    if (VAR == 7) { asm volatile("" :: "v"(af_[0]), "v"(af_[1]), "v"(bf_[0]), "v"(bf_[1])); }                        \
    else {                                                                                                           \
      _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                                  \
        _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                                \
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af_[a], bf_[b], acc[a][b], 0, 0, 0);                    \
    }                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  }

  const char* nbase; long nstride;
  DMA_SEGMENT(0, nbase, nstride)
#pragma unroll
  for (int t = 0; t < 7; ++t) DMA_HALO_PIECE(t, nbase, nstride, 0)
  DMA_SLAB(0, 0)
  if (nsteps > 1) DMA_SLAB(1, 1)
  DMA_WAIT_BARRIER();

  // Software pipeline.  Step s (tap t of chunk c) runs four k-steps of 4 MFMAs per wave; the fragments of a k-step are
  // read from LDS one k-step ahead (two register sets), across step boundaries too.  The ONE barrier of a step sits
  // between its k-steps 1 and 2: it publishes slab s+1 and the halo pieces issued a step ago (every DMA in flight at that
  // point was issued after the previous step's barrier, so vmcnt(0) is the exact wait), and right after it -- every wave
  // has finished step s-1 -- the wave issues the DMA of slab s+2 (into the buffer of slab s-1) and its halo piece of the
  // next chunk.  No k-step waits on the barrier for its operands, so the MFMA pipe does not drain around it.
  int a_row[2], a_x[2];
  half8 af0[2], bf0[2], af1[2], bf1[2];
  if (VAR == 5) { const half8 one = {1, 1, 1, 1, 1, 1, 1, 1}; af0[0] = af0[1] = bf0[0] = bf0[1] = af1[0] = af1[1] = bf1[0] = bf1[1] = one; }
  SET_TAP(0)
  LOAD_FRAGS(af0, bf0, lds, lds + 2 * DA_BYTES, 0)
  int bcur = 0;                                          // slab buffer of the current step: step % 3
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    if (more) DMA_SEGMENT(c + 1, nbase, nstride)
    const char* const Acur = lds + (c & 1) * DA_BYTES;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int step = c * 9 + t;
      const char* const Bcur = lds + 2 * DA_BYTES + bcur * DB_BYTES;
      const int bnext = bcur == 2 ? 0 : bcur + 1;
      if (VAR != 2) {
        LOAD_FRAGS(af1, bf1, Acur, Bcur, 1)
        MFMA_2X2(af0, bf0)
        LOAD_FRAGS(af0, bf0, Acur, Bcur, 2)
        MFMA_2X2(af1, bf1)
      }
      if (VAR != 6) DMA_WAIT_BARRIER();
#define ISSUE_HALO() if (t < 7 && more && (t < 7 ? t : 0) * 8 + wave < DA_PIECES) DMA_HALO_PIECE(t < 7 ? t : 0, nbase, nstride, (c + 1) & 1)
#define ISSUE_SLAB() if (step + 2 < nsteps) { const int b2 = bcur == 0 ? 2 : bcur - 1; DMA_SLAB(step + 2, b2) }
      if (VAR != 1 && VAR < 5) { ISSUE_HALO() ISSUE_SLAB() }
      if (VAR != 2) {
        LOAD_FRAGS(af1, bf1, Acur, Bcur, 3)
        MFMA_2X2(af0, bf0)
        SET_TAP(t == 8 ? 0 : t + 1)
        LOAD_FRAGS(af0, bf0, (t == 8 ? lds + ((c + 1) & 1) * DA_BYTES : Acur), lds + 2 * DA_BYTES + bnext * DB_BYTES, 0)
        MFMA_2X2(af1, bf1)
      }
#undef ISSUE_HALO
#undef ISSUE_SLAB
      bcur = bnext;
    }
  }
#undef SET_TAP
#undef LOAD_FRAGS
#undef MFMA_2X2
#undef DMA_SEGMENT
#undef DMA_HALO_PIECE
#undef DMA_SLAB
#undef DMA_WAIT_BARRIER
  conv_epilogue<EPI, 2, 2>(P, acc, M, m0, n0, wm0, wn0, lane, HW);
}
#endif  // DH_ABLATION


// ---- 3x3, 128-cout tile, second form: weights by LDS-DMA, halo in 64-byte runs ----------------------------------------
// The ablation of conv3x3_halo_kernel says its fetch path is the bound (no halo fetch 7.2 -> 5.2 ms, no weight fetch
// 5.8 ms, neither 4.2 ms at 448->256 / 1024 edges) and that the cost goes with the 128-byte lines a wave load touches; a
// 64-byte run per halo pixel would halve the halo's lines but its staging registers do not fit next to the weight
// staging registers in the 128-VGPR budget of two workgroups per CU.  So here the WEIGHTS do not pass through registers
// at all: a (32-channel chunk, kernel row dy) group = 3 taps x 128 couts x 64 B = 24 KB arrives by
// `global_load_lds_dwordx4` into one of two LDS buffers while the previous group is being multiplied, and the halo is
// fetched 32 channels at a time (one 64-byte run per pixel, 16 registers) and staged once per chunk.  LDS rows are 64 B
// unpadded, 16-byte slots XOR-swizzled with bits 2..3 of the row index (conflict-free ds_read_b128; the weights are
// pre-packed in that order: [cout tile][chunk][dy][dx][128][4 slots][8]).  74.5 KB of LDS, two workgroups per CU, one
// barrier per 24 MFMAs per wave plus one per chunk for the halo hand-over; the epilogue is the LDS-staged one.
// Measured at 448->256, 1024 edges: 6.84 ms against 7.18-7.33 ms; without the halo fetch 5.64, without the weight DMA 5.96,
// without both 4.55 ms (the first halo kernel: 5.24 / 5.83 / 4.17) -- the fetch path costs 2.3 instead of 3.0 ms, the
// skeleton with its smaller steps 0.4 ms more; inside the full iteration 80.9 against 84.4 ms per step.  Those numbers are
// from the version whose loop still spilled (hoisted tap addresses, see the `opaque` notes below); spill-free it runs
// 448->256 in 5.95 ms (1.09 PFLOP/s) and the full iteration in 70.1 ms.
// ---- the ConvGRU's global-context reduction on a tile in LDS (used by glo_reduce_kernel below and by the q gate, EPI_GRU_Q + glo_red) ----
constexpr int GLD = 128 + 8;                          // LDS row stride (halves)

// the reduction proper on a tile [256 px][128 ch] fp16 that sits in LDS (row stride GLD) -- shared by glo_reduce_kernel (tile staged from
// HBM) and the q gate's fused form (round 6: the tile is the hidden state the gate has just written; ConvParams::glo_red).  Eight waves,
// wave tile 64 px x 64 couts; the weights [128][Kpad] come from L2 straight into B fragments (two halves of K); ends with ONE atomic
// per (workgroup, cout) into red_row[128].  Every thread of the workgroup has to have passed its last write of the tile's rows it owns;
// the function synchronises before the first fragment read.
__device__ __forceinline__ void glo_tile_reduce(const __half* __restrict__ wt, int Kpad, const float* __restrict__ bias_, float* __restrict__ red_row,
                                                __half* __restrict__ s_conv, int tid, half8 (&bf)[2][4]) {
  const int lane = tid & 63, wave = tid >> 6;
  const int wrow = wave & 3, wn0 = (wave >> 2) * 64, wm0 = wrow * 64;
  const int p = lane & 31, kh = lane >> 5;
  auto load_b = [&](int half) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        bf[b][ks] = *reinterpret_cast<const half8*>(wt + (long)(wn0 + b * 32 + p) * Kpad + (half * 4 + ks) * 16 + kh * 8);
  };
  f32x16 acc[2][2];
  zero_acc<2, 2>(acc);
  __syncthreads();
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      half8 af[2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
        af[a] = *reinterpret_cast<const half8*>(s_conv + (wm0 + a * 32 + p) * GLD + (half * 4 + ks) * 16 + kh * 8);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b][ks], acc[a][b], 0, 0, 0);
    }
    if (half == 0) load_b(1);
  }
  // gate * feature, summed over the wave's 64 pixels per cout (same roundings as conv_epilogue's EPI_GLO)
  float sum[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int co = wn0 + b * 32 + p;
    const float bias = bias_[co];
    float t = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 16; q += 2) {
        if ((q & 7) == 0) __builtin_amdgcn_sched_barrier(0);      // (keeps the 32 LDS reads from being hoisted in one block: spills)
        // two rows at a time: the gate values rounded to fp16 as a pair, the products as ONE packed fp16 multiply (the
        // exact product of two fp16 numbers has 22 significant bits, so rounding it from fp32 or inside v_pk_mul_f16 is the
        // same single rounding), their fp32 sum as one v_dot2 against (1, 1)
        const int row = wm0 + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
        const half2v g = {(_Float16)sigmoidf_(acc[a][b][q] + bias), (_Float16)sigmoidf_(acc[a][b][q + 1] + bias)};
        const half2v n = {__builtin_bit_cast(_Float16, s_conv[row * GLD + co]), __builtin_bit_cast(_Float16, s_conv[(row + 1) * GLD + co])};
        t = __builtin_amdgcn_fdot2(g * n, half2v{(_Float16)1.f, (_Float16)1.f}, t, false);
      }
    sum[b] = t + __shfl_xor(t, 32, 64);
  }
  __syncthreads();                                    // the tile is dead: its head becomes [8 waves][64] partial sums
  float* sR = reinterpret_cast<float*>(s_conv);
  if (lane < 32) { sR[wave * 64 + p] = sum[0]; sR[wave * 64 + 32 + p] = sum[1]; }
  __syncthreads();
  if (tid < 128) {
    const int hb = tid >> 6, c = tid & 63;            // cout half -> waves 4*hb .. 4*hb+3
    const float t = (sR[(hb * 4 + 0) * 64 + c] + sR[(hb * 4 + 1) * 64 + c]) + (sR[(hb * 4 + 2) * 64 + c] + sR[(hb * 4 + 3) * 64 + c]);
    atomicAdd(&red_row[tid], t);
  }
}
// first half of K of the wave's B fragments (issued before the tile is complete, so that they arrive under the staging)
__device__ __forceinline__ void glo_load_b0(const __half* __restrict__ wt, int Kpad, int tid, half8 (&bf)[2][4]) {
  const int lane = tid & 63, wn0 = (tid >> 8) * 64, p = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      bf[b][ks] = *reinterpret_cast<const half8*>(wt + (long)(wn0 + b * 32 + p) * Kpad + ks * 16 + kh * 8);
}

constexpr int H2CK = 32;
constexpr int H2A_BYTES = HPIX * H2CK * 2;               // 25,344: halo [396 px][32 ch]
constexpr int H2B_BYTES = 3 * 128 * H2CK * 2;            // 24,576: one (chunk, dy) weight group
constexpr int H2_LDS_BYTES = H2A_BYTES + 2 * H2B_BYTES;  // 74,496 (the staged epilogue tile needs 69,632)

// ABL (only instantiated != 0 in -DDH_ABLATION builds, option "conv_abl", WRONG results by construction): timing / power
// attribution of the main loop.  bit 0: fragment reads only in the first step (operands stay in registers); bit 1: no weight
// DMA after the first two groups; bit 2: no halo fetch after the first chunk; bit 3: no epilogue; bit 4: no MFMAs.
// TWO (round 6, CINIT == 0 only; option conv_two_tiles): a workgroup computes TWO vertically adjacent pixel tiles one after the other.  The layers
// with 128 input channels (corr_encoder.2, agg.conv1, the heads' first layer) run 12 steps between a 4.8 us prologue and a 5.3 / 8.3 us
// epilogue (profiles/r05_n_conv_phase_timeline.txt: 1.3-1.5 of the CU's two workgroups in their main loop): here the second tile's halo
// fetch is issued BEFORE the first tile's epilogue (three 16-byte loads per thread that land under it), its first weight group as soon as the
// staged tile has been read, and the workgroup re-enters the main loop without a dispatch in between.  Same MFMAs per tile: same results.
template <int EPI, bool STAGED = true, int CINIT = 0, int ABL = 0, bool GLO = false, bool TWO = false>      // CINIT: 0 zero, 1 pixel-major start values, 2 start values in the accumulator-tile
                                                                         // layout, 3 (STAGED = false) zero start, OUTPUT in that layout (out_f32 == 3)
                                                                         // GLO (EPI_GRU_Q): the next iteration's global-context reduction behind the state update
__global__ __launch_bounds__(512, 4) void conv3x3_halo2_kernel(ConvParams P) {
  extern __shared__ __half s_conv[];
  DH_CTS(0); DH_CTS_ID();
  char* const lds = reinterpret_cast<char*>(s_conv);
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)s_conv;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wrow = wave & 3, wn0 = (wave >> 2) * 64, wm0 = wrow * 64;
  long m0; int n0;
  if (!xcd_decode(P, m0, n0, 128, TWO ? 2 * BM : BM)) return;
  const int HW = P.H * P.W;
  int img = (int)(m0 / HW), y0 = (int)((m0 - (long)img * HW) / 64);
  const int nchunks = P.Ctot / H2CK, nsteps = nchunks * 3;

  // halo roles.  The first and last column of the 6 x 66 halo lie outside the 64-pixel-wide image for every tile: they are
  // zeroed once, and the 6 x 64 interior is exactly 3 pieces per thread: piece id = tid + 512*i -> (row 2i + (tid >> 8),
  // column (tid >> 2) & 63, 16-byte slot tid & 3 of the pixel's 64-byte run)
  constexpr int A_PIECES = 3;
  int aq = tid & 3, ax = (tid >> 2) & 63, ahy = tid >> 8;
  // (offsets are recomputed from these three where they are used: the kernel sits at its 128-register budget, and a
  // spilled address register is reloaded with a vmcnt(0) wait that serialises the halo loads)
  int a_pix0 = (img * P.H + y0 - 1 + ahy) * 64 + ax;      // image pixel of piece 0; piece i is two rows further down
  if (tid < 48) {                                        // 12 edge pixels x 4 slots
    const int hp = (tid >> 3) * HCOLS + ((tid >> 2) & 1) * (HCOLS - 1);
    *reinterpret_cast<uint4*>(lds + hp * 64 + aq * 16) = uint4{0u, 0u, 0u, 0u};
  }
  // weight DMA: per-lane offset (one register for every piece) + wave-uniform base pointer in SGPRs
  int b_voff = lane * 16;
  const char* const bsrc = reinterpret_cast<const char*>(P.wt_halo) + (long)(n0 >> 7) * nsteps * H2B_BYTES;

  // fragment addressing: 16-byte slot s of row r sits at r*64 + ((s ^ ((r >> 2) & 3)) << 4)
  int pl = wrow * HCOLS + (lane & 31), kh = lane >> 5;
  int b_row[2], b_x[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int r = wn0 + b * 32 + (lane & 31);
    b_row[b] = r * 64; b_x[b] = ((r >> 2) & 3) ^ kh;
  }

  f32x16 acc[2][2];
  if constexpr (CINIT == 2) init_acc_tile_tiled(P, acc, m0, n0, wave, lane, HW);
  else if constexpr (CINIT == 1) init_acc_tile<2, 2>(P, acc, m0, n0, wm0, wn0, lane, HW);
  else zero_acc<2, 2>(acc);

  u32x4 ra[A_PIECES];
#define H2_FETCH_A(chunk_)                                                                                           \
  {                                                                                                                  \
    int cs = (chunk_) * H2CK, sgi = 0;                                                                               \
    _Pragma("unroll") for (int q = 0; q < MAXSEG - 1; ++q)                                                           \
      if (sgi == q && q + 1 < P.nseg && cs >= P.segC[q]) { cs -= P.segC[q]; sgi = q + 1; }                          \
    const __half* base = P.in[0]; int segs = P.segS[0];                                                              \
    _Pragma("unroll") for (int q = 1; q < MAXSEG; ++q) if (sgi == q) { base = P.in[q]; segs = P.segS[q]; }           \
    int pix0 = a_pix0;                                                                                               \
    asm volatile("" : "+v"(pix0));          /* opaque: hipcc would hoist the three offsets out of the loop and spill them */ \
    _Pragma("unroll") for (int i = 0; i < A_PIECES; ++i) {    /* raw loads: masking them here would wait for them here */ \
      const bool ok = (unsigned)(y0 - 1 + ahy + 2 * i) < (unsigned)P.H;                                              \
      ra[i] = *reinterpret_cast<const u32x4*>(base + (long)(ok ? pix0 + 128 * i : 0) * segs + cs + aq * 8);          \
    }                                                                                                                \
  }
  // one wave-wide LDS-DMA with a uniform base: lane l copies 16 bytes from sbase_ + voff_ (per lane) to LDS ldsaddr_ + l*16
#define GLDS16S(sbase_, voff_, ldsaddr_)                                                                             \
  {                                                                                                                  \
    unsigned keep_;                                                                                                  \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(voff_), "s"(sbase_), "s"(ldsaddr_) : "memory");                                \
  }
#define H2_DMA_B(step_)                                                                                              \
  {                                                                                                                  \
    _Pragma("unroll") for (int q = 0; q < 3; ++q) {                                                                  \
      const unsigned long gaddr = (unsigned long)(bsrc + (long)(step_) * H2B_BYTES + (wave + 8 * q) * 1024);          \
      const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)gaddr);                                          \
      const unsigned ghi = __builtin_amdgcn_readfirstlane((unsigned)(gaddr >> 32));                                  \
      const void* gs = reinterpret_cast<const void*>(((unsigned long)ghi << 32) | glo);                              \
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + H2A_BYTES + ((step_) & 1) * H2B_BYTES + (wave + 8 * q) * 1024); \
      GLDS16S(gs, b_voff, dst)                                                                                       \
    }                                                                                                                \
  }

  half8 abl_af[2] = {}, abl_bf[2] = {};                  // (ABL != 0 only: operands that outlive a step)
  H2_FETCH_A(0)
  H2_DMA_B(0)
  DH_CTS(1);
  for (int tile2 = 0; tile2 < (TWO ? 2 : 1); ++tile2) {
  for (int c = 0; c < nchunks; ++c) {
    // every wave has finished the previous chunk's reads of the halo tile
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int ahy_o = ahy, ax_o = ax;
    asm volatile("" : "+v"(ahy_o), "+v"(ax_o));         // opaque (see H2_FETCH_A): keep the offset arithmetic inside the loop
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
      const int hp = (2 * i + ahy_o) * HCOLS + ax_o + 1;
      const uint32_t m = (unsigned)(y0 - 1 + ahy_o + 2 * i) < (unsigned)P.H ? 0xffffffffu : 0u;  // rows outside the image are zero
      *reinterpret_cast<u32x4*>(lds + hp * 64 + ((aq ^ ((hp >> 2) & 3)) << 4)) = ra[i] & m;
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int step = c * 3 + dy;
      // this wave's DMA pieces of group `step` (issued a step ago) have landed, its halo stores are done; then the barrier.
      // At dy == 1 the halo loads of the next chunk, issued AFTER those pieces one step ago, may stay in flight (vector
      // memory operations complete in order): they get two steps to land instead of one.
      if (dy == 1 && c + 1 < nchunks && !(ABL & 4)) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (step == 0) DH_CTS(2);
      DH_CTS_STEP(step);
      if (step + 1 < nsteps && (!(ABL & 2) || step < 1)) H2_DMA_B(step + 1)           // into the buffer last read in step - 1
      if (dy == 0 && c + 1 < nchunks && !(ABL & 4)) H2_FETCH_A(c + 1)
      const char* const Bcur = lds + H2A_BYTES + (step & 1) * H2B_BYTES;
      int pl_o = pl;
      asm volatile("" : "+v"(pl_o));                      // opaque: the 18 tap addresses are cheap to form, expensive to keep
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        int a_row[2], a_x[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int p = pl_o + dy * HCOLS + dx + a * 32;
          a_row[a] = p * 64; a_x[a] = ((p >> 2) & 3) ^ kh;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          if constexpr (ABL == 0) {
            half8 af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = *reinterpret_cast<const half8*>(lds + a_row[a] + (((ks * 2) ^ a_x[a]) << 4));
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = *reinterpret_cast<const half8*>(Bcur + dx * (128 * 64) + b_row[b] + (((ks * 2) ^ b_x[b]) << 4));
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
          } else {
            if (!(ABL & 1) || step == 0) {                  // (uniform branch: the reads of step 0 stay, the rest reuse the registers)
#pragma unroll
              for (int a = 0; a < 2; ++a) abl_af[a] = *reinterpret_cast<const half8*>(lds + a_row[a] + (((ks * 2) ^ a_x[a]) << 4));
#pragma unroll
              for (int b = 0; b < 2; ++b) abl_bf[b] = *reinterpret_cast<const half8*>(Bcur + dx * (128 * 64) + b_row[b] + (((ks * 2) ^ b_x[b]) << 4));
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int b = 0; b < 2; ++b) {
                if (ABL & 16) asm volatile("" : "+v"(acc[a][b]) : "v"(abl_af[a]), "v"(abl_bf[b]));       // operands stay live, no matrix instruction
                else acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(abl_af[a], abl_bf[b], acc[a][b], 0, 0, 0);
              }
          }
        }
      }
    }
  }
  if constexpr (TWO) {
    // second tile: its first halo chunk is requested now and lands under the first tile's epilogue
    const long m0_done = m0;
    const bool more = tile2 == 0 && m0 + BM < (long)P.N * HW;
    if (more) {
      m0 += BM;
      img = (int)(m0 / HW); y0 = (int)((m0 - (long)img * HW) / 64);
      a_pix0 = (img * P.H + y0 - 1 + ahy) * 64 + ax;
      H2_FETCH_A(0)
    }
    if (tile2 == 0) DH_CTS(3);
    int tid_e = tid;                                     // opaque: the epilogue now sits in a loop, and its per-thread addresses hoisted out of
    asm volatile("" : "+v"(tid_e));                      // that loop as invariants are kept across the main loop -- in scratch
    if constexpr (EPI == EPI_HEADS0) staged_heads0_epilogue(P, acc, s_conv, m0_done, n0, wm0, wn0, tid_e);
    else staged_epilogue<EPI, 2, 128>(P, acc, s_conv, m0_done, n0, wm0, wn0, tid_e, HW);
    if (!more) break;
    __syncthreads();                                     // the staged tile has been read by everybody: the operand buffers are free again
    {
      // the per-lane constants of the main loop are formed AGAIN from an opaque copy of the thread index: kept alive across the
      // epilogue (a dozen registers next to the accumulators, the three halo pieces in flight and the epilogue's own pieces) they spill
      int t_o = tid;
      asm volatile("" : "+v"(t_o));
      const int l_o = t_o & 63;
      aq = t_o & 3; ax = (t_o >> 2) & 63; ahy = t_o >> 8; b_voff = l_o * 16;
      pl = wrow * HCOLS + (l_o & 31); kh = l_o >> 5;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int r = wn0 + b * 32 + (l_o & 31);
        b_row[b] = r * 64; b_x[b] = ((r >> 2) & 3) ^ kh;
      }
      if (t_o < 48) {                                    // (the staged tile lay over the halo's zero columns)
        const int hp = (t_o >> 3) * HCOLS + ((t_o >> 2) & 1) * (HCOLS - 1);
        *reinterpret_cast<uint4*>(lds + hp * 64 + aq * 16) = uint4{0u, 0u, 0u, 0u};
      }
    }
    H2_DMA_B(0)
    zero_acc<2, 2>(acc);
  }
  }
#undef H2_FETCH_A
#undef H2_DMA_B
#undef GLDS16S
  if constexpr (TWO) { DH_CTS(4); return; }
  DH_CTS(3);
  if constexpr ((ABL & 8) != 0) {                        // no epilogue: one never-taken store keeps the accumulators live
    float sum = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) sum += acc[a][b][q];
    if (sum == 1.2345678e30f) reinterpret_cast<__half*>(P.out)[0] = __float2half(sum);
    return;
  }
  if constexpr (EPI == EPI_HEADS0) staged_heads0_epilogue(P, acc, s_conv, m0, n0, wm0, wn0, tid);
  else if constexpr (GLO) {
    // the staged tile [256 px][128 couts] ends up holding the NEW hidden state (every thread writes its finished pieces back), and the
    // tail of glo_reduce_kernel runs on it in place: one 128 x 128 GEMM per tile (1 / 22.5 of the gate's own MFMA work), 64 sigmoids per
    // lane on the vector pipe under the other resident workgroup's matrix work, one atomic per (workgroup, channel)
    staged_epilogue<EPI, 2, 128, 512, true>(P, acc, s_conv, m0, n0, wm0, wn0, tid, HW);
    half8 bf[2][4];                    // (requested only now: 32 more registers during the state update would spill)
    glo_load_b0(P.glo_wt, 128, tid, bf);
    glo_tile_reduce(P.glo_wt, 128, P.glo_bias, P.glo_red + (long)img * 128, s_conv, tid, bf);
  }
  else if constexpr (STAGED) staged_epilogue<EPI, 2, 128>(P, acc, s_conv, m0, n0, wm0, wn0, tid, HW);
  else if constexpr (CINIT == 3) store_acc_tile(P, acc, m0, n0, wn0, wave, lane);      // the context term for the CINIT == 2 launches
  else conv_epilogue<EPI, 2, 2>(P, acc, (long)P.N * HW, m0, n0, wm0, wn0, lane, HW);
  DH_CTS(4);
}

#ifdef DH_ABLATION   // four-wave form of the second kernel (measurement variant, option conv_halo4)
// ---- 3x3, 128-cout tile, the second form with FOUR waves of 64 x 128 instead of eight of 64 x 64 --------------------------------
// Same tile (4 image rows x 64 columns x 128 couts), same LDS image and weight layout, still two workgroups per CU; a wave owns one
// image row and all 128 couts: 2 x 4 accumulator tiles (128 registers), an activation fragment read from LDS feeds four MFMAs
// (0.75 fragment reads per MFMA instead of 1), two waves per SIMD instead of four.
template <int EPI, bool STAGED = true, bool CINIT = false>
__global__ __launch_bounds__(256, 2) void conv3x3_halo4_kernel(ConvParams P) {
  extern __shared__ __half s_conv[];
  char* const lds = reinterpret_cast<char*>(s_conv);
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)s_conv;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wrow = wave, wn0 = 0, wm0 = wrow * 64;
  long m0; int n0;
  if (!xcd_decode(P, m0, n0, 128)) return;
  const int HW = P.H * P.W;
  const int img = (int)(m0 / HW), y0 = (int)((m0 - (long)img * HW) / 64);
  const int nchunks = P.Ctot / H2CK, nsteps = nchunks * 3;
  // halo: 6 x 64 interior = 1536 pieces = 6 per thread: piece tid + 256 i -> (row i, column (tid >> 2) & 63, slot tid & 3)
  constexpr int A_PIECES = 6;
  const int aq = tid & 3, ax = (tid >> 2) & 63;
  const int a_pix0 = (img * P.H + y0 - 1) * 64 + ax;
  if (tid < 48) {
    const int hp = (tid >> 3) * HCOLS + ((tid >> 2) & 1) * (HCOLS - 1);
    *reinterpret_cast<uint4*>(lds + hp * 64 + aq * 16) = uint4{0u, 0u, 0u, 0u};
  }
  const int b_voff = lane * 16;
  const char* const bsrc = reinterpret_cast<const char*>(P.wt_halo) + (long)(n0 >> 7) * nsteps * H2B_BYTES;
  const int pl = wrow * HCOLS + (lane & 31), kh = lane >> 5;
  int b_row[4], b_x[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int r = b * 32 + (lane & 31);
    b_row[b] = r * 64; b_x[b] = ((r >> 2) & 3) ^ kh;
  }
  f32x16 acc[2][4];
  if constexpr (CINIT) init_acc_tile<2, 4>(P, acc, m0, n0, wm0, wn0, lane, HW);
  else zero_acc<2, 4>(acc);
  u32x4 ra[A_PIECES];
#define H4_FETCH_A(chunk_)                                                                                           \
  {                                                                                                                  \
    int cs = (chunk_) * H2CK, sgi = 0;                                                                               \
    _Pragma("unroll") for (int q = 0; q < MAXSEG - 1; ++q)                                                           \
      if (sgi == q && q + 1 < P.nseg && cs >= P.segC[q]) { cs -= P.segC[q]; sgi = q + 1; }                          \
    const __half* base = P.in[0]; int segs = P.segS[0];                                                              \
    _Pragma("unroll") for (int q = 1; q < MAXSEG; ++q) if (sgi == q) { base = P.in[q]; segs = P.segS[q]; }           \
    int pix0 = a_pix0;                                                                                               \
    asm volatile("" : "+v"(pix0));                                                                                   \
    _Pragma("unroll") for (int i = 0; i < A_PIECES; ++i) {                                                           \
      const bool ok = (unsigned)(y0 - 1 + i) < (unsigned)P.H;                                                        \
      ra[i] = *reinterpret_cast<const u32x4*>(base + (long)(ok ? pix0 + 64 * i : 0) * segs + cs + aq * 8);           \
    }                                                                                                                \
  }
#define GLDS16S(sbase_, voff_, ldsaddr_)                                                                             \
  {                                                                                                                  \
    unsigned keep_;                                                                                                  \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(voff_), "s"(sbase_), "s"(ldsaddr_) : "memory");                                \
  }
#define H4_DMA_B(step_)                                                                                              \
  {                                                                                                                  \
    _Pragma("unroll") for (int q = 0; q < 6; ++q) {                                                                  \
      const unsigned long gaddr = (unsigned long)(bsrc + (long)(step_) * H2B_BYTES + (wave + 4 * q) * 1024);          \
      const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)gaddr);                                          \
      const unsigned ghi = __builtin_amdgcn_readfirstlane((unsigned)(gaddr >> 32));                                  \
      const void* gs = reinterpret_cast<const void*>(((unsigned long)ghi << 32) | glo);                              \
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + H2A_BYTES + ((step_) & 1) * H2B_BYTES + (wave + 4 * q) * 1024); \
      GLDS16S(gs, b_voff, dst)                                                                                       \
    }                                                                                                                \
  }
  H4_FETCH_A(0)
  H4_DMA_B(0)
  for (int c = 0; c < nchunks; ++c) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int ax_o = ax;
    asm volatile("" : "+v"(ax_o));
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
      const int hp = i * HCOLS + ax_o + 1;
      const uint32_t m = (unsigned)(y0 - 1 + i) < (unsigned)P.H ? 0xffffffffu : 0u;
      *reinterpret_cast<u32x4*>(lds + hp * 64 + ((aq ^ ((hp >> 2) & 3)) << 4)) = ra[i] & m;
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int step = c * 3 + dy;
      if (dy == 1 && c + 1 < nchunks) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (step + 1 < nsteps) H4_DMA_B(step + 1)
      if (dy == 0 && c + 1 < nchunks) H4_FETCH_A(c + 1)
      const char* const Bcur = lds + H2A_BYTES + (step & 1) * H2B_BYTES;
      int pl_o = pl;
      asm volatile("" : "+v"(pl_o));
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          half8 af[2], bf[4];
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const int p = pl_o + dy * HCOLS + dx + a * 32;
            af[a] = *reinterpret_cast<const half8*>(lds + p * 64 + (((ks * 2) ^ ((p >> 2) & 3) ^ kh) << 4));
          }
#pragma unroll
          for (int b = 0; b < 4; ++b) bf[b] = *reinterpret_cast<const half8*>(Bcur + dx * (128 * 64) + b_row[b] + (((ks * 2) ^ b_x[b]) << 4));
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
      }
    }
  }
#undef H4_FETCH_A
#undef H4_DMA_B
#undef GLDS16S
  if constexpr (STAGED) staged_epilogue<EPI, 4, 128, 256>(P, acc, s_conv, m0, n0, wm0, wn0, tid, HW);
  else conv_epilogue<EPI, 2, 4>(P, acc, (long)P.N * HW, m0, n0, wm0, wn0, lane, HW);
}

#endif  // DH_ABLATION

// ---- 3x3, 64-cout tile in the SECOND form (option conv_halo64, on by default since round 5): the flow encoder's 128 -> 64 layer -------
// That layer ran in the first-generation halo kernel (conv3x3_halo_kernel<.., 64>: eight waves of 64 px x 32 couts, weights staged
// through registers) at 0.75-0.90 PFLOP/s where the 128-cout layers reach 1.0-1.1 on the same operands; measured at 4096 edges on one
// box (profiles/r05_b_conv_halo64_ab.txt): 2.464 ms -> 1.827 ms (0.30 -> 0.41 of the fp16 MFMA peak).  This is conv3x3_halo4_kernel
// with half the couts: four waves, a wave owns one image row and all 64 couts (2 x 2 accumulator tiles, the production wave tile), the
// (32-channel chunk, kernel row) weight groups arrive by LDS-DMA into two 12-KB buffers -- 50 KB of LDS, three workgroups per CU.
// Weights: the conv3x3_halo2_kernel layout of the layer padded to 128 couts (droid_amd.update.pack_conv_halo under this option); a
// (chunk, dy) group there is 3 x 128 rows of 64 bytes, of which this kernel fetches rows 0 .. 63 of every dx.
constexpr int H64B_BYTES = 3 * 64 * H2CK * 2;                 // 12,288: one (chunk, dy) weight group, 64 couts
constexpr int H64_LDS_BYTES = H2A_BYTES + 2 * H64B_BYTES;     // 49,920 (the staged epilogue tile needs 36,864)
// Round 6 (option conv_gate64, off by default): the SAME kernel on 64-cout tiles of a layer with CoutPad % 128 == 0 -- the ConvGRU gates
// (K = 320, 256 / 128 couts: four / two cout tiles per pixel tile, visited back to back on one XCD so that the halo re-reads are L2 hits)
// with the GRU epilogues and the accumulator-tile start values (CINIT == 2), THREE workgroups per CU instead of conv3x3_halo2_kernel's
// two: tile n0 fetches rows (n0 & 64) .. + 63 of every dx of cout tile n0 >> 7 of the halo2 weight layout, and its wave `wrow` holds
// exactly the registers of conv3x3_halo2_kernel's wave wrow + 4 * ((n0 >> 6) & 1) -- same MFMAs in the same k order: bit-identical.
template <int EPI, int CINIT = 0>
__global__ __launch_bounds__(256, 3) void conv3x3_halo64_kernel(ConvParams P) {
  extern __shared__ __half s_conv[];
  DH_CTS(0); DH_CTS_ID();
  char* const lds = reinterpret_cast<char*>(s_conv);
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)s_conv;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wrow = wave, wn0 = 0, wm0 = wrow * 64;
  long m0; int n0;
  if (!xcd_decode(P, m0, n0, 64)) return;
  const int HW = P.H * P.W;
  const int img = (int)(m0 / HW), y0 = (int)((m0 - (long)img * HW) / 64);
  const int nchunks = P.Ctot / H2CK, nsteps = nchunks * 3;
  constexpr int A_PIECES = 6;
  const int aq = tid & 3, ax = (tid >> 2) & 63;
  const int a_pix0 = (img * P.H + y0 - 1) * 64 + ax;
  if (tid < 48) {
    const int hp = (tid >> 3) * HCOLS + ((tid >> 2) & 1) * (HCOLS - 1);
    *reinterpret_cast<uint4*>(lds + hp * 64 + aq * 16) = uint4{0u, 0u, 0u, 0u};
  }
  const int b_voff = lane * 16;
  // 128-cout tile n0 >> 7 of the halo2 layout, rows (n0 & 64) .. + 63 of every dx (64 rows of 64 bytes = 4096)
  const char* const bsrc = reinterpret_cast<const char*>(P.wt_halo) + (long)(n0 >> 7) * nsteps * H2B_BYTES + ((n0 >> 6) & 1) * 4096;
  const int pl = wrow * HCOLS + (lane & 31), kh = lane >> 5;
  int b_row[2], b_x[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int r = b * 32 + (lane & 31);
    b_row[b] = r * 64; b_x[b] = ((r >> 2) & 3) ^ kh;
  }
  f32x16 acc[2][2];
  if constexpr (CINIT == 2) init_acc_tile_tiled(P, acc, m0, n0, wave + 4 * ((n0 >> 6) & 1), lane, HW);
  else zero_acc<2, 2>(acc);
  u32x4 ra[A_PIECES];
#define H64_FETCH_A(chunk_)                                                                                          \
  {                                                                                                                  \
    int cs = (chunk_) * H2CK, sgi = 0;                                                                               \
    _Pragma("unroll") for (int q = 0; q < MAXSEG - 1; ++q)                                                           \
      if (sgi == q && q + 1 < P.nseg && cs >= P.segC[q]) { cs -= P.segC[q]; sgi = q + 1; }                          \
    const __half* base = P.in[0]; int segs = P.segS[0];                                                              \
    _Pragma("unroll") for (int q = 1; q < MAXSEG; ++q) if (sgi == q) { base = P.in[q]; segs = P.segS[q]; }           \
    int pix0 = a_pix0;                                                                                               \
    asm volatile("" : "+v"(pix0));                                                                                   \
    _Pragma("unroll") for (int i = 0; i < A_PIECES; ++i) {                                                           \
      const bool ok = (unsigned)(y0 - 1 + i) < (unsigned)P.H;                                                        \
      ra[i] = *reinterpret_cast<const u32x4*>(base + (long)(ok ? pix0 + 64 * i : 0) * segs + cs + aq * 8);           \
    }                                                                                                                \
  }
#define GLDS16S(sbase_, voff_, ldsaddr_)                                                                             \
  {                                                                                                                  \
    unsigned keep_;                                                                                                  \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(voff_), "s"(sbase_), "s"(ldsaddr_) : "memory");                                \
  }
  // 12 pieces of 1 KB per group: piece p = wave + 4 q -> dx = p >> 2, rows 16 (p & 3) .. of that dx
#define H64_DMA_B(step_)                                                                                             \
  {                                                                                                                  \
    _Pragma("unroll") for (int q = 0; q < 3; ++q) {                                                                  \
      const int pc = wave + 4 * q;                                                                                   \
      const unsigned long gaddr = (unsigned long)(bsrc + (long)(step_) * H2B_BYTES + (pc >> 2) * 8192 + (pc & 3) * 1024); \
      const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)gaddr);                                          \
      const unsigned ghi = __builtin_amdgcn_readfirstlane((unsigned)(gaddr >> 32));                                  \
      const void* gs = reinterpret_cast<const void*>(((unsigned long)ghi << 32) | glo);                              \
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + H2A_BYTES + ((step_) & 1) * H64B_BYTES + (pc >> 2) * 4096 + (pc & 3) * 1024); \
      GLDS16S(gs, b_voff, dst)                                                                                       \
    }                                                                                                                \
  }
  H64_FETCH_A(0)
  H64_DMA_B(0)
  DH_CTS(1);
  for (int c = 0; c < nchunks; ++c) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int ax_o = ax;
    asm volatile("" : "+v"(ax_o));
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
      const int hp = i * HCOLS + ax_o + 1;
      const uint32_t m = (unsigned)(y0 - 1 + i) < (unsigned)P.H ? 0xffffffffu : 0u;
      *reinterpret_cast<u32x4*>(lds + hp * 64 + ((aq ^ ((hp >> 2) & 3)) << 4)) = ra[i] & m;
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int step = c * 3 + dy;
      // (vmcnt: the group of this step was requested one step ago; at dy == 1 the six halo loads of the next chunk are younger)
      if (dy == 1 && c + 1 < nchunks) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (step == 0) DH_CTS(2);
      DH_CTS_STEP(step);
      if (step + 1 < nsteps) H64_DMA_B(step + 1)
      if (dy == 0 && c + 1 < nchunks) H64_FETCH_A(c + 1)
      const char* const Bcur = lds + H2A_BYTES + (step & 1) * H64B_BYTES;
      int pl_o = pl;
      asm volatile("" : "+v"(pl_o));
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          half8 af[2], bf[2];
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const int p = pl_o + dy * HCOLS + dx + a * 32;
            af[a] = *reinterpret_cast<const half8*>(lds + p * 64 + (((ks * 2) ^ ((p >> 2) & 3) ^ kh) << 4));
          }
#pragma unroll
          for (int b = 0; b < 2; ++b) bf[b] = *reinterpret_cast<const half8*>(Bcur + dx * (64 * 64) + b_row[b] + (((ks * 2) ^ b_x[b]) << 4));
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
      }
    }
  }
#undef H64_FETCH_A
#undef H64_DMA_B
#undef GLDS16S
  DH_CTS(3);
  staged_epilogue<EPI, 2, 64, 256>(P, acc, s_conv, m0, n0, wm0, wn0, tid, HW);
  DH_CTS(4);
}

#ifdef DH_ABLATION   // 512-pixel-tile form: measured equal in energy, slower in time (profiles/r04_d_conv_halo3_ab.json): not in release builds
// ---- 3x3, 128-cout tile, THIRD form: 512-pixel tile (8 image rows), 128 x 64 per wave -- opt-in (option conv_halo3) -----------
// Round 4's power measurements (profiles/r04_conv_power.json) put conv3x3_halo2_kernel on the socket's 1400 W cap with 31 % of the
// dynamic energy outside the matrix cores: 11 % fragment reads from LDS, 12 % weight DMA + halo fetch, 5 % epilogue.  Under the cap
// time is energy, so this form spends fewer bytes per flop with the SAME data flow (weights of a (32-channel chunk, kernel row) group
// by LDS-DMA into two buffers, halo staged once per chunk through registers, same weight layout, same swizzles, same epilogues):
//   * a wave owns image rows wrow and wrow + 4 of the 8-row tile x 64 couts = 4 x 2 accumulator tiles (128 registers): a weight
//     fragment read from LDS feeds four MFMAs instead of two -- 0.75 fragment reads per MFMA instead of 1;
//   * one weight group per 512 pixels instead of per 256: half the weight stream per pixel; halo 10 rows for 8 (1.25x) instead
//     of 6 for 4 (1.5x).
// Price: 128 accumulators + staging = ~200 registers -> ONE workgroup per CU (8 waves, 2 per SIMD, 91 KB of LDS): nothing overlaps
// this workgroup's prologue / epilogue.  Needs H % 8 == 0.  The two 256-pixel halves go through the epilogues of the second form
// one after the other.
constexpr int H3ROWS = 10, H3PIX = H3ROWS * HCOLS;
constexpr int H3A_BYTES = H3PIX * H2CK * 2;              // 42,240: halo [660 px][32 ch]
constexpr int H3_LDS_BYTES = H3A_BYTES + 2 * H2B_BYTES;  // 91,392 (the staged epilogue tile needs 69,632)

template <int EPI, bool STAGED = true, bool CINIT = false>
__global__ __launch_bounds__(512, 2) void conv3x3_halo3_kernel(ConvParams P) {
  extern __shared__ __half s_conv[];
  DH_CTS(0); DH_CTS_ID();
  char* const lds = reinterpret_cast<char*>(s_conv);
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)s_conv;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wrow = wave & 3, wn0 = (wave >> 2) * 64, wm0 = wrow * 64;
  long m0; int n0;
  if (!xcd_decode(P, m0, n0, 128, 512)) return;
  const int HW = P.H * P.W;
  const int img = (int)(m0 / HW), y0 = (int)((m0 - (long)img * HW) / 64);
  const int nchunks = P.Ctot / H2CK, nsteps = nchunks * 3;

  // halo: the 10 x 64 interior is exactly 5 pieces per thread: piece id = tid + 512*i -> (row 2i + (tid >> 8), column (tid >> 2) & 63,
  // 16-byte slot tid & 3); the first and last column (outside the 64-pixel-wide image for every tile) are zeroed once
  constexpr int A_PIECES = 5;
  const int aq = tid & 3, ax = (tid >> 2) & 63, ahy = tid >> 8;
  const int a_pix0 = (img * P.H + y0 - 1 + ahy) * 64 + ax;
  if (tid < 8 * H3ROWS) {                                // 20 edge pixels x 4 slots
    const int hp = (tid >> 3) * HCOLS + ((tid >> 2) & 1) * (HCOLS - 1);
    *reinterpret_cast<uint4*>(lds + hp * 64 + aq * 16) = uint4{0u, 0u, 0u, 0u};
  }
  const int b_voff = lane * 16;
  const char* const bsrc = reinterpret_cast<const char*>(P.wt_halo) + (long)(n0 >> 7) * nsteps * H2B_BYTES;

  const int pl = wrow * HCOLS + (lane & 31), kh = lane >> 5;
  int b_row[2], b_x[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int r = wn0 + b * 32 + (lane & 31);
    b_row[b] = r * 64; b_x[b] = ((r >> 2) & 3) ^ kh;
  }

  f32x16 acc[2][2][2];                                   // [tile half: image rows wrow / wrow + 4][32-pixel half of the row][32-cout half]
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if constexpr (CINIT) init_acc_tile<2, 2>(P, acc[h], m0 + h * 256, n0, wm0, wn0, lane, HW);
    else zero_acc<2, 2>(acc[h]);
  }

  u32x4 ra[A_PIECES];
#define H3_FETCH_A(chunk_)                                                                                           \
  {                                                                                                                  \
    int cs = (chunk_) * H2CK, sgi = 0;                                                                               \
    _Pragma("unroll") for (int q = 0; q < MAXSEG - 1; ++q)                                                           \
      if (sgi == q && q + 1 < P.nseg && cs >= P.segC[q]) { cs -= P.segC[q]; sgi = q + 1; }                          \
    const __half* base = P.in[0]; int segs = P.segS[0];                                                              \
    _Pragma("unroll") for (int q = 1; q < MAXSEG; ++q) if (sgi == q) { base = P.in[q]; segs = P.segS[q]; }           \
    int pix0 = a_pix0;                                                                                               \
    asm volatile("" : "+v"(pix0));                                                                                   \
    _Pragma("unroll") for (int i = 0; i < A_PIECES; ++i) {                                                           \
      const bool ok = (unsigned)(y0 - 1 + ahy + 2 * i) < (unsigned)P.H;                                              \
      ra[i] = *reinterpret_cast<const u32x4*>(base + (long)(ok ? pix0 + 128 * i : 0) * segs + cs + aq * 8);          \
    }                                                                                                                \
  }
#define GLDS16S(sbase_, voff_, ldsaddr_)                                                                             \
  {                                                                                                                  \
    unsigned keep_;                                                                                                  \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(voff_), "s"(sbase_), "s"(ldsaddr_) : "memory");                                \
  }
#define H3_DMA_B(step_)                                                                                              \
  {                                                                                                                  \
    _Pragma("unroll") for (int q = 0; q < 3; ++q) {                                                                  \
      const unsigned long gaddr = (unsigned long)(bsrc + (long)(step_) * H2B_BYTES + (wave + 8 * q) * 1024);          \
      const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)gaddr);                                          \
      const unsigned ghi = __builtin_amdgcn_readfirstlane((unsigned)(gaddr >> 32));                                  \
      const void* gs = reinterpret_cast<const void*>(((unsigned long)ghi << 32) | glo);                              \
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + H3A_BYTES + ((step_) & 1) * H2B_BYTES + (wave + 8 * q) * 1024); \
      GLDS16S(gs, b_voff, dst)                                                                                       \
    }                                                                                                                \
  }

  H3_FETCH_A(0)
  H3_DMA_B(0)
  DH_CTS(1);
  for (int c = 0; c < nchunks; ++c) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave has finished the previous chunk's reads of the halo tile
    int ahy_o = ahy, ax_o = ax;
    asm volatile("" : "+v"(ahy_o), "+v"(ax_o));
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
      const int hp = (2 * i + ahy_o) * HCOLS + ax_o + 1;
      const uint32_t m = (unsigned)(y0 - 1 + ahy_o + 2 * i) < (unsigned)P.H ? 0xffffffffu : 0u;
      *reinterpret_cast<u32x4*>(lds + hp * 64 + ((aq ^ ((hp >> 2) & 3)) << 4)) = ra[i] & m;
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int step = c * 3 + dy;
      // as in the second form; the next chunk's 5 halo loads (issued after the DMA pieces one step ago) may stay in flight at dy == 1
      if (dy == 1 && c + 1 < nchunks) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (step == 0) DH_CTS(2);
      if (step + 1 < nsteps) H3_DMA_B(step + 1)
      if (dy == 0 && c + 1 < nchunks) H3_FETCH_A(c + 1)
      const char* const Bcur = lds + H3A_BYTES + (step & 1) * H2B_BYTES;
      int pl_o = pl;
      asm volatile("" : "+v"(pl_o));
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          half8 bf[2];
#pragma unroll
          for (int b = 0; b < 2; ++b) bf[b] = *reinterpret_cast<const half8*>(Bcur + dx * (128 * 64) + b_row[b] + (((ks * 2) ^ b_x[b]) << 4));
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            half8 af[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
              const int p = pl_o + (4 * h + dy) * HCOLS + dx + a * 32;
              af[a] = *reinterpret_cast<const half8*>(lds + p * 64 + (((ks * 2) ^ ((p >> 2) & 3) ^ kh) << 4));
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int b = 0; b < 2; ++b) acc[h][a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[h][a][b], 0, 0, 0);
          }
        }
      }
    }
  }
#undef H3_FETCH_A
#undef H3_DMA_B
#undef GLDS16S
  DH_CTS(3);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if constexpr (EPI == EPI_HEADS0) staged_heads0_epilogue(P, acc[h], s_conv, m0 + h * 256, n0, wm0, wn0, tid);
    else if constexpr (STAGED) staged_epilogue<EPI, 2, 128>(P, acc[h], s_conv, m0 + h * 256, n0, wm0, wn0, tid, HW);
    else conv_epilogue<EPI, 2, 2>(P, acc[h], (long)P.N * HW, m0 + h * 256, n0, wm0, wn0, lane, HW);
    if (h == 0) __syncthreads();                          // the staging tile is reused by the second half
  }
  DH_CTS(4);
}

#endif  // DH_ABLATION

#ifdef DH_ABLATION   // Winograd F(2,3) prototype (0.83-0.98x of the direct kernel): not in release builds
// ---- 3x3 as Winograd F(2,3) along x (direct along y): PROTOTYPE, opt-in (weights_layout = DH_CONV_LAYOUT_WINO) -------------
// One output row pair (x = 2p, 2p+1) needs the four input columns 2p-1 .. 2p+2:
//     V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3 = d1 - d3            (input transform, per kernel row dy)
//     U0 = g0        U1 = (g0 + g1 + g2) / 2   U2 = (g0 - g1 + g2) / 2   U3 = g2   (weight transform, packed by the host)
//     M_t = sum over (channels, dy) of V_t . U_t                             (4 GEMMs instead of 3 taps, on HALF the rows)
//     y(2p) = M0 + M1 + M2     y(2p+1) = M1 - M2 - M3
// i.e. 16 instead of 24 MFMAs per (32-channel chunk, dy) step and wave -- 1.5x fewer MACs -- for twice the accumulators.
// Same tile as conv3x3_halo2_kernel (4 image rows x 64 columns x 128 couts, 8 waves = 4 rows x 2 cout halves, weights of a
// (chunk, dy) group by LDS-DMA, here two steps ahead into three buffers), but the MFMA rows are the 32 column PAIRS of the wave's image row:
// acc[position][cout tile] = 8 tiles = 128 registers -> one workgroup per CU.  The halo is transformed on its way into LDS
// ([6 rows][4 positions][32 pairs][32 channels], 48 KB): a thread loads the four columns of one (row, pair, 8-channel slot)
// and writes the four transformed slots.  Accumulator start values (the gates' context term) enter through the transform's
// null space: M0 = c(2p), M3 = -c(2p+1).  The output transform feeds the same LDS-staged epilogue as the other kernels.
// Precision: V is a sum of two fp16 values rounded to fp16, U is rounded to fp16 from fp32 -- roundings the direct form
// does not have (tests/test_gpu_parity.py::test_conv_winograd_prototype quantifies them).
constexpr int WV_BYTES = 6 * 4 * 32 * 64;                 // 49,152: transformed halo
constexpr int WU_BYTES = 4 * 128 * 64;                    // 32,768: one (chunk, dy) weight group [4 positions][128 couts][32 ch]
constexpr int WU_BUFS = 3;                               // weight groups are fetched TWO steps ahead (a step is only 16 MFMAs per wave)
constexpr int WINO_LDS_BYTES = WV_BYTES + WU_BUFS * WU_BYTES;   // 147,456 (the staged epilogue tile needs 69,632)

__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2v, a) - __builtin_bit_cast(half2v, b));
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2v, a) + __builtin_bit_cast(half2v, b));
}

template <int EPI, bool CINIT = false>
__global__ __launch_bounds__(512, 2) void conv3x3_wino_kernel(ConvParams P) {
  extern __shared__ __half s_conv[];
  char* const lds = reinterpret_cast<char*>(s_conv);
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)s_conv;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wrow = wave & 3, wn0 = (wave >> 2) * 64;
  long m0; int n0;
  if (!xcd_decode(P, m0, n0, 128)) return;
  const int HW = P.H * P.W;
  const int img = (int)(m0 / HW), y0 = (int)((m0 - (long)img * HW) / 64);
  const int nchunks = P.Ctot / H2CK, nsteps = nchunks * 3;
  const int kh = lane >> 5;

  // halo roles: item id = tid + 512*i (i = 0, 1; the second only for tid < 256) -> halo row id >> 7, pair (id >> 2) & 31, slot id & 3
  const int aq = tid & 3, apair = (tid >> 2) & 31;
  // weight DMA: per-lane offset + wave-uniform base pointer
  const int b_voff = lane * 16;
  const char* const bsrc = reinterpret_cast<const char*>(P.wt_halo) + (long)(n0 >> 7) * nsteps * WU_BYTES;
  int b_row[2], b_x[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int r = wn0 + b * 32 + (lane & 31);
    b_row[b] = r * 64; b_x[b] = ((r >> 2) & 3) ^ kh;
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[t][b][q] = 0.f;
  if constexpr (CINIT) {
    // start values c(pixel, cout): M0 = c(even pixel), M3 = -c(odd pixel) (see above).  32 lanes read one 128-byte run.
    const float* base = P.cinit + ((long)P.cinit_idx[img] * HW + (m0 - (long)img * HW) + wrow * 64) * P.cinit_stride + P.cinit_off + n0 + wn0 + (lane & 31);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const bool co_ok = n0 + wn0 + b * 32 + (lane & 31) < P.Cout;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int pair = (q & 3) + 8 * (q >> 2) + 4 * kh;
        acc[0][b][q] = co_ok ? base[(long)(2 * pair) * P.cinit_stride + b * 32] : 0.f;
        acc[3][b][q] = co_ok ? -base[(long)(2 * pair + 1) * P.cinit_stride + b * 32] : 0.f;
      }
    }
  }

  u32x4 ra[2][4];
#define WN_FETCH_A(chunk_)                                                                                           \
  {                                                                                                                  \
    int cs = (chunk_) * H2CK, sgi = 0;                                                                               \
    _Pragma("unroll") for (int q = 0; q < MAXSEG - 1; ++q)                                                           \
      if (sgi == q && q + 1 < P.nseg && cs >= P.segC[q]) { cs -= P.segC[q]; sgi = q + 1; }                          \
    const __half* base = P.in[0]; int segs = P.segS[0];                                                              \
    _Pragma("unroll") for (int q = 1; q < MAXSEG; ++q) if (sgi == q) { base = P.in[q]; segs = P.segS[q]; }           \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                  \
      const int hr = (tid >> 7) + 4 * i;                     /* halo row of item i: 0..3, then 4..5 (tid < 256) */    \
      const int y = y0 - 1 + hr;                                                                                     \
      const bool rok = hr < 6 && (unsigned)y < (unsigned)P.H;                                                        \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                                \
        const int col = 2 * apair - 1 + k;                                                                           \
        const bool ok = rok && (unsigned)col < 64u;                                                                  \
        ra[i][k] = *reinterpret_cast<const u32x4*>(base + (long)(ok ? (img * P.H + y) * 64 + col : 0) * segs + cs + aq * 8); \
      }                                                                                                              \
    }                                                                                                                \
  }
#define WN_GLDS16S(sbase_, voff_, ldsaddr_)                                                                          \
  {                                                                                                                  \
    unsigned keep_;                                                                                                  \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(voff_), "s"(sbase_), "s"(ldsaddr_) : "memory");                                \
  }
#define WN_DMA_B(step_)                                                                                              \
  {                                                                                                                  \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                  \
      const unsigned long gaddr = (unsigned long)(bsrc + (long)(step_) * WU_BYTES + (wave + 8 * q) * 1024);           \
      const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)gaddr);                                          \
      const unsigned ghi = __builtin_amdgcn_readfirstlane((unsigned)(gaddr >> 32));                                  \
      const void* gs = reinterpret_cast<const void*>(((unsigned long)ghi << 32) | glo);                              \
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + WV_BYTES + ((step_) % WU_BUFS) * WU_BYTES + (wave + 8 * q) * 1024); \
      WN_GLDS16S(gs, b_voff, dst)                                                                                    \
    }                                                                                                                \
  }

  WN_FETCH_A(0)
  WN_DMA_B(0)
  if (nsteps > 1) WN_DMA_B(1)
  for (int c = 0; c < nchunks; ++c) {
    // every wave has finished the previous chunk's reads of the transformed halo
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int hr = (tid >> 7) + 4 * i;
      if (hr < 6) {
        const int y = y0 - 1 + hr;
        const bool rok = (unsigned)y < (unsigned)P.H;
        uint32_t msk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) msk[k] = (rok && (unsigned)(2 * apair - 1 + k) < 64u) ? 0xffffffffu : 0u;
        u32x4 v0, v1, v2, v3;
#pragma unroll
        for (int e = 0; e < 4; ++e) {             // component by component: two channels per 32-bit lane
          const uint32_t d0 = ra[i][0][e] & msk[0], d1 = ra[i][1][e] & msk[1], d2 = ra[i][2][e] & msk[2], d3 = ra[i][3][e] & msk[3];
          v0[e] = pk_sub(d0, d2); v1[e] = pk_add(d1, d2); v2[e] = pk_sub(d2, d1); v3[e] = pk_sub(d1, d3);
        }
        const int R0 = (hr * 4) * 32 + apair;                     // row of position 0; position t is 32 rows further
#define WN_PUT(t_, v_) { const int R = R0 + (t_) * 32; *reinterpret_cast<u32x4*>(lds + R * 64 + ((aq ^ ((R >> 2) & 3)) << 4)) = v_; }
        WN_PUT(0, v0) WN_PUT(1, v1) WN_PUT(2, v2) WN_PUT(3, v3)
#undef WN_PUT
      }
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int step = c * 3 + dy;
      // group `step` (DMA issued two steps ago) has landed; younger operations that may stay in flight (vector memory
      // operations complete in order): the 4 pieces of group step + 1 and, from dy == 1 on, the next chunk's 8 halo loads
      if (step + 1 >= nsteps) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else if (dy == 0 || c + 1 >= nchunks) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (step + 2 < nsteps) WN_DMA_B(step + 2)             // into the buffer last read in step - 1
      if (dy == 0 && c + 1 < nchunks) WN_FETCH_A(c + 1)
      const char* const Bcur = lds + WV_BYTES + (step % WU_BUFS) * WU_BYTES;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int R = ((wrow + dy) * 4 + t) * 32 + (lane & 31);
        const int a_row = R * 64, a_x = ((R >> 2) & 3) ^ kh;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const half8 af = *reinterpret_cast<const half8*>(lds + a_row + (((ks * 2) ^ a_x) << 4));
          half8 bf[2];
#pragma unroll
          for (int b = 0; b < 2; ++b) bf[b] = *reinterpret_cast<const half8*>(Bcur + t * (128 * 64) + b_row[b] + (((ks * 2) ^ b_x[b]) << 4));
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf[b], acc[t][b], 0, 0, 0);
        }
      }
    }
  }
#undef WN_FETCH_A
#undef WN_DMA_B
#undef WN_GLDS16S
  // ---- output transform + the staged epilogue's first phase (bias / activation, fp16 into the [256 px][128 cout] LDS tile)
  constexpr int ELD = 128 + 8;
  __half* const sT = s_conv;
  __syncthreads();                        // the operand tiles of the main loop are dead
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int cl = wn0 + b * 32 + (lane & 31), co = n0 + cl;
    const float add = (co < P.CoutPad ? P.bias[co] : 0.f) + ((P.gterm && co < P.CoutPad) ? P.gterm[(long)img * P.CoutPad + co] : 0.f);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int pair = (q & 3) + 8 * (q >> 2) + 4 * kh;
      float ve = acc[0][b][q] + acc[1][b][q] + acc[2][b][q] + add;
      float vo = acc[1][b][q] - acc[2][b][q] - acc[3][b][q] + add;
      switch (EPI) {
        case EPI_RELU: ve = fmaxf(ve, 0.f); vo = fmaxf(vo, 0.f); break;
        case EPI_SIGMOID: case EPI_GRU_ZR: ve = sigmoidf_(ve); vo = sigmoidf_(vo); break;
        case EPI_GRU_Q: ve = tanhf_(ve); vo = tanhf_(vo); break;
        default: break;
      }
      const int row = wrow * 64 + 2 * pair;
      sT[row * ELD + cl] = __float2half(ve);
      sT[(row + 1) * ELD + cl] = __float2half(vo);
    }
  }
  __syncthreads();
  staged_tile_store<EPI, 128>(P, sT, m0, n0, tid);
}
#endif  // DH_ABLATION


// ---- 7x7 on 4 input channels (the flow encoder's first layer: motion features -> 128) -----------------------------------
// In the generic loop this layer costs as much as a 128 -> 64 3x3 convolution: its K = 49 taps x 8 (4 real + 4 padded)
// channels is walked in 64-wide chunks of gathered 16-byte pieces.  Here the whole problem sits in LDS: the 10 x 72 pixel
// halo of a 4-row tile with its 4 real channels (5.6 KB) and ALL weights [128 cout][7 dy][8 dx (7 + one zero tap)][4 ch]
// (57 KB, row stride 464 B: conflict-free ds_read_b128).  A k-step of 16 = four x-adjacent taps of one kernel row, which are
// 16 contiguous bytes of the halo per lane pair: K = 7 x 32 = 224 -> 14 k-steps instead of 25, no staging inside the loop.
// (Round 3 measured a persistent form -- one workgroup per CU walking the tiles, a wave's 28 B fragments resident in 112
// registers, the next tile's halo fetched under the epilogue: 0.555-0.588 ms against 0.565-0.576 ms for this kernel on
// 2048 edges.  Reloading the weights per tile is not what the kernel waits for; not kept.)
constexpr int C7_COLS = 72, C7_ROWS = 10, C7_K = 224, C7_WLD = C7_K + 8;
constexpr int C7_A_BYTES = C7_ROWS * C7_COLS * 8, C7_W_BYTES = 128 * C7_WLD * 2;
constexpr int C7_LDS_BYTES = (C7_A_BYTES + C7_W_BYTES) > BM * (128 + 8) * 2 ? (C7_A_BYTES + C7_W_BYTES) : BM * (128 + 8) * 2;
constexpr int C7_W64_BYTES = 64 * C7_WLD * 2;             // 29,696: the weights of 64 couts
constexpr int C7_LDS64_BYTES = (C7_A_BYTES + C7_W64_BYTES) > BM * (64 + 8) * 2 ? (C7_A_BYTES + C7_W64_BYTES) : BM * (64 + 8) * 2;      // 36,864

// Round 6: BN = 64 (option conv_c7_split, default): a workgroup of FOUR waves computes 256 px x 64 couts -- half the weights (29.7 KB),
// a staged tile of 36.9 KB -- so that FOUR workgroups fit a CU instead of two.  The kernel runs far from either roof (0.63 PFLOP/s, 2.8 TB/s
// of stores): its workgroups spend most of their life in a prologue (weights L2 -> registers -> LDS) and an epilogue with nothing of their
// own to overlap, and unlike the gate convolutions more residency is what it lacks.  Same MFMAs per output element: same results.
template <int EPI, int BN = 128>
__global__ __launch_bounds__(BN == 128 ? 512 : 256, BN == 128 ? 4 : 4) void conv7x7_c4_kernel(ConvParams P) {
  constexpr int NT = BN == 128 ? 512 : 256;              // threads
  extern __shared__ __half s_conv[];
  char* const lds = reinterpret_cast<char*>(s_conv);
  char* const sW = lds + C7_A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wrow = wave & 3, wn0 = (wave >> 2) * 64, wm0 = wrow * 64;
  // BN == 64: 1-D grid, workgroup id -> (pixel tile id >> 1, cout half id & 1)
  const long m0 = (BN == 128 ? (long)blockIdx.x : (long)(blockIdx.x >> 1)) * BM;
  const int n0 = BN == 128 ? 0 : (int)(blockIdx.x & 1) * 64;
  const int HW = P.H * P.W;
  const int img = (int)(m0 / HW), y0 = (int)((m0 - (long)img * HW) / 64);
  DH_CTS(0); DH_CTS_ID();
  // halo: column c holds image column c - 4, row r holds image row y0 - 3 + r; 8 bytes (4 channels) per pixel
#pragma unroll
  for (int i = 0; i < (C7_ROWS * C7_COLS + NT - 1) / NT; ++i) {
    const int id = tid + NT * i;
    if (id < C7_ROWS * C7_COLS) {
      const int r = id / C7_COLS, c = id - r * C7_COLS;
      const int y = y0 - 3 + r, x = c - 4;
      uint2 v{0u, 0u};
      if ((unsigned)y < (unsigned)P.H && (unsigned)x < 64u)
        v = *reinterpret_cast<const uint2*>(P.in[0] + ((long)(img * P.H + y) * 64 + x) * P.segS[0]);
      *reinterpret_cast<uint2*>(lds + id * 8) = v;
    }
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) {                            // BN rows x 28 pieces of 16 bytes (BN * 28 = 7 * NT)
    const int id = tid + NT * i, row = id / 28, pc = id - row * 28;
    *reinterpret_cast<uint4*>(sW + row * (C7_WLD * 2) + pc * 16) = *reinterpret_cast<const uint4*>(P.wt_halo + (long)(n0 + row) * C7_K + pc * 8);
  }
  f32x16 acc[2][2];
  zero_acc<2, 2>(acc);
  DH_CTS(1);
  __syncthreads();
  DH_CTS(2);
  const int p = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int dy = 0; dy < 7; ++dy) {
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) {                       // taps dx = hx*4 + kh*2 + {0, 1}
      const int ks = dy * 2 + hx;
      half8 af[2], bf[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const char* src = lds + ((wrow + dy) * C7_COLS + a * 32 + p + hx * 4 + kh * 2 + 1) * 8;      // 8-byte aligned
        const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 8);
        const uint4 v{lo.x, lo.y, hi.x, hi.y};
        af[a] = *reinterpret_cast<const half8*>(&v);
      }
#pragma unroll
      for (int b = 0; b < 2; ++b)
        bf[b] = *reinterpret_cast<const half8*>(sW + (wn0 + b * 32 + p) * (C7_WLD * 2) + ks * 32 + kh * 16);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
  }
  DH_CTS(3);
  staged_epilogue<EPI, 2, BN, NT>(P, acc, s_conv, m0, n0, wm0, wn0, tid, HW);
  DH_CTS(4);
}

// ---- the stem with SIXTEEN waves of 32 px x 64 couts per workgroup (round 6, option conv_c7_w16) ---------------------------------------------
// The stem is latency-bound with its SIMDs issuing ~45 % of the time (670 instructions per wave and tile in chains of LDS / memory latencies,
// four waves per SIMD): the same tile, the same LDS image and the same two workgroups per CU, but 32 accumulators per wave instead of 64
// (<= 64 registers) double the waves per SIMD to eight.  A wave owns pixel columns 32 a .. 32 a + 31 of one image row and 64 couts.  Same
// MFMAs per output element in the same order: EQUAL results.
template <int EPI>
__global__ __launch_bounds__(1024, 8) void conv7x7_c4_w16_kernel(ConvParams P) {
  extern __shared__ __half s_conv[];
  char* const lds = reinterpret_cast<char*>(s_conv);
  char* const sW = lds + C7_A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wrow = wave & 3, ah = (wave >> 2) & 1, wn0 = (wave >> 3) * 64;
  const long m0 = (long)blockIdx.x * BM;
  const int HW = P.H * P.W;
  const int img = (int)(m0 / HW), y0 = (int)((m0 - (long)img * HW) / 64);
  if (tid < C7_ROWS * C7_COLS) {
    const int r = tid / C7_COLS, c = tid - r * C7_COLS;
    const int y = y0 - 3 + r, x = c - 4;
    uint2 v{0u, 0u};
    if ((unsigned)y < (unsigned)P.H && (unsigned)x < 64u)
      v = *reinterpret_cast<const uint2*>(P.in[0] + ((long)(img * P.H + y) * 64 + x) * P.segS[0]);
    *reinterpret_cast<uint2*>(lds + tid * 8) = v;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {                            // 128 rows x 28 pieces of 16 bytes = 3584 = 3.5 x 1024
    const int id = tid + 1024 * i;
    if (id < 128 * 28) {
      const int row = id / 28, pc = id - row * 28;
      *reinterpret_cast<uint4*>(sW + row * (C7_WLD * 2) + pc * 16) = *reinterpret_cast<const uint4*>(P.wt_halo + (long)row * C7_K + pc * 8);
    }
  }
  f32x16 acc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[b][q] = 0.f;
  __syncthreads();
  const int p = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int dy = 0; dy < 7; ++dy) {
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) {
      const int ks = dy * 2 + hx;
      const char* src = lds + ((wrow + dy) * C7_COLS + ah * 32 + p + hx * 4 + kh * 2 + 1) * 8;
      const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 8);
      const uint4 v{lo.x, lo.y, hi.x, hi.y};
      const half8 af = *reinterpret_cast<const half8*>(&v);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const half8 bf = *reinterpret_cast<const half8*>(sW + (wn0 + b * 32 + p) * (C7_WLD * 2) + ks * 32 + kh * 16);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[b], 0, 0, 0);
      }
    }
  }
  // the staged epilogue of the other kernels with one 32-pixel accumulator tile per wave and 1024 threads
  constexpr int ELD = 128 + 8;
  __syncthreads();
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int cl = wn0 + b * 32 + (lane & 31);
    const float add = P.bias[cl];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = wrow * 64 + ah * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
      float v = acc[b][q] + add;
      if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
      s_conv[row * ELD + cl] = __float2half(v);
    }
  }
  __syncthreads();
  staged_tile_store<EPI, 128, 1024>(P, s_conv, m0, 0, tid);
}

// ---- the stem as PERSISTENT workgroups (round 6, option conv_c7_pp) ---------------------------------------------------------------------
// conv7x7_c4_kernel writes 3.2 GB per launch at 2.9-3.0 TB/s where a write-only stream reaches 6.9 TB/s (scripts/ubench/hbm_rw.py): its
// workgroups live 11 us -- weights L2 -> registers -> LDS, 2.8 us of MFMAs, 5 us of parking and storing -- with nothing of their own to
// overlap and 16 waves per CU whatever the workgroup size.  Here ONE workgroup of 16 waves per CU walks pixel tiles: the 57 KB of weights are
// staged once; the waves form two groups of eight with the roles of the kernel above, and in every "super-step" one group multiplies its
// tile while the other parks and stores the tile it multiplied in the previous super-step and stages the halo of its next one; then
// the roles swap.  The groups meet at four workgroup barriers per super-step (the parking group needs them around its two 128-pixel
// halves of the staged tile; the multiplying group passes them between its kernel rows).  LDS: weights 59.4 KB + 2 x (halo 5.8 KB +
// half a staged tile 34.8 KB) = 140.5 KB.  Same MFMAs in the same order per output element, same epilogue arithmetic: EQUAL results.
constexpr int C7P_S_BYTES = 128 * (128 + 8) * 2;               // 34,816: half a staged tile [128 px][128 couts + 8]
constexpr int C7P_G_BYTES = C7_A_BYTES + C7P_S_BYTES;          // per wave group
constexpr int C7P_LDS_BYTES = C7_W_BYTES + 2 * C7P_G_BYTES;    // 140,544
template <int EPI>
__global__ __launch_bounds__(1024, 4) void conv7x7_c4_pp_kernel(ConvParams P) {
  extern __shared__ __half s_conv[];
  char* const lds = reinterpret_cast<char*>(s_conv);
  char* const sW = lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 3, w8 = wave & 7, tg = tid & 511;           // wave group, wave inside it, thread inside it
  const int wrow = w8 & 3, wn0 = (w8 >> 2) * 64, wm0 = wrow * 64;
  char* const sH = lds + C7_W_BYTES + grp * C7P_G_BYTES;              // this group's halo
  __half* const sS = reinterpret_cast<__half*>(sH + C7_A_BYTES);     // this group's half staged tile
  const int HW = P.H * P.W;
  const long ntiles = (long)P.N * HW / BM;
  // weights once: 128 rows x 28 pieces of 16 bytes
  for (int id = tid; id < 128 * 28; id += 1024) {
    const int row = id / 28, pc = id - row * 28;
    *reinterpret_cast<uint4*>(sW + row * (C7_WLD * 2) + pc * 16) = *reinterpret_cast<const uint4*>(P.wt_halo + (long)row * C7_K + pc * 8);
  }
  // halo of tile t into this group's buffer: column c holds image column c - 4, row r image row y0 - 3 + r; 8 bytes per pixel
  auto halo_fetch = [&](long t, uint2 (&v)[2], int tg) {
    const long m0 = t * BM;
    const int img = (int)(m0 / HW), y0 = (int)((m0 - (long)img * HW) / 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int id = tg + 512 * i;
      v[i] = uint2{0u, 0u};
      if (id < C7_ROWS * C7_COLS) {
        const int r = id / C7_COLS, c = id - r * C7_COLS;
        const int y = y0 - 3 + r, x = c - 4;
        if ((unsigned)y < (unsigned)P.H && (unsigned)x < 64u)
          v[i] = *reinterpret_cast<const uint2*>(P.in[0] + ((long)(img * P.H + y) * 64 + x) * P.segS[0]);
      }
    }
  };
  auto halo_store = [&](const uint2 (&v)[2], int tg) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int id = tg + 512 * i;
      if (id < C7_ROWS * C7_COLS) *reinterpret_cast<uint2*>(sH + id * 8) = v[i];
    }
  };
  // tile s of this workgroup's sequence; group (s & 1) owns it
  auto tile_of = [&](long sidx) { return (long)blockIdx.x + sidx * gridDim.x; };
  f32x16 acc[2][2];
  const int p = lane & 31, kh = lane >> 5;
  {
    uint2 v[2];
    if (tile_of(grp) < ntiles) { halo_fetch(tile_of(grp), v, tg); halo_store(v, tg); }       // each group's first tile
  }
  __syncthreads();
  // super-step k: group (k & 1) multiplies tile k; the other group finishes tile k - 1 and stages the halo of tile k + 1
  for (long k = 0; tile_of(k > 0 ? k - 1 : 0) < ntiles; ++k) {
    const bool mul = (int)(k & 1) == grp;
    const long t_mul = tile_of(k), t_epi = tile_of(k - 1), t_next = tile_of(k + 1);
    const bool do_mul = mul && t_mul < ntiles, do_epi = !mul && k > 0 && t_epi < ntiles;
    // (opaque per super-step: the per-thread addresses of the halo / parking / store code would otherwise be hoisted out of the tile loop
    // as invariants and kept -- in scratch -- across the multiply phase)
    int lane_o = lane, tg_o = tg;
    asm volatile("" : "+v"(lane_o), "+v"(tg_o));
    uint2 hv[2];
    const bool do_halo = !mul && k > 0 && t_next < ntiles;                              // (k == 0: both first halos were staged above)
    if (do_halo) halo_fetch(t_next, hv, tg_o);
    if (do_mul) zero_acc<2, 2>(acc);
    const long m0e = t_epi * BM;
#pragma unroll
    for (int seg = 0; seg < 4; ++seg) {
      if (do_mul) {
        // kernel rows 2 seg, 2 seg + 1 (the last segment: row 6 only)
#pragma unroll
        for (int d2 = 0; d2 < 2; ++d2) {
          const int dy = 2 * seg + d2;
          if (dy < 7) {
#pragma unroll
            for (int hx = 0; hx < 2; ++hx) {
              const int ks = dy * 2 + hx;
              half8 af[2], bf[2];
#pragma unroll
              for (int a = 0; a < 2; ++a) {
                const char* src = sH + ((wrow + dy) * C7_COLS + a * 32 + p + hx * 4 + kh * 2 + 1) * 8;
                const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 8);
                const uint4 v{lo.x, lo.y, hi.x, hi.y};
                af[a] = *reinterpret_cast<const half8*>(&v);
              }
#pragma unroll
              for (int b = 0; b < 2; ++b)
                bf[b] = *reinterpret_cast<const half8*>(sW + (wn0 + b * 32 + p) * (C7_WLD * 2) + ks * 32 + kh * 16);
#pragma unroll
              for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
          }
        }
      } else if (do_epi) {
        const int half = seg >> 1;
        if ((seg & 1) == 0) {
          // park: the waves that own image rows 2 half, 2 half + 1 of the tile write their 64 x 64 sub-tiles (bias, activation, fp16)
          if ((wrow >> 1) == half) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              const int cl = wn0 + b * 32 + (lane_o & 31);
              const float add = P.bias[cl];
#pragma unroll
              for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                  const int row = (wrow & 1) * 64 + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane_o >> 5);
                  float v = acc[a][b][q] + add;
                  if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
                  sS[row * (128 + 8) + cl] = __float2half(v);
                }
            }
          }
        } else {
          staged_tile_store<EPI, 128, 512, false, 128>(P, sS, m0e + half * 128, 0, tg_o);
          if (seg == 3 && do_halo) halo_store(hv, tg_o);
        }
      } else if (seg == 3 && do_halo) halo_store(hv, tg_o);
      __syncthreads();
    }
  }
}

// ---- 1x1 on 128 input channels with many outputs (GraphAgg's upmask head: 128 -> 576 on the source frames, droid_net.py:62) -------
// The generic loop runs this shape as 5 independent cout tiles per pixel tile, each with a two-chunk main loop between a cold
// prologue and a full epilogue (0.80 ms at 512 frames for 2.2 GB of traffic).  Here the pixel tile's activations [256 px][128 ch] are
// staged ONCE and stay in LDS while the workgroup walks the 64-cout tiles (576 = 9 tiles, no padding tile): per tile a 16 KB weight
// slab (next tile's slab fetched into registers under the MFMAs), 16 MFMAs per wave (64 px x 32 couts) and the staged 16-byte-piece
// store of the other kernels, whose stores drain under the next tile.  LDS: 69.6 (A) + 17.4 (B) + 36.9 (staged tile) KB.  Measured at 512
// frames: 0.60-0.68 ms against 0.80-0.86 ms (a persistent form that prefetches the next pixel tile: 0.66-0.73 ms, not kept).
constexpr int K1_LD = 128 + 8;
constexpr int K1_A_BYTES = BM * K1_LD * 2, K1_B_BYTES = 64 * K1_LD * 2, K1_T_BYTES = BM * (64 + 8) * 2;
constexpr int K1_LDS_BYTES = K1_A_BYTES + K1_B_BYTES + K1_T_BYTES;
// Round 6: TM = 128 (option conv_k1_half, default): 128-pixel tiles by FOUR waves -- 34.8 (A) + 17.4 (B) + 18.4 (staged tile) = 70.6 KB, TWO
// workgroups per CU, so that one workgroup's activation fetch and stores run under the other's cout-tile walk (the 256-pixel form has the
// CU to itself: its 64 KB fetch is exposed).  Same MFMAs per output element: same results.
template <int EPI, int TM = 256>
__global__ __launch_bounds__(TM * 2, TM == 256 ? 1 : 2) void conv1x1_c128_kernel(ConvParams P) {
  constexpr int NT = TM * 2, AIT = TM * 16 / NT, BIT = 64 * 16 / NT;        // threads; 16-byte pieces per thread of the A / B tiles
  extern __shared__ __half s_conv[];
  __half* sA = s_conv;
  __half* sB = s_conv + TM * K1_LD;
  __half* sT = s_conv + (TM + 64) * K1_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave % (TM / 64)) * 64, wn0 = (wave / (TM / 64)) * 32;
  const long m0 = (long)blockIdx.x * TM;
  const int HW = P.H * P.W;
  uint4 ra[AIT], rb[BIT];
#pragma unroll
  for (int i = 0; i < AIT; ++i) {
    const int id = tid + NT * i;
    ra[i] = *reinterpret_cast<const uint4*>(P.in[0] + (m0 + (id >> 4)) * P.segS[0] + (id & 15) * 8);
  }
  auto fetch_b = [&](int n0) {
#pragma unroll
    for (int i = 0; i < BIT; ++i) {
      const int id = tid + NT * i, row = id >> 4;
      rb[i] = n0 + row < P.CoutPad ? *reinterpret_cast<const uint4*>(P.wt + (long)(n0 + row) * P.Kpad + (id & 15) * 8) : uint4{0u, 0u, 0u, 0u};
    }
  };
  fetch_b(0);
#pragma unroll
  for (int i = 0; i < AIT; ++i) {
    const int id = tid + NT * i;
    *reinterpret_cast<uint4*>(sA + (id >> 4) * K1_LD + (id & 15) * 8) = ra[i];
  }
  const int p = lane & 31, kh = lane >> 5;
  for (int n0 = 0; n0 < P.CoutPad; n0 += 64) {
#pragma unroll
    for (int i = 0; i < BIT; ++i) {        // (the previous cout tile's B fragments are read: two barriers in its epilogue)
      const int id = tid + NT * i;
      *reinterpret_cast<uint4*>(sB + (id >> 4) * K1_LD + (id & 15) * 8) = rb[i];
    }
    __syncthreads();                       // A (first cout tile) and B staged
    if (n0 + 64 < P.CoutPad) fetch_b(n0 + 64);          // the next cout tile's slab under the MFMAs
    f32x16 acc[2][1];
    zero_acc<2, 1>(acc);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const half8 bf = *reinterpret_cast<const half8*>(sB + (wn0 + p) * K1_LD + ks * 16 + kh * 8);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const half8 af = *reinterpret_cast<const half8*>(sA + (wm0 + a * 32 + p) * K1_LD + ks * 16 + kh * 8);
        acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[a][0], 0, 0, 0);
      }
    }
    staged_epilogue<EPI, 1, 64, NT, false, TM>(P, acc, sT, m0, n0, wm0, wn0, tid, HW);      // (its first barrier: the previous tile's pieces are read)
  }
}

// ---- global-context reduction of the ConvGRU as its own kernel: red[n][c] += sum_px sigmoid(W net + b)[c] * net[c] --------
// (reference gru.py:23-24).  A 128 x 128 1x1 convolution is a streaming problem (3.2 GB of hidden state at 4096 edges
// against 0.4 TFLOP), and in the generic loop a workgroup never has more than one 32 KB chunk in flight and re-reads the
// hidden state for the gate product: 1.86 ms = 1.7 TB/s.  Here a workgroup requests its whole 256 px x 128 ch tile at once
// (8 x 16 bytes per thread: 128 KB in flight per CU with two workgroups), the tile stays in LDS as MFMA operand AND as the
// gate's second factor, and the weights come from L2 straight into B fragments (two halves of K).
// (Round 3 tried persistent workgroups -- one per CU, the next tile's loads issued under the current tile's MFMAs and gate
// algebra, weights resident in 64 registers: 1.66 instead of 1.45 ms.  Two workgroups per CU overlapping each other's phases
// beat one workgroup overlapping its own loads; kept as it was.)
__global__ __launch_bounds__(512, 4) void glo_reduce_kernel(ConvParams P) {
  extern __shared__ __half s_conv[];
  const int tid = threadIdx.x;
  const long m0 = (long)blockIdx.x * BM;
  const int HW = P.H * P.W;
  const int img = (int)(m0 / HW);
  uint4 ra[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int id = tid + 512 * i;
    ra[i] = *reinterpret_cast<const uint4*>(P.in[0] + (m0 + (id >> 4)) * P.segS[0] + (id & 15) * 8);
  }
  half8 bf[2][4];
  glo_load_b0(P.wt, P.Kpad, tid, bf);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int id = tid + 512 * i;
    *reinterpret_cast<uint4*>(s_conv + (id >> 4) * GLD + (id & 15) * 8) = ra[i];
  }
  glo_tile_reduce(P.wt, P.Kpad, P.bias, P.red + (long)img * P.Cout, s_conv, tid, bf);
}

// mean over the rows of each segment (GraphAgg's scatter_mean over the edges of a source frame, reference
// droid_net.py:67): out[k] = mean_{e in order[seg_off[k] .. seg_off[k+1])} x[e]; one thread = 8 channels (16 B),
// fp32 accumulation in a fixed order, one rounding to fp16.
__global__ __launch_bounds__(256) void segment_mean_kernel(const __half* __restrict__ x, const int64_t* __restrict__ order,
                                                           const int64_t* __restrict__ seg_off, __half* __restrict__ out,
                                                           long row8) {
  const int k = blockIdx.y;
  const long o = (long)blockIdx.x * 256 + threadIdx.x;
  if (o >= row8) return;
  const long a = seg_off[k], b = seg_off[k + 1];
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long i = a; i < b; ++i) {
    const uint4 v = reinterpret_cast<const uint4*>(x)[order[i] * row8 + o];
    const __half2* h2 = reinterpret_cast<const __half2*>(&v);
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float2 f = __half22float2(h2[q]); acc[2 * q] += f.x; acc[2 * q + 1] += f.y; }
  }
  const float inv = b > a ? 1.f / (float)(b - a) : 0.f;
  uint4 r;
  __half2* rh = reinterpret_cast<__half2*>(&r);
#pragma unroll
  for (int q = 0; q < 4; ++q) rh[q] = __floats2half2_rn(acc[2 * q] * inv, acc[2 * q + 1] * inv);
  reinterpret_cast<uint4*>(out)[(long)k * row8 + o] = r;
}

template <int WM, int WN, int BN, int EPI>
int launch_epi(const ConvParams& P, hipStream_t st) {
  const long M = (long)P.N * P.H * P.W;
  const dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((P.CoutPad + BN - 1) / BN));
  const size_t lds = (size_t)(BM + BN) * LDT * sizeof(__half);
  if constexpr (BN == 128 && WM == 64 && WN == 64 && (EPI == EPI_LINEAR || EPI == EPI_RELU || EPI == EPI_SIGMOID || EPI == EPI_GLO)) {
    // whole tiles inside one image: the LDS-staged epilogue (its tile is larger than the operand tiles of the loop)
    if (M % BM == 0 && ((long)P.H * P.W) % BM == 0 && staged_epilogue_ok<EPI>(P)) {
      constexpr size_t lds_st = (size_t)BM * (BN + 8) * sizeof(__half);
      DH_LDS_OPTIN((&conv_igemm_kernel<WM, WN, BN, EPI, true>), 80 * 1024);
      hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, BN, EPI, true>), grid, dim3(512), lds_st > lds ? lds_st : lds, st, P);
      DH_LAUNCH_CHECK();
      return DH_OK;
    }
  }
  hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, BN, EPI>), grid, dim3(512), lds, st, P);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

template <int EPI, int BN>
int launch_halo(const ConvParams& P, hipStream_t st) {
  const long M = (long)P.N * P.H * P.W;
  const dim3 grid((unsigned)(M / BM), (unsigned)(P.CoutPad / BN));
  const size_t lds = (size_t)(HPIX + 9 * BN) * (halo_ck(BN) + 8) * sizeof(__half);
  DH_LDS_OPTIN((&conv3x3_halo_kernel<EPI, BN>), 128 * 1024);
  if constexpr ((BN == 128 || BN == 64) && (EPI == EPI_LINEAR || EPI == EPI_RELU || EPI == EPI_SIGMOID || EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q)) {
    if (staged_epilogue_ok<EPI>(P)) {
      DH_LDS_OPTIN((&conv3x3_halo_kernel<EPI, BN, true>), 128 * 1024);
      hipLaunchKernelGGL((conv3x3_halo_kernel<EPI, BN, true>), grid, dim3(512), lds, st, P);
      DH_LAUNCH_CHECK();
      return DH_OK;
    }
  }
  hipLaunchKernelGGL((conv3x3_halo_kernel<EPI, BN>), grid, dim3(512), lds, st, P);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

#ifdef DH_ABLATION
template <int EPI>
int launch_dma(const ConvParams& P, hipStream_t st) {
  const long M = (long)P.N * P.H * P.W;
  const dim3 grid((unsigned)(M / BM), (unsigned)(P.CoutPad / 128));
  DH_LDS_OPTIN((&conv3x3_dma_kernel<EPI>), DMA_LDS_BYTES);
  if (EPI == EPI_RELU) {                                  // timing ablations (scripts/bench_conv.py), never set in production
    const int var = opts().dma_var;
#define DMA_VARIANT(v_)                                                                                             \
    if (var == v_) {                                                                                                 \
      DH_LDS_OPTIN((&conv3x3_dma_kernel<EPI_RELU, v_>), DMA_LDS_BYTES);                                              \
      hipLaunchKernelGGL((conv3x3_dma_kernel<EPI_RELU, v_>), grid, dim3(512), DMA_LDS_BYTES, st, P);                 \
      DH_LAUNCH_CHECK();                                                                                             \
      return DH_OK;                                                                                                  \
    }
    DMA_VARIANT(1) DMA_VARIANT(2) DMA_VARIANT(5) DMA_VARIANT(6) DMA_VARIANT(7)
#undef DMA_VARIANT
  }
  hipLaunchKernelGGL((conv3x3_dma_kernel<EPI>), grid, dim3(512), DMA_LDS_BYTES, st, P);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

#endif  // DH_ABLATION

template <int EPI>
int launch_halo2(const ConvParams& P0, hipStream_t st) {
  ConvParams P = P0;
  const long M = (long)P.N * P.H * P.W;
  dim3 grid((unsigned)(M / BM), (unsigned)(P.CoutPad / 128));
  P.ny = (int)grid.y;
  if (opts().conv_xcd) { P.xcd_tiles = (int)((grid.x + 7) / 8); grid = dim3((unsigned)P.xcd_tiles * 8 * grid.y, 1); }
  if constexpr (EPI == EPI_LINEAR) {
    if (P.out_f32 == 3) {                                 // fp32 output in the accumulator-tile layout: 16-byte stores, 1 KB per wave-store
      DH_LDS_OPTIN((&conv3x3_halo2_kernel<EPI, false, 3>), 80 * 1024);
      hipLaunchKernelGGL((conv3x3_halo2_kernel<EPI, false, 3>), grid, dim3(512), H2_LDS_BYTES, st, P);
      DH_LAUNCH_CHECK();
      return DH_OK;
    }
    if (P.out_f32) {                                      // fp32 output: 32 lanes already store one full 128-byte run per pixel
      DH_LDS_OPTIN((&conv3x3_halo2_kernel<EPI, false>), 80 * 1024);
      hipLaunchKernelGGL((conv3x3_halo2_kernel<EPI, false>), grid, dim3(512), H2_LDS_BYTES, st, P);
      DH_LAUNCH_CHECK();
      return DH_OK;
    }
  }
  if constexpr (EPI == EPI_GRU_Q) {
    if (P.glo_red) {                                      // + the next iteration's global-context reduction (ConvParams::glo_red)
#define H2_GLO(ci_)                                                                                                  \
      {                                                                                                              \
        DH_LDS_OPTIN((&conv3x3_halo2_kernel<EPI, true, ci_, 0, true>), 80 * 1024);                                   \
        hipLaunchKernelGGL((conv3x3_halo2_kernel<EPI, true, ci_, 0, true>), grid, dim3(512), H2_LDS_BYTES, st, P);   \
        DH_LAUNCH_CHECK();                                                                                           \
        return DH_OK;                                                                                                \
      }
      if (P.cinit && P.cinit_stride < 0) H2_GLO(2)
      if (P.cinit) H2_GLO(1)
      H2_GLO(0)
#undef H2_GLO
    }
  }
  if constexpr (EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q) {
    if (P.cinit && P.cinit_stride < 0) {                  // start values in the accumulator-tile layout
      DH_LDS_OPTIN((&conv3x3_halo2_kernel<EPI, true, 2>), 80 * 1024);
      hipLaunchKernelGGL((conv3x3_halo2_kernel<EPI, true, 2>), grid, dim3(512), H2_LDS_BYTES, st, P);
      DH_LAUNCH_CHECK();
      return DH_OK;
    }
    if (P.cinit) {
      DH_LDS_OPTIN((&conv3x3_halo2_kernel<EPI, true, 1>), 80 * 1024);
      hipLaunchKernelGGL((conv3x3_halo2_kernel<EPI, true, 1>), grid, dim3(512), H2_LDS_BYTES, st, P);
      DH_LAUNCH_CHECK();
      return DH_OK;
    }
  }
#ifdef DH_ABLATION
  if constexpr (EPI == EPI_RELU) {                        // timing / power attribution (scripts/conv_power.py): wrong results
    const int abl = opts().conv_abl;
#define H2_ABL(v_)                                                                                                   \
    if (abl == v_) {                                                                                                 \
      DH_LDS_OPTIN((&conv3x3_halo2_kernel<EPI_RELU, true, false, v_>), 80 * 1024);                                   \
      hipLaunchKernelGGL((conv3x3_halo2_kernel<EPI_RELU, true, false, v_>), grid, dim3(512), H2_LDS_BYTES, st, P);   \
      DH_LAUNCH_CHECK();                                                                                             \
      return DH_OK;                                                                                                  \
    }
    H2_ABL(1) H2_ABL(2) H2_ABL(4) H2_ABL(6) H2_ABL(7) H2_ABL(8) H2_ABL(15) H2_ABL(16) H2_ABL(22)
#undef H2_ABL
  }
#endif
#ifndef DH_ABLATION       // (the measurement build's phase timestamps do not fit next to the second tile's registers: one-tile form there)
  if constexpr (EPI == EPI_RELU || EPI == EPI_HEADS0) {
    // two vertically adjacent pixel tiles per workgroup for the layers with a short main loop (see the kernel's TWO parameter)
    if (opts().conv_two_tiles && !P.cinit && P.Ctot <= opts().conv_two_tiles_maxc && M / BM >= 2048) {
      const unsigned pairs = (unsigned)((M / BM + 1) / 2);
      dim3 g2(pairs, (unsigned)P.ny);
      if (opts().conv_xcd) { P.xcd_tiles = (int)((pairs + 7) / 8); g2 = dim3((unsigned)P.xcd_tiles * 8 * P.ny, 1); }
      else P.xcd_tiles = 0;
      DH_LDS_OPTIN((&conv3x3_halo2_kernel<EPI, true, 0, 0, false, true>), 80 * 1024);
      hipLaunchKernelGGL((conv3x3_halo2_kernel<EPI, true, 0, 0, false, true>), g2, dim3(512), H2_LDS_BYTES, st, P);
      DH_LAUNCH_CHECK();
      return DH_OK;
    }
  }
#endif
  DH_LDS_OPTIN((&conv3x3_halo2_kernel<EPI>), 80 * 1024);
  hipLaunchKernelGGL((conv3x3_halo2_kernel<EPI>), grid, dim3(512), H2_LDS_BYTES, st, P);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

#ifdef DH_ABLATION
template <int EPI>
int launch_halo4(const ConvParams& P0, hipStream_t st) {
  ConvParams P = P0;
  const long M = (long)P.N * P.H * P.W;
  dim3 grid((unsigned)(M / BM), (unsigned)(P.CoutPad / 128));
  P.ny = (int)grid.y;
  if (opts().conv_xcd) { P.xcd_tiles = (int)((grid.x + 7) / 8); grid = dim3((unsigned)P.xcd_tiles * 8 * grid.y, 1); }
  if constexpr (EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q) {
    if (P.cinit) {
      DH_LDS_OPTIN((&conv3x3_halo4_kernel<EPI, true, true>), 80 * 1024);
      hipLaunchKernelGGL((conv3x3_halo4_kernel<EPI, true, true>), grid, dim3(256), H2_LDS_BYTES, st, P);
      DH_LAUNCH_CHECK();
      return DH_OK;
    }
  }
  DH_LDS_OPTIN((&conv3x3_halo4_kernel<EPI>), 80 * 1024);
  hipLaunchKernelGGL((conv3x3_halo4_kernel<EPI>), grid, dim3(256), H2_LDS_BYTES, st, P);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

#endif  // DH_ABLATION

// the rule under which droid_amd.update.pack_conv_halo lays a 64-cout 3x3 layer out as the halo2 layout of the layer padded to 128
// couts (read by BOTH sides: a launch that cannot take conv3x3_halo64_kernel must then not hand `weights_halo` to the first halo kernel)
inline bool halo64_layout(const ConvParams& P) {
  return opts().conv_halo64 && opts().conv_halo2 && P.KH == 3 && P.KW == 3 && P.CoutPad == 64 && P.Ctot % H2CK == 0;
}

// conv3x3_halo64_kernel: 3x3, Cout <= 64 = CoutPad, W == 64, H % 4 == 0, 32-channel segments, fp16 output through the staged epilogue;
// `weights_halo` must be the halo2 layout of the layer padded to 128 couts (the packer follows the same option)
inline bool halo64_ok(const ConvParams& P) {
  if (!halo64_layout(P)) return false;
  if (!opts().conv_halo64 || !opts().conv_halo || !P.wt_halo || P.KH != 3 || P.KW != 3 || P.W != 64 || P.H % 4 || P.CoutPad != 64) return false;
  if (P.Ctot % H2CK || P.cinit || P.out_f32 || P.gterm) return false;
  for (int i = 0; i < P.nseg; ++i) if (P.segC[i] % H2CK) return false;
  return P.epi == EPI_RELU ? staged_epilogue_ok<EPI_RELU>(P) : (P.epi == EPI_LINEAR && staged_epilogue_ok<EPI_LINEAR>(P));
}

template <int EPI>
int launch_halo64(const ConvParams& P0, hipStream_t st) {
  ConvParams P = P0;
  const long M = (long)P.N * P.H * P.W;
  dim3 grid((unsigned)(M / BM), 1);
  P.ny = 1;
  if (opts().conv_xcd) { P.xcd_tiles = (int)((grid.x + 7) / 8); grid = dim3((unsigned)P.xcd_tiles * 8, 1); }
  hipLaunchKernelGGL((conv3x3_halo64_kernel<EPI>), grid, dim3(256), H64_LDS_BYTES, st, P);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

// conv_gate64 (bit 0: the gate epilogues, bit 1: relu / linear / sigmoid layers): a layer that conv3x3_halo2_kernel would take, run on
// 64-cout tiles by conv3x3_halo64_kernel at three workgroups per CU.  Same eligibility as halo2_ok (the halo2 weight layout, W == 64,
// H % 4 == 0, 32-channel segments, staged epilogue); start values only in the accumulator-tile layout.
template <int EPI>
bool gate64_ok(const ConvParams& P);

template <int EPI>
int launch_gate64(const ConvParams& P0, hipStream_t st) {
  ConvParams P = P0;
  const long M = (long)P.N * P.H * P.W;
  dim3 grid((unsigned)(M / BM), (unsigned)(P.CoutPad / 64));
  P.ny = (int)grid.y;
  if (opts().conv_xcd) { P.xcd_tiles = (int)((grid.x + 7) / 8); grid = dim3((unsigned)P.xcd_tiles * 8 * grid.y, 1); }
  if (P.cinit) {
    if constexpr (EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q) {
      hipLaunchKernelGGL((conv3x3_halo64_kernel<EPI, 2>), grid, dim3(256), H64_LDS_BYTES, st, P);
      DH_LAUNCH_CHECK();
      return DH_OK;
    }
    return DH_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL((conv3x3_halo64_kernel<EPI>), grid, dim3(256), H64_LDS_BYTES, st, P);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

#ifdef DH_ABLATION
template <int EPI>
int launch_halo3(const ConvParams& P0, hipStream_t st) {
  ConvParams P = P0;
  const long M = (long)P.N * P.H * P.W;
  dim3 grid((unsigned)(M / 512), (unsigned)(P.CoutPad / 128));
  P.ny = (int)grid.y;
  if (opts().conv_xcd) { P.xcd_tiles = (int)((grid.x + 7) / 8); grid = dim3((unsigned)P.xcd_tiles * 8 * grid.y, 1); }
  if constexpr (EPI == EPI_LINEAR) {
    if (P.out_f32) {
      DH_LDS_OPTIN((&conv3x3_halo3_kernel<EPI, false>), 96 * 1024);
      hipLaunchKernelGGL((conv3x3_halo3_kernel<EPI, false>), grid, dim3(512), H3_LDS_BYTES, st, P);
      DH_LAUNCH_CHECK();
      return DH_OK;
    }
  }
  if constexpr (EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q) {
    if (P.cinit) {
      DH_LDS_OPTIN((&conv3x3_halo3_kernel<EPI, true, true>), 96 * 1024);
      hipLaunchKernelGGL((conv3x3_halo3_kernel<EPI, true, true>), grid, dim3(512), H3_LDS_BYTES, st, P);
      DH_LAUNCH_CHECK();
      return DH_OK;
    }
  }
  DH_LDS_OPTIN((&conv3x3_halo3_kernel<EPI>), 96 * 1024);
  hipLaunchKernelGGL((conv3x3_halo3_kernel<EPI>), grid, dim3(512), H3_LDS_BYTES, st, P);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

#endif  // DH_ABLATION

#ifdef DH_ABLATION
template <int EPI>
bool wino_ok(const ConvParams& P) {
  if (!P.wt_halo || P.KH != 3 || P.KW != 3 || P.W != 64 || P.H % 4 || P.CoutPad % 128 || P.Ctot % H2CK) return false;
  for (int i = 0; i < P.nseg; ++i) if (P.segC[i] % H2CK) return false;
  if (P.cinit && EPI != EPI_GRU_ZR && EPI != EPI_GRU_Q) return false;
  if (EPI != EPI_RELU && EPI != EPI_LINEAR && EPI != EPI_SIGMOID && EPI != EPI_GRU_ZR && EPI != EPI_GRU_Q) return false;
  return staged_epilogue_ok<EPI>(P);
}

template <int EPI>
int launch_wino(const ConvParams& P0, hipStream_t st) {
  ConvParams P = P0;
  const long M = (long)P.N * P.H * P.W;
  dim3 grid((unsigned)(M / BM), (unsigned)(P.CoutPad / 128));
  P.ny = (int)grid.y;
  if (opts().conv_xcd) { P.xcd_tiles = (int)((grid.x + 7) / 8); grid = dim3((unsigned)P.xcd_tiles * 8 * grid.y, 1); }
  if constexpr (EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q) {
    if (P.cinit) {
      DH_LDS_OPTIN((&conv3x3_wino_kernel<EPI, true>), WINO_LDS_BYTES);
      hipLaunchKernelGGL((conv3x3_wino_kernel<EPI, true>), grid, dim3(512), WINO_LDS_BYTES, st, P);
      DH_LAUNCH_CHECK();
      return DH_OK;
    }
  }
  DH_LDS_OPTIN((&conv3x3_wino_kernel<EPI>), WINO_LDS_BYTES);
  hipLaunchKernelGGL((conv3x3_wino_kernel<EPI>), grid, dim3(512), WINO_LDS_BYTES, st, P);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

#endif  // DH_ABLATION

// halo2 layout of `weights_halo`: cout tile 128, channel count a multiple of 32 (DH_CONV_HALO2=0, read by both sides,
// selects the 16-channel slabs of the first halo kernel instead)
bool dma_layout(int CoutPad, int Ctot);
bool halo2_layout(int CoutPad, int Ctot) {
  if (!opts().conv_halo2) return false;
  return !dma_layout(CoutPad, Ctot) && CoutPad % 128 == 0 && Ctot % H2CK == 0;
}

template <int EPI>
bool halo2_ok(const ConvParams& P) {
  if (!opts().conv_halo) return false;                      // 0: generic loop only
  if (!halo2_layout(P.CoutPad, P.Ctot)) return false;
  if (!P.wt_halo || P.KH != 3 || P.KW != 3 || P.W != 64 || P.H % 4) return false;
  for (int i = 0; i < P.nseg; ++i) if (P.segC[i] % H2CK) return false;
  if (P.cinit && EPI != EPI_GRU_ZR && EPI != EPI_GRU_Q) return false;
  if (EPI == EPI_LINEAR && P.out_f32) return true;          // per-element fp32 stores (launch_halo2)
  return staged_epilogue_ok<EPI>(P);
}

template <int EPI>
bool gate64_ok(const ConvParams& P) {
  const int want = (EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q) ? 1 : 2;
  if (!(opts().conv_gate64 & want) || P.out_f32 || P.glo_red) return false;       // (the fused global-context reduction needs all 128 channels in one workgroup)
  if (P.cinit && P.cinit_stride >= 0) return false;
  return halo2_ok<EPI>(P);
}

// Layout rule of `weights_halo` (the same rule is applied by the packer, droid_amd/update.py pack_conv_halo): 16-channel
// slabs of the halo kernel, unless DH_CONV_DMA=1 (read by both sides) AND the cout tile is 128 AND the channel count is a
// multiple of 64 with at least four chunks (with fewer, the un-overlapped prologue of the one-workgroup-per-CU kernel
// dominates): then the 64-channel swizzled slabs of the LDS-DMA kernel.
bool dma_layout(int CoutPad, int Ctot) {
  if (opts().conv_dma != 1) return false;
  return CoutPad % 128 == 0 && Ctot % DCK == 0 && Ctot >= 4 * DCK;
}

bool dma_ok(const ConvParams& P) {
  if (!opts().conv_halo || P.cinit) return false;           // 0: generic loop only
  if (!dma_layout(P.CoutPad, P.Ctot)) return false;
  if (!P.wt_halo || P.KH != 3 || P.KW != 3 || P.W != 64 || P.H % 4) return false;
  for (int i = 0; i < P.nseg; ++i) if (P.segC[i] % DCK || P.segS[i] % 8) return false;
  return true;
}

// 7x7 on 4 real input channels: weights_halo = [128][7][8][4] (droid_amd.update.pack_conv_7x7_c4)
// write-dominated launches (option conv_nt_out, default on): outputs of 64 MB and more leave with the non-temporal hint
inline ConvParams with_nt_out(const ConvParams& P) {
  ConvParams q = P;
  q.nt_out = (opts().conv_nt_out && !P.out_f32 && (long)P.N * P.H * P.W * P.out_stride * 2 >= (64L << 20)) ? 1 : 0;
  return q;
}
bool c7_ok(const ConvParams& P) {
  if (!opts().conv_halo || P.cinit || !P.wt_halo || P.KH != 7 || P.KW != 7 || P.nseg != 1 || P.segC[0] != 8) return false;
  if (P.W != 64 || P.H % 4 || P.CoutPad != 128 || P.Cout != 128 || P.epi != EPI_RELU) return false;
  return staged_epilogue_ok<EPI_RELU>(P);
}

// compute units of the current device (grid size of the persistent kernels); queried once per device: hipGetDeviceProperties takes milliseconds
int device_cus() {
  static int cu_count[64];
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
    if (!cu_count[dev]) { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cu_count[dev] = n; }
    if (cu_count[dev]) cus = cu_count[dev];
  }
  return cus;
}

// 1x1, one 128-channel input, at least four 64-cout tiles (fewer: the generic loop's independent tiles fill the chip better)
bool k1_ok(const ConvParams& P) {
  if (!opts().conv_halo || P.KH != 1 || P.KW != 1 || P.nseg != 1 || P.Ctot != 128 || P.Kpad != 128 || P.segS[0] % 8) return false;
  if (P.epi != EPI_LINEAR && P.epi != EPI_RELU) return false;
  if (P.cinit || P.gterm || P.CoutPad < 256 || P.CoutPad % 64 || ((long)P.N * P.H * P.W) % BM || ((uintptr_t)P.in[0]) % 16) return false;
  return P.epi == EPI_LINEAR ? staged_epilogue_ok<EPI_LINEAR>(P) : staged_epilogue_ok<EPI_RELU>(P);
}

bool glo_ok(const ConvParams& P) {
  if (!opts().conv_halo || P.epi != EPI_GLO || P.KH != 1 || P.KW != 1 || P.nseg != 1 || P.segC[0] != 128) return false;
  if (P.Cout != 128 || P.CoutPad != 128 || P.Kpad != 128 || P.cinit) return false;
  if (P.aux0 != P.in[0] || P.aux0_stride != P.segS[0]) return false;        // the gate multiplies the convolution's own input
  return ((long)P.H * P.W) % BM == 0;                                       // whole tiles inside one image
}

// small-Cout convolutions (CoutPad == 32 or 64): bound by re-reading the activations in the generic loop
bool halo_small_ok(const ConvParams& P, int bn) {
  if (P.cinit) return false;
  if (bn == 64 && halo64_layout(P)) return false;           // weights_halo holds the padded halo2 layout (conv3x3_halo64_kernel's): generic loop
  if (!P.wt_halo || P.KH != 3 || P.KW != 3 || P.W != 64 || P.H % 4 || P.CoutPad != bn) return false;
  for (int i = 0; i < P.nseg; ++i) if (P.segC[i] % (halo_afc(bn) * halo_ck(bn))) return false;
  return true;
}

bool halo_ok(const ConvParams& P) {
  // On by default (DH_CONV_HALO=0 falls back to the generic loop).  Measured on MI355X at 4096 edges inside the full
  // update iteration: 91.9 vs 104.1 ms for the update operator (step 100.8 vs 113.1 ms); in isolation at 1024 edges
  // 448->256: 7.3 vs 8.7 ms, 448->128: 3.7 vs 4.3 ms, 128->128: 1.37 vs 1.45 ms.  (A first version with 32-channel
  // chunks -- one workgroup per CU -- and strided weight reads was slower than the generic loop inside the iteration.)
  if (!opts().conv_halo || P.cinit) return false;
  if (dma_layout(P.CoutPad, P.Ctot) || halo2_layout(P.CoutPad, P.Ctot)) return false;     // weights_halo holds another layout
  if (!P.wt_halo || P.KH != 3 || P.KW != 3 || P.W != 64 || P.H % 4 || P.CoutPad % 128 || P.Ctot < 128) return false;
  for (int i = 0; i < P.nseg; ++i) if (P.segC[i] % halo_ck(128)) return false;
  return true;
}

template <int WM, int WN, int BN>
int launch(const ConvParams& P, hipStream_t st) {
  switch (P.epi) {
    case EPI_LINEAR: return launch_epi<WM, WN, BN, EPI_LINEAR>(P, st);
    case EPI_RELU: return launch_epi<WM, WN, BN, EPI_RELU>(P, st);
    case EPI_SIGMOID: return launch_epi<WM, WN, BN, EPI_SIGMOID>(P, st);
    case EPI_GRU_ZR: return launch_epi<WM, WN, BN, EPI_GRU_ZR>(P, st);
    case EPI_GRU_Q: return launch_epi<WM, WN, BN, EPI_GRU_Q>(P, st);
    case EPI_GLO: return launch_epi<WM, WN, BN, EPI_GLO>(P, st);
    case EPI_SOFTPLUS_001: return launch_epi<WM, WN, BN, EPI_SOFTPLUS_001>(P, st);
    case EPI_HEADS: return launch_epi<WM, WN, BN, EPI_HEADS>(P, st);
    default: break;
  }
  return DH_ERR_ARG;
}

// ---- global-context terms of the gates: the three 1x1 convolutions on the pixel mean (gru.py:21-27) as one GEMV -------
// out[e][n] = fp16( bias[n] + sum_k fp16(red[e][k] * scale) * wt[k][n] ),  n < N = 384 (z | r | q), k < 128; wt is k-major
// so that thread n reads it coalesced.  16 edges per workgroup share the weight reads.
constexpr int GV_EDGES = 16;
__global__ __launch_bounds__(384) void glo_gemv_kernel(const float* __restrict__ red, const float* __restrict__ wt,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       int E, int N, float scale) {
  __shared__ float4 s_g[128][GV_EDGES / 4];            // [k][edge]: one k-step reads its 16 edge values as four 16-byte broadcasts
  const int e0 = blockIdx.x * GV_EDGES, n = threadIdx.x;
  for (int o = threadIdx.x; o < GV_EDGES * 128; o += blockDim.x) {
    const int i = o >> 7, k = o & 127, e = e0 + i;
    reinterpret_cast<float*>(&s_g[k][0])[i] = e < E ? __half2float(__float2half(red[(long)e * 128 + k] * scale)) : 0.f;
  }
  __syncthreads();
  if (n >= N) return;
  float acc[GV_EDGES];
  const float b = bias[n];
#pragma unroll
  for (int i = 0; i < GV_EDGES; ++i) acc[i] = b;
#pragma unroll 4
  for (int k = 0; k < 128; ++k) {
    const float wv = wt[(long)k * N + n];
#pragma unroll
    for (int i4 = 0; i4 < GV_EDGES / 4; ++i4) {
      const float4 gk = s_g[k][i4];
      acc[4 * i4 + 0] = fmaf(gk.x, wv, acc[4 * i4 + 0]); acc[4 * i4 + 1] = fmaf(gk.y, wv, acc[4 * i4 + 1]);
      acc[4 * i4 + 2] = fmaf(gk.z, wv, acc[4 * i4 + 2]); acc[4 * i4 + 3] = fmaf(gk.w, wv, acc[4 * i4 + 3]);
    }
  }
#pragma unroll
  for (int i = 0; i < GV_EDGES; ++i)
    if (e0 + i < E) out[(long)(e0 + i) * N + n] = __half2float(__float2half(acc[i]));
}

// ---- corr_encoder.0 on the REFERENCE-layout lookup output: 1x1, 196 -> 128, relu (droid_net.py:83-86) -------------------
// The channel-last lookup pads every level from 49 to 56 channels so that this convolution can read 16-byte pieces; the
// reference layout [E,196,h,w] has no padding (392 instead of 448 bytes per pixel leave the lookup, and are read here),
// but its channels are HW apart.  This kernel turns the tile itself: thread (kg, pg) loads the 8 x 8 block
// (channels 8 kg .. +7) x (pixels 8 pg .. +7) as eight 16-byte pieces -- a wave touches 4 channel rows x 256 contiguous
// bytes per load --, transposes it in registers (32 v_perm_b32) and writes eight 16-byte pixel-major pieces to LDS, where
// the MFMA fragments are plain ds_read_b128 (row stride 432 B: conflict-free).  out[px][co] = relu(b + sum_k x[k][px] w[co][k])
// with i = cout, j = pixel on the 32x32x16 MFMA (K padded to 208), results staged through LDS for 16-byte stores.
// Persistent workgroups (one per CU, the weights stay in LDS): tile = 128 pixels of one edge, the next tile's 8 pieces
// are in flight while this one is multiplied.  HBM-bound: 392 + 256 bytes per pixel.
constexpr int C0_K = 196, C0_KP = 208, C0_ROWB = 432;              // row stride in bytes (216 halves)
constexpr int C0_TILE = 128;
constexpr int C0_LDS_BYTES = 2 * 128 * C0_ROWB;                    // weights + pixel tile (the output tile aliases the pixel tile)
constexpr int C0_OROWB = 272;                                      // output staging row: 128 couts x 2 B + 16

__global__ __launch_bounds__(512, 2) void corr0_nchw_kernel(const __half* __restrict__ x, const __half* __restrict__ wp,
                                                            const float* __restrict__ bias, __half* __restrict__ out,
                                                            int E, int HW) {
  extern __shared__ __half s_conv[];
  char* const Ws = reinterpret_cast<char*>(s_conv);
  char* const Xs = Ws + 128 * C0_ROWB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_per_edge = HW / C0_TILE;
  const long ntiles = (long)E * tiles_per_edge;

  // weights -> LDS once: [128 couts][208 halves] -> rows of 432 B
  for (int o = tid; o < 128 * (C0_KP / 8); o += 512) {
    const int co = o / (C0_KP / 8), pc = o - co * (C0_KP / 8);
    *reinterpret_cast<uint4*>(Ws + co * C0_ROWB + pc * 16) = *reinterpret_cast<const uint4*>(wp + (long)co * C0_KP + pc * 8);
  }
  // staging role: 16 lanes of a DS pass = 4 pixel groups x 4 channel groups (2-way bank conflicts at worst; global side:
  // 16 pixel groups of one channel row = 256 contiguous bytes)
  const int pg = ((lane >> 4) & 3) * 4 + (lane & 3), kg = wave * 4 + ((lane >> 2) & 3);
  const bool stager = kg * 8 < C0_K;                                // kg 0..24
  // MFMA role: wave -> 32 couts x 64 pixels
  const int co0 = (wave & 3) * 32, px0 = (wave >> 2) * 64;
  float bia[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) bia[q] = bias[co0 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)];

  uint4 P[8];
  auto fetch = [&](long tile) {
    const int e = (int)(tile / tiles_per_edge), p0 = (int)(tile - (long)e * tiles_per_edge) * C0_TILE;
    const __half* src = x + ((long)e * C0_K + kg * 8) * HW + p0 + pg * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r)
      P[r] = (stager && kg * 8 + r < C0_K) ? *reinterpret_cast<const uint4*>(src + (long)r * HW) : uint4{0u, 0u, 0u, 0u};
  };
  long tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    // 8 channels x 8 pixels -> 8 pixels x 8 channels
    if (stager) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const uint32_t sel = (p & 1) ? 0x07060302u : 0x05040100u;
        uint4 o;
        const uint32_t* P32 = reinterpret_cast<const uint32_t*>(P);
        o.x = __builtin_amdgcn_perm(P32[1 * 4 + (p >> 1)], P32[0 * 4 + (p >> 1)], sel);
        o.y = __builtin_amdgcn_perm(P32[3 * 4 + (p >> 1)], P32[2 * 4 + (p >> 1)], sel);
        o.z = __builtin_amdgcn_perm(P32[5 * 4 + (p >> 1)], P32[4 * 4 + (p >> 1)], sel);
        o.w = __builtin_amdgcn_perm(P32[7 * 4 + (p >> 1)], P32[6 * 4 + (p >> 1)], sel);
        *reinterpret_cast<uint4*>(Xs + (pg * 8 + p) * C0_ROWB + kg * 16) = o;
        if (kg == 24) *reinterpret_cast<uint4*>(Xs + (pg * 8 + p) * C0_ROWB + 25 * 16) = uint4{0u, 0u, 0u, 0u};   // k 200..207
      }
    }
    const long next = tile + gridDim.x;
    if (next < ntiles) fetch(next);
    __syncthreads();                                      // the pixel tile (and, first time, the weights) are in LDS
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
#pragma unroll
    for (int ks = 0; ks < C0_KP / 16; ++ks) {
      const int slot = 2 * ks + (lane >> 5);
      const half8 wf = *reinterpret_cast<const half8*>(Ws + (co0 + (lane & 31)) * C0_ROWB + slot * 16);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const half8 xf = *reinterpret_cast<const half8*>(Xs + (px0 + t * 32 + (lane & 31)) * C0_ROWB + slot * 16);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xf, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();                                      // every wave has read the pixel tile: the output tile may overwrite it
    char* const Os = Xs;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int px = px0 + t * 32 + (lane & 31);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = co0 + g * 8 + 4 * (lane >> 5);
        __half2 h01 = __floats2half2_rn(fmaxf(acc[t][4 * g + 0] + bia[4 * g + 0], 0.f), fmaxf(acc[t][4 * g + 1] + bia[4 * g + 1], 0.f));
        __half2 h23 = __floats2half2_rn(fmaxf(acc[t][4 * g + 2] + bia[4 * g + 2], 0.f), fmaxf(acc[t][4 * g + 3] + bia[4 * g + 3], 0.f));
        uint2 v; v.x = __builtin_bit_cast(uint32_t, h01); v.y = __builtin_bit_cast(uint32_t, h23);
        *reinterpret_cast<uint2*>(Os + px * C0_OROWB + co * 2) = v;
      }
    }
    __syncthreads();
    {
      const int e = (int)(tile / tiles_per_edge), p0 = (int)(tile - (long)e * tiles_per_edge) * C0_TILE;
      __half* dst = out + ((long)e * HW + p0) * 128;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int id = tid + 512 * i, px = id >> 4, sg = id & 15;
        *reinterpret_cast<uint4*>(dst + (long)px * 128 + sg * 8) = *reinterpret_cast<const uint4*>(Os + px * C0_OROWB + sg * 16);
      }
    }
    __syncthreads();                                      // the output tile is out: the next pixel tile may be written
  }
}

}  // namespace

// -DDH_ABLATION builds: per-workgroup phase timestamps of the 3x3 kernels into `buf` (8 x uint64 per workgroup, `capacity`
// workgroups; nullptr switches them off again).  Launches with more workgroups than `capacity` run without timestamps.
namespace { unsigned long long* g_conv_ts = nullptr; long g_conv_ts_cap = 0; }
extern "C" int dh_conv_set_timestamps(void* buf, long capacity) {
#ifdef DH_ABLATION
  if (capacity < 0 || (buf && ((uintptr_t)buf & 7))) return DH_ERR_ARG;
  g_conv_ts = (unsigned long long*)buf; g_conv_ts_cap = buf ? capacity : 0;
  return DH_OK;
#else
  (void)buf; (void)capacity;
  return DH_ERR_UNSUPPORTED;                                 // measurement hook of the -DDH_ABLATION build
#endif
}

extern "C" int dh_conv2d_nhwc_f16_ex(const void* const* inputs, const int* in_channels, const int* in_strides, int n_inputs,
                                     const void* weights, const void* weights_halo, const float* bias,
                                     int N, int H, int W, int KH, int KW, int Cout, int CoutPad, int Kpad, int epilogue,
                                     void* out, int out_is_f32, int out_stride,
                                     const float* gterm, const void* aux0, int aux0_stride, const void* aux1, int aux1_stride,
                                     float* red, const float* cinit, const int64_t* cinit_idx, int cinit_stride, int cinit_off,
                                     dh_stream_t stream) {
  return dh_conv2d_nhwc_f16_ex2(inputs, in_channels, in_strides, n_inputs, weights, weights_halo, DH_CONV_LAYOUT_AUTO, bias, N, H, W, KH, KW,
                                Cout, CoutPad, Kpad, epilogue, out, out_is_f32, out_stride, gterm, aux0, aux0_stride, aux1, aux1_stride, red,
                                cinit, cinit_idx, cinit_stride, cinit_off, stream);
}

extern "C" int dh_conv2d_nhwc_f16_ex2(const void* const* inputs, const int* in_channels, const int* in_strides, int n_inputs,
                                      const void* weights, const void* weights_halo, int weights_layout, const float* bias,
                                      int N, int H, int W, int KH, int KW, int Cout, int CoutPad, int Kpad, int epilogue,
                                      void* out, int out_is_f32, int out_stride,
                                      const float* gterm, const void* aux0, int aux0_stride, const void* aux1, int aux1_stride,
                                      float* red, const float* cinit, const int64_t* cinit_idx, int cinit_stride, int cinit_off,
                                      dh_stream_t stream) {
  return dh_conv2d_nhwc_f16_ex3(inputs, in_channels, in_strides, n_inputs, weights, weights_halo, weights_layout, bias, N, H, W, KH, KW,
                                Cout, CoutPad, Kpad, epilogue, out, out_is_f32, out_stride, gterm, aux0, aux0_stride, aux1, aux1_stride, red,
                                cinit, cinit_idx, cinit_stride, cinit_off, nullptr, nullptr, nullptr, stream);
}

extern "C" int dh_conv2d_nhwc_f16_ex3(const void* const* inputs, const int* in_channels, const int* in_strides, int n_inputs,
                                      const void* weights, const void* weights_halo, int weights_layout, const float* bias,
                                      int N, int H, int W, int KH, int KW, int Cout, int CoutPad, int Kpad, int epilogue,
                                      void* out, int out_is_f32, int out_stride,
                                      const float* gterm, const void* aux0, int aux0_stride, const void* aux1, int aux1_stride,
                                      float* red, const float* cinit, const int64_t* cinit_idx, int cinit_stride, int cinit_off,
                                      const void* glo_weights, const float* glo_bias, float* glo_red,
                                      dh_stream_t stream) {
  if (weights_layout != DH_CONV_LAYOUT_AUTO && weights_layout != DH_CONV_LAYOUT_WINO) return DH_ERR_ARG;
  if (glo_red && (epilogue != EPI_GRU_Q || !glo_weights || !glo_bias || Cout != 128 || CoutPad != 128 || ((uintptr_t)glo_weights) % 16)) return DH_ERR_ARG;
  if (n_inputs < 1 || n_inputs > MAXSEG || !inputs || !in_channels || !weights || !bias) return DH_ERR_ARG;
  if (N < 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0 || !(KH & 1) || !(KW & 1)) return DH_ERR_ARG;
  if (Cout <= 0 || CoutPad < Cout || CoutPad % 32 || Kpad <= 0 || Kpad % BK) return DH_ERR_ARG;
  if (epilogue < 0 || epilogue > EPI_HEADS0) return DH_ERR_ARG;
  ConvParams P{};
  int ctot = 0;
  for (int i = 0; i < n_inputs; ++i) {
    if (!inputs[i] || in_channels[i] <= 0 || in_channels[i] % 8) return DH_ERR_ARG;
    P.in[i] = (const __half*)inputs[i]; P.segC[i] = in_channels[i]; ctot += in_channels[i];
    P.segS[i] = in_strides ? in_strides[i] : in_channels[i];
    if (P.segS[i] < in_channels[i] || P.segS[i] % 8 || ((uintptr_t)inputs[i]) % 16) return DH_ERR_ARG;
  }
  for (int i = n_inputs; i < MAXSEG; ++i) { P.in[i] = P.in[0]; P.segC[i] = 1 << 30; P.segS[i] = P.segS[0]; }
  P.nseg = n_inputs; P.Ctot = ctot;
  P.Kreal = KH * KW * ctot;
  if (Kpad < P.Kreal) return DH_ERR_ARG;
  if (epilogue == EPI_GLO && (!red || !aux0)) return DH_ERR_ARG;
  if (epilogue == EPI_GRU_ZR && !aux0) return DH_ERR_ARG;
  if (epilogue == EPI_GRU_Q && (!aux0 || !aux1)) return DH_ERR_ARG;
  if (epilogue == EPI_HEADS0 && (!red || !aux1)) return DH_ERR_ARG;
  if (epilogue != EPI_GLO && epilogue != EPI_HEADS0 && !out) return DH_ERR_ARG;
  if (out_is_f32 < 0 || out_is_f32 > 3) return DH_ERR_ARG;
  if (cinit && cinit_stride >= 0 && (!cinit_idx || cinit_stride < cinit_off + Cout || cinit_off < 0)) return DH_ERR_ARG;
  if (cinit && cinit_stride < 0 && (!cinit_idx || (-cinit_stride) % 128 || cinit_off < 0 || cinit_off % 128 || Cout % 128 || cinit_off + Cout > -cinit_stride))
    return DH_ERR_ARG;                                       // accumulator-tile layout: whole 128-cout tiles
  if (out_is_f32 == 3 && (Cout % 128 || CoutPad != Cout || epilogue != EPI_LINEAR)) return DH_ERR_ARG;
  if (N == 0) return DH_OK;
  P.cinit = cinit; P.cinit_idx = cinit_idx; P.cinit_stride = cinit_stride; P.cinit_off = cinit_off;
  P.wt = (const __half*)weights; P.wt_halo = (const __half*)weights_halo; P.bias = bias;
  P.N = N; P.H = H; P.W = W; P.KH = KH; P.KW = KW; P.Cout = Cout; P.CoutPad = CoutPad; P.Kpad = Kpad; P.epi = epilogue;
  P.stride = 1; P.Hin = H; P.Win = W;
  P.out = out; P.out_f32 = out_is_f32; P.out_stride = out_stride;
#ifdef DH_ABLATION
  {
    const int ct = (opts().conv_gate64 && KH == 3 && CoutPad % 128 == 0) ? 64 : 128;      // (conv_gate64: 64-cout tiles, launch_gate64)
    const long nwg = ((long)N * H * W / 256) * ((CoutPad + ct - 1) / ct);
    P.ts = nwg <= g_conv_ts_cap ? g_conv_ts : nullptr;
    // per-step stamps of 256 workgroups from the middle of the launch, in the 2048 rows behind the per-workgroup block
    P.ts_step = (P.ts && nwg + 2048 <= g_conv_ts_cap) ? g_conv_ts + nwg * 8 : nullptr;
    P.ts_step_wg0 = (unsigned)(nwg / 2);
  }
#endif
  P.gterm = gterm; P.aux0 = (const __half*)aux0; P.aux0_stride = aux0_stride;
  P.aux1 = (const __half*)aux1; P.aux1_stride = aux1_stride; P.red = red;
  P.glo_wt = (const __half*)glo_weights; P.glo_bias = glo_bias; P.glo_red = glo_red;
  hipStream_t st = (hipStream_t)stream;
  if (P.glo_red) {                                           // the fused reduction exists in conv3x3_halo2_kernel's q-gate form only
    if (weights_layout != DH_CONV_LAYOUT_AUTO || !halo2_ok<EPI_GRU_Q>(P)) return DH_ERR_UNSUPPORTED;
    return launch_halo2<EPI_GRU_Q>(P, st);
  }
  if ((P.cinit && P.cinit_stride < 0) || P.out_f32 == 3) {    // accumulator-tile layout: conv3x3_halo2_kernel only (writer and reader share its tiling)
    if (weights_layout != DH_CONV_LAYOUT_AUTO) return DH_ERR_UNSUPPORTED;
    switch (P.epi) {
      case EPI_LINEAR: if (P.out_f32 == 3 && !P.cinit && halo2_ok<EPI_LINEAR>(P)) return launch_halo2<EPI_LINEAR>(P, st); break;
      case EPI_GRU_ZR: if (P.out_f32 != 3 && gate64_ok<EPI_GRU_ZR>(P)) return launch_gate64<EPI_GRU_ZR>(P, st);
                       if (P.out_f32 != 3 && halo2_ok<EPI_GRU_ZR>(P)) return launch_halo2<EPI_GRU_ZR>(P, st); break;
      case EPI_GRU_Q: if (P.out_f32 != 3 && gate64_ok<EPI_GRU_Q>(P)) return launch_gate64<EPI_GRU_Q>(P, st);
                      if (P.out_f32 != 3 && halo2_ok<EPI_GRU_Q>(P)) return launch_halo2<EPI_GRU_Q>(P, st); break;
      default: break;
    }
    return DH_ERR_UNSUPPORTED;
  }
  if (weights_layout == DH_CONV_LAYOUT_WINO) {              // prototype: F(2,3) along x; weights_halo = droid_amd.update.pack_conv_wino
#ifdef DH_ABLATION
    switch (P.epi) {
      case EPI_LINEAR: if (!P.out_f32 && wino_ok<EPI_LINEAR>(P)) return launch_wino<EPI_LINEAR>(P, st); break;
      case EPI_RELU: if (wino_ok<EPI_RELU>(P)) return launch_wino<EPI_RELU>(P, st); break;
      case EPI_SIGMOID: if (wino_ok<EPI_SIGMOID>(P)) return launch_wino<EPI_SIGMOID>(P, st); break;
      case EPI_GRU_ZR: if (wino_ok<EPI_GRU_ZR>(P)) return launch_wino<EPI_GRU_ZR>(P, st); break;
      case EPI_GRU_Q: if (wino_ok<EPI_GRU_Q>(P)) return launch_wino<EPI_GRU_Q>(P, st); break;
      default: break;
    }
#endif
    return DH_ERR_UNSUPPORTED;                               // (release builds carry no prototype kernels: -DDH_ABLATION)
  }
#ifdef DH_ABLATION
  if (dma_ok(P)) {
    switch (P.epi) {
      case EPI_LINEAR: return launch_dma<EPI_LINEAR>(P, st);
      case EPI_RELU: return launch_dma<EPI_RELU>(P, st);
      case EPI_SIGMOID: return launch_dma<EPI_SIGMOID>(P, st);
      case EPI_GRU_ZR: return launch_dma<EPI_GRU_ZR>(P, st);
      case EPI_GRU_Q: return launch_dma<EPI_GRU_Q>(P, st);
      default: break;
    }
  }
#endif
  if (halo64_ok(P)) return P.epi == EPI_RELU ? launch_halo64<EPI_RELU>(P, st) : launch_halo64<EPI_LINEAR>(P, st);
#ifdef DH_ABLATION
  if (opts().conv_halo4 && !(P.epi == EPI_LINEAR && P.out_f32)) {                      // measurement variant: four 64 x 128 waves per workgroup
    switch (P.epi) {
      case EPI_LINEAR: if (halo2_ok<EPI_LINEAR>(P)) return launch_halo4<EPI_LINEAR>(P, st); break;
      case EPI_RELU: if (halo2_ok<EPI_RELU>(P)) return launch_halo4<EPI_RELU>(P, st); break;
      case EPI_SIGMOID: if (halo2_ok<EPI_SIGMOID>(P)) return launch_halo4<EPI_SIGMOID>(P, st); break;
      case EPI_GRU_ZR: if (halo2_ok<EPI_GRU_ZR>(P)) return launch_halo4<EPI_GRU_ZR>(P, st); break;
      case EPI_GRU_Q: if (halo2_ok<EPI_GRU_Q>(P)) return launch_halo4<EPI_GRU_Q>(P, st); break;
      default: break;
    }
  }
  if (opts().conv_halo3 && P.H % 8 == 0 && ((long)P.N * P.H * P.W) % 512 == 0) {       // opt-in: the 512-pixel-tile form of the same kernel
    switch (P.epi) {
      case EPI_LINEAR: if (halo2_ok<EPI_LINEAR>(P)) return launch_halo3<EPI_LINEAR>(P, st); break;
      case EPI_RELU: if (halo2_ok<EPI_RELU>(P)) return launch_halo3<EPI_RELU>(P, st); break;
      case EPI_SIGMOID: if (halo2_ok<EPI_SIGMOID>(P)) return launch_halo3<EPI_SIGMOID>(P, st); break;
      case EPI_GRU_ZR: if (halo2_ok<EPI_GRU_ZR>(P)) return launch_halo3<EPI_GRU_ZR>(P, st); break;
      case EPI_GRU_Q: if (halo2_ok<EPI_GRU_Q>(P)) return launch_halo3<EPI_GRU_Q>(P, st); break;
      case EPI_HEADS0: if (halo2_ok<EPI_HEADS0>(P)) return launch_halo3<EPI_HEADS0>(P, st); break;
      default: break;
    }
  }
#endif
  if (opts().conv_gate64) {                                  // 64-cout tiles, three workgroups per CU (A/B switch, see launch_gate64)
    switch (P.epi) {
      case EPI_LINEAR: if (gate64_ok<EPI_LINEAR>(P)) return launch_gate64<EPI_LINEAR>(P, st); break;
      case EPI_RELU: if (gate64_ok<EPI_RELU>(P)) return launch_gate64<EPI_RELU>(P, st); break;
      case EPI_SIGMOID: if (gate64_ok<EPI_SIGMOID>(P)) return launch_gate64<EPI_SIGMOID>(P, st); break;
      case EPI_GRU_ZR: if (gate64_ok<EPI_GRU_ZR>(P)) return launch_gate64<EPI_GRU_ZR>(P, st); break;
      case EPI_GRU_Q: if (gate64_ok<EPI_GRU_Q>(P)) return launch_gate64<EPI_GRU_Q>(P, st); break;
      default: break;
    }
  }
  switch (P.epi) {
    case EPI_LINEAR: if (halo2_ok<EPI_LINEAR>(P)) return launch_halo2<EPI_LINEAR>(P, st); break;
    case EPI_RELU: if (halo2_ok<EPI_RELU>(P)) return launch_halo2<EPI_RELU>(P, st); break;
    case EPI_SIGMOID: if (halo2_ok<EPI_SIGMOID>(P)) return launch_halo2<EPI_SIGMOID>(P, st); break;
    case EPI_GRU_ZR: if (halo2_ok<EPI_GRU_ZR>(P)) return launch_halo2<EPI_GRU_ZR>(P, st); break;
    case EPI_GRU_Q: if (halo2_ok<EPI_GRU_Q>(P)) return launch_halo2<EPI_GRU_Q>(P, st); break;
    case EPI_HEADS0: if (halo2_ok<EPI_HEADS0>(P)) return launch_halo2<EPI_HEADS0>(P, st); return DH_ERR_UNSUPPORTED;
    default: break;
  }
  if (halo_ok(P)) {
    switch (P.epi) {
      case EPI_LINEAR: return launch_halo<EPI_LINEAR, 128>(P, st);
      case EPI_RELU: return launch_halo<EPI_RELU, 128>(P, st);
      case EPI_SIGMOID: return launch_halo<EPI_SIGMOID, 128>(P, st);
      case EPI_GRU_ZR: return launch_halo<EPI_GRU_ZR, 128>(P, st);
      case EPI_GRU_Q: return launch_halo<EPI_GRU_Q, 128>(P, st);
      default: break;
    }
  }
  if (k1_ok(P) && opts().conv_k1_half) {                     // 128-pixel tiles, two workgroups per CU
    const dim3 grid((unsigned)((long)P.N * P.H * P.W / 128));
    constexpr int lds = (128 + 64) * K1_LD * 2 + 128 * (64 + 8) * 2;
    if (P.epi == EPI_LINEAR) {
      DH_LDS_OPTIN((&conv1x1_c128_kernel<EPI_LINEAR, 128>), 80 * 1024);
      hipLaunchKernelGGL((conv1x1_c128_kernel<EPI_LINEAR, 128>), grid, dim3(256), lds, st, with_nt_out(P));
    } else {
      DH_LDS_OPTIN((&conv1x1_c128_kernel<EPI_RELU, 128>), 80 * 1024);
      hipLaunchKernelGGL((conv1x1_c128_kernel<EPI_RELU, 128>), grid, dim3(256), lds, st, with_nt_out(P));
    }
    DH_LAUNCH_CHECK();
    return DH_OK;
  }
  if (k1_ok(P)) {
    const dim3 grid((unsigned)((long)P.N * P.H * P.W / BM));
    if (P.epi == EPI_LINEAR) {
      DH_LDS_OPTIN((&conv1x1_c128_kernel<EPI_LINEAR>), 160 * 1024);
      hipLaunchKernelGGL((conv1x1_c128_kernel<EPI_LINEAR>), grid, dim3(512), K1_LDS_BYTES, st, with_nt_out(P));
    } else {
      DH_LDS_OPTIN((&conv1x1_c128_kernel<EPI_RELU>), 160 * 1024);
      hipLaunchKernelGGL((conv1x1_c128_kernel<EPI_RELU>), grid, dim3(512), K1_LDS_BYTES, st, with_nt_out(P));
    }
    DH_LAUNCH_CHECK();
    return DH_OK;
  }
  if (glo_ok(P)) {
    DH_LDS_OPTIN(&glo_reduce_kernel, 80 * 1024);
    hipLaunchKernelGGL(glo_reduce_kernel, dim3((unsigned)((long)P.N * P.H * P.W / BM)), dim3(512), BM * GLD * 2, st, P);
    DH_LAUNCH_CHECK();
    return DH_OK;
  }
  if (c7_ok(P) && opts().conv_c7_w16) {                      // sixteen 32 px x 64 cout waves per workgroup: eight waves per SIMD
    DH_LDS_OPTIN((&conv7x7_c4_w16_kernel<EPI_RELU>), 80 * 1024);
    hipLaunchKernelGGL((conv7x7_c4_w16_kernel<EPI_RELU>), dim3((unsigned)((long)P.N * P.H * P.W / BM)), dim3(1024), C7_LDS_BYTES, st, P);
    DH_LAUNCH_CHECK();
    return DH_OK;
  }
  if (c7_ok(P) && opts().conv_c7_pp && P.epi == EPI_RELU) {  // persistent workgroups, two wave groups alternating roles
    const long ntiles = (long)P.N * P.H * P.W / BM;
    const unsigned grid = (unsigned)std::min<long>((ntiles + 1) / 2, device_cus());
    DH_LDS_OPTIN((&conv7x7_c4_pp_kernel<EPI_RELU>), 160 * 1024);
    hipLaunchKernelGGL((conv7x7_c4_pp_kernel<EPI_RELU>), dim3(grid), dim3(1024), C7P_LDS_BYTES, st, P);
    DH_LAUNCH_CHECK();
    return DH_OK;
  }
  if (c7_ok(P) && opts().conv_c7_split) {                    // 64-cout halves, four workgroups per CU
    hipLaunchKernelGGL((conv7x7_c4_kernel<EPI_RELU, 64>), dim3((unsigned)((long)P.N * P.H * P.W / BM) * 2), dim3(256), C7_LDS64_BYTES, st, with_nt_out(P));
    DH_LAUNCH_CHECK();
    return DH_OK;
  }
  if (c7_ok(P)) {
    DH_LDS_OPTIN((&conv7x7_c4_kernel<EPI_RELU>), 80 * 1024);
    hipLaunchKernelGGL((conv7x7_c4_kernel<EPI_RELU>), dim3((unsigned)((long)P.N * P.H * P.W / BM)), dim3(512), C7_LDS_BYTES, st, with_nt_out(P));
    DH_LAUNCH_CHECK();
    return DH_OK;
  }
  if (halo_small_ok(P, 64)) {
    switch (P.epi) {
      case EPI_LINEAR: return launch_halo<EPI_LINEAR, 64>(P, st);
      case EPI_RELU: return launch_halo<EPI_RELU, 64>(P, st);
      default: break;
    }
  }
  if (halo_small_ok(P, 32)) {
    switch (P.epi) {
      case EPI_LINEAR: return launch_halo<EPI_LINEAR, 32>(P, st);
      case EPI_HEADS: return launch_halo<EPI_HEADS, 32>(P, st);
      case EPI_SOFTPLUS_001: return launch_halo<EPI_SOFTPLUS_001, 32>(P, st);
      default: break;
    }
  }
  if (P.cinit) return DH_ERR_UNSUPPORTED;                    // accumulator start values: 3x3, W == 64, H % 4 == 0, gate epilogues only
  if (CoutPad >= 128) return launch<64, 64, 128>(P, st);
  if (CoutPad >= 64) return launch<64, 32, 64>(P, st);
  return launch<32, 32, 32>(P, st);
}

extern "C" int dh_conv2d_nhwc_f16(const void* const* inputs, const int* in_channels, const int* in_strides, int n_inputs,
                                  const void* weights, const void* weights_halo, const float* bias,
                                  int N, int H, int W, int KH, int KW, int Cout, int CoutPad, int Kpad, int epilogue,
                                  void* out, int out_is_f32, int out_stride,
                                  const float* gterm, const void* aux0, int aux0_stride, const void* aux1, int aux1_stride,
                                  float* red, dh_stream_t stream) {
  return dh_conv2d_nhwc_f16_ex(inputs, in_channels, in_strides, n_inputs, weights, weights_halo, bias, N, H, W, KH, KW, Cout,
                               CoutPad, Kpad, epilogue, out, out_is_f32, out_stride, gterm, aux0, aux0_stride, aux1, aux1_stride,
                               red, nullptr, nullptr, 0, 0, stream);
}

// Stride-2 "same" convolution of ONE dense NHWC input (k = 1, 3, 7; pad = k / 2): out[n, y, x] = sum_taps w . in[n, 2y + dy - pad, 2x + dx - pad]
// -- exactly the even positions of the stride-1 result, computed at a quarter of its work and without the strided copy
// (extractor.py:140 conv1, :24 / :151 the residual blocks' first convolution and 1x1 down-sampling).  Hin, Win even; linear / relu.
extern "C" int dh_conv2d_s2_nhwc_f16(const void* input, int C, int in_stride, const void* weights, const float* bias,
                                     int N, int Hin, int Win, int KH, int KW, int Cout, int CoutPad, int Kpad, int epilogue,
                                     void* out, int out_stride, dh_stream_t stream) {
  if (!input || !weights || !bias || !out) return DH_ERR_ARG;
  if (N < 0 || Hin <= 0 || Win <= 0 || (Hin & 1) || (Win & 1) || KH <= 0 || KW <= 0 || !(KH & 1) || !(KW & 1)) return DH_ERR_ARG;
  if (C <= 0 || C % 8 || in_stride < C || in_stride % 8 || ((uintptr_t)input) % 16) return DH_ERR_ARG;
  if (Cout <= 0 || CoutPad < Cout || CoutPad % 32 || Kpad <= 0 || Kpad % BK || Kpad < KH * KW * C || out_stride < Cout) return DH_ERR_ARG;
  if (epilogue != EPI_LINEAR && epilogue != EPI_RELU) return DH_ERR_UNSUPPORTED;
  if (N == 0) return DH_OK;
  ConvParams P{};
  for (int i = 0; i < MAXSEG; ++i) { P.in[i] = (const __half*)input; P.segC[i] = i == 0 ? C : 1 << 30; P.segS[i] = in_stride; }
  P.nseg = 1; P.Ctot = C; P.Kreal = KH * KW * C;
  P.wt = (const __half*)weights; P.wt_halo = nullptr; P.bias = bias;
  P.N = N; P.H = Hin / 2; P.W = Win / 2; P.KH = KH; P.KW = KW; P.Cout = Cout; P.CoutPad = CoutPad; P.Kpad = Kpad; P.epi = epilogue;
  P.stride = 2; P.Hin = Hin; P.Win = Win;
  P.out = out; P.out_f32 = 0; P.out_stride = out_stride;
  hipStream_t st = (hipStream_t)stream;
  if (CoutPad >= 128) return launch<64, 64, 128>(P, st);
  if (CoutPad >= 64) return launch<64, 32, 64>(P, st);
  return launch<32, 32, 32>(P, st);
}

extern "C" int dh_corr0_nchw_f16(const void* x, const void* wp, const float* bias, void* out, int E, int HW, dh_stream_t stream) {
  if (E < 0 || HW <= 0 || HW % C0_TILE) return DH_ERR_ARG;
  if (E == 0) return DH_OK;
  if (!x || !wp || !bias || !out) return DH_ERR_ARG;
  DH_LDS_OPTIN(&corr0_nchw_kernel, C0_LDS_BYTES);
  const long ntiles = (long)E * (HW / C0_TILE);
  const unsigned grid = (unsigned)std::min<long>(ntiles, device_cus());
  hipLaunchKernelGGL(corr0_nchw_kernel, dim3(grid), dim3(512), C0_LDS_BYTES, (hipStream_t)stream, (const __half*)x, (const __half*)wp,
                     bias, (__half*)out, E, HW);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_glo_gemv(const float* red, const float* wt, const float* bias, float* out, int E, int N, float scale,
                           dh_stream_t stream) {
  if (E < 0 || N <= 0 || N > 384) return DH_ERR_ARG;
  if (E == 0) return DH_OK;
  if (!red || !wt || !bias || !out) return DH_ERR_ARG;
  hipLaunchKernelGGL(glo_gemv_kernel, dim3((unsigned)((E + GV_EDGES - 1) / GV_EDGES)), dim3(384), 0, (hipStream_t)stream, red, wt,
                     bias, out, E, N, scale);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_heads_gather(const float* partials, const float* bias4, float* dw, int N, int H, int W, int n_cout_tiles, dh_stream_t stream) {
  return dh_heads_gather_ex(partials, bias4, dw, N, H, W, n_cout_tiles, 0, stream);
}

extern "C" int dh_heads_gather_ex(const float* partials, const float* bias4, float* dw, int N, int H, int W, int n_cout_tiles, int mode, dh_stream_t stream) {
  if (mode != 0 && mode != 1) return DH_ERR_ARG;
  if (N < 0 || H <= 0 || W <= 0 || n_cout_tiles < 1) return DH_ERR_ARG;
  if (W != 64 || H % 4) return DH_ERR_UNSUPPORTED;           // the partials are per tile of four full 64-pixel rows (EPI_HEADS0)
  if (N == 0) return DH_OK;
  if (!partials || !bias4 || !dw) return DH_ERR_ARG;
  const long M = (long)N * H * W;
  if (mode == 1) hipLaunchKernelGGL(heads_gather_kernel<1>, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partials, bias4, dw, M, n_cout_tiles, H);
  else hipLaunchKernelGGL(heads_gather_kernel<0>, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partials, bias4, dw, M, n_cout_tiles, H);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_segment_mean_f16(const void* x, const int64_t* order, const int64_t* seg_off, void* out,
                                   int K, long row_elems, dh_stream_t stream) {
  if (K < 0 || row_elems <= 0 || row_elems % 8) return DH_ERR_ARG;
  if (K == 0) return DH_OK;
  if (!x || !order || !seg_off || !out) return DH_ERR_ARG;
  const long row8 = row_elems / 8;
  hipLaunchKernelGGL(segment_mean_kernel, dim3((unsigned)((row8 + 255) / 256), K), dim3(256), 0, (hipStream_t)stream,
                     (const __half*)x, order, seg_off, (__half*)out, row8);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
