// Process-wide switches of libdroid_hip (A/B measurements, fallbacks for the parity tests).  Read from the environment
// ONCE, when the library is loaded; afterwards only dh_set_option() changes them -- nothing on a launch path calls getenv().
#pragma once

namespace dh {

struct Options {
  int debug;            // DH_DEBUG=1: print HIP errors to stderr
  int gram_strips;      // DH_GRAM_STRIPS (0 = automatic): pixel strips per depth block in the Schur-complement kernel
  int chol_regpanel;    // DH_CHOL_REGPANEL (1): look-ahead step keeps the unfinished panel in MFMA accumulator registers; 0 = LDS panel
  int chol_lookahead;   // DH_CHOL_LOOKAHEAD (1): one fused launch per block column of the Cholesky; 0 = two launches; 2 / 3 (-DDH_ABLATION builds) = dataflow schedule, one persistent launch (3: LDS-DMA operands, grouped acquires)
  int conv_epi_staged;  // DH_CONV_EPI_STAGED (1): LDS-staged convolution epilogues
  int conv_halo;        // DH_CONV_HALO (1): 3x3 halo-tile kernels; 0 = generic loop only
  int conv_halo2;       // DH_CONV_HALO2 (1): weights by LDS-DMA for 128-cout tiles; 0 = first halo kernel
  int conv_dma;         // DH_CONV_DMA (0): opt-in one-workgroup-per-CU LDS-DMA experiment
  int conv_xcd;         // DH_CONV_XCD (1): XCD-aware workgroup order of the 3x3 halo2 kernel (conv.hip xcd_decode)
  int dma_var;          // DH_DMA_VAR (0): timing ablations of that experiment
  int pyr_build_chunk;  // DH_PYR_BUILD=chunk (0): first form of the pyramid build kernel
  int pyr_build_waves;  // DH_PYR_BUILD_WAVES (8): waves per workgroup of the row-ring build kernel at w = 64 (4 = the first form)
  int pyr_build_tm;     // DH_PYR_BUILD_TM (0; measured neutral, profiles/r06_pyr_build_ab.txt): level 0 of a 64-column image with tile-major wave roles (a target tile's B fragments read by 2 (1) waves instead of 4); 0 = row-pair-major (rounds 2-5); same records
  int pyr_build_xcd;    // DH_PYR_BUILD_XCD (0): the row-ring build's workgroups of one edge on ONE XCD: fabric reads 11.1 -> 0.9 GB per 256 edges, time 0 .. +7 % (same-line reads queue on one L2; profiles/r06_v_pyr_build_pmc.txt); same records
  int pyr_build_dual;   // DH_PYR_BUILD_DUAL (1): w = 64, one workgroup of 16 waves builds two adjacent source blocks from one staged copy of each target row (row fetches halved: -5 %); same records
  int pyr_lds_pad;      // DH_PYR_LDS_PAD (0): measurement only -- extra dynamic LDS bytes for the 8-wave row-ring build (4096 = one workgroup per CU: +20 %)
  int conv_nt_out;      // DH_CONV_NT_OUT (1): the stem's and the upmask head's output stores (>= 64 MB) carry the non-temporal hint (-6 % on the stem); same results
  int lookup_mode;      // DH_LOOKUP_MODE (0): pyramid lookup variant: 1 = nt tap loads; 2 / 3 = timing ablations (no stores / no loads, wrong results); fused kernel: also 5 (a quarter of the MFMAs) and 6 = synchronous twin (every tap batch waited for at issue; same results, used by the tests)
  int lookup_fused;     // DH_LOOKUP_FUSED (1): read by the host side (droid_amd.factor_graph, bench.py): lookup and the correlation encoder's first layer in one kernel (dh_corr_pyramid_lookup_corr0); 0 = dh_corr_pyramid_lookup + dh_corr0_nchw_f16
  int lookup_fill;      // DH_LOOKUP_FILL (0): fused lookup refills its tap registers 0 = by half level (two batches of 4 window rows), 1 = window row by window row (-DDH_ABLATION builds); same results
  int lookup_mix;       // DH_LOOKUP_MIX (1): fused lookup interpolates through v_fma_mix_f32 / v_fma_mixlo_f16 (no conversion instructions); 0 = conversions spelled out, same results
  int altcorr_v1;       // DH_ALTCORR_V1 (0): first form of the MFMA alt-correlation kernel (register staging) for A/B runs
  int conv_wino;        // DH_CONV_WINO (0): read by the host packer only (droid_amd.update): gate convolutions through the Winograd F(2,3) prototype
  int conv_halo3;       // DH_CONV_HALO3 (0): 3x3 / 128-cout convolutions through the 512-pixel-tile form (conv3x3_halo3_kernel: 128 x 64 per wave, one workgroup per CU) instead of conv3x3_halo2_kernel; same weight layout, same results up to the accumulation order of nothing (identical k order)
  int cinit_tiled;      // DH_CINIT_TILED (1): host side (UpdateModule): the gates' per-frame context term in the accumulator-tile layout (16-byte start-value loads); 0 = pixel-major
  int conv_halo64;      // DH_CONV_HALO64 (1): 3x3 layers with 64 couts through conv3x3_halo64_kernel (0 = first halo kernel); read by the weight packer too
  int conv_gate64;      // DH_CONV_GATE64 (0; round-6 A/B): layers conv3x3_halo2_kernel takes, on 64-cout tiles in conv3x3_halo64_kernel at THREE workgroups per CU: bit 0 = the GRU gates, bit 1 = relu / linear / sigmoid layers; bit-identical results
  int glo_fused;        // DH_GLO_FUSED (1): host side (UpdateModule / FactorGraph / bench.py): the next iteration's global-context reduction inside the q gate's launch (dh_conv2d_nhwc_f16_ex3) instead of a pass over the hidden state at the start of that iteration; 0 = always the stand-alone kernel
  int conv_two_tiles;   // DH_CONV_TWO_TILES (0; round 6): relu / heads layers of conv3x3_halo2_kernel with at most conv_two_tiles_maxc input channels (128) and at least 2048 pixel tiles: TWO vertically adjacent pixel tiles per workgroup, the second tile's first fetches under the first tile's epilogue; same results
  int conv_two_tiles_maxc;
  int conv_k1_half;     // DH_CONV_K1_HALF (round 6): the 128 -> 576 upmask head on 128-pixel tiles by four waves, two workgroups per CU (conv1x1_c128_kernel<.., 128>) instead of one 256-pixel workgroup; same results
  int conv_c7_pp;       // DH_CONV_C7_PP (round 6): the stem as persistent 16-wave workgroups whose two wave groups alternate between multiplying a tile and parking / storing the previous one (conv7x7_c4_pp_kernel); equal results
  int conv_c7_w16;      // DH_CONV_C7_W16 (round 6): the stem with sixteen waves of 32 px x 64 couts per workgroup (<= 64 registers: eight waves per SIMD) instead of eight of 64 x 64; equal results
  int conv_c7_split;    // DH_CONV_C7_SPLIT (round 6): the flow encoder's 7x7 stem as 64-cout halves, four workgroups of four waves per CU (conv7x7_c4_kernel<.., 64>) instead of two of eight; same results
  int eta_fused;        // DH_ETA_FUSED (1): host side (UpdateModule): GraphAgg's eta head as the fused second layer of agg.conv2's launch (EPI_HEADS0 with an output pointer + dh_heads_gather_ex mode 1) instead of a 3x3 convolution with one output channel
  int conv_halo4;       // DH_CONV_HALO4 (0): -DDH_ABLATION builds only: the second kernel with four 64 x 128 waves per workgroup
  int conv_abl;         // DH_CONV_ABL (0): -DDH_ABLATION builds only: timing-ablation mask of conv3x3_halo2_kernel (wrong results)
  int ba_strict;        // DH_BA_STRICT (1): dh_ba / dh_ba_ex / dh_ba_build return DH_ERR_ARG on bad indices / eta rows; the host waits for the argument check (the call's first kernel), not for the solve; 2 = the check behind a stream synchronisation (rounds 1-5); 0 = asynchronous, no signal -- a flagged call applies no update in any mode
};

Options& opts();

}  // namespace dh
