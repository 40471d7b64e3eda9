/*
 * droid_hip.h -- C ABI of libdroid_hip.so: the MI355X (gfx950) implementation of DROID-SLAM's
 * dense bundle-adjustment update operator.
 *
 * This is the drop-in boundary: plain pointers + sizes + a HIP stream, no torch types.  Each entry
 * point replaces one function that the reference registers in its `droid_backends` extension
 * (reference src/droid.cpp:246-259); the torch binding in droid-slam_amd/csrc/droid_backends.cpp
 * re-creates that Python module on top of these calls.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in `_host`;
 *   - tensors are dense, row-major ("contiguous") in the shapes given below;
 *   - `stream` is a hipStream_t (passed as void* so that C callers need no HIP headers); all work is
 *     enqueued on it and the call returns without synchronising;
 *   - poses are world->camera [tx,ty,tz,qx,qy,qz,qw]; tangent vectors are (tau,phi); updates are
 *     LEFT multiplications exp(xi)*T (reference src/droid_kernels.cu:886-904);
 *   - return value: DH_OK (0) or a DH_ERR_* code; dh_status_string() names it.  Argument errors are
 *     detected before anything is enqueued.  Like the reference, a failed Cholesky factorisation is
 *     NOT an error: the pose update of that iteration is zero (src/droid_kernels.cu:1211-1219).
 */
#ifndef DROID_HIP_H
#define DROID_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DH_OK 0
#define DH_ERR_ARG 1
#define DH_ERR_WORKSPACE 2
#define DH_ERR_LAUNCH 3
#define DH_ERR_UNSUPPORTED 4

/* element types of correlation volumes / feature maps */
#define DH_F16 0
#define DH_F32 1
#define DH_F64 2   /* dh_corr_index_fwd / _bwd only (the reference dispatches double there too, correlation_kernels.cu:146,167) */

typedef void* dh_stream_t;

const char* dh_version(void);
const char* dh_status_string(int status);

/* Process-wide switches (A/B measurements and the fallbacks the parity tests exercise).  Initialised from the
 * environment variables DH_<NAME> once, when the library is loaded; nothing on a launch path reads the environment.
 * names: "debug", "chol_lookahead", "conv_epi_staged", "conv_halo", "conv_halo2", "conv_dma", "dma_var",
 * "pyr_build_chunk", "ba_strict" (1 by default: dh_ba / dh_ba_ex / dh_ba_build return DH_ERR_ARG when an edge index lies
 * outside the frame buffer or eta does not have one row per depth block -- the reference would read out of bounds / fail
 * its broadcast, src/droid_kernels.cu:1407.  The flags are final after the call's first kernel: the call returns once THAT
 * kernel has run (an event behind a copy to pinned memory), with the rest of its work still queued on the stream -- like any
 * other entry point it does NOT synchronise the stream.  2 = the check behind a stream synchronisation after the last launch
 * (rounds 1-5); 0 = fully asynchronous call, no signal to the caller.  In every mode a flagged call applies NO update).
 * Returns DH_ERR_ARG for an unknown name. */
int dh_set_option(const char* name, int value);
int dh_get_option(const char* name, int* value);
/* number of dh_set_option calls that changed a value so far: lets a holder of option-dependent state (the kernel-ordered
 * weight copies of dh_conv2d_nhwc_f16, whose layout follows "conv_dma" / "conv_halo2") notice cheaply that it must re-check */
int dh_options_epoch(void);

/* ------------------------------------------------------------------------------------------------
 * Correlation-volume lookup.  Replaces corr_index_forward / corr_index_backward
 * (reference src/droid.cpp:175-196 -> src/correlation_kernels.cu:127-186).
 *   volume [N,h1,w1,h2,w2] (dtype), coords [N,2,h1,w1] f32 (x,y) -> corr [N,2r+1,2r+1,h1,w1] (dtype)
 * corr is fully overwritten (no pre-zeroing needed).  First window index = x offset.
 */
int dh_corr_index_fwd(const void* volume, const float* coords, void* corr, int dtype,
                      int N, int h1, int w1, int h2, int w2, int radius, dh_stream_t stream);
/* volume_grad [N,h1,w1,h2,w2] (dtype) is fully overwritten with the adjoint of the lookup. */
int dh_corr_index_bwd(const float* coords, const void* corr_grad, void* volume_grad, int dtype,
                      int N, int h1, int w1, int h2, int w2, int radius, dh_stream_t stream);

/* All-pairs correlation volume in the reference layout and its 2x2 average pooling, any image size.  Replaces
 * CorrBlock.corr + F.avg_pool2d of CorrBlock.__init__ (reference droid_slam/modules/corr.py:23-38,63-71).
 *   fmap1, fmap2 [E,C,h,w] (dtype) -> volume [E,h,w,h,w] (dtype): sum_c (f1/4)(f2/4), fp32 accumulation
 *   in [n_slices,h2,w2] -> out [n_slices,h2/2,w2/2] (floor), mean of each 2x2 block */
int dh_corr_volume_build(const void* fmap1, const void* fmap2, void* volume, int dtype,
                         int E, int C, int h, int w, dh_stream_t stream);
int dh_corr_volume_pool(const void* in, void* out, int dtype, long n_slices, int h2, int w2, dh_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * On-the-fly ("alt") correlation.  Replaces altcorr_forward / altcorr_backward
 * (reference src/droid.cpp:198-226 -> src/altcorr_kernel.cu:132-225).
 *   fmap1 [B,N1,C,H,W], fmap2 [B,N2,C,H2,W2] (dtype), coords [B,M,2,H,W] f32, ii,jj [M] i64
 *   -> corr [B,M,2r+1,2r+1,H,W] (dtype), x offset outer (the layout the reference returns as a
 *   permuted view, altcorr_kernel.cu:171, here written densely).  Features are scaled by 1/4 each.
 */
int dh_altcorr_fwd(const void* fmap1, const void* fmap2, const float* coords,
                   const int64_t* ii, const int64_t* jj, void* corr, int dtype,
                   int B, int N1, int N2, int C, int H, int W, int H2, int W2, int M, int radius,
                   dh_stream_t stream);
/* Same lookup (radius 3) on the fp16 MFMA from channel-last features (the layout droid_amd.corr.AltCorrBlock keeps):
 *   fmap1 [N1,H,W,128] f16, fmap2 [N2,H2,W2,128] f16, coords [M,2,H,W] f32 (already in fmap2's resolution),
 *   -> corr [M,7,7,H,W] f16, x offset outer.  H % 8 == 0, W % 8 == 0, C == 128 (else DH_ERR_UNSUPPORTED). */
int dh_altcorr_fwd_nhwc(const void* fmap1, const void* fmap2, const float* coords,
                        const int64_t* ii, const int64_t* jj, void* corr,
                        int N1, int N2, int C, int H, int W, int H2, int W2, int M, dh_stream_t stream);
/* one pyramid level of AltCorrBlock.__call__ (corr.py:104-117) written in place: coords [M,2,H,W] are FULL-resolution
 * coordinates, divided by 2^level here (exact); edge m's 49 planes go to corr + m * corr_stride_m (elements), so the four
 * levels of an edge land side by side in one [M,196,H,W] tensor (corr = base + level * 49 * H * W, corr_stride_m = 196 * H * W)
 * without the stack / flatten copies of the reference formulation. */
int dh_altcorr_fwd_nhwc_level(const void* fmap1, const void* fmap2, const float* coords,
                              const int64_t* ii, const int64_t* jj, void* corr,
                              int N1, int N2, int C, int H, int W, int H2, int W2, int M,
                              int level, long corr_stride_m, dh_stream_t stream);
/* corr_grad [B,M,2r+1,2r+1,H,W] f32 (x offset outer); fmap1_grad/fmap2_grad f32, ACCUMULATED into
 * (caller zero-fills), shapes of fmap1/fmap2. */
int dh_altcorr_bwd(const void* fmap1, const void* fmap2, const float* coords,
                   const int64_t* ii, const int64_t* jj, const float* corr_grad,
                   float* fmap1_grad, float* fmap2_grad, int dtype,
                   int B, int N1, int N2, int C, int H, int W, int H2, int W2, int M, int radius,
                   dh_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * MI355X-native correlation pyramid (own layout, built and consumed only by this library).
 * Replaces CorrBlock.__init__ / CorrBlock.__call__ (reference droid_slam/modules/corr.py:23-50,63-71):
 * per level the all-pairs contraction (f1/4)^T pool_l(f2/4) on the fp16 MFMA, written in a
 * displacement-skewed, 8x8-source-block interleaved layout (DESIGN.md "correlation pyramid layout"),
 * then ONE lookup kernel over all 4 levels.
 *   fmap1, fmap2 [E,C=128,h,w] f16 -> pyramid (opaque f16, dh_corr_pyramid_bytes(E,h,w) bytes; edge-major,
 *     so concatenating / indexing edges is a plain copy of dh_corr_pyramid_bytes(1,h,w)-sized records)
 *   coords [E,h,w,2] f32 (x,y; what CorrBlock.__call__ receives) -> out [E,4*49,h,w] f16,
 *     channel = level*49 + xoff*7 + yoff (corr.py:46-50).  radius 3, 4 levels, h % 8 == 0, w in {16,32,64}
 *     (otherwise DH_ERR_ARG: use dh_corr_index_fwd on a reference-layout volume).
 *   build needs dh_corr_pyramid_workspace_bytes(E,h,w) bytes of scratch (channel-last feature copies).
 */
size_t dh_corr_pyramid_bytes(int E, int h, int w);
size_t dh_corr_pyramid_workspace_bytes(int E, int h, int w);
int dh_corr_pyramid_build(const void* fmap1, const void* fmap2, void* pyramid, void* workspace,
                          size_t workspace_bytes, int E, int C, int h, int w, dh_stream_t stream);
/* an h_real x w_real image (any size that fits) on an h x w canvas that satisfies the layout (h % 8 == 0, w in {16,32,64}):
 * fmap1 / fmap2 are canvas-sized and ZERO outside the image; pooled levels are cut at (h_real >> l) x (w_real >> l) as
 * avg_pool2d floors them (reference modules/corr.py:36: 30x40 -> 15x20 -> 7x10 -> 3x5).  Lookups then run on the canvas
 * (coords of canvas pixels outside the image are don't-cares, their outputs are to be dropped). */
int dh_corr_pyramid_build_canvas(const void* fmap1, const void* fmap2, void* pyramid, void* workspace,
                                 size_t workspace_bytes, int E, int C, int h, int w, int h_real, int w_real, dh_stream_t stream);
/* Frame-level form of the build (round 5): the channel-last transpose and the three pooled levels of a feature map are computed
 * once per FRAME -- fmaps [F,C,h,w] f16 -> prepared [F][sum_l (h>>l)(w>>l)][C] f16, dh_corr_pyramid_prepared_bytes(F, h, w) bytes --
 * and the pyramid of edge e is built from the prepared rows of frames idx1[e] (source) and idx2[e] (target; for the reference's
 * stereo pairs the second camera's maps are frames of the same tensor: factor_graph.py:128-133 fmaps[jj, c]).  Same records, bit
 * for bit, as dh_corr_pyramid_build_canvas on the gathered per-edge features; no per-edge feature copies, no workspace. */
size_t dh_corr_pyramid_prepared_bytes(int F, int h, int w);
int dh_corr_pyramid_prepare_frames(const void* fmaps, void* prepared, int F, int C, int h, int w, int h_real, int w_real,
                                   dh_stream_t stream);
int dh_corr_pyramid_build_indexed(const void* prepared, const int64_t* idx1, const int64_t* idx2, void* pyramid,
                                  int F, int E, int h, int w, dh_stream_t stream);
int dh_corr_pyramid_lookup(const void* pyramid, const float* coords, void* out,
                           int E, int h, int w, dh_stream_t stream);
/* same lookup written channel-last for the update operator of this library (dh_conv2d_nhwc_f16):
 *   out [4,E,h,w,56] f16 (level-planar), channel = yoff*7 + xoff, channels 49..55 are zero. */
int dh_corr_pyramid_lookup_nhwc(const void* pyramid, const float* coords, void* out,
                                int E, int h, int w, dh_stream_t stream);
/* the lookup fused with the layer that consumes it, the correlation encoder's Conv2d(196, 128, 1) + ReLU
 * (reference droid_slam/droid_net.py:96-100, first layer of corr_encoder on the output of corr.py:46-50): the 196 window
 * samples are rounded to fp16 as dh_corr_pyramid_lookup stores them and multiplied out of registers on the matrix cores.
 *   wpk  [13][128][16] f16: k-step l*3+s (s < 3) = channels kk = 16s..16s+15 of level l, kk = yoff*7 + xoff (reference
 *        input channel l*49 + xoff*7 + yoff); k-step 12: k = l < 4 is channel kk = 48 of level l, the rest zero.
 *   bias [128] f32;  out [E,h,w,128] f16 (channel-last). */
int dh_corr_pyramid_lookup_corr0(const void* pyramid, const float* coords, const void* wpk, const float* bias, void* out,
                                 int E, int h, int w, dh_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Dense bundle adjustment.  Replaces ba (reference src/droid.cpp:93-122 -> ba_cuda,
 * src/droid_kernels.cu:1323-1443).
 *   poses [num_frames,7] f32      IN/OUT (poses t0..t1-1 are retracted in place)
 *   disps [num_frames,ht,wd] f32  IN/OUT (disps of every frame in unique(arange(t0,t1) U ii))
 *   intrinsics [4] f32 (fx,fy,cx,cy), disps_sens [num_frames,ht,wd] f32
 *   targets, weights [E,2,ht,wd] f32;  eta [K,ht,wd] f32, K = |unique(arange(t0,t1) U ii)|
 *   ii, jj [E] i64
 *   dx_out [t1-t0,6] f32, dz_out [K,ht*wd] f32 (may be NULL): updates of the LAST iteration
 * Everything (index building, Schur complement, fp64 Cholesky, back-substitution, retraction) runs
 * on the device with no host round trip.  `workspace` must hold dh_ba_workspace_bytes(...) bytes and be 256-byte aligned
 * (DH_ERR_WORKSPACE otherwise).
 */
size_t dh_ba_workspace_bytes(int num_frames, int n_edges, int ht, int wd, int t0, int t1, int motion_only);
int dh_ba(float* poses, float* disps, const float* intrinsics, const float* disps_sens,
          const float* targets, const float* weights, const float* eta,
          const int64_t* ii, const int64_t* jj,
          int num_frames, int n_edges, int n_eta_rows, int ht, int wd,
          int t0, int t1, int iterations, float lm, float ep, int motion_only,
          float* dx_out, float* dz_out, void* workspace, size_t workspace_bytes, dh_stream_t stream);

/* Split form used by the edge-sharded multi-GPU solver (one RCCL all-reduce between the two calls):
 * dh_ba_build fills the reduced camera system of THIS rank's edges,
 *   Hsys [6P,6P] f64 (row-major, LOWER triangle -- all the factorisation reads; the strict upper triangle is not
 *   assembled), bsys [6P] f64, and keeps per-frame depth terms in the
 *   workspace; dh_ba_finish damps, solves, back-substitutes the depths of the frames this rank owns
 *   and retracts.  dh_ba() == dh_ba_build + dh_ba_finish per iteration. */
/* dh_ba with a PER-PIXEL weight of the sensor-depth prior (BASELINE.json configs[4], "per-pixel depth-confidence weights";
 * SURVEY.md Q10b): alpha [num_frames,ht,wd] f32 replaces the reference's constant 0.05 (src/droid_kernels.cu:1405-1408) where
 * disps_sens > 0 and alpha > 0 (zero confidence = no prior at that pixel, damped by eta like a pixel without sensor depth):
 * C = sum Cii + alpha,  w = sum bz - alpha * (disps - disps_sens).  alpha == NULL or alpha == 0.05
 * everywhere is exactly dh_ba.  The positional signature of the reference's `ba` is fixed (src/droid.cpp:93-108), hence a
 * new entry point (droid_backends.ba_ex). */
int dh_ba_ex(float* poses, float* disps, const float* intrinsics, const float* disps_sens, const float* alpha,
             const float* targets, const float* weights, const float* eta,
             const int64_t* ii, const int64_t* jj,
             int num_frames, int n_edges, int n_eta_rows, int ht, int wd,
             int t0, int t1, int iterations, float lm, float ep, int motion_only,
             float* dx_out, float* dz_out, void* workspace, size_t workspace_bytes, dh_stream_t stream);

/* shape of the system buffer behind Hsys_out: [rows, cols] f64, row-major; rows [0, 6P) x cols [0, 6P) = Hsys,
 * row `cols` = bsys (the right-hand side travels as an extra block row), everything else padding. */
int dh_ba_system_shape(int t0, int t1, int* rows, int* cols);
int dh_ba_build(const float* poses, const float* disps, const float* intrinsics, const float* disps_sens,
                const float* targets, const float* weights, const float* eta,
                const int64_t* ii, const int64_t* jj,
                int num_frames, int n_edges, int n_eta_rows, int ht, int wd,
                int t0, int t1, int motion_only,
                double** Hsys_out, double** bsys_out, void* workspace, size_t workspace_bytes, dh_stream_t stream);
/* Views into a workspace after dh_ba_build (non-motion-only): Qinv = 1 / C and w of the depth blocks, [K, ht*wd] f32, rows
 * ordered like kx [K] (sorted source frames; *K_out points at the DEVICE int K) = `C`, `w` of the reference's ba_cuda
 * (src/droid_kernels.cu:1407-1408; the stage its accum_cuda calls feed, :957-1007). */
int dh_ba_depth_blocks(const void* workspace, size_t workspace_bytes, int num_frames, int n_edges, int ht, int wd,
                       int t0, int t1, const float** Qinv_out, const float** w_out, const int** kx_out, const int** K_out);
int dh_ba_finish(float* poses, float* disps, const int64_t* jj,
                 int num_frames, int n_edges, int ht, int wd, int t0, int t1,
                 float lm, float ep, int motion_only, float* dx_out, float* dz_out,
                 void* workspace, size_t workspace_bytes, dh_stream_t stream);

/* Edge-sharded solver (north_star: "global BA partitioned by edge batches across the 8 GPUs of one node with RCCL
 * all-reduce of the 6x6 pose Hessian blocks and residuals"; the reference has no such path -- its ba is droid.cpp:93-122).
 * Per Gauss-Newton iteration a rank calls
 *   dh_ba_build_shard   = dh_ba_build WITHOUT the ba_strict host synchronisation (a rank-local error ahead of the collective
 *                         would leave the other ranks waiting in it);
 *   dh_ba_pack_blocks   the n_blocks lower-triangular 6x6 blocks (bp[b], bq[b]) of its partial system, the rhs and two
 *                       status words -> packed f64 [36 n_blocks + 6 P + 2]; status[0] = this rank's argument flag,
 *                       status[1] = host_flags != 0 (the host's "block pattern is stale");
 *   (one RCCL all-reduce, SUM, of `packed`)
 *   dh_ba_unpack_blocks writes blocks and rhs back; a non-zero status word of ANY rank turns the iteration into a no-op
 *                       update on EVERY rank (the host reads the two words once, after the last iteration, and raises /
 *                       falls back on all ranks together);
 *   dh_ba_finish_owned  = dh_ba_finish, moving only the depths of the frames [own_lo, own_hi) this rank owns.
 * Dense fallback (no pattern): all-reduce the whole system and a 2-word status buffer filled / applied by
 * dh_ba_exchange_flags(set = 0 / 1). */
int dh_ba_build_shard(const float* poses, const float* disps, const float* intrinsics, const float* disps_sens,
                      const float* targets, const float* weights, const float* eta,
                      const int64_t* ii, const int64_t* jj,
                      int num_frames, int n_edges, int n_eta_rows, int ht, int wd,
                      int t0, int t1, int motion_only,
                      double** Hsys_out, double** bsys_out, void* workspace, size_t workspace_bytes, dh_stream_t stream);
/* dh_ba_build_shard with dh_ba_ex's per-pixel weight of the sensor-depth prior (alpha [num_frames,ht,wd] f32, NULL = the
 * reference's constant 0.05, src/droid_kernels.cu:1405-1408): BASELINE.json configs[4] (stereo + per-pixel depth-confidence
 * weights) is defined on the edge-sharded path.  The prior of a frame enters on the rank that holds the frame's edges. */
int dh_ba_build_shard_ex(const float* poses, const float* disps, const float* intrinsics, const float* disps_sens,
                         const float* alpha, const float* targets, const float* weights, const float* eta,
                         const int64_t* ii, const int64_t* jj,
                         int num_frames, int n_edges, int n_eta_rows, int ht, int wd,
                         int t0, int t1, int motion_only,
                         double** Hsys_out, double** bsys_out, void* workspace, size_t workspace_bytes, dh_stream_t stream);
size_t dh_ba_packed_len(int n_blocks, int t0, int t1);
int dh_ba_pack_blocks(const void* workspace, size_t workspace_bytes, int num_frames, int n_edges, int ht, int wd,
                      int t0, int t1, int motion_only, const int32_t* bp, const int32_t* bq, int n_blocks,
                      int host_flags, double* packed, dh_stream_t stream);
int dh_ba_unpack_blocks(void* workspace, size_t workspace_bytes, int num_frames, int n_edges, int ht, int wd,
                        int t0, int t1, int motion_only, const int32_t* bp, const int32_t* bq, int n_blocks,
                        const double* packed, dh_stream_t stream);
int dh_ba_exchange_flags(void* workspace, size_t workspace_bytes, int num_frames, int n_edges, int ht, int wd,
                         int t0, int t1, int motion_only, int host_flags, double* status, int set, dh_stream_t stream);
int dh_ba_finish_owned(float* poses, float* disps, const int64_t* jj,
                       int num_frames, int n_edges, int ht, int wd, int t0, int t1,
                       float lm, float ep, int motion_only, int own_lo, int own_hi, float* dx_out, float* dz_out,
                       void* workspace, size_t workspace_bytes, dh_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Encoder glue (csrc/norm.hip): the normalisation / residual steps of the feature and context encoders
 * (reference droid_slam/modules/extractor.py:6-50,120-198), whose convolutions go through dh_conv2d_nhwc_f16.
 *   y = act( instance_norm(x) )   (normalize = 1: per image and channel over the H*W pixels, eps 1e-5, fp32 statistics)
 *   y = act( x + residual )       (normalize = 0; residual may be NULL: plain activation)
 * x, residual, y [N,H,W,C] f16 channel-last, C % 8 == 0; stats_ws [N*C*2] f32 scratch; relu = 0 / 1. */
int dh_norm_act_nhwc_f16(const void* x, const void* residual, void* y, float* stats_ws, int N, int HW, int C,
                         int normalize, int relu, dh_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Factor-graph side kernels (csrc/graph.hip): the glue of FactorGraph.update / add_proximity_factors and
 * DepthVideo.upsample (reference droid_slam/factor_graph.py:214-263,346-412, depth_video.py:155-159, droid_net.py:21-35).
 *   dh_motion_features: coords1, target [E,ht,wd,2] f32 -> flow [E,ht,wd,8] f16 =
 *     cat(coords1 - coords0, target - coords1).clamp(-64, 64), channels 4..7 zero (factor_graph.py:221-222)
 *   dh_ba_inputs: coords1 [E,ht,wd,2], dw [E,ht,wd,4] f32 = (delta, weight) of the update operator ->
 *     target = coords1 + delta, weight [E,ht,wd,2] (optional, NULL to skip) and the same in ba's [E,2,ht,wd] layout
 *     (factor_graph.py:233-234,253-254)
 *   dh_cvx_upsample: disp [K,ht,wd] f32, mask [K,ht,wd,576] f16 channel-last (channel = (dy*3+dx)*64 + sy*8 + sx)
 *     -> out [K,8ht,8wd] f32 (droid_net.py:21-35: softmax over the 9 neighbours, weights rounded to fp16)
 *   dh_proximity_nms: the candidate selection of add_proximity_factors on the device.  dist [(t-t0) x (t-t1)] f32 is
 *     MODIFIED.  stage 0: dist[i - rad < j] = dist[dist > 100] = inf, temporal neighbours / stereo self edges and the
 *     |di|+|dj| <= max(min(|i-j|-2, nms), 0) neighbourhood of every edge in edges_i/j [n_edges] i64 = inf.
 *     stage 1 (after the caller sorted the masked matrix: sorted [n] f32 values, order [n] i64 indices): greedy walk in
 *     ascending distance (entries suppressed since the sort are skipped), every accepted
 *     (i,j) is written as (i,j),(j,i) to out_edges [2*max_new][2] i64 and suppresses its neighbourhood; stops at
 *     dist > thresh, when n_es0 + 2*accepted > max_factors (max_factors > 0), or at max_new; out_count[0] = accepted. */
int dh_motion_features(const float* coords1, const float* target, void* flow, int E, int ht, int wd, dh_stream_t stream);
/* x [N,Hc,Wc,C] f16 channel-last (C % 8 == 0, 16-byte aligned): zero the pixels outside the h x w image in the canvas' top-left
 * corner.  Image sizes outside the production tiling (w != 64 or h % 4 != 0; TUM's 30x40, ...) run the update operator
 * (reference droid_net.py:78-143) on a zero-padded canvas with this mask between the layers. */
int dh_canvas_mask_f16(void* x, int N, int Hc, int Wc, int C, int h, int w, dh_stream_t stream);
int dh_ba_inputs(const float* coords1, const float* dw, float* target, float* weight, float* target_ba, float* weight_ba,
                 int E, int ht, int wd, dh_stream_t stream);
int dh_cvx_upsample(const float* disp, const void* mask, float* out, int K, int ht, int wd, dh_stream_t stream);
int dh_proximity_nms(float* dist, const float* sorted, const int64_t* order, const int64_t* edges_i, const int64_t* edges_j, int n_edges,
                     int t0, int t1, int t, int rad, int nms, float thresh, int max_factors, int n_es0, int stereo,
                     int64_t* out_edges, int max_new, int* out_count, int stage, dh_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * ConvGRU update operator building block.  Replaces the cuDNN convolutions + elementwise GRU algebra of
 * UpdateModule / ConvGRU / GraphAgg (reference droid_slam/droid_net.py:78-143,44-75, modules/gru.py:5-33).
 * Implicit-GEMM convolution, stride 1, "same" padding, fp16 MFMA with fp32 accumulation, NHWC activations.
 *   inputs[i] [N,H,W,in_channels[i]] f16 (1..4 tensors, concatenated along channels on the fly; channels % 8 == 0;
 *   in_strides[i] = elements between consecutive pixels, NULL = dense: lets a segment be a channel slice of a wider tensor)
 *   weights [CoutPad, Kpad] f16 with k = (dy*KW + dx) * sum(in_channels) + c, zero padded (Kpad % 64 == 0,
 *   CoutPad % 32 == 0);  bias [CoutPad] f32
 *   weights_halo (optional, NULL = none): second copy for the 3x3 halo-tile fast path (W == 64, H % 4 == 0, every
 *   in_channels[i] % 32 == 0; CoutPad == 32, 64 or a multiple of 128), laid out [CoutPad/BN][Ctot/16][9 taps][BN][16]
 *   f16 with BN = CoutPad if CoutPad is 32 or 64, else 128
 *   out [N,H,W,out_stride] (f16, or f32 holding fp16-rounded values when out_is_f32), channels [0,Cout) written
 *   epilogue: 0 linear, 1 relu, 2 sigmoid,
 *     3 GRU z|r  (Cout = 256: z = sigmoid(.+g) for cout < 128, r*net for cout >= 128; aux0 = net [.,128]),
 *     4 GRU q    (out = (1-z)*net + z*tanh(.+g); aux0 = net, aux1 = z|r tensor; out may alias aux0),
 *     5 global-context reduction (red[N,Cout] f32 += sum_pixels sigmoid(.)*aux0; nothing else is written),
 *     6 0.01*softplus, 7 (delta_x, delta_y, sigmoid w_x, sigmoid w_y),
 *     8 first head layer fused with the second layer's channel contraction (3x3, W == 64, H % 4 == 0, Cout = CoutPad a
 *       multiple of 128; DH_ERR_UNSUPPORTED otherwise): aux1 = second-layer weights packed [CoutPad/128][64][128] f16
 *       (row = tap*4 + output, 36 used), red = partial sums [CoutPad/128][N*H/4 tiles of four rows][6][64][4] f32 (what the
 *       tile's pixels contribute to the output rows -1 .. 4 relative to the tile), `out` unused; finish with dh_heads_gather
 *   gterm [N,CoutPad] f32 or NULL: per-image additive term (the ConvGRU's 1x1 global-context convolutions).
 */
int dh_conv2d_nhwc_f16(const void* const* inputs, const int* in_channels, const int* in_strides, int n_inputs,
                       const void* weights, const void* weights_halo, const float* bias,
                       int N, int H, int W, int KH, int KW, int Cout, int CoutPad, int Kpad, int epilogue,
                       void* out, int out_is_f32, int out_stride,
                       const float* gterm, const void* aux0, int aux0_stride, const void* aux1, int aux1_stride,
                       float* red, dh_stream_t stream);

/* Stride-2 "same" convolution of one dense NHWC fp16 input (k = 1, 3 or 7, pad = k / 2; Hin, Win even): the down-sampling layers of
 * the feature / context encoders (reference droid_slam/modules/extractor.py:140 conv1 7x7 / 2, :24 the residual blocks' first 3x3 / 2,
 * :151 their 1x1 / 2 shortcut).  out [N, Hin/2, Win/2, out_stride] f16 = the even positions of the stride-1 result, at a quarter of
 * its work; weights / bias / CoutPad / Kpad as dh_conv2d_nhwc_f16; epilogue 0 (linear) or 1 (relu). */
int dh_conv2d_s2_nhwc_f16(const void* input, int C, int in_stride, const void* weights, const float* bias,
                          int N, int Hin, int Win, int KH, int KW, int Cout, int CoutPad, int Kpad, int epilogue,
                          void* out, int out_stride, dh_stream_t stream);
/* dw [N,H,W,4] f32 = (delta_x, delta_y, sigmoid w_x, sigmoid w_y) from the partial sums of epilogue 8: bias4 + over the cout
 * tiles the pixel's own four-row tile's row + the row the tile above / below contributes (same image)
 * (= the heads' second 3x3 convolution, reference droid_slam/droid_net.py:95-106).  W == 64, H % 4 == 0, else DH_ERR_UNSUPPORTED. */
int dh_heads_gather(const float* partials, const float* bias4, float* dw, int N, int H, int W, int n_cout_tiles, dh_stream_t stream);
/* Round 6, mode 1: the same gather for a ONE-output second layer with the softplus of GraphAgg's eta head (reference
 * droid_slam/droid_net.py:57-60 `eta`: Conv2d(128, 1, 3) + GradientClip + Softplus, scaled by 0.01 at :71): partials as above (outputs 1..3
 * of every tap unused), dw [N*H*W] f32 = 0.01 * softplus(fp16(bias4[0] + sum)).  The first layer is then GraphAgg's conv2 launched with
 * epilogue 8 AND an `out` pointer: its relu'd activations are stored as well (the upmask head reads them).  mode 0 = dh_heads_gather. */
int dh_heads_gather_ex(const float* partials, const float* bias4, float* dw, int N, int H, int W, int n_cout_tiles, int mode, dh_stream_t stream);
/* The same convolution with accumulator start values:  acc(image n, pixel r, cout) starts from
 *   cinit[(cinit_idx[n] * H*W + r) * cinit_stride + cinit_off + cout]   (fp32; cinit_idx [N] i64)
 * instead of zero, and out_is_f32 == 2 stores unrounded fp32.  This is how the per-source-frame context features of the
 * ConvGRU (reference `inp`, droid_slam/factor_graph.py:135: video.inps[ii], identical for all edges of a source frame)
 * leave the per-edge convolutions: their contribution to the z, r, q gates is one convolution per FRAME
 * (out_is_f32 = 2), and the per-edge gate convolutions run over 320 instead of 448 input channels starting from it --
 * the same sum, associated differently.  Supported where the gate convolutions take their production kernel (3x3, W == 64,
 * H % 4 == 0, every in_channels[i] % 32 == 0, epilogue 3 or 4); DH_ERR_UNSUPPORTED otherwise (the caller then gathers the
 * context features per edge and uses dh_conv2d_nhwc_f16 on all 448 channels).
 * Round 5, the ACCUMULATOR-TILE layout of that per-frame tensor: out_is_f32 == 3 (writer; epilogue 0, Cout % 128 == 0) stores the
 * 64 x 64 wave tiles of the kernel as its registers hold them -- [pixel tile of 256][cout tile of 128][wave 8][a*2+b][q>>2]
 * [lane 64][q&3] fp32, the same N*H*W*Cout floats -- and cinit_stride = -(Cout of that tensor) (reader; cinit_off % 128 == 0)
 * restores them with sixteen 16-byte loads per lane (1 KB contiguous per wave-load) instead of 64 dword loads: the same values,
 * i.e. bit-identical results; both only in the production 3x3 kernel, DH_ERR_UNSUPPORTED elsewhere. */
int dh_conv2d_nhwc_f16_ex(const void* const* inputs, const int* in_channels, const int* in_strides, int n_inputs,
                          const void* weights, const void* weights_halo, const float* bias,
                          int N, int H, int W, int KH, int KW, int Cout, int CoutPad, int Kpad, int epilogue,
                          void* out, int out_is_f32, int out_stride,
                          const float* gterm, const void* aux0, int aux0_stride, const void* aux1, int aux1_stride,
                          float* red, const float* cinit, const int64_t* cinit_idx, int cinit_stride, int cinit_off,
                          dh_stream_t stream);
/* the same with an explicit layout of `weights_halo`: DH_CONV_LAYOUT_AUTO = what dh_conv2d_nhwc_f16_ex does (layout follows
 * the options "conv_dma" / "conv_halo2"); DH_CONV_LAYOUT_WINO = PROTOTYPE: 3x3 as Winograd F(2,3) along x, weights_halo =
 * [CoutPad/128][Ctot/32][3 dy][4 positions][128][4 slots (XOR-swizzled like the halo2 layout)][8] f16 with the transformed
 * taps (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2); W == 64, H % 4 == 0, staged epilogues only, else DH_ERR_UNSUPPORTED. */
#define DH_CONV_LAYOUT_AUTO 0
#define DH_CONV_LAYOUT_WINO 4
int dh_conv2d_nhwc_f16_ex2(const void* const* inputs, const int* in_channels, const int* in_strides, int n_inputs,
                          const void* weights, const void* weights_halo, int weights_layout, const float* bias,
                          int N, int H, int W, int KH, int KW, int Cout, int CoutPad, int Kpad, int epilogue,
                          void* out, int out_is_f32, int out_stride,
                          const float* gterm, const void* aux0, int aux0_stride, const void* aux1, int aux1_stride,
                          float* red, const float* cinit, const int64_t* cinit_idx, int cinit_stride, int cinit_off,
                          dh_stream_t stream);
/* Round 6: the same, with the NEXT iteration's global-context reduction of the ConvGRU fused behind the q gate (epilogue 4 only).
 * The reference's ConvGRU.forward begins with glo = mean_px(sigmoid(w(net)) * net) (droid_slam/modules/gru.py:23-24) on the hidden state
 * the previous iteration's last statement wrote (gru.py:31: net = (1-z) * net + z * q).  With glo_red != NULL the q-gate launch, whose
 * workgroups hold the new state of their 256 pixels x 128 channels in LDS, also adds
 *     glo_red[image][c] += sum over the tile's pixels of fp16(fp16(sigmoid(glo_weights[c] . net' + glo_bias[c])) * net'[c])
 * (glo_weights [128][128] f16, k contiguous = the packed 1x1 layer `gru.w`; glo_bias [128] f32; glo_red [N][128] f32, zeroed by the caller)
 * -- exactly what epilogue 5 (EPI_GLO) computes from HBM on the same values, so that the next iteration's standalone reduction (3.2 GB of
 * reads at 4096 edges) is not launched.  Cout == CoutPad == 128; production 3x3 kernel only (W == 64, H % 4 == 0, 32-channel segments,
 * staged epilogue), DH_ERR_UNSUPPORTED elsewhere; glo_red == NULL: dh_conv2d_nhwc_f16_ex2. */
int dh_conv2d_nhwc_f16_ex3(const void* const* inputs, const int* in_channels, const int* in_strides, int n_inputs,
                          const void* weights, const void* weights_halo, int weights_layout, const float* bias,
                          int N, int H, int W, int KH, int KW, int Cout, int CoutPad, int Kpad, int epilogue,
                          void* out, int out_is_f32, int out_stride,
                          const float* gterm, const void* aux0, int aux0_stride, const void* aux1, int aux1_stride,
                          float* red, const float* cinit, const int64_t* cinit_idx, int cinit_stride, int cinit_off,
                          const void* glo_weights, const float* glo_bias, float* glo_red,
                          dh_stream_t stream);

/* Measurement hook (-DDH_ABLATION builds; DH_ERR_UNSUPPORTED in the shipped library): per-workgroup phase timestamps of the 3x3
 * convolution kernels -- 8 x uint64 per workgroup: 100 MHz wall clock at kernel entry / first fetches issued / first barrier passed /
 * main loop left / epilogue done, then HW_ID and XCC_ID (which CU).  scripts/conv_timeline.py turns them into per-CU timelines. */
int dh_conv_set_timestamps(void* buf, long capacity_workgroups);

/* corr_encoder.0 (reference droid_net.py:83-86: 1x1 convolution 196 -> 128 + relu) on the REFERENCE-layout correlation
 * features: x [E,196,HW] f16 (what dh_corr_pyramid_lookup / corr_index_forward produce, channel = level*49 + xoff*7 + yoff),
 * wp [128,208] f16 = weight[cout][cin] zero-padded to 208 input channels, bias [128] f32 -> out [E,HW,128] f16 channel-last.
 * HW % 128 == 0 (DH_ERR_ARG otherwise: use dh_conv2d_nhwc_f16 on the channel-last lookup output). */
int dh_corr0_nchw_f16(const void* x, const void* wp, const float* bias, void* out, int E, int HW, dh_stream_t stream);

/* Global-context terms of the ConvGRU gates (reference modules/gru.py:21-27: three 1x1 convolutions on the pixel mean of
 * sigmoid(w(net)) * net) as one GEMV: out [E,N] f32 = fp16(bias + fp16(red * scale) wt), red [E,128] f32 (pixel SUMS, scale =
 * 1 / pixels), wt [128,N] f32 k-major, N <= 384 (z | r | q).  Values are rounded to fp16 where autocast rounds them. */
int dh_glo_gemv(const float* red, const float* wt, const float* bias, float* out, int E, int N, float scale,
                dh_stream_t stream);

/* GraphAgg's scatter_mean (reference droid_net.py:67, torch_scatter): out[k,:] = mean of the rows x[order[i],:],
 * i in [seg_off[k], seg_off[k+1]); x [E,row_elems] f16, out [K,row_elems] f16, row_elems % 8 == 0. */
int dh_segment_mean_f16(const void* x, const int64_t* order, const int64_t* seg_off, void* out,
                        int K, long row_elems, dh_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Geometry kernels of the droid_backends API (reference src/droid.cpp:125-171,228-242).
 *   frame_distance: dist [M] f32   (src/droid_kernels.cu:527-666, 1447-1469)
 *   projmap: coords [M,ht,wd,3] f32 (channel 2 = 0), valid [M,ht,wd,1] f32  (:436-525, 1472-1497)
 *   iproj: points [N,ht,wd,3] f32  (:788-859, 1527-1550)
 *   depth_filter: counter [M,ht,wd] f32 (:670-784, 1500-1524)
 */
int dh_frame_distance(const float* poses, const float* disps, const float* intrinsics,
                      const int64_t* ii, const int64_t* jj, float* dist,
                      int M, int ht, int wd, float beta, dh_stream_t stream);
int dh_projmap(const float* poses, const float* disps, const float* intrinsics,
               const int64_t* ii, const int64_t* jj, float* coords, float* valid,
               int M, int ht, int wd, dh_stream_t stream);
int dh_iproj(const float* poses, const float* disps, const float* intrinsics, float* points,
             int N, int ht, int wd, dh_stream_t stream);
int dh_depth_filter(const float* poses, const float* disps, const float* intrinsics,
                    const int64_t* ix, const float* thresh, float* counter,
                    int M, int num_frames, int ht, int wd, dh_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * SE(3) group ops and the fused reprojection: the subset of the un-vendored `lietorch` that sits on
 * the path (reference droid_slam/geom/projective_ops.py:165-198, depth_video.py:171-179).
 * All arrays f32; poses/elements are 7-vectors, n = number of group elements.
 *   dh_se3_inv: out[i] = a[i]^-1          dh_se3_mul: out[i] = a[i]*b[i]
 *   dh_se3_exp: out[i] = exp(xi[i])       dh_se3_retr: out[i] = exp(xi[i])*a[i]
 *   dh_se3_act4: Y[i,p,:] = a[i] * X[i,p,:]  (homogeneous 4-vectors, npts per element)
 *   dh_se3_adjT: Y[i,p,:] = Adj(a[i])^T X[i,p,:]  (6-vectors)
 *   dh_reproject: coords [E,ht,wd,2], valid [E,ht,wd,1] with the Python thresholds
 *                 (MIN_DEPTH 0.2, Z<0.1 -> 1, stereo override for ii==jj; projective_ops.py:6,52,176-185)
 */
int dh_se3_inv(const float* a, float* out, int n, dh_stream_t stream);
int dh_se3_mul(const float* a, const float* b, float* out, int n, dh_stream_t stream);
int dh_se3_exp(const float* xi, float* out, int n, dh_stream_t stream);
int dh_se3_retr(const float* xi, const float* a, float* out, int n, dh_stream_t stream);
int dh_se3_act4(const float* a, const float* X, float* Y, int n, int npts, dh_stream_t stream);
/* out [n,6] = log(a[i]) = (tau, phi): the inverse of dh_se3_exp (lietorch SE3.log; used by the frontend's motion model,
 * reference droid_slam/droid_frontend.py:59-63, and the trajectory filler, trajectory_filler.py:55-65) */
int dh_se3_log(const float* a, float* out, int n, dh_stream_t stream);
int dh_se3_adjT(const float* a, const float* X, float* Y, int n, int npts, dh_stream_t stream);
int dh_reproject(const float* poses, const float* disps, const float* intrinsics,
                 const int64_t* ii, const int64_t* jj, float* coords, float* valid,
                 int E, int ht, int wd, dh_stream_t stream);
/* per_frame_intrinsics != 0: intrinsics is [num_frames,4]; pixels of frame ii[e] are back-projected with intrinsics[ii[e]]
 * and projected with intrinsics[jj[e]], as projective_transform does (projective_ops.py:180,183) */
int dh_reproject_ex(const float* poses, const float* disps, const float* intrinsics, int per_frame_intrinsics,
                    const int64_t* ii, const int64_t* jj, float* coords, float* valid,
                    int E, int ht, int wd, dh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DROID_HIP_H */
