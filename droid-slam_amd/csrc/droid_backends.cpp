// Python module `droid_backends` on top of the C ABI of libdroid_hip.so.
//
// Same nine functions, positional signatures, in-place semantics and error convention as the
// reference's pybind module (reference src/droid.cpp:93-259), so droid_slam/depth_video.py:196-222 and
// droid_slam/modules/corr.py:12,19,79,86 call it unchanged.  This file only does what a torch binding
// has to do: contiguity checks (TORCH_CHECK -> RuntimeError, as droid.cpp:89-90), output allocation on
// the inputs' device, workspace allocation through torch's caching allocator, and handing the current
// HIP stream to the C ABI.  No arithmetic happens here, and there is no CPU fallback: CPU tensors are
// rejected.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <vector>
#include <algorithm>
#include "../../include/droid_hip.h"

namespace {

#define CHECK_CONTIGUOUS(x) TORCH_CHECK(x.is_contiguous(), #x " must be contiguous")
#define CHECK_DEVICE(x) TORCH_CHECK(x.is_cuda(), #x " must be a ROCm device tensor (droid_backends has no CPU path)")
#define CHECK_INPUT(x) \
  CHECK_DEVICE(x);     \
  CHECK_CONTIGUOUS(x)
#define CHECK_F32(x) TORCH_CHECK(x.scalar_type() == torch::kFloat32, #x " must be float32")
#define CHECK_I64(x) TORCH_CHECK(x.scalar_type() == torch::kInt64, #x " must be int64")

dh_stream_t cur_stream() { return (dh_stream_t)c10::hip::getCurrentHIPStream().stream(); }

void check_status(int rc, const char* what) {
  TORCH_CHECK(rc == DH_OK, what, ": ", dh_status_string(rc));
}

int dtype_code(const torch::Tensor& t, const char* name, bool allow_f64 = false) {
  if (t.scalar_type() == torch::kFloat16) return DH_F16;
  if (t.scalar_type() == torch::kFloat32) return DH_F32;
  if (allow_f64 && t.scalar_type() == torch::kFloat64) return DH_F64;
  TORCH_CHECK(false, name, allow_f64 ? " must be float16, float32 or float64" : " must be float16 or float32");
  return -1;
}

// ---- ba (droid.cpp:93-122) --------------------------------------------------------------------
std::vector<torch::Tensor> ba_impl(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics,
                                   torch::Tensor disps_sens, c10::optional<torch::Tensor> alpha, torch::Tensor targets,
                                   torch::Tensor weights, torch::Tensor eta, torch::Tensor ii, torch::Tensor jj, const int t0,
                                   const int t1, const int iterations, const float lm, const float ep, const bool motion_only) {
  CHECK_INPUT(targets); CHECK_INPUT(weights); CHECK_INPUT(poses); CHECK_INPUT(disps);
  CHECK_INPUT(intrinsics); CHECK_INPUT(disps_sens); CHECK_INPUT(ii); CHECK_INPUT(jj);
  CHECK_DEVICE(eta);
  CHECK_F32(poses); CHECK_F32(disps); CHECK_F32(intrinsics); CHECK_F32(disps_sens);
  CHECK_F32(targets); CHECK_F32(weights); CHECK_F32(eta); CHECK_I64(ii); CHECK_I64(jj);
  eta = eta.contiguous();                           // the reference does not check eta (droid.cpp:110-117)
  const int F = (int)disps.size(0), ht = (int)disps.size(1), wd = (int)disps.size(2);
  const int E = (int)ii.size(0);
  TORCH_CHECK(poses.size(0) >= F && disps_sens.size(0) >= F, "poses / disps_sens shorter than disps");
  TORCH_CHECK(jj.size(0) == E && targets.size(0) == E && weights.size(0) == E, "edge count mismatch");
  const int HW = ht * wd;
  const int K = HW > 0 ? (int)(eta.numel() / HW) : 0;
  const int P = t1 - t0;
  auto fopts = poses.options();
  torch::Tensor dx = torch::zeros({P > 0 ? P : 0, 6}, fopts);
  torch::Tensor dz = torch::zeros({K, HW}, fopts);
  const size_t wsb = dh_ba_workspace_bytes(F, E, ht, wd, t0, t1, motion_only ? 1 : 0);
  TORCH_CHECK(wsb > 0, "ba: invalid arguments (t0/t1/buffer sizes)");
  torch::Tensor ws = torch::empty({(int64_t)wsb}, poses.options().dtype(torch::kUInt8));
  const float* al = nullptr;
  if (alpha.has_value()) {
    const torch::Tensor& a = *alpha;
    CHECK_INPUT(a); CHECK_F32(a);
    TORCH_CHECK(a.numel() == disps.numel(), "ba_ex: alpha must have the shape of disps");
    al = a.data_ptr<float>();
  }
  const int rc = dh_ba_ex(poses.data_ptr<float>(), disps.data_ptr<float>(), intrinsics.data_ptr<float>(),
                       disps_sens.data_ptr<float>(), al, targets.data_ptr<float>(), weights.data_ptr<float>(),
                       eta.data_ptr<float>(), ii.data_ptr<int64_t>(), jj.data_ptr<int64_t>(),
                       F, E, K, ht, wd, t0, t1, iterations, lm, ep, motion_only ? 1 : 0,
                       dx.data_ptr<float>(), motion_only ? nullptr : dz.data_ptr<float>(),
                       ws.data_ptr(), wsb, cur_stream());
  check_status(rc, "ba");
  return {dx, dz};
}


std::vector<torch::Tensor> ba(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics,
                              torch::Tensor disps_sens, torch::Tensor targets, torch::Tensor weights,
                              torch::Tensor eta, torch::Tensor ii, torch::Tensor jj, const int t0, const int t1,
                              const int iterations, const float lm, const float ep, const bool motion_only) {
  return ba_impl(poses, disps, intrinsics, disps_sens, c10::nullopt, targets, weights, eta, ii, jj, t0, t1, iterations, lm, ep, motion_only);
}

// ba with a per-pixel depth-prior weight alpha [buf,ht,wd] (MI355X extension, include/droid_hip.h dh_ba_ex)
std::vector<torch::Tensor> ba_ex(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics,
                                 torch::Tensor disps_sens, torch::Tensor alpha, torch::Tensor targets, torch::Tensor weights,
                                 torch::Tensor eta, torch::Tensor ii, torch::Tensor jj, const int t0, const int t1,
                                 const int iterations, const float lm, const float ep, const bool motion_only) {
  return ba_impl(poses, disps, intrinsics, disps_sens, alpha, targets, weights, eta, ii, jj, t0, t1, iterations, lm, ep, motion_only);
}


// ---- split BA for the edge-sharded multi-GPU solver: build -> (all-reduce of `system`) -> finish ------
// returns {workspace (opaque, keeps everything alive), system [(npad+48), npad] f64 view into it}: rows
// 0..npad-1 = reduced camera matrix (6P x 6P in the top-left corner), row npad = right-hand side.
static std::vector<torch::Tensor> ba_build_any(bool shard, torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics,
                                               torch::Tensor disps_sens, c10::optional<torch::Tensor> alpha,
                                               torch::Tensor targets, torch::Tensor weights,
                                               torch::Tensor eta, torch::Tensor ii, torch::Tensor jj, const int t0, const int t1,
                                               const bool motion_only) {
  CHECK_INPUT(targets); CHECK_INPUT(weights); CHECK_INPUT(poses); CHECK_INPUT(disps);
  CHECK_INPUT(intrinsics); CHECK_INPUT(disps_sens); CHECK_INPUT(ii); CHECK_INPUT(jj); CHECK_INPUT(eta);
  CHECK_F32(poses); CHECK_F32(disps); CHECK_F32(intrinsics); CHECK_F32(disps_sens);
  CHECK_F32(targets); CHECK_F32(weights); CHECK_F32(eta); CHECK_I64(ii); CHECK_I64(jj);
  const int F = (int)disps.size(0), ht = (int)disps.size(1), wd = (int)disps.size(2);
  const int E = (int)ii.size(0), HW = ht * wd;
  const int K = HW > 0 ? (int)(eta.numel() / HW) : 0;
  const size_t wsb = dh_ba_workspace_bytes(F, E, ht, wd, t0, t1, motion_only ? 1 : 0);
  TORCH_CHECK(wsb > 0, "ba_build: invalid arguments");
  torch::Tensor ws = torch::empty({(int64_t)wsb}, poses.options().dtype(torch::kUInt8));
  double* H = nullptr; double* b = nullptr;
  const float* al = nullptr;
  if (alpha.has_value()) {
    const torch::Tensor& a = *alpha;
    CHECK_INPUT(a); CHECK_F32(a);
    TORCH_CHECK(shard, "ba_build: the per-pixel prior weight exists on ba_ex and ba_build_shard_ex");
    TORCH_CHECK(a.numel() == disps.numel(), "ba_build_shard_ex: alpha must have the shape of disps");
    al = a.data_ptr<float>();
  }
  if (shard)
    check_status(dh_ba_build_shard_ex(poses.data_ptr<float>(), disps.data_ptr<float>(), intrinsics.data_ptr<float>(),
                                      disps_sens.data_ptr<float>(), al, targets.data_ptr<float>(), weights.data_ptr<float>(),
                                      eta.data_ptr<float>(), ii.data_ptr<int64_t>(), jj.data_ptr<int64_t>(), F, E, K, ht, wd,
                                      t0, t1, motion_only ? 1 : 0, &H, &b, ws.data_ptr(), wsb, cur_stream()),
                 "ba_build_shard");
  else
    check_status(dh_ba_build(poses.data_ptr<float>(), disps.data_ptr<float>(), intrinsics.data_ptr<float>(),
                             disps_sens.data_ptr<float>(), targets.data_ptr<float>(), weights.data_ptr<float>(),
                             eta.data_ptr<float>(), ii.data_ptr<int64_t>(), jj.data_ptr<int64_t>(), F, E, K, ht, wd,
                             t0, t1, motion_only ? 1 : 0, &H, &b, ws.data_ptr(), wsb, cur_stream()),
                 "ba_build");
  int rows = 0, cols = 0;
  check_status(dh_ba_system_shape(t0, t1, &rows, &cols), "ba_system_shape");
  torch::Tensor sys = torch::from_blob(H, {rows, cols}, poses.options().dtype(torch::kFloat64));
  return {ws, sys};
}

std::vector<torch::Tensor> ba_build(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics,
                                    torch::Tensor disps_sens, torch::Tensor targets, torch::Tensor weights,
                                    torch::Tensor eta, torch::Tensor ii, torch::Tensor jj, const int t0, const int t1,
                                    const bool motion_only) {
  return ba_build_any(false, poses, disps, intrinsics, disps_sens, c10::nullopt, targets, weights, eta, ii, jj, t0, t1, motion_only);
}

// the edge-sharded solver's build (dh_ba_build_shard): no host synchronisation, the argument flag travels with the exchange
std::vector<torch::Tensor> ba_build_shard(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics,
                                          torch::Tensor disps_sens, torch::Tensor targets, torch::Tensor weights,
                                          torch::Tensor eta, torch::Tensor ii, torch::Tensor jj, const int t0, const int t1,
                                          const bool motion_only) {
  return ba_build_any(true, poses, disps, intrinsics, disps_sens, c10::nullopt, targets, weights, eta, ii, jj, t0, t1, motion_only);
}

// ba_build_shard with ba_ex's per-pixel depth-prior weight alpha [buf,ht,wd] (dh_ba_build_shard_ex; BASELINE configs[4] on the sharded path)
std::vector<torch::Tensor> ba_build_shard_ex(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics,
                                             torch::Tensor disps_sens, torch::Tensor alpha, torch::Tensor targets, torch::Tensor weights,
                                             torch::Tensor eta, torch::Tensor ii, torch::Tensor jj, const int t0, const int t1,
                                             const bool motion_only) {
  return ba_build_any(true, poses, disps, intrinsics, disps_sens, alpha, targets, weights, eta, ii, jj, t0, t1, motion_only);
}

static std::vector<torch::Tensor> ba_finish_any(torch::Tensor poses, torch::Tensor disps, torch::Tensor jj, torch::Tensor ws,
                                                const int n_eta_rows, const int t0, const int t1, const float lm, const float ep,
                                                const bool motion_only, const int own_lo, const int own_hi) {
  CHECK_INPUT(poses); CHECK_INPUT(disps); CHECK_INPUT(jj); CHECK_INPUT(ws); CHECK_F32(poses); CHECK_F32(disps);
  CHECK_I64(jj);
  const int F = (int)disps.size(0), ht = (int)disps.size(1), wd = (int)disps.size(2), E = (int)jj.size(0);
  const int P = t1 - t0;
  torch::Tensor dx = torch::zeros({P > 0 ? P : 0, 6}, poses.options());
  torch::Tensor dz = torch::zeros({F, ht * wd}, poses.options());
  check_status(dh_ba_finish_owned(poses.data_ptr<float>(), disps.data_ptr<float>(), jj.data_ptr<int64_t>(), F, E, ht, wd,
                                  t0, t1, lm, ep, motion_only ? 1 : 0, own_lo, own_hi, dx.data_ptr<float>(),
                                  motion_only ? nullptr : dz.data_ptr<float>(), ws.data_ptr(), (size_t)ws.numel(),
                                  cur_stream()),
               "ba_finish");
  return {dx, dz.narrow(0, 0, std::min(n_eta_rows, F))};
}

std::vector<torch::Tensor> ba_finish(torch::Tensor poses, torch::Tensor disps, torch::Tensor jj, torch::Tensor ws,
                                     const int n_eta_rows, const int t0, const int t1, const float lm, const float ep,
                                     const bool motion_only) {
  return ba_finish_any(poses, disps, jj, ws, n_eta_rows, t0, t1, lm, ep, motion_only, 0, 1 << 30);
}

std::vector<torch::Tensor> ba_finish_owned(torch::Tensor poses, torch::Tensor disps, torch::Tensor jj, torch::Tensor ws,
                                           const int n_eta_rows, const int t0, const int t1, const float lm, const float ep,
                                           const bool motion_only, const int own_lo, const int own_hi) {
  return ba_finish_any(poses, disps, jj, ws, n_eta_rows, t0, t1, lm, ep, motion_only, own_lo, own_hi);
}

// {Qinv [F,HW], w [F,HW] f32 (rows < K valid, ordered like kx), kx [F] int32, K [1] int32}: views into the workspace of ba_build
std::vector<torch::Tensor> ba_depth_blocks(torch::Tensor ws, torch::Tensor disps, torch::Tensor jj, const int t0, const int t1) {
  CHECK_INPUT(ws); CHECK_INPUT(disps); CHECK_INPUT(jj);
  const int F = (int)disps.size(0), ht = (int)disps.size(1), wd = (int)disps.size(2), E = (int)jj.size(0);
  const float *q = nullptr, *w = nullptr; const int *kx = nullptr, *K = nullptr;
  check_status(dh_ba_depth_blocks(ws.data_ptr(), (size_t)ws.numel(), F, E, ht, wd, t0, t1, &q, &w, &kx, &K), "ba_depth_blocks");
  auto f32 = disps.options().dtype(torch::kFloat32), i32 = disps.options().dtype(torch::kInt32);
  return {torch::from_blob((void*)q, {F, (int64_t)ht * wd}, f32), torch::from_blob((void*)w, {F, (int64_t)ht * wd}, f32),
          torch::from_blob((void*)kx, {F}, i32), torch::from_blob((void*)K, {1}, i32)};
}

// packed exchange of the co-visible 6x6 blocks (dh_ba_pack_blocks / dh_ba_unpack_blocks): `packed` f64
// [36 * n_blocks + 6 (t1 - t0) + 2], bp / bq int32 block rows / columns; disps / jj only give the sizes
void ba_pack_blocks(torch::Tensor ws, torch::Tensor disps, torch::Tensor jj, const int t0, const int t1, const bool motion_only,
                    torch::Tensor bp, torch::Tensor bq, const int host_flags, torch::Tensor packed) {
  CHECK_INPUT(ws); CHECK_INPUT(bp); CHECK_INPUT(bq); CHECK_INPUT(packed);
  TORCH_CHECK(bp.scalar_type() == torch::kInt32 && bq.scalar_type() == torch::kInt32 && bp.numel() == bq.numel(), "ba_pack_blocks: bp / bq int32, same length");
  TORCH_CHECK(packed.scalar_type() == torch::kFloat64 && (size_t)packed.numel() >= dh_ba_packed_len((int)bp.numel(), t0, t1), "ba_pack_blocks: packed buffer");
  check_status(dh_ba_pack_blocks(ws.data_ptr(), (size_t)ws.numel(), (int)disps.size(0), (int)jj.size(0), (int)disps.size(1), (int)disps.size(2),
                                 t0, t1, motion_only ? 1 : 0, bp.data_ptr<int32_t>(), bq.data_ptr<int32_t>(), (int)bp.numel(), host_flags,
                                 packed.data_ptr<double>(), cur_stream()),
               "ba_pack_blocks");
}

void ba_unpack_blocks(torch::Tensor ws, torch::Tensor disps, torch::Tensor jj, const int t0, const int t1, const bool motion_only,
                      torch::Tensor bp, torch::Tensor bq, torch::Tensor packed) {
  CHECK_INPUT(ws); CHECK_INPUT(bp); CHECK_INPUT(bq); CHECK_INPUT(packed);
  TORCH_CHECK(bp.scalar_type() == torch::kInt32 && bq.scalar_type() == torch::kInt32 && bp.numel() == bq.numel(), "ba_unpack_blocks: bp / bq int32, same length");
  TORCH_CHECK(packed.scalar_type() == torch::kFloat64 && (size_t)packed.numel() >= dh_ba_packed_len((int)bp.numel(), t0, t1), "ba_unpack_blocks: packed buffer");
  check_status(dh_ba_unpack_blocks(ws.data_ptr(), (size_t)ws.numel(), (int)disps.size(0), (int)jj.size(0), (int)disps.size(1), (int)disps.size(2),
                                   t0, t1, motion_only ? 1 : 0, bp.data_ptr<int32_t>(), bq.data_ptr<int32_t>(), (int)bp.numel(),
                                   packed.data_ptr<double>(), cur_stream()),
               "ba_unpack_blocks");
}

// dense exchange: status f64 [2]; set = false reads this rank's flags into it, set = true applies the all-reduced ones
void ba_exchange_flags(torch::Tensor ws, torch::Tensor disps, torch::Tensor jj, const int t0, const int t1, const bool motion_only,
                       const int host_flags, torch::Tensor status, const bool set) {
  CHECK_INPUT(ws); CHECK_INPUT(status);
  TORCH_CHECK(status.scalar_type() == torch::kFloat64 && status.numel() >= 2, "ba_exchange_flags: status f64 [2]");
  check_status(dh_ba_exchange_flags(ws.data_ptr(), (size_t)ws.numel(), (int)disps.size(0), (int)jj.size(0), (int)disps.size(1), (int)disps.size(2),
                                    t0, t1, motion_only ? 1 : 0, host_flags, status.data_ptr<double>(), set ? 1 : 0, cur_stream()),
               "ba_exchange_flags");
}

// ---- geometry (droid.cpp:125-171, 228-242) ------------------------------------------------------
torch::Tensor frame_distance(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics,
                             torch::Tensor ii, torch::Tensor jj, const float beta) {
  CHECK_INPUT(poses); CHECK_INPUT(disps); CHECK_INPUT(intrinsics); CHECK_INPUT(ii); CHECK_INPUT(jj);
  CHECK_F32(poses); CHECK_F32(disps); CHECK_F32(intrinsics); CHECK_I64(ii); CHECK_I64(jj);
  const int M = (int)ii.size(0);
  torch::Tensor dist = torch::zeros({M}, poses.options());
  check_status(dh_frame_distance(poses.data_ptr<float>(), disps.data_ptr<float>(), intrinsics.data_ptr<float>(),
                                 ii.data_ptr<int64_t>(), jj.data_ptr<int64_t>(), dist.data_ptr<float>(), M,
                                 (int)disps.size(1), (int)disps.size(2), beta, cur_stream()),
               "frame_distance");
  return dist;
}

std::vector<torch::Tensor> projmap(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics,
                                   torch::Tensor ii, torch::Tensor jj) {
  CHECK_INPUT(poses); CHECK_INPUT(disps); CHECK_INPUT(intrinsics); CHECK_INPUT(ii); CHECK_INPUT(jj);
  CHECK_F32(poses); CHECK_F32(disps); CHECK_F32(intrinsics); CHECK_I64(ii); CHECK_I64(jj);
  const int M = (int)ii.size(0), ht = (int)disps.size(1), wd = (int)disps.size(2);
  torch::Tensor coords = torch::empty({M, ht, wd, 3}, poses.options());
  torch::Tensor valid = torch::empty({M, ht, wd, 1}, poses.options());
  check_status(dh_projmap(poses.data_ptr<float>(), disps.data_ptr<float>(), intrinsics.data_ptr<float>(),
                          ii.data_ptr<int64_t>(), jj.data_ptr<int64_t>(), coords.data_ptr<float>(),
                          valid.data_ptr<float>(), M, ht, wd, cur_stream()),
               "projmap");
  return {coords, valid};
}

torch::Tensor iproj(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics) {
  CHECK_INPUT(poses); CHECK_INPUT(disps); CHECK_INPUT(intrinsics);
  CHECK_F32(poses); CHECK_F32(disps); CHECK_F32(intrinsics);
  const int N = (int)disps.size(0), ht = (int)disps.size(1), wd = (int)disps.size(2);
  torch::Tensor points = torch::empty({N, ht, wd, 3}, disps.options());
  check_status(dh_iproj(poses.data_ptr<float>(), disps.data_ptr<float>(), intrinsics.data_ptr<float>(),
                        points.data_ptr<float>(), N, ht, wd, cur_stream()),
               "iproj");
  return points;
}

torch::Tensor depth_filter(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics,
                           torch::Tensor ix, torch::Tensor thresh) {
  CHECK_INPUT(poses); CHECK_INPUT(disps); CHECK_INPUT(intrinsics); CHECK_INPUT(ix); CHECK_INPUT(thresh);
  CHECK_F32(poses); CHECK_F32(disps); CHECK_F32(intrinsics); CHECK_I64(ix); CHECK_F32(thresh);
  const int M = (int)ix.size(0), N = (int)disps.size(0), ht = (int)disps.size(1), wd = (int)disps.size(2);
  torch::Tensor counter = torch::empty({M, ht, wd}, disps.options());
  check_status(dh_depth_filter(poses.data_ptr<float>(), disps.data_ptr<float>(), intrinsics.data_ptr<float>(),
                               ix.data_ptr<int64_t>(), thresh.data_ptr<float>(), counter.data_ptr<float>(), M, N,
                               ht, wd, cur_stream()),
               "depth_filter");
  return counter;
}

// ---- correlation (droid.cpp:175-226) -------------------------------------------------------------
std::vector<torch::Tensor> corr_index_forward(torch::Tensor volume, torch::Tensor coords, int radius) {
  CHECK_INPUT(volume); CHECK_INPUT(coords); CHECK_F32(coords);
  TORCH_CHECK(volume.dim() == 5 && coords.dim() == 4, "corr_index_forward: volume [N,h1,w1,h2,w2], coords [N,2,h1,w1]");
  const int N = (int)volume.size(0), h1 = (int)volume.size(1), w1 = (int)volume.size(2);
  torch::Tensor corr = torch::empty({N, 2 * radius + 1, 2 * radius + 1, h1, w1}, volume.options());
  check_status(dh_corr_index_fwd(volume.data_ptr(), coords.data_ptr<float>(), corr.data_ptr(),
                                 dtype_code(volume, "volume", true), N, h1, w1, (int)volume.size(3), (int)volume.size(4),
                                 radius, cur_stream()),
               "corr_index_forward");
  return {corr};
}

std::vector<torch::Tensor> corr_index_backward(torch::Tensor volume, torch::Tensor coords,
                                               torch::Tensor corr_grad, int radius) {
  CHECK_INPUT(volume); CHECK_INPUT(coords); CHECK_INPUT(corr_grad); CHECK_F32(coords);
  TORCH_CHECK(corr_grad.scalar_type() == volume.scalar_type(), "corr_grad dtype must match volume");
  torch::Tensor volume_grad = torch::empty_like(volume);
  check_status(dh_corr_index_bwd(coords.data_ptr<float>(), corr_grad.data_ptr(), volume_grad.data_ptr(),
                                 dtype_code(volume, "volume", true), (int)volume.size(0), (int)volume.size(1),
                                 (int)volume.size(2), (int)volume.size(3), (int)volume.size(4), radius, cur_stream()),
               "corr_index_backward");
  return {volume_grad};
}

// ---- reference-layout volume + pooling for any image size (droid_amd.corr.CorrBlockRef) ----------------------
torch::Tensor corr_volume_build(torch::Tensor fmap1, torch::Tensor fmap2) {
  CHECK_INPUT(fmap1); CHECK_INPUT(fmap2);
  TORCH_CHECK(fmap1.dim() == 4 && fmap1.sizes() == fmap2.sizes() && fmap1.scalar_type() == fmap2.scalar_type(),
              "corr_volume_build: fmaps [E,C,h,w] of equal shape and dtype");
  const int E = (int)fmap1.size(0), C = (int)fmap1.size(1), h = (int)fmap1.size(2), w = (int)fmap1.size(3);
  torch::Tensor vol = torch::empty({E, h, w, h, w}, fmap1.options());
  check_status(dh_corr_volume_build(fmap1.data_ptr(), fmap2.data_ptr(), vol.data_ptr(), dtype_code(fmap1, "fmap1"), E, C, h, w, cur_stream()),
               "corr_volume_build");
  return vol;
}

torch::Tensor corr_volume_pool(torch::Tensor vol) {
  CHECK_INPUT(vol);
  TORCH_CHECK(vol.dim() >= 3, "corr_volume_pool: volume [..., h2, w2]");
  const int h2 = (int)vol.size(-2), w2 = (int)vol.size(-1);
  std::vector<int64_t> shape(vol.sizes().begin(), vol.sizes().end());
  shape[shape.size() - 2] = h2 / 2; shape[shape.size() - 1] = w2 / 2;
  torch::Tensor out = torch::empty(shape, vol.options());
  const long n = h2 * w2 > 0 ? vol.numel() / ((long)h2 * w2) : 0;
  check_status(dh_corr_volume_pool(vol.data_ptr(), out.data_ptr(), dtype_code(vol, "volume"), n, h2, w2, cur_stream()), "corr_volume_pool");
  return out;
}

std::vector<torch::Tensor> altcorr_forward(torch::Tensor fmap1, torch::Tensor fmap2, torch::Tensor coords,
                                           torch::Tensor ii, torch::Tensor jj, int radius) {
  CHECK_INPUT(fmap1); CHECK_INPUT(fmap2); CHECK_INPUT(coords); CHECK_DEVICE(ii); CHECK_DEVICE(jj);
  CHECK_F32(coords); CHECK_I64(ii); CHECK_I64(jj);
  TORCH_CHECK(fmap1.scalar_type() == fmap2.scalar_type(), "fmap dtypes differ");
  ii = ii.contiguous(); jj = jj.contiguous();
  const int B = (int)coords.size(0), M = (int)coords.size(1), H = (int)coords.size(3), W = (int)coords.size(4);
  const int rd = 2 * radius + 1;
  torch::Tensor corr = torch::empty({B, M, rd, rd, H, W}, fmap1.options());
  check_status(dh_altcorr_fwd(fmap1.data_ptr(), fmap2.data_ptr(), coords.data_ptr<float>(), ii.data_ptr<int64_t>(),
                              jj.data_ptr<int64_t>(), corr.data_ptr(), dtype_code(fmap1, "fmap1"), B,
                              (int)fmap1.size(1), (int)fmap2.size(1), (int)fmap1.size(2), H, W, (int)fmap2.size(3),
                              (int)fmap2.size(4), M, radius, cur_stream()),
               "altcorr_forward");
  return {corr};
}

// argument order follows the definition the reference actually links (droid.cpp:212-226)
std::vector<torch::Tensor> altcorr_backward(torch::Tensor fmap1, torch::Tensor fmap2, torch::Tensor coords,
                                            torch::Tensor corr_grad, torch::Tensor ii, torch::Tensor jj, int radius) {
  CHECK_INPUT(fmap1); CHECK_INPUT(fmap2); CHECK_INPUT(coords); CHECK_INPUT(corr_grad);
  CHECK_DEVICE(ii); CHECK_DEVICE(jj); CHECK_F32(coords); CHECK_I64(ii); CHECK_I64(jj);
  ii = ii.contiguous(); jj = jj.contiguous();
  torch::Tensor g = corr_grad.to(torch::kFloat32).contiguous();
  torch::Tensor g1 = torch::zeros(fmap1.sizes(), fmap1.options().dtype(torch::kFloat32));
  torch::Tensor g2 = torch::zeros(fmap2.sizes(), fmap2.options().dtype(torch::kFloat32));
  const int B = (int)coords.size(0), M = (int)coords.size(1), H = (int)coords.size(3), W = (int)coords.size(4);
  check_status(dh_altcorr_bwd(fmap1.data_ptr(), fmap2.data_ptr(), coords.data_ptr<float>(), ii.data_ptr<int64_t>(),
                              jj.data_ptr<int64_t>(), g.data_ptr<float>(), g1.data_ptr<float>(),
                              g2.data_ptr<float>(), dtype_code(fmap1, "fmap1"), B, (int)fmap1.size(1),
                              (int)fmap2.size(1), (int)fmap1.size(2), H, W, (int)fmap2.size(3), (int)fmap2.size(4),
                              M, radius, cur_stream()),
               "altcorr_backward");
  return {g1.to(fmap1.scalar_type()), g2.to(fmap2.scalar_type())};
}


// channel-last MFMA variant (droid_amd.corr.AltCorrBlock): fmap1 [N1,H,W,128], fmap2 [N2,H2,W2,128] f16, coords [M,2,H,W]
torch::Tensor altcorr_forward_nhwc(torch::Tensor fmap1, torch::Tensor fmap2, torch::Tensor coords, torch::Tensor ii,
                                   torch::Tensor jj) {
  CHECK_INPUT(fmap1); CHECK_INPUT(fmap2); CHECK_INPUT(coords); CHECK_INPUT(ii); CHECK_INPUT(jj);
  CHECK_F32(coords); CHECK_I64(ii); CHECK_I64(jj);
  TORCH_CHECK(fmap1.scalar_type() == torch::kFloat16 && fmap2.scalar_type() == torch::kFloat16, "fmaps must be float16");
  TORCH_CHECK(fmap1.dim() == 4 && fmap2.dim() == 4 && coords.dim() == 4 && coords.size(1) == 2, "altcorr_forward_nhwc: shapes");
  const int N1 = (int)fmap1.size(0), H = (int)fmap1.size(1), W = (int)fmap1.size(2), C = (int)fmap1.size(3);
  const int N2 = (int)fmap2.size(0), H2 = (int)fmap2.size(1), W2 = (int)fmap2.size(2), M = (int)ii.size(0);
  TORCH_CHECK(coords.size(0) == M && coords.size(2) == H && coords.size(3) == W && fmap2.size(3) == C, "altcorr_forward_nhwc: shapes");
  torch::Tensor out = torch::empty({M, 7, 7, H, W}, fmap1.options());
  check_status(dh_altcorr_fwd_nhwc(fmap1.data_ptr(), fmap2.data_ptr(), coords.data_ptr<float>(), ii.data_ptr<int64_t>(),
                                   jj.data_ptr<int64_t>(), out.data_ptr(), N1, N2, C, H, W, H2, W2, M, cur_stream()),
               "altcorr_forward_nhwc");
  return out;
}

// all levels of AltCorrBlock.__call__ into one [M, 49 * levels, H, W] tensor (no per-level stack copies)
torch::Tensor altcorr_forward_nhwc_levels(torch::Tensor fmap1, std::vector<torch::Tensor> fmap2_levels, torch::Tensor coords,
                                          torch::Tensor ii, torch::Tensor jj) {
  CHECK_INPUT(fmap1); CHECK_INPUT(coords); CHECK_INPUT(ii); CHECK_INPUT(jj);
  CHECK_F32(coords); CHECK_I64(ii); CHECK_I64(jj);
  TORCH_CHECK(fmap1.scalar_type() == torch::kFloat16 && fmap1.dim() == 4 && coords.dim() == 4 && coords.size(1) == 2, "altcorr_forward_nhwc_levels: shapes");
  const int N1 = (int)fmap1.size(0), H = (int)fmap1.size(1), W = (int)fmap1.size(2), C = (int)fmap1.size(3), M = (int)ii.size(0);
  const int L = (int)fmap2_levels.size();
  TORCH_CHECK(coords.size(0) == M && coords.size(2) == H && coords.size(3) == W && L >= 1, "altcorr_forward_nhwc_levels: shapes");
  torch::Tensor out = torch::empty({M, 49 * L, H, W}, fmap1.options());
  for (int l = 0; l < L; ++l) {
    const torch::Tensor& f2 = fmap2_levels[l];
    CHECK_INPUT(f2);
    TORCH_CHECK(f2.scalar_type() == torch::kFloat16 && f2.dim() == 4 && f2.size(3) == C, "altcorr_forward_nhwc_levels: level features [N2,h,w,C] float16");
    check_status(dh_altcorr_fwd_nhwc_level(fmap1.data_ptr(), f2.data_ptr(), coords.data_ptr<float>(), ii.data_ptr<int64_t>(), jj.data_ptr<int64_t>(),
                                           (char*)out.data_ptr() + (size_t)l * 49 * H * W * 2, N1, (int)f2.size(0), C, H, W, (int)f2.size(1), (int)f2.size(2), M,
                                           l, (long)49 * L * H * W, cur_stream()), "altcorr_forward_nhwc_levels");
  }
  return out;
}

// ---- MI355X-native correlation pyramid (own layout; droid_amd.corr.CorrBlock) ----------------------------
// h_real / w_real > 0: the features are an h_real x w_real image, zero-padded to the canvas size of the tensors
// (dh_corr_pyramid_build_canvas); 0 = the tensors ARE the image
// out: optional pre-allocated pyramid storage (float16, contiguous, at least E records) -- the caller's ARENA: a 105 GB hipMalloc
// costs seconds on a cold device, so callers that rebuild pyramids of the same graph (FactorGraph.update_lowmem) keep the storage
torch::Tensor corr_pyramid_build(torch::Tensor fmap1, torch::Tensor fmap2, const int h_real, const int w_real,
                                 const c10::optional<torch::Tensor>& out) {
  CHECK_INPUT(fmap1); CHECK_INPUT(fmap2);
  TORCH_CHECK(fmap1.scalar_type() == torch::kFloat16 && fmap2.scalar_type() == torch::kFloat16, "fmaps must be float16");
  TORCH_CHECK(fmap1.dim() == 4 && fmap1.sizes() == fmap2.sizes(), "corr_pyramid_build: fmaps [E,C,h,w] of equal shape");
  const int E = (int)fmap1.size(0), C = (int)fmap1.size(1), h = (int)fmap1.size(2), w = (int)fmap1.size(3);
  const size_t bytes1 = dh_corr_pyramid_bytes(1, h, w);
  TORCH_CHECK(bytes1 > 0, "corr_pyramid_build: unsupported image size (need h % 8 == 0 and w in {16,32,64})");
  torch::Tensor pyr;
  if (out.has_value() && out->defined()) {
    CHECK_INPUT((*out));
    TORCH_CHECK(out->scalar_type() == torch::kFloat16 && out->dim() == 2 && out->size(0) >= E && (size_t)out->size(1) * 2 == bytes1,
                "corr_pyramid_build: out must be float16 [>= E, record elements] for this image size");
    pyr = out->narrow(0, 0, E);
  } else {
    pyr = torch::empty({E, (int64_t)(bytes1 / 2)}, fmap1.options());
  }
  const int chunk = 256;                                  // bounds the channel-last scratch copies
  const size_t wsb = dh_corr_pyramid_workspace_bytes(std::min(E, chunk), h, w);
  torch::Tensor ws = torch::empty({(int64_t)wsb}, fmap1.options().dtype(torch::kUInt8));
  for (int s = 0; s < E; s += chunk) {
    const int n = std::min(chunk, E - s);
    check_status(dh_corr_pyramid_build_canvas(fmap1[s].data_ptr(), fmap2[s].data_ptr(), pyr[s].data_ptr(), ws.data_ptr(), wsb,
                                              n, C, h, w, h_real > 0 ? h_real : h, w_real > 0 ? w_real : w, cur_stream()),
                 "corr_pyramid_build");
  }
  return pyr;
}

// frame-level build: prepared = corr_pyramid_prepare_frames(fmaps [F,C,h,w] f16 (canvas-sized), h_real, w_real) -> [F, T, C] f16;
// corr_pyramid_build_indexed(prepared, idx1 [E] i64, idx2 [E] i64, h, w, out=None) -> pyramid [E, record elements]
torch::Tensor corr_pyramid_prepare_frames(torch::Tensor fmaps, const int h_real, const int w_real) {
  CHECK_INPUT(fmaps);
  TORCH_CHECK(fmaps.scalar_type() == torch::kFloat16 && fmaps.dim() == 4, "corr_pyramid_prepare_frames: fmaps [F,C,h,w] float16");
  const int F = (int)fmaps.size(0), C = (int)fmaps.size(1), h = (int)fmaps.size(2), w = (int)fmaps.size(3);
  const size_t bytes = dh_corr_pyramid_prepared_bytes(F, h, w);
  TORCH_CHECK(F == 0 || bytes > 0, "corr_pyramid_prepare_frames: unsupported image size (need h % 8 == 0 and w in {16,32,64})");
  torch::Tensor prep = torch::empty({F, F > 0 ? (int64_t)(bytes / 2 / F / C) : 0, C}, fmaps.options());
  check_status(dh_corr_pyramid_prepare_frames(fmaps.data_ptr(), prep.data_ptr(), F, C, h, w, h_real > 0 ? h_real : h, w_real > 0 ? w_real : w,
                                              cur_stream()), "corr_pyramid_prepare_frames");
  return prep;
}

torch::Tensor corr_pyramid_build_indexed(torch::Tensor prepared, torch::Tensor idx1, torch::Tensor idx2, const int h, const int w,
                                         const c10::optional<torch::Tensor>& out) {
  CHECK_INPUT(prepared); CHECK_INPUT(idx1); CHECK_INPUT(idx2); CHECK_I64(idx1); CHECK_I64(idx2);
  TORCH_CHECK(prepared.scalar_type() == torch::kFloat16 && prepared.dim() == 3 && prepared.size(2) == 128, "corr_pyramid_build_indexed: prepared [F,T,128] float16");
  const int F = (int)prepared.size(0), E = (int)idx1.size(0);
  TORCH_CHECK(idx2.size(0) == E, "corr_pyramid_build_indexed: idx1 / idx2 of equal length");
  const size_t bytes1 = dh_corr_pyramid_bytes(1, h, w);
  TORCH_CHECK(bytes1 > 0 && (size_t)prepared.size(1) * 128 * 2 * (size_t)std::max(F, 1) == dh_corr_pyramid_prepared_bytes(std::max(F, 1), h, w),
              "corr_pyramid_build_indexed: prepared does not belong to this image size");
  if (E > 0) {                                            // one read-back for the four bounds (an index outside the tensor would be a wild read)
    torch::Tensor b = torch::stack({idx1.min(), idx1.max(), idx2.min(), idx2.max()}).cpu();
    const int64_t* bp = b.data_ptr<int64_t>();
    TORCH_CHECK(bp[0] >= 0 && bp[1] < F && bp[2] >= 0 && bp[3] < F, "corr_pyramid_build_indexed: frame index out of range");
  }
  torch::Tensor pyr;
  if (out.has_value() && out->defined()) {
    CHECK_INPUT((*out));
    TORCH_CHECK(out->scalar_type() == torch::kFloat16 && out->dim() == 2 && out->size(0) >= E && (size_t)out->size(1) * 2 == bytes1,
                "corr_pyramid_build_indexed: out must be float16 [>= E, record elements] for this image size");
    pyr = out->narrow(0, 0, E);
  } else {
    pyr = torch::empty({E, (int64_t)(bytes1 / 2)}, prepared.options());
  }
  check_status(dh_corr_pyramid_build_indexed(prepared.data_ptr(), idx1.data_ptr<int64_t>(), idx2.data_ptr<int64_t>(), pyr.data_ptr(), F, E, h, w,
                                             cur_stream()), "corr_pyramid_build_indexed");
  return pyr;
}

torch::Tensor corr_pyramid_lookup(torch::Tensor pyramid, torch::Tensor coords) {
  CHECK_INPUT(pyramid); CHECK_INPUT(coords); CHECK_F32(coords);
  TORCH_CHECK(pyramid.scalar_type() == torch::kFloat16, "pyramid must be float16");
  TORCH_CHECK(coords.dim() == 4 && coords.size(3) == 2, "corr_pyramid_lookup: coords [E,h,w,2]");
  const int E = (int)coords.size(0), h = (int)coords.size(1), w = (int)coords.size(2);
  TORCH_CHECK(pyramid.size(0) == E && (size_t)pyramid.size(1) * 2 == dh_corr_pyramid_bytes(1, h, w),
              "corr_pyramid_lookup: pyramid does not match coords");
  torch::Tensor out = torch::empty({E, 4 * 49, h, w}, pyramid.options());
  check_status(dh_corr_pyramid_lookup(pyramid.data_ptr(), coords.data_ptr<float>(), out.data_ptr(), E, h, w, cur_stream()),
               "corr_pyramid_lookup");
  return out;
}

torch::Tensor corr_pyramid_lookup_nhwc(torch::Tensor pyramid, torch::Tensor coords) {
  CHECK_INPUT(pyramid); CHECK_INPUT(coords); CHECK_F32(coords);
  TORCH_CHECK(pyramid.scalar_type() == torch::kFloat16, "pyramid must be float16");
  TORCH_CHECK(coords.dim() == 4 && coords.size(3) == 2, "corr_pyramid_lookup_nhwc: coords [E,h,w,2]");
  const int E = (int)coords.size(0), h = (int)coords.size(1), w = (int)coords.size(2);
  TORCH_CHECK(pyramid.size(0) == E && (size_t)pyramid.size(1) * 2 == dh_corr_pyramid_bytes(1, h, w),
              "corr_pyramid_lookup_nhwc: pyramid does not match coords");
  torch::Tensor out = torch::empty({4, E, h, w, 56}, pyramid.options());
  check_status(dh_corr_pyramid_lookup_nhwc(pyramid.data_ptr(), coords.data_ptr<float>(), out.data_ptr(), E, h, w, cur_stream()),
               "corr_pyramid_lookup_nhwc");
  return out;
}

torch::Tensor corr_pyramid_lookup_corr0(torch::Tensor pyramid, torch::Tensor coords, torch::Tensor wpk, torch::Tensor bias) {
  CHECK_INPUT(pyramid); CHECK_INPUT(coords); CHECK_F32(coords); CHECK_INPUT(wpk); CHECK_INPUT(bias);
  TORCH_CHECK(pyramid.scalar_type() == torch::kFloat16, "pyramid must be float16");
  TORCH_CHECK(coords.dim() == 4 && coords.size(3) == 2, "corr_pyramid_lookup_corr0: coords [E,h,w,2]");
  TORCH_CHECK(wpk.scalar_type() == torch::kFloat16 && wpk.numel() == 13 * 128 * 16, "corr_pyramid_lookup_corr0: wpk [13,128,16] float16");
  TORCH_CHECK(bias.scalar_type() == torch::kFloat32 && bias.numel() == 128, "corr_pyramid_lookup_corr0: bias [128] float32");
  const int E = (int)coords.size(0), h = (int)coords.size(1), w = (int)coords.size(2);
  TORCH_CHECK(pyramid.size(0) == E && (size_t)pyramid.size(1) * 2 == dh_corr_pyramid_bytes(1, h, w),
              "corr_pyramid_lookup_corr0: pyramid does not match coords");
  torch::Tensor out = torch::empty({E, h, w, 128}, pyramid.options());
  check_status(dh_corr_pyramid_lookup_corr0(pyramid.data_ptr(), coords.data_ptr<float>(), wpk.data_ptr(), bias.data_ptr<float>(),
                                            out.data_ptr(), E, h, w, cur_stream()), "corr_pyramid_lookup_corr0");
  return out;
}

torch::Tensor corr0_nchw(torch::Tensor x, torch::Tensor wp, torch::Tensor bias) {
  CHECK_INPUT(x); CHECK_INPUT(wp); CHECK_INPUT(bias);
  TORCH_CHECK(x.scalar_type() == torch::kFloat16 && x.dim() == 4 && x.size(1) == 196, "corr0_nchw: x [E,196,h,w] float16");
  TORCH_CHECK(wp.scalar_type() == torch::kFloat16 && wp.dim() == 2 && wp.size(0) == 128 && wp.size(1) == 208, "corr0_nchw: wp [128,208] float16");
  TORCH_CHECK(bias.scalar_type() == torch::kFloat32 && bias.numel() == 128, "corr0_nchw: bias [128] float32");
  const int64_t E = x.size(0), h = x.size(2), w = x.size(3);
  torch::Tensor out = torch::empty({E, h, w, 128}, x.options());
  check_status(dh_corr0_nchw_f16(x.data_ptr(), wp.data_ptr(), bias.data_ptr<float>(), out.data_ptr(), (int)E, (int)(h * w), cur_stream()),
               "corr0_nchw");
  return out;
}

torch::Tensor glo_gemv(torch::Tensor red, torch::Tensor wt, torch::Tensor bias, double scale) {
  CHECK_INPUT(red); CHECK_INPUT(wt); CHECK_INPUT(bias);
  TORCH_CHECK(red.scalar_type() == torch::kFloat32 && red.dim() == 2 && red.size(1) == 128, "glo_gemv: red [E,128] float32");
  TORCH_CHECK(wt.scalar_type() == torch::kFloat32 && wt.dim() == 2 && wt.size(0) == 128, "glo_gemv: wt [128,N] float32");
  const int64_t E = red.size(0), N = wt.size(1);
  TORCH_CHECK(bias.scalar_type() == torch::kFloat32 && bias.numel() == N && N <= 384, "glo_gemv: bias [N], N <= 384");
  torch::Tensor out = torch::empty({E, N}, red.options());
  check_status(dh_glo_gemv(red.data_ptr<float>(), wt.data_ptr<float>(), bias.data_ptr<float>(), out.data_ptr<float>(), (int)E, (int)N,
                           (float)scale, cur_stream()), "glo_gemv");
  return out;
}

torch::Tensor segment_mean(torch::Tensor x, torch::Tensor order, torch::Tensor seg_off) {
  CHECK_INPUT(x); CHECK_INPUT(order); CHECK_INPUT(seg_off); CHECK_I64(order); CHECK_I64(seg_off);
  TORCH_CHECK(x.scalar_type() == torch::kFloat16 && x.dim() >= 2, "segment_mean: x [E,...] float16");
  const int64_t E = x.size(0), K = seg_off.numel() - 1;
  TORCH_CHECK(order.numel() == E && K >= 0, "segment_mean: order [E], seg_off [K+1]");
  const int64_t row = E > 0 ? x.numel() / E : 0;
  auto sizes = x.sizes().vec(); sizes[0] = K;
  torch::Tensor out = torch::empty(sizes, x.options());
  if (E > 0) check_status(dh_segment_mean_f16(x.data_ptr(), order.data_ptr<int64_t>(), seg_off.data_ptr<int64_t>(), out.data_ptr(),
                                              (int)K, (long)row, cur_stream()), "segment_mean");
  return out;
}

// ---- implicit-GEMM convolution with fused epilogues (droid_amd.update.UpdateModule) ----------------------
// stride-2 "same" convolution of one dense NHWC input (dh_conv2d_s2_nhwc_f16: the encoders' down-sampling layers) -> [N,H/2,W/2,Cout] f16
torch::Tensor conv2d_s2_nhwc(torch::Tensor input, torch::Tensor weight, torch::Tensor bias, int64_t KH, int64_t KW, int64_t Cout, int64_t epilogue) {
  CHECK_INPUT(input); CHECK_INPUT(weight); CHECK_INPUT(bias); CHECK_F32(bias);
  TORCH_CHECK(input.scalar_type() == torch::kFloat16 && input.dim() == 4, "conv2d_s2_nhwc: input [N,H,W,C] float16, dense");
  TORCH_CHECK(weight.scalar_type() == torch::kFloat16 && weight.dim() == 2, "conv2d_s2_nhwc: packed weight [CoutPad,Kpad] float16");
  const int N = (int)input.size(0), H = (int)input.size(1), W = (int)input.size(2), C = (int)input.size(3);
  TORCH_CHECK(H % 2 == 0 && W % 2 == 0, "conv2d_s2_nhwc: even image size");
  torch::Tensor out = torch::empty({N, H / 2, W / 2, Cout}, input.options());
  check_status(dh_conv2d_s2_nhwc_f16(input.data_ptr(), C, C, weight.data_ptr(), bias.data_ptr<float>(), N, H, W, (int)KH, (int)KW, (int)Cout,
                                     (int)weight.size(0), (int)weight.size(1), (int)epilogue, out.data_ptr(), (int)Cout, cur_stream()),
               "conv2d_s2_nhwc");
  return out;
}

void conv2d_nhwc(std::vector<torch::Tensor> inputs, torch::Tensor weight, c10::optional<torch::Tensor> weight_halo,
                 torch::Tensor bias, int64_t KH, int64_t KW,
                 int64_t Cout, int64_t epilogue, c10::optional<torch::Tensor> out, int64_t out_stride,
                 c10::optional<torch::Tensor> gterm, c10::optional<torch::Tensor> aux0, c10::optional<torch::Tensor> aux1,
                 c10::optional<torch::Tensor> red, c10::optional<torch::Tensor> cinit, c10::optional<torch::Tensor> cinit_idx,
                 int64_t cinit_off, bool out_raw_f32, int64_t weights_layout, bool out_tiled, bool cinit_tiled,
                 c10::optional<torch::Tensor> glo_weight, c10::optional<torch::Tensor> glo_bias, c10::optional<torch::Tensor> glo_red) {
  TORCH_CHECK(!inputs.empty() && inputs.size() <= 4, "conv2d_nhwc: 1..4 inputs");
  const void* ptrs[4]; int chans[4], strides[4];
  const int64_t N = inputs[0].size(0), H = inputs[0].size(1), W = inputs[0].size(2);
  for (size_t i = 0; i < inputs.size(); ++i) {
    const torch::Tensor& t = inputs[i];
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == torch::kFloat16 && t.dim() == 4, "conv2d_nhwc: inputs [N,H,W,C] float16 device tensors");
    TORCH_CHECK(t.size(0) == N && t.size(1) == H && t.size(2) == W, "conv2d_nhwc: input shapes differ");
    // dense NHWC or a channel slice of one: stride(3) == 1, pixels equally spaced
    // (the stride of a size-1 dimension is arbitrary in torch: only dimensions that are actually stepped through count)
    TORCH_CHECK(t.stride(3) == 1 && (H == 1 || t.stride(1) == t.stride(2) * W) && (N == 1 || t.stride(0) == t.stride(2) * W * H),
                "conv2d_nhwc: input must be NHWC (channel slices allowed)");
    ptrs[i] = t.data_ptr(); chans[i] = (int)t.size(3); strides[i] = (int)t.stride(2);
  }
  CHECK_INPUT(weight); CHECK_INPUT(bias); CHECK_F32(bias);
  TORCH_CHECK(weight.scalar_type() == torch::kFloat16 && weight.dim() == 2, "conv2d_nhwc: packed weight [CoutPad,Kpad] float16");
  auto opt_ptr = [](const c10::optional<torch::Tensor>& t) -> void* { return t.has_value() ? t->data_ptr() : nullptr; };
  auto last = [](const c10::optional<torch::Tensor>& t) -> int { return t.has_value() ? (int)t->stride(2) : 0; };   // pixel stride
  const bool out_f32 = out.has_value() && out->scalar_type() == torch::kFloat32;
  if (gterm.has_value()) { const torch::Tensor& gt = *gterm; CHECK_INPUT(gt); CHECK_F32(gt); }
  if (red.has_value()) { const torch::Tensor& rt = *red; CHECK_INPUT(rt); CHECK_F32(rt); }
  const float* ci = nullptr; const int64_t* cidx = nullptr; int cstride = 0;
  if (cinit.has_value()) {
    const torch::Tensor& ct = *cinit;
    TORCH_CHECK(cinit_idx.has_value(), "conv2d_nhwc: cinit needs cinit_idx");
    const torch::Tensor& it = *cinit_idx;
    CHECK_INPUT(ct); CHECK_F32(ct); CHECK_INPUT(it); CHECK_I64(it);
    if (cinit_tiled) {      // accumulator-tile layout (include/droid_hip.h): [K, H*W/256 pixel tiles, C/128 cout tiles, 32768 floats per tile]
      TORCH_CHECK(ct.dim() == 4 && ct.size(1) * 256 == H * W && ct.size(3) == 32768 && it.numel() == N,
                  "conv2d_nhwc: tiled cinit [K, H*W/256, C/128, 32768] f32, cinit_idx [N] i64");
      cstride = -(int)ct.size(2) * 128;
    } else {
      TORCH_CHECK(ct.dim() == 4 && ct.size(1) == H && ct.size(2) == W && it.numel() == N, "conv2d_nhwc: cinit [K,H,W,C] f32, cinit_idx [N] i64");
      cstride = (int)ct.size(3);
    }
    ci = ct.data_ptr<float>(); cidx = it.data_ptr<int64_t>();
  }
  TORCH_CHECK(!out_raw_f32 || out_f32, "conv2d_nhwc: out_raw_f32 needs a float32 output");
  TORCH_CHECK(!out_tiled || (out_f32 && out_raw_f32), "conv2d_nhwc: out_tiled needs a raw float32 output");
  const void* gw = nullptr; const float* gb = nullptr; float* gr = nullptr;
  if (glo_red.has_value()) {          // the next iteration's global-context reduction behind the q gate (dh_conv2d_nhwc_f16_ex3)
    TORCH_CHECK(glo_weight.has_value() && glo_bias.has_value(), "conv2d_nhwc: glo_red needs glo_weight and glo_bias");
    const torch::Tensor& w = *glo_weight; const torch::Tensor& b = *glo_bias; const torch::Tensor& r = *glo_red;
    CHECK_INPUT(w); CHECK_INPUT(b); CHECK_F32(b); CHECK_INPUT(r); CHECK_F32(r);
    TORCH_CHECK(w.scalar_type() == torch::kFloat16 && w.dim() == 2 && w.size(0) == 128 && w.size(1) == 128 && b.numel() == 128 &&
                r.dim() == 2 && r.size(0) == N && r.size(1) == 128, "conv2d_nhwc: glo_weight [128,128] f16, glo_bias [128] f32, glo_red [N,128] f32");
    gw = w.data_ptr(); gb = b.data_ptr<float>(); gr = r.data_ptr<float>();
  }
  check_status(dh_conv2d_nhwc_f16_ex3(ptrs, chans, strides, (int)inputs.size(), weight.data_ptr(), opt_ptr(weight_halo), (int)weights_layout, bias.data_ptr<float>(),
                                     (int)N, (int)H, (int)W, (int)KH, (int)KW, (int)Cout, (int)weight.size(0), (int)weight.size(1),
                                     (int)epilogue, opt_ptr(out), out_f32 ? (out_tiled ? 3 : out_raw_f32 ? 2 : 1) : 0, (int)out_stride,
                                     gterm.has_value() ? gterm->data_ptr<float>() : nullptr,
                                     opt_ptr(aux0), last(aux0), opt_ptr(aux1), last(aux1),
                                     red.has_value() ? red->data_ptr<float>() : nullptr, ci, cidx, cstride, (int)cinit_off,
                                     gw, gb, gr, cur_stream()),
               "conv2d_nhwc");
}

// ---- factor-graph side kernels (droid_amd.factor_graph / droid_amd.depth_video) ---------------------------------
// zero the canvas pixels outside the h x w image, in place (dh_canvas_mask_f16)
void canvas_mask_(torch::Tensor x, const int h, const int w) {
  CHECK_INPUT(x);
  TORCH_CHECK(x.scalar_type() == torch::kFloat16 && x.dim() == 4, "canvas_mask_: [N,Hc,Wc,C] float16");
  check_status(dh_canvas_mask_f16(x.data_ptr(), (int)x.size(0), (int)x.size(1), (int)x.size(2), (int)x.size(3), h, w, cur_stream()), "canvas_mask_");
}

torch::Tensor motion_features(torch::Tensor coords1, torch::Tensor target) {
  CHECK_INPUT(coords1); CHECK_INPUT(target); CHECK_F32(coords1); CHECK_F32(target);
  TORCH_CHECK(coords1.dim() == 4 && coords1.size(3) == 2 && target.sizes() == coords1.sizes(), "motion_features: [E,h,w,2]");
  const int E = (int)coords1.size(0), ht = (int)coords1.size(1), wd = (int)coords1.size(2);
  torch::Tensor flow = torch::empty({E, ht, wd, 8}, coords1.options().dtype(torch::kFloat16));
  check_status(dh_motion_features(coords1.data_ptr<float>(), target.data_ptr<float>(), flow.data_ptr(), E, ht, wd, cur_stream()), "motion_features");
  return flow;
}

// -> {target [E,h,w,2], weight [E,h,w,2], target_ba [E,2,h,w], weight_ba [E,2,h,w]}
std::vector<torch::Tensor> ba_inputs(torch::Tensor coords1, torch::Tensor dw) {
  CHECK_INPUT(coords1); CHECK_INPUT(dw); CHECK_F32(coords1); CHECK_F32(dw);
  TORCH_CHECK(coords1.dim() == 4 && coords1.size(3) == 2 && dw.dim() == 4 && dw.size(3) == 4 && dw.size(0) == coords1.size(0), "ba_inputs: shapes");
  const int E = (int)coords1.size(0), ht = (int)coords1.size(1), wd = (int)coords1.size(2);
  torch::Tensor target = torch::empty({E, ht, wd, 2}, coords1.options()), weight = torch::empty({E, ht, wd, 2}, coords1.options());
  torch::Tensor tb = torch::empty({E, 2, ht, wd}, coords1.options()), wb = torch::empty({E, 2, ht, wd}, coords1.options());
  check_status(dh_ba_inputs(coords1.data_ptr<float>(), dw.data_ptr<float>(), target.data_ptr<float>(), weight.data_ptr<float>(),
                            tb.data_ptr<float>(), wb.data_ptr<float>(), E, ht, wd, cur_stream()), "ba_inputs");
  return {target, weight, tb, wb};
}

torch::Tensor cvx_upsample(torch::Tensor disp, torch::Tensor mask) {
  CHECK_INPUT(disp); CHECK_INPUT(mask); CHECK_F32(disp);
  TORCH_CHECK(mask.scalar_type() == torch::kFloat16 && mask.dim() == 4 && mask.size(3) == 576 && disp.dim() == 3 &&
              mask.size(0) == disp.size(0) && mask.size(1) == disp.size(1) && mask.size(2) == disp.size(2),
              "cvx_upsample: disp [K,h,w] f32, mask [K,h,w,576] f16");
  const int K = (int)disp.size(0), ht = (int)disp.size(1), wd = (int)disp.size(2);
  torch::Tensor out = torch::empty({K, 8 * ht, 8 * wd}, disp.options());
  check_status(dh_cvx_upsample(disp.data_ptr<float>(), mask.data_ptr(), out.data_ptr<float>(), K, ht, wd, cur_stream()), "cvx_upsample");
  return out;
}

// dist [(t-t0)*(t-t1)] f32 (modified), edges_i/j: existing edges; returns {new_edges [2*max_new,2] i64, count [1] i32} (device)
std::vector<torch::Tensor> proximity_nms(torch::Tensor dist, torch::Tensor edges_i, torch::Tensor edges_j, int64_t t0, int64_t t1,
                                         int64_t t, int64_t rad, int64_t nms, double thresh, int64_t max_factors, int64_t n_es0,
                                         bool stereo, int64_t max_new) {
  CHECK_INPUT(dist); CHECK_F32(dist); CHECK_INPUT(edges_i); CHECK_INPUT(edges_j); CHECK_I64(edges_i); CHECK_I64(edges_j);
  TORCH_CHECK(dist.numel() == (t - t0) * (t - t1) && edges_i.numel() == edges_j.numel(), "proximity_nms: shapes");
  const int ne = (int)edges_i.numel();
  check_status(dh_proximity_nms(dist.data_ptr<float>(), nullptr, nullptr, edges_i.data_ptr<int64_t>(), edges_j.data_ptr<int64_t>(), ne, (int)t0,
                                (int)t1, (int)t, (int)rad, (int)nms, (float)thresh, (int)max_factors, (int)n_es0, stereo ? 1 : 0,
                                nullptr, 0, nullptr, 0, cur_stream()), "proximity_nms (mask)");
  auto srt = torch::sort(dist.view({-1}), /*stable=*/true, 0, false);
  torch::Tensor sorted = std::get<0>(srt).contiguous(), order = std::get<1>(srt).contiguous();
  torch::Tensor out = torch::zeros({2 * max_new, 2}, edges_i.options());
  torch::Tensor count = torch::zeros({1}, dist.options().dtype(torch::kInt32));
  check_status(dh_proximity_nms(dist.data_ptr<float>(), sorted.data_ptr<float>(), order.data_ptr<int64_t>(), nullptr, nullptr, 0, (int)t0, (int)t1, (int)t, (int)rad,
                                (int)nms, (float)thresh, (int)max_factors, (int)n_es0, stereo ? 1 : 0, out.data_ptr<int64_t>(),
                                (int)max_new, count.data_ptr<int>(), 1, cur_stream()), "proximity_nms (walk)");
  return {out, count};
}

torch::Tensor heads_gather(torch::Tensor partials, torch::Tensor bias4, int64_t H, int64_t W, int64_t mode) {
  CHECK_INPUT(partials); CHECK_INPUT(bias4); CHECK_F32(partials); CHECK_F32(bias4);
  TORCH_CHECK(partials.dim() == 5 && partials.size(2) == 6 && partials.size(3) == 64 && partials.size(4) == 4 && bias4.numel() >= 4 && W == 64 &&
              H % 4 == 0 && (partials.size(1) * 4) % H == 0,
              "heads_gather: partials [cout tiles, N*H/4 pixel tiles, 6, 64, 4] f32 of images with 64 columns");
  const int N = (int)(partials.size(1) * 4 / H);
  // mode 1: GraphAgg's eta head (one output, 0.01 * softplus) -> [N,H,W]
  torch::Tensor dw = mode == 1 ? torch::empty({N, H, W}, partials.options()) : torch::empty({N, H, W, 4}, partials.options());
  check_status(dh_heads_gather_ex(partials.data_ptr<float>(), bias4.data_ptr<float>(), dw.data_ptr<float>(), N, (int)H, (int)W,
                                  (int)partials.size(0), (int)mode, cur_stream()), "heads_gather");
  return dw;
}

// ---- encoder glue (droid_amd.encoder) ---------------------------------------------------------------------------------
torch::Tensor norm_act(torch::Tensor x, c10::optional<torch::Tensor> residual, bool normalize, bool relu) {
  CHECK_INPUT(x);
  TORCH_CHECK(x.scalar_type() == torch::kFloat16 && x.dim() == 4, "norm_act: x [N,H,W,C] float16");
  const int N = (int)x.size(0), HW = (int)(x.size(1) * x.size(2)), C = (int)x.size(3);
  const void* r = nullptr;
  if (residual.has_value()) {
    const torch::Tensor& rt = *residual; CHECK_INPUT(rt);
    TORCH_CHECK(rt.scalar_type() == torch::kFloat16 && rt.sizes() == x.sizes(), "norm_act: residual must match x");
    r = rt.data_ptr();
  }
  torch::Tensor y = torch::empty_like(x);
  torch::Tensor ws = torch::empty({normalize ? (int64_t)N * C * 2 : 1}, x.options().dtype(torch::kFloat32));
  check_status(dh_norm_act_nhwc_f16(x.data_ptr(), r, y.data_ptr(), ws.data_ptr<float>(), N, HW, C, normalize ? 1 : 0, relu ? 1 : 0,
                                    cur_stream()), "norm_act");
  return y;
}

// ---- extensions beyond the reference module (used by droid_amd / lietorch compat) ----------------
std::vector<torch::Tensor> reproject(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics,
                                     torch::Tensor ii, torch::Tensor jj) {
  CHECK_INPUT(poses); CHECK_INPUT(disps); CHECK_INPUT(intrinsics); CHECK_INPUT(ii); CHECK_INPUT(jj);
  CHECK_F32(poses); CHECK_F32(disps); CHECK_F32(intrinsics); CHECK_I64(ii); CHECK_I64(jj);
  const int E = (int)ii.size(0), ht = (int)disps.size(1), wd = (int)disps.size(2);
  torch::Tensor coords = torch::empty({E, ht, wd, 2}, poses.options());
  torch::Tensor valid = torch::empty({E, ht, wd, 1}, poses.options());
  // intrinsics [4]: one camera; [frames,4]: per frame (source frame's for the back-projection, target frame's for the projection)
  const int per_frame = intrinsics.dim() == 2 ? 1 : 0;
  TORCH_CHECK(intrinsics.size(-1) == 4 && (!per_frame || intrinsics.size(0) >= disps.size(0)),
              "reproject: intrinsics must be [4] or [frames,4] with one row per frame of disps");
  check_status(dh_reproject_ex(poses.data_ptr<float>(), disps.data_ptr<float>(), intrinsics.data_ptr<float>(), per_frame,
                               ii.data_ptr<int64_t>(), jj.data_ptr<int64_t>(), coords.data_ptr<float>(),
                               valid.data_ptr<float>(), E, ht, wd, cur_stream()),
               "reproject");
  return {coords, valid};
}

torch::Tensor se3_op(const std::string& op, torch::Tensor a, torch::Tensor b) {
  CHECK_INPUT(a); CHECK_F32(a);
  const bool two = (op == "mul" || op == "retr");
  if (two) { CHECK_INPUT(b); CHECK_F32(b); }
  const int64_t n = (op == "exp" || op == "retr") ? a.numel() / 6 : a.numel() / 7;
  auto shape = a.sizes().vec();
  shape.back() = (op == "log") ? 6 : 7;
  torch::Tensor out = torch::empty(shape, a.options());
  int rc;
  if (op == "log") rc = dh_se3_log(a.data_ptr<float>(), out.data_ptr<float>(), (int)n, cur_stream());
  else if (op == "inv") rc = dh_se3_inv(a.data_ptr<float>(), out.data_ptr<float>(), (int)n, cur_stream());
  else if (op == "mul") rc = dh_se3_mul(a.data_ptr<float>(), b.data_ptr<float>(), out.data_ptr<float>(), (int)n, cur_stream());
  else if (op == "exp") rc = dh_se3_exp(a.data_ptr<float>(), out.data_ptr<float>(), (int)n, cur_stream());
  else if (op == "retr") rc = dh_se3_retr(a.data_ptr<float>(), b.data_ptr<float>(), out.data_ptr<float>(), (int)n, cur_stream());
  else { TORCH_CHECK(false, "se3_op: unknown op ", op); rc = DH_ERR_ARG; }
  check_status(rc, "se3_op");
  return out;
}

// a [n,7], X [n,npts,D] with D = 4 (act) or 6 (adjT)
torch::Tensor se3_map(const std::string& op, torch::Tensor a, torch::Tensor X) {
  CHECK_INPUT(a); CHECK_INPUT(X); CHECK_F32(a); CHECK_F32(X);
  const int n = (int)(a.numel() / 7);
  const int D = (op == "act4") ? 4 : 6;
  TORCH_CHECK(X.size(-1) == D, "se3_map: last dim must be ", D);
  const int npts = n > 0 ? (int)(X.numel() / D / n) : 0;
  torch::Tensor Y = torch::empty_like(X);
  int rc = (op == "act4") ? dh_se3_act4(a.data_ptr<float>(), X.data_ptr<float>(), Y.data_ptr<float>(), n, npts, cur_stream())
                          : dh_se3_adjT(a.data_ptr<float>(), X.data_ptr<float>(), Y.data_ptr<float>(), n, npts, cur_stream());
  check_status(rc, "se3_map");
  return Y;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  // bundle adjustment kernels (names and order as reference src/droid.cpp:246-259)
  m.def("ba", &ba, "bundle adjustment");
  m.def("frame_distance", &frame_distance, "frame_distance");
  m.def("projmap", &projmap, "projmap");
  m.def("depth_filter", &depth_filter, "depth_filter");
  m.def("iproj", &iproj, "back projection");
  // correlation volume kernels
  m.def("altcorr_forward", &altcorr_forward, "ALTCORR forward");
  m.def("altcorr_backward", &altcorr_backward, "ALTCORR backward");
  m.def("corr_index_forward", &corr_index_forward, "INDEX forward");
  m.def("corr_index_backward", &corr_index_backward, "INDEX backward");
  // MI355X extensions (not in the reference module)
  m.def("ba_ex", &ba_ex, "bundle adjustment with a per-pixel depth-prior weight");
  m.def("ba_build", &ba_build, "per-rank reduced camera system (split BA)");
  m.def("ba_depth_blocks", &ba_depth_blocks, "views of Qinv = 1/C, w, kx, K in the workspace of ba_build (the keep-alive is the caller's: hold `ws`)");
  m.def("ba_finish", &ba_finish, "damp + solve + back-substitute + retract (split BA)");
  m.def("ba_build_shard", &ba_build_shard, "ba_build without a host synchronisation (edge-sharded BA)");
  m.def("ba_build_shard_ex", &ba_build_shard_ex, "ba_build_shard with a per-pixel depth-prior weight");
  m.def("ba_finish_owned", &ba_finish_owned, "ba_finish that only moves the depths of frames [own_lo, own_hi)");
  m.def("ba_pack_blocks", &ba_pack_blocks, "co-visible 6x6 blocks + rhs + flags -> one contiguous f64 buffer");
  m.def("ba_unpack_blocks", &ba_unpack_blocks, "all-reduced buffer -> system + flags");
  m.def("ba_exchange_flags", &ba_exchange_flags, "argument / pattern flags of the dense exchange");
  m.def("altcorr_forward_nhwc_levels", &altcorr_forward_nhwc_levels, "all pyramid levels of the on-the-fly correlation into one [M,49L,H,W] tensor");
  m.def("altcorr_forward_nhwc", &altcorr_forward_nhwc, "on-the-fly correlation on the fp16 MFMA, channel-last features");
  m.def("corr_volume_build", &corr_volume_build, "all-pairs volume in the reference layout (any image size)");
  m.def("corr_volume_pool", &corr_volume_pool, "2x2 average pooling of the last two dims of a reference-layout volume");
  m.def("corr_pyramid_build", &corr_pyramid_build, "all-pairs correlation pyramid, MI355X layout", py::arg("fmap1"), py::arg("fmap2"),
        py::arg("h_real") = 0, py::arg("w_real") = 0, py::arg("out") = py::none());
  m.def("corr_pyramid_prepare_frames", &corr_pyramid_prepare_frames, "channel-last transpose + pooled levels once per frame", py::arg("fmaps"),
        py::arg("h_real") = 0, py::arg("w_real") = 0);
  m.def("corr_pyramid_build_indexed", &corr_pyramid_build_indexed, "pyramid of edges (idx1[e], idx2[e]) from prepared frames", py::arg("prepared"),
        py::arg("idx1"), py::arg("idx2"), py::arg("h"), py::arg("w"), py::arg("out") = py::none());
  m.def("corr_pyramid_lookup", &corr_pyramid_lookup, "fused 4-level lookup on the MI355X pyramid");
  m.def("corr_pyramid_lookup_nhwc", &corr_pyramid_lookup_nhwc, "fused 4-level lookup, channel-last output for the update operator");
  m.def("corr_pyramid_lookup_corr0", &corr_pyramid_lookup_corr0, "4-level lookup fused with the correlation encoder's first layer (1x1, 196 -> 128, ReLU)");
  m.def("corr0_nchw", &corr0_nchw, "corr_encoder.0 (1x1, 196 -> 128, relu) on the reference-layout lookup output [E,196,h,w] -> [E,h,w,128]");
  m.def("glo_gemv", &glo_gemv, "global-context GEMV of the ConvGRU gates: fp16(bias + fp16(red * scale) wt)");
  m.def("segment_mean", &segment_mean, "mean over row segments (GraphAgg scatter_mean)");
  m.def("conv2d_s2_nhwc", &conv2d_s2_nhwc, "stride-2 same convolution (encoders' down-sampling layers)");
  m.def("conv2d_nhwc", &conv2d_nhwc, "implicit-GEMM NHWC convolution on the fp16 MFMA with fused epilogues",
        py::arg("inputs"), py::arg("weight"), py::arg("weight_halo"), py::arg("bias"), py::arg("KH"), py::arg("KW"), py::arg("Cout"),
        py::arg("epilogue"), py::arg("out"), py::arg("out_stride"), py::arg("gterm"), py::arg("aux0"), py::arg("aux1"), py::arg("red"),
        py::arg("cinit") = py::none(), py::arg("cinit_idx") = py::none(), py::arg("cinit_off") = 0, py::arg("out_raw_f32") = false,
        py::arg("weights_layout") = 0, py::arg("out_tiled") = false, py::arg("cinit_tiled") = false,
        py::arg("glo_weight") = py::none(), py::arg("glo_bias") = py::none(), py::arg("glo_red") = py::none());
  m.def("conv_set_timestamps", [](c10::optional<torch::Tensor> buf) {
    if (buf.has_value() && buf->defined()) {
      CHECK_INPUT((*buf));
      TORCH_CHECK(buf->scalar_type() == torch::kInt64 && buf->dim() == 2 && buf->size(1) == 8, "conv_set_timestamps: int64 [workgroups, 8]");
      check_status(dh_conv_set_timestamps(buf->data_ptr(), (long)buf->size(0)), "conv_set_timestamps");
    } else {
      check_status(dh_conv_set_timestamps(nullptr, 0), "conv_set_timestamps");
    }
  }, "-DDH_ABLATION builds: phase timestamps of the 3x3 convolution kernels into an int64 [workgroups, 8] tensor (None = off)");
  m.def("canvas_mask_", &canvas_mask_, "zero the canvas pixels outside the h x w image (fp16 NHWC, in place)");
  m.def("motion_features", &motion_features, "cat(coords1 - coords0, target - coords1).clamp(-64, 64) as fp16 NHWC");
  m.def("ba_inputs", &ba_inputs, "target = coords1 + delta, weight; also in ba's [E,2,h,w] layout");
  m.def("cvx_upsample", &cvx_upsample, "convex 8x upsampling of depth maps");
  m.def("proximity_nms", &proximity_nms, "candidate masking + greedy NMS of add_proximity_factors on the device");
  m.def("heads_gather", &heads_gather, "second head layer from the fused first layer's partial products",
        py::arg("partials"), py::arg("bias4"), py::arg("H"), py::arg("W"), py::arg("mode") = 0);
  m.def("norm_act", &norm_act, "instance norm / residual add + activation on channel-last fp16 (encoders)");
  m.def("reproject", &reproject, "fused reprojection (Python thresholds)");
  m.def("se3_op", &se3_op, "SE3 inv/mul/exp/retr");
  m.def("se3_map", &se3_map, "SE3 act4/adjT");
  m.def("version", []() { return std::string(dh_version()); });
  m.def("set_option", [](const std::string& name, int value) { check_status(dh_set_option(name.c_str(), value), "set_option"); });
  m.def("options_epoch", []() { return dh_options_epoch(); });
  m.def("get_option", [](const std::string& name) { int v = 0; check_status(dh_get_option(name.c_str(), &v), "get_option"); return v; });
}
