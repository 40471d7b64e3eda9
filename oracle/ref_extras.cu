// Test infrastructure only (oracle/): never linked into the product (droid-slam_amd/), never timed.
//
// This translation unit IS the reference's src/droid_kernels.cu (included where it lies under /root/reference, not
// copied) plus two probe functions that launch the reference's own kernels / host functions and hand back the
// intermediate quantities the reference's `ba` keeps private: the per-edge blocks of projective_transform_kernel
// (droid_kernels.cu:185-433) and the reduced camera system right before SparseBlock::solve (droid_kernels.cu:1385-1415).
// The tests compare droid_backends.ba_build / the oracle against them.
// Registered as torch.ops.droid_ref.* by the shared object oracle/_ref/droid_backends_ref.so (oracle/build_ref.py).
#include "droid_kernels.cu"
#include <torch/library.h>

static std::vector<torch::Tensor> ref_edge_blocks(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics,
                                                  torch::Tensor targets, torch::Tensor weights,
                                                  torch::Tensor ii, torch::Tensor jj) {
  auto opts = poses.options();
  const int num = ii.size(0), ht = disps.size(1), wd = disps.size(2);
  torch::Tensor Hs = torch::zeros({4, num, 6, 6}, opts), vs = torch::zeros({2, num, 6}, opts);
  torch::Tensor Eii = torch::zeros({num, 6, ht * wd}, opts), Eij = torch::zeros({num, 6, ht * wd}, opts);
  torch::Tensor Cii = torch::zeros({num, ht * wd}, opts), wi = torch::zeros({num, ht * wd}, opts);
  projective_transform_kernel<<<num, THREADS>>>(
      targets.packed_accessor32<float, 4, torch::RestrictPtrTraits>(), weights.packed_accessor32<float, 4, torch::RestrictPtrTraits>(),
      poses.packed_accessor32<float, 2, torch::RestrictPtrTraits>(), disps.packed_accessor32<float, 3, torch::RestrictPtrTraits>(),
      intrinsics.packed_accessor32<float, 1, torch::RestrictPtrTraits>(),
      ii.packed_accessor32<LongType, 1, torch::RestrictPtrTraits>(), jj.packed_accessor32<LongType, 1, torch::RestrictPtrTraits>(),
      Hs.packed_accessor32<float, 4, torch::RestrictPtrTraits>(), vs.packed_accessor32<float, 3, torch::RestrictPtrTraits>(),
      Eii.packed_accessor32<float, 3, torch::RestrictPtrTraits>(), Eij.packed_accessor32<float, 3, torch::RestrictPtrTraits>(),
      Cii.packed_accessor32<float, 2, torch::RestrictPtrTraits>(), wi.packed_accessor32<float, 2, torch::RestrictPtrTraits>());
  return {Hs, vs, Eii, Eij, Cii, wi};
}

// [H (6P x 6P, fp64, = A - S before damping), b (6P), C (K x HW), w (K x HW)] of the first iteration of ba_cuda
static std::vector<torch::Tensor> ref_reduced_system(torch::Tensor poses, torch::Tensor disps, torch::Tensor intrinsics,
                                                     torch::Tensor disps_sens, torch::Tensor targets, torch::Tensor weights,
                                                     torch::Tensor eta, torch::Tensor ii, torch::Tensor jj,
                                                     int64_t t0, int64_t t1, bool motion_only) {
  const int num = ii.size(0), ht = disps.size(1), wd = disps.size(2), P = t1 - t0;
  auto blk = ref_edge_blocks(poses, disps, intrinsics, targets, weights, ii, jj);
  SparseBlock A(P, 6);
  A.update_lhs(blk[0].reshape({-1, 6, 6}), torch::cat({ii, ii, jj, jj}) - t0, torch::cat({ii, jj, ii, jj}) - t0);
  A.update_rhs(blk[1].reshape({-1, 6}), torch::cat({ii, jj}) - t0);
  auto dense = [&](SparseBlock& B) {
    torch::Tensor H = torch::from_blob(B.A.d.data(), {6 * P, 6 * P}, torch::kFloat64).clone().t().contiguous();   // column-major store
    torch::Tensor b = torch::from_blob(B.b.data(), {6 * P}, torch::kFloat64).clone();
    return std::make_pair(H, b);
  };
  if (motion_only) { auto hb = dense(A); return {hb.first, hb.second}; }
  torch::Tensor ts = torch::arange(t0, t1).to(torch::kCUDA);
  torch::Tensor ii_exp = torch::cat({ts, ii}, 0), jj_exp = torch::cat({ts, jj}, 0);
  auto kuniq = torch::_unique(ii_exp, true, true);
  torch::Tensor kx = std::get<0>(kuniq), kk_exp = std::get<1>(kuniq);
  const float alpha = 0.05;
  torch::Tensor m = (disps_sens.index({kx, "..."}) > 0).to(torch::kFloat32).view({-1, ht * wd});
  torch::Tensor C = accum_cuda(blk[4], ii, kx) + m * alpha + (1 - m) * eta.view({-1, ht * wd});
  torch::Tensor w = accum_cuda(blk[5], ii, kx) - m * alpha * (disps.index({kx, "..."}) - disps_sens.index({kx, "..."})).view({-1, ht * wd});
  torch::Tensor Q = 1.0 / C;
  torch::Tensor Ei = accum_cuda(blk[2].view({num, 6 * ht * wd}), ii, ts).view({P, 6, ht * wd});
  torch::Tensor E = torch::cat({Ei, blk[3]}, 0);
  SparseBlock S = schur_block(E, Q, w, ii_exp, jj_exp, kk_exp, t0, t1);
  SparseBlock R = A - S;
  auto hb = dense(R);
  return {hb.first, hb.second, C, w};
}

TORCH_LIBRARY(droid_ref, m) {
  m.def("edge_blocks(Tensor poses, Tensor disps, Tensor intrinsics, Tensor targets, Tensor weights, Tensor ii, Tensor jj) -> Tensor[]", &ref_edge_blocks);
  m.def("reduced_system(Tensor poses, Tensor disps, Tensor intrinsics, Tensor disps_sens, Tensor targets, Tensor weights, Tensor eta, Tensor ii, Tensor jj, int t0, int t1, bool motion_only) -> Tensor[]", &ref_reduced_system);
}
