"""CPU oracle for the DROID-SLAM dense-BA update operator.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The product path
(``droid-slam_amd/``) never imports this package and fails loudly when its HIP
library is missing.

What it is: a plain numpy / torch-CPU restatement of the reference's algorithm
for the hot path (every function cites the reference file:line it follows):

* ``oracle.se3``    SE(3) helpers            (src/droid_kernels.cu:67-184, 886-904)
* ``oracle.ba``     ``ba_cuda``              (src/droid_kernels.cu:185-433, 863-1443)
* ``oracle.corr``   volume build + lookups   (src/correlation_kernels.cu, src/altcorr_kernel.cu,
                                              droid_slam/modules/corr.py)
* ``oracle.geom``   frame_distance / projmap / iproj / depth_filter / reproject
                                             (src/droid_kernels.cu:436-859,
                                              droid_slam/geom/projective_ops.py)
* ``oracle.update`` UpdateModule / ConvGRU / GraphAgg in fp32 torch-CPU
                                             (droid_slam/droid_net.py:44-143, modules/gru.py)
* ``oracle/build_ref.py`` + ``ref_extras.cu`` + ``ref_shims/``: recipe that compiles the REFERENCE's own
  ``src/droid.cpp`` + three ``.cu`` files for gfx950 where they lie under /root/reference into
  ``oracle/_ref/droid_backends_ref.so`` (git-ignored; travels to the GPU box).  The four CUDA include names map to
  the HIP / ATen-hip headers; Eigen (absent) is stood in by a dense fp64 LLT with SparseBlock's contract.

Pinning status: **pinned by the reference itself.**  The reference ships no golden vectors,
so they were produced here from its own code:

1. ``tests/golden/ref_cuda.npz`` -- outputs of the reference's CUDA kernels (``ba`` incl. stereo / sensor-depth /
   t0 > 1 / motion-only / failure cases, the per-edge blocks of ``projective_transform_kernel``, the reduced camera
   system before ``SparseBlock::solve``, ``corr_index_forward/backward`` in fp32 and fp16, ``altcorr_forward/backward``,
   ``frame_distance``, ``projmap``, ``iproj``, ``depth_filter``) written on an MI355X by
   ``tests/golden/make_ref_golden.py`` from ``oracle/_ref``; ``tests/test_oracle_ref_golden.py`` checks the oracle
   against them on CPU.  This pins ``ba_cuda``'s damping placement, MIN_DEPTH = 0.25, the ``EvT6x1`` row skip, the
   fp16 lookup and the unscaled ``altcorr`` backward.
2. the reference's own *Python* formulation (``droid_slam/geom/ba.py``, ``geom/projective_ops.py``,
   ``modules/corr.py``, ``droid_net.py``) imported from /root/reference by ``tests/golden/make_golden.py`` on CPU
   (small shims for the un-vendored ``lietorch`` / ``torch_scatter``) -> ``tests/golden/*_python.npz``,
   checked by ``tests/test_oracle_golden.py``.
3. on the GPU, ``tests/test_ref_parity.py`` runs the HIP product and ``oracle/_ref`` side by side on identical
   inputs up to the headline 512-keyframe / 4096-edge configuration.

What remains unpinned: Eigen's SimplicialLLT itself (a dense LLT computes the same factorisation up to fp64
round-off; AMD ordering only permutes the operations), and the un-vendored ``lietorch`` beyond the SE3 formulas that
``src/droid_kernels.cu:67-184`` restates in-tree.
"""
