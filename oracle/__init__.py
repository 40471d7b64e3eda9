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
* ``oracle/c``      C restatement of the BA kernels used as the timed CPU baseline.

Pinning status: the reference ships NO golden vectors, known-answer tests or
fixtures for this path, and its CUDA extension cannot be
built here (no nvcc, no Eigen, no NVIDIA GPU).  What pins the oracle instead:

1. the reference's own *Python* formulation of the same math
   (``droid_slam/geom/ba.py``, ``geom/projective_ops.py``, ``modules/corr.py``,
   ``droid_net.py``) imported from /root/reference by
   ``tests/golden/make_golden.py`` on CPU (with small shims for the un-vendored
   ``lietorch`` / ``torch_scatter``); its outputs are committed under
   ``tests/golden/`` and ``tests/test_oracle_golden.py`` checks the oracle
   against them;
2. self-consistency properties (finite-difference Jacobians, sparse-vs-dense
   Gauss-Newton agreement, cost decrease).

The CUDA-only semantics that the Python formulation does not exercise
(``ba_cuda``'s damping placement, MIN_DEPTH=0.25, the ``EvT6x1`` row skip, the
fp16 accumulation order of the lookup kernels) are restated from the source
but are **parity unpinned** by any reference-produced vector.
"""
