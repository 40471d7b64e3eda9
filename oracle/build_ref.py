#!/usr/bin/env python
"""Test infrastructure only: build the REFERENCE's own droid_backends for gfx950 as oracle/_ref/droid_backends_ref.so.

The reference's extension (src/droid.cpp + three .cu files) is plain CUDA C whose only CUDA-specific pieces are four
include names; its one external dependency is Eigen (SparseBlock, droid_kernels.cu:1126-1228), absent here.  This
recipe compiles the sources WHERE THEY LIE under /root/reference/src (nothing is copied, nothing is hipified):

  -I oracle/ref_shims   cuda.h / cuda_runtime.h / cuda_fp16.h / THC/THCAtomics.cuh -> the HIP / ATen-hip headers,
                        Eigen/{Sparse,SparseCore,SparseCholesky} -> a dense fp64 LLT with SparseBlock's contract
  oracle/ref_extras.cu  = `#include "droid_kernels.cu"` + two probes (per-edge blocks, reduced camera system)
  correlation_kernels.cu, altcorr_kernel.cu, droid.cpp (TORCH_EXTENSION_NAME=droid_backends_ref) as they are

The reference's own build system (setup.py) is not run.  Output goes to oracle/_ref/ only (git-ignored; it travels to
the GPU box with the gpurun snapshot).  Only tests/ may load it.  It cannot be built on the GPU box (/root/reference
is absent there): the committed recipe is run here, by __graft_entry__.build().
"""
import os
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DROID_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "src")
OUT = os.path.join(HERE, "_ref")
SO = os.path.join(OUT, "droid_backends_ref.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = os.environ.get("DROID_HIP_ARCH", "gfx950")


def available():
    return os.path.isdir(SRC)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout[-6000:]))


def build(force=False):
    if not available():
        return SO if os.path.exists(SO) else None
    import torch
    from torch.utils import cpp_extension as ce
    os.makedirs(OUT, exist_ok=True)
    extras = os.path.join(HERE, "ref_extras.cu")
    shims = os.path.join(HERE, "ref_shims")
    deps = [extras, os.path.abspath(__file__)] + [os.path.join(SRC, f) for f in os.listdir(SRC)] + \
           [os.path.join(d, f) for d, _, fs in os.walk(shims) for f in fs]
    if not force and os.path.exists(SO) and os.path.getmtime(SO) >= max(map(os.path.getmtime, deps)):
        return SO
    inc = ["-I", shims, "-I", SRC]
    for p in ce.include_paths():
        inc += ["-isystem", p]
    inc += ["-isystem", "/opt/rocm/include", "-I", sysconfig.get_paths()["include"]]
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    common = ["-fPIC", "-std=c++17", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DHIPBLAS_V2",
              "-DTORCH_EXTENSION_NAME=droid_backends_ref", "-DTORCH_API_INCLUDE_EXTENSION_H",
              "-D_GLIBCXX_USE_CXX11_ABI=%d" % abi, "-Wno-deprecated-declarations", "-w"]
    hipflags = ["-x", "hip", "--offload-arch=" + ARCH, "-O3", "-DCUDA_HAS_FP16=1", "-D__HIP_NO_HALF_OPERATORS__=1",
                "-D__HIP_NO_HALF_CONVERSIONS__=1", "-fno-gpu-rdc"]
    jobs, objs = [], []
    for name, src in (("ref_extras", extras), ("correlation_kernels", os.path.join(SRC, "correlation_kernels.cu")),
                      ("altcorr_kernel", os.path.join(SRC, "altcorr_kernel.cu"))):
        o = os.path.join(OUT, name + ".o")
        objs.append(o)
        jobs.append([HIPCC] + hipflags + common + inc + ["-c", src, "-o", o])
    o = os.path.join(OUT, "droid.o")
    objs.append(o)
    jobs.append([os.environ.get("CXX", "g++"), "-O2"] + common + inc + ["-c", os.path.join(SRC, "droid.cpp"), "-o", o])
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(_run, jobs))
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", SO] + objs +
         ["-L" + tlib, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python",
          "-Wl,-rpath," + tlib])
    return SO


PYZIP = os.path.join(OUT, "ref_py.zip")
# the reference's hot-path CALLERS, to be run UNCHANGED on this repo's droid_backends / lietorch / torch_scatter on a GPU
# (tests/test_ref_callers_gpu.py).  /root/reference does not exist on the GPU box, so they travel the way the compiled
# reference does: as one git-ignored archive under oracle/_ref/ (python imports straight from it, zipimport); nothing
# of the reference is unpacked into the tree or committed.
PY_FILES = ["factor_graph.py", "depth_video.py", "droid_net.py", "droid_frontend.py", "droid_backend.py", "motion_filter.py",
            "trajectory_filler.py", "cuda_timer.py", "geom/__init__.py", "geom/ba.py", "geom/chol.py", "geom/graph_utils.py",
            "geom/projective_ops.py", "geom/losses.py", "modules/__init__.py", "modules/clipping.py", "modules/corr.py",
            "modules/extractor.py", "modules/gru.py",
            "data_readers/__init__.py", "data_readers/rgbd_utils.py"]          # (imported by geom/graph_utils.py at module scope)


def stage_python(force=False):
    """oracle/_ref/ref_py.zip <- the reference's droid_slam/*.py listed above, byte for byte"""
    pydir = os.path.join(REF, "droid_slam")
    if not os.path.isdir(pydir):
        return PYZIP if os.path.exists(PYZIP) else None
    import zipfile
    srcs = [os.path.join(pydir, f) for f in PY_FILES]
    if not force and os.path.exists(PYZIP) and os.path.getmtime(PYZIP) >= max(map(os.path.getmtime, srcs + [os.path.abspath(__file__)])):
        return PYZIP
    os.makedirs(OUT, exist_ok=True)
    with zipfile.ZipFile(PYZIP, "w", zipfile.ZIP_DEFLATED) as z:
        for f, src in zip(PY_FILES, srcs):
            z.write(src, "droid_slam/" + f)
    return PYZIP


def load():
    """Import the reference module (tests only).  Returns (droid_backends_ref, torch.ops.droid_ref) or None."""
    if not os.path.exists(SO):
        return None
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("droid_backends_ref", SO)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, torch.ops.droid_ref


if __name__ == "__main__":
    print("built:", build(force="--force" in sys.argv), stage_python(force="--force" in sys.argv))
