"""Profiler ranges and event timers for the hot path (the MI355X counterpart of the reference's
droid_slam/cuda_timer.py:3-23).

``roctx_range(name)``  a ROCTX range (librocprofiler-sdk-roctx, else libroctx64: roctxRangePushA / roctxRangePop) around a stage of FactorGraph.update /
                       update_lowmem / DistBA.ba.  `rocprofv3 --marker-trace --kernel-trace -- <cmd>` shows the kernels of a
                       step grouped under reproject / corr_lookup / update_operator / ba / upsample.  A push/pop pair costs
                       ~100 ns with no tool attached; DROID_HIP_ROCTX=0 turns the ranges into no-ops.
``HipTimer(name)``     the reference's CudaTimer on HIP events of the CURRENT stream: `with HipTimer("ba"): ...` prints the
                       elapsed milliseconds (synchronises, like the reference's; for ad-hoc measurements, not the product path).
"""
import contextlib
import ctypes
import os

import torch

_lib = None
_enabled = os.environ.get("DROID_HIP_ROCTX", "1") != "0"


def _roctx():
    global _lib, _enabled
    if _lib is None and _enabled:
        # rocprofv3 (rocprofiler-sdk) records the ranges of ITS roctx library; the roctracer-era libroctx64 is the fall-back for older tools
        for name in ("librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so", "libroctx64.so.4",
                     "/opt/rocm/lib/libroctx64.so"):
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.restype = ctypes.c_int
                _lib = lib
                break
            except (OSError, AttributeError):
                continue
        if _lib is None:
            _enabled = False            # no ROCTX library next to this ROCm: ranges are no-ops (kernels are unaffected)
    return _lib


def roctx_available():
    return _roctx() is not None


@contextlib.contextmanager
def roctx_range(name):
    lib = _roctx()
    if lib is None:
        yield
        return
    lib.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        lib.roctxRangePop()


class HipTimer:
    """`with HipTimer("stage", enabled): ...` -- HIP events around the block on the current stream; prints `stage <ms>` at exit
    (reference cuda_timer.py:3-23) and keeps the value in `.ms`"""

    def __init__(self, name, enabled=True, quiet=False):
        self.name, self.enabled, self.quiet, self.ms = name, enabled, quiet, None
        if enabled:
            self.start = torch.cuda.Event(enable_timing=True)
            self.end = torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.enabled:
            self.start.record()
        return self

    def __exit__(self, exc_type, exc, tb):
        if self.enabled:
            self.end.record()
            self.end.synchronize()
            self.ms = self.start.elapsed_time(self.end)
            if not self.quiet:
                print(self.name, self.ms)
