"""The host side of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer (the sanitizer host-test build).

`DROID_HIP_SANITIZE=1 python droid-slam_amd/build.py` builds libdroid_hip.so with its HOST code instrumented into
droid-slam_amd/sanitize/ (device code unchanged).  This test runs the CPU-side boundary tests of tests/test_capi_cpu.py -- argument
checks of every entry point, workspace / pyramid / packed-exchange size queries, the option store, weight packing against the
layout rules -- in a python process that has the matching ASAN runtime preloaded; any heap overflow, use-after-free, signed
overflow or misaligned access in those paths aborts it.  Skipped when the sanitize build is absent (it is git-ignored and takes
~1.5 min; `scripts/sanitize_run.sh` builds and runs it, on a GPU box also the ctypes raw-pointer launches)."""
import os
import subprocess
import sys
import importlib.util

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "droid-slam_amd")


def _asan_runtime():
    spec = importlib.util.spec_from_file_location("droid_hip_build", os.path.join(PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.asan_runtime()


def test_c_abi_host_paths_under_asan_and_ubsan():
    lib = os.path.join(PKG, "sanitize", "libdroid_hip.so")
    if not os.path.exists(lib):
        pytest.skip("sanitize build absent (DROID_HIP_SANITIZE=1 python droid-slam_amd/build.py)")
    srcs = [os.path.join(PKG, "csrc", f) for f in os.listdir(os.path.join(PKG, "csrc"))] + [os.path.join(ROOT, "include", "droid_hip.h")]
    if os.path.getmtime(lib) < max(os.path.getmtime(f) for f in srcs):
        pytest.skip("sanitize build is older than the sources (DROID_HIP_SANITIZE=1 python droid-slam_amd/build.py)")
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("no AddressSanitizer runtime next to hipcc's clang")
    env = dict(os.environ, LD_PRELOAD=rt, DROID_HIP_TEST_SANITIZE="1",
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=97",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=98")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_capi_cpu.py"), "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "argument_errors or exports_every or round3_entry or option_store or weight_packing or shipped_library"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    tail = r.stdout[-3000:]
    assert "AddressSanitizer" not in r.stdout and "runtime error:" not in r.stdout, tail
    assert r.returncode == 0, tail
    assert " passed" in r.stdout, tail
