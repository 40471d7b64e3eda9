import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "droid-slam_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
# DROID_HIP_TEST_ABLATION=1: run the suite against the -DDH_ABLATION build (droid-slam_amd/ablation/, built by
# DROID_HIP_ABLATION=1 python droid-slam_amd/build.py): the tests of the prototype / measurement kernels then run instead of skipping
if os.environ.get("DROID_HIP_TEST_ABLATION", "0") == "1":
    sys.path.insert(0, os.path.join(PKG, "ablation"))

# DROID_HIP_TEST_SANITIZE=1: the ASAN + UBSAN host build (droid-slam_amd/sanitize/, DROID_HIP_SANITIZE=1 python droid-slam_amd/build.py);
# python must run under the matching runtime (tests/test_sanitize_cpu.py starts such a process)
if os.environ.get("DROID_HIP_TEST_SANITIZE", "0") == "1":
    sys.path.insert(0, os.path.join(PKG, "sanitize"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
