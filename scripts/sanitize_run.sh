#!/bin/bash
# ASAN + UBSAN run of the host side of the C ABI.  Builds droid-slam_amd/sanitize/ if it is missing, then
#   * CPU (anywhere):  tests/test_sanitize_cpu.py  -- argument checks, size queries, option store, weight packing
#   * GPU box (if a device is visible): attempts the raw-pointer ctypes launches and the BA / lookup parity cases with the sanitized
#     host code around the real kernels.  MEASURED on the pool's boxes (profiles/r05_e_sanitizer_gpu_attempt.txt): not possible with
#     this image -- the ROCm compiler-rt ASAN runtime intercepts hsa_amd_memory_pool_allocate and aborts the first HIP allocation
#     ("out of memory: allocator is trying to allocate 0x400000 bytes"); it needs the ASAN builds of the ROCm runtime
#     (/opt/rocm/lib/asan, not installed).  The host-side coverage is therefore the CPU test.
# usage: bash scripts/sanitize_run.sh [OUTDIR]
OUT=${1:-gpurun_out/sanitize}; mkdir -p $OUT
cd "$(dirname "$0")/.."
[ -f droid-slam_amd/sanitize/libdroid_hip.so ] || DROID_HIP_SANITIZE=1 python droid-slam_amd/build.py
RT=$(python - <<'PY'
import importlib.util, os
s = importlib.util.spec_from_file_location("b", os.path.join("droid-slam_amd", "build.py")); m = importlib.util.module_from_spec(s); s.loader.exec_module(m)
print(m.asan_runtime() or "")
PY
)
[ -n "$RT" ] || { echo "no ASAN runtime"; exit 1; }
python -m pytest tests/test_sanitize_cpu.py -q 2>&1 | tail -n 3
if python -c "import torch,sys; sys.exit(0 if torch.cuda.is_available() else 1)" 2>/dev/null; then
  export LD_PRELOAD=$RT DROID_HIP_TEST_SANITIZE=1
  export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:protect_shadow_gap=0:exitcode=97 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1:exitcode=98
  # (1) does torch + the HIP runtime come up at all under the ASAN runtime on this box?  (2) one raw C-ABI launch  (3) the parity cases
  timeout 300 python -X faulthandler -u -c "import torch; x = torch.zeros(4, device='cuda'); torch.cuda.synchronize(); print('hip under asan ok', x.sum().item())" > $OUT/gpu_asan_probe.log 2>&1
  echo "hip-under-asan probe rc=$?"; tail -n 5 $OUT/gpu_asan_probe.log
  timeout 600 python -X faulthandler -u -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -p no:cacheprovider -k "raw_c_abi_call" > $OUT/gpu_asan_one.log 2>&1
  echo "one raw C-ABI test under asan rc=$?"; tail -n 8 $OUT/gpu_asan_one.log
  timeout 900 python -X faulthandler -u -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider \
      -k "raw_c_abi or ba_small or ba_config_c2 or pyramid_vs_oracle or lookup_fused or 64_cout" > $OUT/gpu_asan.log 2>&1
  echo "gpu tests under asan rc=$?"; tail -n 6 $OUT/gpu_asan.log; grep -c "AddressSanitizer\|runtime error:" $OUT/gpu_asan.log
fi
