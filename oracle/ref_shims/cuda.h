// Test infrastructure only (oracle/): lets the reference's CUDA sources compile with hipcc where they lie under
// /root/reference/src.  The reference includes <cuda.h>; on ROCm the same declarations come from the HIP runtime.
#pragma once
#include <hip/hip_runtime.h>
