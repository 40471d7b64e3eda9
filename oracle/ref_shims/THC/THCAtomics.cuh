// Test infrastructure only (oracle/): the reference's altcorr_kernel.cu includes <THC/THCAtomics.cuh> for
// atomicAdd(at::Half*, at::Half); PyTorch-ROCm ships the same overloads in ATen/hip/Atomic.cuh.
#pragma once
#include <ATen/hip/Atomic.cuh>
