// Test infrastructure only (oracle/): see cuda.h in this directory.
#pragma once
#include <hip/hip_runtime.h>
