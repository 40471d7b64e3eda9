// Probe: (1) largest dynamic LDS size a kernel may opt into on gfx950, (2) semantics of global_load_lds_dwordx4 issued
// from inline asm: destination = M0 + lane * 16 (lane-linear), M0 beyond 64 KB, per-lane source addresses.
// build: hipcc --offload-arch=gfx950 -O3 -o glds_probe glds_probe.hip ; run: ./glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef __attribute__((address_space(3))) char lds_char;

__global__ __launch_bounds__(512) void touch_kernel(uint32_t* out, int bytes) {
  extern __shared__ char s[];
  if (threadIdx.x == 0) { s[bytes - 1] = 7; out[0] = (uint32_t)s[bytes - 1] + (uint32_t)(uintptr_t)(lds_char*)s; }
}

// every wave copies 1 KB pieces: piece q of the workgroup goes to LDS byte offset dst_off + q * 1024, lane l fetching
// the 16 bytes src[perm(q, l)]; then the LDS image is written back linearly.
__global__ __launch_bounds__(512) void dma_kernel(const uint4* __restrict__ src, uint4* __restrict__ out, int pieces, int dst_off) {
  extern __shared__ char s[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)s;
  for (int q = wave; q < pieces; q += 8) {
    const uint4* g = src + q * 64 + (lane ^ (q & 7));                    // a per-lane permutation of the source piece
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + dst_off + q * 1024);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  for (int i = tid; i < pieces * 64; i += 512) out[i] = *reinterpret_cast<const uint4*>(s + dst_off + i * 16);
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("sharedMemPerBlock %zu optin %zu perMultiprocessor %zu\n", prop.sharedMemPerBlock, prop.sharedMemPerBlockOptin, prop.sharedMemPerMultiprocessor);
  uint32_t* out; CK(hipMalloc(&out, 64));
  for (int kb : {64, 96, 128, 132, 136, 144, 152, 156, 159, 160}) {
    hipError_t a = hipFuncSetAttribute(reinterpret_cast<const void*>(&touch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
    hipError_t l = hipSuccess, s = hipSuccess;
    if (a == hipSuccess) {
      hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(512), kb * 1024, 0, out, kb * 1024);
      l = hipGetLastError(); s = hipDeviceSynchronize();
    } else (void)hipGetLastError();
    printf("lds %3d KB: attr %s launch %s sync %s\n", kb, hipGetErrorName(a), hipGetErrorName(l), hipGetErrorName(s));
  }
  const int pieces = 20;
  std::vector<uint32_t> h(pieces * 64 * 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)i * 2654435761u + 12345u;
  uint4 *dsrc, *dout; CK(hipMalloc(&dsrc, h.size() * 4)); CK(hipMalloc(&dout, h.size() * 4));
  CK(hipMemcpy(dsrc, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  for (int dst_off : {0, 40 * 1024, 100 * 1024}) {
    const int lds = dst_off + pieces * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipMemset(dout, 0, h.size() * 4));
    hipLaunchKernelGGL(dma_kernel, dim3(1), dim3(512), lds, 0, dsrc, dout, pieces, dst_off);
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
    std::vector<uint32_t> r(h.size()); CK(hipMemcpy(r.data(), dout, h.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int q = 0; q < pieces; ++q) for (int l = 0; l < 64; ++l) for (int k = 0; k < 4; ++k)
      if (r[(q * 64 + l) * 4 + k] != h[(q * 64 + (l ^ (q & 7))) * 4 + k]) ++bad;
    printf("dma dst_off %6d: %zu mismatching dwords of %zu\n", dst_off, bad, h.size());
  }
  return 0;
}
