// Micro-benchmark: vector-memory instruction rates per CU on gfx950 (L2-resident data).
// build: hipcc --offload-arch=gfx950 -O3 -o ta_rate ta_rate.hip ; run: ./ta_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename T> __device__ inline uint32_t fold(T v);
template <> __device__ inline uint32_t fold<unsigned short>(unsigned short v) { return (uint32_t)v * 2654435761u; }
template <> __device__ inline uint32_t fold<uint32_t>(uint32_t v) { return v; }
template <> __device__ inline uint32_t fold<uint2>(uint2 v) { return v.x ^ v.y; }
template <> __device__ inline uint32_t fold<uint4>(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// each wave streams over its own 16 KB window (L1-miss / L2-hit after first touch when `span` is large)
template <typename T, int UNROLL>
__global__ __launch_bounds__(256) void load_kernel(const T* __restrict__ buf, uint32_t* out, int iters, int span_elems, int lane_stride) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const T* p = buf + (size_t)wave * span_elems;
  uint32_t acc = 0;
  int off = lane * lane_stride;
  for (int it = 0; it < iters; ++it) {
    T v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = p[(off + u * 64 * lane_stride) & (span_elems - 1)];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= fold<T>(v[u]);
    off = (off + UNROLL * 64 * lane_stride) & (span_elems - 1);
  }
  if (acc == 0x9e3779b9u || iters < 0) out[wave] = acc;
}

template <typename T, int UNROLL>
__global__ __launch_bounds__(256) void store_kernel(T* __restrict__ buf, int iters, int span_elems) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  T* p = buf + (size_t)wave * span_elems;
  T val; __builtin_memset(&val, 0, sizeof(T));
  int off = lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) p[(off + u * 64) & (span_elems - 1)] = val;
    off = (off + UNROLL * 64) & (span_elems - 1);
  }
}

template <typename F> float time_ms(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
  int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  const int CUs = prop.multiProcessorCount; const double ghz = prop.clockRate * 1e-6;
  printf("CUs %d clock %.2f GHz\n", CUs, ghz);
  const int waves_per_cu = 16, blocks = CUs * waves_per_cu / 4;
  const size_t nwaves = (size_t)blocks * 4;
  void* buf; const size_t bytes_per_wave = 16384; CK(hipMalloc(&buf, nwaves * bytes_per_wave)); CK(hipMemset(buf, 1, nwaves * bytes_per_wave));
  uint32_t* out; CK(hipMalloc(&out, 4 * nwaves));
  const int iters = 512;
  auto report = [&](const char* name, float ms, int instr_per_wave, int bytes_per_instr) {
    double instr_per_cu = (double)nwaves * instr_per_wave / CUs;
    double cyc = ms * 1e-3 * ghz * 1e9 / instr_per_cu;
    printf("%-48s %8.3f ms  %6.1f cycles/wave-instr/CU  %7.1f B/clk/CU  %6.2f TB/s chip\n", name, ms, cyc, bytes_per_instr / cyc,
           (double)nwaves * instr_per_wave * bytes_per_instr / ms / 1e9);
  };
#define LOADT(T, name, stride) for (size_t span : {(size_t)1024, (size_t)16384}) { float ms = time_ms([&] { hipLaunchKernelGGL((load_kernel<T, 8>), dim3(blocks), dim3(256), 0, 0, (const T*)buf, out, iters, (int)(span / sizeof(T)), stride); }); char nm[96]; snprintf(nm, 96, "%s span %zu", name, span); report(nm, ms, iters * 8, 64 * sizeof(T)); }
  LOADT(unsigned short, "load ushort contiguous (128B/wave)", 1)
  LOADT(uint32_t, "load dword contiguous (256B/wave)", 1)
  LOADT(uint2, "load dwordx2 contiguous (512B)", 1)
  LOADT(uint4, "load dwordx4 contiguous (1KB)", 1)
  LOADT(unsigned short, "load ushort stride 2 (256B span)", 2)
  LOADT(unsigned short, "load ushort stride 64 (64 lines)", 64)
  LOADT(uint32_t, "load dword stride 32 (64 lines)", 32)
#define STORET(T, name) for (size_t span : {(size_t)1024, (size_t)16384}) { float ms = time_ms([&] { hipLaunchKernelGGL((store_kernel<T, 8>), dim3(blocks), dim3(256), 0, 0, (T*)buf, iters, (int)(span / sizeof(T))); }); char nm[96]; snprintf(nm, 96, "%s span %zu", name, span); report(nm, ms, iters * 8, 64 * sizeof(T)); }
  STORET(unsigned short, "store short contiguous")
  STORET(uint32_t, "store dword contiguous")
  STORET(uint2, "store dwordx2 contiguous")
  STORET(uint4, "store dwordx4 contiguous")
  return 0;
}
