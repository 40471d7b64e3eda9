// Micro-benchmark behind the Winograd decision for the 3x3 gate convolutions of the update operator:
// how fast does a CU run v_mfma_f32_32x32x16_f16 when every MFMA's operands come from LDS at the ratios the candidate
// formulations need?  The production kernel (conv3x3_halo2_kernel) keeps a 64x64 wave tile: per 16-channel k-step it reads
// 2 A + 2 B fragments (1 KB each, one ds_read_b128 per lane) for 4 MFMAs = 1 fragment read per MFMA.  Winograd multiplies
// the accumulators per output by 4 (F(2x2,3x3): 16 transform positions per 2x2 outputs) or 2 (F(2,3) along x only), so at
// the same register budget the wave tile shrinks and the fragment reads per MFMA grow:
//
//   direct    64x64 tile                          4 reads / 4 MFMA   = 1.0 read per MFMA     x1.00 MACs
//   wino1d    F(2,3) along x, 32 pairs x 64 cout   (1 A + 2 B) / 2    = 1.5 reads per MFMA    /1.50 MACs
//   wino2d    F(2x2,3x3), 32 tiles x 32 cout x 16 positions: (1 A + 1 B) / 1 = 2.0 reads per MFMA    /2.25 MACs
//
// The kernels below issue exactly those instruction mixes (operands really read from LDS, conflict-free addresses, results
// kept alive), 8 waves per CU like the production kernel, zeros as data (no DVFS effect: this is the upper bound).
// "effective" = MFMA rate x the formulation's MAC saving = the direct-convolution-equivalent rate an ideal kernel of that
// shape could reach BEFORE paying for the input / output transforms, the 4x larger transformed-weight stream and the
// cross-wave exchange.
// build: hipcc --offload-arch=gfx950 -O3 -o wino_lds wino_lds.hip ; run: ./wino_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// NA / NB fragment reads and NM MFMAs per step; accumulators: NACC independent 32x32 tiles (round robin)
template <int NA, int NB, int NM, int NACC>
__global__ __launch_bounds__(512) void mix_kernel(float* out, int steps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int o = tid * 16; o < 64 * 1024; o += 512 * 16) *reinterpret_cast<uint4*>(lds + o) = uint4{0u, 0u, 0u, 0u};
  __syncthreads();
  f32x16 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
  // a fragment = 64 lanes x 16 B = 1 KB; fragments of a step at consecutive KBs, offset by the wave so that the 8 waves
  // walk different parts of the 64 KB window (like different cout / pixel tiles)
  int base = wave * 8192;
  for (int s = 0; s < steps; ++s) {
    half8 a[NA], b[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) a[i] = *reinterpret_cast<const half8*>(lds + ((base + i * 1024 + lane * 16) & 65535));
#pragma unroll
    for (int i = 0; i < NB; ++i) b[i] = *reinterpret_cast<const half8*>(lds + ((base + (NA + i) * 1024 + lane * 16) & 65535));
#pragma unroll
    for (int m = 0; m < NM; ++m)
      acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m % NA], b[m % NB], acc[m % NACC], 0, 0, 0);
    base = (base + (NA + NB) * 1024) & 65535;
  }
  float r = 0.f;
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc[a][i];
  if (r == 12345.f) out[tid] = r;
}

template <int NA, int NB, int NM, int NACC>
int run(const char* name, double mac_saving, int cus, double ghz) {
  float* out; CK(hipMalloc(&out, 4096));
  const int steps = 20000 / NM * 4;
  CK(hipFuncSetAttribute((const void*)mix_kernel<NA, NB, NM, NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((mix_kernel<NA, NB, NM, NACC>), dim3(cus), dim3(512), 64 * 1024, 0, out, steps);
  CK(hipDeviceSynchronize());
  hipEventRecord(e0);
  hipLaunchKernelGGL((mix_kernel<NA, NB, NM, NACC>), dim3(cus), dim3(512), 64 * 1024, 0, out, steps);
  hipEventRecord(e1); CK(hipEventSynchronize(e1));
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = 2.0 * 32 * 32 * 16 * (double)NM * steps * 8 * cus;
  const double tf = flop / (ms * 1e-3) / 1e12;
  const double lds_bpc = (double)(NA + NB) * 1024 * steps * 8 / (ms * 1e-3 * ghz * 1e9);
  printf("%-8s %2d+%2d fragment reads / %2d MFMA  %.3f ms  MFMA %7.1f TFLOP/s (%.0f%% of 2500)  LDS %5.1f B/clk/CU  effective %7.1f TFLOP/s direct-equivalent\n",
         name, NA, NB, NM, ms, tf, 100.0 * tf / 2500.0, lds_bpc, tf * mac_saving);
  hipFree(out);
  return 0;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount; const double ghz = prop.clockRate * 1e-6;
  printf("CUs %d clock %.2f GHz, 8 waves per CU, operands = zeros\n", cus, ghz);
  if (run<2, 2, 4, 4>("direct", 1.0, cus, ghz)) return 1;          // 64x64 wave tile
  if (run<1, 2, 2, 2>("wino1d", 1.5, cus, ghz)) return 1;          // per (position, dy): 32 pairs x 64 cout
  if (run<4, 4, 4, 4>("wino2d", 2.25, cus, ghz)) return 1;         // 4 of the 16 positions per step: 1 A + 1 B per MFMA
  if (run<0 + 1, 1, 4, 4>("nolds", 1.0, cus, ghz)) return 1;       // reference point: (almost) no operand traffic
  return 0;
}
