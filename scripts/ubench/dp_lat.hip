// fp64 VALU latency / issue rate on gfx950 (one wave, inline asm so nothing moves): dependent chains and independent
// streams of v_fma_f64, v_mul_f64, v_rsq_f64, v_rcp_f64, fp32 fma; v_mfma_f64_16x16x4 dependent / independent.
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/dp_lat.hip -o scripts/ubench/dp_lat && scripts/ubench/dp_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#define TIC(v) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 7\n s_nop 7\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
__global__ void k(double* out, unsigned long long* t, double a0) {
  double a = a0 + threadIdx.x * 1e-9, b = 1.0000001, c = 1e-9;
  double x1 = a + 1, x2 = a + 2, x3 = a + 3;
  unsigned long long t0, t1;
  TIC(t0);
  asm volatile(".rept 64\n v_fma_f64 %0, %0, %1, %2\n .endr" : "+v"(a) : "v"(b), "v"(c));
  TIC(t1); if (threadIdx.x == 0) t[0] = t1 - t0;
  TIC(t0);
  asm volatile(".rept 16\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n .endr"
               : "+v"(a), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(c));
  TIC(t1); if (threadIdx.x == 0) t[1] = t1 - t0;
  TIC(t0);
  asm volatile(".rept 64\n v_mul_f64 %0, %0, %1\n .endr" : "+v"(a) : "v"(b));
  TIC(t1); if (threadIdx.x == 0) t[2] = t1 - t0;
  TIC(t0);
  asm volatile(".rept 64\n v_rsq_f64 %0, %0\n .endr" : "+v"(a));
  TIC(t1); if (threadIdx.x == 0) t[3] = t1 - t0;
  TIC(t0);
  asm volatile(".rept 64\n v_rcp_f64 %0, %0\n .endr" : "+v"(a));
  TIC(t1); if (threadIdx.x == 0) t[4] = t1 - t0;
  TIC(t0);
  asm volatile(".rept 16\n v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n .endr" : "+v"(a), "+v"(x1), "+v"(x2), "+v"(x3));
  TIC(t1); if (threadIdx.x == 0) t[5] = t1 - t0;
  float f = (float)a;
  TIC(t0);
  asm volatile(".rept 64\n v_fma_f32 %0, %0, %0, %0\n .endr" : "+v"(f));
  TIC(t1); if (threadIdx.x == 0) t[6] = t1 - t0;
  TIC(t0);
  asm volatile(".rept 32\n v_fma_f64 %0, %0, %2, %3\n v_mul_f64 %1, %0, %2\n .endr" : "+v"(a), "+v"(x1) : "v"(b), "v"(c));   // fma -> mul -> (next fma independent of mul)
  TIC(t1); if (threadIdx.x == 0) t[7] = t1 - t0;
  TIC(t0);
  asm volatile(".rept 32\n v_fma_f64 %0, %0, %1, %2\n v_mul_f64 %0, %0, %1\n .endr" : "+v"(a) : "v"(b), "v"(c));            // fma -> mul dependent
  TIC(t1); if (threadIdx.x == 0) t[8] = t1 - t0;
  typedef double f64x4 __attribute__((ext_vector_type(4)));
  f64x4 acc = {a, x1, x2, x3};
  TIC(t0);
  asm volatile(".rept 64\n v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n .endr\n s_nop 15\n s_nop 15" : "+v"(acc) : "v"(b), "v"(c));
  TIC(t1); if (threadIdx.x == 0) t[9] = t1 - t0;
  out[threadIdx.x] = a + x1 + x2 + x3 + f + acc[0] + acc[1] + acc[2] + acc[3];
}
int main() {
  double* out; unsigned long long* t;
  (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&t, 16 * 8);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, t, 1.5);
  unsigned long long h[16]; (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"dependent v_fma_f64 (x64)", "4 independent v_fma_f64 streams (x64 ops)", "dependent v_mul_f64 (x64)", "dependent v_rsq_f64 (x64)",
                         "dependent v_rcp_f64 (x64)", "4 independent v_rsq_f64 streams (x64 ops)", "dependent v_fma_f32 (x64)", "fma->(mul off chain) x32 (64 ops)", "fma->mul dependent x32 (64 ops)", "dependent v_mfma_f64_16x16x4_f64 (x64)"};
  for (int i = 0; i < 10; ++i) printf("%-48s %6.1f ticks per op\n", names[i], h[i] / 64.0);
  return 0;
}
