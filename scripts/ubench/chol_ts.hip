// Phase timestamps of the look-ahead Cholesky step (workgroup 0) on a 48-block-column system, the size of the
// 512-keyframe global BA.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=fast -DDH_CHOL_TS -I include -I droid-slam_amd/csrc \
//         scripts/ubench/chol_ts.hip droid-slam_amd/csrc/options.hip -o /tmp/chol_ts && /tmp/chol_ts
#include "../../droid-slam_amd/csrc/ba.hip"
#include <cstdio>
#include <vector>
#include <random>

int main(int argc, char** argv) {
  const int nbk = 48, npad = nbk * NB, ld = npad, rows = npad + NB;
  const int regp = argc > 1 ? atoi(argv[1]) : 1;
  std::vector<double> h((size_t)rows * ld);
  std::mt19937_64 rng(1);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  for (int i = 0; i < rows; ++i)
    for (int j = 0; j < ld; ++j) h[(size_t)i * ld + j] = (j <= i || i >= npad) ? 0.01 * U(rng) : 0.0;
  for (int i = 0; i < npad; ++i) h[(size_t)i * ld + i] = 10.0 + U(rng);
  double *H, *Ldiag; int* meta;
  hipMalloc(&H, sizeof(double) * h.size());
  hipMalloc(&Ldiag, sizeof(double) * nbk * NB * NB);
  hipMalloc(&meta, 64);
  hipMemset(meta, 0, 64);
  const size_t lds = sizeof(double) * 2 * NB * LDB;
  hipFuncSetAttribute((const void*)&chol_panel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  hipFuncSetAttribute((const void*)&chol_step_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  hipFuncSetAttribute((const void*)&chol_step_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  unsigned long long ts[128];
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(H, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(chol_panel_kernel, dim3(nbk), dim3(256), lds, 0, H, ld, 0, meta, Ldiag);
    for (int j = 0; j + 1 < nbk; ++j) {
      const int nP = nbk + 1 - j - 2;
      if (regp) hipLaunchKernelGGL(chol_step_kernel<true>, dim3(nP + nP * (nP + 1) / 2), dim3(256), lds + 2048, 0, H, ld, j, nbk, nP, meta, Ldiag);
      else hipLaunchKernelGGL(chol_step_kernel<false>, dim3(nP + nP * (nP + 1) / 2), dim3(256), lds, 0, H, ld, j, nbk, nP, meta, Ldiag);
      if (j == 10 || j == 40) {
        hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_chol_ts), sizeof(ts));
        if (rep == 2) {
          const double cyc = double(ts[4] - ts[0]), wall = double(ts[64 + 4] - ts[64 + 0]) * 10.0;   // ns (100 MHz)
          printf("step j=%d regp=%d: kernel body %.0f counter ticks, %.0f ns wall -> %.3f ticks/ns\n", j, regp, cyc, wall, cyc / wall);
          printf("  load+stage %.0f | 2 products %.0f | panel %.0f | store %.0f   (ns)\n", (ts[65] - ts[64]) * 10.0, (ts[66] - ts[65]) * 10.0,
                 (ts[67] - ts[66]) * 10.0, (ts[68] - ts[67]) * 10.0);
          printf("  ticks: load %llu products %llu panel %llu store %llu\n", ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3]);
          if (regp)
            for (int k = 0; k < 8; ++k)
              printf("   k=%d: lazy+pivot+subst %llu | write+barrier %llu | operands+urgent+slab %llu | to next barrier %llu\n", k,
                     ts[9 + 4 * k] - ts[8 + 4 * k], ts[10 + 4 * k] - ts[9 + 4 * k], ts[11 + 4 * k] - ts[10 + 4 * k],
                     k < 7 ? ts[12 + 4 * k] - ts[11 + 4 * k] : 0ull);
        }
      }
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int m[4]; hipMemcpy(m, meta, 16, hipMemcpyDeviceToHost);
    printf("rep %d: factorisation %.3f ms (%d launches), fail flag %d\n", rep, ms, nbk, m[1]);
  }
  return 0;
}
