// Phase timestamps of the pyramid build's row-ring loop (one workgroup in the middle of a 512-edge launch, level 0, steps 16-31).
// Build + run on the GPU box (DH_PYR_TS = the edge whose workgroup 5 is stamped, DH_PYR_TS_WAVE = the second stamped wave):
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=fast -fno-slp-vectorize -DDH_PYR_TS=300 -DDH_PYR_TS_WAVE=5 -I include \
//         -I droid-slam_amd/csrc scripts/ubench/pyr_ts.hip droid-slam_amd/csrc/options.hip -o /tmp/pyr_ts && /tmp/pyr_ts [dual=0|1]
// Stamps per step: 0 step start, 1 staged row written to LDS, 2 past barrier 1, 3 next row's fetch issued, 4 B reads + MFMAs issued,
// 5 scatter issued, 6 past barrier 2, 7 finished row read out and its store issued.
#include "../../droid-slam_amd/csrc/corr_pyramid.hip"
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>

int main(int argc, char** argv) {
  const int dual = argc > 1 ? atoi(argv[1]) : 0;
  const int F = 64, E = 512, h = 48, w = 64;
  dh_set_option("pyr_build_dual", dual);
  dh_set_option("pyr_build_xcd", 0);
  std::vector<__half> fm((size_t)F * 128 * h * w);
  std::mt19937 rng(1); std::normal_distribution<float> N(0.f, 1.f);
  for (auto& v : fm) v = __float2half(N(rng));
  std::vector<int64_t> i1(E), i2(E);
  for (int e = 0; e < E; ++e) { i1[e] = rng() % F; i2[e] = rng() % F; }
  void *dfm, *prep, *pyr; int64_t *d1, *d2;
  hipMalloc(&dfm, fm.size() * 2); hipMemcpy(dfm, fm.data(), fm.size() * 2, hipMemcpyHostToDevice);
  hipMalloc(&prep, dh_corr_pyramid_prepared_bytes(F, h, w));
  hipMalloc(&pyr, dh_corr_pyramid_bytes(E, h, w));
  hipMalloc(&d1, E * 8); hipMalloc(&d2, E * 8);
  hipMemcpy(d1, i1.data(), E * 8, hipMemcpyHostToDevice); hipMemcpy(d2, i2.data(), E * 8, hipMemcpyHostToDevice);
  if (dh_corr_pyramid_prepare_frames(dfm, prep, F, 128, h, w, h, w, nullptr) != 0) { printf("prepare failed\n"); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  unsigned long long ts[2 * 16 * 8];
  for (int rep = 0; rep < 3; ++rep) {
    hipDeviceSynchronize();
    hipEventRecord(e0);
    if (dh_corr_pyramid_build_indexed(prep, d1, d2, pyr, F, E, h, w, nullptr) != 0) { printf("build failed\n"); return 1; }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("rep %d: %.3f ms per %d edges (dual=%d, with the stamps)\n", rep, ms, E, dual);
  }
  hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_pyr_ts), sizeof(ts));
  const char* names[8] = {"start", "staged", "barrier1", "fetch", "mfma", "scatter", "barrier2", "readout"};
  for (int wv = 0; wv < 2; ++wv) {
    printf("wave %d: cycles between stamps (s_memtime), level-0 steps 16..31\n  step", wv ? DH_PYR_TS_WAVE : 0);
    for (int i = 1; i < 8; ++i) printf(" %9s", names[i]);
    printf("   total  gap-to-next\n");
    double sum[9] = {0};
    for (int k = 0; k < 16; ++k) {
      const unsigned long long* t = ts + (wv * 16 + k) * 8;
      printf("  %4d", 16 + k);
      for (int i = 1; i < 8; ++i) { printf(" %9lld", (long long)(t[i] - t[i - 1])); sum[i] += (double)(t[i] - t[i - 1]); }
      printf(" %7lld", (long long)(t[7] - t[0])); sum[0] += (double)(t[7] - t[0]);
      if (k < 15) { printf(" %7lld", (long long)(ts[(wv * 16 + k + 1) * 8] - t[7])); sum[8] += (double)(ts[(wv * 16 + k + 1) * 8] - t[7]); }
      printf("\n");
    }
    printf("  mean");
    for (int i = 1; i < 8; ++i) printf(" %9.0f", sum[i] / 16);
    printf(" %7.0f %7.0f\n", sum[0] / 16, sum[8] / 15);
  }
  return 0;
}
