mkdir -p gpurun_out/r6y
F="-O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=fast -fno-slp-vectorize -DDH_PYR_TS=300 -I include -I droid-slam_amd/csrc scripts/ubench/pyr_ts.hip droid-slam_amd/csrc/options.hip"
hipcc $F -DDH_PYR_TS_WAVE=5 -o /tmp/pyr_ts5 2>/dev/null && /tmp/pyr_ts5 0 > gpurun_out/r6y/ts_single.txt 2>&1
hipcc $F -DDH_PYR_TS_WAVE=13 -o /tmp/pyr_ts13 2>/dev/null && /tmp/pyr_ts13 1 > gpurun_out/r6y/ts_dual.txt 2>&1
cat gpurun_out/r6y/ts_single.txt gpurun_out/r6y/ts_dual.txt
