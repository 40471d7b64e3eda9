mkdir -p gpurun_out/r6v
python scripts/bench_pyr_build.py 256 5 64 > gpurun_out/r6v/build_ab_64frames.txt 2>&1
python scripts/bench_pyr_build.py 256 5 512 "8 waves" > gpurun_out/r6v/build_ab_512frames.txt 2>&1
tail -8 gpurun_out/r6v/build_ab_64frames.txt; tail -6 gpurun_out/r6v/build_ab_512frames.txt
bash scripts/pmc_pyr_build.sh gpurun_out/r6v/pmc_plain 512 "row-pair-major, 8 waves (rounds" > gpurun_out/r6v/pmc_plain.txt 2>&1
bash scripts/pmc_pyr_build.sh gpurun_out/r6v/pmc_xcd 512 "row-pair-major, 8 waves, an edge" > gpurun_out/r6v/pmc_xcd.txt 2>&1
tail -45 gpurun_out/r6v/pmc_plain.txt
