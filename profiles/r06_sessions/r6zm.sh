bash scripts/pmc_pyr_build.sh gpurun_out/r6zm/pmc_dual 512 "plain order" > gpurun_out/r6zm_pmc_dual.txt 2>&1
tail -42 gpurun_out/r6zm_pmc_dual.txt
