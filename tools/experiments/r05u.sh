mkdir -p gpurun_out/r05u
bash tools/pmc_step.sh 4 r05 > gpurun_out/r05u/pmc.log 2>&1
cp gpurun_out/pmc_step/summary.txt gpurun_out/r05u/step_pmc_counters.md
cp gpurun_out/pmc_step/step_hbm_traffic.json gpurun_out/r05u/
rm -rf gpurun_out/pmc_step/*/
head -32 gpurun_out/r05u/step_pmc_counters.md | cut -c1-230
