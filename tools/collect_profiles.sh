#!/bin/bash
# Copy the outputs of tools/gpu_full_pass.sh / gpu_pmc_sq.sh (gpurun_out/full_<cfg>/) into profiles/ under the round's prefix
# and rebuild the traffic records bench.py reads.   tools/collect_profiles.sh r03
R=${1:?round prefix, e.g. r03}
for c in c1 c2 c3 c5 c2_bf16; do
  d=gpurun_out/full_$c
  [ -d $d ] || continue
  [ -s $d/bench.json ] && cp $d/bench.json profiles/${R}_${c}_bench_line.json
  [ -s $d/kernel_stats.csv ] && cp $d/kernel_stats.csv profiles/${R}_${c}_rocprofv3_kernel_stats.csv
  [ -s $d/steady_state_per_step.txt ] && cp $d/steady_state_per_step.txt profiles/${R}_${c}_steady_state_per_step.txt
  [ -s $d/pmc_FETCH_SIZE.txt ] && cp $d/pmc_FETCH_SIZE.txt profiles/${R}_${c}_pmc_fetch_size.txt
  [ -s $d/pmc_WRITE_SIZE.txt ] && cp $d/pmc_WRITE_SIZE.txt profiles/${R}_${c}_pmc_write_size.txt
  [ -s $d/pmc_sq.txt ] && cp $d/pmc_sq.txt profiles/${R}_${c}_pmc_sq_counters.txt
done
python tools/pmc_json.py profiles $R
