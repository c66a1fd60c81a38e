#!/bin/bash
# Round-end evidence pass on the GPU box: [full GPU test suite,] the bench line of one workload, rocprofv3 kernel stats
# and the FETCH_SIZE / WRITE_SIZE counter passes of the same bench command -> gpurun_out/<outdir>
#   tools/gpu_full_pass.sh [config=c2] [outdir=full] [tests=1]
CFG=${1:-c2}
R=$PWD
O=$R/gpurun_out/${2:-full}
mkdir -p $O
if [ "${3:-1}" = "1" ]; then
  timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5 > $O/pytest_gpu.txt
fi
python bench.py --config $CFG > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o st -- python $R/bench.py --config $CFG --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python $R/tools/trace_steps.py $(find /tmp/prof -name "*kernel_trace.csv" | head -1) --last 5 --top 80 > $O/steady_state_per_step.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc
  timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc -o pm -- python $R/bench.py --config $CFG --no-cpu-baseline --steps 4 --warmup 2 > /dev/null 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pmc -name "*counter_collection.csv" | head -1) --top 400 > $O/pmc_$c.txt
done
