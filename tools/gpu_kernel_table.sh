#!/bin/bash
# On the GPU box: per-kernel steady-state table of one bench workload -> gpurun_out/<outdir>/steady_state_per_step.txt
#   tools/gpu_kernel_table.sh [config=c2] [outdir=ktable] [extra bench flags...]
CFG=${1:-c2}; OUT=$PWD/gpurun_out/${2:-ktable}; shift 2
R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o tr -- python $R/bench.py --config $CFG --no-cpu-baseline --no-chamfer-standalone --steps 12 --warmup 8 "$@" > $OUT/bench_under_rocprof.json 2>/dev/null
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_steps.py $f --last 4 --top 60 > $OUT/steady_state_per_step.txt
head -${TOPN:-32} $OUT/steady_state_per_step.txt | cut -c1-170
