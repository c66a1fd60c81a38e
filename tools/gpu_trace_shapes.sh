#!/bin/bash
# On the GPU box: per-launch-shape durations of the named kernels of one bench workload -> gpurun_out/<outdir>/shapes.txt
#   tools/gpu_trace_shapes.sh <config> <outdir> <name substring> [...]
CFG=$1; OUT=$PWD/gpurun_out/$2; shift 2
R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o tr -- python $R/bench.py --config $CFG --no-cpu-baseline --no-chamfer-standalone --steps 8 --warmup 8 > $OUT/bench_under_rocprof.json 2>/dev/null
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_by_shape.py $f "$@" > $OUT/shapes.txt
head -70 $OUT/shapes.txt | cut -c1-200
