#!/bin/bash
# per-step kernel table of one bench config under rocprofv3 (dev tool): tools/gpu_steps_table.sh c2 [top=45]
CFG=${1:-c2}; TOP=${2:-45}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o st -- python $R/bench.py --config $CFG --no-cpu-baseline --no-chamfer-standalone > /dev/null 2>&1
python $R/tools/trace_steps.py $(find /tmp/prof -name "*kernel_trace.csv" | head -1) --last 5 --top $TOP | cut -c1-150
