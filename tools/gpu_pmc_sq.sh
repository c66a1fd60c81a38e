#!/bin/bash
# SQ counter pass of the bench command (dev tool) -> gpurun_out/<outdir>/pmc_sq.txt
#   tools/gpu_pmc_sq.sh [config=c2] [outdir=.]
CFG=${1:-c2}
R=$PWD
O=$R/gpurun_out/${2:-.}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d /tmp/pmc -o pm -- python $R/bench.py --config $CFG --no-cpu-baseline --steps 3 --warmup 2 > /dev/null 2>&1
python $R/tools/pmc_summary.py $(find /tmp/pmc -name "*counter_collection.csv" | head -1) --top 400 > $O/pmc_sq.txt
