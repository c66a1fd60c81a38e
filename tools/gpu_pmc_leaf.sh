#!/bin/bash
# SQ counter pass of the fused loss forward under the leaf search (dev tool) -> gpurun_out/pmc_leaf.txt
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d /tmp/pmc -o pm -- python $R/tools/probe_leaf.py leaf > /dev/null 2>&1
python $R/tools/pmc_summary.py $(find /tmp/pmc -name "*counter_collection.csv" | head -1) --top 400 | grep -i "leaf" > $R/gpurun_out/pmc_leaf.txt
rm -rf /tmp/pmc
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR --output-format csv -d /tmp/pmc -o pm -- python $R/tools/probe_leaf.py leaf > /dev/null 2>&1
python $R/tools/pmc_summary.py $(find /tmp/pmc -name "*counter_collection.csv" | head -1) --top 400 | grep -i "leaf" >> $R/gpurun_out/pmc_leaf.txt
