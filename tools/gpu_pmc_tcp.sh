#!/bin/bash
# L1 (TCP) / texture-addresser counter passes of the bench command (dev tool) -> gpurun_out/<outdir>/pmc_tcp.txt
#   tools/gpu_pmc_tcp.sh [config=c2] [outdir=.]      (one counter group per pass: see the guide's PMC section)
CFG=${1:-c2}
R=$PWD
O=$R/gpurun_out/${2:-.}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/pmc_tcp.txt
PGROUPS=${PMC_GROUPS:-"TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum|TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum|TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum|GRBM_GUI_ACTIVE TA_TA_BUSY_sum"}
IFS="|"; for grp in $PGROUPS; do IFS=" "
  rm -rf /tmp/pmc
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc -o pm -- python $R/bench.py --config $CFG --no-cpu-baseline --no-chamfer-standalone --steps 3 --warmup 2 > /tmp/pmc_log.txt 2>&1
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_summary.py $f --top 400 | grep -E "grid_search|assembly_nn|pn_fwd_split|leaf_search" >> $O/pmc_tcp.txt; else echo "no output for: $grp ($(grep -i -m1 "error\|invalid\|not" /tmp/pmc_log.txt))" >> $O/pmc_tcp.txt; fi
done
cut -c1-110 $O/pmc_tcp.txt
