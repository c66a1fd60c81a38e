#!/bin/bash
# tools/variant_trace.sh for another workload: tools/variant_trace_cfg.sh <config> <kernel pattern>
R=$PWD
CFG=$1; PAT=${2:-knn}
cp multi_part_assembly_amd/libmpa_hip.so /tmp/orig.so
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/variants.txt
for v in $R/build_variants/*.so; do
  cp $v $R/multi_part_assembly_amd/libmpa_hip.so
  rm -rf /tmp/prof
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o tr -- python $R/bench.py --config $CFG --no-cpu-baseline --no-chamfer-standalone --steps 5 --warmup 3 > /tmp/bench.out 2>&1
  f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
  echo "== $(basename $v)" >> $R/gpurun_out/variants.txt
  python $R/tools/trace_steps.py $f --last 3 --top 70 | grep -E "$PAT|steps=" >> $R/gpurun_out/variants.txt
done
cp /tmp/orig.so $R/multi_part_assembly_amd/libmpa_hip.so
