#!/bin/bash
# dev experiment loop: for every build_variants/*.so, swap it in as libmpa_hip.so, trace a short bench run and
# print the kernels matching $1 (grep -E pattern)  -> gpurun_out/variants.txt
R=$PWD
PAT=${1:-pn_}
cp multi_part_assembly_amd/libmpa_hip.so /tmp/orig.so
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/variants.txt
for v in $R/build_variants/*.so; do
  cp $v $R/multi_part_assembly_amd/libmpa_hip.so
  rm -rf /tmp/prof
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o tr -- python $R/bench.py --no-cpu-baseline --steps 6 --warmup 3 > /tmp/bench.out 2>&1
  f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
  echo "== $(basename $v)  $(python -c "import json,sys; d=json.loads(open('/tmp/bench.out').read().strip().splitlines()[-1]); print('ms/step', round(d['ms_per_step'],3))" 2>/dev/null)" >> $R/gpurun_out/variants.txt
  python $R/tools/trace_steps.py $f --last 3 --top 70 | grep -E "$PAT" >> $R/gpurun_out/variants.txt
done
cp /tmp/orig.so $R/multi_part_assembly_amd/libmpa_hip.so
