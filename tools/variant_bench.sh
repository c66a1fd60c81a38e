#!/bin/bash
# A/B on the GPU box: for every build_variants/*.so, swap it in as libmpa_hip.so and print the bench's step time and
# the timers matching $2 (grep -E pattern) for config $1  -> gpurun_out/variant_bench.txt
CFG=${1:-c3}; PAT=${2:-dgcnn}
R=$PWD
cp multi_part_assembly_amd/libmpa_hip.so /tmp/orig.so
: > $R/gpurun_out/variant_bench.txt
for v in /tmp/orig.so $R/build_variants/*.so; do
  cp $v $R/multi_part_assembly_amd/libmpa_hip.so
  echo "== $(basename $v)" >> $R/gpurun_out/variant_bench.txt
  python bench.py --config $CFG --no-cpu-baseline 2>/dev/null | python -c "
import json, sys, re
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  ms/step', round(d['ms_per_step'], 3))
for k, v in d['kernels'].items():
    if re.search(r'$PAT', k): print('  ', k, round(v['avg_ms'], 3))" >> $R/gpurun_out/variant_bench.txt
done
cp /tmp/orig.so $R/multi_part_assembly_amd/libmpa_hip.so
cat $R/gpurun_out/variant_bench.txt
