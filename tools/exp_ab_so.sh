#!/bin/bash
# A/B of whole-library builds by the bench's step time on ONE box, alternating three times (boxes differ by more than most
# single changes): the shipped libmpa_hip.so against every build_variants/*.so.   tools/exp_ab_so.sh [config=c2]
CFG=${1:-c2}
R=$PWD
cp multi_part_assembly_amd/libmpa_hip.so /tmp/new.so
for r in 1 2 3; do
  for v in /tmp/new.so $R/build_variants/*.so; do
    cp $v multi_part_assembly_amd/libmpa_hip.so
    echo -n "$(basename $v) "
    python bench.py --config $CFG --no-cpu-baseline --no-chamfer-standalone 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))"
  done
done
cp /tmp/new.so multi_part_assembly_amd/libmpa_hip.so
