#!/bin/bash
# every build_variants/*.so against each other on c3 (or $1), two alternations on one box, by the bench's step time
CFG=${1:-c3}
cp multi_part_assembly_amd/libmpa_hip.so /tmp/o.so
for rep in 1 2; do
  for v in build_variants/*.so; do
    cp $v multi_part_assembly_amd/libmpa_hip.so
    python bench.py --config $CFG --no-cpu-baseline --no-chamfer-standalone --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$(basename $v)', round(d['ms_per_step'],3))"
  done
done
cp /tmp/o.so multi_part_assembly_amd/libmpa_hip.so
