#!/bin/bash
# Is the c2 training trajectory reproducible run to run?  N bench runs per library (the shipped one and build_variants/*.so)
N=${1:-12}
R=$PWD
cp multi_part_assembly_amd/libmpa_hip.so /tmp/orig.so
for v in /tmp/orig.so $R/build_variants/*.so; do
  cp $v $R/multi_part_assembly_amd/libmpa_hip.so
  echo "== $(basename $v)"
  for i in $(seq $N); do python bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-chamfer-standalone 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['final_loss'])"; done | sort | uniq -c
done
cp /tmp/orig.so $R/multi_part_assembly_amd/libmpa_hip.so
