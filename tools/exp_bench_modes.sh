#!/bin/bash
# step time of the bench configs under the search modes of the fused loss (one box): tools/exp_bench_modes.sh c2 c3 c5
for cfg in "$@"; do
  for mode in grid leaf auto grid leaf auto; do
    echo -n "$cfg $mode: "
    MPA_SHAPE_SEARCH=$mode python bench.py --config $cfg --no-cpu-baseline --no-chamfer-standalone 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print(round(d['ms_per_step'],4), 'ms/step; roofline kernel', r.get('avg_launch_ms'), 'ms frac', r.get('frac'))"
  done
done
