#!/bin/bash
# Same-box A/B of the whole tree against round 4's (build_variants/r04_tree: `git archive bc307bf` + its own build):
# bench step times, alternating, for the configs given.   tools/exp_vs_round4.sh c2 c3 c5
R=$PWD
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; k=d.get('kernel_table') or {}
print(round(d['ms_per_step'],4), 'ms/step | roofline', round(r.get('avg_launch_ms') or 0,4), '| part-cd', round((k.get('assembly_part_chamfer') or {}).get('ms_per_step',0),4))"; }
for cfg in "$@"; do
  for rep in 1 2 3; do
    echo -n "$cfg r04 : "; (cd $R/build_variants/r04_tree && python bench.py --config $cfg --no-cpu-baseline --no-chamfer-standalone 2>/dev/null | line)
    echo -n "$cfg HEAD: "; (cd $R && python bench.py --config $cfg --no-cpu-baseline --no-chamfer-standalone 2>/dev/null | line)
  done
done
