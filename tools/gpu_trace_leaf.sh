#!/bin/bash
# kernel-trace stats of one case of tools/probe_leaf.py (dev tool): tools/gpu_trace_leaf.sh everyday:untrained [modes]
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr
PROBE_ONLY=$1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o tr -- python $R/tools/probe_leaf.py ${2:-leaf} > /tmp/tr_out.txt 2>&1
tail -4 /tmp/tr_out.txt
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/tr/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
