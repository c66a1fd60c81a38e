#!/bin/bash
# dev tool: dg_agg_bwd_kernel variants on one box (tools/build_variant.sh agg_* dgcnn_enc.hip ...): phase cycles of the
# instrumented build, then the kernel's average time per variant under rocprofv3.  PARTS=353|514
R=$PWD
cp multi_part_assembly_amd/libmpa_hip.so /tmp/orig.so
cp build_variants/agg_stats.so multi_part_assembly_amd/libmpa_hip.so
python tools/probe_agg_stats.py
cd /tmp && export TMPDIR=/tmp
for v in /tmp/orig.so $R/build_variants/agg_[a-rt-z]*.so; do
  cp $v $R/multi_part_assembly_amd/libmpa_hip.so
  rm -rf /tmp/prof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o st -- python $R/tools/probe_agg_stats.py > /dev/null 2>&1
  echo "== $(basename $v)"; python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dg_agg_" in r["Name"]: print("   ", r["Name"][:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
cp /tmp/orig.so $R/multi_part_assembly_amd/libmpa_hip.so
