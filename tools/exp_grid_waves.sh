#!/bin/bash
# A/B of MPA_GRID_WAVES variants (build_variants/gw*.so): fused search inside the c2 step and the stand-alone operator
R=$PWD
cp multi_part_assembly_amd/libmpa_hip.so /tmp/orig.so
for v in /tmp/orig.so $R/build_variants/gw*.so; do
  cp $v $R/multi_part_assembly_amd/libmpa_hip.so
  python bench.py --steps 40 --warmup 12 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$(basename $v)', round(d['ms_per_step'],3), 'search', round(d['roofline']['avg_launch_ms'],4), 'phase', round(d['roofline']['whole_phase_avg_ms'],4), [round(c['avg_call_ms'],3) for c in d['chamfer_standalone']['cases']])"
done
cp /tmp/orig.so $R/multi_part_assembly_amd/libmpa_hip.so
