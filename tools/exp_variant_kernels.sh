#!/bin/bash
# A/B of library variants by per-kernel time: tools/exp_variant_kernels.sh <grep pattern> [config=c2]
PAT=${1:-pn_}; CFG=${2:-c2}
R=$PWD
cp multi_part_assembly_amd/libmpa_hip.so /tmp/orig.so
for v in /tmp/orig.so $R/build_variants/*.so; do
  cp $v $R/multi_part_assembly_amd/libmpa_hip.so
  echo "== $(basename $v)"
  TOPN=80 bash tools/gpu_kernel_table.sh $CFG ktable_v 2>/dev/null | grep -E "$PAT" | cut -c1-150
done
cp /tmp/orig.so $R/multi_part_assembly_amd/libmpa_hip.so
