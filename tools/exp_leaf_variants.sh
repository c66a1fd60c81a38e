#!/bin/bash
# A/B of build_variants/leaf_*.so on tools/probe_leaf.py (one box): tools/exp_leaf_variants.sh [case=everyday:untrained] variants...
CASE=${1:-everyday:untrained}; shift
cp multi_part_assembly_amd/libmpa_hip.so /tmp/base.so
export PROBE_ONLY=$CASE
for r in 1 2; do
  for v in base "$@"; do
    if [ $v = base ]; then cp /tmp/base.so multi_part_assembly_amd/libmpa_hip.so; else cp build_variants/$v.so multi_part_assembly_amd/libmpa_hip.so; fi
    echo -n "$v: "; python tools/probe_leaf.py leaf 2>/dev/null | grep "+order"
  done
done
cp /tmp/base.so multi_part_assembly_amd/libmpa_hip.so
