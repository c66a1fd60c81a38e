#!/bin/bash
# Build a variant of the library for an A/B run on the GPU box: recompile ONE source with extra flags and link it
# with the regular objects -> build_variants/<name>.so (git-ignored; tools/variant_bench.sh swaps them in).
#   tools/build_variant.sh <name> <source.hip> [extra hipcc flags...]
set -e
NAME=$1; SRC=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd)
C=$R/multi_part_assembly_amd/csrc
mkdir -p $R/build_variants /tmp/variant_$NAME
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w $(grep -m1 "^// hipcc-flags:" $C/$SRC | cut -d: -f2-) "$@" -c $C/$SRC -o /tmp/variant_$NAME/${SRC%.hip}.o
OBJS=$(ls $C/build/*.o | grep -v "/${SRC%.hip}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_variants/$NAME.so $OBJS /tmp/variant_$NAME/${SRC%.hip}.o
echo built build_variants/$NAME.so
