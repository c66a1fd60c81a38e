#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of the stand-alone Chamfer operator -> gpurun_out/chamfer/
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/chamfer
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/tools/chamfer_standalone.py 20 > $OUT/standalone.json 2> $OUT/standalone.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o cs -- python $GRAFT_REPO_ROOT/tools/chamfer_standalone.py 20 > $OUT/prof_run.json 2> $OUT/prof.err
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -12 $OUT/kernel_stats.csv | cut -c1-200
