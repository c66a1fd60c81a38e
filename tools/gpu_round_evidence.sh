#!/bin/bash
# One gpurun call that collects the round's evidence (tools/collect_profiles.sh copies it into profiles/):
#   bench line + rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of c2, c3, c5, c1 and the bf16 variant,
#   the SQ counter passes of c2 / c3 / c5, the HIP-graph lines, the stand-alone Chamfer operator profile.
R=$PWD
for c in c2 c3 c5 c1; do bash tools/gpu_full_pass.sh $c full_$c 0; cd $R; done
mkdir -p gpurun_out/full_c2_bf16
python bench.py --dtype bf16 --no-chamfer-standalone > gpurun_out/full_c2_bf16/bench.json 2> gpurun_out/full_c2_bf16/bench.err
for c in c2 c3 c5; do
  python bench.py --config $c --graph --no-cpu-baseline --no-chamfer-standalone > gpurun_out/full_$c/bench_graph.json 2>/dev/null
  bash tools/gpu_pmc_sq.sh $c full_$c; cd $R
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-chamfer-standalone > gpurun_out/full_c2/bench_driver_protocol.json 2>/dev/null
python bench.py --self-check 300 --no-cpu-baseline --no-chamfer-standalone > gpurun_out/full_c2/bench_self_check.json 2>/dev/null
bash tools/gpu_chamfer_profile.sh > /dev/null; cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc -o pm -- python $R/tools/chamfer_standalone.py 4 > /dev/null 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pmc -name "*counter_collection.csv" | head -1) --top 40 > $R/gpurun_out/chamfer/pmc_$c.txt
  cd $R
done
ls gpurun_out/full_c2 gpurun_out/full_c3 gpurun_out/full_c5 gpurun_out/full_c1 gpurun_out/chamfer
