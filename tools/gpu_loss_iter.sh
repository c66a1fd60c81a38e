#!/bin/bash
# dev loop on the GPU box: loss + chamfer tests, bench line (eager and graph), kernel trace
R=$PWD
timeout 900 python -m pytest tests/test_loss_gpu.py tests/test_chamfer_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5 > gpurun_out/pytest_loss.txt
python bench.py --no-cpu-baseline > gpurun_out/bench_tf.json 2>gpurun_out/bench_tf.err
python bench.py --no-cpu-baseline --graph > gpurun_out/bench_graph.json 2>gpurun_out/bench_graph.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o tr -- python $R/bench.py --no-cpu-baseline --steps 6 --warmup 3 > /dev/null 2>&1
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_steps.py $f --last 3 --top 70 > $R/gpurun_out/trace_tf.txt
