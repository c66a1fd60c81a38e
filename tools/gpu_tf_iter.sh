#!/bin/bash
# dev loop on the GPU box: model tests, bench line, per-step kernel trace -> gpurun_out/
R=$PWD
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5 > gpurun_out/pytest_tf.txt
python bench.py --no-cpu-baseline > gpurun_out/bench_tf.json 2>gpurun_out/bench_tf.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o tr -- python $R/bench.py --no-cpu-baseline --steps 6 --warmup 3 > /dev/null 2>&1
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_steps.py $f --last 3 --top 70 --seq "${SEQ:-wgrad_kernel}" > $R/gpurun_out/trace_tf.txt
