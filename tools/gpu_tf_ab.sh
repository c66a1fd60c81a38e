#!/bin/bash
# dev loop for the transformer: its GPU tests, then a traced bench run -> per-step kernel table rows matching $1
R=$PWD
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "transformer or step or graph or trajectory" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o tr -- python $R/bench.py --no-cpu-baseline --steps 6 --warmup 3 > /tmp/bench.out 2>&1
python -c "import json; d=json.loads(open('/tmp/bench.out').read().strip().splitlines()[-1]); print('ms/step under rocprof', round(d['ms_per_step'],3))"
python $R/tools/trace_steps.py $(find /tmp/prof -name "*kernel_trace.csv" | head -1) --last 3 --top 80 | grep -E "${1:-attn|gemm|ln_|wgrad}"
