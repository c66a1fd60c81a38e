R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o tr -- python $R/bench.py --no-cpu-baseline --steps 6 --warmup 3 > /dev/null 2>&1
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_steps.py $f --last 3 --top 5 --gaps 60 > $R/gpurun_out/gaps.txt
