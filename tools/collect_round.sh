#!/bin/bash
# Everything tools/gpu_round_evidence.sh left under gpurun_out/ -> profiles/<round>_*   (tools/collect_round.sh r04)
R=${1:?round prefix}
bash tools/collect_profiles.sh $R > /dev/null
cp gpurun_out/chamfer/pmc_FETCH_SIZE.txt profiles/${R}_chamfer_standalone_pmc_fetch_size.txt
cp gpurun_out/chamfer/pmc_WRITE_SIZE.txt profiles/${R}_chamfer_standalone_pmc_write_size.txt
cp gpurun_out/chamfer/kernel_stats.csv profiles/${R}_chamfer_standalone_rocprofv3_kernel_stats.csv
cp gpurun_out/chamfer/standalone.json profiles/${R}_chamfer_standalone.json
for c in c2 c3 c5; do cp gpurun_out/full_$c/bench_graph.json profiles/${R}_${c}_graph_bench_line.json; done
cp gpurun_out/full_c2/bench_driver_protocol.json profiles/${R}_c2_driver_protocol_bench_line.json
cp gpurun_out/full_c2/bench_self_check.json profiles/${R}_c2_self_check_line.json
python - <<PY
import json
for c in ["c2","c3","c5","c1","c2_bf16"]:
    d=json.loads(open(f"profiles/${R}_{c}_bench_line.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print(c, round(d["ms_per_step"],3),"ms", round(d["value"]), "parts/s; valid", d["config"]["valid_parts_per_batch_rank0"], "roofline", (round(r["avg_launch_ms"],4), round(r["achieved"],1), round(r["frac"],5), r.get("traffic")), "cpu", (round(d["cpu_baseline"]["value"],1), d["cpu_baseline"]["cores"]) if "cpu_baseline" in d else None)
for c in ["c2","c3","c5"]:
    d=json.loads(open(f"profiles/${R}_{c}_graph_bench_line.json").read().strip().splitlines()[-1]); print("graph",c, round(d["ms_per_step"],3))
d=json.loads(open("profiles/${R}_c2_driver_protocol_bench_line.json").read()); print("driver protocol c2", round(d["ms_per_step"],3), round(d["value"]))
d=json.loads(open("profiles/${R}_c2_self_check_line.json").read()); print("self check", d["self_check"], round(d["ms_per_step"],3))
d=json.loads(open("profiles/${R}_c2_bench_line.json").read().strip().splitlines()[-1])
for c in d["chamfer_standalone"]["cases"]: print(c["case"][:45], round(c["avg_call_ms"],3), round(c["GBps"],1), c.get("exhaustive_scan_ms"))
print({k:round(v["ms_per_step"],4) for k,v in d["kernel_table"].items()})
PY
