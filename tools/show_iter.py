import json, sys
print(open("gpurun_out/pytest_tf.txt").read().strip())
d = json.loads(open("gpurun_out/bench_tf.json").read().strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"], 3), "parts/s", round(d["value"]), "loss", d["final_loss"])
for k, v in d["kernels"].items():
    print("   ", k, round(v["avg_ms"], 4), v["launches"])
pat = sys.argv[1:] or ["steps="]
for line in open("gpurun_out/trace_tf.txt"):
    if any(p in line for p in pat):
        print(line.rstrip()[:150])
