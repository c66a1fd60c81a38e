"""A/B of the edge MLP's first layer on one box: two part-row GEMMs + broadcast sum (mpa_pair_layer_*) against the GEMM over
the materialised pair rows.   python tools/exp_pair_layer.py [c3|c5] [steps=60]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = """
import sys
sys.path.insert(0, %r)
sys.argv = ['bench.py', '--config', %r, '--steps', %r, '--warmup', '15', '--no-cpu-baseline', '--no-chamfer-standalone']
import multi_part_assembly_amd.gnn as g
g._PairMLP.PAIR_LAYER = %r
import bench
bench.main()
"""
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = sys.argv[2] if len(sys.argv) > 2 else "60"
for rep in range(2):
    for on in (True, False):
        out = subprocess.run([sys.executable, "-c", CODE % (ROOT, cfg, steps, on)], capture_output=True, text=True, cwd=ROOT)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(cfg, "pair layer" if on else "pair rows ", round(json.loads(line[-1])["ms_per_step"], 3) if line else out.stderr[-300:])
