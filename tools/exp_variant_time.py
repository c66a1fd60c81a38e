#!/usr/bin/env python3
"""A/B of library variants on the stand-alone Chamfer call: python tools/exp_variant_time.py [variant.so ...]"""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CODE = r'''
import sys
sys.path.insert(0, "%s")
import torch
from multi_part_assembly_amd import _lib
if "%s":
    from pathlib import Path
    _lib.LIB_PATH = Path("%s")
import bench
d = bench.chamfer_standalone(torch.device("cuda", 0), reps=30)
print("%s", " | ".join(f"{c['avg_call_ms']:.3f} ms" for c in d["cases"]))
'''
for v in [""] + sys.argv[1:]:
    path = str(ROOT / "build_variants" / v) if v else ""
    subprocess.run([sys.executable, "-c", CODE % (ROOT, v, path, v or "libmpa_hip.so")], check=True)
