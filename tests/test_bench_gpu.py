"""The bench contract (task statement §4, bench.py's docstring) checked on a short run: ONE JSON line on stdout with the
mandated keys, the `roofline` object of the dominant kernel measured with HIP events over exactly the K timed steps, the
bounded CPU baseline — and the KernelTimer scoping that keeps every other event out of the timed region."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # exactly one line on stdout
    return json.loads(lines[0])


def test_bench_line_has_the_contract_keys(cuda_device):
    d = _run("--steps", "3", "--warmup", "1", "--cpu-batch", "1")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 640.0 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]  # B x P parts per step
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["launches"] == 3  # the dominant kernel was timed in the timed region: once per timed step
    assert 0.0 < r["avg_launch_ms"] < d["ms_per_step"]
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


def test_bench_variants_are_labelled(cuda_device):
    d = _run("--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--dtype", "bf16")
    assert d["dtype"] == "bf16" and "precision_note" in d["config"] and "cpu_baseline" not in d
    d = _run("--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--config", "dgl_dgcnn")
    assert d["config"]["name"] == "c3" and "knn" in d["roofline"]["kernel"]
    assert d["roofline"]["launches"] == 2


def test_kernel_timer_scope(cuda_device):
    from multi_part_assembly_amd import _lib

    t = _lib.KernelTimer(only=("grid_search_kernel",))
    _lib.KernelTimer.active = t
    try:
        assert _lib.KernelTimer.start("pointnet_forward[1x2x3]") is None
        assert _lib.KernelTimer.phase_events(["assembly_pose[x]", "assembly_finalize[x]"]) is None
        evs = _lib.KernelTimer.phase_events(["grid_search_kernel[x]"])
        assert evs is not None and len(evs) == 2 and all(e is not None for e in evs)
        arr = _lib.KernelTimer.handles([None] * 5 + evs)
        assert len(arr) == 7 and arr[0] is None and arr[5] is not None
        _lib.KernelTimer.add_phases(["grid_search_kernel[x]"], evs)
        torch.cuda.synchronize()
        assert list(t.summary()) == ["grid_search_kernel[x]"]
    finally:
        _lib.KernelTimer.active = None
