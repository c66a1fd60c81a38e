"""The bench contract (task statement §4, bench.py's docstring) checked on a short run: ONE JSON line on stdout with the
mandated keys, the `roofline` object of the dominant kernel measured with HIP events over exactly the K timed steps, the
bounded CPU baseline — and the KernelTimer scoping that keeps every other event out of the timed region.

Collected LAST (file name + the hook in conftest.py) so that no bench run can stop `pytest -x` before a parity test, and
every wall-clock figure here is RECORDED (printed into the run's log), never asserted: a stopwatch must not be able to
void a correctness run (round-3 verdict, item 1)."""
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
    assert len(d["config"]["valid_parts_per_batch_rank0"]) == 4  # distinct batches rotated through the timed loop
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
    # every row of the per-kernel table names the peak its fraction refers to, and no fraction exceeds 1 (round-5 verdict: a
    # fraction above 1 means the denominator is not what the kernel does)
    kt = d["kernel_table"]
    assert {"pointnet_forward", "pointnet_backward", "transformer_forward", "assembly_part_chamfer"} <= set(kt)
    for name, row in kt.items():
        assert "peak" in row and 0.0 < row["frac"] <= 1.0, (name, row)
    assert kt["assembly_part_chamfer"]["exhaustive_equivalent"]["speedup_over_valu_peak_scan"] > 0
    # the drop-in operator on its own: SURVEY.md 8(d)'s two standalone shapes through the C ABI
    cs = d["chamfer_standalone"]["cases"]
    assert len(cs) == 3 and all(x["GBps"] > 0 and x["calls"] == 20 for x in cs)
    assert cs[0]["algorithmic_bytes_per_call"] == 24.0 * 640 * 2000 and cs[1]["algorithmic_bytes_per_call"] == 24.0 * 32 * 40000
    # the per-part call goes to the matrix-core gated search (gate_nn.hip), the whole-shape calls to the grid-pruned one;
    # every case carries the exhaustive scan's time and the bit-equality of all four outputs with it
    assert cs[0]["search"].startswith("matrix-core gated") and all(x["search"].startswith("grid") for x in cs[1:])
    assert all(x["bit_equal_to_exhaustive_scan"] is True and x["exhaustive_scan_ms"] > x["avg_call_ms"] for x in cs)


def test_one_stdout_line_under_the_drivers_launcher(cuda_device):
    """The driver's N>1 form (`python -m torch.distributed.run ... bench.py --gpus N`) with one rank: the process group is
    RCCL, whose version banner goes to stdout when the first communicator comes up — the bench keeps it off ITS stdout."""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-chamfer-standalone"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["rccl_ranks"] == 1 and d["steps"] == 3


def test_two_rank_branch_runs_on_one_gpu_over_gloo(cuda_device):
    """bench.py's world > 1 branch (process group, bucketed all-reduce hooks, `measure_collectives`, the `collectives`
    object, max-over-ranks timing, one stdout line) executed end to end: `--gpus 2` under the driver's launcher with
    MPA_DP_BACKEND=gloo, both ranks on this box's one GPU.  No RCCL rank exists here, and the line says so."""
    env = dict(os.environ, MPA_DP_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "2"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["config"]["rccl_ranks"] == 0 and d["config"]["dp_backend"] == "gloo" and d["config"]["parallelism"] == "dp2"
    assert abs(d["value"] - 2 * 32 * 20 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]  # whole-job aggregate
    c = d["collectives"]
    assert c["world"] == 2 and len(c["bucket_bytes"]) == 2 and all(b > 0 for b in c["bucket_bytes"])
    assert len(c["allreduce_ms"]) == 2 and all(t > 0 for t in c["allreduce_ms"])
    assert c["local_ms_per_step"] > 0 and c["exposed_ms"] >= 0 and 0.0 <= c["overlap_frac"] <= 1.0
    assert "cpu_baseline" not in d and "chamfer_standalone" not in d  # rank-0-at-N=1 legs only
    assert d["roofline"] is not None and d["roofline"]["launches"] == 3


def _committed_ms(tag):
    """ms_per_step of the newest committed bench line profiles/rNN_<tag>_bench_line.json."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}_bench_line.json")))
    assert files, f"no committed bench line for {tag}"
    with open(files[-1]) as f:
        return json.loads(f.read().strip().splitlines()[-1])["ms_per_step"], os.path.basename(files[-1])


@pytest.mark.parametrize("tag,flags", [("c3", ("--config", "c3")), ("c5", ("--config", "c5")),
                                       ("c2_bf16", ("--dtype", "bf16"))])
def test_other_configs_hold_their_committed_step_time(cuda_device, capsys, tag, flags):
    """BASELINE.json configs[2] / configs[4] and the bf16 variant line, 10 timed steps each under the driver's own GPU
    test run: the JSON line is printed (so the run's log witnesses the number) next to the builder-run line committed
    under profiles/ and their ratio.  Only the structure of the line is asserted."""
    d = _run("--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-chamfer-standalone", *flags)
    want, src = _committed_ms(tag)
    with capsys.disabled():
        slim = {k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "dtype")}
        slim["config"] = d["config"]["name"]
        slim["roofline"] = {k: d["roofline"][k] for k in ("kernel", "avg_launch_ms", "frac")} if d["roofline"] else None
        print(f"\n  BENCH {tag}: {json.dumps(slim)}  (committed: {want:.3f} ms in profiles/{src}; "
              f"ratio {d['ms_per_step'] / want:.3f}, recorded not asserted)", end="")
    assert d["ms_per_step"] > 0
    if d["ms_per_step"] > 2.0 * want:  # a stopwatch never fails the run (round-3 verdict), but a 2x regression is said aloud
        import warnings
        warnings.warn(f"BENCH {tag}: {d['ms_per_step']:.3f} ms/step is more than twice the committed {want:.3f} ms "
                      f"(profiles/{src})")
    if tag == "c2_bf16":
        assert d["dtype"] == "bf16" and "precision_note" in d["config"] and "cpu_baseline" not in d
    else:
        assert d["config"]["name"] == tag and "knn" in d["roofline"]["kernel"] and d["roofline"]["launches"] == 10
        assert d["roofline"]["binding"]["bound"] == "mfma_bf16"


def test_plumbing_config_with_its_cpu_baseline_and_self_check(cuda_device, capsys):
    """configs[0] (B-Global, semantic flags, P = 2, B = 4) with its full-size CPU baseline on all physical cores.  The
    step time and CPU rate are printed, not judged."""
    d = _run("--config", "c1", "--steps", "10", "--warmup", "5", "--no-chamfer-standalone")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "B=4," in c["sample"]
    r = d["roofline"]  # the whole-shape search of the fused loss (round 6: semantic models use it too): 5 min-of-N samples per step
    assert "fused loss" in r["kernel"] and r["launches"] == 50 and r["algorithmic_bytes_per_launch"] == 24.0 * 4 * 4000
    with capsys.disabled():
        print(f"\n  BENCH c1: {d['ms_per_step']:.3f} ms/step, {d['value']:.0f} parts/s; CPU {c['value']:.1f} parts/s on "
              f"{c['cores']} cores", end="")


def test_self_check_leg_reports_the_ratio(cuda_device, capsys):
    """`--self-check M` on the headline config (c2, device-bound) after 20 warm-up steps: the leg runs M further steps
    and reports their mean against the K timed ones.  The ratio is printed for the record; only its presence and the
    step count are asserted."""
    d = _run("--steps", "20", "--warmup", "20", "--self-check", "40", "--no-cpu-baseline", "--no-chamfer-standalone")
    sc = d["self_check"]
    assert sc["steps"] == 40 and sc["ms_per_step"] > 0 and sc["ratio_to_timed_mean"] > 0
    with capsys.disabled():
        print(f"\n  BENCH c2 self-check: {d['ms_per_step']:.3f} ms/step timed; {sc}", end="")


def test_kernel_timer_scope(cuda_device):
    from multi_part_assembly_amd import _lib

    t = _lib.KernelTimer(only=("shape_search_kernel",))
    _lib.KernelTimer.active = t
    try:
        assert _lib.KernelTimer.start("pointnet_forward[1x2x3]") is None
        assert _lib.KernelTimer.phase_events(["assembly_pose[x]", "assembly_finalize[x]"]) is None
        evs = _lib.KernelTimer.phase_events(["shape_search_kernel[x]"])
        assert evs is not None and len(evs) == 2 and all(e is not None for e in evs)
        arr = _lib.KernelTimer.handles([None] * 5 + evs)
        assert len(arr) == 7 and arr[0] is None and arr[5] is not None
        _lib.KernelTimer.add_phases(["shape_search_kernel[x]"], evs)
        torch.cuda.synchronize()
        assert list(t.summary()) == ["shape_search_kernel[x]"]
    finally:
        _lib.KernelTimer.active = None
