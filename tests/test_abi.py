"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports exactly what
include/mpa_hip.h declares (no compute calls — those need a GPU)."""
import ctypes
import subprocess

import pytest

from multi_part_assembly_amd import _build, _lib


@pytest.fixture(scope="module")
def built():
    return _build.build()


def test_header_and_signature_table_agree():
    assert sorted(_lib.SIGNATURES) == _lib.declared_functions()


def test_library_exports_every_declared_symbol(built):
    out = subprocess.run(["nm", "-D", "--defined-only", str(built)], capture_output=True, text=True,
                         check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [f for f in _lib.declared_functions() if f not in exported]
    assert not missing, f"declared in mpa_hip.h but not exported: {missing}"
    stray = sorted(s for s in exported if s.startswith("mpa_") and s not in _lib.SIGNATURES)
    assert not stray, f"exported but undeclared: {stray}"


def test_library_loads_and_reports_abi(built):
    handle = _lib.lib()
    assert handle.mpa_abi_version() == _lib.ABI_VERSION
    assert handle.mpa_last_error() == b""


def test_argument_validation_needs_no_gpu(built):
    L = _lib.lib()
    # negative sizes are rejected before anything touches the device
    st = L.mpa_chamfer_forward(None, None, -1, 4, 4, None, None, None, None, None, 0, None)
    assert st == -1 and b"negative" in L.mpa_last_error()
    st = L.mpa_chamfer_forward(None, None, 2, 4, 4, None, None, None, None, None, 0, None)
    assert st == -1 and b"null" in L.mpa_last_error()
    # empty problems are a no-op success
    assert L.mpa_chamfer_forward(None, None, 0, 4, 4, None, None, None, None, None, 0, None) == 0
    assert L.mpa_pose_apply_forward(None, None, None, None, ctypes.c_float(0), 0, 10, None, None) == 0


def test_graph_network_glue_validates_its_arguments(built):
    """csrc/gnn_glue.hip: shapes outside the kernels' instantiation and null pointers are refused with a message, before
    anything touches the device; the workspace queries are pure arithmetic."""
    L = _lib.lib()
    n = ctypes.c_int64()
    assert L.mpa_narrow_linear_relu_workspace(640, 7, 256, ctypes.byref(n)) == 0 and n.value == 5 * 256 * 17
    assert L.mpa_narrow_linear_relu_workspace(640, 17, 256, ctypes.byref(n)) == -1 and b"K=17" in L.mpa_last_error()
    assert L.mpa_narrow_linear_relu_forward(None, None, None, 640, 17, 256, None, None) == -1
    assert L.mpa_narrow_linear_relu_forward(None, None, None, 640, 7, 256, None, None) == -1
    assert b"null" in L.mpa_last_error()
    assert L.mpa_relation_head_workspace(12800, 512, ctypes.byref(n)) == 0 and n.value == 12800 + 800 * 513
    assert L.mpa_relation_head_workspace(12800, 510, ctypes.byref(n)) == -1 and b"multiple of 4" in L.mpa_last_error()
    assert L.mpa_relation_head_forward(None, None, None, None, 12800, 512, None, None, None) == -1
    assert L.mpa_relation_mean_forward(None, None, 640, 65, 128, None, None) == -1 and b"P=65" in L.mpa_last_error()
    assert L.mpa_relation_mean_backward(None, None, None, None, 640, 20, 128, None, None, None) == -1
    assert L.mpa_pair_rows_forward(None, None, 32, 20, 126, 0, None, None) == -1 and b"F=126" in L.mpa_last_error()
    assert L.mpa_pair_rows_backward(None, 32, 20, 128, 0, None, None, None) == -1 and b"null" in L.mpa_last_error()
    # the pose head takes any input width now (odd widths are padded inside its workspace)
    assert L.mpa_pose_head_workspace(640, 135, ctypes.byref(n)) == 0
    assert n.value == 640 * 780 + 64 + 2 * (640 + 256) * 192
    assert L.mpa_pose_head_workspace(640, 128, ctypes.byref(n)) == 0 and n.value == 640 * 780 + 64
    assert L.mpa_pose_head_workspace(640, 0, ctypes.byref(n)) == -1


def test_graph_network_glue_wrappers_reject_cpu_tensors():
    import torch
    from multi_part_assembly_amd import gnn_ops

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gnn_ops.narrow_linear_relu(torch.zeros(4, 7), torch.zeros(8, 7))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gnn_ops.relation_head(torch.zeros(4, 8), torch.zeros(1, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gnn_ops.relation_mean(torch.zeros(1, 2, 2, 4), torch.zeros(1, 2, 2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gnn_ops.pair_rows(torch.zeros(1, 2, 4), torch.zeros(1, 2, 4))


def test_code_object_targets_gfx950_only(built):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", str(built)],
                         capture_output=True, text=True).stdout
    blob = built.read_bytes()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_"):
        assert other not in blob, f"unexpected target {other!r} in the fat binary"


def test_host_wrappers_reject_cpu_tensors():
    import torch
    from multi_part_assembly_amd import chamfer

    a = torch.zeros(1, 4, 3)
    with pytest.raises((RuntimeError, AssertionError)):
        chamfer.chamfer_distance(a, a)
    with pytest.raises(RuntimeError):
        chamfer.chamfer_forward(a, a)


def test_no_kernel_uses_scratch_memory(built):
    """The compiler's resource-usage remarks of the last build (csrc/build/*.usage.json, written by _build.py):
    registers demoted to private (scratch) memory are a 2-3x slowdown no correctness test notices — it happened to
    the transformer GEMM (staging arrays stored through a reference) and to the grid search (parameter struct
    copied by value and indexed with a runtime shape index).  No exception is left: the one of rounds 2-3 (two spilled
    registers of the 64 -> 64 fused backward at the 256-register budget) went away with round 4's epilogue."""
    from multi_part_assembly_amd import _build

    usage = _build.resource_usage()
    if not usage:
        pytest.skip("objects were built without the usage report")
    assert len(usage) > 80  # every kernel of the library is in the report
    allowed = {}
    bad = {}
    for name, u in usage.items():
        limit = next((v for k, v in allowed.items() if k in name), 0)
        if u.get("scratch_bytes_per_lane", 0) > limit:
            bad[name] = u
    assert not bad, f"kernels using scratch memory: {bad}"


def test_bench_batches_have_representative_part_counts():
    """bench.py rotates four synthetic batches; except the historical batch 0 of rank 0 their part counts are redrawn until
    the sum is within 4 of the generator's expectation (the metric counts B x P slots, the work follows the valid parts)."""
    import importlib.util
    import pathlib
    spec = importlib.util.spec_from_file_location("bench", pathlib.Path(__file__).resolve().parents[1] / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for preset, want in (("everyday", 352), ("artifact", 512)):
        for seed in (1234, 2234, 1235, 99999):
            parts = bench.representative_parts(preset, seed)
            assert len(parts) == 32 and abs(sum(parts) - want) <= 4
            assert parts == bench.representative_parts(preset, seed)  # a function of (preset, seed) only
    lo, hi = 2, 20
    assert all(lo <= p <= hi for p in bench.representative_parts("everyday", 7))
