"""Parity of the HIP Chamfer operator (through the C ABI) with the oracle and the golden fixtures.

Bars (the reference's own, utils/chamfer/test_chamfer.py:72-76,92-101): indices exactly equal,
distances atol 1e-6 — met here bit for bit; backward checked against the fp64 autograd fixture and,
in fp32, against the sequential oracle within summation-order noise."""
import numpy as np
import pytest
import torch

from multi_part_assembly_amd import chamfer as C
from oracle import chamfer as oc

pytestmark = pytest.mark.gpu


def _dev(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


@pytest.mark.parametrize("case", ["a", "b", "c", "d", "hand", "tie"])
@pytest.mark.parametrize("variant", [0, 1, 2, None])
def test_forward_matches_golden(golden, cuda_device, case, variant):
    z = golden("chamfer")
    out = C.chamfer_forward(_dev(z[f"{case}_xyz1"], cuda_device), _dev(z[f"{case}_xyz2"], cuda_device),
                            variant=variant)
    for got, name in zip(out, ["dist1", "idx1", "dist2", "idx2"]):
        np.testing.assert_array_equal(got.cpu().numpy(), z[f"{case}_{name}"])
    assert out[1].dtype == torch.int64 and out[0].dtype == torch.float32


@pytest.mark.parametrize("shape", [(1, 1, 1), (5, 1, 9), (2, 7, 0), (2, 0, 5), (0, 4, 4), (3, 257, 255),
                                   (2, 1031, 300), (7, 1000, 1000), (1, 2500, 3100)])
def test_forward_matches_oracle_ragged(cuda_device, shape):
    B, n1, n2 = shape
    rng = np.random.default_rng(hash(shape) % 2**32)
    a = rng.standard_normal((B, n1, 3)).astype(np.float32)
    b = rng.standard_normal((B, n2, 3)).astype(np.float32)
    ref = oc.chamfer_forward(a, b)
    for variant in (0, 1, 2):
        out = C.chamfer_forward(_dev(a, cuda_device), _dev(b, cuda_device), variant=variant)
        for got, want in zip(out, ref):
            np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_forward_special_values(cuda_device):
    a = np.zeros((1, 3, 3), np.float32)
    b = np.array([[[np.nan, 0, 0], [1, 0, 0], [np.inf, 0, 0], [1, 0, 0]]], np.float32)
    ref = oc.chamfer_forward(a, b)
    out = C.chamfer_forward(_dev(a, cuda_device), _dev(b, cuda_device))
    np.testing.assert_array_equal(out[1].cpu().numpy(), ref[1])
    np.testing.assert_array_equal(out[0].cpu().numpy(), ref[0])
    # the NaN / inf targets themselves find their nearest query (NaN compares false everywhere)
    np.testing.assert_array_equal(out[3].cpu().numpy(), ref[3])
    # coordinates so large that every distance exceeds 1e32: nothing wins, (1e32, -1)
    far = np.full((1, 2, 3), 3e16, np.float32)
    out = C.chamfer_forward(_dev(a, cuda_device), _dev(far, cuda_device))
    assert (out[1].cpu().numpy() == -1).all() and (out[0].cpu().numpy() == np.float32(1e32)).all()


def test_forward_double_matches_oracle(cuda_device):
    rng = np.random.default_rng(5)
    a, b = rng.random((2, 513, 3)), rng.random((2, 300, 3))
    ref = oc.chamfer_forward(a, b)
    out = C.chamfer_forward(_dev(a, cuda_device), _dev(b, cuda_device))
    for got, want in zip(out, ref):
        np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_backward_matches_fp64_fixture(golden, cuda_device):
    z = golden("chamfer")
    x1 = _dev(z["bwd_xyz1"], cuda_device).requires_grad_()
    x2 = _dev(z["bwd_xyz2"], cuda_device).requires_grad_()
    d1, d2 = C.chamfer_distance(x1, x2)
    np.testing.assert_allclose(d1.detach().cpu().numpy(), z["bwd_dist1"], rtol=1e-14)
    ((d1 * _dev(z["bwd_g1"], cuda_device)).sum() + (d2 * _dev(z["bwd_g2"], cuda_device)).sum()).backward()
    np.testing.assert_allclose(x1.grad.cpu().numpy(), z["bwd_gxyz1"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(x2.grad.cpu().numpy(), z["bwd_gxyz2"], rtol=1e-12, atol=1e-14)


def test_gradcheck_double(cuda_device):
    # the reference's own backward test (test_chamfer.py:92-101)
    g = torch.Generator().manual_seed(3)
    x1 = torch.rand(2, 64, 3, generator=g, dtype=torch.float64).to(cuda_device).requires_grad_()
    x2 = torch.rand(2, 64, 3, generator=g, dtype=torch.float64).to(cuda_device).requires_grad_()
    assert torch.autograd.gradcheck(C.chamfer_distance, (x1, x2, False))
    x1t = x1.detach().transpose(1, 2).contiguous().requires_grad_()
    x2t = x2.detach().transpose(1, 2).contiguous().requires_grad_()
    assert torch.autograd.gradcheck(C.chamfer_distance, (x1t, x2t, True))


def test_backward_fp32_vs_oracle(cuda_device):
    rng = np.random.default_rng(11)
    a = rng.random((4, 1000, 3)).astype(np.float32)
    b = rng.random((4, 700, 3)).astype(np.float32)
    g1 = rng.standard_normal((4, 1000)).astype(np.float32)
    g2 = rng.standard_normal((4, 700)).astype(np.float32)
    d1, i1, d2, i2 = oc.chamfer_forward(a, b)
    r1, r2 = oc.chamfer_backward(g1, g2, a, b, i1, i2)
    o1, o2 = C.chamfer_backward(*[_dev(x, cuda_device) for x in (g1, g2, a, b, i1, i2)])
    # different summation order of colliding scatter contributions only
    np.testing.assert_allclose(o1.cpu().numpy(), r1, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(o2.cpu().numpy(), r2, rtol=1e-5, atol=1e-6)


def test_nn_distance_and_wrapper_options(cuda_device):
    g = torch.Generator().manual_seed(9)
    a = torch.rand(3, 3, 50, generator=g).to(cuda_device)   # BCN
    b = torch.rand(3, 3, 60, generator=g).to(cuda_device)
    d1, i1, d2, i2 = C.nn_distance(a, b)                     # default transpose=True
    e1, e2 = C.chamfer_distance(a, b, transpose=True)
    assert torch.equal(d1, e1) and torch.equal(d2, e2)
    s1, _ = C.chamfer_distance(a, b, transpose=True, sqrt=True)
    assert torch.allclose(s1, d1.clamp_min(1e-12).sqrt())
    u1, u2 = C.chamfer_distance(a[0].t(), b[0].t())           # 2-D inputs get a batch axis
    assert torch.equal(u1[0], d1[0]) and u1.shape == (1, 50)


def test_autocast_keeps_fp32(cuda_device):
    a = torch.rand(2, 40, 3, device=cuda_device)
    b = torch.rand(2, 30, 3, device=cuda_device)
    with torch.autocast("cuda", dtype=torch.float16):
        d1, _ = C.chamfer_distance(a.half(), b.half())
    assert d1.dtype == torch.float32


def test_full_size_properties(cuda_device):
    """BASELINE sizes, checked through size-independent properties instead of the (slow) oracle:
    (1) all three scan variants agree bit for bit; (2) dist equals the distance to the returned index
    recomputed in torch; (3) symmetry: swapping the clouds swaps the outputs; (4) a cloud against
    itself gives zero distance and the identity (lowest duplicate) index."""
    g = torch.Generator().manual_seed(21)
    for B, n in ((640, 1000), (32, 20000)):
        a = (torch.rand(B, n, 3, generator=g) - 0.5).to(cuda_device)
        b = (torch.rand(B, n, 3, generator=g) - 0.5).to(cuda_device)
        o0 = C.chamfer_forward(a, b, variant=0)
        o1 = C.chamfer_forward(a, b, variant=1)
        o2 = C.chamfer_forward(a, b, variant=2)
        for x, y, w in zip(o0, o1, o2):
            assert torch.equal(x, y) and torch.equal(x, w)
        d1, i1, d2, i2 = o1
        near = torch.gather(b, 1, i1[..., None].expand(-1, -1, 3))
        diff = a - near
        rec = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
        assert torch.equal(rec, d1)
        s = C.chamfer_forward(b, a)
        assert torch.equal(s[0], d2) and torch.equal(s[1], i2) and torch.equal(s[2], d1)
        z = C.chamfer_forward(a, a)
        assert (z[0] == 0).all()
        assert torch.equal(z[1], torch.arange(n, device=cuda_device).expand(B, n))


# ---- the grid-pruned search behind mpa_chamfer_forward (csrc/grid_nn.hip: cloud_sort_kernel + grid_search_kernel<true>) ----
def _cloud(kind, rng, B, n):
    """Clouds that stress what the pruning assumes: clumps, flat / degenerate extents, exact duplicates, far outliers and
    the padded-part fill of shape_cd_loss (utils/loss.py:173-175)."""
    if kind == "normal":
        return rng.standard_normal((B, n, 3)).astype(np.float32)
    if kind == "uniform_offset":  # an origin far larger than the cell size
        return (rng.random((B, n, 3)) * 0.3 + np.array([500.0, -700.0, 90.0])).astype(np.float32)
    if kind == "flat":  # a plane: one axis has no extent at all
        x = rng.random((B, n, 3)).astype(np.float32)
        x[..., 2] = 0.25
        return x
    if kind == "line":
        t = rng.random((B, n, 1)).astype(np.float32)
        return (t * np.array([1.0, 2.0, -1.0], np.float32)).astype(np.float32)
    if kind == "clumps":
        c = rng.standard_normal((B, 6, 3)).astype(np.float32)
        pick = rng.integers(0, 6, (B, n))
        return (np.take_along_axis(c, pick[..., None].repeat(3, -1), 1) + 1e-3 * rng.standard_normal((B, n, 3))).astype(np.float32)
    if kind == "duplicates":  # every point four times, scattered: ties resolved by the lowest index
        base = rng.random((B, (n + 3) // 4, 3)).astype(np.float32)
        x = np.concatenate([base] * 4, 1)[:, :n]
        return np.stack([x[b][rng.permutation(n)] for b in range(B)])
    if kind == "one_point":
        return np.broadcast_to(rng.random((B, 1, 3)).astype(np.float32), (B, n, 3)).copy()
    if kind == "outliers":  # 5 % of the points up to 1e6 away
        x = rng.standard_normal((B, n, 3)).astype(np.float32)
        far = rng.random((B, n)) < 0.05
        x[far] *= np.float32(10.0) ** rng.integers(1, 7, (int(far.sum()), 1)).astype(np.float32)
        return x
    if kind == "padded_fill":  # shape_cd_loss: the last parts are N copies of (1e3, 1e3, 1e3) + t
        x = (rng.random((B, n, 3)) - 0.5).astype(np.float32)
        for b in range(B):
            cut = int(n * rng.uniform(0.1, 0.9))
            part = max(1, (n - cut) // 4)
            for s in range(cut, n, part):
                x[b, s:s + part] = np.float32(1e3) + rng.uniform(-0.5, 0.5, 3).astype(np.float32)
        return x
    raise KeyError(kind)


KINDS = ["normal", "uniform_offset", "flat", "line", "clumps", "duplicates", "one_point", "outliers", "padded_fill"]


@pytest.mark.parametrize("kind2", ["normal", "padded_fill", "outliers"])
@pytest.mark.parametrize("kind1", KINDS)
def test_grid_search_matches_oracle(cuda_device, kind1, kind2):
    """variant 3 forces the pruned search at sizes the oracle finishes in a moment: every output bit-equal to the
    sequential strict-`<` scan of oracle/chamfer_ref.c."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"{kind1}/{kind2}".encode()))
    for B, n1, n2 in ((3, 700, 900), (2, 1, 1500), (1, 2300, 64)):
        a, b = _cloud(kind1, rng, B, n1), _cloud(kind2, rng, B, n2)
        ref = oc.chamfer_forward(a, b)
        out = C.chamfer_forward(_dev(a, cuda_device), _dev(b, cuda_device), variant=3)
        for got, want, name in zip(out, ref, ("dist1", "idx1", "dist2", "idx2")):
            np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg=f"{name} {kind1} {kind2} {(B, n1, n2)}")


@pytest.mark.parametrize("offset, spacing", [(0.0, 1.0), (30.0, 1e-2), (1000.0, 1e-3), (-3e4, 0.5), (2e7, 4.0), (1e12, 1e6)])
def test_grid_search_gate_far_from_the_origin(cuda_device, offset, spacing):
    """The scan of the pruned search rejects candidates with three FMAs on |t|^2 - 2 q.t (grid_nn.hip: scan_cand); the
    cancellation in that form grows with the distance of the clouds from the origin, and the gate's margin must grow with
    it.  Lattice points (many exact ties, lowest index wins) and jittered points at offsets where one ulp of a coordinate is
    a sizeable fraction of the spacing: every output bit-equal to the oracle's in-order scan."""
    rng = np.random.default_rng(int(abs(offset)) % 1000 + 7)
    n1, n2 = 700, 640
    lat = rng.integers(0, 9, (2, n1 + n2, 3)).astype(np.float32) * np.float32(spacing)
    jit = (rng.standard_normal((2, n1 + n2, 3)) * spacing * 3).astype(np.float32)
    pts = np.concatenate([lat[:1], jit[1:]], 0) + np.float32(offset) * np.array([1.0, -0.5, 0.25], np.float32)
    a = np.ascontiguousarray(pts[:, :n1]).astype(np.float32)
    b = np.ascontiguousarray(pts[:, n1:]).astype(np.float32)
    ref = oc.chamfer_forward(a, b)
    out = C.chamfer_forward(_dev(a, cuda_device), _dev(b, cuda_device), variant=3)
    for got, want, name in zip(out, ref, ("dist1", "idx1", "dist2", "idx2")):
        np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg=f"{name} offset {offset} spacing {spacing}")


def test_grid_search_hands_non_finite_samples_to_the_scan(cuda_device):
    """A sample with a NaN / inf / > 1e15 coordinate is answered by the exhaustive scan (same contract as the scan's own
    special-value test); its neighbours in the batch still go through the grid."""
    rng = np.random.default_rng(77)
    a = rng.standard_normal((5, 600, 3)).astype(np.float32)
    b = rng.standard_normal((5, 800, 3)).astype(np.float32)
    a[1, 17, 0] = np.nan
    b[2, 5, 2] = np.inf
    b[3, 9, 1] = -3e16
    a[4, 100] = 9e14  # large but legal: stays on the grid path
    ref = oc.chamfer_forward(a, b)
    out = C.chamfer_forward(_dev(a, cuda_device), _dev(b, cuda_device), variant=3)
    for got, want in zip(out, ref):
        np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_grid_search_full_size_equals_the_scan(cuda_device):
    """§8(d)'s whole-shape call, [32, 20000, 3] against itself with the 1e3 padding fill applied: the default dispatch
    (pruned search) against the exhaustive scan, bit for bit, on shape-like clouds whose prediction is (a) far from and
    (b) close to the target; plus unequal cloud sizes."""
    g = torch.Generator().manual_seed(5)
    B, P, N = 32, 20, 1000
    parts = torch.randint(2, P + 1, (B,), generator=g)
    base = (torch.rand(B, P, N, 3, generator=g) - 0.5) * torch.rand(B, P, 1, 3, generator=g) * 0.6
    off1, off2 = (torch.rand(B, P, 1, 3, generator=g) - 0.5) * 0.8, (torch.rand(B, P, 1, 3, generator=g) - 0.5) * 0.8
    for close in (False, True):
        s1 = base + off1
        s2 = base + (off1 + 0.01 * off2 if close else off2)
        for b in range(B):
            s1[b, parts[b]:] = 1e3 + off1[b, parts[b]:]
            s2[b, parts[b]:] = 1e3
        x1 = s1.reshape(B, P * N, 3).contiguous().to(cuda_device)
        x2 = s2.reshape(B, P * N, 3).contiguous().to(cuda_device)
        fast = C.chamfer_forward(x1, x2)
        slow = C.chamfer_forward(x1, x2, variant=2)
        for f, s in zip(fast, slow):
            assert torch.equal(f, s)
    y1 = x1[:, :12345].contiguous()
    fast, slow = C.chamfer_forward(y1, x2), C.chamfer_forward(y1, x2, variant=2)
    for f, s in zip(fast, slow):
        assert torch.equal(f, s)


def test_chamfer_workspace_contract(cuda_device):
    """mpa_chamfer_workspace sizes the scratch; without it (or with too little) mpa_chamfer_forward still answers, by the
    exhaustive scan; variant 3 without workspace is an error, not a silent fallback."""
    import ctypes
    from multi_part_assembly_amd import _lib
    L = _lib.lib()
    nb = ctypes.c_int64(-1)
    assert L.mpa_chamfer_workspace(32, 20000, 20000, ctypes.byref(nb)) == 0 and nb.value > 0
    assert L.mpa_chamfer_workspace(4, 0, 100, ctypes.byref(nb)) == 0 and nb.value == 0
    # sizes the exhaustive scan answers reserve nothing (the per-part call used to ask for 725 MiB of cell tables) ...
    assert L.mpa_chamfer_workspace(640, 1000, 1000, ctypes.byref(nb)) == 0 and nb.value == 0
    assert L.mpa_chamfer_workspace(8000, 100, 100, ctypes.byref(nb)) == 0 and nb.value == 0
    # ... unless the pruned search is pinned for them
    assert L.mpa_chamfer_workspace_variant(640, 1000, 1000, 3, ctypes.byref(nb)) == 0 and nb.value > 0
    assert L.mpa_chamfer_workspace_variant(640, 1000, 1000, 2, ctypes.byref(nb)) == 0 and nb.value == 0
    assert L.mpa_chamfer_workspace_variant(32, 20000, 20000, -1, ctypes.byref(nb)) == 0 and nb.value > 0
    a = torch.rand(2, 4000, 3, device=cuda_device)
    b = torch.rand(2, 4000, 3, device=cuda_device)
    d1 = torch.empty(2, 4000, device=cuda_device)
    d2 = torch.empty(2, 4000, device=cuda_device)
    i1 = torch.empty(2, 4000, dtype=torch.int64, device=cuda_device)
    i2 = torch.empty(2, 4000, dtype=torch.int64, device=cuda_device)
    s = _lib.current_stream(cuda_device)
    args = (a.data_ptr(), b.data_ptr(), 2, 4000, 4000, d1.data_ptr(), i1.data_ptr(), d2.data_ptr(), i2.data_ptr())
    assert L.mpa_chamfer_forward(*args, None, 0, s) == 0
    want = C.chamfer_forward(a, b)
    assert torch.equal(d1, want[0]) and torch.equal(i1, want[1]) and torch.equal(i2, want[3])
    assert L.mpa_chamfer_forward_variant(*args, 3, None, 0, s) == _lib.lib().mpa_chamfer_forward_variant(*args, 3, None, 0, s) != 0
    assert b"workspace" in L.mpa_last_error()


def test_grid_search_far_beyond_the_loss_sizes(cuda_device):
    """Clouds ten times the whole-shape call's size (150 000 and 200 000 points, unequal, with far outliers and a long run
    of one repeated point): the pruned search's work lists, record strides and run heads at sizes the fused loss never
    reaches — bit-equal to the exhaustive scan."""
    g = torch.Generator().manual_seed(17)
    a = torch.randn(2, 150_000, 3, generator=g) * torch.tensor([1.0, 0.3, 2.0])
    b = torch.randn(2, 200_000, 3, generator=g) * 0.8 + 0.1
    a[:, 1000:1200] *= 1e4
    b[0, 50_000:90_000] = torch.tensor([1e3, 1e3, -1e3])
    a, b = a.to(cuda_device), b.to(cuda_device)
    fast = C.chamfer_forward(a, b)
    slow = C.chamfer_forward(a, b, variant=2)
    for f, s in zip(fast, slow):
        assert torch.equal(f, s)
