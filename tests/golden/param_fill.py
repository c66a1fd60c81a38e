"""Deterministic parameter filler shared by the fixture generator (make_golden.py, run against the reference) and
the parity tests (run against this repository's modules): both sides rebuild the SAME weights from the parameter
names, so the fixtures of the multi-million-parameter callers (DGL / RGL-NET) need not store them.

Also the compact gradient record: small tensors in full, large ones as a strided sample plus two norms."""
import zlib

import numpy as np
import torch

SAMPLE = 1024
FULL_LIMIT = 4096


def fill_parameters(module, seed):
    """Overwrites every parameter and buffer of `module` in place, keyed by its state_dict name."""
    with torch.no_grad():
        for name, t in sorted(module.state_dict().items()):
            g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
            if not t.is_floating_point():
                t.zero_()  # num_batches_tracked
            elif t.dim() >= 2:
                fan_in = int(np.prod(t.shape[1:]))
                t.copy_(torch.randn(t.shape, generator=g) / np.sqrt(fan_in))
            elif name.endswith("running_var"):
                t.copy_(torch.rand(t.shape, generator=g) + 0.5)
            elif name.endswith(".weight"):  # 1-d weights are normalisation scales
                t.copy_(torch.rand(t.shape, generator=g) + 0.5)
            else:  # biases, running means
                t.copy_(torch.randn(t.shape, generator=g) * 0.1)


def compact(prefix, name, array):
    """-> dict of arrays describing `array` (numpy)."""
    a = np.asarray(array, dtype=np.float32).reshape(-1)
    if a.size <= FULL_LIMIT:
        return {f"{prefix}{name}": a.copy()}
    idx = np.linspace(0, a.size - 1, SAMPLE).astype(np.int64)
    return {f"{prefix}{name}#sample": a[idx].copy(),
            f"{prefix}{name}#norms": np.array([np.abs(a).sum(), np.sqrt((a.astype(np.float64) ** 2).sum())], dtype=np.float64)}


def compare(record, prefix, name, array, rel, floor=0.0):
    """Asserts that `array` matches what `compact` recorded under (prefix, name): max |a - ref| < rel * max(|ref|,
    floor).  `floor` keeps pure-noise tensors (e.g. the gradient of a conv bias in front of a BatchNorm, which is
    zero up to rounding) from being compared relative to their own noise."""
    a = np.asarray(array, dtype=np.float32).reshape(-1)
    key = f"{prefix}{name}"
    if key in record:
        ref = record[key]
        err = np.abs(a - ref).max() / max(np.abs(ref).max(), floor, 1e-12)
        assert err < rel, (key, err)
        return
    idx = np.linspace(0, a.size - 1, SAMPLE).astype(np.int64)
    ref = record[key + "#sample"]
    err = np.abs(a[idx] - ref).max() / max(np.abs(ref).max(), floor, 1e-12)
    assert err < rel, (key, "sample", err)
    norms = record[key + "#norms"]
    mine = np.array([np.abs(a).sum(), np.sqrt((a.astype(np.float64) ** 2).sum())])
    assert np.all(np.abs(mine - norms) / np.maximum(np.abs(norms), floor * a.size ** 0.5 + 1e-12) < rel), (key, "norms", mine, norms)


def anchored_errors(record, name, array, floor=0.0):
    """(err_mine, err_ref32, outliers): max-abs distance of `array`, and of the float32 reference gradient recorded under
    "grad." + name, from the float64 evaluation of the same step recorded under "grad64." + name — both relative to
    max(|float64 gradient|, floor), on the recorded entries (the full tensor or its strided sample); `outliers` = the
    number of recorded entries of `array` further than 1e-2 from float64 (an isolated ReLU / arg-max flip shows up as one
    or two entries, a wiring error as most of them) and the number of recorded entries."""
    a = np.asarray(array, dtype=np.float64).reshape(-1)
    if "grad64." + name in record:
        t, r = record["grad64." + name], record["grad." + name]
    else:
        idx = np.linspace(0, a.size - 1, SAMPLE).astype(np.int64)
        a, t, r = a[idx], record[f"grad64.{name}#sample"], record[f"grad.{name}#sample"]
    t = t.astype(np.float64)
    scale = max(np.abs(t).max(), floor, 1e-12)
    e = np.abs(a - t) / scale
    return float(e.max()), float(np.abs(r.astype(np.float64) - t).max() / scale), (int((e > 1e-2).sum()), int(e.size))


def grad64_scale(record, name):
    """Largest |float64 gradient| recorded for parameter `name` (0.0 if there is no record)."""
    for key in ("grad64." + name, f"grad64.{name}#sample"):
        if key in record:
            return float(np.abs(record[key]).max())
    return 0.0
