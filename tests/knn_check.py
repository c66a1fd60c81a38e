"""Shared check of a kNN graph against the reference's own `knn` output (tests/golden/dgcnn_graphs.npz, captured from
multi_part_assembly/models/modules/encoder/dgcnn.py:8-15 by make_golden.py:gen_dgcnn_graphs).

torch.topk leaves the order among equal scores open and, in 64/128-d, the reference's Gram matrix comes out of a blocked
BLAS whose summation order is not defined — so a neighbour list may legitimately differ from the reference's in a
near-tie.  The rule checked here: the neighbour SETS are equal, except where every neighbour we picked instead of one of
the reference's has a float64 score within `gap` (relative to |x_i|^2 + |x_j|^2, the magnitude the fp32 score is rounded
at) of the pick it replaces.  Returns the statistics so the caller can print / bound them."""
import numpy as np


def compare_with_reference_graph(x, got, ref, gap=1e-6):
    """x [n, N, C] float32, got / ref [n, N, k] integer.  -> dict(rows, set_mismatch_rows, swapped_picks,
    in_order_equal, worst_gap); raises AssertionError on a mismatch that is not a float64 near-tie."""
    x = np.asarray(x, np.float32)
    got = np.asarray(got).astype(np.int64)
    ref = np.asarray(ref).astype(np.int64)
    n, N, _ = x.shape
    assert got.shape == ref.shape == (n, N, ref.shape[-1])
    assert got.min() >= 0 and got.max() < N
    xd = x.astype(np.float64)
    sq = (xd ** 2).sum(-1)
    stats = dict(rows=n * N, set_mismatch_rows=0, swapped_picks=0, worst_gap=0.0,
                 in_order_equal=float((got == ref).mean()))
    gs, rs = np.sort(got, -1), np.sort(ref, -1)
    assert (np.diff(gs, axis=-1) > 0).all(), "duplicate neighbour in a list"
    for c, i in zip(*np.nonzero((gs != rs).any(-1))):
        a, b = set(got[c, i].tolist()), set(ref[c, i].tolist())
        ours, theirs = sorted(a - b), sorted(b - a)
        score = lambda js: -np.sort(sq[c, i] + sq[c, js] - 2.0 * xd[c, js] @ xd[c, i])  # descending scores
        scale = sq[c, i] + sq[c, ours + theirs].max()
        g = float(np.abs(score(ours) - score(theirs)).max() / scale)
        assert g < gap, f"cloud {c} point {i}: picks {ours} vs the reference's {theirs}, float64 score gap {g:.3e}"
        stats["set_mismatch_rows"] += 1
        stats["swapped_picks"] += len(ours)
        stats["worst_gap"] = max(stats["worst_gap"], g)
    return stats
