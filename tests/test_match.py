"""Device-side matching of equivalent parts: the restated assignment algorithm against scipy
(the reference's host call, base_model.py:175), the HIP solver against scipy, and the batched `match_parts` against
the reference's per-group composition (sub-sample, transform, Chamfer cost matrix, scipy, permute)."""
import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment as scipy_lsa

from oracle.lsap import linear_sum_assignment_square


def _cases(seed, count, max_n):
    g = np.random.default_rng(seed)
    for t in range(count):
        n = int(g.integers(1, max_n + 1))
        kind = t % 3
        if kind == 0:
            c = g.random((n, n))
        elif kind == 1:  # tie-heavy: the tie-breaking rules decide the answer
            c = g.integers(0, 4, size=(n, n)).astype(np.float64)
        else:  # near-duplicate rows (equivalent parts in similar poses)
            c = np.repeat(g.random((1, n)), n, 0) + 1e-3 * g.random((n, n))
        yield c.astype(np.float32)


def test_restated_algorithm_equals_scipy():
    for c in _cases(0, 300, 20):
        np.testing.assert_array_equal(linear_sum_assignment_square(c), scipy_lsa(c.astype(np.float64))[1])


@pytest.mark.gpu
def test_hip_lsap_equals_scipy(cuda_device):
    from multi_part_assembly_amd import matching
    mats = list(_cases(1, 200, 20)) + list(_cases(2, 8, 64))
    ld = 64
    cost = np.full((len(mats), ld, ld), np.nan, dtype=np.float32)  # entries outside a problem are never read
    sizes = np.zeros(len(mats), dtype=np.int32)
    for i, c in enumerate(mats):
        n = c.shape[0]
        cost[i, :n, :n] = c
        sizes[i] = n
    got = matching.linear_sum_assignment(torch.from_numpy(cost).to(cuda_device), torch.from_numpy(sizes)).cpu().numpy()
    for i, c in enumerate(mats):
        n = c.shape[0]
        np.testing.assert_array_equal(got[i, :n], scipy_lsa(c.astype(np.float64))[1], err_msg=f"problem {i}")
        assert (got[i, n:] == -1).all()


@pytest.mark.gpu
def test_match_parts_equals_reference_composition(cuda_device):
    from multi_part_assembly_amd import config, matching
    from multi_part_assembly_amd.base_model import BaseModel
    from multi_part_assembly_amd.rotation import Rotation3D

    g = torch.Generator().manual_seed(3)
    B, P, N = 6, 9, 160
    ids = torch.tensor([[0, 1, 1, 1, 2, 2, 0, 0, 0],
                        [1, 1, 0, 0, 0, 0, 0, 0, 0],
                        [0, 0, 0, 0, 0, 0, 0, 0, 0],
                        [1, 2, 1, 2, 3, 3, 3, 3, 0],
                        [2, 2, 2, 2, 2, 0, 0, 0, 0],   # group 1 absent
                        [1, 1, 1, 1, 1, 1, 1, 1, 1]])
    # equivalent parts: the members of a group share one point cloud
    base = torch.randn(B, 4, N, 3, generator=g) * 0.2
    pcs = base[torch.arange(B)[:, None], ids.clamp(max=3)].contiguous()
    quat = lambda: torch.nn.functional.normalize(torch.randn(B, P, 4, generator=g), dim=-1)
    gt_t, gt_q = torch.rand(B, P, 3, generator=g) - 0.5, quat()
    # predictions = a shuffled copy of the GT poses inside each group, plus noise: a non-trivial optimum
    pr_t, pr_q = gt_t.clone(), gt_q.clone()
    for b in range(B):
        for grp in range(1, int(ids[b].max()) + 1):
            m = (ids[b] == grp).nonzero()[:, 0]
            if len(m):
                sh = m[torch.randperm(len(m), generator=g)]
                pr_t[b, m], pr_q[b, m] = gt_t[b, sh], gt_q[b, sh]
    pr_t += 0.01 * torch.randn(pr_t.shape, generator=g)
    G, n = int(ids.max()), 100
    sample_idx = torch.stack([torch.stack([torch.randperm(N, generator=g)[:n] for _ in range(G)]) for _ in range(B)])
    dev = cuda_device
    new_t, new_q, perm, cost, col4row = matching.match_parts(
        pcs.to(dev), pr_t.to(dev), pr_q.to(dev), gt_t.to(dev), gt_q.to(dev), ids.to(dev), sample_idx, ret_aux=True)

    # the reference's composition, one group at a time, with the same sub-samples
    cfg = config.global_partnet_chair()
    cfg.data.max_num_part = P
    model = BaseModel(cfg)
    want_t, want_q = gt_t.clone(), gt_q.clone()
    moved = 0
    for b in range(B):
        for grp in range(1, int(ids[b].max()) + 1):
            m = (ids[b] == grp).nonzero()[:, 0].tolist()
            if not m:
                continue
            torch.randperm_backup = torch.randperm
            try:
                torch.randperm = lambda N_, _s=sample_idx[b, grp - 1]: _s  # the method takes [:n] of it
                _, matched = model._linear_sum_assignment(pcs[b, m].to(dev), pr_t[b, m].to(dev), pr_q[b, m].to(dev),
                                                          gt_t[b, m].to(dev), gt_q[b, m].to(dev))
            finally:
                torch.randperm = torch.randperm_backup
            matched = matched.cpu()
            want_t[b, m], want_q[b, m] = gt_t[b, m][matched], gt_q[b, m][matched]
            moved += int((matched != torch.arange(len(m))).sum())
            np.testing.assert_array_equal(col4row[b, grp - 1, : len(m)].cpu().numpy(), matched.numpy())
    assert moved > 5  # the optimum is not the identity
    np.testing.assert_array_equal(new_t.cpu().numpy(), want_t.numpy())
    np.testing.assert_array_equal(new_q.cpu().numpy(), want_q.numpy())
    # perm is a permutation inside every group and the identity elsewhere
    pm = perm.cpu()
    assert (pm.sort(1).values == torch.arange(P)).all()
    assert (pm[ids == 0] == torch.arange(P).expand(B, P)[ids == 0]).all()
    assert (torch.gather(ids, 1, pm.long()) == ids).all()
    # through the model: same result, one host copy of match_ids at most
    rt, rr = model._match_parts(pcs.to(dev), pr_t.to(dev), Rotation3D(pr_q.to(dev)), gt_t.to(dev),
                                Rotation3D(gt_q.to(dev)), ids.to(dev))
    assert rt.shape == (B, P, 3) and rr.rot.shape == (B, P, 4)
