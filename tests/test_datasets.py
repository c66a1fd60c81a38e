"""Batch producers against fixtures produced by the reference's own
`GeometryPartDataset.__getitem__` / `PartNetPartDataset.__getitem__` (tests/golden/make_golden.py:gen_batch_producer;
the PartNet mini dataset under tests/golden/partnet_mini/ was written by that script in the reference's format)."""
import random

import numpy as np
import pytest
import torch

from pathlib import Path

from multi_part_assembly_amd import datasets

GOLDEN = Path(__file__).resolve().parent / "golden"

PARTS = [2, 5, 4]
PARTNET_KEYS = ("part_label", "part_ids", "match_ids", "contact_points", "sym", "valid_matrix")


def _partnet(device):
    return datasets.PartNetBatchProducer(str(GOLDEN / "partnet_mini"), "Chair.train.npy", PARTNET_KEYS,
                                         num_part_category=5, min_num_part=2, max_num_part=8, device=device)


def test_partnet_items_match_reference(golden):
    z = golden("batch_producer")
    ds = _partnet("cpu")
    np.testing.assert_array_equal(np.array(ds.shape_ids), z["partnet.shape_ids"])  # 104 has too many parts
    assert len(ds) == 4
    for i in range(len(ds)):
        item = ds.item(i)
        want = {k[len(f"partnet.{i}."):]: v for k, v in z.items() if k.startswith(f"partnet.{i}.")}
        assert set(item) == set(want)
        for k, v in want.items():
            got = np.asarray(item[k])
            assert got.dtype == v.dtype and got.shape == v.shape, k
            np.testing.assert_array_equal(got, v, err_msg=k)


def test_label_derivations_documented_cases():
    # the two examples of the reference's docstring (partnet_data.py:113-125)
    np.testing.assert_array_equal(datasets.match_ids(np.array([0, 4, 4, 4, 1, 2, 3]), 7), [0, 1, 1, 1, 0, 0, 0])
    np.testing.assert_array_equal(datasets.match_ids(np.array([0, 1, 1, 2, 3, 4, 4, 4]), 9),
                                  [0, 1, 1, 0, 0, 2, 2, 2, 0])
    inst = datasets.instance_labels(np.array([0, 4, 4, 4, 1, 2, 3]), 7)
    np.testing.assert_array_equal(inst.argmax(1), [0, 0, 1, 2, 0, 0, 0])
    assert inst.sum() == 7


def test_partnet_collate_cpu(golden):
    z = golden("batch_producer")
    b = _partnet("cpu").batch([0, 2])
    assert b["part_pcs"].shape == (2, 8, 32, 3) and b["part_pcs"].dtype == torch.float32
    assert b["data_id"].dtype == torch.int64 and b["data_id"].tolist() == [0, 2]
    np.testing.assert_array_equal(b["match_ids"][1].numpy(), z["partnet.2.match_ids"])


def test_geometry_randomness_matches_reference(golden):
    """The host side draws rotations and point orders with the reference's RNG calls: same seeds, same quaternions."""
    z = golden("batch_producer")
    for tag, rot_range in (("free", -1), ("range", 30.0)):
        prod = datasets.GeometryBatchProducer(num_points=64, max_num_part=6, rot_range=rot_range, device="cpu")
        np.random.seed(77)
        random.seed(77)
        for b, p in enumerate(PARTS):
            for i in range(p):
                _, q = prod._draw_rotation()
                prod._draw_order(64)
                np.testing.assert_allclose(q.astype(np.float32), z[f"geo.{tag}.part_quat"][b, i], rtol=0, atol=1e-7)


def test_geometry_producer_rejects_cpu():
    prod = datasets.GeometryBatchProducer(num_points=8, max_num_part=3, device="cpu")
    with pytest.raises(RuntimeError):
        prod.produce([np.zeros((2, 8, 3))])
    with pytest.raises(RuntimeError):
        datasets.GeometryBatchProducer(device="cpu", data_list=["x"]).batch([0])  # no mesh sampler in this image


@pytest.mark.gpu
def test_geometry_batches_match_reference(golden, cuda_device):
    z = golden("batch_producer")
    raws = [z[f"geo.raw{i}"] for i in range(len(PARTS))]
    for tag, rot_range in (("free", -1), ("range", 30.0)):
        prod = datasets.GeometryBatchProducer(num_points=64, max_num_part=6, rot_range=rot_range,
                                              data_keys=("part_ids", "valid_matrix"), device=cuda_device)
        np.random.seed(77)
        random.seed(77)
        batch = prod.produce(raws)
        want = {k[len(f"geo.{tag}."):]: v for k, v in z.items() if k.startswith(f"geo.{tag}.")}
        assert set(batch) == set(want)
        for k, v in want.items():
            got = batch[k].cpu().numpy()
            assert got.shape == v.shape, (k, got.shape, v.shape)
            assert got.dtype == v.dtype, (k, got.dtype, v.dtype)
            if k in ("part_pcs", "part_trans"):  # float64 pipeline cast to float32: at most the last bit moves
                np.testing.assert_allclose(got, v, rtol=0, atol=1e-6, err_msg=k)
            elif k == "part_quat":
                np.testing.assert_allclose(got, v, rtol=0, atol=1e-7, err_msg=k)
            else:
                np.testing.assert_array_equal(got, v, err_msg=k)
        # the producer's output is a valid input of the hot path: zero-centred parts, unit quaternions
        pcs = batch["part_pcs"].cpu().numpy()
        assert np.abs(pcs.mean(2)).max() < 1e-6
        vq = batch["part_quat"].norm(dim=-1)[batch["part_valids"] > 0]
        assert torch.allclose(vq, torch.ones_like(vq), atol=1e-6)


def _write_box_obj(path, lo, hi, quads=True):
    """An axis-aligned box as .obj (quad faces: exercises the fan triangulation; `v/vt/vn` index syntax on one face)."""
    (x0, y0, z0), (x1, y1, z1) = lo, hi
    v = [(x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)]
    f = [(1, 2, 3, 4), (5, 8, 7, 6), (1, 5, 6, 2), (2, 6, 7, 3), (3, 7, 8, 4), (4, 8, 5, 1)]
    with open(path, "w") as fh:
        fh.write("# box\n")
        for p in v:
            fh.write("v %r %r %r\n" % p)
        for k, q in enumerate(f):
            fh.write("f " + " ".join((f"{i}/1/1" if k == 0 else str(i)) for i in q) + "\n")


def test_obj_surface_sampler_restates_the_published_sampling(tmp_path):
    """`ObjSurfaceSampler` (the `_get_pcs` of geometry_data.py:109-131 without trimesh): sorted part files, N points per
    part ON the mesh surface, faces hit in proportion to their area, reproducible under `np.random.seed`, part-count
    limits enforced.  (trimesh itself is absent: its sampling is restated from its published algorithm, parity unpinned.)"""
    import random
    from multi_part_assembly_amd.datasets import ObjSurfaceSampler, load_obj, sample_surface
    folder = tmp_path / "plate" / "x" / "fractured_0"
    folder.mkdir(parents=True)
    _write_box_obj(folder / "piece_1.obj", (0, 0, 0), (1, 2, 4))
    _write_box_obj(folder / "piece_0.obj", (-1, -1, -1), (0, 0, 0))
    v, f = load_obj(folder / "piece_1.obj")
    assert v.shape == (8, 3) and f.shape == (12, 3)  # six quads -> twelve triangles
    np.random.seed(3)
    pts = sample_surface(v, f, 40000)
    # on the surface: at least one coordinate sits on a face plane, all inside the box
    lo, hi = np.array([0, 0, 0.0]), np.array([1, 2, 4.0])
    assert ((pts >= lo - 1e-12) & (pts <= hi + 1e-12)).all()
    on_face = (np.abs(pts - lo) < 1e-12) | (np.abs(pts - hi) < 1e-12)
    assert on_face.any(axis=1).all()
    # area-proportional: the two 2 x 4 faces (x = 0, x = 1) carry 16 of the 28 units of area, the 1 x 2 faces (z) 4
    frac_x = ((np.abs(pts[:, 0]) < 1e-12) | (np.abs(pts[:, 0] - 1) < 1e-12)).mean()
    frac_z = ((np.abs(pts[:, 2]) < 1e-12) | (np.abs(pts[:, 2] - 4) < 1e-12)).mean()
    assert abs(frac_x - 16 / 28) < 0.01 and abs(frac_z - 4 / 28) < 0.01
    # uniform inside a face: the mean of the x = 0 face's points is the face centre
    face = pts[np.abs(pts[:, 0]) < 1e-12]
    assert np.abs(face[:, 1:].mean(0) - np.array([1.0, 2.0])).max() < 0.03
    sampler = ObjSurfaceSampler(str(tmp_path), num_points=64, min_num_part=2, max_num_part=20)
    np.random.seed(11)
    a = sampler("plate/x/fractured_0")
    np.random.seed(11)
    b = sampler("plate/x/fractured_0")
    assert a.shape == (2, 64, 3) and np.array_equal(a, b)
    assert (a[0] <= 1e-12).all() and (a[1] >= -1e-12).all()  # sorted file order: piece_0 (negative octant) first
    with pytest.raises(ValueError):
        ObjSurfaceSampler(str(tmp_path), num_points=8, min_num_part=3)("plate/x/fractured_0")
    random.seed(0)
    ObjSurfaceSampler(str(tmp_path), num_points=8, shuffle_parts=True)("plate/x/fractured_0")
