"""A fixed handful of cases of every family of tools/fuzz_parity.py (randomised differential checks: index-exact kernels
against the C oracles, modules and whole training steps against the float64 oracle, graph replay against eager launches,
bit-reproducibility) under the GPU test run; the long runs are recorded in profiles/r03_fuzz_parity.txt."""
import importlib.util
import pathlib

import pytest

pytestmark = pytest.mark.gpu

FAMILIES = {"loss": 4, "chamfer": 4, "knn": 4, "glue": 6, "repro": 6, "nets": 4, "dgcnn": 3, "step": 3, "gnn": 2, "global": 3,
            "adam": 6, "graph": 2, "cgrid": 12}


@pytest.fixture(scope="module")
def fuzz(cuda_device):
    path = pathlib.Path(__file__).resolve().parents[1] / "tools" / "fuzz_parity.py"
    spec = importlib.util.spec_from_file_location("fuzz_parity", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_fuzz_family(fuzz, family):
    assert family in [k for k, _ in fuzz.families]
    for seed in range(7000, 7000 + FAMILIES[family]):
        ok, what = fuzz.run_case(family, seed)
        assert ok, (family, seed, what)
