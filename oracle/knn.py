"""ctypes front-end of oracle/knn_ref.c — the kNN graph of DGCNN with pinned score arithmetic.  Test infrastructure."""
from __future__ import annotations

import ctypes

import numpy as np

from . import chamfer as _oc


def knn_exact(x: np.ndarray, k: int = 20) -> np.ndarray:
    """x [n, N, C] float32 -> int32 [n, N, k], best first, ties to the lower index.  C = 3 uses the reference's CPU
    arithmetic (mode 0), wider features the matrix-core chain order (mode 1); see knn_ref.c."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, N, C = x.shape
    assert C == 3 or C % 2 == 0
    idx = np.empty((n, N, k), np.int32)
    fn = _oc._load().oracle_knn
    fn.restype = None
    i64 = ctypes.c_int64
    fn(x.ctypes.data_as(ctypes.c_void_p), i64(n), i64(N), i64(C), i64(C), i64(k), ctypes.c_int32(0 if C == 3 else 1),
       idx.ctypes.data_as(ctypes.c_void_p))
    return idx
