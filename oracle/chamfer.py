"""ctypes front-end of oracle/chamfer_ref.c (numpy in, numpy out).  Test infrastructure only."""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_SO = _DIR / "liboracle.so"
_lib = None


def build(force: bool = False) -> Path:
    """(Re)build liboracle.so with the committed Makefile."""
    src_m = max(p.stat().st_mtime for p in [_DIR / "chamfer_ref.c", _DIR / "knn_ref.c", _DIR / "Makefile"])
    if force or not _SO.exists() or _SO.stat().st_mtime < src_m:
        subprocess.run(["make", "-C", str(_DIR), "-B", "liboracle.so"], check=True,
                       capture_output=True)
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not _SO.exists():
            build()
        _lib = ctypes.CDLL(str(_SO))
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def chamfer_forward(xyz1, xyz2):
    """(dist1, idx1, dist2, idx2) for xyz1 [B,n1,3], xyz2 [B,n2,3] (float32 or float64)."""
    xyz1 = np.ascontiguousarray(xyz1)
    xyz2 = np.ascontiguousarray(xyz2)
    assert xyz1.dtype == xyz2.dtype and xyz1.dtype in (np.float32, np.float64)
    assert xyz1.ndim == 3 and xyz2.ndim == 3 and xyz1.shape[2] == 3 and xyz2.shape[2] == 3
    assert xyz1.shape[0] == xyz2.shape[0]
    B, n1, n2 = xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]
    dist1 = np.empty((B, n1), xyz1.dtype)
    dist2 = np.empty((B, n2), xyz1.dtype)
    idx1 = np.empty((B, n1), np.int64)
    idx2 = np.empty((B, n2), np.int64)
    fn = getattr(_load(), "oracle_chamfer_forward_" + ("f32" if xyz1.dtype == np.float32 else "f64"))
    fn.restype = None
    i64 = ctypes.c_int64
    fn(_p(xyz1), _p(xyz2), i64(B), i64(n1), i64(n2), _p(dist1), _p(idx1), _p(dist2), _p(idx2))
    return dist1, idx1, dist2, idx2


def chamfer_backward(g1, g2, xyz1, xyz2, idx1, idx2):
    """(grad_xyz1, grad_xyz2); sequential accumulation order."""
    xyz1 = np.ascontiguousarray(xyz1)
    xyz2 = np.ascontiguousarray(xyz2)
    dt = xyz1.dtype
    g1 = np.ascontiguousarray(g1, dtype=dt)
    g2 = np.ascontiguousarray(g2, dtype=dt)
    idx1 = np.ascontiguousarray(idx1, dtype=np.int64)
    idx2 = np.ascontiguousarray(idx2, dtype=np.int64)
    B, n1, n2 = xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]
    gx1 = np.empty((B, n1, 3), dt)
    gx2 = np.empty((B, n2, 3), dt)
    fn = getattr(_load(), "oracle_chamfer_backward_" + ("f32" if dt == np.float32 else "f64"))
    fn.restype = None
    i64 = ctypes.c_int64
    fn(_p(g1), _p(g2), _p(xyz1), _p(xyz2), _p(idx1), _p(idx2), i64(B), i64(n1), i64(n2), _p(gx1),
       _p(gx2))
    return gx1, gx2
