"""ctypes binding of libmpa_hip.so — the only way the Python host reaches the HIP kernels.

There is deliberately no fallback: if the shared library is missing or fails to load, every
operator raises.  `import torch` happens first so that the library resolves `libamdhip64.so.7`
to the HIP runtime torch already loaded (one runtime per process, shared streams and pointers).
"""
from __future__ import annotations

import ctypes
import re
from pathlib import Path

import torch  # noqa: F401  (must precede the dlopen, see module docstring)

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "libmpa_hip.so"
HEADER_PATH = PKG_DIR.parent / "include" / "mpa_hip.h"

_c = ctypes
_P = _c.c_void_p
_I64 = _c.c_int64
_INT = _c.c_int
_F32 = _c.c_float
_U64 = _c.c_uint64

# name -> (restype, argtypes); must list every function include/mpa_hip.h declares
# (tests/test_abi.py cross-checks this table against the header and the .so's dynamic symbols).
SIGNATURES: dict[str, tuple] = {
    "mpa_abi_version": (_INT, []),
    "mpa_last_error": (_c.c_char_p, []),
    "mpa_chamfer_workspace": (_INT, [_I64, _I64, _I64, _P]),
    "mpa_chamfer_workspace_variant": (_INT, [_I64, _I64, _I64, _INT, _P]),
    "mpa_chamfer_forward": (_INT, [_P, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _I64, _P]),
    "mpa_chamfer_forward_variant": (_INT, [_P, _P, _I64, _I64, _I64, _P, _P, _P, _P, _INT, _P, _I64, _P]),
    "mpa_chamfer_backward": (_INT, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _P]),
    "mpa_chamfer_forward_f64": (_INT, [_P, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P]),
    "mpa_chamfer_backward_f64": (_INT, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _P]),
    "mpa_pose_apply_forward": (_INT, [_P, _P, _P, _P, _F32, _I64, _I64, _P, _P]),
    "mpa_pose_apply_backward": (_INT, [_P, _P, _P, _P, _F32, _I64, _I64, _P, _P, _P, _P]),
    "mpa_assembly_loss_workspace": (_INT, [_I64, _I64, _I64, _P, _P]),
    "mpa_linear_sum_assignment": (_INT, [_P, _P, _I64, _I64, _P, _P]),
    "mpa_match_parts": (_INT, [_P] * 7 + [_I64] * 5 + [_P] * 6),
    "mpa_quat_sanitize": (_INT, [_P, _I64, _P, _P, _P]),
    "mpa_loss_reduce_forward": (_INT, [_P, _P, _I64, _I64, _P, _P, _P]),
    "mpa_loss_reduce_backward": (_INT, [_P, _P, _P, _I64, _I64, _P, _P]),
    "mpa_part_batch_transform": (_INT, [_P] * 4 + [_I64, _I64, _P, _P, _P]),
    "mpa_assembly_loss_forward": (_INT, [_P] * 6 + [_I64, _I64, _I64, _INT, _INT, _P, _P, _P, _P]),
    "mpa_assembly_loss_forward_timed": (_INT, [_P] * 6 + [_I64, _I64, _I64, _INT, _INT, _P, _P, _P, _P, _P]),
    "mpa_assembly_loss_backward": (_INT, [_P] * 7 + [_I64, _I64, _I64, _INT, _P, _P, _P, _P, _P]),
    "mpa_assembly_order_elems": (_INT, [_I64, _I64, _I64, _P]),
    "mpa_assembly_order": (_INT, [_P, _P, _I64, _I64, _I64, _P, _P]),
    "mpa_assembly_loss_forward_ordered": (_INT, [_P] * 6 + [_I64, _I64, _I64, _INT, _INT, _P, _INT, _P, _P, _P, _P, _P]),
    "mpa_pointnet_workspace": (_INT, [_I64, _I64, _I64, _P, _P]),
    "mpa_pointnet_forward": (_INT, [_P] * 7 + [_INT, _F32, _F32, _I64, _I64, _I64, _P, _P, _P, _P]),
    "mpa_pointnet_backward": (_INT, [_P] * 5 + [_I64, _I64, _I64] + [_P] * 6),
    "mpa_pointnet_workspace_bf16": (_INT, [_I64, _I64, _I64, _P]),
    "mpa_pointnet_forward_bf16": (_INT, [_P] * 7 + [_INT, _F32, _F32, _I64, _I64, _I64, _P, _P, _P]),
    "mpa_pointnet_backward_bf16": (_INT, [_P] * 5 + [_I64, _I64, _I64] + [_P] * 5),
    "mpa_dgcnn_workspace": (_INT, [_I64, _I64, _I64, _P]),
    "mpa_dgcnn_forward": (_INT, [_P] * 9 + [_INT, _F32, _F32, _I64, _I64, _I64, _P, _P, _P, _P]),
    "mpa_dgcnn_forward_graphs": (_INT, [_P] * 9 + [_INT, _F32, _F32, _I64, _I64, _I64, _P, _P, _P, _P]),
    "mpa_dgcnn_export_graph": (_INT, [_P, _I64, _I64, _I64, _I64, _P, _P]),
    "mpa_dgcnn_backward": (_INT, [_P] * 4 + [_I64, _I64, _I64] + [_P] * 8),
    "mpa_knn_exact_workspace": (_INT, [_I64, _I64, _P]),
    "mpa_knn_exact": (_INT, [_P, _I64, _I64, _I64, _I64, _P, _P, _P]),
    "mpa_mlp_layer_workspace": (_INT, [_I64, _I64, _I64, _P]),
    "mpa_mlp_layer_forward": (_INT, [_P, _I64, _P, _P, _P, _P, _P, _P, _INT, _F32, _F32, _INT, _I64, _I64, _I64, _P, _P, _P]),
    "mpa_mlp_layer_backward": (_INT, [_P, _P, _I64, _P, _P, _P, _INT, _I64, _I64, _I64] + [_P] * 7),
    "mpa_pair_layer_workspace": (_INT, [_I64, _I64, _I64, _I64, _P]),
    "mpa_pair_layer_forward": (_INT, [_P] * 8 + [_INT, _F32, _F32, _INT, _I64, _I64, _I64, _I64, _P, _P, _P]),
    "mpa_pair_layer_backward": (_INT, [_P] * 6 + [_INT, _I64, _I64, _I64, _I64] + [_P] * 8),
    "mpa_pair_rows_forward": (_INT, [_P, _P, _I64, _I64, _I64, _INT, _P, _P]),
    "mpa_pair_rows_backward": (_INT, [_P, _I64, _I64, _I64, _INT, _P, _P, _P]),
    "mpa_narrow_linear_relu_forward": (_INT, [_P, _P, _P, _I64, _I64, _I64, _P, _P]),
    "mpa_narrow_linear_relu_workspace": (_INT, [_I64, _I64, _I64, _P]),
    "mpa_narrow_linear_relu_backward": (_INT, [_P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P]),
    "mpa_relation_head_workspace": (_INT, [_I64, _I64, _P]),
    "mpa_relation_head_forward": (_INT, [_P, _P, _P, _P, _I64, _I64, _P, _P, _P]),
    "mpa_relation_head_backward": (_INT, [_P, _P, _P, _P, _I64, _I64, _P, _P, _P, _P, _P]),
    "mpa_relation_mean_forward": (_INT, [_P, _P, _I64, _I64, _I64, _P, _P]),
    "mpa_relation_mean_backward": (_INT, [_P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _P]),
    "mpa_gru_workspace": (_INT, [_I64, _I64, _I64, _I64, _P]),
    "mpa_gru_resident": (_INT, [_I64, _I64, _I64, _P]),
    "mpa_gru_forward": (_INT, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, _P, _P, _P]),
    "mpa_gru_backward": (_INT, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P]),
    "mpa_debug_occupy": (_INT, [_I64, _I64, _I64, _P]),
    "mpa_transformer_workspace": (_INT, [_I64] * 6 + [_P]),
    "mpa_transformer_forward": (_INT, [_P, _P, _P] + [_I64] * 6 + [_F32, _U64, _P, _P, _P, _P]),
    "mpa_transformer_backward": (_INT, [_P, _P, _P] + [_I64] * 6 + [_F32, _U64, _P, _P, _P, _P, _P]),
    "mpa_pose_head_workspace": (_INT, [_I64, _I64, _P]),
    "mpa_pose_head_forward": (_INT, [_P, _P, _I64, _I64, _P, _P, _P, _P]),
    "mpa_pose_head_backward": (_INT, [_P, _P, _P, _P, _I64, _I64, _P, _P, _P, _P]),
    "mpa_adam_step": (_INT, [_P, _P, _P, _P, _I64, _F32, _F32, _F32, _F32, _F32, _INT, _I64, _F32, _P, _P]),
    "mpa_adam_step_dev": (_INT, [_P, _P, _P, _P, _I64, _P, _F32, _F32, _F32, _F32, _INT, _P, _P]),
    "mpa_grad_clip_workspace": (_INT, [_P]),
    "mpa_grad_clip_coef": (_INT, [_P, _I64, _F32, _P, _F32, _P, _P, _P]),
}

ABI_VERSION = 8
_lib = None


class MpaError(RuntimeError):
    """A libmpa_hip.so entry point returned a nonzero status."""


def declared_functions() -> list[str]:
    """Function names declared in include/mpa_hip.h."""
    text = HEADER_PATH.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpa_[a-z0-9_]+)\s*\(", text)))


def lib() -> ctypes.CDLL:
    """Load (once) and return the operator library; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise MpaError(
            f"{LIB_PATH} is missing — build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (there is no non-HIP fallback for these operators)"
        )
    handle = ctypes.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    got = handle.mpa_abi_version()
    if got != ABI_VERSION:
        raise MpaError(f"libmpa_hip.so ABI version {got}, python host expects {ABI_VERSION}")
    _lib = handle
    return handle


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().mpa_last_error().decode("utf-8", "replace")
        raise MpaError(f"{what} failed (status {status}): {msg}")


def ptr_array(tensors) -> ctypes.Array:
    """Host array of raw device pointers (for the `const float* const*` parameters)."""
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def ptr(t: torch.Tensor | None) -> int | None:
    """Raw device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream(device: torch.device) -> int:
    """hipStream_t of torch's current stream on `device`, as an integer handle."""
    return torch.cuda.current_stream(device).cuda_stream


class KernelTimer:
    """Optional per-launch HIP-event timing of named kernels, on the stream they are launched on.

    bench.py sets `KernelTimer.active = KernelTimer()` around its timed region; the operator
    wrappers then bracket their launches with torch.cuda.Event records (torch events are recorded
    on torch's current stream, which is the stream the launch uses).  Inactive (None) by default:
    zero overhead on the training path.
    """

    active: "KernelTimer | None" = None

    def __init__(self, only=None):
        """`only`: name prefixes to time (None: every instrumented launch).  bench.py times just the roofline kernel
        inside its timed region — two event records per step instead of ~60 — and the rest in a second pass."""
        self.events: dict[str, list] = {}
        self.only = None if only is None else tuple(only)

    @classmethod
    def wants(cls, name: str) -> bool:
        a = cls.active
        return a is not None and (a.only is None or name.startswith(a.only))

    @classmethod
    def start(cls, name: str):
        if not cls.wants(name):
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return (name, ev)

    @classmethod
    def stop(cls, token) -> None:
        if token is None or cls.active is None:
            return
        end = torch.cuda.Event(enable_timing=True)
        end.record()
        cls.active.events.setdefault(token[0], []).append((token[1], end))

    @classmethod
    def phase_events(cls, names):
        """Timing events for a library call that records its own phase boundaries: phase i runs from event i to event
        i + 1 and is called names[i].  Returns a list of len(names) + 1 events with None where no wanted phase touches
        the boundary (the library skips null entries), or None when nothing is wanted."""
        want = [cls.wants(n) for n in names]
        if not any(want):
            return None
        evs = []
        for i in range(len(names) + 1):
            need = (i < len(names) and want[i]) or (i > 0 and want[i - 1])
            ev = torch.cuda.Event(enable_timing=True) if need else None
            if ev is not None:
                ev.record()  # materialise the hipEvent_t handle; the library re-records it at the right place
            evs.append(ev)
        return evs

    @classmethod
    def add_phases(cls, names, evs) -> None:
        if evs is None or cls.active is None:
            return
        for i, name in enumerate(names):
            if evs[i] is not None and evs[i + 1] is not None and cls.wants(name):
                cls.active.events.setdefault(name, []).append((evs[i], evs[i + 1]))

    @staticmethod
    def handles(evs):
        """ctypes array of the hipEvent_t handles (null where the event is None), or None."""
        if evs is None:
            return None
        return (ctypes.c_void_p * len(evs))(*[None if e is None else e.cuda_event for e in evs])

    def summary(self) -> dict[str, dict]:
        """name -> {launches, avg_ms, total_ms}; call after torch.cuda.synchronize()."""
        out = {}
        for name, pairs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in pairs]
            out[name] = {"launches": len(ms), "avg_ms": sum(ms) / len(ms), "total_ms": sum(ms)}
        return out
