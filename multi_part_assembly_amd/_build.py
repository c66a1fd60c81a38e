"""In-tree build of libmpa_hip.so (the gfx950 operator library behind include/mpa_hip.h).

`hipcc` cross-compiles for gfx950 without a GPU, so this runs in the CPU-only build container;
the resulting .so is git-ignored but travels to the GPU box with the tree.  Nothing here falls
back to anything: if hipcc is missing the build raises.
"""
from __future__ import annotations

import concurrent.futures
import json
import os
import shutil
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_PATH = PKG_DIR / "libmpa_hip.so"
OBJ_DIR = CSRC / "build"

# -ffp-contract=off: the Chamfer/transform arithmetic is pinned op by op (include/mpa_hip.h);
# kernels that want an FMA say so with __builtin_fmaf.
HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",
    "-Wall",
    "-Wno-unused-function",
]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libmpa_hip.so cannot be built (no fallback exists)")
    return exe


def sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def file_flags(src: Path) -> list[str]:
    """Extra hipcc flags a source asks for on a line `// hipcc-flags: ...` (e.g. leaf_nn.hip turns the SLP vectoriser
    off: packed v_pk_*_f32 arithmetic cannot carry the DPP operand its inner loop is built on)."""
    for line in src.read_text().splitlines()[:60]:
        if line.startswith("// hipcc-flags:"):
            return line.split(":", 1)[1].split()
    return []


_USAGE_KEYS = {"VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
               "Occupancy [waves/SIMD]": "waves_per_simd", "LDS Size [bytes/block]": "lds_bytes", "VGPRs Spill": "vgpr_spill"}


def parse_resource_usage(stderr: str) -> dict:
    """{mangled kernel name: {vgprs, agprs, scratch_bytes_per_lane, waves_per_simd, lds_bytes, vgpr_spill}} from the
    compiler's kernel-resource-usage remarks.  Written next to every object (csrc/build/<file>.usage.json) and
    checked by tests/test_abi.py: a kernel that silently starts using scratch memory (registers demoted to
    private memory by an innocent-looking edit) is a 2-3x slowdown that no correctness test sees."""
    out, cur = {}, None
    for line in stderr.splitlines():
        if "remark:" not in line:
            continue
        body = line.split("remark:", 1)[1].replace("[-Rpass-analysis=kernel-resource-usage]", "").strip()
        if body.startswith("Function Name:"):
            cur = out.setdefault(body.split(":", 1)[1].strip(), {})
        elif cur is not None and ":" in body:
            k, v = body.rsplit(":", 1)
            if k.strip() in _USAGE_KEYS:
                cur[_USAGE_KEYS[k.strip()]] = int(v)
    return out


def resource_usage() -> dict:
    """Merged usage records of the last build (empty if the objects were built by an older _build.py)."""
    merged = {}
    for f in sorted(OBJ_DIR.glob("*.usage.json")):
        merged.update(json.loads(f.read_text()))
    return merged


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every csrc/*.hip for gfx950 and link libmpa_hip.so; returns its path."""
    hipcc = _hipcc()
    OBJ_DIR.mkdir(exist_ok=True)
    headers = sorted(CSRC.glob("*.h")) + [PKG_DIR.parent / "include" / "mpa_hip.h"]
    srcs = sources()
    if not srcs:
        raise RuntimeError(f"no HIP sources under {CSRC}")
    jobs = []
    for src in srcs:
        obj = OBJ_DIR / (src.stem + ".o")
        if force or _stale(obj, [src, *headers, Path(__file__)]):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc, *HIPCC_FLAGS, *file_flags(src), "-Rpass-analysis=kernel-resource-usage", "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n{r.stdout}\n{r.stderr}")
        usage = parse_resource_usage(r.stderr)
        obj.with_suffix(".usage.json").write_text(json.dumps(usage, indent=1))
        return "\n".join(l for l in r.stderr.splitlines() if "-Rpass-analysis" not in l)

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for warn in ex.map(compile_one, jobs):
            if verbose and warn.strip():
                print(warn)

    objs = [OBJ_DIR / (s.stem + ".o") for s in srcs]
    if force or jobs or _stale(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB_PATH)] + [
            str(o) for o in objs
        ]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build(verbose=True))
