"""Build the per-kernel HBM-traffic records bench.py reads (`profiles/rNN_pmc_dominant_kernel.json`,
`profiles/rNN_pmc_knn_kernel.json`) out of the FETCH_SIZE / WRITE_SIZE summaries of tools/gpu_full_pass.sh.
    python tools/pmc_json.py profiles r02
FETCH_SIZE is doubled (gfx950 reports half of wide coalesced reads: MI355X_MICROARCH.md, HBM section); both counters are in KB."""
import json
import re
import sys
from pathlib import Path


def per_kernel(path, pattern):
    """avg/launch (KB) of the first line of a pmc_summary file whose kernel name matches `pattern`."""
    for line in Path(path).read_text().splitlines():
        m = re.match(r"\s*(\w+)\s+avg/launch\s+([\d.]+)\s+launches\s+(\d+)\s+(.*)", line)
        if m and re.search(pattern, m.group(4)):
            return float(m.group(2))
    return None


def clouds_of(prof, rnd, cfg):
    """valid parts per launch (mean over the rotated batches) of the bench line committed beside the counter passes."""
    f = prof / f"{rnd}_{cfg}_bench_line.json"
    if not f.exists():
        return None
    return json.loads(f.read_text().strip().splitlines()[-1])["config"].get("valid_parts_rank0")


def main():
    prof, rnd = Path(sys.argv[1]), sys.argv[2]
    note = ("FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md HBM section); "
            "WRITE_SIZE as reported")
    f = per_kernel(prof / f"{rnd}_c2_pmc_fetch_size.txt", r"grid_search_kernel")
    w = per_kernel(prof / f"{rnd}_c2_pmc_write_size.txt", r"grid_search_kernel")
    if f is not None and w is not None:
        rec = {"kernel": "grid_search_kernel",
               "command": "python bench.py --config c2 --no-cpu-baseline --steps 4 --warmup 2 (rocprofv3 --pmc FETCH_SIZE "
                          "and --pmc WRITE_SIZE, separate passes; tools/gpu_full_pass.sh)",
               "config": "c2", "clouds_per_launch": clouds_of(prof, rnd, "c2"),
               "fetch_size_kb_per_launch": f, "write_size_kb_per_launch": w,
               "correction": "FETCH_SIZE at face value: the kernel gathers 16-byte records, and tools/probes/gather16.hip "
                             "(profiles/r06_gather16_fetch_size.txt) shows rocprofv3 counting one 64-byte fetch per gathered "
                             "record with NO halving (a coalesced stream of the same bytes is reported at 1/2); WRITE_SIZE as "
                             "reported (scattered 4-byte result stores count a full 64-byte line each); upper bound with the "
                             "streaming correction (FETCH x 2): traffic_bytes_per_launch_upper",
               "traffic_bytes_per_launch": (1.0 * f + w) * 1024.0,
               "traffic_bytes_per_launch_upper": (2.0 * f + w) * 1024.0}
        (prof / f"{rnd}_pmc_dominant_kernel.json").write_text(json.dumps(rec, indent=1))
        print(rec)
    # the wide kNN stages are a pipeline of kernels (csrc/dg_knn_fast.h): sum them per feature width; one record per
    # workload (c3: everyday-like, c5: artifact-like clouds — different cloud counts per launch)
    for cfg, out in (("c3", f"{rnd}_pmc_knn_kernel.json"), ("c5", f"{rnd}_pmc_knn_kernel_c5.json")):
        ff, wf = prof / f"{rnd}_{cfg}_pmc_fetch_size.txt", prof / f"{rnd}_{cfg}_pmc_write_size.txt"
        if not (ff.exists() and wf.exists()):
            continue
        names = []
        for ln in ff.read_text().splitlines():
            m = re.match(r"\s*\w+\s+avg/launch\s+[\d.]+\s+launches\s+\d+\s+(?:void\s+)?(.*)", ln)
            if m and ("knn_" in m.group(1) or "rownorm" in m.group(1)) and "knn3" not in m.group(1):
                names.append(m.group(1).split("(")[0].strip())
        widths = {}
        for C in (128, 64):
            pats = sorted({re.escape(n) for n in names if re.search(rf"<{C}[,>]", n)})
            if not pats:
                continue
            fs = [per_kernel(ff, pat) for pat in pats]
            ws = [per_kernel(wf, pat) for pat in pats]
            missing = [pat for pat, a, b in zip(pats, fs, ws) if a is None or b is None]
            fs = [0.0 if v is None else v for v in fs]  # a kernel below the summary's cut-off moved (almost) nothing
            ws = [0.0 if v is None else v for v in ws]
            # round 5b: the pass that writes the stage's output (dg_apply_knn_kernel<C>) also writes the search's operands
            # (bf16 rows, norms).  Its reads of the pre-activation rows and its write of the output exist with or without
            # a kNN search behind it; what the search adds is what the pass writes BEYOND the output it reads in:
            # WRITE_SIZE - 2 x FETCH_SIZE (input and output have the same element count).
            pf = per_kernel(ff, re.escape(f"dg_apply_knn_kernel<{C}>"))
            pw = per_kernel(wf, re.escape(f"dg_apply_knn_kernel<{C}>"))
            extra = max(0.0, pw - 2.0 * pf) if pf is not None and pw is not None else 0.0
            widths[str(C)] = {"fetch_size_kb_per_launch": sum(fs), "write_size_kb_per_launch": sum(ws),
                              "per_kernel_fetch_kb": dict(zip(pats, fs)), "per_kernel_write_kb": dict(zip(pats, ws)),
                              "below_the_summary_cutoff": missing,
                              "producer_pass": None if pf is None else {
                                  "kernel": f"dg_apply_knn_kernel<{C}>", "fetch_size_kb_per_launch": pf,
                                  "write_size_kb_per_launch": pw, "operand_bytes_written_for_the_search_kb": extra},
                              "traffic_bytes_per_launch": (2.0 * sum(fs) + sum(ws) + extra) * 1024.0}
        if widths:
            rec = {"kernel": "dg::knn_wide = every kernel of one wide kNN search (names in per_kernel_*)",
                   "config": cfg, "clouds_per_launch": clouds_of(prof, rnd, cfg),
                   "command": f"python bench.py --config {cfg} --no-cpu-baseline --steps 4 --warmup 2 (rocprofv3 --pmc "
                              "FETCH_SIZE and --pmc WRITE_SIZE, separate passes; tools/gpu_full_pass.sh)",
                   "correction": note, "per_width": widths,
                   "note": "sum over the kernels of one search (+ the operand bytes the producing pass writes for it, "
                           "producer_pass); the query blocks of a cloud run on one XCD (dg_knn.h: knn_block), so one L2 "
                           "streams the cloud's split features in both Gram passes"}
            (prof / out).write_text(json.dumps(rec, indent=1))
            print(rec)


if __name__ == "__main__":
    main()
