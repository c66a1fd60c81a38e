"""Dev helper: per-kernel-name launch counts and average / minimum durations out of a rocprofv3 results database."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = [dict(zip(cols, r)) for r in cur.execute("select * from kernels order by start")]
agg = {}
for d in rows:
    n = d["name"]
    if "pb_" in n or "gemm_tn_reduce" in n or "pn_" in n:
        m = re.search(r"((?:pb|pn|dg)_\w+|gemm_tn_reduce_kernel)(<[^>]*>)?", n)
        k = (m.group(0) if m else n)[:60]
        a = agg.setdefault(k, [0, 0.0, 1e9]); a[0] += 1; a[1] += (d["end"] - d["start"]) / 1e3; a[2] = min(a[2], (d["end"] - d["start"]) / 1e3)
for k, (c, t, mn) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:62s} calls {c:4d}  avg {t / c:8.1f} us  min {mn:8.1f}")
