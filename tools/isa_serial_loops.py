"""Flag the latency-serial loops of a gfx950 assembly listing (hipcc -S --cuda-device-only): innermost loops (a label
and a backward branch to it with no other label between) whose body issues only a few global loads and then waits for
all of them (s_waitcnt vmcnt(0)).  One wave then has a single row of loads in flight per trip: with ~2 us of HBM
latency a streaming kernel built this way stops at ~2 TB/s however many waves the chip holds.
   python tools/isa_serial_loops.py file.s [file.s ...] [--max-loads 4]"""
import re
import sys


def scan(path, max_loads):
    kernel = None
    out = []
    lines = open(path).read().splitlines()
    label_at = {}
    body_start = None
    cur_label = None
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kernel = m.group(1)
            cur_label = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            cur_label = m.group(1)
            body_start = i
            continue
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln)
        if m and cur_label is not None and m.group(1) == cur_label:
            body = lines[body_start:i]
            loads = sum(1 for b in body if re.search(r"\b(global_load|buffer_load|flat_load)", b))
            stores = sum(1 for b in body if re.search(r"\b(global_store|buffer_store|flat_store)", b))
            lds = sum(1 for b in body if re.search(r"\bds_(read|write|load|store)", b))
            mfma = sum(1 for b in body if "mfma" in b)
            wait0 = any(re.search(r"s_waitcnt.*vmcnt\(0\)", b) for b in body)
            if 1 <= loads <= max_loads and wait0 and mfma == 0:
                out.append((kernel, cur_label, loads, stores, lds, len(body)))
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    max_loads = 4
    if "--max-loads" in sys.argv:
        max_loads = int(sys.argv[sys.argv.index("--max-loads") + 1])
        args.remove(str(max_loads))
    import subprocess
    for p in args:
        for kernel, label, loads, stores, lds, n in scan(p, max_loads):
            name = subprocess.run(["c++filt", kernel], capture_output=True, text=True).stdout.strip()
            print(f"{p.split('/')[-1]:18s} {label:12s} loads {loads} stores {stores} lds {lds} instr {n:4d}  {name[:110]}")


if __name__ == "__main__":
    main()
