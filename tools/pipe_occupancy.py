#!/usr/bin/env python3
"""Matrix-pipe occupancy table of a workload's largest matrix-core kernels out of two committed files:
    python tools/pipe_occupancy.py profiles/r06_c2_steady_state_per_step.txt profiles/r06_c2_pmc_sq_counters.txt
SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's 1024 SIMDs) / (1024 SIMDs x average launch duration x 2.4 GHz)."""
import re
import sys

steady, pmc = sys.argv[1], sys.argv[2]
dur = {}
for line in open(steady):
    m = re.match(r"\s*[\d.]+\s+[\d.]+\s+([\d.]+)\s+(.*)", line)
    if m:
        dur[m.group(2).strip()[:70]] = float(m.group(1))
busy, valu = {}, {}
for line in open(pmc):
    m = re.match(r"\s*(\w+)\s+avg/launch\s+([\d.]+)\s+launches\s+\d+\s+(.*)", line)
    if not m:
        continue
    key = m.group(3).strip()[:70]
    if m.group(1) == "SQ_VALU_MFMA_BUSY_CYCLES":
        busy[key] = float(m.group(2))
    if m.group(1) == "SQ_INSTS_VALU":
        valu[key] = float(m.group(2))
print("Matrix-pipe occupancy of the largest matrix-core kernels (SQ_VALU_MFMA_BUSY_CYCLES summed over the chip's 1024 SIMDs; 32 per")
print("v_mfma_f32_32x32x16_bf16, 64 per v_mfma_f32_32x32x2_f32) / (1024 SIMDs x average launch duration x 2.4 GHz).  Durations: " + steady)
print("(rocprofv3 --kernel-trace); counters: " + pmc + " (separate pass; the profiled clock is ~1.9-2.1 GHz, so occupancy in the")
print("kernel's own cycles is ~15-20 % higher than the figure at the 2.4 GHz peak).\n")
print("us/launch  MFMA busy cyc  pipe occ @2.4GHz   VALU instr  kernel")
rows = []
for k, b in busy.items():
    d = next((v for kk, v in dur.items() if kk[:60] == k[:60]), None)
    if d is None or b <= 0:
        continue
    rows.append((b, d, k))
for b, d, k in sorted(rows, reverse=True)[:14]:
    print(f"{d:9.1f} {b:14.0f} {b / (1024 * d * 1e-6 * 2.4e9):17.1%} {valu.get(k, 0):12.0f}  {k}")
