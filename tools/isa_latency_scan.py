#!/usr/bin/env python3
"""Per kernel of one .s listing (hipcc --cuda-device-only -S): counts that point at exposed latency in the small, launch-bound
kernels — full `s_waitcnt vmcnt(0)` waits (each can be a dependent memory round trip), LDS cross-lane permutes
(`ds_bpermute`: ~100 cycles each in a reduction chain; DPP does the same in the VALU), branches.
usage: tools/isa_latency_scan.py listing.s [name filter]"""
import re
import sys

text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):", text, re.M)]
for i, (pos, name) in enumerate(starts):
    end = text.find("s_endpgm", pos)
    if end < 0 or (i + 1 < len(starts) and end > starts[i + 1][0]):
        continue
    if flt and flt not in name:
        continue
    lines = text[pos:end].split("\n")
    cnt = lambda pat: sum(1 for l in lines if re.search(pat, l))
    loads, waits = cnt(r"(global|buffer)_load"), cnt(r"s_waitcnt vmcnt[(]0[)]")
    perm, br = cnt(r"ds_bpermute|ds_swizzle"), cnt(r"s_cbranch")
    print(f"{name[:64]:64s} vmem loads {loads:4d}  vmcnt(0) {waits:3d}  bpermute {perm:3d}  branches {br:3d}  instr {len(lines)}")
