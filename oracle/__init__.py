"""CPU oracle for the multi-part-assembly hot path — TEST INFRASTRUCTURE, not product code.

Restates, on the CPU, the reference algorithm of every operator the HIP library implements, each
function citing the reference file:line it follows.  Only `tests/`, `__graft_entry__.smoke()` and
the `cpu_baseline` leg of `bench.py` may import this package, and only as the checker / the timed
baseline — never as the thing shipped: nothing under `multi_part_assembly_amd/` imports it.

Pinning status (details in DESIGN.md §Oracle):
  * chamfer (chamfer_ref.c)        — pinned against tests/golden/chamfer_*.npz, generated from the
                                     reference's own brute-force definition (test_chamfer.py:8-31).
"""
