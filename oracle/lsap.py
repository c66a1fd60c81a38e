"""TEST INFRASTRUCTURE ONLY — CPU restatement of scipy.optimize.linear_sum_assignment (square case).

The reference calls scipy on the host (multi_part_assembly/models/modules/base_model.py:6,175); scipy is a third-party
dependency that is not vendored under /root/reference (unpinned in its setup.py; 1.15.3 is installed here).  Its
solver is the shortest-augmenting-path algorithm of D. F. Crouse, "On implementing 2D rectangular assignment
algorithms", IEEE TAES 52(4), 2016 (scipy/optimize/rectangular_lsap/rectangular_lsap.cpp).  This file restates that
algorithm step by step — including the reversed `remaining` list and the "prefer a new sink among equal costs" rule
that decide ties — so that `csrc/match.hip:lsap_solve` can be checked line against line; the restatement itself is
pinned against scipy (tests/test_match.py), which is importable on both boxes.
"""
from __future__ import annotations

import numpy as np


def linear_sum_assignment_square(cost) -> np.ndarray:
    """col4row of the minimum-cost perfect matching of a square matrix (float64 arithmetic, as scipy)."""
    c = np.asarray(cost, dtype=np.float64)
    n = c.shape[0]
    assert c.shape == (n, n)
    u = np.zeros(n)
    v = np.zeros(n)
    path = np.full(n, -1, dtype=np.int64)
    col4row = np.full(n, -1, dtype=np.int64)
    row4col = np.full(n, -1, dtype=np.int64)
    for cur in range(n):
        min_val = 0.0
        remaining = [n - it - 1 for it in range(n)]
        num_remaining = n
        SR = np.zeros(n, dtype=bool)
        SC = np.zeros(n, dtype=bool)
        spc = np.full(n, np.inf)
        sink, i = -1, cur
        while sink == -1:
            index, lowest = -1, np.inf
            SR[i] = True
            for it in range(num_remaining):
                j = remaining[it]
                r = min_val + c[i, j] - u[i] - v[j]
                if r < spc[j]:
                    path[j] = i
                    spc[j] = r
                if spc[j] < lowest or (spc[j] == lowest and row4col[j] == -1):
                    lowest = spc[j]
                    index = it
            min_val = lowest
            if min_val == np.inf:
                raise ValueError("cost matrix is infeasible")
            j = remaining[index]
            if row4col[j] == -1:
                sink = j
            else:
                i = row4col[j]
            SC[j] = True
            num_remaining -= 1
            remaining[index] = remaining[num_remaining]
        u[cur] += min_val
        for r in range(n):
            if SR[r] and r != cur:
                u[r] += min_val - spc[col4row[r]]
        for j in range(n):
            if SC[j]:
                v[j] -= min_val - spc[j]
        j = sink
        while True:
            r = path[j]
            row4col[j] = r
            col4row[r], j = j, col4row[r]
            if r == cur:
                break
    return col4row
