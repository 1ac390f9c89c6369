#!/usr/bin/env python
"""Workload for the rocprofv3 kernel trace of the small-LP path: 20 cuOptSolve calls on the 50v-10 relaxation
(resident single-workgroup loop + fused major-iteration kernel).  GPU only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import decode_problem  # noqa: E402
from cuopt_amd import capi  # noqa: E402

g = json.load(open(os.path.join(ROOT, "tests", "golden", "problems.json")))
p = decode_problem(g["mip-50v-10-free-bound-relaxation"])
p.pop("var_types", None)
for _ in range(20):
    r = capi.solve(p, method=1, tol=1e-4)
    assert r["status"] == "Optimal"
print("20 solves,", r["steps_taken"], "iterations each, objective", r["objective"])
