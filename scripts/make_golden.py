#!/usr/bin/env python
"""Generates tests/golden/*.json.  RUN IN THE BUILD CONTAINER ONLY (needs /root/reference and the
compiled-in-place reference library oracle/_ref/libcuopt_ref.so, see oracle/Makefile).

What is recorded, and where it comes from:
  problems.json  : small LPs parsed BY THE REFERENCE's own libmps_parser (free format, the mode
                   cuOptReadProblem uses) + the objective of THE REFERENCE's own CPU dual simplex on
                   them + the known answers pinned in the reference's tests (cited per entry) + the
                   results of our C oracle (iterations, objective, initial step size / primal
                   weight) so the GPU tests can run without /root/reference.
  mps_parser.json: for every file under datasets/linear_programming: what the reference parser
                   returns (sizes, CSR, bounds, objective ...) or that it rejects the file.
Nothing but numbers derived by running reference code on its public fixtures is stored; no
reference source is copied."""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orcbind, refbind  # noqa: E402

REF = "/root/reference/datasets"
OUT = os.path.join(ROOT, "tests", "golden")


def enc(a):
    a = np.asarray(a)
    if a.dtype.kind == "f":
        return [("inf" if v == np.inf else "-inf" if v == -np.inf else float(v)) for v in a.tolist()]
    return a.tolist()


def problem_entry(path, relax=True, pinned=None, source=None):
    p = refbind.parse_mps(path, fixed_format=False)
    d = dict(file=os.path.relpath(path, REF), m=p["m"], n=p["n"], nnz=p["nnz"],
             maximize=p["maximize"], objective_offset=p["objective_offset"],
             offsets=enc(p["offsets"]), indices=enc(p["indices"]), values=enc(p["values"]),
             c=enc(p["c"]), lo=enc(p["lo"]), hi=enc(p["hi"]), lb=enc(p["lb"]), ub=enc(p["ub"]),
             var_types=enc(p["var_types"]), row_names=p["row_names"], var_names=p["var_names"],
             objective_name=p["objective_name"], problem_name=p["problem_name"])
    if p["flags"] & 2:
        d["row_types"] = enc(p["row_types"])
        d["rhs"] = enc(p["rhs"])
    ds = refbind.dual_simplex(p)
    d["reference_dual_simplex"] = dict(status=ds["status"], objective=ds["objective"],
                                       iterations=ds["iterations"])
    if pinned is not None:
        d["pinned_objective"] = pinned
        d["pinned_source"] = source
    orc = {}
    for tol in (1e-4, 1e-8):
        # bounded: PDLP (reference rule, pdlp_restart_strategy.cu:684-750) can stall at 1e-8 on tiny degenerate LPs
        # (minrep_inf: the primal weight collapses); the recorded status says so
        s = orcbind.solve(p, tol=tol, num_threads=1, iteration_limit=200000)
        orc["%g" % tol] = {k: s[k] for k in ("status", "steps_taken", "attempted_steps",
                                             "primal_objective", "dual_objective", "gap",
                                             "l2_primal_residual", "l2_dual_residual",
                                             "initial_step_size", "initial_primal_weight",
                                             "num_restarts", "returned_average")}
    d["oracle"] = orc
    return d


def parser_goldens(lp):
    parsed = {}
    for path in sorted(glob.glob(os.path.join(lp, "*.mps"))):
        name = os.path.basename(path)
        try:
            p = refbind.parse_mps(path, fixed_format=False)
            parsed[name] = dict(ok=True, m=p["m"], n=p["n"], nnz=p["nnz"], maximize=p["maximize"],
                                objective_offset=p["objective_offset"], offsets=enc(p["offsets"]),
                                indices=enc(p["indices"]), values=enc(p["values"]), c=enc(p["c"]),
                                lo=enc(p["lo"]), hi=enc(p["hi"]), lb=enc(p["lb"]), ub=enc(p["ub"]),
                                var_types=enc(p["var_types"]), row_names=p["row_names"],
                                var_names=p["var_names"])
            if p["m"] > 0:  # what the reference's own CPU dual simplex says about the LP in the file
                ds = refbind.dual_simplex(p)
                obj = ds["objective"]
                parsed[name]["reference_dual_simplex"] = dict(
                    status=ds["status"], iterations=ds["iterations"],
                    objective=obj if np.isfinite(obj) else None)
        except refbind.RefMpsError as e:
            parsed[name] = dict(ok=False, error=str(e)[:200])
    json.dump(parsed, open(os.path.join(OUT, "mps_parser.json"), "w"), indent=0)
    return parsed


def main():
    os.makedirs(OUT, exist_ok=True)
    only_names = set(filter(None, os.environ.get("GOLDEN_ONLY", "").split(",")))
    if os.environ.get("GOLDEN_PARSER_ONLY"):  # mps_parser.json alone
        parser_goldens(os.path.join(REF, "linear_programming"))
        return
    lp = os.path.join(REF, "linear_programming")
    mip = os.path.join(REF, "mip")
    specs = [
        ("afiro", os.path.join(lp, "afiro_original.mps"),
         dict(pinned=-464.7531, source="python/cuopt/cuopt/tests/linear_programming/test_lp_solver.py:119; "
                                       "cpp/tests/linear_programming/pdlp_test.cu:58-84")),
        ("good-max", os.path.join(lp, "good-max.mps"),
         dict(pinned=17.0, source="cpp/tests/linear_programming/pdlp_test.cu:909-925")),
        ("max_offset", os.path.join(lp, "max_offset.mps"),
         dict(pinned=0.0, source="cpp/tests/linear_programming/pdlp_test.cu:927-943")),
        ("good-mps-1", os.path.join(lp, "good-mps-1.mps"), {}),
        ("lp_model_with_var_bounds", os.path.join(lp, "lp_model_with_var_bounds.mps"), {}),
        ("mip-sample-relaxation", os.path.join(mip, "sample.mps"), {}),
        ("mip-bb_optimality-relaxation", os.path.join(mip, "bb_optimality.mps"), {}),
        # BASELINE config 5 inputs (LP relaxations of datasets/mip; integrality dropped by the tests)
        ("mip-50v-10-free-bound-relaxation", os.path.join(mip, "50v-10-free-bound.mps"), {}),
        ("mip-neos5-free-bound-relaxation", os.path.join(mip, "neos5-free-bound.mps"), {}),
        ("mip-sudoku-relaxation", os.path.join(mip, "sudoku.mps"), {}),
        # the reference's dual simplex needs ~5 minutes on this degenerate relaxation (6347 pivots)
        ("mip-cod105_max-relaxation", os.path.join(mip, "cod105_max.mps"), {}),
        # small LP-shaped fixtures of the MIP tests (cpp/tests/mip/empty_fixed_problems_test.cu:65,
        # termination_test.cu:68, feasibility_jump_tests.cu:273): fixed variables, a trivial row, a tiny relaxation
        ("mip-fixed-problem-relaxation", os.path.join(mip, "fixed-problem.mps"), {}),
        ("mip-trivial-presolve-optimality-relaxation", os.path.join(mip, "trivial-presolve-optimality.mps"), {}),
        ("mip-minrep_inf-relaxation", os.path.join(mip, "minrep_inf.mps"), {}),
    ]
    # GOLDEN_ONLY=name1,name2: regenerate these entries only and keep the rest of problems.json as it is
    problems = json.load(open(os.path.join(OUT, "problems.json"))) if only_names else {}
    for name, path, kw in specs:
        if only_names and name not in only_names:
            continue
        problems[name] = problem_entry(path, **kw)
        print("  ", name, problems[name]["reference_dual_simplex"], flush=True)
    if only_names:
        json.dump(problems, open(os.path.join(OUT, "problems.json"), "w"), indent=0)
        return
    # goldens of the reference's initial-solution test (afiro, Methodical1): step size / primal weight
    afiro = refbind.parse_mps(os.path.join(lp, "afiro_original.mps"))
    s = orcbind.solve(afiro, mode=1, iteration_limit=0)
    problems["afiro"]["pinned_initial"] = dict(
        step_size=1.4893, primal_weight=0.0141652, tolerance=1e-4, mode="Methodical1",
        source="cpp/tests/linear_programming/pdlp_test.cu:237-239,276-283",
        oracle_stable2_step_size=s["initial_step_size"], oracle_stable2_primal_weight=s["initial_primal_weight"])
    h = orcbind.hyper_preset(2)
    h[orcbind.H["ORC_H_RESTART_STRATEGY"]] = 1  # scaling/init only differ by Ruiz iterations; restart unused at it=0
    s2 = orcbind.solve(afiro, hyper=h, iteration_limit=0)
    problems["afiro"]["pinned_initial"].update(oracle_methodical1_step_size=s2["initial_step_size"],
                                               oracle_methodical1_primal_weight=s2["initial_primal_weight"])
    json.dump(problems, open(os.path.join(OUT, "problems.json"), "w"), indent=0)

    parsed = parser_goldens(lp)
    print("problems:", {k: (v["m"], v["n"], v["nnz"], v["reference_dual_simplex"]["objective"]) for k, v in problems.items()})
    print("parser fixtures:", sum(v["ok"] for v in parsed.values()), "ok /", len(parsed))


if __name__ == "__main__":
    main()
