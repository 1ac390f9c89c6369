"""TEST INFRASTRUCTURE ONLY: ctypes binding over oracle/liboracle_pdlp.so (pdlp_oracle.c), the
plain-C CPU restatement of the reference PDLP.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product never does."""
import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_pdlp.so")
_lib = None


def _enum(prefix):
    """parse the index enums out of pdlp_oracle.h so Python can never drift from C."""
    txt = open(os.path.join(_HERE, "pdlp_oracle.h")).read()
    names = re.findall(r"\b(%s[A-Z0-9_]+)\b" % prefix, txt)
    out, seen = {}, 0
    for nm in names:
        if nm not in out:
            out[nm] = seen
            seen += 1
    return out


H = _enum("ORC_H_")
S = _enum("ORC_S_")
O = _enum("ORC_O_")


def available():
    return os.path.exists(_SO)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_SO)
        vp, dp, ci, cd = C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_double
        _lib.orc_hyper_preset.argtypes = [ci, vp]
        _lib.orc_default_settings.argtypes = [vp]
        _lib.orc_spmv.argtypes = [ci, vp, vp, vp, vp, vp]
        _lib.orc_csr_transpose.argtypes = [ci, ci] + [vp] * 6
        _lib.orc_compute_scaling.argtypes = [ci, ci] + [vp] * 9
        _lib.orc_eval.argtypes = [ci, ci] + [vp] * 11 + [cd, cd, ci, cd, cd] + [vp] * 4
        _lib.orc_trust_region_bounds.argtypes = [ci, ci] + [vp] * 11 + [cd, cd, cd] + [vp] * 3
        _lib.orc_eval_infeasibility.argtypes = [ci, ci] + [vp] * 11 + [ci] + [vp] * 3
        _lib.orc_pdlp_solve.argtypes = [ci, ci] + [vp] * 8 + [ci, cd] + [vp] * 8
        _lib.orc_pdlp_solve.restype = ci
        _lib.orc_pdhg_fixed_steps.argtypes = [ci, ci] + [vp] * 11 + [cd, cd, ci, vp, vp]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def default_threads(cap=8):
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(cap, avail))


def hyper_preset(mode=1):
    h = np.zeros(H["ORC_H_COUNT"])
    lib().orc_hyper_preset(int(mode), _p(h))
    return h


def default_settings():
    s = np.zeros(S["ORC_S_COUNT"])
    lib().orc_default_settings(_p(s))
    return s


def spmv(offsets, indices, values, x):
    offsets, indices, values, x = _i32(offsets), _i32(indices), _f64(values), _f64(x)
    y = np.zeros(len(offsets) - 1)
    lib().orc_spmv(len(y), _p(offsets), _p(indices), _p(values), _p(x), _p(y))
    return y


def transpose(m, n, offsets, indices, values):
    offsets, indices, values = _i32(offsets), _i32(indices), _f64(values)
    to, ti, tv = np.zeros(n + 1, np.int32), np.zeros(len(indices), np.int32), np.zeros(len(values))
    lib().orc_csr_transpose(m, n, _p(offsets), _p(indices), _p(values), _p(to), _p(ti), _p(tv))
    return to, ti, tv


def compute_scaling(m, n, offsets, indices, values, hyper):
    offsets, indices, values = _i32(offsets), _i32(indices), _f64(values)
    to, ti, tv = transpose(m, n, offsets, indices, values)
    dr, dc = np.zeros(m), np.zeros(n)
    lib().orc_compute_scaling(m, n, _p(offsets), _p(indices), _p(values), _p(to), _p(ti), _p(tv),
                              _p(_f64(hyper)), _p(dr), _p(dc))
    return dr, dc


def evaluate(p, x, y, finite_bounds_rule=True, rel_primal_tol=1e-4, rel_dual_tol=1e-4):
    """orc_eval on problem dict p (user form; maximize handled here like problem_t does)."""
    m, n = int(p["m"]), int(p["n"])
    offsets, indices, values = _i32(p["offsets"]), _i32(p["indices"]), _f64(p["values"])
    to, ti, tv = transpose(m, n, offsets, indices, values)
    c = _f64(p["c"]).copy()
    scale = float(p.get("objective_scaling_factor", 1.0))
    if p.get("maximize", False):
        c, scale = -c, -scale
    lo, hi, lb, ub = (_f64(p[k]) for k in ("lo", "hi", "lb", "ub"))
    rc, out = np.zeros(n), np.zeros(10)
    x, y = _f64(x), _f64(y)
    lib().orc_eval(m, n, _p(offsets), _p(indices), _p(values), _p(to), _p(ti), _p(tv), _p(c),
                   _p(lo), _p(hi), _p(lb), _p(ub), scale, float(p.get("objective_offset", 0.0)),
                   int(finite_bounds_rule), rel_primal_tol, rel_dual_tol, _p(x), _p(y), _p(rc), _p(out))
    keys = ["primal_objective", "dual_objective", "gap", "abs_objective", "l2_primal_residual",
            "l2_dual_residual", "l2_x", "l2_y", "linf_rel_primal_residual", "linf_rel_dual_residual"]
    r = dict(zip(keys, out.tolist()))
    r["reduced_cost"] = rc
    return r


def trust_region_bounds(p, x, y, wp, wd, radius):
    m, n = int(p["m"]), int(p["n"])
    offsets, indices, values = _i32(p["offsets"]), _i32(p["indices"]), _f64(p["values"])
    to, ti, tv = transpose(m, n, offsets, indices, values)
    c = _f64(p["c"]).copy()
    if p.get("maximize", False):
        c = -c
    lo, hi, lb, ub = (_f64(p[k]) for k in ("lo", "hi", "lb", "ub"))
    out = np.zeros(3)
    x, y = _f64(x), _f64(y)
    lib().orc_trust_region_bounds(m, n, _p(offsets), _p(indices), _p(values), _p(to), _p(ti), _p(tv), _p(c),
                                  _p(lo), _p(hi), _p(lb), _p(ub), float(wp), float(wd), float(radius), _p(x),
                                  _p(y), _p(out))
    return dict(lagrangian=out[0], lower_bound=out[1], upper_bound=out[2])


def evaluate_infeasibility(p, x, y, finite_bounds_rule=True):
    m, n = int(p["m"]), int(p["n"])
    offsets, indices, values = _i32(p["offsets"]), _i32(p["indices"]), _f64(p["values"])
    to, ti, tv = transpose(m, n, offsets, indices, values)
    c = _f64(p["c"]).copy()
    if p.get("maximize", False):
        c = -c
    lo, hi, lb, ub = (_f64(p[k]) for k in ("lo", "hi", "lb", "ub"))
    out = np.zeros(4)
    x, y = _f64(x), _f64(y)
    lib().orc_eval_infeasibility(m, n, _p(offsets), _p(indices), _p(values), _p(to), _p(ti), _p(tv), _p(c),
                                 _p(lo), _p(hi), _p(lb), _p(ub), int(finite_bounds_rule), _p(x), _p(y), _p(out))
    return dict(zip(["max_primal_ray_infeasibility", "primal_ray_linear_objective",
                     "max_dual_ray_infeasibility", "dual_ray_linear_objective"], out.tolist()))


STATUS = {0: "NoTermination", 1: "Optimal", 2: "PrimalInfeasible", 3: "DualInfeasible",
          4: "IterationLimit", 5: "TimeLimit", 6: "NumericalError", 7: "PrimalFeasible",
          8: "FeasibleFound", 9: "ConcurrentLimit"}


def solve(p, mode=1, hyper=None, init_x=None, init_y=None, **settings):
    """Full oracle PDLP solve on problem dict p = {m,n,offsets,indices,values,c,lo,hi,lb,ub,
    maximize,objective_offset}.  settings: tol=..., iteration_limit=..., time_limit=...,
    per_constraint_residual=..., first_primal_feasible=..., num_threads=..., or any of
    abs_gap_tol/rel_gap_tol/abs_primal_tol/rel_primal_tol/abs_dual_tol/rel_dual_tol."""
    m, n = int(p["m"]), int(p["n"])
    h = hyper_preset(mode) if hyper is None else _f64(hyper)
    s = default_settings()
    # never let OpenMP spawn one thread per *visible* CPU: on cgroup-limited hosts that is a
    # pathological oversubscription.  Default: at most 8 threads (override with num_threads=...).
    settings.setdefault("num_threads", default_threads())
    if "tol" in settings:
        s[:6] = settings.pop("tol")
    for k, v in settings.items():
        s[S["ORC_S_" + k.upper()]] = float(v)
    offsets, indices, values = _i32(p["offsets"]), _i32(p["indices"]), _f64(p["values"])
    c, lo, hi, lb, ub = (_f64(p[k]) for k in ("c", "lo", "hi", "lb", "ub"))
    x, y, rc, st = np.zeros(n), np.zeros(m), np.zeros(n), np.zeros(O["ORC_O_COUNT"])
    ix = None if init_x is None else _f64(init_x)
    iy = None if init_y is None else _f64(init_y)
    r = lib().orc_pdlp_solve(m, n, _p(offsets), _p(indices), _p(values), _p(c), _p(lo), _p(hi),
                             _p(lb), _p(ub), int(bool(p.get("maximize", False))),
                             float(p.get("objective_offset", 0.0)), _p(h), _p(s), _p(ix), _p(iy),
                             _p(x), _p(y), _p(rc), _p(st))
    if r != 0:
        raise RuntimeError("oracle: unsupported configuration (rc=%d)" % r)
    out = {k[6:].lower(): float(st[i]) for k, i in O.items() if k != "ORC_O_COUNT"}
    out["status"] = STATUS[int(st[0])]
    out["status_code"] = int(st[0])
    out.update(x=x, y=y, reduced_cost=rc)
    return out
