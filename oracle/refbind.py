"""TEST INFRASTRUCTURE ONLY: ctypes binding over oracle/_ref/libcuopt_ref.so, i.e. the reference's
own libmps_parser and CPU dual simplex compiled in place (oracle/Makefile, oracle/ref_driver.cpp).
Only tests/, scripts/make_golden.py, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libcuopt_ref.so")
_lib = None


def available():
    return os.path.exists(_SO)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_SO)
        _lib.ref_mps_parse.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.c_char_p, C.c_int]
        _lib.ref_mps_free.argtypes = [C.c_void_p]
        _lib.ref_mps_dims.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4 + [C.POINTER(C.c_double)] * 2
        _lib.ref_mps_flags.argtypes = [C.c_void_p]
        _lib.ref_mps_arrays.argtypes = [C.c_void_p] + [C.c_void_p] * 11
        _lib.ref_mps_name.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
        _lib.ref_dual_simplex.argtypes = (
            [C.c_int, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_double, C.c_double]
            + [C.POINTER(C.c_double)] + [C.c_void_p] * 3 + [C.POINTER(C.c_int)])
    return _lib


class RefMpsError(Exception):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def parse_mps(path, fixed_format=False):
    """Parse with the REFERENCE parser. Returns a dict of numpy arrays (constraint bounds always
    materialised as lo/hi the way problem_helpers.cuh:33-58 does)."""
    L = lib()
    h = C.c_void_p()
    err = C.create_string_buffer(1024)
    rc = L.ref_mps_parse(os.fsencode(path), int(fixed_format), C.byref(h), err, 1024)
    if rc != 0:
        raise RefMpsError(rc, err.value.decode(errors="replace"))
    try:
        m, n, nnz, mx = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        off, sc = C.c_double(), C.c_double()
        L.ref_mps_dims(h, C.byref(m), C.byref(n), C.byref(nnz), C.byref(mx), C.byref(off), C.byref(sc))
        m, n, nnz = m.value, n.value, nnz.value
        flags = L.ref_mps_flags(h)
        d = dict(
            m=m, n=n, nnz=nnz, maximize=bool(mx.value), objective_offset=off.value,
            objective_scaling_factor=sc.value, flags=flags,
            offsets=np.zeros(m + 1, np.int32), indices=np.zeros(nnz, np.int32),
            values=np.zeros(nnz), c=np.zeros(n), lo=np.zeros(m), hi=np.zeros(m),
            lb=np.zeros(n), ub=np.zeros(n), var_types=np.zeros(n, np.uint8),
            row_types=np.zeros(m if flags & 2 else 0, np.uint8),
            rhs=np.zeros(m if flags & 2 else 0))
        p = lambda a: a.ctypes.data_as(C.c_void_p) if a.size else None
        L.ref_mps_arrays(h, p(d["offsets"]), p(d["indices"]), p(d["values"]), p(d["c"]), p(d["lo"]),
                         p(d["hi"]), p(d["lb"]), p(d["ub"]), p(d["var_types"]), p(d["row_types"]),
                         p(d["rhs"]))
        buf = C.create_string_buffer(4096)
        def name(kind, idx=0):
            L.ref_mps_name(h, kind, idx, buf, 4096)
            return buf.value.decode(errors="replace")
        d["problem_name"] = name(0)
        d["objective_name"] = name(1)
        d["var_names"] = [name(2, j) for j in range(n)]
        d["row_names"] = [name(3, i) for i in range(m)]
        return d
    finally:
        L.ref_mps_free(h)


DS_STATUS = {0: "OPTIMAL", 1: "INFEASIBLE", 2: "UNBOUNDED", 3: "ITERATION_LIMIT", 4: "TIME_LIMIT",
             5: "NUMERICAL_ISSUES", 6: "CUTOFF", 7: "CONCURRENT_LIMIT", 8: "UNSET"}


def dual_simplex(p, time_limit=0.0):
    """Reference CPU dual simplex on a problem dict (keys as returned by parse_mps)."""
    L = lib()
    m, n = int(p["m"]), int(p["n"])
    arr = lambda k, t: np.ascontiguousarray(p[k], dtype=t)
    offsets, indices, values = arr("offsets", np.int32), arr("indices", np.int32), arr("values", np.float64)
    c, lo, hi, lb, ub = (arr(k, np.float64) for k in ("c", "lo", "hi", "lb", "ub"))
    x, y, z = np.zeros(n), np.zeros(m), np.zeros(n)
    obj, its = C.c_double(), C.c_int()
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    st = L.ref_dual_simplex(m, n, ptr(offsets), ptr(indices), ptr(values), ptr(c), ptr(lo), ptr(hi),
                            ptr(lb), ptr(ub), int(bool(p.get("maximize", False))),
                            float(p.get("objective_offset", 0.0)), float(time_limit),
                            C.byref(obj), ptr(x), ptr(y), ptr(z), C.byref(its))
    return dict(status=DS_STATUS.get(st, str(st)), objective=obj.value, iterations=its.value, x=x, y=y, z=z)
