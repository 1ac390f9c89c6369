import json
import os
import sys

import numpy as np
import pytest

# the oracle's OpenMP threads sleep between its (many, short) parallel regions instead of spinning: the GPU boxes give a test run a
# CPU quota, and spinning workers eat it while the test thread waits for the device
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
# The sharded tests run up to eight RANKS as host threads on ONE device, each with a stream of its own; under the direct peer transport a
# rank's consumer kernel SPINS until its peers' flags arrive.  HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default
# 4): two ranks on one queue = a spinning kernel in front of the producer it waits for = a 5-second timeout (found in round 6, by a
# test that ran world 2 and world 4 in one process).  One queue per rank; production runs one rank per device and never shares one.
# Must be in the environment before the HIP runtime initialises, i.e. before cuopt_amd.capi loads the library.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


def set_tune(monkeypatch, **kw):
    """Merge keys into CUOPT_AMD_TUNE (the one string of test / harness knobs, host_parallel.hpp); value None removes a key."""
    cur = dict(item.split("=", 1) for item in os.environ.get("CUOPT_AMD_TUNE", "").split(",") if "=" in item)
    for k, v in kw.items():
        if v is None:
            cur.pop(k, None)
        else:
            cur[k] = str(v)
    if cur:
        monkeypatch.setenv("CUOPT_AMD_TUNE", ",".join("%s=%s" % kv for kv in cur.items()))
    else:
        monkeypatch.delenv("CUOPT_AMD_TUNE", raising=False)


def _dec(v):
    return np.array([np.inf if x == "inf" else -np.inf if x == "-inf" else x for x in v], dtype=np.float64)


def decode_problem(d):
    p = dict(m=d["m"], n=d["n"], offsets=np.array(d["offsets"], np.int32),
             indices=np.array(d["indices"], np.int32), values=_dec(d["values"]), c=_dec(d["c"]),
             lo=_dec(d["lo"]), hi=_dec(d["hi"]), lb=_dec(d["lb"]), ub=_dec(d["ub"]),
             maximize=bool(d["maximize"]), objective_offset=float(d["objective_offset"]))
    for k in ("row_names", "var_names", "objective_name", "problem_name", "var_types"):
        if k in d:
            p[k] = d[k]
    return p


@pytest.fixture(scope="session")
def golden_problems():
    raw = json.load(open(os.path.join(GOLDEN, "problems.json")))
    return {k: dict(problem=decode_problem(v), meta=v) for k, v in raw.items()}


@pytest.fixture(scope="session")
def golden_parser():
    return json.load(open(os.path.join(GOLDEN, "mps_parser.json")))


def write_mps(path, p, name="LP"):
    """Own minimal free-format MPS writer (test utility): ranged rows become E/L/G + RANGES."""
    m, n = p["m"], p["n"]
    rows = p.get("row_names") or ["R%d" % i for i in range(m)]
    cols = p.get("var_names") or ["X%d" % j for j in range(n)]
    lo, hi = p["lo"], p["hi"]
    with open(path, "w") as f:
        f.write("NAME %s\n" % name)
        if p.get("maximize"):
            f.write("OBJSENSE\n    MAX\n")
        f.write("ROWS\n N COST\n")
        kinds, rhs, rng = [], [], []
        for i in range(m):
            if lo[i] == hi[i]:
                kinds.append("E"), rhs.append(lo[i]), rng.append(None)
            elif np.isinf(lo[i]) and not np.isinf(hi[i]):
                kinds.append("L"), rhs.append(hi[i]), rng.append(None)
            elif np.isinf(hi[i]) and not np.isinf(lo[i]):
                kinds.append("G"), rhs.append(lo[i]), rng.append(None)
            elif np.isinf(lo[i]) and np.isinf(hi[i]):
                raise ValueError("free rows are not written")
            else:
                kinds.append("G"), rhs.append(lo[i]), rng.append(hi[i] - lo[i])
            f.write(" %s %s\n" % (kinds[-1], rows[i]))
        f.write("COLUMNS\n")
        col_entries = [[] for _ in range(n)]
        for i in range(m):
            for k in range(p["offsets"][i], p["offsets"][i + 1]):
                col_entries[p["indices"][k]].append((rows[i], p["values"][k]))
        for j in range(n):
            if p["c"][j] != 0.0 or not col_entries[j]:
                f.write("    %s COST %.17g\n" % (cols[j], p["c"][j]))
            for r, v in col_entries[j]:
                f.write("    %s %s %.17g\n" % (cols[j], r, v))
        f.write("RHS\n")
        if p.get("objective_offset", 0.0) != 0.0:
            f.write("    RHS COST %.17g\n" % (-p["objective_offset"]))
        for i in range(m):
            if rhs[i] != 0.0:
                f.write("    RHS %s %.17g\n" % (rows[i], rhs[i]))
        if any(r is not None for r in rng):
            f.write("RANGES\n")
            for i in range(m):
                if rng[i] is not None:
                    f.write("    RNG %s %.17g\n" % (rows[i], rng[i]))
        f.write("BOUNDS\n")
        for j in range(n):
            lb, ub = p["lb"][j], p["ub"][j]
            if lb == 0.0 and np.isinf(ub) and ub > 0:
                continue
            if np.isinf(lb) and np.isinf(ub):
                f.write(" FR BND %s\n" % cols[j])
                continue
            if np.isinf(lb):
                f.write(" MI BND %s\n" % cols[j])
            elif lb != 0.0:
                f.write(" LO BND %s %.17g\n" % (cols[j], lb))
            if not np.isinf(ub):
                f.write(" UP BND %s %.17g\n" % (cols[j], ub))
        f.write("ENDATA\n")


def has_gpu():
    try:
        from cuopt_amd import capi
        return capi.device_count() > 0
    except Exception:
        return False
