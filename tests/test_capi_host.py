"""CPU: the C-ABI library loads, exports every symbol the headers declare, and the host-only part
of the libcuopt C API (problem builder, getters, parameter registry, MPS reader) behaves like the
reference (cpp/src/linear_programming/cuopt_c.cpp, cpp/tests/linear_programming/c_api_tests)."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import ROOT, decode_problem, write_mps, set_tune
from cuopt_amd import capi

INF = np.inf


def _declared_functions(header):
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    return set(re.findall(r"\b((?:cuOpt|cuoptamd_|pdlpdev_)\w+)\s*\(", txt))


def test_every_declared_symbol_is_exported():
    headers = glob.glob(os.path.join(ROOT, "include", "**", "*.h"), recursive=True)
    assert len(headers) >= 4
    names = set()
    for h in headers:
        names |= _declared_functions(h)
    assert len([n for n in names if n.startswith("cuOpt") and not n.startswith("cuOptAmd")]) == 41  # cuopt_c.h:89-668
    assert {n for n in names if n.startswith("cuOptAmd")} == {"cuOptAmdGetPdlpStats", "cuOptAmdGetSolveInfo", "cuOptAmdReadSolutionFile", "cuOptAmdGetName"}  # cuopt_c_ext.h
    missing = [n for n in sorted(names) if not hasattr(capi.lib, n)]
    assert not missing, missing


def test_scalar_sizes():  # c_api_tests.cpp:27-29
    assert capi.lib.cuOptGetFloatSize() == 8 and capi.lib.cuOptGetIntSize() == 4


def test_problem_round_trip_and_null_checks(golden_problems):
    p = golden_problems["afiro"]["problem"]
    prob = capi.Problem.from_dict(p)
    assert (prob.m, prob.n, prob.nnz, prob.is_mip) == (27, 32, 83, False)
    d = prob.to_dict()
    for k in ("offsets", "indices", "values", "c", "lo", "hi", "lb", "ub"):
        np.testing.assert_array_equal(d[k], p[k])
    v = C.c_int32()
    assert capi.lib.cuOptGetNumConstraints(None, C.byref(v)) == capi.CUOPT_INVALID_ARGUMENT
    assert capi.lib.cuOptGetNumConstraints(prob.handle, None) == capi.CUOPT_INVALID_ARGUMENT
    h = C.c_void_p()
    assert capi.lib.cuOptCreateProblem(1, 1, 1, 0.0, None, None, None, None, None, None, None, None, None,
                                       C.byref(h)) == capi.CUOPT_INVALID_ARGUMENT
    prob.close()
    assert not prob.handle  # Destroy NULLs the handle (cuopt_c.cpp:200-206)
    capi.lib.cuOptDestroyProblem(None)  # tolerated
    capi.lib.cuOptDestroyProblem(C.byref(C.c_void_p()))


def test_sense_rhs_builder_materialises_bounds():
    p = dict(m=3, n=2, offsets=[0, 2, 4, 6], indices=[0, 1, 0, 1, 0, 1], values=[1.0] * 6, c=[1.0, 1.0],
             lb=[0.0, 0.0], ub=[INF, INF], row_types=np.frombuffer(b"LGE", np.uint8), rhs=[1.0, 2.0, 3.0])
    prob = capi.Problem.from_dict(p, ranged=False)
    d = prob.to_dict()
    np.testing.assert_array_equal(d["lo"], [-INF, 2.0, 3.0])
    np.testing.assert_array_equal(d["hi"], [1.0, INF, 3.0])
    sense = np.zeros(3, np.uint8)
    capi.lib.cuOptGetConstraintSense(prob.handle, sense.ctypes.data_as(C.c_void_p))
    assert bytes(sense) == b"LGE"


def test_integer_variables_are_reported():
    p = dict(m=1, n=2, offsets=[0, 2], indices=[0, 1], values=[1.0, 1.0], c=[1.0, 1.0], lo=[0.0], hi=[1.0],
             lb=[0.0, 0.0], ub=[1.0, 1.0], var_types=np.frombuffer(b"CI", np.uint8))
    assert capi.Problem.from_dict(p).is_mip


def test_parameter_registry():
    """names/ranges/defaults: cpp/src/math_optimization/solver_settings.cu:66-118"""
    s = capi.Settings()
    for k in ("absolute_dual_tolerance", "relative_dual_tolerance", "absolute_primal_tolerance",
              "relative_primal_tolerance", "absolute_gap_tolerance", "relative_gap_tolerance"):
        assert float(s.get(k)) == pytest.approx(1e-4)
    assert int(s.get("iteration_limit")) == 2 ** 31 - 1
    assert int(s.get("pdlp_solver_mode")) == 1 and int(s.get("method")) == 0
    assert s.get("time_limit") == "inf" and s.get("crossover") == "false"
    s.set("time_limit", 2.5)
    assert float(s.get("time_limit")) == 2.5
    s.set("per_constraint_residual", "T")
    assert s.get("per_constraint_residual") == "true"
    v = C.c_int32()
    # integer setter falls back to a boolean parameter (cuopt_c.cpp:493-505)
    assert capi.lib.cuOptSetIntegerParameter(s.handle, b"crossover", 1) == 0
    assert capi.lib.cuOptGetIntegerParameter(s.handle, b"crossover", C.byref(v)) == 0 and v.value == 1
    # c_api_tests.cpp:82: bad parameter name
    assert capi.lib.cuOptSetParameter(s.handle, b"bad_parameter_name", b"1") == capi.CUOPT_INVALID_ARGUMENT
    assert capi.lib.cuOptSetFloatParameter(s.handle, b"absolute_gap_tolerance", 0.5) == capi.CUOPT_INVALID_ARGUMENT
    assert capi.lib.cuOptSetIntegerParameter(s.handle, b"pdlp_solver_mode", 7) == capi.CUOPT_INVALID_ARGUMENT
    assert capi.lib.cuOptSetParameter(s.handle, b"iteration_limit", b"abc") == capi.CUOPT_INVALID_ARGUMENT
    assert capi.lib.cuOptSetParameter(s.handle, None, b"1") == capi.CUOPT_INVALID_ARGUMENT


def test_read_problem_error_codes(tmp_path):
    h = C.c_void_p()
    # c_api_tests.cpp:86: missing file -> CUOPT_MPS_FILE_ERROR
    assert capi.lib.cuOptReadProblem(b"/nonexistent/file.mps", C.byref(h)) == capi.CUOPT_MPS_FILE_ERROR
    assert not h
    bad = tmp_path / "bad.mps"
    bad.write_text("NAME x\nROWS\n N obj\n L r1\nCOLUMNS\n    x nosuchrow 1.0\nRHS\nENDATA\n")
    assert capi.lib.cuOptReadProblem(os.fsencode(str(bad)), C.byref(h)) == capi.CUOPT_MPS_PARSE_ERROR


def test_mps_round_trip_through_own_writer(golden_problems, tmp_path):
    for name, g in golden_problems.items():
        path = str(tmp_path / (name + ".mps"))
        write_mps(path, g["problem"])
        d = capi.Problem.read(path).to_dict()
        p = g["problem"]
        assert (d["m"], d["n"], d["maximize"]) == (p["m"], p["n"], p["maximize"])
        assert d["objective_offset"] == pytest.approx(p["objective_offset"])
        for k in ("offsets", "indices", "values", "c", "lo", "hi", "lb", "ub"):
            np.testing.assert_allclose(d[k], p[k], rtol=0, atol=0, err_msg="%s:%s" % (name, k))


REF_DIR = "/root/reference/datasets/linear_programming"


@pytest.mark.skipif(not os.path.isdir(REF_DIR), reason="reference datasets only exist in the build container")
def test_mps_reader_matches_reference_parser_on_all_fixtures(golden_parser):
    """every datasets/linear_programming/*.mps: same accept/reject decision as the reference's parser
    in free-format mode, and identical CSR / bounds / objective when accepted."""
    checked = 0
    for name, ref in sorted(golden_parser.items()):
        h = C.c_void_p()
        rc = capi.lib.cuOptReadProblem(os.fsencode(os.path.join(REF_DIR, name)), C.byref(h))
        if not ref["ok"]:
            assert rc == capi.CUOPT_MPS_PARSE_ERROR, name
            continue
        assert rc == 0, name
        d = capi.Problem(h).to_dict()
        r = decode_problem(ref)
        assert (d["m"], d["n"], d["maximize"]) == (r["m"], r["n"], r["maximize"]), name
        assert d["objective_offset"] == r["objective_offset"], name
        for k in ("offsets", "indices", "values", "c", "lo", "hi", "lb", "ub"):
            np.testing.assert_array_equal(d[k], r[k], err_msg="%s:%s" % (name, k))
        ref_types = bytes(bytearray(ref["var_types"])).replace(b"\x00", b"C")
        assert bytes(bytearray(d["var_types"])) == ref_types, name
        checked += 1
    assert checked >= 20


def test_no_gpu_means_loud_failure_not_fallback(golden_problems):
    """the product has no CPU path: without a HIP device cuOptSolve must report an error"""
    if capi.device_count() > 0:
        pytest.skip("a GPU is visible")
    r = capi.solve(golden_problems["afiro"]["problem"])
    assert r["return_code"] == capi.CUOPT_RUNTIME_ERROR
    assert "no HIP device" in r["error_string"]


def test_partition_rows_balances_nonzeros():
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 50, size=1000)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    for world in (1, 2, 4, 8):
        b = capi.partition_rows(1000, off, world)
        assert b[0] == 0 and b[-1] == 1000 and np.all(np.diff(b) >= 0)
        per = np.diff(off[b])
        assert per.max() - per.min() <= 2 * 50


def test_csr_transpose_matches_scipy():
    import scipy.sparse as sp
    rng = np.random.default_rng(1)
    a = sp.random(40, 30, density=0.2, random_state=rng, format="csr")
    to, ti, tv = capi.csr_transpose(40, 30, a.indptr, a.indices, a.data)
    t = sp.csr_matrix((tv, ti, to), shape=(30, 40))
    assert abs(t - a.T).max() == 0
    assert np.all(np.diff(to) >= 0) and all(np.all(np.diff(ti[to[j]:to[j + 1]]) > 0) for j in range(30))


def test_blocked_transposition_of_large_matrices_is_the_direct_one(monkeypatch):
    """from 2^22 nonzeros on the host transposition deals the entries into column-block buckets first (cache-sized scatters):
    the result is the direct counting sort's, entry for entry -- skewed columns, empty columns and rows included"""
    import scipy.sparse as sp
    rng = np.random.default_rng(3)
    m, n, nnz = 400_000, 300_000, 4_400_000
    rows = rng.integers(0, m, size=nnz)
    cols = np.where(rng.random(nnz) < 0.2, rng.integers(0, 50, size=nnz), rng.integers(1000, n - 1000, size=nnz))  # hot columns, empty ones
    a = sp.csr_matrix((rng.standard_normal(nnz), (rows, cols)), shape=(m, n))  # (duplicates summed)
    a.sort_indices()
    blocked = capi.csr_transpose(m, n, a.indptr, a.indices, a.data)
    set_tune(monkeypatch, transpose_direct="1")
    direct = capi.csr_transpose(m, n, a.indptr, a.indices, a.data)
    for x, y in zip(blocked, direct):
        assert np.array_equal(x, y)
    t = a.T.tocsr()
    t.sort_indices()
    assert np.array_equal(blocked[0], t.indptr) and np.array_equal(blocked[1], t.indices) and np.array_equal(blocked[2], t.data)


def test_user_problem_file_round_trip(tmp_path):
    """CUOPT_USER_PROBLEM_FILE (solve.cu:586-589, test_lp_solver.py:675-700): cuOptSolve writes the problem as MPS before
    anything else happens (so this runs without a GPU: the solve itself then fails loudly); reading the file back
    gives the same LP -- every bound flavour, ranged rows, both senses, objective offset, 17 significant digits.
    Free rows have no MPS row type of their own: they are written as additional N rows, which MPS readers skip."""
    from test_random_lps_gpu import random_lp
    for seed in range(6):
        p, _ = random_lp(seed)
        if seed == 3:
            p["lb"][0] = p["ub"][0] = 1.5  # FX
        path = str(tmp_path / ("user_%d.mps" % seed))
        prob = capi.Problem.from_dict(p)
        st = capi.Settings(user_problem_file=path, iteration_limit=1)
        sol = C.c_void_p()
        capi.lib.cuOptSolve(prob.handle, st.handle, C.byref(sol))
        capi.lib.cuOptDestroySolution(C.byref(sol))
        back = capi.Problem.read(path).to_dict()
        keep = np.isfinite(p["lo"]) | np.isfinite(p["hi"])
        A = sp.csr_matrix((p["values"], p["indices"], p["offsets"]), shape=(p["m"], p["n"]))[keep]
        A.sort_indices()
        assert back["m"] == int(keep.sum()) and back["n"] == p["n"]
        np.testing.assert_array_equal(back["offsets"], A.indptr)
        np.testing.assert_array_equal(back["indices"], A.indices)
        np.testing.assert_array_equal(back["values"], A.data)
        for k in ("c", "lb", "ub"):
            np.testing.assert_array_equal(back[k], p[k])
        # (a two-sided row travels as bound + range: exact whenever one of the two encodings is, else within an ulp)
        np.testing.assert_allclose(back["lo"], p["lo"][keep], rtol=3e-16, atol=0)
        np.testing.assert_allclose(back["hi"], p["hi"][keep], rtol=3e-16, atol=0)
        assert back["maximize"] == p["maximize"] and back["objective_offset"] == p["objective_offset"]


def test_solution_file_reader(tmp_path):
    """cuOptAmdReadSolutionFile vs the behaviour of solution_reader.cu:57-145 (own writer format and MIPLIB flavour)"""
    mps = tmp_path / "p.mps"
    mps.write_text("NAME T\nROWS\n N COST\n L R1\nCOLUMNS\n X1 COST 1 R1 1\n YY COST 2 R1 1\n Z COST 3 R1 1\nRHS\n RHS R1 4\nENDATA\n")
    prob = capi.Problem.read(str(mps))
    sol = tmp_path / "a.sol"
    sol.write_text("# Status: Optimal\n# Objective value: -12.5\nZ 3.25\nX1 1e-3\n\nYY -7\n# trailing comment\n")
    vals = np.zeros(3)
    obj = C.c_double()
    status = C.create_string_buffer(64)
    fn = capi.lib.cuOptAmdReadSolutionFile
    fn.restype = C.c_int32
    fn.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_double), C.c_char_p, C.c_int32]
    assert fn(prob.handle, str(sol).encode(), capi._ptr(vals), C.byref(obj), status, 64) == capi.CUOPT_SUCCESS
    np.testing.assert_array_equal(vals, [1e-3, -7.0, 3.25])
    assert obj.value == -12.5 and status.value == b"Optimal"
    mip = tmp_path / "b.sol"
    mip.write_text("=obj= 42\nX1 1\nYY 2\nZ 3\nZ 4\n")  # MIPLIB flavour; a repeated name: last one wins
    assert fn(prob.handle, str(mip).encode(), capi._ptr(vals), C.byref(obj), None, 0) == capi.CUOPT_SUCCESS
    np.testing.assert_array_equal(vals, [1.0, 2.0, 4.0])
    assert obj.value == 42.0
    short = tmp_path / "c.sol"
    short.write_text("X1 1\nZ 3\n")
    assert fn(prob.handle, str(short).encode(), capi._ptr(vals), None, None, 0) == capi.CUOPT_VALIDATION_ERROR
    assert fn(prob.handle, str(tmp_path / "none.sol").encode(), capi._ptr(vals), None, None, 0) == capi.CUOPT_MPS_FILE_ERROR
    prob.close()


def _snapshot(primal, dual):
    d = {k: np.array(primal, dtype=np.float64) for k in capi.WarmStart.PRIMAL}
    d.update({k: np.array(dual, dtype=np.float64) for k in capi.WarmStart.DUAL})
    d.update({k: -1 for k in capi.WarmStart.SCALARS})
    d.update(n_variables=len(primal), n_constraints=len(dual))
    return d


PRIMAL_SIDE = [k for k in capi.WarmStart.PRIMAL if not k.endswith("_scaled")]
DUAL_SIDE = [k for k in capi.WarmStart.DUAL if not k.endswith("_scaled")]


def test_warm_start_smaller_vector():
    """unit_tests/solver_settings_test.cu:84-181: two of four variables kept (0 - 1 swapped), three of four constraints
    (1 - 2 swapped); every primal-side / dual-side vector of the snapshot follows, the scalars travel unchanged"""
    out = capi.remap_warm_start(_snapshot([0.0, 1.0, 2.0, 3.0], [0.0, 1.0, 2.0, 3.0]), [1, 0], [0, 2, 1])
    for k in PRIMAL_SIDE:
        assert out[k].tolist() == [1.0, 0.0], k
    for k in DUAL_SIDE:
        assert out[k].tolist() == [0.0, 2.0, 1.0], k
    assert (out["n_variables"], out["n_constraints"]) == (2, 3)
    assert out["initial_step_size"] == -1 and out["total_pdlp_iterations"] == -1
    assert out["current_primal_solution_scaled"] is None and out["current_dual_solution_scaled"] is None


def test_warm_start_bigger_vector():
    """unit_tests/solver_settings_test.cu:183-280: six variables for four, seven constraints for three -> zero padding"""
    out = capi.remap_warm_start(_snapshot([0.0, 1.0, 2.0, 3.0], [0.0, 1.0, 2.0]), list(range(6)), list(range(7)))
    for k in PRIMAL_SIDE:
        assert out[k].tolist() == [0.0, 1.0, 2.0, 3.0, 0.0, 0.0], k
    for k in DUAL_SIDE:
        assert out[k].tolist() == [0.0, 1.0, 2.0, 0.0, 0.0, 0.0, 0.0], k
    assert (out["n_variables"], out["n_constraints"]) == (6, 7)


def test_warm_start_mapping_edge_cases():
    """no mapping = untouched side (solver_settings.cu:98 `size() != 0`); where the reference would scatter out of range, an error"""
    snap = _snapshot([0.0, 1.0, 2.0, 3.0], [5.0, 6.0, 7.0])
    out = capi.remap_warm_start(snap, None, [1, 0])
    assert out["current_primal_solution"].tolist() == [0.0, 1.0, 2.0, 3.0]
    assert out["current_primal_solution_scaled"] is None  # a size changed: the old scaling's iterate is dropped
    assert out["current_dual_solution"].tolist() == [6.0, 5.0]
    same = capi.remap_warm_start(snap, [3, 2, 1, 0], None)  # same length: neither branch of the reference runs
    assert same["current_primal_solution"].tolist() == [0.0, 1.0, 2.0, 3.0]
    assert same["current_primal_solution_scaled"].tolist() == [0.0, 1.0, 2.0, 3.0]
    for bad in ([0, 2], [1, 1], [-1, 0]):
        with pytest.raises(capi.CuOptError):
            capi.remap_warm_start(snap, bad, None)


def test_problem_checking_through_the_c_api():
    """unit_tests/optimization_problem_test.cu:341-414 (test_csr_validity cases 1-4, test_row_type_invalidity_char) through the
    boundary: cuOptSolve answers CUOPT_VALIDATION_ERROR with the reference's message (utilities/problem_checking.cu) and still
    hands out a solution handle holding it (cuopt_c.cpp:613-618) -- before any device is touched, so this runs without a GPU"""
    base = dict(m=2, n=1, offsets=[0, 1, 2], indices=[0, 0], values=[1.0, 1.0], c=[1.0], lb=[0.0], ub=[np.inf],
                row_types=np.frombuffer(b"EE", np.uint8), rhs=[1.0, 1.0])
    cases = [(dict(offsets=[1, 1, 2]), "A_offsets first value should be 0"),
             (dict(offsets=[0, 2, 1], indices=[0]), "increasing order"),
             (dict(indices=[0, -1]), "A_indices"),
             (dict(indices=[0, 1]), "A_indices"),
             (dict(row_types=np.frombuffer(b"EN", np.uint8)), "row_types values must equal to 'E', 'G' or 'L'")]
    for change, message in cases:
        prob = capi.Problem.from_dict(dict(base, **change), ranged=False)
        r = capi.solve(prob, method=1)
        prob.close()
        assert r["return_code"] == capi.CUOPT_VALIDATION_ERROR, change
        assert r["error_status"] == capi.CUOPT_VALIDATION_ERROR and message in r["error_string"], r["error_string"]


@pytest.mark.parametrize("shape,transposed", [((200000, 30000, 3), False), ((200000, 30000, 3), True), ((30000, 200000, 16), False), ((30000, 200000, 16), True),
                                              ((1500000, 4400000, 3), False), ((1500000, 4400000, 3), True)])
def test_wide_bins_of_the_gather_free_layout_on_the_host(shape, transposed):
    """build_pb_wide (kernels_pb.hip) walked on the CPU exactly as phase P and phase R order the work -- pieces to image slots, steps of
    1024 slots, one addition per row and level: the row sums are the sequential CSR sums bit for bit (the oracle's), every slot is
    written once, no level exceeds its step's; rows that crowd into one step become serial rows; chunks beyond 16 bits are refused"""
    from cuopt_amd import synthetic
    from oracle import orcbind
    fn = capi.lib.pdlpdev_debug_pb_wide_host
    fn.restype = C.c_int
    p = synthetic.generate(*shape, seed=23)
    m, n, off, idx, val = p["m"], p["n"], p["offsets"], p["indices"], p["values"]
    if transposed:
        off, idx, val = capi.csr_transpose(m, n, off, idx, val)
        m, n = n, m
    off, idx, val = (np.ascontiguousarray(a, t) for a, t in ((off, np.int32), (idx, np.int32), (val, np.float64)))
    x = np.random.default_rng(1).standard_normal(n)
    out, info = np.zeros(m), np.zeros(5, np.int64)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    assert fn(C.c_int32(m), C.c_int32(n), ptr(off), ptr(idx), ptr(val), ptr(x), ptr(out), ptr(info)) == 0
    np.testing.assert_array_equal(out, orcbind.spmv(off, idx, val, x))
    assert info[1] == -(-m // 8192) and info[0] % 1024 == 0 and info[0] <= (1.06 if shape[1] < 4400000 else 1.25) * len(val) + 1024 * info[1] and info[3] <= 6
    assert info[4] == 0
    # ten entries per row over seven panels: a few rows have more than seven entries inside one step -- serial rows, still exact
    q = synthetic.generate(60000, 50000, 10, seed=23)
    off, idx, val = (np.ascontiguousarray(q[k], t) for k, t in (("offsets", np.int32), ("indices", np.int32), ("values", np.float64)))
    x, out = np.random.default_rng(3).standard_normal(q["n"]), np.zeros(q["m"])
    assert fn(C.c_int32(q["m"]), C.c_int32(q["n"]), ptr(off), ptr(idx), ptr(val), ptr(x), ptr(out), ptr(info)) == 0
    np.testing.assert_array_equal(out, orcbind.spmv(off, idx, val, x))
    assert 0 < info[4] < 600
    # forty entries per row over three panels: chunks of more than 65535 entries -- refused
    q = synthetic.generate(20000, 20000, 40, seed=23)
    off, idx, val = (np.ascontiguousarray(q[k], t) for k, t in (("offsets", np.int32), ("indices", np.int32), ("values", np.float64)))
    assert fn(C.c_int32(q["m"]), C.c_int32(q["n"]), ptr(off), ptr(idx), ptr(val), ptr(np.zeros(q["n"])), ptr(np.zeros(q["m"])), None) == 1

@pytest.mark.parametrize("transposed", [False, True])
def test_wide_bins_hand_long_clustered_rows_to_single_lanes_on_the_host(transposed):
    """40 rows that run through 60 consecutive columns each, among rows of three entries: in the A side's image such a row has dozens
    of entries inside one step -- it leaves the steps (its slots read as padding) and is summed by one lane from its list of slots,
    left to right: still the sequential CSR sum, bit for bit.  (On the A^T side the same entries are one per row.)"""
    from cuopt_amd import synthetic
    from oracle import orcbind
    fn = capi.lib.pdlpdev_debug_pb_wide_host
    fn.restype = C.c_int
    p = synthetic.generate_clustered(200000, 30000, 3, heavy=40, width=60, seed=11, empty_rows=(8192, 16384))  # (+ a bin without a product)
    m, n, off, idx, val = p["m"], p["n"], p["offsets"], p["indices"], p["values"]
    if transposed:
        off, idx, val = capi.csr_transpose(m, n, off, idx, val)
        m, n = n, m
    off, idx, val = (np.ascontiguousarray(a, t) for a, t in ((off, np.int32), (idx, np.int32), (val, np.float64)))
    x = np.random.default_rng(2).standard_normal(n)
    out, info = np.zeros(m), np.zeros(5, np.int64)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    assert fn(C.c_int32(m), C.c_int32(n), ptr(off), ptr(idx), ptr(val), ptr(x), ptr(out), ptr(info)) == 0
    np.testing.assert_array_equal(out, orcbind.spmv(off, idx, val, x))
    assert (36 <= info[4] <= 40) if not transposed else (info[4] < 10), info  # (A^T: a column of twenty entries now and then crowds a step too)
