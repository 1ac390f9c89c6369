"""GPU: the device-side set-up (cuopt_amd/csrc/kernels_setup.hip; pdlp_device.h "device-side set-up").  The reference transposes on
the device (cpp/src/mip/problem/problem.cu:277-309, raft csr_transpose) and hands its SpMV analysis to cusparseSpMV_preprocess
(cpp/src/linear_programming/cusparse_view.cu:92-115,254-265); here both are hand-written and pinned:
  * the primitives (stable radix sort of pairs, exclusive scan) against numpy;
  * A^T built on the device == the host transposition, bit for bit (which itself is pinned on scipy, test_capi_host.py);
  * the panel layout built on the device == the host construction's arrays, bit for bit (checksums of every array);
  * the ordering search: a band / staircase / block-angular LP whose rows AND columns arrive shuffled is recognised, the permuted
    pair on the device is P A Q and its transpose exactly (scipy), a random matrix is turned away;
  * SpMV of the permuted layout bit-exact against the oracle on the permuted CSR; a full solve's x, y (un-permuted by the interface)
    against the known optimum and the reference's termination inequalities on the ORIGINAL LP; warm-start snapshots of a reordered
    solver are in the caller's order and restore bit-exactly."""
import numpy as np
import pytest
import scipy.sparse as sp

from cuopt_amd import capi, synthetic
from oracle import orcbind
from conftest import set_tune
from test_solve_gpu import host_check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,bits", [(1, 8), (63, 8), (4096, 16), (4097, 20), (300_000, 20), (1_000_003, 32)])
def test_radix_sort_of_pairs_is_numpy_s_stable_sort(n, bits):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 2 ** min(bits, 31), size=n, dtype=np.int64).astype(np.uint32)
    if n > 100:
        keys[: n // 3] = keys[0]  # a long run of equal keys: stability is visible
    for vals in (None, rng.integers(0, 2 ** 31, size=n).astype(np.uint32)):
        ko, vo = capi.device_sort_pairs(keys, vals, bits=bits)
        order = np.argsort(keys, kind="stable")
        np.testing.assert_array_equal(ko, keys[order])
        np.testing.assert_array_equal(vo, order.astype(np.uint32) if vals is None else vals[order])


@pytest.mark.parametrize("n", [1, 4095, 4096, 4097, 1_000_003])
def test_exclusive_scan(n):
    a = np.random.default_rng(n).integers(0, 200, size=n).astype(np.int32)
    np.testing.assert_array_equal(capi.device_exclusive_scan(a), np.concatenate([[0], np.cumsum(a)]).astype(np.int32))


def _ragged_lp(seed=3):
    """rows of 0 ... 5000 entries, empty columns, unsorted would-be duplicates removed: what a transposition must survive"""
    rng = np.random.default_rng(seed)
    m, n = 5000, 7000
    lens = np.minimum((2.0 * (1.0 - rng.random(m)) ** (-1.0 / 1.3)).astype(np.int64), 5000)
    lens[rng.random(m) < 0.1] = 0
    rows = np.repeat(np.arange(m), lens)
    cols = rng.integers(0, n - 500, size=len(rows))  # the last 500 columns stay empty
    a = sp.csr_matrix((rng.standard_normal(len(rows)), (rows, cols)), shape=(m, n))
    a.sum_duplicates()
    a.sort_indices()
    return dict(m=m, n=n, offsets=a.indptr.astype(np.int32), indices=a.indices.astype(np.int32), values=a.data.astype(np.float64))


@pytest.mark.parametrize("which", ["tiny", "ragged", "c2"])
def test_transpose_on_the_device_is_the_host_transposition(which):
    p = _ragged_lp() if which == "ragged" else synthetic.generate(**synthetic.CONFIGS[which])
    an = capi.Analysis(p, reorder=False)
    to, ti, tv = an.download(transposed=True)
    ho, hi, hv = capi.csr_transpose(p["m"], p["n"], p["offsets"], p["indices"], p["values"])
    np.testing.assert_array_equal(to, ho)
    np.testing.assert_array_equal(ti, hi)
    np.testing.assert_array_equal(tv, hv)
    ao, ai, av = an.download()
    np.testing.assert_array_equal(ai, p["indices"])
    an.close()


@pytest.mark.parametrize("family", ["random", "powerlaw"])
def test_panels_built_on_the_device_are_the_host_construction(family, monkeypatch):
    """every array of the panel layout (row0, tile pointers, 16-bit row pointers, columns, permutation; the long-tail variant's packed
    entries) and A^T itself: FNV-1a checksums of the host path (pdlpdev_create: host transposition + build_panels + uploads) against
    the device path (pdlpdev_analyze + pdlpdev_create_from_analysis)"""
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "panel")
    set_tune(monkeypatch, slab_bytes=64 * 1024, panel_nnz=6000)
    p = synthetic.generate(60000, 50000, 9, seed=4) if family == "random" else synthetic.generate_structured("powerlaw", m=60000, n=60000, k=8, seed=5)
    host = capi.Device(p)
    dev = capi.Device(p, analysis=capi.Analysis(p, reorder=False))
    a, b = host.layout_checksums(), dev.layout_checksums()
    assert int(a[15]) & 3 == 3, "both sides in panels"
    if family == "powerlaw":
        assert int(a[15]) & 0x10, "the long-tail variant on the A side"
    np.testing.assert_array_equal(a, b)
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(p["n"]), rng.standard_normal(p["m"])
    np.testing.assert_array_equal(host.spmv(x, False, p["m"]), dev.spmv(x, False, p["m"]))
    np.testing.assert_array_equal(host.spmv(y, True, p["n"]), dev.spmv(y, True, p["n"]))
    host.close(), dev.close()


@pytest.mark.parametrize("shape", ["square", "wide", "ragged", "square_sorted_counts"])
def test_gather_free_layout_built_on_the_device_is_the_host_construction(shape, monkeypatch):
    """every array of the gather-free layout (phase-P order + local columns + pieces, the rows by length, the jagged diagonals of
    positions, bins, groups, P workgroups): FNV-1a checksums of the host construction (build_pb, CUOPT_AMD_TUNE=pb_device=0) against
    the device construction (sorts and scans per bin), on both matrices; then the products and a solve"""
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "pb")
    if shape == "square_sorted_counts":  # the chunk sizes from the sort instead of the LDS histogram (the path of > 12 288 panels)
        set_tune(monkeypatch, pb_hist=0)
    if shape.startswith("square"):
        p = synthetic.generate(120000, 100000, 9, seed=4)
    elif shape == "wide":
        p = synthetic.generate(40000, 300000, 8, seed=6)  # (37 panels of 8192 columns on the A side, rows of ~60 entries on it)
    else:
        p = synthetic.generate_structured("powerlaw", m=60000, n=60000, k=8, seed=5)  # (row lengths up to the forced limit)
    set_tune(monkeypatch, pb_device=0)
    try:
        host = capi.Device(p, analysis=capi.Analysis(p, reorder=False))
    except capi.CuOptError as e:
        assert shape == "ragged", e  # a row beyond the layout's limit: both constructions must refuse alike
        set_tune(monkeypatch, pb_device=None)
        with pytest.raises(capi.CuOptError):
            capi.Device(p, analysis=capi.Analysis(p, reorder=False))
        return
    set_tune(monkeypatch, pb_device=None)
    dev = capi.Device(p, analysis=capi.Analysis(p, reorder=False))
    a, b = host.layout_checksums(), dev.layout_checksums()
    assert (int(a[15]) >> 6) & 3 == 3, "both sides gather-free"
    np.testing.assert_array_equal(a, b)
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(p["n"]), rng.standard_normal(p["m"])
    np.testing.assert_array_equal(host.spmv(x, False, p["m"]), dev.spmv(x, False, p["m"]))
    np.testing.assert_array_equal(host.spmv(y, True, p["n"]), dev.spmv(y, True, p["n"]))
    host.close(), dev.close()
    r = capi.Solver(p, tol=1e-4, iteration_limit=4000)
    assert r.device.layout()["A"]["layout"] == "pb"
    out = r.advance()
    assert out["status_name"] == "Optimal" and abs(out["primal_objective"] - p["objective_star"]) <= 2e-3 * (1 + abs(p["objective_star"]))
    r.close()


@pytest.mark.parametrize("kind", ["banded", "staircase", "block_angular", "multiband", "random_forced", "powerlaw_forced"])
def test_jagged_layout_built_on_the_device_is_the_host_construction(kind, monkeypatch):
    """every array of the jagged layout (block boundaries, the column sets -- contiguous windows and sorted lists --, row descriptors,
    16-bit LDS slots and the permutation along the jagged diagonals, the long-row lists): FNV-1a checksums of the host construction
    (build_jag, CUOPT_AMD_TUNE=jag_device=0) against the device construction, on both matrices; then the products"""
    if kind.endswith("_forced"):
        monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "jag")
        p = synthetic.generate(150000, 140000, 9, seed=4) if kind == "random_forced" else synthetic.generate_structured("powerlaw", m=140000, n=140000, k=8, seed=5)
    elif kind == "banded":
        p = synthetic.generate(262144, 262144, 10, seed=2, band=2000)
    else:
        p = synthetic.generate_structured(kind, m=262144, n=262144, k=10, seed=7)
    set_tune(monkeypatch, jag_device=0)
    host = capi.Device(p, analysis=capi.Analysis(p, reorder=False))
    set_tune(monkeypatch, jag_device=None)
    dev = capi.Device(p, analysis=capi.Analysis(p, reorder=False))
    a, b = host.layout_checksums(), dev.layout_checksums()
    assert (int(a[15]) >> 2) & 3 == 3, ("both sides jagged", host.layout())
    np.testing.assert_array_equal(a, b)
    assert host.layout() == dev.layout()
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(p["n"]), rng.standard_normal(p["m"])
    np.testing.assert_array_equal(host.spmv(x, False, p["m"]), dev.spmv(x, False, p["m"]))
    np.testing.assert_array_equal(host.spmv(y, True, p["n"]), dev.spmv(y, True, p["n"]))
    host.close(), dev.close()


def _family(kind, m):
    if kind == "banded":
        return synthetic.generate(m, m, 10, seed=2, band=2000)
    return synthetic.generate_structured(kind, m=m, n=m, k=10, seed=7)


@pytest.fixture(scope="module", params=["banded", "staircase", "block_angular"])
def shuffled_lp(request):
    return request.param, synthetic.shuffled(_family(request.param, 262144), seed=5)


def test_the_analysis_pass_finds_the_order_and_builds_p_a_q(shuffled_lp):
    kind, q = shuffled_lp
    an = capi.Analysis(q, reorder=True)
    info = an.info()
    assert info["permuted"], info
    assert info["method"] == ("groups" if kind == "block_angular" else "chains"), info
    assert min(info["estimate_natural"]) < 0.35  # the shuffled matrix itself is no case for the jagged layout
    rn2o, cn2o = an.maps()
    assert sorted(rn2o.tolist()) == list(range(q["m"])) and sorted(cn2o.tolist()) == list(range(q["n"]))
    a = sp.csr_matrix((q["values"], q["indices"], q["offsets"]), shape=(q["m"], q["n"]))
    ref = a[rn2o][:, cn2o].tocsr()
    ref.sort_indices()
    off, idx, val = an.download()
    np.testing.assert_array_equal(off, ref.indptr)
    np.testing.assert_array_equal(idx, ref.indices)
    np.testing.assert_array_equal(val, ref.data)
    reft = ref.T.tocsr()
    reft.sort_indices()
    off, idx, val = an.download(transposed=True)
    np.testing.assert_array_equal(off, reft.indptr)
    np.testing.assert_array_equal(idx, reft.indices)
    np.testing.assert_array_equal(val, reft.data)
    # the same matrix, the same order: nothing in the search depends on timing
    an2 = capi.Analysis(q, reorder=True)
    r2, c2 = an2.maps()
    np.testing.assert_array_equal(r2, rn2o)
    np.testing.assert_array_equal(c2, cn2o)
    an.close(), an2.close()


def test_the_estimate_on_the_device_is_the_host_s(shuffled_lp, monkeypatch):
    """k_jag_estimate restates build_jag's sampled estimate (greedy row blocks, range / list pricing) on the device: the same numbers
    as the host evaluation of the same samples, for the matrix as given and for the accepted candidate"""
    kind, q = shuffled_lp
    infos = []
    for host in (1, None):
        set_tune(monkeypatch, estimate_host=host)
        an = capi.Analysis(q, reorder=True)
        infos.append(an.info())
        an.close()
    assert infos[0] == infos[1], infos


def test_a_random_matrix_is_turned_away():
    p = synthetic.generate(262144, 262144, 10, seed=9)
    an = capi.Analysis(p, reorder=True)
    info = an.info()
    assert not info["permuted"] and an.maps() is None, info
    an.close()


def test_spmv_of_the_permuted_layout_is_bit_exact_against_the_oracle_on_the_permuted_csr(shuffled_lp):
    kind, q = shuffled_lp
    an = capi.Analysis(q, reorder=True)
    rn2o, cn2o = an.maps()
    off, idx, val = an.download()
    to, ti, tv = an.download(transposed=True)
    pq = dict(q, c=q["c"][cn2o], lb=q["lb"][cn2o], ub=q["ub"][cn2o], lo=q["lo"][rn2o], hi=q["hi"][rn2o])
    dev = capi.Device(pq, analysis=an)
    lay = dev.layout()
    assert lay["A"]["layout"] == "jag" and lay["At"]["layout"] == "jag", lay
    rng = np.random.default_rng(3)
    x, y = rng.standard_normal(q["n"]), rng.standard_normal(q["m"])
    for got, ref, lens in ((dev.spmv(x, False, q["m"]), orcbind.spmv(off, idx, val, x), np.diff(off)),
                           (dev.spmv(y, True, q["n"]), orcbind.spmv(to, ti, tv, y), np.diff(to))):
        np.testing.assert_array_equal(got[lens <= 128], ref[lens <= 128])  # short rows: left to right in the permuted CSR's order
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12 * (1 + np.abs(ref).max()))
    dev.close()


def test_a_reordered_solve_answers_in_the_caller_s_order(shuffled_lp):
    """x, y, reduced costs of a solve on P A Q come back un-permuted: the known optimum of the ORIGINAL (shuffled) LP, the
    reference's termination inequalities re-verified on the host with the caller's arrays; same with the ordering search off"""
    kind, q = shuffled_lp
    s = capi.Solver(q, mode=1, tol=1e-5)
    r = s.advance()
    info = s.reorder_info(maps=True)
    assert info["reordered"] and r["status_name"] == "Optimal", (info, r["status_name"])
    x, y, z = s.solution()
    scale = 1 + abs(q["objective_star"])
    assert abs(r["primal_objective"] - q["objective_star"]) <= 2e-4 * scale
    host_check(q, dict(r, x=x, y=y, reduced_cost=z, objective=r["primal_objective"]), eps=1e-5)
    # warm-start snapshots are in the caller's order: restoring one into a fresh (reordered) solver continues bit for bit
    # (the pattern of pdlp_test.cu:803-854: iterations to a coarse tolerance + iterations from its snapshot = iterations from scratch)
    full = capi.Solver(q, mode=1, tol=1e-4)
    r_full = full.advance()
    a = capi.Solver(q, mode=1, tol=1e-2)
    ra = a.advance()
    ws = a.get_warm_start()
    assert ws["current_primal_solution"].shape == (q["n"],) and ws["total_pdlp_iterations"] == ra["steps_taken"]
    b = capi.Solver(q, mode=1, tol=1e-4, warm_start=ws)
    rb = b.advance()
    assert ra["steps_taken"] + rb["steps_taken"] == r_full["steps_taken"]
    assert rb["primal_objective"] == r_full["primal_objective"]
    np.testing.assert_array_equal(full.solution()[0], b.solution()[0])
    full.close()
    s.close(), a.close(), b.close()


def test_without_the_search_the_same_lp_takes_the_panels(shuffled_lp, monkeypatch):
    kind, q = shuffled_lp
    set_tune(monkeypatch, reorder=0)
    s = capi.Solver(q, mode=1, tol=1e-5)
    r = s.advance()
    assert not s.reorder_info()["reordered"] and r["status_name"] == "Optimal"
    assert abs(r["primal_objective"] - q["objective_star"]) <= 2e-4 * (1 + abs(q["objective_star"]))
    s.close()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_solve_of_a_reordered_lp(shuffled_lp, world, monkeypatch):
    """row-block sharding AFTER the reordering (every rank finds the same order, slices P A Q): the in-process communicator on one
    device, same optimum, y gathered back into the caller's row order"""
    kind, q = shuffled_lp
    set_tune(monkeypatch, soft_communicator=1)
    r = capi.solve(q, method=1, tol=1e-5, amd_num_gpus=world)
    assert r["status"] == "Optimal", r["status"]
    assert abs(r["objective"] - q["objective_star"]) <= 2e-4 * (1 + abs(q["objective_star"]))
    host_check(q, r, eps=1e-5)
