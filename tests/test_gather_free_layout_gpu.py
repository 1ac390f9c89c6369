"""GPU: the gather-free SpMV layout ("pb", spmv_pb.hpp): several source panels and bins, both piece sizes, both panel
widths, the automatic choice for gathered vectors beyond the panels' 16 slabs, a full solve."""
import numpy as np
import pytest

from conftest import set_tune
from cuopt_amd import capi, synthetic
from oracle import orcbind

pytestmark = pytest.mark.gpu


def _both_products_bit_exact(p, dev, seed=1):
    rng = np.random.default_rng(seed)
    x, y = rng.standard_normal(p["n"]), rng.standard_normal(p["m"])
    to, ti, tv = orcbind.transpose(p["m"], p["n"], p["offsets"], p["indices"], p["values"])
    np.testing.assert_array_equal(dev.spmv(x, False, p["m"]), orcbind.spmv(p["offsets"], p["indices"], p["values"], x))
    np.testing.assert_array_equal(dev.spmv(y, True, p["n"]), orcbind.spmv(to, ti, tv, y))


@pytest.mark.parametrize("shape", [(60000, 50000, 10), (200000, 30000, 3), (30000, 200000, 16)], ids=["square", "tall", "wide"])
def test_products_are_bit_exact_and_the_solve_reaches_the_optimum(shape, monkeypatch):
    """several 8192-column panels and dozens of bins on both sides; chunks of a few entries (4-entry pieces) on the tall / wide
    matrices, longer ones (8-entry pieces) on the square one"""
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "pb")
    m, n, k = shape
    p = synthetic.generate(m, n, k, seed=23)
    dev = capi.Device(p)
    lay = dev.layout()
    assert lay["A"]["layout"] == lay["At"]["layout"] == "pb" and lay["A"]["workgroups"] > 8
    _both_products_bit_exact(p, dev)
    r = capi.solve(p, method=1, tol=1e-6)
    assert r["status"] == "Optimal"
    assert abs(r["objective"] - p["objective_star"]) <= 2e-5 * (1 + abs(p["objective_star"]))
    # the same decisions as the CSR stream layout over the first iterations (only the grouping of the reduction partials differs)
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "stream")
    a = capi.Solver(p, tol=0.0, iteration_limit=40).advance()
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "pb")
    b = capi.Solver(p, tol=0.0, iteration_limit=40).advance()
    assert (a["steps_taken"], a["attempted_steps"]) == (b["steps_taken"], b["attempted_steps"])
    assert b["step_size"] == pytest.approx(a["step_size"], rel=1e-9)


def test_rows_of_every_length_and_empty_rows(monkeypatch):
    """empty rows, rows of hundreds and of two thousand nonzeros: one lane adds up its row left to right, so EVERY row -- not only
    those up to 128 nonzeros as in the other layouts -- is bit-identical to the sequential sum"""
    from test_kernels_gpu import ragged_problem
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "pb")
    p = ragged_problem(m=3000, n=2500)
    dev = capi.Device(p)
    assert dev.layout()["A"]["layout"] == "pb"
    _both_products_bit_exact(p, dev)


def test_a_matrix_the_layout_cannot_hold_is_refused_loudly(monkeypatch):
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "pb")
    rng = np.random.default_rng(0)
    m, n = 50, 9000
    lens = np.full(m, 4)
    lens[7] = 8000  # more than half a bin in one row
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = np.concatenate([np.sort(rng.choice(n, size=l, replace=False)) for l in lens]).astype(np.int32)
    p = dict(m=m, n=n, offsets=off, indices=idx, values=rng.standard_normal(off[-1]), c=np.ones(n), lo=np.full(m, -1.0), hi=np.full(m, 1.0),
             lb=np.zeros(n), ub=np.ones(n))
    with pytest.raises(Exception, match="gather-free"):
        capi.Device(p)


@pytest.mark.parametrize("shape", [(200000, 30000, 3), (30000, 200000, 16), (400000, 400000, 6), (1500000, 4400000, 3)], ids=["tall", "wide", "square", "short_chunks"])
@pytest.mark.parametrize("where", ["device", "host"])
def test_wide_bins_products_are_bit_exact_and_the_solve_reaches_the_optimum(shape, where, monkeypatch):
    """the geometry 'auto' takes beyond 2 M columns (bins of 8192 rows whose accumulators live in LDS, the image streamed in steps of
    1024 products, one addition per row and level), forced here on matrices of test size: both products bit-identical to the sequential
    CSR sums, the first 40 iterations take the stream layout's decisions, a solve reaches the optimum -- with the layout built on the
    device and on the host"""
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "pb")
    set_tune(monkeypatch, pb_wide=1, pb_device=None if where == "device" else 0)
    m, n, k = shape
    p = synthetic.generate(m, n, k, seed=23)
    dev = capi.Device(p)
    lay = dev.layout()
    assert lay["A"]["layout"] == lay["At"]["layout"] == "pb"
    assert lay["A"]["workgroups"] == -(-m // 8192) and lay["At"]["workgroups"] == -(-n // 8192), "bins of 8192 rows: the wide geometry on both sides"
    # ("short_chunks": 269 panels of 16384 columns x 184 bins, chunks of ~90 / ~45 entries: 8-entry pieces, and a bin's rounding to whole steps shows)
    assert shape[1] == 4400000 or (lay["A"]["padding_pct"] <= 12 and lay["At"]["padding_pct"] <= 12)
    _both_products_bit_exact(p, dev)
    _both_products_bit_exact(p, dev, seed=2)
    dev.close()
    if shape[1] == 4400000:
        return  # (the geometry's case; a solve of this LP to 1e-6 takes minutes and adds nothing to it)
    r = capi.solve(p, method=1, tol=1e-6)
    assert r["status"] == "Optimal"
    assert abs(r["objective"] - p["objective_star"]) <= 2e-5 * (1 + abs(p["objective_star"]))
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "stream")
    a = capi.Solver(p, tol=0.0, iteration_limit=40).advance()
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "pb")
    b = capi.Solver(p, tol=0.0, iteration_limit=40).advance()
    assert (a["steps_taken"], a["attempted_steps"]) == (b["steps_taken"], b["attempted_steps"])
    assert b["step_size"] == pytest.approx(a["step_size"], rel=1e-9)


def test_wide_bins_built_on_the_device_are_the_host_construction(monkeypatch):
    """every array of the wide geometry (phase-P order, local columns, pieces, slot words with their levels, the steps' levels, bins,
    P workgroups): FNV-1a checksums of build_pb_wide against the device construction (segment counts, ballots, ds_min rounds)"""
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "pb")
    for shape, seed in (((200000, 30000, 3), 4), ((30000, 200000, 16), 6), ((300000, 250000, 5), 8), ((1500000, 4400000, 3), 9)):
        p = synthetic.generate(*shape, seed=seed)
        set_tune(monkeypatch, pb_wide=1, pb_device=0)
        host = capi.Device(p, analysis=capi.Analysis(p, reorder=False))
        set_tune(monkeypatch, pb_wide=1, pb_device=None)
        dev = capi.Device(p, analysis=capi.Analysis(p, reorder=False))
        a, b = host.layout_checksums(), dev.layout_checksums()
        assert (int(a[15]) >> 6) & 3 == 3, "both sides gather-free"
        assert host.layout()["A"]["workgroups"] == -(-shape[0] // 8192)
        np.testing.assert_array_equal(a, b)
        host.close(), dev.close()


def test_a_matrix_the_wide_bins_cannot_hold_takes_the_other_geometry(monkeypatch):
    """forty entries per row over three panels make chunks of more than 65535 entries: both constructions fall back to the
    image-in-LDS bins, and agree"""
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "pb")
    p = synthetic.generate(20000, 20000, 40, seed=23)
    sums = []
    for pb_device in (0, None):
        set_tune(monkeypatch, pb_wide=1, pb_device=pb_device)
        dev = capi.Device(p, analysis=capi.Analysis(p, reorder=False))
        lay = dev.layout()
        assert lay["A"]["layout"] == "pb" and lay["A"]["workgroups"] > -(-20000 // 8192)
        _both_products_bit_exact(p, dev)
        sums.append(dev.layout_checksums())
        dev.close()
    np.testing.assert_array_equal(sums[0], sums[1])


def test_wide_bins_with_serial_rows(monkeypatch):
    """short rows + 40 rows through 60 consecutive columns: the long rows leave the steps of their bins and are summed by single lanes
    (PbView ser_*): products bit-exact on both sides, the device construction == the host's, the solve reaches the optimum"""
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "pb")
    # (the third LP: rows 8192 ... 16383 -- the whole second bin of A -- have no entry at all)
    for p in (synthetic.generate_clustered(200000, 30000, 3, heavy=40, width=60, seed=11), synthetic.generate(60000, 50000, 10, seed=23),
              synthetic.generate_clustered(100000, 30000, 3, heavy=0, width=1, seed=12, empty_rows=(8192, 16384))):
        _serial_rows_case(p, monkeypatch)


def _serial_rows_case(p, monkeypatch):
    sums = []
    for pb_device in (0, None):
        set_tune(monkeypatch, pb_wide=1, pb_device=pb_device)
        dev = capi.Device(p, analysis=capi.Analysis(p, reorder=False))
        lay = dev.layout()
        assert lay["A"]["layout"] == lay["At"]["layout"] == "pb" and lay["A"]["workgroups"] == -(-p["m"] // 8192), lay
        _both_products_bit_exact(p, dev)
        _both_products_bit_exact(p, dev, seed=3)
        sums.append(dev.layout_checksums())
        dev.close()
    np.testing.assert_array_equal(sums[0], sums[1])
    r = capi.solve(p, method=1, tol=1e-6)
    assert r["status"] == "Optimal"
    assert abs(r["objective"] - p["objective_star"]) <= 2e-5 * (1 + abs(p["objective_star"]))
