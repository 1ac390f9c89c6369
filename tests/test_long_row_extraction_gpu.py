"""GPU: with CUOPT_AMD_LONG_ROWS=1 rows of more than 2048 nonzeros are taken out of the layouts of THEIR matrix and multiplied in chunks
by kernels of their own (pdlp_device.hip "long rows"), every layout adds the result ahead of its epilogue (opt-in: on the
power-law / block-angular workloads the two extra launches cost more than the imbalance they remove, profiles/r03_long_rows.txt).  Such rows were compared at the long-row
tolerance before (fixed tree) and still are; everything else stays bit-identical."""
import numpy as np
import pytest

from cuopt_amd import capi, synthetic
from oracle import orcbind
from test_kernels_gpu import ragged_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def extraction_on(monkeypatch):
    monkeypatch.setenv("CUOPT_AMD_LONG_ROWS", "1")


def _products(p, dev, exact_short_rows=True):
    rng = np.random.default_rng(2)
    x, y = rng.standard_normal(p["n"]), rng.standard_normal(p["m"])
    to, ti, tv = orcbind.transpose(p["m"], p["n"], p["offsets"], p["indices"], p["values"])
    for got, ref, lens in ((dev.spmv(x, False, p["m"]), orcbind.spmv(p["offsets"], p["indices"], p["values"], x), np.diff(p["offsets"])),
                           (dev.spmv(y, True, p["n"]), orcbind.spmv(to, ti, tv, y), np.diff(to))):
        if exact_short_rows:
            np.testing.assert_array_equal(got[lens <= 128], ref[lens <= 128])
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12 * (1 + np.abs(ref).max()))
    return int((np.diff(p["offsets"]) > 2048).sum()), int((np.diff(to) > 2048).sum())


@pytest.mark.parametrize("layout", ["auto", "stream", "panel", "jag"])
@pytest.mark.parametrize("kind", ["powerlaw", "block_angular"])
def test_families_with_hub_rows(kind, layout, monkeypatch):
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", layout)
    monkeypatch.setenv("CUOPT_AMD_SLAB_BYTES", str(64 * 1024))
    p = synthetic.generate_structured(kind, m=200000, n=200000, k=10, seed=11)
    dev = capi.Device(p)
    la, lat = _products(p, dev)
    assert la + lat >= 1, "the family is supposed to have rows beyond the extraction threshold at this size"
    dev.close()


def test_ragged_rows_and_the_switch(monkeypatch):
    """rows of 2400 and 2049 nonzeros next to empty ones; with CUOPT_AMD_LONG_ROWS=0 the layouts keep them (same numbers to the
    tolerance, same decisions)"""
    p = ragged_problem(m=3000, n=2500)
    p["lb"] = np.zeros(p["n"])
    monkeypatch.setenv("CUOPT_AMD_SMALL", "0")
    got = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("CUOPT_AMD_LONG_ROWS", flag)
        dev = capi.Device(p)
        assert _products(p, dev)[0] == 2
        dev.close()
        r = capi.Solver(p, tol=0.0, iteration_limit=80).advance()
        got[flag] = (r["steps_taken"], r["attempted_steps"], r["primal_objective"])
    assert got["0"][:2] == got["1"][:2] and got["0"][2] == pytest.approx(got["1"][2], rel=1e-9, abs=1e-9)


def test_dense_segments_and_long_rows_together(monkeypatch):
    """a hub row that ALSO contains a dense run: the extracted remainder starts from the segment's share"""
    monkeypatch.setenv("CUOPT_AMD_DENSE", "1")
    rng = np.random.default_rng(3)
    m, n = 4000, 9000
    lens = rng.integers(2, 12, size=m)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    rows = np.repeat(np.arange(m), lens)
    cols = rng.integers(0, n, size=off[-1])
    hub = 17  # 600 consecutive columns + 3000 scattered ones
    rows = np.concatenate([rows, np.full(600, hub), np.full(3000, hub), np.arange(n) % m])
    cols = np.concatenate([cols, 2000 + np.arange(600), rng.choice(n, size=3000, replace=False), np.arange(n)])
    p = synthetic._finish(m, n, rows, cols, rng.standard_normal(len(rows)), rng, dict(seed=3, k=0, kind="hub", hard=False, band=0))
    dev = capi.Device(p)
    assert dev.dense_info()["on"]
    la, lat = _products(p, dev, exact_short_rows=False)  # (the columns under the dense run are split sums: tolerance)
    assert la == 1
    dev.close()
    r = capi.solve(p, method=1, tol=1e-6)
    assert r["status"] == "Optimal" and abs(r["objective"] - p["objective_star"]) <= 2e-5 * (1 + abs(p["objective_star"]))
