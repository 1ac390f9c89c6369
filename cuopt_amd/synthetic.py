"""Synthetic random sparse LPs with a KNOWN primal-dual optimal pair (SURVEY.md 8(d) family S(m,n,k,seed)).

    min c.x   s.t.  lo <= A x <= hi,  x >= 0
    A: m x n CSR, k nonzeros per row (distinct sorted columns, values ~ N(0,1)), every column used
    rows [0, m/2): equalities lo = hi = A x*;  rows [m/2, m): lo = A x* - s, hi = +inf
    x*: 50 % zeros, else U(0,1];  y*: N(0,1) on equalities, |N(0,1)| with 30 % zeros on '>=' rows
    s_i = 0 where y*_i > 0 else U(0,1);  z*_j = 0 where x*_j > 0 else U(0,1);  c = A^T y* + z*
so (x*, y*) satisfies the KKT conditions and obj* = c.x* = lo.y* is the known answer.
`hard=True` rescales rows by 10^U(-1,1) and columns by 10^U(-2,2) first, which makes PDLP need
thousands of iterations and several restarts.  `band=B` draws the columns of row i within +-B of the
diagonal position instead of uniformly (a STRUCTURED LP: staircase / time-expanded models look like this);
the uniform family is the gather-adversarial worst case for SpMV."""
import numpy as np


def _draw(rng, m, n, k, band, rows=None):
    """k column indices per row: uniform on [0, n), or within +-band of the row's scaled diagonal position"""
    rows = np.arange(m, dtype=np.int64) if rows is None else rows
    if not band:
        return rng.integers(0, n, size=(len(rows), k), dtype=np.int64)
    centre = (rows * n) // m
    width = min(2 * band + 1, n)
    lo = np.clip(centre - band, 0, n - width)  # the window slides at the edges instead of collapsing
    return lo[:, None] + rng.integers(0, width, size=(len(rows), k), dtype=np.int64)


def _columns(rng, m, n, k, band=0):
    if n > m * k:
        raise ValueError("need n <= m*k so that every column can get an entry")
    cols = _draw(rng, m, n, k, band)
    j = np.arange(n, dtype=np.int64)
    forced_rows, forced_slots = j % m, j // m
    cols[forced_rows, forced_slots] = j
    nforced = np.zeros(m, dtype=np.int64)
    np.add.at(nforced, forced_rows, 1)
    for _ in range(100):
        srt = np.sort(cols, axis=1)
        bad = np.nonzero((srt[:, 1:] == srt[:, :-1]).any(axis=1))[0]
        if bad.size == 0:
            return srt
        for i in bad:  # redraw only the free slots of the offending rows
            f = nforced[i]
            cols[i, f:] = _draw(rng, m, n, k, band, rows=np.array([i]))[0, : k - f]
    raise RuntimeError("could not draw distinct columns")


def generate(m, n, k=10, seed=1, hard=False, band=0):
    rng = np.random.default_rng(seed)
    cols = _columns(rng, m, n, k, band)
    vals = rng.standard_normal((m, k))
    if hard:
        rs = 10.0 ** rng.uniform(-1.0, 1.0, size=m)
        cs = 10.0 ** rng.uniform(-2.0, 2.0, size=n)
        vals = vals * rs[:, None] * cs[cols]
    offsets = (np.arange(m + 1, dtype=np.int64) * k).astype(np.int32)
    indices = cols.reshape(-1).astype(np.int32)
    values = np.ascontiguousarray(vals.reshape(-1))
    xs = np.where(rng.random(n) < 0.5, 0.0, 1.0 - rng.random(n))  # U(0,1]
    half = m // 2
    ys = rng.standard_normal(m)
    ys[half:] = np.abs(ys[half:]) * (rng.random(m - half) >= 0.3)
    ax = (vals * xs[cols]).sum(axis=1)
    lo, hi = ax.copy(), ax.copy()
    slack = np.where(ys[half:] > 0.0, 0.0, rng.random(m - half))
    lo[half:] = ax[half:] - slack
    hi[half:] = np.inf
    zs = np.where(xs > 0.0, 0.0, rng.random(n))
    aty = np.zeros(n)
    np.add.at(aty, cols.reshape(-1), (vals * ys[:, None]).reshape(-1))
    c = aty + zs
    return dict(m=m, n=n, offsets=offsets, indices=indices, values=values, c=c, lo=lo, hi=hi,
                lb=np.zeros(n), ub=np.full(n, np.inf), maximize=False, objective_offset=0.0,
                x_star=xs, y_star=ys, objective_star=float(c @ xs), seed=seed, k=k, hard=hard, band=band)


# the configurations BASELINE.json names
CONFIGS = {
    "tiny": dict(m=2000, n=2000, k=10, seed=3),
    "c2": dict(m=100_000, n=100_000, k=10, seed=1),        # 1e5 x 1e5, 1e6 nnz
    "c3": dict(m=1_000_000, n=1_000_000, k=10, seed=2),    # 1e6 x 1e6, 1e7 nnz
    "banded": dict(m=1_000_000, n=1_000_000, k=10, seed=2, band=2000),  # same size, structured columns
    "banded4": dict(m=4_000_000, n=4_000_000, k=10, seed=2, band=2000),  # 4e7 nnz banded: where the 8-GPU model of DESIGN 5 crosses 3.5x
    "c3x10": dict(m=10_000_000, n=10_000_000, k=10, seed=2),   # 1e7 x 1e7, 1e8 nnz: ten times C3 (robustness / capacity check)
}


def spmv_bytes(rows, cols, nnz):
    """algorithmic bytes of one CSR SpMV (fp64 values, int32 indices): SURVEY.md 8(d)"""
    return 12 * nnz + 4 * (rows + 1) + 8 * cols + 8 * rows


def iteration_bytes_min(m, n, nnz):
    """fused floor of one accepted PDHG iteration: 24 nnz + 4(m+n+2) + 8(14 n + 7 m)"""
    return 24 * nnz + 4 * (m + n + 2) + 8 * (14 * n + 7 * m)


# ---- structured families (stand-ins for the Mittelmann LPs of pdlp_test.cu:189-235, which cannot be fetched) ----------------
def _finish(m, n, rows, cols, vals, rng, meta):
    """CSR + a primal-dual optimal pair by construction (same recipe as generate()) from a ragged pattern"""
    import scipy.sparse as sp
    a = sp.csr_matrix((vals, (rows, cols)), shape=(m, n))
    a.sum_duplicates()
    a.sort_indices()
    xs = np.where(rng.random(n) < 0.5, 0.0, 1.0 - rng.random(n))
    half = m // 2
    ys = rng.standard_normal(m)
    ys[half:] = np.abs(ys[half:]) * (rng.random(m - half) >= 0.3)
    ax = a @ xs
    lo, hi = ax.copy(), ax.copy()
    slack = np.where(ys[half:] > 0.0, 0.0, rng.random(m - half))
    lo[half:] = ax[half:] - slack
    hi[half:] = np.inf
    zs = np.where(xs > 0.0, 0.0, rng.random(n))
    c = a.T @ ys + zs
    out = dict(m=m, n=n, offsets=a.indptr.astype(np.int32), indices=a.indices.astype(np.int32),
               values=np.ascontiguousarray(a.data, dtype=np.float64), c=c, lo=lo, hi=hi, lb=np.zeros(n),
               ub=np.full(n, np.inf), maximize=False, objective_offset=0.0, x_star=xs, y_star=ys,
               objective_star=float(c @ xs))
    out.update(meta)
    return out


def generate_structured(kind, m=1_000_000, n=1_000_000, k=10, seed=7):
    """kind = 'staircase'     time-expanded model: stages of 1000 rows x 1000 columns, a row couples its own stage (60 %)
                              with the next one (40 %)
              'block_angular' 100 independent diagonal blocks + 0.02 % linking rows that run through EVERY block
                              (thousands of nonzeros each: the long-row path) + 1 % linking columns
              'multiband'     three bands of +-500 columns at lags 0, +n/10 and -3n/10
              'powerlaw'      row lengths ~ Pareto(1.5) with mean ~k, capped at 20000, uniform columns (hub constraints)
              'dense_rows'    a random sparse LP (k - 2 nonzeros per row) + m / 25000 DENSE constraints that run through a
                              contiguous block of n / 20 variables each (budget / convexity / linking rows): ~17 % of the
                              nonzeros sit in dense row segments (index-free storage, pdlp_device.hip "dense")
    every column gets at least one entry; (x*, y*) is optimal by construction, objective_star is the known answer."""
    rng = np.random.default_rng(seed)
    r_forced = np.arange(n, dtype=np.int64) % m
    c_forced = np.arange(n, dtype=np.int64)
    if kind == "staircase":
        stage = 1000
        rows = np.repeat(np.arange(m, dtype=np.int64), k)
        st = (rows * n // m) // stage
        nxt = rng.random(len(rows)) < 0.4
        base = (st + nxt) * stage
        cols = np.minimum(base + rng.integers(0, stage, size=len(rows)), n - 1)
    elif kind == "block_angular":
        blocks = 100
        bw_r, bw_c = m // blocks, n // blocks
        nlink_rows = max(m // 5000, 1)
        rows = np.repeat(np.arange(m - nlink_rows, dtype=np.int64), k - 1)
        blk = rows // bw_r
        link_col = rng.random(len(rows)) < 0.01  # linking columns: the last 1 % of the columns
        cols = np.where(link_col, n - 1 - rng.integers(0, max(n // 100, 1), size=len(rows)),
                        np.minimum(blk, blocks - 1) * bw_c + rng.integers(0, bw_c, size=len(rows)))
        per_link = (m * k - len(rows)) // nlink_rows  # the remaining nonzeros go to the linking rows
        lr = np.repeat(np.arange(m - nlink_rows, m, dtype=np.int64), per_link)
        lc = rng.integers(0, n, size=len(lr))
        rows, cols = np.concatenate([rows, lr]), np.concatenate([cols, lc])
    elif kind == "multiband":
        # three far-apart bands (a time-expanded model with two lags, or a 3-D grid): no contiguous window of columns holds a row
        # block, but every row block re-uses a small SET of columns
        rows = np.repeat(np.arange(m, dtype=np.int64), k)
        centre = rows * n // m
        lag = rng.choice(np.array([0, 0, 0, 0, n // 10, n // 10, n // 10, -(3 * n) // 10, -(3 * n) // 10, -(3 * n) // 10]), size=len(rows))
        cols = (centre + lag + rng.integers(-500, 501, size=len(rows))) % n
    elif kind == "powerlaw":
        xm = k / 3.0
        lens = np.minimum((xm * (1.0 - rng.random(m)) ** (-1.0 / 1.5)).astype(np.int64) + 1, 20000)
        rows = np.repeat(np.arange(m, dtype=np.int64), lens)
        cols = rng.integers(0, n, size=len(rows))
    elif kind == "dense_rows":
        rows = np.repeat(np.arange(m, dtype=np.int64), max(k - 2, 1))
        cols = rng.integers(0, n, size=len(rows))
        nd, width = max(m // 25000, 2), max(n // 20, 300)
        d_rows = np.sort(rng.choice(m, size=nd, replace=False))
        starts = rng.integers(0, n - width, size=nd)
        rows = np.concatenate([rows, np.repeat(d_rows, width)])
        cols = np.concatenate([cols, (starts[:, None] + np.arange(width)[None, :]).reshape(-1)])
    else:
        raise ValueError(kind)
    rows = np.concatenate([rows, r_forced])
    cols = np.concatenate([cols, c_forced])
    vals = rng.standard_normal(len(rows))
    return _finish(m, n, rows, cols, vals, rng, dict(seed=seed, k=k, kind=kind, hard=False, band=0))


def generate_clustered(m, n, k=3, heavy=40, width=60, seed=11, empty_rows=None):
    """S(m, n, k) plus `heavy` rows that also run through `width` CONSECUTIVE columns each (a budget / convexity row among short
    ones): the rows the wide bins of the gather-free layout hand to single lanes (more than 7 entries of a row inside one step).
    empty_rows = (first, last): these rows lose all their entries (a whole bin without a product)"""
    rng = np.random.default_rng(seed)
    base = generate(m, n, k, seed=seed)
    rows = np.repeat(np.arange(m, dtype=np.int64), k)
    cols = base["indices"].astype(np.int64)
    if empty_rows is not None:
        keep = (rows < empty_rows[0]) | (rows >= empty_rows[1])
        rows, cols = rows[keep], cols[keep]
    h_rows = np.sort(rng.choice(m, size=heavy, replace=False))
    starts = rng.integers(0, n - width, size=heavy)
    rows = np.concatenate([rows, np.repeat(h_rows, width)])
    cols = np.concatenate([cols, (starts[:, None] + np.arange(width)[None, :]).reshape(-1)])
    vals = rng.standard_normal(len(rows))
    return _finish(m, n, rows, cols, vals, rng, dict(seed=seed, k=k, kind="clustered", hard=False, band=0, heavy_rows=h_rows))


def shuffled(p, seed=5):
    """The same LP under a seeded random row AND column permutation -- how a structured model arrives when its MPS file lists rows
    and columns in modelling order rather than in the order of its structure.  Row i of the result is row rp[i] of p, column j is
    column cp[j]; indices ascend inside every row; the known optimal pair moves along.  `shuffle` = (rp, cp) is kept for tests."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    m, n = p["m"], p["n"]
    rp, cp = rng.permutation(m), rng.permutation(n)
    a = sp.csr_matrix((p["values"], p["indices"], p["offsets"]), shape=(m, n))[rp][:, cp].tocsr()
    a.sort_indices()
    out = dict(p)
    out.update(offsets=a.indptr.astype(np.int32), indices=a.indices.astype(np.int32), values=np.ascontiguousarray(a.data, dtype=np.float64),
               c=np.ascontiguousarray(p["c"][cp]), lb=np.ascontiguousarray(p["lb"][cp]), ub=np.ascontiguousarray(p["ub"][cp]),
               lo=np.ascontiguousarray(p["lo"][rp]), hi=np.ascontiguousarray(p["hi"][rp]), shuffle=(rp, cp))
    if "x_star" in p:
        out["x_star"], out["y_star"] = p["x_star"][cp], p["y_star"][rp]
    return out
