"""GPU: K small LPs in K workgroups of one launch (pdlpdev_small_batch_*, cuoptamd_batch_* over resident solvers -- BASELINE config 5
at branch-and-bound scale; the reference's counterpart is the thread pool of cython_solve.cu:264-296 / the relaxation streams of
relaxed_lp.cu:53-127).  The contract: every LP of the batch gets, BIT FOR BIT, what its own Solver.advance gives it -- results,
iterates, verdicts -- whatever the other members do (other matrices, other resident tiers, finishing early, infeasible, out of budget)."""
import numpy as np
import pytest

from cuopt_amd import capi, synthetic

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]

LIMIT = 6000  # a perturbed relaxation may be infeasible: nobody iterates forever (IterationLimit is a verdict like any other)

SKIP_FIELDS = ("setup_seconds", "loop_seconds")


def relaxation(golden_problems, name):
    p = dict(golden_problems[name]["problem"])
    p.pop("var_types", None)
    return p


def perturbed(p, rng, share=0.2):
    """the MIP heuristics' move: some finite variable ranges tightened (relaxed_lp.cu:74-108 re-solves under other bounds)"""
    q = dict(p)
    lb, ub = np.array(p["lb"], dtype=float), np.array(p["ub"], dtype=float)
    pick = rng.random(len(lb)) < share
    both = pick & np.isfinite(lb) & np.isfinite(ub)
    with np.errstate(invalid="ignore"):
        mid = 0.5 * (lb + ub)
    up = rng.random(len(lb)) < 0.5
    lb = np.where(both & up, mid, lb)
    ub = np.where(both & ~up, mid, ub)
    q["lb"], q["ub"] = lb, ub
    return q


def family(golden_problems, count, seed=0):
    """`count` LPs over four different matrices and all three resident tiers"""
    rng = np.random.default_rng(seed)
    base = [relaxation(golden_problems, "mip-50v-10-free-bound-relaxation"), relaxation(golden_problems, "mip-neos5-free-bound-relaxation"),
            relaxation(golden_problems, "afiro"), synthetic.generate(1000, 1000, 8, seed=4), synthetic.generate(1800, 2000, 2, seed=5),
            synthetic.generate(200, 500, 6, seed=6)]
    out = []
    for i in range(count):
        p = base[i % len(base)]
        out.append(p if i < len(base) else perturbed(p, rng))
    return out


def same(a, b):
    for k, v in a.items():
        if k in SKIP_FIELDS:
            continue
        w = b[k]
        assert v == w or (isinstance(v, float) and np.isnan(v) and np.isnan(w)), (k, v, w)


def singles(problems, chunks=(2 ** 31 - 1,), **kw):
    out = []
    for p in problems:
        s = capi.Solver(p, **kw)
        r = None
        for c in chunks:
            r = s.advance(c)
        out.append((r, s.solution()))
        s.close()
    return out


@pytest.mark.parametrize("K", [2, 64, 256])
def test_each_lp_of_the_batch_gets_its_own_solve_bit_for_bit(golden_problems, K):
    problems = family(golden_problems, K, seed=K)
    want = singles(problems, tol=1e-6, iteration_limit=LIMIT)
    solvers = [capi.Solver(p, tol=1e-6, iteration_limit=LIMIT) for p in problems]
    for s in solvers:
        assert s.device.layout()["resident"]
    batch = capi.SmallBatch(solvers)
    got = batch.advance()
    assert len({r["steps_taken"] for r in got}) > 1  # the members finish at different times and rest
    for l, s in enumerate(solvers):
        same(want[l][0], got[l])
        for u, v in zip(want[l][1], s.solution()):
            np.testing.assert_array_equal(u, v)
    batch.close()
    for s in solvers:
        s.close()


def test_mixed_verdicts_and_settings(golden_problems):
    """an infeasible member (detected by its own infeasibility evaluation), one that runs out of iterations, one under another preset,
    one with per-constraint residuals -- next to ordinary ones"""
    rng = np.random.default_rng(3)
    base = relaxation(golden_problems, "mip-50v-10-free-bound-relaxation")
    from test_solve_gpu import infeasible_lp_of_the_c_api_test
    infeasible = infeasible_lp_of_the_c_api_test()  # (the reference's own infeasible LP, c_api_test.c:625-757)
    members = [(base, dict(tol=1e-6)), (infeasible, dict(tol=1e-6, detect_infeasibility=1, iteration_limit=20000)), (perturbed(base, rng), dict(tol=1e-9, iteration_limit=170)),
               (synthetic.generate(1000, 1000, 8, seed=4), dict(tol=1e-6, mode=0)), (perturbed(base, rng), dict(tol=1e-5, per_constraint_residual=1, iteration_limit=LIMIT)),
               (synthetic.generate(200, 500, 6, seed=6), dict(tol=1e-6, mode=3)), (relaxation(golden_problems, "afiro"), dict(tol=1e-8))]
    want = []
    for p, kw in members:
        s = capi.Solver(p, **kw)
        want.append((s.advance(), s.solution()))
        s.close()
    assert want[1][0]["status_name"] == "PrimalInfeasible" and want[2][0]["status_name"] == "IterationLimit" and want[0][0]["status_name"] == "Optimal"
    solvers = [capi.Solver(p, **kw) for p, kw in members]
    batch = capi.SmallBatch(solvers)
    got = batch.advance()
    for l, s in enumerate(solvers):
        same(want[l][0], got[l])
        for u, v in zip(want[l][1], s.solution()):
            np.testing.assert_array_equal(u, v)
    batch.close()


def test_budgets_resume_where_they_stopped(golden_problems):
    """advance in pieces of 90 iterations (budget ends between and ON major iterations, right after restarts): same pieces as a single solver"""
    problems = family(golden_problems, 12, seed=5)
    chunks = (90, 90, 120, 2 ** 31 - 1)
    want = singles(problems, chunks=chunks, tol=1e-6, iteration_limit=LIMIT)
    solvers = [capi.Solver(p, tol=1e-6, iteration_limit=LIMIT) for p in problems]
    batch = capi.SmallBatch(solvers)
    for c in chunks:
        got = batch.advance(c)
    for l, s in enumerate(solvers):
        same(want[l][0], got[l])
        for u, v in zip(want[l][1], s.solution()):
            np.testing.assert_array_equal(u, v)
    batch.close()


def test_re_solves_through_one_batch(golden_problems):
    """config 5's pattern: the same solvers are reset to new bounds, warm-started from the previous primal / dual, and advanced again
    through the SAME batch object (relaxed_lp.cu:74-108)"""
    rng = np.random.default_rng(11)
    base = relaxation(golden_problems, "mip-50v-10-free-bound-relaxation")
    K = 16
    rounds = [[perturbed(base, rng, share=0.05 * (r + 1)) for _ in range(K)] for r in range(3)]
    ones = [capi.Solver(base, tol=1e-5, iteration_limit=LIMIT) for _ in range(K)]
    many = [capi.Solver(base, tol=1e-5, iteration_limit=LIMIT) for _ in range(K)]
    batch = capi.SmallBatch(many)
    prev = [None] * K
    for r, lps in enumerate(rounds):
        want = []
        for l, s in enumerate(ones):
            s.reset(lb=lps[l]["lb"], ub=lps[l]["ub"], init_x=None if prev[l] is None else prev[l][0], init_y=None if prev[l] is None else prev[l][1])
            want.append((s.advance(), s.solution()))
        # the batch's own reset and read-back: one launch each for all K (cuoptamd_batch_reset / cuoptamd_batch_get_solutions)
        batch.reset(lb=[q["lb"] for q in lps], ub=[q["ub"] for q in lps], init_x=None if r == 0 else [v[0] for v in prev],
                    init_y=None if r == 0 else [v[1] for v in prev])
        got = batch.advance()
        sols = batch.solutions()
        for l, s in enumerate(many):
            same(dict(want[l][0], setup_seconds=0), dict(got[l], setup_seconds=0))
            for u, v, w in zip(want[l][1], sols[l], s.solution()):
                np.testing.assert_array_equal(u, v)
                np.testing.assert_array_equal(u, w)
            prev[l] = sols[l]
    batch.close()


def test_branching_on_the_device_equals_the_trip_through_the_host(golden_problems):
    """cuoptamd_batch_branch (one variable's bounds + a start from the solver's own last solution, all on the device) against
    Solver.reset(full bounds, the solution read back) per solver: same re-solves bit for bit, over three levels of a tree"""
    rng = np.random.default_rng(23)
    base = relaxation(golden_problems, "mip-50v-10-free-bound-relaxation")
    K = 24
    ones = [capi.Solver(base, tol=1e-5, iteration_limit=LIMIT) for _ in range(K)]
    many = [capi.Solver(base, tol=1e-5, iteration_limit=LIMIT) for _ in range(K)]
    batch = capi.SmallBatch(many)
    lbs, ubs = [np.array(base["lb"], float) for _ in range(K)], [np.array(base["ub"], float) for _ in range(K)]
    want = [(s.advance(), s.solution()) for s in ones]
    batch.advance()
    for level in range(3):
        views = batch.solution_views()
        var, lo, hi = np.full(K, -1, np.int32), np.zeros(K), np.zeros(K)
        for l in range(K):
            np.testing.assert_array_equal(views[l][0], want[l][1][0])
            if l % 5 == 4:
                continue  # this node is re-solved as it is
            x = views[l][0]
            cand = np.flatnonzero(np.isfinite(lbs[l]) & np.isfinite(ubs[l]) & (ubs[l] - lbs[l] >= 1.0))
            j = int(rng.choice(cand))
            if l % 2:
                lbs[l][j] = min(np.ceil(x[j]), ubs[l][j])
            else:
                ubs[l][j] = max(np.floor(x[j]), lbs[l][j])
            var[l], lo[l], hi[l] = j, lbs[l][j], ubs[l][j]
        for l, s in enumerate(ones):
            s.reset(lb=lbs[l], ub=ubs[l], init_x=want[l][1][0], init_y=want[l][1][1])
        want = [(s.advance(), s.solution()) for s in ones]
        batch.branch(var, lo, hi)
        got = batch.advance()
        sols = batch.solutions()
        for l in range(K):
            same(dict(want[l][0], setup_seconds=0), dict(got[l], setup_seconds=0))
            for u, v in zip(want[l][1], sols[l]):
                np.testing.assert_array_equal(u, v)
    batch.close()

def test_who_is_turned_away(golden_problems):
    big = synthetic.generate(3000, 3000, 6, seed=2)  # not resident
    small = relaxation(golden_problems, "afiro")
    a, b = capi.Solver(small), capi.Solver(big)
    with pytest.raises(capi.CuOptError) as e:
        capi.SmallBatch([a] + [b] * 17)
    assert e.value.code == -7
    with pytest.raises(capi.CuOptError) as e:
        capi.SmallBatch([a, a])  # the same solver twice
    assert e.value.code == -1


def test_batch_solve_routes_small_lps_through_the_workgroup_batch(golden_problems, monkeypatch):
    problems = family(golden_problems, 40, seed=9)
    monkeypatch.setenv("CUOPT_AMD_TUNE", "small_batch=0")
    want = capi.batch_solve(problems, tol=1e-6, iteration_limit=LIMIT)
    monkeypatch.setenv("CUOPT_AMD_TUNE", "small_batch=1")
    got = capi.batch_solve(problems, tol=1e-6, iteration_limit=LIMIT)
    for w, g in zip(want, got):
        for k in ("status", "steps_taken", "attempted_steps", "primal_objective", "num_restarts"):
            assert w[k] == g[k], (k, w[k], g[k])
        np.testing.assert_array_equal(w["x"], g["x"])
        np.testing.assert_array_equal(w["y"], g["y"])
