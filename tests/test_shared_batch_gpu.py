"""GPU: K LPs over ONE matrix in lockstep (cuoptamd_solver_clone / cuoptamd_batch_*; kernels_batch.hip) -- BASELINE config 5's
pattern: the MIP heuristics re-solve the same A and c under different bounds (cpp/src/mip/relaxed_lp/relaxed_lp.cu:53-127); the
reference's batch entry point is a thread pool of independent solves (cython_solve.cu:264-296), each of which must give what a
single solve gives.  Pinned here: every LP of a batch takes, BIT FOR BIT, the trajectory of a solver freshly created on that LP
-- iterates, step sizes, restarts, verdicts -- although the two products of an attempt serve all K LPs from one pass over the
matrix (interleaved gather vectors, the panel kernels' reduction trees reproduced)."""
import numpy as np
import pytest

from cuopt_amd import capi, synthetic

pytestmark = pytest.mark.gpu

LIMIT = 4000


def variants(p, k, seed=3):
    """k LPs over p's matrix: the first is p itself, the others tighten some bounds around the known optimum (feasible, same
    optimal value, different trajectories) -- and one of them loosens nothing but fixes a few variables at 0 (may change the
    optimum: verdicts need not agree between the LPs, only with their own single solves)"""
    rng = np.random.default_rng(seed)
    x = p["x_star"]
    out = [(np.array(p["lb"], float), np.array(p["ub"], float))]
    for l in range(1, k):
        lb, ub = np.array(p["lb"], float), np.array(p["ub"], float)
        cols = rng.choice(p["n"], size=p["n"] // (4 + l), replace=False)
        for j in cols:
            if l % 3 == 2:
                ub[j] = lb[j] if np.isfinite(lb[j]) else ub[j]
            elif rng.random() < 0.5:
                ub[j] = x[j] + 0.3 * rng.random()
            else:
                lb[j] = max(lb[j], x[j] - 0.3 * rng.random())
        out.append((lb, ub))
    return out


KEYS_INT = ("status", "steps_taken", "attempted_steps", "num_restarts", "num_major_iterations")
KEYS_F64 = ("primal_objective", "dual_objective", "gap", "l2_primal_residual", "l2_dual_residual", "step_size", "primal_weight",
            "initial_step_size", "initial_primal_weight")


def same(a, b, sa, sb, what):
    for k in KEYS_INT + KEYS_F64:
        assert a[k] == b[k], (what, k, a[k], b[k])
    for u, v, name in zip(sa, sb, "xyz"):
        np.testing.assert_array_equal(u, v, err_msg="%s: %s" % (what, name))


@pytest.mark.parametrize("layout", ["panel", "stream"])
@pytest.mark.parametrize("k", [2, 4, 8, 16])
def test_batch_trajectories_are_bit_identical_to_single_solves(k, layout, monkeypatch):
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", layout)
    p = synthetic.generate(30000, 24000, 10, seed=21)
    bounds = variants(p, k)
    # the single solves: the state after 130 iterations (mid-way between two major iterations) and the end
    single = []
    for lb, ub in bounds:
        s = capi.Solver(dict(p, lb=lb, ub=ub), tol=1e-5, iteration_limit=LIMIT)
        lay = s.device.layout()
        assert lay["A"]["layout"] == layout and lay["At"]["layout"] == layout, lay
        a = s.advance(130)
        sa = s.solution()
        b = s.advance()
        single.append((a, sa, b, s.solution()))
        s.close()
    parent = capi.Solver(dict(p, lb=bounds[0][0], ub=bounds[0][1]), tol=1e-5, iteration_limit=LIMIT)
    solvers = [parent] + [parent.clone(lb=lb, ub=ub) for lb, ub in bounds[1:]]
    batch = capi.SharedMatrixBatch(solvers)
    got = batch.advance(130)
    for l in range(k):
        same(got[l], single[l][0], solvers[l].solution(), single[l][1], "LP %d after 130 iterations" % l)
    got = batch.advance()
    for l in range(k):
        same(got[l], single[l][2], solvers[l].solution(), single[l][3], "LP %d at the end" % l)
    assert got[0]["status_name"] == "Optimal"
    assert len({g["steps_taken"] for g in got}) > 1  # (the LPs finish at different times: some rested while others went on)
    # a second round through the same objects: the clones reset to other bounds, the batch re-created
    batch.close()
    for l in range(1, k):
        lb, ub = bounds[(l + 1) % k if (l + 1) % k else 1]
        solvers[l].reset(lb=lb, ub=ub, tol=1e-5, iteration_limit=LIMIT)
    parent.reset(tol=1e-5, iteration_limit=LIMIT)
    batch = capi.SharedMatrixBatch(solvers)
    got = batch.advance()
    same(got[0], single[0][2], parent.solution(), single[0][3], "parent, second round")
    for l in range(1, k):
        src = (l + 1) % k if (l + 1) % k else 1
        same(got[l], single[src][2], solvers[l].solution(), single[src][3], "LP %d, second round" % l)
    batch.close()
    for s in solvers[1:]:
        s.close()
    parent.close()


def test_clone_alone_is_a_fresh_solver_and_row_bounds_travel(monkeypatch):
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "panel")
    p = synthetic.generate(9000, 7000, 8, seed=4)
    lo = np.where(np.isfinite(p["lo"]), p["lo"] - 0.25, p["lo"])
    hi = np.where(np.isfinite(p["hi"]), p["hi"] + 0.5, p["hi"])
    parent = capi.Solver(p, tol=1e-6, iteration_limit=LIMIT)
    child = parent.clone(lo=lo, hi=hi)
    fresh = capi.Solver(dict(p, lo=lo, hi=hi), tol=1e-6, iteration_limit=LIMIT)
    a, b = child.advance(), fresh.advance()
    same(a, b, child.solution(), fresh.solution(), "clone with new row bounds")
    # the parent is untouched by its clone's solve
    c = parent.advance()
    again = capi.Solver(p, tol=1e-6, iteration_limit=LIMIT)
    d = again.advance()
    same(c, d, parent.solution(), again.solution(), "parent after the clone's solve")
    # the two in a batch of 2 (different row bounds)
    parent.reset(tol=1e-6, iteration_limit=LIMIT)
    child.reset(tol=1e-6, iteration_limit=LIMIT)
    batch = capi.SharedMatrixBatch([parent, child])
    got = batch.advance()
    same(got[0], d, parent.solution(), again.solution(), "batch of 2: parent")
    same(got[1], b, child.solution(), fresh.solution(), "batch of 2: clone")
    batch.close(), child.close(), parent.close()


def test_not_eligible_layouts_are_refused(monkeypatch):
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "jag")
    p = synthetic.generate(20000, 20000, 10, seed=2, band=500)
    parent = capi.Solver(p, tol=1e-4, iteration_limit=200)
    lay = parent.device.layout()
    assert lay["A"]["layout"] == "jag" or lay["At"]["layout"] == "jag", lay  # (a band: jagged rows with LDS column sets)
    child = parent.clone()
    with pytest.raises(capi.CuOptError) as e:
        capi.SharedMatrixBatch([parent, child])
    assert e.value.code == -7
    # ... and the solvers are still good one by one
    a, b = parent.advance(), child.advance()
    same(a, b, parent.solution(), child.solution(), "clone of a solver outside the panels")
    child.close(), parent.close()
    with pytest.raises(capi.CuOptError):
        capi.SharedMatrixBatch([capi.Solver(synthetic.generate(600, 500, 6, seed=1))] * 3)  # K = 3


def test_batch_solve_routes_lps_over_one_matrix_through_the_lockstep_batch(monkeypatch):
    """cuoptamd_batch_solve (the C entry under BatchSolve): LPs that share matrix and objective go through one set-up and the
    lockstep batch -- 16 + 8 + 4 + 1 here -- and every answer is the one the independent solves give"""
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "panel")
    p = synthetic.generate(12000, 10000, 8, seed=9)
    lps = [dict(p, lb=lb, ub=ub) for lb, ub in variants(p, 29, seed=5)]
    together = capi.batch_solve(lps, tol=1e-5, iteration_limit=LIMIT)
    monkeypatch.setenv("CUOPT_AMD_TUNE", "shared_batch=0")
    apart = capi.batch_solve(lps, tol=1e-5, iteration_limit=LIMIT, max_threads=2)
    assert len(together) == len(apart) == 29
    for l, (a, b) in enumerate(zip(together, apart)):
        for k in KEYS_INT + KEYS_F64:
            assert a[k] == b[k], (l, k, a[k], b[k])
        for name in ("x", "y", "reduced_cost"):
            np.testing.assert_array_equal(a[name], b[name], err_msg="LP %d: %s" % (l, name))
    assert together[0]["status_name"] == "Optimal"
    # LPs over different matrices keep the independent path
    q = synthetic.generate(12000, 10000, 8, seed=10)
    mixed = capi.batch_solve([lps[0], q, lps[1], lps[2]], tol=1e-5, iteration_limit=LIMIT)
    assert mixed[0]["steps_taken"] == apart[0]["steps_taken"] and mixed[1]["status_name"] == "Optimal"


@pytest.mark.parametrize("mode", [0, 2, 3])  # Stable1, Methodical1 (trust-region restarts), Fast1 (artificial restarts, a step per trip)
def test_batch_under_the_other_presets(mode, monkeypatch):
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "panel")
    p = synthetic.generate(6000, 5000, 8, seed=33)
    bounds = variants(p, 4, seed=7)
    limit = 1500
    single = []
    for lb, ub in bounds:
        s = capi.Solver(dict(p, lb=lb, ub=ub), mode=mode, tol=1e-4, iteration_limit=limit)
        single.append((s.advance(), s.solution()))
        s.close()
    parent = capi.Solver(dict(p, lb=bounds[0][0], ub=bounds[0][1]), mode=mode, tol=1e-4, iteration_limit=limit)
    solvers = [parent] + [parent.clone(lb=lb, ub=ub) for lb, ub in bounds[1:]]
    batch = capi.SharedMatrixBatch(solvers)
    got = batch.advance()
    for l in range(4):
        same(got[l], single[l][0], solvers[l].solution(), single[l][1], "mode %d, LP %d" % (mode, l))
    batch.close()
    for s in solvers[1:]:
        s.close()
    parent.close()


def test_an_infeasible_member_gets_its_own_verdict(monkeypatch):
    """one LP of the batch is infeasible (a variable's bounds exclude every feasible point of its rows): with infeasibility
    detection on it ends PrimalInfeasible exactly where its single solve does, the others are Optimal"""
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "panel")
    p = synthetic.generate(3000, 2500, 6, seed=12)
    lb_bad, ub_bad = np.array(p["lb"], float), np.array(p["ub"], float)
    big = np.argsort(-p["x_star"])[:50]
    ub_bad[big] = 0.0  # the rows these variables carry cannot be met any more (with high probability: checked by the single solve)
    sets = [(np.array(p["lb"], float), np.array(p["ub"], float)), (lb_bad, ub_bad)] + variants(p, 3, seed=2)[1:]
    kw = dict(tol=1e-4, iteration_limit=LIMIT, detect_infeasibility=1)
    single = []
    for lb, ub in sets:
        s = capi.Solver(dict(p, lb=lb, ub=ub), **kw)
        single.append((s.advance(), s.solution()))
        s.close()
    parent = capi.Solver(dict(p, lb=sets[0][0], ub=sets[0][1]), **kw)
    solvers = [parent] + [parent.clone(lb=lb, ub=ub) for lb, ub in sets[1:]]
    batch = capi.SharedMatrixBatch(solvers)
    got = batch.advance()
    for l in range(4):
        same(got[l], single[l][0], solvers[l].solution(), single[l][1], "LP %d" % l)
    assert got[0]["status_name"] == "Optimal"
    batch.close()
    for s in solvers[1:]:
        s.close()
    parent.close()


def test_row_bounds_are_shared_until_someone_changes_them(monkeypatch):
    """clones read the parent's row bounds (one copy for all LPs of a batch); a reset with other row bounds -- of a clone or of the
    parent -- gives that solver arrays of its own and leaves the others alone"""
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "panel")
    p = synthetic.generate(5000, 4000, 8, seed=17)
    lo2 = np.where(np.isfinite(p["lo"]), p["lo"] - 0.5, p["lo"])
    hi2 = np.where(np.isfinite(p["hi"]), p["hi"] + 0.5, p["hi"])
    kw = dict(tol=1e-6, iteration_limit=LIMIT)
    ref = capi.Solver(p, **kw)
    a = ref.advance()
    sa = ref.solution()
    ref2 = capi.Solver(dict(p, lo=lo2, hi=hi2), **kw)
    b = ref2.advance()
    sb = ref2.solution()
    parent = capi.Solver(p, **kw)
    child = parent.clone()
    parent.reset(lo=lo2, hi=hi2, **kw)  # the parent moves away; the clone keeps the bounds it was created with
    same(child.advance(), a, child.solution(), sa, "clone after the parent's row bounds changed")
    same(parent.advance(), b, parent.solution(), sb, "parent with new row bounds")
    child.reset(lo=lo2, hi=hi2, **kw)
    parent.reset(lo=p["lo"], hi=p["hi"], **kw)
    same(child.advance(), b, child.solution(), sb, "clone with new row bounds")
    same(parent.advance(), a, parent.solution(), sa, "parent back on the first row bounds")
    child.close(), parent.close()


def test_a_parent_reset_again_and_again_next_to_a_clone_does_not_grow(monkeypatch):
    """(round-5 advisor) the copy-on-change of shared row bounds happens ONCE per clone generation, not at every reset of the parent:
    a MIP loop that keeps resetting the parent next to live clones must not collect 32 m bytes per reset -- and the clone made AFTER
    such a reset aliases the parent's new arrays, so the parent's next reset copies again (and both keep their own bounds)"""
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "panel")
    p = synthetic.generate(5000, 4000, 8, seed=19)
    kw = dict(tol=1e-6, iteration_limit=LIMIT)
    bounds = [(np.where(np.isfinite(p["lo"]), p["lo"] - 0.1 * k, p["lo"]), np.where(np.isfinite(p["hi"]), p["hi"] + 0.1 * k, p["hi"])) for k in range(1, 7)]
    parent = capi.Solver(p, **kw)
    first = parent.clone()
    used = []
    for lo, hi in bounds[:4]:
        parent.reset(lo=lo, hi=hi, **kw)
        used.append(capi.lib.pdlpdev_device_bytes(parent.device.handle))
    assert used[0] == used[1] == used[2] == used[3], used  # one copy (at the first reset), then in place
    second = parent.clone()  # aliases the arrays the parent holds NOW (bounds[3])
    parent.reset(lo=bounds[4][0], hi=bounds[4][1], **kw)
    grown = capi.lib.pdlpdev_device_bytes(parent.device.handle)
    assert grown > used[3]  # the parent moved off the arrays its second clone reads ...
    parent.reset(lo=bounds[5][0], hi=bounds[5][1], **kw)
    assert capi.lib.pdlpdev_device_bytes(parent.device.handle) == grown  # ... once
    ref = {}
    for name, q in (("first", p), ("second", dict(p, lo=bounds[3][0], hi=bounds[3][1])), ("parent", dict(p, lo=bounds[5][0], hi=bounds[5][1]))):
        s = capi.Solver(q, **kw)
        ref[name] = (s.advance(), s.solution())
        s.close()
    same(first.advance(), ref["first"][0], first.solution(), ref["first"][1], "the first clone still reads the original row bounds")
    same(second.advance(), ref["second"][0], second.solution(), ref["second"][1], "the second clone reads the bounds of its creation")
    same(parent.advance(), ref["parent"][0], parent.solution(), ref["parent"][1], "the parent on its latest bounds")
    first.close(), second.close(), parent.close()

def test_members_may_be_reset_while_the_batch_lives(monkeypatch):
    """a reset of a member between two advances of ONE batch object -- new variable bounds (their uniform-bound summary travels with
    the LP's table entry) and new row bounds (the member's lo / hi move to arrays of its own) -- is seen by the next advance"""
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "panel")
    p = synthetic.generate(6000, 5000, 8, seed=29)
    bounds = variants(p, 4, seed=5)
    lo2 = np.where(np.isfinite(p["lo"]), p["lo"] - 0.5, p["lo"])
    hi2 = np.where(np.isfinite(p["hi"]), p["hi"] + 0.5, p["hi"])
    kw = dict(tol=1e-6, iteration_limit=LIMIT)
    def single(lb, ub, lo, hi):
        s = capi.Solver(dict(p, lb=lb, ub=ub, lo=lo, hi=hi), **kw)
        r = s.advance()
        out = (r, s.solution())
        s.close()
        return out
    parent = capi.Solver(dict(p, lb=bounds[0][0], ub=bounds[0][1]), **kw)
    solvers = [parent] + [parent.clone(lb=lb, ub=ub) for lb, ub in bounds[1:]]
    batch = capi.SharedMatrixBatch(solvers)
    got = batch.advance()
    for l in range(4):
        r, sol = single(bounds[l][0], bounds[l][1], p["lo"], p["hi"])
        same(got[l], r, solvers[l].solution(), sol, "LP %d, first round" % l)
    # second round through the SAME batch: LP 1 takes LP 3's variable bounds, LP 2 other row bounds, LP 3 both; the parent starts over
    solvers[1].reset(lb=bounds[3][0], ub=bounds[3][1], **kw)
    solvers[2].reset(lb=bounds[2][0], ub=bounds[2][1], lo=lo2, hi=hi2, **kw)
    solvers[3].reset(lb=bounds[1][0], ub=bounds[1][1], lo=lo2, hi=hi2, **kw)
    parent.reset(**kw)
    got = batch.advance()
    want = [(bounds[0], (p["lo"], p["hi"])), (bounds[3], (p["lo"], p["hi"])), (bounds[2], (lo2, hi2)), (bounds[1], (lo2, hi2))]
    for l, ((lb, ub), (lo, hi)) in enumerate(want):
        r, sol = single(lb, ub, lo, hi)
        same(got[l], r, solvers[l].solution(), sol, "LP %d, second round of the same batch" % l)
    batch.close()
    for s in solvers[1:]:
        s.close()
    parent.close()


def test_batches_repeat_themselves_next_to_other_processes():
    """four processes share the GPU, each solving the same LPs singly and as lockstep batches: whatever the interleaving of their
    kernels, a batch gives the single solves' results (scripts/r05_contention_probe.py).  Pins the barrier between the first chunk's
    reads and the third chunk's staging in batch_block_sums: waves of a workgroup drift far enough apart to need it only when the
    scheduler takes them off the compute unit -- i.e. next to other processes' kernels"""
    import os
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts", "r05_contention_probe.py")
    procs = [subprocess.Popen([sys.executable, probe, "4", "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for _ in range(4)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out[-2000:]
        rounds = [ln for ln in out.splitlines() if " round " in ln]
        assert len(rounds) == 2, out[-2000:]
        for ln in rounds:
            assert "False" not in ln, ln


def test_timing_the_kernels_leaves_the_batch_where_it_was(monkeypatch):
    """pdlpdev_batch_time_kernels (bench.py's roofline leg) forces every LP active for its measurement and puts control blocks and
    running sums back: the solves go on as if nothing had happened"""
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "panel")
    p = synthetic.generate(6000, 5000, 8, seed=31)
    bounds = variants(p, 4, seed=7)
    kw = dict(tol=1e-6, iteration_limit=LIMIT)
    parent = capi.Solver(dict(p, lb=bounds[0][0], ub=bounds[0][1]), **kw)
    solvers = [parent] + [parent.clone(lb=lb, ub=ub) for lb, ub in bounds[1:]]
    batch = capi.SharedMatrixBatch(solvers)
    batch.advance(130)
    t = batch.time_kernels(3)
    assert all(v > 0.0 for v in t.values()), t
    got = batch.advance()
    for l, (lb, ub) in enumerate(bounds):
        s = capi.Solver(dict(p, lb=lb, ub=ub), **kw)
        r = s.advance()
        same(got[l], r, solvers[l].solution(), s.solution(), "LP %d after the timed attempts" % l)
        s.close()
    batch.close()
    for s in solvers[1:]:
        s.close()
    parent.close()
