"""One layout's bit-exact checks against the oracle, over and over, while FIVE other processes do the same for the other layouts on the
same GPU (scripts/gpu_session.sh <tag> soak:<rounds>): the double-buffered LDS stages of the kernels (DESIGN: LDS hazard audit) must
not depend on how the workgroups of a launch happen to be scheduled.  Round-5 review, item 4: two races had shown up only under
exactly this kind of load.  Prints one line per round; "MISMATCH" anywhere fails the soak.

  python scripts/contention_layouts.py panel|panel_seg|jag|pb|pb_wide|stream|resident [rounds]"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
layout = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tune = {"panel": "slab_bytes=16384,panel_seg=0,panel_nnz=6000", "panel_seg": "slab_bytes=16384,panel_seg=1,panel_nnz=6000", "jag": "jag_waves=8", "pb": "", "pb_wide": "pb_wide=1", "stream": "",
        "resident": ""}[layout]
os.environ["CUOPT_AMD_SPMV_LAYOUT"] = {"panel_seg": "panel", "resident": "auto", "pb_wide": "pb"}.get(layout, layout)
os.environ["CUOPT_AMD_SMALL"] = "1" if layout == "resident" else "0"
if tune:
    os.environ["CUOPT_AMD_TUNE"] = tune
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("OMP_NUM_THREADS", "2")
from cuopt_amd import capi, synthetic  # noqa: E402
from oracle import orcbind  # noqa: E402

if layout == "resident":
    p = synthetic.generate(1000, 1000, 8, seed=4)
elif layout == "jag":
    p = synthetic.generate(60000, 60000, 10, seed=2, band=300)
elif layout == "pb_wide":  # (three entries per row over 25 bins x 4 panels: the wide bins hold it; steps with levels 0 ... 6 on the A^T side)
    p = synthetic.generate(200000, 30000, 3, seed=21)
else:
    p = synthetic.generate(60000, 50000, 10, seed=21)
to, ti, tv = orcbind.transpose(p["m"], p["n"], p["offsets"], p["indices"], p["values"])
rng = np.random.default_rng(5)
vecs = [(rng.standard_normal(p["n"]), rng.standard_normal(p["m"])) for _ in range(3)]
refs = [(orcbind.spmv(p["offsets"], p["indices"], p["values"], x), orcbind.spmv(to, ti, tv, y)) for x, y in vecs]
oracle = {its: orcbind.solve(p, tol=0.0, iteration_limit=its) for its in (5, 40)}
exact = layout != "panel_seg"  # (the long-tail panels: a fixed tree, rtol 1e-12 per row, the same bits on every launch)
first = None
bad = 0
for r in range(rounds):
    dev = capi.Device(p)
    lay = dev.layout()
    want = {"panel_seg": "panel", "pb_wide": "pb"}.get(layout, layout)
    assert lay["A"]["layout"] == want or (layout == "resident" and lay["resident"]), lay
    assert layout != "pb_wide" or lay["A"]["workgroups"] == -(-p["m"] // 8192), lay
    ok_spmv, bits = True, []
    for (x, y), (ra, rt) in zip(vecs, refs):
        for vec, tr, rows, ref in ((x, False, p["m"], ra), (y, True, p["n"], rt)):
            for _ in range(4):
                got = dev.spmv(vec, tr, rows)
                bits.append(hash(got.tobytes()))
                if exact:
                    ok_spmv &= bool(np.array_equal(got, ref))
                else:
                    ok_spmv &= bool(np.allclose(got, ref, rtol=1e-12, atol=1e-12 * (1.0 + np.abs(ref).max())))
    dev.close()
    if first is None:
        first = bits
    ok_repeat = bits == first and all(bits[i] == bits[i - i % 4] for i in range(len(bits)))
    ok_traj = True
    for its, o in oracle.items():
        s = capi.Solver(p, tol=0.0, iteration_limit=its)
        q = s.advance()
        s.close()
        ok_traj &= (q["steps_taken"], q["attempted_steps"]) == (int(o["steps_taken"]), int(o["attempted_steps"]))
        ok_traj &= abs(q["step_size"] - o["final_step_size"]) <= 1e-9 * abs(o["final_step_size"])
        ok_traj &= abs(q["primal_objective"] - o["primal_objective"]) <= 1e-9 * (1 + abs(o["primal_objective"]))
    good = ok_spmv and ok_repeat and ok_traj
    bad += not good
    print("pid %d layout %-9s round %d: products %s the oracle [%s], launches repeat themselves [%s], first 5 / 40 iterations follow the oracle [%s] %s" % (
        os.getpid(), layout, r, "equal" if exact else "within rtol 1e-12 of", ok_spmv, ok_repeat, ok_traj, "" if good else "MISMATCH"), flush=True)
sys.exit(1 if bad else 0)
