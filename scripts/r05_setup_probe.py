"""Set-up laps (CUOPT_AMD_TIMING) + layouts + a short rate measurement for the workloads given on the command line.
   python scripts/r05_setup_probe.py c3 banded_shuffled staircase_shuffled block_angular_shuffled [--natural]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CUOPT_AMD_TIMING"] = "1"
import numpy as np  # noqa: E402
from cuopt_amd import capi, synthetic  # noqa: E402


def make(name):
    shuffle = name.endswith("_shuffled")
    base = name[:-9] if shuffle else name
    if base in synthetic.CONFIGS:
        p = synthetic.generate(**synthetic.CONFIGS[base])
    else:
        p = synthetic.generate_structured(base, m=1_000_000, n=1_000_000, k=10, seed=7)
    return synthetic.shuffled(p, seed=5) if shuffle else p


for name in [a for a in sys.argv[1:] if not a.startswith("--")]:
    p = make(name)
    for rep in range(2):  # the second solve shows the set-up with warm pools (hipMalloc, host arrays)
        print("==== %s (solve %d)" % (name, rep), flush=True)
        t0 = time.perf_counter()
        s = capi.Solver(p, mode=1)
        r = s.advance()
        wall = time.perf_counter() - t0
        print("RESULT %s: status %s, %d iterations, wall %.4f s (setup %.4f + loop %.4f), objective %.6g (known %.6g), layout %s, reorder %s" % (
            name, r["status_name"], r["steps_taken"], wall, r["setup_seconds"], r["loop_seconds"], r["primal_objective"], p["objective_star"],
            s.device.layout(), s.reorder_info()), flush=True)
        s.close()
    if os.environ.get("PROBE_NO_RATE"):  # (profsetup: rocprofv3 7.2 falls over the 256-node replay graphs of a fixed-budget run)
        continue
    s = capi.Solver(p, mode=1, tol=0.0)
    s.advance(400)
    s.device.call("synchronize")
    t0 = time.perf_counter()
    s.advance(2000)
    s.device.call("synchronize")
    dt = time.perf_counter() - t0
    ks = {k: s.device.time_kernel(k, 50) for k in ("SPMV_A_DUAL", "SPMV_AT_STEP", "PRIMAL")}
    print("RATE %s: %.1f it/s, kernels ms %s" % (name, 2000 / dt, {k: round(v, 5) for k, v in ks.items()}), flush=True)
    s.close()
