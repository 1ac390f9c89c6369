"""Calibration of the structural layout rule (pdlp_device.hip gather_working_set): for a sweep of matrices, the live gather set
of the CSR stream kernel next to the device timings of both layouts (CUOPT_AMD_SPMV_LAYOUT=timed, CUOPT_AMD_TIMING=1).
Run on the GPU box:  python scripts/r02_layout_rule.py 2> gpurun_out/r02/layout_rule.txt"""
import os
import sys

os.environ["CUOPT_AMD_SPMV_LAYOUT"] = "timed"
os.environ["CUOPT_AMD_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuopt_amd import capi, synthetic  # noqa: E402

cases = [("random n=2e5", dict(m=200_000, n=200_000, k=10, seed=1)),
         ("random n=3e5", dict(m=300_000, n=300_000, k=10, seed=1)),
         ("random n=4e5", dict(m=400_000, n=400_000, k=10, seed=1)),
         ("random n=5e5", dict(m=500_000, n=500_000, k=10, seed=1)),
         ("random n=7e5", dict(m=700_000, n=700_000, k=10, seed=1)),
         ("random n=1e6", dict(m=1_000_000, n=1_000_000, k=10, seed=2)),
         ("random 2e6x5e5", dict(m=2_000_000, n=500_000, k=5, seed=1)),
         ("band 6000 n=1e6", dict(m=1_000_000, n=1_000_000, k=10, seed=2, band=6000)),
         ("band 20000 n=1e6", dict(m=1_000_000, n=1_000_000, k=10, seed=2, band=20000)),
         ("band 60000 n=1e6", dict(m=1_000_000, n=1_000_000, k=10, seed=2, band=60000)),
         ("band 150000 n=1e6", dict(m=1_000_000, n=1_000_000, k=10, seed=2, band=150000)),
         ("band 300000 n=1e6", dict(m=1_000_000, n=1_000_000, k=10, seed=2, band=300000))]
for name, cfg in cases:
    p = synthetic.generate(**cfg)
    print(f"==== {name}", file=sys.stderr, flush=True)
    dev = capi.Device(p)
    print(f"     chosen: {dev.layout()}", file=sys.stderr, flush=True)
    dev.close()
for kind in ("staircase", "block_angular", "powerlaw"):
    p = synthetic.generate_structured(kind)
    print(f"==== {kind}", file=sys.stderr, flush=True)
    dev = capi.Device(p)
    print(f"     chosen: {dev.layout()}", file=sys.stderr, flush=True)
    dev.close()
