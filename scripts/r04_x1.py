"""round 4 experiment driver: time the loop kernels of one workload through a given library build / tune string (child process per
variant: CUOPT_AMD_LIB and CUOPT_AMD_TUNE are read at import / create time)."""
import json
import os
import subprocess
import sys

CHILD = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from cuopt_amd import capi, synthetic
wl = %(wl)r
structured = wl in ("staircase", "block_angular", "powerlaw", "multiband", "dense_rows")
if wl.startswith("rand:"):
    _, mm, kk = wl.split(":")
    cfg = dict(m=int(mm), n=int(mm), k=int(kk), seed=5)
else:
    cfg = dict(kind=wl, m=1_000_000, n=1_000_000, k=10, seed=7) if structured else dict(synthetic.CONFIGS[wl])
p = synthetic.generate_structured(**cfg) if structured else synthetic.generate(**cfg)
dev = capi.Device(p)
out = dict(layout=dev.layout())
for k in ("SPMV_A_DUAL", "SPMV_AT_STEP", "SPMV_A_PLAIN", "SPMV_AT_PLAIN", "PRIMAL", "STEP_DECISION"):
    out[k] = round(1e3 * dev.time_kernel(k, 50), 2)
print(json.dumps(out))
'''


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    variants = json.loads(sys.argv[1])  # [[name, workload, {env}], ...]
    for name, wl, env in variants:
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, "-c", CHILD % dict(root=root, wl=wl)], capture_output=True, text=True, env=e, cwd=root, timeout=600)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-800:]
        print(name, wl, line, flush=True)


if __name__ == "__main__":
    main()
