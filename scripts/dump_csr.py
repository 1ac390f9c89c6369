"""Dumps a bench workload's A and A^T as raw arrays for tools/rocsparse_yardstick.cpp:  python scripts/dump_csr.py c3 /tmp/c3csr"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from cuopt_amd import capi, synthetic  # noqa: E402

name, out = sys.argv[1], sys.argv[2]
p = synthetic.generate(**synthetic.CONFIGS[name]) if name in synthetic.CONFIGS else synthetic.generate_structured(name, m=1_000_000, n=1_000_000, k=10, seed=7)
os.makedirs(out, exist_ok=True)
to, ti, tv = capi.csr_transpose(p["m"], p["n"], p["offsets"], p["indices"], p["values"])
for tag, arr in (("a_off.i32", p["offsets"]), ("a_idx.i32", p["indices"]), ("a_val.f64", p["values"]), ("at_off.i32", to), ("at_idx.i32", ti), ("at_val.f64", tv)):
    np.ascontiguousarray(arr).tofile(os.path.join(out, tag))
open(os.path.join(out, "dims.txt"), "w").write("%d %d\n" % (p["m"], p["n"]))
print("dumped", name, p["m"], p["n"], len(p["values"]))
