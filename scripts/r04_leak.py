"""Is the Concurrent path leaking?  Current RSS (not the high-water mark) over many solves of the 2e4 x 2e4 LP."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cuopt_amd import capi, synthetic
def rss():
    return int(open("/proc/self/statm").read().split()[1]) * 4096 / float(1 << 20)
mid = synthetic.generate(20000, 20000, 10, seed=3)
for method in (1, 0, 0):
    out = []
    for i in range(300):
        r = capi.solve(mid, method=method, tol=1e-4, iteration_limit=100000)
        if i % 50 == 0: out.append(round(rss(), 1))
    out.append(round(rss(), 1))
    print("method", method, "RSS MB every 50 solves:", out, flush=True)
