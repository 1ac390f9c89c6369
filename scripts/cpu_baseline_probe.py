import os, time, sys
sys.path.insert(0, os.getcwd())
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip().replace("\n", " | "))
    except Exception as e:
        print(f, "n/a")
os.system("lscpu | egrep 'Socket|NUMA|Model name|Thread|Core' | head -12")
from cuopt_amd import synthetic
from oracle import orcbind
p = synthetic.generate(**synthetic.CONFIGS["c3"])
def stat():
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().strip().split("\n"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except Exception:
        return (0, 0)
for pol in ("active", "passive"):
    os.environ["OMP_WAIT_POLICY"] = pol
    for t in (4, 8, 16, 32, 64, 128):
        a = stat()
        o = orcbind.solve(p, tol=0.0, iteration_limit=12, num_threads=t)
        b = stat()
        print(pol, "threads", t, "its/s %.2f" % (o["steps_taken"] / o["loop_seconds"]), "throttled periods +%d, throttled ms +%.0f" % (b[0] - a[0], (b[1] - a[1]) / 1e3), flush=True)
