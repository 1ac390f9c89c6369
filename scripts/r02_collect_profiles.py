"""gpurun_out/r02/final/* (scripts/r02_final_profiles.sh) -> profiles/r02_*: kernel statistics as a fixed-width table, PMC json +
summary.  Bench lines are appended by the caller AFTER the PMC files are in place (bench.py reads `traffic` from them).
usage: python scripts/r02_collect_profiles.py c3 banded"""
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r02", "final")
for w in sys.argv[1:]:
    files = sorted(glob.glob(os.path.join(SRC, "prof_" + w, "*", "*kernel_stats.csv")), key=os.path.getmtime)
    rows = list(csv.DictReader(open(files[-1])))
    out = ["# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --workload %s --steps 800 --warmup 100 --no-cpu-baseline --no-convergence-run" % w,
           "# (round 2, one MI355X; scripts/r02_final_profiles.sh; kernel-trace durations = dispatch start/stop timestamps inside the solver's loop)",
           "%-64s %8s %16s %14s %8s %10s %10s" % ("kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns")]
    for r in rows:
        name = r["Name"].split("(")[0]
        out.append("%-64s %8d %16d %14.1f %8.2f %10d %10d" % (name[:64], int(r["Calls"]), int(r["TotalDurationNs"]), float(r["AverageNs"]),
                                                             float(r["Percentage"]), int(r["MinNs"]), int(r["MaxNs"])))
    open(os.path.join(ROOT, "profiles", "r02_bench_%s_kernel_stats.txt" % w), "w").write("\n".join(out) + "\n")
    shutil.copy(os.path.join(SRC, "pmc_%s.json" % w), os.path.join(ROOT, "profiles", "r02_pmc_%s.json" % w))
    shutil.copy(os.path.join(SRC, "pmc_%s_summary.txt" % w), os.path.join(ROOT, "profiles", "r02_pmc_%s_summary.txt" % w))
    print(w, "<-", os.path.relpath(files[-1], ROOT))
    print("\n".join(out[2:8]))
