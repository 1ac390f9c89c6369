# round 4: the split device layer through the whole GPU suite + all bench workloads (one line each) + sanitizer runs on the GPU box
O=$GRAFT_REPO_ROOT/gpurun_out/r04_run9; mkdir -p $O
timeout -k 5 1200 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; tail -6 $O/pytest_all.log
for W in c3 c2 hard banded staircase block_angular powerlaw multiband dense_rows c3x10; do
  timeout -k 5 400 python bench.py --workload $W --no-cpu-baseline > $O/line_$W.json 2> $O/line_$W.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/line_$W.json").read().strip().splitlines()[-1])
    r=d["roofline"]; c=d.get("time_to_1e-4") or {}
    print("$W", d["value"], "it/s", r["kernel"], r["avg_launch_ms"], "frac", r["frac"], "ksum/attempt", r.get("attempt_kernels_over_ms_per_attempt"), "layout", d["spmv_layout"]["A"]["layout"], d["spmv_layout"]["A"].get("row_sums",""), "wall", c.get("wall_s"), c.get("iterations"), c.get("status"))
except Exception as e:
    print("$W FAILED", e); print(open("$O/line_$W.err").read()[-800:])
PY
done 2>&1 | tee $O/r04_lines.txt
cat $O/line_*.json > $O/r04_bench_lines.jsonl
