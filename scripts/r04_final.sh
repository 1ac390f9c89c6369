# round 4, final tree: the whole GPU suite, smoke, every bench workload (one line each), rocprof kernel stats of c3 / dense_rows
O=$GRAFT_REPO_ROOT/gpurun_out/r04_final2; mkdir -p $O
export TMPDIR=/tmp
timeout -k 5 1200 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; tail -4 $O/pytest_all.log; grep -E "^FAILED" $O/pytest_all.log | head
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
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
for W in c3 dense_rows; do
  (cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$W -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --workload $W --no-cpu-baseline --no-convergence-run --steps 400 --warmup 80" > $O/trace_$W.log 2>&1)
  F=$(ls $O/trace_$W/*/*kernel_stats.csv 2>/dev/null | tail -1); [ -n "$F" ] && cp $F $O/r04_bench_${W}_kernel_stats.csv && head -6 $F | cut -c1-50,180-330
  rm -rf $O/trace_$W
done
