# structured 1e6 x 1e6 LPs (generate_structured) + the BASELINE workloads: one bench line each, then kernel stats of each
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02/structured; mkdir -p $O
for W in "$@"; do
  timeout 900 python bench.py --workload $W --steps 400 --warmup 100 --no-cpu-baseline > $O/bench_$W.json 2> $O/bench_$W.err || tail -5 $O/bench_$W.err
  python - <<PY
import json
d=json.load(open("$O/bench_$W.json"))
r=d["roofline"]; c=d["time_to_1e-4"]
print("$W", "it/s", d["value"], "| layout A/At", d["spmv_layout"]["A"]["layout"], d["spmv_layout"]["At"]["layout"], "| dom", r["kernel"], r["frac"], "|", {k: round(v*1e3,1) for k,v in r["per_kernel_ms"].items()}, "| 1e-4:", c["status"], c["iterations"], "its", c["wall_s"], "s, obj err %.2e" % (abs(c["objective"]-c["objective_known"])/(1+abs(c["objective_known"]))), "| nnz", d["config"]["nnz"])
PY
done
