# round 3: the dense_rows workload (1e6 x 1e6; 40 dense constraints of 50 000 consecutive variables = 2e6 of 1.1e7 nonzeros) with the
# index-free path on (default) and off
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/dense
mkdir -p $O
for D in 0 auto 0 auto; do
  if [ $D = auto ]; then unset CUOPT_AMD_DENSE; else export CUOPT_AMD_DENSE=$D; fi
  CUOPT_AMD_TIMING=1 timeout 900 python bench.py --workload dense_rows --no-cpu-baseline > $O/dense_$D.json 2> $O/dense_$D.err
  python - <<PY
import json
d = json.load(open("$O/dense_$D.json"))
print("dense=$D", d["value"], "it/s", d["spmv_layout"]["A"]["layout"], d["spmv_layout"]["At"]["layout"], {k: round(v * 1e3, 1) for k, v in d["roofline"]["per_kernel_ms"].items()}, d["time_to_1e-4"]["status"], d["time_to_1e-4"]["iterations"], d["time_to_1e-4"]["wall_s"])
PY
  grep "dense:" $O/dense_$D.err | head -1
done
