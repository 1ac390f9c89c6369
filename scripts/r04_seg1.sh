# round 4: first run of the long-tail panels (row sums by nonzero) -- parity tests, then power-law / c3 / dense_rows lines, seg off / on
O=gpurun_out/r04_seg1; mkdir -p $O
timeout 900 python -m pytest tests/test_panel_seg_gpu.py -m gpu -q -x > $O/pytest_seg.log 2>&1; tail -15 $O/pytest_seg.log
run() { name=$1; wl=$2; shift 2; env "$@" timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-convergence-run > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d["roofline"]["per_kernel_ms"], d.get("spmv_layout"))
except Exception as e:
    print("$name FAILED", e); print(open("$O/$name.err").read()[-1500:])
PY
}
run powerlaw_seg0 powerlaw CUOPT_AMD_TUNE=panel_seg=0
run powerlaw_auto powerlaw X=1
run c3_seg0 c3 X=1
run c3_seg1 c3 CUOPT_AMD_TUNE=panel_seg=1
run dense_rows_auto dense_rows X=1
run dense_rows_seg1 dense_rows CUOPT_AMD_TUNE=panel_seg=1
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; tail -8 $O/pytest_all.log
