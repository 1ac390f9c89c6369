# round 4: the long-tail panels with three chunks in flight, DPP scan, LDS-atomic emission -- parity tests, then bench lines
O=gpurun_out/r04_seg2; mkdir -p $O
timeout 900 python -m pytest tests/test_panel_seg_gpu.py -m gpu -q > $O/pytest_seg.log 2>&1; tail -25 $O/pytest_seg.log
run() { name=$1; wl=$2; shift 2; env "$@" timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-convergence-run > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d["roofline"]["per_kernel_ms"], d.get("spmv_layout")["A"].get("row_sums"))
except Exception as e:
    print("$name FAILED", e); print(open("$O/$name.err").read()[-1500:])
PY
}
run powerlaw_seg0 powerlaw CUOPT_AMD_TUNE=panel_seg=0
run powerlaw_auto powerlaw X=1
run c3_seg0 c3 X=1
run c3_seg1 c3 CUOPT_AMD_TUNE=panel_seg=1
run dense_rows_seg1 dense_rows CUOPT_AMD_TUNE=panel_seg=1
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; tail -12 $O/pytest_all.log
