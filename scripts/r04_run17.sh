O=$GRAFT_REPO_ROOT/gpurun_out/r04_run17; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_persistent_resolve_gpu.py tests/test_kernels_gpu.py tests/test_random_lps_gpu.py tests/test_resident_small_gpu.py -m gpu -q > $O/pytest_b.log 2>&1; tail -4 $O/pytest_b.log
for i in 1 2 3; do timeout -k 5 200 python bench.py --no-cpu-baseline --no-convergence-run | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', d['value'], d['roofline']['per_kernel_ms']['PRIMAL'])"; done
