# round 3 (late): coalesced scaling kernels -- parity tests (scaling is compared with the oracle bit for bit), set-up laps, c3 / banded lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/check5
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_solve_gpu.py tests/test_structured_gpu.py tests/test_panel_layout_gpu.py tests/test_dense_segments_gpu.py tests/test_long_row_extraction_gpu.py tests/test_sharded_gpu.py tests/test_random_lps_gpu.py tests/test_full_size_gpu.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -3
for W in c3 banded block_angular; do
CUOPT_AMD_TIMING=1 timeout 600 python bench.py --workload $W --no-cpu-baseline --steps 20 --warmup 5 > $O/${W}_timing.json 2> $O/${W}_timing.err
grep "scaling_compute\|scale_problem" $O/${W}_timing.err | tail -2
python -c "
import json; d = json.load(open('$O/${W}_timing.json')); c = d['time_to_1e-4']
print('$W', d['value'], 'it/s', c['status'], c['iterations'], 'wall', c['wall_s'], 'setup', c['setup_s'], 'loop', c['loop_s'])"
done
