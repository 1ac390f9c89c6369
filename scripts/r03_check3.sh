# round 3 (late): the A^T side of the set-up on a thread of its own -- layout / parity tests, the set-up laps, the c3 line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/check3
mkdir -p $O
timeout 900 python -m pytest tests/test_panel_layout_gpu.py tests/test_gather_free_layout_gpu.py tests/test_dense_segments_gpu.py tests/test_long_row_extraction_gpu.py tests/test_full_size_gpu.py tests/test_structured_gpu.py tests/test_sharded_gpu.py tests/test_kernels_gpu.py tests/test_solve_gpu.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -3
for W in c3 banded; do
CUOPT_AMD_TIMING=1 timeout 600 python bench.py --workload $W --no-cpu-baseline --steps 20 --warmup 5 > $O/${W}_timing.json 2> $O/${W}_timing.err
grep "cuopt_amd setup" $O/${W}_timing.err | tail -22 | head -17
timeout 600 python bench.py --workload $W --no-cpu-baseline > $O/$W.json 2> $O/$W.err
python -c "
import json; d = json.load(open('$O/$W.json')); c = d['time_to_1e-4']
print('$W', d['value'], 'it/s', c['status'], c['iterations'], 'wall', c['wall_s'], 'setup', c['setup_s'], 'loop', c['loop_s'])"
done
