# round 3 (late): the method / simplex tests on the GPU after the sparse-LU simplex, the set-up laps again, the c3 line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/check2
mkdir -p $O
timeout 600 python -m pytest tests/test_method_and_multigpu_gpu.py tests/test_python_api.py tests/test_solve_gpu.py tests/test_dense_segments_gpu.py -m gpu -q 2>&1 | tail -4
CUOPT_AMD_TIMING=1 timeout 600 python bench.py --workload c3 --no-cpu-baseline --steps 20 --warmup 5 > $O/c3_timing.json 2> $O/c3_timing.err
grep "cuopt_amd setup" $O/c3_timing.err | tail -23 | head -12
timeout 600 python bench.py --workload c3 --no-cpu-baseline > $O/c3.json 2> $O/c3.err
python -c "
import json; d = json.load(open('$O/c3.json')); c = d['time_to_1e-4']
print('c3', d['value'], 'it/s', c['status'], c['iterations'], 'wall', c['wall_s'], 'setup', c['setup_s'], 'loop', c['loop_s'])"
