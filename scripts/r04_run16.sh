# round 4: k_primal without the bound arrays where every bound is the same 0 / infinity; dense-segment duplicate test; c3 line
O=$GRAFT_REPO_ROOT/gpurun_out/r04_run16; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_kernels_gpu.py tests/test_solve_gpu.py tests/test_dense_segments_gpu.py tests/test_persistent_resolve_gpu.py tests/test_panel_seg_gpu.py -m gpu -q > $O/pytest_b.log 2>&1; tail -5 $O/pytest_b.log
timeout -k 5 300 python scripts/r04_x1.py '[["auto", "c3", {}], ["auto", "c2", {}]]' | sed 's/"layout": {.*"resident": false}, //' | cut -c1-300
timeout -k 5 300 python bench.py --no-cpu-baseline > $O/c3.json 2> $O/c3.err; python -c "
import json; d=json.loads(open('$O/c3.json').read().strip().splitlines()[-1]); print('c3', d['value'], d['roofline']['per_kernel_ms'], d['roofline']['attempt_kernels_over_ms_per_attempt'], d['time_to_1e-4']['wall_s'])"
