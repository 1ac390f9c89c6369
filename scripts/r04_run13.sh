# round 4: dense row segments folded into the panel kernels (own-row workgroups on the A side, the column epilogue on the A^T side)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_run13; mkdir -p $O
timeout -k 5 400 python -m pytest tests/test_dense_segments_gpu.py tests/test_panel_seg_gpu.py tests/test_panel_layout_gpu.py -m gpu -q > $O/pytest_dense.log 2>&1; tail -8 $O/pytest_dense.log
timeout -k 5 200 python bench.py --workload dense_rows --no-cpu-baseline > $O/dense_rows.json 2> $O/dense_rows.err; python -c "
import json; d=json.loads(open('$O/dense_rows.json').read().strip().splitlines()[-1]); print('dense_rows', d['value'], d['roofline']['per_kernel_ms'], d['spmv_layout']['A'], d['time_to_1e-4'])" | cut -c1-700
timeout -k 5 200 python scripts/r04_x1.py '[["seg", "dense_rows", {"CUOPT_AMD_TUNE": "panel_seg=1"}]]' | cut -c1-400
