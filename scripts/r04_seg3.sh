# round 4: long-tail panels v3 (round-major storage, per-round wave scans, edge records joined by wave 0) + C2 latency cuts + p2p own-slice
O=gpurun_out/r04_seg3; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_panel_seg_gpu.py tests/test_kernels_gpu.py tests/test_p2p_transport_gpu.py -m gpu -q > $O/pytest_a.log 2>&1; tail -25 $O/pytest_a.log
timeout -k 5 300 python scripts/r04_x1.py '[
 ["row", "c3", {"CUOPT_AMD_TUNE": "panel_seg=0"}],
 ["seg", "c3", {"CUOPT_AMD_TUNE": "panel_seg=1"}],
 ["row", "powerlaw", {"CUOPT_AMD_TUNE": "panel_seg=0"}],
 ["seg", "powerlaw", {"CUOPT_AMD_TUNE": "panel_seg=1"}],
 ["seg", "dense_rows", {"CUOPT_AMD_TUNE": "panel_seg=1"}],
 ["auto", "dense_rows", {}],
 ["auto", "c2", {}]
]' 2>&1 | cut -c1-400 | tee $O/table.txt
timeout -k 5 120 python bench.py --workload c2 --no-cpu-baseline --no-convergence-run > $O/c2.json 2> $O/c2.err; python -c "
import json; d=json.loads(open('$O/c2.json').read().strip().splitlines()[-1]); print('c2', d['value'], d['roofline']['per_kernel_ms'], d['roofline'].get('attempt_kernels_over_ms_per_attempt'))"
