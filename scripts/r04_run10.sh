# round 4: hybrid (range + list) column sets of the jagged layout on the block-angular workload + its parity test; PMC summaries
O=$GRAFT_REPO_ROOT/gpurun_out/r04_run10; mkdir -p $O
timeout -k 5 300 python -m pytest tests/test_panel_layout_gpu.py -m gpu -q -k "range_plus_list or jag" > $O/pytest_jag.log 2>&1; tail -5 $O/pytest_jag.log
timeout -k 5 300 python scripts/r04_x1.py '[
 ["list", "block_angular", {"CUOPT_AMD_TUNE": "jag_hybrid=0"}],
 ["auto", "block_angular", {}],
 ["auto", "multiband", {}],
 ["auto", "staircase", {}],
 ["auto", "banded", {}]
]' 2>&1 | cut -c1-420 | tee $O/r04_jag_hybrid.txt
CUOPT_AMD_TIMING=1 timeout -k 5 200 python bench.py --workload block_angular --no-cpu-baseline > $O/ba.json 2> $O/ba.err; grep -E "build_jag|jag" $O/ba.err | head -8; python -c "
import json; d=json.loads(open('$O/ba.json').read().strip().splitlines()[-1]); print('block_angular', d['value'], d['roofline']['per_kernel_ms'], d['spmv_layout'], d['time_to_1e-4'])" | cut -c1-600
bash scripts/r04_pmc.sh 2>&1 | tail -30
