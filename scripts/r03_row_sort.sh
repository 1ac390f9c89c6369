# round 3 (end): panel rows sorted by length inside each panel (lane slot -> row through PanelView::rperm) on matrices with long rows
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/row_sort
mkdir -p $O
timeout 900 python -m pytest tests/test_panel_layout_gpu.py tests/test_full_size_gpu.py tests/test_structured_gpu.py tests/test_dense_segments_gpu.py tests/test_long_row_extraction_gpu.py tests/test_kernels_gpu.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -3
run() {
  local name=$1 w=$2; shift 2
  env "$@" timeout 600 python bench.py --workload $w --no-cpu-baseline > $O/$name.json 2> $O/$name.err
  python -c "
import json; d = json.load(open('$O/$name.json')); r = d['roofline']; c = d['time_to_1e-4']
print('$name', d['value'], 'it/s', {k: round(v * 1e3, 1) for k, v in r['per_kernel_ms'].items()}, c['status'], c['iterations'], c['wall_s'])" || tail -3 $O/$name.err
}
run powerlaw_sorted powerlaw A=1
run powerlaw_natural powerlaw CUOPT_AMD_PANEL_ROW_SORT=0
run dense_rows_off_sorted dense_rows CUOPT_AMD_DENSE=0
run dense_rows_off_natural dense_rows CUOPT_AMD_DENSE=0 CUOPT_AMD_PANEL_ROW_SORT=0
run c3 c3 A=1
