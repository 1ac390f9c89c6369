# round 3 (end): the step decision at the head of the following primal step (CUOPT_AMD_FUSED_DECISION=1 second stream, =2 one stream; this file: the run of =2): parity tests with it on, c3 / c2 lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/fused_decision
mkdir -p $O
CUOPT_AMD_FUSED_DECISION=2 timeout 600 python -m pytest tests/test_solve_gpu.py tests/test_structured_gpu.py tests/test_panel_layout_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -3
run() {
  local name=$1 w=$2; shift 2
  env "$@" timeout 600 python bench.py --workload $w --no-cpu-baseline > $O/$name.json 2> $O/$name.err
  python -c "
import json; d = json.load(open('$O/$name.json')); r = d['roofline']; c = d['time_to_1e-4']
print('$name', d['value'], 'it/s', {k: round(v * 1e3, 1) for k, v in r['per_kernel_ms'].items()}, c['status'], c['iterations'], c['wall_s'], c['objective'])" || tail -5 $O/$name.err
}
run c3_fused2 c3 CUOPT_AMD_FUSED_DECISION=2
run c3_plain c3 A=1
run c2_fused2 c2 CUOPT_AMD_FUSED_DECISION=2
run c2_plain c2 A=1
