# round 3: the power-law family through the gather-free layout once its long rows are out of the hot CSR
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/powerlaw_pb
mkdir -p $O
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --workload powerlaw --no-cpu-baseline > $O/$name.json 2> $O/$name.err
  python -c "
import json; d = json.load(open('$O/$name.json')); r = d['roofline']; c = d['time_to_1e-4']
print('$name', d['value'], 'it/s', d['spmv_layout']['A']['layout'], d['spmv_layout']['At']['layout'], {k: round(v * 1e3, 1) for k, v in r['per_kernel_ms'].items()}, c['status'], c['iterations'], c['wall_s'])" || tail -3 $O/$name.err
}
run default A=1
run long256_pb CUOPT_AMD_LONG_ROWS=256 CUOPT_AMD_SPMV_LAYOUT=pb
run long256_panel CUOPT_AMD_LONG_ROWS=256
run long1024_pb CUOPT_AMD_LONG_ROWS=1024 CUOPT_AMD_SPMV_LAYOUT=pb
