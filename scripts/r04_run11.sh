O=$GRAFT_REPO_ROOT/gpurun_out/r04_run11; mkdir -p $O
for H in 0 1; do CUOPT_AMD_TIMING=1 CUOPT_AMD_TUNE=jag_hybrid=$H timeout -k 5 200 python bench.py --workload block_angular --no-cpu-baseline > $O/ba$H.json 2> $O/ba$H.err; grep -E "build_jag|jagged rows" $O/ba$H.err | head -6; python -c "
import json; d=json.loads(open('$O/ba$H.json').read().strip().splitlines()[-1]); print('block_angular hybrid=$H', d['value'], d['roofline']['per_kernel_ms'], d['spmv_layout'], d['time_to_1e-4']['wall_s'], d['time_to_1e-4']['setup_s'])" | cut -c1-500; done
CUOPT_AMD_TIMING=1 timeout -k 5 200 python bench.py --workload block_angular --no-cpu-baseline --no-convergence-run 2>&1 | grep -E "jagged rows|build_jag" | head
