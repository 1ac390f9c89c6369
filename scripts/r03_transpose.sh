# round 3 (late): blocked against direct host transposition, set-up time of c3 (same box, alternating)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/transpose
mkdir -p $O
for R in 1 2 3; do
for MODE in blocked direct; do
  if [ $MODE = direct ]; then export CUOPT_AMD_TRANSPOSE_DIRECT=1; else unset CUOPT_AMD_TRANSPOSE_DIRECT; fi
  CUOPT_AMD_TIMING=1 timeout 600 python bench.py --workload c3 --no-cpu-baseline --steps 20 --warmup 5 > $O/c3_${MODE}_$R.json 2> $O/c3_${MODE}_$R.err
  python -c "
import json; d = json.load(open('$O/c3_${MODE}_$R.json')); c = d['time_to_1e-4']
print('$MODE', 'wall', c['wall_s'], 'setup', c['setup_s'])"
  grep "wait for the A^T side" $O/c3_${MODE}_$R.err | tail -1
done; done
