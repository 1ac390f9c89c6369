# round 3 (experiment): the panel kernel's wave-shared segment sums from 24 entries on instead of 128 (library built with
# -DPANEL_SHARE_FROM=24), power-law and dense_rows-without-segments
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/panel_share
mkdir -p $O
for W in powerlaw c3; do
  timeout 600 python bench.py --workload $W --no-cpu-baseline > $O/$W.json 2> $O/$W.err
  python -c "
import json; d = json.load(open('$O/$W.json')); r = d['roofline']; c = d['time_to_1e-4']
print('$W', d['value'], 'it/s', {k: round(v * 1e3, 1) for k, v in r['per_kernel_ms'].items()}, c['status'], c['iterations'])" || tail -3 $O/$W.err
done
