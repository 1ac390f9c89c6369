# round 3: the whole GPU suite, smoke, and the bench lines quoted in DESIGN / README
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/final
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > $O/c3_driver_flags.json 2> $O/c3_driver_flags.err; tail -c 700 $O/c3_driver_flags.json
for W in c3 banded staircase block_angular multiband powerlaw dense_rows c2; do
  timeout 900 python bench.py --workload $W --no-cpu-baseline > $O/$W.json 2> $O/$W.err
  python -c "
import json; d = json.load(open('$O/$W.json')); r = d['roofline']; c = d['time_to_1e-4']
print('$W', d['value'], 'it/s', d['spmv_layout']['A']['layout'], d['spmv_layout']['At']['layout'], {k: round(v * 1e3, 1) for k, v in r['per_kernel_ms'].items()}, 'frac', r['frac'], c['status'], c['iterations'], c['wall_s'])"
done
