# round 4, last tree: whole GPU suite + smoke + the default bench line
O=$GRAFT_REPO_ROOT/gpurun_out/r04_run20; mkdir -p $O
timeout -k 5 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; tail -4 $O/pytest_all.log; grep -E "^FAILED|^ERROR" $O/pytest_all.log | head -20
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 5 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
