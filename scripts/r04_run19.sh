# round 4, after the simplex changes (bound flipping, helper thread) and the two formerly refused hyper-parameter combinations: whole GPU suite, smoke, the default bench line
O=$GRAFT_REPO_ROOT/gpurun_out/r04_run19; mkdir -p $O
timeout -k 5 900 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; tail -4 $O/pytest_all.log; grep -E "^FAILED|^ERROR" $O/pytest_all.log | head -20
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 5 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
