# round 4: validation -- the whole GPU suite, smoke(), the driver's bench command, the thread-sanitizer flavour on the racing engines
O=$GRAFT_REPO_ROOT/gpurun_out/r04_run14; mkdir -p $O
timeout -k 5 1200 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; tail -5 $O/pytest_all.log
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -3 $O/bench_driver.err; cut -c1-1500 $O/bench_driver.json
( timeout -k 5 300 scripts/run_sanitized.sh thread tests/test_method_and_multigpu_gpu.py tests/test_simplex_through_cuoptsolve_gpu.py -m gpu -k "dual_simplex or concurrent or race or maximisation" > $O/tsan.log 2>&1; echo "tsan exit $?" ); tail -6 $O/tsan.log | cut -c1-200
