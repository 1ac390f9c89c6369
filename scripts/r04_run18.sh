# round 4: the two hyper-parameter combinations that were refused until now, on the device against the oracle
O=$GRAFT_REPO_ROOT/gpurun_out/r04_run18; mkdir -p $O
timeout -k 5 500 python -m pytest tests/test_solve_gpu.py -m gpu -q -x -k "initial or trust_region or methodical or warm" > $O/pytest.log 2>&1; tail -15 $O/pytest.log
