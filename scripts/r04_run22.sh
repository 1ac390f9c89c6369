# round 4: soak -- back-to-back cuOptSolve calls, PDLP alone and the Concurrent race (simplex thread + helper started and cancelled per call)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_run22; mkdir -p $O
timeout -k 5 420 python scripts/soak.py > $O/soak.txt 2>&1; cat $O/soak.txt | tail -8
