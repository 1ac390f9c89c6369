# round 4: long-row workgroups first in the jagged layout's grid; the whole suite on the re-split device layer
O=$GRAFT_REPO_ROOT/gpurun_out/r04_run15; mkdir -p $O
timeout -k 5 200 python scripts/r04_x1.py '[["auto", "block_angular", {}]]' | cut -c1-420 | tee $O/ba.txt
timeout -k 5 1200 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; tail -5 $O/pytest_all.log; grep -E "^FAILED" $O/pytest_all.log | head
