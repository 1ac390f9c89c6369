# round 4: PMC passes (FETCH / WRITE / L2 / SQ; no TA_* counters) of the block-angular and dense_rows workloads
O=$GRAFT_REPO_ROOT/gpurun_out/r04_pmc2; mkdir -p $O
export TMPDIR=/tmp
for W in block_angular dense_rows; do
  B="python bench.py --workload $W --no-cpu-baseline --no-convergence-run --steps 200 --warmup 40"
  i=0
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_LDS"; do
    i=$((i+1))
    (cd /tmp && timeout -k 5 60 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${W}_pmc_$i -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $O/${W}_pmc_$i.log 2>&1)
  done
  python scripts/pmc_summary.py $O/r04_pmc_$W.json $O/${W}_pmc_1 $O/${W}_pmc_2 $O/${W}_pmc_3 $O/${W}_pmc_4 > $O/r04_pmc_${W}_summary.txt
  grep -E "traffic MB" $O/r04_pmc_${W}_summary.txt | head -6
  rm -rf $O/${W}_pmc_?
done
