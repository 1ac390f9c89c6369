# round 3 (end): PMC passes of the power-law workload (where do the 14 us over c3's per-nonzero cost go?)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03/pmc_powerlaw
mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --workload powerlaw --no-cpu-baseline --no-convergence-run --steps 200 --warmup 40"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$i -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $O/pmc_$i.log 2>&1)
done
python scripts/pmc_summary.py $O/r03_pmc_powerlaw.json $O/pmc_1 $O/pmc_2 $O/pmc_3 $O/pmc_4 $O/pmc_5 > $O/r03_pmc_powerlaw_summary.txt
grep -E "^k_panel_a_dual|^k_panel_at_step" $O/r03_pmc_powerlaw_summary.txt
