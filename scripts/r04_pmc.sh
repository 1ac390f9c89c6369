# round 4: PMC passes (FETCH / WRITE / L2 / SQ; no TA_* counters: that pass hung rocprofv3 on this pool) of c3 and the power-law workload
O=$GRAFT_REPO_ROOT/gpurun_out/r04_pmc; mkdir -p $O
export TMPDIR=/tmp
for W in powerlaw c3; do
  B="python bench.py --workload $W --no-cpu-baseline --no-convergence-run --steps 200 --warmup 40"
  i=0
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU"; do
    i=$((i+1))
    (cd /tmp && timeout -k 5 90 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${W}_pmc_$i -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $O/${W}_pmc_$i.log 2>&1)
  done
  python scripts/pmc_summary.py $O/r04_pmc_$W.json $O/${W}_pmc_1 $O/${W}_pmc_2 $O/${W}_pmc_3 $O/${W}_pmc_4 $O/${W}_pmc_5 > $O/r04_pmc_${W}_summary.txt
  grep -E "^k_panel_a_dual|^k_panel_at_step" $O/r04_pmc_${W}_summary.txt | cut -c1-120
  rm -rf $O/${W}_pmc_?   # (the raw per-dispatch tables are tens of MB: only the summaries travel back)
done
# the pull of the peer transport at two ranks on one device (in-process communicator): what a 4 MB remote half costs
(cd /tmp && CUOPT_AMD_SHARD_DATAFLOW=owner CUOPT_AMD_SHARD_TRANSPORT=p2p CUOPT_AMD_TUNE=soft_communicator=1 GPU_MAX_HW_QUEUES=16 timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/soft2 -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
from cuopt_amd import capi, synthetic
p = synthetic.generate(**synthetic.CONFIGS['c3'])
r = capi.solve(p, method=1, tol=0.0, iteration_limit=400, amd_num_gpus=2)
print(r['status'], r['steps_taken'])
" > $O/soft2.log 2>&1)
F=$(ls $O/soft2/*/*kernel_stats.csv 2>/dev/null | tail -1); [ -n "$F" ] && cp $F $O/r04_soft2_p2p_kernel_stats.csv && head -12 $F | cut -c1-70,150-260
tail -3 $O/soft2.log | cut -c1-200
rm -rf $O/soft2
