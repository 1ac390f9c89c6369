# usage: r02_pmc.sh <workload> <tag> : rocprofv3 --pmc passes (one per counter group, --kernel-trace only) of the bench command
W=$1; T=$2
O=$GRAFT_REPO_ROOT/gpurun_out/r02/pmc_$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 80 --warmup 20 --no-cpu-baseline --no-convergence-run > $O/p$i.log 2>&1
  tail -1 $O/p$i.log | cut -c1-200
done
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $O/summary.json $O/p1 $O/p2 $O/p3 $O/p4 > $O/summary.txt
grep -E "k_jag_a_dual|k_jag_at_step|k_panel_a_dual|k_panel_at_step|k_primal " $O/summary.txt
