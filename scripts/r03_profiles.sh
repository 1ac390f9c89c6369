# round 3: kernel statistics + PMC passes of the bench command (C3), P2P / owner dataflow at one RCCL rank, structured lines
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03/prof
mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --workload c3 --no-cpu-baseline --no-convergence-run --steps 400 --warmup 80"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $O/trace_c3.log 2>&1)
find $O/trace_c3 -name '*kernel_stats.csv' | head -1 | xargs cat | head -8
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_LDS"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_c3_$i -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $O/pmc_c3_$i.log 2>&1)
done
python scripts/pmc_summary.py $O/r03_pmc_c3.json $O/pmc_c3_1 $O/pmc_c3_2 $O/pmc_c3_3 $O/pmc_c3_4 > $O/r03_pmc_c3_summary.txt
grep -E "k_panel_a_dual|k_panel_at_step" $O/r03_pmc_c3_summary.txt | grep -E "traffic|FETCH|WRITE"
# the sharded dataflows through real RCCL at ONE rank (what the communicator calls and the peer transport cost on top of one GPU)
for F in "allreduce collective" "rsag collective" "owner collective" "owner p2p"; do
  set -- $F
  CUOPT_AMD_SHARD_DATAFLOW=$1 CUOPT_AMD_SHARD_TRANSPORT=$2 timeout 600 python bench.py --gpus 1 --self-launch --workload c3 --no-cpu-baseline --no-convergence-run > $O/rank1_$1_$2.json 2> $O/rank1_$1_$2.err
  python -c "
import json; d = json.load(open('$O/rank1_$1_$2.json')); print('one RCCL rank, $1 / $2:', d['value'], 'it/s', d['config']['parallelism'])"
done
