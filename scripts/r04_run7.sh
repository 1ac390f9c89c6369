# round 4: peer transport with the flag raised at the head of the consuming kernel; C2 floor; profiles (kernel stats + PMC) c3 / powerlaw / c2
O=$GRAFT_REPO_ROOT/gpurun_out/r04_run7; mkdir -p $O
export TMPDIR=/tmp
timeout -k 5 500 python -m pytest tests/test_p2p_transport_gpu.py tests/test_sharded_gpu.py tests/test_method_and_multigpu_gpu.py -m gpu -q > $O/pytest_p2p.log 2>&1; tail -6 $O/pytest_p2p.log
for F in "owner p2p" "owner collective" "rsag collective" "allreduce collective"; do
  set -- $F
  CUOPT_AMD_SHARD_DATAFLOW=$1 CUOPT_AMD_SHARD_TRANSPORT=$2 timeout -k 5 200 python bench.py --gpus 1 --self-launch --workload c3 --no-cpu-baseline --no-convergence-run > $O/rank1_$1_$2.json 2> $O/rank1_$1_$2.err
  python -c "
import json; d = json.loads(open('$O/rank1_$1_$2.json').read().strip().splitlines()[-1]); print('one RCCL rank, $1 / $2:', d['value'], 'it/s, ms per attempt', d['roofline']['ms_per_attempt'], d['roofline']['per_kernel_ms'])" 2>&1 | tail -1
done 2>&1 | tee $O/r04_rank1_dataflows.txt
hipcc -O3 --offload-arch=gfx950 tools/launch_floor.hip -o /tmp/launch_floor 2>/dev/null && timeout -k 5 60 /tmp/launch_floor | tee $O/r04_launch_floor.txt
for W in c2 c3 powerlaw; do
  (cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$W -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --workload $W --no-cpu-baseline --no-convergence-run --steps 400 --warmup 80" > $O/trace_$W.log 2>&1)
  F=$(ls $O/trace_$W/*/*kernel_stats.csv 2>/dev/null | tail -1); [ -n "$F" ] && cp $F $O/r04_bench_${W}_kernel_stats.csv && head -7 $F | cut -c1-60,200-330
  grep -h "^{" $O/trace_$W.log | tail -1 > $O/r04_bench_${W}_line_under_rocprof.json
done
