# round 4: where the peer transport's time goes at one RCCL rank (kernel trace)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_run6; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --gpus 1 --self-launch --workload c3 --no-cpu-baseline --no-convergence-run --steps 400 --warmup 80"
(cd /tmp && CUOPT_AMD_SHARD_DATAFLOW=owner CUOPT_AMD_SHARD_TRANSPORT=p2p timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p2p_trace -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $O/p2p_trace.log 2>&1)
F=$(ls $O/p2p_trace/*/*kernel_stats.csv 2>/dev/null | tail -1); [ -n "$F" ] && head -14 $F | cut -c1-150
grep -h "^{" $O/p2p_trace.log | cut -c1-200
CUOPT_AMD_SHARD_DATAFLOW=owner CUOPT_AMD_SHARD_TRANSPORT=collective timeout -k 5 240 $B > $O/coll.json 2> $O/coll.err; cut -c1-120 $O/coll.json
