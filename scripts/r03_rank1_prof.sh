cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03/prof
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && CUOPT_AMD_SHARD_DATAFLOW=owner CUOPT_AMD_SHARD_TRANSPORT=p2p timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_p2p -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --gpus 1 --force-comm --workload c3 --no-cpu-baseline --no-convergence-run --steps 400 --warmup 80" > $O/trace_p2p.log 2>&1)
find $O/trace_p2p -name '*kernel_stats.csv' | head -1 | xargs cat | head -12 | cut -c1-50,150-400
