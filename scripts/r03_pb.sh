# round 3: the gather-free SpMV harness on the GPU box (sweep + kernel trace of one configuration)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03/pb
mkdir -p $O
B=$GRAFT_REPO_ROOT/tools/bin/spmv_pb
timeout 1200 $B 1000000 10 20 ${2:-sweep} > $O/sweep.txt 2>&1
tail -40 $O/sweep.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B 1000000 10 30 prof ${1:-0} > $O/trace.log 2>&1
find $O/trace -name '*kernel_stats.csv' | head -1 | xargs cat | head -12
