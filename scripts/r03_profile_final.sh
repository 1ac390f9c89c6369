# round 3 (end): kernel statistics of the bench command with the final set-up kernels (k_row_norm_blocks, k_scale_matrix_blocks)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03/prof_final
mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --workload c3 --no-cpu-baseline --steps 400 --warmup 80"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $O/trace_c3.log 2>&1)
F=$(find $O/trace_c3 -name '*kernel_stats.csv' | head -1)
cp $F $O/r03_bench_c3_kernel_stats_final.csv
head -12 $F | cut -c1-160
grep -E "k_row_norm|k_scale_matrix" $F | cut -c1-200
tail -c 300 $O/trace_c3.log
