# rocprofv3 kernel statistics of the structured workloads (same command as r02_final_profiles.sh, no PMC passes)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r02/final; mkdir -p $O
for W in "$@"; do
  rm -rf $O/prof_$W
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 800 --warmup 100 --no-cpu-baseline --no-convergence-run > $O/prof_$W.log 2>&1 )
  ls $O/prof_$W/*/*kernel_stats.csv | tail -1
done
