# round 4: the racing engines (PDLP on the GPU, the simplex with its helper thread on the host) under AddressSanitizer + UBSan on the GPU box;
# rocprof kernel statistics of the block-angular workload on the final tree
O=$GRAFT_REPO_ROOT/gpurun_out/r04_run21; mkdir -p $O
export TMPDIR=/tmp
CUOPT_AMD_TUNE="simplex_helper_rows=1" timeout -k 5 500 bash scripts/run_sanitized.sh address tests/test_method_and_multigpu_gpu.py tests/test_simplex_through_cuoptsolve_gpu.py tests/test_doc_examples_gpu.py -m gpu > $O/asan_gpu.log 2>&1; tail -6 $O/asan_gpu.log
rm -f cuopt_amd/lib/libcuopt_san_*.so
(cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ba -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --workload block_angular --no-cpu-baseline --no-convergence-run --steps 400 --warmup 80" > $O/trace_ba.log 2>&1)
F=$(ls $O/trace_ba/*/*kernel_stats.csv 2>/dev/null | tail -1); [ -n "$F" ] && cp $F $O/r04_bench_block_angular_kernel_stats.csv && head -6 $F | cut -c1-50,150-300
rm -rf $O/trace_ba
