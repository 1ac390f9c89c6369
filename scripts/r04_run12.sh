# round 4: the sanitizer builds on the GPU box -- the racing threads of Concurrent solves, sharded thread-per-GPU solves (in-process communicator)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_run12; mkdir -p $O
export ASAN_OPTIONS_EXTRA=protect_shadow_gap=0
( timeout -k 5 420 scripts/run_sanitized.sh address tests/test_method_and_multigpu_gpu.py tests/test_doc_examples_gpu.py -m gpu > $O/asan.log 2>&1; echo "asan exit $?" ) ; tail -6 $O/asan.log | cut -c1-200
( timeout -k 5 420 scripts/run_sanitized.sh thread tests/test_method_and_multigpu_gpu.py -m gpu -k "dual_simplex or concurrent or race" > $O/tsan.log 2>&1; echo "tsan exit $?" ); tail -8 $O/tsan.log | cut -c1-200
