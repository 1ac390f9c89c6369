# round 3: the gather-free layout inside the product: full-size tests, bench lines at C3 (forced) and at ten times C3 (auto)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/pbprod
mkdir -p $O
timeout 1500 python -m pytest tests/test_full_size_gpu.py -x -q 2>&1 | tail -5
CUOPT_AMD_TIMING=1 timeout 600 python bench.py --workload c3 --spmv-layout pb --no-cpu-baseline > $O/c3_pb.json 2> $O/c3_pb.err; tail -c 1500 $O/c3_pb.json
timeout 600 python bench.py --workload c3 --no-cpu-baseline > $O/c3_auto.json 2> $O/c3_auto.err; tail -c 600 $O/c3_auto.json
CUOPT_AMD_TIMING=1 timeout 1500 python bench.py --workload c3x10 --no-cpu-baseline > $O/c3x10_auto.json 2> $O/c3x10_auto.err; tail -c 1500 $O/c3x10_auto.json
timeout 1500 python bench.py --workload c3x10 --spmv-layout panel --no-cpu-baseline > $O/c3x10_panel.json 2> $O/c3x10_panel.err; tail -c 1200 $O/c3x10_panel.json
grep "setup\]" $O/c3x10_auto.err | head -40
