# round 4: where the peer transport's time goes at one RCCL rank (kernel trace), captured collectives in the product, fixed tests
O=gpurun_out/r04_run5; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --gpus 1 --self-launch --workload c3 --no-cpu-baseline --no-convergence-run --steps 400 --warmup 80"
(cd /tmp && CUOPT_AMD_SHARD_DATAFLOW=owner CUOPT_AMD_SHARD_TRANSPORT=p2p timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p2p_trace -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $O/p2p_trace.log 2>&1)
F=$(ls $O/p2p_trace/*/*kernel_stats.csv 2>/dev/null | tail -1); [ -n "$F" ] && head -14 $F | cut -c1-150
tail -2 $O/p2p_trace.log | cut -c1-300
for G in "" "--graph-comm"; do
  CUOPT_AMD_SHARD_DATAFLOW=owner CUOPT_AMD_SHARD_TRANSPORT=collective timeout -k 5 240 $B $G > $O/coll$G.json 2> $O/coll$G.err
  python -c "
import json; d = json.loads(open('$O/coll$G.json').read().strip().splitlines()[-1]); print('one RCCL rank, owner / collective $G:', d['value'], 'it/s')" 2>&1 | tail -1
  tail -3 $O/coll$G.err | cut -c1-300
done
timeout -k 5 600 python -m pytest tests/test_full_size_gpu.py tests/test_method_and_multigpu_gpu.py tests/test_doc_examples_gpu.py -m gpu -q > $O/pytest_b.log 2>&1; tail -8 $O/pytest_b.log
