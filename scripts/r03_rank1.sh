cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03/prof
mkdir -p $O
for F in "owner p2p" "owner collective"; do
  set -- $F
  CUOPT_AMD_SHARD_DATAFLOW=$1 CUOPT_AMD_SHARD_TRANSPORT=$2 timeout 600 python bench.py --gpus 1 --self-launch --workload c3 --no-cpu-baseline --no-convergence-run > $O/rank1_$1_$2.json 2> $O/rank1_$1_$2.err
  python -c "
import json; d = json.load(open('$O/rank1_$1_$2.json')); print('one RCCL rank, $1 / $2:', d['value'], 'it/s', d['config']['parallelism'])"
done
timeout 900 python -m pytest tests/test_p2p_transport_gpu.py -x -q 2>&1 | tail -3
