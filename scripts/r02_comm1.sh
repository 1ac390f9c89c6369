# single-rank RCCL path (torchrun, 1 process): the two sharded dataflows through the real RCCL calls
# (ncclAllReduce | ncclReduceScatter + ncclAllGather + ncclAllReduce of 3 scalars)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02
for F in allreduce rsag; do
  CUOPT_AMD_SHARD_DATAFLOW=$F timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --force-comm --steps 400 --warmup 100 --no-cpu-baseline --no-convergence-run > gpurun_out/r02/comm1_$F.json 2> gpurun_out/r02/comm1_$F.err || tail -5 gpurun_out/r02/comm1_$F.err
  python -c "
import json; d=json.load(open('gpurun_out/r02/comm1_$F.json')); print('dataflow=$F', d['value'], 'it/s', d['ms_per_step'], 'ms/step', d['config']['parallelism'])"
done
