# single-rank RCCL path (torchrun, 1 process): graphs on / off
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02
for G in 1 0; do
  CUOPT_AMD_GRAPH_COMM=$G timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --force-comm --steps 400 --warmup 100 --no-cpu-baseline --no-convergence-run > gpurun_out/r02/comm1_graph$G.json 2> gpurun_out/r02/comm1_graph$G.err || tail -5 gpurun_out/r02/comm1_graph$G.err
  python -c "
import json; d=json.load(open('gpurun_out/r02/comm1_graph$G.json')); print('graph_comm=$G', d['value'], d['ms_per_step'])"
done
