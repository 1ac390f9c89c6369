# round 4: own rows only beyond a panel's size (long-tail panels), the peer transport at one RCCL rank, RCCL capture probe, full suite
O=gpurun_out/r04_run4; mkdir -p $O
timeout -k 5 200 python scripts/r04_x1.py '[
 ["seg", "powerlaw", {"CUOPT_AMD_TUNE": "panel_seg=1"}],
 ["seg_denseoff", "dense_rows", {"CUOPT_AMD_TUNE": "panel_seg=1,dense=0"}]
]' 2>&1 | cut -c1-420 | tee $O/table.txt
for F in "owner p2p" "owner collective"; do
  set -- $F
  CUOPT_AMD_SHARD_DATAFLOW=$1 CUOPT_AMD_SHARD_TRANSPORT=$2 timeout -k 5 240 python bench.py --gpus 1 --self-launch --workload c3 --no-cpu-baseline --no-convergence-run > $O/rank1_$1_$2.json 2> $O/rank1_$1_$2.err
  python -c "
import json; d = json.loads(open('$O/rank1_$1_$2.json').read().strip().splitlines()[-1]); print('one RCCL rank, $1 / $2:', d['value'], 'it/s', d['roofline']['per_kernel_ms'])" 2>&1 | tail -1
done 2>&1 | tee $O/r04_rank1_dataflows.txt
TL=$(python -c "import torch,os; print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
hipcc -O1 --offload-arch=gfx950 tools/rccl_capture_repro.cpp -o /tmp/rccl_capture_repro -ldl 2>/dev/null
for M in 0 1 2; do echo "== torch rccl, mode $M"; LD_LIBRARY_PATH=$TL:$LD_LIBRARY_PATH timeout -k 5 60 /tmp/rccl_capture_repro $M 2>&1 | tail -6; echo "exit $?"; done 2>&1 | tee $O/rccl_capture.txt
for M in 1 2; do echo "== rocm rccl, mode $M"; LD_LIBRARY_PATH=/opt/rocm/lib timeout -k 5 60 /tmp/rccl_capture_repro $M 2>&1 | tail -6; echo "exit $?"; done 2>&1 | tee -a $O/rccl_capture.txt
timeout -k 5 1200 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; tail -12 $O/pytest_all.log
