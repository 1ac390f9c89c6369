# usage: r02_ab2.sh <workload> [ENV=VAL ...] : bench line with layout info under extra environment
W=$1; shift
mkdir -p gpurun_out/r02
env "$@" timeout 400 python bench.py --workload $W --steps 400 --warmup 100 --no-cpu-baseline --no-convergence-run > gpurun_out/r02/ab2.json 2> gpurun_out/r02/ab2.err || tail -3 gpurun_out/r02/ab2.err
python -c "
import json; d=json.load(open('gpurun_out/r02/ab2.json')); print('$W', '$*', d['value'], {k: round(v*1e3,1) for k,v in d['roofline']['per_kernel_ms'].items()}, d['spmv_layout'])"
