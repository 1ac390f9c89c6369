# A/B of library builds on the banded workload: prints per-kernel times
mkdir -p gpurun_out/r02
for L in "$@"; do
  CUOPT_AMD_LIB=$PWD/cuopt_amd/lib/$L timeout 300 python bench.py --workload ${WL:-banded} --steps 400 --warmup 100 --no-cpu-baseline --no-convergence-run > gpurun_out/r02/ab_$L.json 2> gpurun_out/r02/ab_$L.err || tail -3 gpurun_out/r02/ab_$L.err
  python -c "
import json,sys; d=json.load(open('gpurun_out/r02/ab_$L.json')); print('$L', d['value'], d['roofline']['per_kernel_ms'])"
done
