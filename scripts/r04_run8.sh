# round 4: the stream layout with ordinary (L2-keeping) loads on cache-resident matrices: C2 and two sizes above it, both ways
O=$GRAFT_REPO_ROOT/gpurun_out/r04_run8; mkdir -p $O
timeout -k 5 400 python scripts/r04_x1.py '[
 ["nt", "c2", {"CUOPT_AMD_TUNE": "stream_keep=0"}],
 ["keep", "c2", {"CUOPT_AMD_TUNE": "stream_keep=1"}],
 ["nt", "rand:150000:10", {"CUOPT_AMD_TUNE": "stream_keep=0"}],
 ["keep", "rand:150000:10", {"CUOPT_AMD_TUNE": "stream_keep=1"}],
 ["nt", "rand:250000:10", {"CUOPT_AMD_TUNE": "stream_keep=0", "CUOPT_AMD_SPMV_LAYOUT": "stream"}],
 ["keep", "rand:250000:10", {"CUOPT_AMD_TUNE": "stream_keep=1", "CUOPT_AMD_SPMV_LAYOUT": "stream"}],
 ["nt", "rand:50000:10", {"CUOPT_AMD_TUNE": "stream_keep=0"}],
 ["keep", "rand:50000:10", {"CUOPT_AMD_TUNE": "stream_keep=1"}]
]' 2>&1 | sed 's/"layout": {.*"resident": false}, //' | cut -c1-300 | tee $O/r04_stream_keep.txt
for K in 0 1; do CUOPT_AMD_TUNE=stream_keep=$K timeout -k 5 120 python bench.py --workload c2 --no-cpu-baseline --no-convergence-run > $O/c2_keep$K.json 2> $O/c2_keep$K.err; python -c "
import json; d=json.loads(open('$O/c2_keep$K.json').read().strip().splitlines()[-1]); print('c2 stream_keep=$K', d['value'], d['roofline']['per_kernel_ms'])"; done | tee -a $O/r04_stream_keep.txt
