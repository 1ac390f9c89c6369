# usage: r02_prof.sh <workload> <tag>   -> bench line + rocprofv3 kernel stats of the same command
W=$1; T=$2
O=$GRAFT_REPO_ROOT/gpurun_out/r02
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --workload $W --steps 800 --warmup 100 --no-cpu-baseline > $O/bench_${T}.json 2> $O/bench_${T}.err || tail -5 $O/bench_${T}.err
python -c "
import json; d=json.load(open('$O/bench_${T}.json')); print('$T', d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['per_kernel_ms'])"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T} -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 800 --warmup 100 --no-cpu-baseline --no-convergence-run > $O/prof_${T}.log 2>&1
cd $GRAFT_REPO_ROOT
F=$(ls $O/prof_${T}/*/*kernel_stats.csv | head -1)
head -12 $F | cut -d, -f1-8
