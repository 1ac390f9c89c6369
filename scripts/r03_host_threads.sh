# round 3 (late): set-up time of c3 against the number of host threads of the set-up passes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/host_threads
mkdir -p $O
for T in 8 16 32 64; do
  CUOPT_AMD_HOST_THREADS=$T timeout 600 python bench.py --workload c3 --no-cpu-baseline --steps 20 --warmup 5 > $O/c3_$T.json 2> $O/c3_$T.err
  python -c "
import json; d = json.load(open('$O/c3_$T.json')); c = d['time_to_1e-4']
print('threads $T', 'wall', c['wall_s'], 'setup', c['setup_s'], 'loop', c['loop_s'])"
done
