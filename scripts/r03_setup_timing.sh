# round 3: where the set-up time of a solve goes (CUOPT_AMD_TIMING laps), C3 and the banded LP
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/setup
mkdir -p $O
for W in c3 banded; do
  CUOPT_AMD_TIMING=1 timeout 600 python bench.py --workload $W --no-cpu-baseline --steps 20 --warmup 5 > $O/$W.json 2> $O/$W.err
  grep -n "cuopt_amd" $O/$W.err | tail -80
done
