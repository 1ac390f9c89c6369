# Round-2 evidence for profiles/: for each workload the bench line (default flags and the driver's --steps 20 --warmup 5),
# the rocprofv3 --kernel-trace --stats summary of the same command and the --pmc passes (separate runs, kernel-trace only).
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r02/final; mkdir -p $O
for W in "$@"; do
  timeout 900 python bench.py --workload $W > $O/bench_${W}_default.json 2> $O/bench_${W}_default.err || tail -3 $O/bench_${W}_default.err
  timeout 900 python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${W}_driver.json 2> $O/bench_${W}_driver.err || tail -3 $O/bench_${W}_driver.err
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 800 --warmup 100 --no-cpu-baseline --no-convergence-run > $O/prof_$W.log 2>&1 )
  i=0
  for C in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_LDS"; do
    i=$((i+1))
    ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$W/p$i -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 80 --warmup 20 --no-cpu-baseline --no-convergence-run > $O/pmc_$W.p$i.log 2>&1 )
  done
  python scripts/pmc_summary.py $O/pmc_$W.json $O/pmc_$W/p1 $O/pmc_$W/p2 $O/pmc_$W/p3 $O/pmc_$W/p4 > $O/pmc_${W}_summary.txt
  python - <<PY
import json
for f in ("default", "driver"):
    d = json.load(open("$O/bench_${W}_%s.json" % f)); r = d["roofline"]
    print("$W", f, "it/s", d["value"], "timed", d["timed_steps"], "dom", r["kernel"], "frac", r["frac"], {k: round(v * 1e3, 1) for k, v in r["per_kernel_ms"].items()})
PY
done
