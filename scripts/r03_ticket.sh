# round 3, item 8: the step decision in the tail of the A^T y' kernel (last workgroup by ticket) against its own launch, C2
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/ticket
mkdir -p $O
for T in 0 1 0 1; do
  CUOPT_AMD_TICKET_DECISION=$T timeout 600 python bench.py --workload c2 --no-cpu-baseline --no-convergence-run > $O/c2_t$T.json 2> $O/c2_t$T.err
  python - <<PY
import json
d = json.load(open("$O/c2_t$T.json"))
print("ticket=$T", d["value"], "it/s", d["ms_per_step"], "ms", {k: round(v * 1e3, 2) for k, v in d["roofline"]["per_kernel_ms"].items()})
PY
done
CUOPT_AMD_TICKET_DECISION=1 timeout 900 python -m pytest tests/test_solve_gpu.py -x -q -k "small_lps or oracle or first" 2>&1 | tail -3
