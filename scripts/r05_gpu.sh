#!/bin/bash
# One parameterised runner for the round-5 GPU sessions (replaces the per-run scripts of earlier rounds).
#   scripts/r05_gpu.sh <tag> <step> [<step> ...]      outputs under gpurun_out/<tag>/
# steps: setup_tests | profsetup:<workload> | probe:<workloads,comma separated> | tests:<pytest -k expression or file> | bench:<workload> | prof:<workload> | pmc:<workload>
set -u
cd "$(dirname "$0")/.."
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for step in "$@"; do
  kind=${step%%:*}; arg=${step#*:}
  case $kind in
    setup_tests) timeout 1500 python -m pytest tests/test_device_setup_gpu.py -q -m gpu -x > "$OUT/setup_tests.log" 2>&1; tail -n 25 "$OUT/setup_tests.log" ;;
    tests) timeout 2400 python -m pytest $arg -q -m gpu > "$OUT/tests_$(echo "$arg" | tr -c 'A-Za-z0-9' _).log" 2>&1; tail -n 15 "$OUT"/tests_*.log ;;
    probe) timeout 1500 python scripts/r05_setup_probe.py $(echo "$arg" | tr ',' ' ') > "$OUT/probe.log" 2>&1; grep -E "RESULT|RATE|analysis:|ordering:|Traceback|Error" "$OUT/probe.log" | cut -c1-400 ;;
    bench) timeout 900 python bench.py --workload "$arg" > "$OUT/bench_$arg.json" 2> "$OUT/bench_$arg.err"; cut -c1-600 "$OUT/bench_$arg.json" ;;
    prof) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof_$arg" -- python "$OLDPWD/bench.py" --workload "$arg" --no-cpu-baseline --no-convergence-run > "$OLDPWD/$OUT/prof_$arg.json" 2> "$OLDPWD/$OUT/prof_$arg.err"); find "$OUT/prof_$arg" -name "*kernel_stats.csv" | head -1 | xargs -r head -n 12 ;;
    profsetup) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/profsetup_$arg" -- python "$OLDPWD/scripts/r05_setup_probe.py" "$arg" > "$OLDPWD/$OUT/profsetup_$arg.log" 2>&1); find "$OUT/profsetup_$arg" -name "*kernel_stats.csv" | head -1 | xargs -r head -n 45 | cut -c1-200 ;;
    *) echo "unknown step $step" ;;
  esac
done
