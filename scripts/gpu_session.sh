#!/bin/bash
# One parameterised runner for the GPU sessions (rounds 5 and 6; replaces the per-run scripts of earlier rounds).
#   scripts/gpu_session.sh <tag> <step> [<step> ...]      outputs under gpurun_out/<tag>/      (RND=r06: prefix of the files meant for profiles/)
# steps: contention:<rounds> | setup_tests | profsetup:<workload> | probe:<workloads,comma separated> | tests:<pytest -k expression or file> | bench:<workload> | prof:<workload> | pmc:<workload>
set -u
cd "$(dirname "$0")/.."
TAG=$1; shift
RND=${RND:-r06}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export CUOPT_AMD_LP_CACHE=/tmp/lpcache
for step in "$@"; do
  kind=${step%%:*}; arg=${step#*:}
  case $kind in
    setup_tests) timeout 1500 python -m pytest tests/test_device_setup_gpu.py -q -m gpu -x > "$OUT/setup_tests.log" 2>&1; tail -n 25 "$OUT/setup_tests.log" ;;
    tests) timeout 2400 python -m pytest $arg -q -m gpu > "$OUT/tests_$(echo "$arg" | tr -c 'A-Za-z0-9' _).log" 2>&1; tail -n 15 "$OUT"/tests_*.log ;;
    probe) timeout 1500 python scripts/r05_setup_probe.py $(echo "$arg" | tr ',' ' ') > "$OUT/probe.log" 2>&1; grep -E "RESULT|RATE|analysis:|ordering:|Traceback|Error" "$OUT/probe.log" | cut -c1-400 ;;
    bench) timeout 900 python bench.py --workload "$arg" > "$OUT/bench_$arg.json" 2> "$OUT/bench_$arg.err"; cut -c1-700 "$OUT/bench_$arg.json" ;;
    benchfast) timeout 900 python bench.py --workload "$arg" --no-cpu-baseline > "$OUT/bench_$arg.json" 2> "$OUT/bench_$arg.err"; cut -c1-700 "$OUT/bench_$arg.json" ;;
    prof) R=$PWD; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/prof_$arg" -- bash -c "cd $R && python bench.py --workload $arg --no-cpu-baseline --no-convergence-run --steps 400 --warmup 80 --min-seconds 0.2" > "$R/$OUT/prof_$arg.json" 2> "$R/$OUT/prof_$arg.err")
          F=$(find "$OUT/prof_$arg" -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" "$OUT/${RND}_bench_${arg}_kernel_stats.csv" && head -n 8 "$F" | cut -c1-160; rm -rf "$OUT/prof_$arg" ;;
    pmc) R=$PWD; i=0
         for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_LDS"; do
           i=$((i+1))
           (cd /tmp && timeout -k 5 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$R/$OUT/${arg}_pmc_$i" -- bash -c "cd $R && BENCH_USE_GRAPH=${BENCH_USE_GRAPH:-1} python bench.py --workload $arg --no-cpu-baseline --no-convergence-run --steps 200 --warmup 40 --min-seconds 0.05" > "$R/$OUT/${arg}_pmc_$i.log" 2>&1)
         done
         python scripts/pmc_summary.py "$OUT/${RND}_pmc_$arg.json" "$OUT/${arg}_pmc_1" "$OUT/${arg}_pmc_2" "$OUT/${arg}_pmc_3" "$OUT/${arg}_pmc_4" > "$OUT/${RND}_pmc_${arg}_summary.txt"
         grep -E "traffic MB|FETCH_SIZE|WRITE_SIZE" "$OUT/${RND}_pmc_${arg}_summary.txt" | head -12 | cut -c1-120
         rm -rf "$OUT/${arg}"_pmc_? ;;
    profsetup) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/profsetup_$arg" -- env PROBE_NO_RATE=1 python "$OLDPWD/scripts/r05_setup_probe.py" "$arg" > "$OLDPWD/$OUT/profsetup_$arg.log" 2>&1); find "$OUT/profsetup_$arg" -name "*kernel_stats.csv" | head -1 | xargs -r head -n 45 | cut -c1-200 ;;
    apitrace) (cd /tmp && timeout 900 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace -d "$OLDPWD/$OUT/apitrace_$arg" -- python "$OLDPWD/scripts/r05_setup_probe.py" "$arg" > "$OLDPWD/$OUT/apitrace_$arg.log" 2>&1); ls -la "$OUT/apitrace_$arg"/* | head ;;
    yardstick) python scripts/dump_csr.py "$arg" /tmp/csr_$arg > "$OUT/yardstick_$arg.log" 2>&1
               /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-deprecated-declarations tools/rocsparse_yardstick.cpp -lrocsparse -o /tmp/rocsparse_yardstick >> "$OUT/yardstick_$arg.log" 2>&1
               for alg in default csr_rowsplit; do timeout 60 /tmp/rocsparse_yardstick /tmp/csr_$arg $alg >> "$OUT/${RND}_rocsparse_yardstick_$arg.txt" 2>&1; done; cat "$OUT/${RND}_rocsparse_yardstick_$arg.txt" ;;
    gatherprobe) /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o /tmp/gather_probe > "$OUT/gather_probe_build.log" 2>&1
               timeout 600 /tmp/gather_probe planes > "$OUT/${RND}_gather_planes.txt" 2>&1; cat "$OUT/${RND}_gather_planes.txt"
               R=$PWD; (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/$OUT/gcal_fetch" -- /tmp/gather_probe calibrate > "$R/$OUT/${RND}_gather_calibration.txt" 2>&1)
               (cd /tmp && timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$R/$OUT/gcal_req" -- /tmp/gather_probe calibrate > /dev/null 2>&1)
               python - "$OUT" >> "$OUT/${RND}_gather_calibration.txt" <<'PYEOF'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("gcal_fetch", "gcal_req"):
    for f in glob.glob("%s/%s/*/*_counter_collection.csv" % (out, d)):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("counters per launch, in launch order (3 launches per configuration: strides 64 / 128 / 256 B, then the 512 MB stream):")
for k, cs in acc.items():
    for c, v in cs.items():
        print("  %-18s %-24s %s" % (k, c, " ".join("%.0f" % x for x in v)))
PYEOF
               cat "$OUT/${RND}_gather_calibration.txt"; rm -rf "$OUT/gcal_fetch" "$OUT/gcal_req" ;;
    contention) python -c "from cuopt_amd import synthetic; synthetic.generate(30000, 24000, 10, seed=21)"
               for k in 4 8; do for i in 1 2 3 4 5 6; do timeout 150 python scripts/r05_contention_probe.py $k ${arg:-3} > "$OUT/contention_k${k}_$i.log" 2>&1 & done; wait; done
               cat "$OUT"/contention_k*.log > "$OUT/${RND}_batch_contention.txt"; grep -c " round " "$OUT/${RND}_batch_contention.txt"; grep -v "batch equals singles \[True\(, True\)*\]" "$OUT/${RND}_batch_contention.txt" | cut -c1-300 ;;
    batchprobe) timeout 900 python scripts/r05_batch_probe.py $(echo "$arg" | tr ',' ' ') > "$OUT/batchprobe.log" 2>&1; grep -E "RATE|Traceback|Error|assert" "$OUT/batchprobe.log" | cut -c1-300 ;;
    batchprof) R=$PWD; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/batchprof" -- env BATCH_PROBE_GRAPH=0 BATCH_PROBE_BASE=${BATCH_PROBE_BASE:-c3} python "$R/scripts/r05_batch_probe.py" $arg > "$R/$OUT/batchprof.log" 2>&1)
          F=$(find "$OUT/batchprof" -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" "$OUT/${RND}_batch_${arg}_kernel_stats.csv" && head -n 12 "$F" | cut -c1-160; rm -rf "$OUT/batchprof" ;;
    batchspmv) /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/batch_spmv_probe.hip -o /tmp/batch_spmv_probe > "$OUT/batch_spmv_build.log" 2>&1
               for a in $(echo "$arg" | tr ',' ' '); do timeout 300 /tmp/batch_spmv_probe $a >> "$OUT/${RND}_batch_spmv_probe.txt" 2>&1; done; cat "$OUT/${RND}_batch_spmv_probe.txt" ;;
    batchspmvpmc) /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/batch_spmv_probe.hip -o /tmp/batch_spmv_probe > "$OUT/batch_spmv_build.log" 2>&1
               R=$PWD; i=0
               for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
                 i=$((i+1))
                 (cd /tmp && timeout -k 5 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$R/$OUT/bsp_pmc_$i" -- /tmp/batch_spmv_probe $arg > "$R/$OUT/bsp_pmc_$i.log" 2>&1)
               done
               python - "$OUT" > "$OUT/${RND}_batch_spmv_pmc.txt" <<'PYEOF'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("%s/bsp_pmc_*/*/*_counter_collection.csv" % out):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-32s mean %.4g  (first %.4g, n %d)" % (c, sum(v) / len(v), v[0], len(v)))
PYEOF
               cat "$OUT/${RND}_batch_spmv_pmc.txt" | cut -c1-150; rm -rf "$OUT"/bsp_pmc_? ;;
    cell) # the sorted-cell probe (round 6, second design): cell:<workload>[:<NP S E PE PG atomic dbg reps>]  -- sweep when no configuration is given
          W=${arg%%:*}; CFG=""; [ "$W" != "$arg" ] && CFG=${arg#*:}
          [ -d /tmp/csr_$W ] || python scripts/dump_csr.py "$W" /tmp/csr_$W > "$OUT/cell_dump_$W.log" 2>&1
          /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/sorted_cell_probe.hip -o /tmp/sorted_cell_probe > "$OUT/cell_build.log" 2>&1
          timeout 900 /tmp/sorted_cell_probe /tmp/csr_$W $CFG >> "$OUT/${RND}_cell_probe_$W.txt" 2>&1; tail -n 40 "$OUT/${RND}_cell_probe_$W.txt" | cut -c1-260 ;;
    cellpmc) W=${arg%%:*}; CFG=${arg#*:}; R=$PWD; i=0
          for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
            i=$((i+1))
            (cd /tmp && timeout -k 5 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$R/$OUT/cell_pmc_$i" -- /tmp/sorted_cell_probe /tmp/csr_$W $CFG > "$R/$OUT/cell_pmc_$i.log" 2>&1)
          done
          python - "$OUT" > "$OUT/${RND}_cell_probe_${W}_pmc.txt" <<'PYEOF'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("%s/cell_pmc_*/*/*_counter_collection.csv" % out):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-32s mean %.4g  (first %.4g, n %d)" % (c, sum(v) / len(v), v[0], len(v)))
PYEOF
          cat "$OUT/${RND}_cell_probe_${W}_pmc.txt" | cut -c1-150; rm -rf "$OUT"/cell_pmc_? ;;
    tall) # the tall-panel probe (round 6): tall:<workload>[:<NP S SW LW CW D reps>]  -- sweep when no configuration is given
          W=${arg%%:*}; CFG=""; [ "$W" != "$arg" ] && CFG=${arg#*:}
          [ -d /tmp/csr_$W ] || python scripts/dump_csr.py "$W" /tmp/csr_$W > "$OUT/tall_dump_$W.log" 2>&1
          /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/tall_panel_probe.hip -o /tmp/tall_panel_probe > "$OUT/tall_build.log" 2>&1
          timeout 900 /tmp/tall_panel_probe /tmp/csr_$W $CFG >> "$OUT/${RND}_tall_probe_$W.txt" 2>&1; tail -n 40 "$OUT/${RND}_tall_probe_$W.txt" | cut -c1-260 ;;
    tallprof) # rocprofv3 kernel stats of ONE configuration: tallprof:<workload>:<NP S SW LW CW D reps>
          W=${arg%%:*}; CFG=${arg#*:}; R=$PWD
          (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/tallprof" -- /tmp/tall_panel_probe /tmp/csr_$W $CFG > "$R/$OUT/tallprof.log" 2>&1)
          F=$(find "$OUT/tallprof" -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" "$OUT/${RND}_tall_probe_${W}_kernel_stats.csv" && head -n 8 "$F" | cut -c1-200; rm -rf "$OUT/tallprof" ;;
    tallpmc) W=${arg%%:*}; CFG=${arg#*:}; R=$PWD; i=0
          for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_REQ_sum"; do
            i=$((i+1))
            (cd /tmp && timeout -k 5 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$R/$OUT/tall_pmc_$i" -- /tmp/tall_panel_probe /tmp/csr_$W $CFG > "$R/$OUT/tall_pmc_$i.log" 2>&1)
          done
          python - "$OUT" > "$OUT/${RND}_tall_probe_${W}_pmc.txt" <<'PYEOF'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("%s/tall_pmc_*/*/*_counter_collection.csv" % out):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-32s mean %.4g  (first %.4g, n %d)" % (c, sum(v) / len(v), v[0], len(v)))
PYEOF
          cat "$OUT/${RND}_tall_probe_${W}_pmc.txt" | cut -c1-150; rm -rf "$OUT"/tall_pmc_? ;;
    soak) # the contention soak (round 6): soak:<rounds>  -- seven processes, one per layout, each comparing with the oracle; then the WHOLE
          # gpu suite under pytest -n 4, <rounds> times
          : > "$OUT/${RND}_contention.txt"
          for L in panel panel_seg jag pb pb_wide stream resident; do timeout 900 python scripts/contention_layouts.py $L ${arg:-3} > "$OUT/soak_$L.log" 2>&1 & done; wait
          { echo "== seven processes side by side, one per layout (scripts/contention_layouts.py), ${arg:-3} rounds each"; cat "$OUT"/soak_*.log | grep -E "round|Error|Traceback|assert"; } >> "$OUT/${RND}_contention.txt"
          for i in $(seq 1 ${arg:-3}); do
            timeout 2400 python -m pytest tests -q -m gpu -n 4 -p no:cacheprovider > "$OUT/soak_suite_$i.log" 2>&1
            { echo "== the whole -m gpu suite under pytest -n 4, run $i"; tail -n 6 "$OUT/soak_suite_$i.log"; grep -E "^(FAILED|ERROR)" "$OUT/soak_suite_$i.log"; } >> "$OUT/${RND}_contention.txt"
          done
          echo "mismatches: $(grep -c MISMATCH "$OUT/${RND}_contention.txt")" >> "$OUT/${RND}_contention.txt"
          cat "$OUT/${RND}_contention.txt" | cut -c1-250 ;;
    *) echo "unknown step $step" ;;
  esac
done
