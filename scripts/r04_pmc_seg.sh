# round 4: PMC passes, c3 through the row-per-lane panels (panel_seg=0) and the long-tail panels (panel_seg=1): where do the 5 us go?
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_pmc_seg; mkdir -p $O
export TMPDIR=/tmp
for V in 0 1; do
  i=0
  for C in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_LDS" "TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_NC_READ_REQ_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum"; do
    i=$((i+1))
    (cd /tmp && CUOPT_AMD_TUNE=panel_seg=$V timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/v${V}_pmc_$i -- python $GRAFT_REPO_ROOT/scripts/r04_x1.py '[["x","c3",{}]]' > $O/v${V}_pmc_$i.log 2>&1)
  done
  python scripts/pmc_summary.py $O/r04_pmc_c3_seg$V.json $O/v${V}_pmc_1 $O/v${V}_pmc_2 $O/v${V}_pmc_3 $O/v${V}_pmc_4 $O/v${V}_pmc_5 > $O/r04_pmc_c3_seg${V}_summary.txt
  grep -E "^k_panel_plain|^k_panel_a_dual" $O/r04_pmc_c3_seg${V}_summary.txt
done
tail -3 $O/v1_pmc_4.log $O/v1_pmc_5.log
# the RCCL capture probe, bundled RCCL of PyTorch and ROCm's own
TL=$(python -c "import torch,os; print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
hipcc -O1 --offload-arch=gfx950 tools/rccl_capture_repro.cpp -o /tmp/rccl_capture_repro -ldl 2>/dev/null
for M in 0 1 2; do echo "== torch rccl, mode $M"; LD_LIBRARY_PATH=$TL:$LD_LIBRARY_PATH timeout 60 /tmp/rccl_capture_repro $M 2>&1 | tail -6; echo "exit $?"; done 2>&1 | tee $O/rccl_capture.txt
for M in 1 2; do echo "== rocm rccl, mode $M"; LD_LIBRARY_PATH=/opt/rocm/lib timeout 60 /tmp/rccl_capture_repro $M 2>&1 | tail -6; echo "exit $?"; done 2>&1 | tee -a $O/rccl_capture.txt
