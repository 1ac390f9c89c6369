# round 3: the gather-free SpMV at ten times the size of C3 (1e7 x 1e7, 1e8 nnz) + PMC passes at C3 size
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03/pb
mkdir -p $O
B=$GRAFT_REPO_ROOT/tools/bin/spmv_pb
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc$i -- $B 1000000 10 5 prof 1 > $O/pmc$i.log 2>&1
  tail -1 $O/pmc$i.log
done
timeout 1500 $B 10000000 10 10 pr:1,10,6 > $O/big.txt 2>&1
tail -8 $O/big.txt
