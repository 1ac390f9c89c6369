for sb in 1000000 1200000 1398102 1600000 2000000; do for pn in 40000 60000 80000; do
  CUOPT_AMD_TUNE=slab_bytes=$sb,panel_nnz=$pn python bench.py --no-cpu-baseline --no-convergence-run --min-seconds 1.5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slab_bytes $sb panel_nnz $pn: %.1f it/s  a_dual %.1f us at_step %.1f us  wg %s slabs %s' % (d['value'], 1e3*d['roofline']['per_kernel_ms']['SPMV_A_DUAL'], 1e3*d['roofline']['per_kernel_ms']['SPMV_AT_STEP'], d['spmv_layout']['A']['workgroups'], d['spmv_layout']['A'].get('slabs')))"
done; done
