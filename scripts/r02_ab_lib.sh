# A/B of library builds: bash scripts/r02_ab_lib.sh "<lib suffixes>" "<workloads>"   (suffix "" = the default library)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02/ab; mkdir -p $O
for W in $2; do
  for L in $1; do
    LIB=$GRAFT_REPO_ROOT/cuopt_amd/lib/libcuopt$L.so
    [ "$L" = "default" ] && LIB=$GRAFT_REPO_ROOT/cuopt_amd/lib/libcuopt.so
    CUOPT_AMD_LIB=$LIB timeout 150 python bench.py --workload $W --steps 400 --warmup 100 --no-cpu-baseline > $O/$W$L.json 2> $O/$W$L.err || tail -3 $O/$W$L.err
    python - <<PY
import json
d=json.load(open("$O/$W$L.json")); r=d["roofline"]
print("$W", "$L", "it/s", d["value"], {k: round(v*1e3,1) for k,v in r["per_kernel_ms"].items()})
PY
  done
done
