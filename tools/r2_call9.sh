set -e
R=$GRAFT_REPO_ROOT
cd $R
for v in du2; do
  if [ $v = base ]; then L=$R/tfrec_amd/lib/libtfrec_amd.so; else L=$R/var_libs/$v/libtfrec_amd.so; fi
  TFREC_AMD_LIB=$L timeout -k 10 300 python bench.py --no-eval --no-other --no-cpu-baseline --no-extras --steps 100 > gpurun_out/r2_bench9_$v.json 2>/dev/null
  python - $v <<'PY'
import json, sys
o = json.loads(open("gpurun_out/r2_bench9_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(o["ms_per_step"], 3), "ms frac", round(o["roofline"]["frac"], 3), {k: round(v, 3) for k, v in o["roofline"]["phases_ms"].items()})
PY
done
