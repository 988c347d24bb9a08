set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_eval.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_t6.log 2>&1 || { tail -40 gpurun_out/r2_t6.log; exit 1; }
tail -2 gpurun_out/r2_t6.log
A="--users 262144 --U 262144 --I 2000000 --d 128 --k 100 --reps 2"
A2="--users 262144 --U 262144 --I 200000 --d 64 --k 50 --reps 2"
for m in random biased; do
  echo "== c5 $m"; timeout -k 10 120 python tools/eval_probe.py $A --model $m | tail -1
  echo "== c5 f32 $m"; timeout -k 10 120 python tools/eval_probe.py $A --model $m --mode f32 --reps 1 | tail -1
  echo "== c2 $m"; timeout -k 10 120 python tools/eval_probe.py $A2 --model $m | tail -1
  echo "== c2 f32 $m"; timeout -k 10 120 python tools/eval_probe.py $A2 --model $m --mode f32 | tail -1
done
for v in ch64 ch128; do
TFREC_AMD_LIB=$R/var_libs/$v/libtfrec_amd.so timeout -k 10 300 python bench.py --no-eval --no-other --no-cpu-baseline > gpurun_out/r2_bench6_$v.json 2>/dev/null
python - $v <<'PY'
import json, sys
o = json.loads(open("gpurun_out/r2_bench6_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(o["ms_per_step"], 3), "ms frac", round(o["roofline"]["frac"], 3), {k: round(v, 3) for k, v in o["roofline"]["phases_ms"].items()})
PY
done
