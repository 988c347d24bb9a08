set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_eval.py tests/test_gpu_models.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_t5.log 2>&1 || { tail -40 gpurun_out/r2_t5.log; exit 1; }
tail -2 gpurun_out/r2_t5.log
A="--users 262144 --U 262144 --I 2000000 --d 128 --k 100 --reps 2"
A2="--users 262144 --U 262144 --I 200000 --d 64 --k 50 --reps 2"
for m in random biased; do
  echo "== shipped c5 $m"; timeout -k 10 120 python tools/eval_probe.py $A --model $m | tail -1
  echo "== ti3 c5 $m"; TFREC_AMD_LIB=$R/var_libs/ti3/libtfrec_amd.so timeout -k 10 120 python tools/eval_probe.py $A --model $m | tail -1
  echo "== shipped c2 $m"; timeout -k 10 120 python tools/eval_probe.py $A2 --model $m | tail -1
  echo "== ti3 c2 $m"; TFREC_AMD_LIB=$R/var_libs/ti3/libtfrec_amd.so timeout -k 10 120 python tools/eval_probe.py $A2 --model $m | tail -1
done
echo "== prof c5 random"; TFREC_AMD_LIB=$R/var_libs/eprof/libtfrec_amd.so timeout -k 10 120 python tools/eval_probe.py $A --model random | tail -2
echo "== ch64 step"; TFREC_AMD_LIB=$R/var_libs/ch64/libtfrec_amd.so timeout -k 10 300 python bench.py --no-eval --no-other --no-cpu-baseline > gpurun_out/r2_bench5_ch64.json 2>/dev/null
python - <<'PY'
import json
o = json.loads(open("gpurun_out/r2_bench5_ch64.json").read().strip().splitlines()[-1])
print("ch64", round(o["ms_per_step"], 3), "ms frac", round(o["roofline"]["frac"], 3), {k: round(v, 3) for k, v in o["roofline"]["phases_ms"].items()})
PY
