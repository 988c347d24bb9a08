set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_utable.py tests/test_gpu_sharded.py tests/test_gpu_train.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_t4.log 2>&1 || { tail -40 gpurun_out/r2_t4.log; exit 1; }
tail -3 gpurun_out/r2_t4.log
timeout -k 10 600 python bench.py --no-eval --no-other --no-cpu-baseline > gpurun_out/r2_bench4.json 2> gpurun_out/r2_bench4.err || { tail -20 gpurun_out/r2_bench4.err; exit 1; }
timeout -k 10 600 python bench.py --no-eval --no-other --no-cpu-baseline --plain-table > gpurun_out/r2_bench4p.json 2> gpurun_out/r2_bench4p.err || { tail -20 gpurun_out/r2_bench4p.err; exit 1; }
python - <<'PY'
import json
for f in ("r2_bench4", "r2_bench4p"):
    o = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(o["value"] / 1e9, 3), "G/s", round(o["ms_per_step"], 3), "ms frac", round(o["roofline"]["frac"], 3), {k: round(v, 3) for k, v in o["roofline"]["phases_ms"].items()})
PY
