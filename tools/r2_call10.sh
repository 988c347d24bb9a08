set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_train.py tests/test_gpu_utable.py tests/test_gpu_models.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_t10.log 2>&1 || { tail -40 gpurun_out/r2_t10.log; exit 1; }
tail -2 gpurun_out/r2_t10.log
timeout -k 10 300 python bench.py --no-eval --no-other --no-cpu-baseline --steps 100 > gpurun_out/r2_bench10.json 2>/dev/null
python - <<'PY'
import json
o = json.loads(open("gpurun_out/r2_bench10.json").read().strip().splitlines()[-1])
print(round(o["ms_per_step"], 3), "ms frac", round(o["roofline"]["frac"], 3), {k: round(v, 3) for k, v in o["roofline"]["phases_ms"].items()}, o.get("reference_batch_512"))
PY
timeout -k 10 300 python tools/perf_probe.py --U 10000000 --I 2000000 --d 128 --batches 1048576,4194304 --versioned 2>&1 | grep "G triples"
timeout -k 10 300 python tools/perf_probe.py --U 10000000 --I 2000000 --d 128 --batches 4194304 2>&1 | grep "G triples"
timeout -k 10 300 python tools/perf_probe.py --batches 1048576,16777216 --versioned 2>&1 | grep "G triples"
