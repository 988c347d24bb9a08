set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -u -m pytest tests/test_gpu_sharded.py tests/test_gpu_train.py tests/test_gpu_models.py tests/test_gpu_sampler.py tests/test_zz_no_alloc.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_t8.log 2>&1 || { tail -40 gpurun_out/r2_t8.log; exit 1; }
tail -2 gpurun_out/r2_t8.log
timeout -k 10 300 python bench.py --no-eval --no-other --no-cpu-baseline --extras --steps 50 > gpurun_out/r2_bench8.json 2>gpurun_out/r2_bench8.err || { tail gpurun_out/r2_bench8.err; exit 1; }
python - <<'PY'
import json
o = json.loads(open("gpurun_out/r2_bench8.json").read().strip().splitlines()[-1])
print(round(o["ms_per_step"], 3), o.get("reference_batch_512"))
PY
