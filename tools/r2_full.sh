# Developer script (GPU box, repository root): whole GPU suite, bench, two-rank bench rehearsal.
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_gputests.log 2>&1 || { tail -40 gpurun_out/r2_gputests.log; exit 1; }
tail -3 gpurun_out/r2_gputests.log
timeout -k 10 600 python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err || { tail -20 gpurun_out/r2_bench.err; exit 1; }
echo bench done
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --single-device --steps 5 --warmup 2 --no-other --no-cpu-baseline --eval-users 32768 > gpurun_out/r2_bench_2r.json 2> gpurun_out/r2_bench_2r.err || { tail -20 gpurun_out/r2_bench_2r.err; exit 1; }
echo two-rank bench done
python - <<'PY'
import json
o = json.loads(open("gpurun_out/r2_bench.json").read().strip().splitlines()[-1])
print("BPR", round(o["value"] / 1e9, 3), "G/s", round(o["ms_per_step"], 3), "ms frac", round(o["roofline"]["frac"], 3), {k: round(v, 3) for k, v in o["roofline"]["phases_ms"].items()})
e = o["eval"]
for shape in (e, e["c5"]):
    for case in ("independent", "popularity_biased"):
        print(shape["shape"][:8], case, {m: (round(shape[case][m]["ms"], 2), round(shape[case][m]["roofline"]["frac"], 3)) for m in ("f32", "bf16")})
print({k: (round(v.get("ms_per_step", v.get("ms", v.get("ms_per_epoch", v.get("us_per_step", 0) / 1e3))), 3), round(v.get("roofline", {}).get("frac", 0), 3)) for k, v in o["other_paths"].items()})
print(o["cpu_baseline"]["value"], o["cpu_baseline"]["cores"], o["eval"]["cpu_baseline"]["value"])
t = json.loads(open("gpurun_out/r2_bench_2r.json").read().strip().splitlines()[-1])
print("2 ranks (one GPU, gloo):", t["n_gpus"], round(t["ms_per_step"], 2), "ms")
PY
