# Developer script (GPU box, repository root): tests, bench, two-rank bench rehearsal.
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r2_noalloc.log 2>&1 || { tail -30 gpurun_out/r2_noalloc.log; exit 1; }
tail -3 gpurun_out/r2_noalloc.log
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_gputests.log 2>&1 || { tail -40 gpurun_out/r2_gputests.log; exit 1; }
tail -3 gpurun_out/r2_gputests.log
timeout -k 10 600 python bench.py > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err || { tail -20 gpurun_out/r2_bench1.err; exit 1; }
echo bench done
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --single-device --steps 5 --warmup 2 --no-other --no-cpu-baseline --eval-users 32768 > gpurun_out/r2_bench_2r.json 2> gpurun_out/r2_bench_2r.err || { tail -20 gpurun_out/r2_bench_2r.err; exit 1; }
echo two-rank bench done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 60 rocprofv3 -L > $R/gpurun_out/r2_counters.txt 2>&1 || true
echo ok
