# Developer script (run on the GPU box from the repository root): kernel stats and the two counter
# passes of the bench workload, outputs under gpurun_out/.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $R/gpurun_out/bench.json 2> $R/gpurun_out/bench.err
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_stats.log 2>&1
echo "stats done"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-eval --no-cpu-baseline --no-other > $R/gpurun_out/pmc_fetch.log 2>&1
echo "fetch done"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-eval --no-cpu-baseline --no-other > $R/gpurun_out/pmc_write.log 2>&1
echo "write done"
