# Developer script, part A (GPU box, repository root): the un-profiled bench line and the kernel statistics of the same command.
set -e
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r3}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 python3 $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
echo "bench done"
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_prof_stats.log 2>&1
echo "stats done"
