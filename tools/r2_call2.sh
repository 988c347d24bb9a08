# Developer script (GPU box, repository root): tests, bench, two-rank bench rehearsal, step probes.
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_gputests.log 2>&1 || { tail -40 gpurun_out/r2_gputests.log; exit 1; }
tail -3 gpurun_out/r2_gputests.log
timeout -k 10 600 python bench.py > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err || { tail -20 gpurun_out/r2_bench1.err; exit 1; }
echo bench done
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --single-device --steps 5 --warmup 2 --no-other --no-cpu-baseline --eval-users 32768 > gpurun_out/r2_bench_2r.json 2> gpurun_out/r2_bench_2r.err || { tail -20 gpurun_out/r2_bench_2r.err; exit 1; }
echo two-rank bench done
timeout -k 10 200 python tools/perf_probe.py --batches 1048576,4194304 > gpurun_out/r2_probe_base.log 2>&1
TFREC_AMD_LIB=$R/var_libs/nostage/libtfrec_amd.so timeout -k 10 200 python tools/perf_probe.py --batches 1048576,4194304 > gpurun_out/r2_probe_nostage.log 2>&1
timeout -k 10 200 python tools/perf_probe.py --U 10000000 --I 2000000 --d 128 --batches 4194304 > gpurun_out/r2_probe_base_c4.log 2>&1
TFREC_AMD_LIB=$R/var_libs/nostage/libtfrec_amd.so timeout -k 10 200 python tools/perf_probe.py --U 10000000 --I 2000000 --d 128 --batches 4194304 > gpurun_out/r2_probe_nostage_c4.log 2>&1
echo probes done
