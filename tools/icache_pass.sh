# Developer script (GPU box, repository root): instruction-cache counters of the scoring kernel at the 2M-item shape.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
EA="--users 262144 --U 262144 --I 2000000 --d 128 --k 100 --model ${MODEL:-random} --mode bf16 --reps 1"
timeout -s KILL 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/r3_pmc_icache -- python3 $R/tools/eval_probe.py $EA > $R/gpurun_out/r3_pmc_icache.log 2>&1
echo "rc=$?"
tail -3 $R/gpurun_out/r3_pmc_icache.log
