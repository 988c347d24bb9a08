# Developer script (run on the GPU box from the repository root): kernel stats and the counter passes of the
# bench workload and of the scoring kernel, outputs under gpurun_out/ (TAG names the round, e.g. r2).
set -e
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r3}
cd /tmp && export TMPDIR=/tmp
# (part A, tools/collect_profiles_a.sh, runs the bench and its kernel statistics)



timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-eval --no-cpu-baseline --no-other --no-extras > $R/gpurun_out/${TAG}_pmc_fetch.log 2>&1
echo "fetch done"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-eval --no-cpu-baseline --no-other --no-extras > $R/gpurun_out/${TAG}_pmc_write.log 2>&1
echo "write done"
# scoring kernel at the 2M-item, d=128, k=100 shape, independent factors: matrix-pipe and vector-pipe busy cycles,
# wave cycles, and (own pass) the clock the chip held
EA="--users 262144 --U 262144 --I 2000000 --d 128 --k 100 --model random --mode bf16 --reps 1"
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/${TAG}_pmc_eval1 -- python3 $R/tools/eval_probe.py $EA > $R/gpurun_out/${TAG}_pmc_eval1.log 2>&1
echo "eval pass 1 done"
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${TAG}_pmc_eval2 -- python3 $R/tools/eval_probe.py $EA > $R/gpurun_out/${TAG}_pmc_eval2.log 2>&1
echo "eval pass 2 done"
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/${TAG}_pmc_eval3 -- python3 $R/tools/eval_probe.py $EA > $R/gpurun_out/${TAG}_pmc_eval3.log 2>&1 || echo "eval pass 3 failed (counter names)"
echo "eval pass 3 done"
# the BPR step at the large configuration's table shape (10M users x 2M items, d=128, 4M triples, versioned table):
# kernel durations and the two traffic counters, one pass each
LA="--U 10000000 --I 2000000 --d 128 --batches 4194304 --versioned --steps 6 --plain-only"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_large_stats -- python3 $R/tools/perf_probe.py $LA > $R/gpurun_out/${TAG}_large_stats.log 2>&1
echo "large stats done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_large_fetch -- python3 $R/tools/perf_probe.py $LA > $R/gpurun_out/${TAG}_large_fetch.log 2>&1
echo "large fetch done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_large_write -- python3 $R/tools/perf_probe.py $LA > $R/gpurun_out/${TAG}_large_write.log 2>&1
echo "large write done"
