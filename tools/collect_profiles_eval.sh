# Developer script (GPU box, repository root): only the three counter passes over the scoring kernel of tools/collect_profiles.sh
# (for a change that touches the scoring kernel alone).
set -e
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r3}
cd /tmp && export TMPDIR=/tmp
EA="--users 262144 --U 262144 --I 2000000 --d 128 --k 100 --model random --mode bf16 --reps 1"
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/${TAG}_pmc_eval1 -- python3 $R/tools/eval_probe.py $EA > $R/gpurun_out/${TAG}_pmc_eval1.log 2>&1
echo "eval pass 1 done"
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${TAG}_pmc_eval2 -- python3 $R/tools/eval_probe.py $EA > $R/gpurun_out/${TAG}_pmc_eval2.log 2>&1
echo "eval pass 2 done"
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/${TAG}_pmc_eval3 -- python3 $R/tools/eval_probe.py $EA > $R/gpurun_out/${TAG}_pmc_eval3.log 2>&1 || echo "eval pass 3 failed (counter names)"
echo "eval pass 3 done"
