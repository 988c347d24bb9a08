# Developer script (GPU box, repository root): counter passes over the scoring kernel.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
ARGS="${EVAL_ARGS:---model biased --mode bf16 --reps 1}"
timeout -k 10 120 python3 $R/tools/eval_probe.py $ARGS --reps 3
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/epmc1 -- python3 $R/tools/eval_probe.py $ARGS > $R/gpurun_out/epmc1.log 2>&1
echo pass1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/epmc2 -- python3 $R/tools/eval_probe.py $ARGS > $R/gpurun_out/epmc2.log 2>&1
echo pass2
