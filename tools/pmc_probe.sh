set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name counters
  timeout -s KILL 150 rocprofv3 --pmc $2 --kernel-include-regex "seg_tile" --output-format csv -d $R/gpurun_out/pmc_$1 -- python3 $R/tools/perf_probe.py --batches 1048576 --steps 3 --plain-only > $R/gpurun_out/pmc_$1.log 2>&1
}
run sq1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES"
run tcc1 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
run tcc2 "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_32B_sum"
run grbm "GRBM_GUI_ACTIVE GRBM_COUNT"
