set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 150 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $R/gpurun_out/cal1 -- python3 $R/tools/pmc_calibrate.py > $R/gpurun_out/cal1.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/cal2 -- python3 $R/tools/pmc_calibrate.py > $R/gpurun_out/cal2.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/cal3 -- python3 $R/tools/pmc_calibrate.py > $R/gpurun_out/cal3.log 2>&1
