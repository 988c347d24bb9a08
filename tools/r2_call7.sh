set -e
R=$GRAFT_REPO_ROOT
cd $R
A="--users 262144 --U 262144 --I 2000000 --d 128 --k 100 --reps 2"
for m in random biased; do
  echo "== shipped c5 $m"; timeout -k 10 120 python tools/eval_probe.py $A --model $m | tail -1
  echo "== uw2nw8 c5 $m"; TFREC_AMD_LIB=$R/var_libs/uw2nw8/libtfrec_amd.so timeout -k 10 120 python tools/eval_probe.py $A --model $m | tail -1
done
TFREC_AMD_LIB=$R/var_libs/uw2nw8/libtfrec_amd.so timeout -k 10 300 python -u -m pytest tests/test_gpu_eval.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -2
