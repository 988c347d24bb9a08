# Developer script (GPU box): ablation ladder of the scoring kernel at the 2M-item, d=128, k=100 shape.
set -e
R=$GRAFT_REPO_ROOT
cd $R
A="--users 262144 --U 262144 --I 2000000 --d 128 --k 100 --reps 2"
for m in random biased; do
  echo "== shipped $m"; timeout -k 10 120 python tools/eval_probe.py $A --model $m | tail -1
  for v in abl1 abl2 abl3 abl5; do
    echo "== $v $m"; TFREC_AMD_LIB=$R/var_libs/$v/libtfrec_amd.so timeout -k 10 120 python tools/eval_probe.py $A --model $m | tail -1
  done
  echo "== prof $m"; TFREC_AMD_LIB=$R/var_libs/prof/libtfrec_amd.so timeout -k 10 120 python tools/eval_probe.py $A --model $m | tail -2
done
