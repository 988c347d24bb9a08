#!/bin/bash
# C2 shape only (200k items, d=64), independent and biased factors, k given by K (default 50)
K=${K:-50}
for rep in 0 1; do
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  for model in random biased; do
    echo "[round $rep] $name c2k$K $model: $(TFREC_AMD_LIB=$lib timeout -k 10 200 python tools/eval_probe.py --users 262144 --I 200000 --d 64 --k $K --model $model --mode bf16 2>&1 | grep -E '^eval|Error|error' | tail -1)"
  done
done
done
