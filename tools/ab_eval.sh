#!/bin/bash
# Developer script: the scoring kernel of several builds of the library on one box (C5 shape and C2 shape, independent
# and biased models).  Usage: tools/ab_eval.sh name=lib.so ...
for rep in 0 1; do
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  for model in random biased; do
    echo "[round $rep] $name c5 $model: $(TFREC_AMD_LIB=$lib timeout -k 10 200 python tools/eval_probe.py --users 262144 --I 2000000 --d 128 --k 100 --model $model --mode bf16 2>&1 | grep -E '^eval|per wave|Error|error' | tail -2 | tr '\n' ' ')"
  done
  echo "[round $rep] $name c2 random: $(TFREC_AMD_LIB=$lib timeout -k 10 200 python tools/eval_probe.py --users 262144 --I 200000 --d 64 --k 50 --model random --mode bf16 2>&1 | grep -E '^eval|Error|error' | tail -1)"
done
done
