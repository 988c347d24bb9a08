#!/bin/bash
# Developer script: like tools/ab_eval.sh, but only the independent-factor models (C5 and C2 shapes).
for rep in 0 1; do
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  echo "[round $rep] $name c5 random: $(TFREC_AMD_LIB=$lib timeout -k 10 200 python tools/eval_probe.py --users 262144 --I 2000000 --d 128 --k 100 --model random --mode bf16 2>&1 | grep -E '^eval|per wave|Error|error' | tail -2 | tr '\n' ' ')"
  echo "[round $rep] $name c2 random: $(TFREC_AMD_LIB=$lib timeout -k 10 200 python tools/eval_probe.py --users 262144 --I 200000 --d 64 --k 50 --model random --mode bf16 2>&1 | grep -E '^eval|Error|error' | tail -1)"
done
done
