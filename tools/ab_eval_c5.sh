#!/bin/bash
# C5 random only
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  echo "$name c5 random: $(TFREC_AMD_LIB=$lib timeout -k 10 200 python tools/eval_probe.py --users 262144 --I 2000000 --d 128 --k 100 --model random --mode bf16 2>&1 | grep -E '^eval|Error|error' | tail -1)"
done
