#!/bin/bash
# Development: the 4096-fit headline solve under scheduling switches of the fused kernel (environment, read once per process).
for cfg in "" "D2D_LM_PRIO_AT=24" "D2D_LM_PRIO_AT=32" "D2D_LM_PRIO_AT=40" "D2D_LM_PRIO_AT=64" "D2D_LM_PRIO_AT=1000" "D2D_LM_SLICE=8" "D2D_LM_SLICE=16" "D2D_LM_SLICE=24"; do
  echo "== ${cfg:-default}"
  env $cfg timeout -k 10 200 python tools/dev_k50.py 2>&1 | grep "B=4096 minpack\|B=32768 minpack"
done
