#!/bin/bash
# Development: in-kernel phase stamps of the fused LM kernel (D2D_LM_STAMPS=1) for the baseline and the new library.
tag=${1:-st}
out=gpurun_out/r4
mkdir -p $out
for lib in base new; do
  L=$PWD/drone-sim-python_amd/lib/libd2dhip.so
  [ $lib = base ] && L=$PWD/drone-sim-python_amd/lib/libd2dhip_base.so
  [ -f $L ] || continue
  D2D_LIB=$L D2D_LM_STAMPS=1 timeout -k 10 300 python tools/dev_k50.py > $out/${tag}_stamps_$lib.log 2>&1
  echo "--- $lib"; grep -E "stamps|fits/s" $out/${tag}_stamps_$lib.log | awk '!seen[$0]++' | head -24
done
