#!/bin/bash
# Development A/B of the long-horizon kernel: long / groups tests on the new library, then tools/dev_seg.py (121 and 301 nodes) on both.
tag=${1:-long}
out=gpurun_out/r4
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_long.py tests/test_gpu_groups.py -x -q -m gpu > $out/${tag}_long_tests.log 2>&1
echo "long+groups tests rc=$?"; tail -3 $out/${tag}_long_tests.log
for lib in base new; do
  L=$PWD/drone-sim-python_amd/lib/libd2dhip.so
  [ $lib = base ] && L=$PWD/drone-sim-python_amd/lib/libd2dhip_base.so
  [ -f $L ] || continue
  D2D_LIB=$L timeout -k 10 400 python tools/dev_seg.py $out/${tag}_seg_$lib.npz 121 301 > $out/${tag}_seg_$lib.log 2>&1
  echo "--- $lib"; grep "K=" $out/${tag}_seg_$lib.log
done
