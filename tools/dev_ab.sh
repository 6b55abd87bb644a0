#!/bin/bash
# Development A/B on the GPU box: the fit tests on the new library, then tools/dev_k50.py on the baseline and the new library.
# usage (through gpurun): bash tools/dev_ab.sh <tag>
tag=${1:-ab}
out=gpurun_out/r4
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_fit.py -x -q -m gpu > $out/${tag}_fit_tests.log 2>&1
echo "fit tests rc=$?" | tee -a $out/${tag}_fit_tests.log
tail -3 $out/${tag}_fit_tests.log
if [ -f drone-sim-python_amd/lib/libd2dhip_base.so ]; then
  D2D_LIB=$PWD/drone-sim-python_amd/lib/libd2dhip_base.so timeout -k 10 300 python tools/dev_k50.py > $out/${tag}_k50_base.log 2>&1
  echo "--- base"; cat $out/${tag}_k50_base.log
fi
timeout -k 10 300 python tools/dev_k50.py > $out/${tag}_k50_new.log 2>&1
echo "--- new"; cat $out/${tag}_k50_new.log
