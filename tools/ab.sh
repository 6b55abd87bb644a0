#!/bin/bash
# A/B of two library builds on the GPU box: tools/ab.sh [libB.so]   (A = the in-tree default)
set -o pipefail
B=${1:-drone-sim-python_amd/lib/libd2dhip_b.so}
run() {
  echo "== $1 (${D2D_LIB:-default})"
  D2D_LM_STAMPS=1 timeout -k 10 200 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --check-every 200 2>&1 >/dev/null | grep stamps | head -1
  timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$1.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/bench_$1.json'));r=d['roofline'];print(round(d['value']), 'fits/s ms/step', round(d['ms_per_step'],2), 'frac', round(r['frac'],4), 'conv', d['converged_frac'], 'iso', round(d['roofline_isolated']['frac'],3), d['mean_iters'])"
}
run A
[ -f "$B" ] && D2D_LIB=$PWD/$B run B
