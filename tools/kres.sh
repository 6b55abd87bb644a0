#!/bin/bash
# Resource usage (VGPRs, scratch, occupancy) of every kernel of one translation unit of csrc/, as hipcc reports it.
# usage: tools/kres.sh fit_kernels.hip [grep pattern]
cd "$(dirname "$0")/../drone-sim-python_amd/csrc" || exit 1
src=${1:-fit_kernels.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -disable-machine-licm -Rpass-analysis=kernel-resource-usage -c "$src" -o /tmp/kres.o 2> /tmp/kres.txt
python3 ../../tools/res_usage.py /tmp/kres.txt | grep -E "${2:-.}"
