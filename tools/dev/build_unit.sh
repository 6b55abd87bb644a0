#!/bin/bash
# usage: tools/dev/build_unit.sh [csrc dir] [out]   -- the unit test / microbenchmark of the solve against a given csrc tree
cd "$(dirname "$0")/../.."
src=${1:-drone-sim-python_amd/csrc}
out=${2:-drone-sim-python_amd/lib/chol_unit}
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -disable-machine-licm -Wno-unused-result -Wno-unused-value -I $src -I include $3 tools/dev/chol_unit.hip -o $out 2>&1 | grep -E "error" -A5 | head -30
ls -la $out
