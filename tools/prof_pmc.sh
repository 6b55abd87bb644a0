#!/bin/bash
# rocprofv3 over one command on the GPU box: kernel trace + stats, then PMC passes (each in its own run, never with a trace domain
# other than --kernel-trace).  The program itself follows `--` (python3 ...), never a shell or env wrapper.
#   tools/prof_pmc.sh <outdir under gpurun_out> "<pmc pass 1>;<pmc pass 2>;..." python3 <script> [args]
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/$1; PASSES=$2; shift 2
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- "$@" > $OUT/run.out 2> $OUT/run.err || { echo "trace pass failed"; tail -5 $OUT/run.err; exit 1; }
IFS=';' read -ra PP <<< "$PASSES"
i=0
for pass in "${PP[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc$i -o run -- "$@" > $OUT/pmc$i.out 2> $OUT/pmc$i.err || echo "pmc pass $i failed"
done
find $OUT -name "*.csv" | head -30
