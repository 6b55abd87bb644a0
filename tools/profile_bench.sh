#!/bin/bash
# rocprofv3 passes over the headline bench (GPU box): kernel trace + stats, then PMC passes (each in its own run).
#   tools/profile_bench.sh <outdir under gpurun_out>
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/${1:-prof_bench}
mkdir -p $OUT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || exit 1
tail -1 $OUT/bench.json | cut -c1-300
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$tag -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sim --no-nlp --no-groups --config3-batch 0 --no-extra-modes > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err || echo "pass $tag failed"
  echo "pass $tag done"
done
# the simulation kernels (BASELINE configs[4]) and the collocation kernel: HBM traffic passes only
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_sim_$pass -o bench -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-groups --config3-batch 0 --no-extra-modes > $OUT/pmc_sim_$pass.json 2> $OUT/pmc_sim_$pass.err || echo "pass sim $pass failed"
  echo "pass sim $pass done"
done
# executed fp64 flop of the simulation and collocation kernels (fp64-VALU bound): one pass
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sim_fp64 -o bench -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-groups --config3-batch 0 --no-extra-modes > $OUT/pmc_sim_fp64.json 2> $OUT/pmc_sim_fp64.err || echo "pass sim fp64 failed"
echo "pass sim fp64 done"
ls -R $OUT | head -40
# the headline alone (4096 fits, default solver, index order): every launch of fit_lm_kernel in this trace is a headline launch, so
# the AverageNs of its kernel_stats row is the per-launch duration that bench.py's HIP events must agree with
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_headline -o bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sim --no-nlp --no-groups --config3-batch 0 --no-extra-modes > $OUT/headline.json 2> $OUT/headline.err || echo "headline trace failed"
echo "headline trace done"
