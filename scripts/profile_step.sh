#!/bin/bash
# Reproduces the tables under profiles/ for the flagship step (ResNet-50 bs256 bf16, one MI355X).
# Each rocprofv3 pass is a run of its own (kernel statistics, then one counter group per run);
# raw output goes to $OUT (default gpurun_out/profile_step), the condensed tables are printed.
#
#   sh scripts/profile_step.sh [extra bench.py flags]
set -e
cd "$(dirname "$0")/.."
OUT=${OUT:-gpurun_out/profile_step}
mkdir -p "$OUT"
export TMPDIR=${TMPDIR:-/tmp}

echo "== kernel statistics, single lane (serial kernel durations) =="
DTF_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- \
    python bench.py --steps 10 --warmup 2 --no_graph "$@" > "$OUT/stats.log" 2>&1
python scripts/summarize_profile.py stats "$OUT/stats" 20

echo "== MFMA / LDS counters of the convolution kernels =="
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 \
    SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS \
    --output-format csv -d "$OUT/pmc_sq" -- python scripts/bench_layers.py resnet50 256 > "$OUT/pmc_sq.log" 2>&1
python scripts/summarize_profile.py pmc "$OUT/pmc_sq"

for c in FETCH_SIZE WRITE_SIZE; do
  echo "== L2 <-> memory traffic: $c (KB, summed per kernel) =="
  DTF_WGRAD_STREAM=0 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -- \
      python bench.py --steps 2 --warmup 0 --no_graph "$@" > "$OUT/pmc_$c.log" 2>&1
  python scripts/summarize_profile.py pmc "$OUT/pmc_$c"
done
