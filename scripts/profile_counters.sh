#!/bin/bash
# Matrix-core and LDS counters of the convolution kernels, with the known-peak yardstick next to
# them: one rocprofv3 counter run over the yardstick (scripts/microbench/mfma_peak, built with the
# hipcc line in its source) and one over every layer of ResNet-50 bs256 launched alone
# (scripts/bench_layers.py).  Prints the tables of profiles/conv_pmc_lds_current.md.
set -e
cd "$(dirname "$0")/.."
OUT=${OUT:-gpurun_out/profile_counters}
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=${TMPDIR:-/tmp}
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS"
echo "== yardstick: nothing but MFMAs, four waves per SIMD =="
./scripts/microbench/mfma_peak 20000
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/peak" -- ./scripts/microbench/mfma_peak 20000 > "$OUT/peak.log" 2>&1
python scripts/summarize_profile.py pmc "$OUT/peak"
echo
echo "== every convolution launch of ResNet-50 bs256, alone =="
timeout -s KILL 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/layers" -- python scripts/bench_layers.py resnet50 256 > "$OUT/layers.log" 2>&1
python scripts/summarize_profile.py pmc "$OUT/layers"
rm -rf "$OUT/peak" "$OUT/layers"
