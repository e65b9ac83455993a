#!/bin/bash
# Serial kernel tables of the ResNet-50 step with the backward BatchNorm reductions in the
# data-gradient epilogues for every layer (A) and for small tensors only (B, the default).
set -e
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/bstats_ab
rm -rf $OUT; mkdir -p $OUT
T=${T:-gpurun_out/t2.json}
DTF_AUTOTUNE_CACHE=$T DTF_BWD_STATS_SMALL_BYTES=1099511627776 DTF_WGRAD_STREAM=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/a -- python bench.py --steps 10 --warmup 2 --no_graph > $OUT/a.log 2>&1
python scripts/summarize_profile.py stats $OUT/a 20 > $OUT/a.md
DTF_AUTOTUNE_CACHE=$T DTF_WGRAD_STREAM=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b -- python bench.py --steps 10 --warmup 2 --no_graph > $OUT/b.log 2>&1
python scripts/summarize_profile.py stats $OUT/b 20 > $OUT/b.md
rm -rf $OUT/a $OUT/b
