#!/bin/bash
# Re-measures the backward-statistics fusion (sum g, sum g*y in the data-gradient epilogue):
#   1. drops every "+bstats" entry from a copy of the tile table and tunes those shapes again
#      (all models of the table; ResNet-50 bs256 with every layer fused so each shape is seen);
#   2. times bench.py for a range of size thresholds (DTF_BWD_STATS_SMALL_BYTES).
# Output: gpurun_out/bstats_sweep/{tuned.json, tune.log, sweep.txt}
set -e
cd "$(dirname "$0")/../.."
OUT=gpurun_out/bstats_sweep
mkdir -p $OUT
T=$OUT/tuned.json
python - <<PY
import json
t = json.load(open("distributed_tensorflow_amd/ops/tuned_gfx950.json"))
keep = {k: v for k, v in t.items() if "+bstats" not in k}
json.dump(keep, open("$T", "w"), indent=0, sort_keys=True)
print("dropped", len(t) - len(keep), "of", len(t), "entries")
PY
export DTF_AUTOTUNE_CACHE=$T DTF_AUTOTUNE_OUT=$T
DTF_AUTOTUNE=1 timeout -k 10 500 python scripts/autotune_models.py > $OUT/tune.log 2>&1
DTF_AUTOTUNE=1 DTF_BWD_STATS_SMALL_BYTES=1099511627776 timeout -k 10 300 python scripts/autotune_models.py resnet50:256 >> $OUT/tune.log 2>&1
: > $OUT/sweep.txt
for mb in 8 16 32 64 128 256 1048576; do
  ms=$(DTF_BWD_STATS_SMALL_BYTES=$((mb << 20)) timeout -k 10 200 python bench.py --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])")
  echo "threshold ${mb} MB: ${ms} ms/step" | tee -a $OUT/sweep.txt
done
