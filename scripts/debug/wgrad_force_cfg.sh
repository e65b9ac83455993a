#!/bin/bash
# Two-lane step time with the side lane's weight-gradient tile rule on and off
# (DTF_WGRAD_LANE_TILE; the per-layer table is tuned on kernels running alone).
# Measured while the rule was derived (ResNet-50 bs256, one MI355X, ms per step):
#   table as tuned 20.81 | 128x128 x 2 groups forced on every layer with K >= 256, Cout >= 128: 20.44
#   only layers tuned to 128x64: 20.65 | to 64x128: 20.67 | to 128x128 x 4 groups: 20.65 | to 256x128: 20.83
#   forced tile with split counts x2: 20.74, x1.5: 20.60, x0.5: 20.68; 256x128 forced: 20.52; 3-stage ring: 20.67
run() {
  echo "== $*"
  env "$@" timeout -k 10 120 python bench.py --steps 40 --warmup 10 $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['final_loss'])" || exit 1
}
run DTF_WGRAD_LANE_TILE=0
run DTF_WGRAD_LANE_TILE=1
run DTF_WGRAD_LANE_TILE=0
run DTF_WGRAD_LANE_TILE=1
run DTF_WGRAD_LANE_TILE=1 DTF_WGRAD_STREAM=0
EXTRA="--dtype fp16" run DTF_WGRAD_LANE_TILE=1
EXTRA="--dtype fp16" run DTF_WGRAD_LANE_TILE=0
