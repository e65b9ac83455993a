#!/bin/bash
# A/B runs around the side lane's weight-gradient tile rule (HipExecutor._lane_tile): other tiles for
# the layers it leaves alone (DTF_WGRAD_LANE_MAP=tuned:forced,...), scaled split counts
# (DTF_WGRAD_LANE_SPLITS_MUL).  A 128x256 tile (four waves of 64x128, two groups) for layers with
# Cout % 256 == 0 was built and measured at 20.40 against 20.33 ms: not kept.
# Measured (ResNet-50 bs256, ms per step): rule 20.24 | 16->24 20.43 |
# 16->8 20.58 | 16->0 20.77 | 24,88->16 20.30 | 24,88->8 20.29; splits x0.75 20.38, x1.25 20.45, x0.5 20.50 (rule 20.33).
run() {
  echo "== $*"
  env "$@" timeout -k 10 120 python bench.py --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['final_loss'])" || exit 1
}
run DTF_X=1
run DTF_WGRAD_LANE_MAP=16:24
run DTF_WGRAD_LANE_SPLITS_MUL=0.75
run DTF_X=1
