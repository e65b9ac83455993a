#!/bin/bash
# A/B runs around the side lane's weight-gradient tile rule (HipExecutor._lane_tile): other tiles for
# the layers it leaves alone (DTF_WGRAD_LANE_MAP=tuned:forced,...) and scaled split counts
# (DTF_WGRAD_LANE_SPLITS_MUL).  Measured (ResNet-50 bs256, ms per step): rule 20.24 | 16->24 20.43 |
# 16->8 20.58 | 16->0 20.77 | 24,88->16 20.30 | 24,88->8 20.29.
run() {
  echo "== $*"
  env "$@" timeout -k 10 120 python bench.py --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['final_loss'])" || exit 1
}
run DTF_X=1
run DTF_WGRAD_LANE_SPLITS_MUL=0.75
run DTF_WGRAD_LANE_SPLITS_MUL=1.25
run DTF_WGRAD_LANE_SPLITS_MUL=0.5
run DTF_X=1
