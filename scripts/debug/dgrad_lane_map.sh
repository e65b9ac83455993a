#!/bin/bash
# A/B runs: other tiles for the data-gradient launches of the backward plan (they run next to the
# side lane's weight gradients; the table is tuned on kernels running alone).
# DTF_DGRAD_LANE_MAP=tuned:forced,...
run() {
  echo "== $*"
  env "$@" timeout -k 10 120 python bench.py --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['final_loss'])" || exit 1
}
run DTF_X=1
run DTF_DGRAD_LANE_MAP=0:6
run DTF_DGRAD_LANE_MAP=18:31
run DTF_DGRAD_LANE_MAP=18:30
run DTF_DGRAD_LANE_MAP=31:30
run DTF_DGRAD_LANE_MAP=30:31
run DTF_DGRAD_LANE_MAP=30:17
run DTF_DGRAD_LANE_MAP=10:21
run DTF_DGRAD_LANE_MAP=5:10
run DTF_X=1
