#!/bin/bash
# A/B of planner knobs inside one GPU call: every setting twice, interleaved with the default.
#   bash scripts/debug/knob_sweep.sh "DTF_BWD_STATS_SMALL_BYTES=33554432" "DTF_BN_NT_BYTES=60000000" ...
cd "$(dirname "$0")/../.."
run() { env "$@" timeout -k 10 120 python bench.py --steps 30 --warmup 5 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
for rep in 1 2; do
  echo "default: $(run DTF_NOOP=1)"
  for kv in "$@"; do echo "$kv: $(run $kv)"; done
done
