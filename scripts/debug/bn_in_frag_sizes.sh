#!/bin/bash
# Smallest input tensor that takes the fragment BatchNorm path (DTF_BN_IN_FRAG_BYTES), two alternating rounds:
#   sh scripts/debug/bn_in_frag_sizes.sh   (one MI355X, ~4 minutes)
b() { timeout -k 10 120 python bench.py --steps 50 --warmup 10 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for round in 1 2; do
  echo "off                 $(DTF_BN_IN_FRAG=0 b)"
  for t in 60000000 40000000 20000000 0; do
    echo "from $t bytes $(DTF_BN_IN_FRAG_BYTES=$t b)"
  done
done
