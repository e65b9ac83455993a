#!/bin/bash
run() { printf "%-60s" "$*"; env "$@" timeout -k 10 120 python bench.py --model resnet56 --batch 128 --steps 300 --warmup 50 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
run DTF_X=0 && run DTF_BN_GRID_CAP=256 && run DTF_BN_GRID_CAP=1024 && run DTF_BN_GRID_CAP=128 && run DTF_WGRAD_GROUP=3 && run DTF_BWD_STATS_SMALL_BYTES=0
for i in 1 2; do printf "resnet50 "; timeout -k 10 120 python bench.py --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; done
