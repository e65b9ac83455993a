#!/bin/bash
run() { printf "%-60s" "$*"; env "$@" timeout -k 10 120 python bench.py --model resnet56 --batch 128 --steps 300 --warmup 50 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
run DTF_WGRAD_BATCH=0 && run DTF_WGRAD_BATCH=1 && run DTF_WGRAD_BATCH=0 DTF_WGRAD_GROUP=2 && run DTF_WGRAD_BATCH=0 DTF_WGRAD_GROUP=3 && run DTF_WGRAD_BATCH=0 DTF_WGRAD_GROUP=6 && run DTF_WGRAD_BATCH=0 DTF_HIP_GRAPH=0 && run DTF_WGRAD_BATCH=0 DTF_WGRAD_STREAM=0
