#!/bin/bash
# Every bench.py configuration quoted in the result tables, one JSON line each.
cd "$(dirname "$0")/.."
run() { echo "# bench.py $*"; timeout -k 10 300 python bench.py "$@" 2>/dev/null | tail -1; }
run --steps 50 --warmup 10 && run --dtype fp32 --steps 20 --warmup 5 && run --dtype fp16 --steps 30 --warmup 10 && \
run --executor torch --dtype fp32 --steps 10 --warmup 3 && run --executor torch --steps 10 --warmup 3 && \
run --model resnet56 --batch 128 --steps 200 --warmup 30 && run --model resnet56 --batch 128 --dtype fp32 --steps 100 --warmup 30 && \
run --rehearse_collectives --steps 30 --warmup 10 && run --model resnet56 --batch 128 --rehearse_collectives --num_packs 1 --steps 200 --warmup 30
