#!/bin/bash
# ResNet-56 bs128: launch mode / weight-gradient grouping / lane matrix (ms per step).
run() {
  echo "== $*"
  env "$@" timeout -k 10 120 python bench.py --model resnet56 --batch 128 --steps 300 --warmup 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['launch'])" || exit 1
}
run DTF_X=1
run DTF_HIP_GRAPH=1
run DTF_HIP_GRAPH=0
run DTF_WGRAD_GROUP=2
run DTF_WGRAD_GROUP=8
run DTF_WGRAD_GROUP=16
run DTF_WGRAD_BATCH=1
run DTF_WGRAD_BATCH=1 DTF_WGRAD_GROUP=9
run DTF_WGRAD_STREAM=0
run DTF_SWEEP_ORDER=0
run DTF_X=1
