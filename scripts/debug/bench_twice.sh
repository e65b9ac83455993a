#!/bin/bash
# bench.py twice (ResNet-50 bs256) plus the serial single-lane time, for quick A/B runs of a build.
mkdir -p gpurun_out/s5
for i in 1 2; do
  env "$@" timeout -k 10 120 python bench.py --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two-lane', d['ms_per_step'], d['config']['final_loss'])" || exit 1
done
env "$@" DTF_WGRAD_STREAM=0 timeout -k 10 120 python bench.py --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-lane', d['ms_per_step'])"
