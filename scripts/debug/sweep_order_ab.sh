#!/bin/bash
# A/B of the cache-aware sweep directions (DTF_SWEEP_ORDER=0: every launch front to back).
mkdir -p gpurun_out/s5
out=gpurun_out/s5/sweep_ab.txt
: > $out
for m in 0 auto 0 auto; do
  echo "== DTF_SWEEP_ORDER=$m" >> $out
  DTF_SWEEP_ORDER=$m timeout -k 10 120 python bench.py --steps 40 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['final_loss'])" >> $out || exit 1
done
cat $out
