#!/bin/bash
# A/B of the non-temporal cache policy of the BatchNorm sweeps (DTF_BN_NT bit mask, DTF_BN_NT_BYTES
# threshold, DTF_BN_GRID_CAP blocks).
mkdir -p gpurun_out/s5
out=gpurun_out/s5/nt_ab.txt
: > $out
run() {
  echo "== $*" >> $out
  env "$@" timeout -k 10 120 python bench.py --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['final_loss'])" >> $out || exit 1
}
run DTF_BN_NT=0
run DTF_BN_NT=0x3f
run DTF_BN_NT=0x15
run DTF_BN_NT=0x2a
run DTF_BN_NT=0x1f
run DTF_BN_NT=0x3f DTF_BN_GRID_CAP=256
run DTF_BN_NT=0x3f DTF_BN_NT_BYTES=90000000
run DTF_BN_NT=0x3f DTF_BN_NT_BYTES=300000000
run DTF_BN_NT=0x15 DTF_BN_NT_BYTES=90000000
run DTF_BN_NT=0
cat $out
