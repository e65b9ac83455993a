#!/bin/bash
b() { timeout -k 10 120 python bench.py --steps 50 --warmup 10 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
# A/B of BatchNorm + ReLU on the operand fragments of the pointwise consumers (DTF_BN_IN_FRAG):
#   sh scripts/debug/bn_in_frag_ab.sh   (one MI355X, ~3 minutes)
echo "off        $(DTF_BN_IN_FRAG=0 b)"
echo "on cfg17   $(DTF_BN_IN_FRAG=1 b)"
echo "on cfg30   $(DTF_BN_IN_FRAG=1 DTF_BN_IN_FRAG_CFG=30 b)"
echo "on cfg31   $(DTF_BN_IN_FRAG=1 DTF_BN_IN_FRAG_CFG=31 b)"
echo "on cfg23   $(DTF_BN_IN_FRAG=1 DTF_BN_IN_FRAG_CFG=23 b)"
echo "big only17 $(DTF_BN_IN_FRAG=1 DTF_BN_IN_FRAG_BYTES=60000000 b)"
echo "off        $(DTF_BN_IN_FRAG=0 b)"
echo "on cfg17   $(DTF_BN_IN_FRAG=1 b)"
