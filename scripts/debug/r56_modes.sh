#!/bin/bash
# ResNet-56 bs128 step time by launch mode: eager / hipGraph replay x two lanes / one lane.
cd "$(dirname "$0")/../.."
run() { env "$@" timeout -k 10 120 python bench.py --model resnet56 --batch 128 --steps 300 --warmup 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['config']['launch'])"; }
echo "eager, two lanes:  $(run DTF_HIP_GRAPH=0)"
echo "graph, two lanes:  $(run DTF_HIP_GRAPH=1)"
echo "eager, one lane:   $(run DTF_HIP_GRAPH=0 DTF_WGRAD_STREAM=0)"
echo "graph, one lane:   $(run DTF_HIP_GRAPH=1 DTF_WGRAD_STREAM=0)"
echo "graph, two lanes, fork groups of 8: $(run DTF_HIP_GRAPH=1 DTF_WGRAD_GROUP=8)"
echo "graph, two lanes, fork groups of 16: $(run DTF_HIP_GRAPH=1 DTF_WGRAD_GROUP=16)"
echo "graph, two lanes, fenced events: $(run DTF_HIP_GRAPH=1 DTF_EVENT_FLAGS=fence)"
