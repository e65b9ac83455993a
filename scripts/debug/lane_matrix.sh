#!/bin/bash
# bench.py under a few lane-scheduling settings (one line each)
run() {
  printf "%-70s" "$*"
  env "$@" timeout -k 10 120 python bench.py --steps 30 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
}
run DTF_X=0 && run DTF_SIDE_PRIORITY=low && run DTF_SIDE_PRIORITY=high && \
run DTF_SIDE_PRIORITY=low DTF_WGRAD_AFTER_DGRAD=1 && run DTF_SIDE_PRIORITY=low DTF_HIP_GRAPH=0 && \
run DTF_SIDE_PRIORITY=low DTF_WGRAD_AFTER_DGRAD=1 DTF_HIP_GRAPH=0 && run DTF_HIP_GRAPH=0 && run DTF_SIDE_PRIORITY=high DTF_HIP_GRAPH=0
