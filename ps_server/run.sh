#! /bin/bash
# Starts a 1 ps + N worker asynchronous parameter-server job, one process per GPU, stderr of
# process i in log$i.log - the job layout of the reference's ps_server/run.sh (16 processes on
# 4 hosts x 4 GPUs over ssh), sized by environment variables:
#
#   HOSTS="h0 h1"  GPUS_PER_HOST=8  PORT0=1111  sh ps_server/run.sh [extra driver flags]
#
# HOSTS defaults to this machine only (no ssh).  Process 0 is the ps task, process k >= 1 is
# worker k-1 and uses GPU k mod GPUS_PER_HOST.  PIDs go to ps_server/pids so that kill.sh can
# stop exactly these processes.
HERE="$(cd "$(dirname "$0")" && pwd)"
HOSTS=${HOSTS:-localhost}
GPUS_PER_HOST=${GPUS_PER_HOST:-8}
PORT0=${PORT0:-1111}
PY=${PYTHON:-python}
set -- --distribution_strategy parameter_server --batch_size 192 --train_epochs 1 "$@"

addrs=""
for h in $HOSTS; do
  for g in $(seq 0 $((GPUS_PER_HOST - 1))); do
    addrs="$addrs $h:$((PORT0 + g))"
  done
done
set_ps=$(echo $addrs | cut -d' ' -f1)
workers=$(echo $addrs | cut -d' ' -f2- | tr ' ' ',')

: > "$HERE/pids"
i=0
for a in $addrs; do
  h=${a%%:*}
  cmd="DTF_GPUS_PER_HOST=$GPUS_PER_HOST $PY $HERE/resnet_imagenet_main_dist_ps.py --proc_index $i --ps_hosts $set_ps --worker_hosts $workers $*"
  if [ "$h" = "localhost" ] || [ "$h" = "127.0.0.1" ]; then
    sh -c "$cmd" 2>"log$i.log" 1>/dev/null &
    echo "$h $!" >> "$HERE/pids"
  else
    ssh "$h" "$cmd" 2>"log$i.log" 1>/dev/null &
    echo "$h ssh:$!" >> "$HERE/pids"
  fi
  sleep 1
  i=$((i + 1))
done
echo "started $i processes (1 ps + $((i - 1)) workers); logs: log0.log .. log$((i - 1)).log"
