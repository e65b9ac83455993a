#! /bin/bash
# Stops the processes started by run.sh (the reference's kill.sh runs `pkill resnet_imagenet`
# on every host because its ps task never exits; here the ps task leaves on its own once all
# workers are done, and this script only terminates the exact PIDs recorded by run.sh).
HERE="$(cd "$(dirname "$0")" && pwd)"
[ -f "$HERE/pids" ] || { echo "no ps_server/pids file: nothing to stop"; exit 0; }
while read -r host pid; do
  pid=${pid#ssh:}
  kill "$pid" 2>/dev/null && echo "stopped $pid ($host)"
done < "$HERE/pids"
rm -f "$HERE/pids"
