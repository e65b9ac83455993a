#!/bin/bash
# Kernel trace of the worker of a 1 ps + 1 worker parameter-server job sharing one GPU (gloo).
# The ps rank runs as a plain background process and exits when the worker is done.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp DTF_BACKEND=gloo MASTER_ADDR=127.0.0.1 MASTER_PORT=29731 WORLD_SIZE=2 LOCAL_WORLD_SIZE=2 DTF_NUM_PS=1
ARGS="--use_synthetic_data --dtype bf16 --batch_size 256 --distribution_strategy parameter_server --num_gpus 2 --train_steps 24 --skip_eval --log_steps 8"
OUT=gpurun_out/ps_trace
rm -rf $OUT; mkdir -p $OUT
RANK=0 LOCAL_RANK=0 timeout -k 10 240 python resnet_imagenet_main.py $ARGS > $OUT/ps.log 2>&1 &
PS_PID=$!
RANK=1 LOCAL_RANK=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/w -- \
    python resnet_imagenet_main.py $ARGS > $OUT/worker.log 2>&1
rc=$?
wait $PS_PID
grep -E "global step|waiting|updates" $OUT/worker.log $OUT/ps.log | cut -c 1-300
exit $rc
