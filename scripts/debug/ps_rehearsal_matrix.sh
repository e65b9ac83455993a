#!/bin/bash
# One-GPU rehearsal of the parameter server (1 ps + 1 worker over gloo) under a few settings.
run() {
  echo "== $*"
  env "$@" DTF_BACKEND=gloo timeout -k 10 200 python resnet_imagenet_main.py --use_synthetic_data --dtype bf16 \
      --batch_size 256 --distribution_strategy parameter_server --num_gpus 2 --train_steps 60 --skip_eval \
      --log_steps 20 > gpurun_out/ps_m.log 2>&1 || { tail -5 gpurun_out/ps_m.log; return 1; }
  grep -E "global step':60|waiting|updates|eager" gpurun_out/ps_m.log | cut -c 40-400
}
run DTF_X=0 && run DTF_HIP_GRAPH=0 && run DTF_HIP_GRAPH=0 DTF_PS_PUSH_DTYPE=fp32 && run DTF_HIP_GRAPH=0 DTF_PS_BUCKET_MB=200
