#! /bin/bash
# Two-host multi_worker_mirrored CIFAR run, 8 GPUs per host (the recipe of the reference's
# image_classification/run.sh).  Run it on every host with its own TASK_INDEX:
#
#   WORKER_HOSTS=hostA:8080,hostB:8080 TASK_INDEX=0 sh run.sh      # on hostA
#   WORKER_HOSTS=hostA:8080,hostB:8080 TASK_INDEX=1 sh run.sh      # on hostB
#
# Proxies are cleared because rendezvous connections must not go through an HTTP proxy.
export http_proxy=''
export https_proxy=''
export HTTP_PROXY=''
export HTTPS_PROXY=''
HERE="$(cd "$(dirname "$0")" && pwd)"
export PYTHONPATH="$PYTHONPATH:$HERE"

python "$HERE/resnet_cifar_main_dist.py" --data_dir cifar-10-batches-bin/ \
  --distribution_strategy multi_worker_mirrored --num_gpus ${NUM_GPUS:-8} \
  --worker_hosts ${WORKER_HOSTS:-localhost:8080} --task_index ${TASK_INDEX:-0} "$@"
#python "$HERE/resnet_cifar_main_dist_ps.py" --data_dir cifar-10-batches-bin/ --distribution_strategy parameter_server
