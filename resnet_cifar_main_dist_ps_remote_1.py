"""ResNet-56 / CIFAR-10, asynchronous parameter server: the worker task of a two-host job.

Entry point with the role fixed by the file name, like `image_classification/resnet_cifar_main_dist_ps_remote_1.py:36-48`
(one ps + one worker; `_0` is the ps task, `_1` the worker).  The two hosts come from the environment: `DTF_PS_HOST` (default `localhost:8080`) and
`DTF_WORKER_HOST` (default `localhost:8081`) - the reference hard-codes two machine addresses.
Everything else is `resnet_cifar_main_dist_ps.py`.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import resnet_cifar_main_dist_ps as base  # noqa: E402  pylint: disable=wrong-import-position
from distributed_tensorflow_amd.utils import flags as flags_core  # noqa: E402
from distributed_tensorflow_amd.utils import logging  # noqa: E402

if __name__ == "__main__":
  logging.set_verbosity(logging.INFO)
  base.define_flags()
  flags_core.set_defaults(ps_hosts=os.environ.get("DTF_PS_HOST", "localhost:8080"),
                          worker_hosts=os.environ.get("DTF_WORKER_HOST", "localhost:8081"),
                          task_type='worker', task_index=0)
  flags_core.run(base.main)
