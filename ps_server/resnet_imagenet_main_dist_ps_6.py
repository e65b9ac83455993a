"""ResNet-50 / ImageNet parameter-server job, process 6 of 16: worker 5.

Same numbering as `ps_server/resnet_imagenet_main_dist_ps_6.py:38-50` of the reference (file 0
is the ps task, file k is worker k-1 on GPU k mod 4).  The cluster is 1 ps + 15 workers; its
addresses come from `DTF_PS_HOSTS` / `DTF_WORKER_HOSTS` (defaults: ports 1111.. on this
machine) instead of a literal in the file.  Everything else is
`ps_server/resnet_imagenet_main_dist_ps.py` with `--proc_index 6`.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import resnet_imagenet_main_dist_ps as base  # noqa: E402  pylint: disable=wrong-import-position

if __name__ == "__main__":
  base.run_numbered(6)
