"""MI355X-native distributed ResNet training framework.

Python package layout:

* `models/`     layer-graph IR and the model zoo (ResNet-50 v1.5, ResNet-20/32/56/110, trivial)
* `engine/`     flat variable store, PyTorch reference executor, native HIP executor
* `ops/`        hand-written HIP kernels for gfx950 (`csrc/`), their Python bindings and
                the PyTorch reference implementations used as numerical oracles
* `parallel/`   cluster spec, communicator (RCCL / gloo), bucketed all-reduce, strategies,
                asynchronous parameter server
* `train/`      Keras-like Model, callbacks, optimizers, schedules, stats, shared driver
* `data/`       synthetic, CIFAR-10 binary, ImageNet TFRecord input pipelines
* `checkpoint/` pure-Python HDF5 writer/reader with the Keras model layout
* `utils/`      flags, logging, benchmark logger, TFRecord/proto codec
* `hvd`         Horovod-style facade;  `launch`  one-process-per-GPU launcher
"""
__version__ = "0.1.0"


# HIP maps every stream of a process onto GPU_MAX_HW_QUEUES hardware queues (4 by default).
# A training rank uses a compute stream, the collective library's streams, a hipGraph capture
# stream and the prefetcher's copy stream; when they share a queue, a wait packet of one
# stream holds back kernels of another (observed as a stalled step with gloo's host-staging
# streams).  Eight queues keep them apart.  Only effective when set before the HIP runtime
# initialises, which importing this package precedes.
import os as _os

_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# RCCL (and device-tensor sharing between processes) exchange memory handles through the
# dmabuf IPC path; hosts whose driver only supports that path need the legacy mode switched off
# or `hipIpcGetMemHandle` fails with "invalid argument" as soon as two ranks talk.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
