"""Pull / push latency of the parameter server for a model's flat buffers (1 ps + 1 worker).

    PS_MODEL=resnet50 PS_ITERS=5 python -m distributed_tensorflow_amd.launch --nproc 2 --num_ps 1 scripts/bench_ps_latency.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(2)
from distributed_tensorflow_amd import models
from distributed_tensorflow_amd.parallel.parameter_server import ParameterServerStrategy
from distributed_tensorflow_amd.train.model import Model
from distributed_tensorflow_amd.train.optimizers import SGD
st = ParameterServerStrategy(use_gpu=os.environ.get("PS_GPU") == "1")
m = Model(getattr(models, os.environ.get("PS_MODEL", "resnet20"))(), strategy=st, dtype="fp32", device=st.device, executor="torch")
m.compile(SGD(0.1, momentum=0.9), "sparse_categorical_crossentropy")
if st.is_ps_task:
  st.serve(m)
else:
  m._sync_initial_variables()
  t0 = time.time()
  for i in range(int(os.environ.get("PS_ITERS", "50"))):
    st.pull(m.store)
  t1 = time.time()
  for i in range(int(os.environ.get("PS_ITERS", "50"))):
    st.push(m.store, 0.1, 1.0)
  t2 = time.time()
  print("pull %.2f ms  push %.2f ms" % ((t1 - t0) * 1000 / int(os.environ.get("PS_ITERS", "50")), (t2 - t1) * 1000 / int(os.environ.get("PS_ITERS", "50"))), flush=True)
  st.shutdown()
