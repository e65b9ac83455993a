"""One-rank RCCL rehearsal of the reducer path with a stack dump if it stalls."""
import faulthandler, os, sys
faulthandler.dump_traceback_later(int(os.environ.get("DTF_DUMP_AFTER", "60")), exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["DTF_REHEARSE_COLLECTIVES"] = "1"
import torch
from distributed_tensorflow_amd import models
from distributed_tensorflow_amd.parallel import strategy as S
from distributed_tensorflow_amd.train.model import Model
from distributed_tensorflow_amd.train.optimizers import SGD
from distributed_tensorflow_amd.ops import native
dev = torch.device("cuda")
g = torch.Generator().manual_seed(3)
x = (torch.randn(16, 32, 32, 3, generator=g) * 50 + 100).to(dev)
y = torch.randint(0, 10, (16, 1), generator=g).float().to(dev)
plain_first = "plain" in sys.argv
for name in (["plain"] if plain_first else []) + ["rehearsed"]:
  strat = None if name == "plain" else S.MirroredStrategy(use_gpu=True, num_packs=3)
  m = Model(models.resnet20(), strategy=strat, dtype="bf16", device=dev, seed=1)
  m.compile(SGD(0.05, momentum=0.9), "sparse_categorical_crossentropy")
  for i in range(10):
    l = m.train_on_batch(x, y)[0]
    print(name, i, float(l), [round(v, 2) for v in native.ext().last_lane_probe()], flush=True)
print("done", flush=True)
