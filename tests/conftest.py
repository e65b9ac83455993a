import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")
  config.addinivalue_line("markers", "slow: long-running test")


def pytest_collection_modifyitems(config, items):
  """GPU tests are skipped when no GPU is visible (they are selected with -m gpu)."""
  import torch
  # device_count() does not initialise the GPU (is_available() would, and a process that has
  # initialised the GPU must not start child programs on the test machines)
  if torch.cuda.device_count() > 0:
    return
  skip = pytest.mark.skip(reason="needs a GPU")
  for item in items:
    if "gpu" in item.keywords:
      item.add_marker(skip)


@pytest.fixture(autouse=True)
def _fresh_flags():
  """Every test starts from unparsed flags with library defaults."""
  from distributed_tensorflow_amd.utils import flags
  yield
  flags.FLAGS.unparse_flags()
