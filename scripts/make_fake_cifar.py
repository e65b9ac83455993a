"""Writes CIFAR-10-shaped binary files with random content (3073-byte records: label + 3072 CHW
bytes), for pipeline tests and throughput measurements without the dataset.

    python scripts/make_fake_cifar.py <dir> [records_per_file]
"""
import os, sys
import numpy as np

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
os.makedirs(d, exist_ok=True)
rng = np.random.default_rng(0)
for name in ["data_batch_%d.bin" % i for i in range(1, 6)] + ["test_batch.bin"]:
  rec = rng.integers(0, 256, size=(n, 3073), dtype=np.uint8)
  rec[:, 0] = rng.integers(0, 10, size=n, dtype=np.uint8)
  rec.tofile(os.path.join(d, name))
print("wrote 6 files of %d records to %s" % (n, d))
