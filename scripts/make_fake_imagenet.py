"""Writes an ImageNet-shaped TFRecord tree with synthetic JPEGs (1024 `train-*` and 128
`validation-*` shards, `image/encoded` + label + one bounding box per record), for end-to-end
runs and input-pipeline throughput measurements without the dataset.

    python scripts/make_fake_imagenet.py <dir> [images_per_shard] [height] [width]
"""
import io, os, sys
import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import tfrecord

d = sys.argv[1]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 2
H = int(sys.argv[3]) if len(sys.argv) > 3 else 375
W = int(sys.argv[4]) if len(sys.argv) > 4 else 500
os.makedirs(d, exist_ok=True)
rng = np.random.default_rng(0)
# a few smooth images (JPEG cost like photographs, not like noise), re-used with different labels
pool = []
for _ in range(16):
  small = rng.integers(0, 256, size=(H // 8 + 1, W // 8 + 1, 3), dtype=np.uint8)
  buf = io.BytesIO()
  Image.fromarray(small).resize((W, H), Image.BICUBIC).save(buf, format="JPEG", quality=90)
  pool.append(buf.getvalue())
k = 0
for pattern, n in (("train-%05d-of-01024", 1024), ("validation-%05d-of-00128", 128)):
  for i in range(n):
    with open(os.path.join(d, pattern % i), "wb") as f:
      for _ in range(per):
        ex = tfrecord.encode_example({
            "image/encoded": pool[k % len(pool)], "image/class/label": [k % 1000 + 1],
            "image/class/text": b"synthetic",
            "image/object/bbox/xmin": [0.1], "image/object/bbox/ymin": [0.1],
            "image/object/bbox/xmax": [0.9], "image/object/bbox/ymax": [0.9]})
        tfrecord.write_record(f, ex)
        k += 1
print("wrote %d records (%d bytes per image on average) to %s"
      % (k, sum(map(len, pool)) // len(pool), d))
