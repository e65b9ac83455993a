"""Input pipelines on small generated files (CIFAR-10 binary records, ImageNet TFRecords)."""
import io
import os

import numpy as np
import pytest
import torch

from distributed_tensorflow_amd.data import cifar_preprocessing as cifar
from distributed_tensorflow_amd.data import imagenet_preprocessing as imnet
from distributed_tensorflow_amd.data.synthetic import get_synth_input_fn
from distributed_tensorflow_amd.utils import tfrecord


def _write_cifar(dirname, per_file=24):
  rng = np.random.default_rng(0)
  os.makedirs(dirname, exist_ok=True)
  names = ["data_batch_%d.bin" % i for i in range(1, 6)] + ["test_batch.bin"]
  k = 0
  for n in names:
    with open(os.path.join(dirname, n), "wb") as f:
      for _ in range(per_file):
        label = k % 10
        img = rng.integers(0, 256, size=3072, dtype=np.uint8)
        img[0] = k % 256                    # tag the record so duplicates are detectable
        f.write(bytes([label]) + img.tobytes())
        k += 1
  return names


def test_cifar_parse_and_preprocess():
  raw = bytes([7]) + np.arange(3072, dtype=np.uint32).astype(np.uint8).tobytes()
  img, lab = cifar.parse_record(raw, is_training=False)
  assert img.shape == (32, 32, 3) and img.dtype == np.float32
  assert lab.tolist() == [0, 0, 0, 0, 0, 0, 0, 1, 0, 0]
  # CHW -> HWC: pixel (0,0) holds bytes 0, 1024, 2048 before standardisation
  assert abs(img.mean()) < 1e-5 and abs(img.std() - 1.0) < 1e-3
  a, _ = cifar.parse_record(raw, True, rng=np.random.default_rng(1))
  b, _ = cifar.parse_record(raw, True, rng=np.random.default_rng(2))
  assert a.shape == (32, 32, 3) and not np.array_equal(a, b)       # random crop / flip


@pytest.mark.parametrize("workers", [0, 2])
def test_cifar_input_fn(tmp_path, workers):
  _write_cifar(str(tmp_path))
  ds = cifar.input_fn(True, str(tmp_path), batch_size=8, datasets_num_private_threads=workers,
                      drop_remainder=True)
  it = iter(ds)
  for _ in range(3):
    x, y = next(it)
    assert tuple(x.shape) == (8, 32, 32, 3) and x.dtype == torch.float32
    assert tuple(y.shape) == (8, 10) and torch.all(y.sum(1) == 1)
  ev = cifar.input_fn(False, str(tmp_path), batch_size=6, datasets_num_private_threads=workers)
  n = sum(x.shape[0] for x, _ in ev)
  assert n == 24                                            # one pass over test_batch.bin


def test_cifar_sharding_is_disjoint(tmp_path):
  _write_cifar(str(tmp_path))
  seen = []
  for shard in range(2):
    ev = cifar.input_fn(False, str(tmp_path), batch_size=4, datasets_num_private_threads=0,
                        shard_index=shard, num_shards=2)
    seen.append(sum(x.shape[0] for x, _ in ev))
  assert seen == [12, 12]


def _jpeg(h, w, seed):
  from PIL import Image
  rng = np.random.default_rng(seed)
  arr = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
  buf = io.BytesIO()
  Image.fromarray(arr).save(buf, format="JPEG", quality=90)
  return buf.getvalue()


def _write_imagenet(dirname, nfiles=3, per_file=5):
  files = []
  k = 0
  for i in range(nfiles):
    p = os.path.join(dirname, "train-%05d-of-01024" % i)
    with open(p, "wb") as f:
      for _ in range(per_file):
        ex = tfrecord.encode_example({
            "image/encoded": _jpeg(300 + 10 * k, 400, k),
            "image/class/label": [k + 1],
            "image/class/text": b"thing",
            "image/object/bbox/xmin": [0.1], "image/object/bbox/ymin": [0.2],
            "image/object/bbox/xmax": [0.9], "image/object/bbox/ymax": [0.8]})
        tfrecord.write_record(f, ex)
        k += 1
    files.append(p)
  return files


def test_tfrecord_and_example_codec(tmp_path):
  assert tfrecord.crc32c(b"123456789") == 0xE3069283         # CRC-32C check value
  p = tmp_path / "x.tfrecord"
  ex = tfrecord.encode_example({"a": b"bytes", "b": [1, -2, 3], "c": [0.5, 1.5]})
  with open(p, "wb") as f:
    tfrecord.write_record(f, ex)
    tfrecord.write_record(f, b"")
  with open(p, "rb") as f:
    recs = list(tfrecord.read_records(f, verify=True))
  assert recs[1] == b""
  d = tfrecord.decode_example(recs[0])
  assert d["a"] == [b"bytes"] and d["b"] == [1, -2, 3] and d["c"] == pytest.approx([0.5, 1.5])
  bad = bytearray(open(p, "rb").read())
  bad[14] ^= 0xFF
  with pytest.raises(IOError):
    list(tfrecord.read_records(io.BytesIO(bytes(bad)), verify=True))


def test_imagenet_parse_record(tmp_path):
  files = _write_imagenet(str(tmp_path), 1, 2)
  with open(files[0], "rb") as f:
    rec = next(tfrecord.read_records(f))
  img, lab = imnet.parse_record(rec, is_training=True, rng=np.random.default_rng(0))
  assert img.shape == (224, 224, 3) and img.dtype == np.float32
  assert lab.tolist() == [0.0] and lab.dtype == np.float32         # label - 1, float32, shape [1]
  ev, _ = imnet.parse_record(rec, is_training=False)
  assert ev.shape == (224, 224, 3)
  # mean subtraction: values are centred around zero, inside [-124, 152]
  assert -124.0 <= ev.min() and ev.max() <= 152.0
  ev2, _ = imnet.parse_record(rec, is_training=False)
  assert np.array_equal(ev, ev2)                                    # eval path is deterministic


def test_sample_distorted_bounding_box_constraints():
  rng = np.random.default_rng(0)
  box = np.array([[0.2, 0.1, 0.8, 0.9]], dtype=np.float32)
  for _ in range(200):
    y, x, h, w = imnet.sample_distorted_bounding_box(300, 400, box, rng)
    assert 0 <= y and y + h <= 300 and 0 <= x and x + w <= 400
    assert 0.05 * 300 * 400 * 0.9 <= h * w <= 300 * 400
    assert 0.74 <= w / h <= 1.34
  assert imnet._smallest_size_at_least(300, 400, 256) == (256, 341)


def test_imagenet_input_fn(tmp_path):
  files = _write_imagenet(str(tmp_path))
  ds = imnet.input_fn(True, str(tmp_path), batch_size=4, datasets_num_private_threads=2,
                      drop_remainder=True, filenames=files)
  it = iter(ds)
  x, y = next(it)
  assert tuple(x.shape) == (4, 224, 224, 3) and tuple(y.shape) == (4, 1)
  assert y.min() >= 0 and y.max() < 15
  assert len(imnet.get_filenames(True, "/d")) == 1024 and len(imnet.get_filenames(False, "/d")) == 128
  assert imnet.get_filenames(True, "/d")[5] == "/d/train-00005-of-01024"


def test_synthetic_input_matches_reference_contract():
  fn = get_synth_input_fn(224, 224, 3, 1001)
  ds = fn(True, None, 4)
  x, y = next(iter(ds))
  assert tuple(x.shape) == (4, 224, 224, 3) and tuple(y.shape) == (4, 1) and y.dtype == torch.float32
  assert torch.equal(x[0], x[3]) and 0 <= y[0, 0] < 1000        # one sample repeated; last class never used
  assert 0 <= x.min() and x.max() <= 255
  x2, _ = next(iter(ds))
  assert x2.data_ptr() == x.data_ptr()                           # same resident batch every step


def test_native_cifar_loader_matches_python_pipeline(tmp_path):
  """The C++ loader yields exactly the records / preprocessing of the Python map function."""
  from distributed_tensorflow_amd.data import native_loader
  if not native_loader.available():
    pytest.skip("_dtf_data not built")
  _write_cifar(str(tmp_path), per_file=20)
  files = cifar.get_filenames(False, str(tmp_path))
  ds = native_loader.NativeCifarDataset(files, False, batch_size=7, num_epochs=1)
  got_x, got_y = [], []
  for x, y in ds:
    got_x.append(x.clone()); got_y.append(y.clone())
  gx, gy = torch.cat(got_x), torch.cat(got_y)
  ref = [cifar.parse_record(r, False) for r in cifar.read_fixed_length_records(files[0])]
  rx = torch.from_numpy(np.stack([a for a, _ in ref]))
  ry = torch.from_numpy(np.stack([b for _, b in ref]))
  assert gx.shape == rx.shape and torch.allclose(gx, rx, atol=1e-4)
  assert torch.equal(gy, ry)
  # a second pass starts from the first record again
  assert torch.allclose(next(iter(ds))[0], rx[:7], atol=1e-4)
  # training: every record once per epoch, standardised crops, batches span epochs
  tr = native_loader.NativeCifarDataset(cifar.get_filenames(True, str(tmp_path)), True, 16,
                                        num_epochs=None, seed=3, num_threads=3)
  assert tr.num_records == 100
  it = iter(tr)
  labels = []
  for _ in range(25):                                     # 400 samples = 4 epochs
    x, y = next(it)
    assert x.shape == (16, 32, 32, 3)
    assert torch.allclose(x.mean(dim=(1, 2, 3)), torch.zeros(16), atol=1e-4)
    assert torch.allclose(x.std(dim=(1, 2, 3), unbiased=False), torch.ones(16), atol=1e-3)
    labels.append(y.argmax(1))
  labels = torch.cat(labels)
  all_ref = torch.cat([torch.tensor([cifar.parse_record(r, False)[1].argmax()
                                     for r in cifar.read_fixed_length_records(f)])
                       for f in cifar.get_filenames(True, str(tmp_path))])
  for e in range(4):                                      # same label histogram each epoch
    assert torch.equal(torch.bincount(labels[100 * e:100 * (e + 1)], minlength=10),
                       torch.bincount(all_ref, minlength=10))
  assert not torch.equal(labels[:100], labels[100:200])   # reshuffled between epochs
  # shards are disjoint and cover the file
  n = [native_loader.NativeCifarDataset(files, False, 4, num_epochs=1, shard_index=s,
                                        num_shards=3).num_records for s in range(3)]
  assert sum(n) == 20 and max(n) - min(n) <= 1


def test_native_crc_and_tfrecord_reader(tmp_path):
  from distributed_tensorflow_amd.data import native_loader
  from distributed_tensorflow_amd.utils import tfrecord
  if not native_loader.available():
    pytest.skip("_dtf_data not built")
  ext = native_loader.ext()
  rng = np.random.default_rng(0)
  for n in (0, 1, 7, 8, 9, 63, 1000):
    data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    assert ext.crc32c(data) == tfrecord.crc32c(data)
    assert ext.masked_crc32c(data) == tfrecord.masked_crc32c(data)
  assert ext.crc32c(b"123456789") == 0xE3069283              # CRC-32C check value
  path = str(tmp_path / "x.tfrecord")
  payloads = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (5, 0, 300, 17)]
  with open(path, "wb") as f:
    for p in payloads:
      tfrecord.write_record(f, p)
  assert native_loader.read_tfrecords(path) == payloads
  raw = bytearray(open(path, "rb").read())
  raw[12 + 2] ^= 0x55                                          # flip a payload byte
  open(path, "wb").write(bytes(raw))
  with pytest.raises(RuntimeError, match="checksum"):
    native_loader.read_tfrecords(path)
  assert len(native_loader.read_tfrecords(path, verify=False)) == 4


def _tf_legacy_bilinear(crop, oh, ow):
  """`tf.compat.v1.image.resize(..., align_corners=False)`: src = dst * in / out, two taps."""
  H, W, _ = crop.shape
  ys, xs = np.arange(oh) * (H / oh), np.arange(ow) * (W / ow)
  y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
  y1, x1 = np.minimum(y0 + 1, H - 1), np.minimum(x0 + 1, W - 1)
  fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
  t0 = crop[y0][:, x0] * (1 - fx) + crop[y0][:, x1] * fx
  t1 = crop[y1][:, x0] * (1 - fx) + crop[y1][:, x1] * fx
  return t0 * (1 - fy) + t1 * fy


def test_native_crop_resize_matches_tensorflow_legacy_bilinear():
  from distributed_tensorflow_amd.data import native_loader
  if not native_loader.available():
    pytest.skip("native input module not built")
  ext = native_loader.ext()
  rng = np.random.default_rng(3)
  img = rng.integers(0, 256, size=(97, 130, 3), dtype=np.uint8)
  means = [123.68, 116.78, 103.94]
  # training flavour: crop box -> 64 x 48, with and without flip; an upscaling case too
  for (y, x, h, w, oh, ow) in [(5, 7, 80, 100, 64, 48), (0, 0, 97, 130, 224, 224), (10, 20, 33, 31, 17, 90)]:
    ref = _tf_legacy_bilinear(img[y:y + h, x:x + w].astype(np.float64), oh, ow) - np.asarray(means)
    out = ext.crop_resize_bilinear(img, y, x, h, w, oh, ow, 0, 0, oh, ow, False, means)
    assert out.dtype == np.float32 and out.shape == (oh, ow, 3)
    assert np.abs(out - ref).max() < 0.02
    flipped = ext.crop_resize_bilinear(img, y, x, h, w, oh, ow, 0, 0, oh, ow, True, means)
    assert np.abs(flipped - ref[:, ::-1]).max() < 0.02
  # evaluation flavour: the whole image resized to 150 x 201, central 112 x 112 window
  full = _tf_legacy_bilinear(img.astype(np.float64), 150, 201)
  top, left = (150 - 112) // 2, (201 - 112) // 2
  out = ext.crop_resize_bilinear(img, 0, 0, 97, 130, 150, 201, top, left, 112, 112, False, [])
  assert np.abs(out - full[top:top + 112, left:left + 112]).max() < 0.02
  with pytest.raises(ValueError):
    ext.crop_resize_bilinear(img, 90, 0, 20, 20, 8, 8, 0, 0, 8, 8, False, [])


def test_imagenet_preprocess_shapes_and_mean_subtraction():
  """Train / eval preprocessing of a constant-colour JPEG: every output pixel is the colour minus
  the channel means, whatever the crop, flip and resize did."""
  from PIL import Image
  buf = io.BytesIO()
  Image.new("RGB", (320, 200), (200, 100, 50)).save(buf, format="JPEG", quality=95)
  bbox = np.zeros((0, 4), dtype=np.float32)
  want = np.array([200, 100, 50]) - np.array([123.68, 116.78, 103.94])
  for training in (True, False):
    out = imnet.preprocess_image(buf.getvalue(), bbox, 224, 224, 3, training,
                                                  np.random.default_rng(0))
    assert out.shape == (224, 224, 3) and out.dtype == np.float32
    assert np.abs(out - want).max() < 3.0         # JPEG rounding of a flat colour


def _line_reader(path):
  with open(path, "rb") as f:
    for line in f:
      yield line.strip()


def _line_parser(raw, is_training, dtype, rng=None):
  v = int(raw)
  return np.full((2, 2, 1), v, dtype=np.float32), np.array([v], dtype=np.float32)


@pytest.mark.parametrize("workers", [0, 4])
def test_batches_are_cut_after_merging_the_worker_streams(tmp_path, workers):
  """47 records, batch 8, 4 workers: 5 full batches come out (47 // 8), every record at most
  once - not the 4 batches left when each worker drops its own remainder - and asking for
  more steps than there is data is a descriptive error, not a bare StopIteration."""
  from distributed_tensorflow_amd.data import pipeline
  files = []
  n = 0
  for i in range(4):
    p = tmp_path / ("part-%d" % i)
    cnt = 12 if i < 3 else 11
    p.write_text("".join("%d\n" % (n + j) for j in range(cnt)))
    n += cnt
    files.append(str(p))
  assert n == 47
  ds = pipeline.RecordDataset(files, _line_reader, _line_parser, is_training=False,
                              shuffle_buffer=0, num_epochs=1)
  it = pipeline.process_record_dataset(ds, False, 8, device=None,
                                       datasets_num_private_threads=workers, drop_remainder=True)
  batches = list(it)
  assert len(batches) == 47 // 8
  assert all(tuple(x.shape) == (8, 2, 2, 1) for x, _ in batches)
  seen = torch.cat([y.reshape(-1) for _, y in batches]).tolist()
  assert len(set(seen)) == len(seen) == 40
  # without drop_remainder the tail arrives as one short batch
  ds2 = pipeline.RecordDataset(files, _line_reader, _line_parser, is_training=False,
                               shuffle_buffer=0, num_epochs=1)
  sizes = [int(x.shape[0]) for x, _ in pipeline.process_record_dataset(
      ds2, False, 8, device=None, datasets_num_private_threads=workers, drop_remainder=False)]
  assert sizes == [8] * 5 + [7]


def test_evaluate_reports_exhausted_input():
  from distributed_tensorflow_amd import models
  from distributed_tensorflow_amd.train.model import Model
  from distributed_tensorflow_amd.train.optimizers import SGD
  m = Model(models.resnet20(), dtype="fp32")
  m.compile(SGD(0.1), "sparse_categorical_crossentropy", ["sparse_categorical_accuracy"])
  data = [(torch.zeros(2, 32, 32, 3), torch.zeros(2, 1))] * 2
  with pytest.raises(RuntimeError, match="ran out of data at step 2 of 3"):
    m.evaluate(data, steps=3)


def test_training_decode_uses_scaled_jpeg_decoding_for_large_photographs(monkeypatch):
  """A crop that is at least twice the output resolution is decoded at reduced scale
  (`Image.draft`, the inverse DCT emits the smaller image): same picture within resampling
  differences as the full-size decode, and `DTF_JPEG_DRAFT=0` decodes at full size."""
  import io
  import numpy as np
  from PIL import Image
  from distributed_tensorflow_amd.data import imagenet_preprocessing as ip
  yy, xx = np.mgrid[0:1200, 0:1600]
  pix = np.stack([(xx / 8) % 256, (yy / 6) % 256, ((xx + yy) / 12) % 256], -1).astype(np.uint8)
  buf = io.BytesIO()
  Image.fromarray(pix).save(buf, format="JPEG", quality=90)
  raw = buf.getvalue()
  whole = np.zeros((0, 4), np.float32)        # no boxes: the sampler may pick any crop

  class FullCrop:
    """Stands in for the random generator: forces the whole image as the crop, no flip."""
    def random(self):
      return 0.9
  monkeypatch.setattr(ip, "sample_distorted_bounding_box", lambda h, w, b, r, **k: (0, 0, h, w))
  img, box = ip._decode_for_crop(raw, FullCrop(), whole, 224, 224)
  assert img.size == (400, 300) and box == (0, 0, 300, 400)          # decoded at 1/4 scale
  a = ip.preprocess_image(raw, whole, 224, 224, 3, True, FullCrop())
  monkeypatch.setenv("DTF_JPEG_DRAFT", "0")
  img, box = ip._decode_for_crop(raw, FullCrop(), whole, 224, 224)
  assert img.size == (1600, 1200) and box == (0, 0, 1200, 1600)
  b = ip.preprocess_image(raw, whole, 224, 224, 3, True, FullCrop())
  assert a.shape == b.shape == (224, 224, 3)
  assert np.abs(a - b).mean() < 6.0            # of 255: same content, different resampling


def test_training_decode_inside_the_crop_box_matches_the_full_decode(monkeypatch):
  """The region decoder (`ops/csrc/jpeg_region.h`: only the crop box's rows and MCU-aligned
  columns are decoded, as `tf.image.decode_and_crop_jpeg` does for the reference) yields the
  pixels of a full PIL decode inside the box, so a training sample is the same either way;
  files it does not handle (here: PNG content) fall back to PIL with the same random numbers."""
  import io
  import numpy as np
  from PIL import Image
  from distributed_tensorflow_amd.data import imagenet_preprocessing as ip, native_loader
  if not native_loader.available() or not native_loader.ext().jpeg_region_available():
    pytest.skip("no libjpeg-turbo with partial decoding on this machine: %s"
                % (native_loader.ext().jpeg_region_status(),) if native_loader.available() else "")
  ext = native_loader.ext()
  rng0 = np.random.default_rng(3)
  small = rng0.integers(0, 256, size=(48, 64, 3), dtype=np.uint8)
  picture = Image.fromarray(small).resize((500, 375), Image.BICUBIC)
  buf = io.BytesIO()
  picture.save(buf, format="JPEG", quality=90)
  raw = buf.getvalue()
  assert ext.jpeg_header(raw) == (375, 500, 3)
  full = np.asarray(Image.open(io.BytesIO(raw)).convert("RGB"))
  for (y, x, h, w) in [(0, 0, 375, 500), (37, 101, 200, 250), (300, 434, 75, 66), (5, 9, 17, 33)]:
    pix, y0, x0, ih, iw = ext.jpeg_decode_region(raw, y, x, h, w, 1)
    assert (ih, iw) == (375, 500) and y0 == y and x0 <= x and x0 + pix.shape[1] >= x + w
    box = pix[:, x - x0:x - x0 + w].astype(int)
    want = full[y:y + h, x:x + w].astype(int)
    # identical except next to a cut that is not on an MCU boundary (chroma upsampling there
    # sees no neighbour): within a few levels in the outermost columns
    assert np.array_equal(box[:, 1:-1], want[:, 1:-1]) and np.abs(box - want).max() <= 24
  # half scale: the scaled image's size and a picture close to the full decode reduced by two
  pix, y0, x0, ih, iw = ext.jpeg_decode_region(raw, 10, 24, 100, 120, 2)
  assert (ih, iw) == (188, 250) and pix.shape[0] == 100
  ref = np.asarray(picture.resize((250, 188), Image.BILINEAR)).astype(int)
  assert np.abs(pix[:, 24 - x0:24 - x0 + 120].astype(int) - ref[10:110, 24:144]).mean() < 8.0
  # the whole training sample: same random numbers, region decoder against PIL
  bbox = np.array([[0.2, 0.1, 0.9, 0.8]], np.float32)
  for seed in range(6):
    a = ip.preprocess_image(raw, bbox, 224, 224, 3, True, np.random.default_rng(seed))
    monkeypatch.setenv("DTF_JPEG_REGION", "0")
    b = ip.preprocess_image(raw, bbox, 224, 224, 3, True, np.random.default_rng(seed))
    monkeypatch.delenv("DTF_JPEG_REGION")
    assert a.shape == b.shape == (224, 224, 3)
    assert np.abs(a - b).max() <= 24.0 and np.abs(a - b).mean() < 0.05
  # evaluation: the central crop of the resized image is decoded as a region as well
  for size in ((500, 375), (375, 500), (300, 1000), (224, 224)):
    b2 = io.BytesIO()
    Image.fromarray(small).resize(size, Image.BICUBIC).save(b2, format="JPEG", quality=90)
    a = ip.preprocess_image(b2.getvalue(), bbox, 224, 224, 3, False)
    monkeypatch.setenv("DTF_JPEG_REGION", "0")
    b = ip.preprocess_image(b2.getvalue(), bbox, 224, 224, 3, False)
    monkeypatch.delenv("DTF_JPEG_REGION")
    assert a.shape == b.shape == (224, 224, 3)
    assert np.abs(a - b).max() <= 24.0 and np.abs(a - b).mean() < 0.05, size
  # the resize of a part of the image (origin_y / origin_x) equals the resize of the whole
  whole = ext.crop_resize_bilinear(full, 0, 0, 375, 500, 256, 341, 16, 58, 224, 224, False, [1., 2., 3.])
  part = ext.crop_resize_bilinear(full[20:360, 80:430].copy(), 0, 0, 375, 500, 256, 341, 16, 58, 224, 224,
                                  False, [1., 2., 3.], 20, 80)
  assert np.array_equal(whole, part)
  with pytest.raises(ValueError):       # the window needs rows the part does not hold
    ext.crop_resize_bilinear(full[40:360, 80:430].copy(), 0, 0, 375, 500, 256, 341, 16, 58, 224, 224,
                             False, [1., 2., 3.], 40, 80)
  # not a JPEG: the header probe declines, PIL decodes
  png = io.BytesIO()
  picture.save(png, format="PNG")
  assert ext.jpeg_header(png.getvalue()) is None
  c = ip.preprocess_image(png.getvalue(), bbox, 224, 224, 3, True, np.random.default_rng(1))
  monkeypatch.setenv("DTF_JPEG_REGION", "0")
  d = ip.preprocess_image(png.getvalue(), bbox, 224, 224, 3, True, np.random.default_rng(1))
  assert np.array_equal(c, d)
  # a truncated file does not crash the worker: the decoder reports failure or pads like PIL
  assert ext.jpeg_decode_region(raw[:200], 0, 0, 10, 10, 1) is None


def test_rebatched_stream_reports_its_length_or_says_why_not():
  """`Model.evaluate(x, steps=None)` asks `len(x)`: a stream whose sample count is known answers,
  one without it raises an error that names the remedy (not a bare TypeError from `len`)."""
  from distributed_tensorflow_amd.data.pipeline import DevicePrefetcher, _Rebatcher
  batches = [(torch.zeros(4, 2), torch.zeros(4)), (torch.zeros(3, 2), torch.zeros(3))]
  rb = _Rebatcher(batches, 4, drop_remainder=False)
  pf = DevicePrefetcher(rb, torch.device("cpu"))
  with pytest.raises(TypeError, match="no known length"):
    len(pf)
  rb.num_samples = 7
  assert len(pf) == 2 and len(list(pf)) == 2
  rb.drop_remainder = True
  assert len(pf) == 1
