"""TFRecord framing, CRC-32C and a minimal protobuf wire codec.

Enough of the TensorFlow on-disk formats to (a) read ImageNet shards written as
`tf.train.Example` records (reference `imagenet_preprocessing.py:156-223` parses
`image/encoded`, `image/class/label`, `image/object/bbox/*`) and (b) write
TensorBoard event files for `--enable_tensorboard` (`common.py:187-190`) without
TensorFlow or the tensorboard package.

Record framing: `uint64 length | uint32 masked_crc32c(length) | data |
uint32 masked_crc32c(data)`, little endian.
"""
from __future__ import annotations

import struct
from typing import BinaryIO, Dict, Iterator, List, Tuple, Union

__all__ = ["crc32c", "masked_crc32c", "write_record", "read_records", "encode_example",
           "decode_example", "encode_event_scalar", "encode_event_file_version",
           "decode_message"]

# -- CRC-32C (Castagnoli), table driven -------------------------------------------
_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
  _c = _i
  for _ in range(8):
    _c = (_c >> 1) ^ _POLY if _c & 1 else _c >> 1
  _TABLE.append(_c)


def crc32c(data: bytes, crc: int = 0) -> int:
  crc ^= 0xFFFFFFFF
  tbl = _TABLE
  for b in data:
    crc = tbl[(crc ^ b) & 0xFF] ^ (crc >> 8)
  return crc ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
  c = crc32c(data)
  return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def write_record(f: BinaryIO, data: bytes):
  hdr = struct.pack("<Q", len(data))
  f.write(hdr)
  f.write(struct.pack("<I", masked_crc32c(hdr)))
  f.write(data)
  f.write(struct.pack("<I", masked_crc32c(data)))


def read_records(f: BinaryIO, verify: bool = False) -> Iterator[bytes]:
  while True:
    hdr = f.read(8)
    if len(hdr) < 8:
      return
    (n,) = struct.unpack("<Q", hdr)
    (hcrc,) = struct.unpack("<I", f.read(4))
    if verify and hcrc != masked_crc32c(hdr):
      raise IOError("corrupt TFRecord length header")
    data = f.read(n)
    if len(data) < n:
      raise IOError("truncated TFRecord")
    (dcrc,) = struct.unpack("<I", f.read(4))
    if verify and dcrc != masked_crc32c(data):
      raise IOError("corrupt TFRecord payload")
    yield data


# -- protobuf wire format -----------------------------------------------------------
def _varint(n: int) -> bytes:
  if n < 0:
    n += 1 << 64
  out = bytearray()
  while True:
    b = n & 0x7F
    n >>= 7
    if n:
      out.append(b | 0x80)
    else:
      out.append(b)
      return bytes(out)


def _read_varint(buf: bytes, pos: int) -> Tuple[int, int]:
  shift = 0
  val = 0
  while True:
    b = buf[pos]
    pos += 1
    val |= (b & 0x7F) << shift
    if not b & 0x80:
      return val, pos
    shift += 7


def _key(field: int, wire: int) -> bytes:
  return _varint((field << 3) | wire)


def _ld(field: int, payload: bytes) -> bytes:
  return _key(field, 2) + _varint(len(payload)) + payload


def decode_message(buf: bytes) -> Dict[int, List[Union[int, bytes]]]:
  """Generic decode: field number -> list of raw values (ints or bytes)."""
  out: Dict[int, List[Union[int, bytes]]] = {}
  pos = 0
  n = len(buf)
  while pos < n:
    key, pos = _read_varint(buf, pos)
    field, wire = key >> 3, key & 7
    if wire == 0:
      v, pos = _read_varint(buf, pos)
    elif wire == 1:
      v = buf[pos:pos + 8]
      pos += 8
    elif wire == 2:
      ln, pos = _read_varint(buf, pos)
      v = buf[pos:pos + ln]
      pos += ln
    elif wire == 5:
      v = buf[pos:pos + 4]
      pos += 4
    else:
      raise ValueError("unsupported wire type %d" % wire)
    out.setdefault(field, []).append(v)
  return out


# tf.train.Example = { features(1): Features{ feature(1): map<string, Feature> } }
# Feature = oneof { bytes_list(1){value(1)}, float_list(2){value(1) packed}, int64_list(3){value(1) packed} }
def encode_example(features: Dict[str, Union[bytes, List[bytes], List[int], List[float]]]) -> bytes:
  entries = b""
  for name, val in features.items():
    if isinstance(val, (bytes, str)):
      val = [val]
    val = list(val)
    if val and isinstance(val[0], (bytes, str)):
      inner = b"".join(_ld(1, v.encode() if isinstance(v, str) else v) for v in val)
      feat = _ld(1, inner)
    elif val and isinstance(val[0], float):
      feat = _ld(2, _ld(1, b"".join(struct.pack("<f", v) for v in val)))
    else:
      feat = _ld(3, _ld(1, b"".join(_varint(int(v)) for v in val)))
    entry = _ld(1, name.encode()) + _ld(2, feat)
    entries += _ld(1, entry)
  return _ld(1, entries)


def decode_example(buf: bytes) -> Dict[str, list]:
  out: Dict[str, list] = {}
  top = decode_message(buf)
  for feats in top.get(1, []):
    for entry in decode_message(feats).get(1, []):
      e = decode_message(entry)
      name = e[1][0].decode()
      feat = decode_message(e[2][0])
      if 1 in feat:
        out[name] = list(decode_message(feat[1][0]).get(1, []))
      elif 2 in feat:
        vals = []
        for chunk in decode_message(feat[2][0]).get(1, []):
          if isinstance(chunk, bytes) and len(chunk) % 4 == 0:
            vals.extend(struct.unpack("<%df" % (len(chunk) // 4), chunk))
        out[name] = vals
      elif 3 in feat:
        vals = []
        for chunk in decode_message(feat[3][0]).get(1, []):
          if isinstance(chunk, int):
            vals.append(chunk)
          else:
            pos = 0
            while pos < len(chunk):
              v, pos = _read_varint(chunk, pos)
              vals.append(v if v < (1 << 63) else v - (1 << 64))
        out[name] = vals
      else:
        out[name] = []
  return out


# TensorBoard Event = { wall_time(1,double), step(2,int64), file_version(3,string),
#                       summary(5): Summary{ value(1): Value{ tag(1), simple_value(2,float) } } }
def encode_event_file_version(wall_time: float) -> bytes:
  return _key(1, 1) + struct.pack("<d", wall_time) + _ld(3, b"brain.Event:2")


def encode_event_scalar(wall_time: float, step: int, tag: str, value: float) -> bytes:
  val = _ld(1, tag.encode()) + _key(2, 5) + struct.pack("<f", float(value))
  summary = _ld(1, val)
  return (_key(1, 1) + struct.pack("<d", wall_time) + _key(2, 0) + _varint(int(step)) +
          _ld(5, summary))
