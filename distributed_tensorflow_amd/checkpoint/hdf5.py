"""Minimal pure-Python HDF5 writer and reader (no h5py / libhdf5 needed).

Implements the subset of the HDF5 file format (specification version 1.x structures) that
Keras `.h5` model files use, so checkpoints can be exchanged with `tf.keras` models:

* superblock version 0, 8-byte offsets and lengths;
* "old style" groups: object header v1 + symbol-table message, one level-0 v1 B-tree node,
  symbol-table nodes (`SNOD`) and a local heap (`HEAP`) for the link names;
* datasets: dataspace v1, datatype (IEEE float32/float64, signed/unsigned integers,
  fixed-length strings), fill-value v2, contiguous data layout v3;
* attributes (message v1) holding scalars, numeric arrays, fixed-length strings or arrays of
  fixed-length strings - the form `h5py` gives `numpy` `S` arrays such as Keras'
  `layer_names` / `weight_names`.

The reader additionally understands what h5py-written files contain beyond that: object
header continuation blocks, compact layout, variable-length strings in attributes (global
heap), multi-level group B-trees.  Chunked or compressed datasets and version-2 object
headers are rejected with a clear error.

Writer API:   `w = Writer(path); g = w.root.create_group("a"); g.create_dataset("x", arr);
              g.attrs["name"] = value; w.close()`
Reader API:   `f = File(path); f["a/x"][...] -> ndarray; f["a"].attrs; f["a"].keys()`
"""
from __future__ import annotations

import struct
from typing import Dict, Iterator, List, Optional, Tuple, Union

import numpy as np

__all__ = ["Writer", "File", "Group", "Dataset", "HDF5Error"]

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K = 64          # symbol-table node holds up to 2*LEAF_K entries
INTERNAL_K = 64      # B-tree node holds up to 2*INTERNAL_K children
OHDR_MSG_LIMIT = 65000


class HDF5Error(IOError):
  pass


def _pad8(b: bytes) -> bytes:
  return b + b"\x00" * (-len(b) % 8)


# ---------------------------------------------------------------------------------------
# datatype / dataspace encoding
# ---------------------------------------------------------------------------------------
def _encode_datatype(dt: np.dtype) -> bytes:
  dt = np.dtype(dt)
  if dt.kind == "f":
    size = dt.itemsize
    if size == 4:
      props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
      sign = 31
    elif size == 8:
      props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
      sign = 63
    elif size == 2:
      props = struct.pack("<HHBBBBI", 0, 16, 10, 5, 0, 10, 15)
      sign = 15
    else:
      raise HDF5Error("unsupported float size %d" % size)
    return struct.pack("<BBBBI", 0x11, 0x20, sign, 0, size) + props
  if dt.kind in "iu":
    bits0 = 0x08 if dt.kind == "i" else 0x00
    return struct.pack("<BBBBI", 0x10, bits0, 0, 0, dt.itemsize) + struct.pack(
        "<HH", 0, dt.itemsize * 8)
  if dt.kind == "b":
    return struct.pack("<BBBBI", 0x10, 0x00, 0, 0, 1) + struct.pack("<HH", 0, 8)
  if dt.kind == "S":
    # fixed-length string, null padded (what h5py writes for numpy 'S'), ASCII
    return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, max(dt.itemsize, 1))
  raise HDF5Error("unsupported dtype %s" % dt)


def _encode_dataspace(shape: Tuple[int, ...]) -> bytes:
  rank = len(shape)
  return struct.pack("<BBBB4x", 1, rank, 0, 0) + b"".join(struct.pack("<Q", int(d)) for d in shape)


def _as_array(value) -> np.ndarray:
  """Normalises attribute / dataset values to numpy arrays of a supported dtype."""
  if isinstance(value, str):
    value = value.encode("utf8")
  if isinstance(value, bytes):
    return np.array(value, dtype="S%d" % max(len(value), 1))
  if isinstance(value, (list, tuple)) and value and isinstance(value[0], (str, bytes)):
    items = [v.encode("utf8") if isinstance(v, str) else v for v in value]
    return np.array(items, dtype="S%d" % max(max(len(i) for i in items), 1))
  arr = np.asarray(value)
  if arr.dtype.kind == "U":
    arr = np.char.encode(arr, "utf8")
  if arr.dtype == np.bool_:
    arr = arr.astype(np.uint8)
  if arr.dtype.kind == "O":
    raise HDF5Error("object arrays are not supported")
  return arr


# ---------------------------------------------------------------------------------------
# writer
# ---------------------------------------------------------------------------------------
class _WGroup:
  def __init__(self, writer: "Writer", name: str):
    self._w = writer
    self.name = name
    self.attrs: Dict[str, object] = {}
    self._children: Dict[str, Union["_WGroup", "_WDataset"]] = {}

  def create_group(self, name: str) -> "_WGroup":
    node = self
    for part in [p for p in name.split("/") if p]:
      child = node._children.get(part)
      if child is None:
        child = _WGroup(self._w, part)
        node._children[part] = child
      elif not isinstance(child, _WGroup):
        raise HDF5Error("%s exists and is not a group" % part)
      node = child
    return node

  require_group = create_group

  def create_dataset(self, name: str, data) -> "_WDataset":
    parts = [p for p in name.split("/") if p]
    node = self.create_group("/".join(parts[:-1])) if len(parts) > 1 else self
    ds = _WDataset(parts[-1], _as_array(data))
    node._children[parts[-1]] = ds
    return ds

  def __getitem__(self, name: str):
    node = self
    for part in [p for p in name.split("/") if p]:
      node = node._children[part]
    return node

  def __contains__(self, name):
    return name in self._children

  def keys(self):
    return list(self._children)


class _WDataset:
  def __init__(self, name: str, data: np.ndarray):
    self.name = name
    self.data = np.ascontiguousarray(data)
    self.attrs: Dict[str, object] = {}


def _message(mtype: int, body: bytes, flags: int = 0) -> bytes:
  body = _pad8(body)
  if len(body) > OHDR_MSG_LIMIT:
    raise HDF5Error("object header message of %d bytes exceeds the 64 KB limit" % len(body))
  return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _attr_message(name: str, value) -> bytes:
  arr = _as_array(value)
  nm = name.encode("utf8") + b"\x00"
  dt = _encode_datatype(arr.dtype)
  ds = _encode_dataspace(arr.shape)
  body = struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(ds))
  body += _pad8(nm) + _pad8(dt) + _pad8(ds) + arr.tobytes()
  return _message(0x000C, body)


def _object_header(messages: List[bytes]) -> bytes:
  payload = b"".join(messages)
  return struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(payload)) + payload


class Writer:
  """Builds the tree in memory and serialises it on `close()`."""

  def __init__(self, path: str):
    self.path = path
    self.root = _WGroup(self, "/")
    self._closed = False

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    if exc[0] is None:
      self.close()

  # Layout pass: every object gets an address; containers reference children by address.
  def close(self):
    if self._closed:
      return
    buf = bytearray(b"\x00" * 96)          # superblock placeholder
    root_entry = self._write_group(buf, self.root)
    eof = len(buf)
    sb = SIGNATURE + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0)
    sb += struct.pack("<HHI", LEAF_K, INTERNAL_K, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += root_entry
    assert len(sb) == 96
    buf[0:96] = sb
    with open(self.path, "wb") as f:
      f.write(buf)
    self._closed = True

  @staticmethod
  def _alloc(buf: bytearray, data: bytes) -> int:
    buf.extend(b"\x00" * (-len(buf) % 8))
    addr = len(buf)
    buf.extend(data)
    return addr

  def _write_dataset(self, buf: bytearray, ds: _WDataset) -> int:
    raw = ds.data.tobytes()
    data_addr = self._alloc(buf, raw) if raw else UNDEF
    msgs = [
        _message(0x0001, _encode_dataspace(ds.data.shape)),
        _message(0x0003, _encode_datatype(ds.data.dtype), flags=1),
        _message(0x0005, struct.pack("<BBBB", 2, 2, 2, 0)),          # fill value v2, undefined
        _message(0x0008, struct.pack("<BBQQ", 3, 1, data_addr, len(raw))),
    ]
    for k, v in ds.attrs.items():
      msgs.append(_attr_message(k, v))
    return self._alloc(buf, _object_header(msgs))

  def _write_group(self, buf: bytearray, grp: _WGroup) -> bytes:
    """Writes the group; returns its 40-byte symbol-table entry (name offset left 0)."""
    names = sorted(grp._children, key=lambda s: s.encode("utf8"))
    if len(names) > 2 * LEAF_K * 2 * INTERNAL_K:
      raise HDF5Error("too many links in one group")
    child_entries: List[Tuple[str, int, bytes]] = []    # name, header address, scratch/cache
    for n in names:
      c = grp._children[n]
      if isinstance(c, _WGroup):
        entry = self._write_group(buf, c)
        child_entries.append((n, struct.unpack_from("<Q", entry, 8)[0], entry[16:]))
      else:
        addr = self._write_dataset(buf, c)
        child_entries.append((n, addr, struct.pack("<II16x", 0, 0)))
    # local heap with the names (offset 0 = empty string)
    heap = bytearray(b"\x00" * 8)
    offsets = {}
    for n in names:
      offsets[n] = len(heap)
      heap.extend(_pad8(n.encode("utf8") + b"\x00"))
    heap_data_addr = self._alloc(buf, bytes(heap))
    heap_addr = self._alloc(buf, b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), 1, heap_data_addr))
    # symbol table nodes
    per = 2 * LEAF_K
    snods: List[Tuple[int, int]] = []       # (address, heap offset of the last name)
    for i in range(0, max(len(names), 1), per):
      chunk = child_entries[i:i + per]
      body = b"SNOD" + struct.pack("<BBH", 1, 0, len(chunk))
      for n, addr, tail in chunk:
        body += struct.pack("<QQ", offsets[n], addr) + tail
      body += b"\x00" * (40 * (per - len(chunk)))
      a = self._alloc(buf, body)
      snods.append((a, offsets[chunk[-1][0]] if chunk else 0))
    # one level-0 B-tree node over the SNODs
    node = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(snods), UNDEF, UNDEF)
    node += struct.pack("<Q", 0)
    for a, last in snods:
      node += struct.pack("<QQ", a, last)
    full = 24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8
    node += b"\x00" * (full - len(node))
    btree_addr = self._alloc(buf, node)
    msgs = [_message(0x0011, struct.pack("<QQ", btree_addr, heap_addr))]
    for k, v in grp.attrs.items():
      msgs.append(_attr_message(k, v))
    hdr_addr = self._alloc(buf, _object_header(msgs))
    return struct.pack("<QQII", 0, hdr_addr, 1, 0) + struct.pack("<QQ", btree_addr, heap_addr)


# ---------------------------------------------------------------------------------------
# reader
# ---------------------------------------------------------------------------------------
class _Reader:
  def __init__(self, data: bytes):
    self.d = data
    if data[:8] != SIGNATURE:
      raise HDF5Error("not an HDF5 file")
    ver = data[8]
    if ver not in (0, 1):
      raise HDF5Error("superblock version %d is not supported (file written with a newer "
                      "libver); re-save with libver='earliest'" % ver)
    self.so, self.sl = data[13], data[14]
    if self.so != 8 or self.sl != 8:
      raise HDF5Error("only 8-byte offsets/lengths are supported")
    pos = 24 if ver == 0 else 28
    self.base = struct.unpack_from("<Q", data, pos)[0]
    pos += 32
    self.root_entry = data[pos:pos + 40]

  def u(self, fmt, off):
    return struct.unpack_from(fmt, self.d, off)

  # -- object headers ------------------------------------------------------------------
  def messages(self, addr: int) -> List[Tuple[int, bytes]]:
    d = self.d
    if d[addr:addr + 4] == b"OHDR":
      raise HDF5Error("version-2 object headers are not supported")
    ver, _, nmsgs, _, size = self.u("<BBHII", addr)
    if ver != 1:
      raise HDF5Error("bad object header version %d" % ver)
    blocks = [(addr + 16, size)]
    out: List[Tuple[int, bytes]] = []
    while blocks and len(out) < nmsgs:
      pos, left = blocks.pop(0)
      end = pos + left
      while pos + 8 <= end and len(out) < nmsgs:
        mtype, msize, _flags = self.u("<HHB", pos)
        body = d[pos + 8:pos + 8 + msize]
        pos += 8 + msize
        if mtype == 0x0010:                       # continuation
          off, ln = struct.unpack_from("<QQ", body, 0)
          blocks.append((off + self.base, ln))
        out.append((mtype, body))
    return out

  # -- groups -----------------------------------------------------------------------------
  def heap_name(self, heap_addr: int, off: int) -> str:
    if self.d[heap_addr:heap_addr + 4] != b"HEAP":
      raise HDF5Error("bad local heap")
    data_addr = self.u("<Q", heap_addr + 24)[0] + self.base
    end = self.d.index(b"\x00", data_addr + off)
    return self.d[data_addr + off:end].decode("utf8")

  def btree_entries(self, node: int, heap: int) -> Iterator[Tuple[str, int]]:
    d = self.d
    if d[node:node + 4] != b"TREE":
      raise HDF5Error("bad B-tree node")
    _ntype, level, used = self.u("<BBH", node + 4)
    pos = node + 24
    for i in range(used):
      child = self.u("<Q", pos + 8)[0] + self.base
      pos += 16
      if level > 0:
        yield from self.btree_entries(child, heap)
      else:
        if d[child:child + 4] != b"SNOD":
          raise HDF5Error("bad symbol table node")
        n = self.u("<H", child + 6)[0]
        for j in range(n):
          e = child + 8 + 40 * j
          name_off, ohdr = self.u("<QQ", e)
          yield self.heap_name(heap, name_off), ohdr + self.base

  # -- datatypes ----------------------------------------------------------------------------
  def decode_datatype(self, b: bytes):
    """Returns (numpy dtype or ('vlen_str',) marker, element size)."""
    cls = b[0] & 0x0F
    bits0 = b[1]
    size = struct.unpack_from("<I", b, 4)[0]
    order = ">" if (bits0 & 1) else "<"
    if cls == 0:
      signed = bool(bits0 & 0x08)
      return np.dtype("%s%s%d" % (order, "i" if signed else "u", size)), size
    if cls == 1:
      return np.dtype("%sf%d" % (order, size)), size
    if cls == 3:
      return np.dtype("S%d" % size), size
    if cls == 9:
      vtype = bits0 & 0x0F
      if vtype == 1:
        return ("vlen_str",), size
      raise HDF5Error("variable-length sequences are not supported")
    raise HDF5Error("unsupported datatype class %d" % cls)

  def decode_dataspace(self, b: bytes) -> Tuple[int, ...]:
    ver, rank, flags = b[0], b[1], b[2]
    if ver == 1:
      pos = 8
    elif ver == 2:
      pos = 4
      if b[3] == 2:      # null dataspace
        return (0,)
    else:
      raise HDF5Error("dataspace version %d" % ver)
    return tuple(struct.unpack_from("<Q", b, pos + 8 * i)[0] for i in range(rank))

  def global_heap_object(self, addr: int, index: int) -> bytes:
    d = self.d
    addr += self.base
    if d[addr:addr + 4] != b"GCOL":
      raise HDF5Error("bad global heap collection")
    csize = self.u("<Q", addr + 8)[0]
    pos = addr + 16
    end = addr + csize
    while pos + 16 <= end:
      idx, _ref, osize = self.u("<HH4xQ", pos)
      if idx == 0:
        break
      if idx == index:
        return d[pos + 16:pos + 16 + osize]
      pos += 16 + osize + (-osize % 8)
    raise HDF5Error("global heap object %d not found" % index)

  def decode_values(self, dtype, esize, shape, raw: bytes):
    n = int(np.prod(shape)) if shape else 1
    if isinstance(dtype, tuple):      # variable-length strings
      vals = []
      for i in range(n):
        ln, gaddr, gidx = struct.unpack_from("<IQI", raw, i * 16)
        vals.append(self.global_heap_object(gaddr, gidx)[:ln] if ln else b"")
      if not shape:
        return vals[0]
      return np.array(vals, dtype=object).reshape(shape)
    arr = np.frombuffer(raw[:n * esize], dtype=dtype).reshape(shape)
    if not shape:
      v = arr[()]
      return bytes(v) if dtype.kind == "S" else v
    return arr.copy()

  def decode_attribute(self, b: bytes):
    ver = b[0]
    if ver == 1:
      nsz, dsz, ssz = struct.unpack_from("<HHH", b, 2)
      pos = 8
      name = b[pos:pos + nsz].split(b"\x00")[0].decode("utf8")
      pos += nsz + (-nsz % 8)
      dt = b[pos:pos + dsz]
      pos += dsz + (-dsz % 8)
      sp = b[pos:pos + ssz]
      pos += ssz + (-ssz % 8)
    elif ver in (2, 3):
      nsz, dsz, ssz = struct.unpack_from("<HHH", b, 2)
      pos = 8 + (1 if ver == 3 else 0)
      name = b[pos:pos + nsz].split(b"\x00")[0].decode("utf8")
      pos += nsz
      dt = b[pos:pos + dsz]
      pos += dsz
      sp = b[pos:pos + ssz]
      pos += ssz
    else:
      raise HDF5Error("attribute message version %d" % ver)
    dtype, esize = self.decode_datatype(dt)
    shape = self.decode_dataspace(sp)
    return name, self.decode_values(dtype, esize, shape, b[pos:])


class _Node:
  def __init__(self, reader: _Reader, addr: int, name: str):
    self._r = reader
    self._addr = addr
    self.name = name
    self._msgs = reader.messages(addr)
    self._attrs: Optional[Dict[str, object]] = None

  @property
  def attrs(self) -> Dict[str, object]:
    if self._attrs is None:
      self._attrs = {}
      for t, body in self._msgs:
        if t == 0x000C:
          try:
            k, v = self._r.decode_attribute(body)
            self._attrs[k] = v
          except HDF5Error:
            continue          # attribute of an unsupported type: skip, keep the rest usable
    return self._attrs


class Dataset(_Node):
  def _meta(self):
    shape = dtype = None
    esize = 0
    layout = None
    for t, body in self._msgs:
      if t == 0x0001:
        shape = self._r.decode_dataspace(body)
      elif t == 0x0003:
        dtype, esize = self._r.decode_datatype(body)
      elif t == 0x0008:
        layout = body
    if shape is None or dtype is None or layout is None:
      raise HDF5Error("dataset %s lacks dataspace/datatype/layout" % self.name)
    return shape, dtype, esize, layout

  @property
  def shape(self):
    return self._meta()[0]

  @property
  def dtype(self):
    return self._meta()[1]

  def read(self) -> np.ndarray:
    shape, dtype, esize, layout = self._meta()
    n = int(np.prod(shape)) if shape else 1
    ver = layout[0]
    if ver == 3:
      cls = layout[1]
      if cls == 1:
        addr, size = struct.unpack_from("<QQ", layout, 2)
        raw = b"" if addr == UNDEF else self._r.d[addr + self._r.base:addr + self._r.base + size]
        if addr == UNDEF:
          raw = b"\x00" * (n * esize)
      elif cls == 0:
        size = struct.unpack_from("<H", layout, 2)[0]
        raw = layout[4:4 + size]
      else:
        raise HDF5Error("dataset %s uses chunked storage (compression/chunking is not "
                        "supported by this reader)" % self.name)
    elif ver in (1, 2):
      rank, cls = layout[1], layout[2]
      if cls != 1:
        raise HDF5Error("dataset %s: only contiguous layout is supported" % self.name)
      addr = struct.unpack_from("<Q", layout, 8)[0]
      raw = self._r.d[addr + self._r.base:addr + self._r.base + n * esize]
    else:
      raise HDF5Error("data layout version %d" % ver)
    out = self._r.decode_values(dtype, esize, shape, raw)
    return out

  def __getitem__(self, key):
    return self.read()[key]

  def __array__(self, dtype=None):
    a = self.read()
    return a.astype(dtype) if dtype is not None else a


class Group(_Node):
  def __init__(self, reader, addr, name, scratch: Optional[Tuple[int, int]] = None):
    super().__init__(reader, addr, name)
    self._links: Optional[Dict[str, int]] = None
    self._scratch = scratch

  def _load(self):
    if self._links is not None:
      return
    bt = hp = None
    for t, body in self._msgs:
      if t == 0x0011:
        bt, hp = struct.unpack_from("<QQ", body, 0)
      elif t in (0x0002, 0x0006):
        raise HDF5Error("new-style (link message) groups are not supported; write the file "
                        "with libver='earliest'")
    if bt is None and self._scratch:
      bt, hp = self._scratch
    self._links = {}
    if bt is None:
      return
    for name, ohdr in self._r.btree_entries(bt + self._r.base, hp + self._r.base):
      self._links[name] = ohdr

  def keys(self) -> List[str]:
    self._load()
    return list(self._links)

  def __contains__(self, name: str) -> bool:
    try:
      self[name]
      return True
    except KeyError:
      return False

  def __iter__(self):
    return iter(self.keys())

  def _child(self, name: str):
    self._load()
    if name not in self._links:
      raise KeyError(name)
    addr = self._links[name]
    msgs = self._r.messages(addr)
    full = (self.name.rstrip("/") + "/" + name)
    if any(t == 0x0011 for t, _ in msgs):
      return Group(self._r, addr, full)
    if any(t == 0x0008 for t, _ in msgs):
      return Dataset(self._r, addr, full)
    return Group(self._r, addr, full)

  def __getitem__(self, path: str):
    node = self
    for part in [p for p in path.split("/") if p]:
      if not isinstance(node, Group):
        raise KeyError(path)
      node = node._child(part)
    return node

  def visit_datasets(self, prefix: str = "") -> Iterator[Tuple[str, Dataset]]:
    for k in self.keys():
      c = self._child(k)
      p = prefix + k
      if isinstance(c, Group):
        yield from c.visit_datasets(p + "/")
      else:
        yield p, c


class File(Group):
  def __init__(self, path: str):
    with open(path, "rb") as f:
      data = f.read()
    r = _Reader(data)
    _name_off, ohdr, cache = struct.unpack_from("<QQI", r.root_entry, 0)
    scratch = None
    if cache == 1:
      scratch = struct.unpack_from("<QQ", r.root_entry, 24)
    super().__init__(r, ohdr + r.base, "/", scratch)

  def close(self):
    pass

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    return False
