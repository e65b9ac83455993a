"""Builds the native extension `_dtf_native` for gfx950 with hipcc (no torch headers needed).

    python -m distributed_tensorflow_amd.ops.build [--force] [--jobs N]

Sources live in `ops/csrc`; objects go to `ops/csrc/build`, the shared object is written
next to this file so that it travels with the source tree.  hipcc cross-compiles for
gfx950 on a machine without a GPU.
"""
from __future__ import annotations

import argparse
import concurrent.futures
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
ARCH = "gfx950"
SOURCES = ["conv_igemm.hip", "conv_pipe.hip", "conv_halo.hip", "conv_rows.hip", "conv_wgrad_rows.hip", "conv_wgrad.hip", "bn.hip", "pool.hip", "loss.hip", "optim.hip", "dense.hip",
           "rccl_comm.cpp", "bindings.cpp"]
HEADERS = ["common.h", "conv_params.h", "conv_epilogue.h", "elementwise.h", "rccl_comm.h"]


def so_path() -> str:
  return os.path.join(HERE, "_dtf_native" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def _includes():
  import pybind11  # pylint: disable=import-outside-toplevel
  return ["-I" + sysconfig.get_paths()["include"], "-I" + pybind11.get_include(), "-I" + CSRC]


def _stale(target, deps):
  if not os.path.exists(target):
    return True
  t = os.path.getmtime(target)
  return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, force):
  obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
  deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS]
  if not force and not _stale(obj, deps):
    return obj, False
  cmd = ["hipcc", "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
         "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj] + _includes()
  r = subprocess.run(cmd, capture_output=True, text=True)
  if r.returncode != 0:
    raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
  return obj, True


def data_so_path() -> str:
  return os.path.join(HERE, "_dtf_data" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_data(force: bool = False, verbose: bool = True) -> str:
  """Host-only module `_dtf_data` (input pipeline): plain g++, no GPU code."""
  src = os.path.join(CSRC, "data_loader.cpp")
  target = data_so_path()
  if force or _stale(target, [src, os.path.join(CSRC, "jpeg_region.h")]):
    cmd = ["g++", "-O3", "-std=c++17", "-shared", "-fPIC", "-pthread", "-fvisibility=hidden",
           src, "-o", target, "-ldl"] + _includes()
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
      raise RuntimeError("g++ failed for data_loader.cpp:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
      print("built", target)
  elif verbose:
    print("up to date:", target)
  return target


def build(force: bool = False, jobs: int = 4, verbose: bool = True) -> str:
  os.makedirs(OBJ, exist_ok=True)
  build_data(force, verbose)
  with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
    results = list(ex.map(lambda s: _compile(s, force), SOURCES))
  objs = [o for o, _ in results]
  target = so_path()
  if force or any(c for _, c in results) or _stale(target, objs):
    cmd = ["hipcc", "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", target] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
      raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
      print("built", target)
  elif verbose:
    print("up to date:", target)
  return target


if __name__ == "__main__":
  ap = argparse.ArgumentParser()
  ap.add_argument("--force", action="store_true")
  ap.add_argument("--jobs", type=int, default=4)
  a = ap.parse_args()
  build(a.force, a.jobs)
