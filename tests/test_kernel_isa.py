"""Static checks on the gfx950 machine code of the convolution kernels (CPU only).

The convolution kernels fill their LDS tiles with direct-to-LDS loads (`global_load_lds`).
Nothing orders a `ds_read` behind such a load except the issuing wave's own `s_waitcnt vmcnt`
followed by a workgroup barrier, and the compiler's wait insertion is not a guarantee for this
instruction.  The kernels therefore state the wait in the source; this test pins the result in
the built code objects: in every instantiation each `s_barrier` is directly preceded by a
vector-memory wait (`vmcnt(0)` for the two-stage kernels, a counted `vmcnt(n)` for
the multi-stage pipeline), and the multi-stage kernels never drain the counter between two loads
of the loop with anything but their own counted waits.
"""
import os
import re
import shutil
import subprocess

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"
BUILD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                     "distributed_tensorflow_amd", "ops", "csrc", "build")
WINDOW = 8          # instructions before a barrier in which the wait must appear

pytestmark = pytest.mark.skipif(
    not (os.path.exists(os.path.join(LLVM, "llvm-objdump")) and shutil.which("hipcc")),
    reason="needs the ROCm LLVM tools")


def _disassemble(obj, tmp_path):
  if not os.path.exists(obj):
    from distributed_tensorflow_amd.ops import build
    build.build(verbose=False)
  fb = str(tmp_path / "fatbin")
  co = str(tmp_path / "code.co")
  subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fb, obj,
                  str(tmp_path / "stripped.o")], check=True)
  subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle",
                  "--input=" + fb, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co],
                 check=True)
  out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True,
                       capture_output=True, text=True).stdout
  funcs = {}
  name = None
  for line in out.splitlines():
    m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
    if m:
      name = m.group(1)
      funcs[name] = []
    elif name and line.startswith("\t"):
      ins = line.strip().split("//")[0].strip()
      if ins:
        funcs[name].append(ins)
  return funcs


def _loop_barriers(ins):
  """Indices of the barriers of a kernel that uses direct-to-LDS loads.  The compiler lays basic
  blocks out in any order, so barriers are not classified by position: every one of them has to
  carry the wait (those that only exchange accumulators have nothing in flight; the wait is free
  there)."""
  dma = [i for i, s in enumerate(ins) if s.startswith("global_load_lds")]
  bars = [i for i, s in enumerate(ins) if s.startswith("s_barrier")]
  return (bars if dma else []), bars


def _has_vm_wait(ins, b):
  """A vmcnt wait within WINDOW instructions before the barrier, or further back (a kernel that
  transforms its fragments in registers has some forty vector instructions between the wait and the
  barrier) with nothing but arithmetic in between: no memory instruction, branch or barrier."""
  for k, s in enumerate(reversed(ins[max(0, b - 8 * WINDOW):b])):
    if s.startswith("s_waitcnt") and "vmcnt(" in s:
      return True
    if k >= WINDOW and s.startswith(("global_", "buffer_", "flat_", "scratch_", "ds_", "s_cbranch", "s_branch",
                                     "s_barrier", "s_load", "s_endpgm")):
      return False
  return False


@pytest.mark.parametrize("obj,kernel", [("conv_igemm.o", "conv_igemm_kernel"),
                                        ("conv_wgrad.o", "conv_wgrad_kernel"),
                                        ("conv_pipe.o", "conv_pipe_kernel"),
                                        ("conv_rows.o", "conv_rows_kernel"),
                                        ("conv_wgrad_rows.o", "conv_wgrad_rows_kernel")])
def test_every_dma_barrier_waits_for_vector_memory(obj, kernel, tmp_path):
  funcs = _disassemble(os.path.join(BUILD, obj), tmp_path)
  kernels = {n: i for n, i in funcs.items() if kernel in n}
  assert len(kernels) >= (2 if "rows" in obj else 8), "expected the kernel instantiations in %s" % obj
  checked = 0
  for name, ins in kernels.items():
    loop_bars, _ = _loop_barriers(ins)
    assert loop_bars, "no barrier orders the DMA tiles of %s" % name
    for b in loop_bars:
      assert _has_vm_wait(ins, b), "%s: s_barrier at instruction %d without a vmcnt wait" % (name, b)
      checked += 1
  assert checked >= len(kernels)


def test_pipeline_kernels_keep_loads_in_flight(tmp_path):
  """The three- and four-stage instantiations wait with a non-zero count somewhere in the loop
  (that is the point of the ring), and only through the waits written in the source: the
  compiler adds no vmcnt(0) of its own between the first load and the last barrier."""
  funcs = _disassemble(os.path.join(BUILD, "conv_pipe.o"), tmp_path)
  deep = 0
  for name, ins in funcs.items():
    if "conv_pipe_kernel" not in name:
      continue
    m = re.search(r"conv_pipe_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name)
    stages = int(m.group(6))
    if int(m.group(8)) == 3:
      # the addend epilogue (one load wait per pixel block by design) is laid out in the middle of
      # the K loop's address range by the compiler; the loop itself is the same code as in the
      # other epilogue variants, which are checked
      continue
    loop_bars, _ = _loop_barriers(ins)
    dma = [i for i, s in enumerate(ins) if s.startswith("global_load_lds")]
    # the K loop: from the first direct-to-LDS load to the first barrier after the last one
    end = next(b for b in loop_bars if b > dma[-1]) if any(b > dma[-1] for b in loop_bars) else dma[-1]
    waits = [s for s in ins[dma[0]:end] if s.startswith("s_waitcnt") and "vmcnt(" in s]
    counts = sorted({int(re.search(r"vmcnt\((\d+)\)", s).group(1)) for s in waits})
    if stages >= 3:
      deep += 1
      assert any(c > 0 for c in counts), "%s never leaves a stage in flight (%s)" % (name, counts)
      # one wait per distinct depth of the drain ladder, nothing else
      assert len(waits) <= stages, "%s: unexpected vector-memory waits in the loop: %s" % (name, waits)
  assert deep >= 8


@pytest.mark.parametrize("obj", ["conv_igemm.o", "conv_pipe.o", "conv_halo.o", "conv_wgrad.o",
                                 "conv_rows.o", "conv_wgrad_rows.o"])
def test_no_convolution_kernel_spills_registers(obj, tmp_path):
  """Every instantiation of the convolution kernels fits its register budget: no scratch memory
  and no spilled vector register in the code object's kernel metadata.  (A fused epilogue that
  overflowed the 128 registers of the four-waves-per-SIMD tiles once spilled 72 - 91 of them to
  scratch and made the fusion slower than the sweep it replaced.)"""
  path = os.path.join(BUILD, obj)
  if not os.path.exists(path):
    from distributed_tensorflow_amd.ops import build
    build.build(verbose=False)
  fb, co = str(tmp_path / "fatbin"), str(tmp_path / "code.co")
  subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fb, path,
                  str(tmp_path / "stripped.o")], check=True)
  subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle",
                  "--input=" + fb, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co],
                 check=True)
  notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True,
                         capture_output=True, text=True).stdout
  kernels = notes.split("- .agpr_count:")[1:]
  assert len(kernels) >= (3 if "wgrad_rows" in obj else 8)
  bad = []
  for k in kernels:
    scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", k).group(1))
    spilled = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", k).group(1))
    if scratch or spilled:
      bad.append((re.search(r"\.name:\s+(\S+)", k).group(1), scratch, spilled))
  assert not bad, "%d kernels of %s use scratch: %s" % (len(bad), obj, bad[:4])
