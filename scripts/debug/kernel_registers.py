"""Register and scratch use of every kernel in the built convolution code objects (CPU only):
unbundles the gfx950 code object of each listed .o under ops/csrc/build and prints the kernels
that spill (the twelve worst per object, `--all`: every one).
Usage: python scripts/debug/kernel_registers.py [--all] [conv_pipe conv_igemm ...]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
BUILD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..",
                     "distributed_tensorflow_amd", "ops", "csrc", "build")
ALL = "--all" in sys.argv
for o in [a for a in sys.argv[1:] if a != "--all"] or ["conv_pipe", "conv_igemm", "conv_wgrad", "bn"]:
  with tempfile.TemporaryDirectory() as d:
    fb, co = os.path.join(d, "fb"), os.path.join(d, "co")
    subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fb,
                    os.path.join(BUILD, o + ".o"), os.path.join(d, "s.o")], check=True)
    subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--unbundle", "--input=" + fb,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    t = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], check=True, capture_output=True,
                       text=True).stdout
  rows = []
  for b in t.split("- .agpr_count:")[1:]:
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, b).group(1))
    rows.append((g("private_segment_fixed_size"), g("vgpr_spill_count"), g("vgpr_count"),
                 int(b.split("\n")[0].strip()), re.search(r"\.name:\s+(\S+)", b).group(1)))
  bad = sorted((r for r in rows if r[0] or r[1]), reverse=True)
  print("%s: %d kernels, %d with scratch or spills, most registers %d"
        % (o, len(rows), len(bad), max(r[2] for r in rows)))
  for r in (bad if ALL else bad[:12]):
    print("   scratch %4d B  spilled %3d  vgpr %3d agpr %3d  %s" % (r[0], r[1], r[2], r[3], r[4][:100]))
