#!/usr/bin/env python
"""Per-(kernel, grid) table from a rocprofv3 `--kernel-trace` CSV: which launch shapes of one
kernel template the time goes to (the statistics table merges them).

    python scripts/debug/kernel_shapes.py <dir with *_kernel_trace.csv> <steps> [name filter]
"""
from __future__ import annotations

import collections
import csv
import glob
import os
import re
import sys


def short(name: str) -> str:
  name = re.sub(r"^void ", "", name)
  name = name.replace("dtf::", "").replace("(anonymous namespace)::", "")
  name = re.sub(r"\(.*\)$", "", name)
  return name[:70]


def main():
  d, steps = sys.argv[1], int(sys.argv[2])
  flt = sys.argv[3] if len(sys.argv) > 3 else ""
  hits = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True))
  if not hits:
    sys.exit("no *_kernel_trace.csv under " + d)
  acc = collections.defaultdict(lambda: [0, 0.0])
  for r in csv.DictReader(open(hits[0])):
    k = short(r["Kernel_Name"])
    if flt and flt not in k:
      continue
    grid = "%sx%sx%s/%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    a = acc[(k, grid)]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
  print("%-72s %-22s %8s %9s %9s" % ("kernel", "grid/wg", "calls", "avg us", "us/step"))
  tot = 0.0
  for (k, g), (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-72s %-22s %8.1f %9.1f %9.1f" % (k, g, n / steps, t / n, t / steps))
    tot += t / steps
  print("total us/step: %.1f" % tot)


if __name__ == "__main__":
  main()
