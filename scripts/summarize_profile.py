#!/usr/bin/env python
"""Condenses rocprofv3 output into the small text tables kept under `profiles/`.

    python scripts/summarize_profile.py stats  <dir with *_kernel_stats.csv> <steps> > profiles/x.md
    python scripts/summarize_profile.py pmc    <dir with *_counter_collection.csv>  > profiles/y.md

`stats` groups the per-kernel totals of a `--kernel-trace --stats` run (templated kernels
keep their tile configuration) and divides by the number of profiled training steps; `pmc`
sums every collected counter per kernel over all dispatches.
"""
from __future__ import annotations

import collections
import csv
import glob
import os
import re
import sys


def _find(d, suffix):
  hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
  if not hits:
    sys.exit("no *%s under %s" % (suffix, d))
  return hits[0]


def _short(name: str) -> str:
  name = re.sub(r"^void ", "", name)
  name = name.replace("dtf::", "").replace("(anonymous namespace)::", "")
  name = re.sub(r"\(.*\)$", "", name)
  return name[:90]


def stats(d: str, steps: int):
  rows = list(csv.DictReader(open(_find(d, "_kernel_stats.csv"))))
  total = sum(float(r["TotalDurationNs"]) for r in rows)
  print("| kernel | calls/step | ms/step | avg us | % |")
  print("|---|---:|---:|---:|---:|")
  for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    t = float(r["TotalDurationNs"])
    if t / total < 0.0015:
      continue
    print("| `%s` | %.1f | %.3f | %.1f | %.1f |" % (
        _short(r["Name"]), int(r["Calls"]) / steps, t / steps / 1e6,
        float(r["AverageNs"]) / 1e3, 100 * t / total))
  print("\nGPU kernel time per step: %.2f ms over %d steps" % (total / steps / 1e6, steps))


def pmc(d: str):
  path = _find(d, "_counter_collection.csv")
  acc = collections.defaultdict(lambda: collections.defaultdict(float))
  calls = collections.Counter()
  seen = set()
  for r in csv.DictReader(open(path)):
    k = _short(r["Kernel_Name"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r.get("Dispatch_Id"), k)
    if key not in seen:
      seen.add(key)
      calls[k] += 1
  counters = sorted({c for v in acc.values() for c in v})
  print("| kernel | dispatches | " + " | ".join(counters) + " |")
  print("|---|---:|" + "---:|" * len(counters))
  for k in sorted(acc, key=lambda k: -calls[k]):
    print("| `%s` | %d | " % (k, calls[k]) +
          " | ".join("%.4g" % acc[k].get(c, 0.0) for c in counters) + " |")


if __name__ == "__main__":
  if len(sys.argv) < 3:
    sys.exit(__doc__)
  if sys.argv[1] == "stats":
    stats(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1)
  else:
    pmc(sys.argv[2])
