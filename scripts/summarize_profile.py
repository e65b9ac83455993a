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
  """Per-kernel sums of every counter, plus what they say about the matrix cores and the LDS:

  * `MFMA busy %`: SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES).  The busy-cycle counter is
    kept per XCD, the matrix-core counter per compute unit (32 to an XCD); with that factor a
    kernel that does nothing but MFMAs with four waves on every SIMD reads 94 - 100 %
    (`scripts/microbench/mfma_peak.hip` under the same counters: the yardstick rows, 2.0 PFLOP/s
    sustained on non-zero operands for both bf16 shapes);
  * `LDS wait %`: SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES, the share of its resident time a wave
    spends waiting for an LDS instruction to return;
  * `TFLOP/s`: 512 * SQ_INSTS_VALU_MFMA_MOPS_BF16 (one count is 512 FLOP: pinned by the
    yardstick, whose FLOP count is known) over the summed durations of the kernel's dispatches,
    when the run also recorded timestamps (`--kernel-trace`)."""
  path = _find(d, "_counter_collection.csv")
  acc = collections.defaultdict(lambda: collections.defaultdict(float))
  calls = collections.Counter()
  dur = collections.defaultdict(float)
  seen = set()
  for r in csv.DictReader(open(path)):
    k = _short(r["Kernel_Name"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r.get("Dispatch_Id"), k)
    if key not in seen:
      seen.add(key)
      calls[k] += 1
      try:
        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
      except (KeyError, ValueError):
        pass
  # durations from the kernel trace of the same run, when the counter file has none
  if not any(dur.values()):
    try:
      trace = _find(d, "_kernel_trace.csv")
      for r in csv.DictReader(open(trace)):
        dur[_short(r["Kernel_Name"])] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    except SystemExit:
      pass
  counters = sorted({c for v in acc.values() for c in v})
  derived = []
  if {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"} <= set(counters):
    derived.append(("MFMA busy %", lambda v, k: 100.0 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) /
                    max(32.0 * v.get("SQ_BUSY_CYCLES", 0.0), 1.0)))
  if {"SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"} <= set(counters):
    derived.append(("LDS wait %", lambda v, k: 100.0 * v.get("SQ_WAIT_INST_LDS", 0.0) /
                    max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0)))
  if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in counters and any(dur.values()):
    derived.append(("TFLOP/s", lambda v, k: 512.0 * v.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) /
                    max(dur[k], 1.0) / 1e3))
  if any(dur.values()):
    derived.append(("ms total", lambda v, k: dur[k] / 1e6))
  print("| kernel | dispatches | " + " | ".join(n for n, _ in derived) + (" | " if derived else "") +
        " | ".join(counters) + " |")
  print("|---|---:|" + "---:|" * (len(counters) + len(derived)))
  for k in sorted(acc, key=lambda k: -(dur[k] or calls[k])):
    print("| `%s` | %d | " % (k, calls[k]) +
          "".join("%.1f | " % f(acc[k], k) for _, f in derived) +
          " | ".join("%.4g" % acc[k].get(c, 0.0) for c in counters) + " |")


if __name__ == "__main__":
  if len(sys.argv) < 3:
    sys.exit(__doc__)
  if sys.argv[1] == "stats":
    stats(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1)
  else:
    pmc(sys.argv[2])
