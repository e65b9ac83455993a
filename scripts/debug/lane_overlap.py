"""Occupancy of the two lanes over time from a rocprofv3 kernel trace (CSV): per training step,
the wall time, the time with no kernel running, with one running, with two or more running.

    python scripts/debug/lane_overlap.py <dir with *_kernel_trace.csv> [first kernel name substring]
"""
import csv, glob, sys
d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "preprocess_pad"
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
starts = [i for i, r in enumerate(rows) if marker in r[2]]
out = []
for a, b in zip(starts[:-1], starts[1:]):
  seg = rows[a:b]
  t0, t1 = seg[0][0], rows[b][0]
  ev = []
  for s, e, _ in seg:
    ev.append((s, 1)); ev.append((min(e, t1), -1))
  ev.sort()
  depth, last, acc = 0, t0, [0, 0, 0]
  for t, dlt in ev:
    acc[min(depth, 2)] += t - last
    last = t
    depth += dlt
  acc[min(depth, 2)] += t1 - last
  out.append(((t1 - t0) / 1e6, acc[0] / 1e6, acc[1] / 1e6, acc[2] / 1e6, sum(e - s for s, e, _ in seg) / 1e6))
for o in out[-5:]:
  print("step %.3f ms: idle %.3f, one kernel %.3f, two or more %.3f; sum of kernel durations %.3f" % o)
