"""Reads a rocprofv3 kernel trace (CSV) and prints, for the steady-state part of a run, the time
covered by kernels, the idle time between them and the mean kernel duration / gap:
  python scripts/debug/trace_gaps.py <dir with *_kernel_trace.csv> [kernels per step]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
  with open(f) as fh:
    for r in csv.DictReader(fh):
      rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[len(rows) // 2:]                     # second half: steady state
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for s, e, _ in rows[1:]:
  if s > cur_e:
    busy += cur_e - cur_s
    cur_s, cur_e = s, e
  else:
    cur_e = max(cur_e, e)
busy += cur_e - cur_s
dur = sum(e - s for s, e, _ in rows)
print("kernels %d  span %.3f ms  covered %.3f ms (%.1f %%)  idle %.3f ms  sum of durations %.3f ms"
      % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, dur / 1e6))
print("mean duration %.2f us, mean idle per kernel %.2f us" % (dur / len(rows) / 1e3, (span - busy) / len(rows) / 1e3))
by = {}
for s, e, n in rows:
  k = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
  a = by.setdefault(k, [0, 0])
  a[0] += 1
  a[1] += e - s
for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:24]:
  print("  %6d x %6.2f us  %s" % (n, t / n / 1e3, k))
