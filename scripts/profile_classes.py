#!/usr/bin/env python
"""Per kernel class: calls, L2 <-> memory traffic and serial time per training step, from the
tables `scripts/profile_step.sh` prints (its stdout saved to a file).

    python scripts/profile_classes.py gpurun_out/profile_step_tables.txt [steps in a counter run = 10]
"""
import collections
import re
import sys

CLASSES = [
    ("conv forward + data gradient (conv_pipe, conv_igemm, conv_rows, stem rows)",
     r"^conv_(pipe|igemm|rows|stem_rows|halo)"),
    ("weight gradient (conv_wgrad, conv_wgrad_rows + its summing launch)", r"^(conv_wgrad|wgrad_rows_reduce)"),
    ("bn_bwd_apply", r"^bn_bwd_apply"),
    ("bn_apply", r"^bn_apply"),
    ("bn_bwd_reduce", r"^bn_bwd_reduce"),
    ("stem BatchNorm + max-pool (bn_pool fwd + bwd)", r"^bn_pool"),
    ("bn finalize launches", r"finalize"),
    ("buffer fills at planning time (torch)", r"^at::|rocclr"),
    ("sgd, weight packing, input staging, loss, pooling, zero fills", r"."),
]


def cls(name):
  for label, pat in CLASSES:
    if re.search(pat, name):
      return label


def main():
  text = open(sys.argv[1]).read()
  steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
  parts = re.split(r"^== (.*) ==$", text, flags=re.M)
  sec = {parts[i]: parts[i + 1] for i in range(1, len(parts) - 1, 2)}
  calls, ms = collections.Counter(), collections.Counter()
  fetch, write = collections.Counter(), collections.Counter()
  for title, body in sec.items():
    for line in body.splitlines():
      m = re.match(r"\| `(.*)` \|(.*)\|$", line)
      if not m:
        continue
      c, f = cls(m.group(1)), [x.strip() for x in m.group(2).split("|")]
      if title.startswith("kernel statistics"):
        calls[c] += float(f[0])
        ms[c] += float(f[1])
      elif "FETCH_SIZE" in title:
        fetch[c] += float(f[2]) / steps / 1e6          # KB -> GB per step
      elif "WRITE_SIZE" in title:
        write[c] += float(f[2]) / steps / 1e6
  print("| kernel class | calls/step | fetch GB/step | write GB/step | ms/step (serial) |")
  print("|---|---:|---:|---:|---:|")
  for label, _ in CLASSES:
    print("| %s | %d | %.2f | %.2f | %.2f |" % (label, round(calls[label]), fetch[label], write[label], ms[label]))
  print("| **total** | | **%.1f** | **%.1f** | %.1f |" % (sum(fetch.values()), sum(write.values()), sum(ms.values())))
  once = "buffer fills at planning time (torch)"      # once per process, here spread over the profiled steps
  print("| **total of a steady-state step** (without the planning-time fills) | | **%.1f** | **%.1f** | %.1f |"
        % (sum(fetch.values()) - fetch[once], sum(write.values()) - write[once], sum(ms.values()) - ms[once]))


if __name__ == "__main__":
  main()
