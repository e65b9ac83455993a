"""Per-layer kernel timing for a model: forward / data-gradient / weight-gradient of every
distinct convolution shape, with achieved TFLOP/s and GB/s against a simple roofline
(max(compute at 1.2 PF/s, bytes at 5 TB/s)).

    python scripts/bench_layers.py [resnet50|resnet56] [batch] [--default_cfg] [--fwd_cfg=N] ...

Every launch uses the tile configuration the training step uses for that shape (the measured
table `ops/tuned_gfx950.json`) and the forward launches accumulate the BatchNorm statistics of
their output as they do there; `--default_cfg` keeps the planner's untuned default, `--no_stats`
times the bare convolution.
"""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd import models
from distributed_tensorflow_amd.ops import autotune, native
from distributed_tensorflow_amd.ops.conv_plan import ConvGeom, round_up, pick_fwd_cfg, pick_wgrad_cfg

PEAK_TF = 1200.0
PEAK_GBS = 5000.0


def timeit(fn, iters=20, warm=3):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(iters):
    fn()
  e.record(); torch.cuda.synchronize()
  return s.elapsed_time(e) / iters * 1e3     # microseconds


def main():
  name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "resnet50"
  N = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 256
  overrides = {}
  tuned = "--default_cfg" not in sys.argv
  plain = "--no_stats" in sys.argv       # forward launches without the fused statistics
  pick = (lambda kind, d: autotune.apply(kind, d)) if tuned else (lambda kind, d: d)
  for a in sys.argv:
    if a.startswith("--fwd_cfg="): overrides["fwd"] = int(a.split("=")[1])
    if a.startswith("--wgrad_cfg="): overrides["wgrad"] = int(a.split("=")[1])
    if a.startswith("--splits="): overrides["splits"] = int(a.split("=")[1])
    # --sweep_fwd=0,6,16,17: time forward and data gradient with each tile configuration and
    # print the per-configuration times under the layer's row (best one marked)
    if a.startswith("--sweep_wgrad="): overrides["sweep_w"] = [int(c) for c in a.split("=")[1].split(",")]
    if a.startswith("--sweep_fwd="): overrides["sweep"] = [int(c) for c in a.split("=")[1].split(",")]
  spec = getattr(models, name)()
  dev = torch.device("cuda")
  shapes = collections.OrderedDict()
  first = True
  for op in spec.ops:
    if op.kind != "conv":
      continue
    a = op.attrs
    h, w, c = spec.shapes[op.inputs[0]]
    key = (h, w, c, a["kh"], a["cout"], a["stride"], tuple(a["pad"]), first)
    shapes.setdefault(key, []).append(op.name)
    first = False
  tot = collections.Counter()
  floor = collections.Counter()
  print("%-34s %3s %9s | %8s %7s %6s | %8s %7s | %8s %7s %4s | roofline us (f/d/w)" % (
      "shape (HxWxC k s ->Cout cfg f/d/w)", "n", "GMAC", "fwd us", "TF/s", "GB/s", "dgrad us", "TF/s", "wgrad us", "TF/s", "spl"))
  for key, names in shapes.items():
    h, w, c, k, cout, stride, pad, is_first = key
    merged = is_first and c <= 4 and stride % 2 == 0
    g = ConvGeom(N, h, w, c, k, k, cout, stride, pad, mode="merged" if merged else "normal")
    x = torch.randn(g.input_shape, device=dev).to(torch.bfloat16)
    wf = torch.randn(cout * g.k_fwd + 64, device=dev).to(torch.bfloat16) * 0.05
    out = torch.empty(g.output_shape, dtype=torch.bfloat16, device=dev)
    dy = torch.randn(g.output_shape, device=dev).to(torch.bfloat16)
    dw = torch.zeros(k, k, c, cout, device=dev)
    macs = N * g.OH * g.OW * cout * k * k * c
    # as in a training step: the forward launch accumulates the BatchNorm statistics of its output
    stats = torch.zeros(32 * 2 * cout, device=dev)
    fd = pick("conv", g.fwd(x.data_ptr(), wf.data_ptr(), out.data_ptr(),
                            stats=None if plain else stats.data_ptr(), stats_replicas=32))
    if "fwd" in overrides: fd["cfg"] = overrides["fwd"]
    t_f = timeit(lambda: native.run_op("conv", fd))
    bytes_f = (x.numel() + out.numel()) * 2
    if not is_first:
      wd = torch.randn(c * k * k * g.cout_pad + 64, device=dev).to(torch.bfloat16) * 0.05
      dx = torch.empty(N, h, w, c, dtype=torch.bfloat16, device=dev)
      dds = [pick("conv", d) for d in g.dgrad(dy.data_ptr(), wd.data_ptr(), dx.data_ptr())]
      if "fwd" in overrides:
        for d in dds: d["cfg"] = overrides["fwd"]
      t_d = timeit(lambda: [native.run_op("conv", d) for d in dds])
    else:
      t_d = 0.0
    wgd = pick("wgrad", g.wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr()))
    if "wgrad" in overrides: wgd["cfg"] = overrides["wgrad"]
    if "splits" in overrides: wgd["splits"] = overrides["splits"]
    if os.environ.get("DTF_WGRAD_PLAIN") == "1": wgd["debug_plain_store"] = 1
    t_w = timeit(lambda: native.run_op("wgrad", wgd))
    n = len(names)
    sweep_txt = ""
    if "sweep" in overrides:
      rows = []
      for sc in overrides["sweep"]:
        fd2 = dict(fd, cfg=sc)
        tf = timeit(lambda: native.run_op("conv", fd2), iters=10, warm=2)
        td = 0.0
        if not is_first:
          dd2 = [dict(d, cfg=sc) for d in dds]
          td = timeit(lambda: [native.run_op("conv", d) for d in dd2], iters=10, warm=2)
        rows.append((sc, tf, td))
      bf = min(rows, key=lambda r: r[1]); bd = min(rows, key=lambda r: r[2])
      sweep_txt = "    sweep fwd: " + " ".join("%d:%.1f%s" % (sc, tf, "*" if sc == bf[0] else "") for sc, tf, _ in rows)
      sweep_txt += "\n    sweep dgrad: " + " ".join("%d:%.1f%s" % (sc, td, "*" if sc == bd[0] else "") for sc, _, td in rows)
      tot["fwd_best"] += n * min(bf[1], t_f); tot["dgrad_best"] += n * (min(bd[2], t_d) if t_d else 0)
    rf = max(2 * macs / PEAK_TF / 1e6, bytes_f / PEAK_GBS / 1e3)
    print("%-34s %3d %9.2f | %8.1f %7.1f %6.0f | %8.1f %7.1f | %8.1f %7.1f %4d | %6.1f" % (
        "%dx%dx%d k%d s%d ->%d c%d/%s/%d" % (h, w, c, k, stride, cout, fd["cfg"],
                                             dds[0]["cfg"] if not is_first else "-", wgd["cfg"]), n, macs / 1e9,
        t_f, 2 * macs / t_f / 1e6, bytes_f / t_f / 1e3, t_d, (2 * macs / t_d / 1e6) if t_d else 0,
        t_w, 2 * macs / t_w / 1e6, wgd["splits"], rf))
    if "sweep_w" in overrides:
      rows = []
      for wc in overrides["sweep_w"]:
        for sp in sorted({wgd["splits"], max(1, wgd["splits"] // 2), wgd["splits"] * 2}):
          w2 = dict(wgd, cfg=wc, splits=sp)
          try:
            rows.append((wc, sp, timeit(lambda: native.run_op("wgrad", w2), iters=10, warm=2)))
          except Exception:
            pass
      bw = min(rows, key=lambda r: r[2])
      sweep_txt += ("\n" if sweep_txt else "") + "    sweep wgrad (cfg/splits:us): " + " ".join(
          "%d/%d:%.1f%s" % (wc, sp, t, "*" if (wc, sp) == bw[:2] else "") for wc, sp, t in rows)
      tot["wgrad_best"] += n * min(bw[2], t_w)
    if sweep_txt: print(sweep_txt, flush=True)
    tot["fwd"] += n * t_f; tot["dgrad"] += n * t_d; tot["wgrad"] += n * t_w
    floor["fwd"] += n * rf; floor["dgrad"] += n * (rf if t_d else 0); floor["wgrad"] += n * rf
  print("totals per step (ms): fwd %.2f dgrad %.2f wgrad %.2f  | roofline fwd %.2f dgrad %.2f wgrad %.2f" % (
      tot["fwd"] / 1e3, tot["dgrad"] / 1e3, tot["wgrad"] / 1e3, floor["fwd"] / 1e3, floor["dgrad"] / 1e3, floor["wgrad"] / 1e3))
  if "sweep_w" in overrides:
    print("best of sweep per layer (ms): wgrad %.2f" % (tot["wgrad_best"] / 1e3))
  if "sweep" in overrides:
    print("best of sweep per layer (ms): fwd %.2f dgrad %.2f" % (tot["fwd_best"] / 1e3, tot["dgrad_best"] / 1e3))


if __name__ == "__main__":
  main()
