// Python bindings and the launch-plan executor of the native engine.
//
// The Python side plans a model once (shapes, tile configurations, buffer addresses) and
// records every kernel launch of a training step into a `Plan`.  Running a step is then a
// single call that enqueues the recorded launches on a HIP stream with no per-kernel Python
// or dispatcher overhead; any contiguous range of the plan can also be captured into a
// hipGraph and replayed with one `hipGraphLaunch`.
//
// Every op is described by a (kind, args-dict) pair.  The same description either runs
// immediately (`run_op`, used by the unit tests and one-off calls) or is appended to a plan
// (`Plan.add`).  Tensor arguments are raw device addresses (Python ints from
// `tensor.data_ptr()`); the Python wrapper owns the tensors and keeps them alive.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdlib>
#include <functional>
#include <stdexcept>
#include <memory>
#include <string>
#include <vector>

#include "common.h"
#include "conv_params.h"
#include "elementwise.h"
#include "rccl_comm.h"

namespace py = pybind11;
using Launch = std::function<void(hipStream_t)>;

namespace {

template <typename T>
T get(const py::dict& d, const char* key) {
  if (!d.contains(key)) throw std::invalid_argument(std::string("missing argument '") + key + "'");
  return d[key].cast<T>();
}
template <typename T>
T get_or(const py::dict& d, const char* key, T dflt) {
  if (!d.contains(key) || d[key].is_none()) return dflt;
  return d[key].cast<T>();
}
template <typename T>
T* ptr(const py::dict& d, const char* key, bool required = true) {
  if (!d.contains(key) || d[key].is_none()) {
    if (required) throw std::invalid_argument(std::string("missing tensor '") + key + "'");
    return nullptr;
  }
  return reinterpret_cast<T*>(d[key].cast<uintptr_t>());
}

ConvParams conv_from_dict(const py::dict& d) {
  ConvParams p;
  p.dt = get_or<int>(d, "dt", DT_BF16);
  p.in = ptr<const void>(d, "in");
  p.wgt = ptr<const void>(d, "wgt");
  p.out = ptr<void>(d, "out");
  p.addend = ptr<const void>(d, "addend", false);
  p.addend_mask = ptr<const uint8_t>(d, "addend_mask", false);
  p.bias = ptr<const float>(d, "bias", false);
  p.scale = ptr<const float>(d, "scale", false);
  p.relu = get_or<int>(d, "relu", 0);
  p.stats = ptr<float>(d, "stats", false);
  p.stats_replicas = get_or<int>(d, "stats_replicas", 0);
  p.stats_y = ptr<const bf16_t>(d, "stats_y", false);
  p.stats_mask = ptr<const uint8_t>(d, "stats_mask", false);
  p.in_scale = ptr<const float>(d, "in_scale", false);
  p.in_shift = ptr<const float>(d, "in_shift", false);
  p.pro.sums = ptr<const float>(d, "pro_sums", false);
  if (p.pro.sums) {
    p.pro.replicas = get<int>(d, "pro_replicas");
    p.pro.gamma = ptr<const float>(d, "pro_gamma");
    p.pro.beta = ptr<const float>(d, "pro_beta");
    p.pro.moving_mean = ptr<float>(d, "pro_mm", false);
    p.pro.moving_var = ptr<float>(d, "pro_mv", false);
    p.pro.save_mean = ptr<float>(d, "pro_save_mean");
    p.pro.save_invstd = ptr<float>(d, "pro_save_invstd");
    p.pro.scale = ptr<float>(d, "pro_scale");
    p.pro.shift = ptr<float>(d, "pro_shift");
    p.pro.res = ptr<const void>(d, "pro_res", false);
    p.pro.z = ptr<void>(d, "pro_z", false);
    p.pro.mask = ptr<uint8_t>(d, "pro_mask", false);
    p.pro.P = get<int>(d, "pro_P");
    p.pro.eps = get<float>(d, "pro_eps");
    p.pro.momentum = get<float>(d, "pro_momentum");
    p.pro.relu = get_or<int>(d, "pro_relu", 0);
  }
  p.bpro.sums = ptr<const float>(d, "bpro_sums", false);
  if (p.bpro.sums) {
    p.bpro.replicas = get<int>(d, "bpro_replicas");
    p.bpro.y = ptr<const void>(d, "bpro_y");
    p.bpro.mask = ptr<const uint8_t>(d, "bpro_mask", false);
    p.bpro.mean = ptr<const float>(d, "bpro_mean");
    p.bpro.invstd = ptr<const float>(d, "bpro_invstd");
    p.bpro.gamma = ptr<const float>(d, "bpro_gamma");
    p.bpro.dgamma = ptr<float>(d, "bpro_dgamma");
    p.bpro.dbeta = ptr<float>(d, "bpro_dbeta");
    p.bpro.dy = ptr<void>(d, "bpro_dy");
    p.bpro.P = get<int>(d, "bpro_P");
  }
  p.N = get<int>(d, "N"); p.H = get<int>(d, "H"); p.W = get<int>(d, "W");
  p.Cpix = get<int>(d, "Cpix");
  p.OH = get<int>(d, "OH"); p.OW = get<int>(d, "OW");
  p.M = p.N * p.OH * p.OW;
  p.OHfull = get_or<int>(d, "OHfull", p.OH); p.OWfull = get_or<int>(d, "OWfull", p.OW);
  p.Cout = get<int>(d, "Cout");
  p.ldc = get_or<int>(d, "ldc", p.Cout);
  p.OSH = get_or<int>(d, "OSH", 1); p.OSW = get_or<int>(d, "OSW", 1);
  p.OOH = get_or<int>(d, "OOH", 0); p.OOW = get_or<int>(d, "OOW", 0);
  p.TI = get<int>(d, "TI"); p.TJ = get<int>(d, "TJ");
  p.Ctap = get<int>(d, "Ctap");
  p.Ktotal = p.TI * p.TJ * p.Ctap;
  p.ISH = get_or<int>(d, "ISH", 1); p.ISW = get_or<int>(d, "ISW", 1);
  p.dh0 = get_or<int>(d, "dh0", 0); p.dhs = get_or<int>(d, "dhs", 1);
  p.dw0 = get_or<int>(d, "dw0", 0); p.dws = get_or<int>(d, "dws", 1);
  p.ldw = get<int>(d, "ldw");
  p.wk0 = get_or<int>(d, "wk0", 0);
  p.wsi = get_or<int>(d, "wsi", p.TJ * p.Ctap);
  p.wsj = get_or<int>(d, "wsj", p.Ctap);
  p.rev = get_or<int>(d, "rev", 0);
  p.divOW = make_fastdiv((uint32_t)p.OW);
  p.divOH = make_fastdiv((uint32_t)p.OH);
  return p;
}

WgradParams wgrad_from_dict(const py::dict& d) {
  WgradParams p;
  p.dt = get_or<int>(d, "dt", DT_BF16);
  // "batch": device array of (x, dy, dw) pointer triples, "nbatch" layers of this geometry
  p.batch = reinterpret_cast<const WgradBatchItem*>(ptr<const void>(d, "batch", false));
  p.nbatch = get_or<int>(d, "nbatch", 1);
  p.x = ptr<const void>(d, "x", p.batch == nullptr);
  p.dy = ptr<const void>(d, "dy", p.batch == nullptr);
  p.dw = ptr<float>(d, "dw", p.batch == nullptr);
  p.N = get<int>(d, "N"); p.H = get<int>(d, "H"); p.W = get<int>(d, "W");
  p.Cpix = get<int>(d, "Cpix");
  p.OH = get<int>(d, "OH"); p.OW = get<int>(d, "OW");
  p.P = p.N * p.OH * p.OW;
  p.Cout = get<int>(d, "Cout");
  p.ldy = get_or<int>(d, "ldy", p.Cout);
  p.TI = get<int>(d, "TI"); p.TJ = get<int>(d, "TJ");
  p.Ctap = get<int>(d, "Ctap");
  p.Ktotal = p.TI * p.TJ * p.Ctap;
  p.ISH = get_or<int>(d, "ISH", 1); p.ISW = get_or<int>(d, "ISW", 1);
  p.dh0 = get_or<int>(d, "dh0", 0); p.dhs = get_or<int>(d, "dhs", 1);
  p.dw0 = get_or<int>(d, "dw0", 0); p.dws = get_or<int>(d, "dws", 1);
  p.Cpad = get_or<int>(d, "Cpad", p.Ctap);
  p.Creal = get_or<int>(d, "Creal", p.Cpad);
  p.TJpad = get_or<int>(d, "TJpad", p.TJ);
  p.TJreal = get_or<int>(d, "TJreal", p.TJpad);
  p.splits = get_or<int>(d, "splits", 1);
  p.debug_plain_store = get_or<int>(d, "debug_plain_store", 0);
  p.x_scale = ptr<const float>(d, "x_scale", false);
  p.x_shift = ptr<const float>(d, "x_shift", false);
  p.rev = get_or<int>(d, "rev", 0);
  int pps = (p.P + p.splits - 1) / p.splits;
  pps = ((pps + 31) / 32) * 32;
  p.pix_per_split = pps;
  p.divOW = make_fastdiv((uint32_t)p.OW);
  p.divOH = make_fastdiv((uint32_t)p.OH);
  p.divCpad = make_fastdiv((uint32_t)p.Cpad);
  p.divTJpad = make_fastdiv((uint32_t)p.TJpad);
  return p;
}

// Workspace of a whole-rows weight gradient (conv_wgrad_rows.hip).  An op recorded in a plan owns
// one for the plan's lifetime (ops of different plans and lanes may run at the same time; 75 MB
// each is small change in 288 GB); immediate launches share one, they run in stream order.
std::shared_ptr<float> wgrad_rows_workspace(bool planned) {
  auto alloc = [] {
    float* p = nullptr;
    HIP_CHECK(hipMalloc(&p, wgrad_rows_workspace_floats() * sizeof(float)));
    return std::shared_ptr<float>(p, [](float* q) { (void)hipFree(q); });
  };
  if (planned) return alloc();
  static std::shared_ptr<float> shared = alloc();
  return shared;
}

Launch make_op(const std::string& kind, const py::dict& d, bool planned = false) {
  // element type of the activation tensors of the op: 0 bf16 (default), 1 fp16, 2 fp32
  const int dt = get_or<int>(d, "dt", DT_BF16);
  if (kind == "conv") {
    const ConvParams p = conv_from_dict(d);
    const int cfg = get_or<int>(d, "cfg", 0);
    return [p, cfg](hipStream_t s) { launch_conv_igemm(p, cfg, s); };
  }
  if (kind == "wgrad") {
    WgradParams p = wgrad_from_dict(d);
    const int cfg = get_or<int>(d, "cfg", 0);
    if (wgrad_rows_serves(p, cfg) && !get_or<int>(d, "atomic_flush", 0)) {
      std::shared_ptr<float> ws = wgrad_rows_workspace(planned);
      p.ws = ws.get();
      return [p, cfg, ws](hipStream_t s) { launch_conv_wgrad(p, cfg, s); };
    }
    return [p, cfg](hipStream_t s) { launch_conv_wgrad(p, cfg, s); };
  }
  if (kind == "bn_fwd_train") {
    auto y = ptr<const void>(d, "y"); auto res = ptr<const void>(d, "res", false);
    auto z = ptr<void>(d, "z");
    auto gamma = ptr<const float>(d, "gamma"); auto beta = ptr<const float>(d, "beta");
    auto mm = ptr<float>(d, "mm"); auto mv = ptr<float>(d, "mv");
    auto smean = ptr<float>(d, "save_mean"); auto sinv = ptr<float>(d, "save_invstd");
    auto scale = ptr<float>(d, "scale"); auto shift = ptr<float>(d, "shift");
    auto ws = ptr<float>(d, "ws");
    auto mask = ptr<uint8_t>(d, "mask", false);
    // > 0: the producing convolution left that many replicas of [2][C] sums in `ws`
    const int fused = get_or<int>(d, "fused_replicas", 0);
    const int P = get<int>(d, "P"), C = get<int>(d, "C");
    const float eps = get<float>(d, "eps"), mom = get<float>(d, "momentum");
    const int relu = get<int>(d, "relu");
    // residual = input of another BatchNorm, normalised on the fly with that layer's constants
    auto rscale = ptr<const float>(d, "rscale", false); auto rshift = ptr<const float>(d, "rshift", false);
    const int apply = get_or<int>(d, "apply", 1);      // 0: statistics + finalize only
    const int rev = get_or<int>(d, "rev", 0);          // the apply sweep runs back to front
    return [=](hipStream_t s) {
      launch_bn_fwd_train(y, res, z, mask, gamma, beta, mm, mv, smean, sinv, scale, shift, ws,
                          fused, P, C, eps, mom, relu, rscale, rshift, apply, rev, dt, s);
    };
  }
  if (kind == "bn_pool_fwd_train" || kind == "bn_pool_bwd") {
    const int N = get<int>(d, "N"), H = get<int>(d, "H"), W = get<int>(d, "W"), C = get<int>(d, "C");
    const int OH = get<int>(d, "OH"), OW = get<int>(d, "OW"), k = get<int>(d, "k");
    const int st = get<int>(d, "stride"), pt = get<int>(d, "pt"), pl = get<int>(d, "pl");
    auto y = ptr<const void>(d, "y"); auto idx = ptr<uint8_t>(d, "idx");
    auto gamma = ptr<const float>(d, "gamma"); auto ws = ptr<float>(d, "ws");
    auto smean = ptr<float>(d, "save_mean"); auto sinv = ptr<float>(d, "save_invstd");
    if (kind == "bn_pool_fwd_train") {
      auto zp = ptr<void>(d, "zp"); auto beta = ptr<const float>(d, "beta");
      auto mm = ptr<float>(d, "mm"); auto mv = ptr<float>(d, "mv");
      auto scale = ptr<float>(d, "scale"); auto shift = ptr<float>(d, "shift");
      const int fused = get_or<int>(d, "fused_replicas", 0);
      const float eps = get<float>(d, "eps"), mom = get<float>(d, "momentum");
      return [=](hipStream_t s) {
        launch_bn_pool_fwd_train(y, zp, idx, gamma, beta, mm, mv, smean, sinv, scale, shift, ws,
                                 fused, N, H, W, C, OH, OW, k, st, pt, pl, eps, mom, dt, s);
      };
    }
    auto dzp = ptr<const void>(d, "dzp"); auto dy = ptr<void>(d, "dy");
    auto dgamma = ptr<float>(d, "dgamma"); auto dbeta = ptr<float>(d, "dbeta");
    return [=](hipStream_t s) {
      launch_bn_pool_bwd(dzp, idx, y, smean, sinv, gamma, dgamma, dbeta, dy, ws, N, H, W, C, OH, OW,
                         k, st, pt, pl, dt, s);
    };
  }
  if (kind == "bn_fold") {
    auto params = ptr<const float>(d, "params"); auto state = ptr<const float>(d, "state");
    auto out = ptr<float>(d, "out"); auto descs = ptr<const uint8_t>(d, "descs");
    const int n = get<int>(d, "nlayers");
    return [=](hipStream_t s) { launch_bn_fold(params, state, out, descs, n, s); };
  }
  if (kind == "bn_infer") {
    auto y = ptr<const void>(d, "y"); auto res = ptr<const void>(d, "res", false);
    auto z = ptr<void>(d, "z");
    auto gamma = ptr<const float>(d, "gamma"); auto beta = ptr<const float>(d, "beta");
    auto mm = ptr<const float>(d, "mm"); auto mv = ptr<const float>(d, "mv");
    const int P = get<int>(d, "P"), C = get<int>(d, "C");
    const float eps = get<float>(d, "eps");
    const int relu = get<int>(d, "relu");
    return [=](hipStream_t s) { launch_bn_infer(y, res, z, gamma, beta, mm, mv, P, C, eps, relu, dt, s); };
  }
  if (kind == "bn_bwd") {
    auto dz = ptr<const void>(d, "dz"); auto y = ptr<const void>(d, "y");
    auto z = ptr<const void>(d, "z", false);
    auto mask = ptr<const uint8_t>(d, "mask", false);
    auto mean = ptr<const float>(d, "mean"); auto inv = ptr<const float>(d, "invstd");
    auto gamma = ptr<const float>(d, "gamma");
    auto dgamma = ptr<float>(d, "dgamma"); auto dbeta = ptr<float>(d, "dbeta");
    auto dy = ptr<void>(d, "dy"); auto dres = ptr<void>(d, "dres", false);
    auto ws = ptr<float>(d, "ws");
    const int P = get<int>(d, "P"), C = get<int>(d, "C");
    // > 0: `ws` already holds that many replica rows of [sum g | sum g*y] left by the data-
    // gradient launches that produced dz; the reduction sweep is skipped
    const int pre = get_or<int>(d, "pre_replicas", 0);
    auto tail = ptr<float>(d, "tail", false);
    // direction of the two sweeps: bit 0 reduction, bit 1 apply run back to front
    const int rev = get_or<int>(d, "rev_reduce", 0) | (get_or<int>(d, "rev_apply", 0) << 1);
    // ReLU mask from the sign of y * relu_scale + relu_shift instead of mask bits / z
    auto rsc = ptr<const float>(d, "relu_scale", false); auto rsh = ptr<const float>(d, "relu_shift", false);
    return [=](hipStream_t s) {
      launch_bn_bwd(dz, y, z, mask, mean, inv, gamma, dgamma, dbeta, dy, dres, ws, P, C, pre, tail, rev, rsc, rsh,
                    dt, s);
    };
  }
  if (kind == "bn_bwd_dual") {
    auto dz = ptr<const void>(d, "dz"); auto mask = ptr<const uint8_t>(d, "mask");
    auto ws = ptr<float>(d, "ws"); auto ws2 = ptr<float>(d, "ws2");
    const int P = get<int>(d, "P"), C = get<int>(d, "C");
    auto y1 = ptr<const void>(d, "y"); auto y2 = ptr<const void>(d, "y2");
    auto m1 = ptr<const float>(d, "mean"); auto m2 = ptr<const float>(d, "mean2");
    auto i1 = ptr<const float>(d, "invstd"); auto i2 = ptr<const float>(d, "invstd2");
    auto g1 = ptr<const float>(d, "gamma"); auto g2 = ptr<const float>(d, "gamma2");
    auto dg1 = ptr<float>(d, "dgamma"); auto dg2 = ptr<float>(d, "dgamma2");
    auto db1 = ptr<float>(d, "dbeta"); auto db2 = ptr<float>(d, "dbeta2");
    auto dy1 = ptr<void>(d, "dy"); auto dy2 = ptr<void>(d, "dy2");
    auto tail = ptr<float>(d, "tail", false);
    const int rev = get_or<int>(d, "rev_reduce", 0) | (get_or<int>(d, "rev_apply", 0) << 1);
    return [=](hipStream_t s) {
      launch_bn_bwd_dual(dz, mask, P, C, ws, ws2, y1, m1, i1, g1, dg1, db1, dy1, y2, m2, i2, g2, dg2,
                         db2, dy2, tail, rev, dt, s);
    };
  }
  if (kind == "mask_apply") {
    auto x = ptr<const void>(d, "x"); auto mask = ptr<const uint8_t>(d, "mask");
    auto out = ptr<void>(d, "out");
    const size_t n = get<size_t>(d, "n");
    return [=](hipStream_t s) { launch_mask_apply(x, mask, out, n, dt, s); };
  }
  if (kind == "maxpool_fwd" || kind == "maxpool_bwd") {
    const int N = get<int>(d, "N"), H = get<int>(d, "H"), W = get<int>(d, "W"), C = get<int>(d, "C");
    const int OH = get<int>(d, "OH"), OW = get<int>(d, "OW"), k = get<int>(d, "k");
    const int st = get<int>(d, "stride"), pt = get<int>(d, "pt"), pl = get<int>(d, "pl");
    auto idx = ptr<uint8_t>(d, "idx");
    if (kind == "maxpool_fwd") {
      auto x = ptr<const void>(d, "x"); auto out = ptr<void>(d, "out");
      return [=](hipStream_t s) { launch_maxpool_fwd(x, out, idx, N, H, W, C, OH, OW, k, st, pt, pl, dt, s); };
    }
    auto dy = ptr<const void>(d, "dy"); auto dx = ptr<void>(d, "dx");
    return [=](hipStream_t s) { launch_maxpool_bwd(dy, idx, dx, N, H, W, C, OH, OW, k, st, pt, pl, dt, s); };
  }
  if (kind == "mean_fwd") {
    auto x = ptr<const void>(d, "x"); auto out = ptr<void>(d, "out");
    const int N = get<int>(d, "N"), HW = get<int>(d, "HW"), C = get<int>(d, "C");
    return [=](hipStream_t s) { launch_global_mean_fwd(x, out, N, HW, C, dt, s); };
  }
  if (kind == "mean_bwd") {
    auto dy = ptr<const void>(d, "dy"); auto dx = ptr<void>(d, "dx");
    const int N = get<int>(d, "N"), HW = get<int>(d, "HW"), C = get<int>(d, "C");
    return [=](hipStream_t s) { launch_global_mean_bwd(dy, dx, N, HW, C, dt, s); };
  }
  if (kind == "softmax_xent") {
    auto logits = ptr<const void>(d, "logits"); auto labels = ptr<const int32_t>(d, "labels");
    auto dl = ptr<void>(d, "dlogits", false); auto acc = ptr<float>(d, "accum");
    const int N = get<int>(d, "N"), C = get<int>(d, "C"), ld = get<int>(d, "ld");
    // grad_scale may change per step (dynamic loss scale): an optional device scalar carries it.
    const float gs = get_or<float>(d, "grad_scale", 1.f);
    auto gsd = ptr<const float>(d, "grad_scale_dev", false);   // multiplies grad_scale
    // accum[0] += loss * loss_weight: 1 (the default) accumulates the sum, 1 / N the batch mean
    const float lw = get_or<float>(d, "loss_weight", 1.f);
    return [=](hipStream_t s) { launch_softmax_xent(logits, labels, dl, acc, N, C, ld, gs, gsd, lw, dt, s); };
  }
  if (kind == "softmax_probs") {
    auto logits = ptr<const void>(d, "logits"); auto probs = ptr<float>(d, "probs");
    const int N = get<int>(d, "N"), C = get<int>(d, "C"), ld = get<int>(d, "ld");
    return [=](hipStream_t s) { launch_softmax_probs(logits, probs, N, C, ld, dt, s); };
  }
  if (kind == "colsum") {
    auto x = ptr<const void>(d, "x"); auto out = ptr<float>(d, "out");
    const int N = get<int>(d, "N"), C = get<int>(d, "C"), ld = get<int>(d, "ld");
    const int slices = get_or<int>(d, "slices", 0);
    return [=](hipStream_t s) { launch_colsum(x, out, N, C, ld, dt, slices, s); };
  }
  if (kind == "sgd") {
    auto p = ptr<float>(d, "p"); auto g = ptr<const float>(d, "g"); auto v = ptr<float>(d, "v");
    auto l2 = ptr<const float>(d, "l2x2");
    const size_t n = get<size_t>(d, "n");
    const float lr = get_or<float>(d, "lr", 0.f), mom = get_or<float>(d, "momentum", 0.f);
    const float gs = get_or<float>(d, "grad_scale", 1.f);
    auto hyper = ptr<const float>(d, "hyper", false);   // device [lr, momentum, grad_scale]
    auto skip = ptr<const float>(d, "skip", false);     // device [skip flag, inverse loss scale]
    const int um = get_or<int>(d, "use_momentum", mom != 0.f ? 1 : 0);
    return [=](hipStream_t s) { launch_sgd_momentum_l2(p, g, v, l2, n, lr, mom, gs, um, hyper, skip, s); };
  }
  if (kind == "dense_fwd") {
    auto x = ptr<const void>(d, "x"); auto w = ptr<const float>(d, "w");
    auto bias = ptr<const float>(d, "bias", false); auto out = ptr<void>(d, "out");
    const int N = get<int>(d, "N"), K = get<int>(d, "K"), C = get<int>(d, "Cout");
    const int ldx = get_or<int>(d, "ldx", K), ldo = get_or<int>(d, "ldo", C);
    return [=](hipStream_t s) { launch_dense_fwd(x, w, bias, out, N, K, C, ldx, ldo, dt, s); };
  }
  if (kind == "dense_wgrad") {
    auto x = ptr<const void>(d, "x"); auto dy = ptr<const void>(d, "dy"); auto dw = ptr<float>(d, "dw");
    const int N = get<int>(d, "N"), K = get<int>(d, "K"), C = get<int>(d, "Cout");
    const int ldx = get_or<int>(d, "ldx", K), ldy = get_or<int>(d, "ldy", C);
    return [=](hipStream_t s) { launch_dense_wgrad(x, dy, dw, N, K, C, ldx, ldy, dt, s); };
  }
  if (kind == "dense_dgrad") {
    auto dy = ptr<const void>(d, "dy"); auto w = ptr<const float>(d, "w"); auto dx = ptr<void>(d, "dx");
    const int N = get<int>(d, "N"), K = get<int>(d, "K"), C = get<int>(d, "Cout");
    const int ldy = get_or<int>(d, "ldy", C), ldx = get_or<int>(d, "ldx", K);
    return [=](hipStream_t s) { launch_dense_dgrad(dy, w, dx, N, K, C, ldy, ldx, dt, s); };
  }
  if (kind == "nonfinite_flag") {
    auto g = ptr<const float>(d, "g"); auto flag = ptr<unsigned int>(d, "flag");
    const size_t n = get<size_t>(d, "n");
    return [=](hipStream_t s) { launch_nonfinite_flag(g, n, flag, s); };
  }
  if (kind == "loss_scale_update") {
    auto state = ptr<float>(d, "state"); auto flag = ptr<unsigned int>(d, "flag");
    auto lsd = ptr<float>(d, "loss_scale_dev", false);
    const float growth = get_or<float>(d, "growth", 2000.f), mult = get_or<float>(d, "multiplier", 2.f);
    return [=](hipStream_t s) { launch_loss_scale_update(state, flag, lsd, growth, mult, s); };
  }
  if (kind == "pack_weights") {
    auto params = ptr<const float>(d, "params"); auto packed = ptr<void>(d, "packed");
    auto descs = ptr<const PackDesc>(d, "descs");
    const int nd = get<int>(d, "ndesc"), nb = get<int>(d, "nblocks");
    return [=](hipStream_t s) { launch_pack_weights(params, packed, descs, nd, nb, dt, s); };
  }
  if (kind == "preprocess_pad") {
    auto in = ptr<const float>(d, "in"); auto out = ptr<void>(d, "out");
    const int N = get<int>(d, "N"), H = get<int>(d, "H"), W = get<int>(d, "W");
    const int Cin = get<int>(d, "Cin"), pt = get<int>(d, "pt"), pl = get<int>(d, "pl");
    const int OH = get<int>(d, "OH"), OW = get<int>(d, "OW"), Cpad = get<int>(d, "Cpad");
    return [=](hipStream_t s) { launch_preprocess_pad(in, out, N, H, W, Cin, pt, pl, OH, OW, Cpad, dt, s); };
  }
  if (kind == "cast_bf16" || kind == "cast") {      // fp32 -> the element type `dt`
    auto in = ptr<const float>(d, "in"); auto out = ptr<void>(d, "out");
    const size_t n = get<size_t>(d, "n");
    return [=](hipStream_t s) { launch_cast_from_f32(in, out, n, dt, s); };
  }
  if (kind == "allreduce") {
    // a collective as a plan entry: `comm` is the address of a live RcclComm (kept alive by its
    // Python owner); mode 0 one-shot, 1 reduce-scatter + all-gather, 2 by message size, 3 all-link mesh exchange
    auto* comm = reinterpret_cast<dtf::RcclComm*>(get<uintptr_t>(d, "comm"));
    auto buf = ptr<void>(d, "buf");
    const size_t count = get<size_t>(d, "count");
    const int mode = get_or<int>(d, "mode", 2);
    if (!comm) throw std::invalid_argument("allreduce: null communicator");
    return [=](hipStream_t s) {
      if (mode == 0) comm->all_reduce(buf, count, dt, s);
      else if (mode == 1) comm->all_reduce_direct(buf, count, dt, s);
      else if (mode == 3) comm->all_reduce_mesh(buf, count, dt, s);
      else comm->all_reduce_auto(buf, count, dt, s);
    };
  }
  if (kind == "memset") {
    auto p = ptr<uint8_t>(d, "ptr");
    const size_t n = get<size_t>(d, "nbytes");
    return [=](hipStream_t s) { launch_zero(p, n, s); };
  }
  throw std::invalid_argument("unknown op kind '" + kind + "'");
}

void check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) throw std::runtime_error(std::string("kernel launch failed: ") + hipGetErrorString(e));
}

// ---- streams that really run next to each other ------------------------------------------------
// HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, round-robin at stream
// creation).  Two streams that land on the same queue execute strictly in submission order:
// a side lane created after a communication library has taken its streams was measured to
// share the main stream's queue, which turned the overlapped step (23.2 ms) into a serial one
// with extra event traffic (24.4 ms).  So a lane is not just created, it is *probed*: a short
// spin kernel runs on the main stream and on the candidate at the same time, and the candidate
// is kept only if the pair takes about one spin, not two.
__global__ void spin_kernel(unsigned long long ticks, unsigned int* sink) {
  const unsigned long long t0 = wall_clock64();
  unsigned int it = 0;
  while (wall_clock64() - t0 < ticks && it < (1u << 20)) ++it;     // bounded: always exits
  if (sink != nullptr && it == 0xffffffffu) *sink = it;
}

class LaneRegistry {
 public:
  // A stream that executes concurrently with `main` and with every lane handed out before.
  // Lanes live as long as the process.
  hipStream_t pick(hipStream_t main, int priority) {
    std::vector<hipStream_t> peers(lanes_);
    peers.push_back(main);
    std::vector<hipStream_t> rejected;
    hipStream_t chosen = nullptr;
    last_probe_.clear();
    for (int attempt = 0; attempt < 12 && !chosen; ++attempt) {
      hipStream_t c = nullptr;
      HIP_CHECK(hipStreamCreateWithPriority(&c, hipStreamNonBlocking, priority));
      bool ok = true;
      for (hipStream_t peer : peers) {
        const float ratio = overlap_ratio(peer, c);
        last_probe_.push_back(ratio);
        if (ratio > 1.5f) { ok = false; break; }
      }
      if (ok) chosen = c; else rejected.push_back(c);
    }
    if (!chosen) {                       // every candidate shares a queue with someone: keep one
      chosen = rejected.back();
      rejected.pop_back();
    }
    for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
    lanes_.push_back(chosen);
    return chosen;
  }
  const std::vector<float>& last_probe() const { return last_probe_; }

 private:
  // (time of two simultaneous spins) / (time of one): ~1 when concurrent, ~2 when serialised.
  static float overlap_ratio(hipStream_t a, hipStream_t b) {
    int dev = 0, khz = 0;
    HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    if (khz <= 0) khz = 100000;
    const float spin_ms = 0.25f;
    const unsigned long long ticks = (unsigned long long)(spin_ms * (float)khz);
    hipEvent_t e0, ea, eb;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&ea));
    HIP_CHECK(hipEventCreate(&eb));
    float worst = 0.f;
    for (int trial = 0; trial < 2; ++trial) {
      HIP_CHECK(hipEventRecord(e0, a));
      HIP_CHECK(hipStreamWaitEvent(b, e0, 0));
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, ticks, (unsigned int*)nullptr);
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, ticks, (unsigned int*)nullptr);
      HIP_CHECK(hipEventRecord(ea, a));
      HIP_CHECK(hipEventRecord(eb, b));
      HIP_CHECK(hipEventSynchronize(ea));
      HIP_CHECK(hipEventSynchronize(eb));
      float ta = 0.f, tb = 0.f;
      HIP_CHECK(hipEventElapsedTime(&ta, e0, ea));
      HIP_CHECK(hipEventElapsedTime(&tb, e0, eb));
      const float r = (ta > tb ? ta : tb) / spin_ms;
      if (trial == 0 || r < worst) worst = r;      // best of two: a probe can be delayed, never sped up
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    return worst;
  }
  std::vector<hipStream_t> lanes_;
  std::vector<float> last_probe_;
};

LaneRegistry& lane_registry() {
  static LaneRegistry r;
  return r;
}

// A plan has two lanes.  Lane 0 is the caller's stream; lane 1 is a side stream owned by the
// process (shared by all plans, chosen by LaneRegistry), used for work that is independent of the main dependency chain (weight gradients next to
// the BatchNorm / data-gradient chain of the backward pass: atomics- and MFMA-bound kernels
// next to bandwidth-bound ones).  "fork" makes the side lane wait for everything enqueued on
// the main lane so far, "join" makes the main lane wait for the side lane.  Under stream capture
// the same event calls become graph edges, so a replayed hipGraph has the two branches too.
// Fork / join events order kernels of two streams of one device; nothing on the host or on
// another device synchronises on them (collectives and host copies are ordered through the
// framework's own stream waits).  They are therefore created without the system-scope fence an
// event record performs by default: its cache write-back / invalidation left a ~10 us bubble on
// the main lane at every fork (22.00 -> 21.83 ms per ResNet-50 step; gradients of the two-lane
// and the single-lane plan agree to 5e-7, scripts/debug/check_lane_consistency.py).
// DTF_EVENT_FLAGS=fence restores the default events.
unsigned event_flags() {
  unsigned f = hipEventDisableTiming | hipEventDisableSystemFence;
  const char* e = std::getenv("DTF_EVENT_FLAGS");
  if (e && std::string(e) == "fence") f = hipEventDisableTiming;
  return f;
}

class Plan {
 public:
  ~Plan() {
    for (auto& g : graphs_) {
      if (g.exec) (void)hipGraphExecDestroy(g.exec);
      if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    for (auto e : events_) (void)hipEventDestroy(e);
  }
  int add(const std::string& kind, const py::dict& args, const std::string& name, int lane) {
    if (lane != 0 && lane != 1) throw std::invalid_argument("Plan.add: lane must be 0 or 1");
    Entry e;
    e.lane = lane;
    if (kind == "fork") e.type = FORK;
    else if (kind == "join") e.type = JOIN;
    else if (kind == "soft_join") e.type = SOFT_JOIN;
    else e.launch = make_op(kind, args, true);
    if (e.type != OP) {
      hipEvent_t ev;
      HIP_CHECK(hipEventCreateWithFlags(&ev, event_flags()));
      events_.push_back(ev);
      e.event = ev;
    }
    ops_.push_back(std::move(e));
    names_.push_back(name.empty() ? kind : name);
    return (int)ops_.size() - 1;
  }
  int size() const { return (int)ops_.size(); }
  std::string name(int i) const { return names_.at(i); }
  int lane(int i) const { return ops_.at(i).lane; }
  // `skip_soft_joins`: the caller orders whatever consumes the side lane's results itself (see
  // side_stream()), so "soft_join" entries do not make the main lane wait.  The range may then
  // end with side-lane work in flight; a later hard "join" (or run without the flag) collects it.
  void run(int begin, int end, uintptr_t stream, bool skip_soft_joins) {
    if (begin < 0 || end > (int)ops_.size() || begin > end) throw std::out_of_range("Plan.run range");
    enqueue(begin, end, reinterpret_cast<hipStream_t>(stream), skip_soft_joins);
    check_launch();
  }
  // The side lane of all plans (0 before its first use).
  static uintptr_t side_stream() { return reinterpret_cast<uintptr_t>(shared_side()); }
  // Captures ops [begin, end) on `stream` into a hipGraph; returns a handle for replay().
  // The range must end with the side lane joined (the capture cannot end otherwise).
  int capture(int begin, int end, uintptr_t stream) {
    if (begin < 0 || end > (int)ops_.size() || begin > end) throw std::out_of_range("Plan.capture range");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    Graph g;
    HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    try {
      enqueue(begin, end, s, false);
    } catch (...) {
      hipGraph_t tmp = nullptr;
      (void)hipStreamEndCapture(s, &tmp);
      if (tmp) (void)hipGraphDestroy(tmp);
      throw;
    }
    HIP_CHECK(hipStreamEndCapture(s, &g.graph));
    HIP_CHECK(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
    graphs_.push_back(g);
    return (int)graphs_.size() - 1;
  }
  void replay(int handle, uintptr_t stream) {
    HIP_CHECK(hipGraphLaunch(graphs_.at(handle).exec, reinterpret_cast<hipStream_t>(stream)));
  }

 private:
  enum Type { OP, FORK, JOIN, SOFT_JOIN };
  struct Entry {
    Type type = OP;
    int lane = 0;
    Launch launch;
    hipEvent_t event = nullptr;
  };
  struct Graph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; };

  static hipStream_t& shared_side() {
    static hipStream_t s = nullptr;
    return s;
  }
  hipStream_t side(hipStream_t main) {
    hipStream_t& side_ = shared_side();
    if (!side_) {
      // DTF_SIDE_PRIORITY: "low" / "high" place the side lane below / above the main lane
      int lo = 0, hi = 0;
      HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));   // lo: least urgent, hi: most
      const char* e = std::getenv("DTF_SIDE_PRIORITY");
      int prio = 0;
      if (e && std::string(e) == "low") prio = lo;
      if (e && std::string(e) == "high") prio = hi;
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      HIP_CHECK(hipStreamIsCapturing(main, &cs));
      if (cs == hipStreamCaptureStatusNone && !std::getenv("DTF_NO_LANE_PROBE"))
        side_ = lane_registry().pick(main, prio);
      else                                   // no synchronisation is allowed under capture
        HIP_CHECK(hipStreamCreateWithPriority(&side_, hipStreamNonBlocking, prio));
    }
    return side_;
  }
  void enqueue(int begin, int end, hipStream_t main, bool skip_soft_joins) {
    // side lane holds work the main lane has not waited for (possibly from an earlier range
    // that ended on a skipped soft join)
    bool side_busy = side_pending_;
    for (int i = begin; i < end; ++i) {
      Entry& e = ops_[i];
      if (e.type == FORK) {
        HIP_CHECK(hipEventRecord(e.event, main));
        HIP_CHECK(hipStreamWaitEvent(side(main), e.event, 0));
      } else if (e.type == SOFT_JOIN && skip_soft_joins) {
        // nothing: the consumer of the side lane's results waits on side_stream() itself
      } else if (e.type == JOIN || e.type == SOFT_JOIN) {
        if (side_busy) {
          HIP_CHECK(hipEventRecord(e.event, side(main)));
          HIP_CHECK(hipStreamWaitEvent(main, e.event, 0));
          side_busy = false;
        }
      } else if (e.lane == 1) {
        e.launch(side(main));
        side_busy = true;
      } else {
        e.launch(main);
      }
    }
    if (side_busy && !skip_soft_joins)
      throw std::logic_error("Plan: range ends with unjoined side-lane work");
    side_pending_ = side_busy;
  }

  std::vector<Entry> ops_;
  std::vector<std::string> names_;
  std::vector<Graph> graphs_;
  std::vector<hipEvent_t> events_;
  bool side_pending_ = false;
};

}  // namespace

PYBIND11_MODULE(_dtf_native, m) {
  m.doc() = "gfx950 (MI355X) kernels and launch-plan executor";
  m.def("run_op", [](const std::string& kind, const py::dict& args, uintptr_t stream) {
    make_op(kind, args)(reinterpret_cast<hipStream_t>(stream));
    check_launch();
  }, py::arg("kind"), py::arg("args"), py::arg("stream") = 0);
  m.def("concurrent_stream", [](uintptr_t main, int priority) {
    return reinterpret_cast<uintptr_t>(
        lane_registry().pick(reinterpret_cast<hipStream_t>(main), priority));
  }, py::arg("main"), py::arg("priority") = 0,
        "A new stream probed to execute concurrently with `main` and every lane handed out "
        "before (the plans' side lane included); lives as long as the process.");
  m.def("last_lane_probe", []() { return lane_registry().last_probe(); },
        "Overlap ratios measured by the most recent lane pick (1: concurrent, 2: serialised).");
  py::class_<dtf::RcclComm>(m, "RcclComm")
      .def(py::init([](py::bytes id, int nranks, int rank) {
             return new dtf::RcclComm(std::string(id), nranks, rank);
           }), py::arg("unique_id"), py::arg("nranks"), py::arg("rank"))
      .def_static("unique_id", []() { return py::bytes(dtf::RcclComm::unique_id()); })
      .def_static("load_library", &dtf::RcclComm::load_library)
      .def_static("group_start", &dtf::RcclComm::group_start)
      .def_static("group_end", &dtf::RcclComm::group_end)
      .def_property_readonly("rank", &dtf::RcclComm::rank)
      .def_property_readonly("size", &dtf::RcclComm::size)
      .def_property_readonly("handle", [](dtf::RcclComm& c) { return reinterpret_cast<uintptr_t>(&c); })
      .def_readwrite("direct_min_bytes", &dtf::RcclComm::direct_min_bytes)
      .def_readonly("collectives", &dtf::RcclComm::collectives)
      .def("async_error", &dtf::RcclComm::async_error, py::call_guard<py::gil_scoped_release>(),
           "'' while the communicator is healthy; else the error text (the communicator is aborted)")
      .def("abort", &dtf::RcclComm::abort, py::call_guard<py::gil_scoped_release>())
      .def("start_watchdog", &dtf::RcclComm::start_watchdog, py::arg("interval_ms") = 1000,
           py::arg("exit_code") = 0)
      .def("stop_watchdog", &dtf::RcclComm::stop_watchdog, py::call_guard<py::gil_scoped_release>())
      .def_property_readonly("failed", &dtf::RcclComm::failed)
      .def("all_reduce_mesh", [](dtf::RcclComm& c, uintptr_t p, size_t n, int dt, uintptr_t s) {
             c.all_reduce_mesh(reinterpret_cast<void*>(p), n, dt, reinterpret_cast<hipStream_t>(s)); },
           py::arg("ptr"), py::arg("count"), py::arg("dt") = 2, py::arg("stream") = 0)
      .def("all_reduce", [](dtf::RcclComm& c, uintptr_t p, size_t n, int dt, uintptr_t s) {
             c.all_reduce(reinterpret_cast<void*>(p), n, dt, reinterpret_cast<hipStream_t>(s)); },
           py::arg("ptr"), py::arg("count"), py::arg("dt") = 2, py::arg("stream") = 0)
      .def("all_reduce_direct", [](dtf::RcclComm& c, uintptr_t p, size_t n, int dt, uintptr_t s) {
             c.all_reduce_direct(reinterpret_cast<void*>(p), n, dt, reinterpret_cast<hipStream_t>(s)); },
           py::arg("ptr"), py::arg("count"), py::arg("dt") = 2, py::arg("stream") = 0)
      .def("all_reduce_auto", [](dtf::RcclComm& c, uintptr_t p, size_t n, int dt, uintptr_t s) {
             c.all_reduce_auto(reinterpret_cast<void*>(p), n, dt, reinterpret_cast<hipStream_t>(s)); },
           py::arg("ptr"), py::arg("count"), py::arg("dt") = 2, py::arg("stream") = 0)
      .def("reduce_scatter", [](dtf::RcclComm& c, uintptr_t a, uintptr_t b, size_t n, int dt, uintptr_t s) {
             c.reduce_scatter(reinterpret_cast<const void*>(a), reinterpret_cast<void*>(b), n, dt,
                              reinterpret_cast<hipStream_t>(s)); },
           py::arg("send"), py::arg("recv"), py::arg("recv_count"), py::arg("dt") = 2, py::arg("stream") = 0)
      .def("all_gather", [](dtf::RcclComm& c, uintptr_t a, uintptr_t b, size_t n, int dt, uintptr_t s) {
             c.all_gather(reinterpret_cast<const void*>(a), reinterpret_cast<void*>(b), n, dt,
                          reinterpret_cast<hipStream_t>(s)); },
           py::arg("send"), py::arg("recv"), py::arg("send_count"), py::arg("dt") = 2, py::arg("stream") = 0)
      .def("broadcast", [](dtf::RcclComm& c, uintptr_t p, size_t n, int dt, int root, uintptr_t s) {
             c.broadcast(reinterpret_cast<void*>(p), n, dt, root, reinterpret_cast<hipStream_t>(s)); },
           py::arg("ptr"), py::arg("count"), py::arg("dt") = 2, py::arg("root") = 0, py::arg("stream") = 0)
      .def("reduce", [](dtf::RcclComm& c, uintptr_t p, size_t n, int dt, int root, uintptr_t s) {
             c.reduce(reinterpret_cast<void*>(p), n, dt, root, reinterpret_cast<hipStream_t>(s)); },
           py::arg("ptr"), py::arg("count"), py::arg("dt") = 2, py::arg("root") = 0, py::arg("stream") = 0)
      .def("send", [](dtf::RcclComm& c, uintptr_t p, size_t n, int dt, int peer, uintptr_t s) {
             c.send(reinterpret_cast<const void*>(p), n, dt, peer, reinterpret_cast<hipStream_t>(s)); },
           py::arg("ptr"), py::arg("count"), py::arg("dt"), py::arg("peer"), py::arg("stream") = 0)
      .def("recv", [](dtf::RcclComm& c, uintptr_t p, size_t n, int dt, int peer, uintptr_t s) {
             c.recv(reinterpret_cast<void*>(p), n, dt, peer, reinterpret_cast<hipStream_t>(s)); },
           py::arg("ptr"), py::arg("count"), py::arg("dt"), py::arg("peer"), py::arg("stream") = 0);
  m.def("conv_rows_serves", [](const py::dict& d) {
    return conv_rows_serves(conv_from_dict(d), get_or<int>(d, "cfg", 0));
  }, "True when the whole-rows 3x3 kernel (conv_rows.hip) runs this convolution descriptor with its cfg");
  m.def("wgrad_rows_serves", [](const py::dict& d) {
    return wgrad_rows_serves(wgrad_from_dict(d), get_or<int>(d, "cfg", 0));
  }, "True when the whole-rows weight-gradient kernel (conv_wgrad_rows.hip) runs this descriptor with its cfg");
  m.def("fastdiv_check", [](uint32_t n, uint32_t d) { return fd_div(n, make_fastdiv(d)); });
  m.def("pack_desc_size", []() { return (int)sizeof(PackDesc); });
  m.def("bn_workspace_floats", []() { return bn_workspace_floats(); });
  m.def("bn_tail_floats", []() { return bn_tail_floats(); });
  m.def("bn_fold_desc_size", []() { return bn_fold_desc_size(); });
  m.attr("PACK_PER_BLOCK") = 256 * 8;
  m.attr("arch") = "gfx950";
  py::class_<Plan>(m, "Plan")
      .def(py::init<>())
      .def("add", &Plan::add, py::arg("kind"), py::arg("args"), py::arg("name") = "",
           py::arg("lane") = 0)
      .def("lane", &Plan::lane)
      .def("size", &Plan::size)
      .def("name", &Plan::name)
      .def("run", &Plan::run, py::arg("begin"), py::arg("end"), py::arg("stream") = 0,
           py::arg("skip_soft_joins") = false)
      .def_static("side_stream", &Plan::side_stream)
      .def("capture", &Plan::capture)
      .def("replay", &Plan::replay);
}
