// Descriptors of the implicit-GEMM convolution kernels (shared by kernels, launch plan and
// Python bindings).
#pragma once
#include "common.h"

// Training BatchNorm folded into the *consumer* convolution (small, latency-bound layers): `in`
// is the raw output y of the producing convolution and the kernel normalises it on its way into
// LDS, x = relu?(y * scale + shift (+ residual)).  Every block derives scale / shift itself from the sums the
// producer's epilogue accumulated (`sums`: [replicas][2][C] sum / sum of squares, C = Cpix);
// block 0 publishes what the backward pass and the moving averages need.  The blocks of the first
// channel tile also write the normalised values of their own pixels to `z` (the operand of this
// layer's weight gradient) and the 1-bit ReLU mask.
struct BnPrologue {
  const float* sums = nullptr;     // selects the prologue
  int replicas = 0;
  const float* gamma = nullptr;
  const float* beta = nullptr;
  float* moving_mean = nullptr;    // optional (null: no moving-average update, e.g. while timing)
  float* moving_var = nullptr;
  float* save_mean = nullptr;
  float* save_invstd = nullptr;
  float* scale = nullptr;          // published scale / shift (same slots the sweep kernels fill)
  float* shift = nullptr;
  const void* res = nullptr;       // optional residual added before the ReLU, laid out like `in`
  void* z = nullptr;               // optional side outputs, laid out like `in`
  uint8_t* mask = nullptr;         // [pixels][C/8]
  int P = 0;                       // pixels of `in` (N*H*W): the statistics' sample count
  float eps = 0.f, momentum = 0.f;
  int relu = 0;
};

// The backward counterpart, carried by a data-gradient launch: `in` is dz, the gradient of the
// *output* of the BatchNorm behind this layer, and the kernel turns it into the gradient of the
// layer's own output on the way into LDS,
//   dy = gamma * invstd * (g - dbeta / P - xhat * dgamma / P),   g = dz gated by the ReLU mask,
// with dbeta = sum g and dgamma = (sum g*y - mean * sum g) * invstd from the replica sums
// [replicas][2][C] that the launch which produced dz accumulated.  Block 0 stores dgamma / dbeta
// into the gradient buffer; the blocks of channel tile 0 write dy for their own rows (the operand
// of this layer's weight gradient).
struct BnBackwardPrologue {
  const float* sums = nullptr;     // selects the prologue
  int replicas = 0;
  const void* y = nullptr;         // the BatchNorm's input (this layer's output), laid out like `in`
  const uint8_t* mask = nullptr;   // ReLU bits [pixels][C/8]; null: no ReLU
  const float* mean = nullptr;
  const float* invstd = nullptr;
  const float* gamma = nullptr;
  float* dgamma = nullptr;
  float* dbeta = nullptr;
  void* dy = nullptr;              // side output, laid out like `in`
  int P = 0;
};

// Forward / data-gradient gather GEMM; see conv_igemm.hip for the formula.
struct ConvParams {
  int dt = DT_BF16;                // element type of in / wgt / out / addend (common.h)
  const void* in = nullptr;        // [N, H, W, Cpix]
  const void* wgt = nullptr;       // packed [Cout][ldw], K contiguous
  void* out = nullptr;             // [N, OHfull, OWfull, ldc]
  const void* addend = nullptr;    // optional tensor added to the result (same layout as out)
  // optional 1-bit mask of the addend, [pixels][ldc/8] bytes in the layout of `out`: only
  // elements whose bit is set are added (the ReLU mask of a residual branch's gradient)
  const uint8_t* addend_mask = nullptr;
  const float* bias = nullptr;     // optional per-output-channel bias
  // Whole-rows kernels only (conv_rows.hip): `in` is the raw output of the layer in front and the
  // training BatchNorm + ReLU between the two layers is applied inside the halo tile,
  // x := max(in * in_scale[c] + in_shift[c], 0), once per element, so that the normalised tensor is
  // never written (scale / shift: the per-channel constants the BatchNorm's finalize step stored)
  const float* in_scale = nullptr;
  const float* in_shift = nullptr;
  // Inference epilogue (folded BatchNorm): out = relu?(acc * scale[co] + bias[co] (+ addend)).
  // `scale` selects it; `bias` then carries the shift and `relu` the activation.
  const float* scale = nullptr;
  int relu = 0;
  // optional fused batch-norm statistics: per-channel sum / sum of squares of the (bf16
  // rounded) output, added atomically into replica (m-tile % replicas) of
  // stats[replicas][2][Cout]; the buffer is zeroed by the caller
  float* stats = nullptr;
  int stats_replicas = 0;
  // Backward flavour of the fused statistics (data-gradient launches): with `stats_y` set the
  // two accumulated quantities are  sum g  and  sum g * y,  where g is the (bf16 rounded)
  // output gated by the 1-bit `stats_mask` (null: ungated) and y is `stats_y` at the same
  // position - the reductions of the BatchNorm backward pass whose incoming gradient this
  // launch produces.  Bias-free; an addend is allowed (it is part of the gradient).
  const bf16_t* stats_y = nullptr;
  const uint8_t* stats_mask = nullptr;
  BnPrologue pro;
  BnBackwardPrologue bpro;
  int N = 0, H = 0, W = 0, Cpix = 0;
  int OH = 0, OW = 0, M = 0;       // logical output grid; M = N*OH*OW
  int OHfull = 0, OWfull = 0, ldc = 0;
  int OSH = 1, OSW = 1, OOH = 0, OOW = 0;
  int Cout = 0;
  int TI = 1, TJ = 1, Ctap = 0, Ktotal = 0;
  int ISH = 1, ISW = 1;
  int dh0 = 0, dhs = 0, dw0 = 0, dws = 0;
  int ldw = 0, wk0 = 0, wsi = 0, wsj = 0;
  // 1: pixel tiles are handed out back to front.  The launch plan alternates the direction of
  // consecutive launches over one tensor, so that each starts on the rows the one before it
  // touched last (still held by the 256 MB memory-side cache).
  int rev = 0;
  FastDiv divOW, divOH;
};

// Weight gradient: dW[k][co] += sum over pixels of X[pixel, k] * dY[pixel, co], where
// k = (tap i, tap j, channel c) gathers X exactly like the forward pass.
// One layer of a batched weight-gradient launch: layers of identical geometry (a stage of a
// CIFAR ResNet has eighteen) differ only in their three tensors.
struct WgradBatchItem {
  const void* x; const void* dy; float* dw;
};

struct WgradParams {
  int dt = DT_BF16;                // element type of x / dy
  // batched launch: device array of `nbatch` items, one grid z-slice each (x / dy / dw unused)
  const WgradBatchItem* batch = nullptr;
  int nbatch = 1;
  const void* x = nullptr;         // forward input [N, H, W, Cpix]
  const void* dy = nullptr;        // output gradient [N, OH, OW, ldy]
  float* dw = nullptr;             // fp32 [TIr*TJr*Creal][Cout] (Keras HWIO), accumulated atomically
  int N = 0, H = 0, W = 0, Cpix = 0;
  int OH = 0, OW = 0, P = 0;       // P = N*OH*OW pixels (the reduction dimension)
  int ldy = 0, Cout = 0;
  int TI = 1, TJ = 1, Ctap = 0, Ktotal = 0;
  int ISH = 1, ISW = 1, dh0 = 0, dhs = 1, dw0 = 0, dws = 1;
  // epilogue decode of the padded k index into the real HWIO index
  int Cpad = 0, Creal = 0, TJpad = 1, TJreal = 1;
  int splits = 1, pix_per_split = 0;
  int debug_plain_store = 0;       // timing experiments only: plain stores instead of atomics
  int rev = 0;                     // 1: pixel splits are handed out back to front (see ConvParams::rev)
  // whole-rows kernel only: wgrad_rows_workspace_floats() floats the blocks store their partial
  // gradients to (summed into `dw` by a second launch); null: every block adds to `dw` atomically
  float* ws = nullptr;
  // whole-rows kernel only: x := max(x * x_scale[c] + x_shift[c], 0) inside the halo tile (the
  // forward launch normalised its input the same way, see ConvParams::in_scale)
  const float* x_scale = nullptr;
  const float* x_shift = nullptr;
  FastDiv divOW, divOH, divCpad, divTJpad;
};

void launch_conv_igemm(const ConvParams& p, int cfg, hipStream_t stream);
// conv_pipe.hip: false when the descriptor needs the generic gather of conv_igemm.hip
bool launch_conv_pipe(const ConvParams& p, int cfg, hipStream_t stream);
int pipe_fallback_cfg(int cfg);
// conv_halo.hip: false when the tile cannot serve the descriptor
bool launch_conv_halo(const ConvParams& p, int cfg, hipStream_t stream);
int halo_fallback_cfg(int cfg);
// conv_rows.hip: 3x3 unit-stride layers over whole image rows; false when the tile cannot serve
// the descriptor
bool launch_conv_rows(const ConvParams& p, int cfg, hipStream_t stream);
bool conv_rows_serves(const ConvParams& p, int cfg);
int rows_fallback_cfg(int cfg);
void launch_conv_wgrad(const WgradParams& p, int cfg, hipStream_t stream);
// conv_wgrad_rows.hip: the whole-gradient-per-block kernel for large 3x3 unit-stride layers; false
// when it cannot serve the descriptor (the caller falls back to a tile of conv_wgrad.hip)
constexpr int WGRAD_ROWS_CFG = 192;
bool launch_conv_wgrad_rows(const WgradParams& p, int cfg, hipStream_t stream);
bool wgrad_rows_serves(const WgradParams& p, int cfg);
size_t wgrad_rows_workspace_floats();
