// Launchers of the bandwidth-bound kernels (BatchNorm, pooling, loss, optimizer, packing).
// Activation tensors are `void*` plus the element type `dt` (DT_BF16 / DT_F16 / DT_F32, common.h).
#pragma once
#include "common.h"

size_t bn_workspace_floats();
// `mask` (optional): one bit per element of the ReLU mask, [P][C/8] bytes, written as 32-bit
// words: the buffer must be 4-byte aligned with room for P*C/8 rounded up to 4 bytes.
// `fused_replicas` > 0: `ws` holds that many rows of [2][C] sums left by the convolution.
void launch_bn_fwd_train(const void* y, const void* res, void* z, uint8_t* mask,
                         const float* gamma, const float* beta, float* mm, float* mv,
                         float* save_mean, float* save_invstd, float* scale, float* shift,
                         float* ws, int fused_replicas, int P, int C, float eps, float momentum,
                         int relu, const float* rscale, const float* rshift, int apply, int rev,
                         int dt, hipStream_t s);
// Folded inference constants of every BatchNorm layer (descriptor layout: see bn.hip).
int bn_fold_desc_size();
void launch_bn_fold(const float* params, const float* state, float* out, const void* descs,
                    int nlayers, hipStream_t s);
// BatchNorm + ReLU + max-pool in one sweep (forward) and the matching backward, see bn.hip.
void launch_bn_pool_fwd_train(const void* y, void* zp, uint8_t* idx, const float* gamma,
                              const float* beta, float* mm, float* mv, float* save_mean,
                              float* save_invstd, float* scale, float* shift, float* ws,
                              int fused_replicas, int N, int H, int W, int C, int OH, int OW, int k,
                              int stride, int pt, int pl, float eps, float momentum, int dt,
                              hipStream_t s);
void launch_bn_pool_bwd(const void* dzp, const uint8_t* idx, const void* y, const float* mean,
                        const float* invstd, const float* gamma, float* dgamma, float* dbeta,
                        void* dy, float* ws, int N, int H, int W, int C, int OH, int OW, int k,
                        int stride, int pt, int pl, int dt, hipStream_t s);
void launch_bn_bwd_dual(const void* dz, const uint8_t* mask, int P, int C, float* ws, float* ws2,
                        const void* y1, const float* mean1, const float* invstd1,
                        const float* gamma1, float* dgamma1, float* dbeta1, void* dy1,
                        const void* y2, const float* mean2, const float* invstd2,
                        const float* gamma2, float* dgamma2, float* dbeta2, void* dy2,
                        float* tail, int rev, int dt, hipStream_t s);
void launch_bn_infer(const void* y, const void* res, void* z, const float* gamma,
                     const float* beta, const float* mm, const float* mv, int P, int C, float eps,
                     int relu, int dt, hipStream_t s);
void launch_bn_bwd(const void* dz, const void* y, const void* z, const uint8_t* mask,
                   const float* mean, const float* invstd, const float* gamma, float* dgamma,
                   float* dbeta, void* dy, void* dres, float* ws, int P, int C,
                   int pre_replicas, float* tail, int rev, const float* relu_scale, const float* relu_shift, int dt,
                   hipStream_t s);
// `rev`: bit 0 - the reduction sweep, bit 1 - the apply sweep walks the rows back to front (the
// launch plan alternates directions so that a sweep starts on the rows still in the memory-side
// cache; `rev` of the forward launch is the apply sweep's direction).
// `tail` (optional, bn_tail_floats() zeroed floats, shared by launches of one stream): the
// reduction sweep adds its per-block sums to a few replica rows and its last block derives
// dgamma / dbeta, so no finalize launch sits between the two sweeps.  The sums then depend on
// the order of the atomic additions; without `tail` they are deterministic.
size_t bn_tail_floats();

// out[i] = bit (i % 8) of mask[i / 8] ? x[i] : 0, n a multiple of 8 (materialises a masked
// gradient when its consumer cannot apply the mask itself).
void launch_mask_apply(const void* x, const uint8_t* mask, void* out, size_t n, int dt,
                       hipStream_t s);

void launch_maxpool_fwd(const void* x, void* out, uint8_t* idx, int N, int H, int W, int C,
                        int OH, int OW, int k, int stride, int pt, int pl, int dt, hipStream_t s);
void launch_maxpool_bwd(const void* dy, const uint8_t* idx, void* dx, int N, int H, int W,
                        int C, int OH, int OW, int k, int stride, int pt, int pl, int dt,
                        hipStream_t s);
void launch_global_mean_fwd(const void* x, void* out, int N, int HW, int C, int dt, hipStream_t s);
void launch_global_mean_bwd(const void* dy, void* dx, int N, int HW, int C, int dt, hipStream_t s);

// Fused softmax + cross-entropy (+ gradient) over bf16 logits with row pitch ld.
// accum[0] += sum of per-row losses, accum[1] += number of rows whose argmax equals the label.
void launch_softmax_xent(const void* logits, const int32_t* labels, void* dlogits,
                         float* accum, int N, int C, int ld, float grad_scale,
                         const float* grad_scale_dev, float loss_weight, int dt, hipStream_t s);
void launch_softmax_probs(const void* logits, float* probs, int N, int C, int ld, int dt,
                          hipStream_t s);
void launch_colsum(const void* x, float* out, int N, int C, int ld, int dt, int slices_req,
                   hipStream_t s);

// Dense layers of any shape on fp32 master weights [in, out] (dense.hip).
void launch_dense_fwd(const void* x, const float* w, const float* bias, void* out, int N, int K,
                      int Cout, int ldx, int ldo, int dt, hipStream_t s);
void launch_dense_wgrad(const void* x, const void* dy, float* dw, int N, int K, int Cout, int ldx,
                        int ldy, int dt, hipStream_t s);
void launch_dense_dgrad(const void* dy, const float* w, void* dx, int N, int K, int Cout, int ldy,
                        int ldx, int dt, hipStream_t s);
// Dynamic loss scaling without a host round trip (dense.hip): flag[0] |= any non-finite gradient;
// the state kernel consumes the flag and updates [scale, clean, skipped, skip, 1/scale used];
// the fused optimizer takes state + 3 as `skip` ([skip, inverse scale]).
void launch_nonfinite_flag(const float* g, size_t n, unsigned int* flag, hipStream_t s);
void launch_loss_scale_update(float* state, unsigned int* flag, float* loss_scale_dev, float growth,
                              float multiplier, hipStream_t s);

void launch_sgd_momentum_l2(float* p, const float* g, float* v, const float* l2x2, size_t n,
                            float lr, float momentum, float grad_scale, int use_momentum,
                            const float* hyper, const float* skip, hipStream_t s);

// One entry per packed bf16 weight copy (see optim.hip).
struct PackDesc {
  int64_t src_off;      // element offset into the flat fp32 parameter buffer (HWIO / [in,out])
  int64_t dst_off;      // element offset into the flat bf16 packed-weight buffer
  int32_t TI, TJ, Cin, Cout;          // real dims of the HWIO tensor
  int32_t TJpad, Cpad, CoutPad;       // padded dims of the packed copy
  int32_t mode;                       // 0: [Cout][TI][TJpad][Cpad]   1: [Cin][TI][TJ][CoutPad]
  int64_t count;                      // elements of the packed copy
  int64_t first_block;                // first thread block that works on this entry
};
void launch_pack_weights(const float* params, void* packed, const PackDesc* descs, int ndesc,
                         int nblocks, int dt, hipStream_t s);

void launch_preprocess_pad(const float* in, void* out, int N, int H, int W, int Cin, int pt,
                           int pl, int OH, int OW, int Cpad, int dt, hipStream_t s);
void launch_zero(void* ptr, size_t nbytes, hipStream_t s);
void launch_cast_from_f32(const float* in, void* out, size_t n, int dt, hipStream_t s);
