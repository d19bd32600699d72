/*
 * uemda_hip.h -- C ABI of libuemda_hip.so: hand-written HIP kernels (gfx950 / MI355X) for the
 * UemDA hot path.  The reference has no FFI of its own: its "operator surface"
 * is Python calling third-party native kernels (cuDNN through torch, torch_scatter).  Every entry
 * point below replaces one such native call site; the reference file:line it replaces is cited.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch allocations); the library
 *     allocates nothing and keeps no state; all launches are asynchronous on `stream`
 *     (a hipStream_t passed as void*), safe for hipGraph capture.
 *   - return value: 0 = launched, <0 = error (UEM_ERR_*); never throws, never syncs.
 *   - activations are NHWC fp32 ("channels_last"): x[n][y][x][c]; conv weights are OHWI fp32:
 *     w[o][ky][kx][i]  (= torch OIHW tensor in channels_last memory format).
 *   - full-resolution class maps (soft labels) are NCHW planar fp32; label maps are int64.
 */
#ifndef UEMDA_HIP_H
#define UEMDA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UEM_OK 0
#define UEM_ERR_INVALID (-1)     /* bad shape / null pointer / misalignment */
#define UEM_ERR_UNSUPPORTED (-2) /* configuration outside what the kernels cover */
#define UEM_ERR_LAUNCH (-3)      /* hipGetLastError() != success after the launch */

#define UEM_MAX_CLASSES 16

int uem_version(void);
const char* uem_last_error(void);

/* ---- convolution: implicit GEMM on f32-input MFMA (v_mfma_f32_32x32x2_f32) ---------------------
 * replaces cuDNN conv fwd/dgrad/wgrad behind nn.Conv2d at _resnets.py:95-110,149,209 and
 * Encoder.py:74-75,81-83 (ASPP), Encoder.py:19,35-36,40 (PPM).                                   */
typedef struct {
    int N, H, W, Cin;      /* input  x[N][H][W][Cin]                      */
    int Ho, Wo, Cout;      /* output y[N][Ho][Wo][Cout]                   */
    int KH, KW;            /* filter taps                                 */
    int stride, pad, dil;  /* same in both spatial dims                   */
    int x_ld, y_ld;        /* channel stride (floats) of one pixel in x / y: >= Cin / Cout; lets a
                              conv read / write a channel slice of a wider NHWC tensor (PPM concat) */
} uem_conv_shape;

/* flags for uem_conv2d_fwd / wgrad */
#define UEM_CONV_IN_AFFINE 1   /* operand prologue: x' = x*in_scale[c] + in_shift[c]  (fused BN apply) */
#define UEM_CONV_IN_RELU 2     /* operand prologue: x' = max(x', 0)                                  */
#define UEM_CONV_ACCUMULATE 4  /* epilogue: y += result (ASPP branch sum, Encoder.py:83)               */
/* (16 was UEM_CONV_PREC_BF16X3, the split-operand mode of rounds 1-3: retired in round 4, the flag is rejected) */
#define UEM_CONV_PREC_BF16 32   /* bf16 operands on the bf16 MFMA, fp32 tensors and fp32 accumulate: the operand mode of the fp32
                                  islands (7x7 stem, ASPP heads' GEMM) of a bf16-STORAGE model; never the fp32 parity path   */
#define UEM_CONV_TRANSPOSED 8  /* gather of the data-gradient of a strided conv: shape describes the
                                  FORWARD conv, x is dY (N,Ho,Wo,Cout), y is dX (N,H,W,Cin), w is
                                  W'[Cin][KH][KW][Cout] (uem_weight_transpose of the forward weights) */

/* y = conv(x', w) (+ bias).  w: [Cout][KH][KW][Cin].  Requires Cin % 32 == 0 except the 7x7 stem,
 * which is expressed as KH=7, KW=1, Cin=32 over an NHWC4 image (see uem_stem_pack_*).             */
int uem_conv2d_fwd(const float* x, const float* w, const float* bias, const float* in_scale,
                   const float* in_shift, float* y, const uem_conv_shape* s, int flags, void* stream);
/* same forward conv, and the epilogue also leaves the per-128-row-tile column sums of y and y*y in
 * tile_stats[2][Cout][M/128] (fused BatchNorm statistics: saves the stand-alone pass over y).  Needs
 * M % 128 == 0 and Cout % 64 == 0 (UEM_ERR_UNSUPPORTED otherwise: use uem_bn_stats).                   */
int uem_conv2d_fwd_stats(const float* x, const float* w, const float* in_scale, const float* in_shift, float* y,
                         const uem_conv_shape* s, int flags, float* tile_stats, void* stream);
/* data gradient of a stride-1 conv (dx = dA of the producing layer) whose epilogue also computes the first pass
 * of that layer's BatchNorm+ReLU backward: with bn_z = the layer's raw conv output (N,H,W,Cin) and bn_vec =
 * (4,Cin) [scale, shift, mean, invstd], tile_partials[2][Cin][M/128] receives per-tile sums of dp = dA*[relu mask]
 * and dp*xhat (replaces uem_bn_bwd_reduce's pass over z and dA).  s describes the FORWARD conv; flags may carry
 * a UEM_CONV_PREC_* bit only.                                                                            */
int uem_conv2d_dgrad_bnbwd(const float* dy, const float* w_t, float* dx, const uem_conv_shape* s, const float* bn_z,
                           const float* bn_vec, float* tile_partials, int flags, void* stream);
/* The data gradient that closes a bottleneck block's backward (dx of its first 1x1 conv, or of its downsample conv) with
 * the residual bookkeeping in its epilogue (reference autograd through uemda/_resnets.py:92-112):
 *   dx  = dgrad(dy)  +  acc_src * [acc_bits]     identity gradient of THIS block gated by its output ReLU mask (packed bits
 *                                                of uem_affine_act), never materialised; acc_src may alias dx.  With
 *                                                acc_src == NULL and UEM_CONV_ACCUMULATE in flags: dx += dgrad(dy).
 *   tile_partials[2][Cin][M/128] = per-tile sums of dp = dx * [bn_bits] and dp * xhat(bn_z)   -- the reduction pass of the
 *                                                PREVIOUS block's bn3 backward (its z3, (4,Cin) vectors and output mask),
 *                                                taken while dx is still on chip; bn_* may be NULL.
 * Needs stride 1, M % 128 == 0, Cin % 64 == 0 (UEM_ERR_UNSUPPORTED otherwise).                                         */
int uem_conv2d_dgrad_tail(const float* dy, const float* w_t, float* dx, const uem_conv_shape* s, const float* acc_src,
                          const uint32_t* acc_bits, const float* bn_z, const float* bn_vec, const uint32_t* bn_bits,
                          float* tile_partials, int flags, void* stream);
/* stem: x4 is the NHWC4 image (C padded 3->4), w8 is [64][7][8][4] (kx padded 7->8, c 3->4).     */
int uem_conv2d_stem_fwd(const float* x4, const float* w8, float* y, int N, int H, int W, void* stream);
/* the same with the per-tile BatchNorm statistics of uem_conv2d_fwd_stats out of the epilogue ([2][64][M/128]; N*Ho*Wo % 128 == 0):
 * conv1 + the statistics pass of bn1, uemda/_resnets.py:149-150;
 * flags: 0 or one UEM_CONV_PREC_* operand precision                                                                      */
int uem_conv2d_stem_fwd_stats(const float* x4, const float* w8, float* y, int N, int H, int W, float* tile_stats, int flags,
                              void* stream);
/* dw[o][ky][kx][i] += sum_m dy[m][o] * x'[m@tap][i]   (fp32 atomics: callers zero / accumulate)   */
int uem_conv2d_wgrad(const float* x, const float* dy, const float* in_scale, const float* in_shift,
                     float* dw, const uem_conv_shape* s, int flags, void* stream);
int uem_conv2d_stem_wgrad(const float* x4, const float* dy, float* dw8, int N, int H, int W, void* stream);
/* the same with an operand precision (0 or one UEM_CONV_PREC_* flag) */
int uem_conv2d_stem_wgrad_prec(const float* x4, const float* dy, float* dw8, int N, int H, int W, int flags, void* stream);
/* bf16 storage (BASELINE config 5; round 5): the stem's conv output z -- 64 channels at half resolution, the network's largest
 * tensor -- is a bf16 tensor: rounded (RNE) at the conv's store with the BatchNorm tile statistics taken over the rounded values,
 * read as bf16 by the pool and by both passes of the BatchNorm backward, whose dz is bf16 too and is widened at the weight
 * gradient's load.  bf16 operands on the bf16 matrix cores, fp32 accumulation.            uemda/_resnets.py:149-153,205-212 */
int uem_conv2d_stem_fwd_stats_bf16(const float* x4, const float* w8, uint16_t* y, int N, int H, int W, float* tile_stats, void* stream);
int uem_conv2d_stem_wgrad_bf16(const float* x4, const uint16_t* dy, float* dw8, int N, int H, int W, void* stream);
/* The stem as its own kernels (round 5, csrc/stem.hip): conv1 = Conv2d(3, 64, 7, stride 2, padding 3), uemda/_resnets.py:149-153,205-212.
 * x4: the NHWC4 image; w_ohwi / dw_ohwi: the (64, 7, 7, 3) filter bank and its gradient as the model holds them (no packed copy);
 * exact fp32 arithmetic, 148 reduction elements walked for 147.  The output must be whole 8 x 32 pixel tiles (Ho % 8 == 0,
 * Wo % 32 == 0): UEM_ERR_UNSUPPORTED otherwise, nothing launched, the caller takes uem_conv2d_stem_fwd[_stats] / _wgrad.
 *   fwd   : z (N, Ho, Wo, 64); tile_stats (optional) = the partial sums of uem_conv2d_fwd_stats, [2][64][N*Ho*Wo/128].
 *   wgrad : dw_ohwi += sum over pixels; dz fp32 or bf16 (bf16 storage); deterministic (per-block banks summed in a fixed order);
 *           workspace: uem_stem_conv_wgrad_workspace_floats() floats, 16-byte aligned.                                        */
int uem_stem_conv_fwd(const float* x4, const float* w_ohwi, float* z, int N, int H, int W, float* tile_stats, void* stream);
int64_t uem_stem_conv_wgrad_workspace_floats(void);
int uem_stem_conv_wgrad(const float* x4, const float* dz, float* dw_ohwi, float* workspace, int N, int H, int W, void* stream);
int uem_stem_conv_wgrad_bf16(const float* x4, const uint16_t* dz, float* dw_ohwi, float* workspace, int N, int H, int W, void* stream);
/* weight re-layouts (tiny): transposed copy for dgrad; stem pack / unpack-add                     */
int uem_weight_transpose(const float* w /*[Cout][KH][KW][Cin]*/, float* wt /*[Cin][KH][KW][Cout]*/, int Cout,
                         int KH, int KW, int Cin, void* stream);
int uem_stem_pack_weight(const float* w_ohwi /*[64][7][7][3]*/, float* w8, void* stream);
int uem_stem_unpack_grad(const float* dw8, float* dw_ohwi /* += */, void* stream);
int uem_nchw3_to_nhwc4(const float* x, float* x4, int N, int H, int W, void* stream);
int uem_bias_grad(const float* dy, float* db /* += */, int M, int C, int ld, void* stream);
/* ASPP heads (Encoder.py:68-84) as one dense 1x1 GEMM G = feat x Wall plus a gather:
 *   out[n,y,x,j] = sum_d ( bias[d][j] + sum_tap G[n, y+(ky-1)*dil_d, x+(kx-1)*dil_d, (d*9+tap)*K2 + j] )
 * G is (N,h,w,R) with R >= nd*9*K2 (padded to the GEMM tile); K2 = heads*classes; dil is a HOST array.
 * bwd fills dG (all R columns; the padding gets zeros) from dout (N,h,w,K2).                         */
int uem_aspp_gather_fwd(const float* G, const float* bias /* [nd][K2] */, float* out, float* out2 /* NULL: out is (N,h,w,K2);
                        else head 0 -> out, head 1 -> out2, each (N,h,w,K2/2) */, int N, int h, int w, int K2,
                        int R, int nd, const int* dil, void* stream);
int uem_aspp_gather_bwd(const float* dout, const float* dout2 /* as out2 */, float* dG, int N, int h, int w, int K2, int R, int nd,
                        const int* dil, void* stream);
/* The heads' filters <-> the GEMM's filter bank in one launch each way: w / b (HOST arrays of nheads*nd device pointers, index
 * head*nd + d) are the (C,3,3,cin) OHWI filters and (C,) biases of Classifier_Module.conv2d_list (Encoder.py:74-78).
 * nheads: 2 = the layer5 / layer6 pair of the multi-layer model, 1 = one Classifier_Module (Deeplabv2's single-head default
 * `cls_pred` and each head of its cascade branch, Encoder.py:93-102,129-143,156-165).
 * pack: wall (R,cin) row (d*9+tap)*nheads*C + head*C + c = w[head][d][c][tap][:], rows >= nd*9*nheads*C zero; bias [nd][nheads][C].
 * unpack_grad: gw[..] += the matching rows of dwall; gb[head][d][:] += db[head*C ..] (every dilation's bias sees the same gradient). */
int uem_aspp_pack(void* const* w, void* const* b, float* wall, float* bias, int C, int cin, int nd, int R, int nheads, void* stream);
int uem_aspp_unpack_grad(const float* dwall, const float* db, void* const* gw, void* const* gb, int C, int cin, int nd, int nheads,
                         void* stream);

/* ---- BatchNorm2d (training + eval), fused ReLU / residual -- _resnets.py:96-110, Encoder.py:20,37 --
 * stats: per-channel batch mean / biased var of x[M][C]; also updates running stats
 * (momentum, unbiased var) when running_mean != NULL; writes scale = gamma*rsqrt(var+eps),
 * shift = beta - mean*scale (the conv prologue operands) and save_mean / save_invstd.            */
int uem_bn_stats(const float* x, int M, int C, int ld, const float* gamma, const float* beta,
                 float eps, float momentum, float* running_mean, float* running_var,
                 float* save_mean, float* save_invstd, float* scale, float* shift,
                 float* workspace /* >= uem_bn_workspace_floats(M, C) floats */, void* stream);
int64_t uem_bn_workspace_floats(int M, int C); /* also covers uem_bn_bwd_reduce's workspace */
/* the same outputs from uem_conv2d_fwd_stats' tile sums (tiles = M/128)                              */
int uem_bn_stats_from_tiles(const float* tile_stats, int tiles, int M, int C, const float* gamma, const float* beta,
                            float eps, float momentum, float* running_mean, float* running_var, float* save_mean,
                            float* save_invstd, float* scale, float* shift, void* stream);
/* eval-mode BatchNorm as an affine map: scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale; mean /
 * invstd (may be NULL) = running_mean, 1 / sqrt(running_var + eps), what uem_bn_bwd_reduce needs for the gamma / beta
 * gradients of a BatchNorm that runs in eval mode inside a training graph (reference resnet.py:112-117,183-190)        */
int uem_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                       float* scale, float* shift, float* mean, float* invstd, int C, void* stream);
/* y = act(x*scale + shift (+ r)); r = res, or res*res_scale + res_shift (downsample branch BN) when
 * res_scale != NULL; act = relu if relu != 0.  In place allowed (y == x).  relu_bits (optional, C % 32 == 0,
 * M*C/32 words): bit (e & 31) of word e >> 5 = [y_e > 0], e = row*C + c -- the 1/32-size ReLU mask the backward
 * passes read with relu = UEM_RELU_BITS instead of re-reading y.                                  */
int uem_affine_act(const float* x, const float* scale, const float* shift, const float* res,
                   const float* res_scale, const float* res_shift, float* y, int64_t M, int C, int relu,
                   uint32_t* relu_bits, void* stream);
#define UEM_RELU_BITS 2
/* backward of y = relu?(bn(x) (+res)):  given dy (grad wrt y).  relu mask: relu == 1: (ymask > 0) when ymask (the
 * materialised y, float) is given, else (x*scale+shift > 0) recomputed (operand-prologue layers); relu ==
 * UEM_RELU_BITS: ymask is uem_affine_act's relu_bits.
 * pass 1: dgamma[c] = sum dp*xhat, dbeta[c] = sum dp   (dp = dy masked by relu)
 * pass 2: dx = scale*(dp - dbeta/M - xhat*dgamma/M);  dres (optional) = dp                         */
int uem_bn_bwd_reduce(const float* x, const float* dy, const void* ymask, const float* scale, const float* shift,
                      const float* save_mean, const float* save_invstd, int M, int C, int relu,
                      float* dgamma /* = */, float* dbeta /* = */, float* grad_gamma /* +=, may be NULL */,
                      float* grad_beta /* +=, may be NULL */, float* workspace, void* stream);
int uem_bn_bwd_from_tiles(const float* tile_partials, int tiles, int C, float* dgamma /* = */, float* dbeta /* = */,
                          float* grad_gamma /* += or NULL */, float* grad_beta /* += or NULL */, void* stream);
int uem_bn_bwd_apply(const float* x, const float* dy, const void* ymask, const float* scale, const float* shift,
                     const float* save_mean, const float* save_invstd, const float* dgamma,
                     const float* dbeta, int M, int C, int relu, float* dx, float* dres, void* stream);
/* bn3's and the downsample BatchNorm's backward apply in ONE pass (round 5): the bottleneck blocks with a downsample branch end in
 * y = relu(bn3(z3) + bn_ds(zd)) (uemda/_resnets.py:104-112), so both read the same dy gated by the same packed ReLU bits.
 * dx1 = bn backward of x1 under vectors 1, dx2 = of x2 under vectors 2 (dgamma / dbeta: the finished channel sums of the reduction
 * passes); dx2 may alias dy.  Same element arithmetic as uem_bn_bwd_apply(relu = UEM_RELU_BITS).  Large power-of-two-channel maps only
 * (what the rows kernels take): UEM_ERR_UNSUPPORTED otherwise, nothing launched -- run the two uem_bn_bwd_apply passes.            */
int uem_bn_bwd_apply_pair(const float* x1, const float* x2, const float* dy, const uint32_t* relu_bits, const float* scale1,
                          const float* mean1, const float* invstd1, const float* dgamma1, const float* dbeta1, const float* scale2,
                          const float* mean2, const float* invstd2, const float* dgamma2, const float* dbeta2, int M, int C, float* dx1,
                          float* dx2, void* stream);
int uem_bn_bwd_apply_pair_bf16(const uint16_t* x1, const uint16_t* x2, const uint16_t* dy, const uint32_t* relu_bits, const float* scale1,
                               const float* mean1, const float* invstd1, const float* dgamma1, const float* dbeta1, const float* scale2,
                               const float* mean2, const float* invstd2, const float* dgamma2, const float* dbeta2, int M, int C,
                               uint16_t* dx1, uint16_t* dx2, void* stream);
/* BatchNorm(+ReLU, mask recomputed from x) backward of the layer in front of that max-pool (autograd of bn1 / relu / maxpool,
 * uemda/_resnets.py:150-153), reading the POOLED gradient
 * dy_pool (N, Ho, Wo, C) and the argmax taps in gather form instead of uem_maxpool3x3s2_bwd's (N, H, W, C) output: same
 * sums and the same dx as uem_bn_bwd_reduce / uem_bn_bwd_apply on that tensor                                             */
int uem_bn_bwd_reduce_pool(const float* x, const float* dy_pool, const uint8_t* idx, const float* scale, const float* shift,
                           const float* save_mean, const float* save_invstd, int N, int H, int W, int C, int relu, float* dgamma,
                           float* dbeta, float* grad_gamma, float* grad_beta, float* workspace, void* stream);
int uem_bn_bwd_apply_pool(const float* x, const float* dy_pool, const uint8_t* idx, const float* scale, const float* shift,
                          const float* save_mean, const float* save_invstd, const float* dgamma, const float* dbeta, int N, int H,
                          int W, int C, int relu, float* dx, void* stream);
/* the same two passes on bf16 tensors (x = the stem's bf16 z, dy_pool / dx bf16; sums, scale / shift and the arithmetic fp32) */
int uem_bn_bwd_reduce_pool_bf16(const uint16_t* x, const uint16_t* dy_pool, const uint8_t* idx, const float* scale, const float* shift,
                                const float* save_mean, const float* save_invstd, int N, int H, int W, int C, int relu, float* dgamma,
                                float* dbeta, float* grad_gamma, float* grad_beta, float* workspace, void* stream);
int uem_bn_bwd_apply_pool_bf16(const uint16_t* x, const uint16_t* dy_pool, const uint8_t* idx, const float* scale, const float* shift,
                               const float* save_mean, const float* save_invstd, const float* dgamma, const float* dbeta, int N, int H,
                               int W, int C, int relu, uint16_t* dx, void* stream);
/* eval-mode / frozen-statistics backward: dx = dp * scale, dp = dy masked as in uem_bn_bwd_apply (relu 0 / 1 / UEM_RELU_BITS) */
int uem_affine_act_bwd(const float* x, const float* dy, const float* ymask, const float* scale,
                       const float* shift, int64_t M, int C, int relu, float* dx, float* dres, void* stream);

/* ---- MaxPool 3x3 s2 p1 (_resnets.py:153), InstanceNorm2d (Encoder.py:123,147) ---------------------- */
int uem_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx /* argmax tap 0..8, may be NULL */, int N, int H,
                         int W, int C, void* stream);
/* y = maxpool(relu(x*scale + shift)): the stem's BatchNorm + ReLU (_resnets.py:150-153) applied in the pool's fetch, the
 * normalised map is never written; same first-max-wins argmax as uem_maxpool3x3s2_fwd on the materialised tensor          */
int uem_maxpool3x3s2_affine_fwd(const float* x, const float* scale, const float* shift, float* y, uint8_t* idx, int N, int H,
                                int W, int C, void* stream);
int uem_maxpool3x3s2_affine_fwd_bf16(const uint16_t* x, const float* scale, const float* shift, uint16_t* y, uint8_t* idx, int N,
                                     int H, int W, int C, void* stream);      /* bf16 z in, bf16 pooled map out (bf16 storage) */
int uem_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx /* = */, int N, int H, int W, int C,
                         void* stream);
int uem_instnorm_fwd(const float* x, float* y, float* save_mean, float* save_invstd, int N, int HW, int C,
                     float eps, void* stream);
int uem_instnorm_bwd(const float* y, const float* dy, const float* save_invstd, float* dx, int N, int HW,
                     int C, void* stream);
/* the InstanceNorm at the end of the bf16-storage region: bf16 layer4 output in, fp32 features out (what the heads and the
 * mining read); its backward writes the bf16 gradient -- no cast passes on either side.               Encoder.py:123,146-147 */
int uem_instnorm_fwd_bf16(const uint16_t* x, float* y, float* save_mean, float* save_invstd, int N, int HW, int C,
                          float eps, void* stream);
int uem_instnorm_bwd_bf16(const float* y, const float* dy, const float* save_invstd, uint16_t* dx, int N, int HW,
                          int C, void* stream);
/* ---- PPM head pieces: adaptive avg-pool + bilinear (align_corners=False) (Encoder.py:18,48-51) ------ */
int uem_adaptive_avgpool_fwd(const float* x, float* y, int N, int H, int W, int C, int S, void* stream);
int uem_adaptive_avgpool_bwd(const float* dy, float* dx /* += */, int N, int H, int W, int C, int S, void* stream);
/* the feature gradient of a PPM head in one pass (autograd of the concat and the pooled branches, uemda/models/Encoder.py:45-54):
 * dfeat (N,H,W,C) = dcat[..., :C] (row stride dcat_ld) + sum_i
 * adaptive_avgpool_bwd(dp[i] (N,S_i,S_i,C)) -- replaces the slice copy and one uem_adaptive_avgpool_bwd pass per branch; dp and
 * scales are HOST arrays of nbranch <= 4 entries (device pointers / bin counts)                                          */
int uem_ppm_feat_grad(const float* dcat, int dcat_ld, const float* const* dp, const int* scales, int nbranch, float* dfeat, int N,
                      int H, int W, int C, void* stream);
int uem_bilinear_up_fwd(const float* x, float* y, int N, int h, int w, int C, int H, int W, int y_ld,
                        int align_corners, const float* scale, const float* shift, int relu, void* stream);
int uem_bilinear_up_bwd(const float* dy, float* dx /* = */, int N, int h, int w, int C, int H, int W, int dy_ld,
                        int align_corners, void* stream);
int uem_dropout2d(const float* x, float* y, float* mask /*[N][C]*/, int N, int HW, int C, float p,
                  uint64_t seed, void* stream);
int uem_add_inplace(float* a, const float* b, int64_t n, void* stream);
/* a += b; b = 0 in one pass: the fold of the shadow gradient arena (the step's second graph, backward on its own stream, accumulates
 * its parameter gradients there -- the host-side ordering this package adds around torch autograd's per-stream backward; the
 * reference accumulates both graphs into one .grad on one stream, tools/train_ssl_uem.py:224-227). */
int uem_add_clear(float* a, float* b, int64_t n, void* stream);
int uem_nhwc_to_nchw(const float* x, float* y, int N, int HW, int C, void* stream);
int uem_nchw_to_nhwc(const float* x, float* y, int N, int HW, int C, void* stream);

/* ---- pseudo-label mining ------------------------------------------------------------------------
 * pearson_sim: sim[n][c] = 1 / pearson_dist(feat[n][:], protos[c][:])       alignment.py:216,424-451
 *   feat is NHWC (n = B*h*w rows of k floats), protos [C][k].                                      */
int uem_pearson_sim(const float* feat, const float* protos, float* sim, float* workspace /* C*k+C floats */,
                    int n, int k, int C, void* stream);
int uem_pearson_dist(const float* a, const float* b, float* dist, float* workspace /* m*k+m floats */, int n,
                     int m, int k, void* stream);
/* batch-global max of an int64 index map -> *out (device)                   alignment.py:241       */
int uem_index_max(const int64_t* idx, int64_t count, int64_t* out, void* stream);
/* torch_scatter.scatter(src, index, dim=1, reduce) for src (B,N,C) / index (B,N,1)
 * alignment.py:187,245.  reduce: 0 = max, 1 = sum, 2 = mean.  out (B,S,C) is fully written.        */
int uem_scatter(const float* src, const int64_t* index, float* out, float* workspace /* B*S floats */,
                int B, int N, int C, int S, int reduce, void* stream);
/* per-superpixel max of an NCHW-planar soft label: seg_keys[b][s][c] (order-preserving uint keys,
 * caller zero-fills), LDS-table pre-reduction per 64x16 pixel tile.         alignment.py:244-245
 * The reference sizes the table from the batch (dim_size = index.max()+1, alignment.py:241-245); here S is the
 * caller's capacity and `out_of_range` (device int, caller zero-fills, may be NULL) receives the largest id
 * outside [0, S) (INT_MAX for a negative one): such ids are skipped here and uem_label_refine leaves their
 * pixels' weights untouched, so a too-small table is reported, never folded into another segment.          */
int uem_segment_max_planar(const float* soft, const int64_t* sup, uint32_t* seg_keys, int B, int C,
                           int H, int W, int S, int* out_of_range, void* stream);
/* fused three-view refinement (alignment.py:209-293) for modes all / s / p / l:
 *   soft_out = normalise( weight * soft ), and per-(b,c) max of soft_out into plane_max (uint bits of
 *   non-negative floats; written, not accumulated) for the selection pass.  sim / logits are (B,h,w,C). */
#define UEM_REFINE_ALL 0
#define UEM_REFINE_S 1
#define UEM_REFINE_P 2
#define UEM_REFINE_L 3
int uem_label_refine(const float* soft, const int64_t* sup, const float* sim, const float* logits1,
                     const float* logits2 /* may be NULL */, const uint32_t* seg_keys,
                     const int64_t* ignore_id /* device */, float* soft_out, uint32_t* plane_max,
                     float* workspace /* uem_label_refine_workspace_floats(B,C,H,W,S) */, int B, int C, int h, int w,
                     int H, int W, int S, float temp, int mode, void* stream);
/* uem_label_refine followed by uem_pseudo_select on the refined map (train_ssl_uem.py:209-214; alignment.py:194-293 then
 * pseudo_generation.py:62-84), three launches: the refinement kernel also leaves each pixel's candidate (the one class above cutoff_low
 * and its value) in cand_workspace (uem_label_refine_select_workspace_bytes(B, H, W) bytes, 16-byte aligned) and the selection pass
 * reduces the blocks' maxima itself and reads 5 bytes per pixel instead of 4 * C.  hard: (B, H, W) int64.  Bit-identical to the two
 * entries called in sequence.  Needs H * W % 4 == 0 (UEM_ERR_UNSUPPORTED otherwise, nothing launched).                         */
int64_t uem_label_refine_select_workspace_bytes(int B, int H, int W);
int uem_label_refine_select(const float* soft, const int64_t* sup, const float* sim, const float* logits1, const float* logits2,
                            const uint32_t* seg_keys, const int64_t* ignore_id, float* soft_out, uint32_t* plane_max, float* workspace,
                            void* cand_workspace, int64_t* hard, int B, int C, int h, int w, int H, int W, int S, float temp, int mode,
                            float cutoff_top, float cutoff_low, int64_t ignore_label, void* stream);
int64_t uem_label_refine_workspace_floats(int B, int C, int H, int W, int S);
/* per-(b,c) max over H*W of an NCHW map                                pseudo_generation.py:76     */
int uem_plane_max(const float* mask, uint32_t* plane_max, int B, int C, int64_t HW, void* stream);
/* hard[b][p] = the unique c with mask > max(cutoff_top*max_c, cutoff_low), else ignore
 * pseudo_generation.py:76-88.  range_flag (device int, caller zero-fills) is set to 1 if any value
 * is outside [0,1] (the reference asserts, pseudo_generation.py:71).                               */
int uem_pseudo_select(const float* mask, const uint32_t* plane_max, int64_t* hard, int* range_flag, int B,
                      int C, int64_t HW, float cutoff_top, float cutoff_low, int64_t ignore_label, void* stream);
/* DownscaleLabel: majority vote over scale x scale cells             alignment.py:484-509          */
int uem_downscale_label(const int64_t* label, int64_t* out, int B, int H, int W, int scale, int n_classes,
                        int64_t ignore_label, float min_ratio, void* stream);
/* superpixel edge shrinking: keep an id only where the whole (2*win+1)^2 window (clipped) agrees, else ignore_id
 * (gast/superpixels.py:129-152; label and out are (B,H,W) int32, the on-disk .tif dtype; out != label)            */
int uem_superpixel_shrink(const int32_t* label, int32_t* out, int B, int H, int W, int win_size, int32_t ignore_id,
                          void* stream);
/* class-masked feature sums: sums[c][k], counts[c]                    alignment.py:340-348         */
int uem_proto_sums(const float* feat, const int64_t* label_ds, float* sums, float* counts,
                   float* workspace /* >= UEM_PROTO_SPLIT*C*k + UEM_PROTO_SPLIT*C floats */, int n, int k,
                   int C, int64_t ignore_label, void* stream);
#define UEM_PROTO_SPLIT 256
/* local = sums/(n+eps) (kept where n<1); protos = (1-decay)*local + decay*protos   alignment.py:348-353 */
int uem_proto_ema(const float* sums, const float* counts, float* protos, int k, int C, float decay, void* stream);

/* ---- losses: bilinear(align_corners=True) upsample + CE / UVEM, forward AND backward in one pass -------
 * logits are low-res (B,h,w,C); the full-resolution logits are never materialised.
 * CE (tools.py:240-254 + balance.py:88-101): loss = mean over ALL pixels of CE(ignore);
 *   dlogits (B,h,w,C) receives d(loss*loss_scale)/dlogits (written, not accumulated).
 *   workspace: >= uem_loss_workspace_floats(B,C,h,w) floats.                                      */
int uem_ce_upsampled(const float* logits1, const float* logits2 /* NULL: one head */, const int64_t* label,
                     const float* pixel_weight /* NULL */, float* loss_out /* [1] = mean over heads */,
                     float* dlogits1, float* dlogits2, float* workspace, int B, int C, int h, int w, int H, int W,
                     int64_t ignore_label, float loss_scale, void* stream);
/* UVEM (balance.py:356-423,437-451): u = entropy(soft); gate u>t; weight w(u); CE on `hard`;
 *   loss = sum(w*ce) / (#{u<=t & hard!=ignore} + 1e-7).  One call handles BOTH heads (logits2 may
 *   be NULL): loss_out[0] = mean over heads; dlogits1/2 get the gradient of loss*loss_scale.      */
int uem_uvem_upsampled(const float* logits1, const float* logits2, const int64_t* hard, const float* soft,
                       const float* pixel_weight /* NULL */, float* loss_out, float* dlogits1,
                       float* dlogits2, float* workspace, int B, int C, int h, int w, int H, int W,
                       float m, float t, float gamma, int64_t ignore_label, float loss_scale, void* stream);
int64_t uem_loss_workspace_floats(int B, int C, int h, int w); /* per-band gradient images + loss partials */
/* a[i] *= *scalar (and b[i] when b != NULL); scalar lives on the device (no host sync)             */
int uem_scale_by_scalar(float* a, float* b, int64_t n, const float* scalar, void* stream);
/* eval-mode output: (softmax(up(x1)) + softmax(up(x2)))/2 -> NCHW prob        Encoder.py:153-155   */
int uem_upsample_softmax_avg(const float* logits1, const float* logits2, float* prob, int B, int C, int h,
                             int w, int H, int W, void* stream);
/* uvem sample weight w(u) on a vector (UVEMLoss.get_weight, balance.py:396-423)                    */
int uem_uvem_weight(const float* u, float* w, int64_t n, float m, float t, float gamma, void* stream);
/* class histogram of a label map (ClassBalance._local_freq, balance.py:45-53): counts[C+1] (+=)     */
int uem_class_count(const int64_t* label, int64_t n, int C, int64_t ignore_label, float* counts, void* stream);
int uem_class_weight_gather(const int64_t* label, const float* class_w, float* out, int64_t n, int C,
                            int64_t ignore_label, void* stream);

/* ---- rows either side of the step: sliding-window / TTA inference, evaluation, prototype init ----
 * window_accumulate: dst[b,c,y1+i,x1+j] += src[b,c,i,j], cnt[b,y1+i,x1+j] += 1     tools.py:91-92
 *   (src may be a padded tile of size src_h x src_w >= th x tw; tools.py:77,90)
 * window_normalize: dst /= cnt                                                      tools.py:94          */
int uem_window_accumulate(float* dst, float* cnt, const float* src, int B, int C, int H, int W, int y1, int x1,
                          int th, int tw, int src_h, int src_w, void* stream);
int uem_window_normalize(float* dst, const float* cnt, int B, int C, int H, int W, void* stream);
int uem_scale(float* a, int64_t n, float s, void* stream);
/* pred[b,p] = argmax_c prob[b,c,p] (NULL to skip) and cm[gt][pred] += 1 over pixels with 0 <= gt < C
 * (cm: C*C int64, caller zero-fills; gt/cm NULL to skip)             utils/eval.py:41-50, metrics.py:26 */
int uem_argmax_confusion(const float* prob, const int64_t* gt, int64_t* pred, int64_t* cm, int B, int C, int64_t HW,
                         void* stream);
/* prototypes = sums / (counts + 1e-7)   (Aligner.init_avg, alignment.py:121-122)                       */
int uem_proto_mean(const float* sums, const float* counts, float* protos, int k, int C, void* stream);

/* ---- stage-2 alignment losses, forward + backward fused ----------------------------------------
 * PrototypeContrastiveLoss (uemda/loss.py:10-47): rows with label == ignore are dropped; feat (n,k) and the
 * prototypes (C,k) are L2-normalised (eps 1e-12), logits = feat.Proto^T / temperature, loss = mean CE.
 * dfeat (n,k) receives d loss / d feat (zero rows for ignored labels).                                     */
int uem_pcl_loss(const float* protos, const float* feat, const int64_t* labels, float* loss_out /* [1] */, float* dfeat,
                 float* workspace /* uem_pcl_workspace_floats(k, C) */, int n, int k, int C, float temperature,
                 int64_t ignore_label, void* stream);
int64_t uem_pcl_workspace_floats(int k, int C);
/* CoralLoss (uemda/gast/coral.py:15-47).  gram_* = X^T X (d x d, from uem_conv2d_wgrad with x = dy = the (n,d)
 * features), mean_* = column means.  Outputs the loss and the two pre-scaled symmetric matrices of the backward:
 * d source = (source - mean_s) . g_s,  d target = (target - mean_t) . g_t  (1x1 uem_conv2d_fwd with the affine
 * prologue scale = 1, shift = -mean).                                                                       */
int uem_coral_finish(const float* gram_s, const float* gram_t, const float* mean_s, const float* mean_t, int ns, int nt,
                     int d, float* g_s, float* g_t, float* loss_out, float* partial /* >= 1024 floats */, void* stream);
int uem_negate(const float* a, float* b, int n, void* stream);

/* ---- optimizer: clip_grad_norm_(max_norm, L2) + SGD(momentum, weight_decay) over a flat arena ----------
 * train_ssl_uem.py:169-170,228-232.  sqnorm: partial sums (>= UEM_NORM_BLOCKS floats) -> norm_out[0].  */
#define UEM_NORM_BLOCKS 1024
int uem_grad_sqnorm(const float* grad, int64_t n, float* partial, float* norm_out, void* stream);
int uem_sgd_clip_step(float* param, float* grad, float* momentum_buf, int64_t n, const float* norm /* device */,
                      float max_norm, float lr, float momentum, float weight_decay, int first_step,
                      float grad_prescale, const float* lr_dev /* NULL, or the learning rate as a device scalar (overrides lr):
                      what a step captured in a hipGraph needs, by-value arguments being baked into the capture */, void* stream);

/* ---- bf16 STORAGE (BASELINE config 5: "bf16 weights, CDNA4 bf16 MFMA") -------------------------------------------
 * Activations and weights bf16 in HBM (uint16_t = the raw bf16 bits, NHWC / OHWI as above), fp32 accumulation on
 * v_mfma_f32_32x32x16_bf16, outputs rounded to nearest-even.  There is no operand prologue on this path: BatchNorm +
 * ReLU are materialised by uem_bn_apply_bf16 (same HBM bytes per activation as fp32 storage + prologue).  Replaces the
 * same cuDNN call sites as uem_conv2d_fwd.
 * uem_conv2d_bf16: forward, or with UEM_CONV_TRANSPOSED the data gradient (x = dY, w = transposed weights, y = dX, shape =
 *   the FORWARD conv); UEM_CONV_ACCUMULATE: y += result.  tile_stats (forward, may be NULL): [2][Cout][M/128] per-tile
 *   column sums of the STORED (rounded) y and y*y for uem_bn_stats_from_tiles.  Needs reduction channels % 64 == 0,
 *   output channels % 64 == 0.                                                                                     */
int uem_conv2d_bf16(const uint16_t* x, const uint16_t* w, uint16_t* y, const uem_conv_shape* s, int flags,
                    float* tile_stats, void* stream);
/* bf16 twin of uem_conv2d_dgrad_bnbwd / uem_conv2d_dgrad_tail (same argument meaning and restrictions: stride 1,
 * N*H*W % 128 == 0, Cin % 64 == 0): dx = dgrad(dy) [+ acc_src * [acc_bits]] (or dx += with UEM_CONV_ACCUMULATE), and with
 * bn_z / bn_vec (4, Cin) / tile_partials [2][Cin][M/128] the per-tile sums of dp and dp*xhat over the ROUNDED dx, dp = dx
 * masked by bn_bits (packed) or by [bn_z*scale + shift > 0].  Tensors bf16; vectors, partial sums fp32.             */
int uem_conv2d_dgrad_tail_bf16(const uint16_t* dy, const uint16_t* w_t, uint16_t* dx, const uem_conv_shape* s,
                               const uint16_t* acc_src, const uint32_t* acc_bits, const uint16_t* bn_z, const float* bn_vec,
                               const uint32_t* bn_bits, float* tile_partials, int flags, void* stream);
/* dw[o][tap][i] (fp32, the gradient arena) += sum_m dY[m][o] * x[m'(m,tap)][i] with bf16 x and dY: 1x1 layers (stride 1; stride
 * 2 on output rows that are a multiple of 32 pixels) and 3x3 layers (stride 1 dilation 1 / 2, stride 2) on such rows;
 * channel counts % 64 == 0 (UEM_ERR_UNSUPPORTED otherwise).                                                          */
int uem_conv2d_wgrad_bf16(const uint16_t* x, const uint16_t* dy, float* dw, const uem_conv_shape* s, void* stream);
/* BatchNorm / residual passes on bf16 tensors (statistics, scale / shift, gradients of gamma / beta stay fp32):
 * the bf16 twins of uem_affine_act (relu(bn(z)) materialised for the next conv; with `res` the block output and its packed
 * ReLU bits), uem_bn_bwd_reduce and uem_bn_bwd_apply.  relu: 0 = none, 1 = mask recomputed from x*scale+shift > 0 (pass no
 * bits), UEM_RELU_BITS = packed output mask of uem_affine_act_bf16.                        _resnets.py:96-112          */
int uem_affine_act_bf16(const uint16_t* x, const float* scale, const float* shift, const uint16_t* res /* may be NULL */,
                        const float* res_scale, const float* res_shift, uint16_t* y, int64_t M, int C, int relu,
                        uint32_t* relu_bits /* may be NULL */, void* stream);
int uem_bn_bwd_reduce_bf16(const uint16_t* x, const uint16_t* dy, const uint32_t* relu_bits, const float* scale,
                           const float* shift, const float* save_mean, const float* save_invstd, int M, int C, int relu,
                           float* dgamma, float* dbeta, float* grad_gamma /* += */, float* grad_beta /* += */,
                           float* workspace /* uem_bn_workspace_floats(M,C) */, void* stream);
int uem_bn_bwd_apply_bf16(const uint16_t* x, const uint16_t* dy, const uint32_t* relu_bits, const float* scale,
                          const float* shift, const float* save_mean, const float* save_invstd, const float* dgamma,
                          const float* dbeta, int M, int C, int relu, uint16_t* dx, uint16_t* dres /* dy*mask, may be NULL */,
                          void* stream);
/* fp32 <-> bf16 (round to nearest even): the bf16 copy of the fp32 master weights after every optimizer step, and the two
 * ends of the bf16-storage region of the network (max-pool output in, layer4 output out).                             */
int uem_weight_transpose_bf16(const float* w /* [Cout][KH][KW][Cin] fp32 master */, uint16_t* wt /* [Cin][KH][KW][Cout] bf16 */,
                              int Cout, int KH, int KW, int Cin, void* stream);
int uem_cast_f32_bf16(const float* x, uint16_t* y, int64_t n, void* stream);
int uem_cast_bf16_f32(const uint16_t* x, float* y, int64_t n, void* stream);

/* ---- Winograd F(2x2, 3x3) and F(4x4, 3x3) for the stride-1 3x3 convolutions of the deep layers (csrc/winograd.hip) ------
 * Same cuDNN call sites as uem_conv2d_fwd for the 3x3 convs (reference uemda/_resnets.py:100-103 -- conv2 of a Bottleneck,
 * dilation 1 or 2 under output stride 16 -- and Encoder.py:35, the PPM head's 4096 -> 512 conv).  Exact fp32 arithmetic:
 * Y = A^T [(G g G^T) (.) (B^T d B)] A.  `m` is the output tile edge: 2 = F(2x2,3x3), 16 positions, 2.25x fewer multiplies, 4e-7
 * relative L2 against float64; 4 = F(4x4,3x3) on the points (0, 1, -1, 1/2, -2, inf), 36 positions, 4x fewer multiplies, 1.4e-6.
 * Transform-domain tensors are [npos][T][C] fp32, npos = (m+2)^2, T = N * H * W / m^2 tiles (dilation d: tiles of the d*d
 * interleaved sub-images).  Needs H, W % (m*dil) == 0, C % 64 == 0, T % 32 == 0 (the GEMMs: T % 128 == 0).
 *   uem_wino_filter      U[npos][Cout][Cin] = G w G^T from w (Cout,3,3,Cin); transposed != 0: U'[npos][Cin][Cout] of the flipped
 *                        taps, the filter bank of the data gradient dX = conv(dY, W'), W'[ci][ky][kx][co] = W[co][2-ky][2-kx][ci]
 *   uem_wino_input       V = B^T d B of x (N,H,W,C), optionally through relu(x*in_scale + in_shift) first (the producer's
 *                        BatchNorm, as the conv kernels' operand prologue; zero padding applies AFTER it)
 *   uem_wino_gemm        M[npos][T][N] = V[npos][T][K] x U[npos][N][K]^T: the npos products as one launch of the f32-MFMA 1x1 kernel
 *   uem_wino_output      y (N,H,W,C) = A^T M A; tile_stats [2][C][N*H*W/128]: per-128-pixel sums of y and y*y (forward; for
 *                        uem_bn_stats_from_tiles), OR bn_z / bn_vec (4,C) / tile_bnbwd [2][C][N*H*W/128]: per-group sums of
 *                        dp = y*[bn_z*scale + shift > 0] and dp*xhat (data gradient; for uem_bn_bwd_from_tiles); all may be NULL
 *   uem_wino_dy          dM[npos][T][C] = A dY A^T (weight gradient); optionally clears dU in the same launch
 *   uem_wino_wgrad_gemm  dU[npos][N][K] += sum_tiles dM[pos][tile][n] * V[pos][tile][k] (split-K fp32 atomics: zero dU first)
 *   uem_wino_filter_grad dw (Cout,3,3,Cin) += G^T dU G                                                                   */
int uem_wino_filter(const float* w_ohwi, float* U, int Cout, int Cin, int transposed, int m, void* stream);
int uem_wino_input(const float* x, const float* in_scale, const float* in_shift, int relu, float* V, int N, int H, int W, int C,
                   int dil, int m, int pass /* 0 forward, 1 data gradient (x = dY), 2 weight gradient recomputing V: same arithmetic,
                   the kernel instantiation is named after the pass so that profiles attribute it */, void* stream);
int uem_wino_gemm(const float* V, const float* U, float* M, int T, int K, int N, int npos /* 16 or 36 */,
                  int data_gradient /* 0 forward, 1: the same product through the data-gradient kernel instantiation */, void* stream);
int uem_wino_output(const float* M, float* y, int N, int H, int W, int C, int dil, int m, float* tile_stats, const float* bn_z,
                    const float* bn_vec, float* tile_bnbwd, void* stream);
int uem_wino_dy(const float* dy, float* dM, int N, int H, int W, int C, int dil, int m, float* zero /* optional: a buffer the launch also
                clears -- the dU accumulator of uem_wino_wgrad_gemm -- */, int64_t zero_floats, void* stream);
int uem_wino_wgrad_gemm(const float* V, const float* dM, float* dU, int T, int K, int N, int npos, void* stream);
int uem_wino_filter_grad(const float* dU, float* dw_ohwi, int Cout, int Cin, int m, void* stream);

/* ---- every per-step weight re-layout in one launch ----------------------------------------------------------------------
 * The kernels read their filter banks in layouts derived from the OIHW / OHWI parameters (reference: nn.Conv2d weights,
 * uemda/_resnets.py:21-29): the transposed bank of a direct data gradient (uem_weight_transpose), the Winograd banks U / U'
 * (uem_wino_filter, m = 2 / 4), the stem's padded taps (uem_stem_pack_weight).  After an optimizer step all of them are stale at once;
 * uem_weight_prep refreshes a whole table of such jobs in ONE launch.  `jobs` and `block_starts` (njobs + 1 prefix sums of
 * uem_weight_prep_blocks over the jobs) live in DEVICE memory; the caller owns them and every src / dst buffer.                  */
#define UEM_PREP_TRANSPOSE 0   /* src w[cout][taps][cin] -> dst wt[cin][taps][cout]                                        */
#define UEM_PREP_WINO2 1       /* src w[cout][3][3][cin] -> dst U[16][cout][cin]      (uem_wino_filter, m = 2)             */
#define UEM_PREP_WINO2_T 2     /*                        -> dst U'[16][cin][cout], flipped taps                            */
#define UEM_PREP_WINO4 3       /*                        -> dst U[36][cout][cin]      (m = 4)                              */
#define UEM_PREP_WINO4_T 4     /*                        -> dst U'[36][cin][cout]                                          */
#define UEM_PREP_STEM_PACK 5   /* src w[64][7][7][3] -> dst w8[64][7][8][4] (cout 64, cin 3, taps 49)                       */
#define UEM_PREP_TRANSPOSE_BF16 6 /* as UEM_PREP_TRANSPOSE, dst in bf16 (round to nearest even): uem_weight_transpose_bf16         */
typedef struct uem_prep_job {
    const void* src;
    void* dst;
    int kind, cout, cin, taps;
} uem_prep_job;
int uem_weight_prep_blocks(int kind, int cout, int cin, int taps);   /* blocks the job takes; -1: shape not supported by this kind */
int uem_weight_prep(const uem_prep_job* jobs /* device */, const int* block_starts /* device, njobs + 1 */, int njobs,
                    int total_blocks, void* stream);

/* ---- data parallel (new relative to the reference, which is single-GPU: SURVEY 2a, 8e) --------------------------
 * all-reduce(sum, in place) of a flat fp32 buffer -- the gradient arena, 98 MB for R50-ASPP -- over RCCL on `stream`:
 * one process per GPU, xGMI inside the node.  Rank 0 draws an id with uem_comm_unique_id, the host ships those 128
 * bytes to every rank (any side channel: uemda_amd.dp uses the torch.distributed store), every rank calls
 * uem_comm_init.  This is the library's only state (the communicator handles the caller holds); there is nothing else
 * to shut down, which is why the header has no uem_shutdown().  RCCL is taken from the host process (PyTorch-ROCm
 * carries its own librccl.so) and loaded from /opt/rocm only when the process has none.
 * uemda_amd.dp uses torch.distributed (backend "nccl" = the same RCCL) as its ONLY transport (round 4 removed its
 * UEM_DP_NATIVE switch); these entry points are for hosts without torch.distributed and are driven directly by
 * tests/test_gpu_dp.py (one rank: with one GPU per box the multi-rank leg cannot be rehearsed here).                 */
int uem_comm_unique_id(void* id_out_128_bytes /* host */);
int uem_comm_init(void** comm_out, const void* id_128_bytes /* host */, int rank, int world);
int uem_allreduce_flat(void* comm, float* buf /* device, in place */, int64_t count, void* stream);
int uem_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* UEMDA_HIP_H */
