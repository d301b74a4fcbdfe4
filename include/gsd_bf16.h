/*
 * gsd_bf16.h -- C ABI of libgsd.so, bf16 mixed-precision family (BASELINE.json configs[4]: "bf16 mixed-precision
 * training ... with MFMA im2col conv path").  Same conventions as gsd.h (caller-owned memory, explicit stream, negative
 * status + gsd_last_error(), re-entrant).  The reference has no reduced-precision path; these entry points stand in for
 * the same ATen operators as their fp32 counterparts in gsd.h (cited per function), with
 *   - activations and activation gradients stored as bfloat16, NHWC ("pixel-major"), described by gsd_nhwc;
 *   - fp32 accumulation in every contraction and reduction, fp32 BatchNorm statistics;
 *   - fp32 master parameters, gradients, Adam and EMA state (gsd_adam_ema is shared with the fp32 path).
 */
#ifndef GSD_BF16_H
#define GSD_BF16_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* A bf16 NHWC tensor, possibly a channel slice of a wider buffer: element (n,h,w,c) lives at
 * ptr + (((n*H + h)*W + w) * pitch + c).  ptr 16-byte aligned, pitch a multiple of 8. */
typedef struct {
  void* ptr;
  int64_t pitch;   /* elements from one pixel to the next (>= C) */
  int32_t N, H, W, C;
} gsd_nhwc;

/* Optional fusion for dX launches: the output is the gradient w.r.t. the activation a = relu(bn(y)) of a conv+BN+ReLU
 * unit, so pass 1 of that unit's BatchNorm+ReLU backward runs in the epilogue: out <- dz = dX where y*scale+shift > 0
 * else 0, and `partials` receives [sum dz | sum dz*xhat] per block in the layout of the forward statistics (reduce with
 * gsd_bn_reduce_partials, then gsd_bn_bwd_finalize).  Replaces gsd_bf16_bn_bwd_reduce(mode 0) and its 8 B/element. */
typedef struct {
  const gsd_nhwc* y;    /* raw conv output of the unit, same (N,H,W,C) as the launch's out */
  const float* scale;
  const float* shift;
  const float* mean;
  const float* invstd;
} gsd_bf16_bnbwd;

/* Row padding of the m (output channel) dimension in every bf16 weight image: gsd_bf16_conv_mpad(M) rows. */
int gsd_bf16_conv_mpad(int M);

/* conv3x3 p1 s1, no bias (unet.py:11,14; with a dgrad weight image the dX half of its backward):
 *   out[n,h,w,m] = sum_{t,k} in[n,h+t/3-1,w+t%3-1,k] * wt[t][m][k];  wt: [9][mpad(M)][K] bf16, K % 32 == 0, M % 16 == 0.
 * partials (or NULL): BatchNorm partial sums of the STORED (rounded) values, one row of 2*mpad(M) floats per (persistent
 * block, wave): gsd_bf16_conv_partial_rows rows (a few hundred, it depends on the device's CU count), to be reduced
 * with gsd_bn_reduce_partials(partials, rows, mpad(M), M, ...) from gsd.h. */
int gsd_bf16_conv_partial_rows(int N, int H, int W, int M);
int gsd_bf16_conv3x3(const gsd_nhwc* in, const void* wt, const gsd_nhwc* out, int K, int M, float* partials,
                     const gsd_bf16_bnbwd* bw, void* stream);

/* Eval-mode "3x3 conv + BatchNorm + ReLU" in one kernel (unet.py:11-13 / :14-16 under model.eval()): scale/shift are the
 * running-statistics coefficients from gsd_bn_eval_coeffs; out = relu(conv3x3(in) * scale[m] + shift[m]) is written
 * directly (the raw convolution output is never stored).  gsd_bf16_conv1x1_bnrelu is the same for the first layer on
 * the im2col'd input. */
int gsd_bf16_conv3x3_bnrelu(const gsd_nhwc* in, const void* wt, const gsd_nhwc* out, int K, int M, const float* scale,
                            const float* shift, void* stream);
int gsd_bf16_conv1x1_bnrelu(const gsd_nhwc* in, const void* wt, const gsd_nhwc* out, int K, int M, const float* scale,
                            const float* shift, void* stream);

/* Taps without spatial reuse on an (N,H,W) pixel grid:
 *   acc[n,h,w,m] = sum_{t<ntaps} sum_k in[n, stride*h+ty[t], stride*w+tx[t], k] * wt[t][m][k]   (zeros outside in)
 * scatter_cs == 0: out[n,h,w,m] = acc (+ bias[m]).   -- 1x1 convolution (first layer after gsd_bf16_im2col3x3)
 * scatter_cs  > 0: M == 4*scatter_cs, m = q*Cs + co: out[n, 2h+(q>>1)+oy, 2w+(q&1)+ox, co] = acc + bias[co]
 *                  -- ConvTranspose2d(k=2,s=2) forward written straight into its F.pad position inside the concat
 *                     buffer (unet.py:36,41,46-48).
 * With 4 taps at stride 2 and a dgrad weight image it is the dX of that ConvTranspose2d. */
int gsd_bf16_conv_dense(const gsd_nhwc* in, const void* wt, const gsd_nhwc* out, int K, int M, int ntaps, int stride,
                        const int* ty, const int* tx, int H, int W, int scatter_cs, int oy, int ox, const float* bias,
                        float* partials, const gsd_bf16_bnbwd* bw, void* stream);
/* BatchNorm partial rows a gsd_bf16_conv_dense launch of this shape writes (plain output, scatter_cs == 0): the transposed
 * convolutions' shapes (4 taps at stride 2, K % 64 == 0, M == 128 or M % 256 == 0) run on a large-tile kernel with its own
 * row count (csrc/gsd_bf16_ctgemm.hip; GSD_BF16_CTGEMM=0 keeps them on the general kernel), everything else writes
 * gsd_bf16_conv_partial_rows(N, H, W, M) rows. */
int gsd_bf16_conv_dense_partial_rows(int N, int H, int W, int K, int M, int ntaps, int stride);

/* ---- weight images (fp32 master -> bf16 GEMM layout [T][mpad(M)][round_up(K,32)], zero padded) ---------------------
 * mode 0 conv3x3 forward  from (Cout,Cin,3,3):  [t][co][ci]
 *      1 conv3x3 dX       from (Cout,Cin,3,3):  [t][ci][co] with the tap flipped (8-t)
 *      2 first layer on the im2col'd input:     [0][co][ci*9+t]
 *      3 ConvTranspose2d forward from (Cin,Cout,2,2): [0][(kh*2+kw)*Cout+co][ci]
 *      4 ConvTranspose2d dX:                         [kh*2+kw][ci][co]                                             */
int64_t gsd_bf16_weight_image_size(int mode, int Cout, int Cin);
int gsd_bf16_weight_image(int mode, const float* w, int Cout, int Cin, void* out, void* stream);
/* n images in one launch per 32 jobs (a train step wants 43: as separate launches between the convolutions they are pure
 * latency); same values as gsd_bf16_weight_image job by job.  `jobs` is a host array. */
typedef struct gsd_bf16_wimg_job {
  const float* w;   /* fp32 master weights */
  void* out;        /* gsd_bf16_weight_image_size(mode, Cout, Cin) bf16 elements */
  int32_t mode, Cout, Cin, reserved;
} gsd_bf16_wimg_job;
int gsd_bf16_weight_images(const gsd_bf16_wimg_job* jobs, int n, void* stream);

/* First layer WITHOUT the im2col tensor (unet.py:11 for `inc`, 9*C <= 32 and M in {32, 64}: gsd_bf16_conv3x3_first_supported):
 *   out[n,h,w,m] = sum_{c,t} bf16(x[n,c,h+t/3-1,w+t%3-1]) * wt[m][c*9+t]       x (N,C,H,W) fp32, wt: weight image mode 2
 * straight from x -- the layer is bound by the HBM write of its output (K = 27 is one MFMA k-step), and the im2col tensor
 * costs more traffic than the input.  Bit-identical to gsd_bf16_im2col3x3 + gsd_bf16_conv_dense.  partials (train): BatchNorm
 * partial sums, gsd_bf16_conv3x3_first_partial_rows rows of 2*gsd_bf16_conv_mpad(M) floats; ep_scale/ep_shift (eval): out =
 * relu(acc*scale+shift) instead of the raw output.  out == NULL (with partials): the statistics only, nothing is stored.
 * gsd_bf16_wgrad_first: dW (M,C,3,3) fp32 of the same layer from x and the gradient of its output.  With y != NULL the
 * BatchNorm backward of the layer's output is applied on the fly -- d_raw = scale*(dz - c1 - (y-mean)*invstd*c2), rounded to
 * bf16 exactly as gsd_bf16_bn_bwd_apply stores it: the first layer has no dX, so dW is d_raw's only reader and the apply
 * pass (and the im2col tensor) disappear.  y == NULL: dz already is d_raw.  workspace: gsd_bf16_wgrad_first_workspace floats
 * (one slab per block, summed in a fixed order: bitwise reproducible). */
int gsd_bf16_conv3x3_first_supported(int C, int M);
int gsd_bf16_conv3x3_first_partial_rows(int N, int H, int W, int M);
int gsd_bf16_conv3x3_first(const float* x, int N, int C, int H, int W, const void* wt, const gsd_nhwc* out, int M,
                           float* partials, const float* ep_scale, const float* ep_shift, void* stream);
int64_t gsd_bf16_wgrad_first_workspace(int N, int H, int W, int M);
int gsd_bf16_wgrad_first(const float* x, int N, int C, int H, int W, const gsd_nhwc* dz, const gsd_nhwc* y,
                         const float* scale, const float* mean, const float* invstd, const float* c1, const float* c2,
                         float* dw, float* workspace, int64_t workspace_elems, void* stream);

/* ---- the `inc` double convolution in train mode without the raw output of its first convolution ------------------------
 * (unet.py:7-20 for `inc`, :67: 3 -> 64 -> 64 at full resolution -- the north star's "320x427 double-conv kernel".)
 * Unfused, the block writes the first convolution's raw output y0, reads it back to write a0 = relu(bn(y0)), and reads a0
 * again (with halo overlap) in the 64 -> 64 convolution: 2.5x its algorithmic bytes.  Fused:
 *   1. gsd_bf16_conv3x3_first(x, ..., out = NULL, partials): the statistics of y0 from a write-free pass over x (same kernel,
 *      same partial rows as the storing form); gsd_bn_reduce_finalize turns them into (scale0, shift0, mean0, invstd0).
 *   2. gsd_bf16_inc_conv: per pixel tile, a0 is rebuilt over the tile's halo from x (one MFMA k-step per 16 pixels), rounded
 *      exactly as the two unfused passes round it, staged in LDS as the operand of the 64 -> 64 convolution, whose weights
 *      (wt1: weight image mode 0) stay resident in LDS.  Writes a0 once (the second layer's dW reads it in backward), y1
 *      (raw output of the second convolution) and y1's BatchNorm partial sums: gsd_bf16_inc_conv_partial_rows rows of
 *      2*gsd_bf16_conv_mpad(64) floats (gsd_bn_reduce_partials / gsd_bn_reduce_finalize from gsd.h).  a0 and y1 are bit-identical
 *      to gsd_bf16_conv3x3_first + gsd_bf16_bn_apply(relu) + gsd_bf16_conv3x3 under the same (scale0, shift0).
 *   3. backward of the first layer, y0 recomputed from x the same way:
 *      gsd_bf16_first_bn_bwd_reduce: pass 1 of its BatchNorm+ReLU backward from da (the gradient w.r.t. a0, as the second
 *        layer's plain dX launch leaves it): partials = [sum dz | sum dz*xhat] with dz = da where y0*scale+shift > 0, in the
 *        layout of the forward statistics (gsd_bf16_conv3x3_first_partial_rows rows); nothing is written back.
 *      gsd_bf16_wgrad_first_recompute: gsd_bf16_wgrad_first with (da, wt) in place of (dz, y): the mask, xhat and
 *        d_raw = scale*(dz - c1 - xhat*c2) are formed from the recomputed y0 -- the same d_raw values, bit for bit.
 * Supported: 9*C <= 32 and 64 channels (gsd_bf16_inc_supported); the first-layer pieces also serve M = 32. */
int gsd_bf16_inc_supported(int C, int M);
int gsd_bf16_inc_conv_partial_rows(int N, int H, int W);
int gsd_bf16_inc_conv(const float* x, int N, int C, int H, int W, const void* wt0, const float* scale0, const float* shift0,
                      const void* wt1, const gsd_nhwc* a0, const gsd_nhwc* y1, float* partials, void* stream);
int gsd_bf16_first_bn_bwd_reduce(const float* x, int N, int C, int H, int W, const void* wt, const gsd_nhwc* da,
                                 const float* scale, const float* shift, const float* mean, const float* invstd,
                                 float* partials, void* stream);
int gsd_bf16_wgrad_first_recompute(const float* x, int N, int C, int H, int W, const void* wt, const gsd_nhwc* da,
                                   const float* scale, const float* shift, const float* mean, const float* invstd,
                                   const float* c1, const float* c2, float* dw, float* workspace, int64_t workspace_elems,
                                   void* stream);

/* ---- conv3x3 64 -> 64 channels with the weights resident in LDS -------------------------------------------------------------
 * The 64-channel 3x3 convolutions of the first level (unet.py:14 for `inc` and `up.3.conv`, forward and dX) as a persistent
 * kernel that loads its 72 KiB of weights once per CU and fills the whole 64-channel halo of a pixel tile by one LDS-DMA fill
 * under the previous tile's epilogue: no operand fill and no barrier inside the K loop.  Same products in the same order as
 * gsd_bf16_conv3x3 (outputs bit-identical); `wt`, `partials` (gsd_bf16_conv3x3_c64_partial_rows rows of 2*gsd_bf16_conv_mpad(64)
 * floats, one per block; NULL: a plain convolution) and `bw` as there. */
int gsd_bf16_conv3x3_c64_supported(int K, int M);
int gsd_bf16_conv3x3_c64_partial_rows(int N, int H, int W);
int gsd_bf16_conv3x3_c64(const gsd_nhwc* in, const void* wt, const gsd_nhwc* out, float* partials, const gsd_bf16_bnbwd* bw,
                         void* stream);

/* First layer: x (N,C,H,W) fp32 NCHW -> col (N,H,W,round_up(9C,32)) bf16 with col[..,c*9+t] = x[n,c,h+t/3-1,w+t%3-1]
 * (zero padded), so that conv3x3(x) is a 1x1 convolution of col (gsd_bf16_conv_dense). unet.py:11 for `inc`. */
int gsd_bf16_im2col3x3(const float* x, int N, int C, int H, int W, const gsd_nhwc* col, void* stream);

/* a = y*scale[c] + shift[c], then max(0,.) if relu: BatchNorm2d (batch or running statistics folded into scale/shift
 * by gsd_bn_finalize / gsd_bn_eval_coeffs) + ReLU (unet.py:12-13,15-16). `a` may be a channel slice of a concat buffer. */
int gsd_bf16_bn_apply(const gsd_nhwc* y, const float* scale, const float* shift, const gsd_nhwc* a, int relu, void* stream);

/* The two above in one pass where a unit's activation feeds a max-pool (the encoder's skip units, unet.py:15-16 then :26):
 * a = relu(y*scale+shift) is written (into the concat buffer's skip slice) AND pooled = MaxPool2d(2)(a), floor mode, from the
 * same read of y -- the stand-alone pool's re-read of the activation disappears.  Bit-identical to gsd_bf16_bn_apply(relu=1)
 * followed by gsd_bf16_maxpool2. */
int gsd_bf16_bn_apply_pool(const gsd_nhwc* y, const float* scale, const float* shift, const gsd_nhwc* a,
                           const gsd_nhwc* pooled, void* stream);
/* ... and, for the backward, which of its window's four activations each pooled value is: idx = (N, H/2, W/2, C/8) uint16, two
 * bits per channel (channel c of a group in bits 2c..2c+1; 0..3 = (0,0),(0,1),(1,0),(1,1), the first maximum of the stored
 * values).  gsd_bf16_bn_bwd_reduce_pool_idx then routes the pooled gradient from 2 bytes per (window, 8 channels) instead
 * of re-reading the window's four activations (64 bytes).  idx == NULL: gsd_bf16_bn_apply_pool. */
int gsd_bf16_bn_apply_pool_idx(const gsd_nhwc* y, const float* scale, const float* shift, const gsd_nhwc* a,
                               const gsd_nhwc* pooled, void* idx, void* stream);

/* MaxPool2d(2), floor mode (unet.py:26). */
int gsd_bf16_maxpool2(const gsd_nhwc* a, const gsd_nhwc* pooled, void* stream);

/* OutConv: out (N,K,H,W) fp32 NCHW = conv1x1(a; w (K,C), bias) (unet.py:54), K <= 4. */
int gsd_bf16_conv1x1_out(const gsd_nhwc* a, const float* w, const float* bias, int K, float* out, void* stream);
/* The same with the last unit's BatchNorm + ReLU folded in (train mode): y is that unit's RAW output, scale / shift its batch
 * coefficients; the activation bf16(relu(y*scale+shift)) -- bit for bit what gsd_bf16_bn_apply stores -- is formed in registers
 * and never written (its only other reader, the backward's gsd_bf16_bn_bwd_reduce mode 2, recomputes it from y too). */
int gsd_bf16_bn_relu_conv1x1_out(const gsd_nhwc* y, const float* scale, const float* shift, const float* w, const float* bias, int K,
                                 float* out, void* stream);

/* BatchNorm+ReLU backward, pass 1 (+ the adjoining max-pool / output-conv backward), as gsd_bn_bwd_reduce in gsd.h:
 *   da = g                                  (mode 0)
 *      = g + maxpool2 backward of dpool     (mode 1: arg-max over the STORED activations a, first maximum wins)
 *      = dout[n,0,h,w] * wout[c]            (mode 2: n_classes == 1; third partial = sum dout * a -> dW of the OutConv)
 *   dz = da where y*scale+shift > 0 else 0, written to `dz` (may alias g);
 *   partials: gsd_bf16_bn_bwd_partial_rows rows of [sum dz | sum dz*xhat | third] (3*C floats), to be reduced with
 *   gsd_bn_bwd_reduce_partials / gsd_bn_bwd_finalize from gsd.h. */
int gsd_bf16_bn_bwd_partial_rows(int N, int H, int W);
int gsd_bf16_bn_bwd_reduce(int mode, const gsd_nhwc* y, const float* scale, const float* shift, const float* mean,
                           const float* invstd, const gsd_nhwc* g, const gsd_nhwc* a, const gsd_nhwc* dpool,
                           const float* dout, const float* wout, const gsd_nhwc* dz, float* partials, void* stream);
/* mode 1 with the arg-max codes of gsd_bf16_bn_apply_pool_idx instead of the activations; same dz, same sums. */
int gsd_bf16_bn_bwd_reduce_pool_idx(const gsd_nhwc* y, const float* scale, const float* shift, const float* mean,
                                    const float* invstd, const gsd_nhwc* g, const void* pool_idx, const gsd_nhwc* dpool,
                                    const gsd_nhwc* dz, float* partials, void* stream);
/* pass 2, in place: dz <- scale * (dz - c1 - xhat * c2), the gradient w.r.t. the raw convolution output. */
int gsd_bf16_bn_bwd_apply(const gsd_nhwc* dz, const gsd_nhwc* y, const float* scale, const float* mean, const float* invstd,
                          const float* c1, const float* c2, void* stream);

/* out[c] = sum over n and the window rows [y0,y0+hh) x cols [x0,x0+ww) of t[n,y,x,c] (fp32): the bias gradient of
 * ConvTranspose2d over the un-padded part of its output slice (unet.py:36). workspace: ..._workspace(N,hh,ww,C) floats. */
int64_t gsd_bf16_channel_sums_workspace(int N, int hh, int ww, int C);
int gsd_bf16_channel_sums(const gsd_nhwc* t, int y0, int x0, int hh, int ww, float* out, float* workspace,
                          int64_t workspace_elems, void* stream);
/* The same bias gradient without a pass over the gradient slice: the dX launch that WROTE the slice (the decoder's first
 * convolution, gsd_bf16_conv3x3 with `partials` and no `bw`) left per-channel sums of the whole plane in its statistics rows
 * (`rows` rows of `ld` floats; the slice's channel c is column col0 + c); what lies outside the transposed convolution's window
 * -- the F.pad border, (oy,ox)+(hh,ww) inside g's (H,W) -- is summed from g (thin strips) and subtracted.
 * out[c] = sum_rows partials[r][col0+c] - sum_{n, (y,x) outside the window} g[n,y,x,c]. */
int64_t gsd_bf16_convT_bias_grad_workspace(int N, int H, int W, int oy, int ox, int hh, int ww, int C);
int gsd_bf16_convT_bias_grad(const float* partials, int rows, int ld, int col0, const gsd_nhwc* g, int oy, int ox, int hh, int ww,
                             float* out, float* workspace, int64_t workspace_elems, void* stream);

/* ---- weight gradients --------------------------------------------------------------------------------------------
 *   D[t][m][n] = sum_{n_img,h,w} a[n_img,h,w,m] * b[n_img, stride*h+ty[t], stride*w+tx[t], n]   (zeros outside b)
 *   dw[(m*ncols_out + n)*ntaps + t] = D[t][m][n]   for n < ncols_out          (fp32; overwritten, not accumulated)
 * conv3x3 (unet.py:11,14):  a = dy, b = layer input, 9 taps (t/3-1, t%3-1)      -> dw == dW (Cout,Cin,3,3)
 * first layer:              a = dy, b = im2col'd input, 1 tap, ncols_out = 9*Cin -> dw == dW (Cout,Cin,3,3)
 * ConvTranspose2d (unet.py:36): a = x, b = gradient of its (padded) output slice, taps (kh+oy, kw+ox) at stride 2
 *                                                                               -> dw == dW (Cin,Cout,2,2)
 * workspace: gsd_bf16_wgrad_workspace(ntaps, N, H, W, a->C, b->C) floats (split-K slabs, summed in a fixed order). */
int64_t gsd_bf16_wgrad_workspace(int ntaps, int N, int H, int W, int M, int Ncols);
int gsd_bf16_wgrad(const gsd_nhwc* a, const gsd_nhwc* b, int ntaps, int stride, const int* ty, const int* tx, float* dw,
                   int ncols_out, float* workspace, int64_t workspace_elems, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSD_BF16_H */
