/*
 * gsd.h -- C ABI of libgsd.so: MI355X (gfx950) kernels for the gelslim_depth U-Net train step.
 *
 * The reference (MMintLab/gelslim_depth) is pure Python and has NO FFI of its own: every device
 * operation of its hot path is a torch (ATen) operator call.  Each entry point below therefore
 * cites the reference call site whose ATen operator(s) it replaces (paths relative to
 * /root/reference/).  A maintainer binds this library with ctypes; the
 * shipped binding is gelslim_depth_amd/_lib.py.
 *
 * Conventions
 *   - all tensors fp32, NCHW, allocated by the caller (PyTorch caching allocator); the library
 *     never allocates, frees or retains device memory;
 *   - `stream` is a hipStream_t passed as void*; kernels are launched on it and nothing
 *     synchronises (safe for hipGraph capture and for the autograd thread);
 *   - return 0 on success, a negative gsd_status otherwise; gsd_last_error() returns a
 *     thread-local message; no exceptions, no exit();
 *   - every call is re-entrant and no result depends on library state: tuning knobs (GSD_* environment
 *     variables) are read on every call; the only process-wide state is an atomic per-device cache of an
 *     idempotent launch attribute (hipFuncSetAttribute(MaxDynamicSharedMemorySize), gsd_common.h).
 *
 * "Deferred BatchNorm": a convolution never materialises relu(bn(raw)).  It writes the raw
 * convolution output plus per-block partial sums; gsd_bn_finalize turns the partials into a
 * per-channel (scale, shift); every consumer applies  max(0, raw*scale+shift)  while it loads
 * the tensor (gsd_src.scale/shift/relu).  F.pad + torch.cat (unet.py:46-48) are likewise never
 * materialised: a consumer takes up to two channel segments, each with its own extent/offset,
 * and reads zeros outside a segment's extent.
 */
#ifndef GSD_H
#define GSD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  GSD_OK = 0,
  GSD_ERR_BAD_ARG = -1,       /* null pointer, non-positive size, inconsistent shapes          */
  GSD_ERR_UNSUPPORTED = -2,   /* shape outside what the kernels tile (message says which)      */
  GSD_ERR_HIP = -3,           /* a HIP runtime call failed (message carries hipGetErrorString) */
  GSD_ERR_WORKSPACE = -4      /* caller-provided workspace too small                           */
} gsd_status;

/* Non-finite guard of a train step -- the device-side form of the reference's `if pred_loss.isnan()`
 * (train_utils/train_unet.py:371-372; there it costs a host sync and makes backward() raise).
 * words: two int32 on the device, zero-initialised by the caller.  A kernel that sees a non-finite BatchNorm
 * batch statistic (gsd_bn_finalize) or loss (gsd_loss_fwd_bwd) stores `tick` (the caller's step number,
 * never 0) into words[0]; gsd_adam_ema called with the same tick then leaves parameters, moments and EMA
 * untouched and adds 1 to words[1].  No host synchronisation; the caller reads words[1] when it likes.
 * Independently of the guard, non-finite batch statistics never reach running_mean/running_var.
 * Running statistics of a SKIPPED step: layers in front of the first bad one (and, racily, its finite channels) have
 * already been updated when the step is found bad.  A caller that wants a skipped step to leave no trace -- and every
 * data-parallel rank with the same buffers -- brackets the step with gsd_guard_snapshot / gsd_guard_restore over its
 * BatchNorm buffer arena: the restore puts the snapshot back iff words[0] == tick. */
typedef struct {
  int32_t* words;
  int32_t tick;
} gsd_guard;
int gsd_guard_snapshot(const float* live, float* snapshot, int64_t n, void* stream);
int gsd_guard_restore(const gsd_guard* guard, float* live, const float* snapshot, int64_t n, void* stream);

/* One channel segment of an input operand as a consumer sees it. */
typedef struct {
  const float* ptr;     /* element (n=0, c=0, h=0, w=0) of the stored tensor                    */
  const float* scale;   /* per-channel affine applied on load, or NULL for identity             */
  const float* shift;
  int32_t C;            /* channels in this segment                                             */
  int32_t H, W;         /* stored extent                                                        */
  int32_t off_h, off_w; /* where stored (0,0) sits in the consumer's grid (F.pad top/left)      */
  int32_t relu;         /* max(0, .) after the affine                                           */
  int32_t w_stride;     /* elements between rows (row pitch, >= W).  Kernels that take a pitched
                           operand say so; the others require w_stride == W.  With a pitch that is
                           a multiple of 4 floats (and 16-byte aligned strides) the Winograd kernels
                           move the operand as aligned 16-byte LDS-DMA pieces; columns W .. pitch-1
                           must then hold the operand's padding value (0 for a plain tensor).      */
  int32_t slack;        /* the caller vouches for this many READABLE floats before the first and after
                           the last element of the stored tensor (0: none).  With slack >= 4 the Winograd
                           dW kernel moves activation windows as 16-byte pieces straight from unaligned
                           rows -- a piece that straddles an image edge reads up to 3 floats of the
                           neighbouring row, at the tensor's two ends of the slack; the two-dimensional
                           dW form also parks masked lanes on the 4 floats in front of a plane -- a quarter
                           of the fill instructions.  (Keeps the 64-bit members aligned.)           */
  int64_t n_stride;     /* elements between images                                              */
  int64_t c_stride;     /* elements between channels                                            */
} gsd_src;

/* One channel segment of an output operand. Positions outside (H, W) after subtracting the
 * offset are dropped (this is the backward of F.pad: a crop). */
typedef struct {
  float* ptr;
  int32_t C;
  int32_t H, W;
  int32_t off_h, off_w;
  int32_t w_stride;     /* elements between rows (row pitch, >= W); see gsd_src                  */
  int64_t n_stride;
  int64_t c_stride;
} gsd_dst;

/* ---- library ------------------------------------------------------------------------------ */
const char* gsd_version(void);
const char* gsd_last_error(void);
/* MFMA fragment-layout self test (v_mfma_f32_16x16x4_f32 A/B/D lane maps used by every conv
 * kernel); writes 16x16 D = A*B for asymmetric integer A (16x4), B (4x16) to `out`. */
int gsd_selftest_mfma(const float* a, const float* b, float* out, void* stream);

/* ---- weight re-layouts (cheap, once per optimiser step) ----------------------------------- */
/* mode 0: conv3x3 forward   W[Co][Ci][3][3]  -> k = ci*9+t,           m = co
 * mode 1: conv3x3 dgrad     W[Co][Ci][3][3]  -> k = co*9+(8-t) flipped, m = ci
 * mode 2: convT   forward   W[Ci][Co][2][2]  -> k = ci,               m = co*4+kh*2+kw
 * mode 3: convT   dgrad     W[Ci][Co][2][2]  -> k = co*4+kh*2+kw,     m = ci
 * mode 4: conv3x3 forward, Winograd F(4,3) rows: k = ci*18+r*6+f, m = co   (see gsd_conv3x3_w43)
 * mode 5: conv3x3 dgrad,   Winograd F(4,3) rows: k = co*18+r*6+f (flipped kernel), m = ci
 * mode 6: convT   forward, LDS-DMA kernel: k = ci (rows padded to 32), m = co*4+kh*2+kw in 128-column blocks
 * mode 7: convT   dgrad,   LDS-DMA kernel: k = co*4+kh*2+kw (rows padded to 32), m = ci in 128-column blocks
 * mode 8: conv3x3 forward, Winograd F(2x4,3x3): k = ci*24+fr*6+fc, m = co   (see gsd_conv3x3_w2d)
 * mode 9: conv3x3 dgrad,   Winograd F(2x4,3x3): k = co*24+fr*6+fc (flipped kernel), m = ci
 * Modes 0/1 are tiled for the LDS-DMA kernel: [m-block][k row][BM] with BM = 64 (M <= 64) or 128, columns
 * permuted inside each 64-group (slot l*4+t = column t*16+l), so one K-chunk of one m-block is a contiguous LDS
 * image whose A operands are aligned float4s; modes 2/3 are [k row][M rounded up to 64].
 * k rows are zero-padded to whole K-chunks. gsd_weight_layout_size returns the element count; the buffer
 * must be 16-byte aligned. */
int64_t gsd_weight_layout_size(int mode, int Co, int Ci);
int gsd_weight_layout(int mode, const float* w, int Co, int Ci, float* wt, void* stream);
/* Several layouts in as few launches as possible: the 2-D Winograd images (modes 8 / 9) of all jobs in ONE launch (up to 40 per
 * launch), every other mode as gsd_weight_layout would.  `jobs` is read on the host during the call. */
typedef struct {
  const float* w;   /* (Co, Ci, 3, 3) weights */
  float* wt;        /* gsd_weight_layout_size(mode, Co, Ci) floats */
  int32_t mode, Co, Ci, reserved;
} gsd_wl_job;
int gsd_weight_layout_batch(const gsd_wl_job* jobs, int n, void* stream);

/* ---- convolution family (implicit GEMM on v_mfma_f32_16x16x4_f32) -------------------------- */
/* conv3x3, stride 1, pad 1, no bias.  Replaces aten::convolution at unet.py:11,14 (forward,
 * weights from mode 0) and the dX half of aten::convolution_backward (weights from mode 1,
 * src = gradient w.r.t. the raw output).  `partials` (optional) receives per-block
 * (sum, sum of squares) of the raw output per output channel for BatchNorm:
 * layout [n_partial_rows][2*Mpad]; gsd_conv3x3_partial_rows gives n_partial_rows. */
int gsd_conv3x3_partial_rows(int N, int H, int W, int Cout);
int gsd_conv3x3(const gsd_src* src, int nsrc, const float* wt, int Cin, int Cout,
                const gsd_dst* dst, int ndst, float* partials, int N, int H, int W, void* stream);

/* conv3x3 dX fused with the backward of the relu(bn(raw)) that produced this convolution's INPUT: instead of
 * da it writes dz = da * [raw*scale+shift > 0] to dst (one full (Cout,H,W) buffer with raw's strides) and the
 * partials hold (sum dz, sum dz*xhat) per channel, layout as gsd_conv3x3 -- saves the separate
 * gsd_bn_bwd_reduce pass (one read and one write of the tensor). aten: convolution_backward(dX) +
 * threshold_backward + the reduction half of native_batch_norm_backward (unet.py:11-16). */
int gsd_conv3x3_dgrad_bnrelu(const gsd_src* src, const float* wt, int Cin, int Cout, const gsd_dst* dst,
                             const float* raw, const float* scale, const float* shift, const float* mean,
                             const float* invstd, float* partials, int N, int H, int W, void* stream);

/* The same two operators with the Winograd F(4,3) minimal-filtering identity along image rows (half the MFMA work,
 * same fp32 storage and accumulation; weights from gsd_weight_layout modes 4 (forward) / 5 (dX):
 * [m-block of 64][k row = ci*18 + r*6 + f][64], U = G g per kernel row r).  gsd_conv3x3_algo says which form the
 * library prefers for a shape (0 direct, 1 Winograd: tile padding can eat the gain on small images); the caller
 * lays the weights out for the form it calls.  Partial-row layout as gsd_conv3x3, its own row count. */
int gsd_conv3x3_algo(int N, int H, int W, int Cin, int Cout);
int gsd_conv3x3_w43_partial_rows(int N, int H, int W, int Cout);
int64_t gsd_conv3x3_w43_mfma_count(int N, int H, int W, int Cin, int Cout);
int gsd_conv3x3_w43(const gsd_src* src, int nsrc, const float* wt, int Cin, int Cout,
                    const gsd_dst* dst, int ndst, float* partials, int N, int H, int W, void* stream);
int gsd_conv3x3_w43_dgrad_bnrelu(const gsd_src* src, const float* wt, int Cin, int Cout, const gsd_dst* dst,
                                 const float* raw, const float* scale, const float* shift, const float* mean,
                                 const float* invstd, float* partials, int N, int H, int W, void* stream);
/* "K slab" form of the two Winograd entry points for launches whose tile grid leaves most of the chip idle (the 40x53 and
 * 20x26 levels at small per-GPU batches: 150-600 blocks for 512 block slots): the input channels are cut into S slabs, block
 * (tile, slab) writes its un-reduced output tile to `ws`, and a second launch adds the slabs in slab order (run-to-run
 * bitwise) and runs the usual epilogue (crop, statistics, fused BatchNorm-backward).  gsd_conv3x3_w43_workspace gives the
 * floats of scratch the library would like for a shape (0: it runs the shape unsplit); a smaller or null `ws` reduces S, down
 * to the plain launch.  The rounding differs from the unsplit launch (two partial sums instead of one), so a caller that
 * needs batch-size-independent bits (eval-mode inference) passes ws = NULL.  Same operators as above (unet.py:11,14). */
int64_t gsd_conv3x3_w43_workspace(int N, int H, int W, int Cin, int Cout);
int gsd_conv3x3_w43_ws(const gsd_src* src, int nsrc, const float* wt, int Cin, int Cout, const gsd_dst* dst, int ndst,
                       float* partials, float* ws, int64_t ws_elems, int N, int H, int W, void* stream);
int gsd_conv3x3_w43_dgrad_bnrelu_ws(const gsd_src* src, const float* wt, int Cin, int Cout, const gsd_dst* dst,
                                    const float* raw, const float* scale, const float* shift, const float* mean,
                                    const float* invstd, float* partials, float* ws, int64_t ws_elems, int N, int H, int W,
                                    void* stream);

/* The same two operators with the TWO-dimensional Winograd identity F(2x4, 3x3): F(4,3) along the rows combined with F(2,3)
 * down the columns -- 24 products per 2x4 outputs and input channel, a third of the direct form's and two thirds of the
 * row-only form's MFMA work; same fp32 storage and accumulation.  Weights from gsd_weight_layout modes 8 (forward) / 9 (dX):
 * [m-block of 64][k row = ci*24 + fr*6 + fc][64], U = G2 g G4^T.  Needs Cin % 4 == 0 and a first source segment of a multiple
 * of 4 channels (gsd_conv3x3_w2d_supported); no row folding (an eval-mode forward in this form gives image i of a batch the bits
 * the image alone gets).  Partial-row layout as gsd_conv3x3, its own row count (two rows per pixel tile).
 * K slabs (the _ws entries, as gsd_conv3x3_w43_ws): with scratch lent by the caller a launch whose tile grid leaves most of the
 * chip idle is cut along the input channels, the slabs summed in a fixed order by a second kernel that also runs the epilogue
 * (run-to-run bitwise; not bit-equal to the unsplit launch).  gsd_conv3x3_w2d_workspace: the floats such a launch wants (0: it runs
 * unsplit); any capacity is safe -- the launcher shrinks the slab count to what fits. */
int gsd_conv3x3_w2d_supported(int Cin, int C0);
int gsd_conv3x3_prefers_w2d(int N, int H, int W, int Cin, int Cout, int train);   /* 1: run this Winograd launch in the 2-D form */
double gsd_conv3x3_w2d_estimate_us(int N, int H, int W, int Cin, int Cout);          /* the planner's run-time models */
double gsd_conv3x3_w43_estimate_us(int N, int H, int W, int Cin, int Cout, int slabs);
int gsd_conv3x3_w2d_partial_rows(int N, int H, int W, int Cout);
int64_t gsd_conv3x3_w2d_mfma_count(int N, int H, int W, int Cin, int Cout);
int gsd_conv3x3_w2d(const gsd_src* src, int nsrc, const float* wt, int Cin, int Cout,
                    const gsd_dst* dst, int ndst, float* partials, int N, int H, int W, void* stream);
int gsd_conv3x3_w2d_dgrad_bnrelu(const gsd_src* src, const float* wt, int Cin, int Cout, const gsd_dst* dst,
                                 const float* raw, const float* scale, const float* shift, const float* mean,
                                 const float* invstd, float* partials, int N, int H, int W, void* stream);
double gsd_conv3x3_w2d_estimate_slabs_us(int N, int H, int W, int Cin, int Cout);    /* ... with the K-slab form where it pays */
int64_t gsd_conv3x3_w2d_workspace(int N, int H, int W, int Cin, int Cout);
int gsd_conv3x3_w2d_ws(const gsd_src* src, int nsrc, const float* wt, int Cin, int Cout, const gsd_dst* dst, int ndst,
                       float* partials, float* workspace, int64_t workspace_elems, int N, int H, int W, void* stream);
int gsd_conv3x3_w2d_dgrad_bnrelu_ws(const gsd_src* src, const float* wt, int Cin, int Cout, const gsd_dst* dst,
                                    const float* raw, const float* scale, const float* shift, const float* mean,
                                    const float* invstd, float* partials, float* workspace, int64_t workspace_elems,
                                    int N, int H, int W, void* stream);

/* ConvTranspose2d(k=2,s=2)+bias. Replaces aten::convolution(transposed) at unet.py:36,41.
 * src is the (h,w) input (deferred BN allowed), dst the (2h,2w) output. weights: mode 6. */
int gsd_convT2x2(const gsd_src* src, const float* wt, const float* bias, int Cin, int Cout,
                 const gsd_dst* dst, int N, int H, int W, void* stream);
/* dX of the above: src = gradient w.r.t. the (2h,2w) output (plain), dst (h,w).  weights: the gsd_weight_layout mode that
 * gsd_convT2x2_dgrad_layout returns for the same arguments -- 7 (LDS-DMA kernel; an odd-width plane needs src->slack >= 2:
 * its last pixel pair reads two floats past the end of a row) or 3 (register-staged kernel). */
int gsd_convT2x2_dgrad_layout(const gsd_src* src, int Cin, int Cout, int N, int H, int W);
int gsd_convT2x2_dgrad(const gsd_src* src, const float* wt, int Cin, int Cout,
                       const gsd_dst* dst, int N, int H, int W, void* stream);
/* The same with the layout mode of `wt` (3 or 7) STATED by the caller instead of re-derived at launch: a caller that lays its
 * weights out once per buffer shape cannot be handed the other kernel by GSD_CONVT_DG_DMA or a different `slack` at launch
 * time; a mode the arguments do not admit is refused with GSD_ERR_BAD_ARG. */
int gsd_convT2x2_dgrad_as(int wt_mode, const gsd_src* src, const float* wt, int Cin, int Cout,
                          const gsd_dst* dst, int N, int H, int W, void* stream);
/* dX of the transposed convolution fused with the backward of the relu(bn(raw)) that produced its INPUT (unet.py:41 reads the
 * output of the DoubleConv below, unet.py:12-16): dst receives dz = dx * [raw*scale+shift > 0] and `partials` the per-channel
 * (sum dz, sum dz*xhat) as rows of 2*round_up(Cin,64) floats -- the layout gsd_conv3x3_dgrad_bnrelu leaves, for
 * gsd_bn_reduce_partials / gsd_bn_bwd_reduce_finalize -- so gsd_bn_bwd_reduce(mode 0) over that unit disappears.  `raw` has dst's
 * strides; wt: layout mode 7 (the LDS-DMA kernel only).  gsd_convT2x2_dgrad_bnrelu_partial_rows: the row count, 0 when the
 * arguments do not admit the kernel (odd W without src->slack >= 2); it does not read the environment. */
int gsd_convT2x2_dgrad_bnrelu_partial_rows(const gsd_src* src, int Cin, int Cout, int N, int H, int W);
int gsd_convT2x2_dgrad_bnrelu(const gsd_src* src, const float* wt, int Cin, int Cout, const gsd_dst* dst, const float* raw,
                              const float* scale, const float* shift, const float* mean, const float* invstd,
                              float* partials, int N, int H, int W, void* stream);

/* dW of conv3x3: dW[co][ci][kh][kw] = sum_{n,h,w} dy[n,co,h,w] * a[n,ci,h+kh-1,w+kw-1].
 * `a` is given as up to two segments with deferred BN (recomputed on load), `dy` plain.
 * Deterministic split-K: partial slabs in `workspace` (gsd_conv3x3_wgrad_workspace elements),
 * then an ordered reduction writes dW in the reference's (Co,Ci,3,3) layout.
 * The library picks the arithmetic form per shape (direct taps, or the transposed Winograd F(4,3) identity
 * dg = G^T[(A dy).(B^T d)] along rows for Cin, Cout >= 16: half the MFMA work, fp32 throughout; GSD_WGRAD_ALGO=0|1
 * forces one); the workspace query covers either. */
int64_t gsd_conv3x3_wgrad_workspace(int N, int H, int W, int Cin, int Cout);
int gsd_conv3x3_wgrad(const gsd_src* a, int nsrc, const gsd_src* dy, int Cin, int Cout,
                      float* dw, float* workspace, int64_t workspace_elems,
                      int N, int H, int W, void* stream);
/* Which arithmetic form gsd_conv3x3_wgrad takes for exactly these operands: 0 direct taps, 1 Winograd F(4,3) along rows,
 * 2 the two-dimensional F(2x4,3x3) form dg = G2^T[(A2 dY A4^T).(B2^T d B4)]G4 (a third of the direct form's MFMA work; it needs
 * Cout % 128 == 0 or Cout == 64, Cin a multiple of its 32/64-channel block, a dy with 16-byte aligned rows (w_stride % 4 == 0,
 * pad columns 0) and slack >= 4 around every activation segment; GSD_WGRAD_W2D=0 switches it off).  Reads no device memory.
 * gsd_conv3x3_wgrad_mfma_count: v_mfma_f32_16x16x4_f32 instructions that form executes (tile padding included). */
int gsd_conv3x3_wgrad_form(const gsd_src* a, int nsrc, const gsd_src* dy, int Cin, int Cout, int N, int H, int W);
int64_t gsd_conv3x3_wgrad_mfma_count(int form, int N, int H, int W, int Cin, int Cout);
/* 1 when gsd_conv3x3_wgrad serves this shape with a kernel that takes a pitched dy (dy->w_stride > W, see gsd_src). */
int gsd_conv3x3_wgrad_takes_pitched_dy(int N, int H, int W, int Cin, int Cout);
/* dW of a conv3x3 with FEW input channels (Cin * 9 <= 32: the network's first layer, unet.py:15) with the BatchNorm
 * backward of its output applied on the fly: d_raw = scale * (dz - c1 - (raw - mean) * invstd * c2) is formed in registers
 * from dz and raw -- what gsd_bn_bwd_apply would have written.  The first layer has no dX, so dW is d_raw's only reader and
 * the apply pass disappears.  With scale == raw == NULL dz is taken as the conv output's gradient as it is.
 * `a`: ONE plain segment (no deferred BN: the input image); dz, raw: contiguous (N,Cout,H,W).  Deterministic (slabs in
 * `workspace`, ordered reduction); dW in the reference's (Co,Ci,3,3) layout.
 * gsd_conv3x3_wgrad_bn_supported: 1 when the shape is served (0: use gsd_bn_bwd_apply + gsd_conv3x3_wgrad). */
int gsd_conv3x3_wgrad_bn_supported(int N, int H, int W, int Cin, int Cout);
int64_t gsd_conv3x3_wgrad_bn_workspace(int N, int H, int W, int Cin, int Cout);
int gsd_conv3x3_wgrad_bn(const gsd_src* a, const float* dz, const float* raw, const float* scale,
                         const float* mean, const float* invstd, const float* c1, const float* c2,
                         int Cin, int Cout, float* dw, float* workspace, int64_t workspace_elems,
                         int N, int H, int W, void* stream);
/* dW and db of ConvTranspose2d(k2,s2): x (h,w) with deferred BN, dy (2h,2w) plain;
 * dW in the reference's (Ci,Co,2,2) layout. */
int64_t gsd_convT2x2_wgrad_workspace(int N, int H, int W, int Cin, int Cout);
int gsd_convT2x2_wgrad(const gsd_src* x, const gsd_src* dy, int Cin, int Cout,
                       float* dw, float* dbias, float* workspace, int64_t workspace_elems,
                       int N, int H, int W, void* stream);

/* ---- BatchNorm2d (unet.py:12,15; aten::native_batch_norm / _backward) ---------------------- */
/* Reduce conv partials -> sums[0..2*C) (fp64: sum, sum of squares), two ordered stages. Separate from
 * finalize so a data-parallel run can all-reduce sums[0..2C) (SyncBN) in between.
 * `sums` must hold 65*2*C doubles: the result followed by 64*2*C doubles of stage-1 scratch. */
int gsd_bn_reduce_partials(const float* partials, int rows, int Mpad, int C, double* sums, void* stream);
/* Per-channel sums (fp32) of what a conv launch STORED, channels [c_begin, c_begin+c_count): the first halves of its partial
 * rows summed in fp64 in the same two ordered stages.  The ConvT bias gradient (aten::convolution_backward of unet.py:36,41)
 * comes out of the statistics epilogue of the dX launch that writes the up-sampled tensor's gradient this way, without a
 * second pass over it.  `sums`: 65*2*C doubles of scratch (sums[0..C) receives all C channel sums). */
int gsd_partials_channel_sums(const float* partials, int rows, int Mpad, int C, int c_begin, int c_count, float* out,
                              double* sums, void* stream);
/* *counters[i] += delta for n int64 device counters (host array of device pointers): BatchNorm2d.num_batches_tracked of
 * every layer of a train-mode forward (aten::native_batch_norm's `num_batches_tracked += 1`, unet.py:12,15), one launch. */
int gsd_add_counters(int64_t* const* counters, int n, int64_t delta, void* stream);
/* sums -> mean, invstd, (scale, shift) = (gamma*invstd, beta - mean*scale); updates running stats
 * (momentum 0.1, unbiased variance) when running_mean != NULL and the batch statistics are finite.
 * count = N*H*W (global if synced). guard may be NULL. */
int gsd_bn_finalize(const double* sums, int C, double count, const float* gamma, const float* beta,
                    float eps, float momentum, float* running_mean, float* running_var,
                    float* mean, float* invstd, float* scale, float* shift, const gsd_guard* guard, void* stream);
/* eval mode: (scale, shift) from running stats. */
int gsd_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, int C, float* scale, float* shift, void* stream);

/* Backward of relu(bn(raw)), pass 1.  dz = da * [raw*scale+shift > 0]; writes dz and per-block
 * partials of (sum dz, sum dz*xhat) [+ sum dout*a for mode OUTC].
 *   mode 0 PLAIN: da given by `da` (gsd_src, plain).
 *   mode 1 POOL : da = da (may be NULL ptr => 0) + max-pool routed `dpool` (N,C,H/2,W/2):
 *                 the 2x2 arg-max is recomputed from raw (first maximum wins, like aten).
 *   mode 2 OUTC : da[c] = sum_k dout[k]*wout[k][c]   (1x1 output conv unet.py:54, n_classes K <= 8), extra
 *                 partial dWout[0][c] = sum dout[0]*a[c]  (a = relu(bn(raw))): all of dWout for K == 1; for K > 1 see
 *                 gsd_conv1x1_out_wgrad.
 * partial layout: [rows][3*C] (third block only meaningful in mode 2), rows from
 * gsd_bn_bwd_partial_rows. */
int gsd_bn_bwd_partial_rows(int N, int C, int H, int W);
int gsd_bn_bwd_reduce(int mode, const float* raw, const float* scale, const float* shift,
                      const float* mean, const float* invstd,
                      const gsd_src* da, const float* dpool, const float* dout, const float* wout,
                      int K, float* dz, float* partials, int N, int C, int H, int W, void* stream);
/* pass 2: reduce partials -> sums[0..3*C) (fp64; exposed for the SyncBN all-reduce), then
 * dgamma, dbeta (+ dWout), and the coefficient vectors c1 = sum_dz/n, c2 = sum_dz_xhat/n.
 * `sums` must hold 65*3*C doubles (result + stage-1 scratch). dwout may be NULL.
 * finalize: dgamma/dbeta/dwout come from sums_local (this rank's batch: data-parallel averaging of
 * parameter gradients happens later), c1/c2 from sums_global/count (sums_global == NULL: local). */
int gsd_bn_bwd_reduce_partials(const float* partials, int rows, int C, double* sums, void* stream);
int gsd_bn_bwd_finalize(const double* sums_local, const double* sums_global, int C, double count,
                        float* dgamma, float* dbeta, float* dwout, float* c1, float* c2, void* stream);

/* One-launch forms of (gsd_bn_reduce_partials + gsd_bn_finalize) and (gsd_bn_[bwd_]reduce_partials +
 * gsd_bn_bwd_finalize) for launches that leave only a few hundred partial rows (the persistent bf16 kernels; no SyncBN
 * exchange in between).  `sums` (3*C doubles) still receives the totals.  layout_mpad: 0 for rows of 3*C floats from
 * the stand-alone backward reduce kernels, mpad for rows of 2*mpad floats from a dX epilogue. */
int gsd_bn_reduce_finalize(const float* partials, int rows, int Mpad, int C, double* sums, double count, const float* gamma,
                           const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                           float* mean, float* invstd, float* scale, float* shift, const gsd_guard* guard, void* stream);
int gsd_bn_bwd_reduce_finalize(const float* partials, int rows, int layout_mpad, int C, double* sums, double count,
                               float* dgamma, float* dbeta, float* dwout, float* c1, float* c2, void* stream);
/* pass 3: d_raw = scale_g * (dz - c1 - xhat*c2); scale_g = gamma*invstd = scale.
 * out == NULL: in place on dz.  Otherwise the result goes to `out`, an (N,C,H,out_w_stride) buffer whose rows are
 * PITCHED (out_w_stride >= W, a multiple of 4 floats, base 16-byte aligned; columns W.. are written 0) and dz is left
 * as it is: the dW / dX kernels that read d_raw next move a pitched operand as aligned 16-byte pieces. */
int gsd_bn_bwd_apply(float* dz, const float* raw, const float* scale, const float* mean,
                     const float* invstd, const float* c1, const float* c2,
                     int N, int C, int H, int W, float* out, int out_w_stride, void* stream);
/* out[k] = sum over n, p of x[n][k][p] (contiguous [N][K][HW]); deterministic two-stage.
 * workspace >= 64*K floats. Used for dbias of the output conv / transposed conv. */
int gsd_sum_planes(const float* x, int N, int K, int64_t HW, float* out, float* workspace, void* stream);

/* ---- MaxPool2d(2) floor mode (unet.py:26; aten::max_pool2d_with_indices) -------------------- */
/* y = maxpool(max(0, raw*scale+shift)) materialised at (H/2, W/2). No index tensor is kept. */
int gsd_maxpool2(const gsd_src* src, float* y, int N, int C, int H, int W, void* stream);

/* ---- output conv + loss (unet.py:54; train_unet.py:51-52) ----------------------------------- */
/* out[n,k,p] = b[k] + sum_c w[k][c] * max(0, raw[c]*scale[c]+shift[c]). */
int gsd_conv1x1_out(const gsd_src* src, const float* w, const float* b, int C, int K,
                    float* out, int N, int H, int W, void* stream);
/* dW of the output conv for n_classes K > 1 (1 <= K <= 8; for K == 1 it is the third sum of gsd_bn_bwd_reduce mode 2):
 * dw[k][c] = sum_{n,p} dout[n,k,p] * max(0, raw[n,c,p]*scale[c]+shift[c])  (aten::convolution_backward of unet.py:54).
 * raw: dense (N,C,H,W); partials: gsd_conv1x1_out_wgrad_rows(N,H,W) * K*C floats; sums: 65*K*C doubles of scratch. */
int gsd_conv1x1_out_wgrad_rows(int N, int H, int W);
int gsd_conv1x1_out_wgrad(const float* raw, const float* scale, const float* shift, const float* dout, int C, int K,
                          float* dw, float* partials, double* sums, int N, int H, int W, void* stream);
/* loss = mean((o-t)^2) (kind 0) or mean(|o-t|) (kind 1); grad = d loss/d o * grad_scale.
 * loss_out: 1 float (device). workspace >= 2048 floats. grad and guard may be NULL. */
int gsd_loss_fwd_bwd(int kind, const float* o, const float* t, int64_t numel, float grad_scale,
                     float* loss_out, float* grad, float* workspace, const gsd_guard* guard, void* stream);

/* ---- optimiser (train_unet.py:306,375-376) -------------------------------------------------- */
/* Fused torch.optim.Adam(lr, betas, eps, weight_decay: coupled L2) + torch_ema update over a flat
 * parameter arena. step is the 1-based Adam step; ema may be NULL; ema_decay already resolved
 * (min(decay,(1+n)/(10+n))). grad_scale multiplies g first (1/world_size after all-reduce).
 * guard (may be NULL): skip the whole update when words[0] == tick (see gsd_guard). */
int gsd_adam_ema(float* p, const float* g, float* m, float* v, float* ema, int64_t numel,
                 int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                 float ema_decay, float grad_scale, const gsd_guard* guard, void* stream);

/* ---- inference pre/post-processing (test_utils/test_depth_estimation.py:14-20, complete_prediction.py:4-10) ---- */
/* F.interpolate(mode='area') (image_utils.py:12-15; == adaptive_avg_pool2d) fused with the difference image
 * (image_utils.py:6-10, when base != NULL: pre(x) = (x - base + pre_add) * pre_mul) and a per-channel affine
 * (normalize_tactile_image / denormalize_depth_image, normalization_utils.py:4-35,101-129):
 *   out[n,c,oh,ow] = A[min(c,nab-1)] * mean_window(pre(in)) + B[min(c,nab-1)]. */
int gsd_area_resize_affine(const float* in, const float* base, int N, int C, int H, int W, float* out, int OH, int OW,
                           const float* A, const float* B, int nab, float pre_add, float pre_mul, void* stream);

/* ---- device-resident dataset path (gelslim_depth/datasets/general_dataset.py) ----------------- */
/* Ingest of one object file's images into the dataset arena: finger split (the caller passes the channel view
 * through strides; general_dataset.py:69-72), difference image (image_utils.py:6-10, when base != NULL:
 * pre(x) = (x - base + pre_add) * pre_mul) and F.interpolate(mode='area') (image_utils.py:12-15) in one pass.
 * dtype: 0 = float32, 1 = uint8 source (strides in ELEMENTS). out is contiguous (N,C,OH,OW) float32. */
int gsd_ingest_images(const void* in, const void* base, int dtype, int N, int C, int H, int W, int64_t in_n_stride,
                      int64_t in_c_stride, int64_t base_n_stride, int64_t base_c_stride, float* out, int OH, int OW,
                      float pre_add, float pre_mul, void* stream);
/* torchvision.transforms.functional.gaussian_blur of `planes` contiguous HxW planes (blur_depth_images, image_utils.py:17-19;
 * general_dataset.py:74-76,84-86 when depth_image_blur_kernel > 1): reflect padding of K/2, then the depthwise correlation
 * with the K x K kernel (device pointer, row-major; the caller builds it as torchvision does -- gelslim_depth_amd/dataset.py).
 * Out of place (in != out).  torchvision is a third-party dependency the reference does not pin and this image lacks:
 * its published algorithm is restated, parity unpinned for this one function. */
int gsd_gaussian_blur(const float* in, int64_t planes, int H, int W, const float* kernel2d, int K, float* out, void* stream);
/* Per-channel {min, max, mean, unbiased std} over x (N,C,HW) -> out[4*C] doubles
 * (calculate_image_normalization_params / calculate_depth_normalization_params, general_dataset.py:199-220).
 * workspace: gsd_channel_stats_workspace(C) doubles. Deterministic (fixed reduction order). */
int64_t gsd_channel_stats_workspace(int C);
int gsd_channel_stats(const float* x, int64_t N, int C, int64_t HW, double* out, double* workspace, void* stream);
/* Batch assembly (__getitem__ + normalize_sample, general_dataset.py:222-236, behind DataLoader's shuffle):
 *   out[b,c,:] = A[min(c,nab-1)] * src[idx[b],c,:] + B[min(c,nab-1)],  src (M,C,HW), idx int64[B] on the device.
 * An index outside [0,M) fills its row with NaN (never reads out of bounds). */
int gsd_gather_affine(const float* src, const int64_t* idx, int64_t M, int B, int C, int64_t HW, const float* A,
                      const float* Bc, int nab, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSD_H */
