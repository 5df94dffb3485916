/* tramba_hip.h -- C ABI of libtramba_hip.so: the MI355X (gfx950) hot path of Tramba.
 *
 * Plain pointers and sizes only (no torch types).  Every device pointer is a HIP device
 * pointer owned by the caller; nothing is retained after the call; every launch goes onto
 * the caller's `stream` (a hipStream_t passed as void*), with no implicit synchronisation,
 * no allocation and no host<->device copy inside the call (graph-capture safe).
 * Return value: 0 on success, <0 on a rejected argument (see tramba_last_error()); the
 * library never aborts the process.
 *
 * Reference interface each entry replaces (paths relative to the mj129/Tramba tree):
 *   tramba_selective_scan_fwd   selective_scan_cuda_oflex.fwd   Models/SS2D/csms6s.py:910
 *   tramba_selective_scan_bwd   selective_scan_cuda_oflex.bwd   Models/SS2D/csms6s.py:920-922
 *   tramba_scan_table           generate_indices / generate_window_indices /
 *                               generate_dilation_indices       SpiralLine.py:27-82, Window.py:3-35,
 *                                                               Dilation.py:3-45 (+ csms6s.py:18-22)
 *   tramba_cross_scan           CrossScan*.forward (= CrossMerge*.backward)
 *                                                               csms6s.py:13-22,64-72,113-121,161-172
 *   tramba_cross_merge          CrossMerge*.forward (= CrossScan*.backward)
 *                                                               csms6s.py:34-42,83-91,132-140,188-201
 *   tramba_ss2d_scan_cl /       SS2Dv2.forward_corev2 (scan -> x_proj split -> dt_proj ->
 *   tramba_ss2d_merge_norm_cl   selective scan -> merge -> out_norm) vmamba.py:230-273,
 *                               fused, channels-last
 *   tramba_layernorm_cl         LayerNorm2d.forward             Models/modules.py:22-27
 *   tramba_shuffle_norm_cl      rearrange(...)+norm in PatchExpand / FinalPatchExpand_X4 /
 *                               FreqExpand2D                    Models/modules.py:209-218,240-249,687-696
 *   tramba_dwconv_cl            SS2D.conv2d (+SiLU)             Models/vmamba.py:283-285
 *   tramba_dw_pack              weight re-layout + DWMSMlp fold (h+dw3+dw5+dw7, vmamba.py:624)
 *   tramba_dct_split_cl         DCT2D.forward                   Models/DCT_2D.py:12-29
 *   tramba_linear_cl            Linear2d.forward (1x1 conv)     Models/modules.py:10-13
 *   tramba_conv3x3s2_cl /       patch_embed + downsample convs  Models/vmamba.py:454,481-486
 *   tramba_stem_conv_ln_gelu
 *
 * "_cl" = channels-last: activations are (B, H*W, C) row-major, C contiguous.
 */
#ifndef TRAMBA_HIP_H
#define TRAMBA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { TRAMBA_F32 = 0, TRAMBA_F16 = 1, TRAMBA_BF16 = 2 } tramba_dtype;

typedef enum {
    TRAMBA_SCAN_RASTER = 0,   /* K=4 */
    TRAMBA_SCAN_LINE = 1,     /* K=4, the Bresenham half of Helix */
    TRAMBA_SCAN_HELIX = 2,    /* K=8 = raster + line */
    TRAMBA_SCAN_WINDOW = 3,   /* K=4, param = window size (0 = default rule) */
    TRAMBA_SCAN_DILATION = 4  /* K=4, param = dilation rate (0 = 4) */
} tramba_scan_family;

typedef enum { TRAMBA_ACT_NONE = 0, TRAMBA_ACT_SILU = 1, TRAMBA_ACT_GELU = 2,
               TRAMBA_ACT_SIGMOID_GATE = 3 /* GEMM epilogue only: y = sigmoid(acc + bias) * residual */,
               TRAMBA_ACT_GELU_GRAD_MUL = 4 /* GEMM epilogue only: y = (acc + bias) * gelu'(residual) -- the input gradient
                                               of `Linear(GELU(h))` in one launch, residual = h (training path) */
             } tramba_act;

#define TRAMBA_OK 0
#define TRAMBA_ERR_ARG (-1)
#define TRAMBA_ERR_UNSUPPORTED (-2)
#define TRAMBA_ERR_HIP (-3)

/* ------------------------------------------------------------------ library */
const char *tramba_last_error(void);      /* thread-local message of the last failure */
int tramba_abi_version(void);
/* Failures a kernel can only detect on the device (today: the carry mailbox of the fused scans timing out, which leaves NaN
 * in that launch's output) raise a device error word in host-mapped memory.  No call synchronises for it: EVERY entry point
 * checks the word after its own launch and, when set, clears it and returns TRAMBA_ERR_HIP with tramba_last_error() naming
 * the cause -- i.e. the failure is reported by the next library call issued after the failing kernel has run.  This entry
 * reads (and clears) the word on demand, e.g. after the caller's own stream synchronisation: TRAMBA_OK or TRAMBA_ERR_HIP.
 * (No reference counterpart: the reference's extension reports through TORCH_CHECK on the host only, csms6s.py:857.) */
int tramba_device_error(void);
/* HIP-event timing of the kernels launched by this library (used by bench.py for the
 * roofline figure).  enable=1 brackets every launch of kernel class `which` with events on
 * the launch stream; tramba_profile_read() synchronises those events and returns the
 * number of launches, writing the summed duration in milliseconds and the summed ALGORITHMIC
 * work of those launches: bytes for the scan classes, flop (2*M*N*K) for TRAMBA_PROF_GEMM
 * (DESIGN.md states the per-kernel formula). */
int tramba_profile_enable(int which, int enable);
int tramba_profile_read(int which, double *total_ms, double *total_bytes);
/* Time only the launches of class `which` that account for at least `min_units` bytes (flops): singles out one shape
 * (e.g. the Helix 96x96 fused scan inside a model forward).  0 = every launch (default). */
int tramba_profile_min_units(int which, double min_units);
/* Kernel-variant selection for A/B timing from scripts/ and for tests that pin ONE form of a kernel against the oracle
 * (never needed for correctness: 0 = the library's own choice).  MEASUREMENT-ONLY STATE: the knobs and the profile
 * classes above are the library's only process-global mutable state (SURVEY 8(b) asks for none on the operator path):
 * nothing on the product path (tramba_amd/) ever sets a knob, every knob defaults to 0, and with all knobs at 0 a call's
 * result and kernel choice depend on its arguments alone.  Not thread-safe against concurrent launches by design: set a
 * knob, time, reset.
 * knob TRAMBA_TUNE_MERGE_FORM: 1 = one wave per pixel (deep row pipeline), 2 = streaming (several pixels per wave). */
int tramba_tune_set(int knob, int value);
int tramba_tune_get(int knob);
#define TRAMBA_TUNE_MERGE_FORM 0
#define TRAMBA_TUNE_SCAN_FORM 1      /* 1 = chained (register ring), 2 = wave-segment, 3 = chained on LDS-DMA staged operands */
#define TRAMBA_TUNE_SCAN_W 2         /* waves per sequence of the register-ring chained scan (capped by the library's own choice) */
#define TRAMBA_TUNE_GEMM_TILE 3      /* plain GEMMs with K % 64 == 0: 0 = 64x64 staged by LDS-DMA, 3 or 4 stages by shape (the default), 6 / 7 = 3 / 4 stages;
                                        register-staged
                                        forms: 1 = 64x64, 2 = 128x128, 3 = 128x64, 4 = 96x64 where it saves a round of the chip,
                                        5 = 64x64 on a 4-stage ring; 13 / 14 = the LDS-DMA kernel on 2 stages everywhere / nowhere (default:
                                        K <= 256 on grids of >= 1024 tiles); 15 = 96x64 LDS-DMA tiles (3 compute waves + a loader wave) wherever M >= 96.
                                        weight-gradient TN GEMMs (tramba_wgrad_cl): 0 = token tiles staged by LDS-DMA on 3 stages, one
                                        workgroup per CU (the default); 8 = register-staged, one tile in flight; 9 = LDS-DMA on 4 stages;
                                        10 / 11 / 12 = 384 / 512 / 768 workgroups wanted by the token split */
#define TRAMBA_TUNE_MAILBOX_SKIP 4   /* tests only: > 0 withholds the carry hand-over of that tile (chain order) in the fused scans, so that
                                        the wave waiting for it runs out of polls (~0.1 s) and the device error word is raised; 0 = off */
#define TRAMBA_TUNE_WGRAD_FORM 5     /* weight-gradient TN GEMM alone: 1 = the register-staged kernel (as TRAMBA_TUNE_GEMM_TILE 8, which also switches
                                        the projections); 2 = the LDS-DMA kernel with r03's counted lgkmcnt waits on its transposed reads
                                        (NOT safe beside other kernels on the same CU: scripts/dev/debug_wgrad_concurrent.py) */
#define TRAMBA_TUNE_DW_FORM 6        /* 7x7 depth-wise stencil and its weight gradient: 1 = the r03 kernels (one output row per thread / tap row outer),
                                        0 = the kernels that march down a band of rows (default) */
#define TRAMBA_TUNE_DW_ROWS 7        /* rows per band of the marching 7x7 kernels (0 = the library's choice) */
#define TRAMBA_TUNE_COUNT 8
#define TRAMBA_PROF_SCAN_BOUNDARY 0
#define TRAMBA_PROF_SCAN_FUSED 1
#define TRAMBA_PROF_GEMM 2          /* tramba_linear_cl (1x1-conv projections) */
#define TRAMBA_PROF_MERGE 3         /* tramba_ss2d_merge_norm_cl: K*L*D ys bytes read + L*D written, per image */
#define TRAMBA_PROF_SCAN_BWD 4      /* tramba_ss2d_scan_bwd_cl: SURVEY 8(d) backward bytes of the op it replaces, 12 B per (b,k,d,l)
                                       element at 16-bit activations (u, delta, dout read; du, ddelta written), 20 B at fp32 */
#define TRAMBA_PROF_WGRAD 5         /* tramba_wgrad_cl: 2*M*N*K flop per group and batch */
#define TRAMBA_PROF_LAYERNORM 6     /* the LayerNorm family (tramba_layernorm_cl, tramba_add_layernorm_cl, tramba_shuffle_norm_cl and the
                                       tramba_layernorm_bwd_* / tramba_shuffle_norm_bwd_cl launches): bytes of every activation-sized tensor
                                       the call reads or writes once (x, dy, dx, the residual and its sum ... ), the per-channel rows not counted */
#define TRAMBA_PROF_DW 7            /* the depth-wise stencils (tramba_dwconv_cl / _dual_cl: x read, y (and y_pre) written;
                                       tramba_dwconv_wgrad_cl: x and gy read) */
#define TRAMBA_PROF_COUNT 8

/* ------------------------------------------------------------------ scan-order tables (host) */
/* Number of directions K of a family. */
int tramba_scan_family_k(int family);
/* Default window size this library uses for feature size h (reference values for
 * 12/24/48/96, csms6s.py:107-108; documented rule otherwise). */
int tramba_default_window(int h);
/* Fills out[K*h*w] (HOST memory) with the flat pixel index (row*w+col) read at each
 * sequence position of each direction.  Returns K, or <0. */
int tramba_scan_table(int family, int h, int w, int param, int32_t *out);
/* Inverse (CSR) of a table for the deterministic merge: for pixel p,
 * entries inv_idx[inv_ptr[p] .. inv_ptr[p+1]) hold k*L+l with table[k][l]==p, ascending.
 * inv_ptr has L+1 entries, inv_idx has K*L.  HOST memory. */
int tramba_scan_table_inverse(const int32_t *table, int k, int l, int32_t *inv_ptr, int32_t *inv_idx);

/* ------------------------------------------------------------------ L0: selective scan, reference layout */
/* u, delta: (B, KD, L) io_dtype;  A: (KD, N) f32;  Bm, Cm: (B, K, N, L) io_dtype;
 * D, delta_bias: (KD) f32 or NULL;  out: (B, KD, L) out_dtype (f32 = "oflex");
 * ckpt: (B, KD, nchunk, N) f32 chunk-end states for the backward, or NULL;
 * nchunk = tramba_selective_scan_nchunk(L, io_dtype). */
int tramba_selective_scan_nchunk(int l, int io_dtype);
int tramba_selective_scan_fwd(const void *u, const void *delta, const float *A, const void *Bm,
                              const void *Cm, const float *D, const float *delta_bias, void *out,
                              float *ckpt, int batch, int kd, int k, int n, int l, int io_dtype,
                              int out_dtype, int delta_softplus, void *stream);
/* du, ddelta: (B, KD, L) io_dtype; dA (KD,N), dD, ddelta_bias (KD): f32, ACCUMULATED into
 * (caller zeroes);  dB, dC: (ncopy, B, K, N, L) f32, accumulated into (caller zeroes) -- the rows of a
 * direction group spread their atomic adds over `ncopy` private copies which the caller sums.
 * Both directions: d_state N = 1 (everything Tramba builds), 2 and 4. */
int tramba_selective_scan_bwd(const void *u, const void *delta, const float *A, const void *Bm,
                              const void *Cm, const float *D, const float *delta_bias,
                              const float *dout, const float *ckpt, void *du, void *ddelta,
                              float *dA, float *dB, float *dC, float *dD, float *ddelta_bias,
                              int batch, int kd, int k, int n, int l, int io_dtype,
                              int delta_softplus, int ncopy, void *stream);

/* ------------------------------------------------------------------ L1: scan-order gather / merge, NCHW */
/* xs[b,k,c,l] = x[b,c,table[k,l]];  x: (B, C, L), xs: (B, K, C, L);  table: DEVICE int32 (K, L). */
int tramba_cross_scan(const void *x, const int32_t *table, void *xs, int batch, int c, int l, int k,
                      int dtype, void *stream);
/* y[b,c,p] = sum over entries e=k*L+l of inv[p] of ys[b,k,c,l];  fp32 accumulation, fixed order. */
int tramba_cross_merge(const void *ys, const int32_t *inv_ptr, const int32_t *inv_idx, void *y,
                       int batch, int c, int l, int k, int dtype, void *stream);

/* ------------------------------------------------------------------ fused SS2D core, channels-last */
/* x:    (B, L, D)           dtype   -- conv+SiLU output, spatial order
 * xdbl: (B, L, K*RG)        f32     -- x_proj output in SPATIAL order: per group k
 *                                      [dt_0..dt_{R-1}, 0-pad to R8, B, C, 0, 0]  (d_state N = 1 only);
 *                                      R8 = R rounded up to 8, RG = tramba_ss2d_group_stride(R) = R8 + 4
 * table (K, L) int32 device; dt_w (K, D, R) f32; dt_bias (K*D) f32; A (K*D) f32 (= -exp(A_logs));
 * Ds (K*D) f32.   ys: (B, K, L, D) ys_dtype in SEQUENCE order.                                   */
int tramba_ss2d_group_stride(int r);
/* workspace (device memory, 16-byte aligned, >= tramba_ss2d_scan_workspace() bytes) selects the
 * wave-segment form: segment reduce -> carry scan -> segment replay, every wave independent.  With
 * workspace == NULL the chained single-pass kernel (one workgroup per sequence) runs instead.
 * states (training; NULL otherwise, requires ys_dtype == dtype): tramba_ss2d_scan_bwd_workspace() bytes that receive the
 * recurrence state entering every 32-position tile, (B, K, ceil(L/32) + 8, D) f32 -- handed to tramba_ss2d_scan_bwd_cl as
 * its workspace with have_states = 1, the backward skips the sweep that would recompute them.
 * a_log (training, chained forms): `A` holds the PARAMETER A_logs (K*D) and the kernel forms A = -exp(A_logs) itself
 * (vmamba.py:246) -- no exp / negation launch per block and step. */
size_t tramba_ss2d_scan_workspace(int batch, int l, int d, int k);
int tramba_ss2d_scan_cl(const void *x, const float *xdbl, const int32_t *table, const float *dt_w,
                        const float *dt_bias, const float *A, const float *Ds, void *ys, void *workspace,
                        size_t workspace_bytes, int batch, int l, int d, int k, int r, int dtype,
                        int ys_dtype, float *states, int a_log, void *stream);
/* Training: backward of tramba_ss2d_scan_cl + the merge that follows it.  gym (B, L, D), f32 or dtype (gym_dtype), is the
 * gradient of the MERGED map (CrossMerge output, before out_norm); the kernel gathers it through `table`.  Outputs, in SEQUENCE
 * order like ys: gu (B,K,L,D) dtype = dL/d(gathered x) -- merge it with tramba_ss2d_merge_norm_cl(eps < 0) to get
 * dL/dx; graw (B,K,L,D) dtype = dL/d(x_proj ranks . dt_w) before bias and softplus; gB, gC (B,K,L) f32 ACCUMULATED
 * (zero them first), element (b,k,l) at gB[((b*K + k)*L + l) * bc_stride] -- bc_stride = 1 for packed arrays, or the row
 * stride of a (B,K,L,RG) x_dbl-gradient table whose B / C columns the two pointers address; gpar (B,3,K,D) f32 = per-channel dA, dD, d(dt_bias) planes (sum over B for the parameters).
 * workspace: tramba_ss2d_scan_bwd_workspace() bytes of device scratch -- or, with have_states = 1, the `states` buffer the
 * forward launch filled. */
size_t tramba_ss2d_scan_bwd_workspace(int batch, int l, int d, int k);
int tramba_ss2d_scan_bwd_cl(const void *x, const float *xdbl, const int32_t *table, const float *dt_w,
                            const float *dt_bias, const float *A, const float *Ds, const void *gym, void *gu,
                            void *graw, float *gB, float *gC, int bc_stride, float *gpar, void *workspace,
                            size_t workspace_bytes, int have_states, int batch, int l, int d, int k, int r, int dtype,
                            int gym_dtype, int flags, void *stream);
/* flags of tramba_ss2d_scan_bwd_cl:
 *   1  `A` holds A_logs (as a_log of the forward) and gpar's first plane is dL/dA_logs (= dL/dA * A)
 *   2  gB / gC are (B, K, ceil(D/32), L) f32 tables of per-channel-tile partial sums, every element written by exactly one
 *      wave: no zero fill and no atomics (reproducible); tramba_ss2d_bwd_prep_cl adds the partials in a fixed order.
 *      bc_stride is ignored. */
#define TRAMBA_SCAN_BWD_A_LOG 1
#define TRAMBA_SCAN_BWD_BC_PARTIALS 2
/* Training, after tramba_ss2d_scan_bwd_cl(flags & 2), r = dt_rank, R8 = 8*ceil(r/8), RG = R8 + 4:
 *   ranks (B, K, L, R8) dtype = xdbl[b, table[k][i], k*RG .. k*RG + R8): the dt-rank rows in SEQUENCE order, the operand of
 *         the dt_projs_weight gradient (tramba_wgrad_cl with groups = K);
 *   gseq  (B, K, L, RG) f32: columns R8, R8 + 1 = dL/dB, dL/dC summed over the channel tiles (fixed order), R8 + 2, R8 + 3 = 0;
 *         the rank columns 0 .. r are written afterwards by tramba_rows_gemm_cl(graw, dt_projs_weight^T, ldy = RG). */
int tramba_ss2d_bwd_prep_cl(const float *xdbl, const int32_t *table, const float *bpart, const float *cpart, void *ranks,
                            float *gseq, int batch, int l, int k, int r, int ctiles, int dtype, void *stream);
/* out (B, L, K*RG) dtype = the x_dbl-row gradients of gseq (B, K, L, RG) f32 brought back from sequence to spatial order:
 * out[b, p, k*RG + j] = sum over the entries (k, i) of pixel p in the inverse table of gseq[b, k, i, j] for j < r and
 * j in {R8, R8 + 1}, 0 elsewhere -- a gather-sum in CSR order (deterministic; the Helix lines revisit pixels), the adjoint of
 * the scan kernels' gather of x_dbl rows, cast to the dtype the x_proj gradient GEMMs read. */
int tramba_ss2d_bwd_assemble_cl(const float *gseq, const int32_t *inv_ptr, const int32_t *inv_idx, void *out, int batch,
                                int l, int k, int r, int dtype, void *stream);
/* y[b,p,:] = act(LayerNorm_D(sum_{e in inv[p]} ys[b, e/L, e%L, :]));  y: (B, L, D) dtype.
 * eps < 0: the plain sum (CrossMerge alone), ln_w / ln_b / act ignored. */
int tramba_ss2d_merge_norm_cl(const void *ys, const int32_t *inv_ptr, const int32_t *inv_idx,
                              const float *ln_w, const float *ln_b, void *y, int batch, int l, int d,
                              int k, float eps, int act, int ys_dtype, int dtype, void *stream);
/* The same merge with the epilogue of SS2D's input gradient (training; eps < 0 only): y = (sum + addend) * silu'(zpre),
 * addend (B, L, D) dtype or NULL = the x_proj branch's gradient, zpre (B, L, D) dtype = the pre-activation of the SiLU in front
 * of the core (vmamba.py:283-285). */
int tramba_ss2d_merge_grad_cl(const void *ys, const int32_t *inv_ptr, const int32_t *inv_idx, const float *ln_w,
                              const float *ln_b, void *y, const void *addend, const void *zpre, int batch, int l, int d,
                              int k, float eps, int act, int ys_dtype, int dtype, void *stream);

/* ------------------------------------------------------------------ element / stencil kernels, channels-last */
/* y = act(LayerNorm_C(x));  x, y: (rows, C).  w, b f32. */
int tramba_layernorm_cl(const void *x, const float *w, const float *b, void *y, int64_t rows, int c,
                        float eps, int act, int dtype, void *stream);
/* Training: LayerNorm backward over the last dim of (rows, C): dx in dtype; part (P, 2, C) f32 receives P partial
 * (dgamma, dbeta) rows (one per wave, or per workgroup for short rows), P = tramba_layernorm_bwd_parts(rows, c, dtype): the
 * caller sums over P (tramba_slab_sum: fixed order -> reproducible).  mean / rstd are recomputed from x. */
int64_t tramba_layernorm_bwd_parts(int64_t rows, int c, int dtype);
int tramba_layernorm_bwd_cl(const void *x, const void *dy, const float *w, void *dx, float *part, int64_t rows,
                            int c, float eps, int dtype, void *stream);
/* The same on the residual stream of a block (vmamba.py:384-396 under autograd): dx = LayerNorm-backward + gres (the gradient
 * that reaches the block input through the skip connection; NULL: none), and dxm (NULL: not wanted) = dx * mask[row /
 * rows_per_sample] -- the gradient of the PREVIOUS residual branch under stochastic depth (mask (B) f32 = keep / keep_prob
 * per sample; NULL: 1).  One pass where autograd issued the LayerNorm backward, an add and a multiply. */
int tramba_layernorm_bwd_res_cl(const void *x, const void *dy, const float *w, void *dx, float *part, const void *gres,
                                const float *mask, int64_t rows_per_sample, void *dxm, int64_t rows, int c, float eps,
                                int dtype, void *stream);
/* The general form behind the two entries above and tramba_shuffle_norm_bwd_cl (P > 1: dy is indexed through the pixel
 * shuffle of x's row). */
int tramba_layernorm_bwd_any_cl(const void *x, const void *dy, const float *w, void *dx, float *part, const void *gres,
                                const float *mask, int64_t rows_per_sample, void *dxm, int64_t rows, int c, float eps, int P,
                                int H, int W, int dtype, void *stream);
/* Training: backward of tramba_shuffle_norm_cl.  x (B, H, W, P*P*C) the un-shuffled input, dy (B, H*P, W*P, C) the gradient of
 * the shuffled, normalised map -> dx like x; part as tramba_layernorm_bwd_cl with rows = B*H*W*P*P.  The gradient is read
 * through the shuffle: no permuted copy of either map exists (modules.py:209-218, 687-696 under autograd). */
int tramba_shuffle_norm_bwd_cl(const void *x, const void *dy, const float *w, void *dx, float *part, int batch, int h,
                               int wd, int c, int p, float eps, int dtype, void *stream);
/* Residual add + stochastic depth + LayerNorm of the training path in one pass: xsum = x + y * mask[row / rows_per_sample]
 * (y NULL: no add, xsum unused), n = LayerNorm_C(xsum), n_act (NULL: not wanted) = act(n) -- the pre-activation / activation
 * pair a following `Linear(act(.))` needs under autograd.  x, y, xsum, n, n_act: (rows, C) dtype; w, b, mask f32. */
int tramba_add_layernorm_cl(const void *x, const void *y, const float *mask, int64_t rows_per_sample, const float *w,
                            const float *b, void *xsum, void *n, void *n_act, int64_t rows, int c, float eps, int act,
                            int dtype, void *stream);
/* x: (B, H, W, P*P*C) -> y: (B, H*P, W*P, C) with y[b,hP+p1,wP+p2,:] = LN(x[b,h,w,(p1*P+p2)*C : +C]). */
int tramba_shuffle_norm_cl(const void *x, const float *w, const float *b, void *y, int batch, int h,
                           int wd, int c, int p, float eps, int dtype, void *stream);
/* shuffle_norm followed by a 1x1 convolution to ONE channel (FinalPatchExpand_X4 + the full-resolution
 * seg_layers head, Trambav6.py:132-137): y (B, H*P, W*P) f32 = <LayerNorm(shuffled row) rounded to dtype, head_w>
 * + head_b.  The normalised (B, H*P, W*P, C) map is never written. */
int tramba_shuffle_norm_head_cl(const void *x, const float *w, const float *b, const float *head_w, float head_b,
                                float *y, int batch, int h, int wd, int c, int p, float eps, int dtype,
                                void *stream);
/* Training: backward of tramba_shuffle_norm_head_cl in one pass over x (the normalised (B, H*P, W*P, C) map and its gradient
 * never exist).  g (B, H*P, W*P) f32 = d loss / d logits; dx (B, H, W, P*P*C) dtype; part (P, C + 4) f32 receives
 * P = tramba_shuffle_norm_head_bwd_parts() partial rows of A_c = sum over rows of g * xhat_c (slots 0..C) and G = sum g
 * (slot C); summed over the rows (tramba_slab_sum) they give d ln_w = head_w A, d ln_b = head_w G,
 * d head_w = ln_w A + ln_b G, d head_b = G.  C % 8 == 0 (4 for f32), C <= 512 (256). */
int64_t tramba_shuffle_norm_head_bwd_parts(int batch, int h, int wd, int c, int p, int dtype);
int tramba_shuffle_norm_head_bwd_cl(const void *x, const float *g, const float *ln_w, const float *head_w, void *dx,
                                    float *part, int batch, int h, int wd, int c, int p, float eps, int dtype, void *stream);
/* y[row] = <x[row, :], w> + bias: nn.Conv2d(C, 1, 1) on a channels-last map (decoder seg_layers,
 * Trambav6.py:62,67).  x (rows, C) dtype, w (C) f32, y (rows) f32. */
int tramba_rowdot_cl(const void *x, const float *w, float bias, float *y, int64_t rows, int c, int dtype,
                     void *stream);
/* Training: backward of tramba_rowdot_cl in one pass over x: gx (rows, C) dtype = gy[row] * w; part (P, C + 4) f32 receives P
 * partial rows, P = tramba_rowdot_bwd_parts(rows, c, dtype) (0: the shape is not served -- rows wider than 64 x 16 bytes),
 * columns 0 .. C = sum gy[row] * x[row, :] (the weight gradient), column C = sum gy[row] (the bias gradient); the caller sums
 * over P (tramba_slab_sum).  gy (rows) f32. */
int64_t tramba_rowdot_bwd_parts(int64_t rows, int c, int dtype);
int tramba_rowdot_bwd_cl(const void *x, const float *gy, const float *w, void *gx, float *part, int64_t rows, int c,
                         int dtype, void *stream);
/* Depth-wise stencils take TAP-MAJOR weights wt (ks*ks, C) f32 + bias bt (C) f32 produced by
 * tramba_dw_pack from the reference layout w (C, ks, ks), bias (C) or NULL.  With w3/b3/w5/b5
 * non-NULL (ks = 7) it packs the multi-scale stencil of DWMSMlp: identity + 3x3 + 5x5 + 7x7 folded
 * into one 7x7 (and b3 + b5 + b7), so y = GELU(h + dw3(h) + dw5(h) + dw7(h)) is ONE dwconv pass. */
int tramba_dw_pack(const float *w, const float *bias, const float *w3, const float *b3, const float *w5,
                   const float *b5, float *wt, float *bt, int c, int ks, void *stream);
/* `count` such packs in ceil(count / 40) launches (every depth-wise stencil of a model after an optimizer step): item i packs
 * w[i] (+ bias[i], and w3 / b3 / w5 / b5[i] for the folded multi-scale form; NULL entries as in tramba_dw_pack) into
 * wt[i] / bt[i].  All ten arguments are HOST arrays (device pointers, channel counts, kernel sizes); the items travel to the
 * kernel by value (hipGraph-capture safe). */
int tramba_dw_pack_multi(const float *const *w, const float *const *bias, const float *const *w3, const float *const *b3,
                         const float *const *w5, const float *const *b5, float *const *wt, float *const *bt, const int *c,
                         const int *ks, int count, void *stream);
/* Training path of the dense 3x3 convolutions of the VMamba stem / downsample layers (vmamba.py:454,481,486; the reference
 * gets their autograd from cuDNN through nn.Conv2d): the data movement around the library's GEMMs.
 *   im2col: x (B,H,W,C) dtype -> cols (B*Ho*Wo, CKp) dtype, column (ky*3 + kx)*C + ci (the k-major order of
 *           tramba_conv3x3s2_cl's weight), zeros in the padding taps and in columns 9C..CKp;  Ho = (H + 2 pad - 3)/stride + 1
 *   col2im: gcols (B*Ho*Wo, CKp) -> gx (B,H,W,C) = the adjoint gather (fp32 sums, fixed order, no atomics). */
int tramba_im2col3x3_cl(const void *x, void *cols, int batch, int h, int wd, int c, int stride, int pad, int ckp,
                        int dtype, void *stream);
int tramba_col2im3x3_cl(const void *gcols, void *gx, int batch, int h, int wd, int c, int stride, int pad, int ckp,
                        int dtype, void *stream);
/* Backward of F.interpolate(mode="bilinear", align_corners=False) from (planes, h, w) to (planes, hout, wout), fp32 -- the
 * resize of the deep-supervision outputs in the loss (train.py:76-85): gin = the adjoint, gathered per input pixel. */
int tramba_upsample_bilinear_bwd(const float *gout, float *gin, int planes, int h, int w, int hout, int wout,
                                 void *stream);
/* depth-wise ks x ks, stride 1, "same" padding, y = act(conv(x) + bt). */
int tramba_dwconv_cl(const void *x, const float *wt, const float *bt, void *y, int batch, int h,
                     int wd, int c, int ks, int act, int dtype, void *stream);
/* Training forms of the same stencil: y_pre (NULL: not wanted) also receives conv(x) + bt BEFORE the activation (the backward
 * of SiLU / GELU needs it; y = act of the value as stored); flip_taps = 1 mirrors the taps through the centre -- the input
 * gradient of the stencil is tramba_dwconv_dual_cl(gy, wt, zero bias, NULL, gx, ..., ACT_NONE, 1). */
int tramba_dwconv_dual_cl(const void *x, const float *wt, const float *bt, void *y_pre, void *y, int batch, int h, int wd,
                          int c, int ks, int act, int flip_taps, int dtype, void *stream);
/* The weight gradient of a stencil in the parameters' own layout: gwt (ks*ks + 1, C) f32 = tap-major taps + bias row (the
 * summed output of tramba_dwconv_wgrad_cl) -> g7 (C, ks*ks); with g5 / g3 non-NULL (ks = 7) the folded multi-scale stencil's
 * gradient restricted to the 5x5 / 3x3 parameters' supports (C, 25) / (C, 9) (the adjoint of tramba_dw_pack's fold); gb
 * (nb, C) or NULL = nb copies of the bias row. */
int tramba_dw_unpack_grad(const float *gwt, float *g7, float *g5, float *g3, float *gb, int nb, int c, int ks,
                          void *stream);
/* `count` such unpacks in ceil(count / 64) launches (the depth-wise parameter gradients of a whole backward pass: they are
 * leaves, nothing reads them before the optimizer); HOST arrays, items by value as in tramba_dw_pack_multi. */
int tramba_dw_unpack_grad_multi(const float *const *gwt, float *const *g7, float *const *g5, float *const *g3,
                                float *const *gb, const int *nb, const int *c, const int *ks, int count, void *stream);
/* Training: gradients of the same stencil w.r.t. its tap-major weights and bias (what autograd computes for
 * nn.Conv2d(groups=C), vmamba.py:301, 595-603).  part (P, ks*ks + 1, C) f32, P = tramba_dwconv_wgrad_parts(batch, h, wd,
 * c, ks): one partial row per workgroup (image, row band, column range), planes 0..ks*ks-1 = taps, last plane = bias; the
 * caller sums over P (tramba_slab_sum: fixed order).  The input gradient is tramba_dwconv_cl(gy, flipped taps, zero bias). */
int64_t tramba_dwconv_wgrad_parts(int batch, int h, int wd, int c, int ks);
int tramba_dwconv_wgrad_cl(const void *x, const void *gy, float *part, int batch, int h, int wd, int c, int ks,
                           int dtype, void *stream);
/* x: (B, n, n, C).  Y = Wy X Wx^T per channel; low = Y[:n/2,:n/2], high = Y[n/2:,n/2:],
 * both (B, n/2, n/2, C).  wx, wy: (n, n) f32.  tmp: (B, n, n, C) f32 workspace. */
int tramba_dct_split_cl(const void *x, const float *wx, const float *wy, float *tmp, void *high,
                        void *low, int batch, int n, int c, int dtype, void *stream);
/* y[m, :] = epilogue(x[m, :] @ W^T + bias);  x: (M, K) dtype, W: (N, K) dtype, bias (N) f32 or
 * NULL, y: (M, N) out_dtype.  act applied after bias.  residual (M, N) dtype or NULL is added
 * last.  MFMA path for f16/bf16. */
int tramba_linear_cl(const void *x, const void *w, const float *bias, const void *residual, void *y,
                     int64_t m, int n, int k, int act, int dtype, int out_dtype, void *stream);
/* LayerNorm2d followed by Linear2d as ONE launch (VSSBlock: norm -> op.in_proj, norm2 -> mlp.fc1, vmamba.py:384-396;
 * the decoder blocks likewise): y = epilogue(LayerNorm_K(x) @ W^T + b) computed as rstd_m * (x @ W'^T - mean_m * colsum) + t
 * with W' = W * gamma (rows of W scaled, in dtype), colsum (N) f32 = row sums of W' as rounded, bias (N) f32 = W beta + b.
 * The block derives mean / rstd of its own rows; the normalised map is never written.  16-bit dtypes, K % 64 == 0,
 * K <= 2048, N % 8 == 0; act / residual as tramba_linear_cl. */
int tramba_linear_ln_cl(const void *x, const void *w_folded, const float *colsum, const float *bias, const void *residual,
                        void *y, int64_t m, int n, int k, float eps, int act, int dtype, int out_dtype, void *stream);
/* Same with the K dimension of x split over two tensors, x = [x1 (M, k1) | x2 (M, k - k1)]: Linear2d applied to
 * torch.cat((x1, x2), dim=-1) without materialising the concatenation (decoder concat_back_dim, Trambav6.py:124;
 * FreqSS2Dv6, freq_mamba.py:52).  16-bit dtypes, k1 % 64 == 0 and (k - k1) % 64 == 0. */
int tramba_linear2_cl(const void *x1, const void *x2, int k1, const void *w, const float *bias, const void *residual,
                      void *y, int64_t m, int n, int k, int act, int dtype, int out_dtype, void *stream);

/* y_pre = x @ w^T + bias and y_act = act(y_pre) from ONE launch (act = GELU or SiLU): the forward of `act(Linear(x))` on the
 * training path, whose backward needs the pre-activation (Models/modules.py:134-153 under autograd).  16-bit dtypes,
 * K % 64 == 0, N % 8 == 0; both outputs (M, N) in dtype. */
int tramba_linear_dual_cl(const void *x, const void *w, const float *bias, void *y_pre, void *y_act, int64_t m, int n,
                          int k, int act, int dtype, void *stream);
/* Weight (and bias) gradient of a 1x1 convolution under autograd -- what `loss.backward()` computes for every
 * Linear2d (Models/modules.py:10-19; the reference gets it from cuDNN/cuBLAS through F.conv2d's autograd):
 *   out[g][n*K + k]  = sum over batches b and tokens t of gy[g][b][t][n] * x[g][b][t][k]       (N x K, fp32)
 *   out[g][N*K + n]  = sum over b, t of gy[g][b][t][n]                                          (bias gradient, if want_bias)
 * Element (g, b, t, c) of an operand sits at base + b*bs + g*gs + t*ld + c (element strides): groups / nbatch serve the
 * per-direction contractions of the SS2D backward on (B, K, L, C) tensors; a plain Linear2d has groups = nbatch = 1.
 * out: (groups, N*K + N) f32.  workspace: tramba_wgrad_workspace() bytes of fp32 partial slabs (the token range is split
 * over workgroups, slabs are summed in a fixed order).  16-bit dtypes; N, K and every stride multiples of 8. */
size_t tramba_wgrad_workspace(int64_t m, int n, int k, int groups, int nbatch);
int tramba_wgrad_cl(const void *gy, const void *x, float *out, void *workspace, size_t workspace_bytes, int64_t m, int n,
                    int k, int groups, int nbatch, int64_t gy_bs, int64_t gy_gs, int gy_ld, int64_t x_bs, int64_t x_gs,
                    int x_ld, int want_bias, int dtype, void *stream);
/* The same launch stopped at the partial slabs: *nslab = slabs per group left in `workspace` as (groups, *nslab, N*K + N)
 * fp32, or 0 when a single slab was written straight to `out`.  The caller sums them later -- tramba_slab_sum, or
 * tramba_multi_sum together with the other pending sums of a training step (the gradient of a Linear2d hangs off the backward
 * chain: nothing reads it before the optimizer, reference train.py:86-89). */
int tramba_wgrad_parts_cl(const void *gy, const void *x, float *out, void *workspace, size_t workspace_bytes, int64_t m,
                          int n, int k, int groups, int nbatch, int64_t gy_bs, int64_t gy_gs, int gy_ld, int64_t x_bs,
                          int64_t x_gs, int x_ld, int want_bias, int dtype, void *stream, int *nslab);
/* y[z][t][0..N) = x[z][t][:] . w[z % groups][n][:]  for z < nz, t < m: x (nz, m, K) dtype, w (groups, N, K) dtype,
 * y fp32 rows of stride ldy (>= N; the columns past N are left untouched).  N <= 64, K % 8 == 0.  The dt_rank projection
 * of the SS2D backward (vmamba.py:236 under autograd): d(x_dbl ranks) = d(dt_raw) @ dt_projs_weight[k]. */
int tramba_rows_gemm_cl(const void *x, const void *w, float *y, int nz, int64_t m, int n, int k, int groups, int ldy,
                        int dtype, void *stream);
/* The 16-bit copies of a model's fp32 master weights that the next training step reads (the reference keeps fp32 weights
 * and lets autocast cast per call, Trambav6.py:151-154 / train.py:74-89): for every table entry t,
 *   dst[t] (rows, cols) = (dtype) src[t],   dst_t[t] (cols, rows) = its transpose      (either pointer may be 0: skipped)
 * table: DEVICE array of ntensors x 8 int64 {src, dst, dst_t, rows, cols, first_tile, dst_ld, dst_t_ld}; src is a dense
 * (rows, cols) matrix, the rows of dst / dst_t are dst_ld / dst_t_ld elements apart (= cols / rows for dense outputs;
 * larger when the matrix is a block of a padded layout, e.g. one direction of the x_proj weight); a tensor owns
 * ceil(rows/64)*ceil(cols/64) consecutive tiles starting at first_tile (ascending), total_tiles = their sum.  8-byte
 * alignment of every dst / dst_t row start when the leading dimensions are multiples of 4. */
int tramba_shadow_cast_multi(const void *table, int ntensors, int64_t total_tiles, int dtype, void *stream);
/* out[i] = sum over s < nslab of part[s*n + i], i < n, summed in slab order (deterministic): the per-workgroup partial sums
 * of the LayerNorm / depth-wise / scan parameter gradients.  n % 4 == 0, 16-byte aligned. */
int tramba_slab_sum(const float *part, float *out, int64_t n, int nslab, void *stream);
/* `count` such sums in ceil(count / 64) launches: outs[i][j] = sum over s < nslab[i] of parts[i][s*n[i] + j], the same
 * summation order as tramba_slab_sum / the slab sums of tramba_wgrad_cl.  parts / outs / n / nslab are HOST arrays of device
 * pointers and sizes (they travel to the kernel by value: nothing is copied to the device, hipGraph-capture safe).
 * _strided: slab s of table i starts stride[i] floats after slab s - 1 (>= n[i], a multiple of 4; NULL = dense) -- a row range
 * of a wider table summed straight into its place in another layout (the x_proj weight gradient leaves the GEMM in the scan
 * kernels' padded layout and lands in the parameter's own, vmamba.py:236: no concatenation kernel). */
int tramba_multi_sum_strided(const float *const *parts, float *const *outs, const int64_t *n, const int64_t *stride,
                             const int *nslab, int count, void *stream);
int tramba_multi_sum(const float *const *parts, float *const *outs, const int64_t *n, const int *nslab, int count,
                     void *stream);

/* The whole last decoder stage in one kernel (FinalPatchExpand_X4 + seg_layers[-1], Trambav6.py:132-137):
 * y (B, H*P, W*P) f32 = head(LayerNorm_128(pixel_shuffle_P(x @ w^T))).  x (B, H, W, Cin) dtype, w (P*P*128, Cin) dtype
 * (the expand Linear2d, no bias), ln_w / ln_b / head_w (128) f32.  16-bit dtypes, Cin % 64 == 0. */
int tramba_expand_norm_head_cl(const void *x, const void *w, const float *ln_w, const float *ln_b,
                               const float *head_w, float head_b, float *y, int batch, int h, int wd, int cin, int p,
                               float eps, int dtype, void *stream);

/* ------------------------------------------------------------------ dense convs of the VMamba stem */
/* Implicit-GEMM 3x3 / stride 2 / pad 1 convolution on a channels-last map (patch_embed[5] and the three
 * downsample convs, Models/vmamba.py:454,486): x (B, Hin, Win, Cin) -> y (B, ceil(Hin/2), ceil(Win/2), Cout).
 * w is K-major (Cout, 3, 3, Cin) = reference weight.permute(0,2,3,1), same dtype as x; Cin % 64 == 0. */
int tramba_conv3x3s2_cl(const void *x, const void *w, const float *bias, void *y, int batch, int hin,
                        int win, int cin, int cout, int dtype, void *stream);
/* patch_embed[0..4] fused (vmamba.py:481-485): conv 3x3/s2/p1 (3 -> 64) + bias + LayerNorm2d(64) + GELU.
 * img (B, 3, H, W) NCHW in f32 or `dtype`; w (64, 3, 3, 3) f32 reference layout; y (B, H/2, W/2, 64). */
int tramba_stem_conv_ln_gelu(const void *img, const float *w, const float *bias, const float *ln_w,
                             const float *ln_b, void *y, int batch, int h, int wd, float eps,
                             int img_dtype, int dtype, void *stream);

/* ------------------------------------------------------------------ evaluation (train.py:101-139 test_one_epoch) */
/* Per-image statistics of saliency maps against their masks, from which MAE, F-measure, E-measure and S-measure of
 * Evaluation/metrics.py follow without the maps leaving the GPU.  pred (B, H, W) f32 = sigmoid(logits) (NOT yet
 * min-max normalised: the kernel does Evaluation/metrics.py:13-19 itself); gt (B, H, W) u8, 0 / 1.
 * ints (B, TRAMBA_EVAL_NINT) i64: [0] mask area, [1] sum g*col, [2] sum g*row, [3] cx, [4] cy (S-measure centroid, 1-based
 *   split point as metrics.py:201-212), [5] / [6] pixels >= adaptive threshold inside / outside the mask, [7] H*W,
 *   [8 .. 264) histogram of uint8(p*255) inside the mask, [264 .. 520) outside.
 * dbl (B, TRAMBA_EVAL_NDBL) f64: [0] min, [1] max of pred, [2] sum p, [3] sum |p - g|, [4] [5] sum p, p^2 inside the mask,
 *   [6] [7] sum (1-p), (1-p)^2 outside, [8 + 4q + {0,1,2,3}] sum p, p^2, g, p*g over quadrant q (LT, RT, LB, RB),
 *   [24] the adaptive threshold, [32 + 3q + {0,1,2}] centred sums over quadrant q: (p - mean p)^2, (g - mean g)^2,
 *   (p - mean p)(g - mean g), [44] sum (p - mean)^2 inside the mask, [45] sum ((1-p) - mean)^2 outside;
 *   p = normalised prediction.  NaN where a region is empty, as numpy gives. */
#define TRAMBA_EVAL_NINT 520
#define TRAMBA_EVAL_NDBL 48
int tramba_saliency_stats(const float *pred, const unsigned char *gt, long long *ints, double *dbl, int batch, int h,
                          int w, void *stream);

/* ------------------------------------------------------------------ loss and optimizer of the training step */
/* The deep-supervision loss (train.py:76-85; utils/loss.py:6-11) of ONE output: logits (planes, h, w) f32 bilinearly resized
 * (F.interpolate(mode="bilinear"), align_corners=False; identity when the sizes agree) to the label (planes, hout, wout) f32,
 * then per plane the three sums behind binary_cross_entropy_with_logits + iou_loss:
 *   part[plane][blk][0..3) = sum of { max(z,0) - z y + log(1 + exp(-|z|)),  sigmoid(z) y,  sigmoid(z) + y }
 * over the pixels workgroup blk of nblk walked (any nblk >= 1; fixed summation order).  The resized map is never stored. */
int tramba_sod_loss_sums(const float *logits, const float *label, float *part, int planes, int h, int w, int hout, int wout,
                         int nblk, void *stream);
/* loss[0] = sum over outputs o < nout (<= 8) of weights[o] * ( mean bce + mean over planes of 1 - (I + 1) / (U - I + 1) ),
 * from the tables of tramba_sod_loss_sums (parts[o]: (planes, nblk[o], 3)), accumulated in fp64, and coefs[o] (planes, 4) f32 =
 * the per-plane coefficients { a, cI, cU, 0 } of d loss / d resized logit = a (p - y) + p (1 - p) (cI y + cU).
 * parts / nblk / weights (NULL: all 1) / coefs are HOST arrays; they travel by value (hipGraph-capture safe).
 * planes <= 512; npix = hout * wout. */
int tramba_sod_loss_finish(const float *const *parts, const int *nblk, const float *weights, float *const *coefs, int nout,
                           int planes, int64_t npix, float *loss, void *stream);
/* glogits (planes, h, w) f32 = gscale[0] * d loss / d logits of one output (gscale: device scalar, the incoming gradient of
 * the loss; NULL = 1).  A resized output goes through the adjoint of the bilinear resize, which is separable: the gradient
 * at label resolution is formed once per label pixel in LDS, folded along x into `workspace` (planes, hout, w) f32 and then
 * along y (no atomics, fixed order; cf. tramba_upsample_bilinear_bwd).  hout >= h, wout >= w, wout <= 4096. */
size_t tramba_sod_loss_grad_workspace(int planes, int h, int w, int hout, int wout);
int tramba_sod_loss_grad(const float *logits, const float *label, const float *coef, const float *gscale, float *glogits,
                         void *workspace, size_t workspace_bytes, int planes, int h, int w, int hout, int wout, void *stream);
/* One Adam step (torch.optim.Adam as train.py:266-280 builds it: amsgrad off, maximize off) on `count` fp32 tensors:
 *   steps[i][0] += 1 (device counters, part of the optimizer's state_dict);  g += weight_decay p;
 *   exp_avg = lerp(exp_avg, g, 1 - beta1);  exp_avg_sq = beta2 exp_avg_sq + (1 - beta2) g g;
 *   p -= lr / (1 - beta1^t) * exp_avg / (sqrt(exp_avg_sq) / sqrt(1 - beta2^t) + eps)       (bias corrections in fp64)
 * The six arrays are HOST arrays of device pointers / element counts; the tensors travel to the kernels by value, 72 per
 * launch (nothing is copied to the device: hipGraph-capture safe).  Any alignment; 16-byte aligned tensors take the
 * vector path. */
int tramba_adam_step(float *const *params, const float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                     float *const *steps, const int64_t *numel, int count, double lr, double beta1, double beta2, double eps,
                     double weight_decay, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TRAMBA_HIP_H */
