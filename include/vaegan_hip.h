/* vaegan_hip.h -- C ABI of libvaegan_hip.so: the MI355X (gfx950) kernels behind
 * the CelebA 64x64 beta-VAE-GAN training iteration.
 *
 * The reference (RicoFio/disentangle_mlp) has no FFI of its own: its hot path is
 * Python torch.nn modules (models/model.py) whose arithmetic is dispatched to
 * ATen.  Each entry point below replaces one ATen call site of that path; the
 * citation next to it is the reference line that reaches the op.  A maintainer
 * binds them with ctypes (INTEGRATION.md shows the stub) -- no torch types
 * cross this boundary, only device pointers, sizes and a hipStream_t.
 *
 * Conventions
 *   - all tensors are contiguous fp32, NCHW (activations), (Cout,Cin,5,5)
 *     (Conv2d weights), (Cin,Cout,5,5) (ConvTranspose2d weights), (out,in)
 *     (Linear weights): exactly the reference's state_dict layouts.
 *   - kernels never allocate; outputs / workspaces are caller-owned device
 *     buffers; `stream` is a hipStream_t (NULL = default stream).
 *   - every function returns 0 on success, VG_ERR_BAD_ARG (-1) for a rejected
 *     argument, VG_ERR_WORKSPACE (-2) for a too-small workspace, otherwise the
 *     hipError_t of the failed launch.  No exceptions cross the ABI.
 *   - launches are asynchronous, hold no global mutable state and are
 *     HIP-graph capturable.  (The tile-forcing knobs tests and tuning scripts use are process-globals,
 *     so they exist only in a second build of the same sources, libvaegan_hip_tuning.so, compiled with
 *     -DVG_TUNING: the section at the end of this header.)
 */
#ifndef VAEGAN_HIP_H
#define VAEGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VG_ERR_BAD_ARG (-1)
#define VG_ERR_WORKSPACE (-2)

/* activation codes for the fused BatchNorm kernels */
#define VG_ACT_NONE 0
#define VG_ACT_RELU 1  /* nn.ReLU          model.py:452,463,493 */
#define VG_ACT_LRELU 2 /* nn.LeakyReLU(.2) model.py:391,404     */

/* ABI version of this header: bumped on every incompatible change of a signature, of a packed-filter layout or of a
 * workspace contract.  vg_version() returns the value the library was built with; a binding must refuse a library
 * whose version is not the header's (the .so files are build products that travel with the working tree: a stale one
 * still exports every old symbol).  3: round 3.  4: round 4 (fp16 planes: VG_PLANES_F16, the *_amax arguments). */
#define VG_ABI_VERSION 5
int vg_version(void);

/* ---- 5x5 convolutions, padding 2, stride 1 or 2 ----------------------------
 * y[B,Cout,OH,OW] = conv2d(x[B,Cin,H,W], w[Cout,Cin,5,5]) + bias;  OH=(H-1)/s+1.
 * nn.Conv2d forward: model.py:450,453,456 (encoder), :389,392,395,398 (discr.).
 * Also the input gradient of vg_convT5x5_fwd.  bias may be NULL. */
int vg_conv5x5_fwd(const float* x, const float* w, const float* bias, float* y,
                   int B, int Cin, int H, int W, int Cout, int stride, void* stream);

/* y[B,Cout,s*H,s*W] = conv_transpose2d(x[B,Cin,H,W], w[Cin,Cout,5,5], stride s,
 * padding 2, output_padding s-1) + bias.  nn.ConvTranspose2d forward with the
 * literal output_size of model.py:558-564 (deconv1..4, :495-507).  Also the
 * input gradient of vg_conv5x5_fwd (then w is the Conv2d weight [Cout,Cin,5,5]
 * read as [Cin_T=Cout, Cout_T=Cin], H*s must equal the conv input size). */
int vg_convT5x5_fwd(const float* x, const float* w, const float* bias, float* y,
                    int B, int Cin, int H, int W, int Cout, int stride, void* stream);

/* Pre-packed filters.  The implicit-GEMM kernels run fastest when the filter slab of a K
 * chunk is a verbatim 16-byte copy: vg_conv5x5_pack rewrites a weight tensor once per
 * weight version into [parity class][ci][tap][cout] (cout innermost, zero padded to a
 * multiple of 128 output / 8 input channels), vg_conv5x5_fwd_packed /
 * vg_convT5x5_fwd_packed consume it.  On the same tile variant the results are bit-identical
 * to the plain entry points (same K order); the tile heuristics of the two may differ.
 *   transposed = 0: w is [Cout,Cin,5,5], for vg_conv5x5_fwd_packed (stride ignored);
 *   transposed = 1: w is [Cin,Cout,5,5], for vg_convT5x5_fwd_packed with the SAME stride
 *                   (the stride-2 kernel walks the 4 output-parity classes separately).
 * `packed` holds vg_conv5x5_packed_floats(Cout, Cin) floats, 16-byte aligned. */
size_t vg_conv5x5_packed_floats(int Cout, int Cin);
int vg_conv5x5_pack(const float* w, float* packed, int Cout, int Cin, int transposed, int stride,
                    void* stream);
int vg_conv5x5_fwd_packed(const float* x, const float* packed, const float* bias, float* y,
                          int B, int Cin, int H, int W, int Cout, int stride, void* stream);
int vg_convT5x5_fwd_packed(const float* x, const float* packed, const float* bias, float* y,
                           int B, int Cin, int H, int W, int Cout, int stride, void* stream);
/* vg_conv5x5_fwd_packed that also leaves the statistics of its output for the BatchNorm that follows
 * (vg_conv_fusion.stats semantics: [slot][Cout][2] partial sums; vg_bn_finalize_stats consumes them) -- the
 * 3-channel first layers (model.py:389-390, 450-451), whose 67 MB / 34 MB outputs would otherwise be read once
 * more only to be summed.  `stats` holds vg_conv5x5_fwd_packed_stats_floats(...) floats. */
size_t vg_conv5x5_fwd_packed_stats_floats(int B, int Cin, int H, int W, int Cout, int stride);
int vg_conv5x5_fwd_packed_stats(const float* x, const float* packed, const float* bias, float* y,
                                int B, int Cin, int H, int W, int Cout, int stride,
                                float* stats, size_t stats_floats, void* stream);

/* Split arithmetics of vg_conv5x5_fwd / vg_convT5x5_fwd: every fp32 operand is split
 * into `planes` 16-bit values and the products whose plane indices sum to < planes are evaluated on the 16-bit
 * MFMA (v_mfma_f32_32x32x16_bf16 / _f16) with fp32 accumulation:
 *   planes = 2 | VG_PLANES_F16   fp16 hi/lo = 11 + 11 significand bits (residual <= 2^-24 relative), 3 MFMAs per product
 *               ("fp16x3"): fp32-equivalent (4e-7..6e-7 against fp64) at half the matrix work of bf16x6.  fp16 has 5
 *               exponent bits: each operand tensor is multiplied by an exact power of two taken from an UPPER BOUND of
 *               its largest magnitude, which the caller supplies in device memory (vg_conv_fusion.in_amax, the *_amax
 *               arguments below; vg_absmax* compute one, vg_bn_act_bwd / vg_bn_finalize_stats / vg_affine_act emit one
 *               on their way out); the scales are undone on the fp32 accumulators.  A bound that is too small overflows
 *               fp16 (inf / NaN in the output, never a silently wrong number); elements more than 2^-16 below the bound
 *               keep fewer than 22 bits: their error is bounded by 2^-40 of the bound.  THE PRODUCT DEFAULT of the
 *               Python layer.
 *   planes = 3  bf16 hi/mid/lo = the whole 24-bit mantissa, 6 MFMAs per product ("bf16x6"): every dropped
 *               term is below 2^-24 -- fp32-equivalent at any dynamic range (4e-7..9e-7 against fp64; the exact
 *               fp32-input MFMA kernels above: 5e-7..1e-6).  Opt-in; the 3-channel edge kernels always use it.
 *   planes = 2  bf16 hi/lo, 3 MFMAs per product ("bf16x3"): ~4.5e-6 relative error; opt-in.
 * Requires Cin % 16 == 0.  `packed` holds vg_conv5x5_packed_bf16split_bytes(Cout, Cin, planes) bytes
 * (16-byte aligned), written by vg_conv5x5_pack_bf16split once per weight version:
 *   transposed = 0: from w[Cout,Cin,5,5] for vg_conv5x5_fwd_bf16split with the SAME stride (the stride-2
 *                   kernel pairs taps differently from the stride-1 one);
 *   transposed = 1: from w[Cin,Cout,5,5] for vg_convT5x5_fwd_bf16split with the SAME stride.
 * Stride 2 runs on the 8-wavefront kernel of conv_ring.hip (both MFMA operands from LDS, the filter through a
 * global_load_lds DMA ring); layers whose tile grid would leave CUs idle split their input channels over
 * workgroups and sum the partial outputs in a fixed order: query the workspace (0 for most shapes). */
/* Fused BatchNorm around a stride-2 split-bf16 convolution (replaces the separate statistics and
 * normalise passes of F.batch_norm reached from model.py:451-458, 496-505, 390-400):
 *   in_scale / in_shift / in_act: the input is read as act(x * in_scale[c] + in_shift[c]) -- the producing layer's
 *     train-mode BatchNorm (scale = gamma * invstd, shift = beta - mean * scale) and activation (VG_ACT_*) applied
 *     while the patch is staged; zero padding pads the activated tensor.  NULL: the input is used as it is.
 *   stats: per-channel partial sums of THIS layer's output, [slot][Cout][2] floats (sum y, sum y^2 over the slot's
 *     pixels, bias included), `stats_floats` = vg_conv*_stats_floats(...) of them; vg_bn_finalize_stats turns them
 *     into the next BatchNorm's coefficients.  NULL: none.
 * vg_conv5x5_bf16split_fusable says whether a layer's kernel takes a non-empty fusion; *_stats_floats returns 0 for
 * layers that cannot emit statistics (other kernels, K-split layers). */
#define VG_PLANES_F16 0x100 /* flag of the `planes` arguments: the planes are IEEE half precision (with 2 planes only) */
typedef struct vg_conv_fusion {
  const float* in_scale;
  const float* in_shift;
  int in_act;
  float* stats;
  size_t stats_floats;
  /* fp16 planes (required then, ignored otherwise; not a fusion: every kernel of the family takes it): in_amax[0], DEVICE
   * memory, >= the largest |value| of the input as the kernel reads it (after in_scale / in_shift / in_act) */
  const float* in_amax;
} vg_conv_fusion;
int vg_conv5x5_bf16split_fusable(int transposed, int Cin, int Cout, int stride);
/* (the tile a layer runs on, and with it the slot count and the K split, depends on the arithmetic: two fp16 planes leave
 * room in the LDS for larger tiles) */
size_t vg_conv5x5_fwd_bf16split_stats_floats(int B, int Cin, int H, int W, int Cout, int stride, int planes);
size_t vg_convT5x5_fwd_bf16split_stats_floats(int B, int Cin, int H, int W, int Cout, int stride, int planes);
size_t vg_conv5x5_packed_bf16split_bytes(int Cout, int Cin, int planes);
/* w_amax (fp16 planes; NULL otherwise): w_amax[0] >= max |w|, DEVICE memory (vg_absmax / vg_absmax_multi); the pack
 * keeps the inverse of the filter's scale in a 16-byte trailer for the convolution's epilogue */
int vg_conv5x5_pack_bf16split(const float* w, void* packed, int Cout, int Cin, int transposed, int stride,
                           int planes, const float* w_amax, void* stream);
/* The same for several filters in one launch (`entries` is a HOST array; 24 filters per kernel launch): what a
 * training iteration calls after each optimizer step for every filter that step has changed. */
typedef struct {
  const float* w;
  void* packed;
  int Cout, Cin, transposed, stride;
  const float* w_amax; /* as vg_conv5x5_pack_bf16split */
} VgPackEntry;
int vg_conv5x5_pack_bf16split_multi(const VgPackEntry* entries, int count, int planes, void* stream);
size_t vg_conv5x5_fwd_bf16split_workspace_bytes(int B, int Cin, int H, int W, int Cout, int stride, int planes);
int vg_conv5x5_fwd_bf16split(const float* x, const void* packed, const float* bias, float* y,
                          int B, int Cin, int H, int W, int Cout, int stride, int planes,
                          void* workspace, size_t workspace_bytes, const vg_conv_fusion* fuse, void* stream);
size_t vg_convT5x5_fwd_bf16split_workspace_bytes(int B, int Cin, int H, int W, int Cout, int stride, int planes);
int vg_convT5x5_fwd_bf16split(const float* x, const void* packed, const float* bias, float* y,
                           int B, int Cin, int H, int W, int Cout, int stride, int planes,
                           void* workspace, size_t workspace_bytes, const vg_conv_fusion* fuse, void* stream);
/* Stride-1 transposed convolution 32 -> (1..3) channels in the same arithmetic, on the PLAIN filter w (Cin, Cout, 5, 5):
 * the decoder's last layer (ConvTranspose2d(32, 3, 5, 1, 2), /root/reference/models/model.py:507) and the data
 * gradient of the discriminator's first layer (model.py:389).  One pass over x with the filter resident in registers
 * (csrc/conv_thin_mfma.hip).  in_scale / in_shift / in_act: x is read as act(x * in_scale[ci] + in_shift[ci]) (the
 * producing layer's BatchNorm + activation, vg_conv_fusion semantics); NULL: as it is.  _ok: 1 when the shape is taken
 * (Cin == 32, Cout <= 3, W a multiple of 16 up to 128); otherwise use vg_convT5x5_fwd. */
int vg_convT5x5_s1_thin_bf16split_ok(int Cin, int H, int W, int Cout);
int vg_convT5x5_s1_thin_bf16split(const float* x, const float* w, const float* bias, float* y, int B, int Cin,
                                  int H, int W, int Cout, int planes, const float* in_scale, const float* in_shift,
                                  int in_act, void* stream);
/* Convolution from 1..3 input channels in the same arithmetic, on the PLAIN filter w (Cout, Cin, 5, 5): the first
 * layers of the discriminator (Conv2d(3, 32, 5, 1, 2), model.py:389) and the encoder (Conv2d(3, 64, 5, 2, 2),
 * model.py:450).  One write pass over y with the filter resident in registers (csrc/conv_thin_fwd.hip); `stats`
 * (or NULL) receives the per-channel partial sums of y for the BatchNorm that follows, vg_conv_fusion.stats layout
 * ([slot][Cout][2], vg_conv5x5_thin_bf16split_stats_floats(...) floats, slots = floats / (2 Cout)).  _ok: 1 when the
 * shape is taken (output width a multiple of 32, (output width / 32) * ceil(Cout / 32) <= 8); otherwise use
 * vg_conv5x5_fwd. */
int vg_conv5x5_thin_bf16split_ok(int Cin, int H, int W, int Cout, int stride);
size_t vg_conv5x5_thin_bf16split_stats_floats(int B, int Cin, int H, int W, int Cout, int stride);
int vg_conv5x5_thin_bf16split(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W,
                              int Cout, int stride, int planes, float* stats, size_t stats_floats, void* stream);
/* ... and its weight gradient (dw (Cout, Cin, 5, 5), Cin <= 3, Cout <= 64; also the decoder's ConvTranspose2d(32, 3)
 * with the roles of x and gy swapped): one read pass over gy, x split once per workgroup into shifted bf16 plane
 * copies in LDS (csrc/conv_thin_wgrad.hip).  The workspace query returns 0 for shapes it does not take (output width
 * not a multiple of 16, ...): use vg_conv5x5_wgrad then.  workspace: 16-byte aligned; gy 16-byte aligned. */
size_t vg_conv5x5_thin_wgrad_bf16split_workspace_bytes(int B, int Cin, int H, int W, int Cout, int stride, int planes);
int vg_conv5x5_thin_wgrad_bf16split(const float* x, const float* gy, float* dw, int B, int Cin, int H, int W, int Cout,
                                    int stride, int planes, void* workspace, size_t workspace_bytes,
                                    /* gy read as act(gy * gy_scale[c] + gy_shift[c]) (NULL, NULL, 0: plain): the weight
                                     * gradient of ConvTranspose2d(32, 3) passes the layer's input here, the train-mode
                                     * BatchNorm + ReLU of its producer (model.py:505) is applied on load */
                                    const float* gy_scale, const float* gy_shift, int gy_act,
                                    int accumulate /* dw += (see vg_conv5x5_wgrad) */, void* stream);
/* vg_conv5x5_wgrad in the same arithmetic.  The reduction runs over images in groups of 16: gy is re-laid
 * batch-innermost inside the call (B zero-padded to a multiple of 16), x is staged straight from NCHW; needs
 * OW % 8 == 0 -- the workspace query returns 0 for shapes it does not take (use vg_conv5x5_wgrad).
 * workspace: 16-byte aligned. */
size_t vg_conv5x5_wgrad_bf16split_workspace_bytes(int B, int Cin, int H, int W, int Cout, int stride, int planes);
int vg_conv5x5_wgrad_bf16split(const float* x, const float* gy, float* dw, int B, int Cin, int H, int W,
                            int Cout, int stride, int planes, void* workspace, size_t workspace_bytes,
                            const float* in_scale, const float* in_shift, int in_act, int affine_on_gy,
                            /* fp16 planes (NULL otherwise): x_amax[0] >= max |x|, gy_amax[0] >= max |gy|, each of the
                             * operand as the kernel reads it (after the affine + activation where one applies), DEVICE */
                            const float* x_amax, const float* gy_amax,
                            int accumulate /* dw += (see vg_conv5x5_wgrad) */, void* stream);
/* in_scale / in_shift / in_act: one operand is read as act(v * scale[c] + shift[c]) (vg_conv_fusion semantics) --
 * x (Cin coefficients) when affine_on_gy == 0, gy (Cout coefficients; the weight gradient of a transposed
 * convolution passes the layer's input there) otherwise.  NULL, NULL, 0, 0: both operands as they are. */

/* ---- Linear layers in the fp16x3 arithmetic (csrc/gemm_split.hip) -------------------------------------------------
 * C[m][n] = sum_k A(m,k) * B(n,k) (+ bias[n]), A(m,k) = A[m*a_row_stride + k*a_k_stride], B alike; C row-major [M][N].
 * Per operand either the reduction index is contiguous (k stride 1; row stride % 4 == 0, 16-byte aligned base) or the
 * row index is (row stride 1).  K % 32 == 0.  One entry point serves nn.Linear (/root/reference/models/model.py:460-471,
 * 402-408, 490-492; replaces torch.nn.functional.linear and the two backward GEMMs autograd derives from it): forward
 * y = x W^T (A = x, B = W), data gradient gx = gy W (A = gy, B = W with b_row_stride = 1, b_k_stride = in_features),
 * weight gradient gW = gy^T x (A = gy with a_row_stride = 1, B = x with b_row_stride = 1; K = batch).  Arithmetic: the
 * convolutions' fp16x3 (planes = 2 | VG_PLANES_F16 above) -- a_amax[0] >= max |A|, b_amax[0] >= max |B|, DEVICE memory.
 * Short output grids split the reduction over workgroups: query the workspace (partial tiles, summed in a fixed order;
 * no atomics -- results are reproducible run to run). */
size_t vg_gemm_nt_f16x3_workspace_bytes(int M, int N, int K);
int vg_gemm_nt_f16x3(const float* A, const float* B, const float* bias, float* C, int M, int N, int K,
                     long a_row_stride, long a_k_stride, long b_row_stride, long b_k_stride,
                     const float* a_amax, const float* b_amax, void* workspace, size_t workspace_bytes, void* stream);

/* Bounds for the fp16 planes: amax[0] = max(amax[0], max |x|) over n floats -- an atomic maximum on the bit pattern
 * (order-independent; a NaN in x ends up in the bound); the caller zeroes amax[0] first (or keeps accumulating a bound
 * over several tensors).  _affine: over act(x[b][c][hw] * scale[c] + shift[c]) (vg_conv_fusion semantics).  _multi: many
 * tensors in one launch (`entries` is a HOST array) -- the filters an optimizer step has changed. */
int vg_absmax(const float* x, size_t n, float* amax, void* stream);
int vg_absmax_affine(const float* x, const float* scale, const float* shift, int act, int B, int C, int HW, float* amax,
                     void* stream);
typedef struct {
  const float* x;
  size_t n;
  float* amax;
} VgAbsmaxEntry;
int vg_absmax_multi(const VgAbsmaxEntry* entries, int count, void* stream);

/* dw[Cout,Cin,5,5] = sum_{b,oh,ow} gy[b,co,oh,ow] * x[b,ci,s*oh+kh-2,s*ow+kw-2].
 * Weight gradient of nn.Conv2d (autograd of model.py:450...; new_betavaegan.py:103,121)
 * and, with the roles swapped (x := gy_T, gy := x_T), of nn.ConvTranspose2d.
 * x is [B,Cin,H,W], gy is [B,Cout,OH,OW] with OH=(H-1)/s+1.  Deterministic:
 * split-K partial slabs in `workspace` are summed in a fixed order. */
size_t vg_conv5x5_wgrad_workspace_bytes(int B, int Cin, int H, int W, int Cout, int stride);
int vg_conv5x5_wgrad(const float* x, const float* gy, float* dw, int B, int Cin, int H, int W,
                     int Cout, int stride, void* workspace, size_t workspace_bytes,
                     /* accumulate != 0: the result is ADDED to what dw holds (in the final fixed-order slab sum) -- a layer
                      * applied twice before one backward (D on the real and the generated batch, the decoder on the prior
                      * sample and the reconstruction: new_betavaegan.py:99-121, 144-163) gets its second weight gradient
                      * without the separate addition autograd would launch; fp addition is commutative: same bits */
                     int accumulate, void* stream);

/* out[c] = sum_{b,hw} g[b,c,hw]   (bias gradient of the convolutions).
 * workspace >= vg_bn_workspace_bytes(C). */
int vg_channel_sum(const float* g, float* out, int B, int C, int HW,
                   void* workspace, size_t workspace_bytes, void* stream);

/* ---- train-mode BatchNorm (+ReLU / LeakyReLU(0.2)) --------------------------
 * x is [B,C,HW] (HW=1 for BatchNorm1d).  Batch statistics (biased variance,
 * eps), y = act(gamma*(x-mean)*invstd + beta); running_mean/var updated with
 * `momentum` (unbiased variance), as F.batch_norm(training=True) does.
 * nn.BatchNorm2d/1d in train mode: model.py:451-458,462,468,492,496-505,390-400
 * (the reference never calls .eval()).  save_mean / save_invstd ([C]) are kept
 * for the backward.  running_mean / running_var may be NULL.
 * y_amax (may be NULL; ignored for HW == 1): max |y| is added to y_amax[0] (atomic maximum, DEVICE memory, zeroed by the
 * caller) -- the bound the fp16-plane consumers of y (a convolution, vg_gemm_nt_f16x3) scale it by.
 * workspace >= vg_bn_workspace_bytes(C). */
size_t vg_bn_workspace_bytes(int C);
int vg_bn_act_fwd(const float* x, const float* gamma, const float* beta, float* y,
                  float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                  int B, int C, int HW, float eps, float momentum, int act, float* y_amax,
                  void* workspace, size_t workspace_bytes, void* stream);
/* gx, dgamma, dbeta from gy, the saved x and statistics (the activation mask is
 * recomputed from x, bit-identically to the forward). gx must not alias gy. */
int vg_bn_act_bwd(const float* gy, const float* x, const float* gamma, const float* beta,
                  const float* save_mean, const float* save_invstd,
                  float* gx, float* dgamma, float* dbeta,
                  int B, int C, int HW, int act,
                  int accumulate_param_grads /* dgamma, dbeta += (a layer used twice before one backward) */,
                  float* gx_amax /* NULL, or DEVICE: gx_amax[0] = max(gx_amax[0], max |gx|) -- see vg_absmax */,
                  void* workspace, size_t workspace_bytes, void* stream);
/* Fused-BatchNorm helpers.  vg_bn_finalize_stats turns a convolution's statistics slots
 * ([nslots][C][2] floats: vg_conv_fusion.stats) into the coefficients of the train-mode BatchNorm that follows it
 * -- save_mean / save_invstd for backward, scale = gamma * invstd and shift = beta - mean * scale for the consumer's
 * in_scale / in_shift -- and updates the running statistics (momentum, unbiased variance) as vg_bn_act_fwd does;
 * count = B * H * W.  vg_bn_stats does the same from a pass over x (layers whose producer cannot emit statistics).
 * vg_affine_act materialises y = act(x * scale[c] + shift[c]) (16-byte accesses when HW % 4 == 0) for consumers that cannot apply the
 * coefficients while they load.
 * act_amax (NULL, or DEVICE, zeroed by the caller): receives an upper bound of max |act(BN(x))| over the tensor -- from
 * the coefficients alone (|x - mean| <= sigma sqrt(count - 1), so |BN(x)| <= |gamma| sqrt(count) + |beta|): what an
 * fp16-plane convolution that applies scale / shift on load needs as vg_conv_fusion.in_amax, without a pass over x.
 * y_amax of vg_affine_act: the exact max |y| (see vg_absmax). */
int vg_bn_finalize_stats(const float* stats, int nslots, int C, double count, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                         float* scale, float* shift, float eps, float momentum, float* act_amax,
                         void* workspace, size_t workspace_bytes, /* vg_bn_workspace_bytes(C) */ void* stream);
int vg_bn_stats(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                float* save_mean, float* save_invstd, float* scale, float* shift, int B, int C, int HW,
                float eps, float momentum, float* act_amax, void* workspace, size_t workspace_bytes, void* stream);
int vg_affine_act(const float* x, const float* scale, const float* shift, float* y, int B, int C, int HW, int act,
                  float* y_amax, void* stream);


/* ---- elementwise activations ------------------------------------------------
 * LeakyReLU(0.2) after lth_features (model.py:404), tanh after deconv4
 * (model.py:509,565), both with an optional per-channel bias add fused in:
 * y = act(x + bias[c]) for x [B,C,HW]; bias may be NULL. */
int vg_bias_act_fwd(const float* x, const float* bias, float* y, int B, int C, int HW, int act_kind, void* stream);
/* gx_amax (may be NULL): max |gx| is added to gx_amax[0] (atomic maximum; DEVICE memory, zeroed by the caller) -- the bound an
 * fp16-plane consumer of gx (the Linear layers' backward GEMMs) scales it by */
int vg_act_bwd(const float* gy, const float* y, float* gx, size_t n, int act_kind, float* gx_amax, void* stream);
#define VG_EW_LRELU 0
#define VG_EW_TANH 1
#define VG_EW_SIGMOID 2

/* ---- fused losses -----------------------------------------------------------
 * Loss scalars are written to device memory (no host sync). */

/* z = mu + eps*exp(logvar/2) (model.py:532-535) and
 * kl[0] = beta * (-0.5 * sum(1 + logvar - mu^2 - exp(logvar))) (new_betavaegan.py:64-65);
 * kl_rows ([B], may be NULL) gets the un-scaled per-sample KL (model.py:321). */
int vg_reparam_kl_fwd(const float* mu, const float* logvar, const float* eps, float* z, float* kl,
                      float* kl_rows, int B, int D, float beta, void* stream);
/* gmu = gz + gkl*beta*mu ; glogvar = gz*eps*0.5*exp(logvar/2) + gkl*beta*0.5*(exp(logvar)-1).
 * gz (tensor) and gkl (DEVICE scalar, the upstream gradient of kl[0]) may each be
 * NULL (treated as 0). */
int vg_reparam_kl_bwd(const float* gz, const float* mu, const float* logvar, const float* eps,
                      const float* gkl, float beta, float* gmu, float* glogvar, int B, int D, void* stream);

/* out[i] = g[i] * s[0] with s a DEVICE scalar: applies the upstream gradient of a
 * loss scalar to a gradient precomputed by the fused loss kernels, without a
 * host read of s (out may alias g). */
int vg_scale_by_scalar(const float* g, const float* s, float* out, size_t n, void* stream);

/* loss[0] = scale * sum((a-b)^2);  ga = gscale * 2*scale*(a-b)  (ga may be NULL).
 * scale=0.5: Dis_l / SIM (new_betavaegan.py:67-69); scale=1: pixel MSE (:71-75). */
int vg_sqdiff_loss(const float* a, const float* b, float* loss, float* ga, size_t n,
                   float scale, float gscale, void* workspace, size_t workspace_bytes, void* stream);
size_t vg_sqdiff_workspace_bytes(size_t n);

/* nn.BCELoss() against a constant label (new_betavaegan.py:53,101,118,153-154):
 * loss[0] = (1/divisor) * sum(-(t*max(log p,-100) + (1-t)*max(log(1-p),-100)));
 * gp = gscale/divisor * (p-t)/max(p*(1-p),1e-12)   (gp may be NULL).
 * divisor = B locally; the global batch under data parallelism. */
int vg_bce_loss(const float* p, float target, float* loss, float* gp, int B, float divisor,
                float gscale, void* stream);
/* The same with the label read from DEVICE memory (target_dev[0]): the per-iteration soft / flipped label of
 * new_betavaegan.py:89-90 changes between replays of an iteration captured in a HIP graph. */
int vg_bce_loss_dev(const float* p, const float* target_dev, float* loss, float* gp, int B, float divisor,
                    float gscale, void* stream);

/* the discriminator's head and its GAN loss fused (rows kernel + fixed-order sum forward, one kernel backward) -- Linear(2048 -> 1) + Sigmoid
 * (model.py:406-408) followed by nn.BCELoss against the iteration's label (new_betavaegan.py:101,118,153-154):
 *   p[b] = sigmoid(<feat[b,:], w> + bias[0]);  loss[0] = the vg_bce_loss of p;
 *   dlogit[b] = d loss / d logit[b] = (1/divisor) (p-t)/max(p(1-p),1e-12) * p (1-p)      (may be NULL)
 * (the same fp32 expressions as vg_bias_act_fwd(sigmoid) -> vg_bce_loss -> vg_act_bwd).  The label is `target`, or
 * target_dev[0] when target_dev != NULL (captured iterations).  Backward, given g = gloss[0] (device; NULL = 1):
 *   gfeat[b,k] = g dlogit[b] w[k];  gw[k] = g sum_b dlogit[b] feat[b,k];  gb[0] = g sum_b dlogit[b]
 * (each output may be NULL: a frozen discriminator only relays gfeat). */
size_t vg_dot_sigmoid_bce_workspace_bytes(int B);      /* the row terms of the loss (double), summed in a fixed order */
int vg_dot_sigmoid_bce_fwd(const float* feat, const float* w, const float* bias, float target, const float* target_dev,
                           float* p, float* loss, float* dlogit, int B, int K, float divisor,
                           void* workspace, size_t workspace_bytes, void* stream);
int vg_dot_sigmoid_bce_bwd(const float* dlogit, const float* gloss, const float* feat, const float* w, float* gfeat,
                           float* gw, float* gb, int B, int K, int accumulate_param_grads /* gw, gb += */, void* stream);

/* ---- Adam (experiments/new_betavaegan.py:49-50: optim.Adam defaults, stepped 3x per iteration)
 * For each tensor: m += (1-beta1)(g-m); v = beta2 v + (1-beta2) g^2;
 * p -= (lr / bias_correction1) * m / (sqrt(v) / bias_correction2_sqrt + eps),
 * with bias_correction1 = 1 - beta1^step and bias_correction2_sqrt = sqrt(1 - beta2^step) computed by
 * the caller (double).  `tensors` is a HOST array of DEVICE pointers (fp32, n elements each);
 * any number of tensors, 24 per kernel launch.  No weight decay / amsgrad / maximize. */
typedef struct {
  float* p;
  const float* g;
  float* m;
  float* v;
  size_t n;
  /* NULL, or DEVICE memory: amax[0] = max(amax[0], max |p| after the update) -- the bound the fp16x3 Linear GEMMs
   * (vg_gemm_nt_f16x3) scale this weight by; the caller zeroes it before the step */
  float* amax;
} VgAdamTensor;
int vg_adam_step(const VgAdamTensor* tensors, int count, double lr, double beta1, double beta2, double eps,
                 double bias_correction1, double bias_correction2_sqrt, void* stream);
/* The same step with its scalars on the DEVICE, for an iteration captured in a HIP graph (kernel arguments are frozen
 * at capture; the step count is not).  vg_adam_prepare (one thread) writes scalars[0] = lr / (1 - beta1^step) and
 * scalars[1] = sqrt(1 - beta2^step), formed in double: with advance_device_counter the device counter step_dev[0] is
 * advanced by one and used (captured steps); without, the host's `step` (>= 1) is used and -- step_dev may be NULL --
 * stored in the counter (eager steps between replays keep it current; same expression, same bits).
 * vg_adam_step_dev is vg_adam_step reading those two scalars. */
int vg_adam_prepare(double step, double* step_dev, int advance_device_counter, double lr, double beta1, double beta2,
                    float* scalars, void* stream);
int vg_adam_step_dev(const VgAdamTensor* tensors, int count, double beta1, double beta2, double eps,
                     const float* scalars, void* stream);

/* ---- image I/O either side of the step -----------------
 * Input pipeline of dataloader/dataset.py:37-43 (ToTensor + Normalize(mean, std) of a
 * shuffled batch) on a uint8 image cache resident in HBM, layout [N][H][W][C] as decoded:
 *   out[b][c][h][w] = (cache[index[b]][h][w][c] / 255 - mean) / std      (IEEE fp32, bit-exact)
 * index holds B int64 image numbers (DEVICE memory), all in [0, N). */
int vg_u8_gather_normalize(const uint8_t* cache, const int64_t* index, float* out, int B, int C,
                           int H, int W, float mean, float stdv, void* stream);

/* out2[0] = min(x), out2[1] = max(x) over n floats (DEVICE scalars; exact in any order). */
size_t vg_minmax_workspace_bytes(size_t n);
int vg_minmax(const float* x, size_t n, float* out2, void* workspace, size_t workspace_bytes,
              void* stream);

/* torchvision.utils.save_image / make_grid (pinned 0.2.1) as called by utils/utils.py:12-36:
 * x[B,C,H,W] (C = 1 or 3) -> uint8 HWC grid[GH][GW][3].  minmax (DEVICE, from vg_minmax; NULL =
 * normalize=False): v = (clamp(x, min, max) - min) / (max - min + 1e-5); byte = trunc(clamp(
 * v * 255, 0, 255)).  B == 1 gives the un-padded image; otherwise nrow images per row, `padding`
 * pixels of pad_value between and around them.  vg_image_grid_shape returns GH, GW. */
int vg_image_grid_shape(int B, int H, int W, int nrow, int padding, int* grid_h, int* grid_w);
int vg_image_grid_u8(const float* x, const float* minmax, uint8_t* grid, int B, int C, int H, int W,
                     int nrow, int padding, float pad_value, void* stream);

#ifdef VG_TUNING
/* ---- tuning build only (libvaegan_hip_tuning.so): process-global knobs, never in the product library ---- */
/* Diagnostics / tuning only: force tile variant `variant` (>= 0; -1 restores the
 * heuristic) for mode 0 (vg_conv5x5_fwd) or 1 (vg_convT5x5_fwd).  Process-global;
 * never used by the product path. */
int vg_debug_set_conv_tile(int mode, int variant);
/* Diagnostics / tuning only: what=0 caps the wgrad cout tile (32/64/128, -1 = heuristic);
 * what=1 sets the split-K workgroup target (-1 = heuristic); what=2 the K groups of the 128-row
 * tile (1 or 2); what=3 the input channels per column tile (5 or 10); what=4: 0 = scalar gy loads;
 * what=5: output-pixel rows per chunk of vg_conv5x5_wgrad_bf16split (1 or 2; 0 = chosen by its plan). */
int vg_debug_set_wgrad(int what, int value);
/* split-bf16 kernels of conv_bf16split.hip: 0 = 128x128, 1 = 64x128, 2 = 64x64, 3 = 32x128, 4 = 32x256 (transposed),
 * 5 = 128x128 with the 4 wavefronts along cout, -1 = heuristic */
int vg_debug_set_conv_bf16split_tile(int variant);
/* stride-2 ring kernel of conv_ring.hip: 0 = 256 cout x 128 px, 1 / 2 = 128 x 128 (2x4 / 4x2 wavefronts),
 * 3 = 128 x 256 (transposed), -1 = heuristic */
int vg_debug_set_conv_ring_tile(int variant);
#endif

#ifdef __cplusplus
}
#endif
#endif /* VAEGAN_HIP_H */
