/* jspsr_hip.h -- C ABI of the MI355X (gfx950) JSPSR hot-path library, libjspsr_hip.so.
 *
 * Plain pointers and sizes only; every pointer is DEVICE memory unless stated otherwise.
 * Every launcher is asynchronous on `stream` (a hipStream_t passed as void*), allocates
 * nothing, keeps no pointer after return and is safe to call from any host thread
 * (the reference's backward runs on PyTorch's autograd thread -- SURVEY.md section 8b).
 * Return value: 0 on success, otherwise a negative JSPSR_E* code or a positive hipError_t;
 * jspsr_last_error() gives the text for the calling thread.
 *
 * The reference (xandercai/JSPSR) is pure Python and has no FFI of its own; the seam these
 * entry points replace is the operator call inside its nn.Modules (cited per function,
 * paths relative to the reference tree).  INTEGRATION.md shows the ctypes binding.
 */
#ifndef JSPSR_HIP_H
#define JSPSR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JSPSR_OK 0
#define JSPSR_EINVAL (-1)   /* bad shape / null pointer / unsupported combination */
#define JSPSR_EALIGN (-2)   /* pointer not aligned as documented */

typedef void* jspsr_stream_t; /* hipStream_t */

/* ABI version of this header (bumped on any signature change). */
int jspsr_abi_version(void);
/* Diagnostics: launches so far (this process) of the kernel family named `what` -- the names the error texts use:
 * "conv64_resident" (K2r), "conv128_resident" (K2q), "conv_patch", "conv_patch_16x16", "conv_igemm", "conv2d_wgrad_patch", "prop_forward (dma)",
 * "prop_backward (dma)", "prop_forward", "prop_head_forward", ...  Lets a parity test assert that a shape really took
 * the kernel it is meant to exercise.  -1 for a NULL name, 0 for a name never launched. */
long long jspsr_launch_count(const char* what);
/* The persistent register-resident conv kernels (K2r: "conv64_resident", K2q: "conv128_resident") hand their tiles out either
 * by a static stride walk (default alone on the GPU: neighbouring tiles share an XCD's L2) or from a global ticket in runs of
 * four (on = 1).  One workgroup of these kernels needs a WHOLE compute unit; where another kernel holds some CUs for long --
 * RCCL's all-reduce beside the backward pass of a data-parallel step -- the workgroups that start late would, with the static
 * walk, still do their full share after everybody else has finished; drawing from the ticket they find it empty and leave.
 * on = -1: back to the environment's default (JSPSR_CONV_DYNQ / JSPSR_CONV_DYNQ128).  Returns the previous setting.
 * jspsr_amd.ddp.GradReducer switches it on for world sizes > 1 -- a precaution (no multi-GPU box to measure it on; on one GPU
 * the two walks time the same inside the step).  (ABI v15) */
int jspsr_conv_dynamic_queue(int on);
/* Text of the last error raised on the calling thread ("" if none). */
const char* jspsr_last_error(void);

/* ---- K1: fused spatial propagation ------------------------------------------------------
 * Replaces PostProcessor.forward, models/components/spn.py:99-118, i.e. the sequence
 *   weight - mean_k(weight)                                   (spn.py:100-101)
 *   torchvision.ops.deform_conv2d(dem, offset, w, b, pad 1, mask=weight)   (spn.py:105-114)
 *   + scale * dem                                             (spn.py:116-117)
 * and the identical Post_process_deconv.forward, models/LRRU.py:267-298.
 *
 * dem    [B][H][W]      fp32   (B,1,H,W contiguous)
 * weight [B][9][H][W]   fp32   affinities after the sigmoid, tap k row-major over the 3x3 window
 * offset [B][OC][H][W]  fp32   OC = 18: channel 2k = dy_k, 2k+1 = dx_k (torchvision layout);
 *                              OC = 16: the 8 learned taps only (k = 0..3, 5..8), centre tap
 *                              implicitly (0,0) -- what Generator emits before spn.py:70-73
 * wk     [9], b0 [1]    fp32   PostProcessor.w / .b (device pointers: no host sync)
 * out    [B][H][W]      fp32
 * All pointers 4-byte aligned; 16-byte aligned pointers with W % 4 == 0 take the fast path.
 */
int jspsr_prop_forward_f32(const float* dem, const float* weight, const float* offset,
                           int offset_channels, const float* wk, const float* b0, float scale,
                           float* out, int B, int H, int W, jspsr_stream_t stream);

/* Bytes of scratch jspsr_prop_backward_f32 needs for a (B,H,W) problem. */
size_t jspsr_prop_backward_workspace_bytes(int B, int H, int W);

/* Backward of the above (the autograd of spn.py:99-118).
 * grad_out [B][H][W]; grad_weight [B][9][H][W]; grad_offset [B][OC][H][W];
 * grad_wk [9], grad_b0 [1] are overwritten (not accumulated).  grad wrt dem is not produced:
 * every caller detaches it (models/JSPSR.py:372, models/LRRU.py:453,467,481,496).
 * workspace: jspsr_prop_backward_workspace_bytes() bytes, 16-byte aligned.
 * Two launches: the streaming kernel (writes grad_weight / grad_offset and one row of 10 partial sums per
 * workgroup into the workspace) and a 10-workgroup fold of those rows into grad_wk / grad_b0.  With
 * grad_wk == grad_b0 == NULL only the first is launched; jspsr_prop_backward_fold_f32 is the second on its own.
 */
int jspsr_prop_backward_f32(const float* grad_out, const float* dem, const float* weight,
                            const float* offset, int offset_channels, const float* wk,
                            float* grad_weight, float* grad_offset, float* grad_wk,
                            float* grad_b0, void* workspace, int B, int H, int W,
                            jspsr_stream_t stream);

int jspsr_prop_backward_fold_f32(const void* workspace, int B, int H, int W, float* grad_wk, float* grad_b0,
                                 jspsr_stream_t stream);

/* ---- K1s: one propagation step in its general form, for chains of steps ------------------------------------------
 * Replaces NLSPN._propagate_once, models/components/nlspn.py:177-187, called prop_time times on its own output
 * (nlspn.py:226-233) with affinities normalised once outside the loop (nlspn.py:158-173) -- and, with normalize != 0,
 * is the same operator as jspsr_prop_forward_f32 / jspsr_prop_backward_f32 plus the gradient they do not produce.
 *   normalize == 0:  out = b0 + sum_k wk[k] weight_k S_k + scale * dem      (affinities taken as they are)
 *   normalize != 0:  out = b0 + sum_k wk[k] (weight_k - mean_k weight) S_k + scale * dem      (spn.py:99-118)
 * Operand layout as for jspsr_prop_forward_f32.  out must not alias dem.
 * Backward: grad_weight / grad_offset are overwritten, or ADDED to when accumulate != 0 (the affinities and offsets
 * of an N-step chain are shared by all steps: their gradients sum over the steps).  grad_dem (may be NULL): the
 * gradient with respect to dem is ADDED into it (bilinear scatter + scale * grad_out; the caller zero-fills it or
 * lets it carry another contribution); it uses float atomics, so its last bits depend on the execution order.
 * grad_wk / grad_b0: both NULL (partial rows stay in the workspace) or both valid (overwritten).
 * workspace: jspsr_prop_step_backward_workspace_bytes() bytes, 16-byte aligned. */
int jspsr_prop_step_forward_f32(const float* dem, const float* weight, const float* offset, int offset_channels,
                                const float* wk, const float* b0, float scale, int normalize, float* out,
                                int B, int H, int W, jspsr_stream_t stream);
size_t jspsr_prop_step_backward_workspace_bytes(int B, int H, int W);
int jspsr_prop_step_backward_f32(const float* grad_out, const float* dem, const float* weight, const float* offset,
                                 int offset_channels, const float* wk, float scale, int normalize, int accumulate,
                                 float* grad_weight, float* grad_offset, float* grad_dem, float* grad_wk,
                                 float* grad_b0, void* workspace, int B, int H, int W, jspsr_stream_t stream);

/* ---- K1h: the same propagation step fed straight from the generator head's NHWC output ----------------------
 * Inside the models the two 1x1 heads of Generator.forward (models/components/spn.py:41-52,66-68; LRRU.py:238-247) run
 * as ONE 32-channel convolution; these entry points read its output where it lies and fold in what the reference does
 * between the heads and deform_conv2d: the Sigmoid of conv_weight (spn.py:43), the zero centre offset (spn.py:69-73),
 * the mean subtraction (spn.py:100-101) and the residual (spn.py:116-117).  The public planar entries above stay the
 * boundary of PostProcessor.forward.
 *
 * head [B][H][W][32], dtype JSPSR_F32 or JSPSR_BF16 (defined below), 16-byte aligned, channel c = 4 t + j:
 *   t = 0..7 the learned taps in window order without the centre (k = t < 4 ? t : t + 1),
 *   j = 0 affinity LOGIT of tap k (pre-sigmoid), j = 1 dy_k, j = 2 dx_k,
 *   j = 3: the centre tap's affinity logit for t == 0; ignored for t > 0.
 * dem, out [B][H][W] fp32; wk [9], b0 [1] as above.
 * Backward: grad_head in the same layout and dtype = d/d(head) (sigmoid derivative included; channels 4t+3, t > 0,
 * are written as zeros); grad_wk / grad_b0 overwritten, or both NULL to leave the per-workgroup partial rows in the
 * workspace (jspsr_prop_head_backward_workspace_bytes; the fold is the 10-workgroup launch jspsr_prop_backward_f32 also uses). */
int jspsr_prop_head_forward(int dtype, const float* dem, const void* head, const float* wk, const float* b0,
                            float scale, float* out, int B, int H, int W, jspsr_stream_t stream);
size_t jspsr_prop_head_backward_workspace_bytes(int B, int H, int W);
int jspsr_prop_head_backward(int dtype, const float* grad_out, const float* dem, const void* head, const float* wk,
                             void* grad_head, float* grad_wk, float* grad_b0, void* workspace, int B, int H, int W,
                             jspsr_stream_t stream);

/* ---- K8: the steps either side of the model call, on device rasters -----------------------------
 * jspsr_tiles_crop_f32: TileCrop's square cover, data/data_utils.py:87-194 -- x [C][H][W] -> out [n_x*n_x][C][k][k], tile
 *   (r, c) = rows stride*r .., columns stride*c .., row-major over the cover.
 * jspsr_tiles_merge_f32: merge_dem(method = copyto_add) with gen_weight_row / gen_weight_col, utils/utils.py:802-967 --
 *   tiles [n_x*n_x][k][k] predictions, each first losing border_px pixels per side, weighted by the linear ramps over the
 *   p = (k - 2 border_px) - stride pixels neighbours share (ramp [p] = linspace(1, 0, p + 2) without its ends, from the
 *   caller) and summed into out [S][S], S = stride (n_x - 1) + k - 2 border_px.  Every mosaic pixel gathers its <= 4 tiles
 *   in tile order: the same additions in the same order as the reference's tile-by-tile accumulation.
 * jspsr_mirror_pad_f32: add_padding, utils/utils.py:1501-1520, index for index -- x [C][H][W] -> out [C][H+2n][W+2n].
 * jspsr_elev_scale_f32: ToTensor.scale_data (descale = 0; data/data_utils.py:289-312: (z - base - min) / (max - min), or
 *   log(z - base - min) / log(max - min) + 1e-8) and ToDEM.descale_data (descale = 1; data_utils.py:441-457). */
int jspsr_tiles_crop_f32(const float* x, float* out, int C, int H, int W, int k, int stride, int n_x, jspsr_stream_t stream);
int jspsr_tiles_merge_f32(const float* tiles, const float* ramp, float* out, int n_x, int k, int border_px, int stride,
                          jspsr_stream_t stream);
int jspsr_mirror_pad_f32(const float* x, float* out, int C, int H, int W, int n, jspsr_stream_t stream);
int jspsr_elev_scale_f32(const float* in, float* out, long long n, int descale, int elev_log, double elev_min, double elev_max,
                         double base_elev, jspsr_stream_t stream);

/* ---- K1 in the models (round 4): logits + offsets as PLANES of one tensor ---------------------------------------
 * The same operator as jspsr_prop_forward_f32 / jspsr_prop_backward_f32 -- same kernels, same 108 / 208 algorithmic bytes
 * per pixel -- with what the reference does between the generator's heads and deform_conv2d folded in: the Sigmoid of
 * conv_weight (models/components/spn.py:43), the zero centre offset (spn.py:69-73), the mean subtraction
 * (spn.py:100-101), the residual (spn.py:116-117).
 *   head [B][25][H][W] fp32: planes 0..8 affinity LOGITS of tap k (row-major over the 3x3 window), planes 9..24 the sixteen
 *   learned offsets in Generator's order (dy, dx of taps 0..3, 5..8) -- what jspsr_head_forward writes.
 * Backward: grad_head in the same layout = d/d(head) (sigmoid derivative included); grad_wk / grad_b0 overwritten, or both
 * NULL to leave the partial rows in the workspace (jspsr_prop_backward_workspace_bytes; jspsr_prop_backward_fold_f32).
 * Any W and 4-byte aligned pointers; 16-byte aligned pointers with W % 4 == 0 take the persistent LDS-DMA kernel. */
int jspsr_prop_logits_forward_f32(const float* dem, const float* head, const float* wk, const float* b0, float scale,
                                  float* out, int B, int H, int W, jspsr_stream_t stream);
int jspsr_prop_logits_backward_f32(const float* grad_out, const float* dem, const float* head, const float* wk,
                                   float* grad_head, float* grad_wk, float* grad_b0, void* workspace, int B, int H, int W,
                                   jspsr_stream_t stream);

/* ---- K1c: the generator's two 1x1 heads as one convolution that writes planes -------------------------------------
 * Replaces conv_weight (without its Sigmoid) and conv_offset of Generator.forward, models/components/spn.py:41-52,66-68
 * (and BasicDepthEncoder's heads, models/LRRU.py:238-247), and their autograd.
 *   x      NHWC feature (B,H,W,Cin) of `dtype` (0 fp32 / 1 bf16, as JSPSR_F32 / JSPSR_BF16 below), channel pitch x_cstride,
 *          first channel x_coff (elements; multiples of 4 fp32 / 8 bf16), 16-byte aligned
 *   w25    [25][Cin] fp32: rows 0..8 = conv_weight.0.weight, rows 9..24 = conv_offset.conv.0.weight;  b25 [25] the biases
 *   planes [B][25][H][W] fp32 = the `head` operand of jspsr_prop_logits_forward_f32
 * jspsr_head_ok(dtype, B, H, W, Cin) != 0: H * W a multiple of 32, Cin in {32, 64, 128}.
 * Backward, one pass over grad_planes [B][25][H][W]: grad_x (NHWC, `dtype`, pitch / offset as x; NULL = not wanted),
 * grad_nhwc32 [B][H][W][32] in `dtype` (channels 25..31 zero) = the G operand jspsr_conv2d_wgrad takes for the weight
 * gradient, grad_b25 [25] (overwritten).  workspace: jspsr_head_backward_workspace_bytes(), 16-byte aligned. */
int jspsr_head_ok(int dtype, int B, int H, int W, int Cin);
int jspsr_head_forward(int dtype, const void* x, int x_cstride, int x_coff, int Cin, const float* w25, const float* b25,
                       float* planes, int B, int H, int W, jspsr_stream_t stream);
size_t jspsr_head_backward_workspace_bytes(int B, int H, int W);
int jspsr_head_backward(int dtype, const float* grad_planes, const float* w25, int Cin, void* grad_x, int gx_cstride,
                        int gx_coff, void* grad_nhwc32, float* grad_b25, void* workspace, int B, int H, int W,
                        jspsr_stream_t stream);

/* ---- K2: convolutions on the matrix cores (implicit GEMM, NHWC) ---------------------------
 * Replace the reference's nn.Conv2d / nn.ConvTranspose2d calls and their autograd
 * (models/components/basics.py:6-20,39-47,69-77; every conv of models/JSPSR.py:66-180 and
 * models/components/spn.py:16-52).  Activations are NHWC; a tensor argument is described by
 * (pointer, C, channel pitch, channel offset) so a conv can read or write a channel slice of a
 * wider buffer (the reference's torch.cat fusion, basics.py:134, needs no copy).
 * dtype: JSPSR_F32 -> v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, fp32 storage);
 *        JSPSR_BF16 -> v_mfma_f32_32x32x16_bf16 (bf16 storage, fp32 accumulate).
 * Gathered channel counts / pitches / offsets must be multiples of 4 (fp32) or 8 (bf16).
 */
#define JSPSR_F32 0
#define JSPSR_BF16 1

/* Re-lay a master weight (O, I, KH, KW) fp32 (PyTorch Conv2d layout) for the kernels:
 *   mode 0: packed[o][ky][kx][i]  (i zero-padded to c_pad) -- forward of Conv2d(I->O)
 *   mode 1: packed[i][ky][kx][o]  (o zero-padded to c_pad) -- data gradient of Conv2d(I->O);
 *           also the forward of ConvTranspose2d whose weight is stored (I_T=O, O_T=I, KH, KW).
 */
int jspsr_pack_weight(int dtype, const float* w, void* packed, int O, int I, int KH, int KW,
                      int mode, int c_pad, jspsr_stream_t stream);

/* The same re-lay for MANY weights in one launch (after an optimizer step: 76 conv weights x 2 layouts for the
 * image+mask JSPSR instead of 152 launches).  `descs` is DEVICE memory, n descriptors sorted by `start`; descriptor i
 * is served by workgroups [start, start + ceil(total / jspsr_pack_chunk())) of the launch, total = rows * KH * KW *
 * c_pad elements; total_blocks = the sum of those block counts.  Each descriptor names its own output dtype. */
typedef struct jspsr_pack_desc {
  const float* w;     /* (O, I, KH, KW) fp32 master */
  void* out;          /* packed output */
  long long start;    /* first workgroup of this descriptor = sum of ceil(total / jspsr_pack_chunk()) over the preceding ones */
  long long total;    /* elements of this packed output */
  int O, I, KH, KW;
  int mode;           /* 0 / 1 as for jspsr_pack_weight */
  int c_pad;
  int dtype;          /* JSPSR_F32 / JSPSR_BF16 */
  int reserved;
} jspsr_pack_desc;
int jspsr_pack_chunk(void);
int jspsr_pack_weights_multi(const jspsr_pack_desc* descs, int n, long long total_blocks, jspsr_stream_t stream);

/* out[b,oy,ox,n] = bias[n] + sum_{ky,kx,c} in[b, oy*stride-pad+ky, ox*stride-pad+kx, c] * W[n,ky,kx,c]
 * (+ ReLU if relu != 0).  wpack: mode-0 packing with c_pad = Cin.  bias may be NULL.
 * stats (may be NULL; requires bias == NULL and relu == 0): the epilogue also writes the BatchNorm batch
 * statistics of the result taken from the fp32 accumulators -- jspsr_conv2d_stats_rows(B,OH,OW) partial
 * rows of [sum over the row's pixels | sum of squares] x Cout floats, to be handed to jspsr_bn_forward
 * (ext_partial / ext_rows), which then skips its own pass over the tensor. */
/* Inference epilogue (all optional, NULL = absent): out = [relu](acc * scale[c] + bias[c] + addend), the ReLU always
 * last.  With scale/bias from jspsr_bn_fold and addend = the shortcut tensor this is conv -> BatchNorm(eval)
 * (-> + residual) (-> ReLU) of basics.py:49-53,111-123 in one launch.  Not combinable with `stats`. */
/* Input transform (in_affine, may be NULL): [2][Cin] fp32 (scale | shift), 16-byte aligned.  The gathered tensor is
 * then read as [relu](in * scale[c] + shift[c]) -- applied between the patch registers and LDS, zero padding stays
 * zero -- so a conv can consume the RAW output of the preceding conv plus that layer's BatchNorm affine
 * (jspsr_bn_forward: affine_out) instead of a normalised copy: conv -> BN -> ReLU -> conv of BasicBlock,
 * basics.py:111-117, without the normalised activation ever existing in memory.  Available where
 * jspsr_conv2d_in_affine_ok(dtype, Cin, KH, KW, stride) != 0 (the patch kernel: stride 1, 2..9 taps, Cin a multiple
 * of 32 fp32 / 64 bf16); JSPSR_EINVAL otherwise. */
int jspsr_conv2d_stats_rows(int B, int OH, int OW);
int jspsr_conv2d_in_affine_ok(int dtype, int Cin, int KH, int KW, int stride);
int jspsr_conv2d_forward(int dtype, const void* in, const void* wpack, const float* bias, void* out,
                         int B, int IH, int IW, int Cin, int in_cstride, int in_coff, int Cout,
                         int out_cstride, int out_coff, int KH, int KW, int stride, int pad, int relu,
                         float* stats, const float* scale, const void* addend, int add_cstride,
                         const float* in_affine, int in_relu, jspsr_stream_t stream);

/* gin[b,y,x,c] = bias[c] + sum_{ky,kx,n} gout[b,(y+pad-ky)/stride,(x+pad-kx)/stride,n] * W[n,c,ky,kx]
 * over the taps where the division is exact: the data gradient of the conv above, and equally
 * ConvTranspose2d(stride, pad) forward with gout := its input (IH, IW = its output size).
 * wpack_t: mode-1 packing with c_pad = Cg.  One launch per stride phase (no wasted taps).
 * addend (may be NULL): a tensor on gin's grid, Cin channels at pitch add_cstride, added in the epilogue --
 * gin = [relu](...) + addend.  It carries the gradient that reaches the same tensor along another path (the
 * residual branch of a BasicBlock, basics.py:113-122), which autograd would otherwise add in a separate pass.
 * scale (may be NULL): per-channel factor as in jspsr_conv2d_forward -- ConvTranspose2d -> BatchNorm(eval) -> ReLU
 * of the decoder (basics.py:69-85) in one launch at inference.  gin = [relu](acc * scale + bias + addend).
 * red_x / red_cstride / red_par / red_out (round 4; all NULL / 0: off): the REDUCE pass of the BatchNorm whose output
 * gradient this launch produces (conv -> BN -> ReLU -> conv of BasicBlock, basics.py:111-117: gin is the gradient of the
 * ReLU's output), taken in the epilogue instead of by a pass of its own.  red_x: that BatchNorm's saved input on gin's grid
 * (Cin channels at pitch red_cstride); red_par [4][Cin] from jspsr_bn_reduce_params; red_out: partial rows
 * [jspsr_conv2d_stats_rows(B, IH, IW)][2][Cin] = per 8x16-pixel tile the sums of dz and dz * xhat, dz = the stored gin where
 * the ReLU was open -- handed to jspsr_bn_backward as ext_partial.  Available where
 * jspsr_conv2d_dgrad_reduce_ok(...) != 0 (3x3, stride 1, pad 1 on the patch kernel). */
int jspsr_conv2d_dgrad_reduce_ok(int dtype, int B, int IH, int IW, int Cg, int Cin, int KH, int KW, int stride, int pad);
int jspsr_conv2d_dgrad(int dtype, const void* gout, const void* wpack_t, const float* bias, void* gin,
                       int B, int OH, int OW, int Cg, int g_cstride, int g_coff, int IH, int IW,
                       int Cin, int in_cstride, int in_coff, int KH, int KW, int stride, int pad,
                       int relu, const void* addend, int add_cstride, const float* scale,
                       const void* red_x, int red_cstride, const float* red_par, float* red_out, jspsr_stream_t stream);

/* Weight gradient (autograd of nn.Conv2d / nn.ConvTranspose2d w.r.t. .weight):
 *   dW[r][c][ky][kx] = sum_{b,oy,ox} G[b,oy,ox,r] * X[b, oy*stride-pad+ky, ox*stride-pad+kx, c]
 * G lives on the conv's output grid (B,OH,OW,Cg), X on its input grid (B,IH,IW,Cx); Cg, Cx are the
 * chunk-padded channel counts, R <= Cg and C <= Cx the real ones; dW is fp32 in PyTorch's
 * (R, C, KH, KW) layout and is overwritten (accumulate == 0) or added to (accumulate != 0).
 * Conv2d(I->O): G = grad_out, X = input, R = O, C = I.  ConvTranspose2d(I->O, weight (I,O,KH,KW)):
 * G = its input, X = grad of its output, R = I, C = O.
 * workspace: jspsr_conv2d_wgrad_workspace_bytes() bytes (split-K slabs, summed in a fixed order).
 * x_affine (may be NULL): [2][Cx] (scale | shift) -- X is read as [relu](X * scale + shift), the counterpart of
 * jspsr_conv2d_forward's in_affine for the weight gradient of a conv whose normalised input was never materialised.
 * Available where jspsr_conv2d_wgrad_x_affine_ok(...) != 0 (the nine-tap kernel: 3x3, stride 1, pad 1).
 */
size_t jspsr_conv2d_wgrad_workspace_bytes(int dtype, int B, int OH, int OW, int Cg, int Cx, int KH, int KW);
int jspsr_conv2d_wgrad_x_affine_ok(int dtype, int B, int OH, int OW, int Cg, int Cx, int KH, int KW, int stride, int pad);
int jspsr_conv2d_wgrad(int dtype, const void* G, int Cg, int g_cstride, int g_coff, const void* X, int Cx,
                       int x_cstride, int x_coff, float* dW, int R, int C, int B, int OH, int OW,
                       int IH, int IW, int KH, int KW, int stride, int pad, int accumulate,
                       const float* x_affine, int x_relu, void* workspace, jspsr_stream_t stream);

/* ---- K4/K5: per-channel operators around the convolutions (HBM-bound, NHWC) -----------------
 * Tensors are (pointer, channel pitch, channel offset) slices of NHWC buffers, `npix` pixels,
 * `C` channels (multiple of 4 for fp32 / 8 for bf16); statistics and parameters are fp32.
 * workspace: jspsr_reduce_workspace_bytes(dtype, C, nseg) bytes, 16-byte aligned
 * (nseg = 1, or the batch size for the per-image gate kernels).
 */
size_t jspsr_reduce_workspace_bytes(int dtype, int C, int nseg);

/* nn.BatchNorm2d forward fused with the residual add and ReLU of BasicBlock
 * (models/components/basics.py:49-53,81-85,111-123):
 *   y = [relu]( bn(x) * res_scale + res )      (res may be NULL)
 * training != 0: batch statistics (biased var), running stats updated in place with `momentum`
 * (unbiased var), save_mean / save_invstd [C] written for the backward.  training == 0: running
 * stats are used (and copied to save_*).  ext_partial / ext_rows: statistics already accumulated by
 * jspsr_conv2d_forward (NULL / 0: this call makes its own pass over x).
 * The shortcut of a projecting BasicBlock is itself conv1x1 -> BatchNorm (basics.py:118-119): instead of normalising
 * it in a pass of its own, that BatchNorm is called with y == NULL and affine_out [2][C] (statistics, running stats
 * and its per-channel scale | shift only), and the block's second BatchNorm takes the RAW conv1x1 output as `res`
 * together with res_affine = that [2][C]:  y = [relu]( bn(x) * res_scale + (res * res_affine[0] + res_affine[1]) ). */
int jspsr_bn_forward(int dtype, const void* x, int x_cs, int x_coff, const void* res, int r_cs, int r_coff,
                     void* y, int y_cs, int y_coff, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, int training,
                     int relu, float res_scale, float* save_mean, float* save_invstd, long long npix, int C,
                     const float* ext_partial, int ext_rows, const float* res_affine, float* affine_out,
                     void* workspace, jspsr_stream_t stream);

/* BatchNorm in eval mode as a per-channel affine: scale = gamma / sqrt(var + eps) * res_scale,
 * shift = (beta - mean * gamma / sqrt(var + eps)) * res_scale. */
int jspsr_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                  float eps, float res_scale, int C, float* scale, float* shift, jspsr_stream_t stream);

/* Backward of the above.  dy is the gradient w.r.t. y.  relu: 0 = none; 1 = mask from the saved output
 * (y > 0); 2 = mask recomputed from x as gamma*xhat + beta > 0 (valid without a residual; y is not read).
 * dx (dense, pitch C) = grad w.r.t. x; dres (dense, may be NULL) = grad w.r.t. res;
 * dgamma, dbeta [C]: overwritten, or added to when accumulate != 0 (gradients landing directly in a
 * caller-owned accumulation buffer).
 * ext_partial / ext_rows (NULL / 0: this call makes its own reduce pass): the per-tile sums of dz and dz * xhat, rows of
 * [2][C], that the producing data gradient's epilogue wrote (jspsr_conv2d_dgrad: red_out) -- relu mode 2 only. */
int jspsr_bn_reduce_params(const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, int C,
                           float* par, jspsr_stream_t stream);
int jspsr_bn_backward(int dtype, const void* dy, int dy_cs, int dy_coff, const void* y, int y_cs, int y_coff,
                      const void* x, int x_cs, int x_coff, const float* gamma, const float* beta, const float* save_mean,
                      const float* save_invstd, int training, int relu, float res_scale, void* dx, void* dres,
                      float* dgamma, float* dbeta, int accumulate, long long npix, int C, void* workspace,
                      const float* ext_partial, int ext_rows, jspsr_stream_t stream);

/* Backward of the conv epilogue `y = [relu](conv + bias)` of the BN-free Basic2d (basics.py:36-53):
 * dz = dy * [y > 0] (y read with channel pitch y_cs; dz written with pitch dz_cs if dz != NULL),
 * dbias[c] = sum dz (if dbias != NULL). */
int jspsr_act_backward(int dtype, const void* dy, int dy_cs, int dy_coff, const void* y, int y_cs, int relu, void* dz,
                       int dz_cs, float* dbias, long long npix, int C, void* workspace, jspsr_stream_t stream);

/* ChannelAttention (models/components/resnet_cbam.py:36-53; applied in basics.py:57-58).
 * gate_pool: avg[b,c], mx[b,c] over the npix pixels of image b, amax = smallest pixel index of the
 * max.  gate_scale: y = x * s[b,c].  Backward: ds[b,c] = sum_p dy*x (reduce), then
 * dx = dy*s + davg/npix + [p == amax] dmax (apply).  The C -> C/16 -> C MLP between pool and
 * scale works on B x C vectors: jspsr_gate_mlp_* below. */
int jspsr_gate_pool(int dtype, const void* x, int B, long long npix, int C, float* avg, float* mx, int* amax,
                    void* workspace, jspsr_stream_t stream);
int jspsr_gate_scale(int dtype, const void* x, const float* s, void* y, int B, long long npix, int C,
                     jspsr_stream_t stream);
int jspsr_gate_backward_reduce(int dtype, const void* dy, const void* x, float* ds, int B, long long npix, int C,
                               void* workspace, jspsr_stream_t stream);
int jspsr_gate_backward_apply(int dtype, const void* dy, const float* s, const float* davg, const float* dmax,
                              const int* amax, void* dx, int B, long long npix, int C, jspsr_stream_t stream);

/* ---- the two ends of the training step ----------------------------------------------------
 * Fused loss of the reference configs (MultiLoss, losses/loss_schemes.py:55-72; weights from
 * configs/<name>.yml:67-70): losses[4] = {L1, L2, Grad, Total = w1*L1 + w2*L2 + wg*Grad} on fp32
 * (B,1,H,W) tensors; Grad = L1 between normalised Sobel gradients with replicate padding
 * (losses/loss_functions.py:171-185).  The backward writes d(Total)/d(pred) * grad_total[0]
 * (grad_total may be NULL = 1).  workspace: jspsr_loss_workspace_bytes(), shared by both calls.
 */
size_t jspsr_loss_workspace_bytes(int B, int H, int W);
int jspsr_loss_forward(const float* pred, const float* gt, float w1, float w2, float wg, float* losses,
                       void* workspace, int B, int H, int W, jspsr_stream_t stream);
int jspsr_loss_backward(const float* pred, const float* gt, const float* grad_total, float w1, float w2,
                        float wg, float* grad_pred, const void* workspace, int B, int H, int W,
                        jspsr_stream_t stream);

/* Evaluation scores of one tile on the device (evaluation/metrics.py: MeterBase._prepare :147-199, MeterPSNR :229-235,
 * MeterRMSE :372-384, MeterMedian :453, MeterNMAD :508-510, MeterLE95 :565-568; ToDEM.descale_data,
 * data/data_utils.py:441-457).  pred, gt: fp32 [H][W] in the network's [0,1] range (the reference evaluates one tile
 * at a time).  border: fraction cropped on every side (int(H * border) rows, int(W * border) columns); the prediction is
 * clamped to [0,1]; both are de-scaled to metres with (value_min, value_max, elev_log).
 * scores[5] (device) = {PSNR on the [0,1] tensors, RMSE, median, NMAD, LE95 of the elevation differences}.  The order
 * statistics are exact (radix select; torch.median's lower-middle convention, kthvalue with k = 1 + round(0.95 (n-1))).
 * No host synchronisation.  workspace: jspsr_metrics_workspace_bytes(H, W) bytes, 16-byte aligned. */
size_t jspsr_metrics_workspace_bytes(int H, int W);
int jspsr_metrics_forward(const float* pred, const float* gt, int H, int W, float border, float value_min,
                          float value_max, int elev_log, float* scores, void* workspace, jspsr_stream_t stream);

/* One AdamW step (torch.optim.AdamW semantics: decoupled weight decay, bias correction) over a flat
 * fp32 parameter / gradient / moment buffer of n elements (utils/common_config.py:241-291).  The four pointers are
 * 4-byte aligned and share one offset from a 16-byte boundary (sub-ranges of four identically laid out buffers). */
int jspsr_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int step, jspsr_stream_t stream);
/* The same step with its scalars in DEVICE memory, for launches captured in a hipGraph (jspsr_amd/graph.py): hyper[7] =
 * { lr, beta1, beta2, eps, weight_decay, 1 - beta1^step, sqrt(1 - beta2^step) } (the bias corrections computed by the caller
 * in double, as torch.optim.AdamW does), refreshed by the caller before every replay. */
int jspsr_adamw_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                         const float* hyper, jspsr_stream_t stream);

/* The MLP between gate_pool and gate_scale (resnet_cbam.py:41-53: two bias-free 1x1 convs C -> Ch -> C shared by the
 * average- and the max-pooled vector, ReLU between, Sigmoid of the sum): s[b,c] = sigmoid(W2 relu(W1 avg[b]) + W2 relu(W1
 * mx[b])).  w1 (Ch, C), w2 (C, Ch) fp32 row-major; hid (B, 2, Ch) receives the two hidden vectors for the backward pass.
 * Backward: from ds (B, C) the gradients of avg, mx (B, C) and of the two weights (summed over the batch in image order);
 * workspace of jspsr_gate_mlp_backward_workspace_bytes.  Needs (C + 2 Ch) * 4 bytes <= 64 KiB. */
int jspsr_gate_mlp_forward(const float* avg, const float* mx, const float* w1, const float* w2, int B, int C, int Ch,
                           float* s, float* hid, jspsr_stream_t stream);
size_t jspsr_gate_mlp_backward_workspace_bytes(int B, int C, int Ch);
int jspsr_gate_mlp_backward(const float* ds, const float* s, const float* hid, const float* avg, const float* mx,
                            const float* w1, const float* w2, int B, int C, int Ch, float* davg, float* dmax, float* dw1,
                            float* dw2, void* workspace, jspsr_stream_t stream);

/* The models' first step with every input (the reference hands them contiguous planar fp32 tensors, utils/utils.py:156-179;
 * models/JSPSR.py:208-222 then feeds the stems): src (B,C,H,W) fp32 -> dst (B,H,W,c_pad) in `dtype`, channels last and
 * zero-padded to c_pad (a multiple of 4 fp32 / 8 bf16 channels, >= C).  One pass; dst 16-byte aligned. */
int jspsr_nchw_to_nhwc(int dtype, const float* src, void* dst, int B, int C, int H, int W, int c_pad, jspsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* JSPSR_HIP_H */
