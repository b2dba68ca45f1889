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
 */
int jspsr_prop_backward_f32(const float* grad_out, const float* dem, const float* weight,
                            const float* offset, int offset_channels, const float* wk,
                            float* grad_weight, float* grad_offset, float* grad_wk,
                            float* grad_b0, void* workspace, int B, int H, int W,
                            jspsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* JSPSR_HIP_H */
