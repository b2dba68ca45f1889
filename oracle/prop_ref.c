/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the JSPSR propagation step.
 *
 * Restates PostProcessor.forward (/root/reference/models/components/spn.py:99-118) on top of
 * the documented semantics of torchvision 0.16 `deform_conv2d` (third-party, not vendored;
 * SURVEY.md section 8c): C_in = C_out = 1, 3x3 window, pad 1, stride 1, one offset group,
 * modulated.  Pinned by tests/golden/g1_postprocessor.npz (made by the reference's own
 * PostProcessor, oracle/gen_golden.py).  Never linked into the product library.
 *
 * Layouts (contiguous, row-major): dem [B][H][W], weight [B][9][H][W], offset [B][18][H][W]
 * (channel 2k = dy, 2k+1 = dx of tap k, k row-major over the window), wk[9], out [B][H][W].
 */
#include <math.h>
#include <stddef.h>
#include <string.h>

#define DEFINE_PROP(NAME, T, FLOOR)                                                              \
  static inline T NAME##_at(const T* im, long H, long W, long y, long x) {                       \
    return (y >= 0 && y < H && x >= 0 && x < W) ? im[y * W + x] : (T)0;                          \
  }                                                                                              \
  /* forward: out = b0 + sum_k wk[k]*(a_k - mean a)*S_k + scale*dem   (spn.py:100-117) */        \
  void NAME##_forward(const T* dem, const T* weight, const T* offset, const T* wk, T b0,         \
                      T scale, T* out, long B, long H, long W) {                                 \
    const long P = H * W;                                                                        \
    for (long b = 0; b < B; ++b)                                                                 \
      for (long y = 0; y < H; ++y)                                                               \
        for (long x = 0; x < W; ++x) {                                                           \
          const T* im = dem + b * P;                                                             \
          const T* a = weight + b * 9 * P + y * W + x;                                           \
          const T* o = offset + b * 18 * P + y * W + x;                                          \
          T mean = 0;                                                                            \
          for (int k = 0; k < 9; ++k) mean += a[k * P];                                          \
          mean /= (T)9;                                                                          \
          T acc = b0;                                                                            \
          for (int k = 0; k < 9; ++k) {                                                          \
            T py = (T)(y - 1 + k / 3) + o[(2 * k) * P];                                          \
            T px = (T)(x - 1 + k % 3) + o[(2 * k + 1) * P];                                      \
            T S = 0;                                                                             \
            if (py > (T)-1 && py < (T)H && px > (T)-1 && px < (T)W) {                            \
              T fy = FLOOR(py), fx = FLOOR(px);                                                  \
              long y0 = (long)fy, x0 = (long)fx;                                                 \
              T ly = py - fy, lx = px - fx, hy = (T)1 - ly, hx = (T)1 - lx;                      \
              S = hy * hx * NAME##_at(im, H, W, y0, x0) + hy * lx * NAME##_at(im, H, W, y0, x0 + 1) + \
                  ly * hx * NAME##_at(im, H, W, y0 + 1, x0) + ly * lx * NAME##_at(im, H, W, y0 + 1, x0 + 1); \
            }                                                                                    \
            acc += wk[k] * (a[k * P] - mean) * S;                                                \
          }                                                                                      \
          out[b * P + y * W + x] = acc + scale * im[y * W + x];                                  \
        }                                                                                        \
  }                                                                                              \
  /* backward: grad_weight [B][9][H][W], grad_offset [B][18][H][W],  */   \
  /* grad_wk[9], grad_b0[1]; coordinate derivative = torchvision get_coordinate_weight      */   \
  /* (per-corner validity, no inside gate).                                                  */  \
  void NAME##_backward(const T* gout, const T* dem, const T* weight, const T* offset,            \
                       const T* wk, T* gweight, T* goffset, double* gwk, double* gb0, long B,    \
                       long H, long W) {                                                         \
    const long P = H * W;                                                                        \
    for (int k = 0; k < 9; ++k) gwk[k] = 0;                                                      \
    gb0[0] = 0;                                                                                  \
    for (long b = 0; b < B; ++b)                                                                 \
      for (long y = 0; y < H; ++y)                                                               \
        for (long x = 0; x < W; ++x) {                                                           \
          const T* im = dem + b * P;                                                             \
          const long pix = y * W + x;                                                            \
          const T* a = weight + b * 9 * P + pix;                                                 \
          const T* o = offset + b * 18 * P + pix;                                                \
          const T g = gout[b * P + pix];                                                         \
          T mean = 0;                                                                            \
          for (int k = 0; k < 9; ++k) mean += a[k * P];                                          \
          mean /= (T)9;                                                                          \
          T gm[9], gsum = 0;                                                                     \
          for (int k = 0; k < 9; ++k) {                                                          \
            T py = (T)(y - 1 + k / 3) + o[(2 * k) * P];                                          \
            T px = (T)(x - 1 + k % 3) + o[(2 * k + 1) * P];                                      \
            T fy = FLOOR(py), fx = FLOOR(px);                                                    \
            long y0 = (long)fy, x0 = (long)fx;                                                   \
            if (!(fy > (T)-1e9 && fy < (T)1e9)) { y0 = -10; fy = py; }                           \
            if (!(fx > (T)-1e9 && fx < (T)1e9)) { x0 = -10; fx = px; }                           \
            T ly = py - fy, lx = px - fx, hy = (T)1 - ly, hx = (T)1 - lx;                        \
            T v00 = NAME##_at(im, H, W, y0, x0), v01 = NAME##_at(im, H, W, y0, x0 + 1);          \
            T v10 = NAME##_at(im, H, W, y0 + 1, x0), v11 = NAME##_at(im, H, W, y0 + 1, x0 + 1);  \
            int inside = (py > (T)-1 && py < (T)H && px > (T)-1 && px < (T)W);                   \
            T S = inside ? hy * hx * v00 + hy * lx * v01 + ly * hx * v10 + ly * lx * v11 : (T)0; \
            T dSdy = hx * (v10 - v00) + lx * (v11 - v01);                                        \
            T dSdx = hy * (v01 - v00) + ly * (v11 - v10);                                        \
            T m = a[k * P] - mean;                                                               \
            T c = g * wk[k] * m;                                                                 \
            goffset[b * 18 * P + (2 * k) * P + pix] = c * dSdy;                                  \
            goffset[b * 18 * P + (2 * k + 1) * P + pix] = c * dSdx;                              \
            gm[k] = g * wk[k] * S;                                                               \
            gsum += gm[k];                                                                       \
            gwk[k] += (double)(g * m * S);                                                       \
          }                                                                                      \
          gsum /= (T)9;                                                                          \
          for (int k = 0; k < 9; ++k) gweight[b * 9 * P + k * P + pix] = gm[k] - gsum;           \
          gb0[0] += (double)g;                                                                   \
        }                                                                                        \
  }

DEFINE_PROP(prop_ref_f64, double, floor)
DEFINE_PROP(prop_ref_f32, float, floorf)
