/* sgr_loss.h -- C ABI of the image-space colour loss that consumes the rasterizer's RGB output.
 *
 * train.py:100-104 of the reference:
 *     Ll1  = l1_loss(image, gt_image, mask)                                   lib/utils/loss_utils.py:21-37
 *     loss = (1 - lambda_dssim) * lambda_l1 * Ll1 + lambda_dssim * (1 - ssim(image, gt_image, mask=mask))   :80-125
 * ssim() runs five depthwise 11x11 Gaussian convolutions (zero padding) plus a dozen elementwise ops through
 * MIOpen / autograd; its backward repeats them.  Here each of the two losses is one forward and one backward kernel.
 *
 * Images are planar float32 [C,H,W] DEVICE arrays; mask is [H,W] bytes (0/1) or NULL; results are device scalars.
 * Reductions are two-stage with a fixed order (bit-reproducible). */
#ifndef SGR_LOSS_H
#define SGR_LOSS_H
#include <stddef.h>
#include <stdint.h>
#include "sgr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* workspace (floats) the SSIM forward needs for its per-block partial sums */
size_t sgr_ssim_workspace_floats(int C, int H, int W);

/* ssim(img1, img2, window_size = 11, size_average = True, mask)  (loss_utils.py:80-125).
 * out_ssim[0] = mean of the SSIM map over all C*H*W positions (masked pixels of both images are zeroed first, :92-94).
 * partials: NULL, or 3*C*H*W floats receiving dM/dmu1, dM/dsigma1^2, dM/dsigma12 per position for the backward. */
int sgr_ssim_forward(int C, int H, int W, const float* img1, const float* img2, const uint8_t* mask, float* out_ssim,
                     float* partials, float* workspace, void* stream);
/* dL/dimg1 [C,H,W] = upstream[0] / (C*H*W) * d(sum of the SSIM map)/dimg1  (zero where the mask is 0). */
int sgr_ssim_backward(int C, int H, int W, const float* img1, const float* img2, const uint8_t* mask,
                      const float* partials, const float* upstream, float* dL_dimg1, void* stream);

/* workspace (floats) of the L1 forward */
size_t sgr_l1_workspace_floats(int C, int H, int W);
/* l1_loss(network_output, gt, mask)  (loss_utils.py:21-37): mean of |a - b| over the masked pixels' C values.
 * out[0] = loss, out[1] = number of values averaged (C * masked pixels).  An empty mask gives NaN like torch's mean. */
int sgr_l1_forward(int C, int H, int W, const float* a, const float* b, const uint8_t* mask, float* out, float* workspace,
                   void* stream);
/* dL/da = upstream[0] * sign(a - b) / out[1] on masked pixels, 0 elsewhere. */
int sgr_l1_backward(int C, int H, int W, const float* a, const float* b, const uint8_t* mask, const float* out,
                    const float* upstream, float* dL_da, void* stream);

/* The whole colour loss of train.py:100-104 in ONE backward kernel -- the producer of the rasterizer's dL/dout_color:
 *   dL/dimg1 = upstream[0] * ( w_l1 * d l1_loss/dimg1  +  w_ssim * d ssim/dimg1 ),
 * w_l1 = (1 - lambda_dssim) * lambda_l1, w_ssim = -lambda_dssim (the loss holds 1 - ssim).  partials / l1_out are what
 * sgr_ssim_forward / sgr_l1_forward left behind; no intermediate gradient image is materialised. */
int sgr_color_loss_backward(int C, int H, int W, const float* img1, const float* img2, const uint8_t* mask,
                            const float* partials, const float* l1_out, float w_l1, float w_ssim, const float* upstream,
                            float* dL_dimg1, void* stream);

/* ---- the accumulation and depth terms of train.py:106-133 ---------------------------------------------------------
 * Binary-cross-entropy style terms on the accumulated opacity acc [H*W] (clamped to [1e-6, 1-1e-6] first):
 *   SGR_BCE_SKY     where(mask, -log(1 - acc), -log(acc)).mean()                                   train.py:107-109
 *   SGR_BCE_OBJECT  where(mask, -(acc log acc + (1-acc) log(1-acc)), -log(1 - acc)).mean()         train.py:118-121
 * out[0] = the mean.  workspace: sgr_l1_workspace_floats floats. */
#define SGR_BCE_SKY 0
#define SGR_BCE_OBJECT 1
int sgr_bce_forward(int n, int mode, const float* acc, const uint8_t* mask, float* out, float* workspace, void* stream);
/* dL/dacc = upstream[0] / n * d(term)/dacc, zero where the clamp is active (torch.clamp's backward). */
int sgr_bce_backward(int n, int mode, const float* acc, const uint8_t* mask, const float* upstream, float* dL_dacc,
                     void* stream);

/* LiDAR depth term (train.py:124-131): over the pixels with lidar_depth > 0 and mask, the error
 * |depth / (acc + 1e-10) - lidar_depth|; the mean of its int(keep * count) SMALLEST values (keep is a double, like the
 * Python float of train.py:128: 0.95f * 100 truncates to 94, 0.95 * 100 to 95) (keep = 0.95: the largest 5 %
 * are dropped).  The k-th smallest error is found with a 4-pass radix select on the float bits (no sort, no host
 * sync).  out[0] = loss, out[1] = k, out[2] = threshold error, out[3] = weight of the errors equal to the threshold
 * (their share of the remaining slots).  work: sgr_lidar_work_bytes(n) bytes (keeps the per-pixel errors for the backward). */
size_t sgr_lidar_work_bytes(int n);
int sgr_lidar_depth_forward(int n, const float* depth, const float* acc, const float* lidar_depth, const uint8_t* mask,
                            double keep, float* out, char* work, void* stream);
int sgr_lidar_depth_backward(int n, const float* depth, const float* acc, const float* lidar_depth, const float* out,
                             const char* work, const float* upstream, float* dL_ddepth, float* dL_dacc, void* stream);

#ifdef __cplusplus
}
#endif
#endif
