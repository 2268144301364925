/* Training-loop helpers of libhdrnet_amd.so that are not part of the bilateral-grid operator boundary
 * (include/hdrnet_amd.h): the optimizer update of the reference's training loop.
 *
 * hdrnet/bin/train.py:108-115 minimises the l2 loss with tf.train.AdamOptimizer; one update of the whole model
 * (~482 k parameters) is 2 MB of state.  As a multi-tensor launch over 35 separate tensors it takes ~40 us of a
 * 0.7-ms training step on MI355X (few, long-running workgroups); over ONE flat buffer it is a 2-us kernel.
 *
 * hdrnet_adam_step_f32: Adam (Kingma & Ba; the update of torch.optim.Adam without amsgrad / weight decay) on flat
 * fp32 buffers of n elements, in place:
 *   t = step[0] + 1;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2
 *   param -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
 * hdrnet_adam_step_tf_f32: the same with TensorFlow's placement of epsilon -- tf.train.AdamOptimizer, the optimizer the
 * reference constructs (hdrnet/bin/train.py:113), applies "epsilon hat" (tensorflow/python/training/adam.py):
 *   param -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps)
 * i.e. the formula above with eps / sqrt(1 - b2^t) in place of eps.  The two agree to rounding once sqrt(v) >> eps.
 * `step` is a DEVICE float holding the number of updates done so far; the call increments it (a second, one-thread
 * launch), so a captured hipGraph replays correctly.  Returns 0, or 1 for a bad argument (null / misaligned buffer,
 * n <= 0); no host synchronisation. */
#ifndef HDRNET_AMD_TRAIN_H_
#define HDRNET_AMD_TRAIN_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The l2 loss of hdrnet/metrics.py:21-24 WITH its unit gradient in one pass over the batch (the training step's form of
 * hdrnet_l2_loss_f32 + hdrnet_l2_loss_grad_f32 of include/hdrnet_amd.h, which read prediction and target twice):
 *   loss[0] = mean((target - prediction)^2);   dprediction_unit = (2 / n) * (prediction - target)
 * `workspace`: hdrnet_l2_loss_workspace_bytes(n) bytes.  hdrnet_l2_loss_grad_scale_f32 then turns the unit gradient into
 * the gradient, dprediction *= grad_output[0] (a DEVICE scalar), and does nothing but read that scalar when it is 1 --
 * the loss as the root of the backward pass.  Tensors 16-byte aligned; 0 on success, 1 for a bad argument. */
int hdrnet_l2_loss_with_grad_f32(const float* prediction, const float* target, long long n, float* loss,
                                 float* dprediction_unit, void* workspace, size_t workspace_bytes, void* stream);
int hdrnet_l2_loss_grad_scale_f32(float* dprediction, const float* grad_output, long long n, void* stream);

/* The pyramid model's up-add and its VJP (hdrnet/models.py:283-287: `current = tf.image.resize_images(current, sz, BILINEAR,
 * align_corners=True) + out_lvl`), NHWC fp32, the resize of hdrnet_resize_bilinear_f32 (include/hdrnet_amd.h):
 *   hdrnet_resize_add_f32            output[B, OH, OW, C] = resize(coarse[B, IH, IW, C]) + fine[B, OH, OW, C], one pass
 *   hdrnet_resize_bilinear_grad_f32  dinput[B, IH, IW, C] = the transpose of the resize applied to doutput[B, OH, OW, C], as a
 *                                    gather in a fixed order (no atomics: bit-reproducible)
 * (the gradient of the up-add with respect to `fine` is doutput itself).  0 on success, 1 for a bad argument. */
int hdrnet_resize_add_f32(const float* coarse, const float* fine, float* output, int batch, int in_height, int in_width,
                          int out_height, int out_width, int channels, void* stream);
int hdrnet_resize_bilinear_grad_f32(const float* doutput, float* dinput, int batch, int in_height, int in_width,
                                    int out_height, int out_width, int channels, void* stream);

int hdrnet_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                         float* step, float lr, float beta1, float beta2, float eps, void* stream);
int hdrnet_adam_step_tf_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                            float* step, float lr, float beta1, float beta2, float eps, void* stream);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* HDRNET_AMD_TRAIN_H_ */
