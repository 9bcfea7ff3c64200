/*
 * crossloc_loss.h — C ABI of the fused per-pixel regression losses (libcrossloc_hip.so).
 *
 * Replaces the PyTorch-eager bodies of
 *   scene_coords_regression_loss   /root/reference/loss/coord.py:87-188
 *   depth_regression_loss          /root/reference/loss/depth.py:7-76
 *   normal_regression_loss         /root/reference/loss/normal.py:8-127
 *   semantics_classification_loss  /root/reference/loss/semantics.py:44-91 (with CrossEntropyLoss2d :10-18)
 * with one streaming kernel each that produces the loss terms AND the analytic gradients w.r.t. the
 * prediction and the uncertainty map in the same pass (what autograd derived in the reference), plus a
 * fixed-order finalisation.  No host synchronisation (the reference syncs 5x per call, coord.py:132,170-175).
 *
 * Tensors are float32 device pointers, contiguous NCHW as the network emits them:
 *   pred [B,C,Ho,Wo], unc [B,1,Ho,Wo] (positive channel; NULL allowed when mode == 0), labels [B,C,Ho,Wo]
 *   with `nodata` marking cells without ground truth (-1 in CrossLoc, utils/learning.py:38-46).
 * mode: 0 = `uncertainty is None`, 1 = `uncertainty == 'MLE'`.
 * per_image_scale: 0 -> gradients scaled for reduction='mean' (1/(B*N)); 1 -> reduction=None (1/N per image).
 * out (device, 2+B floats): [0] mean loss, [1] valid prediction rate, [2+b] per-image mean loss.
 * workspace: xl_loss_workspace_doubles(B,Ho,Wo) doubles of device scratch.  dpred/dunc may be NULL.
 * All calls are asynchronous on `stream` and return 0 or a negative xl status (crossloc_dsac.h).
 */
#ifndef CROSSLOC_LOSS_H
#define CROSSLOC_LOSS_H

#ifdef __cplusplus
extern "C" {
#endif

int xl_loss_workspace_doubles(int B, int Ho, int Wo);

/* focal/cx/cy: get_cam_mat (coord.py:7-17); subsample: get_pixel_grid cell size (utils/learning.py:20-35);
 * min_depth, soft_clamp, hard_clamp, init_tolerance: train_single_task.py:93-107 defaults 0.1/100/1000/50. */
int xl_loss_coord(const float *pred, const float *unc, const float *gt_poses, const float *gt_coords,
                  int B, int Ho, int Wo, float focal, float cx, float cy, float subsample,
                  float min_depth, float soft_clamp, float hard_clamp, float init_tolerance, float nodata,
                  int mode, int per_image_scale, float *dpred, float *dunc, double *workspace, float *out,
                  void *stream);

int xl_loss_depth(const float *pred, const float *unc, const float *gt_depth, int B, int Ho, int Wo,
                  float min_depth, float hard_clamp, float nodata, int mode, int per_image_scale,
                  float *dpred, float *dunc, double *workspace, float *out, void *stream);

int xl_loss_normal(const float *logits, const float *unc, const float *gt_normals, int B, int Ho, int Wo,
                   float hard_clamp, float nodata, int mode, int per_image_scale,
                   float *dlogits, float *dunc, double *workspace, float *out, void *stream);

/* semantics_classification_loss with CrossEntropyLoss2d (loss/semantics.py:10-18, 44-91; no class weights, no
 * uncertainty): logits [B,C,H,W], labels [B,H,W] class ids stored as floats (what the loader yields); out[1] is the
 * share of pixels whose arg-max class equals the label; dlogits = (softmax - onehot) * scale. */
int xl_loss_semantics(const float *logits, const float *labels, int B, int C, int H, int W, int per_image_scale,
                      float *dlogits, double *workspace, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif
