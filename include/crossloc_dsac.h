/*
 * crossloc_dsac.h — C ABI of the MI355X-native DSAC* pose solver (libcrossloc_hip.so).
 *
 * Drop-in boundary for the reference's `dsacstar` extension module
 * (/root/reference/dsacstar/dsacstar.cpp:887-892, PYBIND11_MODULE exporting forward_rgb,
 * backward_rgb, forward_rgbd, backward_rgbd).  The reference binds at a pybind11/ATen level;
 * here the compute core sits behind plain C entry points (pointers + sizes, no torch types) and
 * `dsacstar.py` is the thin Python shim a maintainer would keep in place of the extension.
 *
 * All pointers named *_dev are device (HIP) pointers; *_host are host pointers.
 * Strides are in ELEMENTS (floats), exactly what at::TensorAccessor would honour
 * (dsacstar.cpp:78-79).  Poses are cam->world 4x4, float32, row-major (dsacstar.cpp:172-177).
 * Every function returns 0 on success or a negative xl status (see xl_status_string).
 */
#ifndef CROSSLOC_DSAC_H
#define CROSSLOC_DSAC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XL_OK 0
#define XL_ERR_ARG (-1)        /* null pointer / non-positive size */
#define XL_ERR_GRID (-2)       /* Ho*Wo larger than the kernel supports (16384 cells) */
#define XL_ERR_HIP (-3)        /* a HIP runtime call failed; see xl_last_hip_error() */
#define XL_ERR_UNSUPPORTED (-4)

#define XL_DSAC_MAX_REF_STEPS 100          /* dsacstar.cpp:47 */
#define XL_DSAC_MAX_HYP_TRIES 1000000u     /* dsacstar.cpp:48 */
#define XL_DSAC_DBG_DOUBLES 28             /* per-image debug record, layout below */

/* The launch form xl_dsac_forward_rgb_batch selects for a batch of B images and n_hyp hypotheses: 1 = one fused launch (a
 * workgroup per image: sample, score, select, refine), S > 1 = the split form (S sub-blocks per image sample and score, a second
 * launch selects and refines).  Results are bit-identical across forms; parity tests assert which one they exercised.
 * No reference counterpart (the reference loops over hypotheses with OpenMP, dsacstar.cpp:112-130). */
int xl_dsac_forward_sub_blocks(int B, int n_hyp);

/*
 * Batched forward_rgb: replaces dsacstar_rgb_forward (dsacstar.cpp:63-178) for B independent
 * images in one launch (the reference supports batch 1 only, dsacstar_util.h:161).
 *
 *   coords_dev     [B,3,Ho,Wo] float32 scene coordinates, strides (sb, sc, sy, sx) in elements
 *   out_poses_dev  [B,16] float32, written in place (caller-owned, like outPoseSrc)
 *   n_hyp, thr, focal, ppx, ppy, alpha, max_reproj, sub
 *                  ransacHypotheses, inlierThreshold, focalLength, ppointX, ppointY,
 *                  inlierAlpha, maxReproj, subSampling of dsacstar.cpp:63-73
 *   focals_dev     optional [B] per-image focal lengths (NULL -> `focal` for every image)
 *   seed           RANSAC seed (reference: ThreadRand seed 1305, thread_rand.h:101)
 *   image0, image_stride   global index of image b is image0 + b*image_stride; the sampler is
 *                  keyed on it so results do not depend on batch composition or rank count
 *   max_tries      MAX_HYPOTHESES_TRIES (dsacstar.cpp:48); pass XL_DSAC_MAX_HYP_TRIES to mirror
 *   stream         hipStream_t to launch on (NULL = default stream); asynchronous
 *   cells_dev      optional [B,n_hyp,4] int32: sampled cells (y*Wo + x) of the accepted try
 *   tries_dev      optional [B,n_hyp] int32: tries used (negative: budget exhausted)
 *   scores_dev     optional [B,n_hyp] float64 soft-inlier scores
 *   dbg_dev        optional [B,28] float64: [0] winner, [1] refinement rounds, [2] inliers,
 *                  [3] LM evaluations, [4..15] winner pose before refinement (R row-major, t),
 *                  [16..27] refined pose (world->camera)
 */
int xl_dsac_forward_rgb_batch(const float *coords_dev, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                              int B, int Ho, int Wo, float *out_poses_dev,
                              int n_hyp, float thr, float focal, float ppx, float ppy,
                              float alpha, float max_reproj, int sub, const float *focals_dev,
                              uint64_t seed, uint64_t image0, uint64_t image_stride, uint32_t max_tries,
                              void *stream,
                              int32_t *cells_dev, int32_t *tries_dev, double *scores_dev, double *dbg_dev);

/*
 * Reference-shaped single-image call on HOST memory: same argument meaning as
 * dsacstar.forward_rgb(sceneCoordinates[1,3,Ho,Wo] CPU, outPose[4,4] CPU, ...)
 * (utils/evaluation.py:162-172).  Copies in, runs the batch kernel with B=1 on `stream`,
 * copies the pose out and synchronises.  Debug outputs are host pointers (nullable).
 */
int xl_dsac_forward_rgb_host(const float *coords_host, int64_t sc, int64_t sy, int64_t sx, int Ho, int Wo,
                             float *out_pose_host, int n_hyp, float thr, float focal, float ppx, float ppy,
                             float alpha, float max_reproj, int sub,
                             uint64_t seed, uint64_t image, uint32_t max_tries,
                             int32_t *cells_host, int32_t *tries_host, double *scores_host, double *dbg_host);

/*
 * Batched backward_rgb: replaces dsacstar_rgb_backward (dsacstar.cpp:200-215, 215-483) for B independent images.
 * Computes the DSAC* expectation of the pose loss over the soft-max distribution of n_hyp hypotheses and ACCUMULATES
 * (+=, like the reference) its gradient w.r.t. the scene coordinates.
 *
 *   coords_dev     [B,3,Ho,Wo] float32, strides (sb, sc, sy, sx) in elements
 *   grad_dev       [B,3,Ho,Wo] float32 gradient, strides (gsb, gsc, gsy, gsx); accumulated in place
 *   gt_poses_dev   [B,16] float32 ground-truth cam->world poses (gtPoseSrc)
 *   out_loss_dev   [B] float64: the expected loss the reference returns
 *   w_rot, w_trans, soft_clamp   wLossRot, wLossTrans, softClamp of dsacstar.cpp:210-212
 *   seed           randomSeed of dsacstar.cpp:215 (keys the counter-based sampler together with the image index)
 *   rec_dev        optional [B,n_hyp,XL_DSAC_BWD_REC] float64 per-hypothesis records for parity tests:
 *                  [0] prob [1] loss [2] active (prob >= 1e-3) [3] accepted inliers [4] path I clamped [5] soft-max
 *                  gradient [6..17] refined pose (R row-major, t) [18..23] dLoss/d(rvec,tvec) [24..29] pinv(JtJ) dLoss^T
 *                  [30..41] support-point gradients [42] max|jacobeanR| [43] max|dPNP| [44..49] sum_c dRepro J
 *                  [50..52] rvec of the refined pose; the rest is workspace
 *   other arguments as in xl_dsac_forward_rgb_batch.  Asynchronous on `stream`.
 */
#define XL_DSAC_BWD_REC 128
int xl_dsac_backward_rgb_batch(const float *coords_dev, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                               int B, int Ho, int Wo,
                               float *grad_dev, int64_t gsb, int64_t gsc, int64_t gsy, int64_t gsx,
                               const float *gt_poses_dev, double *out_loss_dev,
                               int n_hyp, float thr, float focal, float ppx, float ppy,
                               float w_rot, float w_trans, float soft_clamp, float alpha, float max_reproj, int sub,
                               const float *focals_dev, uint64_t seed, uint64_t image0, uint64_t image_stride,
                               uint32_t max_tries, void *stream, double *rec_dev);

/* Exported for symmetry with the reference module (dsacstar.cpp:889-891); not on CrossLoc's
 * path (no call site in the reference).  They return XL_ERR_UNSUPPORTED. */
int xl_dsac_forward_rgbd(void);
int xl_dsac_backward_rgbd(void);

const char *xl_status_string(int status);
const char *xl_last_hip_error(void);

#ifdef __cplusplus
}
#endif
#endif
