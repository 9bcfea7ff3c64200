/*
 * crossloc_data.h — C ABI of the GPU-side frame / label preparation (libcrossloc_hip.so), SURVEY.md §8(f3).
 *
 * Replaces the CPU-worker bodies of /root/reference/dataloader/dataloader.py:
 *   image_transform / cur_image_transform   :189-232, :349-393   ToPILImage -> Resize(image_height) -> [ColorJitter
 *                                                                 (brightness, contrast)] -> ToTensor -> [Normalize]
 *   batch_resize (collate_fn)               :512-563             one common scale + rotation per mini-batch
 * Decoding the PNG and reading the text files stays on the host (file formats, crossloc_amd/dataset.py); everything
 * after the decoded uint8 frame runs here, so a training batch is uploaded once as bytes.
 *
 * Arithmetic contracts (checked bit for bit against oracle/data_oracle.py, which is pinned against Pillow itself):
 *   resize      Pillow's two-pass bilinear resampler on uint8, 22-bit fixed-point coefficients, horizontal pass first
 *               (what torchvision's Resize does on a PIL image); a pass whose size does not change is skipped
 *   jitter      PIL.ImageEnhance.Brightness / Contrast on uint8 (Image.blend against black / against the rounded mean
 *               of the 'L' conversion), in the order torchvision drew; `jitter` = host array [B][3] of
 *               {brightness factor, contrast factor, 1.0f if contrast is applied first else 0.0f}, NULL = none
 *   to tensor   uint8 / 255 in float32, then (x - mean[c]) / std[c] when mean/std are given (host float[3]); NCHW out
 *   batch aug   images: bilinear resize (align_corners = false, like F.interpolate) to (oh, ow) composed with the
 *               nearest-neighbour rotation of torchvision's tensor `rotate` about the image centre, `fill` outside;
 *               labels: nearest resize (F.interpolate 'nearest') composed with the same rotation
 * All pointers are device pointers unless stated; calls are asynchronous on `stream` (the per-frame jitter records
 * travel as kernel arguments: the host arrays may be freed when a call returns) and return 0 or a negative xl
 * status (crossloc_dsac.h).  Frames of one call share their size (the reference's collate stacks them, :563).
 */
#ifndef CROSSLOC_DATA_H
#define CROSSLOC_DATA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Output size of Resize(image_height) for a stored Hs x Ws frame (smaller edge -> image_height, torchvision rule). */
int xl_data_resized_shape(int Hs, int Ws, int image_height, int *H, int *W);

/* Bytes of device scratch xl_data_prepare_images needs for B frames of Hs x Ws resized to H x W. */
long long xl_data_prepare_workspace_bytes(int B, int Hs, int Ws, int H, int W);

/* src: uint8 [B][Hs][Ws][Cs] (Cs = 3, or 4 with the alpha channel ignored: dataloader.py:305-307);
 * out: float32 [B][3][H][W] with (H, W) = xl_data_resized_shape(Hs, Ws, image_height). */
int xl_data_prepare_images(const uint8_t *src, int B, int Hs, int Ws, int Cs, int image_height,
                           const float *jitter_host, const float *mean_host, const float *std_host,
                           float *out, void *workspace, void *stream);

/* The grayscale pipeline (dataloader.py:171-187, 359-373: ... Resize -> Grayscale -> [ColorJitter] -> ToTensor ->
 * Normalize(mean[1], std[1])): Pillow's 'L' conversion (R*19595 + G*38470 + B*7471 + 0x8000) >> 16 of the resized frame,
 * the jitter on that one channel; out: float32 [B][1][H][W]; mean_host / std_host: one float each (or NULL). */
int xl_data_prepare_images_gray(const uint8_t *src, int B, int Hs, int Ws, int Cs, int image_height,
                                const float *jitter_host, const float *mean_host, const float *std_host,
                                float *out, void *workspace, void *stream);

/* in: float32 [B][C][H][W] -> out [B][C][oh][ow]: resize (bilinear = 1: images; 0: nearest, labels) + rotation by
 * angle_deg (counter-clockwise, nearest, `fill` outside the rotated frame). */
int xl_data_batch_augment(const float *in, float *out, int B, int C, int H, int W, int oh, int ow,
                          double angle_deg, float fill, int bilinear, void *stream);

#ifdef __cplusplus
}
#endif
#endif
