"""`dsacstar` module replacement: same names and positional signatures as the reference's pybind11
extension (/root/reference/dsacstar/dsacstar.cpp:887-892), backed by the HIP kernels in
libcrossloc_hip.so through the C ABI of include/crossloc_dsac.h.

    import torch, dsacstar                      # README.md:51 of the reference: torch first
    dsacstar.forward_rgb(scene_coords, out_pose, 64, 10.0, f, 360.0, 240.0, 100.0, 100.0, 8)

is the exact call utils/evaluation.py:162-172 makes.  Differences, all additive:
  * `sceneCoordinates` / `outPose` may live on the GPU (the reference dereferences them as host
    memory, dsacstar.cpp:78-79); CPU tensors take the host entry point (H2D, kernel, D2H, sync).
  * no stdout chatter (the reference prints ANSI-coloured timings, dsacstar.cpp:97-169).
  * the sampler is counter-based.  The reference's std::mt19937 state carries across calls
    (thread_rand.cpp:17); here every call consumes one "image index" from a module counter
    (reset with `set_image_index`), and results do not depend on thread or rank count.
  * `forward_rgb_batch` localises B images per launch (the reference is batch-1 only).
  * `backward_rgb` (dsacstar.cpp:200-483, exported by the reference but never called by CrossLoc) and its batched
    form `backward_rgb_batch` run on the GPU as well; the RGB-D pair stays unimplemented.
There is no CPU fallback: without the HIP library the call raises.
"""
import ctypes

import torch

from . import _lib

RANSAC_SEED = 1305                      # thread_rand.h:101 default seed of the reference
MAX_HYPOTHESES_TRIES = 1000000          # dsacstar.cpp:48
MAX_REF_STEPS = 100                     # dsacstar.cpp:47 (compiled into the kernel)
_image_index = 0


def set_image_index(index):
    """Next forward_rgb call is keyed as image `index` (then index+1, ...)."""
    global _image_index
    _image_index = int(index)


def _take_image_index():
    """The image index of the next single-image call, consumed (shared by the ctypes shim and the compiled binding)."""
    global _image_index
    image = _image_index
    _image_index += 1
    return image


def _check_coords(t, batched):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError("sceneCoordinates must be a torch.Tensor")
    if t.dim() != 4:
        # at::Tensor::accessor<float,4>() throws c10::Error -> RuntimeError (dsacstar.cpp:78)
        raise RuntimeError("expected 4 dims but tensor has %d" % t.dim())
    if t.dtype != torch.float32:
        raise RuntimeError("expected scalar type Float but found %s" % str(t.dtype).replace("torch.", ""))
    if t.size(1) != 3:
        raise RuntimeError("sceneCoordinates must be [B,3,H,W], got %s" % (tuple(t.shape),))
    if not batched and t.size(0) != 1:
        raise RuntimeError("forward_rgb supports batch size 1 only (dsacstar_util.h:161); use forward_rgb_batch")


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def forward_rgb_batch(sceneCoordinates, outPoses, ransacHypotheses, inlierThreshold, focalLength, ppointX,
                      ppointY, inlierAlpha, maxReproj, subSampling, image0=0, image_stride=1, focals=None,
                      seed=None, max_tries=None, debug=False):
    """Localise B images in one launch. sceneCoordinates [B,3,Ho,Wo] float32 CUDA (any strides),
    outPoses [B,4,4] float32 CUDA contiguous, written in place (asynchronous on the current stream).
    Image b is keyed as image0 + b*image_stride.  With debug=True returns a dict of device tensors:
    cells [B,nHyp,4] int32, tries [B,nHyp] int32, scores [B,nHyp] float64, dbg [B,28] float64."""
    _check_coords(sceneCoordinates, True)
    if not sceneCoordinates.is_cuda or not outPoses.is_cuda:
        raise RuntimeError("forward_rgb_batch needs CUDA(HIP) tensors; there is no CPU fallback")
    B, _, Ho, Wo = sceneCoordinates.shape
    if outPoses.dtype != torch.float32 or tuple(outPoses.shape) != (B, 4, 4) or not outPoses.is_contiguous():
        raise RuntimeError("outPoses must be a contiguous float32 [B,4,4] tensor")
    dev = sceneCoordinates.device
    if focals is not None:
        focals = focals.to(device=dev, dtype=torch.float32).contiguous()
        assert focals.numel() == B
    out = None
    cells = tries = scores = dbg = None
    if debug:
        cells = torch.zeros((B, ransacHypotheses, 4), dtype=torch.int32, device=dev)
        tries = torch.zeros((B, ransacHypotheses), dtype=torch.int32, device=dev)
        scores = torch.zeros((B, ransacHypotheses), dtype=torch.float64, device=dev)
        dbg = torch.zeros((B, 28), dtype=torch.float64, device=dev)
        out = dict(cells=cells, tries=tries, scores=scores, dbg=dbg)
    sb, sc, sy, sx = sceneCoordinates.stride()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        rc = _lib.lib().xl_dsac_forward_rgb_batch(
            _ptr(sceneCoordinates), sb, sc, sy, sx, B, Ho, Wo, _ptr(outPoses),
            int(ransacHypotheses), float(inlierThreshold), float(focalLength), float(ppointX), float(ppointY),
            float(inlierAlpha), float(maxReproj), int(subSampling), _ptr(focals),
            int(RANSAC_SEED if seed is None else seed), int(image0), int(image_stride),
            int(MAX_HYPOTHESES_TRIES if max_tries is None else max_tries), ctypes.c_void_p(stream),
            _ptr(cells), _ptr(tries), _ptr(scores), _ptr(dbg))
    _lib.check(rc)
    return out


def forward_rgb(sceneCoordinates, outPose, ransacHypotheses, inlierThreshold, focalLength, ppointX, ppointY,
                inlierAlpha, maxReproj, subSampling):
    """dsacstar_rgb_forward (dsacstar.cpp:63-73): estimate the camera pose of ONE image from its scene
    coordinate prediction [1,3,H,W]; writes the 4x4 cam->world matrix into outPose in place."""
    _check_coords(sceneCoordinates, False)
    if not isinstance(outPose, torch.Tensor) or outPose.dim() != 2 or outPose.dtype != torch.float32 \
            or tuple(outPose.shape) != (4, 4):
        raise RuntimeError("outPose must be a float32 [4,4] tensor")
    image = _take_image_index()
    _, _, Ho, Wo = sceneCoordinates.shape
    if sceneCoordinates.is_cuda:
        dst = outPose if (outPose.is_cuda and outPose.is_contiguous()) else \
            torch.empty((4, 4), dtype=torch.float32, device=sceneCoordinates.device)
        forward_rgb_batch(sceneCoordinates, dst.view(1, 4, 4), ransacHypotheses, inlierThreshold, focalLength,
                          ppointX, ppointY, inlierAlpha, maxReproj, subSampling, image0=image)
        if dst is not outPose:
            outPose.copy_(dst)          # synchronising D2H copy, like the reference's blocking call
        return None
    if outPose.is_cuda:
        raise RuntimeError("outPose on the GPU needs sceneCoordinates on the GPU too")
    host_pose = outPose if outPose.is_contiguous() else torch.empty((4, 4), dtype=torch.float32)
    _, sc, sy, sx = sceneCoordinates.stride()
    rc = _lib.lib().xl_dsac_forward_rgb_host(
        _ptr(sceneCoordinates), sc, sy, sx, Ho, Wo, _ptr(host_pose), int(ransacHypotheses),
        float(inlierThreshold), float(focalLength), float(ppointX), float(ppointY), float(inlierAlpha),
        float(maxReproj), int(subSampling), int(RANSAC_SEED), int(image), int(MAX_HYPOTHESES_TRIES),
        None, None, None, None)
    _lib.check(rc)
    if host_pose is not outPose:
        outPose.copy_(host_pose)
    return None


BWD_REC = 128                            # XL_DSAC_BWD_REC: doubles per hypothesis in the debug record


def backward_rgb_batch(sceneCoordinates, outSceneCoordinatesGrad, gtPoses, ransacHypotheses, inlierThreshold,
                       focalLength, ppointX, ppointY, wLossRot, wLossTrans, softClamp, inlierAlpha, maxReproj,
                       subSampling, randomSeed, image0=0, image_stride=1, focals=None, max_tries=None, debug=False):
    """DSAC* expected pose loss of B images and its gradient w.r.t. their scene coordinates in one launch sequence.
    sceneCoordinates [B,3,Ho,Wo] and outSceneCoordinatesGrad (same shape, any strides; ACCUMULATED like the
    reference's `+=`) are float32 CUDA tensors, gtPoses [B,4,4] cam->world.  Returns the expected losses as a
    float64 CUDA tensor [B] (asynchronous on the current stream); with debug=True also the per-hypothesis records
    [B,nHyp,BWD_REC] (layout: include/crossloc_dsac.h)."""
    _check_coords(sceneCoordinates, True)
    g = outSceneCoordinatesGrad
    if not sceneCoordinates.is_cuda or not isinstance(g, torch.Tensor) or not g.is_cuda:
        raise RuntimeError("backward_rgb_batch needs CUDA(HIP) tensors; there is no CPU fallback")
    if g.dtype != torch.float32 or tuple(g.shape) != tuple(sceneCoordinates.shape):
        raise RuntimeError("outSceneCoordinatesGrad must be float32 with the shape of sceneCoordinates")
    B, _, Ho, Wo = sceneCoordinates.shape
    dev = sceneCoordinates.device
    gt = gtPoses.to(device=dev, dtype=torch.float32).reshape(B, 16).contiguous()
    if focals is not None:
        focals = focals.to(device=dev, dtype=torch.float32).contiguous()
        assert focals.numel() == B
    loss = torch.zeros((B,), dtype=torch.float64, device=dev)
    rec = torch.zeros((B, ransacHypotheses, BWD_REC), dtype=torch.float64, device=dev) if debug else None
    sb, sc, sy, sx = sceneCoordinates.stride()
    gb, gc, gy, gx = g.stride()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        rc = _lib.lib().xl_dsac_backward_rgb_batch(
            _ptr(sceneCoordinates), sb, sc, sy, sx, B, Ho, Wo, _ptr(g), gb, gc, gy, gx, _ptr(gt), _ptr(loss),
            int(ransacHypotheses), float(inlierThreshold), float(focalLength), float(ppointX), float(ppointY),
            float(wLossRot), float(wLossTrans), float(softClamp), float(inlierAlpha), float(maxReproj),
            int(subSampling), _ptr(focals), int(randomSeed), int(image0), int(image_stride),
            int(MAX_HYPOTHESES_TRIES if max_tries is None else max_tries), ctypes.c_void_p(stream), _ptr(rec))
    _lib.check(rc)
    return (loss, rec) if debug else loss


def backward_rgb(sceneCoordinates, outSceneCoordinatesGrad, gtPose, ransacHypotheses, inlierThreshold, focalLength,
                 ppointX, ppointY, wLossRot, wLossTrans, softClamp, inlierAlpha, maxReproj, subSampling, randomSeed):
    """dsacstar_rgb_backward (dsacstar.cpp:200-215): pose estimation of ONE image plus the gradient of the expected
    pose loss w.r.t. its scene coordinates, accumulated into outSceneCoordinatesGrad [1,3,H,W]; returns the DSAC
    expectation of the pose loss as a Python float.  Tensors may live on the GPU or (like the reference) the CPU."""
    _check_coords(sceneCoordinates, False)
    g = outSceneCoordinatesGrad
    if not isinstance(g, torch.Tensor) or g.dtype != torch.float32 or tuple(g.shape) != tuple(sceneCoordinates.shape):
        raise RuntimeError("outSceneCoordinatesGrad must be float32 with the shape of sceneCoordinates")
    if not isinstance(gtPose, torch.Tensor) or tuple(gtPose.shape) != (4, 4):
        raise RuntimeError("gtPose must be a [4,4] tensor")
    if sceneCoordinates.is_cuda and g.is_cuda:
        loss = backward_rgb_batch(sceneCoordinates, g, gtPose.view(1, 4, 4), ransacHypotheses, inlierThreshold,
                                  focalLength, ppointX, ppointY, wLossRot, wLossTrans, softClamp, inlierAlpha,
                                  maxReproj, subSampling, randomSeed)
        return float(loss.item())
    if sceneCoordinates.is_cuda or g.is_cuda:
        raise RuntimeError("sceneCoordinates and outSceneCoordinatesGrad must be on the same device")
    dev = torch.device("cuda")                       # raises if there is no HIP device: no CPU fallback
    co = sceneCoordinates.to(dev)
    gd = torch.zeros(tuple(g.shape), dtype=torch.float32, device=dev)
    gd.copy_(g)
    loss = backward_rgb_batch(co, gd, gtPose.view(1, 4, 4), ransacHypotheses, inlierThreshold, focalLength, ppointX,
                              ppointY, wLossRot, wLossTrans, softClamp, inlierAlpha, maxReproj, subSampling, randomSeed)
    g.copy_(gd)
    return float(loss.item())


def forward_rgbd(*args, **kwargs):
    """dsacstar_rgbd_forward (dsacstar.cpp:495-629): RGB-D variant, no call site in CrossLoc."""
    raise NotImplementedError("dsacstar.forward_rgbd is not on CrossLoc's path and is not implemented")


def backward_rgbd(*args, **kwargs):
    """dsacstar_rgbd_backward (dsacstar.cpp:631-885): RGB-D variant, no call site in CrossLoc."""
    raise NotImplementedError("dsacstar.backward_rgbd is not on CrossLoc's path and is not implemented")
