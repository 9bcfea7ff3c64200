"""MI355X-native `TransPoseNet` — same constructor, attributes and state_dict keys as the reference
(/root/reference/networks/networks.py:375-502) so `load_state_dict` on reference checkpoints is strict-clean
(116 tensors single-task, 270 for the 3-encoder CrossLoc net), but `forward` does not run PyTorch ops: it is
lowered once per input shape to an op list (include/crossloc_cnn.h) of hand-written HIP kernels — implicit-GEMM
convolutions on fp32 MFMA, two-pass GroupNorm with fused ReLU/residual epilogues, fused decoder head — and
executed with one C call on the current HIP stream.  NCHW at the module boundary, NHWC inside.  With autograd enabled
and trainable parameters, forward keeps the activations and `loss.backward()` runs a second op list (data
gradients through the same MFMA kernel, split-K weight gradients, fused GroupNorm/ReLU/residual backward).

The nn.Conv2d / nn.GroupNorm children are parameter containers only (they give identical keys, shapes and
default initialisation); they are never called.  There is no CPU/eager fallback: forward on a non-GPU tensor
raises.  The backward pass covers the single-task and the 3-encoder MLR networks (frozen encoders are skipped).
"""
import ctypes
import math
import os
import weakref

import torch
import torch.nn as nn

from . import _lib

XL_OP_CONV1, XL_OP_CONV, XL_OP_GN_STATS, XL_OP_GN_APPLY, XL_OP_HEAD = 0, 1, 2, 3, 4
GN_RELU_IN, GN_ADD, GN_RELU_OUT, GN_ACC_AUX, GN_NO_CONV_BIAS = 1, 2, 4, 8, 16
XL_OP_WGRAD, XL_OP_GNB_STATS, XL_OP_GNB_APPLY, XL_OP_GNB_PARAMS, XL_OP_HEAD_BWD, XL_OP_CONV1_WGRAD = 5, 6, 7, 8, 9, 10
XL_OP_GN_FINAL = 11
XL_OP_WINO_IN, XL_OP_WINO_OUT = 12, 13
XL_OP_DUC_HEAD = 14
XL_OP_DUC_HEAD_BWD = 15
XL_OP_WINO_DY, XL_OP_WINO_WFINAL, XL_OP_GNB_FINAL = 16, 17, 18
XL_OP_STEM12 = 19
XL_OP_S2_DGRAD = 20
CONV_DGRAD, CONV_ACCUMULATE, CONV_SPLIT_BF16 = 1, 2, 64
CONV_NORM_IN, CONV_NORM_RELU = 128, 256
CONV_SPLIT_IL = 512
CONV_SPLIT_ACT = 1024
CONV_M_TILE_MAJOR = 2048
CONV_NORM_ADD = 4096
CONV_PAIR_F16 = 8192
CONV_PAIR_AMAX = 16384
XL_OP_FILL0 = 21
XL_OP_GNB_PARAMS_LIST = 22
# entries of the device tables of xl_cnn_repack_pairs / XL_OP_GNB_PARAMS_LIST (include/crossloc_cnn.h: xl_pair_item, xl_gnb_params_item)
import numpy as _np   # noqa: E402
PAIR_ITEM_DTYPE = _np.dtype([("src", "<u8"), ("dst", "<u8"), ("rows", "<i4"), ("K", "<i4"), ("kind", "<i4"), ("pad", "<i4")])
GNB_PARAMS_ITEM_DTYPE = _np.dtype([("sums", "<u8"), ("gamma", "<u8"), ("dgamma", "<u8"), ("dbeta", "<u8"), ("dbias", "<u8"),
                                   ("B", "<i4"), ("C", "<i4"), ("G", "<i4"), ("HW", "<i4")])
XL_ERR_UNSUPPORTED = -4            # include/crossloc_dsac.h


class XlOp(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("type", "B", "Hi", "Wi", "Cin", "Ho", "Wo", "Cout", "ksize", "stride", "groups", "nchunks",
                 "flags", "ld_in", "ld_out", "ld_aux", "n_task", "n_pos", "nchunks2", "reserved_i")] + \
               [(n, ctypes.c_float) for n in ("eps", "clamp_lo", "clamp_hi", "reserved")] + \
               [(n, ctypes.c_void_p) for n in ("in_", "w", "bias", "aux", "stats", "out", "aux2", "out2", "stats2", "scale")]


def _bind():
    L = _lib.lib()
    if not hasattr(L, "_cnn_bound"):
        L.xl_cnn_run.restype = ctypes.c_int
        L.xl_cnn_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.xl_cnn_op_size.restype = ctypes.c_int
        L.xl_cnn_pack_conv_weight.restype = ctypes.c_int
        L.xl_cnn_pack_conv_weight.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_void_p]
        L.xl_cnn_pack_conv_weight_dgrad.restype = ctypes.c_int
        L.xl_cnn_pack_conv_weight_dgrad.argtypes = L.xl_cnn_pack_conv_weight.argtypes
        L.xl_cnn_pack_wino_weight.restype = ctypes.c_int
        L.xl_cnn_pack_wino_weight.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
        L.xl_cnn_split_weight.restype = ctypes.c_int
        L.xl_cnn_split_weight.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.xl_cnn_pack_wino_weight_pair.restype = ctypes.c_int
        L.xl_cnn_pack_wino_weight_pair.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        L.xl_cnn_pair_weight.restype = ctypes.c_int
        L.xl_cnn_pair_weight.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.xl_cnn_repack_pairs.restype = ctypes.c_int
        L.xl_cnn_repack_pairs.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
        L.xl_cnn_pair_activation.restype = ctypes.c_int
        L.xl_cnn_pair_activation.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.xl_cnn_pair_scales.restype = ctypes.c_int
        L.xl_cnn_pair_scales.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_void_p]
        L.xl_cnn_graph_capture.restype = ctypes.c_int
        L.xl_cnn_graph_capture.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        L.xl_cnn_graph_launch.restype = ctypes.c_int
        L.xl_cnn_graph_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.xl_cnn_graph_destroy.restype = ctypes.c_int
        L.xl_cnn_graph_destroy.argtypes = [ctypes.c_void_p]
        L.xl_cnn_last_error.restype = ctypes.c_char_p
        L.xl_cnn_item_size.restype = ctypes.c_int
        L.xl_cnn_item_size.argtypes = [ctypes.c_int]
        if (L.xl_cnn_item_size(0), L.xl_cnn_item_size(1)) != (PAIR_ITEM_DTYPE.itemsize, GNB_PARAMS_ITEM_DTYPE.itemsize):
            raise _lib.XlError("device-table entry layout mismatch: C %d / %d vs numpy %d / %d" % (
                L.xl_cnn_item_size(0), L.xl_cnn_item_size(1), PAIR_ITEM_DTYPE.itemsize, GNB_PARAMS_ITEM_DTYPE.itemsize))
        if L.xl_cnn_op_size() != ctypes.sizeof(XlOp):
            raise _lib.XlError("xl_op layout mismatch: C %d vs ctypes %d" % (L.xl_cnn_op_size(), ctypes.sizeof(XlOp)))
        L._cnn_bound = True
    return L


def _check(rc):
    if rc != 0:
        L = _lib.lib()
        raise _lib.XlError("crossloc_hip cnn: %s %s (status %d)" % (
            L.xl_status_string(rc).decode(), L.xl_cnn_last_error().decode(), rc))


# ------------------------------------------------------------------------------------------ parameter containers

def _create_res_block(tiny, num_gn_channel, ch_down_factor=1):
    """networks.py:133-146 (ReLU entries keep the Sequential indices 0,1,3,4,6,7 of the reference keys)."""
    num_ch = (512, 128)[tiny] // ch_down_factor
    g = min(num_gn_channel, num_ch)
    return nn.Sequential(nn.Conv2d(num_ch, num_ch, 3, 1, 1), nn.GroupNorm(g, num_ch), nn.ReLU(),
                         nn.Conv2d(num_ch, num_ch, 1, 1, 0), nn.GroupNorm(g, num_ch), nn.ReLU(),
                         nn.Conv2d(num_ch, num_ch, 3, 1, 1), nn.GroupNorm(g, num_ch), nn.ReLU())


def _create_mlr_concatenator(num_mlr, tiny, num_gn_channel):
    """networks.py:149-163"""
    cin, cout = (512, 128)[tiny] * num_mlr, (512, 128)[tiny]
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1), nn.GroupNorm(num_gn_channel, cout), nn.ReLU(),
                         nn.Conv2d(cout, cout, 1, 1, 0), nn.GroupNorm(num_gn_channel, cout), nn.ReLU(),
                         nn.Conv2d(cout, cout, 3, 1, 1), nn.GroupNorm(num_gn_channel, cout), nn.ReLU())


def _create_mlr_skip_layer(num_mlr, tiny, num_gn_channel):
    """networks.py:166-172"""
    cin, cout = (512, 128)[tiny] * num_mlr, (512, 128)[tiny]
    return nn.Sequential(nn.Conv2d(cin, cout, 1, 1, 0), nn.GroupNorm(num_gn_channel, cout))


class TransPoseNetEncoder(nn.Module):
    """Parameter layout of networks.py:175-219."""

    def __init__(self, tiny, grayscale, enc_add_res_block=0, num_gn_channel=32):
        super().__init__()
        self.tiny, self.grayscale = tiny, grayscale
        self.enc_add_res_block, self.num_gn_channel = enc_add_res_block, num_gn_channel
        g = num_gn_channel
        c4, c5 = (256, 128)[tiny], (512, 128)[tiny]
        self.conv1 = nn.Conv2d(1 if grayscale else 3, g, 3, 1, 1)
        self.norm1 = nn.GroupNorm(g, g)
        self.conv2 = nn.Conv2d(g, 64, 3, 2, 1)
        self.norm2 = nn.GroupNorm(g, 64)
        self.conv3 = nn.Conv2d(64, 128, 3, 2, 1)
        self.norm3 = nn.GroupNorm(g, 128)
        self.conv4 = nn.Conv2d(128, c4, 3, 2, 1)
        self.norm4 = nn.GroupNorm(g, c4)
        self.res1_conv1 = nn.Conv2d(c4, c4, 3, 1, 1)
        self.res1_norm1 = nn.GroupNorm(g, c4)
        self.res1_conv2 = nn.Conv2d(c4, c4, 1, 1, 0)
        self.res1_norm2 = nn.GroupNorm(g, c4)
        self.res1_conv3 = nn.Conv2d(c4, c4, 3, 1, 1)
        self.res1_norm3 = nn.GroupNorm(g, c4)
        self.res2_conv1 = nn.Conv2d(c4, c5, 3, 1, 1)
        self.res2_norm1 = nn.GroupNorm(g, c5)
        self.res2_conv2 = nn.Conv2d(c5, c5, 1, 1, 0)
        self.res2_norm2 = nn.GroupNorm(g, c5)
        self.res2_conv3 = nn.Conv2d(c5, c5, 3, 1, 1)
        self.res2_norm3 = nn.GroupNorm(g, c5)
        if not tiny:
            self.res2_skip = nn.Conv2d(256, 512, 1, 1, 0)
            self.res2_skip_norm = nn.GroupNorm(g, 512)
        self.enc_add_res_block_ls = [_create_res_block(tiny, g) for _ in range(enc_add_res_block)]
        for i, block in enumerate(self.enc_add_res_block_ls):
            self.add_module('enc_add_res_block{:d}'.format(i + 1), block)


class DenseUpsamplingConvolution(nn.Module):
    """Parameter layout of networks.py:259-273 (conv 3x3 -> GroupNorm -> ReLU -> pixel shuffle x down_sampling_rate)."""

    def __init__(self, down_sampling_rate, in_channel, num_classes, num_gn_channel=32):
        super().__init__()
        self.conv = nn.Conv2d(in_channel, (down_sampling_rate ** 2) * num_classes, 3, 1, 1)
        self.norm = nn.GroupNorm(num_gn_channel, (down_sampling_rate ** 2) * num_classes)


class TransPoseNetDecoder(nn.Module):
    """Parameter layout of networks.py:276-317, including the full_size_output (DUC / semantics) branch."""

    def __init__(self, mean, tiny, dec_add_res_block=0, num_task_channel=3, num_pos_channel=1, num_gn_channel=32,
                 full_size_output=False):
        super().__init__()
        self.register_buffer('mean', mean.clone().float())
        self.tiny, self.dec_add_res_block = tiny, dec_add_res_block
        self.num_task_channel, self.num_pos_channel = num_task_channel, num_pos_channel
        self.num_gn_channel, self.full_size_output = num_gn_channel, full_size_output
        c = (512, 128)[tiny]
        g = num_gn_channel
        self.dec_add_res_block_ls = [_create_res_block(tiny, g) for _ in range(dec_add_res_block)]
        for i, block in enumerate(self.dec_add_res_block_ls):
            self.add_module('dec_add_res_block{:d}'.format(i + 1), block)
        self.res3_conv1 = nn.Conv2d(c, c, 1, 1, 0)
        self.res3_norm1 = nn.GroupNorm(g, c)
        self.res3_conv2 = nn.Conv2d(c, c, 1, 1, 0)
        self.res3_norm2 = nn.GroupNorm(g, c)
        self.res3_conv3 = nn.Conv2d(c, c, 1, 1, 0)
        self.res3_norm3 = nn.GroupNorm(g, c)
        self.fc1 = nn.Conv2d(c, c, 1, 1, 0)
        self.fc1_norm = nn.GroupNorm(min(c, g), c)
        self.fc2 = nn.Conv2d(c, c, 1, 1, 0)
        self.fc2_norm = nn.GroupNorm(min(c, g), c)
        assert num_task_channel > 0 and num_pos_channel >= 0
        assert num_task_channel == len(mean)
        if full_size_output:
            nc = num_task_channel + num_pos_channel
            self.duc_upsample = DenseUpsamplingConvolution(8, c, nc)
            self.fc3 = nn.Conv2d(nc, nc, 1, 1, 0)
        else:
            self.fc3 = nn.Conv2d(c, num_task_channel + num_pos_channel, 1, 1, 0)


# ------------------------------------------------------------------------------------------ lowering

class _Plan:
    """One pass for a fixed (B, H, W): op array + workspace, replayed on every call.

    train=False: inference plan (GroupNorm in place, buffers recycled as soon as they are dead).
    train=True:  every conv output (pre-norm) and every activation is kept, a tape of the layers is recorded and a
                 second op array for the backward pass is lowered from it (data gradients through the same
                 implicit-GEMM kernel, weight gradients, GroupNorm/epilogue backward, head backward)."""

    SPLIT_DEFAULT = "il"           # default of XL_GEMM_SPLIT_BF16 (see conv_wino); "0" = fp32 MFMA everywhere

    def __init__(self, net, B, H, W, device, train=False):
        self.B, self.H, self.W, self.device, self.train = B, H, W, device, train
        # separate per-image statistics passes instead of the conv-epilogue ones (whose partial sums are grouped by
        # conv tile, i.e. by the position of a frame inside the batch): results bitwise independent of the batch
        self.separate_stats = bool(getattr(net, "batch_invariant", False) or os.environ.get("XL_NO_FUSED_STATS"))
        self.ops = []
        self.packed_split = {}
        self.packed_1x1 = {}
        self.packed_c1 = {}
        self.keep = []                      # tensors the op pointers reference
        self.free = {}                      # numel -> [tensor]
        self.packed = {}                    # (id(param), kind) -> packed weight tensor
        self.net = net
        self.max_stats = 0
        self.stats_ops = []                 # (op index) of GN ops using the shared stats scratch (inference)
        self.out_op_index = None
        self.image_op_indices = []
        self.tape = []
        # round 5: the activation scales of the fp16-pair GEMMs (csrc/xl_gemm_pair.hip): {s, 1/s} for GroupNorm outputs and sums of
        # them, {s/256, 256/s} for their Winograd transforms; written by _update_pair_scales() once the plan's GroupNorm layers are
        # known, and again whenever the parameters change
        self.packed_pair = {}
        self.pair_scales = torch.zeros(8, dtype=torch.float32, device=device)
        self._lower(net)
        self._update_pair_scales()
        self.op_array = (XlOp * len(self.ops))(*self.ops)
        self.stats = torch.zeros(max(self.max_stats, 1), dtype=torch.float64, device=device)
        for i in self.stats_ops:
            self.op_array[i].stats = self.stats.data_ptr()
        self.coeff = torch.zeros(max(getattr(self, "max_coeff", 0), 1), dtype=torch.float32, device=device)
        for i, op in enumerate(self.ops):
            # every GN_FINAL writes the shared coefficient buffer; its consumer - the following GN_APPLY, or the
            # Winograd input transform of the next layer when the apply was deferred - runs before the next GN_FINAL
            if op.type == XL_OP_GN_FINAL and not train:
                self.op_array[i].out = self.coeff.data_ptr()
            elif op.type == XL_OP_GN_APPLY and not train:
                self.op_array[i].aux2 = self.coeff.data_ptr()
        for i in getattr(self, "deferred_gn_consumers", []):
            self.op_array[i].aux2 = self.coeff.data_ptr()
        if getattr(self, "aux_final_ops", None):      # the second table: residuals normalised by the pass that adds them
            self.coeff_aux = torch.zeros_like(self.coeff)
            for i in self.aux_final_ops:
                self.op_array[i].out = self.coeff_aux.data_ptr()
            for i in getattr(self, "aux_apply_ops", []):
                self.op_array[i].aux2 = self.coeff_aux.data_ptr()
            for i in getattr(self, "aux_coef_consumers", []):
                self.op_array[i].w = self.coeff_aux.data_ptr()
        assert not getattr(self, "pending_gn", None), "a deferred GroupNorm was never consumed"
        assert not getattr(self, "pending_aux", None), "a deferred residual GroupNorm was never consumed"
        assert not getattr(self, "pending_fold", None), "a folded GroupNorm apply was never consumed"
        if train:
            self._lower_backward()
            self._update_pair_scales()          # (the backward GEMMs may be the plan's first pair operands: small maps)

    # -- workspace
    def alloc(self, numel):
        lst = self.free.get(numel)
        if lst:
            return lst.pop()
        t = torch.empty(numel, dtype=torch.float32, device=self.device)
        self.keep.append(t)
        return t

    def release(self, t):
        if self.train:                      # training keeps every forward tensor for the backward pass
            return
        held = getattr(self, "held", {}).get(id(t))
        if held is not None:                # an input of a GroupNorm apply that a later op performs (fold): free it then
            held[1] = True
            return
        self.free.setdefault(t.numel(), []).append(t)

    # -- folded GroupNorm applies (inference plans).  A GroupNorm(+ReLU, +residual, +ReLU) whose result has SEVERAL consumers
    # - the first convolution of the next block and, later, a residual branch - used to be a pass of its own (read the raw
    # conv output and the residual, write the activation).  With `share` the pass is left to the first consumer when that is
    # an F(6x6,3x3) layer: its input transform reads the raw tensor and the residual anyway, applies the normalisation on load
    # and writes the activation (to a buffer of its own) for the consumers that follow.  Same arithmetic, same bits.
    def _fold_begin(self, ap, raw, aux):
        t, H, W, C, ld, off = raw
        mat = self.alloc(self.B * H * W * C)
        res = (mat, H, W, C, C, 0)
        if not hasattr(self, "pending_fold"):
            self.pending_fold, self.held = {}, {}
        tensors = [raw[0]] + ([aux[0]] if aux is not None else [])
        for x in tensors:
            self.held[id(x)] = [x, x is raw[0]]           # the raw conv output has no other owner: released with the fold
        self.pending_fold[self._act_key(res)] = dict(ap=ap, raw=raw, tensors=tensors, aux_ap=self._aux_take(aux))
        return res

    def _unhold(self, t):
        """End of a hold on a tensor (see `held`): release it now if its owner released it meanwhile."""
        if t is None:
            return
        entry = getattr(self, "held", {}).pop(id(t), None)
        if entry is not None and entry[1]:
            self.release(t)

    def _fold_end(self, fold):
        for x in fold["tensors"]:
            entry = self.held.pop(id(x))
            if entry[1]:
                self.release(x)

    def _fold_materialise(self, fold, act):
        """The first consumer cannot apply it on load: run the GroupNorm apply as a pass (raw -> the activation's buffer)."""
        ap = fold["ap"]
        ap.out, ap.ld_out = act[0].data_ptr() + 4 * act[5], act[4]
        if fold.get("aux_ap") is not None:
            self._aux_apply(fold["aux_ap"])
        self.stats_ops.append(len(self.ops))
        self.ops.append(ap)
        self._fold_end(fold)

    # -- a residual whose own GroupNorm + ReLU has no other consumer than the addition (res2_conv3 -> res2_norm3 -> ReLU, added to
    # the normalised skip branch, networks.py:247-250 of the reference): its apply pass is left to the pass that performs the
    # addition - the fold form of the next block's input transform reads the RAW residual and normalises it while loading.  Its
    # {scale, shift} pairs must outlive the GN_FINAL of the skip branch: they go to a second coefficient table.
    def _aux_defer(self, act):
        """`act` was produced by cgr(..., defer=True): move its pending GroupNorm apply to the residual slot."""
        ap = getattr(self, "pending_gn", {}).pop(self._act_key(act), None)
        if ap is None:
            return                                    # (the layer's form could not defer it: it ran as a pass)
        assert not getattr(self, "pending_aux", None)
        fin = max(i for i, op in enumerate(self.ops) if op.type == XL_OP_GN_FINAL)
        self.aux_final_ops = getattr(self, "aux_final_ops", []) + [fin]
        self.pending_aux = {self._act_key(act): ap}

    def _aux_take(self, aux):
        return getattr(self, "pending_aux", {}).pop(self._act_key(aux), None) if aux is not None else None

    def _aux_apply(self, ap):
        """The addition is not performed by a fold: apply the residual's GroupNorm as a pass after all (in place)."""
        self.aux_apply_ops = getattr(self, "aux_apply_ops", []) + [len(self.ops)]
        self.stats_ops.append(len(self.ops))
        self.ops.append(ap)

    def release_grad(self, t):
        self.free.setdefault(t.numel(), []).append(t)

    # -- weights
    # Winograd F(2x2,3x3) weight transform U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] (Lavin & Gray 2016)
    _WINO_G = {2: ((1.0, 0.0, 0.0), (0.5, 0.5, 0.5), (0.5, -0.5, 0.5), (0.0, 0.0, 1.0)),
               # F(4x4,3x3), interpolation points 0, +-1, +-2, inf
               4: ((1 / 4, 0.0, 0.0), (-1 / 6, -1 / 6, -1 / 6), (-1 / 6, 1 / 6, -1 / 6), (1 / 24, 1 / 12, 1 / 6),
                   (1 / 24, -1 / 12, 1 / 6), (0.0, 0.0, 1.0)),
               # F(6x6,3x3), interpolation points 0, +-1, +-2, +-1/2, inf (the scaling of Lavin's wincnn set)
               6: ((1.0, 0.0, 0.0), (-2 / 9, -2 / 9, -2 / 9), (-2 / 9, 2 / 9, -2 / 9), (1 / 90, 1 / 45, 2 / 45),
                   (1 / 90, -1 / 45, 2 / 45), (1 / 45, 1 / 90, 1 / 180), (1 / 45, -1 / 90, 1 / 180), (0.0, 0.0, 1.0))}

    def pack_conv_wino(self, conv, m, dgrad=False):
        """[(m+2)^2][Cout][Cin] transformed weights of a 3x3 convolution: one plain [Cout][Cin] GEMM operand per
        frequency of F(m x m, 3x3).  dgrad: the data gradient is the convolution with the flipped kernel and the
        channel roles swapped, so its operands are [(m+2)^2][Cin][Cout]."""
        w = conv.weight
        kind = "wino%d%s" % (m, "d" if dgrad else "")
        key = (id(w), kind)
        if key not in self.packed:
            src = w.detach().to(device=self.device, dtype=torch.float32).contiguous()
            dst = torch.empty((m + 2) ** 2 * src.shape[0] * src.shape[1], dtype=torch.float32, device=self.device)
            self.packed[key] = (dst, src, kind)
            self._pack(dst, src, kind)
        return self.packed[key][0]

    @staticmethod
    def split_bf16(x):
        """fp32 tensor -> [3, ...] bf16 planes with x = p0 + p1 + p2 exactly (8 + 8 + 8 mantissa bits), as int16 storage."""
        p0 = x.to(torch.bfloat16)
        r1 = x - p0.to(torch.float32)
        p1 = r1.to(torch.bfloat16)
        p2 = (r1 - p1.to(torch.float32)).to(torch.bfloat16)
        return torch.stack([p0, p1, p2]).contiguous().view(torch.int16)

    @staticmethod
    def split_bf16_interleaved(x, C):
        """fp32 tensor whose last dimension runs over C channels -> the interleaved-plane operand layout of the 256 x 256
        split GEMM, [..., C/16, 3, 16] bf16 (as int16 storage): the three planes of a 16-channel chunk next to each other."""
        planes = _Plan.split_bf16(x.reshape(-1, C // 16, 16)).view(3, -1, C // 16, 16)
        return planes.permute(1, 2, 0, 3).contiguous()

    def pack_conv_wino_split(self, conv, m, interleaved=False, dgrad=False):
        """The transformed weights of pack_conv_wino as three bf16 planes (operands of csrc/xl_gemm_split.hip): separate
        planes, or - `interleaved` - [(m+2)^2][Cout][Cin/16][3][16] for the 256 x 256 kernel."""
        key = (id(conv.weight), "wino%d%s_split%s" % (m, "d" if dgrad else "", "_il" if interleaved else ""))
        if key not in self.packed_split:
            src = conv.weight.detach().to(device=self.device, dtype=torch.float32).contiguous()   # aliases the live parameter
            planes = torch.empty(3 * (m + 2) ** 2 * src.shape[0] * src.shape[1], dtype=torch.int16, device=self.device)
            self.packed_split[key] = (planes, src, m, interleaved, dgrad)
            self._pack_wino_split(*self.packed_split[key])
        return self.packed_split[key][0]

    def _pack_wino_split(self, planes, src, m, interleaved, dgrad=False):
        """U = G g G^T (float64 inside) and its exact three-term bf16 split in one HIP launch (csrc/xl_pack.hip); dgrad: the
        data-gradient operand [(m+2)^2][Cin][Cout] (flipped kernel, channel roles swapped)."""
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _check(_bind().xl_cnn_pack_wino_weight(src.data_ptr(), planes.data_ptr(), src.shape[0], src.shape[1], m,
                                               1 if dgrad else 0, 2 if interleaved else 1, stream))

    def _split_weight(self, planes, src, transposed=False):
        """fp32 [rows][K] (1x1; `transposed`: rows and K swapped) or OIHW 3x3 (K tap-major) -> interleaved bf16 planes, one
        HIP launch."""
        taps = 9 if (src.dim() == 4 and src.shape[2] == 3) else 1
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if transposed:
            _check(_bind().xl_cnn_split_weight(src.data_ptr(), planes.data_ptr(), src.shape[1], src.shape[0], 0, stream))
        else:
            _check(_bind().xl_cnn_split_weight(src.data_ptr(), planes.data_ptr(), src.shape[0], src.shape[1] * taps, taps, stream))


    # -- fp16 pair / triple operands (round 5, XL_CONV_PAIR_F16): half the matrix-pipe passes of the split-bf16 GEMMs
    def pair_ok(self):
        """The forward GEMMs of a plan run as three fp16 passes instead of six bf16 ones (csrc/xl_gemm_pair.hip) unless
        XL_GEMM_PAIR=0.  Every convolution this applies to reads GroupNorm outputs (and sums of them): their magnitude is
        bounded by the GroupNorm parameters, which is what makes ONE static power-of-two scale per plan safe for fp16."""
        # (training plans, round 5: the FORWARD GEMMs only - their operands are GroupNorm outputs like an inference plan's; the
        #  gradients the backward GEMMs read have no such bound and stay on the six-pass bf16 kernels.  XL_TRAIN_PAIR=0: off)
        if self.train and os.environ.get("XL_TRAIN_PAIR", "1") in ("", "0"):
            return False
        return (os.environ.get("XL_GEMM_PAIR", "1") not in ("", "0")
                and os.environ.get("XL_GEMM_SPLIT_BF16", self.SPLIT_DEFAULT) not in ("", "0", "1") and self.split_train_ok())

    def _gn_layers(self):
        """(gamma, beta, C, sqrt(N)) of every GroupNorm of the network: N = the elements of a group, from the op that applies
        the layer in this plan, or - layers the plan does not run - from the image size (no map is larger)."""
        seen = {}
        for op in self.ops:
            if op.type in (XL_OP_GN_FINAL, XL_OP_GN_APPLY) and op.w and op.bias and op.groups > 0:
                n = (op.Cin // op.groups) * op.Hi * op.Wi
                seen[op.w] = max(seen.get(op.w, 0), n)
        out = []
        for mod in self.net.modules():
            if not isinstance(mod, nn.GroupNorm):
                continue
            if mod.weight is None:
                # affine=False (no reference network has one): gamma = 1, beta = 0 in the bound - constants of the plan
                if not hasattr(self, "_unit_gn"):
                    self._unit_gn = {}
                if mod.num_channels not in self._unit_gn:
                    self._unit_gn[mod.num_channels] = (torch.ones(mod.num_channels, dtype=torch.float32, device=self.device),
                                                       torch.zeros(mod.num_channels, dtype=torch.float32, device=self.device))
                g, b = self._unit_gn[mod.num_channels]
                n = (mod.num_channels // mod.num_groups) * self.H * self.W
            else:
                # (not self.dev(): this runs with every weight refresh and must not grow the plan's keep list - the tensors alias the
                #  live parameters and are held by the caller's table)
                g = mod.weight.detach().to(device=self.device, dtype=torch.float32).contiguous()
                b = mod.bias.detach().to(device=self.device, dtype=torch.float32).contiguous()
                n = seen.get(g.data_ptr(), (mod.num_channels // mod.num_groups) * self.H * self.W)
            out.append((g, b, mod.num_channels, float(n) ** 0.5))
        return out

    def _update_pair_scales(self):
        """s = the largest power of two with s * sum over the GroupNorm layers of (sqrt(N) max|gamma| + max|beta|) <= 2^14:
        |gn(x)| <= sqrt(N - 1) |gamma| + |beta|, an activation is a GroupNorm output plus residuals that are activations
        themselves (the sum over ALL layers bounds any chain), and |B^T d B| <= 225 max|d| for F(6x6,3x3).  One tiny launch
        reading the live parameters - no host synchronisation, so a training loop can call it every step."""
        if not self.pair_ok():
            return
        # (the table holds raw device pointers of gamma / beta: rebuilt whenever a parameter's storage moved - net.to(),
        #  load_state_dict(assign=True) -, as _repack_pairs re-validates its own tables)
        layers = self._gn_layers()
        sig = tuple((g.data_ptr(), b.data_ptr()) for g, b, _, _ in layers)
        if getattr(self, "_pair_gn_sig", None) != sig:
            self._pair_gn_keep = [(g, b) for g, b, _, _ in layers]
            self._pair_gn_sig = sig
            n = len(layers)
            self._pair_gn = ((ctypes.c_void_p * n)(*[g.data_ptr() for g, _, _, _ in layers]),
                             (ctypes.c_void_p * n)(*[b.data_ptr() for _, b, _, _ in layers]),
                             (ctypes.c_int * n)(*[c for _, _, c, _ in layers]),
                             (ctypes.c_float * n)(*[r for _, _, _, r in layers]), n)
        g, b, c, r, n = self._pair_gn
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _check(_bind().xl_cnn_pair_scales(g, b, c, r, n, self.pair_scales.data_ptr(), stream))

    def pack_conv_wino_pair(self, conv, m, dgrad=False):
        """The transformed weights of pack_conv_wino as fp16 pairs {hi, lo} [(m+2)^2][Cout][Cin/16][2][16], each frequency scaled by its
        own power of two, + 2 (m+2)^2 floats (scratch, inverse scales).  dgrad: the data-gradient operand [(m+2)^2][Cin][Cout/16]..."""
        key = (id(conv.weight), "wino%d%s_pair" % (m, "d" if dgrad else ""))
        if key not in self.packed_pair:
            src = conv.weight.detach().to(device=self.device, dtype=torch.float32).contiguous()   # aliases the live parameter
            nf = (m + 2) ** 2
            planes = torch.empty(2 * nf * src.shape[0] * src.shape[1] + 4 * nf, dtype=torch.int16, device=self.device)
            self.packed_pair[key] = (planes, src, m, "d" if dgrad else "")
            self._pack_pair(*self.packed_pair[key])
        return self.packed_pair[key][0]

    def pack_conv_1x1_pair(self, conv, transposed=False):
        """[Cout][Cin/16][2][16] fp16 pairs {hi, lo} of a 1x1 convolution's weight (one power-of-two scale) + 2 floats;
        `transposed`: [Cin][Cout/16][2][16], the operand of its data gradient."""
        key = (id(conv.weight), "1x1_pair" + ("_t" if transposed else ""))
        if key not in self.packed_pair:
            src = conv.weight.detach().to(device=self.device, dtype=torch.float32).contiguous()   # aliases the live parameter
            planes = torch.empty(2 * src.numel() + 4, dtype=torch.int16, device=self.device)
            self.packed_pair[key] = (planes, src, 0, "t" if transposed else "")
            self._pack_pair(*self.packed_pair[key])
        return self.packed_pair[key][0]

    def pack_conv_stem_pair(self, conv):
        """[Cout][9 Cin / 16][2][16] fp16 pairs {hi, lo} of a 3x3 convolution's weight, K ordered tap-major, + 2 floats."""
        key = (id(conv.weight), "stem_pair")
        if key not in self.packed_pair:
            src = conv.weight.detach().to(device=self.device, dtype=torch.float32).contiguous()   # aliases the live parameter
            planes = torch.empty(2 * src.numel() + 4, dtype=torch.int16, device=self.device)
            self.packed_pair[key] = (planes, src, 0, "")
            self._pack_pair(*self.packed_pair[key])
        return self.packed_pair[key][0]

    def _pack_pair(self, planes, src, m, mode=""):
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if m:
            _check(_bind().xl_cnn_pack_wino_weight_pair(src.data_ptr(), planes.data_ptr(), src.shape[0], src.shape[1], m,
                                                        1 if mode == "d" else 0, stream))
        elif mode == "t":                             # [Cin][Cout]: rows and K swapped
            _check(_bind().xl_cnn_pair_weight(src.data_ptr(), planes.data_ptr(), src.shape[1], src.shape[0], 0, stream))
        elif src.dim() == 4 and src.shape[2] == 3:
            _check(_bind().xl_cnn_pair_weight(src.data_ptr(), planes.data_ptr(), src.shape[0], src.shape[1] * 9, 9, stream))
        else:
            _check(_bind().xl_cnn_pair_weight(src.data_ptr(), planes.data_ptr(), src.shape[0], src.shape[1], 1, stream))

    @staticmethod
    def _pair_item(planes, src, m, mode=""):
        """(list key, rows, K, kind) of a packed_pair entry for xl_cnn_repack_pairs - the arguments _pack_pair passes per matrix."""
        if m:
            return m, src.shape[0], src.shape[1], 1 if mode == "d" else 0
        if mode == "t":
            return 0, src.shape[1], src.shape[0], 0
        if src.dim() == 4 and src.shape[2] == 3:
            return 0, src.shape[0], src.shape[1] * 9, 9
        return 0, src.shape[0], src.shape[1], 1

    def _repack_pairs(self):
        """Every fp16-pair operand again (refresh_weights) in three launches per list - the F(6x6,3x3) layers, the F(4x4,3x3) ones,
        the plain matrices - instead of a memset and two launches per matrix (xl_cnn_repack_pairs; 24 + 23 matrices per step of
        the batch-16 training plan).  The device tables hold pointers: they are rebuilt when an entry or an address changed."""
        if not self.packed_pair:
            return
        if os.environ.get("XL_NO_BATCHED_REPACK"):
            for entry in self.packed_pair.values():
                self._pack_pair(*entry)
            return
        entries = list(self.packed_pair.values())
        sig = tuple((e[0].data_ptr(), e[1].data_ptr()) for e in entries)
        if getattr(self, "_pair_tables_sig", None) != sig:
            import numpy as np
            lists = {}
            for planes, src, m, mode in entries:
                key, rows, K, kind = self._pair_item(planes, src, m, mode)
                lists.setdefault(key, []).append((src.data_ptr(), planes.data_ptr(), rows, K, kind, 0))
            dt = PAIR_ITEM_DTYPE
            self._pair_tables = []
            for key, items in sorted(lists.items()):
                table = torch.from_numpy(np.array(items, dtype=dt).view(np.uint8).copy()).to(self.device)
                self._pair_tables.append((key, len(items), max(i[2] * i[3] for i in items), table))
            self._pair_tables_sig = sig
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for key, n, biggest, table in self._pair_tables:
            _check(_bind().xl_cnn_repack_pairs(table.data_ptr(), n, key, biggest, stream))

    def wino_pick(self, H, W, chan_max, allowed=(6, 4)):
        """Output tile m of F(m x m, 3x3) for an H x W map: the allowed size (capped by XL_WINOGRAD) with the fewest
        multiplies, (m+2)^2 * ceil(H/m) * ceil(W/m); 0 if none.  The transformed tensors V / M hold (m+2)^2 independent
        GEMM operands of [tiles][channels] each: the batched GEMM launch gives every one of them its own buffer
        descriptor (64-bit base, 32-bit offsets inside), so only ONE operand has to stay below 2 GiB, not the tensor."""
        want = int(os.environ.get("XL_WINOGRAD", "6"))
        cands = [m for m in allowed if m <= want]
        cands.sort(key=lambda m: ((m + 2) ** 2 * -(-H // m) * -(-W // m), -m))
        for m in cands:
            T = self.B * -(-H // m) * -(-W // m)
            if T * chan_max * 4 < 2 ** 31 - 1:
                return m
        return 0

    def wino_wgrad_ok(self, conv, H, W, C, m):
        """The weight gradient of this F(m x m,3x3) layer will be a Winograd one (the conditions of _lower_backward): it then
        reads the normalised V and never the layer's input tensor."""
        Cout = conv.out_channels
        # (the SAME conditions _lower_backward applies - including H * W >= 64 - and, when the apply is left to this
        #  layer, the backward pass takes the forward's tile size `wm` instead of re-deriving one: see `xnorm` there)
        return (not conv.weight.requires_grad) or (
            m in (4, 6) and C % 64 == 0 and Cout % 128 == 0 and H * W >= 64 and self.B * -(-H // m) * -(-W // m) >= 64
            and not os.environ.get("XL_NO_WINOGRAD") and not os.environ.get("XL_NO_WINOGRAD_TRAIN")
            and not os.environ.get("XL_NO_WINOGRAD_WGRAD"))

    def wino_dgrad_m(self, conv, H, W, C):
        """Data gradient of a stride-1 3x3 layer as F(m x m, 3x3) (C = the layer's input channels = gradient channels):
        the tile size, or 0 for the direct MODE 1 kernel."""
        if (conv.kernel_size[0] != 3 or conv.stride[0] != 1 or C not in (128, 256, 512, 1024)
                or conv.out_channels % 32 != 0 or H * W < 64
                or os.environ.get("XL_NO_WINOGRAD") or os.environ.get("XL_NO_WINOGRAD_TRAIN")):
            return 0
        return self.wino_pick(H, W, max(C, conv.out_channels))

    def pack_conv(self, conv, dgrad=False):
        w = conv.weight
        key = (id(w), dgrad)
        if key not in self.packed:
            src = w.detach().to(device=self.device, dtype=torch.float32).contiguous()     # aliases the live parameter
            cout, cin, k, _ = src.shape
            kind = "conv1" if (cin in (1, 3) and k == 3) else ("dgrad" if dgrad else "fwd")
            dst = torch.empty(src.numel(), dtype=torch.float32, device=self.device)
            self.packed[key] = (dst, src, kind)
            self._pack(dst, src, kind)
        return self.packed[key][0]

    def _pack(self, dst, src, kind):
        L = _bind()
        cout, cin, k, _ = src.shape
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if kind.startswith("wino"):                  # U = G g G^T in float64, rounded once: one HIP launch (csrc/xl_pack.hip)
            _check(L.xl_cnn_pack_wino_weight(src.data_ptr(), dst.data_ptr(), cout, cin, int(kind[4]),
                                             1 if kind.endswith("d") else 0, 0, stream))
        elif kind == "conv1":                     # [(ky*3+kx)*Cin + c][Cout]
            dst.view(k, k, cin, cout).copy_(src.permute(2, 3, 1, 0))
        elif kind == "dgrad":
            _check(L.xl_cnn_pack_conv_weight_dgrad(src.data_ptr(), dst.data_ptr(), cout, cin, k, stream))
        else:
            _check(L.xl_cnn_pack_conv_weight(src.data_ptr(), dst.data_ptr(), cout, cin, k, stream))

    def refresh_weights(self):
        """Parameters changed in place (optimizer step, load_state_dict): re-pack the conv operands.  Biases,
        GroupNorm affine parameters and fc3 are read through pointers to the live parameters."""
        for dst, src, kind in self.packed.values():
            self._pack(dst, src, kind)
        for entry in self.packed_split.values():
            self._pack_wino_split(*entry)
        for entry in self.packed_1x1.values():
            self._split_weight(*entry)
        self._repack_pairs()
        self._update_pair_scales()
        for key, (planes, src) in self.packed_c1.items():
            kind = key[1] if isinstance(key, tuple) else "c1"
            planes.copy_(self.conv2_pair_fragments(src) if kind == "c2pair" else      # (src: the conv module)
                         self.s2_dgrad_fragments(src) if kind == "s2dgrad" else self.conv2_fragments(src) if kind == "c2frag"
                         else self.conv1_fragments(src))

    def dev(self, p):
        t = p.detach().to(device=self.device, dtype=torch.float32).contiguous()
        self.keep.append(t)
        return t

    # -- op emitters; an activation is (tensor, H, W, C, ld, channel_offset)
    def norm_on_load_ok(self, act, conv):
        """The 1x1 forward conv kernel can apply the producer's GroupNorm(+ReLU) to its A operand while loading it (no
        separate apply pass): 128-row x 128-column tiles, whole 32-channel K-steps, at most two images per tile."""
        t, H, W, C, ld, off = act
        cout = conv.out_channels
        # (the tile-count term - small launches run the 64-row form, which has no operand normalisation - depends on the
        #  batch: batch-invariant plans make the choice from the layer alone, see split_1x1_ok; and every apply site, fused
        #  or not, computes fmaf(x, scale, shift), so the two forms agree to the bit anyway)
        fills = self.separate_stats or -(-self.B * H * W // 128) * (cout // 128) > 256
        return (conv.kernel_size[0] == 1 and conv.stride[0] == 1 and cout % 128 == 0 and C % 32 == 0 and H * W >= 128
                and fills and not self.train and not os.environ.get("XL_NO_NORM_ON_LOAD"))

    def split_1x1_ok(self, act, conv):
        """1x1 stride-1 layers of inference plans on the bf16 matrix pipe (csrc/xl_gemm_split.hip, split_conv1x1_kernel):
        weights split once on the host, activations split by the kernel on their way into LDS - fp32-accurate like the
        Winograd GEMMs.  The choice depends on the layer only, never on the batch: a frame's result must not change with
        the batch it is in (a single frame is 44 tiles of 256 x 256: one short round on 44 CUs, about the time the fp32
        kernel needs for its 340 small tiles)."""
        t, H, W, C, ld, off = act
        cout = conv.out_channels
        return (conv.kernel_size[0] == 1 and conv.stride[0] == 1 and self.split_train_ok() and C % 32 == 0 and cout % 256 == 0
                and cout <= 1024 and H * W >= 256 and ld % 4 == 0 and off % 4 == 0
                and os.environ.get("XL_GEMM_SPLIT_BF16", self.SPLIT_DEFAULT) not in ("", "0", "1")
                and not os.environ.get("XL_NO_SPLIT_1X1"))

    def wgrad_split_ok(self, C, Cout):
        """Weight gradients of 1x1 layers and of the batched Winograd products on the split pipe (256 x 256 tiles)."""
        return (C % 256 == 0 and Cout % 256 == 0 and self.split_train_ok()
                and os.environ.get("XL_GEMM_SPLIT_BF16", self.SPLIT_DEFAULT) not in ("", "0", "1")
                and not os.environ.get("XL_NO_SPLIT_WGRAD"))

    @staticmethod
    def wgrad_splits(tiles, K):
        """Split-K factor of the split-pipe weight gradient: fill the 256 CUs once, at least 256 rows of K per split."""
        return max(1, min(256 // max(tiles, 1), K // 256))

    def split_tile_form(self, M, N, Z=1, HW=1 << 30):
        """Tile form of a 1x1 layer / of the Z batched GEMMs of a Winograd layer on the split pipe, as the op's reserved_i:
        256 = 256 x 256 tiles (8 waves), 192 = 256 rows x 128 columns (8 waves), 128 = 128 x 128 (4 waves), 384 = 256 x 256
        for the full rounds + 256 x 128 for the last partial round (two launches over disjoint tile ranges).  The persistent
        kernels run one workgroup per CU, so a launch costs (rounds of 256 tiles) x (time of a tile); the tile times per
        K-step were measured at 60 x 90 (2.05 / 1.30 / 1.10 us).  At 47 frames the large tiles win everywhere; a single frame
        has 44 large tiles for a 1x1 layer (172 small ones: one round at half the tile time) and 128 for a Winograd layer
        (256 of the 256 x 128 form).  Every output element accumulates in the same order in all forms - the convolution
        results are bitwise the same - but the GroupNorm partial sums are per tile: batch-invariant plans keep 256."""
        if self.separate_stats or self.train or os.environ.get("XL_NO_SMALL_TILES"):
            return 256
        forced = os.environ.get("XL_TILE_FORM_WINO" if Z > 1 else "XL_TILE_FORM_1X1")      # measurement switch
        if forced:
            return int(forced) if (HW >= 128 or forced != "128") else 256
        big = -(-M // 256) * (N // 256) * Z
        cands = [(256, -(-big // 256) * 2.05), (192, -(-2 * big // 256) * 1.30)]
        if big > 256 and 0 < 2 * (big % 256) <= 256:     # 384: full rounds of large tiles, the rest as ONE round of 256 x 128
            cands.append((384, big // 256 * 2.05 + 1.30 + 0.15))          # (+ the second launch's pipeline fill)
        if HW >= 128:
            cands.append((128, -(-(-(-M // 128) * (N // 128) * Z) // 256) * 1.10))
        return min(cands, key=lambda c: (c[1], -c[0]))[0]

    def split_train_ok(self):
        """Training plans run their forward GEMMs (and the Winograd data gradients) on the split pipe too (round 3);
        XL_NO_SPLIT_TRAIN=1: fp32 MFMA throughout, the round-2 training plans."""
        return not self.train or not os.environ.get("XL_NO_SPLIT_TRAIN")

    def stem_split_ok(self, act, conv):
        """The stride-2 3x3 stem layers of inference plans on the bf16 matrix pipe (csrc/xl_stem_split.hip): a choice by
        layer, never by batch."""
        t, H, W, C, ld, off = act
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        return (conv.kernel_size[0] == 3 and conv.stride[0] == 2 and self.split_train_ok() and C in (32, 64, 128)
                and conv.out_channels in (64, 128, 256) and Ho * Wo >= 256 and ld % 4 == 0 and off % 4 == 0
                and 2 * H * W * ld * 4 < 2 ** 31 - 1
                and os.environ.get("XL_GEMM_SPLIT_BF16", self.SPLIT_DEFAULT) not in ("", "0", "1")
                and not os.environ.get("XL_NO_SPLIT_STEM"))

    def pack_conv_stem_split(self, conv):
        """[Cout][9 Cin / 16][3][16] bf16: the weight of a 3x3 convolution with K ordered tap-major (K = (3 ky + kx) Cin + c)
        as interleaved bf16 planes."""
        w = conv.weight
        key = id(w)
        if key not in self.packed_1x1:
            src = w.detach().to(device=self.device, dtype=torch.float32).contiguous()     # aliases the live parameter
            planes = torch.empty(3 * src.numel(), dtype=torch.int16, device=self.device)
            self.packed_1x1[key] = (planes, src, False)
            self._split_weight(planes, src)
        return self.packed_1x1[key][0]

    @staticmethod
    def _stem_rows(src):
        return src.permute(0, 2, 3, 1).reshape(src.shape[0], -1) if src.dim() == 4 and src.shape[2] == 3 else src.reshape(src.shape[0], src.shape[1])

    def pack_conv_1x1_split(self, conv, transposed=False):
        """[Cout][Cin/16][3][16] bf16: the weight of a 1x1 convolution as interleaved bf16 planes; `transposed`:
        [Cin][Cout/16][3][16], the operand of its data gradient."""
        w = conv.weight
        key = (id(w), "t") if transposed else id(w)
        if key not in self.packed_1x1:
            src = w.detach().to(device=self.device, dtype=torch.float32).contiguous()     # aliases the live parameter
            planes = torch.empty(3 * src.numel(), dtype=torch.int16, device=self.device)
            self.packed_1x1[key] = (planes, src, transposed)
            self._split_weight(planes, src, transposed)
        return self.packed_1x1[key][0]

    _conv1_fragment_index = {}                   # device -> gather indices of conv1_fragments

    @staticmethod
    def conv1_fragments(weight):
        """[3 planes][3 window rows][64 lanes][8] bf16 (int16 storage): the MFMA weight fragments of conv1_mfma_kernel.  Lane =
        32 * K-half + output channel; slot i of a lane = (dx = 2 * K-half + i // 4, c = i % 4), zero for c = 3 and dx = 3."""
        w = weight.detach().to(torch.float32)                                    # [32][3][ky][kx]
        # one gather through an index table (a training loop rebuilds the fragments after every step): slot (dy, K-half, channel,
        # i) <- w[channel][c][dy][dx], or the zero appended behind the weights
        idx = _Plan._conv1_fragment_index.get(w.device)
        if idx is None:
            host = torch.full((3, 2, 32, 8), 32 * 27, dtype=torch.int64)
            for dy in range(3):
                for kh in range(2):
                    for i in range(8):
                        dx, c = 2 * kh + i // 4, i % 4
                        if c < 3 and dx < 3:
                            host[dy, kh, :, i] = torch.arange(32) * 27 + c * 9 + dy * 3 + dx
            idx = _Plan._conv1_fragment_index[w.device] = host.reshape(-1).to(w.device)
        f = torch.cat([w.reshape(-1), w.new_zeros(1)])[idx]                      # [dy][K-half][channel][slot]
        planes = _Plan.split_bf16(f.reshape(3, 64, 8))                           # [3 planes][3 dy][64][8]
        return planes.contiguous()

    @staticmethod
    def conv2_fragments(weight):
        """[18 K-steps][3 planes][2 column blocks][64 lanes][8] bf16 (int16 storage): the MFMA weight fragments of stem12_kernel
        (csrc/xl_stem_fused.hip) for the 32 -> 64 3x3 stride-2 layer.  K = tap * 32 + channel (tap = 3 ky + kx), a K-step = 16
        channels of one tap; lane = 32 * K-half + output channel within the 32-channel block; a lane's 8 values are consecutive K."""
        w = weight.detach().to(torch.float32)                                    # [64][32][ky][kx]
        rows = w.permute(0, 2, 3, 1).reshape(64, 288)
        planes = _Plan.split_bf16(rows).view(3, 2, 32, 18, 2, 8)                 # [plane][j][fr][kk][kh][8]
        return planes.permute(3, 0, 1, 4, 2, 5).contiguous()                     # [kk][plane][j][kh][fr][8]

    @staticmethod
    def s2_dgrad_fragments(weight):
        """[9 taps][CO/16][3 planes][CI/32][64 lanes][8] bf16 (int16 storage): the MFMA weight fragments of s2_dgrad_kernel
        (csrc/xl_stem_dgrad.hip) for a stride-2 3x3 layer with weight [CO][CI][3][3]: rows = the layer's INPUT channels (the
        gradient's channels), K = its output channels; lane = 32 * K-half + row within the 32-row block."""
        w = weight.detach().to(torch.float32)
        CO, CI = w.shape[0], w.shape[1]
        wt = w.permute(2, 3, 1, 0).reshape(9, CI, CO)                            # [tap][ci][co]
        planes = _Plan.split_bf16(wt).view(3, 9, CI // 32, 32, CO // 16, 2, 8)   # [plane][tap][j][fr][c][kh][8]
        return planes.permute(1, 4, 0, 2, 5, 3, 6).contiguous()                  # [tap][c][plane][j][kh][fr][8]

    def pack_s2_dgrad_fragments(self, conv):
        key = (id(conv.weight), "s2dgrad")
        if key not in self.packed_c1:
            src = conv.weight.detach().to(device=self.device, dtype=torch.float32).contiguous()   # aliases the live parameter
            self.packed_c1[key] = (self.s2_dgrad_fragments(src), src)
        return self.packed_c1[key][0]

    def conv2_pair_fragments(self, conv):
        """[18 K-steps][2 planes {hi, lo}][2 column blocks][64 lanes][8] fp16 + the inverse weight scale (one float, then padding
        to 16 bytes): the MFMA weight fragments of stem12_kernel<.., true> - the pair form of the 32 -> 64 stride-2 layer, cut
        from the operand xl_cnn_pair_weight packs for pair_conv3x3s2_kernel (same scale, same split: the fused and the
        two-kernel stem stay bitwise equal).  (int16 storage)"""
        packed = self.pack_conv_stem_pair(conv)                                  # (a refresh re-packs packed_pair first: _repack_pairs)
        n = 64 * 288
        frag = packed[:2 * n].view(2, 32, 18, 2, 2, 8).permute(2, 3, 0, 4, 1, 5).contiguous().reshape(-1)   # [kk][p][j][kh][fr][8]
        tail = torch.zeros(8, dtype=torch.int16, device=self.device)
        tail[:2] = packed[2 * n + 2:2 * n + 4]                                   # the inverse scale (behind the maxima word)
        return torch.cat([frag, tail])

    def pack_conv2_pair_fragments(self, conv):
        key = (id(conv.weight), "c2pair")
        if key not in self.packed_c1:
            self.packed_c1[key] = (self.conv2_pair_fragments(conv), conv)
        return self.packed_c1[key][0]

    def pack_conv2_fragments(self, conv):
        key = (id(conv.weight), "c2frag")
        if key not in self.packed_c1:
            src = conv.weight.detach().to(device=self.device, dtype=torch.float32).contiguous()   # aliases the live parameter
            self.packed_c1[key] = (self.conv2_fragments(src), src)
        return self.packed_c1[key][0]

    def pack_conv1_split(self, conv):
        key = id(conv.weight)
        if key not in self.packed_c1:
            src = conv.weight.detach().to(device=self.device, dtype=torch.float32).contiguous()   # aliases the live parameter
            self.packed_c1[key] = (self.conv1_fragments(src), src)
        return self.packed_c1[key][0]

    def conv(self, act, conv, out=None, out_ld=None, out_off=0, norm_in=None, split=False):
        t, H, W, C, ld, off = act
        k, s = conv.kernel_size[0], conv.stride[0]
        cout = conv.out_channels
        Ho = (H + 2 * (k // 2) - k) // s + 1
        Wo = (W + 2 * (k // 2) - k) // s + 1
        if out is None:
            out = self.alloc(self.B * Ho * Wo * cout)
            out_ld = cout
        op = XlOp()
        op.type = XL_OP_CONV
        op.B, op.Hi, op.Wi, op.Cin, op.Ho, op.Wo, op.Cout = self.B, H, W, C, Ho, Wo, cout
        op.ksize, op.stride, op.ld_in, op.ld_out = k, s, ld, out_ld
        op.in_ = t.data_ptr() + 4 * off
        if not split:
            op.w = self.pack_conv(conv).data_ptr()
        op.bias = self.dev(conv.bias).data_ptr()
        op.out = out.data_ptr() + 4 * out_off
        # 64-row tiles when 128-row tiles would not even fill one wave of workgroups over the 256 CUs
        bn = 128 if cout % 128 == 0 else 64
        if -(-self.B * Ho * Wo // 128) * -(-cout // bn) <= 256 and norm_in is None:
            op.reserved_i = 64                        # (the normalise-on-load form exists with 128-row tiles only)
        if split and k == 3:                          # stride-2 stem layer on the split pipe (no statistics epilogue)
            op.flags |= CONV_SPLIT_BF16 | CONV_SPLIT_IL
            if (self.pair_ok() and (norm_in is not None or (self.train and os.environ.get("XL_TRAIN_PAIR_STEM", "1") not in ("", "0")))
                    and not os.environ.get("XL_NO_PAIR_STEM")):
                # round 5: fp16 pairs, three passes (csrc/xl_stem_pair.hip); the operand is a GroupNorm output normalised on load.
                # Training plans (materialised GroupNorm + ReLU outputs: the same bound holds) run them too since round 6
                # (XL_TRAIN_PAIR_STEM=0: the six-pass kernels; -0.2 ms of a 32.4 ms step.  Round 5 left them off because one small-map
                # test counted ReLU-kink flips as errors: tests/test_semantics_gpu.py now uses the criterion of the other gradient tests)
                op.flags |= CONV_PAIR_F16
                op.w = self.pack_conv_stem_pair(conv).data_ptr()
                op.scale = self.pair_scales.data_ptr()
            else:
                op.w = self.pack_conv_stem_split(conv).data_ptr()
            op.reserved_i = 0
            if (cout == 256 and -(-self.B * Ho * Wo // 256) < 128 and not os.environ.get("XL_NO_SMALL_TILES")):
                op.reserved_i = 128                  # latency form: 128 x 128 tiles when 256-row tiles leave the chip idle
        elif split and self.pair_ok():
            # round 5: three fp16 passes instead of six bf16 ones; the operand is a GroupNorm output (or normalised on load)
            op.flags |= CONV_SPLIT_BF16 | CONV_SPLIT_IL | CONV_PAIR_F16
            op.w = self.pack_conv_1x1_pair(conv).data_ptr()
            op.scale = self.pair_scales.data_ptr()
            op.reserved_i = -256 if self.separate_stats else self.split_tile_form(self.B * Ho * Wo, cout, 1, Ho * Wo)
        elif split:
            op.flags |= CONV_SPLIT_BF16 | CONV_SPLIT_IL
            op.w = self.pack_conv_1x1_split(conv).data_ptr()
            # rows per tile (the statistics epilogue writes one entry per tile); negative: tiles start at image boundaries,
            # so the grouping of the partial sums does not depend on where a frame sits in the batch (batch-invariant plans)
            op.reserved_i = -256 if self.separate_stats else self.split_tile_form(self.B * Ho * Wo, cout, 1, Ho * Wo)
        if norm_in is not None:                       # the producer's deferred GroupNorm apply, folded into the operand load
            op.flags |= CONV_NORM_IN | (CONV_NORM_RELU if norm_in.flags & GN_RELU_IN else 0)
            if norm_in.flags & GN_ADD:                # ... + residual + ReLU (XL_CONV_NORM_ADD, split 1x1 kernel only)
                assert split and k == 1 and (norm_in.flags & GN_RELU_OUT) and (norm_in.flags & GN_RELU_IN)
                op.flags |= CONV_NORM_ADD
                op.aux, op.ld_aux = norm_in.aux, norm_in.ld_aux
            if self.train:
                op.aux2 = norm_in.aux2                # (training plans: the producer's own coefficient table)
            else:
                self.deferred_gn_consumers = getattr(self, "deferred_gn_consumers", []) + [len(self.ops)]
        self.ops.append(op)
        res = (out, Ho, Wo, cout, out_ld, out_off)
        self.tape.append(dict(kind="conv", conv=conv, x=act, raw=res, xnorm=norm_in if self.train else None))
        return res

    def gn(self, act, norm, flags, aux=None, out=None, pre_stats=None, stat_tile=0, defer=False, share=False, stat_mult=1):
        """GroupNorm (+fused epilogue) of `act`; in place unless `out` (tensor, ld, off) is given or training.
        pre_stats = (stats tensor, nchunks): the partial sums were already produced (Winograd output transform or conv
        epilogue of a training plan; stat_tile = rows per conv tile in the latter case), no statistics pass is emitted.
        Training plans keep one coefficient table per layer ({scale, shift} and {mean, rstd} per image and channel,
        written by GN_FINAL): the apply pass and the three backward passes read it instead of re-reducing the partial
        sums in the prologue of every workgroup."""
        t, H, W, C, ld, off = act
        G = norm.num_groups
        HW = H * W
        nchunks = max(1, min(128, (HW + 255) // 256)) if pre_stats is None else pre_stats[1]
        st = XlOp()
        st.type = XL_OP_GN_STATS
        st.B, st.Hi, st.Wi, st.Cin, st.groups, st.nchunks, st.ld_in = self.B, H, W, C, G, nchunks, ld
        st.in_ = t.data_ptr() + 4 * off
        ap = XlOp()
        ap.type = XL_OP_GN_APPLY
        ap.B, ap.Hi, ap.Wi, ap.Cin, ap.groups, ap.nchunks, ap.ld_in = self.B, H, W, C, G, nchunks, ld
        ap.flags, ap.eps = flags, norm.eps
        ap.in_ = t.data_ptr() + 4 * off
        gamma, beta = self.dev(norm.weight), self.dev(norm.bias)
        ap.w, ap.bias = gamma.data_ptr(), beta.data_ptr()
        stats_t = None
        if pre_stats is not None:
            assert self.train
            stats_t = pre_stats[0]
            ap.stats = stats_t.data_ptr()
        elif self.train:                    # the forward statistics are inputs of the backward pass: keep them
            stats_t = torch.zeros(self.B * nchunks * G * 2, dtype=torch.float64, device=self.device)
            self.keep.append(stats_t)
            st.stats = ap.stats = stats_t.data_ptr()
        else:
            self.max_stats = max(self.max_stats, self.B * nchunks * G * 2)
            self.stats_ops += [len(self.ops), len(self.ops) + 1]
        if pre_stats is None:
            self.ops.append(st)
        table = None
        if self.train:
            table = torch.zeros(self.B * C * 4, dtype=torch.float32, device=self.device)
            self.keep.append(table)
        self._emit_final(ap, gamma, beta, stat_tile, table, mult=stat_mult)
        if aux is not None:
            ap.aux = aux[0].data_ptr() + 4 * aux[5]
            ap.ld_aux = aux[4]
        # round 4, training plans: a GroupNorm + ReLU whose only consumer is a convolution that can apply it while loading its
        # operand (an F(m x m,3x3) layer: input transform, the normalised V is kept for the weight gradient; a 1x1 layer on the
        # split pipe: forward and weight-gradient kernels normalise on load) is NOT materialised: no apply pass, no activation
        # tensor.  Its backward pass needs the raw conv output and the coefficient table only.  cgr() materialises it after all
        # when the consumer turns out not to be able to (XL_NO_TRAIN_DEFER=1: never deferred).
        train_defer = (self.train and defer and out is None and flags == GN_RELU_IN and aux is None and not self.separate_stats
                       and HW % 8 == 0 and not os.environ.get("XL_NO_TRAIN_DEFER"))
        if out is None and self.train and not train_defer:
            out = (self.alloc(self.B * HW * C), C, 0)
        if out is None:
            ap.out, ap.ld_out = ap.in_, ld
            res = act
        else:
            ot, old, ooff = out
            ap.out, ap.ld_out = ot.data_ptr() + 4 * ooff, old
            res = (ot, H, W, C, old, ooff)
        if not self.train and res is act:
            if defer:                                 # the only consumer applies it while loading its operand
                if not hasattr(self, "pending_gn"):
                    self.pending_gn = {}
                self.pending_gn[self._act_key(res)] = ap
                return res
            if share and self.fold_ok():              # ... or the first of several consumers does, and materialises it
                return self._fold_begin(ap, act, aux)
        if train_defer:
            entry = dict(kind="gn", norm=norm, raw=act, out=res, aux=None, flags=flags, table=table, gamma=gamma, beta=beta)
            self.tape.append(entry)
            if not hasattr(self, "pending_gn"):
                self.pending_gn, self.pending_entry = {}, {}
            self.pending_gn[self._act_key(res)] = ap
            self.pending_entry = getattr(self, "pending_entry", {})
            self.pending_entry[self._act_key(res)] = entry
            return res
        aux_ap = self._aux_take(aux)
        if aux_ap is not None:
            self._aux_apply(aux_ap)
        self.ops.append(ap)
        self.tape.append(dict(kind="gn", norm=norm, raw=act, out=res, aux=aux, flags=flags, table=table,
                              gamma=gamma, beta=beta))
        return res

    def _train_materialise(self, pend, act):
        """A GroupNorm apply deferred in a training plan whose consumer cannot apply it on load: run it as a pass into a buffer of
        its own (the raw conv output stays: the backward pass reads it) and continue with that activation."""
        t, H, W, C, ld, off = act
        out = self.alloc(self.B * H * W * C)
        pend.out, pend.ld_out = out.data_ptr(), C
        self.ops.append(pend)
        res = (out, H, W, C, C, 0)
        entry = getattr(self, "pending_entry", {}).pop(self._act_key(act), None)
        if entry is not None:
            entry["out"] = res
        return res

    def wino_tile(self, act, conv):
        """Stride-1 3x3 convolutions run as Winograd F(m x m, 3x3).  Returns the output tile size m, or 0 for the direct
        kernel.  Inference plans choose between F(6x6,3x3) (64 multiplies per 36 outputs) and F(4x4,3x3) (36 per 16)
        by the number of multiplies the feature map needs with each tiling - (m+2)^2 * ceil(H/m) * ceil(W/m): 9600 vs
        12420 per channel pair at 60x90, where 6 divides both sides, but F(4x4) wins on small maps with ragged 6x6
        tiles.  XL_WINOGRAD=4 / 2 forces F(4x4,3x3) / F(2x2,3x3) (the latter for inference only).  Training plans make the
        same choice, for the forward pass and for both gradients."""
        t, H, W, C, ld, off = act
        if (conv.kernel_size[0] != 3 or conv.stride[0] != 1 or C % 32 != 0 or H * W < 64
                or conv.out_channels not in (128, 256, 512, 1024) or os.environ.get("XL_NO_WINOGRAD")):
            return 0
        want = int(os.environ.get("XL_WINOGRAD", "6"))
        if self.train and (os.environ.get("XL_NO_WINOGRAD_TRAIN") or want not in (4, 6)):
            return 0
        if want == 2:
            return 2 if not (H % 2 or W % 2) and self.B * (H // 2) * (W // 2) * max(C, conv.out_channels) * 4 < 2 ** 31 - 1 else 0
        return self.wino_pick(H, W, max(C, conv.out_channels))

    def wino_gemm_form(self, C, cout, m, T):
        """(split, split_il, split_act) of the GEMMs of an F(m x m,3x3) layer with T tiles.  XL_GEMM_SPLIT_BF16: "il" =
        interleaved planes + 256 x 256 persistent kernels, "1" = separate planes + 128 x 128 register-staged kernel (the
        first form), "0" = fp32 MFMA.  split_act (round 3): V stays fp32 in HBM (4 bytes per element instead of 6, written
        once and read once) and the GEMM kernel splits it on its way into LDS, like the activations of a 1x1 layer
        (XL_CONV_SPLIT_ACT); XL_WINO_V_SPLIT=1: the round-2 form, V written as interleaved bf16 planes by the input transform."""
        nf = (m + 2) ** 2
        mode = os.environ.get("XL_GEMM_SPLIT_BF16", self.SPLIT_DEFAULT)
        split = (mode not in ("", "0") and C % 32 == 0 and self.split_train_ok())
        split_il = split and mode != "1" and (T + 256) * max(C * 6, cout * 4) < 2 ** 31 - 1 and C % 128 == 0 and cout % 256 == 0
        split_act = split_il and cout <= 1024 and not os.environ.get("XL_WINO_V_SPLIT")
        if m != 6 or self.train:
            # the forms that read V as bf16 planes exist for F(6x6,3x3) inference layers only (wino6_in_kernel writes them);
            # the form that splits an fp32 V inside the GEMM does not care about the tile size, and leaves V as the weight
            # gradient of a training plan wants it
            split = split_il = split_act
        if split and not split_il:
            split = nf * T * max(C, cout) * 6 < 2 ** 31 - 1             # (the first form addresses a plane as a whole)
        return split, split_il, split_act

    def conv_wino(self, act, conv, norm, flags, aux, m, deferred=None, defer=False, fold=None, share=False, dst=None):
        """conv3x3 + GroupNorm(+epilogue) as F(m x m, 3x3): input transform, (m+2)^2 GEMMs in one batched launch, output
        transform that also emits the GroupNorm partial sums, GN_FINAL, GN_APPLY (in place; `dst` = (tensor, ld, channel
        offset): the apply writes the activation there instead - an encoder's last layer into its slice of the concat buffer)."""
        t, H, W, C, ld, off = act
        B, cout = self.B, conv.out_channels
        Th, Tw = -(-H // m), -(-W // m)
        T = B * Th * Tw
        nf = (m + 2) ** 2
        # opt-in: the GEMMs on the bf16 matrix pipe with every fp32 operand split into three bf16 terms (fp32-accurate)
        # the GEMMs on the bf16 matrix pipe with every fp32 operand split into three bf16 terms (fp32-accurate).
        # XL_GEMM_SPLIT_BF16: "il" = interleaved planes + 256 x 256 persistent kernel, "1" = separate planes + 128 x 128
        # register-staged kernel (the first form), "0" = fp32 MFMA
        split, split_il, split_act = self.wino_gemm_form(C, cout, m, T)
        assert fold is None or not split or split_act
        V = self.alloc(nf * T * C * 3 // 2 if (split and not split_act) else nf * T * C)
        op = XlOp()
        op.type = XL_OP_WINO_IN
        op.ksize = m
        op.B, op.Hi, op.Wi, op.Cin, op.Ho, op.Wo, op.ld_in = B, H, W, C, Th, Tw, ld
        op.in_, op.out = t.data_ptr() + 4 * off, V.data_ptr()
        # round 5: fp16 pairs (XL_CONV_PAIR_F16).  With full 256 x 256 tiles V is written as pairs by the input transform and both
        # operands of the GEMM arrive by DMA (pair_gemm_kernel); the small-batch tile forms keep V in fp32 and form the pairs in
        # the GEMM (pair_conv1x1_kernel with XL_CONV_SPLIT_ACT)
        pair = split_act and m == 6 and self.pair_ok()
        tile_form = self.split_tile_form(T, cout, nf) if split_act else 0
        # (training plans keep V in fp32: it is the left operand of the Winograd weight gradient)
        pair_dma = pair and tile_form == 256 and not self.train and not os.environ.get("XL_PAIR_NO_DMA")
        if pair_dma:
            op.flags = CONV_PAIR_F16
            op.scale = self.pair_scales.data_ptr() + 8
        if split and not split_act:
            op.flags = CONV_SPLIT_BF16 | (CONV_SPLIT_IL if split_il else 0)
        if deferred is not None:                      # the producer's GroupNorm(+ReLU) is applied while gathering
            op.flags |= deferred.flags
            if self.train:
                op.aux2 = deferred.aux2               # (training plans: the producer's own coefficient table)
            else:
                self.deferred_gn_consumers = getattr(self, "deferred_gn_consumers", []) + [len(self.ops)]
        if fold is not None:                          # ... and, fold: the activation `act` is written by this transform
            fap, raw = fold["ap"], fold["raw"]
            op.in_, op.ld_in = raw[0].data_ptr() + 4 * raw[5], raw[4]
            op.flags |= fap.flags & (GN_RELU_IN | GN_ADD | GN_RELU_OUT)
            op.out2, op.ld_out = t.data_ptr() + 4 * off, ld
            if fap.flags & GN_ADD:
                op.aux, op.ld_aux = fap.aux, fap.ld_aux
                if fold.get("aux_ap") is not None:    # the residual is a raw conv output: its {scale, shift} pairs in `w`
                    self.aux_coef_consumers = getattr(self, "aux_coef_consumers", []) + [len(self.ops)]
            self.deferred_gn_consumers = getattr(self, "deferred_gn_consumers", []) + [len(self.ops)]
        self.ops.append(op)
        if fold is not None:
            self._fold_end(fold)
        Mb = self.alloc(nf * T * cout)
        op = XlOp()
        op.type = XL_OP_CONV
        op.B, op.Hi, op.Wi, op.Cin, op.Ho, op.Wo, op.Cout = B, Th, Tw, C, Th, Tw, cout
        op.ksize, op.stride, op.ld_in, op.ld_out, op.nchunks2 = 1, 1, C, cout, nf
        op.in_, op.out = V.data_ptr(), Mb.data_ptr()
        # XL_WINO_M_TILE_MAJOR=1: the product M as [tiles][64][C], so that the block the output transform reads per tile is one
        # contiguous piece.  Measured at 47 frames: output transforms 3.21 -> 3.05 ms per step, GEMM epilogues +0.17 ms: no net
        # gain, so [64][tiles][C] (what every other form reads and writes) stays the default
        m_tile_major = CONV_M_TILE_MAJOR if (split_act and m == 6 and os.environ.get("XL_WINO_M_TILE_MAJOR") and not pair_dma) else 0
        if split:
            op.flags = (CONV_SPLIT_BF16 | (CONV_SPLIT_IL if split_il else 0) | (CONV_SPLIT_ACT if split_act else 0)
                        | m_tile_major)
            if pair:
                op.flags |= CONV_PAIR_F16
                if pair_dma:
                    op.flags &= ~CONV_SPLIT_ACT
                op.w = self.pack_conv_wino_pair(conv, m).data_ptr()
                op.scale = self.pair_scales.data_ptr() + 8
            else:
                op.w = self.pack_conv_wino_split(conv, m, split_il).data_ptr()
        else:
            op.w = self.pack_conv_wino(conv, m).data_ptr()
        if -(-T // 128) * (cout // 128) * nf <= 256:
            op.reserved_i = 64
        if split_act:
            op.reserved_i = tile_form
        self.ops.append(op)
        self.wino_gemm_indices = getattr(self, "wino_gemm_indices", []) + [len(self.ops) - 1]
        self.release(V)
        out = self.alloc(B * H * W * cout)
        G = norm.num_groups
        tpb = 32 if m == 2 else 16
        zblocks = max(1, cout // (256 if m == 6 else 512))          # channel blocks of the output-transform grid
        while tpb > 1 and B * -(-(Th * Tw) // tpb) * zblocks < 1024:
            tpb //= 2                                # small batches: more, shorter workgroups (latency-bound otherwise)
        if m == 6 and cout % 512 == 0 and not os.environ.get("XL_WINO_OUT_TPB16"):
            # the two-channels-per-lane form: 2 workgroups of 4 waves are resident per CU (230 VGPRs), so a launch costs
            # (rounds of 512 workgroups) x (tiles per workgroup + a workgroup's start-up, ~half a tile).  47 frames of 150
            # tiles: 15 tiles per workgroup = 470 workgroups of equal length in ONE round (16: nine chunks of 16 and one of 6
            # per image - the round takes 16 tile-times for 13.8 tiles of work per slot)
            zb = cout // 512
            tpb = min(range(1, 17), key=lambda t: (-(-(B * -(-(Th * Tw) // t) * zb) // 512) * (t + 0.5), -t))
        if os.environ.get("XL_WINO_OUT_TPB"):
            tpb = int(os.environ["XL_WINO_OUT_TPB"])
        if self.separate_stats:
            tpb = 4          # batch-invariant plans: the grouping of the partial sums must not depend on the batch size
        nchunks = -(-(Th * Tw) // tpb)
        op = XlOp()
        op.type = XL_OP_WINO_OUT
        op.ksize = m
        op.B, op.Hi, op.Wi, op.Cin, op.ld_out, op.groups, op.nchunks, op.reserved_i = B, H, W, cout, cout, G, nchunks, tpb
        op.in_, op.out = Mb.data_ptr(), out.data_ptr()
        op.flags = m_tile_major
        op.bias = self.dev(conv.bias).data_ptr()
        y = (out, H, W, cout, cout, 0)
        if self.train:
            # the raw conv output and its statistics are inputs of the backward pass: keep both, record the tape
            stats_t = torch.zeros(B * nchunks * G * 2, dtype=torch.float64, device=self.device)
            self.keep.append(stats_t)
            op.stats = stats_t.data_ptr()
            self.ops.append(op)
            self.free.setdefault(Mb.numel(), []).append(Mb)          # M is scratch even in training plans
            # V = B^T x B is also the left operand of the Winograd weight gradient: keep it (407 MB per 512-channel
            # layer at batch 16) instead of transforming the input again, within a fixed budget
            kept_v = None
            self.kept_v_bytes = getattr(self, "kept_v_bytes", 0)
            if (m in (4, 6) and conv.weight.requires_grad and self.kept_v_bytes + 4 * V.numel() <= (16 << 30)
                    and not os.environ.get("XL_NO_KEEP_V") and not os.environ.get("XL_NO_WINOGRAD_WGRAD")):
                kept_v = V
                self.kept_v_bytes += 4 * V.numel()
            else:
                self.free.setdefault(V.numel(), []).append(V)
            self.tape.append(dict(kind="conv", conv=conv, x=act, raw=y, v=kept_v, wm=m, xnorm=deferred))
            return self.gn(y, norm, flags, aux, pre_stats=(stats_t, nchunks), out=dst, defer=defer)
        self.max_stats = max(self.max_stats, B * nchunks * G * 2)
        self.stats_ops.append(len(self.ops))
        self.ops.append(op)
        self.release(Mb)
        ap = XlOp()
        ap.type = XL_OP_GN_APPLY
        ap.B, ap.Hi, ap.Wi, ap.Cin, ap.groups, ap.nchunks, ap.ld_in = B, H, W, cout, G, nchunks, cout
        ap.flags, ap.eps = flags, norm.eps
        ap.in_ = out.data_ptr()
        gamma, beta = self.dev(norm.weight), self.dev(norm.bias)
        ap.w, ap.bias = gamma.data_ptr(), beta.data_ptr()
        self._emit_final(ap, gamma, beta, 0)
        if aux is not None:
            ap.aux = aux[0].data_ptr() + 4 * aux[5]
            ap.ld_aux = aux[4]
        ap.out, ap.ld_out = ap.in_, cout
        if dst is not None:                            # the apply is a pass of its own: raw conv output -> the destination slice
            ot, old, ooff = dst
            ap.out, ap.ld_out = ot.data_ptr() + 4 * ooff, old
            aux_ap = self._aux_take(aux)
            if aux_ap is not None:
                self._aux_apply(aux_ap)
            self.stats_ops.append(len(self.ops))
            self.ops.append(ap)
            self.release(out)
            return (ot, H, W, cout, old, ooff)
        if defer and flags == GN_RELU_IN and aux is None and not os.environ.get("XL_NO_DEFERRED_GN"):
            # the only consumer applies it while loading its operand (a 1x1 conv: norm_on_load_ok; or a Winograd transform)
            if not hasattr(self, "pending_gn"):
                self.pending_gn = {}
            self.pending_gn[self._act_key(y)] = ap
            return y
        if share and self.fold_ok():
            return self._fold_begin(ap, y, aux)
        aux_ap = self._aux_take(aux)
        if aux_ap is not None:
            self._aux_apply(aux_ap)
        self.stats_ops.append(len(self.ops))
        self.ops.append(ap)
        return y

    def fold_ok(self):
        return not self.train and not os.environ.get("XL_NO_DEFERRED_GN") and not os.environ.get("XL_NO_FOLD_GN")

    def cgr(self, act, conv, norm, flags=GN_RELU_IN, aux=None, defer=False, share=False, out=None, defer_add=False):
        """conv -> GroupNorm -> epilogue.  `defer`: the caller promises that the next cgr() is the only consumer of the
        result; when that consumer is an F(4x4,3x3) layer its input transform applies the normalisation and the separate
        GN_APPLY pass (one read + one write of the activation) disappears.  `out` = (tensor, ld, channel offset): the
        activation is written there (an encoder's last layer into its slice of the MLR concat buffer)."""
        pend = getattr(self, "pending_gn", {}).pop(self._act_key(act), None)
        fold = getattr(self, "pending_fold", {}).pop(self._act_key(act), None)
        m = self.wino_tile(act, conv)
        if out is not None and not m:                  # no Winograd form for this layer: direct conv, then the apply into `out`
            if pend is not None and self.train:
                act = self._train_materialise(pend, act)
            elif pend is not None:
                self.stats_ops.append(len(self.ops))
                self.ops.append(pend)
            if fold is not None:
                self._fold_materialise(fold, act)
            y = self.conv(act, conv)
            r = self.gn(y, norm, flags, aux, out=out)
            if r[0] is not y[0]:
                self.release(y[0])
            return r
        stem = self.stem_split_ok(act, conv)
        cpg = conv.out_channels // norm.num_groups
        # a 1x1 layer on the split pipe applies a pending GroupNorm on load at ANY batch size (the tile-count condition of
        # norm_on_load_ok belongs to the fp32 kernel's 64-row form)
        # (the statistics epilogue of the split kernel sums 16-channel groups; a layer with other groups - res1_conv2: 8 per
        #  group - still runs on the split pipe in inference plans, followed by a statistics pass over its output)
        split_1x1 = self.split_1x1_ok(act, conv) and (cpg == 16 or self.separate_stats or not self.train)
        absorbs = (split_1x1 and act[3] <= 512 and act[1] * act[2] >= 256 and not os.environ.get("XL_NO_NORM_ON_LOAD"))
        if pend is not None and self.train:
            # training plans: absorbed by a Winograd layer (V is kept normalised) or by a 1x1 layer whose forward AND
            # weight-gradient kernels normalise on load; anything else gets the activation materialised
            t_ok = (m in (4, 6) and self.wino_wgrad_ok(conv, act[1], act[2], act[3], m)) or \
                   (split_1x1 and absorbs and cpg == 16 and self.wgrad_split_ok(act[3], conv.out_channels)
                    and act[4] % 4 == 0 and act[5] % 4 == 0)
            if not t_ok:
                act = self._train_materialise(pend, act)
                pend = None
            else:
                getattr(self, "pending_entry", {}).pop(self._act_key(act), None)
        pend_add = pend is not None and bool(pend.flags & GN_ADD)
        if pend is not None and not self.train and ((m not in (4, 6) and not self.norm_on_load_ok(act, conv) and not stem and not absorbs)
                                                    or (pend_add and not absorbs)):
            self.stats_ops.append(len(self.ops))       # consumer cannot absorb it: materialise now
            self.ops.append(pend)
            pend = None
        held_res = None
        if pend_add:                                   # the residual is dead once the consumer (or the apply pass) is emitted
            held_res = getattr(self, "pending_res", {}).pop(self._act_key(act), None)
            if pend is None:                           # (materialised just above)
                self._unhold(held_res)
                held_res = None
        if fold is not None:
            # the fold form of the input transform writes V as fp32 (F(6x6,3x3) layers whose GEMMs read fp32 activations)
            sp, _, sp_act = self.wino_gemm_form(act[3], conv.out_channels, m, self.B * -(-act[1] // 6) * -(-act[2] // 6)) if m == 6 else (0, 0, 0)
            if m != 6 or (sp and not sp_act):
                self._fold_materialise(fold, act)
                fold = None
        if m:
            return self.conv_wino(act, conv, norm, flags, aux, m, pend, defer=defer, fold=fold, share=share, dst=out)
        if stem:
            # conv on the split pipe with the producer's GroupNorm applied on load; statistics pass; the apply is left to the
            # consumer (the next stem layer, or - conv4 - the input transform of res1_conv1)
            y = self.conv(act, conv, norm_in=pend, split=True)
            dfr = defer and flags == GN_RELU_IN and aux is None and not os.environ.get("XL_NO_DEFERRED_GN")
            if self.stem_stats_ok(norm, conv.out_channels):
                # round 4: the statistics come from the convolution's epilogue (one entry per tile and row block of waves)
                bm, wm, nchunks = self._stem_stat_shape(self.ops[-1], y)
                return self.gn_fused(y, norm, flags, aux, len(self.ops) - 1, defer=dfr, share=share, stat=(bm, wm, nchunks))
            if self.train and os.environ.get("XL_TRAIN_STEM_STATS") and self.stem_stats_ok(norm, conv.out_channels, train=True):
                # training (opt-in, XL_TRAIN_STEM_STATS=1): the same epilogue statistics in a buffer of the layer's own (GN_FINAL
                # turns them into the table the apply and the backward passes read).  Not the default: the step time does not
                # move (40.5 ms either way, three 47 us passes) and the fp32 trees perturb the statistics by ~1e-7, which is
                # enough to flip ReLUs at the kinks and move the small-map gradient test against float64 autograd
                # (tests/test_semantics_gpu.py) from 0.047 to 0.063 of the max-norm
                cop = self.ops[-1]
                bm, wm, nchunks = self._stem_stat_shape(cop, y)
                stats_t = torch.zeros(self.B * nchunks * norm.num_groups * 2, dtype=torch.float64, device=self.device)
                self.keep.append(stats_t)
                cop.stats, cop.groups, cop.nchunks = stats_t.data_ptr(), norm.num_groups, nchunks
                return self.gn(y, norm, flags, aux, pre_stats=(stats_t, nchunks), stat_tile=bm, stat_mult=wm, defer=dfr)
            return self.gn(y, norm, flags, aux, defer=dfr, share=share)
        split = split_1x1 and (pend is None or absorbs)
        y = self.conv(act, conv, norm_in=pend, split=split)
        self._unhold(held_res)
        bn = 128 if conv.out_channels % 128 == 0 else 64
        # a conv tile's columns cover whole groups, and the statistics epilogue sums 2- or 4-channel pieces
        whole_groups = bn % cpg == 0 and (cpg == 2 or cpg % 4 == 0)
        if (not self.train and y[1] * y[2] >= 128 and whole_groups and (not split or cpg == 16)
                and (not self.separate_stats or (split and cpg == 16))):
            # inference: the conv epilogue produces the GroupNorm statistics, the separate stats pass is dropped
            # defer_add (round 4): the caller promises that the only consumer is a 1x1 layer on the split pipe - it applies the whole
            # GroupNorm + ReLU + residual + ReLU epilogue while it loads its operand (XL_CONV_NORM_ADD), no apply pass
            add_on_load = (defer_add and flags == (GN_RELU_IN | GN_ADD | GN_RELU_OUT) and aux is not None and split
                           and not os.environ.get("XL_NO_DEFERRED_GN") and not os.environ.get("XL_NO_ADD_ON_LOAD"))
            return self.gn_fused(y, norm, flags, aux, len(self.ops) - 1,
                                 defer=(defer and flags == GN_RELU_IN and aux is None
                                        and not os.environ.get("XL_NO_DEFERRED_GN")) or add_on_load, share=share)
        if (self.train and y[1] * y[2] >= 128 and whole_groups and not self.separate_stats
                and self.ops[-1].type == XL_OP_CONV):
            # training: same epilogue statistics, written to a buffer of the layer's own (they are inputs of the
            # backward pass).  Slots a conv tile never touches stay zero, so the consumers may sum all of them.
            cop = self.ops[-1]
            tile = 64 if cop.reserved_i == 64 else (256 if split else 128)
            G = norm.num_groups
            nchunks = (y[1] * y[2] + tile - 1) // tile + 1
            stats_t = torch.zeros(self.B * nchunks * G * 2, dtype=torch.float64, device=self.device)
            self.keep.append(stats_t)
            cop.stats, cop.groups, cop.nchunks = stats_t.data_ptr(), G, nchunks
            return self.gn(y, norm, flags, aux, pre_stats=(stats_t, nchunks), stat_tile=tile, defer=defer)
        if not self.train:
            return self.gn(y, norm, flags, aux, defer=defer and flags == GN_RELU_IN and aux is None
                           and not os.environ.get("XL_NO_DEFERRED_GN"), share=share)
        r = self.gn(y, norm, flags, aux, defer=defer)
        if r[0] is not y[0]:
            self.release(y[0])
        return r

    @staticmethod
    def _act_key(act):
        return (act[0].data_ptr(), act[5], act[3])

    def gn_fused(self, act, norm, flags, aux, conv_index, out=None, defer=False, share=False, stat=None):
        """GroupNorm apply (in place) consuming statistics emitted by the epilogue of the conv op `conv_index`.
        stat = (rows per tile, entries per tile, nchunks) when the producer is not the 1x1 / direct kernel (the stride-2 stem
        kernels: one entry per tile and row block of waves; rows per tile 0: all nchunks entries are written)."""
        t, H, W, C, ld, off = act
        G, HW = norm.num_groups, H * W
        cop = self.ops[conv_index]
        mult = 1
        if stat is not None:
            tile, mult, nchunks = stat
        else:
            tile = cop.reserved_i if cop.reserved_i in (64, 256, -256) else (256 if cop.reserved_i in (192, 384) else 128)
            nchunks = (HW + abs(tile) - 1) // abs(tile) + 1
        self.max_stats = max(self.max_stats, self.B * nchunks * G * 2)
        cop.groups, cop.nchunks = G, nchunks
        ap = XlOp()
        ap.type = XL_OP_GN_APPLY
        ap.B, ap.Hi, ap.Wi, ap.Cin, ap.groups, ap.nchunks, ap.ld_in = self.B, H, W, C, G, nchunks, ld
        ap.flags, ap.eps, ap.reserved_i = flags, norm.eps, tile
        ap.in_ = t.data_ptr() + 4 * off
        gamma, beta = self.dev(norm.weight), self.dev(norm.bias)
        ap.w, ap.bias = gamma.data_ptr(), beta.data_ptr()
        self.stats_ops.append(conv_index)
        self._emit_final(ap, gamma, beta, tile, mult=mult)
        if aux is not None:
            ap.aux = aux[0].data_ptr() + 4 * aux[5]
            ap.ld_aux = aux[4]
        if out is None:
            ap.out, ap.ld_out = ap.in_, ld
            res = act
        else:
            ot, old, ooff = out
            ap.out, ap.ld_out = ot.data_ptr() + 4 * ooff, old
            res = (ot, H, W, C, old, ooff)
        if defer and out is None:
            if not hasattr(self, "pending_gn"):
                self.pending_gn = {}
            self.pending_gn[self._act_key(res)] = ap
            if aux is not None:                        # (the residual must outlive the caller's release until the consumer is emitted)
                if not hasattr(self, "held"):
                    self.pending_fold, self.held = getattr(self, "pending_fold", {}), {}
                self.held[id(aux[0])] = [aux[0], False]
                self.pending_res = getattr(self, "pending_res", {})
                self.pending_res[self._act_key(res)] = aux[0]
            return res
        if share and out is None and self.fold_ok():
            return self._fold_begin(ap, act, aux)
        aux_ap = self._aux_take(aux)
        if aux_ap is not None:
            self._aux_apply(aux_ap)
        self.stats_ops.append(len(self.ops))
        self.ops.append(ap)
        return res

    def _emit_final(self, ap, gamma, beta, stat_tile, table=None, mult=1):
        """GN_FINAL op: one tiny launch turns the partial sums into per-(image, channel) scale/shift so the
        streaming apply kernel does no redundant reduction per workgroup.  Inference: shared statistics and
        coefficient buffers, patched in once their sizes are known.  Training (`table`): the layer's own statistics
        (ap.stats) and its own table, {scale, shift} pairs first, {mean, rstd} pairs behind them."""
        fin = XlOp()
        fin.type = XL_OP_GN_FINAL
        fin.B, fin.Hi, fin.Wi, fin.Cin, fin.groups, fin.nchunks = ap.B, ap.Hi, ap.Wi, ap.Cin, ap.groups, ap.nchunks
        fin.eps, fin.reserved_i, fin.stride = ap.eps, stat_tile, mult
        fin.w, fin.bias = gamma.data_ptr(), beta.data_ptr()
        if table is None:
            self.max_coeff = max(getattr(self, "max_coeff", 0), self.B * ap.Cin * 2)
            self.stats_ops.append(len(self.ops))
        else:
            fin.stats = ap.stats
            fin.out, fin.out2 = table.data_ptr(), table.data_ptr() + 4 * self.B * ap.Cin * 2
            ap.aux2 = table.data_ptr()
        self.ops.append(fin)

    def res_block(self, res, block):
        """relu(res + block(res)), networks.py:252-254 / :332-334"""
        x = self.cgr(res, block[0], block[1], defer=True)
        x2 = self.cgr(x, block[3], block[4], defer=True)
        self.release(x[0])
        x3 = self.cgr(x2, block[6], block[7], GN_RELU_IN | GN_ADD | GN_RELU_OUT, aux=res, share=True)
        self.release(x2[0])
        self.release(res[0])
        return x3

    def encoder(self, enc, image, out=None):
        """networks.py:221-256.  `out` = (tensor, ld, off): write the final activation into a channel slice."""
        B, H, W = self.B, self.H, self.W
        tape_start = len(self.tape)
        frozen = not any(p.requires_grad for p in enc.parameters())
        try:
            return self._encoder_body(enc, image, out)
        finally:
            if frozen:
                for e in self.tape[tape_start:]:
                    e["frozen"] = True

    def _encoder_body(self, enc, image, out=None):
        B, H, W = self.B, self.H, self.W
        cin = enc.conv1.in_channels
        c1 = enc.conv1.out_channels
        if (not self.train and cin == 3 and c1 == 32 and enc.norm1.num_groups == 32
                and not os.environ.get("XL_NO_CONV1_FUSED")):
            if self.stem12_ok(enc):
                return self._encoder_tail(enc, None, out, x2=self._stem12(enc, image))
            x = self._conv1_fused(enc, image, self.alloc(B * H * W * c1))
            return self._encoder_tail(enc, x, out)
        t1 = self.alloc(B * H * W * c1)
        if (self.train and cin == 3 and c1 == 32 and enc.norm1.num_groups == 32 and self.split_train_ok()
                and os.environ.get("XL_GEMM_SPLIT_BF16", self.SPLIT_DEFAULT) not in ("", "0", "1")
                and not os.environ.get("XL_CONV1_VALU")):
            # round 4, training plans: the matrix-pipe form of the inference plans, ONE evaluation that writes the raw output
            # (kept for the backward pass) together with its GroupNorm partial sums - no separate statistics pass over the
            # largest tensor of the network (conv1_direct_kernel + gn_stats: 0.52 + 0.13 ms at batch 16; this: 0.25)
            G = enc.norm1.num_groups
            nchunks = -(-H // 16) * -(-W // 64)
            stats_t = torch.zeros(B * nchunks * G * 2, dtype=torch.float64, device=self.device)
            self.keep.append(stats_t)
            op = XlOp()
            op.type = XL_OP_CONV1
            op.B, op.Hi, op.Wi, op.Cin, op.Ho, op.Wo, op.Cout, op.ld_out = B, H, W, 3, H, W, c1, c1
            op.groups, op.nchunks, op.reserved_i, op.eps = G, nchunks, 0, enc.norm1.eps
            op.in_, op.w, op.bias = image.data_ptr(), self.pack_conv1_split(enc.conv1).data_ptr(), self.dev(enc.conv1.bias).data_ptr()
            op.out, op.stats = t1.data_ptr(), stats_t.data_ptr()
            self.ops.append(op)
            self.image_op_indices.append(len(self.ops) - 1)
            raw1 = (t1, H, W, c1, c1, 0)
            self.tape.append(dict(kind="conv1", conv=enc.conv1, raw=raw1))
            x = self.gn(raw1, enc.norm1, GN_RELU_IN, pre_stats=(stats_t, nchunks))
            return self._encoder_tail(enc, x, out)
        op = XlOp()
        op.type = XL_OP_CONV1
        op.B, op.Hi, op.Wi, op.Cin, op.Ho, op.Wo, op.Cout, op.ld_out = B, H, W, cin, H, W, c1, c1
        op.in_ = image.data_ptr()
        op.w = self.pack_conv(enc.conv1).data_ptr()
        op.bias = self.dev(enc.conv1.bias).data_ptr()
        op.out = t1.data_ptr()
        self.ops.append(op)
        self.image_op_indices.append(len(self.ops) - 1)
        raw1 = (t1, H, W, c1, c1, 0)
        self.tape.append(dict(kind="conv1", conv=enc.conv1, raw=raw1))
        x = self.gn(raw1, enc.norm1, GN_RELU_IN)
        return self._encoder_tail(enc, x, out)

    def stem12_ok(self, enc):
        """conv1 evaluated inside conv2's operand stage (csrc/xl_stem_fused.hip, round 4): inference plans whose stem runs on
        the split pipe.  A choice by layer, never by batch.  XL_NO_STEM12=1: the two-kernel path (conv1 writes its raw output,
        conv2 normalises and splits it on load)."""
        c2 = enc.conv2
        return (c2.in_channels == 32 and c2.out_channels == 64 and c2.kernel_size[0] == 3 and c2.stride[0] == 2
                and self.split_train_ok() and os.environ.get("XL_GEMM_SPLIT_BF16", self.SPLIT_DEFAULT) not in ("", "0", "1")
                and not os.environ.get("XL_NO_SPLIT_STEM") and not os.environ.get("XL_NO_STEM12")
                and not os.environ.get("XL_CONV1_VALU") and not os.environ.get("XL_NO_DEFERRED_GN"))

    @staticmethod
    def _stem_stat_shape(cop, y):
        """(rows per tile, statistics entries per tile, nchunks) of a stride-2 stem convolution on the split pipe
        (csrc/xl_stem_split.hip: one fp64 entry per tile and row block of waves; include/crossloc_cnn.h)."""
        bm, wm = (128, 2) if cop.reserved_i == 128 else {64: (128, 4), 128: (128, 2), 256: (256, 2)}[y[3]]
        return bm, wm, (-(-(y[1] * y[2]) // bm) + 1) * wm

    def stem_stats_ok(self, norm, cout, train=False):
        """The stride-2 stem kernels (split pipe) sum the GroupNorm statistics of their output in the epilogue.
        Not for batch-invariant plans (the partial sums are grouped by tile, i.e. by the frame's position in the batch)."""
        return (self.train == train and not self.separate_stats and norm.num_groups == 32 and cout in (64, 128, 256)
                and not os.environ.get("XL_STEM_FORM") and not os.environ.get("XL_NO_STEM_STATS"))

    def _stem12(self, enc, image):
        """conv1 statistics (one evaluation of conv1, nothing written), GN_FINAL, then the fused kernel: raw conv2 output.  The
        32-channel full-resolution activation (2 GB at 47 frames) is never allocated.  Returns conv2's GroupNorm'ed activation
        with its apply left to the consumer (conv3 on the split pipe)."""
        B, H, W = self.B, self.H, self.W
        c1, G = enc.conv1.out_channels, enc.norm1.num_groups
        nchunks = -(-H // 16) * -(-W // 64)
        w1, b1 = self.pack_conv1_split(enc.conv1), self.dev(enc.conv1.bias)
        gamma, beta = self.dev(enc.norm1.weight), self.dev(enc.norm1.bias)
        st = XlOp()
        st.type = XL_OP_CONV1
        st.B, st.Hi, st.Wi, st.Cin, st.Ho, st.Wo, st.Cout, st.ld_out = B, H, W, 3, H, W, c1, c1
        st.groups, st.nchunks, st.reserved_i, st.eps = G, nchunks, 0, enc.norm1.eps
        st.in_, st.w, st.bias = image.data_ptr(), w1.data_ptr(), b1.data_ptr()
        self.max_stats = max(self.max_stats, B * nchunks * G * 2)
        self.stats_ops.append(len(self.ops))
        self.image_op_indices.append(len(self.ops))
        self.ops.append(st)
        shape = XlOp()                                 # GN_FINAL sees the normalised tensor: c1 channels, G groups
        shape.B, shape.Hi, shape.Wi, shape.Cin, shape.groups, shape.nchunks, shape.eps = B, H, W, c1, G, nchunks, enc.norm1.eps
        self._emit_final(shape, gamma, beta, 0)
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        c2 = enc.conv2.out_channels
        y = self.alloc(B * Ho * Wo * c2)
        op = XlOp()
        op.type = XL_OP_STEM12
        op.B, op.Hi, op.Wi, op.Cin, op.Ho, op.Wo, op.Cout, op.ld_out = B, H, W, 3, Ho, Wo, c2, c2
        op.ksize, op.stride, op.flags = 3, 2, GN_RELU_IN
        op.in_, op.w, op.bias = image.data_ptr(), w1.data_ptr(), b1.data_ptr()
        if self.pair_ok() and not os.environ.get("XL_NO_PAIR_STEM"):
            # round 5: conv2 inside the fused kernel as three fp16 passes (conv1's normalised output is a GroupNorm output)
            op.flags |= CONV_PAIR_F16
            op.aux = self.pack_conv2_pair_fragments(enc.conv2).data_ptr()
            op.scale = self.pair_scales.data_ptr()
        else:
            op.aux = self.pack_conv2_fragments(enc.conv2).data_ptr()
        op.stats2 = self.dev(enc.conv2.bias).data_ptr()
        queue = torch.zeros(4, dtype=torch.int32, device=self.device)      # the launch's tile queue (zero before and after)
        self.keep.append(queue)
        op.out2 = queue.data_ptr()
        op.out = y.data_ptr()
        self.deferred_gn_consumers = getattr(self, "deferred_gn_consumers", []) + [len(self.ops)]
        self.image_op_indices.append(len(self.ops))
        self.ops.append(op)
        raw2 = (y, Ho, Wo, c2, c2, 0)
        if self.stem_stats_ok(enc.norm2, c2) and enc.norm2.num_groups == 32:   # conv2's statistics from the fused kernel's epilogue
            th = 8 if os.environ.get("XL_STEM12_TILE") == "8" else 4
            return self.gn_fused(raw2, enc.norm2, GN_RELU_IN, None, len(self.ops) - 1,
                                 defer=not os.environ.get("XL_NO_DEFERRED_GN"),
                                 stat=(0, 1, -(-Wo // 16) * -(-Ho // th) * (th // 2)))
        return self.gn(raw2, enc.norm2, GN_RELU_IN, None, defer=not os.environ.get("XL_NO_DEFERRED_GN"))

    def _conv1_fused(self, enc, image, t1):
        """Inference form of conv1 + GroupNorm + ReLU: a statistics-only evaluation of the convolution, GN_FINAL, then
        a second evaluation that writes the normalised activation - the raw 32-channel full-resolution tensor (the
        largest of the network) is never written, re-read for statistics or re-read for the apply."""
        B, H, W = self.B, self.H, self.W
        c1, G, ppt = enc.conv1.out_channels, enc.norm1.num_groups, 5
        nchunks = -(-(H * W) // (256 * ppt))
        w, bias = self.pack_conv(enc.conv1), self.dev(enc.conv1.bias)
        if not os.environ.get("XL_CONV1_VALU"):       # matrix-pipe form: one workgroup per 16 x 64 output tile
            ppt, nchunks = 0, -(-H // 16) * -(-W // 64)
            w = self.pack_conv1_split(enc.conv1)
        gamma, beta = self.dev(enc.norm1.weight), self.dev(enc.norm1.bias)

        def conv1_op():
            op = XlOp()
            op.type = XL_OP_CONV1
            op.B, op.Hi, op.Wi, op.Cin, op.Ho, op.Wo, op.Cout, op.ld_out = B, H, W, 3, H, W, c1, c1
            op.groups, op.nchunks, op.reserved_i, op.eps = G, nchunks, ppt, enc.norm1.eps
            op.in_, op.w, op.bias = image.data_ptr(), w.data_ptr(), bias.data_ptr()
            return op
        st = conv1_op()
        self.max_stats = max(self.max_stats, B * nchunks * G * 2)
        act1 = (t1, H, W, c1, c1, 0)
        if (ppt == 0 and self.stem_split_ok(act1, enc.conv2) and not os.environ.get("XL_NO_DEFERRED_GN")
                and not os.environ.get("XL_CONV1_TWO_PASS")):
            # round 3: ONE evaluation - the raw convolution is written together with its statistics, and conv2 (on the split
            # pipe) applies GroupNorm + ReLU while it gathers its operand: the statistics-only evaluation disappears
            st.out = t1.data_ptr()
            self.stats_ops.append(len(self.ops))
            self.image_op_indices.append(len(self.ops))
            self.ops.append(st)
            ap = XlOp()                                # the apply pass, left to the consumer (materialised only if it cannot)
            ap.type = XL_OP_GN_APPLY
            ap.B, ap.Hi, ap.Wi, ap.Cin, ap.groups, ap.nchunks, ap.ld_in = B, H, W, c1, G, nchunks, c1
            ap.flags, ap.eps = GN_RELU_IN, enc.norm1.eps
            ap.in_ = ap.out = t1.data_ptr()
            ap.ld_out = c1
            ap.w, ap.bias = gamma.data_ptr(), beta.data_ptr()
            self._emit_final(ap, gamma, beta, 0)
            if not hasattr(self, "pending_gn"):
                self.pending_gn = {}
            self.pending_gn[self._act_key(act1)] = ap
            return act1
        self.stats_ops.append(len(self.ops))
        self.image_op_indices.append(len(self.ops))
        self.ops.append(st)
        shape = XlOp()                                 # GN_FINAL sees the normalised tensor: c1 channels, G groups
        shape.B, shape.Hi, shape.Wi, shape.Cin, shape.groups, shape.nchunks, shape.eps = B, H, W, c1, G, nchunks, enc.norm1.eps
        self._emit_final(shape, gamma, beta, 0)
        ap = conv1_op()
        ap.out, ap.flags = t1.data_ptr(), GN_RELU_IN
        self.deferred_gn_consumers = getattr(self, "deferred_gn_consumers", []) + [len(self.ops)]
        self.image_op_indices.append(len(self.ops))
        self.ops.append(ap)
        return (t1, H, W, c1, c1, 0)

    def _encoder_tail(self, enc, x, out=None, x2=None):
        if x2 is None:
            x2 = self.cgr(x, enc.conv2, enc.norm2, defer=True); self.release(x[0])
        x3 = self.cgr(x2, enc.conv3, enc.norm3, defer=True); self.release(x2[0])
        res = self.cgr(x3, enc.conv4, enc.norm4, share=True); self.release(x3[0])
        a = self.cgr(res, enc.res1_conv1, enc.res1_norm1, defer=True)
        b = self.cgr(a, enc.res1_conv2, enc.res1_norm2, defer=True); self.release(a[0])
        c = self.cgr(b, enc.res1_conv3, enc.res1_norm3, GN_RELU_IN | GN_ADD | GN_RELU_OUT, aux=res, share=True)
        self.release(b[0]); self.release(res[0])
        res = c
        a = self.cgr(res, enc.res2_conv1, enc.res2_norm1, defer=True)
        b = self.cgr(a, enc.res2_conv2, enc.res2_norm2, defer=True); self.release(a[0])
        n_add = len(enc.enc_add_res_block_ls)
        last_out = out if n_add == 0 else None
        # (c is consumed by the addition below only: when that addition is folded into the next block's input transform, so is
        #  c's own GroupNorm + ReLU)
        aux_fold = last_out is None and self.fold_ok() and not os.environ.get("XL_NO_AUX_FOLD")
        if enc.tiny:
            # networks.py:245-250 with tiny=True: no projection on the skip path - res2 closes like res1, relu(res + x)
            c = self.cgr(b, enc.res2_conv3, enc.res2_norm3, GN_RELU_IN | GN_ADD | GN_RELU_OUT, aux=res,
                         share=last_out is None, out=last_out)
            self.release(b[0]); self.release(res[0])
            res = c
            return self._encoder_add_blocks(enc, res, out)
        c = self.cgr(b, enc.res2_conv3, enc.res2_norm3, defer=aux_fold); self.release(b[0])
        if aux_fold:
            self._aux_defer(c)
        if last_out is None:                        # conv -> GroupNorm with the statistics out of the conv epilogue
            skip_in = res
            res = self.cgr(skip_in, enc.res2_skip, enc.res2_skip_norm, GN_ADD | GN_RELU_OUT, aux=c, share=True)
            self.release(skip_in[0]); self.release(c[0])
        else:
            sk = self.conv(res, enc.res2_skip)
            self.release(res[0])
            res = self.gn(sk, enc.res2_skip_norm, GN_ADD | GN_RELU_OUT, aux=c, out=last_out)
            self.release(c[0])
            if res[0] is not sk[0]:
                self.release(sk[0])
        return self._encoder_add_blocks(enc, res, out)

    def _encoder_add_blocks(self, enc, res, out=None):
        """networks.py:252-254: the encoder's additional residual blocks; the last one writes into `out` if given."""
        n_add = len(enc.enc_add_res_block_ls)
        for i, block in enumerate(enc.enc_add_res_block_ls):
            if i == n_add - 1 and out is not None:
                # (round 4: like every other block - GroupNorm applies deferred to the consumer, the last 3x3 layer as
                #  Winograd - except that the block's final apply writes into the encoder's slice of the concat buffer.
                #  Until round 3 this block ran undeferred and its last layer as the DIRECT fp32-MFMA convolution: 4.3 ms
                #  instead of 0.9 per encoder at 24 frames)
                if os.environ.get("XL_MLR_LAST_DIRECT"):          # the round-3 lowering, kept for the A/B
                    x = self.cgr(res, block[0], block[1])
                    x2 = self.cgr(x, block[3], block[4]); self.release(x[0])
                    y = self.conv(x2, block[6]); self.release(x2[0])
                    r = self.gn(y, block[7], GN_RELU_IN | GN_ADD | GN_RELU_OUT, aux=res, out=out)
                    self.release(y[0]); self.release(res[0])
                    res = r
                    continue
                x = self.cgr(res, block[0], block[1], defer=True)
                x2 = self.cgr(x, block[3], block[4], defer=True); self.release(x[0])
                r = self.cgr(x2, block[6], block[7], GN_RELU_IN | GN_ADD | GN_RELU_OUT, aux=res, out=out)
                self.release(x2[0]); self.release(res[0])
                res = r
            else:
                res = self.res_block(res, block)
        return res

    def _lower(self, net):
        dec = net.decoder
        if net.num_mlr == 0:
            res = self.encoder(net.encoder, _DUMMY)
        else:
            c = (512, 128)[net.tiny]
            # encoders write straight into channel slices of the concat buffer (networks.py:485-488)
            h, w = self.H, self.W
            for _ in range(3):
                h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            Ho, Wo = h, w
            ctot = c * net.num_mlr
            cat = self.alloc(self.B * Ho * Wo * ctot)
            for i, enc in enumerate(net.mlr_encoder_ls):
                self.encoder(enc, _DUMMY, out=(cat, ctot, i * c))
            mlr = (cat, Ho, Wo, ctot, ctot, 0)
            sk = self.cgr(mlr, net.mlr_skip[0], net.mlr_skip[1], 0)
            # (mlr_norm's only consumer is the fusion layer: its input transform applies the normalisation - no apply pass over
            #  the 1536-channel buffer)
            mlr = self.gn(mlr, net.mlr_norm, 0, defer=not os.environ.get("XL_NO_DEFERRED_GN"))
            f = net.mlr_forward
            a = self.cgr(mlr, f[0], f[1], defer=True); self.release(cat)
            b = self.cgr(a, f[3], f[4], defer=True); self.release(a[0])
            res = self.cgr(b, f[6], f[7], GN_RELU_IN | GN_ADD | GN_RELU_OUT, aux=sk, share=True)
            self.release(b[0]); self.release(sk[0])
        for block in dec.dec_add_res_block_ls:
            res = self.res_block(res, block)
        a = self.cgr(res, dec.res3_conv1, dec.res3_norm1, defer=True)
        b = self.cgr(a, dec.res3_conv2, dec.res3_norm2, defer=True); self.release(a[0])
        # (res3's output has ONE consumer, fc1 - a 1x1 layer: its GroupNorm + ReLU + residual + ReLU is applied by fc1's operand load)
        c = self.cgr(b, dec.res3_conv3, dec.res3_norm3, GN_RELU_IN | GN_ADD | GN_RELU_OUT, aux=res, defer_add=True)
        self.release(b[0]); self.release(res[0])
        res = c
        a = self.cgr(res, dec.fc1, dec.fc1_norm, defer=True); self.release(res[0])
        b = self.cgr(a, dec.fc2, dec.fc2_norm, defer=True); self.release(a[0])
        if dec.full_size_output:
            # networks.py:344-349: DUC conv-GN-ReLU; pixel shuffle, bilinear trim and fc3 fused in one kernel
            d = self.cgr(b, dec.duc_upsample.conv, dec.duc_upsample.norm); self.release(b[0])
            t, H, W, C, ld, off = d
            nc = dec.num_task_channel + dec.num_pos_channel
            op = XlOp()
            op.type = XL_OP_DUC_HEAD
            op.B, op.Hi, op.Wi, op.Cin, op.Ho, op.Wo, op.Cout = self.B, H, W, C, self.H, self.W, nc
            op.n_task, op.n_pos, op.ld_in = dec.num_task_channel, dec.num_pos_channel, ld
            op.clamp_lo, op.clamp_hi = -16.10, 13.82
            op.in_ = t.data_ptr() + 4 * off
            w3 = dec.fc3.weight.detach().to(device=self.device, dtype=torch.float32).reshape(nc, nc).contiguous()
            self.keep.append(w3)
            op.w = w3.data_ptr()
            op.bias = self.dev(dec.fc3.bias).data_ptr()
            op.aux = self.dev(dec.mean).data_ptr()
            self.ops.append(op)
            self.out_op_index = len(self.ops) - 1
            self.out_shape = (self.B, nc, self.H, self.W)
            self.tape.append(dict(kind="duc_head", fc3=dec.fc3, x=d, w3=w3, cout=nc, n_task=op.n_task))
            return
        pend = getattr(self, "pending_gn", {}).pop(self._act_key(b), None)
        if pend is not None and self.train:              # (the head's backward pass reads the normalised activation)
            b = self._train_materialise(pend, b)
            pend = None
        t, H, W, C, ld, off = b
        nout = dec.num_task_channel + dec.num_pos_channel
        if pend is not None and not (C == 512 and nout <= 4):
            self.stats_ops.append(len(self.ops))       # the general head form reads a normalised activation
            self.ops.append(pend)
            pend = None
        op = XlOp()
        op.type = XL_OP_HEAD
        if pend is not None:                           # fc2's GroupNorm + ReLU applied by the head while it loads
            op.flags = CONV_NORM_RELU if pend.flags & GN_RELU_IN else 0
            self.deferred_gn_consumers = getattr(self, "deferred_gn_consumers", []) + [len(self.ops)]
        op.B, op.Hi, op.Wi, op.Cin, op.Ho, op.Wo = self.B, H, W, C, H, W
        op.Cout = dec.num_task_channel + dec.num_pos_channel
        op.n_task, op.n_pos, op.ld_in = dec.num_task_channel, dec.num_pos_channel, ld
        op.clamp_lo, op.clamp_hi = -16.10, 13.82
        op.in_ = t.data_ptr() + 4 * off
        w3 = dec.fc3.weight.detach().to(device=self.device, dtype=torch.float32).reshape(op.Cout, C).contiguous()
        self.keep.append(w3)
        op.w = w3.data_ptr()
        op.bias = self.dev(dec.fc3.bias).data_ptr()
        op.aux = self.dev(dec.mean).data_ptr()
        self.ops.append(op)
        self.out_op_index = len(self.ops) - 1
        self.out_shape = (self.B, op.Cout, H, W)
        self.tape.append(dict(kind="head", fc3=dec.fc3, x=b, w3=w3, cout=op.Cout, n_task=op.n_task))

    # ------------------------------------------------------------------ backward lowering (train plans)
    @staticmethod
    def _key(act):
        return (act[0].data_ptr(), act[5], act[3])          # (storage, channel offset, channels)

    def _lower_backward(self):
        B, dev = self.B, self.device
        bops = []
        grads = {}                # activation key -> (gradient tensor, ld, channel offset); layout like the activation
        graw = {}                 # conv-output key -> dense gradient tensor
        self.param_grads = []     # (parameter, tensor) in production order
        scratch_f = 0             # fp32 scratch (wgrad split-K partials, head / conv1 partials)
        scratch_d = 0             # fp64 scratch (GroupNorm backward sums)
        patch_f, patch_d = [], []
        self.conv1_wgrad_indices = []
        producers = {self._key(e["raw"]): e for e in self.tape if e["kind"] in ("conv", "conv1")}
        # round 5: backward GEMMs as fp16 pairs (XL_TRAIN_PAIR_BWD=0: six-pass bf16).  A gradient has no static bound: every
        # GroupNorm-backward apply pass records max |dx| (a float's bits, atomicMax) in a slot of its own, and so does the dY
        # transform of a Winograd weight gradient; the GEMMs that read those tensors derive their power-of-two scale from the
        # slot.  The slots are zeroed by the first op of the list.
        pair_bwd = self.pair_ok() and os.environ.get("XL_TRAIN_PAIR_BWD", "1") not in ("", "0")
        self.bwd_amax = torch.zeros(1024, dtype=torch.int32, device=dev)
        amax_n = [0]
        gamax = {}                # conv-output key -> byte address of the slot holding max |its gradient|

        params_list = None if os.environ.get("XL_GNB_PARAMS_PER_LAYER") else []
        c1_fold, patch_bco = {}, []

        def new_slot():
            assert amax_n[0] < 1024
            amax_n[0] += 1
            return self.bwd_amax.data_ptr() + 4 * (amax_n[0] - 1)
        if pair_bwd:
            z = XlOp()
            z.type, z.Cin, z.out = XL_OP_FILL0, 4 * 1024, self.bwd_amax.data_ptr()
            bops.append(z)

        # every parameter gradient is a slice of ONE flat buffer (16-byte aligned slices): a backward pass hands its result
        # out with one device-to-device copy of that buffer instead of one clone per parameter (run_backward)
        total = sum((p.numel() + 3) // 4 * 4 for p in self.net.parameters() if p.requires_grad)
        self.grad_flat = torch.zeros(max(total, 4), dtype=torch.float32, device=dev)
        self.grad_slices, cursor = [], [0]

        def pgrad(param):
            n = param.numel()
            t = self.grad_flat[cursor[0]:cursor[0] + n]
            self.param_grads.append((param, t))
            self.grad_slices.append((param, cursor[0], n))
            cursor[0] += (n + 3) // 4 * 4
            assert cursor[0] <= self.grad_flat.numel()
            return t

        def find_grad(act):
            """Gradient of an activation: its own entry, or a channel slice of a wider tensor's gradient
            (the three encoder outputs are slices of the MLR concat buffer)."""
            k = self._key(act)
            if k in grads:
                return grads[k]
            ptr, off, C = k
            for (p2, o2, c2), (gt, gld, goff) in grads.items():
                if p2 == ptr and o2 <= off and off + C <= o2 + c2:
                    return (gt, gld, goff + off - o2)
            return None

        for e in reversed(self.tape):
            kind = e["kind"]
            if e.get("frozen"):
                continue                                       # frozen encoder (networks.py:424-428): nothing to do
            if kind == "head":
                t, H, W, C, ld, off = e["x"]
                gin = self.alloc(B * H * W * C)
                op = XlOp()
                op.type = XL_OP_HEAD_BWD
                op.B, op.Hi, op.Wi, op.Cin, op.Cout, op.n_task = B, H, W, C, e["cout"], e["n_task"]
                op.ld_in, op.ld_out = ld, C
                op.clamp_lo, op.clamp_hi = -16.10, 13.82
                op.in_, op.w, op.out = t.data_ptr() + 4 * off, e["w3"].data_ptr(), gin.data_ptr()
                op.out2 = pgrad(e["fc3"].weight).data_ptr()
                op.stats = pgrad(e["fc3"].bias).data_ptr()
                waves = 4 * max(1, min(256, (B * H * W + 63) // 64))
                scratch_f = max(scratch_f, waves * e["cout"] * (C + 1))
                patch_f.append(len(bops))
                self.head_bwd_index = len(bops)
                bops.append(op)
                grads[self._key(e["x"])] = (gin, C, 0)
            elif kind == "duc_head":
                t, H, W, C, ld, off = e["x"]
                gin = self.alloc(B * H * W * C)
                nc = e["cout"]
                op = XlOp()
                op.type = XL_OP_DUC_HEAD_BWD
                op.B, op.Hi, op.Wi, op.Cin, op.Ho, op.Wo, op.Cout, op.n_task = B, H, W, C, self.H, self.W, nc, e["n_task"]
                op.ld_in, op.ld_out = ld, C
                op.clamp_lo, op.clamp_hi = -16.10, 13.82
                op.in_, op.w, op.out = t.data_ptr() + 4 * off, e["w3"].data_ptr(), gin.data_ptr()
                op.out2 = pgrad(e["fc3"].weight).data_ptr()
                op.stats = pgrad(e["fc3"].bias).data_ptr()
                blocks = max(1, min(1024, (B * self.H * self.W + 255) // 256))
                # (+ d(interpolated activation) [B][nc][H][W] when the bilinear trim is active)
                scratch_f = max(scratch_f, (blocks + 1) * (nc * nc + nc) + B * nc * self.H * self.W)
                patch_f.append(len(bops))
                self.head_bwd_index = len(bops)
                bops.append(op)
                grads[self._key(e["x"])] = (gin, C, 0)
            elif kind == "gn":
                t, H, W, C, ld, off = e["raw"]
                gout = find_grad(e["out"])
                if gout is None:
                    continue                                   # output unused downstream of any trainable path
                flags = e["flags"]
                prod = producers.get(self._key(e["raw"]))
                # conv1's GroupNorm: its dx has ONE reader, conv1's weight gradient (the image needs no data gradient) - that kernel
                # applies the backward pass on load and the apply pass (2.1 GB read and written at batch 16) is not run
                fold_c1 = (prod is not None and prod["kind"] == "conv1" and e["aux"] is None and C == 32
                           and not (flags & (GN_ADD | GN_RELU_OUT)) and ld % 4 == 0 and off % 4 == 0
                           and not os.environ.get("XL_NO_CONV1_WGRAD_FOLD"))
                dx = None if fold_c1 else self.alloc(B * H * W * C)
                daux = None
                if e["aux"] is not None:
                    daux = find_grad(e["aux"])
                    if daux is not None:
                        flags |= GN_ACC_AUX
                    else:
                        daux = (self.alloc(B * H * W * C), C, 0)
                        grads[self._key(e["aux"])] = daux
                G = e["norm"].num_groups
                nch2 = max(1, min(128, (H * W + 63) // 64))
                scratch_d = max(scratch_d, B * nch2 * C * 3 + B * C * 6 + (B * C * 3 + 1) // 2)
                # d gamma / d beta / d bias of ALL layers come from one launch at the end of the pass (XL_OP_GNB_PARAMS_LIST): the
                # layer's per-(image, channel) sums then live in a buffer of their own (XL_GNB_PARAMS_PER_LAYER=1: one launch each)
                sums = None
                if params_list is not None:
                    sums = torch.empty(B * C * 6, dtype=torch.float64, device=dev)
                    self.keep.append(sums)
                for typ in (XL_OP_GNB_STATS, XL_OP_GNB_FINAL, XL_OP_GNB_APPLY, XL_OP_GNB_PARAMS):
                    if typ == XL_OP_GNB_APPLY and fold_c1:
                        continue
                    op = XlOp()
                    op.type = typ
                    op.B, op.Hi, op.Wi, op.Cin, op.groups = B, H, W, C, G
                    op.nchunks2, op.flags, op.eps = nch2, flags, e["norm"].eps
                    op.ld_in, op.ld_aux, op.ld_out = ld, gout[1], e["out"][4]
                    op.in_ = t.data_ptr() + 4 * off
                    op.w, op.bias = e["gamma"].data_ptr(), e["beta"].data_ptr()
                    op.stats = e["table"].data_ptr()
                    op.aux = gout[0].data_ptr() + 4 * gout[2]
                    op.aux2 = e["out"][0].data_ptr() + 4 * e["out"][5]
                    if typ == XL_OP_GNB_APPLY:
                        op.out = dx.data_ptr()
                        if pair_bwd and prod is not None:
                            gamax[self._key(e["raw"])] = op.scale = new_slot()
                        if daux is not None:
                            op.out2 = daux[0].data_ptr() + 4 * daux[2]
                            op.Cout = daux[1]                  # pixel stride of the d(residual) tensor
                    elif typ == XL_OP_GNB_FINAL and sums is not None:
                        op.scale = sums.data_ptr()
                    elif typ == XL_OP_GNB_PARAMS:
                        op.out = pgrad(e["norm"].weight).data_ptr()
                        op.out2 = pgrad(e["norm"].bias).data_ptr()
                        if prod is not None:
                            op.aux2 = pgrad(prod["conv"].bias).data_ptr()
                        else:
                            op.flags = flags | GN_NO_CONV_BIAS
                        if sums is not None:
                            params_list.append((sums.data_ptr(), e["gamma"].data_ptr(), op.out, op.out2,
                                                op.aux2 if prod is not None else 0, B, C, G, H * W))
                            continue
                    patch_d.append(len(bops))
                    bops.append(op)
                if fold_c1:
                    # (the gradient w.r.t. the GroupNorm output stays alive until conv1's weight gradient has read it)
                    c1_fold[self._key(e["raw"])] = dict(dout=gout, x=(t, ld, off), fco=e["table"], flags=flags,
                                                       bco_off=B * nch2 * C * 3 + B * C * 6,
                                                       release=grads.pop(self._key(e["out"]), (None,))[0])
                    graw[self._key(e["raw"])] = gout[0]
                    continue
                if self._key(e["out"]) in grads:                # dense, fully consumed: recycle (slices of the
                    self.release_grad(grads.pop(self._key(e["out"]))[0])   # concat gradient stay until the end)
                if prod is not None:
                    graw[self._key(e["raw"])] = dx
                else:
                    # GroupNorm applied directly to an activation (mlr_norm on the concat buffer): dx is a
                    # gradient of that activation
                    assert find_grad(e["raw"]) is None
                    grads[self._key(e["raw"])] = (dx, C, 0)
            elif kind == "conv":
                conv = e["conv"]
                t, H, W, C, ld, off = e["x"]
                rt, Ho, Wo, Cout, rld, roff = e["raw"]
                dy = graw.pop(self._key(e["raw"]), None)
                if dy is None:
                    continue
                dy_amax = gamax.pop(self._key(e["raw"]), None)
                k, s = conv.kernel_size[0], conv.stride[0]
                bo = 128 if Cout % 128 == 0 else 64
                bc = 128 if C % 128 == 0 else (64 if C % 64 == 0 else 32)
                wm = 0
                if (k == 3 and s == 1 and H * W >= 64 and C % 64 == 0 and Cout % 128 == 0
                        and not os.environ.get("XL_NO_WINOGRAD") and not os.environ.get("XL_NO_WINOGRAD_TRAIN")
                        and not os.environ.get("XL_NO_WINOGRAD_WGRAD")):
                    # a V kept by the forward pass fixes the tile size; otherwise the cheapest form for this map
                    # (a deferred GroupNorm - xnorm - was accepted by wino_wgrad_ok() for the FORWARD tile size: keep it, V kept or not)
                    wm = e.get("wm", 0) if (e.get("v") is not None or e.get("xnorm") is not None) else self.wino_pick(H, W, max(C, Cout))
                Tw4 = B * -(-H // wm) * -(-W // wm) if wm else 0          # tiles = K dimension of the GEMMs
                wino_w = wm in (4, 6) and Tw4 >= 64
                nfw = (wm + 2) ** 2
                if wino_w:
                    # weight gradient through F(m x m,3x3): V = B^T x B, dM = A dY A^T, (m+2)^2 GEMMs over the tiles,
                    # dg = G^T dU G
                    Th, Tw = -(-H // wm), -(-W // wm)
                    Vb = e.get("v")
                    if Vb is None:
                        Vb = self.alloc(nfw * Tw4 * C)
                        wi = XlOp()
                        wi.type, wi.ksize = XL_OP_WINO_IN, wm
                        wi.B, wi.Hi, wi.Wi, wi.Cin, wi.Ho, wi.Wo, wi.ld_in = B, H, W, C, Th, Tw, ld
                        wi.in_, wi.out = t.data_ptr() + 4 * off, Vb.data_ptr()
                        if e.get("xnorm") is not None:                # x is a raw conv output whose GroupNorm was left to its consumers
                            wi.aux2, wi.flags = e["xnorm"].aux2, e["xnorm"].flags & GN_RELU_IN
                        bops.append(wi)
                    dMb = self.alloc(nfw * Tw4 * Cout)
                    wd = XlOp()
                    wd.type, wd.ksize = XL_OP_WINO_DY, wm
                    wd.B, wd.Hi, wd.Wi, wd.Cin, wd.Ho, wd.Wo, wd.ld_in = B, H, W, Cout, Th, Tw, Cout
                    wd.in_, wd.out = dy.data_ptr(), dMb.data_ptr()
                    wg_pair = pair_bwd and wm == 6 and self.wgrad_split_ok(C, Cout)
                    if wg_pair:
                        wd.scale = new_slot()                          # max |dM|: the scale of the weight-gradient GEMMs' dY operand
                    bops.append(wd)
                    dU = self.alloc(nfw * Cout * C)
                    tiles = nfw * (Cout // bo) * (C // bc)
                    steps_total = -(-Tw4 // 32)
                    best, splits = None, 1
                    for cand in range(1, 33):
                        if cand > max(1, Tw4 // 256):
                            break
                        rounds = -(-(tiles * cand) // 512)
                        cost = rounds * (-(-steps_total // cand) + 8) + (cand + 1) * tiles * bo * bc * 4 / 4e12 / 1.8e-6
                        if best is None or cost < best - 1e-9:
                            best, splits = cost, cand
                    wg = XlOp()
                    wg.type = XL_OP_WGRAD
                    wg.B, wg.Hi, wg.Wi, wg.Cin, wg.Ho, wg.Wo, wg.Cout = 1, Tw4, 1, C, Tw4, 1, Cout
                    if self.wgrad_split_ok(C, Cout):
                        # on the split pipe (csrc/xl_wgrad_split.hip): 256 x 256 tiles, one workgroup per CU
                        wg.flags = CONV_SPLIT_BF16
                        splits = self.wgrad_splits(nfw * (Cout // 256) * (C // 256), Tw4)
                        if wg_pair:                                    # csrc/xl_wgrad_pair.hip: V at the plan's scale, dM at its own
                            wg.flags |= CONV_PAIR_F16
                            wg.scale, wg.out2 = self.pair_scales.data_ptr() + 8, wd.scale
                    wg.ksize, wg.stride, wg.ld_in, wg.ld_aux, wg.groups, wg.nchunks2 = 1, 1, C, Cout, nfw, splits
                    wg.in_, wg.aux, wg.out = Vb.data_ptr(), dMb.data_ptr(), dU.data_ptr()
                    scratch_f = max(scratch_f, nfw * splits * Cout * C)
                    patch_f.append(len(bops))
                    bops.append(wg)
                    wf = XlOp()
                    wf.type, wf.ksize = XL_OP_WINO_WFINAL, wm
                    wf.Cin, wf.Cout = C, Cout
                    wf.in_, wf.out = dU.data_ptr(), pgrad(conv.weight).data_ptr()
                    bops.append(wf)
                    self.release_grad(Vb); self.release_grad(dMb); self.release_grad(dU)   # a kept V is dead from here on
                op = XlOp()
                op.type = XL_OP_WGRAD
                op.B, op.Hi, op.Wi, op.Cin, op.Ho, op.Wo, op.Cout = B, H, W, C, Ho, Wo, Cout
                op.ksize, op.stride, op.ld_in, op.ld_aux = k, s, ld, Cout
                tiles = k * k * (Cout // bo) * (C // bc)
                M = B * Ho * Wo
                # split-K factor from a cost model in units of one K-step (32 pixels) of a workgroup: rounds of the
                # 512 resident workgroups (2 per CU) x (K-steps per split + ~8 steps of prologue / partial-tile
                # store) + the fixed-order reduce pass over the partial tiles (~4 TB/s, one K-step ~ 1.8 us).
                # E.g. 3x3 512->512 at batch 16: 144 tiles x 7 = 1008 workgroups = 1.97 rounds of 386 steps.
                steps_total = -(-M // 32)
                best, splits = None, 1
                # (round 4: resident workgroups by the tile form's LDS - 2 stages x 32 pixels x (bo + bc) floats: the 64 x 32 form of
                #  conv2's weight gradient, 9 tiles, fits six per CU, and 56 splits = 504 two-wave workgroups had left the chip at one
                #  wave per SIMD: 0.99 -> 0.54 ms; 128 x 64 fits three)
                per_cu = max(2, min(6, (160 * 1024) // (2 * 32 * (bo + bc) * 4)))
                if os.environ.get("XL_WGRAD_SMALL_SPLITS_OLD"):
                    per_cu = 2
                resident, max_splits = 256 * per_cu, 32 * per_cu
                for cand in range(1, max_splits + 1):
                    if cand > max(1, M // 256):
                        break
                    n_wg = tiles * cand
                    rounds = -(-n_wg // resident)
                    cost = rounds * (-(-steps_total // cand) + 8) + (cand + 1) * tiles * bo * bc * 4 / 4e12 / 1.8e-6
                    if best is None or cost < best - 1e-9:
                        best, splits = cost, cand
                if k == 1 and s == 1 and not wino_w and self.wgrad_split_ok(C, Cout) and ld % 4 == 0 and off % 4 == 0:
                    op.flags = CONV_SPLIT_BF16
                    splits = self.wgrad_splits((Cout // 256) * (C // 256), M)
                    if pair_bwd and dy_amax is not None:
                        op.flags |= CONV_PAIR_F16
                        op.scale, op.out2 = self.pair_scales.data_ptr(), dy_amax
                    if e.get("xnorm") is not None:                    # x is a raw conv output: normalise on load
                        op.flags |= CONV_NORM_IN | (CONV_NORM_RELU if e["xnorm"].flags & GN_RELU_IN else 0)
                        op.aux2 = e["xnorm"].aux2
                else:
                    assert e.get("xnorm") is None or wino_w, "a deferred GroupNorm reached a weight-gradient form that cannot apply it"
                op.nchunks2 = splits
                op.in_, op.aux = t.data_ptr() + 4 * off, dy.data_ptr()
                if not wino_w:
                    op.out = pgrad(conv.weight).data_ptr()
                    scratch_f = max(scratch_f, splits * k * k * Cout * C)
                    patch_f.append(len(bops))
                    bops.append(op)
                # data gradient into the gradient of x (second producers accumulate)
                op = XlOp()
                op.type = XL_OP_CONV
                op.flags = CONV_DGRAD
                gx = find_grad(e["x"])
                if gx is not None:
                    op.flags |= CONV_ACCUMULATE
                else:
                    gx = (self.alloc(B * H * W * C), C, 0)
                    grads[self._key(e["x"])] = gx
                m = self.wino_dgrad_m(conv, H, W, C)
                if m:
                    # dX = conv3x3(dY, flipped kernel, channels swapped) as F(m x m,3x3): 4x / 5x fewer multiplies
                    Th, Tw = -(-H // m), -(-W // m)
                    T, nf = B * Th * Tw, (m + 2) ** 2
                    Vb = self.alloc(nf * T * Cout)
                    wi = XlOp()
                    wi.type, wi.ksize = XL_OP_WINO_IN, m
                    wi.B, wi.Hi, wi.Wi, wi.Cin, wi.Ho, wi.Wo, wi.ld_in = B, H, W, Cout, Th, Tw, Cout
                    wi.in_, wi.out = dy.data_ptr(), Vb.data_ptr()
                    bops.append(wi)
                    Mb = self.alloc(nf * T * C)
                    gm = XlOp()
                    gm.type = XL_OP_CONV
                    gm.B, gm.Hi, gm.Wi, gm.Cin, gm.Ho, gm.Wo, gm.Cout = B, Th, Tw, Cout, Th, Tw, C
                    gm.ksize, gm.stride, gm.ld_in, gm.ld_out, gm.nchunks2 = 1, 1, Cout, C, nf
                    gm.in_, gm.out = Vb.data_ptr(), Mb.data_ptr()
                    tile_major = 0
                    if self.wino_gemm_form(Cout, C, m, T)[2]:       # on the split pipe, V(dY) split inside the GEMM kernel
                        tile_major = CONV_M_TILE_MAJOR if (m == 6 and os.environ.get("XL_WINO_M_TILE_MAJOR")) else 0
                        gm.flags = CONV_SPLIT_BF16 | CONV_SPLIT_IL | CONV_SPLIT_ACT | tile_major
                        if pair_bwd and dy_amax is not None and not tile_major:
                            # V(dY) stays fp32; the pairs are formed in the GEMM at the scale of max |dY| / 256 (|B^T d B| <= 225 max|d|)
                            gm.flags |= CONV_PAIR_F16 | CONV_PAIR_AMAX
                            gm.w = self.pack_conv_wino_pair(conv, m, dgrad=True).data_ptr()
                            gm.scale = dy_amax
                        else:
                            gm.w = self.pack_conv_wino_split(conv, m, True, dgrad=True).data_ptr()
                    else:
                        gm.w = self.pack_conv_wino(conv, m, dgrad=True).data_ptr()
                        if -(-T // 128) * (C // 128) * nf <= 256:
                            gm.reserved_i = 64
                    bops.append(gm)
                    wo = XlOp()
                    wo.type, wo.ksize = XL_OP_WINO_OUT, m
                    tpb = 16
                    while tpb > 1 and B * -(-(Th * Tw) // tpb) * max(1, C // (256 if m == 6 else 512)) < 1024:
                        tpb //= 2
                    if m == 6 and C % 512 == 0 and not os.environ.get("XL_WINO_OUT_TPB16"):      # (as conv_wino: equal workgroups)
                        tpb = min(range(1, 17), key=lambda t: (-(-(B * -(-(Th * Tw) // t) * (C // 512)) // 512) * (t + 0.5), -t))
                    wo.B, wo.Hi, wo.Wi, wo.Cin, wo.ld_out, wo.groups = B, H, W, C, gx[1], 1
                    wo.nchunks, wo.reserved_i = -(-(Th * Tw) // tpb), tpb
                    wo.flags = (op.flags & CONV_ACCUMULATE) | tile_major
                    wo.in_, wo.out = Mb.data_ptr(), gx[0].data_ptr() + 4 * gx[2]
                    bops.append(wo)
                    self.release_grad(Vb); self.release_grad(Mb)
                    self.release_grad(dy)
                    continue
                op.B, op.Hi, op.Wi, op.Cin, op.Ho, op.Wo, op.Cout = B, Ho, Wo, Cout, H, W, C
                op.ksize, op.stride, op.ld_in, op.ld_out = k, s, Cout, gx[1]
                op.in_, op.out = dy.data_ptr(), gx[0].data_ptr() + 4 * gx[2]
                if (k == 3 and s == 2 and (Cout, C) in ((64, 32), (128, 64)) and not (op.flags & CONV_ACCUMULATE)
                        and gx[1] % 4 == 0 and gx[2] % 4 == 0 and self.split_train_ok()
                        and os.environ.get("XL_GEMM_SPLIT_BF16", self.SPLIT_DEFAULT) not in ("", "0", "1")
                        and not os.environ.get("XL_NO_S2_DGRAD")):
                    # round 4: the stem's data gradients on the split pipe, one launch over tiles of the result instead of four
                    # parity-class launches of the fp32 implicit GEMM (csrc/xl_stem_dgrad.hip)
                    op.type = XL_OP_S2_DGRAD
                    op.flags = 0
                    op.w = self.pack_s2_dgrad_fragments(conv).data_ptr()
                    queue = torch.zeros(4, dtype=torch.int32, device=dev)
                    self.keep.append(queue)
                    op.stats = queue.data_ptr()
                    bops.append(op)
                    self.release_grad(dy)
                    continue
                if (k == 1 and s == 1 and Cout % 32 == 0 and C % 256 == 0 and C <= 1024 and H * W >= 256
                        and gx[1] % 4 == 0 and gx[2] % 4 == 0 and self.split_train_ok()
                        and os.environ.get("XL_GEMM_SPLIT_BF16", self.SPLIT_DEFAULT) not in ("", "0", "1")
                        and not os.environ.get("XL_NO_SPLIT_1X1")):
                    # dX = dY W on the split pipe: a plain 1x1 "convolution" of dY with the transposed weight matrix, split
                    # once per weight version; a second producer of the gradient accumulates in the epilogue
                    op.flags = CONV_SPLIT_BF16 | CONV_SPLIT_IL | (op.flags & CONV_ACCUMULATE)
                    if pair_bwd and dy_amax is not None:
                        op.flags |= CONV_PAIR_F16 | CONV_PAIR_AMAX
                        op.w = self.pack_conv_1x1_pair(conv, transposed=True).data_ptr()
                        op.scale = dy_amax
                    else:
                        op.w = self.pack_conv_1x1_split(conv, transposed=True).data_ptr()
                    op.reserved_i = 256
                else:
                    op.w = self.pack_conv(conv, dgrad=True).data_ptr()
                bops.append(op)      # 32 result channels (conv2): the kernel masks the padded half of its 64-wide tile
                self.release_grad(dy)
            elif kind == "conv1":
                conv = e["conv"]
                rt, H, W, Cout, rld, roff = e["raw"]
                dy = graw.pop(self._key(e["raw"]), None)
                if dy is None:
                    continue
                op = XlOp()
                op.type = XL_OP_CONV1_WGRAD
                op.B, op.Hi, op.Wi, op.Cin, op.Cout, op.ld_aux = B, H, W, conv.in_channels, Cout, Cout
                op.aux = dy.data_ptr()
                fold = c1_fold.pop(self._key(e["raw"]), None)
                if fold is not None:                                # the GroupNorm-backward apply pass on load (see the "gn" branch)
                    gt, gld, goff = fold["dout"]
                    xt, xld, xoff = fold["x"]
                    op.aux, op.ld_aux = gt.data_ptr() + 4 * goff, gld
                    op.aux2, op.ld_in = xt.data_ptr() + 4 * xoff, xld
                    op.w, op.flags = fold["fco"].data_ptr(), fold["flags"]
                    patch_bco.append((len(bops), fold["bco_off"]))
                op.out = pgrad(conv.weight).data_ptr()
                # the bias gradient of conv1 comes from the GroupNorm backward sums (fp64 closed form); the sum this
                # kernel also produces goes to a scratch vector
                self.conv1_db_unused = torch.empty(Cout, dtype=torch.float32, device=dev)
                op.out2 = self.conv1_db_unused.data_ptr()
                op.reserved_i = 8                                   # image rows per workgroup
                scratch_f = max(scratch_f, B * ((H + 7) // 8) * 28 * Cout)
                patch_f.append(len(bops))
                self.conv1_wgrad_indices.append(len(bops))
                bops.append(op)
                if fold is None:
                    self.release_grad(dy)
                elif fold["release"] is not None:
                    self.release_grad(fold["release"])
        if params_list:
            import numpy as np
            dt = GNB_PARAMS_ITEM_DTYPE
            self.gnb_params_table = torch.from_numpy(np.array(params_list, dtype=dt).view(np.uint8).copy()).to(dev)
            op = XlOp()
            op.type, op.Cin, op.Cout = XL_OP_GNB_PARAMS_LIST, len(params_list), max(i[6] for i in params_list)
            op.in_ = self.gnb_params_table.data_ptr()
            bops.append(op)
        self.bwd_scratch_f = torch.empty(max(scratch_f, 1), dtype=torch.float32, device=dev)
        self.bwd_scratch_d = torch.empty(max(scratch_d, 1), dtype=torch.float64, device=dev)
        self.bwd_array = (XlOp * len(bops))(*bops)
        for i in patch_f:
            self.bwd_array[i].stats2 = self.bwd_scratch_f.data_ptr()
        for i in patch_d:
            self.bwd_array[i].stats2 = self.bwd_scratch_d.data_ptr()
        for i, off_d in patch_bco:                                  # conv1's folded apply: the coefficients its GNB_FINAL op left
            self.bwd_array[i].bias = self.bwd_scratch_d.data_ptr() + 8 * off_d

    GRAPH_MAX_BATCH = 8            # plans of at most this many frames replay their op list as one HIP graph (XL_CNN_GRAPH)

    def _graph_wanted(self, stream):
        env = os.environ.get("XL_CNN_GRAPH")
        if self.train or env == "0":
            return False
        return env == "1" or self.B <= self.GRAPH_MAX_BATCH

    def _run_graph(self, image, stream):
        """Latency path: the ~95 launches of a small-batch forward as one executable HIP graph.  The graph holds pointers,
        so the image is copied into a buffer of the plan (4 MB per frame, one device-to-device copy) and the result is
        handed out as a copy of the plan's result buffer.  Returns None when the eager path has to run (first call of the
        plan: every kernel configures itself on its first launch; per-op profiling)."""
        L = _bind()
        if not hasattr(self, "graph_in"):
            self.graph_in = torch.empty_like(image)
            self.graph_out = torch.empty(self.out_shape, dtype=torch.float32, device=self.device)
            self.graph, self.graph_runs, self.graph_stream, self.graph_off = None, 0, None, False
        if self.graph_off:
            return None
        self.graph_runs += 1
        if self.graph_runs == 1:
            return None                                     # warm-up: eager
        # one stream per plan: the graph's input / result buffers belong to the plan, and a launch on another stream could
        # overlap the copy-in of the next call or the copy-out of this one (callers that drive one network from several
        # streams pass distinct plan_slots).  A call from any other stream than the capturing one runs eagerly.
        if self.graph_stream is not None and stream != self.graph_stream:
            return None
        # round 5: the caller is on the DEFAULT stream - what the reference's unchanged loop uses (test_single_task.py:347,
        # `network(image.cuda())`), and a stream HIP cannot capture on.  The graph then lives on a private stream of the plan,
        # bracketed by events: copy-in on the caller's stream, graph behind it, the result copy back on the caller's stream
        # behind the graph.  Same kernels, same order, same bits as the eager op list.
        private = None
        if stream == 0:
            if not hasattr(self, "graph_private"):
                self.graph_private = torch.cuda.Stream(device=self.device)
            private = self.graph_private
        for i in self.image_op_indices:
            self.op_array[i].in_ = self.graph_in.data_ptr()
        self.op_array[self.out_op_index].out = self.graph_out.data_ptr()
        if self.graph is None:
            h = ctypes.c_void_p()
            rc = L.xl_cnn_graph_capture(self.op_array, len(self.op_array),
                                        ctypes.c_void_p(private.cuda_stream if private is not None else stream), ctypes.byref(h))
            if rc != 0:
                return self._graph_give_up("capture", rc)
            self.graph, self.graph_stream = h, stream
            weakref.finalize(self, L.xl_cnn_graph_destroy, h)
        self.graph_in.copy_(image)
        if private is not None and not getattr(self, "graph_on_null_stream", True):
            caller = torch.cuda.current_stream()
            private.wait_stream(caller)                     # the copy-in (and whatever produced the image) before the graph
            rc = L.xl_cnn_graph_launch(self.graph, ctypes.c_void_p(private.cuda_stream))
            caller.wait_stream(private)                     # the caller's next operation - the result copy - behind the graph
        elif private is not None:
            # round 6: a graph CAPTURED on the private stream may be LAUNCHED on the default stream - only capture is refused
            # there - so the copy-in, the graph and the result copy are simply in stream order: no events, four driver calls
            # fewer per frame.  A runtime that refuses falls back to the bracketed form for the life of the plan.
            rc = L.xl_cnn_graph_launch(self.graph, ctypes.c_void_p(0))
            if rc != 0 and rc != XL_ERR_UNSUPPORTED:
                self.graph_on_null_stream = False
                return self._run_graph_retry(image, stream)
        else:
            rc = L.xl_cnn_graph_launch(self.graph, ctypes.c_void_p(stream))
        if rc == XL_ERR_UNSUPPORTED:                        # per-op profiling is on: this call runs eagerly
            return None
        if rc != 0:
            return self._graph_give_up("launch", rc)
        return self.graph_out.clone()

    def _run_graph_retry(self, image, stream):
        self.graph_runs -= 1                                 # (this call was counted already)
        return self._run_graph(image, stream)

    def _graph_give_up(self, what, rc):
        """A capture / launch failure must not make small-batch inference unusable where the eager path works (a call inside the
        op list that cannot be captured, e.g. a first-use hipFuncSetAttribute on another device): say so once, stop trying."""
        import warnings
        L = _lib.lib()
        warnings.warn("crossloc_amd: HIP-graph %s of a %d-frame plan failed (%s %s); this plan runs its op list eagerly from now on"
                      % (what, self.B, L.xl_status_string(rc).decode(), L.xl_cnn_last_error().decode()))
        self.graph_off = True
        return None

    def run(self, image):
        stream = torch.cuda.current_stream().cuda_stream
        if self._graph_wanted(stream):
            out = self._run_graph(image, stream)
            if out is not None:
                self.last_image, self.last_out = image, out.detach()
                return out
        out = torch.empty(self.out_shape, dtype=torch.float32, device=self.device)
        for i in self.image_op_indices:
            self.op_array[i].in_ = image.data_ptr()
        self.op_array[self.out_op_index].out = out.data_ptr()
        _check(_bind().xl_cnn_run(self.op_array, len(self.op_array), ctypes.c_void_p(stream)))
        # (a detached alias: the returned tensor itself becomes the output of the autograd node in training, and a
        # reference to it from here would keep that graph - and the plan's busy token - alive)
        self.last_image, self.last_out = image, out.detach()
        return out

    def run_backward(self, dout):
        """dout [B, Cout, Ho, Wo] (NCHW, like the forward output).  Returns [(parameter, flat gradient)]."""
        dout = dout.detach().to(torch.float32).contiguous()
        hb = self.bwd_array[self.head_bwd_index]
        hb.aux, hb.aux2 = dout.data_ptr(), self.last_out.data_ptr()
        for i in self.conv1_wgrad_indices:
            self.bwd_array[i].in_ = self.last_image.data_ptr()
        stream = torch.cuda.current_stream().cuda_stream
        _check(_bind().xl_cnn_run(self.bwd_array, len(self.bwd_array), ctypes.c_void_p(stream)))
        # hand the gradients out as slices of a copy of the flat buffer (ONE device-to-device copy; the plan's own buffer
        # is overwritten by the next backward pass).  Two result buffers alternate - their addresses repeat, which keeps
        # the fused optimizer's pointer table valid - but a buffer is reused only when NOTHING else references its
        # storage any more: a .grad that is still alive (accumulation without zero_grad), a result of torch.autograd.grad,
        # a tensor kept by a hook or for logging all hold a view, and then a fresh buffer is handed out instead.
        if not hasattr(self, "grad_results"):
            self.grad_results, self.grad_turn = [None, None], 0
        self.grad_turn ^= 1
        buf = self.grad_results[self.grad_turn]
        if buf is not None and not _sole_owner(buf):
            buf = None
        if buf is None:
            buf = self.grad_results[self.grad_turn] = torch.empty_like(self.grad_flat)
        buf.copy_(self.grad_flat)
        return [(p, buf[o:o + n]) for p, o, n in self.grad_slices]


def _sole_owner(buf):
    """True when `buf` is the only tensor left on its storage (no view of it is alive anywhere): 2 = this tensor + the
    Python storage object the query itself creates.  Unknown (private API missing) counts as shared."""
    try:
        return torch._C._storage_Use_Count(buf.untyped_storage()._cdata) <= 2
    except (AttributeError, RuntimeError, TypeError):
        if not getattr(_sole_owner, "warned", False):
            _sole_owner.warned = True
            import warnings
            warnings.warn("crossloc_amd: torch._C._storage_Use_Count is unavailable in this PyTorch: every backward pass allocates a fresh "
                          "gradient buffer (correct, but the fused optimizer rebuilds its pointer tables each step)")
        return False


class _NetFunction(torch.autograd.Function):
    """Connects the HIP forward/backward plans to autograd: the parameters are passed as inputs so that
    `loss.backward()` (train_single_task.py:298) routes the output gradient into run_backward and accumulates the
    returned parameter gradients into `.grad` like any other op.  The image gradient is not computed."""

    @staticmethod
    def forward(ctx, plan, x, *params):
        ctx.plan, ctx.params = plan, params
        out = plan.run(x)
        # the plan's buffers hold this graph's activations until its backward ran or the graph is dropped
        plan.generation = gen = getattr(plan, "generation", 0) + 1
        plan.busy = True
        ctx.generation = gen
        ctx.token = _Token()
        weakref.finalize(ctx.token, _release_plan, weakref.ref(plan), gen)
        return out

    @staticmethod
    def backward(ctx, gout):
        if ctx.generation != ctx.plan.generation:
            raise RuntimeError("the activations of this graph were overwritten by a later forward of the same plan "
                               "(backward twice through one graph after another forward?)")
        produced = {id(p): g for p, g in ctx.plan.run_backward(gout)}
        ctx.plan.busy = False
        grads = tuple(produced[id(p)].view_as(p) if (p.requires_grad and id(p) in produced) else None
                      for p in ctx.params)
        return (None, None) + grads


class _Token:
    """Lives as long as the autograd node of one training forward (see _NetFunction.forward)."""


def _release_plan(plan_ref, generation):
    plan = plan_ref()
    if plan is not None and getattr(plan, "generation", 0) == generation:
        plan.busy = False


class _Dummy:
    @staticmethod
    def data_ptr():
        return 0


_DUMMY = _Dummy()


class TransPoseNet(nn.Module):
    """Drop-in for networks.py:362-502.  forward(inputs[B,C,H,W] on the GPU) -> [B, n_task+n_pos, H/8, W/8], or
    [B, n_task+n_pos, H, W] with full_size_output (the DUC / semantics head, inference only)."""

    def __init__(self, mean, tiny, grayscale, enc_add_res_block=0, dec_add_res_block=0, num_task_channel=3,
                 num_pos_channel=1, num_gn_channel=32, num_mlr=0, num_unfrozen_encoder=0, full_size_output=False):
        super().__init__()
        # True: inference / training plans run separate per-image GroupNorm statistics passes, which makes every frame's
        # result bitwise independent of the batch it is in (default: conv-epilogue statistics, ~4 % faster, equal to
        # the last fp32 bit or two).  Set before the first forward.
        self.batch_invariant = False
        mean = torch.as_tensor(mean, dtype=torch.float32)
        self.register_buffer('mean', mean.clone())
        self.tiny, self.grayscale = tiny, grayscale
        self.enc_add_res_block, self.dec_add_res_block = enc_add_res_block, dec_add_res_block
        self.num_task_channel, self.num_pos_channel = num_task_channel, num_pos_channel
        self.num_gn_channel, self.num_mlr, self.full_size_output = num_gn_channel, num_mlr, full_size_output
        self.OUTPUT_SUBSAMPLE = 1 if full_size_output else 8
        if num_mlr == 0:
            self.encoder = TransPoseNetEncoder(tiny, grayscale, enc_add_res_block, num_gn_channel)
            self.encoder_ls = [self.encoder]
            self.mlr_encoder_ls = [nn.Identity()]
            self.mlr_norm, self.mlr_forward, self.mlr_skip = nn.Identity(), nn.Identity(), nn.Identity()
        else:
            assert isinstance(num_mlr, int) and 0 <= num_unfrozen_encoder <= num_mlr
            self.encoder = nn.Identity()
            self.encoder_ls = [self.encoder]
            self.mlr_encoder_ls = [TransPoseNetEncoder(tiny, grayscale, enc_add_res_block, num_gn_channel)
                                   for _ in range(num_mlr)]
            for i, block in enumerate(self.mlr_encoder_ls):
                if i >= num_unfrozen_encoder:                       # networks.py:424-428
                    for p in block.parameters():
                        p.requires_grad = False
                self.add_module('mlr_encoder_{:d}'.format(i + 1), block)
            self.mlr_norm = nn.GroupNorm(num_gn_channel, (512, 128)[tiny] * num_mlr)
            self.mlr_forward = _create_mlr_concatenator(num_mlr, tiny, num_gn_channel)
            self.mlr_skip = _create_mlr_skip_layer(num_mlr, tiny, num_gn_channel)
        self.decoder = TransPoseNetDecoder(mean, tiny, dec_add_res_block, num_task_channel, num_pos_channel,
                                           num_gn_channel, full_size_output)
        self.decoder_ls = [self.decoder]
        self._plans = {}
        self._plan_version = None

    def _tensors(self):
        """(parameters, version) - the parameter list of the module tree and the sum of all parameter / buffer version counters.
        `nn.Module.parameters()` / `.buffers()` walk the tree with de-duplicating generators: 0.35 ms of host time per forward for this
        network (versions + the parameter list) - a sixth of a single frame's 2 ms in the reference's per-frame loop (round 6), now 0.02.  The (owner, name, tensor) triples
        are cached; every call re-checks each slot by identity (a replaced parameter object, a moved module: the cache is rebuilt),
        which is a dictionary look-up per tensor instead of a tree walk."""
        cache = self.__dict__.get("_tensor_cache")
        if cache is not None:
            ver = 0
            for owner, name, t, is_param in cache:
                if (owner._parameters if is_param else owner._buffers).get(name) is not t:
                    cache = None
                    break
                ver += t._version
            if cache is not None:
                return self.__dict__["_param_list"], ver
        cache, seen = [], set()
        for mod in self.modules():
            for name, t in mod._parameters.items():
                if t is not None and id(t) not in seen:
                    seen.add(id(t)); cache.append((mod, name, t, True))
            for name, t in mod._buffers.items():
                if t is not None and id(t) not in seen:
                    seen.add(id(t)); cache.append((mod, name, t, False))
        self.__dict__["_tensor_cache"] = cache
        self.__dict__["_param_list"] = list(self.parameters())            # (the order autograd's inputs are bound in)
        return self.__dict__["_param_list"], sum(t._version for _, _, t, _ in cache)

    def _version(self):
        return self._tensors()[1]

    def invalidate(self):
        """Drop cached plans (packed weights); called automatically when a parameter changes in place."""
        self._plans = {}

    def _apply(self, fn, *a, **k):
        self._plans = {}
        self.__dict__.pop("_tensor_cache", None)
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._plans = {}
        self.__dict__.pop("_tensor_cache", None)
        return super().load_state_dict(*a, **k)

    def forward(self, inputs, plan_slot=0):
        """plan_slot: calls that may be in flight at the same time on different HIP streams (evaluation.PipelinedLocalizer
        runs two half-batches concurrently) must use different slots: a plan owns its activation buffers."""
        if not isinstance(inputs, torch.Tensor) or inputs.dim() != 4:
            raise RuntimeError("TransPoseNet.forward expects a 4D tensor [B,C,H,W]")
        if not inputs.is_cuda:
            raise RuntimeError("crossloc_amd.TransPoseNet runs on the MI355X only (no CPU fallback); got a CPU tensor")
        x = inputs.detach().to(torch.float32).contiguous()
        B, C, H, W = x.shape
        if C != (1 if self.grayscale else 3):
            raise RuntimeError("expected %d input channels, got %d" % (1 if self.grayscale else 3, C))
        params, ver = self._tensors()
        if ver != self._plan_version:
            for plan in self._plans.values():
                plan.refresh_weights()
            self._plan_version = ver
        train = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        # The conv kernel addresses a tensor with 32-bit byte offsets; inference launches whose tensors pass 2 GiB (the
        # 32-channel full-resolution activation does at 48 frames of 480x720) are issued per image range inside the
        # library (launch_igemm), so inference batches have no limit here.  The backward kernels are not segmented.
        max_b = max(1, (2 ** 31 - 1) // (H * W * self.num_gn_channel * 4) - 1)
        if B > max_b and train:
            raise RuntimeError("training batch of %d frames exceeds the per-launch limit of %d at %dx%d" % (B, max_b, H, W))
        key = (B, H, W, x.device.index, train) if plan_slot == 0 else (B, H, W, x.device.index, train, plan_slot)
        if getattr(self, "batch_invariant", False):
            key = key + ("batch_invariant",)
        with torch.cuda.device(x.device):
            plan = self._plans.get(key)
            if plan is None:
                plan = _Plan(self, B, H, W, x.device, train=train)
                self._plans[key] = plan
            if not train:
                return plan.run(x)
            # a training plan holds the activations of ONE graph until its backward ran (or the graph was dropped): a
            # second grad-enabled forward before that (gradient accumulation over two forwards, two losses) gets a
            # plan of its own instead of silently overwriting the first graph's activations
            n = 0
            while getattr(plan, "busy", False):
                n += 1
                if n >= 4:
                    raise RuntimeError("4 training forwards of shape %s are outstanding without a backward; run "
                                       "inference under torch.no_grad()" % (tuple(x.shape),))
                k2 = key + ("outstanding", n)
                plan = self._plans.get(k2)
                if plan is None:
                    plan = self._plans[k2] = _Plan(self, B, H, W, x.device, train=True)
            return _NetFunction.apply(plan, x, *params)

