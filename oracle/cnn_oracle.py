"""CPU fp32 restatement of the reference network graph — TEST INFRASTRUCTURE ONLY (see oracle/dsac_oracle.c).

Follows /root/reference/networks/networks.py: encoder :221-256, residual block :133-146 and its callers
:252-254 / :332-334, decoder :319-360 (incl. the full-size DUC head :259-273), MLR fusion :483-494, written as a pure function of a state_dict with the
reference's key names.  Pinned against tests/golden/net_forward.npz and semantics.npz (outputs of the imported reference).
The floating-point kernels of the product path are compared with this at fp32 tolerances.
"""
import torch
import torch.nn.functional as F


def _cgr(sd, x, conv, norm, stride=1, groups=32, relu=True):
    w = sd[conv + ".weight"]
    y = F.conv2d(x, w, sd[conv + ".bias"], stride=stride, padding=w.shape[-1] // 2)
    y = F.group_norm(y, min(groups, w.shape[0]), sd[norm + ".weight"], sd[norm + ".bias"], eps=1e-5)
    return F.relu(y) if relu else y


def _res_block(sd, prefix, x, groups):
    """_create_res_block (networks.py:133-146): conv3x3-GN-ReLU, conv1x1-GN-ReLU, conv3x3-GN-ReLU."""
    for c, n in ((0, 1), (3, 4), (6, 7)):
        x = _cgr(sd, x, "%s.%d" % (prefix, c), "%s.%d" % (prefix, n), groups=groups)
    return x


def encoder_forward(sd, x, prefix="encoder", enc_add=2, groups=32):
    p = prefix + "."
    x = _cgr(sd, x, p + "conv1", p + "norm1", groups=groups)
    x = _cgr(sd, x, p + "conv2", p + "norm2", 2, groups)
    x = _cgr(sd, x, p + "conv3", p + "norm3", 2, groups)
    res = _cgr(sd, x, p + "conv4", p + "norm4", 2, groups)
    x = _cgr(sd, res, p + "res1_conv1", p + "res1_norm1", groups=groups)
    x = _cgr(sd, x, p + "res1_conv2", p + "res1_norm2", groups=groups)
    x = _cgr(sd, x, p + "res1_conv3", p + "res1_norm3", groups=groups)
    res = F.relu(res + x)
    x = _cgr(sd, res, p + "res2_conv1", p + "res2_norm1", groups=groups)
    x = _cgr(sd, x, p + "res2_conv2", p + "res2_norm2", groups=groups)
    x = _cgr(sd, x, p + "res2_conv3", p + "res2_norm3", groups=groups)
    if (p + "res2_skip.weight") in sd:
        res = _cgr(sd, res, p + "res2_skip", p + "res2_skip_norm", groups=groups, relu=False)
    res = F.relu(res + x)
    for i in range(enc_add):
        res = F.relu(res + _res_block(sd, p + "enc_add_res_block%d" % (i + 1), res, groups))
    return res


def decoder_forward(sd, res, dec_add=2, n_task=3, n_pos=1, groups=32, up_hw=None):
    p = "decoder."
    for i in range(dec_add):
        res = F.relu(res + _res_block(sd, p + "dec_add_res_block%d" % (i + 1), res, groups))
    x = _cgr(sd, res, p + "res3_conv1", p + "res3_norm1", groups=groups)
    x = _cgr(sd, x, p + "res3_conv2", p + "res3_norm2", groups=groups)
    x = _cgr(sd, x, p + "res3_conv3", p + "res3_norm3", groups=groups)
    res = F.relu(res + x)
    sc = _cgr(sd, res, p + "fc1", p + "fc1_norm", groups=groups)
    sc = _cgr(sd, sc, p + "fc2", p + "fc2_norm", groups=groups)
    if (p + "duc_upsample.conv.weight") in sd:
        # full_size_output (networks.py:259-273, 344-349): DUC conv-GN-ReLU, pixel shuffle x8, bilinear trim
        sc = _cgr(sd, sc, p + "duc_upsample.conv", p + "duc_upsample.norm", groups=groups)
        sc = F.pixel_shuffle(sc, 8)
        sc = F.interpolate(sc, up_hw, mode="bilinear", align_corners=False)
    sc = F.conv2d(sc, sd[p + "fc3.weight"], sd[p + "fc3.bias"])
    task = sc[:, :n_task] + sd[p + "mean"][None, :, None, None]
    if n_pos:
        pos = torch.exp(F.hardtanh(sc[:, n_task:], min_val=-16.10, max_val=13.82))
        return torch.cat([task, pos], dim=1)
    return task


def transposenet_forward(sd, x, num_mlr=0, enc_add=2, dec_add=2, n_task=3, n_pos=1, groups=32):
    sd = {k: v.detach().float().cpu() for k, v in sd.items()}
    x = x.detach().float().cpu()
    if num_mlr == 0:
        res = encoder_forward(sd, x, "encoder", enc_add, groups)
    else:
        mlr = torch.cat([encoder_forward(sd, x, "mlr_encoder_%d" % (i + 1), enc_add, groups)
                         for i in range(num_mlr)], dim=1)
        res = _cgr(sd, mlr, "mlr_skip.0", "mlr_skip.1", groups=groups, relu=False)
        mlr = F.group_norm(mlr, groups, sd["mlr_norm.weight"], sd["mlr_norm.bias"], eps=1e-5)
        mlr = _res_block(sd, "mlr_forward", mlr, groups)
        res = F.relu(res + mlr)
    return decoder_forward(sd, res, dec_add, n_task, n_pos, groups, up_hw=tuple(x.shape[2:4]))
