"""CPU restatement of the three per-pixel losses in plain torch (autograd supplies the gradients) —
TEST INFRASTRUCTURE ONLY.  Follows /root/reference/loss/coord.py:87-188, loss/depth.py:7-76,
loss/normal.py:8-127, loss/semantics.py:10-18, 44-91 and utils/learning.py:49-71, 401-440; pinned against
tests/golden/losses.npz and semantics.npz
(values and autograd gradients captured from the imported reference)."""
import math

import torch


def _valid(lab, nodata):
    return (lab == nodata).sum(dim=1) == 0


def _mle(err_sq, sigma, k):
    s = sigma.clamp(min=1e-7)
    return k * torch.log(s) + err_sq.clamp(min=1e-7) / (2.0 * s.square().clamp(min=1e-7))


def _reduce(per_cell, valid_mask, reduction):
    B, N = per_cell.shape
    rate = valid_mask.sum().item() / (B * N)
    if reduction is None:
        return per_cell.sum(dim=1) / N, rate
    return per_cell.sum() / (B * N), rate


def coord_loss(pred, sigma, gt_poses, gt, focal, cx, cy, sub, min_depth=0.1, soft=100.0, hard=1000.0, tol=50.0,
               nodata=-1.0, mle=True, reduction='mean'):
    B, _, H, W = pred.shape
    X = pred.reshape(B, 3, -1)
    G = gt.reshape(B, 3, -1)
    P = torch.linalg.inv(gt_poses)[:, :3, :]
    one = torch.ones(B, 1, X.shape[2])
    Xc = torch.bmm(P, torch.cat([X, one], 1))
    Gc = torch.bmm(P, torch.cat([G, one], 1))
    d = torch.norm(Xc - Gc, dim=1)
    K = torch.tensor([[focal, 0, cx], [0, focal, cy], [0, 0, 1.0]])
    p = torch.bmm(K.expand(B, 3, 3), Xc)
    z = torch.clamp(p[:, 2:], min=min_depth)
    uv = p[:, :2] / z
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    grid = torch.stack([xs * sub + sub / 2, ys * sub + sub / 2]).reshape(2, -1)
    e = (uv - grid[None]).norm(dim=1).clamp(min=1e-7)
    g = _valid(G, nodata)
    m = ~(Xc[:, 2] < min_depth) & ~(e > hard) & ~((d > tol) & g)
    lr = 0
    if m.sum() > 0:
        ep = e * m
        l1 = (ep * (ep <= soft)).clamp(min=1e-7)
        lq = torch.sqrt(soft * (ep * (ep > soft)).clamp(min=1e-7) + 1e-7).clamp(min=1e-7)
        lr = l1 + lq
    lu = _mle(d.square(), sigma.reshape(B, -1), 3.0) if mle else d
    return _reduce(lu * g + lr, m, reduction)


def depth_loss(pred, sigma, gt, min_depth=0.1, hard=10.0, nodata=-1.0, mle=True, reduction='mean'):
    B = pred.shape[0]
    D, G = pred.reshape(B, -1), gt.reshape(B, -1)
    err = (D - G).abs()
    g = G != nodata
    valid = ~(D < min_depth) & ~(err > hard) & g
    l = _mle(err.square(), sigma.reshape(B, -1), 1.0) if mle else err
    return _reduce(l * g, valid, reduction)


def normal_loss(logits, sigma, gt, hard=10.0, nodata=-1.0, mle=True, reduction='mean'):
    B = logits.shape[0]
    L, G = logits.reshape(B, 2, -1), gt.reshape(B, 3, -1)
    ae = (torch.sigmoid(L).clamp(min=1e-7, max=1 - 1e-7) * 2 - 1.0) * math.pi
    az_g = torch.atan2(G[:, 1], G[:, 0])
    el_g = torch.atan2(G[:, 2], torch.norm(G[:, 0:2], dim=1))
    dl = (az_g - ae[:, 0]).abs()
    E = (2.0 * torch.min(dl, 2.0 * math.pi - dl).abs() + (ae[:, 1] - el_g).abs()).clamp(min=1e-7)
    g = _valid(G, nodata)
    a = ae.detach()
    xyz = torch.stack([torch.cos(a[:, 0]) * torch.cos(a[:, 1]), torch.sin(a[:, 0]) * torch.cos(a[:, 1]), torch.sin(a[:, 1])], 1)
    xyz = torch.nn.functional.normalize(xyz, dim=1)
    cs = torch.nn.functional.cosine_similarity(xyz, G, dim=1).clamp(min=-1 + 1e-7, max=1 - 1e-7)
    valid = ~(torch.acos(cs) / math.pi * 180.0 > hard) & g
    l = _mle(E.square(), sigma.reshape(B, -1), 2.0) if mle else E
    return _reduce(l * g, valid, reduction)


def semantics_loss(logits, labels, reduction='mean'):
    """loss/semantics.py:44-91 with CrossEntropyLoss2d (:10-18, no class weights): per-pixel -log softmax at the
    label; loss = sum / pixels ('mean') or per image (None); rate = share of pixels whose arg-max class (first maximum)
    equals the label.  labels: [B,1,H,W] (any numeric dtype)."""
    B = logits.shape[0]
    lab = labels.squeeze(1).long()
    ce = torch.nn.functional.nll_loss(torch.log_softmax(logits, dim=1), lab, reduction='none')     # [B,H,W]
    pred = torch.argmax(torch.log_softmax(logits.detach(), dim=1), dim=1)
    rate = (pred == lab).sum().item() / lab.numel()
    per = ce.reshape(B, -1).sum(dim=1)
    if reduction is None:
        return per / lab[0].numel(), rate
    return per.sum() / lab.numel(), rate
