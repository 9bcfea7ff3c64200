"""GPU parity of the fused loss kernels (through the C ABI / crossloc_amd.loss) against the golden outputs of
the imported reference losses and, at the BASELINE batch size [16,*,60,90], against the CPU restatement.
Tolerances: loss value 1e-4 relative, gradients 1e-3 relative (+1e-4 of the largest gradient).  They are set
by the fixture's conditioning, not by the kernels: scene coordinates are O(500 m) in fp32, so the 3-D
distance ||P X - P X_gt|| of a ~3 m error carries ~3e-5 relative rounding noise whatever the op order (the
reference's own value moves by that much between bmm implementations), and one clamped-sigma cell
amplifies it into the loss."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from crossloc_amd import loss as xl_loss, synth      # noqa: E402
from oracle import loss_oracle                        # noqa: E402

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "losses.npz"))


def _c(name, grad=False):
    return torch.tensor(G[name], device="cuda", requires_grad=grad)


def _check(loss, rate, p, u, tag, mle):
    loss.backward()
    assert loss.item() == pytest.approx(float(G[tag + "_loss"]), rel=1e-4)
    assert float(rate) == pytest.approx(float(G[tag + "_rate"]), abs=1e-6)
    gp = G[tag + "_dpred"].copy()
    got = p.grad.cpu().numpy().copy()
    if tag.startswith("coord"):
        # cell (1,:,5,5) is the planted exact hit pred == gt: its reprojection residual is 0 up to rounding, so
        # the direction rho/|rho| of its gradient is undefined (pure rounding noise in the reference as well)
        gp[1, :, 5, 5] = 0
        got[1, :, 5, 5] = 0
    assert np.allclose(got, gp, rtol=1e-3, atol=1e-4 * np.abs(gp).max())
    if mle:
        gu = G[tag + "_dunc"]
        assert np.allclose(u.grad.cpu().numpy(), gu, rtol=1e-3, atol=1e-4 * np.abs(gu).max())


@pytest.mark.parametrize("tag,mode,soft,hard", [("coord_MLE", "MLE", 100.0, 1000.0), ("coord_plain", None, 100.0, 1000.0),
                                                 ("coord_tight", "MLE", 20.0, 60.0)])
def test_coord_vs_reference_golden(tag, mode, soft, hard):
    H, W = (int(v) for v in G["coord_hw"])
    p, u = _c("coord_pred", True), _c("coord_unc", True)
    loss, rate = xl_loss.scene_coords_regression_loss(0.1, soft, hard, 50.0, mode, xl_loss.get_pixel_grid(8), -1,
                                                      xl_loss.get_cam_mat(W, H, float(G["coord_focal"])),
                                                      p, u, _c("coord_poses"), _c("coord_gt"))
    _check(loss, rate, p, u, tag, mode == "MLE")


@pytest.mark.parametrize("tag,mode", [("depth_MLE", "MLE"), ("depth_plain", None)])
def test_depth_vs_reference_golden(tag, mode):
    p, u = _c("depth_pred", True), _c("coord_unc", True)
    loss, rate = xl_loss.depth_regression_loss(0.1, 10.0, mode, -1, p, u, _c("depth_gt"))
    _check(loss, rate, p, u, tag, mode == "MLE")


@pytest.mark.parametrize("tag,mode", [("normal_MLE", "MLE"), ("normal_plain", None)])
def test_normal_vs_reference_golden(tag, mode):
    p, u = _c("normal_logits", True), _c("coord_unc", True)
    loss, rate = xl_loss.normal_regression_loss(10.0, mode, -1, p, u, _c("normal_gt"))
    _check(loss, rate, p, u, tag, mode == "MLE")


def _full_batch():
    B = 16
    coords, gt, poses = synth.make_batch(700, B, noise=3.0, outlier_ratio=0.1)
    rng = np.random.default_rng(1)
    unc = np.exp(rng.uniform(-2, 3, size=(B, 1, 60, 90))).astype(np.float32)
    return coords, gt, poses.astype(np.float32), unc


@pytest.mark.parametrize("reduction", ["mean", None])
def test_coord_full_size_vs_oracle(reduction):
    coords, gt, poses, unc = _full_batch()
    pc, uc = torch.tensor(coords, requires_grad=True), torch.tensor(unc, requires_grad=True)
    lo, ro = loss_oracle.coord_loss(pc, uc, torch.tensor(poses), torch.tensor(gt), 480.0, 360.0, 240.0, 8.0, reduction=reduction)
    w = torch.linspace(0.5, 1.5, 16)
    (lo * w).sum().backward() if reduction is None else lo.backward()
    pg, ug = torch.tensor(coords, device="cuda", requires_grad=True), torch.tensor(unc, device="cuda", requires_grad=True)
    lg, rg = xl_loss.scene_coords_regression_loss(0.1, 100.0, 1000.0, 50.0, "MLE", xl_loss.get_pixel_grid(8), -1,
                                                  xl_loss.get_cam_mat(720, 480, 480.0), pg, ug,
                                                  torch.tensor(poses, device="cuda"), torch.tensor(gt, device="cuda"),
                                                  reduction=reduction)
    (lg * w.cuda()).sum().backward() if reduction is None else lg.backward()
    assert torch.allclose(lg.detach().cpu(), lo.detach(), rtol=1e-4)
    assert float(rg) == pytest.approx(ro, abs=1e-6)
    assert torch.allclose(pg.grad.cpu(), pc.grad, rtol=2e-3, atol=1e-4 * pc.grad.abs().max().item())
    assert torch.allclose(ug.grad.cpu(), uc.grad, rtol=2e-3, atol=1e-4 * uc.grad.abs().max().item())


def test_depth_normal_full_size_vs_oracle():
    rng = np.random.default_rng(2)
    B = 16
    unc = np.exp(rng.uniform(-2, 3, size=(B, 1, 60, 90))).astype(np.float32)
    gd = rng.uniform(100, 300, size=(B, 1, 60, 90)).astype(np.float32)
    pd = gd + rng.normal(0, 4, size=gd.shape).astype(np.float32)
    gd[rng.uniform(size=gd.shape) < 0.1] = -1.0
    gn = rng.normal(size=(B, 3, 60, 90)).astype(np.float32)
    gn /= np.linalg.norm(gn, axis=1, keepdims=True)
    gn[:, :, rng.uniform(size=(60, 90)) < 0.05] = -1.0
    lg_ = rng.normal(0, 2, size=(B, 2, 60, 90)).astype(np.float32)
    for name, fo, fg, pred, gt, kw in [
            ("depth", loss_oracle.depth_loss, lambda p, u, g: xl_loss.depth_regression_loss(0.1, 10.0, "MLE", -1, p, u, g), pd, gd, {}),
            ("normal", loss_oracle.normal_loss, lambda p, u, g: xl_loss.normal_regression_loss(10.0, "MLE", -1, p, u, g), lg_, gn, {})]:
        pc, uc = torch.tensor(pred, requires_grad=True), torch.tensor(unc, requires_grad=True)
        lo, ro = fo(pc, uc, torch.tensor(gt))
        lo.backward()
        pg, ug = torch.tensor(pred, device="cuda", requires_grad=True), torch.tensor(unc, device="cuda", requires_grad=True)
        l2, r2 = fg(pg, ug, torch.tensor(gt, device="cuda"))
        l2.backward()
        assert l2.item() == pytest.approx(lo.item(), rel=1e-4), name
        assert float(r2) == pytest.approx(ro, abs=2e-5), name
        assert torch.allclose(pg.grad.cpu(), pc.grad, rtol=2e-3, atol=1e-4 * pc.grad.abs().max().item()), name
        assert torch.allclose(ug.grad.cpu(), uc.grad, rtol=2e-3, atol=1e-4 * uc.grad.abs().max().item()), name


def test_all_invalid_batch_gates_reprojection_term():
    """coord.py:141: with no valid cell in the batch the reprojection term is skipped entirely."""
    coords, gt, poses, unc = _full_batch()
    far = coords[:2] + 1.0e5                                   # every cell fails the hard clamp / tolerance
    args = (torch.tensor(poses[:2]), torch.tensor(gt[:2]))
    lo, ro = loss_oracle.coord_loss(torch.tensor(far), torch.tensor(unc[:2]), *args, 480.0, 360.0, 240.0, 8.0)
    lg, rg = xl_loss.scene_coords_regression_loss(0.1, 100.0, 1000.0, 50.0, "MLE", xl_loss.get_pixel_grid(8), -1,
                                                  xl_loss.get_cam_mat(720, 480, 480.0), torch.tensor(far, device="cuda"),
                                                  torch.tensor(unc[:2], device="cuda"), args[0].cuda(), args[1].cuda())
    assert ro == 0.0 and float(rg) == 0.0
    assert lg.item() == pytest.approx(lo.item(), rel=1e-4)
