"""The BASELINE sizes and the backward pass pinned to the REFERENCE ITSELF (round 4).

tests/golden/full_size.npz, grads.npz and losses_b16.npz hold what the imported reference modules computed
(tests/golden/make_golden.py: networks/networks.py:466-502 at 480x720; train_single_task.py:262-298 forward + MLE
coordinate loss + loss.backward() on the reference TransPoseNet; loss/coord.py, depth.py, normal.py at [16,*,60,90]); the
inputs are regenerated from seeds on both sides (tests/golden/golden_inputs.py, checksums in the fixtures).

CPU tests (not gpu): the oracle restatements against the fixtures - they pin the checker at these sizes too.
GPU tests: the HIP path through crossloc_amd against the fixtures; cnn_oracle / loss_oracle are not in that loop.

Tolerances.  Forward at 480x720: <= 2e-4 of the coordinate range (SURVEY 8(d) asks 1e-4 relative of the output for the fp32
path; measured 6e-5 m = one fp32 ulp at |X| ~ 500 m), sigma 2e-3 relative.  Parameter gradients: the fixture comes from a
float64 evaluation of the reference module (no ReLU-mask flips of its own); the fp32 forward on the GPU may put a
pre-activation that is within an ulp of 0 on the other side, which moves single elements - so every tensor is held to a
relative L2 error (<= 5e-2; typical 1e-5) on its strided sample, and its L2 norm to 2e-2.  Losses: 1e-4 on the value,
gradient moments 1e-3, samples 2e-3 relative + 1e-4 of the largest element against the float64 evaluation of the reference
functions, plus twice the reference's own fp32-to-float64 distance at that element (fp32 cancellation in |PX - PX_gt| at 500 m).
"""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import golden_inputs                                            # noqa: E402

from crossloc_amd import networks                               # noqa: E402
from crossloc_amd.weights import seeded_state_dict              # noqa: E402

MEAN = torch.tensor([-455.934, 417.50, 520.31])
FULL = np.load(os.path.join(HERE, "golden", "full_size.npz"))
GRADS = np.load(os.path.join(HERE, "golden", "grads.npz"))
L16 = np.load(os.path.join(HERE, "golden", "losses_b16.npz"))


def _net(num_mlr=0):
    net = networks.TransPoseNet(MEAN, False, False, 2, 2, 3, 1, 32, num_mlr, 0, False)
    net.load_state_dict(seeded_state_dict(net, seed=2021))
    return net


def _check_forward(y, ref, tol_range=2e-4):
    y, ref = y.double(), torch.from_numpy(np.asarray(ref)).double()
    rng = (ref[:, :3] - MEAN[None, :, None, None].double()).abs().max().item()
    err = (y[:, :3] - ref[:, :3]).abs().max().item()
    assert err <= tol_range * max(1.0, rng), (err, rng)
    assert torch.allclose(y[:, 3], ref[:, 3], rtol=2e-3), ((y[:, 3] - ref[:, 3]).abs() / ref[:, 3]).max()
    return err


# ------------------------------------------------------------------------------------------ CPU: the oracle at these sizes

def test_inputs_regenerate():
    for tag in ("single", "mlr3"):
        assert golden_inputs.checksum(golden_inputs.full_size_image(tag)) == float(FULL[tag + "_x_checksum"])
    x, poses, _ = golden_inputs.grad_inputs()
    assert golden_inputs.checksum(x) == float(GRADS["x_checksum"])
    assert golden_inputs.checksum(poses) == float(GRADS["poses_checksum"])
    I = golden_inputs.loss_b16_inputs()
    assert [golden_inputs.checksum(I[k]) for k in sorted(I)] == list(L16["input_checksums"])


def test_cnn_oracle_at_480x720_vs_reference_fixture():
    from oracle import cnn_oracle
    y = cnn_oracle.transposenet_forward(seeded_state_dict(_net(), seed=2021),
                                        torch.from_numpy(golden_inputs.full_size_image("single")), 0, 2, 2)
    assert y.shape == (1, 4, 60, 90)
    _check_forward(y, FULL["single_y"])


def _loss16(mod, on_gpu):
    """The three losses x {MLE, plain} at [16,*,60,90] through `mod` (loss_oracle on CPU / crossloc_amd.loss on the GPU):
    yields (tag, loss, rate, dpred, dunc)."""
    I = golden_inputs.loss_b16_inputs()
    dev = "cuda" if on_gpu else "cpu"

    def t(name, grad=False):
        return torch.tensor(I[name], device=dev, requires_grad=grad)
    for mode in ("MLE", None):
        for task in ("coord", "depth", "normal"):
            pname = {"coord": "pred", "depth": "depth_pred", "normal": "normal_logits"}[task]
            p, u = t(pname, True), t("unc", True)
            if on_gpu:
                if task == "coord":
                    loss, rate = mod.scene_coords_regression_loss(0.1, 100.0, 1000.0, 50.0, mode, mod.get_pixel_grid(8), -1,
                                                                  mod.get_cam_mat(720, 480, 480.0), p, u, t("poses"), t("gt"))
                elif task == "depth":
                    loss, rate = mod.depth_regression_loss(0.1, 10.0, mode, -1, p, u, t("depth_gt"))
                else:
                    loss, rate = mod.normal_regression_loss(10.0, mode, -1, p, u, t("normal_gt"))
            else:
                if task == "coord":
                    loss, rate = mod.coord_loss(p, u, t("poses"), t("gt"), 480.0, 360.0, 240.0, 8.0, mle=mode == "MLE")
                elif task == "depth":
                    loss, rate = mod.depth_loss(p, u, t("depth_gt"), mle=mode == "MLE")
                else:
                    loss, rate = mod.normal_loss(p, u, t("normal_gt"), mle=mode == "MLE")
            loss.backward()
            dp = p.grad.detach().double().cpu().numpy()
            du = u.grad.detach().double().cpu().numpy() if u.grad is not None else np.zeros(tuple(u.shape))
            yield "%s_%s" % (task, mode or "plain"), float(loss.item()), float(rate), dp, du


def _check_loss16(mod, on_gpu, rel_loss, rel_mom, rel_s, abs_s=1e-4):
    for tag, loss, rate, dp, du in _loss16(mod, on_gpu):
        assert loss == pytest.approx(float(L16[tag + "_loss"]), rel=rel_loss), tag
        assert rate == pytest.approx(float(L16[tag + "_rate"]), abs=2e-5), tag
        for nm, g in (("dpred", dp), ("dunc", du)):
            ref_m, ref_s = L16["%s_%s_moments" % (tag, nm)], L16["%s_%s_sample" % (tag, nm)].astype(np.float64)
            got_m = np.array([g.sum(), np.abs(g).sum(), np.sqrt((g * g).sum())])
            # (the signed sum cancels: it is held to the absolute sum's scale)
            assert abs(got_m[0] - ref_m[0]) <= rel_mom * max(ref_m[1], 1e-30), (tag, nm, got_m, ref_m)
            assert np.allclose(got_m[1:], ref_m[1:], rtol=rel_mom, atol=1e-30), (tag, nm, got_m, ref_m)
            s = g[:, :, ::4, ::5]
            # element-wise against the float64 evaluation of the reference functions, allowing - on top of the tolerance - twice
            # the distance the reference's OWN fp32 result keeps from it at that element (a near-cancelling sum of the distance
            # and reprojection terms leaves single elements of the fp32 reference 2e-2 off; measured on the MI355X: the kernel
            # lands closer to the float64 value than the fp32 reference does)
            ref64 = L16["%s_%s_sample64" % (tag, nm)].astype(np.float64)
            atol = abs_s * max(np.abs(ref_s).max(), 1e-30)
            excess = np.abs(s - ref64) - rel_s * np.abs(ref64) - 2.0 * np.abs(ref_s - ref64)
            i = np.unravel_index(np.argmax(excess), excess.shape)
            assert excess[i] <= atol, (tag, nm, "worst element", i, float(s[i]), float(ref_s[i]), float(ref64[i]), "excess / atol",
                                       float(excess[i] / atol), "elements over", int((excess > atol).sum()))


def test_loss_oracle_at_batch16_vs_reference_fixture():
    from oracle import loss_oracle
    # (abs_s: the oracle is PyTorch CPU fp32, whose vectorised kernels differ between hosts in their last bits - on the GPU box's
    #  host one near-cancelling element of d pred lands 1.4e-4 of the largest sample off, in the build container below 1e-4; the HIP kernel,
    #  which accumulates those terms in a fixed order, is held to 1e-4 on every host)
    _check_loss16(loss_oracle, False, 2e-6, 1e-5, 1e-4, abs_s=3e-4)


# ------------------------------------------------------------------------------------------ GPU: the HIP path

@pytest.mark.gpu
@pytest.mark.parametrize("tag,num_mlr", [("single", 0), ("mlr3", 3)])
def test_hip_forward_at_480x720_vs_reference_fixture(tag, num_mlr):
    """BASELINE configs[2] / configs[4] frame size against the reference module's own output."""
    net = _net(num_mlr).cuda().eval()
    x = torch.from_numpy(golden_inputs.full_size_image(tag)).cuda()
    with torch.no_grad():
        y = net(x).cpu()
    assert y.shape == (1, 4, 60, 90)
    err = _check_forward(y, FULL[tag + "_y"])
    print("%s 480x720: max coordinate error vs the reference %.2e m" % (tag, err))
    # the plan under test is the benchmarked one: GEMMs on the split-bf16 pipe (unless the fp32-MFMA switch is set)
    plan = list(net._plans.values())[0]
    if os.environ.get("XL_GEMM_SPLIT_BF16", "il") not in ("", "0"):
        assert any(op.flags & networks.CONV_SPLIT_BF16 for op in plan.ops if op.type == networks.XL_OP_CONV)


TINY = np.load(os.path.join(HERE, "golden", "net_forward_tiny.npz"))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["tiny", "tiny_mlr3", "tiny_full"])
def test_hip_forward_of_the_tiny_variant_vs_reference_fixture(tag):
    """`--tiny` (test_single_task.py:43, :253 '-tiny' in the folder name, :299 -> utils/evaluation.py:106): the 128-channel
    network of networks.py:133-135, 194-198 whose res2 block closes without the skip projection (:245-247).  Single-task and
    3-encoder at 64x96 x 2, single-task at 480x720, against the reference module's fp32 output and - the yardstick - its float64
    one: the HIP forward's distance from float64 within 8x / 12x (median / max) of the reference's own fp32 distance on the
    uncertainty channel, as the 512-channel network is held."""
    num_mlr, add, seed, shape = golden_inputs.TINY_CASES[tag]
    net = networks.TransPoseNet(MEAN, True, False, add, add, 3, 1, 32, num_mlr, 0, False)
    net.load_state_dict(seeded_state_dict(net, seed=seed), strict=True)
    net = net.cuda().eval()
    with torch.no_grad():
        y = net(torch.from_numpy(golden_inputs.tiny_input(tag)).cuda()).cpu()
    assert tuple(y.shape) == tuple(TINY[tag + "_y"].shape)
    err = _check_forward(y, TINY[tag + "_y"])
    y, y64, y32 = y.double().numpy(), TINY[tag + "_y64"], TINY[tag + "_y"].astype(np.float64)
    e, e32 = np.abs(y - y64), np.abs(y32 - y64)
    r, r32 = e[:, 3] / np.abs(y64[:, 3]), e32[:, 3] / np.abs(y64[:, 3])
    print("%s: max coordinate error vs the reference %.2e m; uncertainty channel vs float64: median %.2e (reference fp32 %.2e), "
          "max %.2e (%.2e)" % (tag, err, np.median(r), np.median(r32), r.max(), r32.max()))
    assert e[:, :3].mean() <= 1.25 * e32[:, :3].mean() + 1e-6 and e[:, :3].max() <= 2.0 * e32[:, :3].max() + 1e-5
    assert np.median(r) <= 8.0 * np.median(r32) and r.max() <= 12.0 * r32.max(), (np.median(r), r.max())


@pytest.mark.gpu
@pytest.mark.parametrize("tag,num_mlr", [("single", 0), ("mlr3", 3)])
def test_hip_forward_at_480x720_against_the_float64_reference(tag, num_mlr):
    """The dtype claim at the level of the NETWORK: the reference module evaluated in float64 on the fixture's frame
    (full_size.npz `*_y64`) is the truth, the reference's own fp32 forward (`*_y`, PyTorch CPU) the yardstick.  The coordinate
    channels - a mean of ~500 m added to an O(1) head output - sit within half an fp32 ulp of it either way: the HIP forward's mean
    distance must stay within 1.25x the reference's (measured 1.03x / 1.09x).  The uncertainty channel has no such floor and shows
    the arithmetic: median relative distance 2.1e-6 / 3.8e-6 against the reference's 5.3e-7 / 7.2e-7 (4x / 5x), maxima 2.0e-5 /
    3.0e-5 against 2.9e-6 / 4.4e-6 (7x) - held to 8x and 12x.  That distance is the Winograd F(6x6,3x3) transforms' fp32 arithmetic,
    not the GEMM operands: the same network with XL_NO_WINOGRAD=1 lands at 1.8x (9.5e-7), with F(4x4,3x3) at 2.8x, and of the three
    GEMM forms the fp16 pairs are the CLOSEST (pairs 2.1e-6, six-pass bf16 2.7e-6, fp32 MFMA 3.0e-6; tools/forward_error_vs_f64.py)."""
    net = _net(num_mlr).cuda().eval()
    with torch.no_grad():
        y = net(torch.from_numpy(golden_inputs.full_size_image(tag)).cuda()).cpu().double().numpy()
    y64, y32 = FULL[tag + "_y64"], FULL[tag + "_y"].astype(np.float64)
    e, e32 = np.abs(y - y64), np.abs(y32 - y64)
    # (1.25: the measurement switches' forms - six-pass bf16 1.16x, fp32 MFMA 1.20x on the 3-encoder net - pass the same test)
    assert e[:, :3].mean() <= 1.25 * e32[:, :3].mean() and e[:, :3].max() <= 2.0 * e32[:, :3].max(), (e[:, :3].mean(), e[:, :3].max())
    r, r32 = e[:, 3] / np.abs(y64[:, 3]), e32[:, 3] / np.abs(y64[:, 3])
    print("%s: uncertainty channel, relative distance from float64: median %.2e (reference fp32 %.2e), max %.2e (%.2e)"
          % (tag, np.median(r), np.median(r32), r.max(), r32.max()))
    assert np.median(r) <= 8.0 * np.median(r32) and r.max() <= 12.0 * r32.max(), (np.median(r), r.max())


@pytest.mark.gpu
def test_hip_training_step_gradients_vs_reference_fixture():
    """a14: forward + MLE coordinate loss + backward through crossloc_amd (HIP kernels) against the gradients autograd
    produced on the reference module (train_single_task.py:262-298)."""
    from crossloc_amd import loss as xl_loss
    x, poses, _ = golden_inputs.grad_inputs()
    H, W = golden_inputs.GRAD_H, golden_inputs.GRAD_W
    net = _net().cuda().train()
    pred = net(torch.from_numpy(x).cuda())
    _check_forward(pred.detach().cpu(), GRADS["y"], tol_range=1e-3)
    pred.retain_grad()
    sc, unc = torch.split(pred, [3, 1], dim=1)
    loss, rate = xl_loss.scene_coords_regression_loss(0.1, 100.0, 1000.0, 50.0, "MLE", xl_loss.get_pixel_grid(8), -1,
                                                      xl_loss.get_cam_mat(W, H, golden_inputs.GRAD_FOCAL), sc, unc,
                                                      torch.from_numpy(poses).cuda(), torch.from_numpy(GRADS["gt"]).cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert loss.item() == pytest.approx(float(GRADS["loss_f64"]), rel=2e-4)
    assert loss.item() == pytest.approx(float(GRADS["loss_f32"]), rel=2e-4)
    assert float(rate) == pytest.approx(float(GRADS["rate_f64"]), abs=1e-6)
    dref = GRADS["dpred"].astype(np.float64)
    dgot = pred.grad.double().cpu().numpy()
    assert np.linalg.norm(dgot - dref) <= 2e-3 * np.linalg.norm(dref)
    _check_param_grads(net, GRADS, "64x96 x 2")


def _check_param_grads(net, G, tag, tensor_abs=1e-2, tensor_rel=5.0, max_over_1e3=None):
    """All 114 parameter gradients of `net` against the fixture `G` - the reference module's autograd gradients with the network
    in float64.  Per tensor: relative L2 error of a strided 256-element sample and relative error of the L2 norm.  The yardstick
    is the reference's OWN fp32 run of the same step, stored beside the float64 one (round 5): fp32 arithmetic through 29
    GroupNorm layers sits 1e-3 (64x96: 96-pixel maps) / 3e-5 (480x720) from the float64 gradients in the median, and the HIP
    path adds the rounding of its F(6x6,3x3) transforms (~1e-5 per layer where a direct fp32 convolution has 2e-7).  Asserted:
      median over the tensors  <= max(1e-3, 5 x the reference's fp32 median);
      every tensor             <= max(tensor_abs, tensor_rel x that tensor's own fp32 error) on the sample - and never above 5e-2,
                                  whatever the reference's fp32 run does -, <= 2e-2 on the norm;
      the tail (round 6)       at most `max_over_1e3` tensors above 1e-3 (480x720: the typical tensor is at 1e-5 - a systematic
                                  0.9 % error in one tensor must not pass; there tensor_abs = 3e-3, tensor_rel = 3)."""
    names = [n.split(":")[0] for n in G["param_names"]]
    params = dict(net.named_parameters())
    assert names == list(params.keys()) and len(names) == 114
    gmax = float(np.abs(G["param_grad_sample"]).max())
    table, ref32 = [], []
    for i, name in enumerate(names):
        g = params[name].grad
        assert g is not None, name
        g = g.double().cpu().numpy()
        s = golden_inputs.strided(g)
        ref = G["param_grad_sample"][i, :s.size].astype(np.float64)
        if name == "encoder.conv1.bias":
            # GroupNorm(32, 32) is an instance norm: d/d(conv1.bias) is exactly 0; the reference holds rounding noise
            assert np.abs(g).max() <= 1e-4 * gmax and float(G["param_grad_l2"][i]) <= 1e-4 * gmax
            continue
        floor = 1e-4 * gmax * np.sqrt(s.size)
        den = max(np.linalg.norm(ref), floor)
        e2 = np.linalg.norm(s - ref) / den
        r2 = np.linalg.norm(G["param_grad_sample_f32"][i, :s.size].astype(np.float64) - ref) / den
        l2 = np.sqrt((g * g).sum())
        en = abs(l2 - float(G["param_grad_l2"][i])) / max(float(G["param_grad_l2"][i]), 1e-4 * gmax * np.sqrt(g.size))
        table.append((e2, en, r2, name))
        ref32.append(r2)
    table.sort(reverse=True)
    median, med32 = table[len(table) // 2][0], sorted(ref32)[len(ref32) // 2]
    print("%s: parameter gradients vs the reference module in float64: median %.2e (the reference's own fp32 run: %.2e), worst "
          "sample %.2e / norm %.2e, %d tensors above 1e-3" % (tag, median, med32, table[0][0], max(t[1] for t in table),
                                                              sum(1 for t in table if t[0] > 1e-3)))
    for t in table[:6]:
        print("   %-40s sample %.2e (reference fp32 %.2e) norm %.2e" % (t[3], t[0], t[2], t[1]))
    assert median <= max(1e-3, 5.0 * med32), (median, med32)
    for e2, en, r2, name in table:
        bound = min(5e-2, max(tensor_abs, tensor_rel * r2))
        assert e2 <= bound and en <= 2e-2, "%s: sample error %.2e (bound %.2e = %.1fx the measured value; reference fp32 %.2e), norm %.2e" % (
            name, e2, bound, bound / max(e2, 1e-30), r2, en)
    if max_over_1e3 is not None:
        over = [(name, e2) for e2, en, r2, name in table if e2 > 1e-3]
        assert len(over) <= max_over_1e3, "tensors above 1e-3: %d (limit %d): %s" % (len(over), max_over_1e3, over)


@pytest.mark.gpu
def test_hip_training_step_gradients_at_480x720_vs_reference_fixture():
    """Round 5: the same step at the BASELINE configs[1] frame size - 480x720, batch 1 - against gradients autograd produced on
    the reference module in float64 (tests/golden/make_golden.py gradsfull; train_single_task.py:262-300)."""
    from crossloc_amd import loss as xl_loss
    GF = np.load(os.path.join(os.path.dirname(__file__), "golden", "grads_full.npz"))
    x, poses, _ = golden_inputs.grad_full_inputs()
    assert golden_inputs.checksum(x) == pytest.approx(float(GF["x_checksum"]), rel=1e-12)
    net = _net().cuda().train()
    pred = net(torch.from_numpy(x).cuda())
    _check_forward(pred.detach().cpu(), GF["y"], tol_range=1e-3)
    pred.retain_grad()
    sc, unc = torch.split(pred, [3, 1], dim=1)
    loss, rate = xl_loss.scene_coords_regression_loss(0.1, 100.0, 1000.0, 50.0, "MLE", xl_loss.get_pixel_grid(8), -1,
                                                      xl_loss.get_cam_mat(golden_inputs.FULL_W, golden_inputs.FULL_H, 480.0), sc, unc,
                                                      torch.from_numpy(poses).cuda(), torch.from_numpy(GF["gt"]).cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert loss.item() == pytest.approx(float(GF["loss_f64"]), rel=2e-4)
    assert float(rate) == pytest.approx(float(GF["rate_f64"]), abs=1e-6)
    dref = GF["dpred"].astype(np.float64)
    dgot = pred.grad.double().cpu().numpy()
    assert np.linalg.norm(dgot - dref) <= 2e-3 * np.linalg.norm(dref)
    _check_param_grads(net, GF, "480x720 x 1", tensor_abs=3e-3, tensor_rel=3.0, max_over_1e3=30)


@pytest.mark.gpu
def test_hip_losses_at_batch16_vs_reference_fixture():
    from crossloc_amd import loss as xl_loss
    _check_loss16(xl_loss, True, 1e-4, 1e-3, 2e-3)
