"""GPU parity of the training backward kernels against torch-CPU float64 autograd of the
same expressions (oracle/torch_grad.py; in the reference TensorFlow's autodiff supplies these gradients,
mvsnet/train.py:428-429).  All device calls go through the C ABI.  Tolerances: fp32 kernels against a
float64 checker, relative L1 <= 1e-4 unless stated.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import mvsnet_oracle as O
from oracle import torch_grad as TG
from mvsnet_amd import synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _lib(lib_built):
    from mvsnet_amd import _lib as L
    L.load()
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    yield


def t(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(DEV)


def n(x):
    torch.cuda.synchronize()
    return x.detach().cpu().numpy().astype(np.float64)


def rel_l1(a, b):
    return float(np.abs(a - b).sum() / max(np.abs(b).sum(), 1e-30))


def d64(a, grad=False):
    x = torch.tensor(np.asarray(a, np.float64))
    return x.requires_grad_(grad)


def test_softargmin_backward_matches_autograd():
    from mvsnet_amd import backward as B
    rs = np.random.RandomState(0)
    D, H, W = 24, 9, 21
    reg = rs.randn(D, H, W).astype(np.float32)
    g = rs.randn(H, W).astype(np.float32)
    x = d64(reg, True)
    (TG.soft_argmin(x, 425.0, 2.5) * d64(g)).sum().backward()
    got = n(B.softargmin_bwd(t(reg), t(g), 425.0, 2.5))
    assert rel_l1(got, x.grad.numpy()) < 1e-5
    # with a gradient for the 4-bucket probability map as well (bucket indices are constants, as in TensorFlow)
    gp = rs.randn(H, W).astype(np.float32)
    x2 = d64(reg, True)
    P = torch.softmax(-x2, dim=0)
    depth = TG.soft_argmin(x2, 425.0, 2.5)
    idx = ((depth - 425.0) / 2.5).detach()
    l0 = idx.floor().long().clamp(0, D - 1); r0 = idx.ceil().long().clamp(0, D - 1)
    l1 = (l0 - 1).clamp(0, D - 1); r1 = (r0 + 1).clamp(0, D - 1)
    pick = lambda i: torch.gather(P, 0, i[None])[0]
    prob = pick(l0) + pick(r0) + pick(l1) + pick(r1)
    ((depth * d64(g)).sum() + (prob * d64(gp)).sum()).backward()
    got2 = n(B.softargmin_bwd(t(reg), t(g), 425.0, 2.5, g_prob=t(gp)))
    assert rel_l1(got2, x2.grad.numpy()) < 1e-4


@pytest.mark.parametrize("C,two", [(8, True), (16, False), (64, True)])
def test_bn_relu_and_its_backward_match_autograd(C, two):
    from mvsnet_amd import backward as B
    from mvsnet_amd import model as M
    rs = np.random.RandomState(C)
    shape = (6, 10, 12, C)
    y = rs.randn(*shape).astype(np.float32) * 1.5 + 0.3
    gamma = (1.0 + 0.3 * rs.randn(C)).astype(np.float32)
    beta = (0.2 * rs.randn(C)).astype(np.float32)
    g1 = rs.randn(*shape).astype(np.float32)
    g2 = rs.randn(*shape).astype(np.float32) if two else None
    # checker
    yy, gg, bb = d64(y, True), d64(gamma, True), d64(beta, True)
    a = F.relu(F.batch_norm(yy.permute(3, 0, 1, 2)[None], None, None, gg, bb, training=True, eps=1e-5))[0].permute(1, 2, 3, 0)
    gsum = d64(g1) + (d64(g2) if two else 0.0)
    (a * gsum).sum().backward()
    # device: forward statistics exactly as the conv kernels emit them
    yt = t(y)
    stats = torch.stack([yt.double().sum((0, 1, 2)), (yt.double() ** 2).sum((0, 1, 2))]).contiguous()
    aff = M.bn_finalize(stats, yt.numel() // C, t(gamma), t(beta))
    assert rel_l1(n(B.bn_relu(yt, aff)), a.detach().numpy()) < 1e-6
    g_y, g_gamma, g_beta = B.bn_relu_bwd(yt, stats, aff, t(gamma), t(g1), t(g2) if two else None)
    assert rel_l1(n(g_y), yy.grad.numpy()) < 1e-5
    assert rel_l1(n(g_gamma), gg.grad.numpy()) < 1e-5
    assert rel_l1(n(g_beta), bb.grad.numpy()) < 1e-5


# (Cbig, Csmall, stride, kind): every layer shape of RegNetUS0 'normal'
WGRAD_CASES = [(32, 8, 1, "conv"), (8, 32, 1, "conv"), (16, 16, 1, "conv"), (32, 32, 1, "conv"), (64, 64, 1, "conv"), (8, 1, 1, "conv"),
               (32, 16, 2, "conv"), (16, 32, 2, "conv"), (32, 64, 2, "conv"),
               (32, 64, 2, "deconv"), (16, 32, 2, "deconv"), (8, 16, 2, "deconv")]


@pytest.mark.parametrize("cb,cs,stride,kind", WGRAD_CASES)
@pytest.mark.parametrize("dims", [(8, 8, 16), (6, 12, 40)])
def test_conv3d_weight_gradient_matches_autograd(cb, cs, stride, kind, dims):
    from mvsnet_amd import backward as B
    rs = np.random.RandomState(cb * 100 + cs + stride)
    D, H, W = dims
    small_dims = tuple(v // stride for v in dims)
    big = rs.randn(D, H, W, cb).astype(np.float32)
    small = rs.randn(*small_dims, cs).astype(np.float32)
    if kind == "conv":            # y = conv(x=big, w (3,3,3,cb,cs)); small = g_y
        w = d64(np.zeros((3, 3, 3, cb, cs)), True)
        y = TG._conv(d64(big).permute(3, 0, 1, 2)[None], w, stride)[0].permute(1, 2, 3, 0)
        (y * d64(small)).sum().backward()
    else:                         # y = deconv(x=small, w (3,3,3,Cout=cb,Cin=cs)); big = g_y
        w = d64(np.zeros((3, 3, 3, cb, cs)), True)
        y = TG._deconv(d64(small).permute(3, 0, 1, 2)[None], w)[0].permute(1, 2, 3, 0)
        (y * d64(big)).sum().backward()
    got = n(B.conv3d_wgrad(t(big), t(small), stride))
    assert got.shape == (3, 3, 3, cb, cs)
    assert rel_l1(got, w.grad.numpy()) < 2e-5
    # deterministic: a second launch gives the same bits
    assert np.array_equal(got, n(B.conv3d_wgrad(t(big), t(small), stride)))


@pytest.mark.parametrize("cin,cout,kind", [(32, 8, "s1"), (16, 16, "s1"), (8, 1, "s1"), (64, 64, "s1"),
                                           (32, 16, "s2"), (16, 32, "s2"), (32, 64, "s2"),
                                           (64, 32, "deconv"), (32, 16, "deconv"), (16, 8, "deconv")])
@pytest.mark.parametrize("dims", [(8, 8, 16), (6, 12, 40)])
def test_conv3d_input_gradients_match_autograd(cin, cout, kind, dims):
    from mvsnet_amd import backward as B
    rs = np.random.RandomState(cin + 3 * cout)
    D, H, W = dims
    x = d64(rs.randn(D, H, W, cin), True)
    if kind == "deconv":
        w = rs.randn(3, 3, 3, cout, cin).astype(np.float32) * 0.1
        y = TG._deconv(x.permute(3, 0, 1, 2)[None], d64(w))[0].permute(1, 2, 3, 0)
    else:
        w = rs.randn(3, 3, 3, cin, cout).astype(np.float32) * 0.1
        y = TG._conv(x.permute(3, 0, 1, 2)[None], d64(w), 1 if kind == "s1" else 2)[0].permute(1, 2, 3, 0)
    g = rs.randn(*y.shape).astype(np.float32)
    (y * d64(g)).sum().backward()
    fn = {"s1": B.conv_s1_input_grad, "s2": B.conv_s2_input_grad, "deconv": B.deconv_input_grad}[kind]
    got = n(fn(t(g), t(w)))
    assert got.shape == tuple(x.shape)
    assert rel_l1(got, x.grad.numpy()) < 2e-5


def _toy_problem(N=3, D=16, H=16, W=32, C=32):
    cams = S.make_cams(N, H, W, D)
    feats = S.make_features(N, H, W, C, seed=5)
    start, interval = float(cams[0, 1, 3, 0]), float(cams[0, 1, 3, 1])
    Hs = np.stack([O.get_homographies(cams[0], cams[v], D, start, interval, np.float32) for v in range(1, N)])
    t8 = O.homography_to_transform8(Hs, np.float32)
    return feats, t8, start, interval


@pytest.mark.parametrize("method", ["gather", "scatter"])
@pytest.mark.parametrize("shape", [(3, 16, 16, 32, 32), (5, 24, 40, 72, 32), (2, 8, 24, 24, 16), (3, 60, 24, 40, 32)])
def test_cost_volume_backward_matches_autograd(shape, method):
    """2 / 4 / 1 source views, ragged sizes, 32 and 16 channels, several plane chunks; both the atomic-free
    gather kernels and the float-atomic scatter kernel."""
    from mvsnet_amd import backward as B
    feats, t8, _, _ = _toy_problem(*shape)
    rs = np.random.RandomState(3)
    g1 = rs.randn(t8.shape[1], *feats.shape[1:]).astype(np.float32)
    g2 = rs.randn(*g1.shape).astype(np.float32)
    f = d64(feats, True)
    cost = TG.cost_volume(f, d64(t8)).permute(1, 2, 3, 0)               # (D,H,W,C)
    (cost * (d64(g1) + d64(g2))).sum().backward()
    ft = t(feats)
    g_ref, g_src = B.cost_volume_bwd(ft[0], ft[1:], t(t8), t(g1), t(g2), method=method)
    got = np.concatenate([n(g_ref)[None], n(g_src)], 0)
    if method == "gather":                       # bit-reproducible
        r2, s2 = B.cost_volume_bwd(ft[0], ft[1:], t(t8), t(g1), t(g2), method=method)
        assert torch.equal(r2, g_ref) and torch.equal(s2, g_src)
    # the warp is piecewise bilinear: fp32 sample coordinates that land within rounding of a pixel boundary
    # may pick the neighbouring cell; the contribution is continuous, so the L1 error stays at rounding level
    assert rel_l1(got, f.grad.numpy()) < 1e-4


def test_plane_sweep_depth_gradients_match_autograd():
    """features -> depth -> scalar: every RegNetUS0 parameter gradient and the feature gradient."""
    from mvsnet_amd import backward as B
    feats, t8, start, interval = _toy_problem()
    params = S.make_regnet_params("normal", seed=1, random_affine=True)
    rs = np.random.RandomState(11)
    g = rs.randn(feats.shape[1], feats.shape[2]).astype(np.float32)
    # checker
    f64 = d64(feats, True)
    p64 = {k: {kk: d64(vv, True) for kk, vv in v.items()} for k, v in params.items()}
    depth64 = TG.depth_from_features(f64, d64(t8), start, interval, p64)
    (depth64 * d64(g)).sum().backward()
    # device
    ft = t(feats).requires_grad_(True)
    pt = {k: {kk: t(vv).requires_grad_(True) for kk, vv in v.items()} for k, v in params.items()}
    depth, prob = B.plane_sweep_depth(ft, t(t8), start, interval, pt)
    assert rel_l1(n(depth), depth64.detach().numpy()) < 1e-5
    (depth * t(g)).sum().backward()
    assert rel_l1(n(ft.grad), f64.grad.numpy()) < 2e-3
    for name in p64:
        for key in p64[name]:
            ref = p64[name][key].grad.numpy()
            got = n(pt[name][key].grad)
            assert rel_l1(got, ref) < 2e-3, (name, key, rel_l1(got, ref))


def test_rmsprop_step_matches_tensorflow_formula():
    from mvsnet_amd import _lib as L
    lib = L.load()
    rs = np.random.RandomState(2)
    nel = 10007
    w, g = rs.randn(nel).astype(np.float32), rs.randn(nel).astype(np.float32)
    ms, mom = np.ones(nel, np.float32), np.zeros(nel, np.float32)          # TF: rms slot starts at one
    wt, gt, mst, momt = t(w), t(g), t(ms), t(mom)
    lr, decay, momentum, eps = 1e-3, 0.9, 0.0, 1e-10
    for _ in range(3):
        L.check(lib.mvs_rmsprop_step_f32(L.ptr(wt), L.ptr(gt), L.ptr(mst), L.ptr(momt), nel, lr, decay, momentum, eps,
                                         0.5, L.stream_ptr()))
        gs = g.astype(np.float64) * 0.5
        ms = ms + (gs * gs - ms) * (1 - decay)
        mom = momentum * mom + lr * gs / np.sqrt(ms + eps)
        w = w - mom
    assert np.allclose(n(wt), w, rtol=1e-5, atol=1e-7)
    assert np.allclose(n(mst), ms, rtol=1e-5)


def _train_batch(N=3, H=64, W=96, D=16):
    images = S.make_images(N, H, W, seed=0)
    cams = S.make_cams(N, H // 4, W // 4, D)
    start, interval = float(cams[0, 1, 3, 0]), float(cams[0, 1, 3, 1])
    xs = np.linspace(0.3, 0.7, W // 4, dtype=np.float32)[None, :].repeat(H // 4, 0)
    gt = (start + interval * (D - 1) * xs)[:, :, None].astype(np.float32)
    gt[0, :3, 0] = 0.0                                                      # a few invalid pixels
    return images, cams, gt, D


@pytest.mark.parametrize("optimizer", ["rmsprop", "momentum", "adam"])
def test_trainer_reduces_the_loss_on_one_batch(optimizer):
    """images -> towers (torch autograd) -> HIP hot path forward/backward -> flat-buffer optimiser step."""
    from mvsnet_amd import train as T
    images, cams, gt, D = _train_batch()
    lr = {"rmsprop": 1e-3, "momentum": 1e-3, "adam": 1e-3}[optimizer]
    tr = T.Trainer("normal", DEV, optimizer=optimizer, base_lr=lr, seed=0)
    before = tr.params.data.clone()
    losses = [float(tr.train_step(images, cams, gt, D)[0]) for _ in range(10)]
    assert all(np.isfinite(losses)), losses
    assert min(losses[-3:]) < losses[0], losses
    assert tr.global_step == 10
    assert float((tr.params.data - before).abs().max()) > 0
    assert float(tr.params.grad.abs().max()) == 0.0                        # zeroed after the update


@pytest.mark.parametrize("mode,refinement", [("normal", False), ("normal", True), ("lite", False)])
def test_trainer_takes_decoded_uint8_images(mode, refinement):
    """The training loop's input pipeline hands the cropped images over as uint8 (TrainingPrefetcher(center=False)):
    loss and gradients are those of the same images standardised beforehand -- by the HIP towers' own launch pair
    (mvs_center_images_u8_f32, "normal") or by its PyTorch restatement in front of the ATen towers ("lite") and of the
    refinement's guide image."""
    from mvsnet_amd import train as T
    from mvsnet_amd.inference import center_images_device
    _images, cams, gt, D = _train_batch()
    u8 = torch.randint(0, 256, _images.shape, dtype=torch.uint8, generator=torch.Generator().manual_seed(1))
    full = np.repeat(np.repeat(gt, 4, axis=0), 4, axis=1) if refinement else None
    out = []
    for images in (center_images_device(u8.to(DEV)), u8.numpy()):
        tr = T.Trainer(mode, DEV, seed=0, refinement=refinement)
        loss, l1, l3, _d = tr.loss(images, cams, gt, D, full)
        loss.backward()
        out.append((float(loss.detach()), tr.params.grad.clone()))
    assert abs(out[0][0] - out[1][0]) <= 1e-5 * abs(out[0][0])
    assert float((out[0][1] - out[1][1]).abs().max()) <= 1e-4 * float(out[0][1].abs().max())


def test_trainer_checkpoint_round_trip(tmp_path):
    from mvsnet_amd import train as T
    from mvsnet_amd import tf_checkpoint
    images, cams, gt, D = _train_batch()
    tr = T.Trainer("normal", DEV, seed=0)
    for _ in range(2):
        tr.train_step(images, cams, gt, D)
    prefix = tr.save(str(tmp_path))
    saved_w = n(tr.params.group("regnet")["3dconv6_2"]["w"]).astype(np.float32)
    assert prefix.endswith(os.path.join("3DCNN", "normal", "model.ckpt-2"))
    names = {nm for nm, _s, _d in tf_checkpoint.list_variables(prefix)}
    assert {"3dconv0_1/kernel", "3dconv0_1/kernel/RMSProp", "3dconv0_1/bn/gamma/RMSProp_1", "global_step"} <= names
    tr2 = T.Trainer("normal", DEV, seed=123)                               # different initialisation
    tr2.restore(prefix)
    assert tr2.global_step == 2
    assert torch.equal(tr2.params.data, tr.params.data)
    assert all(torch.equal(a, b) for a, b in zip(tr2.slots, tr.slots))
    # the restored replica continues like the original (atomics in the reductions: not bit-identical)
    l1 = float(tr.train_step(images, cams, gt, D)[0]); l2 = float(tr2.train_step(images, cams, gt, D)[0])
    assert abs(l1 - l2) <= 1e-4 * abs(l1)
    # and the inference path loads what training wrote
    loaded = tf_checkpoint.load_mvsnet_params(prefix, "normal", "3DCNN")
    assert np.array_equal(loaded["regnet"]["3dconv6_2"]["w"], saved_w)


def test_trainer_lite_mode_trains_padded_and_saves_native_shapes(tmp_path):
    """'lite' (the reference's training default) runs zero-padded on the 'normal' kernel shapes: the padded
    entries must stay exactly zero and checkpoints must hold the native shapes."""
    from mvsnet_amd import train as T
    from mvsnet_amd import tf_checkpoint
    images, cams, gt, D = _train_batch()
    tr = T.Trainer("lite", DEV, seed=0)
    losses = [float(tr.train_step(images, cams, gt, D)[0]) for _ in range(6)]
    assert all(np.isfinite(losses)) and min(losses[-2:]) < losses[0], losses
    w = tr.params.group("regnet")["3dconv0_1"]["w"]
    assert tuple(w.shape) == (3, 3, 3, 32, 8)
    w = w.detach()
    assert float(w[:, :, :, 16:, :].abs().max()) == 0.0 and float(w[:, :, :, :, 4:].abs().max()) == 0.0
    assert float(w[:, :, :, :16, :4].abs().max()) > 0.0
    g = tr.params.group("regnet")["3dconv0_1"]["gamma"].detach()
    assert float(g[4:].abs().max()) == 0.0 and float(g[:4].min()) > 0.5
    prefix = tr.save(str(tmp_path))
    shapes = {nm: shp for nm, shp, _d in tf_checkpoint.list_variables(prefix)}
    assert shapes["3dconv0_1/kernel"] == (3, 3, 3, 16, 4) and shapes["3dconv0_1/kernel/RMSProp"] == (3, 3, 3, 16, 4)
    assert shapes["2dconv1_0/kernel"][-1] == 8                              # lite towers: base_filter 4
    tr2 = T.Trainer("lite", DEV, seed=9)
    tr2.restore(prefix)
    assert torch.equal(tr2.params.data, tr.params.data)
    for a, b in zip(tr2.slots, tr.slots):            # slots agree on the native entries (padded ones never matter)
        na, nb = tr.params.named_arrays(a), tr.params.named_arrays(b)
        for var, shp in tr.native_shapes.items():
            sl = tuple(slice(0, k) for k in shp)
            assert np.array_equal(na[var][sl], nb[var][sl]), var
    loaded = tf_checkpoint.load_mvsnet_params(prefix, "lite", "3DCNN")       # the inference path reads it back
    assert loaded["regnet"]["3dconv0_1"]["w"].shape == (3, 3, 3, 16, 4)


@pytest.mark.parametrize("C,H,W,relu", [(8, 24, 40, True), (16, 12, 20, False), (64, 6, 10, True), (128, 4, 6, True)])
def test_hip_group_norm_matches_torch_autograd(C, H, W, relu):
    """GroupNorm(+ReLU) of the training towers (network.py:217-276) against torch's float64 group_norm."""
    from mvsnet_amd.feature_net import HipGroupNorm
    rs = np.random.RandomState(C)
    V = 3
    x = (rs.randn(V, C, H, W) * 1.3 + 0.2).astype(np.float32)
    gamma = (1.0 + 0.3 * rs.randn(C)).astype(np.float32); beta = (0.2 * rs.randn(C)).astype(np.float32)
    g = rs.randn(V, C, H, W).astype(np.float32)
    xx, gg, bb = d64(x, True), d64(gamma, True), d64(beta, True)
    y64 = F.group_norm(xx, C // 8, gg, bb, eps=1e-5)
    if relu:
        y64 = F.relu(y64)
    (y64 * d64(g)).sum().backward()
    xt = t(x).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gt_, bt_ = t(gamma).requires_grad_(True), t(beta).requires_grad_(True)
    y = HipGroupNorm.apply(xt, gt_, bt_, 1e-5, relu)
    assert y.is_contiguous(memory_format=torch.channels_last)
    assert rel_l1(n(y), y64.detach().numpy()) < 1e-6
    (y * t(g)).sum().backward()
    assert rel_l1(n(xt.grad), xx.grad.numpy()) < 1e-5
    assert rel_l1(n(gt_.grad), gg.grad.numpy()) < 1e-5
    assert rel_l1(n(bt_.grad), bb.grad.numpy()) < 1e-5


# ---- full-size checks through size-independent properties (train.py's default volume: D=192, 120 x 160) ----------
FULL = (192, 120, 160)


def _dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize("cin,cout,stride", [(32, 8, 1), (32, 16, 2)])
def test_full_size_weight_and_input_gradients_are_adjoints_of_the_forward_convolution(cin, cout, stride):
    """<conv(x, W), g> = <W, wgrad(x, g)> = <x, input_grad(g, W)> with the forward MFMA kernel as the
    left-hand side: ties both backward routes to the (oracle-checked) forward at the full volume size."""
    from mvsnet_amd import backward as B
    from mvsnet_amd.model import conv3d
    torch.manual_seed(cin + cout)
    D, H, W = FULL
    x = torch.randn(D, H, W, cin, device=DEV)
    w = torch.randn(3, 3, 3, cin, cout, device=DEV) * 0.05
    y = conv3d(x, w, stride)
    g = torch.randn_like(y)
    lhs = _dot(y, g)
    dw = B.conv3d_wgrad(x, g, stride)
    assert abs(_dot(dw, w) - lhs) <= 2e-5 * abs(lhs) + 1e-3 * float(dw.abs().double().sum() * 1e-6)
    gx = (B.conv_s1_input_grad if stride == 1 else B.conv_s2_input_grad)(g, w)
    assert gx.shape == x.shape
    assert abs(_dot(gx, x) - lhs) <= 2e-5 * abs(lhs) + 1.0


def test_full_size_cost_volume_backward_is_the_directional_derivative():
    """<grad_f, delta> against a central difference of <cost(f), g> along delta at the full training size.
    (The cost volume is quadratic in the features where the tap pattern is fixed, so the central difference
    is exact up to rounding.)"""
    from mvsnet_amd import backward as B
    from mvsnet_amd.model import cost_volume
    from mvsnet_amd.homography_warping import homography_transforms
    D, H, W = FULL
    N = 3
    cams = S.make_cams(N, H, W, D)
    start, interval = float(cams[0, 1, 3, 0]), float(cams[0, 1, 3, 1])
    t8 = homography_transforms(t(cams), D, start, interval)
    f = t(S.make_features(N, H, W, 32, seed=2))
    torch.manual_seed(0)
    g = torch.randn(D, H, W, 32, device=DEV) * 0.1
    delta = torch.randn_like(f)
    g_ref, g_src = B.cost_volume_bwd(f[0], f[1:], t8, g)
    lhs = _dot(torch.cat([g_ref[None], g_src], 0), delta)
    eps = 0.5
    cp = cost_volume(f[0] + eps * delta[0], f[1:] + eps * delta[1:], t8, variant="eager")
    up = _dot(cp, g); del cp
    cm = cost_volume(f[0] - eps * delta[0], f[1:] - eps * delta[1:], t8, variant="eager")
    rhs = (up - _dot(cm, g)) / (2 * eps)
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), abs(rhs)) + 1e-2, (lhs, rhs)


def test_hip_towers_match_the_torch_towers_forward_and_backward():
    """UNetDS2GN for training as one autograd node (HIP forward with folded GroupNorm, HIP GroupNorm backward,
    ATen convolution backward) against the plain PyTorch towers under autograd."""
    from mvsnet_amd.feature_net import trainable_layers, unet_forward
    from mvsnet_amd.feature_net_train import hip_towers
    params = S.make_unet_params("normal", seed=3)
    rs = np.random.RandomState(1)
    for name in params:                                   # non-trivial GroupNorm affines
        if "gamma" in params[name]:
            params[name]["gamma"] = (1.0 + 0.2 * rs.randn(*params[name]["gamma"].shape)).astype(np.float32)
            params[name]["beta"] = (0.1 * rs.randn(*params[name]["beta"].shape)).astype(np.float32)
    images = t(S.make_images(2, 32, 48, seed=4))
    g = t(rs.randn(2, 8, 12, 32).astype(np.float32))
    mk = lambda: {k: {kk: t(vv).requires_grad_(True) for kk, vv in v.items()} for k, v in params.items()}
    pa, pb = mk(), mk()
    fa = unet_forward(trainable_layers(pa), images)
    (fa * g).sum().backward()
    fb = hip_towers(images, pb)
    assert rel_l1(n(fb), n(fa)) < 1e-5
    (fb * g).sum().backward()
    worst = 0.0
    for name in pa:
        for key in pa[name]:
            e = rel_l1(n(pb[name][key].grad), n(pa[name][key].grad))
            worst = max(worst, e)
            assert e < 2e-3, (name, key, e)
    print("worst tower gradient rel-L1:", worst)


def test_hip_towers_accumulate_their_parameter_gradients_into_the_flat_buffer():
    """`accumulate_into_grads` (the trainer's mode): the backward adds all 94 parameter gradients into the leaves' pre-allocated
    `.grad` slices itself (mvs_transpose_add_many_f32 for ATen's (Cout,Cin,k,k) kernels, mvs_add_f64_many_f32 for the float64
    gamma / beta sums) -- the same values autograd accumulates from the returned tensors, added to what the buffer held."""
    from mvsnet_amd import train as T
    from mvsnet_amd.feature_net_train import hip_towers
    images = t(S.make_images(2, 32, 48, seed=4))
    g = t(np.random.RandomState(1).randn(2, 8, 12, 32).astype(np.float32))
    out = []
    for into in (False, True):
        tr = T.Trainer("normal", DEV, seed=0)
        tr.params.grad.fill_(0.25)                                          # ADDED to, not overwritten
        f = hip_towers(images, tr.params.group("unet"), accumulate_into_grads=into)
        (f * g).sum().backward()
        out.append(tr.params.grad.clone())
    assert float((out[0] - 0.25).abs().max()) > 0
    # (two backward passes differ in the last bits by themselves: float64 atomics of the GroupNorm sums, MIOpen's weight gradients)
    assert float((out[0] - out[1]).abs().max()) <= 1e-5 * float(out[0].abs().max())
    grp = T.Trainer("normal", DEV, seed=0).params.group("unet")
    from mvsnet_amd.feature_net import UNET_LAYERS
    first = UNET_LAYERS[0][0]                                               # the layer that reads the image: (3, 3, 3, 8)
    leaf = torch.zeros(3, 3, 3, 8, device=DEV, requires_grad=True)          # a leaf without .grad
    with pytest.raises(ValueError):
        hip_towers(images, {**grp, first: {**grp[first], "w": leaf}}, accumulate_into_grads=True)


def test_plane_sweep_depth_accumulates_its_parameter_gradients_into_the_flat_buffer():
    """`accumulate_into_grads` of the hot path's autograd node (the trainer's mode): one mvs_add_many_f32 launch adds every
    RegNetUS0 gradient into the leaves' `.grad` slices -- the values autograd accumulates from the returned tensors."""
    from mvsnet_amd import backward as B, train as T
    from mvsnet_amd.homography_warping import homography_transforms
    w = S.make_workload("toy")
    feats = t(w.features[:3])
    cams = S.make_cams(3, w.height, w.width, w.depth_num)
    t8 = homography_transforms(t(cams), w.depth_num, w.depth_start, w.depth_interval)
    out = []
    for into in (False, True):
        tr = T.Trainer("normal", DEV, seed=0)
        tr.params.grad.fill_(0.5)
        f = feats.clone().requires_grad_(True)
        depth, _p = B.plane_sweep_depth(f, t8, w.depth_start, w.depth_interval, tr.params.group("regnet"), accumulate_into_grads=into)
        depth.sum().backward()
        out.append((tr.params.grad.clone(), f.grad.clone()))
    assert float((out[0][0] - 0.5).abs().max()) > 0
    assert float((out[0][0] - out[1][0]).abs().max()) <= 1e-5 * float(out[0][0].abs().max())
    assert float((out[0][1] - out[1][1]).abs().max()) <= 1e-5 * float(out[0][1].abs().max())


def test_many_tensor_launches_match_numpy_beyond_one_table():
    """csrc/multi_tensor.hip: the jobs ride in the kernel arguments, 48 / 96 / 96 per launch -- more jobs than one table holds,
    ragged sizes, a dropped padding channel; results ADDED to what the destinations held (integers: exact)."""
    import ctypes as C
    from mvsnet_amd import _lib as L
    lib = L.load()
    rs = np.random.RandomState(7)
    vp = lambda ts: (C.c_void_p * len(ts))(*[x.data_ptr() for x in ts])
    # (B, A, k, k) -> (k, k, keep, B)
    shapes = [(rs.randint(1, 9), rs.randint(1, 7), int(rs.choice([1, 9, 25]))) for _ in range(101)]
    src = [t(rs.randint(-9, 9, (b, a, kk)).astype(np.float32)) for b, a, kk in shapes]
    keep = [max(1, a - (i % 3 == 0)) for i, (_b, a, _kk) in enumerate(shapes)]
    dst0 = [rs.randint(-9, 9, (kk, kp, b)).astype(np.float32) for (b, _a, kk), kp in zip(shapes, keep)]
    dst = [t(d) for d in dst0]
    dims = (C.c_int * (4 * len(src)))(*[v for (b, a, kk), kp in zip(shapes, keep) for v in (b, a, kk, kp)])
    L.check(lib.mvs_transpose_add_many_f32(len(src), vp(src), vp(dst), dims, L.stream_ptr()), "transpose_add_many")
    for s_, d_, d0, kp in zip(src, dst, dst0, keep):
        assert np.array_equal(n(d_), d0 + n(s_).transpose(2, 1, 0)[:, :kp, :])
    # float64 -> float32 rows, float32 -> float32 tensors
    cnt = [int(rs.randint(1, 300)) for _ in range(197)]
    s64 = [torch.as_tensor(rs.randint(-99, 99, c_).astype(np.float64)).to(DEV) for c_ in cnt]
    d0 = [rs.randint(-9, 9, c_).astype(np.float32) for c_ in cnt]
    d = [t(x) for x in d0]
    L.check(lib.mvs_add_f64_many_f32(len(cnt), vp(s64), vp(d), (C.c_int * len(cnt))(*cnt), L.stream_ptr()), "add_f64_many")
    assert all(np.array_equal(n(a_), b_ + n(c_).astype(np.float32)) for a_, b_, c_ in zip(d, d0, s64))
    cnt = [int(rs.randint(1, 5000)) for _ in range(120)]
    s32 = [t(rs.randint(-99, 99, c_).astype(np.float32)) for c_ in cnt]
    d0 = [rs.randint(-9, 9, c_).astype(np.float32) for c_ in cnt]
    d = [t(x) for x in d0]
    L.check(lib.mvs_add_many_f32(len(cnt), vp(s32), vp(d), (C.c_longlong * len(cnt))(*cnt), L.stream_ptr()), "add_many")
    assert all(np.array_equal(n(a_), b_ + n(c_)) for a_, b_, c_ in zip(d, d0, s32))
    assert lib.mvs_add_many_f32(0, vp(s32), vp(d), (C.c_longlong * 1)(1), L.stream_ptr()) == -1       # MVS_E_BADARG


def test_prepare_many_equals_the_single_preparations():
    """mvs_unet_prepare_many_f32 / mvs_gn_slots_to_channel_sums_many_f64 against their one-at-a-time forms, bit for bit: every
    kind (forward conv with one / two sources, the 3-channel image kernels laid out for 4, 5 x 5, input-gradient form, transposed
    conv), more jobs than one table holds."""
    import ctypes as C
    from mvsnet_amd import _lib as L
    lib = L.load()
    rs = np.random.RandomState(3)
    st = L.stream_ptr()
    jobs = []                                                   # (kind, w, ks, c1, c2, cin_src, cout, single-call reference)
    def conv(ks, c1, c2, cout, cin_src=None):
        src_c = cin_src or (c1 + c2)
        w = t(rs.randn(ks, ks, src_c, cout).astype(np.float32))
        ref = torch.full((lib.mvs_conv2d_prepared_floats(ks, c1, c2, cout),), 7.0, device=DEV)
        wp = w if src_c == c1 + c2 else torch.cat([w, torch.zeros(ks, ks, c1 + c2 - src_c, cout, device=DEV)], 2).contiguous()
        L.check(lib.mvs_conv2d_prepare_f32(L.ptr(wp), ks, c1, c2, cout, L.ptr(ref), st), "single")
        jobs.append((0, w, ks, c1, c2, src_c, cout, ref))
    def dgrad(ks, cin_fwd, cout_fwd):
        w = t(rs.randn(ks, ks, cin_fwd, cout_fwd).astype(np.float32))
        ref = torch.full((lib.mvs_conv2d_prepared_floats(ks, cout_fwd, 0, cin_fwd),), 7.0, device=DEV)
        L.check(lib.mvs_conv2d_prepare_dgrad_f32(L.ptr(w), ks, cin_fwd, cout_fwd, L.ptr(ref), st), "single")
        jobs.append((1, w, ks, cin_fwd, 0, cin_fwd, cout_fwd, ref))
    def deconv(cin, cout):
        w = t(rs.randn(3, 3, cout, cin).astype(np.float32))
        ref = torch.full((lib.mvs_deconv2d_prepared_floats(cin, cout),), 7.0, device=DEV)
        L.check(lib.mvs_deconv2d_prepare_f32(L.ptr(w), cin, cout, L.ptr(ref), st), "single")
        jobs.append((2, w, 3, cin, 0, cin, cout, ref))
    for _ in range(9):
        conv(3, 4, 0, 8, 3); conv(3, 4, 0, 16, 3); conv(3, 8, 0, 8); conv(3, 16, 16, 16); conv(5, 8, 0, 16); conv(3, 64, 64, 64); conv(3, 32, 0, 32)
        dgrad(3, 8, 8); dgrad(3, 16, 8); dgrad(3, 32, 16); deconv(16, 8); deconv(128, 64)
    k = len(jobs)
    assert k > 56
    out = [torch.full_like(j[7], -3.0) for j in jobs]
    ints = lambda col: (C.c_int * k)(*[j[col] for j in jobs])
    vp = lambda ts: (C.c_void_p * k)(*[x.data_ptr() for x in ts])
    L.check(lib.mvs_unet_prepare_many_f32(k, ints(0), vp([j[1] for j in jobs]), ints(2), ints(3), ints(4), ints(5), ints(6), vp(out), st),
               "mvs_unet_prepare_many_f32")
    for j, o in zip(jobs, out):
        assert torch.equal(o, j[7]), j[:7:2]
    # GroupNorm statistics of many layers
    V, nslot = 3, lib.mvs_gn_stat_slots()
    Cs = [int(rs.choice([8, 16, 32, 64, 128])) for _ in range(70)]
    slot_off, stat_off, a, b = [], [], 0, 0
    for c_ in Cs:
        slot_off.append(a); stat_off.append(b)
        a += V * (c_ // 8) * nslot * 2; b += V * 2 * c_
    slots = torch.as_tensor(rs.randn(a)).to(DEV)
    got = torch.zeros(b, dtype=torch.float64, device=DEV)
    L.check(lib.mvs_gn_slots_to_channel_sums_many_f64(len(Cs), L.ptr(slots), (C.c_longlong * len(Cs))(*slot_off), (C.c_int * len(Cs))(*Cs),
                                                         V, nslot, L.ptr(got), (C.c_longlong * len(Cs))(*stat_off), st), "many")
    for c_, so, to in zip(Cs, slot_off, stat_off):
        ref = torch.zeros(V * 2 * c_, dtype=torch.float64, device=DEV)
        L.check(lib.mvs_gn_slots_to_channel_sums_f64(L.ptr(slots[so:]), V, c_, nslot, L.ptr(ref), st), "single")
        assert torch.equal(got[to:to + V * 2 * c_], ref)


def test_train_cli_runs_on_a_synthetic_dataset(tmp_path, capsys):
    """python -m mvsnet_amd.train end to end (train.py:412-535): train/ + val/ session folders -> generator ->
    trainer -> TensorFlow-format checkpoint, two steps."""
    from _helpers import add_depths, make_session
    from mvsnet_amd import train as T
    from mvsnet_amd import tf_checkpoint
    for mode in ("train", "val"):
        s = make_session(str(tmp_path / "data" / mode / "s0"), n_images=4, h=64, w=96, seed=3)
        add_depths(s, n_images=4, h=64, w=96, seed=3)
    out = tmp_path / "model"
    T.main(["--train_data_root", str(tmp_path / "data"), "--model_dir", str(out), "--network_mode", "normal",
            "--view_num", "3", "--max_d", "16", "--width", "96", "--height", "64", "--base_image_size", "32",
            "--epoch", "1", "--max_steps_per_epoch", "2", "--snapshot", "1", "--train_steps_per_val", "2",
            "--val_batch_size", "1"])
    text = capsys.readouterr().out
    assert "total_step 2" in text and "Saving model to" in text and "VAL STEP COMPLETED" in text
    prefix = tf_checkpoint.model_path(tf_checkpoint.ckpt_path(str(out), "3DCNN", "normal"), 2)
    names = {nm for nm, _s, _d in tf_checkpoint.list_variables(prefix)}
    assert "3dconv6_2/kernel" in names and "global_step" in names


def test_device_gradients_match_the_committed_golden_fixture():
    """tests/golden/toy_grad.npz (float64 autograd of the restatement; generator committed beside it)."""
    from mvsnet_amd import backward as B
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "toy_grad.npz"))
    w = S.make_workload("toy")
    rp = S.make_regnet_params("normal", seed=1, random_affine=True)
    ft = t(w.features[:3]).requires_grad_(True)
    pt = {k: {kk: t(vv).requires_grad_(True) for kk, vv in v.items()} for k, v in rp.items()}
    depth, _ = B.plane_sweep_depth(ft, t(g["t8"]), w.depth_start, w.depth_interval, pt)
    (depth * t(g["g"])).sum().backward()
    assert rel_l1(n(depth), g["depth"]) < 1e-5
    assert rel_l1(n(ft.grad), g["g_features"]) < 2e-3
    assert rel_l1(n(pt["3dconv0_1"]["w"].grad), g["g_w01"]) < 2e-3
    assert rel_l1(n(pt["3dconv6_2"]["w"].grad), g["g_w62"]) < 2e-3
    assert rel_l1(n(pt["3dconv3_0"]["gamma"].grad), g["g_gamma30"]) < 2e-3
    assert rel_l1(n(pt["3dconv6_0"]["beta"].grad), g["g_beta60"]) < 2e-3


_SYNC_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from mvsnet_amd import shard as sh, synthetic as S, backward as B
dist = sh.init_process_group("gloo")
rank, _local, world = sh.rank_world()
torch.cuda.set_device(0)                                  # both replicas share the box's one GPU
d = np.load(%(inp)r)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()
params = S.make_regnet_params("normal", seed=1, random_affine=True)
ft = t(d["feats"][rank]).requires_grad_(True)
pt = {k: {kk: t(vv).requires_grad_(True) for kk, vv in v.items()} for k, v in params.items()}
depth, _ = B.plane_sweep_depth(ft, t(d["t8"]), float(d["start"]), float(d["interval"]), pt, sync=B.SyncBN())
(depth * t(d["g"][rank])).sum().backward()
torch.cuda.synchronize()
out = {"depth": depth.detach().cpu().numpy(), "g_features": ft.grad.cpu().numpy()}
for k in pt:
    for kk in pt[k]:
        out["%%s/%%s" %% (k, kk)] = pt[k][kk].grad.cpu().numpy()
np.savez(%(out)r %% rank, **out)
dist.barrier()
dist.destroy_process_group()
"""


def test_sync_batchnorm_two_replicas_match_a_batch_of_two(tmp_path):
    """Cross-replica BatchNorm (backward.SyncBN): two processes, one sample each, against float64 autograd of the
    restatement on the batch of two (BatchNorm statistics over both volumes).  The replicas' variable gradients
    add up to the batch's; each replica's feature gradient is its sample's."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    f0, t8, start, interval = _toy_problem()
    f1 = S.make_features(3, 16, 32, 32, seed=9)
    rs = np.random.RandomState(21)
    g = rs.randn(2, 16, 32).astype(np.float32)
    inp = str(tmp_path / "in.npz")
    np.savez(inp, feats=np.stack([f0, f1]), t8=t8, start=start, interval=interval, g=g)
    outp = str(tmp_path / "rank%d.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", WORLD_SIZE="2")
    from _helpers import run_ranks
    outs = run_ranks(_SYNC_WORKER % {"root": root, "inp": inp, "out": outp}, env, world=2, timeout=300)
    assert all(rc == 0 for rc, _o, _e in outs), outs
    got = [np.load(outp % r) for r in range(2)]
    # checker: the batch of two through the restatement
    params = S.make_regnet_params("normal", seed=1, random_affine=True)
    p64 = {k: {kk: d64(vv, True) for kk, vv in v.items()} for k, v in params.items()}
    f64 = [d64(f0, True), d64(f1, True)]
    costs = torch.stack([TG.cost_volume(f, d64(t8)) for f in f64])
    reg = TG.regnet_us0(costs, p64)
    depth = torch.stack([TG.soft_argmin(reg[b], start, interval) for b in range(2)])
    (depth * d64(g)).sum().backward()
    for r in range(2):
        assert rel_l1(got[r]["depth"].astype(np.float64), depth[r].detach().numpy()) < 1e-5
        assert rel_l1(got[r]["g_features"].astype(np.float64), f64[r].grad.numpy()) < 2e-3
    for k in p64:
        for kk in p64[k]:
            tot = got[0]["%s/%s" % (k, kk)].astype(np.float64) + got[1]["%s/%s" % (k, kk)].astype(np.float64)
            assert rel_l1(tot, p64[k][kk].grad.numpy()) < 2e-3, (k, kk)


_DP_WORKER = r"""
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np, torch
from mvsnet_amd import shard as sh, synthetic as S, train as T
dist = sh.init_process_group("gloo")
rank, _local, world = sh.rank_world()
torch.cuda.set_device(0)
tr = T.Trainer("normal", "cuda", seed=0, sync_bn=%(sync)s, regularization=%(reg)r)       # same seed: identical replicas
N, H, W, D = 3, 64, 96, 16
images = S.make_images(N, H, W, seed=10 + rank)                  # every rank its own sample
cams = S.make_cams(N, H // 4, W // 4, D)
start, interval = float(cams[0, 1, 3, 0]), float(cams[0, 1, 3, 1])
gt = np.full((H // 4, W // 4, 1), start + interval * D * (0.4 + 0.2 * rank), np.float32)
losses = [float(tr.train_step(images, cams, gt, D)[0]) for _ in range(3)]
ck = torch.tensor([float(tr.params.data.double().sum()), float(tr.params.data.double().abs().sum())], dtype=torch.float64)
lo, hi = ck.clone(), ck.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
if rank == 0:
    print(json.dumps({"losses": losses, "equal": bool(torch.equal(lo, hi)), "steps": tr.global_step}))
dist.destroy_process_group()
"""


@pytest.mark.parametrize("sync,reg", [(False, "3DCNN"), (True, "3DCNN"), (False, "GRU")])
def test_data_parallel_training_keeps_two_replicas_identical(sync, reg):
    """Two processes (sharing the box's GPU, collectives over gloo) train on different samples: after the flat
    gradient all-reduce + optimiser step the replicas must hold bit-identical parameters."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29563" if sync else ("29565" if reg == "3DCNN" else "29567"), WORLD_SIZE="2")
    from _helpers import run_ranks
    outs = run_ranks(_DP_WORKER % {"root": root, "sync": sync, "reg": reg}, env, world=2, timeout=400)
    assert all(rc == 0 for rc, _o, _e in outs), outs
    res = json.loads(outs[0][1].strip().splitlines()[-1])
    assert res["equal"] and res["steps"] == 3 and all(np.isfinite(res["losses"]))


@pytest.mark.parametrize("mode,network,stereo", [("all", "unet", False), ("refine_only", "original", False),
                                                 ("main_only", "unet", False), ("all", "original", True)])
def test_trainer_with_refinement(tmp_path, mode, network, stereo):
    """Training through the refinement network (train.py:317-349): the refinement tower's variables are in the flat
    buffer and the checkpoint; refine_only leaves the main network untouched."""
    from mvsnet_amd import train as T
    from mvsnet_amd import tf_checkpoint
    images, cams, gt, D = _train_batch()
    H, W = images.shape[1], images.shape[2]
    full = np.repeat(np.repeat(gt, 4, axis=0), 4, axis=1).astype(np.float32)
    assert full.shape == (H, W, 1)
    tr = T.Trainer("normal", DEV, seed=0, refinement=True, refinement_network=network, refinement_train_mode=mode,
                   refine_with_stereo=stereo)
    before = tr.params.data.clone()
    losses = [float(tr.train_step(images, cams, gt, D, full)[0]) for _ in range(4)]
    assert all(np.isfinite(losses)), losses
    moved = {}
    for (group, _l, _f), _v, o, shape in tr.params.index:
        nel = int(np.prod(shape))
        moved[group] = moved.get(group, 0.0) + float((tr.params.data[o:o + nel] - before[o:o + nel]).abs().sum())
    assert moved["refine"] > 0
    if mode == "refine_only":
        assert moved["regnet"] == 0.0 and moved["unet"] == 0.0
    else:
        assert moved["regnet"] > 0 and moved["unet"] > 0
    prefix = tr.save(str(tmp_path))
    names = {nm for nm, _s, _d in tf_checkpoint.list_variables(prefix)}
    first = "refine_conv0" if network == "original" else "2dconv1_0_refine"
    assert first + "/kernel" in names and first + "/bias" in names
    loaded = tf_checkpoint.load_mvsnet_params(prefix, "normal", "3DCNN", refinement=network)
    assert loaded["refine"][first]["w"].shape[-2] == (8 if stereo else 5)   # image + depth + confidence (+ stereo partner)


# ---- recurrent regulariser (model.py:505-599, loss.py:223-267) -------------------------------------------------

def _gru_t(params, grad=True):
    return {k: ({kk: t(vv).requires_grad_(grad) for kk, vv in v.items()} if isinstance(v, dict) else t(v).requires_grad_(grad))
            for k, v in params.items()}


def _gru_64(params):
    return {k: ({kk: d64(vv, True) for kk, vv in v.items()} if isinstance(v, dict) else d64(v, True))
            for k, v in params.items()}


@pytest.mark.parametrize("mode,C", [("normal", 32), ("lite", 16)])
def test_recurrent_regularisation_gradients_match_autograd(mode, C):
    """features -> cost volume (HIP, both directions) -> 3 ConvGRU cells + prob_conv (x parts hoisted, torch autograd)
    -> classification loss: value, feature gradient and every cell parameter gradient against the float64
    plane-by-plane checker."""
    from mvsnet_amd import gru_train as G
    feats, t8, start, interval = _toy_problem(N=3, D=8, H=16, W=32, C=C)
    gp = S.make_gru_params(mode, in_channels=C, random_affine=True)
    rs = np.random.RandomState(4)
    gt = (start + interval * rs.uniform(0, 7, feats.shape[1:3])).astype(np.float32)
    gt[:2, :5] = 0.0
    f64, p64 = d64(feats, True), _gru_64(gp)
    reg64 = TG.recurrent_reg(f64, d64(t8), p64)
    loss64 = TG.classification_loss(reg64, d64(gt), start, interval)
    loss64.backward()
    ft, pt = t(feats).requires_grad_(True), _gru_t(gp)
    reg = G.recurrent_regularisation(ft, t(t8), pt)
    assert rel_l1(n(reg), reg64.detach().numpy()) < 1e-5
    loss = G.mvsnet_classification_loss(torch.softmax(reg, 0), t(gt)[None, :, :, None], 8, [start], [interval])[0]
    assert abs(float(loss.detach()) - float(loss64.detach())) < 1e-5 * abs(float(loss64.detach()))
    loss.backward()
    assert rel_l1(n(ft.grad), f64.grad.numpy()) < 2e-3
    for cell in ("gru1", "gru2", "gru3"):
        for key in p64[cell]:
            ref, got = p64[cell][key].grad.numpy(), n(pt[cell][key].grad)
            # (a bias / beta in front of a one-channel LayerNorm has a true gradient of zero: absolute floor)
            assert rel_l1(got, ref) < 2e-3 or np.abs(got - ref).max() < 1e-7, (cell, key, rel_l1(got, ref))
    assert rel_l1(n(pt["prob_w"].grad), p64["prob_w"].grad.numpy()) < 2e-3
    # a softmax over the planes does not see a bias shared by all planes: the true gradient of prob_conv/bias is zero
    assert abs(float(p64["prob_b"].grad)) < 1e-12 and abs(float(pt["prob_b"].grad)) < 1e-5


@pytest.mark.parametrize("mode", ["normal", "lite", "semilite-py3", "ultralite"])
def test_gru_trainer_reduces_the_loss_and_round_trips_its_checkpoint(tmp_path, mode):
    from mvsnet_amd import model as M
    from mvsnet_amd import tf_checkpoint
    from mvsnet_amd import train as T
    from mvsnet_amd.mvs_data_generation import flip_cams
    images, cams, gt, D = _train_batch()
    tr = T.Trainer(mode, DEV, regularization="GRU", base_lr=2e-3, seed=0)
    losses = []
    for _ in range(6):                                  # the cluster and its reversed sweep, as the generator yields them
        losses.append(float(tr.train_step(images, cams, gt, D)[0]))
        losses.append(float(tr.train_step(images, flip_cams(cams, D), gt, D)[0]))
    assert all(np.isfinite(losses)) and min(losses[-2:]) < losses[0], losses
    assert abs(losses[0] - np.log(D)) < 0.5             # an untrained softmax over D planes: cross entropy ~ log D
    prefix = tr.save(str(tmp_path))
    assert prefix.endswith(os.path.join("GRU", mode, "model.ckpt-12"))
    names = {nm for nm, _s, _d in tf_checkpoint.list_variables(prefix)}
    assert {"conv_gru1/Gates/conv/kernel", "conv_gru2/Output/LayerNorm/gamma/RMSProp", "prob_conv/bias", "global_step"} <= names
    tr2 = T.Trainer(mode, DEV, regularization="GRU", seed=77)
    tr2.restore(prefix)
    assert tr2.global_step == 12 and torch.equal(tr2.params.data, tr.params.data)
    assert all(torch.equal(a, b) for a, b in zip(tr2.slots, tr.slots))
    # the winner-take-all inference sweep (HIP) runs on what training wrote and agrees with the trainer's own argmax
    loaded = tf_checkpoint.load_mvsnet_params(prefix, mode, "GRU")
    weights = M.MVSNetWeights.from_numpy(mode, unet=loaded["unet"], gru=loaded["gru"], device=DEV)
    start, interval = float(cams[0, 1, 3, 0]), float(cams[0, 1, 3, 1])
    depth, _prob = M.inference_winner_take_all(t(images)[None], t(cams)[None], D, start, start + (D - 1) * interval, weights=weights)
    with torch.no_grad():
        _l, _a, _b, wta = tr.loss(images, cams, gt, D)
    agree = float((torch.abs(depth.reshape(wta.shape) - wta) < 0.5 * interval).float().mean())
    assert agree > 0.9, agree


def test_train_cli_gru_branch(tmp_path, capsys):
    from _helpers import add_depths, make_session
    from mvsnet_amd import train as T
    from mvsnet_amd import tf_checkpoint
    for mode in ("train", "val"):
        s = make_session(str(tmp_path / "data" / mode / "s0"), n_images=4, h=64, w=96, seed=3)
        add_depths(s, n_images=4, h=64, w=96, seed=3)
    out = tmp_path / "model"
    T.main(["--train_data_root", str(tmp_path / "data"), "--model_dir", str(out), "--network_mode", "lite",
            "--regularization", "GRU", "--view_num", "3", "--max_d", "16", "--width", "96", "--height", "64",
            "--base_image_size", "32", "--epoch", "1", "--max_steps_per_epoch", "2", "--snapshot", "2",
            "--train_steps_per_val", "2", "--val_batch_size", "1"])
    text = capsys.readouterr().out
    assert "total_step 4" in text and "Saving model to" in text and "VAL STEP COMPLETED" in text   # 2 clusters x 2 sweeps
    prefix = tf_checkpoint.model_path(tf_checkpoint.ckpt_path(str(out), "GRU", "lite"), 4)
    assert "prob_conv/kernel" in {nm for nm, _s, _d in tf_checkpoint.list_variables(prefix)}


@pytest.mark.parametrize("Cin,Fn,dims", [(32, 16, (5, 21, 37)), (16, 8, (4, 16, 32)), (16, 4, (6, 9, 50)), (4, 2, (5, 33, 17)),
                                         (2, 1, (4, 18, 20)), (8, 16, (3, 40, 16))])
def test_hip_conv_gru_sweep_matches_the_float64_cell_chain(Cin, Fn, dims):
    """mvs_gru_train_cell_fwd_f32 / _bwd_f32 (+ the host's batched convolutions): states and every gradient of one
    ConvGRU cell over all planes against the plane-by-plane concatenated cell of the checker, ragged tile sizes."""
    from mvsnet_amd import gru_train as G
    D, H, W = dims
    rs = np.random.RandomState(Cin * 100 + Fn)
    c = Cin + Fn
    p = {"gates_w": (rs.randn(3, 3, c, 2 * Fn) * (1.5 / np.sqrt(9 * c))).astype(np.float32),
         "out_w": (rs.randn(3, 3, c, Fn) * (1.5 / np.sqrt(9 * c))).astype(np.float32),
         "gates_b": (0.1 * rs.randn(2 * Fn)).astype(np.float32), "out_b": (0.1 * rs.randn(Fn)).astype(np.float32)}
    for nm in ("reset", "update", "out"):
        p[nm + "_gamma"] = (1 + 0.3 * rs.randn(Fn)).astype(np.float32)
        p[nm + "_beta"] = (0.2 * rs.randn(Fn)).astype(np.float32)
    x = rs.randn(D, H, W, Cin).astype(np.float32)
    gh = rs.randn(D, H, W, Fn).astype(np.float32)
    # checker
    x64 = d64(x, True)
    p64 = {k: d64(v, True) for k, v in p.items()}
    h = torch.zeros((1, Fn, H, W), dtype=torch.float64)
    states = []
    for d in range(D):
        h = TG.conv_gru_cell(x64[d].permute(2, 0, 1)[None], h, p64)
        states.append(h[0].permute(1, 2, 0))
    want = torch.stack(states, 0)
    (want * d64(gh)).sum().backward()
    # device
    xt = t(x).requires_grad_(True)
    pt = {k: t(v).requires_grad_(True) for k, v in p.items()}
    got = G.conv_gru_sweep_hip(xt, pt)
    assert tuple(got.shape) == (D, H, W, Fn)
    assert rel_l1(n(got), want.detach().numpy()) < 2e-6
    (got * t(gh)).sum().backward()
    assert rel_l1(n(xt.grad), x64.grad.numpy()) < 1e-4
    for k in p:
        ref, dev = p64[k].grad.numpy(), n(pt[k].grad)
        assert rel_l1(dev, ref) < 1e-4 or np.abs(dev - ref).max() < 1e-5, (k, rel_l1(dev, ref))


def test_full_size_conv_gru_sweep_backward_is_the_directional_derivative():
    """Size-independent property at train.py's full sizes (D = 192 planes of 120 x 160, cell 1: 32 -> 16): the
    gradient the HIP sweep returns for its input and for a kernel is the directional derivative of the forward,
    d/de <f(x + e v), w> = <grad_x, v>, central differences in float64 accumulation over the fp32 forward."""
    from mvsnet_amd import gru_train as G
    D, H, W, Cin, Fn = 192, 120, 160, 32, 16
    gen = torch.Generator(device="cpu").manual_seed(7)
    rnd = lambda *s: torch.randn(*s, generator=gen).to(DEV)
    c = Cin + Fn
    p = {"gates_w": rnd(3, 3, c, 2 * Fn) / np.sqrt(9 * c), "out_w": rnd(3, 3, c, Fn) / np.sqrt(9 * c),
         "gates_b": 0.1 * rnd(2 * Fn), "out_b": 0.1 * rnd(Fn)}
    for nm in ("reset", "update", "out"):
        p[nm + "_gamma"] = 1 + 0.2 * rnd(Fn)
        p[nm + "_beta"] = 0.1 * rnd(Fn)
    x, v, w = rnd(D, H, W, Cin), rnd(D, H, W, Cin), rnd(D, H, W, Fn)
    w[: D - 8] = 0                                        # weight the last planes: the longest back-propagation chains
    xt = x.clone().requires_grad_(True)
    pt = {k: t_.clone().requires_grad_(True) for k, t_ in p.items()}
    out = G.conv_gru_sweep_hip(xt, pt)
    assert torch.isfinite(out).all()
    (out.double() * w.double()).sum().backward()
    dot = lambda a, b: float((a.double() * b.double()).sum())
    with torch.no_grad():
        f = lambda xx, pp: dot(G.conv_gru_sweep_hip(xx, pp), w)
        eps = 1e-2
        num_x = (f(x + eps * v, p) - f(x - eps * v, p)) / (2 * eps)
        vw = rnd(3, 3, c, 2 * Fn) / np.sqrt(9 * c)
        pp, pm = dict(p), dict(p)
        pp["gates_w"], pm["gates_w"] = p["gates_w"] + eps * vw, p["gates_w"] - eps * vw
        num_w = (f(x, pp) - f(x, pm)) / (2 * eps)
    ana_x, ana_w = dot(xt.grad, v), dot(pt["gates_w"].grad, vw)
    assert abs(num_x - ana_x) < 2e-2 * max(abs(ana_x), 1.0), (num_x, ana_x)
    assert abs(num_w - ana_w) < 2e-2 * max(abs(ana_w), 1.0), (num_w, ana_w)


@pytest.mark.parametrize("mode", ["semilite-py3", "ultralite"])
def test_trainer_runs_the_other_narrow_modes(mode):
    """network.py:75-85: base_filter 6 ('semilite-py3': channel counts 6 / 12 / 24 / 48, outside the HIP GroupNorm tiling ->
    torch group_norm) and 2 ('ultralite'); the regulariser trains zero-padded as in 'lite'."""
    from mvsnet_amd import train as T
    images, cams, gt, D = _train_batch()
    tr = T.Trainer(mode, DEV, seed=0)
    losses = [float(tr.train_step(images, cams, gt, D)[0]) for _ in range(6)]
    assert all(np.isfinite(losses)) and min(losses[-2:]) < losses[0], losses


@pytest.mark.parametrize("cin,cout,cg,off", [(32, 48, 48, 0), (16, 32, 48, 0), (16, 16, 48, 32), (32, 32, 32, 0), (32, 16, 20, 4),
                                             (16, 48, 48, 0), (16, 12, 12, 0)])
@pytest.mark.parametrize("dims", [(3, 9, 37), (5, 16, 64), (2, 4, 32)])
def test_conv2d_weight_gradient_matches_autograd(cin, cout, cg, off, dims):
    """mvs_conv2d_wgrad_f32 (MFMA, deterministic) against float64 autograd of the 3x3 SAME convolution, ragged tiles, the
    output gradient as a channel slice of a wider tensor; (16, 12) has no kernel instance and takes the ATen route."""
    from mvsnet_amd import gru_train as G
    N, H, W = dims
    rs = np.random.RandomState(cin + cout + H)
    x = rs.randn(N, H, W, cin).astype(np.float32)
    g = rs.randn(N, H, W, cg).astype(np.float32)
    w64 = torch.zeros((3, 3, cin, cout), dtype=torch.float64, requires_grad=True)
    y = F.conv2d(d64(x).permute(0, 3, 1, 2), w64.permute(3, 2, 0, 1), padding=1)
    (y * d64(g[..., off:off + cout]).permute(0, 3, 1, 2)).sum().backward()
    got = G.conv2d_wgrad(t(x), t(g), off, cout)
    assert tuple(got.shape) == (3, 3, cin, cout)
    assert rel_l1(n(got), w64.grad.numpy()) < 2e-6
    assert torch.equal(got, G.conv2d_wgrad(t(x), t(g), off, cout)) or cout == 12          # bit-reproducible


@pytest.mark.parametrize("chunk", [1, 3, 5])
def test_recurrent_cells_wavefront_matches_the_cell_by_cell_sweeps(monkeypatch, chunk):
    """gru_train.RecurrentCells (three streams, planes handed over in chunks, the state gradient carried between chunk
    calls through dh_out -> dh_in) against the three single-cell sweeps run one after the other, ragged last chunk."""
    from mvsnet_amd import gru_train as G
    monkeypatch.setattr(G, "CHUNK", chunk)
    D, H, W, C = 8, 13, 22, 32
    rs = np.random.RandomState(chunk)
    gp = S.make_gru_params("normal", in_channels=C, random_affine=True)
    x = rs.randn(D, H, W, C).astype(np.float32)
    gh = rs.randn(D, H, W, 2).astype(np.float32)
    cells = ("gru1", "gru2", "gru3")

    def run(fn):
        xt = t(x).requires_grad_(True)
        pt = {k: {f: t(gp[k][f]).requires_grad_(True) for f in G.CELL_FIELDS} for k in cells}
        out = fn(xt, pt)
        (out * t(gh)).sum().backward()
        return n(out), n(xt.grad), {(k, f): n(pt[k][f].grad) for k in cells for f in G.CELL_FIELDS}

    def wavefront(xt, pt):
        return G.RecurrentCells.apply(xt, *[pt[k][f] for k in cells for f in G.CELL_FIELDS])

    def one_by_one(xt, pt):
        s = xt
        for k in cells:
            s = G.conv_gru_sweep_hip(s, pt[k])
        return s

    a, ga, pa = run(wavefront)
    b, gb, pb = run(one_by_one)
    assert np.array_equal(a, b)                                   # the forward is the same kernels on the same data
    assert rel_l1(ga, gb) < 1e-5
    for key in pa:
        assert rel_l1(pa[key], pb[key]) < 1e-4 or np.abs(pa[key] - pb[key]).max() < 1e-6, key


def test_new_training_entry_points_report_errors_instead_of_launching():
    """C ABI error behaviour (include/mvsnet_hip.h): null pointers -> MVS_E_BADARG (-1), shapes without a kernel
    instance -> MVS_E_SHAPE (-2), a workspace that is too small -> MVS_E_WORKSPACE (-3); nothing is launched."""
    import ctypes as C
    from mvsnet_amd import _lib as L
    lib = L.load()
    P = L.ptr
    z = lambda *s: torch.zeros(s, device=DEV)
    D, H, W, Fn = 2, 8, 16, 3                                     # F = 3 has no instance
    px, wgh, woh, ln = z(D, H, W, 3 * Fn), z(3, 3, Fn, 2 * Fn), z(3, 3, Fn, Fn), z(6, Fn)
    g, c, rh, h = z(D, H, W, 2 * Fn), z(D, H, W, Fn), z(D, H, W, Fn), z(D + 1, H, W, Fn)
    stats = torch.zeros((D, 8, 6), device=DEV, dtype=torch.float64)
    st = L.stream_ptr()
    assert lib.mvs_gru_train_cell_fwd_f32(P(px), P(wgh), P(woh), P(ln), D, H, W, Fn, P(g), P(c), P(rh), P(h), P(stats), st) == -2
    assert lib.mvs_gru_train_cell_fwd_f32(None, P(wgh), P(woh), P(ln), D, H, W, 4, P(g), P(c), P(rh), P(h), P(stats), st) == -1
    assert lib.mvs_gru_train_cell_fwd_f32(P(px), P(wgh), P(woh), P(ln), 0, H, W, 4, P(g), P(c), P(rh), P(h), P(stats), st) == -1
    part, scratch = torch.zeros((D, 3, 16, 2, Fn), device=DEV, dtype=torch.float64), z(6, H, W, Fn)
    assert lib.mvs_gru_train_cell_bwd_f32(P(c), P(g), P(c), P(h), P(stats), P(wgh), P(woh), P(ln), D, H, W, Fn, P(px), P(part),
                                          P(scratch), None, None, st) == -2
    assert lib.mvs_gru_train_cell_bwd_f32(P(c), P(g), P(c), P(h), P(stats), P(wgh), P(woh), P(ln), D, H, W, 4, None, P(part),
                                          P(scratch), None, None, st) == -1
    x, gg, dw = z(D, H, W, 16), z(D, H, W, 32), z(3, 3, 16, 32)
    need = lib.mvs_conv2d_wgrad_workspace_bytes(D, H, W, 16, 32)
    ws = torch.empty(need, device=DEV, dtype=torch.uint8)
    wp = C.c_void_p(ws.data_ptr())
    assert lib.mvs_conv2d_wgrad_f32(P(x), P(gg), 32, 0, D, H, W, 16, 32, wp, need, P(dw), st) == 0
    assert lib.mvs_conv2d_wgrad_f32(P(x), P(gg), 32, 0, D, H, W, 16, 32, wp, need - 1, P(dw), st) == -3
    assert lib.mvs_conv2d_wgrad_f32(P(x), P(gg), 32, 0, D, H, W, 16, 24, wp, need, P(dw), st) == -2      # no (16, 24) instance
    assert lib.mvs_conv2d_wgrad_f32(P(x), P(gg), 32, 16, D, H, W, 16, 32, wp, need, P(dw), st) == -1     # slice beyond the tensor
    assert lib.mvs_conv2d_wgrad_f32(P(x), P(gg), 30, 0, D, H, W, 16, 16, wp, need, P(dw), st) == -2      # stride not a multiple of 4
    assert lib.mvs_conv2d_wgrad_f32(None, P(gg), 32, 0, D, H, W, 16, 32, wp, need, P(dw), st) == -1
    torch.cuda.synchronize()


def test_config5_training_step_at_its_own_size():
    """BASELINE.json configs[4]'s per-GPU workload (mvsnet/train.py:412-445 with N = 3 views, D = 128 planes, 640 x 480
    images -> 160 x 120 feature maps): (1) the hot path's gradient at that size is the directional derivative of its forward
    -- <grad_features, v> and <grad_W(3dconv0_1), vw> against central differences of <depth(features), g>, float64
    accumulation over the fp32 forward (the identity the full-size kernel tests use); (2) whole training steps run on
    that batch and reduce the loss."""
    from mvsnet_amd import backward as B, train as T
    from mvsnet_amd.homography_warping import homography_transforms
    N, D, H, W = 3, 128, 120, 160
    cams = S.make_cams(N, H, W, D)
    start, interval = float(cams[0, 1, 3, 0]), float(cams[0, 1, 3, 1])
    t8 = homography_transforms(t(cams), D, start, interval)
    rp = S.make_regnet_params("normal", seed=1, random_affine=True)
    f0 = t(S.make_features(N, H, W, 32, seed=5))
    gen = torch.Generator(device="cpu").manual_seed(11)
    g = torch.randn(H, W, generator=gen).to(DEV)
    v = torch.randn(N, H, W, 32, generator=gen).to(DEV)
    vw = torch.randn(3, 3, 3, 32, 8, generator=gen).to(DEV) * float(np.sqrt(2.0 / (27 * 32)))
    ft = f0.clone().requires_grad_(True)
    pt = {k: {kk: t(vv).requires_grad_(True) for kk, vv in p.items()} for k, p in rp.items()}
    depth, _ = B.plane_sweep_depth(ft, t8, start, interval, pt)
    assert depth.shape == (H, W) and torch.isfinite(depth).all()
    (depth.double() * g.double()).sum().backward()
    dot = lambda a, b: float((a.double() * b.double()).sum())
    ana_f, ana_w = dot(ft.grad, v), dot(pt["3dconv0_1"]["w"].grad, vw)
    scale_f = sum(abs(dot(ft.grad[i], v[i])) for i in range(N))       # the views' contributions cancel: compare at THEIR scale
    with torch.no_grad():
        def fwd(feats, w01):
            p2 = {k: dict(p) for k, p in pt.items()}
            p2["3dconv0_1"]["w"] = w01
            return dot(B.plane_sweep_depth(feats, t8, start, interval, p2)[0], g)
        w01 = pt["3dconv0_1"]["w"].detach()
        num_f = (fwd(f0 + 5e-4 * v, w01) - fwd(f0 - 5e-4 * v, w01)) / (2 * 5e-4)
        num_w = (fwd(f0, w01 + 2e-3 * vw) - fwd(f0, w01 - 2e-3 * vw)) / (2 * 2e-3)
    print("config 5 size: <grad_f, v> %.6g vs central difference %.6g (scale %.4g); <grad_W, vw> %.6g vs %.6g" % (
        ana_f, num_f, scale_f, ana_w, num_w))
    # 15 M ReLU inputs and a soft-argmin over 128 planes sit between the features and the depth: the finite difference of the
    # whole path wanders by a few per cent with the step (tools/dd_probe.py); the strict checks are the float64-autograd
    # comparisons at small sizes and the per-kernel identities at full size above
    assert abs(num_f - ana_f) < 4e-2 * scale_f, (num_f, ana_f, scale_f)
    assert abs(num_w - ana_w) < 2e-2 * max(abs(ana_w), 1.0), (num_w, ana_w)
    images, cams_img, gt, _ = _train_batch(N=3, H=480, W=640, D=D)
    tr = T.Trainer("normal", DEV, seed=0)
    losses = [float(tr.train_step(images, cams_img, gt, D)[0]) for _ in range(4)]
    assert all(np.isfinite(losses)) and min(losses[1:]) < losses[0], losses


def test_narrow_tower_data_gradient_shape_that_made_miopen_over_read():
    """Round-2 GPU memory fault, kept as a regression case: the DATA gradient of a 3x3 convolution with 16 input and 8 output
    channels on a 32 x 48 channels_last map -- the shape for which MIOpen's igemm_bwd_gtcx35_nhwc_fp32 solver read 4 608 bytes
    past its (8,16,3,3) filter (a fault when the filter is the last block of an allocator segment).  With the package's
    switch (mvsnet_amd.ensure_miopen_workaround) the solver is off: the gradient is right and nothing faults, whatever
    the allocator does around the filter."""
    import mvsnet_amd
    assert mvsnet_amd.ensure_miopen_workaround("test")
    rs = np.random.RandomState(3)
    x = rs.randn(1, 16, 32, 48).astype(np.float32)
    g = rs.randn(1, 8, 32, 48).astype(np.float32)
    for rep in range(6):
        torch.cuda.empty_cache()                              # fresh segments: the filter lands at different places
        pad = [torch.empty(257 * (rep + 1), device=DEV) for _ in range(rep)]
        w = torch.as_tensor(rs.randn(8, 16, 3, 3).astype(np.float32)).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        xt = torch.as_tensor(x).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = F.conv2d(xt, w, padding=1)
        y.backward(torch.as_tensor(g).to(DEV).contiguous(memory_format=torch.channels_last))
        torch.cuda.synchronize()
        w64 = w.detach().double().cpu().requires_grad_(True)
        x64 = torch.as_tensor(x).double().requires_grad_(True)
        F.conv2d(x64, w64, padding=1).backward(torch.as_tensor(g).double())
        assert rel_l1(n(xt.grad), x64.grad.numpy()) < 1e-5
        assert rel_l1(n(w.grad), w64.grad.numpy()) < 1e-5
        del pad
