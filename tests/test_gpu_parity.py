"""GPU parity tests: every HIP entry point against the CPU oracle on the same seeded inputs.

All calls go through the C ABI (ctypes) of libmvsnet_hip.so.  Tolerances are stated per test;
the north-star bar is 1e-3 relative L1 on the depth map, the kernels are held to ~1e-5.
"""
import numpy as np
import pytest
import torch

from oracle import mvsnet_oracle as O
from mvsnet_amd import synthetic as S
from mvsnet_amd import _lib as L

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
    return x.detach().cpu().numpy()


def rel_l1(a, b):
    return float(np.abs(a - b).sum() / np.abs(b).sum())


# ---- R1 / R2 -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("inverse", [False, True])
def test_homographies_match_oracle(inverse):
    from mvsnet_amd.homography_warping import homography_transforms
    w = S.make_workload("M")
    cams = w.cams
    D = w.depth_num
    end = w.depth_start + (D - 1) * w.depth_interval
    T, Hm = homography_transforms(t(cams), D, w.depth_start, w.depth_interval, end, inverse, True)
    T, Hm = n(T), n(Hm)
    for v in range(1, w.view_num):
        if inverse:
            Ho = O.get_homographies_inv_depth(cams[0], cams[v], D, w.depth_start, end, np.float64)
        else:
            Ho = O.get_homographies(cams[0], cams[v], D, w.depth_start, w.depth_interval, np.float64)
        To = O.homography_to_transform8(Ho, np.float64)
        # fp32 evaluation of a ~20-flop chain with entries up to ~1e2: absolute 1e-4 on pixel terms
        np.testing.assert_allclose(Hm[v - 1], Ho, rtol=2e-5, atol=2e-4)
        np.testing.assert_allclose(T[v - 1], To, rtol=2e-5, atol=2e-4)


def test_identity_cameras_give_identity_transform():
    from mvsnet_amd.homography_warping import homography_transforms
    K = np.array([[128.0, 0, 64], [0, 128.0, 48], [0, 0, 1]])
    c = np.zeros((2, 2, 4, 4), np.float32)
    c[:, 0] = np.eye(4); c[:, 1, :3, :3] = K
    T = n(homography_transforms(t(c), 4, 100.0, 10.0))
    assert np.array_equal(T, np.broadcast_to(np.array([1, 0, 0, 0, 1, 0, 0, 0], np.float32), (1, 4, 8)))


# ---- R2 warp KATs on the device ------------------------------------------------------------------------
def test_warp_kats_shift_and_zero_fill():
    from mvsnet_amd.homography_warping import transform_image
    rs = np.random.RandomState(1)
    img = rs.randint(-9, 10, size=(5, 12, 8)).astype(np.float32)
    out = n(transform_image(t(img), t([1, 0, -4.0, 0, 1, 0, 0, 0])))
    exp = np.zeros_like(img); exp[:, 4:] = img[:, :-4]
    assert np.array_equal(out, exp)
    out = n(transform_image(t(img), t([1, 0, -2.5, 0, 1, 0, 0, 0])))
    exp = np.zeros_like(img)
    exp[:, 3:] = 0.5 * img[:, :-3] + 0.5 * img[:, 1:-2]
    exp[:, 2] = 0.5 * img[:, 0]
    assert np.array_equal(out, exp)
    out = n(transform_image(t(img), t([1, 0, 0, 0, 1, 0.5, 0, 0])))
    exp = np.zeros_like(img); exp[:-1] = 0.5 * img[:-1] + 0.5 * img[1:]; exp[-1] = 0.5 * img[-1]
    assert np.array_equal(out, exp)
    assert np.array_equal(n(transform_image(t(img), t([1, 0, 0, 0, 1, 0, 0, 0]))), img)
    # far outside: all taps zero
    assert not n(transform_image(t(img), t([1, 0, 500.0, 0, 1, 0, 0, 0]))).any()


@pytest.mark.parametrize("border", ["zeros", "clamp"])
def test_warp_projective_matches_oracle(border):
    from mvsnet_amd.homography_warping import tf_transform_homography
    rs = np.random.RandomState(2)
    img = rs.standard_normal((24, 40, 16)).astype(np.float32)
    for k in range(4):
        Hm = np.eye(3) + 0.03 * rs.standard_normal((3, 3))
        Hm[2, :2] *= 0.01
        Hm[:2, 2] += rs.uniform(-6, 6, 2)
        got = n(tf_transform_homography(t(img), t(Hm), border))
        if border == "zeros":
            exp = O.tf_transform_homography(img, Hm, np.float64)
        else:
            exp = O.homography_warping_clamp(img, Hm, np.float64)
        # a sample within 1e-5 px of an integer may pick the neighbouring tap pair: allow a few
        bad = np.abs(got - exp) > 1e-4 * (1 + np.abs(exp))
        assert bad.mean() < 1e-3, bad.mean()


# ---- R3 cost volume ------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", ["mem", "eager"])
def test_cost_volume_matches_oracle(variant):
    from mvsnet_amd.homography_warping import homography_transforms
    from mvsnet_amd.model import cost_volume
    w = S.make_workload("small")
    feats, cams = w.features, w.cams
    D = w.depth_num
    T = homography_transforms(t(cams), D, w.depth_start, w.depth_interval)
    got = n(cost_volume(t(feats[0]), t(feats[1:]), T, variant=variant))
    To = n(T).astype(np.float64)
    exp = np.empty_like(got, dtype=np.float64)
    fn = O.variance_cost_mem if variant == "mem" else O.variance_cost_eager
    for d in range(D):
        warped = [O.image_projective_transform_bilinear(feats[v + 1], To[v, d], np.float64)
                  for v in range(w.view_num - 1)]
        exp[d] = fn(feats[0], warped, w.view_num, np.float64)
    assert got.shape == (D, w.height, w.width, w.channels)
    err = np.abs(got - exp)
    assert (err > 1e-4).mean() < 1e-3          # tap flips at near-integer samples
    assert rel_l1(got, exp) < 1e-5
    # negated single plane (the GRU step input)
    one = n(cost_volume(t(feats[0]), t(feats[1:]), T, d_begin=3, d_count=1, variant=variant, negate=True))
    np.testing.assert_allclose(one[0], -got[3], rtol=1e-4, atol=5e-5)   # per-plane kernel vs depth sweep (fp32 cancellation noise)


def test_cost_volume_identical_views_is_zero_and_padding_views():
    from mvsnet_amd.model import cost_volume
    rs = np.random.RandomState(3)
    img = rs.randint(-4, 5, size=(8, 8, 32)).astype(np.float32)
    T = np.broadcast_to(np.array([1, 0, 0, 0, 1, 0, 0, 0], np.float32), (3, 5, 8)).copy()
    got = n(cost_volume(t(img), t(np.stack([img] * 3)), t(T)))
    assert got.shape == (5, 8, 8, 32) and not got.any()


@pytest.mark.parametrize("variant", ["mem", "eager"])
def test_cost_volume_border_bands_are_exact(variant):
    """The depth sweep reads ONE clamped 2x2 block per (pixel, view, plane) and moves the bilinear weights with it where
    the sample point lies in the one-pixel band around the image.  Integer features and shifts on a 1/4-pixel grid make
    every product exact, so the volume must equal the oracle's bit for bit: every plane puts the three source views in a
    different band (left / right / top / bottom / corners / two pixels out / far out), homography_warping.py:251-252."""
    from mvsnet_amd.model import cost_volume
    rs = np.random.RandomState(31)
    H, W, C = 6, 10, 32
    ref = rs.randint(-4, 5, size=(H, W, C)).astype(np.float32)
    src = rs.randint(-4, 5, size=(3, H, W, C)).astype(np.float32)
    shifts = [(-0.5, 0.0), (0.0, -0.5), (-0.75, -0.25), (-1.0, -1.0), (-1.25, 0.5), (0.25, -1.75), (-2.0, 0.0), (0.0, -2.5),
              (W - 1.5, 0.0), (W - 1.0, 0.25), (W - 0.75, H - 0.75), (0.5, H - 1.0), (0.0, H - 0.5), (W - 0.25, -0.25),
              (-0.25, H - 0.25), (W + 0.5, 0.0), (0.0, H + 1.0), (300.0, -300.0), (0.0, 0.0), (1.5, 2.25), (-9.0, -9.0),
              (W - 1.25, H - 1.25), (-1.0, H - 1.0), (W - 1.0, -1.0)]
    D = len(shifts)
    T = np.zeros((3, D, 8), np.float32)
    T[:, :, 0] = 1; T[:, :, 4] = 1
    for d in range(D):
        for v in range(3):
            sx, sy = shifts[(d + 7 * v) % D]
            T[v, d, 2] = sx - (v == 2) * (d % 5); T[v, d, 5] = sy + (v == 1) * (d % 3)
    got = n(cost_volume(t(ref), t(src), t(T), variant=variant))
    fn = O.variance_cost_mem if variant == "mem" else O.variance_cost_eager
    for d in range(D):
        warped = [O.image_projective_transform_bilinear(src[v], T[v, d].astype(np.float64), np.float64) for v in range(3)]
        exp = fn(ref, warped, 4, np.float64)
        # sums of at most 4 exact products of small integers and 1/16ths: exact in fp32; the 1/N scalings round once each
        np.testing.assert_allclose(got[d], exp, rtol=0, atol=2e-5, err_msg="plane %d" % d)
        # and a launch of that plane alone gives the same numbers
        one = n(cost_volume(t(ref), t(src), t(T), d_begin=d, d_count=1, variant=variant))
        np.testing.assert_allclose(got[d], one[0], rtol=0, atol=2e-5, err_msg="plane %d" % d)


@pytest.mark.parametrize("C", [4, 8, 16, 32, 64])
def test_cost_volume_wave_tile_shapes_give_the_same_bits(C):
    """The sweep lays a wave's pixels out as a rows x columns tile voted from the transforms (cost_volume.hip); every shape, forced
    through the test hook MVS_HOOK_CV_TILE_ROWS_LOG2 (mvs_set_test_hook), and the voted one compute each voxel with the same instructions: bit-identical volumes,
    ragged image sizes included (tiles hanging over the right and bottom edges)."""
    from mvsnet_amd.homography_warping import homography_transforms
    from mvsnet_amd.model import cost_volume
    w = S.make_workload("small")
    rs = np.random.RandomState(41)
    H, Wd = 11, 13
    feats = rs.standard_normal((w.view_num, H, Wd, C)).astype(np.float32)
    T = homography_transforms(t(w.cams), w.depth_num, w.depth_start, w.depth_interval)
    T = T * t(np.array([1, 1, Wd / w.width, 1, 1, H / w.height, 1, 1], np.float32))      # keep the samples inside the smaller image
    L.set_test_hook("cv_tile_rows_log2", -1)
    voted = n(cost_volume(t(feats[0]), t(feats[1:]), T))
    assert np.isfinite(voted).all() and voted.any()
    To = n(T).astype(np.float64)
    exp = np.stack([O.variance_cost_mem(feats[0], [O.image_projective_transform_bilinear(feats[v + 1], To[v, d], np.float64)
                                                   for v in range(w.view_num - 1)], w.view_num, np.float64) for d in range(w.depth_num)])
    assert rel_l1(voted, exp) < 1e-5 and (np.abs(voted - exp) > 1e-4).mean() < 2e-3      # tap flips at near-integer samples
    for rows_log2 in range(4):
        with L.test_hooks(cv_tile_rows_log2=rows_log2):
            got = n(cost_volume(t(feats[0]), t(feats[1:]), T))
        assert np.array_equal(got, voted), rows_log2
    # a sweep along y instead of x (transposed geometry) votes another shape: same bits as a forced one again
    Tt = T.clone(); Tt[..., [0, 1, 2, 3, 4, 5]] = T[..., [4, 3, 5, 1, 0, 2]]
    with L.test_hooks(cv_tile_rows_log2=0):
        forced = n(cost_volume(t(feats[0]), t(feats[1:]), Tt))
    assert np.array_equal(n(cost_volume(t(feats[0]), t(feats[1:]), Tt)), forced)


@pytest.mark.parametrize("hw", [(2, 2), (2, 9), (7, 2), (1, 6), (5, 1)])
def test_cost_volume_smallest_images(hw):
    """Two rows / two columns is the smallest image the one-block sweep takes (its clamped 2x2 block needs both); a single
    row or column goes to the per-plane kernel.  Exact against the oracle on a 1/4-pixel grid, as the border-band test."""
    from mvsnet_amd.model import cost_volume
    rs = np.random.RandomState(51)
    H, W = hw
    ref = rs.randint(-4, 5, size=(H, W, 32)).astype(np.float32)
    src = rs.randint(-4, 5, size=(2, H, W, 32)).astype(np.float32)
    shifts = [(0.0, 0.0), (-0.5, 0.25), (0.75, -0.75), (-1.25, 1.0), (1.5, 1.5), (-2.0, -2.0), (0.25, 0.5), (3.0, 0.0)]
    T = np.zeros((2, len(shifts), 8), np.float32)
    T[:, :, 0] = 1; T[:, :, 4] = 1
    for d, (sx, sy) in enumerate(shifts):
        T[0, d, 2], T[0, d, 5] = sx, sy
        T[1, d, 2], T[1, d, 5] = -sy, sx
    got = n(cost_volume(t(ref), t(src), t(T)))
    for d in range(len(shifts)):
        warped = [O.image_projective_transform_bilinear(src[v], T[v, d].astype(np.float64), np.float64) for v in range(2)]
        np.testing.assert_allclose(got[d], O.variance_cost_mem(ref, warped, 3, np.float64), rtol=0, atol=2e-5, err_msg="plane %d" % d)


# ---- R4 conv / deconv / BN -----------------------------------------------------------------------------
CONV_CASES = [  # D,H,W,Cin,Cout,stride
    (8, 8, 16, 32, 8, 1), (8, 8, 16, 32, 16, 2), (4, 8, 8, 16, 16, 1), (4, 4, 8, 16, 32, 2),
    (4, 4, 4, 64, 64, 1), (8, 16, 16, 8, 1, 1), (6, 10, 12, 8, 8, 1), (6, 10, 12, 12, 6, 2),
    # widths 40 / 20 of the quarter / eighth resolution levels: 2x8 and 4x4 MFMA column tiles
    (6, 16, 24, 32, 32, 1), (5, 12, 40, 32, 16, 1), (4, 16, 20, 64, 64, 1), (5, 32, 12, 64, 16, 1),
    (7, 8, 16, 32, 8, 1), (6, 16, 32, 16, 16, 1),
    # network_mode 'fat' (base_filter 16): 3dconv1_0, 3_0, 3_1 on the block kernels, ragged blocks
    (6, 10, 14, 64, 32, 2), (5, 9, 12, 64, 128, 2), (3, 6, 10, 128, 128, 1),
]


@pytest.mark.parametrize("shape", [(8, 8, 16), (6, 16, 32), (10, 12, 24), (20, 8, 16)])
def test_conv3d_pair_matches_oracle(shape):
    """The fused pass over the cost volume (3dconv0_1 + 3dconv1_0) against two oracle convolutions;
    shapes cover partial h/w tiles and depth chunks that split the stride-2 output planes."""
    from mvsnet_amd.model import conv3d_pair
    D, H, W = shape
    rs = np.random.RandomState(D * 1000 + H * 10 + W)
    x = rs.standard_normal((D, H, W, 32)).astype(np.float32)
    w1 = (rs.standard_normal((3, 3, 3, 32, 8)) / np.sqrt(27 * 32)).astype(np.float32)
    w2 = (rs.standard_normal((3, 3, 3, 32, 16)) / np.sqrt(27 * 32)).astype(np.float32)
    s1 = torch.zeros((2, 8), dtype=torch.float64, device=DEV)
    s2 = torch.zeros((2, 16), dtype=torch.float64, device=DEV)
    y1, y2 = conv3d_pair(t(x), t(w1), t(w2), s1, s2)
    e1 = O.conv3d_same(x, w1, 1, np.float64)
    e2 = O.conv3d_same(x, w2, 2, np.float64)
    np.testing.assert_allclose(n(y1), e1, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(n(y2), e2, rtol=1e-4, atol=2e-5)
    for st, e, c in ((n(s1), e1, 8), (n(s2), e2, 16)):
        np.testing.assert_allclose(st[0], e.reshape(-1, c).sum(0), rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(st[1], (e.reshape(-1, c) ** 2).sum(0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("impl", ["scalar", "auto"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d_matches_oracle(case, impl):
    from mvsnet_amd import _lib as L
    from mvsnet_amd.model import conv3d, bn_finalize
    D, H, W, Cin, Cout, stride = case
    rs = np.random.RandomState(sum(case))
    x = rs.standard_normal((D, H, W, Cin)).astype(np.float32)
    x2 = rs.standard_normal((D, H, W, Cin)).astype(np.float32)
    wgt = (rs.standard_normal((3, 3, 3, Cin, Cout)) / np.sqrt(27 * Cin)).astype(np.float32)
    sc = (1 + 0.3 * rs.standard_normal(Cin)).astype(np.float32); sh = (0.2 * rs.standard_normal(Cin)).astype(np.float32)
    sc2 = (1 + 0.3 * rs.standard_normal(Cin)).astype(np.float32); sh2 = (0.2 * rs.standard_normal(Cin)).astype(np.float32)
    L.set_conv_impl(impl)
    try:
        stats = torch.zeros((2, Cout), dtype=torch.float64, device=DEV)
        y_plain = n(conv3d(t(x), t(wgt), stride))
        y_fused = n(conv3d(t(x), t(wgt), stride, (t(sc), t(sh)), t(x2), (t(sc2), t(sh2)), stats))
        st = n(stats)
    finally:
        L.set_conv_impl("auto")
    e_plain = O.conv3d_same(x, wgt, stride, np.float64)
    xin = np.maximum(x * sc + sh, 0).astype(np.float64) + np.maximum(x2 * sc2 + sh2, 0)
    e_fused = O.conv3d_same(xin, wgt, stride, np.float64)
    np.testing.assert_allclose(y_plain, e_plain, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(y_fused, e_fused, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(st[0], e_fused.reshape(-1, Cout).sum(0), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(st[1], (e_fused.reshape(-1, Cout) ** 2).sum(0), rtol=1e-4, atol=1e-3)
    # BN finalise of those statistics against the oracle's batch-norm
    gamma = (1 + 0.2 * rs.standard_normal(Cout)).astype(np.float32); beta = (0.1 * rs.standard_normal(Cout)).astype(np.float32)
    scale, shift = bn_finalize(stats, e_fused.size // Cout, t(gamma), t(beta))
    got = np.maximum(e_fused * n(scale) + n(shift), 0)
    exp = O.batch_norm_train(e_fused, gamma, beta, 1e-5, True, np.float64)
    np.testing.assert_allclose(got, exp, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("planes", [0, 1, 2, 3])
@pytest.mark.parametrize("cin,cout", [(16, 16), (16, 32), (32, 16), (32, 64)])
@pytest.mark.parametrize("depth", [2, 4, 5, 6, 10, 14])
def test_stride2_plane_ranges(cin, cout, depth, planes):
    """Pins the code shape of conv3d_s2_kernel: with a separate path for the last even plane of a
    workgroup's range, hipcc hoisted the staging arithmetic above both paths and gfx950 codegen reused a register
    before it was read (wrong results for Cin = 16 only).  Every residue of the 4-way unrolled plane march
    (T = 2n+1 input planes, n = 1..7), ranges that end inside / at the end of the volume, both input widths and an
    odd depth, with the range length forced through the test hook MVS_HOOK_S2_PLANES (0 = the launcher's own choice)."""
    from mvsnet_amd.model import conv3d
    rs = np.random.RandomState(1000 * cin + 10 * depth + planes)
    H, W = 6, 20
    x = rs.standard_normal((depth, H, W, cin)).astype(np.float32)
    wgt = (rs.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    sc = (1 + 0.3 * rs.standard_normal(cin)).astype(np.float32); sh = (0.2 * rs.standard_normal(cin)).astype(np.float32)
    stats = torch.zeros((2, cout), dtype=torch.float64, device=DEV)
    with L.test_hooks(s2_planes=planes):
        y = n(conv3d(t(x), t(wgt), 2, (t(sc), t(sh)), None, None, stats))
    e = O.conv3d_same(np.maximum(x * sc + sh, 0).astype(np.float64), wgt, 2, np.float64)
    assert y.shape == e.shape
    np.testing.assert_allclose(y, e, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(n(stats)[0], e.reshape(-1, cout).sum(0), rtol=1e-4, atol=1e-3)


DECONV_CASES = [(2, 2, 4, 64, 32), (4, 4, 8, 32, 16), (4, 8, 8, 16, 8), (3, 5, 6, 8, 4), (3, 5, 6, 128, 64)]     # last: 'fat' 3dconv4_0


@pytest.mark.parametrize("impl", ["scalar", "auto"])
@pytest.mark.parametrize("case", DECONV_CASES)
def test_deconv3d_matches_oracle(case, impl):
    from mvsnet_amd import _lib as L
    from mvsnet_amd.model import conv3d
    D, H, W, Cin, Cout = case
    rs = np.random.RandomState(sum(case) + 7)
    x = rs.standard_normal((D, H, W, Cin)).astype(np.float32)
    x2 = rs.standard_normal((D, H, W, Cin)).astype(np.float32)
    wgt = (rs.standard_normal((3, 3, 3, Cout, Cin)) / np.sqrt(27 * Cin)).astype(np.float32)
    sc = (1 + 0.3 * rs.standard_normal(Cin)).astype(np.float32); sh = (0.2 * rs.standard_normal(Cin)).astype(np.float32)
    L.set_conv_impl(impl)
    try:
        stats = torch.zeros((2, Cout), dtype=torch.float64, device=DEV)
        y_plain = n(conv3d(t(x), t(wgt), transpose=True))
        y_fused = n(conv3d(t(x), t(wgt), 2, (t(sc), t(sh)), t(x2), (t(sc), t(sh)), stats, transpose=True))
        st = n(stats)
    finally:
        L.set_conv_impl("auto")
    e_plain = O.conv3d_transpose_same(x, wgt, 2, np.float64)
    xin = np.maximum(x * sc + sh, 0).astype(np.float64) + np.maximum(x2 * sc + sh, 0)
    e_fused = O.conv3d_transpose_same(xin, wgt, 2, np.float64)
    assert y_plain.shape == (2 * D, 2 * H, 2 * W, Cout)
    np.testing.assert_allclose(y_plain, e_plain, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(y_fused, e_fused, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(st[0], e_fused.reshape(-1, Cout).sum(0), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(st[1], (e_fused.reshape(-1, Cout) ** 2).sum(0), rtol=1e-4, atol=1e-3)


def test_conv_kat_pad_side_and_crop_on_device():
    from mvsnet_amd.model import conv3d
    x = np.arange(64 * 4, dtype=np.float32).reshape(4, 4, 4, 4)
    w = np.zeros((3, 3, 3, 4, 4), np.float32); w[0, 0, 0] = np.eye(4)
    y = n(conv3d(t(x), t(w), 2))
    assert np.array_equal(y, x[::2, ::2, ::2])
    w = np.zeros((3, 3, 3, 4, 4), np.float32); w[2, 2, 2] = np.eye(4)
    y = n(conv3d(t(x), t(w), 2))
    exp = np.zeros((2, 2, 2, 4), np.float32); exp[0, 0, 0] = x[2, 2, 2]
    assert np.array_equal(y, exp)
    wt = np.zeros((3, 3, 3, 4, 4), np.float32)
    for k, v in enumerate([10, 20, 30]):
        wt[k, 0, 0] = v * np.eye(4)
    xin = np.zeros((2, 2, 2, 4), np.float32); xin[1, 1, 1, 2] = 1
    y = n(conv3d(t(xin), t(wt), transpose=True))
    exp = np.zeros((4, 4, 4, 4), np.float32); exp[2, 2, 2, 2] = 10; exp[3, 2, 2, 2] = 20
    assert np.array_equal(y, exp)


@pytest.mark.parametrize("pad", [True, False])
@pytest.mark.parametrize("mode,shape", [("normal", (8, 16, 16)), ("lite", (16, 8, 24)), ("semilite-py3", (8, 8, 8)),
                                        ("ultralite", (8, 16, 16))])
def test_regnet_matches_oracle(mode, shape, pad):
    """pad=True: narrower modes run zero-padded on the MFMA kernels' shapes (model.pad_regnet_params);
    pad=False: their native channel counts on the shape-generic kernels."""
    from mvsnet_amd.model import RegNetWeights, regnet_us0
    D, H, W = shape
    params = S.make_regnet_params(mode, seed=11, random_affine=True)
    C = 4 * S.base_filter(mode)
    rs = np.random.RandomState(12)
    cost = np.abs(rs.standard_normal((D, H, W, C))).astype(np.float32)
    wts = RegNetWeights(params, DEV, pad_to_mfma=pad)
    assert wts.cin == (32 if pad else C) and wts.cin_native == C
    got = n(regnet_us0(t(cost), wts))
    exp = O.regnet_us0(cost, params, np.float64)
    assert got.shape == (D, H, W)
    assert rel_l1(got, exp) < 2e-5
    np.testing.assert_allclose(got, exp, rtol=1e-3, atol=2e-4)


@pytest.mark.parametrize("shape", [(8, 8, 8), (8, 24, 40), (16, 8, 24), (24, 40, 8), (8, 16, 56)])
def test_regnet_auto_mode_on_small_odd_volumes(shape):
    """with partial BatchNorm rows a layer whose MFMA launcher answers MVS_E_SHAPE for the volume's (D, H, W) used
    to be a hard error in AUTO mode; it now takes the shape-generic kernel for that layer, whose BatchNorm finalisation folds
    the producers' partial rows first.  Volumes whose lower levels are 1..7 voxels wide."""
    from mvsnet_amd.model import RegNetWeights, regnet_us0
    D, H, W = shape
    params = S.make_regnet_params("normal", seed=21, random_affine=True)
    rs = np.random.RandomState(22)
    cost = np.abs(rs.standard_normal((D, H, W, 32))).astype(np.float32)
    got = n(regnet_us0(t(cost), RegNetWeights(params, DEV)))
    exp = O.regnet_us0(cost, params, np.float64)
    assert rel_l1(got, exp) < 2e-5
    np.testing.assert_allclose(got, exp, rtol=1e-3, atol=2e-4)


@pytest.mark.parametrize("shape", [(32, 32, 64), (48, 40, 72), (192, 128, 160)])
def test_regnet_filler_launches_equal_the_layers_apart(shape):
    """Round 4: with prepared weights 3dconv2_1 has no launch of its own -- its blocks ride as filler workgroups behind
    the blocks of 3dconv3_0 / 3_1 / 4_0 (conv3d_os.hip, conv3d_os_filled_kernel; shares: mvs_regnet_filler_shares).
    Without prepared weights (mvs_regnet_us0_f32) the four layers are launched apart.  Same arithmetic per block, so the
    two agree to the order of the float64 BatchNorm atomics; both against the oracle at the small sizes."""
    import ctypes as C
    from mvsnet_amd import _lib as L
    from mvsnet_amd.model import BN_EPSILON, RegNetWeights, regnet_us0
    lib = L.load()
    shares = (C.c_int * 3)()
    assert lib.mvs_regnet_filler_shares(shares) == 0 and sum(shares) == 1000 and min(shares) >= 0
    D, H, W = shape
    params = S.make_regnet_params("normal", seed=31, random_affine=True)
    cost = t(np.abs(np.random.RandomState(32).standard_normal((D, H, W, 32))).astype(np.float32))
    wts = RegNetWeights(params, DEV)
    filled = regnet_us0(cost, wts)
    apart = torch.empty_like(filled)
    ws = torch.empty(lib.mvs_regnet_workspace_bytes(D, H, W, 32, wts.base), device=DEV, dtype=torch.uint8)
    L.check(lib.mvs_regnet_us0_f32(L.ptr(cost), D, H, W, 32, wts.base, wts.w_ptrs, wts.g_ptrs, wts.b_ptrs, BN_EPSILON,
                                   C.c_void_p(ws.data_ptr()), ws.numel(), L.ptr(apart), L.stream_ptr()), "mvs_regnet_us0_f32")
    torch.cuda.synchronize()
    scale = float(apart.abs().max())
    assert float((filled - apart).abs().max()) <= 2e-5 * scale
    if D * H * W <= 48 * 40 * 72:
        exp = O.regnet_us0(n(cost), params, np.float64)
        assert rel_l1(n(filled), exp) < 2e-5


@pytest.mark.parametrize("shape", [(192, 128, 160), (104, 48, 64), (40, 24, 48)])
def test_regnet_span_and_fused_pair_equal_the_plain_schedules(shape):
    """The dominant launch's SPAN schedule (conv3d_c8.hip: workgroups own depth ranges that may cross a tile boundary) against the
    whole-chunk schedule it replaced, and the fused 3dconv1_1 + 3dconv2_0 launch (conv3d_mfma.hip, FUSE2) against the two layers
    apart -- forced through the library's test hooks MVS_HOOK_CONV_NO_SPAN / MVS_HOOK_CONV_NO_FUSE2, at the metric size (where SPAN is
    taken) and at sizes whose half-resolution depth (52, 20) is not a multiple of the planes a workgroup marches (short last
    chunk).  Same arithmetic per voxel; the float64 BatchNorm atomics arrive in another order: 2e-5 of the output's scale."""
    from mvsnet_amd.model import RegNetWeights, regnet_us0
    D, H, W = shape
    params = S.make_regnet_params("normal", seed=33, random_affine=True)
    cost = t(np.abs(np.random.RandomState(34).standard_normal((D, H, W, 32))).astype(np.float32))
    wts = RegNetWeights(params, DEV)
    default = regnet_us0(cost, wts).clone()
    scale = float(default.abs().max())
    assert scale > 0 and bool(torch.isfinite(default).all())
    for hooks in (("conv_no_span",), ("conv_no_fuse2",), ("conv_no_span", "conv_no_fuse2")):
        with L.test_hooks(**{k: 1 for k in hooks}):
            plain = regnet_us0(cost, wts).clone()
        assert float((plain - default).abs().max()) <= 2e-5 * scale, hooks
    if D * H * W <= 48 * 40 * 72:
        assert rel_l1(n(default), O.regnet_us0(n(cost), params, np.float64)) < 2e-5


def test_batch_of_two_shares_batchnorm_statistics():
    """FLAGS.batch_size > 1 (model.py:28,350,431,479): towers / homographies / cost volumes / soft-argmin per sample,
    RegNetUS0's BatchNorm over the whole batch (network.py:496-506) -- against the batched oracle, and NOT equal to
    running the two samples one by one; the recurrent path is per sample throughout."""
    from mvsnet_amd.model import MVSNetWeights, inference_mem, inference_winner_take_all
    wa, wb = S.make_workload("small", seed=0), S.make_workload("small", seed=5)
    rp = S.make_regnet_params("normal", seed=1, random_affine=True)
    gp = S.make_gru_params("normal", seed=2, in_channels=wa.channels, random_affine=True)
    weights = MVSNetWeights.from_numpy("normal", regnet=rp, gru=gp, device=DEV)
    feats = t(np.stack([wa.features, wb.features]))
    cams_b = np.stack([wa.cams, wb.cams]); cams_b[1, :, 1, 3, 0] += 10.0
    starts = np.array([wa.depth_start, wa.depth_start + 10.0], np.float32)
    intervals = np.array([wa.depth_interval, wa.depth_interval * 1.1], np.float32)
    depth, prob = inference_mem(None, t(cams_b), wa.depth_num, starts, intervals, weights=weights, features=feats)
    assert depth.shape == (2, wa.height, wa.width, 1) and prob.shape == depth.shape
    costs = []
    for b, w in enumerate((wa, wb)):
        Hs = np.stack([O.get_homographies(w.cams[0], w.cams[v], w.depth_num, float(starts[b]), float(intervals[b]), np.float64)
                       for v in range(1, w.view_num)])
        costs.append(O.cost_volume(w.features[0], w.features[1:], Hs, w.view_num, "mem", np.float64))
    reg = O.regnet_us0_batch(costs, rp, np.float64)
    for b in range(2):
        ed, ep = O.softargmin_and_prob(reg[b], wa.depth_num, float(starts[b]), float(intervals[b]), False, np.float64)
        d = n(depth)[b, :, :, 0]
        assert float(np.mean(np.abs(d - ed) / ed)) < 1e-4
        assert (np.abs(n(prob)[b, :, :, 0] - ep) > 1e-3).mean() < 0.02
    alone, _ = inference_mem(None, t(wa.cams)[None], wa.depth_num, float(starts[0]), float(intervals[0]), weights=weights,
                             features=t(wa.features))
    assert float((alone[0] - depth[0]).abs().mean()) > 1e-3                 # the batch statistics really couple the samples
    # recurrent path: a batch is its samples one by one
    ends = starts + (wa.depth_num - 1) * intervals
    dg, pg = inference_winner_take_all(None, t(cams_b), wa.depth_num, starts, ends, weights=weights, features=feats)
    d1, p1 = inference_winner_take_all(None, t(cams_b[1])[None], wa.depth_num, float(starts[1]), float(ends[1]), weights=weights,
                                       features=t(wb.features))
    assert dg.shape == (2, wa.height, wa.width, 1) and torch.equal(dg[1], d1[0]) and torch.allclose(pg[1], p1[0])


def test_fat_mode_matches_oracle():
    """network_mode 'fat' (network.py:82-83: base_divisor 0.5, base_filter 16, a 64-channel volume): wider than the shapes
    the MFMA kernels tile, so RegNetUS0 runs on the shape-generic kernels; the regulariser alone and features -> depth."""
    from mvsnet_amd.model import MVSNetWeights, RegNetWeights, regnet_us0, inference_mem
    assert S.base_filter("fat") == 16
    params = S.make_regnet_params("fat", seed=11, random_affine=True)
    rs = np.random.RandomState(13)
    cost = np.abs(rs.standard_normal((8, 8, 16, 64))).astype(np.float32)
    wts = RegNetWeights(params, DEV)
    assert wts.cin == 64 and wts.cin_native == 64
    got = n(regnet_us0(t(cost), wts))
    exp = O.regnet_us0(cost, params, np.float64)
    assert rel_l1(got, exp) < 2e-5
    w = S.make_workload("toy", network_mode="fat")
    assert w.channels == 64
    weights = MVSNetWeights.from_numpy("fat", regnet=params, device=DEV)
    depth, _prob = inference_mem(None, t(w.cams)[None], w.depth_num, w.depth_start, w.depth_interval, "fat",
                                 weights=weights, features=t(w.features))
    ed, _ep = O.inference_mem_from_features(w.features, w.cams, w.depth_num, w.depth_start, w.depth_interval, params, False, np.float64)
    assert float(np.mean(np.abs(n(depth)[0, :, :, 0] - ed) / ed)) < 1e-4


# ---- R6 / R7 -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("inverse", [False, True])
def test_softargmin_prob_matches_oracle(inverse):
    from mvsnet_amd.model import softargmin_prob, get_probability_map
    rs = np.random.RandomState(5)
    D, H, W = 48, 20, 28
    reg = (2.0 * rs.standard_normal((D, H, W))).astype(np.float32)
    reg[7, 3, 4] = -60.0                       # near one-hot pixel
    start, interval = 425.0, 2.65
    depth, prob = softargmin_prob(t(reg), start, interval, inverse)
    depth, prob = n(depth), n(prob)
    ed, ep = O.softargmin_and_prob(reg, D, start, interval, inverse, np.float64)
    np.testing.assert_allclose(depth, ed, rtol=2e-6)
    # bucket indices flip when (depth-start)/interval is within rounding of an integer
    bad = np.abs(prob - ep) > 1e-5
    assert bad.mean() < 5e-3
    P = O.softmax_neg(reg, np.float64)
    pm = n(get_probability_map(t(P), t(ed), start, interval, inverse))
    assert (np.abs(pm - ep) > 1e-5).mean() < 5e-3


def test_softargmin_kats_on_device():
    from mvsnet_amd.model import softargmin_prob
    D, start, interval = 8, 100.0, 2.0
    reg = np.zeros((D, 1, 64), np.float32)
    depth, prob = softargmin_prob(t(reg), start, interval)
    np.testing.assert_allclose(n(depth), 107.0, rtol=1e-6)
    np.testing.assert_allclose(n(prob), 0.5, rtol=1e-6)
    for k, expect in ((0, 3.0), (3, 2.0), (7, 3.0)):
        reg = np.zeros((D, 1, 3), np.float32); reg[k] = -1000.0
        depth, prob = softargmin_prob(t(reg), start, interval)
        assert np.all(n(depth) == start + interval * k) and np.all(n(prob) == expect)


# ---- R8 / R9 -------------------------------------------------------------------------------------------
def test_convgru_cell_stages_match_oracle():
    from mvsnet_amd import _lib as L
    lib = L.load()
    rs = np.random.RandomState(6)
    H, W, Cin, F = 12, 20, 32, 16
    p = S.make_gru_params("normal", seed=8, in_channels=Cin, random_affine=True)["gru1"]
    x = rs.standard_normal((H, W, Cin)).astype(np.float32)
    h = rs.standard_normal((H, W, F)).astype(np.float32)
    dx, dh = t(x), t(h)
    g = torch.empty((H, W, 2 * F), device=DEV); c = torch.empty((H, W, F), device=DEV)
    rh = torch.empty((H, W, F), device=DEV); u = torch.empty((H, W, F), device=DEV)
    st = torch.zeros(6, dtype=torch.float64, device=DEV)
    P = {k: t(v) for k, v in p.items()}
    sp = L.stream_ptr()
    L.check(lib.mvs_conv2d_cat_f32(L.ptr(dx), Cin, L.ptr(dh), F, L.ptr(P["gates_w"]), L.ptr(P["gates_b"]),
                                   H, W, 2 * F, L.ptr(g), L.ptr(st), 2, sp))
    L.check(lib.mvs_gru_gates_f32(L.ptr(g), L.ptr(st), L.ptr(P["reset_gamma"]), L.ptr(P["reset_beta"]),
                                  L.ptr(P["update_gamma"]), L.ptr(P["update_beta"]), L.ptr(dh), H, W, F,
                                  L.ptr(rh), L.ptr(u), sp))
    L.check(lib.mvs_conv2d_cat_f32(L.ptr(dx), Cin, L.ptr(rh), F, L.ptr(P["out_w"]), L.ptr(P["out_b"]),
                                   H, W, F, L.ptr(c), L.ptr(st[4:]), 1, sp))
    L.check(lib.mvs_gru_blend_f32(L.ptr(c), L.ptr(st[4:]), L.ptr(P["out_gamma"]), L.ptr(P["out_beta"]),
                                  L.ptr(u), H, W, F, L.ptr(dh), sp))
    exp = O.conv_gru_cell(x, h, p, np.float64)
    np.testing.assert_allclose(n(dh), exp, rtol=1e-4, atol=2e-5)


def test_gru_wta_pipelined_sweep_with_ragged_depth_matches_oracle():
    """D = 37 planes: three cost-volume batches (16 + 16 + 5), nine full synchronisation groups + one plane,
    several wraps of the 8-plane state ring -- the multi-stream wavefront (toy's D = 8 runs on one stream)."""
    from mvsnet_amd.model import MVSNetWeights, inference_winner_take_all
    w = S.make_workload("toy")
    D = 37
    cams = w.cams.copy()
    interval = 12.0
    cams[:, 1, 3, 1] = interval; cams[:, 1, 3, 2] = D; cams[:, 1, 3, 3] = w.depth_start + interval * D
    end = w.depth_start + (D - 1) * interval
    gp = S.make_gru_params("normal", seed=5, in_channels=w.channels, random_affine=True)
    weights = MVSNetWeights.from_numpy("normal", gru=gp, device=DEV)
    depth, prob = inference_winner_take_all(None, t(cams)[None], D, w.depth_start, end, weights=weights, features=t(w.features))
    ed, ep = O.inference_winner_take_all_from_features(w.features, cams, D, w.depth_start, end, gp, False, np.float64)
    depth, prob = n(depth)[0, :, :, 0], n(prob)[0, :, :, 0]
    same = np.abs(depth - ed) <= 1e-6 * np.abs(ed)
    assert same.mean() > 0.97, same.mean()
    np.testing.assert_allclose(prob[same], ep[same], rtol=5e-4)
    # and it is reproducible run to run
    d2, p2 = inference_winner_take_all(None, t(cams)[None], D, w.depth_start, end, weights=weights, features=t(w.features))
    assert np.array_equal(n(d2)[0, :, :, 0], depth) and np.array_equal(n(p2)[0, :, :, 0], prob)


@pytest.mark.parametrize("inverse,mode", [(False, "normal"), (True, "normal"), (False, "lite"), (False, "fat")])
def test_gru_wta_matches_oracle(inverse, mode):
    """'lite' (16-channel features, GRU filters 8 / 2 / 1) and 'fat' (64-channel features, filters 32 / 8 / 4) take the
    shape-generic conv + gate kernels for cell 1."""
    from mvsnet_amd.model import MVSNetWeights, inference_winner_take_all
    w = S.make_workload("toy", network_mode=mode)
    gp = S.make_gru_params(mode, seed=2, in_channels=w.channels, random_affine=True)
    weights = MVSNetWeights.from_numpy(mode, gru=gp, device=DEV)
    depth, prob = inference_winner_take_all(None, t(w.cams)[None], w.depth_num, w.depth_start, w.depth_end,
                                            inverse_depth=inverse, weights=weights, features=t(w.features))
    ed, ep = O.inference_winner_take_all_from_features(w.features, w.cams, w.depth_num, w.depth_start,
                                                       w.depth_end, gp, inverse, np.float64)
    depth, prob = n(depth)[0, :, :, 0], n(prob)[0, :, :, 0]
    # the arg-max plane may flip where two planes tie within rounding; everything else is exact
    # (depth values are fp32 on the device, fp64 in the oracle: compare to 1e-6 relative)
    same = np.abs(depth - ed) <= 1e-6 * np.abs(ed)
    assert same.mean() > 0.98
    np.testing.assert_allclose(prob[same], ep[same], rtol=2e-4)


@pytest.mark.parametrize("hw", [(27, 41), (9, 47), (31, 17)])
def test_gru_wta_ragged_image_sizes_match_oracle(hw):
    """Image sizes that are not multiples of the fused sweep's 8 x 16 pixel tiles (gru_fused.hip): partial tiles on the right and
    bottom edges, a single tile row, a tile column narrower than a tile -- masked lanes read zeros and drop their stores, the
    LayerNorm sums count own pixels only.  Same cameras, cropped feature maps, against the float64 oracle (convgru.py:82-122)."""
    from mvsnet_amd.model import MVSNetWeights, inference_winner_take_all
    w = S.make_workload("small")
    H, Wd = hw
    feats = np.ascontiguousarray(w.features[:, :H, :Wd])
    gp = S.make_gru_params("normal", seed=7, in_channels=w.channels, random_affine=True)
    weights = MVSNetWeights.from_numpy("normal", gru=gp, device=DEV)
    depth, prob = inference_winner_take_all(None, t(w.cams)[None], w.depth_num, w.depth_start, w.depth_end,
                                            weights=weights, features=t(feats))
    ed, ep = O.inference_winner_take_all_from_features(feats, w.cams, w.depth_num, w.depth_start, w.depth_end, gp, False, np.float64)
    depth, prob = n(depth)[0, :, :, 0], n(prob)[0, :, :, 0]
    assert depth.shape == (H, Wd)
    same = np.abs(depth - ed) <= 1e-6 * np.abs(ed)
    assert same.mean() > 0.97, same.mean()
    np.testing.assert_allclose(prob[same], ep[same], rtol=5e-4)
    # the wavefront kernels of rounds 3-4 on the same input: the same planes up to ties
    import ctypes as C
    from mvsnet_amd import _lib as L
    lib = L.load()
    L.check(lib.mvs_gru_set_formulation(1), "mvs_gru_set_formulation")
    try:
        d1, _p1 = inference_winner_take_all(None, t(w.cams)[None], w.depth_num, w.depth_start, w.depth_end,
                                            weights=weights, features=t(feats))
    finally:
        L.check(lib.mvs_gru_set_formulation(0), "mvs_gru_set_formulation")
    assert (n(d1)[0, :, :, 0] == depth).mean() > 0.97


# ---- R10 end to end -------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,inverse,mode", [("toy", False, "normal"), ("toy", True, "normal"), ("small", False, "normal"),
                                               ("toy", False, "lite"), ("small", False, "semilite-py3")])
def test_inference_mem_from_features_matches_oracle(name, inverse, mode):
    from mvsnet_amd.model import MVSNetWeights, inference_mem
    w = S.make_workload(name, network_mode=mode)
    rp = S.make_regnet_params(mode, seed=1, random_affine=True)
    weights = MVSNetWeights.from_numpy(mode, regnet=rp, device=DEV)
    depth, prob = inference_mem(None, t(w.cams)[None], w.depth_num, w.depth_start, w.depth_interval,
                                inverse_depth=inverse, weights=weights, features=t(w.features))
    ed, ep = O.inference_mem_from_features(w.features, w.cams, w.depth_num, w.depth_start,
                                           w.depth_interval, rp, inverse, np.float64)
    depth, prob = n(depth)[0, :, :, 0], n(prob)[0, :, :, 0]
    abs_rel = float(np.mean(np.abs(depth - ed) / ed))
    assert abs_rel < 1e-4, abs_rel              # north-star bar: 1e-3
    assert (np.abs(prob - ep) > 1e-3).mean() < 0.02


def test_feature_extractor_matches_oracle():
    from mvsnet_amd.feature_net import UNetDS2GN
    params = S.make_unet_params("normal", seed=3)
    imgs = S.make_images(2, 32, 48, seed=0)
    got = n(UNetDS2GN(params, DEV)(t(imgs)))
    assert got.shape == (2, 8, 12, 32)
    for v in range(2):
        exp = O.unet_ds2gn(imgs[v], params, np.float64)
        np.testing.assert_allclose(got[v], exp, rtol=2e-3, atol=2e-4)


def test_inference_mem_from_images_runs_and_writes_outputs(tmp_path):
    from mvsnet_amd.model import MVSNetWeights, inference_mem
    from mvsnet_amd import preprocess as P
    up = S.make_unet_params("normal", seed=3)
    rp = S.make_regnet_params("normal", seed=1)
    weights = MVSNetWeights.from_numpy("normal", unet=up, regnet=rp, device=DEV)
    imgs = S.make_images(3, 64, 64, seed=1)
    cams = S.make_cams(3, 16, 16, 8, interval=60.0)
    depth, prob = inference_mem(t(imgs)[None], t(cams)[None], 8, cams[0, 1, 3, 0], cams[0, 1, 3, 1],
                                weights=weights)
    assert depth.shape == (1, 16, 16, 1) and prob.shape == (1, 16, 16, 1)
    d = n(depth)[0, :, :, 0]
    assert np.isfinite(d).all() and d.min() >= 425.0 and d.max() <= 425.0 + 7 * 60.0
    path = str(tmp_path / "0_init.pfm")
    P.write_pfm(path, d)
    assert np.array_equal(P.load_pfm(path), d)


def test_device_matches_committed_golden_fixtures():
    """tests/golden/toy_*.npz (fp64 oracle outputs, generator committed beside them)."""
    import os
    from mvsnet_amd.model import MVSNetWeights, inference_mem, inference_winner_take_all
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    w = S.make_workload("toy")
    rp = S.make_regnet_params("normal", seed=1, random_affine=True)
    gp = S.make_gru_params("normal", seed=2, in_channels=w.channels, random_affine=True)
    weights = MVSNetWeights.from_numpy("normal", regnet=rp, gru=gp, device=DEV)
    g = np.load(os.path.join(gdir, "toy_3dcnn.npz"))
    depth, prob = inference_mem(None, t(w.cams)[None], w.depth_num, w.depth_start, w.depth_interval,
                                weights=weights, features=t(w.features))
    d = n(depth)[0, :, :, 0]
    assert float(np.mean(np.abs(d - g["depth"]) / g["depth"])) < 1e-4
    g = np.load(os.path.join(gdir, "toy_gru.npz"))
    depth, prob = inference_winner_take_all(None, t(w.cams)[None], w.depth_num, w.depth_start, w.depth_end,
                                            weights=weights, features=t(w.features))
    d = n(depth)[0, :, :, 0]
    assert (np.abs(d - g["depth"]) <= 1e-6 * g["depth"]).mean() > 0.98


def test_gru_batch_entry_point_argument_checks_and_ragged_batches():
    """mvs_gru_wta_batch_f32: 1..8 views, a workspace block per view, per-view depth values; errors are return codes (no fault).
    A ragged volume (tiles cut on both axes, planes not a multiple of the group / batch sizes) with 8 views against the oracle."""
    import ctypes as C
    from mvsnet_amd import _lib
    from mvsnet_amd.model import DepthPlan, MVSNetWeights, wta_depth_values
    w = S.make_workload("small")
    D, Hh, Ww = 19, 21, 37
    gp = S.make_gru_params("normal", seed=2, in_channels=w.channels, random_affine=True)
    weights = MVSNetWeights.from_numpy("normal", gru=gp, device=DEV)
    nv = 8
    plan = DepthPlan(w.view_num, D, Hh, Ww, w.channels, weights, "GRU", DEV, views=nv)
    feats, dvs, ends = [], [], []
    for v in range(nv):
        f = S.make_features(w.view_num, Hh, Ww, w.channels, seed=30 + v)
        feats.append(f)
        ends.append(w.depth_start + (D - 1) * w.depth_interval * (1.0 + 0.1 * v))
        plan.set_cameras(t(w.cams), w.depth_start, (ends[v] - w.depth_start) / (D - 1), ends[v], False, view=v)
        dvs.append(wta_depth_values(D, w.depth_start, ends[v], False))
    d, p = plan.run_gru_batch([t(f) for f in feats], dvs)
    d, p = n(d), n(p)
    for v in (0, 3, 7):
        ed, ep = O.inference_winner_take_all_from_features(feats[v], w.cams, D, w.depth_start, ends[v], gp, False, np.float64)
        same = d[v] == ed.astype(np.float32)
        assert same.mean() > 0.97, (v, float(same.mean()))
        assert np.abs(p[v][same] - ep[same]).max() < 2e-4
    # argument checks
    lib = _lib.load()
    g = weights.gru
    f1, f2, f3 = g.filters
    keep = [(t(f)[0].contiguous(), t(f)[1:].contiguous()) for f in feats]
    ref = _lib.ptr_array([k[0] for k in keep]); src = _lib.ptr_array([k[1] for k in keep])
    tr = _lib.ptr_array([plan.transforms_v[v] for v in range(nv)])
    flat = (C.c_float * (nv * D))(*[float(x) for dv in dvs for x in dv])
    call = lambda views, ws_bytes: lib.mvs_gru_wta_batch_f32(ref, src, tr, views, w.view_num, D, Hh, Ww, w.channels, f1, f2, f3, g.ptrs, flat,
                                                             C.c_void_p(plan.workspace.data_ptr()), ws_bytes, _lib.ptr(plan.depth_v),
                                                             _lib.ptr(plan.prob_v), _lib.stream_ptr())
    assert call(9, plan.workspace.numel()) == -1 and call(0, plan.workspace.numel()) == -1        # MVS_E_BADARG
    assert call(nv, plan.workspace.numel() // nv * (nv - 1)) == -3                                   # MVS_E_WORKSPACE
    assert lib.mvs_gru_set_formulation(4) == -1
    torch.cuda.synchronize()
