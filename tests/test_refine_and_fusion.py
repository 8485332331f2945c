"""CPU tests of the post-processing rows: depth refinement towers against the numpy
oracle (the towers are PyTorch host glue, so torch-CPU is the product path here), the TF1
resize_bilinear KATs, and the Gipuma hand-off files."""
import os
import struct

import numpy as np
import torch

from oracle import mvsnet_oracle as O


def test_resize_bilinear_tf1_kats():
    from mvsnet_amd.refine import resize_bilinear_tf1
    ramp = np.arange(4, dtype=np.float32)[None, :, None].repeat(2, 0)            # value = x
    up = O.resize_bilinear_tf1(ramp, 2, 8)
    # legacy mapping src = dst * in/out: no half-pixel shift, last sample clamps to the edge
    np.testing.assert_allclose(up[0, :, 0], [0, 0.5, 1, 1.5, 2, 2.5, 3, 3])
    down = O.resize_bilinear_tf1(np.arange(8, dtype=np.float32)[None, :, None], 1, 4)
    np.testing.assert_allclose(down[0, :, 0], [0, 2, 4, 6])                       # picks src = 2*dst exactly
    frac = O.resize_bilinear_tf1(np.arange(5, dtype=np.float32)[:, None, None], 3, 1)
    np.testing.assert_allclose(frac[:, 0, 0], [0, 5 / 3, 10 / 3], rtol=1e-6)
    rs = np.random.RandomState(0)
    x = rs.standard_normal((2, 7, 9, 3)).astype(np.float32)
    got = resize_bilinear_tf1(torch.as_tensor(x), 12, 5).numpy()
    for b in range(2):
        np.testing.assert_allclose(got[b], O.resize_bilinear_tf1(x[b], 12, 5), rtol=1e-6, atol=1e-6)
    assert O.resize_bilinear_tf1(x[0], 7, 9) is not None and resize_bilinear_tf1(torch.as_tensor(x), 7, 9).shape == x.shape


def test_refine_tables_match_reference_topology():
    from mvsnet_amd import refine as R
    assert [l[:6] for l in R.REFINE_ORIGINAL] == [l[:6] for l in O.REFINE_ORIGINAL]
    # the oracle builds the U-Net table procedurally, the product writes it out: same layers, any order
    assert sorted(R.REFINE_UNET) == sorted(O.REFINE_UNET) and len(R.REFINE_UNET) == 28


def _case(network_type, conf, upsample, seed, stereo=False, device="cpu"):
    from mvsnet_amd import refine as R
    rs = np.random.RandomState(seed)
    h, w, H, W = 16, 16, 32, 32                         # unet: 4 stride-2 levels need /16 at the tower's resolution
    depth = (425 + 500 * rs.rand(1, h, w, 1)).astype(np.float32)
    prob = rs.rand(1, h, w, 1).astype(np.float32)
    image = rs.standard_normal((1, H, W, 3)).astype(np.float32)
    partner = rs.standard_normal((1, H, W, 3)).astype(np.float32) if stereo else None
    params = R.make_refine_params(network_type, "normal", 3 + 1 + int(conf) + (3 if stereo else 0), seed)
    net = R.RefineNet(params, network_type, device)
    dv = lambda a: torch.as_tensor(a).to(device)
    got, got_res = R.depth_refine(dv(depth), dv(image), dv(prob), 192, 425.0, 2.65,
                                  net, upsample_depth=upsample, refine_with_confidence=conf,
                                  stereo_image=dv(partner) if stereo else None)
    got, got_res = got.cpu(), got_res.cpu()
    exp, exp_res = O.depth_refine(depth[0], image[0], prob[0], 192, 425.0, 2.65, params, network_type,
                                  upsample_depth=upsample, refine_with_confidence=conf, dtype=np.float64,
                                  stereo_image=partner[0] if stereo else None)
    assert got.shape == (1,) + exp.shape
    np.testing.assert_allclose(got_res[0].numpy(), exp_res, rtol=2e-4, atol=2e-3 * np.abs(exp_res).max())
    np.testing.assert_allclose(got[0].numpy(), exp, rtol=2e-4, atol=2e-3 * np.abs(exp_res).max())


def test_depth_refine_original_matches_oracle():
    _case("original", False, False, 1)
    _case("original", True, True, 2)


def test_depth_refine_unet_matches_oracle():
    _case("unet", True, False, 3)
    _case("unet", False, True, 4)


def test_depth_refine_with_stereo_partner_matches_oracle():
    """model.py:777-789: the stereo partner is resized with the image (no up-sampling) or used at full size."""
    _case("original", True, False, 5, stereo=True)
    _case("unet", True, True, 6, stereo=True)


def test_probability_filter_and_gipuma_export(tmp_path):
    from mvsnet_amd import depthfusion as F, preprocess as pp
    dense = str(tmp_path / "dense")
    out = os.path.join(dense, "depths_mvsnet")
    os.makedirs(out)
    rs = np.random.RandomState(0)
    from PIL import Image
    for idx in (0, 3):
        depth = (400 + 100 * rs.rand(6, 8)).astype(np.float32)
        prob = rs.rand(6, 8).astype(np.float32)
        pp.write_pfm(os.path.join(out, "%d_init.pfm" % idx), depth)
        pp.write_pfm(os.path.join(out, "%d_prob.pfm" % idx), prob)
        cam = np.zeros((2, 4, 4)); cam[0] = np.eye(4); cam[0, 0, 3] = 10.0 * idx
        cam[1, :3, :3] = [[50, 0, 4], [0, 50, 3], [0, 0, 1]]; cam[1, 3] = [425, 2.5, 192, 902.5]
        pp.write_cam(os.path.join(out, "%d.txt" % idx), cam)
        Image.fromarray(rs.randint(0, 255, (6, 8, 3)).astype(np.uint8)).save(os.path.join(out, "%d.jpg" % idx))
    F.probability_filter(dense, 0.5)
    filt = pp.load_pfm(os.path.join(out, "3_prob_filtered.pfm"))
    d3, p3 = pp.load_pfm(os.path.join(out, "3_init.pfm")), pp.load_pfm(os.path.join(out, "3_prob.pfm"))
    assert np.array_equal(filt, np.where(p3 < 0.5, 0, d3)) and (filt == 0).any() and (filt > 0).any()
    points = os.path.join(dense, "points_mvsnet")
    assert F.mvsnet_to_gipuma(dense, points) == ["0.jpg", "3.jpg"]
    # .dmb: int32 header (1, H, W, C) + float32 payload; round trip through the reader
    raw = open(os.path.join(points, "2333__3", "disp.dmb"), "rb").read()
    assert struct.unpack("<iiii", raw[:16]) == (1, 6, 8, 1) and len(raw) == 16 + 6 * 8 * 4
    assert np.array_equal(np.frombuffer(raw[16:], np.float32).reshape(6, 8), filt)
    normals = F.read_gipuma_dmb(os.path.join(points, "2333__3", "normals.dmb"))
    assert struct.unpack("<iiii", open(os.path.join(points, "2333__3", "normals.dmb"), "rb").read(16)) == (1, 6, 8, 3)
    assert normals.size == 6 * 8 * 3 and set(np.unique(np.round(normals, 6))) <= {0.0, np.float32(round(1 / 1.732050808, 6))}
    P = np.loadtxt(os.path.join(points, "cams", "3.jpg.P"))
    np.testing.assert_allclose(P, [[50, 0, 4, 1500], [0, 50, 3, 0], [0, 0, 1, 0]])
    assert os.path.isfile(os.path.join(points, "images", "0.jpg"))
    cmd = F.fusibile_command(points, "/opt/fusibile", 0.25, 3)
    assert cmd[0] == "/opt/fusibile" and "--disp_thresh=0.25" in cmd and cmd[2] == points + "/"
    assert F.depth_map_fusion(points, "/nonexistent/fusibile", 0.25, 3) is None
