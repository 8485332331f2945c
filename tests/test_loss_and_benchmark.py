"""CPU tests of the benchmark caller (mvsnet/test.py): forward losses against hand-computed values
and the numpy oracle, ground-truth depth loading of the 'test' generator mode, results CSV."""
import os

import numpy as np
import torch

from oracle import mvsnet_oracle as O
from tests.test_data_and_sharding import make_session


def test_regression_loss_kats():
    from mvsnet_amd import loss as L
    gt = torch.tensor([[[[10.], [0.]], [[20.], [30.]]]])          # (1,2,2,1); 0 = invalid
    pr = torch.tensor([[[[11.], [5.]], [[26.], [30.]]]])
    loss, l1, l3, dbg = L.mvsnet_regression_loss(pr, gt, [0.0], [382.0], grad_loss=False)
    # interval = 382/191 = 2; errors on valid pixels 1, 6, 0 -> MAE/interval = (7/3)/2
    assert abs(float(loss) - 7.0 / 6.0) < 1e-5 and dbg is None
    assert abs(float(l1) - 2.0 / 3.0) < 1e-5 and abs(float(l3) - 1.0) < 1e-5      # |e|/2 <= 1: {0.5, 3, 0}
    # power loss with alpha = 1, beta = 0 reduces to 10 * mean(|e| + 0.005 y) / interval
    pw = L.power_loss(gt, pr, torch.tensor([2.0]), 1.0, 0.0)
    assert abs(float(pw) - 10.0 * ((1 + 6 + 0) + 0.005 * 60) / 3.0 / 2.0) < 1e-4
    g = L.gaussian_loss(gt, gt, torch.tensor([2.0]), 0.02)                          # perfect prediction: -N_all / N_valid
    assert abs(float(g) + 4.0 / 3.0) < 1e-5


def test_losses_match_oracle():
    from mvsnet_amd import loss as L
    rs = np.random.RandomState(0)
    for B in (1, 3, 4):
        gt = (400 + 500 * rs.rand(B, 9, 11, 1)).astype(np.float32)
        gt[rs.rand(*gt.shape) < 0.2] = 0
        pr = (gt + 8 * rs.standard_normal(gt.shape)).astype(np.float32)
        ds, de = (425.0 + rs.rand(B)).astype(np.float32), (900.0 + rs.rand(B)).astype(np.float32)
        got = L.mvsnet_regression_loss(torch.as_tensor(pr), torch.as_tensor(gt), ds, de)
        exp = O.regression_metrics(pr, gt, ds, de)
        for a, b in zip(got, exp):
            assert abs(float(a) - b) <= 1e-4 * max(1.0, abs(b)), (B, float(a), b)


def _session_with_depth(path):
    from PIL import Image
    sess = make_session(path, n_images=4, h=96, w=128)
    os.makedirs(os.path.join(sess, "depths"))
    rs = np.random.RandomState(1)
    for i in range(4):
        d = (300 + 700 * rs.rand(48, 64)).astype(np.uint16)       # half resolution, some values outside [400, 900]
        Image.fromarray(d).save(os.path.join(sess, "depths", "%d.png" % i))
    return sess


def test_generator_test_mode_returns_masked_depth(tmp_path):
    from mvsnet_amd.mvs_data_generation import ClusterGenerator, mask_depth_image
    m = mask_depth_image(np.array([[399, 400, 401], [899, 900, 901]], np.uint16), 400, 900)
    assert m.shape == (2, 3, 1) and m[:, :, 0].tolist() == [[0, 0, 401], [899, 900, 0]]     # (min, max] kept
    sess = _session_with_depth(str(tmp_path / "s"))
    gen = ClusterGenerator(sess, view_num=3, image_width=64, image_height=64, depth_num=8, base_image_size=8,
                           mode="test", output_scale=0.25)
    out_images, in_images, out_cams, full_cams, idx, depth = gen.prepare(gen.clusters[0])
    assert depth.shape == (64, 64, 1) and depth.dtype == np.float32 and in_images.shape == (3, 64, 64, 3)
    vals = depth[depth > 0]
    assert vals.size and vals.min() > 400 and vals.max() <= 900 and (depth == 0).any()
    # nearest-neighbour path: every non-zero value is one of the stored PNG values
    from PIL import Image
    src = np.asarray(Image.open(os.path.join(sess, "depths", "0.png")))
    assert set(np.unique(vals).astype(int)) <= set(np.unique(src).astype(int))


def test_write_results_header_once(tmp_path):
    from mvsnet_amd.test import write_results, RESULTS_HEADER
    p = str(tmp_path / "results.csv")
    write_results(p, "models/x", 400000, 1.5, 0.8, 0.95, 0.1)
    write_results(p, "models/x", 400000, 1.4, 0.81, 0.96, 0.1)
    lines = open(p).readlines()
    assert lines[0] == RESULTS_HEADER and len(lines) == 3 and lines[2].startswith("models/x, 400000, 1.4, 0.81")
