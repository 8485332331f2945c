"""The multi-threaded torch-CPU restatement (bench.py's cpu_baseline) against the strict numpy
oracle, and the committed golden fixture against both."""
import os

import numpy as np
import pytest

from oracle import mvsnet_oracle as O
from oracle import torch_restatement as TR
from mvsnet_amd import synthetic as S

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["toy", "small"])
def test_torch_restatement_matches_numpy_oracle(name):
    w = S.make_workload(name)
    rp = S.make_regnet_params("normal", seed=1, random_affine=True)
    d_t, p_t = TR.inference_mem_from_features(w.features, w.cams, w.depth_num, w.depth_start,
                                              w.depth_interval, rp)
    d_o, p_o = O.inference_mem_from_features(w.features, w.cams, w.depth_num, w.depth_start,
                                             w.depth_interval, rp, False, np.float64)
    assert float(np.mean(np.abs(d_t - d_o) / d_o)) < 1e-4
    assert (np.abs(p_t - p_o) > 1e-3).mean() < 0.02


@pytest.mark.parametrize("name", ["toy", "small"])
def test_float64_torch_restatement_is_the_numpy_oracle(name):
    """The generator of the FULL-SIZE fixtures (tests/golden/make_golden.py --full) is oracle/torch_restatement.py in
    float64; at the sizes the strict numpy oracle finishes in seconds the two must be the same function, 3D-CNN path and
    recurrent path (winning plane identical everywhere, probabilities to rounding)."""
    import torch
    w = S.make_workload(name)
    rp = S.make_regnet_params("normal", seed=1, random_affine=True)
    gp = S.make_gru_params("normal", seed=2, in_channels=w.channels, random_affine=True)
    d_t, p_t = TR.inference_mem_from_features(w.features, w.cams, w.depth_num, w.depth_start, w.depth_interval, rp, torch.float64)
    d_o, p_o = O.inference_mem_from_features(w.features, w.cams, w.depth_num, w.depth_start, w.depth_interval, rp, False, np.float64)
    np.testing.assert_allclose(d_t, d_o, rtol=1e-11)
    np.testing.assert_allclose(p_t, p_o, rtol=1e-9, atol=1e-12)
    d_t, p_t, idx = TR.inference_winner_take_all_from_features(w.features, w.cams, w.depth_num, w.depth_start, w.depth_end, gp, torch.float64)
    d_o, p_o = O.inference_winner_take_all_from_features(w.features, w.cams, w.depth_num, w.depth_start, w.depth_end, gp, False, np.float64)
    assert np.array_equal(d_t, d_o)
    np.testing.assert_allclose(p_t, p_o, rtol=1e-9)
    depths = O.wta_depths(w.depth_num, w.depth_start, w.depth_end, False, np.float64)
    assert np.array_equal(depths[idx], d_o)


def test_full_size_fixtures_are_committed_with_their_input_digests():
    """tests/golden/full_<workload>.npz: shapes, digests of the regenerated inputs, and a rounding-noise floor that makes sense."""
    import hashlib
    for name in ("M", "c1", "c2", "c3"):
        g = np.load(os.path.join(GOLDEN, "full_%s.npz" % name))
        w = S.make_workload(name)
        assert hashlib.sha256(w.features.tobytes() + w.cams.tobytes()).hexdigest() == str(g["input_sha256"])
        assert g["depth"].shape == (w.height, w.width) and g["prob"].shape == (w.height, w.width)
        assert g["depth"].min() >= w.depth_start - 1e-3 and g["depth"].max() <= w.depth_end + 1e-3
        if name == "c3":
            assert g["index"].dtype == np.uint8 and float(g["f32_cpu_plane_agreement"]) > 0.999
        else:
            assert float(g["f32_cpu_abs_rel"]) < 5e-6


def test_fp32_oracle_close_to_fp64_oracle():
    w = S.make_workload("toy")
    rp = S.make_regnet_params("normal", seed=1, random_affine=True)
    a = O.inference_mem_from_features(w.features, w.cams, w.depth_num, w.depth_start, w.depth_interval, rp, False, np.float32)
    b = O.inference_mem_from_features(w.features, w.cams, w.depth_num, w.depth_start, w.depth_interval, rp, False, np.float64)
    assert float(np.mean(np.abs(a[0] - b[0]) / b[0])) < 1e-5


def test_golden_fixture_reproduces():
    """tests/golden/toy_*.npz were written by tests/golden/make_golden.py from the fp64 oracle; the
    inputs are regenerated from their seeds and must hash to the recorded digests."""
    import hashlib
    for tag in ("toy_3dcnn", "toy_gru"):
        g = np.load(os.path.join(GOLDEN, tag + ".npz"))
        w = S.make_workload("toy")
        assert hashlib.sha256(w.features.tobytes() + w.cams.tobytes()).hexdigest() == str(g["input_sha256"])
        if tag == "toy_3dcnn":
            rp = S.make_regnet_params("normal", seed=1, random_affine=True)
            d, p = O.inference_mem_from_features(w.features, w.cams, w.depth_num, w.depth_start,
                                                 w.depth_interval, rp, False, np.float64)
        else:
            gp = S.make_gru_params("normal", seed=2, in_channels=w.channels, random_affine=True)
            d, p = O.inference_winner_take_all_from_features(w.features, w.cams, w.depth_num, w.depth_start,
                                                             w.depth_end, gp, False, np.float64)
        np.testing.assert_allclose(d, g["depth"], rtol=1e-9)
        np.testing.assert_allclose(p, g["prob"], rtol=1e-7, atol=1e-9)


def test_gradient_golden_fixture_reproduces():
    """tests/golden/toy_grad.npz: float64 autograd of the torch restatement (oracle/torch_grad.py), the checker of the
    training backward."""
    import torch
    from oracle import torch_grad as TG
    g = np.load(os.path.join(GOLDEN, "toy_grad.npz"))
    w = S.make_workload("toy")
    rp = S.make_regnet_params("normal", seed=1, random_affine=True)
    d64 = lambda a, req=False: torch.tensor(np.asarray(a, np.float64)).requires_grad_(req)
    f64 = d64(w.features[:3], True)
    p64 = {k: {kk: d64(vv, True) for kk, vv in v.items()} for k, v in rp.items()}
    depth = TG.depth_from_features(f64, d64(g["t8"]), w.depth_start, w.depth_interval, p64)
    (depth * d64(g["g"])).sum().backward()
    np.testing.assert_allclose(depth.detach().numpy(), g["depth"], rtol=1e-10)
    np.testing.assert_allclose(f64.grad.numpy(), g["g_features"], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(p64["3dconv0_1"]["w"].grad.numpy(), g["g_w01"], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(p64["3dconv6_2"]["w"].grad.numpy(), g["g_w62"], rtol=1e-7, atol=1e-10)
