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
