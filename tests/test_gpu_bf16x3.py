"""Opt-in split-precision (bf16x3) regulariser: three bf16 MFMAs per fp32 product.  It must stay
inside the north-star tolerance (1e-3 relative L1 on depth); the default path is exact fp32."""
import numpy as np
import pytest
import torch

from oracle import mvsnet_oracle as O
from mvsnet_amd import synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda"


def t(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(DEV)


def test_regnet_bf16x3_close_to_oracle_and_to_fp32(lib_built):
    from mvsnet_amd import _lib as L
    from mvsnet_amd.model import RegNetWeights, regnet_us0
    D, H, W = 16, 16, 32
    params = S.make_regnet_params("normal", seed=11, random_affine=True)
    rs = np.random.RandomState(12)
    cost = np.abs(rs.standard_normal((D, H, W, 32))).astype(np.float32)
    wts = RegNetWeights(params, DEV)
    exact = regnet_us0(t(cost), wts).cpu().numpy()
    L.set_conv_impl("bf16x3")
    try:
        split = regnet_us0(t(cost), wts).cpu().numpy()
    finally:
        L.set_conv_impl("auto")
    ref = O.regnet_us0(cost, params, np.float64)
    rel = lambda a, b: float(np.abs(a - b).sum() / np.abs(b).sum())
    assert rel(exact, ref) < 2e-5
    assert not np.array_equal(split, exact)            # the split path really ran
    assert rel(split, ref) < 1e-3, rel(split, ref)


@pytest.mark.parametrize("name", ["small"])
def test_inference_mem_bf16x3_within_north_star_tolerance(lib_built, name):
    from mvsnet_amd import _lib as L
    from mvsnet_amd.model import MVSNetWeights, inference_mem
    w = S.make_workload(name)
    rp = S.make_regnet_params("normal", seed=1, random_affine=True)
    weights = MVSNetWeights.from_numpy("normal", regnet=rp, device=DEV)
    L.set_conv_impl("bf16x3")
    try:
        depth, prob = inference_mem(None, t(w.cams)[None], w.depth_num, w.depth_start, w.depth_interval,
                                    weights=weights, features=t(w.features))
        depth = depth.cpu().numpy()[0, :, :, 0]
    finally:
        L.set_conv_impl("auto")
    ed, ep = O.inference_mem_from_features(w.features, w.cams, w.depth_num, w.depth_start,
                                           w.depth_interval, rp, False, np.float64)
    abs_rel = float(np.mean(np.abs(depth - ed) / ed))
    assert abs_rel < 1e-3, abs_rel
