"""LAB variants (opt-in: `pytest -m lab` on a GPU box): kernel families that were built, are exact and measured no faster than
the product's, and therefore live OUTSIDE libmvsnet_hip.so -- csrc/lab/, built by `python -m mvsnet_amd.build --lab` into
mvsnet_amd/variants/libmvsnet_lab.so with entry points of their own (mvs_lab_*).  Each is held to the product kernel
(mvs_cost_volume_f32, itself held to the oracle by tests/test_gpu_parity.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

from mvsnet_amd import synthetic as S

pytestmark = [pytest.mark.lab, pytest.mark.skipif(not torch.cuda.is_available(), reason="lab variants run on the GPU")]


@pytest.fixture(scope="module")
def lab():
    from mvsnet_amd.build import build_lab
    lib = C.CDLL(build_lab(verbose=False))
    p, i = C.c_void_p, C.c_int
    lib.mvs_lab_cost_volume_lds_f32.argtypes = [p, p, p] + [i] * 9 + [p, p]
    lib.mvs_lab_cost_volume_mfma_f32.argtypes = [p, p, p] + [i] * 10 + [p, p]
    lib.mvs_lab_cost_volume_fallback_rounds.argtypes = [C.POINTER(C.c_int)]
    return lib


def cases():
    """(features, transforms, d_begin, d_count, variant, negate): the metric workload, the nearest 32 planes of c3 (a sample
    point moves 1.4 pixels per plane: footprints beyond the LDS budget, windows beyond the unrolled shapes), ragged cases."""
    from mvsnet_amd.homography_warping import homography_transforms
    out = []
    for name in ("M", "c3"):
        w = S.make_workload(name)
        f = torch.as_tensor(w.features).cuda()
        T8 = homography_transforms(torch.as_tensor(w.cams).cuda(), w.depth_num, w.depth_start, w.depth_interval)
        out.append((name, f, T8, 0, w.depth_num if name == "M" else 32, "mem", False))
    k = 0
    for (N, H, W, D, d0, dn, variant, neg, interval) in [(2, 9, 21, 5, 0, 5, "mem", False, 40.0), (3, 17, 33, 13, 2, 9, "eager", True, 25.0),
                                                         (5, 30, 50, 19, 0, 19, "mem", False, 90.0), (7, 12, 70, 10, 3, 6, "eager", False, 15.0),
                                                         (4, 8, 16, 8, 0, 8, "mem", False, 300.0), (6, 5, 130, 7, 1, 5, "mem", True, 60.0)]:
        f = torch.as_tensor(S.make_features(N, H, W, 32, seed=11 + k)).cuda()
        cams = S.make_cams(N, H, W, D, interval=interval)
        T8 = homography_transforms(torch.as_tensor(cams).cuda(), D, float(cams[0, 1, 3, 0]), float(cams[0, 1, 3, 1]))
        out.append(("r%d" % k, f, T8, d0, dn, variant, neg))
        k += 1
    return out


def run_lab(fn, f, T8, d0, dn, variant, neg, *extra):
    N, H, W, Cc = f.shape
    out = torch.empty((dn, H, W, Cc), device=f.device, dtype=torch.float32)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    src = f[1:].contiguous()
    rc = fn(C.c_void_p(f[0].data_ptr()), C.c_void_p(src.data_ptr()), C.c_void_p(T8.data_ptr()), N, T8.shape[1], d0, dn, H, W, Cc,
            0 if variant == "mem" else 1, int(neg), *extra, C.c_void_p(out.data_ptr()), stream)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out


def close(a, b):
    bad = (a - b).abs() > 1e-5 + 1e-4 * a.abs()
    return float(bad.float().mean()) < 1e-3          # rcp vs division: a flipped floor() moves a whole tap on rare samples


def test_lds_staged_cost_volume_matches_the_product_kernel(lab):
    from mvsnet_amd.model import cost_volume
    r = C.c_int(0)
    for name, f, T8, d0, dn, variant, neg in cases():
        if f.shape[0] > 8:
            continue
        ref = cost_volume(f[0], f[1:], T8, d0, dn, variant, negate=neg)
        lab.mvs_lab_cost_volume_fallback_rounds(C.byref(r))
        got = run_lab(lab.mvs_lab_cost_volume_lds_f32, f, T8, d0, dn, variant, neg)
        lab.mvs_lab_cost_volume_fallback_rounds(C.byref(r))
        assert close(ref, got), name
        if name == "M":
            assert r.value == 0                       # every footprint fits the LDS budget: all rounds staged
        if name == "c3":
            assert r.value > 0                        # the near planes exercise the exact direct path


@pytest.mark.parametrize("direct", [0, 1])
def test_mfma_blend_cost_volume_matches_the_product_kernel(lab, direct):
    from mvsnet_amd.model import cost_volume
    for name, f, T8, d0, dn, variant, neg in cases():
        ref = cost_volume(f[0], f[1:], T8, d0, dn, variant, negate=neg)
        got = run_lab(lab.mvs_lab_cost_volume_mfma_f32, f, T8, d0, dn, variant, neg, direct)
        assert torch.isfinite(got).all() and close(ref, got), (name, direct)
