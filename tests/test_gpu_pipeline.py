"""GPU tests at the outer boundary: session directory -> .pfm outputs, full-size properties of
the metric workload, plan re-use under hipGraph capture."""
import os

import numpy as np
import pytest
import torch

from mvsnet_amd import synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda"


def t(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(DEV)


@pytest.mark.parametrize("regularization", ["3DCNN", "GRU"])
def test_compute_depth_maps_on_session_dir(tmp_path, lib_built, regularization):
    from tests.test_data_and_sharding import make_session
    from mvsnet_amd.inference import compute_depth_maps
    from mvsnet_amd import predictlib as pl, preprocess as pp
    sess = make_session(str(tmp_path / "sess"), n_images=4, h=96, w=128)
    cfg = pl.InferenceConfig(view_num=3, max_d=8, width=64, height=64, base_image_size=8,
                             regularization=regularization, max_clusters_per_session=2)
    n = compute_depth_maps(sess, cfg)
    assert n == 2
    out = os.path.join(sess, "depths_mvsnet")
    for idx in (0, 1):
        d = pp.load_pfm(os.path.join(out, "%d_init.pfm" % idx))
        p = pp.load_pfm(os.path.join(out, "%d_prob.pfm" % idx))
        assert d.shape == (16, 16) and p.shape == (16, 16)
        assert np.isfinite(d).all() and d.min() >= 400.0 - 1e-3 and d.max() <= 900.0 + 1e-3
        assert np.isfinite(p).all() and p.min() >= 0.0
        for suffix in ("_depth.png", "_prob.png", ".jpg", ".txt"):
            assert os.path.exists(os.path.join(out, "%d%s" % (idx, suffix)))
        cam = pp.load_cam(os.path.join(out, "%d.txt" % idx))
        assert cam[1, 3, 2] == 8


@pytest.mark.parametrize("network,upsample", [("original", False), ("unet", True)])
def test_compute_depth_maps_with_refinement(tmp_path, lib_built, network, upsample):
    """--refinement: the refined depth lands in <idx>_init.pfm; with
    --upsample_before_refinement the outputs are written at input resolution (predictlib.py:107-115)."""
    from tests.test_data_and_sharding import make_session
    from mvsnet_amd.inference import compute_depth_maps
    from mvsnet_amd import predictlib as pl, preprocess as pp
    sess = make_session(str(tmp_path / "sess"), n_images=4, h=96, w=128)
    cfg = pl.InferenceConfig(view_num=3, max_d=8, width=64, height=64, base_image_size=8, refinement=True,
                             refinement_network=network, upsample_before_refinement=upsample,
                             refine_with_confidence=upsample, max_clusters_per_session=1)
    assert compute_depth_maps(sess, cfg) == 1
    out = os.path.join(sess, "depths_mvsnet")
    size = (64, 64) if upsample else (16, 16)
    d = pp.load_pfm(os.path.join(out, "0_init.pfm")); p = pp.load_pfm(os.path.join(out, "0_prob.pfm"))
    assert d.shape == size and p.shape == size and np.isfinite(d).all() and np.isfinite(p).all()
    cam = pp.load_cam(os.path.join(out, "0.txt"))
    assert cam[1, 0, 0] == pytest.approx(60.0 * 2 / 3 * (1.0 if upsample else 0.25))      # full vs /4 intrinsics


def test_benchmark_driver_on_session_with_gt_depth(tmp_path, lib_built):
    """mvsnet_amd.test (the reference's test.py): depth inference scored against ground-truth depth PNGs."""
    from tests.test_loss_and_benchmark import _session_with_depth
    from mvsnet_amd import test as T
    sess = _session_with_depth(str(tmp_path / "sess"))
    res = str(tmp_path / "results.csv")
    avg = T.main(["--input_dir", sess, "--view_num", "3", "--max_d", "8", "--width", "64", "--height", "64",
                  "--base_image_size", "8", "--max_clusters_per_session", "2", "--results_path", res])
    loss, less_one, less_three, debug = avg
    assert np.isfinite(loss) and loss > 0 and 0.0 <= less_one <= less_three <= 1.0 and np.isfinite(debug)
    lines = open(res).readlines()
    assert lines[0] == T.RESULTS_HEADER and len(lines) == 2


def test_metric_workload_properties_and_mfma_vs_scalar(lib_built):
    """Full-size (N=5, D=192, 160x128) checks that do not need the oracle: the depth map stays
    inside the swept range, probabilities are finite, the MFMA and scalar regularisers agree, and
    identical source views give a cost volume of exactly zero."""
    from mvsnet_amd import _lib as L
    from mvsnet_amd.model import MVSNetWeights, DepthPlan, cost_volume
    w = S.make_workload("M")
    rp = S.make_regnet_params("normal", seed=1)
    weights = MVSNetWeights.from_numpy("normal", regnet=rp, device=DEV)
    feats, cams = t(w.features), t(w.cams)
    plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, weights, "3DCNN", DEV)
    plan.set_cameras(cams, w.depth_start, w.depth_interval, w.depth_end, False)
    d1, p1 = plan.run_3dcnn(feats, w.depth_start, w.depth_interval)
    d1, p1 = d1.clone(), p1.clone()
    cost_sum = float(plan.cost.double().sum())
    assert np.isfinite(cost_sum) and float(plan.cost.min()) > -1e-3       # variance >= 0 up to rounding
    assert float(d1.min()) >= w.depth_start and float(d1.max()) <= w.depth_end
    assert torch.isfinite(p1).all() and float(p1.min()) >= 0
    L.set_conv_impl("scalar")
    try:
        d2, p2 = plan.run_3dcnn(feats, w.depth_start, w.depth_interval)
        d2 = d2.clone()
    finally:
        L.set_conv_impl("auto")
    rel = float(((d1 - d2).abs() / d2).mean())
    assert rel < 1e-4, rel
    # identical views through an identity transform: exact zeros at full size
    same = feats[:1].expand(5, -1, -1, -1).contiguous()
    ident = torch.zeros((4, w.depth_num, 8), device=DEV); ident[..., 0] = 1; ident[..., 4] = 1
    cz = cost_volume(same[0], same[1:], ident)
    assert float(cz.abs().max()) < 1e-5


def test_plan_runs_under_hipgraph_capture(lib_built):
    """The C ABI never allocates or synchronises, so a whole features->depth pass can be captured
    into a hipGraph and replayed."""
    from mvsnet_amd.model import MVSNetWeights, DepthPlan
    w = S.make_workload("small")
    rp = S.make_regnet_params("normal", seed=1)
    weights = MVSNetWeights.from_numpy("normal", regnet=rp, device=DEV)
    feats, cams = t(w.features), t(w.cams)
    plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, weights, "3DCNN", DEV)

    def run():       # the one-call entry (mvs_depth_from_features_f32), as inference_mem uses it
        plan.run_depth(feats, cams, w.depth_start, w.depth_interval, w.depth_end, False)
    plan.set_cameras(cams, w.depth_start, w.depth_interval, w.depth_end, False)
    split = plan.run_3dcnn(feats, w.depth_start, w.depth_interval)[0].clone()       # per-stage entries: same launches
    run()
    torch.cuda.synchronize()
    eager = plan.depth.clone()
    assert torch.allclose(eager, split, rtol=1e-6, atol=1e-4)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            run()
    plan.depth.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.allclose(plan.depth, eager, rtol=1e-6, atol=1e-4)


def test_gru_sweep_prepare_capture_and_release(lib_built):
    """include/mvsnet_hip.h: mvs_gru_prepare(stream) is the ONE call of the recurrent path that creates streams / events and
    synchronises; mvs_gru_wta*_f32 never does.  Round 5: the default sweep is the fused two-launches-per-plane pipeline on the
    caller's stream alone (csrc/gru_fused.hip) -- it needs no set, captures into a hipGraph as it is and replays to the eager
    result bit for bit; the round-4 four-stream wavefront (formulation 1, needs a set) gives the same planes up to near-ties.  prepare is
    refused under capture (MVS_E_NOT_PREPARED = -4); release / prepare recycle the slot; the 17th set of a process is
    MVS_E_NO_SLOT = -5, a warning in Python, and the sweep carries on; DepthPlan.close() gives a set back."""
    import ctypes as C
    import warnings
    from mvsnet_amd import _lib
    from mvsnet_amd.model import DepthPlan, MVSNetWeights, wta_depth_values
    lib = _lib.load()
    base = S.make_workload("c3")
    Hh, Ww, D = 52, 72, 40                               # ragged tiles, enough planes for the wavefront (D > 8)
    gp = S.make_gru_params("normal", seed=2, in_channels=base.channels, random_affine=True)
    weights = MVSNetWeights.from_numpy("normal", gru=gp, device=DEV)
    feats = t(S.make_features(base.view_num, base.height, base.width, base.channels, seed=11)[:, :Hh, :Ww])
    end = base.depth_start + (D - 1) * base.depth_interval
    dv = wta_depth_values(D, base.depth_start, end, False)
    cams = t(base.cams)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        plan = DepthPlan(base.view_num, D, Hh, Ww, base.channels, weights, "GRU", DEV)       # prepares stream s
        pc, us = C.c_int(-9), (C.c_float * 8)()
        assert lib.mvs_gru_stream_layout(_lib.stream_ptr(), C.byref(pc), us) == 0 and -1 <= pc.value <= 3 and min(us) > 0
        assert lib.mvs_gru_prepare(_lib.stream_ptr()) == 0                                   # idempotent
        plan.set_cameras(cams, base.depth_start, base.depth_interval, end, False)
        try:
            _lib.check(lib.mvs_gru_set_formulation(1), "mvs_gru_set_formulation")
            d, p = plan.run_gru(feats, dv)                                                   # the round-4 wavefront over the set's streams
            torch.cuda.synchronize()
            wave_d, wave_p = d.clone(), p.clone()
        finally:
            _lib.check(lib.mvs_gru_set_formulation(0), "mvs_gru_set_formulation")
        d, p = plan.run_gru(feats, dv)                                                       # the default: the fused sweep
        torch.cuda.synchronize()
        eager_d, eager_p = d.clone(), p.clone()
        assert len(torch.unique(eager_d)) > 4
        same = wave_d == eager_d
        assert float((~same).float().mean()) <= 2e-4
        assert float(((wave_p - eager_p).abs() / eager_p)[same].max()) < 2e-3
        g = torch.cuda.CUDAGraph()
        rcs = []
        with torch.cuda.graph(g, stream=s):
            plan.run_gru(feats, dv)                                                          # under capture: the fused sweep
            rcs.append(lib.mvs_gru_prepare(_lib.stream_ptr()))                               # prepare synchronises: refused under capture
            rcs.append(lib.mvs_gru_release(C.c_void_p(12345)))                               # no set for this handle
        assert rcs == [-4, -1]
    for _ in range(2):                                   # a graph is replayable
        plan.depth.zero_(); plan.prob.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(plan.depth, eager_d) and torch.equal(plan.prob, eager_p)
    # a stream nobody prepared: the fused sweep, the same bits
    gw = weights.gru
    f1, f2, f3 = gw.filters
    s2 = torch.cuda.Stream()
    s2.wait_stream(torch.cuda.current_stream())
    dvc = (C.c_float * D)(*[float(v) for v in dv])
    with torch.cuda.stream(s2):
        call = lambda: lib.mvs_gru_wta_f32(_lib.ptr(feats[0]), _lib.ptr(feats[1:]), _lib.ptr(plan.transforms), base.view_num, D, Hh, Ww,
                                           base.channels, f1, f2, f3, gw.ptrs, dvc, C.c_void_p(plan.workspace.data_ptr()),
                                           plan.workspace.numel(), _lib.ptr(plan.depth), _lib.ptr(plan.prob), _lib.stream_ptr())
        assert lib.mvs_gru_stream_layout(_lib.stream_ptr(), None, None) == -4
        assert lib.mvs_gru_release(_lib.stream_ptr()) == -1                                  # nothing to release
        plan.depth.zero_()
        assert call() == 0
        torch.cuda.synchronize()
        assert torch.equal(plan.depth, eager_d) and torch.equal(plan.prob, eager_p)
    # slots are recycled
    with torch.cuda.stream(s):
        _lib.gru_release()
        assert lib.mvs_gru_stream_layout(_lib.stream_ptr(), None, None) == -4
        _lib.gru_prepare()
        assert lib.mvs_gru_stream_layout(_lib.stream_ptr(), None, None) == 0
        d3, _ = plan.run_gru(feats, dv)
        torch.cuda.synchronize()
        assert torch.equal(d3, eager_d)
    assert _lib.load().mvs_error_string(-4).decode().startswith("mvsnet_hip: no side streams")
    # the 17th stream set: its own error code, a warning in Python, and the plan still works
    streams, plans = [torch.cuda.Stream() for _ in range(17)], []
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        for st_ in streams:
            with torch.cuda.stream(st_):
                plans.append(DepthPlan(base.view_num, D, Hh, Ww, base.channels, weights, "GRU", DEV))
    assert any("stream sets" in str(w_.message) for w_ in rec)
    assert lib.mvs_error_string(-5).decode().startswith("mvsnet_hip: all 16 stream sets")
    with torch.cuda.stream(streams[-1]):
        plans[-1].set_cameras(cams, base.depth_start, base.depth_interval, end, False)
        d17, p17 = plans[-1].run_gru(feats, dv)
        torch.cuda.synchronize()
        assert torch.equal(d17, eager_d) and torch.equal(p17, eager_p)                       # no set: the fused sweep
    #  the refused stream is remembered: running again neither asks mvs_gru_prepare again nor grows the plan's token list
    n_tok = len(plans[-1]._gru_keys)
    with torch.cuda.stream(streams[-1]):
        plans[-1].run_gru(feats, dv)
        torch.cuda.synchronize()
    assert len(plans[-1]._gru_keys) == n_tok and all(k_ is not None for k_ in plans[-1]._gru_keys)
    # a stale token (its set was released with gru_release and the stream prepared again) cannot give the NEW set away
    with torch.cuda.stream(s):
        _lib.gru_release()
        tok_old = _lib.gru_prepare()
        _lib.gru_release()
        tok_new = _lib.gru_prepare()
        assert tok_old[:2] == tok_new[:2] and tok_old[2] != tok_new[2]
        _lib.gru_unref(tok_old)
        assert lib.mvs_gru_stream_layout(_lib.stream_ptr(), None, None) == 0                # still prepared
        _lib.gru_unref(tok_new)
        assert lib.mvs_gru_stream_layout(_lib.stream_ptr(), None, None) == -4               # the last real user released it
    # close() re-arms: the plan prepares (a share of) a set again at its next sweep and gives it back at the next close()
    with torch.cuda.stream(s):
        plan.close()
        plan.run_gru(feats, dv)
        torch.cuda.synchronize()
        assert lib.mvs_gru_stream_layout(_lib.stream_ptr(), None, None) == 0 and plan._finalizer is not None
        plan.close()
        assert lib.mvs_gru_stream_layout(_lib.stream_ptr(), None, None) == -4
    n_before = len(_lib._GRU_PREPARED)
    plans[0].close()                                                                          # gives its set back ...
    assert len(_lib._GRU_PREPARED) == n_before - 1
    with torch.cuda.stream(streams[-1]):
        assert _lib.gru_prepare() is not None                                                 # ... and the 17th stream gets it
        _lib.gru_release()
    for pl in plans:
        pl.close()
    plan.close()


def test_lite_mode_from_images_runs_padded_regulariser_and_torch_towers():
    """network_mode 'lite' end to end: narrow towers stay on the PyTorch module (GroupNorm groups < 8 channels),
    the regulariser runs zero-padded on the MFMA shapes; checked against the oracle composition."""
    import numpy as np
    import torch
    from oracle import mvsnet_oracle as O
    from mvsnet_amd import synthetic as S
    from mvsnet_amd.model import MVSNetWeights, inference_mem
    from mvsnet_amd.feature_net import UNetDS2GN
    mode = "lite"
    up, rp = S.make_unet_params(mode, seed=3), S.make_regnet_params(mode, seed=1, random_affine=True)
    weights = MVSNetWeights.from_numpy(mode, unet=up, regnet=rp, device="cuda")
    assert isinstance(weights.unet, UNetDS2GN) and weights.regnet.cin == 32 and weights.regnet.cin_native == 16
    N, H, W, D = 3, 64, 64, 8
    images = S.make_images(N, H, W, seed=2)
    cams = S.make_cams(N, H // 4, W // 4, D, interval=40.0)
    start, interval = float(cams[0, 1, 3, 0]), float(cams[0, 1, 3, 1])
    depth, prob = inference_mem(torch.as_tensor(images).cuda()[None], torch.as_tensor(cams).cuda()[None], D, start, interval,
                                mode, weights=weights, view_num=N)
    feats = np.stack([O.unet_ds2gn(images[v], up, np.float64) for v in range(N)])
    ed, _ep = O.inference_mem_from_features(feats, cams, D, start, interval, rp, False, np.float64)
    d = depth.cpu().numpy()[0, :, :, 0]
    assert float(np.mean(np.abs(d - ed) / ed)) < 1e-3


@pytest.mark.parametrize("stereo", [False, True])
@pytest.mark.parametrize("upsample", [False, True])
@pytest.mark.parametrize("conf", [False, True])
@pytest.mark.parametrize("network", ["original", "unet"])
def test_depth_refine_on_the_device_matches_oracle(network, conf, upsample, stereo):
    """SURVEY 8f f3 on the device: depth_refine (model.py:753-811) with both refinement towers (mvsnetworks.py:178-193,
    261-324) on cuda -- TF1 resize_bilinear, SAME-padded biased convolutions, transposed convolutions cropped at the end,
    residual re-scaling -- against the float64 numpy oracle, every flag combination of --upsample_before_refinement /
    --refine_with_confidence / --refine_with_stereo."""
    from tests.test_refine_and_fusion import _case
    _case(network, conf, upsample, 11 + 2 * conf + upsample, stereo=stereo, device=DEV)


@pytest.mark.parametrize("regularization", ["3DCNN", "GRU"])
def test_session_pipeline_equals_the_per_reference_view_calls(tmp_path, lib_built, regularization):
    """compute_depth_maps (loader threads, per-image decode cache, uint8 upload + standardisation on the device, one tower
    pass per group of reference views, several views per recurrent sweep, asynchronous D2H) writes, reference view by
    reference view, what the plain sequence of the reference's loop gives (mvsnet/inference.py:105-119: generator ->
    inference -> write): towers on the host-standardised images, one hot-path call per view."""
    from mvsnet_amd.inference import build_weights, compute_depth_maps
    from mvsnet_amd.mvs_data_generation import make_generator
    from mvsnet_amd import predictlib as pl, preprocess as pp
    sess = S.write_session(str(tmp_path / "sess"), n_images=7, height=100, width=132, view_num=3, depth_num=24, interval=10.0)
    cfg = pl.InferenceConfig(input_dir=sess, view_num=3, max_d=24, width=128, height=96, base_image_size=8,
                             regularization=regularization, output_dir=str(tmp_path / "out"))
    weights = build_weights(cfg, torch.device("cuda", 0))
    tm = {}
    n = compute_depth_maps(sess, cfg, weights, torch.device("cuda", 0), timings=tm, gru_views=3)
    assert n == 7 and tm["depth_maps"] == 7 and tm["wall"] > 0 and tm["hot_path"] > 0
    assert tm["host_workers"] >= 2 and tm["load"] > 0 and tm["write"] > 0 and tm["towers"] > 0       # worker processes (host_pool)
    # the same session with the loaders / writers on threads of this process (host_workers = 0): the same files, byte for byte
    cfg0 = pl.InferenceConfig(input_dir=sess, view_num=3, max_d=24, width=128, height=96, base_image_size=8,
                              regularization=regularization, output_dir=str(tmp_path / "out_threads"))
    tm0 = {}
    assert compute_depth_maps(sess, cfg0, weights, torch.device("cuda", 0), timings=tm0, gru_views=3, host_workers=0) == 7
    assert tm0["host_workers"] == 0
    for fn in sorted(os.listdir(cfg.output_dir)):
        assert open(os.path.join(cfg.output_dir, fn), "rb").read() == open(os.path.join(cfg0.output_dir, fn), "rb").read(), fn
    assert len(os.listdir(cfg0.output_dir)) == 42
    # round 6: the uploads + towers of a group run on a stream of their own (the default above); in line they give the same maps
    cfg1 = pl.InferenceConfig(input_dir=sess, view_num=3, max_d=24, width=128, height=96, base_image_size=8,
                              regularization=regularization, output_dir=str(tmp_path / "out_inline"))
    assert compute_depth_maps(sess, cfg1, weights, torch.device("cuda", 0), gru_views=3, tower_stream=False) == 7
    for fn in sorted(f_ for f_ in os.listdir(cfg.output_dir) if f_.endswith("_init.pfm")):
        a_, b_ = pp.load_pfm(os.path.join(cfg.output_dir, fn)), pp.load_pfm(os.path.join(cfg1.output_dir, fn))
        if regularization == "3DCNN":
            assert float(np.max(np.abs(a_ - b_) / b_)) < 1e-5, fn
        else:
            assert float((a_ == b_).mean()) > 0.99, fn
    gen = make_generator(sess, 3, 128, 96, 24, 1.0, 8, mode="inference", output_scale=0.25)
    for c in sorted(gen.clusters, key=lambda c_: c_.ref_index):
        out_images, in_images, out_cams, full_cams, index = gen.prepare(c)          # float32, standardised on the host
        assert in_images.dtype == np.float32
        d, p, _ = pl.get_depth_and_prob_map(t(in_images)[None], t(out_cams)[None], float(out_cams[0, 1, 3, 0]),
                                            float(out_cams[0, 1, 3, 1]), cfg, weights, depth_num=int(out_cams[0, 1, 3, 2]),
                                            depth_end=float(out_cams[0, 1, 3, 3]))
        want_d, want_p = d.cpu().numpy()[0, :, :, 0], p.cpu().numpy()[0, :, :, 0]
        got_d = pp.load_pfm(os.path.join(cfg.output_dir, "%d_init.pfm" % index))
        got_p = pp.load_pfm(os.path.join(cfg.output_dir, "%d_prob.pfm" % index))
        assert got_d.shape == want_d.shape == (24, 32)
        if regularization == "3DCNN":
            # device standardisation (float64 moments) against numpy's float32 reductions: ~1e-7 on the inputs
            assert float(np.mean(np.abs(got_d - want_d) / want_d)) < 1e-5
            assert float((np.abs(got_p - want_p) > 1e-3).mean()) < 0.02
        else:
            assert float((got_d == want_d).mean()) > 0.97          # winner-take-all: a near-tie may flip on a few pixels


def test_bench_and_inference_start_their_own_ranks(tmp_path, lib_built):
    """`python bench.py --gpus 2` and `python -m mvsnet_amd.inference --gpus 2` WITHOUT a launcher: the parent stays GPU-less and
    starts one worker per rank (torch.distributed.run); rehearsed on this one-GPU box over gloo with both ranks on cuda:0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MVS_DIST_BACKEND="gloo", MVS_ALLOW_SHARED_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--no-extra", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["ranks"]["world_size"] == 2 and len(rec["ranks"]["devices"]) == 2
    assert len(rec["per_rank_depth_maps_per_s"]["ranks"]) == 2 and rec["value"] > 0
    assert rec["scaling"] == "weak" and "roofline" in rec
    # a rank count the node cannot serve over RCCL is refused with a message, not a hang
    env2 = {k: v for k, v in env.items() if k not in ("MVS_DIST_BACKEND", "MVS_ALLOW_SHARED_GPU")}
    if torch.cuda.device_count() < 2:
        r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--no-extra",
                             "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env2, cwd=root)
        assert r2.returncode != 0 and "needs 2 GPUs" in (r2.stdout + r2.stderr)
    sess = S.write_session(str(tmp_path / "sess"), n_images=5, height=96, width=128, view_num=3, depth_num=8, interval=60.0)
    out = str(tmp_path / "out")
    r3 = subprocess.run([sys.executable, "-m", "mvsnet_amd.inference", "--gpus", "2", "--input_dir", sess, "--output_dir", out,
                         "--view_num", "3", "--max_d", "8", "--width", "128", "--height", "96"],
                        capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r3.returncode == 0, (r3.stdout[-2000:], r3.stderr[-2000:])
    assert sorted(f for f in os.listdir(out) if f.endswith("_init.pfm")) == ["%d_init.pfm" % i for i in range(5)]
    # several worker processes per GPU (the session loop is bound by one Python thread per process): no environment needed,
    # the flag itself selects gloo + shared GPUs for its ranks; the summary line reports all passes after the first
    env3 = {k: v for k, v in env.items() if k not in ("MVS_DIST_BACKEND", "MVS_ALLOW_SHARED_GPU")}
    out2 = str(tmp_path / "out2")
    r4 = subprocess.run([sys.executable, "-m", "mvsnet_amd.inference", "--procs_per_gpu", "2", "--passes", "2", "--input_dir", sess,
                         "--output_dir", out2, "--view_num", "3", "--max_d", "8", "--width", "128", "--height", "96"],
                        capture_output=True, text=True, timeout=600, env=env3, cwd=root)
    assert r4.returncode == 0, (r4.stdout[-2000:], r4.stderr[-2000:])
    rec4 = json.loads([l for l in r4.stdout.splitlines() if l.startswith("{")][-1])
    assert rec4["ranks"] == 2 and rec4["depth_maps"] == 5 and rec4["depth_maps_per_s"] > 0
    assert rec4["devices"] == [0, 0]                        # both worker processes on the ONE requested GPU (MVS_GPUS)
    for i in range(5):                                     # same files as the two-rank run above
        a = open(os.path.join(out, "%d_init.pfm" % i), "rb").read()
        assert a == open(os.path.join(out2, "%d_init.pfm" % i), "rb").read()


def test_bench_with_four_ranks_on_one_card_and_a_failing_rank(lib_built):
    """The N-rank record of `bench.py --gpus N` beyond two ranks, rehearsed over gloo with every rank on cuda:0: the GPU box
    admits six processes on its card (this test process and the launcher count: five ranks were killed by the process guard), so
    four ranks run here and the 8-rank plumbing (launcher, rendezvous, shard of 1078 views, gathers, exit code) runs on the CPU
    (tests/test_data_and_sharding.py).  One failing rank out of four must end the whole run with a non-zero exit code and no
    JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MVS_DIST_BACKEND="gloo", MVS_ALLOW_SHARED_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1", "--workload", "small",
           "--no-extra", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 4 and rec["ranks"]["world_size"] == 4 and rec["ranks"]["backend"] == "gloo"
    assert [d["rank"] for d in rec["ranks"]["devices"]] == list(range(4))
    assert len(rec["per_rank_depth_maps_per_s"]["ranks"]) == 4 and rec["scaling"] == "weak"
    # what a SCALE record needs to prove N distinct GPUs: per rank the device's PCI bus id, per job the RCCL version
    assert all(d.get("pci_bus_id") for d in rec["ranks"]["devices"]) and rec["ranks"]["distinct_pci_bus_ids"] == 1      # four ranks folded onto ONE card here
    assert "rccl_version" in rec["ranks"] and 0 < rec["per_rank_depth_maps_per_s"]["min"] <= rec["per_rank_depth_maps_per_s"]["max"]
    assert abs(rec["value"] - 4 * rec["steps"] / (rec["ms_per_step"] * 1e-3 * rec["steps"])) < 1e-6 * rec["value"]   # whole job / slowest rank
    bad = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(env, MVS_BENCH_FAIL_RANK="2"), cwd=root)
    assert bad.returncode != 0
    assert not [l for l in bad.stdout.splitlines() if l.startswith("{")]


def test_compute_depth_maps_on_an_upstream_pair_txt_project(tmp_path, lib_built):
    """The upstream MVSNet project format (pair.txt, cams/%08d_cam.txt, images/%08d.jpg; README.md:165-215) through the same
    pipeline: its clusters take the host-standardised float32 path (no per-image decode cache), same outputs on disk."""
    from tests.test_data_and_sharding import make_pair_project
    from mvsnet_amd.inference import compute_depth_maps
    from mvsnet_amd import predictlib as pl, preprocess as pp
    proj = make_pair_project(str(tmp_path / "proj"), n_images=4, h=96, w=128)
    cfg = pl.InferenceConfig(view_num=3, max_d=8, width=128, height=96, base_image_size=8, interval_scale=20.0)
    tm = {}
    n = compute_depth_maps(proj, cfg, timings=tm)
    assert n == 4 and tm["depth_maps"] == 4
    out = os.path.join(proj, "depths_mvsnet")
    for idx in range(4):
        d = pp.load_pfm(os.path.join(out, "%d_init.pfm" % idx))
        assert d.shape == (24, 32) and np.isfinite(d).all() and d.min() >= 425.0 - 1e-3
        for suffix in ("_prob.pfm", "_depth.png", "_prob.png", ".jpg", ".txt"):
            assert os.path.exists(os.path.join(out, "%d%s" % (idx, suffix)))


def test_rccl_single_rank_runs_every_collective_of_the_multi_gpu_paths(lib_built):
    """N > 1 on hardware is the driver's run, on a node this box does not have.  What CAN be checked on one GPU:
    that RCCL (torch.distributed backend "nccl") initialises on this image the way bench.py / shard.py / train.py do it
    (device_id given) and accepts every collective those paths issue, with their dtypes and reduce ops, on device tensors --
    a world of one rank, in a child process (a process group is per process)."""
    import subprocess
    import sys
    code = r'''
import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MVS_TEST_PORT", "29631"))
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
dist.barrier(); torch.cuda.synchronize()
tt = torch.tensor([1.25], device=dev, dtype=torch.float64)                 # bench.py: per-rank seconds
every = [torch.zeros_like(tt)]
dist.all_gather(every, tt); dist.all_reduce(tt, op=dist.ReduceOp.MAX)
assert float(every[0]) == 1.25 and float(tt) == 1.25
objs = [None]; dist.all_gather_object(objs, {"rank": 0, "device": torch.cuda.get_device_name(0)})     # bench.py: device names
assert objs[0]["rank"] == 0
g = torch.randn(1 << 20, device=dev); ref = g.clone()                       # train.py / backward.py: gradient buckets, sync-BN sums
dist.all_reduce(g); assert torch.equal(g, ref)
h = torch.randn(257, device=dev, dtype=torch.float64); ref = h.clone()
dist.all_reduce(h); assert torch.equal(h, ref)
n = torch.tensor([7], device=dev); dist.all_reduce(n, op=dist.ReduceOp.MIN); assert int(n) == 7       # train.py: common step count
f = torch.tensor([0], device=dev); dist.all_reduce(f, op=dist.ReduceOp.MAX); assert int(f) == 0       # train.py: stop flag
from mvsnet_amd import shard
assert shard.gather_counts(dist, 3.5, device=dev) == [3.5]
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_over_rccl_with_a_world_of_one_rank(lib_built):
    """bench.py's N > 1 code path (barrier, timed steps, barrier, all_gather + MAX all-reduce of the seconds on DEVICE
    tensors, all_gather_object of the device names) on the backend the driver's 8-GPU run uses -- RCCL -- with the one rank
    a one-GPU box can give it (MVS_BENCH_FORCE_DIST=1)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29633",
               MVS_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MVS_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--no-extra", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["ranks"]["world_size"] == 1 and rec["ranks"]["backend"] == "nccl (RCCL)" and rec["n_gpus"] == 1
    assert rec["ranks"]["devices"][0]["rank"] == 0 and rec["value"] > 100
