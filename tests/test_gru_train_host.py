"""CPU tests of the recurrent-regulariser training host code (model.py:505-599, loss.py:223-267):
the x-part-hoisted, cell-by-cell sweep of mvsnet_amd/gru_train.py against the plane-by-plane concatenated form of the
checker (oracle/torch_grad.py), the checker's forward against the strict numpy restatement, and the classification
loss.  No compute calls into the HIP library here (the cost volume is the only HIP piece of that path; its GPU tests
live in tests/test_gpu_backward.py)."""
import numpy as np
import pytest
import torch

from oracle import mvsnet_oracle as O
from oracle import torch_grad as TG
from mvsnet_amd import synthetic as S


def _d64(a, grad=False):
    return torch.tensor(np.asarray(a, np.float64)).requires_grad_(grad)


def _gru64(params, grad=False):
    return {k: ({kk: _d64(vv, grad) for kk, vv in v.items()} if isinstance(v, dict) else _d64(v, grad))
            for k, v in params.items()}


def _toy(N=3, D=6, H=10, W=14, C=32):
    cams = S.make_cams(N, H, W, D)
    feats = S.make_features(N, H, W, C, seed=5)
    start, interval = float(cams[0, 1, 3, 0]), float(cams[0, 1, 3, 1])
    Hs = np.stack([O.get_homographies(cams[0], cams[v], D, start, interval, np.float64) for v in range(1, N)])
    return feats, Hs, O.homography_to_transform8(Hs, np.float64), start, interval


@pytest.mark.parametrize("mode", ["normal", "lite"])
def test_checker_forward_equals_the_numpy_restatement(mode):
    feats, Hs, t8, _s, _i = _toy(C=32 if mode == "normal" else 16)
    gp = S.make_gru_params(mode, in_channels=feats.shape[-1], random_affine=True)
    reg = TG.recurrent_reg(_d64(feats), _d64(t8), _gru64(gp)).numpy()
    # the numpy restatement, plane by plane (winner_take_all's loop body, model.py:676-702)
    D = Hs.shape[1]
    f = [gp[k]["out_b"].shape[0] for k in ("gru1", "gru2", "gru3")]
    s = [np.zeros(feats.shape[1:3] + (n,), np.float64) for n in f]
    for d in range(D):
        warped = [O.tf_transform_homography(feats[v + 1], Hs[v, d], np.float64) for v in range(Hs.shape[0])]
        cost = O.variance_cost_eager(feats[0], warped, feats.shape[0], np.float64)
        s[0] = O.conv_gru_cell(-cost, s[0], gp["gru1"], np.float64)
        s[1] = O.conv_gru_cell(s[0], s[1], gp["gru2"], np.float64)
        s[2] = O.conv_gru_cell(s[1], s[2], gp["gru3"], np.float64)
        want = O.conv2d_same(s[2], gp["prob_w"], 1, gp["prob_b"], np.float64)[..., 0]
        assert np.abs(reg[d] - want).max() < 1e-9, d


def test_hoisted_sweep_matches_the_concatenated_cells_forward_and_backward():
    """conv([x | h]) = conv_x(x) + conv_h(h), x parts batched over the planes: values and every gradient."""
    from mvsnet_amd import gru_train as G
    rs = np.random.RandomState(3)
    D, H, W = 5, 9, 12
    gp = S.make_gru_params("normal", in_channels=32, random_affine=True)
    x = rs.randn(D, 32, H, W)
    g = rs.randn(D, H, W)

    def run(fn):
        xt, pt = _d64(x, True), _gru64(gp, True)
        reg = fn(xt, pt)
        (reg * _d64(g)).sum().backward()
        return reg.detach().numpy(), xt.grad.numpy(), pt

    def hoisted(xt, pt):
        s3 = G.conv_gru_sweep(G.conv_gru_sweep(G.conv_gru_sweep(xt, pt["gru1"]), pt["gru2"]), pt["gru3"])
        return torch.nn.functional.conv2d(s3, pt["prob_w"].permute(3, 2, 0, 1), pt["prob_b"], padding=1)[:, 0]

    def concatenated(xt, pt):
        s = [torch.zeros((1, pt[k]["out_b"].shape[0], H, W), dtype=torch.float64) for k in ("gru1", "gru2", "gru3")]
        out = []
        for d in range(D):
            s[0] = TG.conv_gru_cell(xt[d:d + 1], s[0], pt["gru1"])
            s[1] = TG.conv_gru_cell(s[0], s[1], pt["gru2"])
            s[2] = TG.conv_gru_cell(s[1], s[2], pt["gru3"])
            out.append(TG._conv2d_same(s[2], pt["prob_w"], pt["prob_b"])[0, 0])
        return torch.stack(out, 0)

    a, ga, pa = run(hoisted)
    b, gb, pb = run(concatenated)
    assert np.abs(a - b).max() < 1e-10 and np.abs(ga - gb).max() < 1e-10 * max(1.0, np.abs(gb).max())
    for cell in ("gru1", "gru2", "gru3"):
        for key in pa[cell]:
            u, v = pa[cell][key].grad.numpy(), pb[cell][key].grad.numpy()
            assert np.abs(u - v).max() < 1e-9 * max(1.0, np.abs(v).max()), (cell, key)
    assert np.abs(pa["prob_w"].grad.numpy() - pb["prob_w"].grad.numpy()).max() < 1e-9


def test_classification_loss_matches_the_checker_and_its_edge_cases():
    from mvsnet_amd import gru_train as G
    rs = np.random.RandomState(1)
    D, H, W = 12, 7, 9
    start, interval = 425.0, 2.5
    reg = rs.randn(D, H, W)
    gt = start + interval * rs.uniform(-1.0, D, (H, W))        # some pixels fall outside [0, D): all-zero one-hot rows
    gt[0, :4] = 0.0                                             # invalid pixels
    gt[1, 0] = start + interval * 2.5                           # a tie: tf.round goes to the even index, 2
    gt[1, 1] = start + interval * 3.5                           # ... and here to 4
    want = float(TG.classification_loss(_d64(reg), _d64(gt), start, interval))
    prob = torch.softmax(_d64(reg), 0)
    loss, mae, l1, l3, wta = G.mvsnet_classification_loss(prob, _d64(gt)[None, :, :, None], D, [start], [interval])
    assert abs(float(loss) - want) < 1e-12 * max(1.0, abs(want))
    # by hand, pixel by pixel (loss.py:233-247)
    p = prob.numpy()
    total, valid = 0.0, 1e-7
    for y in range(H):
        for x in range(W):
            if gt[y, x] == 0.0:
                continue
            valid += 1
            k = int(np.round((gt[y, x] - start) / interval))   # numpy rounds half to even as tf.round does
            if 0 <= k < D:
                total -= np.log(p[k, y, x])
    assert abs(float(loss) - total / valid) < 1e-9
    assert int(np.round((gt[1, 0] - start) / interval)) == 2 and int(np.round((gt[1, 1] - start) / interval)) == 4
    # winner-take-all depth and the accuracies
    idx = p.argmax(0)
    assert np.allclose(wta[0, :, :, 0].numpy(), start + interval * idx)
    m = gt != 0
    err = np.abs(gt - (start + interval * idx)) / interval
    assert abs(float(l1) - (err[m] <= 1).sum() / (m.sum() + 1e-6)) < 1e-9
    assert abs(float(l3) - (err[m] <= 3).sum() / (m.sum() + 1e-6)) < 1e-9
    assert abs(float(mae) - err[m].sum() / (m.sum() + 1e-6)) < 1e-9
    # a reversed sweep (flip_cams: start at the far plane, negative interval) indexes the same planes backwards
    far = start + (D - 1) * interval
    loss_r, _mae, l1r, _l3, wta_r = G.mvsnet_classification_loss(prob.flip(0), _d64(gt)[None, :, :, None], D, [far], [-interval])
    inside = (np.round((gt - start) / interval) >= 0) & (np.round((gt - start) / interval) < D)
    if np.array_equal(inside, (np.round((gt - far) / -interval) >= 0) & (np.round((gt - far) / -interval) < D)):
        ties = np.abs(((gt - start) / interval) % 1.0 - 0.5) < 1e-9
        if not ties[m].any():
            assert abs(float(loss_r) - float(loss)) < 1e-9


def test_gru_trainer_constructs_and_names_its_variables():
    from mvsnet_amd import train as T
    tr = T.Trainer("lite", device="cpu", regularization="GRU", seed=2)
    g = tr.params.group("gru")
    assert tuple(g["gru1"]["gates_w"].shape) == (3, 3, 16 + 8, 16) and tuple(g["prob_w"].shape) == (3, 3, 1, 1)
    assert float(g["gru1"]["gates_b"].detach().abs().max()) == 0.0 and float(g["gru2"]["out_gamma"].detach().min()) == 1.0
    names = tr.params.named_arrays(tr.params.data)
    assert {"conv_gru1/Gates/conv/kernel", "conv_gru3/Output/LayerNorm/gamma", "prob_conv/kernel", "prob_conv/bias",
            "2dconv1_0/kernel"} <= set(names)
    assert not any(k.startswith("3dconv") for k in names)
    with pytest.raises(NotImplementedError):
        T.Trainer("normal", device="cpu", regularization="GRU", refinement=True)
    with pytest.raises(NotImplementedError):
        T.Trainer("normal", device="cpu", regularization="LSTM")
