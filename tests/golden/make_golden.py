"""Writes the golden fixtures tests/golden/toy_*.npz from the float64 CPU oracle.

The reference (ubiquity6/MVSNet) holds no fixtures and cannot be executed here (TensorFlow 1.12 /
python2 absent), so these vectors pin the build's own restatement, which is in turn pinned by the
hand-computed KATs of tests/test_oracle_kat.py.  Inputs are regenerated from seeds
(mvsnet_amd/synthetic.py); their SHA-256 is stored so RNG drift is detected.

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import mvsnet_oracle as O          # noqa: E402
from mvsnet_amd import synthetic as S          # noqa: E402


def main():
    w = S.make_workload("toy")
    sha = hashlib.sha256(w.features.tobytes() + w.cams.tobytes()).hexdigest()
    rp = S.make_regnet_params("normal", seed=1, random_affine=True)
    d, p = O.inference_mem_from_features(w.features, w.cams, w.depth_num, w.depth_start,
                                         w.depth_interval, rp, False, np.float64)
    np.savez_compressed(os.path.join(HERE, "toy_3dcnn.npz"), depth=d, prob=p, input_sha256=sha)
    gp = S.make_gru_params("normal", seed=2, in_channels=w.channels, random_affine=True)
    d, p = O.inference_winner_take_all_from_features(w.features, w.cams, w.depth_num, w.depth_start,
                                                     w.depth_end, gp, False, np.float64)
    np.savez_compressed(os.path.join(HERE, "toy_gru.npz"), depth=d, prob=p, input_sha256=sha)
    # training: gradients of sum(depth * g) w.r.t. the feature maps and a few RegNetUS0 variables, from
    # float64 autograd of the torch restatement (oracle/torch_grad.py)
    import torch
    from oracle import torch_grad as TG
    feats, cams = w.features[:3], w.cams[:3]
    D = 16
    Hs = np.stack([O.get_homographies(cams[0], cams[v], D, w.depth_start, w.depth_interval, np.float32) for v in range(1, 3)])
    t8 = O.homography_to_transform8(Hs, np.float32)
    g = np.random.RandomState(7).randn(*feats.shape[1:3])
    d64 = lambda a, req=False: torch.tensor(np.asarray(a, np.float64)).requires_grad_(req)
    f64 = d64(feats, True)
    p64 = {k: {kk: d64(vv, True) for kk, vv in v.items()} for k, v in rp.items()}
    depth = TG.depth_from_features(f64, d64(t8), w.depth_start, w.depth_interval, p64)
    (depth * d64(g)).sum().backward()
    np.savez_compressed(os.path.join(HERE, "toy_grad.npz"), depth=depth.detach().numpy(), g=g, t8=t8,
                        g_features=f64.grad.numpy(), g_w01=p64["3dconv0_1"]["w"].grad.numpy(),
                        g_w62=p64["3dconv6_2"]["w"].grad.numpy(), g_gamma30=p64["3dconv3_0"]["gamma"].grad.numpy(),
                        g_beta60=p64["3dconv6_0"]["beta"].grad.numpy(), input_sha256=sha)
    print("wrote golden fixtures; input sha256", sha)


if __name__ == "__main__":
    main()
