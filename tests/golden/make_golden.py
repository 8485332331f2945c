"""Writes the golden fixtures from the float64 CPU oracle: tests/golden/toy_*.npz (default) and, with
--full [M c1 c2 c3], the FULL-SIZE fixtures full_<workload>.npz of the BASELINE.json configurations.

The reference (ubiquity6/MVSNet) holds no fixtures and cannot be executed here (TensorFlow 1.12 /
python2 absent), so these vectors pin the build's own restatement, which is in turn pinned by the
hand-computed KATs of tests/test_oracle_kat.py.  Inputs are regenerated from seeds
(mvsnet_amd/synthetic.py); their SHA-256 is stored so RNG drift is detected.

    python tests/golden/make_golden.py                 # toy fixtures (strict numpy oracle, seconds)
    python tests/golden/make_golden.py --full          # M, c1, c2 (3D-CNN) and c3 (ConvGRU sweep): minutes, ~20 GB

Full-size fixtures come from oracle/torch_restatement.py in float64 (the strict numpy oracle's arithmetic,
held to it at ~1e-15 by tests/test_cpu_restatement.py, on all host cores): depth and probability maps as
float32 (c3: also the winning plane index as uint8), the SHA-256 of the regenerated inputs, and the distance of
the float32 CPU restatement from the float64 one (the rounding-noise floor a float32 device path is held to).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import mvsnet_oracle as O          # noqa: E402
from mvsnet_amd import synthetic as S          # noqa: E402


def abs_rel(a, b):
    return float(np.mean(np.abs(np.asarray(a, np.float64) - b) / np.abs(b)))


def full(names):
    import time
    import torch
    from oracle import torch_restatement as TR
    for name in names:
        if name == "c1img":                      # configs[0] from IMAGES: UNetDS2GN towers (numpy oracle) -> hot path (torch restatement)
            w = S.make_workload("c1")
            images = S.make_images(w.view_num, 4 * w.height, 4 * w.width, seed=0)
            up = S.make_unet_params("normal", seed=3)
            rp = S.make_regnet_params("normal", seed=1, random_affine=True)
            sha = hashlib.sha256(images.tobytes() + w.cams.tobytes()).hexdigest()
            t0 = time.time()
            out = {}
            for tag, npdt, tdt in (("f64", np.float64, torch.float64), ("f32", np.float32, torch.float32)):
                feats = np.stack([O.unet_ds2gn(images[v], up, npdt) for v in range(w.view_num)])
                out[tag] = TR.inference_mem_from_features(feats, w.cams, w.depth_num, w.depth_start, w.depth_interval, rp, tdt)
            d, p = out["f64"]
            extra = dict(f32_cpu_abs_rel=abs_rel(out["f32"][0], d), f32_cpu_prob_mismatch=float((np.abs(out["f32"][1] - p) > 1e-3).mean()))
            np.savez_compressed(os.path.join(HERE, "full_c1img.npz"), depth=d.astype(np.float32), prob=p.astype(np.float32),
                                input_sha256=sha, **extra)
            print("wrote full_c1img.npz in %.0f s: %s" % (time.time() - t0, extra), flush=True)
            continue
        inverse = name in ("c3inv", "Minv")      # --inverse_depth (R1'; model.py:706-713 for the sweep, :480-485,83-107 for the 3D-CNN tail): inputs of c3 / M
        w = S.make_workload({"c3inv": "c3", "Minv": "M"}.get(name, name))
        sha = hashlib.sha256(w.features.tobytes() + w.cams.tobytes()).hexdigest()
        t0 = time.time()
        if not name.startswith("c3"):
            rp = S.make_regnet_params("normal", seed=1, random_affine=True)
            d, p = TR.inference_mem_from_features(w.features, w.cams, w.depth_num, w.depth_start, w.depth_interval,
                                                  rp, torch.float64, inverse)
            d32, p32 = TR.inference_mem_from_features(w.features, w.cams, w.depth_num, w.depth_start,
                                                      w.depth_interval, rp, torch.float32, inverse)
            extra = dict(f32_cpu_abs_rel=abs_rel(d32, d), f32_cpu_prob_mismatch=float((np.abs(p32 - p) > 1e-3).mean()))
        else:
            gp = S.make_gru_params("normal", seed=2, in_channels=w.channels, random_affine=True)
            say = lambda a, b: print("  c3 plane %d/%d  %.0f s" % (a, b, time.time() - t0), flush=True)
            d, p, idx = TR.inference_winner_take_all_from_features(w.features, w.cams, w.depth_num, w.depth_start,
                                                                   w.depth_end, gp, torch.float64, say, inverse)
            d32, p32, i32 = TR.inference_winner_take_all_from_features(w.features, w.cams, w.depth_num, w.depth_start,
                                                                       w.depth_end, gp, torch.float32, None, inverse)
            same = i32 == idx
            extra = dict(index=idx.astype(np.uint8), f32_cpu_plane_agreement=float(same.mean()),
                         f32_cpu_prob_rel=float(np.max(np.abs(p32[same] - p[same]) / p[same])))
        np.savez_compressed(os.path.join(HERE, "full_%s.npz" % name), depth=d.astype(np.float32),
                            prob=p.astype(np.float32), input_sha256=sha, **extra)
        print("wrote full_%s.npz in %.0f s: %s" % (name, time.time() - t0,
                                                  {k: v for k, v in extra.items() if k != "index"}), flush=True)


def main():
    if "--full" in sys.argv:
        names = [a for a in sys.argv[1:] if a in S.WORKLOADS or a in ("c3inv", "c1img", "Minv")]
        return full(names or ["c1", "M", "c2", "c3"])
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
