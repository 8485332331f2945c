import sys, numpy as np, torch
sys.path.insert(0, ".")
from mvsnet_amd import synthetic as S, backward as B
from mvsnet_amd.homography_warping import homography_transforms
DEV="cuda"
t=lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(DEV)
def run(N,D,H,W,epss):
    cams = S.make_cams(N, H, W, D)
    start, interval = float(cams[0, 1, 3, 0]), float(cams[0, 1, 3, 1])
    t8 = homography_transforms(t(cams), D, start, interval)
    rp = S.make_regnet_params("normal", seed=1, random_affine=True)
    f0 = t(S.make_features(N, H, W, 32, seed=5))
    gen = torch.Generator(device="cpu").manual_seed(11)
    g = torch.randn(H, W, generator=gen).to(DEV)
    v = torch.randn(N, H, W, 32, generator=gen).to(DEV)
    ft = f0.clone().requires_grad_(True)
    pt = {k: {kk: t(vv).requires_grad_(True) for kk, vv in p.items()} for k, p in rp.items()}
    depth, _ = B.plane_sweep_depth(ft, t8, start, interval, pt)
    (depth.double() * g.double()).sum().backward()
    dot = lambda a, b: float((a.double() * b.double()).sum())
    ana = dot(ft.grad, v)
    # per-view split of the analytic directional derivative
    parts = [dot(ft.grad[i], v[i]) for i in range(N)]
    out = []
    with torch.no_grad():
        for eps in epss:
            fw = lambda ff: dot(B.plane_sweep_depth(ff, t8, start, interval, pt)[0], g)
            num = (fw(f0 + eps * v) - fw(f0 - eps * v)) / (2 * eps)
            nparts = []
            for i in range(N):
                vi = torch.zeros_like(v); vi[i] = v[i]
                nparts.append((fw(f0 + eps * vi) - fw(f0 - eps * vi)) / (2 * eps))
            out.append((eps, num, nparts))
    print("N=%d D=%d %dx%d analytic %.5g parts %s" % (N, D, H, W, ana, ["%.5g" % p for p in parts]))
    for eps, num, nparts in out:
        print("   eps %.0e numeric %.5g parts %s" % (eps, num, ["%.5g" % p for p in nparts]))
run(3,16,32,48,[2e-3,1e-2])
run(3,128,120,160,[5e-4,2e-3,8e-3])
run(3,192,120,160,[2e-3])
run(3,128,128,160,[2e-3])
