import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mvsnet_amd import synthetic as S, train as T
N, H, W, D = 3, 480, 640, 128
images = S.make_images(N, H, W); cams = S.make_cams(N, H // 4, W // 4, D)
start, interval = float(cams[0, 1, 3, 0]), float(cams[0, 1, 3, 1])
gt = np.full((H // 4, W // 4, 1), start + interval * D * 0.5, np.float32)
tr = T.Trainer("normal", "cuda")
for _ in range(10): tr.train_step(images, cams, gt, D)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(10): tr.train_step(images, cams, gt, D)
t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
print("train step (config 5 per GPU): %.2f ms, host enqueue %.2f ms" % ((t2 - t0) / 10 * 1e3, (t1 - t0) / 10 * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): tr.train_step(images, cams, gt, D)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
