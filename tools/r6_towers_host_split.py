"""Host time of the training towers' forward + backward by category (no synchronisation inside: enqueue cost only):
ATen's convolution_backward calls against everything else."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import synthetic as S, train as T, feature_net_train as FT
tr = T.Trainer("normal", "cuda")
images = torch.as_tensor(S.make_images(3, 480, 640)).cuda()
acc = {"conv_bwd": 0.0, "n": 0}
real = FT._conv_bwd
def timed(*a, **k):
    t0 = time.perf_counter(); r = real(*a, **k); acc["conv_bwd"] += time.perf_counter() - t0; acc["n"] += 1
    return r
FT._conv_bwd = timed
def step():
    f = FT.hip_towers(images, tr.params.group("unet"), accumulate_into_grads=True)
    t0 = time.perf_counter(); f.sum().backward(); return time.perf_counter() - t0
for _ in range(5): step()
torch.cuda.synchronize(); acc.update(conv_bwd=0.0, n=0)
t0 = time.perf_counter(); bwd = sum(step() for _ in range(20)); t1 = time.perf_counter(); torch.cuda.synchronize()
print("per step: fwd+bwd host %.2f ms (wall %.2f), backward host %.2f ms, of which %d convolution_backward calls %.2f ms (%.0f us each)"
      % ((t1 - t0) / 20 * 1e3, (time.perf_counter() - t0) / 20 * 1e3, bwd / 20 * 1e3, acc["n"] / 20, acc["conv_bwd"] / 20 * 1e3, acc["conv_bwd"] / max(acc["n"], 1) * 1e6))
