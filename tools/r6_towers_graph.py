"""Round 6: the training towers' forward + backward (HipTowers autograd node: ~660 launches, host-bound: enqueue time = step time)
captured into ONE hipGraph and replayed.   python tools/r6_towers_graph.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import synthetic as S, train as T
from mvsnet_amd.feature_net_train import hip_towers

N, H, W = 3, 480, 640
tr = T.Trainer("normal", "cuda")
images = torch.as_tensor(S.make_images(N, H, W)).cuda()
params = tr.params.group("unet")
leaves = [t_ for p_ in params.values() for t_ in p_.values()]


def step():
    for t_ in leaves:
        t_.grad = None
    f = hip_towers(images, params)
    f.sum().backward()
    return f


for _ in range(15): step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10): f = step()
t1 = time.time(); torch.cuda.synchronize()
print("eager: %.2f ms per step (host enqueue %.2f ms)" % ((time.time() - t0) / 10 * 1e3, (t1 - t0) / 10 * 1e3))
ref_f = f.detach().clone(); ref_g = [t_.grad.detach().clone() for t_ in leaves]

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    for t_ in leaves:
        t_.grad = None
    with torch.cuda.graph(g):
        f_static = hip_towers(images, params)
        f_static.sum().backward()
    torch.cuda.synchronize()
    grads_static = [t_.grad for t_ in leaves]
    g.replay(); torch.cuda.synchronize()
    err_f = float((f_static - ref_f).abs().max() / ref_f.abs().max())
    err_g = max(float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(grads_static, ref_g))
    t0 = time.time()
    for _ in range(20): g.replay()
    t1 = time.time(); torch.cuda.synchronize()
    print("hipGraph replay: %.2f ms per step (host %.2f ms); feature rel err %.1e, worst gradient rel err %.1e" % (
        (time.time() - t0) / 20 * 1e3, (t1 - t0) / 20 * 1e3, err_f, err_g))
except Exception as e:
    print("capture failed:", repr(e)[:400])
