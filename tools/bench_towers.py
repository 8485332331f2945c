"""Forward + backward of the 2D towers alone (torch autograd), steady state: where a training step's time goes."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import synthetic as S, train as T
from mvsnet_amd.feature_net import trainable_layers, unet_forward

N, H, W = 3, 480, 640
tr = T.Trainer("normal", "cuda")
images = torch.as_tensor(S.make_images(N, H, W)).cuda()
from mvsnet_amd.feature_net_train import hip_towers
mode = sys.argv[1] if len(sys.argv) > 1 else "hip"
def step():
    if mode == "hip":
        f = hip_towers(images, tr.params.group("unet"), accumulate_into_grads=os.environ.get("INTO", "1") == "1")
    else:
        f = unet_forward(trainable_layers(tr.params.group("unet")), images, hip_group_norm=(mode == "torch+hipgn"))
    f.sum().backward()
for _ in range(int(os.environ.get("WARM", "3"))): step()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(10): step()
t1 = time.time()
torch.cuda.synchronize()
print({"mode": mode, "towers_fwd_bwd_ms": round((time.time() - t0) / 10 * 1e3, 2), "host_enqueue_ms": round((t1 - t0) / 10 * 1e3, 2)})
