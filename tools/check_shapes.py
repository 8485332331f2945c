import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import _lib as L, synthetic as S
from mvsnet_amd.model import MVSNetWeights, regnet_us0
dev='cuda:0'
rp = S.make_regnet_params("normal", seed=1, random_affine=True)
w = MVSNetWeights.from_numpy("normal", regnet=rp, device=dev)
for (D,H,W) in [(64,104,136),(48,24,40),(16,8,8),(40,72,56)]:
    cost = torch.rand(D,H,W,32, device=dev)
    outs={}
    for impl in ("scalar","auto"):
        L.set_conv_impl(impl)
        outs[impl]=regnet_us0(cost, w.regnet).clone()
    L.set_conv_impl("auto")
    a,b=outs["scalar"],outs["auto"]
    print((D,H,W), float((a-b).abs().max()/a.abs().max()))
