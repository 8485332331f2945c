"""Session loop timing without a profiler: `python tools/session_time.py [n_images] [width height]` -> depth maps/s of the second and
third pass over a synthetic session and the per-depth-map breakdown (ms) compute_depth_maps reports."""
import os, sys, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import synthetic as S
from mvsnet_amd.inference import build_weights, compute_depth_maps
from mvsnet_amd.predictlib import InferenceConfig
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 48
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 512)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
root = tempfile.mkdtemp()
S.write_session(root, n_images=n_img, height=H, width=W, view_num=5, depth_num=192)
cfg = InferenceConfig(input_dir=root, view_num=5, max_d=192, width=W, height=H, sample_scale=0.25)
w = build_weights(cfg, dev)
for k in range(4):
    tm = {}; cfg.output_dir = os.path.join(root, "o%d" % k)
    n = compute_depth_maps(root, cfg, w, dev, timings=tm)
    print("pass %d: %d views, %.1f depth maps/s" % (k, n, n / tm["wall"]), {k_: round(1e3 * v / n, 3) for k_, v in tm.items() if isinstance(v, float)}, flush=True)
