import cProfile, pstats, sys, tempfile, os, torch
sys.path.insert(0, ".")
from mvsnet_amd import synthetic as S
from mvsnet_amd.inference import build_weights, compute_depth_maps
from mvsnet_amd.predictlib import InferenceConfig
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
root = tempfile.mkdtemp()
S.write_session(root, n_images=48)
cfg = InferenceConfig(input_dir=root, view_num=5, max_d=192, width=640, height=512, sample_scale=0.25)
w = build_weights(cfg, dev)
cfg.output_dir = os.path.join(root, "o0"); compute_depth_maps(root, cfg, w, dev)
cfg.output_dir = os.path.join(root, "o1")
pr = cProfile.Profile(); pr.enable(); compute_depth_maps(root, cfg, w, dev); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
for hw in (None, 0):
    tm = {}; cfg.output_dir = os.path.join(root, "o2%s" % hw); n = compute_depth_maps(root, cfg, w, dev, timings=tm, host_workers=hw)
    print("host_workers=%s: %.1f depth maps/s" % (hw, n / tm["wall"]), {k: round(1e3 * v / n, 3) for k, v in tm.items() if isinstance(v, float)})
