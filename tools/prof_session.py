import cProfile, pstats, sys, tempfile, os, torch
sys.path.insert(0, ".")
from mvsnet_amd import synthetic as S
from mvsnet_amd.inference import build_weights, compute_depth_maps
from mvsnet_amd.predictlib import InferenceConfig
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
root = tempfile.mkdtemp()
S.write_session(root, n_images=24)
cfg = InferenceConfig(input_dir=root, view_num=5, max_d=192, width=640, height=512, sample_scale=0.25)
w = build_weights(cfg, dev)
cfg.output_dir = os.path.join(root, "o0"); compute_depth_maps(root, cfg, w, dev)
cfg.output_dir = os.path.join(root, "o1")
pr = cProfile.Profile(); pr.enable(); compute_depth_maps(root, cfg, w, dev); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
