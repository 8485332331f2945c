"""Where the FIRST pass over a session spends its time (the one-offs a one-scan process pays): cProfile of the main thread.
`python tools/r6_first_pass.py [n_images]`"""
import cProfile, os, pstats, sys, tempfile, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import synthetic as S
from mvsnet_amd.inference import build_weights, compute_depth_maps
from mvsnet_amd.predictlib import InferenceConfig
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 48
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
torch.zeros(1, device=dev); torch.cuda.synchronize()
root = tempfile.mkdtemp()
S.write_session(root, n_images=n_img, height=512, width=640, view_num=5, depth_num=192)
cfg = InferenceConfig(input_dir=root, view_num=5, max_d=192, width=640, height=512, sample_scale=0.25)
t0 = time.perf_counter(); w = build_weights(cfg, dev); torch.cuda.synchronize(); print("build_weights %.1f ms" % (1e3 * (time.perf_counter() - t0)))
cfg.output_dir = os.path.join(root, "o")
pr = cProfile.Profile(); tm = {}
t0 = time.perf_counter(); pr.enable(); n = compute_depth_maps(root, cfg, w, dev, timings=tm); pr.disable()
print("first pass %.1f ms for %d views" % (1e3 * (time.perf_counter() - t0), n))
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
