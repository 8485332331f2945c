"""The training loop END TO END on an on-disk dataset in the reference's layout (train/<session>/images|cameras|depths,
covisibility.json): seconds per step of `python -m mvsnet_amd.train` with the input pipeline (worker processes preparing the next
clusters, uint8 upload, standardisation on the device) against one generator on the training thread (--no_prefetch).
Config 5's per-GPU shape: 3 views of 640 x 480, D = 128, from images stored at `src_w x src_h` (DTU stores 1600 x 1200: the
generator decodes, rescales and crops them, mvs_cluster.py:178-192).  `python tools/r6_train_input.py [steps] [src_w src_h]`"""
import io, os, re, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contextlib import redirect_stdout
from PIL import Image
from mvsnet_amd import synthetic as S, train as T

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
SW, SH = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1600, 1200)
root = tempfile.mkdtemp()
for mode, n in (("train", 2), ("val", 1)):
    for k in range(n):
        sdir = os.path.join(root, mode, "s%d" % k)
        S.write_session(sdir, n_images=32, height=SH, width=SW, view_num=3, depth_num=128, seed=k)
        os.makedirs(os.path.join(sdir, "depths"))
        rs = np.random.RandomState(k)
        for i in range(32):                                  # a fronto-parallel plane near the pivot, in uint16 millimetres
            d = (S.PIVOT_DEPTH + 20.0 * rs.standard_normal((SH, SW))).clip(1, 65535).astype(np.uint16)
            Image.fromarray(d).save(os.path.join(sdir, "depths", "%d.png" % i))


def run(extra):
    buf = io.StringIO()
    t0 = time.time()
    with redirect_stdout(buf):
        T.main(["--train_data_root", root, "--model_dir", os.path.join(root, "m" + str(len(extra))), "--network_mode", "normal",
                "--view_num", "3", "--max_d", "128", "--width", "640", "--height", "480", "--epoch", "1",
                "--max_steps_per_epoch", str(steps), "--snapshot", "100000", "--train_steps_per_val", "100000"] + extra)
    wall = time.time() - t0
    per = [float(x) for x in re.findall(r"\(([0-9.]+) sec/step\)", buf.getvalue())]
    tail = per[len(per) // 3:]                               # past the first steps' one-offs
    return {"steps": len(per), "wall_s": round(wall, 2), "median_ms_per_step": round(1e3 * float(np.median(tail)), 2),
            "mean_ms_per_step": round(1e3 * float(np.mean(tail)), 2)}


print("images stored at %d x %d, %d steps per run" % (SW, SH, steps), flush=True)
for name, extra in (("prefetch (default)", []), ("one generator on the training thread", ["--no_prefetch"]),
                    ("prefetch again", [])):
    print(name, run(extra), flush=True)
