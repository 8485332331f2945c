"""Synthetic on-disk fixtures shared by the tests (session folders in the reference's format)."""
import json
import os

import numpy as np


def make_session(path, n_images=4, h=48, w=64, seed=0):
    """A tiny on-disk session in the reference's format (covisibility.json, images/, cameras/)."""
    from PIL import Image
    rs = np.random.RandomState(seed)
    os.makedirs(os.path.join(path, "images"))
    os.makedirs(os.path.join(path, "cameras"))
    covis = {}
    for i in range(n_images):
        img = rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(path, "images", "%d.jpg" % i), quality=95)
        pose = np.eye(4)
        pose[0, 3] = 0.05 * i                                   # metres
        cam = {"pose": {"matrix": {"%d,%d" % (r, c): float(pose[r, c]) for r in range(4) for c in range(4)}},
               "intrinsics": {"fx": 60.0, "fy": 60.0, "px": w / 2.0, "py": h / 2.0}}
        with open(os.path.join(path, "cameras", "%d.json" % i), "w") as f:
            json.dump(cam, f)
        views = [j for j in range(n_images) if j != i][:2] if i != 3 else []
        covis[str(i)] = {"views": views, "min_depth": 400.0, "max_depth": 900.0}
    with open(os.path.join(path, "covisibility.json"), "w") as f:
        json.dump(covis, f)
    return path


def add_depths(session, n_images=4, h=48, w=64, seed=0):
    from PIL import Image
    rs = np.random.RandomState(seed + 100)
    os.makedirs(os.path.join(session, "depths"))
    for i in range(n_images):
        d = rs.randint(300, 1000, size=(h, w)).astype(np.uint16)           # some values fall outside (400, 900]
        Image.fromarray(d).save(os.path.join(session, "depths", "%d.png" % i))


def run_ranks(code, env, world=2, timeout=300):
    """Start `world` copies of `python -c code` (RANK / LOCAL_RANK set per copy) and wait for all of them.  Output goes
    to temporary FILES, not pipes: a rank that fills a pipe while the parent is still draining its peer's would block
    inside a collective the peer is waiting in.  Returns [(returncode, stdout, stderr)] by rank."""
    import subprocess
    import sys
    import tempfile
    import time
    files, procs = [], []
    for r in range(world):
        fo, fe = tempfile.TemporaryFile("w+"), tempfile.TemporaryFile("w+")
        files.append((fo, fe))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=fo, stderr=fe, text=True))
    deadline = time.time() + timeout
    try:
        for p in procs:
            p.wait(timeout=max(1.0, deadline - time.time()))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    outs = []
    for p, (fo, fe) in zip(procs, files):
        fo.seek(0); fe.seek(0)
        outs.append((p.returncode, fo.read(), fe.read()))
        fo.close(); fe.close()
    return outs
