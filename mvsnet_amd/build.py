"""Builds libmvsnet_hip.so (gfx950) in-tree with hipcc.

``python -m mvsnet_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles without a
GPU; the .so is git-ignored but travels to the GPU box with the working tree.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libmvsnet_hip.so")
SOURCES = ["homography.hip", "cost_volume.hip", "conv3d_scalar.hip", "conv3d_mfma.hip", "conv3d_s2_mfma.hip",
           "deconv3d_mfma.hip", "deconv3d_c8.hip", "conv3d_os.hip", "conv3d_out.hip", "conv3d_bf16x3.hip", "conv3d_c8.hip",
           "regnet.hip", "softargmin.hip", "gru.hip", "gru_mfma.hip", "gru_fused.hip", "unet2d.hip", "unet2d_p.hip", "center_images.hip", "multi_tensor.hip",
           "backward.hip", "conv3d_wgrad.hip", "conv3d_c1.hip", "conv3d_k8.hip", "gru_train.hip", "conv2d_wgrad.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("MVS_EXTRA_HIPCC_FLAGS", "").split()      # developer builds (tools/README.md)


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
    headers = [os.path.join(CSRC, "common.h"),
               os.path.join(HERE, "..", "include", "mvsnet_hip.h")]
    headers += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([HIPCC] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB_PATH, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs)
    return LIB_PATH


LAB_LIB_PATH = os.path.join(HERE, "variants", "libmvsnet_lab.so")
LAB_SOURCES = ["cost_volume_lds.hip", "cost_volume_mfma.hip"]


def build_lab(verbose=True):
    """variants/libmvsnet_lab.so: kernel families that were built, are exact, measured no faster than the product's, and
    therefore stay OUT of libmvsnet_hip.so (csrc/lab/, entry points mvs_lab_*; tests/test_gpu_lab.py, `pytest -m lab`)."""
    os.makedirs(os.path.dirname(LAB_LIB_PATH), exist_ok=True)
    lab = os.path.join(CSRC, "lab")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    objs = []
    for s_ in LAB_SOURCES:
        src, obj = os.path.join(lab, s_), os.path.join(lab, s_.replace(".hip", ".o"))
        objs.append(obj)
        if _stale(obj, [src] + deps):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed:\n%s\n%s" % (r.stdout, r.stderr))
    if _stale(LAB_LIB_PATH, objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LAB_LIB_PATH] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LAB_LIB_PATH


if __name__ == "__main__":
    if "--lab" in sys.argv:
        print(build_lab())
    else:
        print(build_library(force="--force" in sys.argv))
