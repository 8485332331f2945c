#!/usr/bin/env python
"""bench.py -- depth maps/sec of the plane-sweep hot path on N MI355X (one process per GPU).

    python bench.py --gpus N --steps 20 --warmup 5          (N > 1: starts the N one-GPU ranks itself, parent GPU-less)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one features->depth pass (one call of mvs_depth_from_features_f32) of the metric workload (BASELINE.json: N=5 views, D=192 planes,
160x128 feature maps, C=32, 3D-CNN regulariser): homography transforms -> fused warp+variance
cost volume -> RegNetUS0 -> soft-argmin + probability map, inputs (feature maps, cameras, weights)
resident in HBM.  Reference views are independent, so ranks shard them with no
data-path collective ("scaling": "weak"); value = depth maps of all ranks / max-over-ranks time.
Rank 0 prints ONE JSON line.  After the timed region (never inside it) the same process adds informative records
to that line: the median of repeated blocks of the same K steps, in-pipeline times of every RegNetUS0 layer
(`roofline_kernels`), and -- single GPU only -- configs[1] (c2, 288x216, D=192) and configs[2] (c3, ConvGRU sweep,
400x300, D=256) with their rates and their distance from the committed float64 fixtures (tests/golden/full_*.npz).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32-input MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2)
MFMA_BF16_PEAK_TFLOPS = 2500.0


def algorithmic_work(view_num, D, H, W, C, base=8):
    """BASELINE.md section 2 / SURVEY 8d: bytes of the warp+variance kernel, FLOPs of RegNetUS0,
    bytes of the soft-argmin kernel, per depth map."""
    vox = D * H * W
    warp_bytes = view_num * H * W * C * 4 + vox * C * 4
    # MACs per full-resolution voxel (mvsnetworks.py:122-158), channels in units of base b, cin = C
    b = base
    mac = 27 * (C * b                      # 0_1   on V0
                + C * 2 * b / 8            # 1_0   on V0/8
                + 2 * b * 2 * b / 8        # 1_1
                + 2 * b * 4 * b / 64       # 2_0
                + 4 * b * 4 * b / 64       # 2_1
                + 4 * b * 8 * b / 512      # 3_0
                + 8 * b * 8 * b / 512      # 3_1
                + 8 * b * 4 * b / 512      # 4_0 (27 taps per INPUT voxel of V0/512)
                + 4 * b * 2 * b / 64       # 5_0
                + 2 * b * b / 8            # 6_0
                + b * 1)                   # 6_2
    conv_flops = 2.0 * mac * vox
    soft_bytes = vox * 4 + 2 * H * W * 4
    return warp_bytes, conv_flops, soft_bytes


# RegNetUS0 layers in the library's weight order (mvs_profile_layers_ms): name, MACs per FULL-resolution voxel
# in units of base^2 * 27 (see algorithmic_work), i.e. flops = 2 * 27 * factor * voxels
LAYERS = ["3dconv1_0", "3dconv2_0", "3dconv3_0", "3dconv0_1", "3dconv1_1", "3dconv2_1", "3dconv3_1", "3dconv4_0",
          "3dconv5_0", "3dconv6_0", "3dconv6_2"]


def layer_flops(C, b, vox):
    mac = {"3dconv0_1": C * b, "3dconv1_0": C * 2 * b / 8, "3dconv1_1": 2 * b * 2 * b / 8, "3dconv2_0": 2 * b * 4 * b / 64,
           "3dconv2_1": 4 * b * 4 * b / 64, "3dconv3_0": 4 * b * 8 * b / 512, "3dconv3_1": 8 * b * 8 * b / 512,
           "3dconv4_0": 8 * b * 4 * b / 512, "3dconv5_0": 4 * b * 2 * b / 64, "3dconv6_0": 2 * b * b / 8, "3dconv6_2": b * 1}
    return {k: 2.0 * 27 * v * vox for k, v in mac.items()}


def fixture(name):
    """tests/golden/full_<name>.npz (float64 CPU oracle outputs, generator committed beside them) or None."""
    path = os.path.join(ROOT, "tests", "golden", "full_%s.npz" % name)
    return np.load(path) if os.path.exists(path) else None


def pmc_traffic():
    """HBM bytes per depth map from the committed rocprofv3 PMC summary (profiles/rNN_pmc.json,
    produced by profiles/collect_rNN.sh in separate --pmc passes; FETCH_SIZE x2 as
    MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950, WRITE_SIZE exact).
    Returns {"warp": bytes, "conv": bytes, "soft": bytes} or {} when no summary is present."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    if not files:
        return {}
    data = json.load(open(files[-1]))
    out = {"warp": 0.0, "conv": 0.0, "soft": 0.0, "pair": 0.0, "source": os.path.basename(files[-1])}
    for name, d in data.items():
        if "cost_volume" in name and "SQ_INSTS_VALU" in d:      # what bounds the warp + variance kernel (not HBM): its SQ / TA counters
            out["warp_counters"] = {k: d[k] for k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY",
                                                      "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "TA_BUSY_avr", "GRBM_GUI_ACTIVE") if k in d}
        if "hbm_write_bytes" not in d:
            continue
        b = (d["hbm_read_bytes_corrected_x2"] + d["hbm_write_bytes"]) * d.get("launches_per_depth_map", 1.0)
        if "::conv3d_c8_kernel" in name:                  # not deconv3d_c8_kernel
            out["pair"] += d["hbm_read_bytes_corrected_x2"] + d["hbm_write_bytes"]       # per launch
        if "cost_volume" in name:
            out["warp"] += b
        elif "conv3d" in name or "deconv3d" in name:
            out["conv"] += b
        elif "softargmin" in name:
            out["soft"] += b
    return out


def gru_pmc_per_plane():
    """Matrix-pipe busy and vector-active microseconds per plane of the 4-view recurrent sweep from the committed SQ counter summary
    (profiles/rNN_gru_pmc_B4.txt, tools/gru_pmc.sh: its TOTAL line), sourced the way roofline.traffic is; {} without a file."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gru_pmc_B4.txt")))
    if not files:
        return {}
    m = None
    for line in open(files[-1]):
        m = re.search(r"TOTAL per plane: active_any ([\d.]+) us, valu ([\d.]+) us, mfma_busy ([\d.]+) us", line) or m
    if not m:
        return {}
    return {"mfma_busy_us": float(m.group(3)), "valu_active_us": float(m.group(2)), "active_any_us": float(m.group(1)),
            "source": os.path.basename(files[-1]) + " (SQ_VALU_MFMA_BUSY_CYCLES / SQ_ACTIVE_INST_VALU per SIMD at 2.4 GHz, summed over the sweep's kernels)"}


def cpu_baseline(workload, rp, budget_s=20.0, gpu_depth=None, max_maps=8):
    """CPU restatement (oracle/torch_restatement.py, fp32, all host cores) timed on a bounded
    sample of the same workload: whole depth maps until ~budget_s have elapsed (at least one)."""
    from oracle import torch_restatement as TR       # measurement only; never on the product path
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:       # cgroup CPU quota (the GPU box gives one GPU's share of the host cores)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    cores = min(cores, int(os.environ.get("MVS_CPU_THREADS", "16")))   # 1-GPU share of the host
    torch.set_num_threads(cores)
    w = workload
    n_done, t0 = 0, time.perf_counter()
    while True:
        cpu_depth, _ = TR.inference_mem_from_features(w.features, w.cams, w.depth_num, w.depth_start, w.depth_interval, rp)
        n_done += 1
        el = time.perf_counter() - t0
        if el > budget_s or n_done >= max_maps:
            break
    extra = {}
    if gpu_depth is not None:      # the metric's "abs-rel vs ref" at full size, against the CPU restatement
        extra["abs_rel_gpu_vs_cpu"] = float(np.mean(np.abs(np.squeeze(gpu_depth) - cpu_depth) / cpu_depth))
    return {**extra, "value": n_done / el, "unit": "depth maps/s", "cores": cores, "kind": "port",
            "sample": "%d whole depth map(s) of workload %s (features->depth, torch-CPU fp32 restatement "
                      "of the reference; TensorFlow reference not runnable offline) in %.1f s" % (n_done, w.name, el)}


def rccl_version():
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:
        return None


def timed_block(step, steps):
    """One block of `steps` steps, bracketed like the timed region (single process); returns seconds."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i, False)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def extra_config_3dcnn(name, dev, steps=10):
    """A second 3D-CNN configuration (c2 = BASELINE.json configs[1]) on this GPU: rate + distance from its fixture.
    Weights are the fixture's (seed 1, random BatchNorm affine); speed does not depend on their values."""
    from mvsnet_amd import synthetic as S
    from mvsnet_amd.model import DepthPlan, MVSNetWeights
    w = S.make_workload(name)
    weights = MVSNetWeights.from_numpy("normal", regnet=S.make_regnet_params("normal", seed=1, random_affine=True), device=dev)
    feats, cams = torch.as_tensor(w.features).to(dev), torch.as_tensor(w.cams).to(dev)
    plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, weights, "3DCNN", dev)

    def step(i, _r):
        plan.run_depth(feats, cams, w.depth_start, w.depth_interval, w.depth_end, False)
    for i in range(3):
        step(i, False)
    el = timed_block(step, steps)
    out = {"workload": "%s: N=%d, D=%d, %dx%d, 3D-CNN" % (name, w.view_num, w.depth_num, w.width, w.height),
           "depth_maps_per_s": steps / el, "ms_per_depth_map": el / steps * 1e3, "steps": steps}
    g = fixture(name)
    if g is not None:
        d = plan.depth.cpu().numpy().astype(np.float64)
        p = plan.prob.cpu().numpy().astype(np.float64)
        out["abs_rel_vs_fixture"] = float(np.mean(np.abs(d - g["depth"]) / g["depth"]))
        out["prob_mismatch_vs_fixture"] = float((np.abs(p - g["prob"]) > 1e-3).mean())
        out["fixture"] = "tests/golden/full_%s.npz (float64 CPU oracle; float32 CPU restatement lands at %.2e)" % (
            name, float(g["f32_cpu_abs_rel"]))
    del plan
    torch.cuda.empty_cache()
    return out


def extra_config_gru(name, dev, steps=5, views=1):
    """configs[2] (c3): the ConvGRU + winner-take-all sweep: rate, time per plane, agreement with its fixture.
    views > 1: that many independent reference views share the sweep's launches (mvs_gru_wta_batch_f32) -- how configuration 3
    shards, ~135 reference views per GPU; view 0 is the fixture's reference view, the others differ in their features."""
    from mvsnet_amd import synthetic as S
    from mvsnet_amd.model import DepthPlan, MVSNetWeights, wta_depth_values
    w = S.make_workload(name)
    gp = S.make_gru_params("normal", seed=2, in_channels=w.channels, random_affine=True)
    weights = MVSNetWeights.from_numpy("normal", gru=gp, device=dev)
    cams = torch.as_tensor(w.cams).to(dev)
    feats = [torch.as_tensor(w.features).to(dev)]
    feats += [torch.as_tensor(S.make_features(w.view_num, w.height, w.width, w.channels, seed=v)).to(dev) for v in range(1, views)]
    dv = wta_depth_values(w.depth_num, w.depth_start, w.depth_end, False)
    plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, weights, "GRU", dev, views=views)

    def step(i, _r):
        for v in range(views):
            plan.set_cameras(cams, w.depth_start, w.depth_interval, w.depth_end, False, view=v)
        plan.run_gru_batch(feats, [dv] * views)
    for i in range(2):
        step(i, False)
    el = timed_block(step, steps)
    maps = steps * views
    flops = 2.0 * 23238 * w.depth_num * w.height * w.width          # 23 238 MAC per pixel and plane
    out = {"workload": "%s: N=%d, D=%d, %dx%d, ConvGRU sweep + winner-take-all, %d reference view(s) per sweep" % (
               name, w.view_num, w.depth_num, w.width, w.height, views),
           "depth_maps_per_s": maps / el, "ms_per_depth_map": el / maps * 1e3, "ms_per_sweep": el / steps * 1e3,
           "ms_per_plane": el / steps / w.depth_num * 1e3, "achieved_tflops": flops * maps / el / 1e12, "peak_tflops": MFMA_F32_PEAK_TFLOPS,
           "frac_of_fp32_mfma_peak": flops * maps / el / 1e12 / MFMA_F32_PEAK_TFLOPS, "sweeps": steps, "views_per_sweep": views,
           "measured": "in this process, after the 3D-CNN records (not a child process)"}
    g = fixture(name)
    if g is not None:
        d = plan.depth_v[0].cpu().numpy()
        p = plan.prob_v[0].cpu().numpy().astype(np.float64)
        same = np.abs(d - g["depth"]) <= 1e-6 * g["depth"]
        out["plane_agreement_vs_fixture"] = float(same.mean())
        out["prob_rel_max_on_agreeing_pixels"] = float(np.max(np.abs(p[same] - g["prob"][same]) / g["prob"][same])) if same.any() else None
        out["fixture"] = "tests/golden/full_%s.npz (float64 CPU oracle; float32 CPU restatement: agreement %.5f, prob rel max %.1e)" % (
            name, float(g["f32_cpu_plane_agreement"]), float(g["f32_cpu_prob_rel"]))
    del plan, feats
    torch.cuda.empty_cache()
    return out


def gru_formulations(dev, name="c3", steps=3):
    """The same sweep (one reference view) under each formulation of the recurrent path: the fused two-launches-per-plane pipeline
    (the default), the round-4 wavefront over a stream set with cell 1's x-part hoisted, and the fused sweep replayed from a hipGraph."""
    from mvsnet_amd import _lib, synthetic as S
    from mvsnet_amd.model import DepthPlan, MVSNetWeights, wta_depth_values
    w = S.make_workload(name)
    gp = S.make_gru_params("normal", seed=2, in_channels=w.channels, random_affine=True)
    weights = MVSNetWeights.from_numpy("normal", gru=gp, device=dev)
    cams = torch.as_tensor(w.cams).to(dev)
    feats = torch.as_tensor(w.features).to(dev)
    dv = wta_depth_values(w.depth_num, w.depth_start, w.depth_end, False)
    out = {}
    lib = _lib.load()
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, weights, "GRU", dev)
        plan.set_cameras(cams, w.depth_start, w.depth_interval, w.depth_end, False)

        def timed(fn):
            fn(); st.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            st.synchronize()
            return (time.perf_counter() - t0) / steps * 1e3
        try:
            for form, tag in ((3, "fused (default)"), (1, "wavefront, hoisted x-part (round 4)")):
                _lib.check(lib.mvs_gru_set_formulation(form), "mvs_gru_set_formulation")
                out[tag] = timed(lambda: plan.run_gru(feats, dv))
        finally:
            _lib.check(lib.mvs_gru_set_formulation(0), "mvs_gru_set_formulation")
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                plan.run_gru(feats, dv)
            out["fused, replayed from a hipGraph"] = timed(g.replay)
        except Exception as e:                             # informative
            out["fused, replayed from a hipGraph"] = repr(e)[:200]
    plan.close()
    return out


def session_record(dev, kernel_rate, n_images=48, regularization="3DCNN", procs_per_gpu=3, width=640, height=512):
    """End-to-end throughput of the reference's own loop (mvsnet/inference.py:105-119: load a cluster -> run the graph -> write
    the outputs, 'Depth inference ... sec/step') on a synthetic on-disk session of the metric's shape: `n_images` JPEGs of
    640x512 with cameras and a covisibility file (mvsnet_amd.synthetic.write_session), view_num 5, max_d 192 -> 160x128 feature
    maps -- configuration 4's per-GPU work, one rank.  compute_depth_maps decodes / resizes / crops in worker processes
    (mvsnet_amd/host_pool.py), runs the UNetDS2GN towers (HIP library, per-image feature cache) and the hot path in this
    process, and the workers write <idx>_init.pfm / _prob.pfm / PNGs / JPG / camera per reference view.  Two passes over the same session: the first pays the lazy one-offs (plans,
    code objects, pinned buffers), the second is reported."""
    import shutil
    import tempfile
    from mvsnet_amd import synthetic as S
    from mvsnet_amd.inference import build_weights, compute_depth_maps
    from mvsnet_amd.predictlib import InferenceConfig
    root = tempfile.mkdtemp(prefix="mvs_session_")
    try:
        S.write_session(root, n_images=n_images, height=height, width=width, view_num=5, depth_num=192)
        cfg = InferenceConfig(input_dir=root, view_num=5, max_d=192, width=width, height=height, sample_scale=0.25,
                              regularization=regularization)
        weights = build_weights(cfg, dev)
        out = {}
        for attempt in range(2):
            tm = {}
            cfg.output_dir = os.path.join(root, "out%d" % attempt)
            n = compute_depth_maps(root, cfg, weights, dev, timings=tm)
            out = {"session": "%d JPEGs %dx%d + cameras, N=5, D=192, %s; %d reference views, second pass" % (n_images, width, height, regularization, n),
                   "session_depth_maps_per_s": n / tm["wall"], "sec_per_step": tm["wall"] / max(n, 1),
                   "fraction_of_kernel_only_rate": (n / tm["wall"]) / kernel_rate if kernel_rate else None,
                   "files_written": len(os.listdir(cfg.output_dir)),
                   "host_workers": tm.get("host_workers"),
                   "host_cpu_ms_per_depth_map": (1e3 * tm["host_cpu"] / max(n, 1)) if tm.get("host_cpu") is not None else None,
                   "host_cores": os.cpu_count(),
                   "breakdown_ms_per_depth_map": {
                       "decode_resize_crop (worker processes, summed)": 1e3 * tm["load"] / max(n, 1),
                       "main_thread_waiting_for_loaders": 1e3 * tm["wait_load"] / max(n, 1),
                       "main_thread_enqueueing_gpu_work": 1e3 * tm["host_gpu_submit"] / max(n, 1),
                       # its parts (round 6): upload + tower launches / feature stack / hot-path launches / result hand-over, where the
                       # last one INCLUDES the wait for a free result slot (8 in flight): time the GPU is behind this thread, not host work
                       "main_thread_enqueueing_parts": {k_: 1e3 * tm.get(k_, 0.0) / max(n, 1) for k_ in ("submit_towers", "submit_features", "submit_depth", "submit_finish")},
                       "gpu_busy_fraction_of_wall": (tm["towers"] + tm["hot_path"] + tm["d2h"]) / tm["wall"],
                       "h2d_and_towers (GPU)": 1e3 * tm["towers"] / max(n, 1),
                       "hot_path (GPU)": 1e3 * tm["hot_path"] / max(n, 1),
                       "d2h (GPU, to pinned buffers)": 1e3 * tm["d2h"] / max(n, 1),
                       "file_writes (worker processes, summed)": 1e3 * tm["write"] / max(n, 1)},
                   "loader_threads": tm["loader_threads"], "writer_threads": tm["writer_threads"]}
        # the same session with several worker PROCESSES sharing this GPU (python -m mvsnet_amd.inference --procs_per_gpu P): the
        # one-process loop above is bound by its Python main thread, not by the GPU
        if procs_per_gpu > 1:
            import subprocess
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
            try:
                r = subprocess.run([sys.executable, "-m", "mvsnet_amd.inference", "--input_dir", root, "--output_dir", os.path.join(root, "outp"),
                                    "--view_num", "5", "--max_d", "192", "--width", "640", "--height", "512", "--regularization", regularization,
                                    "--procs_per_gpu", str(procs_per_gpu), "--passes", "6"], capture_output=True, text=True, timeout=240,
                                   env=env, cwd=ROOT)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                out["processes_sharing_the_gpu"] = json.loads(line[-1]) if line else {"error": (r.stderr or r.stdout)[-300:]}
            except Exception as e:
                out["processes_sharing_the_gpu"] = {"error": repr(e)[:300]}
        return out
    except Exception as e:                                  # informative record: never fail the bench line over it
        return {"error": repr(e)[:400]}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def training_record(dev, iters=5, configs=(("3dcnn_d128_config5", "3DCNN", 128),)):
    """SURVEY 8f f4 / BASELINE configs[4]'s per-GPU work: one training step (images -> towers -> hot path forward + backward ->
    optimiser, mvsnet/train.py:412-445) at train.py's default size (3 views x 640x480, D = 192, batch 1) for both regularisers
    and at configuration 5's depth count (D = 128); milliseconds per step after two warm-up steps."""
    from mvsnet_amd import synthetic as S, train as T
    out = {}
    try:
        for tag, reg, D in configs:
            N, H, W = 3, 480, 640
            images = S.make_images(N, H, W)
            cams = S.make_cams(N, H // 4, W // 4, D)
            start, interval = float(cams[0, 1, 3, 0]), float(cams[0, 1, 3, 1])
            gt = np.full((H // 4, W // 4, 1), start + interval * D * 0.5, np.float32)
            tr = T.Trainer("normal", dev, regularization=reg)
            # inputs resident in HBM when the timed region starts (bench contract; rounds 1-5 handed numpy arrays over and timed a
            # blocking pageable copy of the 11 MB image batch inside every step: ~2 ms of a host-bound 12 ms step).  The cameras
            # stay a host array: Trainer.loss reads the depth range from them.
            images_np = images
            images, gt = torch.as_tensor(images).to(dev), torch.as_tensor(gt).to(dev)
            for _ in range(2):
                tr.train_step(images, cams, gt, D)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                tr.train_step(images, cams, gt, D)
            torch.cuda.synchronize()
            ms_step = (time.perf_counter() - t0) / iters * 1e3
            out[tag] = {"ms_per_step": ms_step, "views": N, "image": "%dx%d" % (W, H), "depth_planes": D}
            if reg == "3DCNN":
                out[tag]["roofline"] = training_roofline(tr, images_np, cams, gt, N, H, W, D, ms_step, iters)
                out[tag]["inputs"] = "images and ground truth resident on the device (uploaded once); the launching thread and the GPU bound the step together (541 launches, 9.2 ms of kernel time per step: profiles/r06_train_step_kernels_after.txt)"
            del tr
            torch.cuda.empty_cache()
    except Exception as e:                                  # informative record: never fail the bench line over it
        out["error"] = repr(e)[:300]
    return out


def training_from_disk_record(steps=30, n_images=16, src=(1600, 1200)):
    """SURVEY 8f f4, the loop AROUND the step: `python -m mvsnet_amd.train` on an on-disk dataset in the reference's layout,
    images stored at DTU's 1600 x 1200 and decoded / rescaled / cropped to configuration 5's 640 x 480 by the generator
    (mvs_cluster.py:178-192), D = 128: seconds per step as the loop prints them, with the input pipeline (the next clusters
    prepared by worker processes, uint8 upload, standardisation on the device: the reference's tf.data parallel_interleave +
    prefetch, train.py:208-247) and with one generator on the training thread."""
    import contextlib, io, re, tempfile
    out = {"images_stored_at": "%dx%d" % src, "trained_at": "3 views x 640x480, D = 128", "steps_per_run": steps}
    try:
        from PIL import Image
        from mvsnet_amd import synthetic as S, train as T
        root = tempfile.mkdtemp()
        for mode in ("train", "val"):
            sdir = os.path.join(root, mode, "s0")
            S.write_session(sdir, n_images=n_images if mode == "train" else 4, height=src[1], width=src[0], view_num=3, depth_num=128)
            os.makedirs(os.path.join(sdir, "depths"))
            rs = np.random.RandomState(0)
            for i in range(n_images if mode == "train" else 4):
                d = (S.PIVOT_DEPTH + 20.0 * rs.standard_normal((src[1], src[0]))).clip(1, 65535).astype(np.uint16)
                Image.fromarray(d).save(os.path.join(sdir, "depths", "%d.png" % i))
        for tag, extra in (("warm-up", []), ("input_pipeline", []), ("one_generator_on_the_training_thread", ["--no_prefetch"])):
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                T.main(["--train_data_root", root, "--model_dir", os.path.join(root, "model"), "--network_mode", "normal",
                        "--view_num", "3", "--max_d", "128", "--width", "640", "--height", "480", "--epoch", "4",
                        "--max_steps_per_epoch", str(steps // 4 + 1), "--snapshot", "1000000", "--train_steps_per_val", "1000000"] + extra)
            per = [float(x) for x in re.findall(r"\(([0-9.]+) sec/step\)", buf.getvalue())]
            if tag != "warm-up":
                out[tag] = {"median_ms_per_step": 1e3 * float(np.median(per[len(per) // 3:])), "steps": len(per)}
        import shutil
        shutil.rmtree(root, ignore_errors=True)
    except (Exception, SystemExit) as e:                    # informative record: never fail the bench line over it
        out["error"] = repr(e)[:300]
    return out


def training_roofline(tr, images, cams, gt, N, H, W, D, ms_step, iters):
    """Roofline of one training step: algorithmic FLOPs of forward + backward -- the backward of a
    convolution is a data-gradient and a weight-gradient convolution of the forward's size, so 3x the forward in all (2x for a
    network's first layer, which needs no data gradient) -- for the hot path and the towers separately, each part timed ALONE
    in steady state (so the parts need not add up to the step: the step also holds the loss, the optimiser and Python glue),
    against the fp32 MFMA peak, and which of it runs on kernels of this library."""
    from mvsnet_amd import backward as B
    from mvsnet_amd.feature_net import unet_macs
    from mvsnet_amd.feature_net_train import hip_towers
    from mvsnet_amd.homography_warping import homography_transforms
    dev = tr.params.data.device
    Hf, Wf = H // 4, W // 4
    hot_fwd = 2.0 * 11448 * D * Hf * Wf                      # 11 448 MAC per voxel (warp + variance: ~54 FLOP per voxel-channel, not counted)
    tow_fwd = 2.0 * unet_macs(H, W) * N
    res = {"algorithmic_gflop": {"hot_path_forward": hot_fwd / 1e9, "hot_path_forward_backward": 3 * hot_fwd / 1e9,
                                 "towers_forward": tow_fwd / 1e9, "towers_forward_backward": 3 * tow_fwd / 1e9},
           "peak_tflops": MFMA_F32_PEAK_TFLOPS}

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3
    try:
        img_t = torch.as_tensor(images).to(dev)
        unet_p = tr.params.group("unet")

        def towers():                                          # as the step runs them: parameter gradients added into the flat buffer
            hip_towers(img_t, unet_p, accumulate_into_grads=True).sum().backward()
        ms_tow = timed(towers)
        feats = hip_towers(img_t, unet_p).detach().requires_grad_(True)
        t8 = homography_transforms(torch.as_tensor(cams).to(dev), D, float(cams[0, 1, 3, 0]), float(cams[0, 1, 3, 1]))
        reg_p = tr.params.group("regnet")
        g1 = torch.ones(Hf, Wf, device=dev)

        def hot():
            d, _ = B.plane_sweep_depth(feats, t8, float(cams[0, 1, 3, 0]), float(cams[0, 1, 3, 1]), reg_p)
            d.reshape(Hf, Wf).backward(g1)
        ms_hot = timed(hot)
        res.update({
            "hot_path_forward_backward": {"ms": ms_hot, "achieved_tflops": 3 * hot_fwd / ms_hot / 1e9, "frac_of_fp32_mfma_peak": 3 * hot_fwd / ms_hot / 1e9 / MFMA_F32_PEAK_TFLOPS,
                                          "kernels": "all HIP (csrc/backward.hip, conv3d_wgrad.hip, the forward's MFMA kernels for the data gradients)"},
            "towers_forward_backward": {"ms": ms_tow, "achieved_tflops": 3 * tow_fwd / ms_tow / 1e9, "frac_of_fp32_mfma_peak": 3 * tow_fwd / ms_tow / 1e9 / MFMA_F32_PEAK_TFLOPS,
                                        "kernels": "forward + GroupNorm backward + weight gradients of the 16/32-channel layers: HIP (unet2d.hip, conv2d_wgrad.hip); "
                                                   "data gradients and the other weight gradients: ATen / MIOpen (feature_net_train.py)"},
            "whole_step": {"ms": ms_step, "achieved_tflops": 3 * (hot_fwd + tow_fwd) / ms_step / 1e9,
                           "frac_of_fp32_mfma_peak": 3 * (hot_fwd + tow_fwd) / ms_step / 1e9 / MFMA_F32_PEAK_TFLOPS},
            "share_of_step": {"hot_path": ms_hot / ms_step, "towers": ms_tow / ms_step}})
    except Exception as e:                                  # informative: the step time above stands
        res["error"] = repr(e)[:300]
    return res


def gru_production_order(dev, n_views=8, views_per_sweep=4):
    """configs[2] in the production order inside ONE process: for each of `n_views` reference views the
    UNetDS2GN towers on the HIP library (5 images of 1600x1200 -> 400x300x32 feature maps), then the recurrent sweep, several
    views per sweep -- towers and sweeps alternate on the same stream, as mvsnet_amd.inference runs them."""
    from mvsnet_amd import synthetic as S
    from mvsnet_amd.feature_net_hip import HipUNetDS2GN
    from mvsnet_amd.model import DepthPlan, MVSNetWeights, wta_depth_values
    try:
        w = S.make_workload("c3")
        weights = MVSNetWeights.from_numpy("normal", gru=S.make_gru_params("normal", seed=2, in_channels=w.channels, random_affine=True), device=dev)
        net = HipUNetDS2GN(S.make_unet_params("normal", seed=3), dev)
        cams = torch.as_tensor(w.cams).to(dev)
        imgs = [torch.as_tensor(S.make_images(w.view_num, 4 * w.height, 4 * w.width, seed=v)).to(dev) for v in range(2)]
        dv = wta_depth_values(w.depth_num, w.depth_start, w.depth_end, False)
        plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, weights, "GRU", dev, views=views_per_sweep)

        ts = torch.cuda.Stream(dev)
        main_s = torch.cuda.current_stream(dev)

        def run(own_stream=False):
            for v0 in range(0, n_views, views_per_sweep):
                feats = []
                for v in range(views_per_sweep):
                    if own_stream:                           # round 6: the towers on a stream of their own, beside the previous sweep
                        with torch.cuda.stream(ts):
                            f_ = net(imgs[(v0 + v) % 2])
                        f_.record_stream(main_s)
                        feats.append(f_)
                    else:
                        feats.append(net(imgs[(v0 + v) % 2]))
                    plan.set_cameras(cams, w.depth_start, w.depth_interval, w.depth_end, False, view=v)
                if own_stream:
                    main_s.wait_event(ts.record_event())
                plan.run_gru_batch(feats, [dv] * views_per_sweep)
        res = {}
        for tag, own in (("", False), ("_towers_on_their_own_stream", True)):
            run(own)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(own)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            res["depth_maps_per_s" + tag] = n_views / el
            res["ms_per_depth_map" + tag] = el / n_views * 1e3
        res["workload"] = ("c3 from IMAGES: %d reference views, each 5 images of %dx%d -> towers (HIP library) -> ConvGRU sweep, %d views per sweep, one process"
                           % (n_views, 4 * w.width, 4 * w.height, views_per_sweep))
        return res
    except Exception as e:                                  # informative record: never fail the bench line over it
        return {"error": repr(e)[:300]}


def gru_config_in_child(name, steps=5):
    """The same recurrent configuration in a fresh process (the GPU is idle here), for comparison with the in-process record:
    round 2 measured 44 ms in this process against 23 ms in a fresh one (hardware-queue / compute-pipe interference, fixed in
    round 3 by the calibrated stream layout of csrc/gru.hip); the two must now agree."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--regularization", "GRU", "--workload", name,
                            "--steps", str(steps)], capture_output=True, text=True, timeout=300)
    except Exception as e:                              # the record is informative: never fail the bench line over it
        return {"error": repr(e)[:300]}
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            d = json.loads(line)
            return {"ms_per_depth_map": d.get("ms_per_depth_map"), "depth_maps_per_s": d.get("depth_maps_per_s")}
    return {"error": (r.stderr or r.stdout)[-400:]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="M", help="M (metric), c1, c2, small, toy")
    ap.add_argument("--network-mode", default="normal")
    ap.add_argument("--conv-impl", default="auto", choices=["auto", "scalar", "mfma", "bf16x3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("MVS_BENCH_STREAMS", "1")),
                    help="independent depth maps in flight per GPU (one plan + HIP stream each).  Keep 1: since the round-4 schedules "
                         "fill every CU with one depth map, two in flight get in each other's way (877 against 929 depth maps/s)")
    ap.add_argument("--regularization", default="3DCNN", choices=["3DCNN", "GRU"],
                    help="GRU = R-MVSNet recurrent sweep (config 3); reported as an extra, not the metric")
    ap.add_argument("--gru-views", type=int, default=1, help="reference views per recurrent sweep (--regularization GRU)")
    ap.add_argument("--extractor", choices=("hip", "torch"), default="hip", help="2D towers for --with-images")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the informative records (repeated blocks, two-stream pass, c2 / c3 configurations)")
    ap.add_argument("--blocks", type=int, default=5, help="repeated blocks of --steps steps after the timed region")
    ap.add_argument("--training-all", action="store_true", help="training record for D = 192 and the recurrent model too")
    ap.add_argument("--with-images", action="store_true",
                    help="also time images->depth (adds the PyTorch UNetDS2GN towers)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly (`python bench.py --gpus N`, as the driver starts --gpus 1): this process stays GPU-less and
        # starts the N one-GPU ranks itself; rank 0's JSON line goes to the inherited stdout, exit code = the ranks'
        from mvsnet_amd.shard import launch_ranks
        raise SystemExit(launch_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # one rank per GPU; MVS_DIST_BACKEND=gloo + MVS_ALLOW_SHARED_GPU=1 rehearse the multi-process path on a box with
    # fewer GPUs than ranks (ranks then share devices)
    ndev = torch.cuda.device_count()
    backend = os.environ.get("MVS_DIST_BACKEND", "nccl")
    if world > 1 and ndev < world and (backend == "nccl" or not os.environ.get("MVS_ALLOW_SHARED_GPU")):
        raise SystemExit("bench.py --gpus %d needs %d GPUs on this node, found %d (one rank per GPU; MVS_DIST_BACKEND=gloo "
                         "MVS_ALLOW_SHARED_GPU=1 rehearses the multi-process path on fewer)" % (args.gpus, world, ndev))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    if os.environ.get("MVS_BENCH_FAIL_RANK") == str(rank):     # test hook: one rank of N dies before the rendezvous -> the
        raise SystemExit("rank %d: MVS_BENCH_FAIL_RANK" % rank)  # launcher tears the others down and the exit code is non-zero
    dev_index = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # MVS_BENCH_FORCE_DIST=1: a world of ONE rank still goes through the process group (RCCL on a one-GPU box: the barriers,
    # the MAX all-reduce and the gathers of the N > 1 path on the real backend; tests/test_gpu_pipeline.py)
    if world > 1 or os.environ.get("MVS_BENCH_FORCE_DIST"):
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from mvsnet_amd import _lib, synthetic as S
    from mvsnet_amd.model import DepthPlan, MVSNetWeights
    _lib.load()
    _lib.set_conv_impl(args.conv_impl)

    # each rank owns a different synthetic reference view (seed = rank): shard by ref view
    w = S.make_workload(args.workload, args.network_mode, seed=rank)
    rp = S.make_regnet_params(args.network_mode, seed=1)
    weights = MVSNetWeights.from_numpy(args.network_mode, regnet=rp, device=dev)
    feats = torch.as_tensor(w.features).to(dev)
    cams = torch.as_tensor(w.cams).to(dev)
    if args.regularization == "GRU":
        # the recurrent sweep alone in this process (also how the 3D-CNN run obtains its `config_c3_gru` record)
        rec = extra_config_gru(args.workload if args.workload != "M" else "c3", dev, args.steps, max(1, args.gru_views))
        print(json.dumps({"metric": "depth maps/sec (GRU regulariser)", "value": rec["depth_maps_per_s"], "unit": "depth maps/s",
                          "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": rec["ms_per_depth_map"],
                          "dtype": "f32", "data": "synthetic", **rec}), flush=True)
        return
    n_streams = max(1, args.streams)
    plans = [DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, weights, "3DCNN", dev)
             for _ in range(n_streams)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    plan = plans[0]
    end = w.depth_end
    from mvsnet_amd.model import cost_volume, regnet_us0, softargmin_prob

    ev = lambda: torch.cuda.Event(enable_timing=True)
    marks = []

    def step(i, record):
        plan = plans[i % n_streams]
        with torch.cuda.stream(streams[i % n_streams]):
            # the product's entry: homographies -> cost volume -> RegNetUS0 -> soft-argmin as ONE library call.  (`record`: the
            # library brackets its three stages with HIP events -- mvs_profile_stages -- for `roofline_kernels`: the split of THIS
            # path, not of a separate pass through the per-stage Python entries as in rounds 1-5)
            plan.run_depth(feats, cams, w.depth_start, w.depth_interval, end, False)

    torch.cuda.synchronize()
    # untimed: the requested warm-up steps, topped up to 10 launches so that lazy one-off work (code-object
    # load, LDS-size attributes) never lands in the timed region, then more of the same steps until 0.25 s have passed:
    # the first block after an idle period measured 1-1.5 % below every later block of the same steps (clock ramp)
    for i in range(max(args.warmup, 10)):
        step(i, False)
    torch.cuda.synchronize()
    t_prime = time.perf_counter()
    while time.perf_counter() - t_prime < 0.25:
        for i in range(10):
            step(i, False)
        torch.cuda.synchronize()
    # the library brackets the dominant kernel (fused 3dconv0_1 + 3dconv1_0 pass) of every step of
    # the timed region with HIP events on the stream it is launched on
    lib = _lib.load()
    _lib.check(lib.mvs_profile_dominant(1), "mvs_profile_dominant")
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, False)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    rank_rates = [args.steps / elapsed]
    if dist:
        tt = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        every = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)                          # per-rank times: stragglers show in the record
        rank_rates = [args.steps / float(e.item()) for e in every]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    props = torch.cuda.get_device_properties(dev_index)
    pci = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", -1) & 0xff, getattr(props, "pci_device_id", 0) & 0xff) \
        if hasattr(props, "pci_bus_id") else None
    rank_devices = [{"rank": rank, "device_index": dev_index, "device": torch.cuda.get_device_name(dev_index), "pci_bus_id": pci,
                     "uuid": str(getattr(props, "uuid", "")) or None, "hip_visible_devices": os.environ.get("HIP_VISIBLE_DEVICES")}]
    if dist:
        every_dev = [None] * world
        dist.all_gather_object(every_dev, rank_devices[0])
        rank_devices = every_dev

    import ctypes
    pair_ms, pair_n = ctypes.c_double(0.0), ctypes.c_int(0)
    _lib.check(lib.mvs_profile_dominant_ms(ctypes.byref(pair_ms), ctypes.byref(pair_n)), "mvs_profile_dominant_ms")
    _lib.check(lib.mvs_profile_dominant(0), "mvs_profile_dominant")
    # stage split (warp / conv stack / soft-argmin) for `roofline_kernels`: the SAME library call as the timed region, with the
    # library's own event brackets between its stages (untimed pass -- every event record costs a microsecond or two of
    # device idle time, and only the dominant kernel's bracket has to live inside the timed region)
    stage_ms, stage_n = (ctypes.c_double * 3)(), ctypes.c_int(0)
    _lib.check(lib.mvs_profile_stages(1), "mvs_profile_stages")
    for i in range(min(args.steps, 20)):
        step(i, True)
    torch.cuda.synchronize()
    _lib.check(lib.mvs_profile_stages_ms(stage_ms, ctypes.byref(stage_n)), "mvs_profile_stages_ms")
    _lib.check(lib.mvs_profile_stages(0), "mvs_profile_stages")
    t_warp, t_conv, t_soft = (max(stage_ms[k], 1e-6) * 1e-3 for k in range(3))      # [0] includes the tiny homography kernel
    warp_bytes, conv_flops, soft_bytes = algorithmic_work(w.view_num, w.depth_num, w.height, w.width,
                                                         w.channels, S.base_filter(args.network_mode))
    depth_np = plan.depth.cpu().numpy()
    # every RegNetUS0 layer inside a depth map (untimed pass; the library brackets each launch with HIP events on the
    # stream it goes to)
    layer_ms = (ctypes.c_double * 11)()
    layer_n = ctypes.c_int(0)
    _lib.check(lib.mvs_profile_layers(1), "mvs_profile_layers")
    for i in range(min(args.steps, 20)):
        step(i, False)
    torch.cuda.synchronize()
    _lib.check(lib.mvs_profile_layers_ms(layer_ms, ctypes.byref(layer_n)), "mvs_profile_layers_ms")
    _lib.check(lib.mvs_profile_layers(0), "mvs_profile_layers")
    # the same K steps again, several times: the headline is one sample of a noisy quantity (clock ramps, DVFS)
    block_rates = []
    if not args.no_extra and args.blocks > 0:
        for _ in range(args.blocks):
            if dist:
                dist.barrier()
            eb = timed_block(step, args.steps)
            if dist:
                tb = torch.tensor([eb], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
                dist.all_reduce(tb, op=dist.ReduceOp.MAX)
                eb = float(tb.item())
            block_rates.append(world * args.steps / eb)

    if rank == 0:
        tr = pmc_traffic()
        kernels = [
            # Priced against HBM (its floor: the volume's write), but NOT HBM-bound: the kernel is bound by instruction issue --
            # vector ALU + vector-memory (TA) instructions at 4 waves per SIMD (counters below: SQ_ACTIVE_INST_ANY of the four
            # waves of a SIMD adds up to ~0.9 of the kernel's duration, TA busy ~0.74 of it, HBM traffic = 1.00 x algorithmic).
            # Four other forms were built and measured no faster (LDS-staged, MFMA blend x 2, parity-slot tap cache).
            {"kernel": "warp+variance cost volume (cost_volume_sweep_kernel)", "bound": "valu+ta",
             "achieved": warp_bytes / t_warp / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": warp_bytes / t_warp / 1e9 / HBM_PEAK_GBS, "ms": t_warp * 1e3,
             "frac_is": "fraction of the HBM peak (the kernel's floor); the binding resource is instruction issue, see counters",
             "algorithmic_bytes": warp_bytes, "traffic": tr.get("warp") or None, "counters": tr.get("warp_counters")},
            {"kernel": "RegNetUS0 3D conv stack (11 conv launches, BatchNorm folded into the consumers)", "bound": "mfma",
             "achieved": conv_flops / t_conv / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
             "frac": conv_flops / t_conv / 1e12 / MFMA_F32_PEAK_TFLOPS, "ms": t_conv * 1e3,
             "algorithmic_flops": conv_flops, "peak_dtype": "fp32-input MFMA (dense)",
             "traffic": tr.get("conv") or None},
            {"kernel": "softmax+soft-argmin+prob (softargmin_prob_kernel)", "bound": "hbm",
             "achieved": soft_bytes / t_soft / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": soft_bytes / t_soft / 1e9 / HBM_PEAK_GBS, "ms": t_soft * 1e3,
             "algorithmic_bytes": soft_bytes, "traffic": tr.get("soft") or None},
        ]
        # The dominant KERNEL (one launch): conv3d_c8_kernel, the fused pass of 3dconv0_1 (32->8, stride 1)
        # and 3dconv1_0 (32->16, stride 2) over the cost volume: 27*32*(8 + 16/8) MAC per full-resolution voxel.
        if pair_n.value > 0 and w.channels == 32 and S.base_filter(args.network_mode) == 8 and args.conv_impl in ("auto", "mfma"):
            pair_flops = 2.0 * 27 * 32 * (8 + 16 / 8.0) * w.depth_num * w.height * w.width
            t_pair = pair_ms.value * 1e-3
            kernels.append({"kernel": "conv3d_c8_kernel<true,false>: fused 3dconv0_1 + 3dconv1_0 pass over the cost volume (one launch)",
                            "bound": "mfma", "achieved": pair_flops / t_pair / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": pair_flops / t_pair / 1e12 / MFMA_F32_PEAK_TFLOPS,
                            "ms": t_pair * 1e3, "algorithmic_flops": pair_flops, "launches_timed": pair_n.value,
                            "peak_dtype": "fp32-input MFMA (dense)", "traffic": tr.get("pair") or None})
            dominant = kernels[-1]
        else:
            dominant = max(kernels, key=lambda k: k["ms"])
        if layer_n.value > 0:
            lf = layer_flops(w.channels, S.base_filter(args.network_mode), w.depth_num * w.height * w.width)
            fused = layer_ms[LAYERS.index("3dconv1_0")] == 0.0 and layer_ms[LAYERS.index("3dconv0_1")] > 0.0
            fused2 = layer_ms[LAYERS.index("3dconv2_0")] == 0.0 and layer_ms[LAYERS.index("3dconv1_1")] > 0.0      # round 4: 3dconv2_0 rides in 3dconv1_1's launch
            # round 4: 3dconv2_1's blocks ride as fillers in the launches of 3dconv3_0 / 3_1 / 4_0 (no launch of its own)
            shares = (ctypes.c_int * 3)()
            _lib.check(lib.mvs_regnet_filler_shares(shares), "mvs_regnet_filler_shares")
            filled = layer_ms[LAYERS.index("3dconv2_1")] == 0.0 and layer_ms[LAYERS.index("3dconv3_1")] > 0.0 and sum(shares) == 1000
            share_of = dict(zip(("3dconv3_0", "3dconv3_1", "3dconv4_0"), (x / 1000.0 for x in shares))) if filled else {}
            for li, name in enumerate(LAYERS):
                ms = layer_ms[li]
                if ms <= 0.0:
                    continue
                rider = "3dconv1_0" if (fused and name == "3dconv0_1") else "3dconv2_0" if (fused2 and name == "3dconv1_1") else None
                fl = lf[name] + (lf[rider] if rider else 0.0) + share_of.get(name, 0.0) * lf["3dconv2_1"]
                label = name + (" + %s fused" % rider if rider else "")
                if name in share_of:
                    label += " + %.2f of 3dconv2_1's blocks as filler workgroups" % share_of[name]
                row = {"kernel": "%s (in-pipeline, HIP events around the launch)" % label,
                       "ms": ms, "algorithmic_flops": fl, "launches_timed": layer_n.value}
                if name == "3dconv6_2":                     # 8 -> 1 channels: reads two 8-channel volumes, writes one channel
                    by = w.depth_num * w.height * w.width * (2 * 8 + 1) * 4
                    row.update({"bound": "hbm", "achieved": by / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": by})
                else:
                    row.update({"bound": "mfma", "achieved": fl / (ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": fl / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS})
                kernels.append(row)
        out = {
            "metric": "depth maps/sec (N=5, D=192, 160x128)" if args.workload == "M" else "depth maps/sec",
            "value": world * args.steps / elapsed,
            "unit": "depth maps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.conv_impl != "bf16x3" else "bf16x3 (fp32 operands split into two bf16, fp32 accumulate; opt-in)",
            "data": "synthetic",
            "config": {"workload": "%s: features->depth, N=%d views, D=%d planes, %dx%d feature maps, C=%d, "
                                   "3D-CNN regulariser (RegNetUS0), network_mode=%s; one reference view per "
                                   "step per GPU, sharded by reference view" % (
                                       w.name, w.view_num, w.depth_num, w.width, w.height, w.channels,
                                       args.network_mode),
                       "conv_impl": args.conv_impl, "streams_per_gpu": n_streams},
            "roofline": {k: dominant[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")},
            "roofline_kernels": kernels,
            "roofline_stage_rows": {"sum_ms": (t_warp + t_conv + t_soft) * 1e3, "ms_per_step": elapsed / args.steps * 1e3,
                                    "launches_timed": stage_n.value,
                                    "source": "the first three rows of roofline_kernels are the library's own HIP-event brackets between the stages of "
                                              "mvs_depth_from_features_f32 (mvs_profile_stages), i.e. of the timed path; their sum exceeds ms_per_step "
                                              "by the device idle time of four event records per depth map"},
            "depth_checksum": float(np.float64(depth_np).sum()),
            "per_rank_depth_maps_per_s": {"min": min(rank_rates), "max": max(rank_rates), "ranks": rank_rates},
            "ranks": {"world_size": dist.get_world_size() if dist else 1, "backend": (backend + (" (RCCL)" if backend == "nccl" else "")) if dist else None,
                      "rccl_version": rccl_version(), "distinct_pci_bus_ids": len({d_.get("pci_bus_id") for d_ in rank_devices}),
                      "devices": rank_devices},
        }
        chain = ["3dconv2_0", "3dconv3_0", "3dconv3_1", "3dconv4_0", "3dconv5_0"]
        if layer_n.value > 0:
            out["low_resolution_chain_us"] = 1e3 * sum(layer_ms[LAYERS.index(n_)] for n_ in chain)     # (3dconv2_0 counts 0 when it rides in 3dconv1_1's launch)
            if layer_ms[LAYERS.index("3dconv2_1")] == 0.0:
                out["low_resolution_chain_includes"] = "3dconv2_1 (its blocks ride in the launches of 3dconv3_0 / 3_1 / 4_0)"
        if block_rates:
            out["repeat_blocks"] = {"blocks": len(block_rates), "steps_per_block": args.steps,
                                    "median": float(np.median(block_rates)), "min": min(block_rates), "max": max(block_rates),
                                    "values": block_rates}
        out["roofline"]["kernel"] = dominant["kernel"]
        out["roofline"]["traffic_source"] = tr.get("source")
        if args.with_images or (world == 1 and n_streams == 1 and not args.no_extra and args.workload == "M" and args.extractor == "hip"):
            up = S.make_unet_params(args.network_mode, seed=3)
            if args.extractor == "hip":
                from mvsnet_amd.feature_net_hip import HipUNetDS2GN as UNetDS2GN
            else:
                from mvsnet_amd.feature_net import UNetDS2GN
            net = UNetDS2GN(up, dev)
            imgs = torch.as_tensor(S.make_images(w.view_num, 4 * w.height, 4 * w.width, seed=0)).to(dev)
            for _ in range(3):
                f = net(imgs)
            torch.cuda.synchronize()
            if args.extractor == "hip":
                # the towers alone: N images -> N feature maps, back to back; events on the stream they run on
                from mvsnet_amd.feature_net import unet_layer_work, unet_macs
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n_pass = 50
                t_h = time.perf_counter()
                e0.record()
                for _ in range(n_pass):
                    f = net(imgs)
                e1.record()
                t_h = time.perf_counter() - t_h
                torch.cuda.synchronize()
                tw_ms = e0.elapsed_time(e1) / n_pass
                gf = 2.0 * unet_macs(4 * w.height, 4 * w.width) * w.view_num / 1e9
                out["towers"] = {"workload": "UNetDS2GN, %d images of %dx%d -> %d feature maps of %dx%dx32, one batched pass" %
                                             (w.view_num, 4 * w.width, 4 * w.height, w.view_num, w.width, w.height),
                                 "ms_per_pass": tw_ms, "algorithmic_gflop": gf, "achieved_tflops": gf / tw_ms,
                                 "peak_tflops": 157.3, "frac_of_fp32_mfma_peak": gf / tw_ms / 157.3,
                                 # 32 layers, each a global barrier (GroupNorm): the floor of the pass is the SUM over layers of the larger
                                 # of the layer's matrix time and its HBM time (several full-resolution layers are HBM-bound, not MFMA-bound)
                                 "floor_ms_sum_over_layers_of_max_mfma_hbm": sum(max(2.0 * m_ * w.view_num / 157.3e12, b_ * w.view_num / HBM_PEAK_GBS / 1e9)
                                                                               for _n, m_, b_ in unet_layer_work(4 * w.height, 4 * w.width)) * 1e3,
                                 "frac_of_that_floor": sum(max(2.0 * m_ * w.view_num / 157.3e12, b_ * w.view_num / HBM_PEAK_GBS / 1e9)
                                                           for _n, m_, b_ in unet_layer_work(4 * w.height, 4 * w.width)) * 1e3 / tw_ms,
                                 "host_enqueue_ms_per_pass": t_h / n_pass * 1e3,
                                 "side_streams": len(getattr(net, "_choice", {}).get((4 * w.height, 4 * w.width), [])),
                                 "launches_per_pass": 32,
                                 "kernels": "csrc/unet2d_p.hip (persistent, 13 layers) + csrc/unet2d.hip (one tile per workgroup, 19 layers)"}
            t1 = time.perf_counter()
            for _ in range(args.steps):
                f = net(imgs)
                plan.set_cameras(cams, w.depth_start, w.depth_interval, end, False)
                plan.run_3dcnn(f, w.depth_start, w.depth_interval)
            torch.cuda.synchronize()
            out["images_to_depth_maps_per_s"] = args.steps / (time.perf_counter() - t1)
            out["extractor"] = args.extractor
            if args.extractor == "hip":
                # the same work pipelined over two streams, as mvsnet_amd.inference runs a session since round 6: the towers of
                # reference view i + 1 on a stream of their own beside the hot path of view i (the host runs ahead of both)
                ts = torch.cuda.Stream(dev)
                main_s = torch.cuda.current_stream(dev)

                def piped(n_):
                    for _ in range(n_):
                        with torch.cuda.stream(ts):
                            f_ = net(imgs)
                            ev_ = ts.record_event()
                        main_s.wait_event(ev_)
                        f_.record_stream(main_s)
                        plan.set_cameras(cams, w.depth_start, w.depth_interval, end, False)
                        plan.run_3dcnn(f_, w.depth_start, w.depth_interval)
                piped(5)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                piped(args.steps)
                torch.cuda.synchronize()
                out["images_to_depth_maps_per_s_towers_on_their_own_stream"] = args.steps / (time.perf_counter() - t1)
        if world == 1 and n_streams == 1 and not args.no_extra and args.workload == "M" and args.network_mode == "normal":
            # BASELINE.json configs[1] and configs[2] on this GPU, each with its distance from the committed fixture
            t_x = [time.perf_counter()]

            def lap(tag):
                t_x.append(time.perf_counter())
                print("bench extra %-28s %6.1f s" % (tag, t_x[-1] - t_x[-2]), file=sys.stderr, flush=True)
            out["config_c2"] = extra_config_3dcnn("c2", dev); lap("config_c2")
            if args.conv_impl == "auto":
                # Opt-in split precision, reported BESIDE `value` and never as it: every fp32 product of 3dconv0_1 as three bf16
                # MFMAs on hi / lo halves (16 significant bits per operand), all other layers and all accumulation in fp32.
                _lib.set_conv_impl("bf16x3")
                try:
                    r = extra_config_3dcnn("M", dev, steps=200)
                    r["dtype"] = "fp32 operands split into two bf16 halves for 3dconv0_1 (conv_impl='bf16x3'), fp32 accumulate; everything else as `value`"
                    r["note"] = "opt-in (mvs_set_conv_impl / --conv-impl bf16x3); `value` is measured with exact fp32 MFMA"
                    out["split_precision_bf16x3"] = r
                finally:
                    _lib.set_conv_impl(args.conv_impl)
                lap("split_precision_bf16x3")
            out["config_c3_gru"] = extra_config_gru("c3", dev, 5, 1); lap("config_c3_gru")
            out["config_c3_gru"]["formulations_ms_per_depth_map"] = gru_formulations(dev); lap("c3 formulations")
            out["config_c3_gru"]["fresh_process"] = gru_config_in_child("c3"); lap("c3 fresh process")
            out["config_c3_gru_4_views"] = extra_config_gru("c3", dev, 3, 4); lap("config_c3_gru_4_views")
            out["config_c3_gru_4_views"]["per_plane_counters"] = gru_pmc_per_plane()
            out["config_c3_gru_from_images"] = gru_production_order(dev); lap("config_c3_gru_from_images")
            out["session"] = session_record(dev, out["value"]); lap("session")
            # the per-GPU share of configuration 4 (1 078 reference views over 8 GPUs = ~135 per rank): the same loop over a session
            # long enough that the pipeline's fill (first decodes) and drain (last file writes) stop dominating the 48-view record
            out["session_144_views"] = session_record(dev, out["value"], n_images=144, procs_per_gpu=0); lap("session_144_views")
            # configuration 2's image size (1152 x 864 JPEGs: ~3x the host decode work per image; 288 x 216 feature maps)
            out["session_config2_images"] = session_record(dev, out["config_c2"].get("depth_maps_per_s"), n_images=24, procs_per_gpu=0,
                                                           width=1152, height=864); lap("session_config2_images")
            # configuration 4 as one rank sees it: 1 078 reference views of 1152 x 864 over 8 GPUs = 135 per rank (mvsnet/inference.py:105-119
            # sharded by reference view); on one GPU this IS that rank's work, the 8-GPU number is the driver's to measure
            out["session_config4_rank_share_135_views"] = session_record(dev, out["config_c2"].get("depth_maps_per_s"), n_images=135, procs_per_gpu=0,
                                                                         width=1152, height=864); lap("session_config4_rank_share_135_views")
            out["training"] = training_record(dev, configs=(("3dcnn_d192", "3DCNN", 192), ("3dcnn_d128_config5", "3DCNN", 128), ("gru_d192", "GRU", 192))
                                              if args.training_all else (("3dcnn_d128_config5", "3DCNN", 128),)); lap("training")
            out["training_from_disk"] = training_from_disk_record(); lap("training_from_disk")
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(w, rp, args.cpu_budget, depth_np)
            if args.workload == "M":       # configs[0]: the configuration BASELINE.json defines AS the CPU run (N=3, D=32, 160x128 features)
                w1 = S.make_workload("c1", args.network_mode, seed=0)
                out["cpu_baseline_c1"] = cpu_baseline(w1, rp, 5.0, None, max_maps=4)
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
