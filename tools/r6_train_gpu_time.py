"""GPU time against wall time of the training step (config 5 per GPU: 3 x 640 x 480, D = 128), from a rocprofv3 kernel trace:
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6_train_trace -- python $GRAFT_REPO_ROOT/tools/r6_train_gpu_time.py run
  python tools/r6_train_gpu_time.py report gpurun_out/r6_train_trace [kernel-name substring ...]
`run`: 10 warm-up + 20 traced-and-timed steps (inputs on the device); `report`: per step, the sum of kernel durations, the span
they cover, the number of launches and the kernels with the largest totals."""
import csv, glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STEPS = 20
if sys.argv[1] == "run":
    import numpy as np, torch
    from mvsnet_amd import synthetic as S, train as T
    N, H, W, D = 3, 480, 640, 128
    images = torch.as_tensor(S.make_images(N, H, W)).cuda(); cams = S.make_cams(N, H // 4, W // 4, D)
    start, interval = float(cams[0, 1, 3, 0]), float(cams[0, 1, 3, 1])
    gt = torch.as_tensor(np.full((H // 4, W // 4, 1), start + interval * D * 0.5, np.float32)).cuda()
    tr = T.Trainer("normal", "cuda")
    for _ in range(10): tr.train_step(images, cams, gt, D)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(STEPS): tr.train_step(images, cams, gt, D)
    torch.cuda.synchronize()
    print("wall per step under the tracer: %.2f ms" % ((time.time() - t0) / STEPS * 1e3))
else:
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = len(rows) // 30                                       # launches per step (10 + 20 steps, the first ones carry one-offs)
    last = rows[-STEPS * n:]
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    busy = sum(dur(r) for r in last) / STEPS
    span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3 / STEPS
    print("launches per step ~%d; kernel time per step %.0f us; span per step %.0f us; GPU busy %.2f of the span" % (n, busy, span, busy / span))
    tot = {}
    for r in last:
        k = r["Kernel_Name"][:90]
        t = tot.setdefault(k, [0, 0.0]); t[0] += 1; t[1] += dur(r)
    for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:28]:
        print("%7.1f us/step %5.1f launches/step  %s" % (t / STEPS, c / STEPS, k))
    for pat in sys.argv[3:]:                                  # every launch of the last step whose kernel name contains `pat`, in order
        print(pat, "per launch (us):", " ".join("%.0f" % dur(r) for r in rows[-n:] if pat in r["Kernel_Name"]))
