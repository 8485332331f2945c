"""Worker PROCESSES for the host side of the session loop (the reference's tf.data prefetch thread,
mvsnet/predictlib.py:48-51, and its write_output, :162-177).

Round 3 ran decode / rescale / crop of the input images and the encoding of the six output files per reference view on
THREADS of the GPU-owning process: 12 worker threads and the thread that enqueues the GPU work took turns on one GIL, the
main thread's 2.4 ms of Python per depth map stretched to 3.9 ms and the loop delivered 0.30 of the kernel-only rate.  Here
the same functions run in worker processes:

  * `image_task`  -- the per-IMAGE part of cluster_generator.py:234-286 (load -> scale-to-cover -> centre-crop -> output
    scale), result handed back through a shared-memory slot (multiprocessing.shared_memory; nothing of the 1 MB image
    goes through a pipe);
  * `write_task`  -- predictlib.write_output_slice (two .pfm, two 16-bit PNGs, the reference .jpg, the camera .txt).

The pool is started with the `spawn` method: the children are fresh interpreters that import numpy / Pillow and this
package's host modules only (never torch, never the HIP library), so it is safe to create after the parent has
initialised the GPU, and nothing of the parent's HIP state is inherited.  One pool per process, kept for the process's
lifetime (`get_pool`); `workers = 0` keeps everything on threads (round 3's path, still used for upstream-format
projects).
"""
from __future__ import annotations

import atexit
import os
import threading
import time
from concurrent.futures import Future, ProcessPoolExecutor

import numpy as np

_ATTACHED = {}          # worker side: shared-memory blocks by name


class PinnedArray(np.ndarray):
    """uint8 image whose memory is a pinned (page-locked) torch tensor, kept in `.pinned`: the GPU-owning process uploads it
    with an asynchronous copy straight from here (HostPool.load_image(pin_device=...))."""
    pinned = None


def _attach(name):
    from multiprocessing import shared_memory
    shm = _ATTACHED.get(name)
    if shm is None:
        for old in list(_ATTACHED):                   # the parent replaced its block: drop the mapping of the old one
            _ATTACHED.pop(old).close()
        shm = _ATTACHED[name] = shared_memory.SharedMemory(name=name)
    return shm


def prepare_image(path, rescale, width, height, base_image_size, output_scale):
    """The per-image part of ClusterGenerator.prepare (cluster_generator.py:234-286): BGR decode (mvs_cluster.py:72-76),
    scale by the cluster's `rescale` (utils.py:107-118), centre-crop (utils.py:121-153), and the output-scaled copy that
    write_output stores as <idx>.jpg.  Returns (cropped uint8 (h,w,3), output image uint8, original shape)."""
    from PIL import Image
    from .mvs_data_generation import crop_mvs_input, scale_image
    rgb = np.asarray(Image.open(path).convert("RGB"))
    img = np.ascontiguousarray(rgb[:, :, ::-1])
    im = scale_image(img, rescale)
    cr, _ = crop_mvs_input([im], [np.zeros((2, 4, 4))], width, height, base_image_size)
    cr = np.ascontiguousarray(cr[0])
    return cr, scale_image(cr, output_scale), img.shape


def image_task(path, rescale, width, height, base_image_size, output_scale, shm_name, offset, capacity):
    """Worker: prepare_image into the shared-memory slot [offset, offset + capacity): cropped image, then output image."""
    t0 = time.perf_counter()
    cr, out, shape = prepare_image(path, rescale, width, height, base_image_size, output_scale)
    if cr.nbytes + out.nbytes > capacity:
        raise ValueError("image slot too small: %d + %d > %d bytes" % (cr.nbytes, out.nbytes, capacity))
    buf = _attach(shm_name).buf
    np.frombuffer(buf, np.uint8, cr.size, offset)[:] = cr.reshape(-1)
    np.frombuffer(buf, np.uint8, out.size, offset + cr.nbytes)[:] = out.reshape(-1)
    return cr.shape, out.shape, shape, time.perf_counter() - t0


def image_task_pickled(path, rescale, width, height, base_image_size, output_scale):
    """Worker: prepare_image with the result returned through the pool's pipe (when /dev/shm cannot hold a slot ring)."""
    t0 = time.perf_counter()
    cr, out, shape = prepare_image(path, rescale, width, height, base_image_size, output_scale)
    return cr, out, shape, time.perf_counter() - t0


def write_task(output_dir, depth, prob, ref_image, ref_cam, index, visualize, prob_upsample):
    """Worker: predictlib.write_output_slice (predictlib.py:105-159)."""
    from . import predictlib as pl
    t0 = time.perf_counter()
    pl.write_output_slice(output_dir, depth, prob, ref_image, ref_cam, index, visualize, prob_upsample)
    return time.perf_counter() - t0


def _worker_init(parent_pid):
    """Runs once in every worker: a daemon thread that ends the worker when its parent is gone (a killed parent would leave
    the workers blocked on their task queue for good)."""
    def watch():
        while True:
            time.sleep(2.0)
            if os.getppid() != parent_pid:
                os._exit(0)
    threading.Thread(target=watch, daemon=True).start()


def _warm(hold=0.0):
    import PIL.Image            # noqa: F401  (the first task should not pay the imports)
    from . import mvs_data_generation, predictlib, preprocess   # noqa: F401
    if hold:
        import time
        time.sleep(hold)        # keeps this worker busy so that the executor has to start another one for the next task
    return os.getpid()


class HostPool:
    """`workers` spawned processes + a ring of shared-memory image slots."""

    def __init__(self, workers, slots=48):
        import multiprocessing as mp
        self.workers = int(workers)
        self.ex = ProcessPoolExecutor(max_workers=self.workers, mp_context=mp.get_context("spawn"),
                                      initializer=_worker_init, initargs=(os.getpid(),))
        self.slots = self.max_slots = int(slots)
        self.slot_bytes = 0
        self.shm = None
        self.free = []
        self.cv = threading.Condition()
        # The children take two things from the moment they are started:
        #  * the environment -- they are given NO GPU (a worker never needs one);
        #  * the description of the parent's __main__ module, which the spawn method would re-import in every child (running an
        #    unguarded main script again: a second copy of the application, GPU work included; and failing outright when the
        #    parent runs from stdin or an embedded interpreter).  The workers only ever call functions of this module, so they
        #    are started with an EMPTY __main__: nothing of the application is imported there.
        # ProcessPoolExecutor starts its workers ON DEMAND (Python 3.9+: one per submit while no worker is idle, inside submit()
        # itself), so the window in which the parent's environment / __main__ are swapped has to cover the submits that start
        # them -- and nothing else: the constructor does NOT wait for the workers to come up (interpreter start + imports,
        # ~0.3 s, used to be half of a one-scan process's first pass); the session's first tasks queue behind the warm-up
        # tasks while the parent goes on with its own one-offs.  Should fewer than `workers` processes exist after the submits
        # (a worker came up and went idle in between: not seen, but nothing forbids it), the warm-up tasks are waited for and a
        # round of holding ones is submitted until the executor's own process table is full (a fast first task used
        # to let later workers start outside the window, with the GPU visible and the real __main__).
        import sys
        import types
        hide = {"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": "", "MVS_HOST_WORKER": "1"}
        saved = {k: os.environ.get(k) for k in hide}
        os.environ.update(hide)
        real_main = sys.modules.get("__main__")
        sys.modules["__main__"] = types.ModuleType("__main__")
        try:
            futs = [self.ex.submit(_warm) for _ in range(self.workers)]
            for _ in range(8):                               # normally not entered
                if len(getattr(self.ex, "_processes", None) or {}) >= self.workers or not hasattr(self.ex, "_processes"):
                    break
                for f in futs:
                    f.result()
                futs = [self.ex.submit(_warm, 0.2) for _ in range(self.workers)]
            procs = getattr(self.ex, "_processes", None)
            if not procs:                                    # an executor without a process table: ask the workers
                for _ in range(8):
                    futs += [self.ex.submit(_warm, 0.2) for _ in range(self.workers)]
                    if len(set(f.result() for f in futs)) >= self.workers:
                        break
            self.pids = sorted(procs.keys()) if procs else sorted(set(f.result() for f in futs))
        finally:
            if real_main is not None:
                sys.modules["__main__"] = real_main
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    def cpu_seconds(self):
        """user + system CPU time the worker processes have used so far (psutil), or None."""
        try:
            import psutil
            return sum(sum(psutil.Process(pid).cpu_times()[:2]) for pid in self.pids)
        except Exception:                              # noqa: BLE001
            return None

    def _ensure_slots(self, nbytes):
        """(Re)allocates the slot ring when a session needs larger slots; waits until no slot is in use.  The ring takes at most
        a quarter of what /dev/shm has free (a tmpfs that runs full kills the writer with SIGBUS: containers often give it
        64 MB, and a 1600 x 1200 slot is 6 MB); with room for fewer than 4 slots the pool hands images back through its pipe
        instead (returns False)."""
        from multiprocessing import shared_memory
        nbytes = (int(nbytes) + 4095) & ~4095
        with self.cv:
            if self.shm is not None and nbytes <= self.slot_bytes:
                return True
            if self.shm is None and self.slot_bytes == -1:
                return False                              # no room was found before: stay on the pipe
            while self.shm is not None and len(self.free) < self.slots:
                self.cv.wait()
            if self.shm is not None:
                self.shm.close(); self.shm.unlink(); self.shm = None
            # a quarter of what /dev/shm has free NOW -- per NODE, not per process: the ranks of one node start together and
            # each would otherwise see (and reserve a quarter of) the same free space, eight ranks twice what exists; the first
            # page touched past the tmpfs limit is a SIGBUS
            room = shm_room()
            slots = min(self.max_slots, room // nbytes)
            if slots < 4:
                self.slot_bytes = -1
                return False
            try:
                self.shm = shared_memory.SharedMemory(create=True, size=slots * nbytes)
            except OSError:
                self.slot_bytes = -1
                return False
            self.slots, self.slot_bytes = slots, nbytes
            self.free = list(range(slots))
            return True

    def load_image(self, path, rescale, width, height, base_image_size, output_scale, pin_device=None):
        """-> Future of (cropped uint8 (h,w,3), output image, original shape, worker seconds); private arrays, the slot is
        free again when the future resolves.  `pin_device` (a CUDA device index): the cropped image is copied out of the slot
        into PINNED host memory (torch's caching host allocator, on the executor's management thread) and comes back as a
        PinnedArray -- the caller's host -> device copy then needs no staging copy on its own thread (round 5 copied every image
        twice: slot -> private array here, private array -> pinned staging buffer on the thread that feeds the GPU)."""
        h_cap = max(height, int(np.ceil(height / base_image_size) * base_image_size)) + base_image_size
        w_cap = max(width, int(np.ceil(width / base_image_size) * base_image_size)) + base_image_size
        need = h_cap * w_cap * 3
        need += int(need * max(output_scale, 0.0) ** 2) + 4096
        if not self._ensure_slots(need):
            return self.ex.submit(image_task_pickled, path, rescale, width, height, base_image_size, output_scale)
        with self.cv:
            while not self.free:
                self.cv.wait()
            slot = self.free.pop()
            shm, cap = self.shm, self.slot_bytes
        out = Future()
        off = slot * cap
        inner = self.ex.submit(image_task, path, rescale, width, height, base_image_size, output_scale, shm.name, off, cap)

        def done(f):                                   # executor's management thread: copy out of the slot, free it
            try:
                cs, os_, shape, sec = f.result()
                n1 = int(np.prod(cs)); n2 = int(np.prod(os_))
                src = np.frombuffer(shm.buf, np.uint8, n1, off).reshape(cs)
                if pin_device is None:
                    cr = src.copy()
                else:
                    import torch
                    with torch.cuda.device(pin_device):
                        tbuf = torch.empty(tuple(cs), dtype=torch.uint8, pin_memory=True)
                    cr = tbuf.numpy().view(PinnedArray)
                    cr.pinned = tbuf
                    np.copyto(cr, src)
                oi = np.frombuffer(shm.buf, np.uint8, n2, off + n1).reshape(os_).copy()
                out.set_result((cr, oi, shape, sec))
            except BaseException as e:                 # noqa: BLE001  (delivered to the consumer)
                out.set_exception(e)
            finally:
                with self.cv:
                    self.free.append(slot)
                    self.cv.notify_all()
        inner.add_done_callback(done)
        return out

    def write_outputs(self, output_dir, depth, prob, ref_image, ref_cam, index, visualize=False, prob_upsample=None):
        """-> Future of the worker's seconds.  The arrays are pickled by the executor's feeder thread: pass private copies."""
        return self.ex.submit(write_task, output_dir, depth, prob, ref_image, ref_cam, index, visualize, prob_upsample)

    def close(self):
        try:
            self.ex.shutdown(wait=True, cancel_futures=True)
        finally:
            if self.shm is not None:
                try:
                    self.shm.close(); self.shm.unlink()
                except FileNotFoundError:
                    pass
                self.shm = None


_POOL = None
_POOL_LOCK = threading.Lock()


def shm_room(statvfs=None):
    """Bytes of /dev/shm this process may reserve for its image slots: a quarter of the free space, shared between the
    LOCAL_WORLD_SIZE ranks of the node."""
    try:
        vfs = (statvfs or os.statvfs)("/dev/shm")
        room = vfs.f_bavail * vfs.f_frsize // 4
    except OSError:
        return 0
    local = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
    return room // max(1, local)


def default_workers():
    """Worker processes per GPU-owning process: the host cores this process may count on (cores / local ranks), less the
    main thread and its helper threads, at most 12 (450 depth maps/s need ~6 busy cores of decode + encode)."""
    n = os.cpu_count() or 4
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    local = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
    return max(2, min(12, n // max(1, local) - 2))


def get_pool(workers=None):
    """The process-wide pool (created on first use; `workers` = 0 -> None: the caller stays on threads)."""
    global _POOL
    w = default_workers() if workers is None else int(workers)
    if w <= 0:
        return None
    with _POOL_LOCK:
        if _POOL is None or _POOL.workers != w:
            if _POOL is not None:
                _POOL.close()
            _POOL = HostPool(w)
    return _POOL


@atexit.register
def _close_pool():
    global _POOL
    if _POOL is not None:
        _POOL.close()
        _POOL = None
