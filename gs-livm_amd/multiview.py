"""View-parallel execution across the GPUs of one node.

The views rendered in one optimiser iteration are independent forward/backward passes over the SAME
Gaussian set (reference src/liw/lioOptimization.cpp:1691-1737) whose gradients are summed
(:1822-1825).  So: one process per GPU, the parameter buffer replicated, rank r renders the views
`shard_views(...)` assigns to it, and the only exchange steps are
  * `reduce_gradients`   -- sum of the per-view parameter gradients onto the optimiser owner, and
  * `broadcast_gaussians` -- the owner's updated parameter buffer back to every replica,
both on one flat f32 buffer (56 B per Gaussian at M = 1) so each is a single large RCCL collective
over xGMI.  The rasterizer itself never communicates.  Backend "nccl" is RCCL on ROCm; the same
code runs on "gloo" for the CPU tests.
"""
from collections import OrderedDict

import torch
import torch.distributed as dist

# parameter groups in buffer order (the six leaves of GaussianModel, include/gs/gs/gaussian.cuh:107-119):
# name -> trailing shape as a function of M
_LAYOUT = OrderedDict([("means3D", lambda M: (3,)), ("features_dc", lambda M: (1, 3)),
                       ("features_rest", lambda M: (M - 1, 3)), ("scales", lambda M: (3,)),
                       ("rotations", lambda M: (4,)), ("opacities", lambda M: (1,))])


def floats_per_gaussian(M):
    return 11 + 3 * M


class GaussianBuffer:
    """One contiguous f32 buffer holding every per-Gaussian parameter group as SoA blocks
    [means3D | features_dc | features_rest | scales | rotations | opacities]; `views` are zero-copy tensors
    into it."""

    def __init__(self, P, M, device, dtype=torch.float32):
        self.P, self.M = int(P), int(M)
        self.flat = torch.zeros(self.P * floats_per_gaussian(self.M), dtype=dtype, device=device)
        self.views = OrderedDict()
        off = 0
        for name, shp in _LAYOUT.items():
            shape = (self.P,) + shp(self.M)
            n = 1
            for d in shape:
                n *= d
            self.views[name] = self.flat[off:off + n].view(shape)
            off += n
        assert off == self.flat.numel()

    def load(self, arrays):
        """arrays: dict with the view names; "shs" [P,M,3] may stand in for features_dc + features_rest."""
        arrays = dict(arrays)
        if "shs" in arrays and "features_dc" not in arrays:
            shs = torch.as_tensor(arrays["shs"])
            arrays["features_dc"], arrays["features_rest"] = shs[:, :1], shs[:, 1:]
        for name, v in self.views.items():
            if v.numel():
                v.copy_(torch.as_tensor(arrays[name]).reshape(v.shape))
        return self

    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()


def shard_views(n_views, rank, world_size):
    """Views of one iteration assigned to `rank`: round-robin, so any n_views >= 0 is covered exactly once."""
    return list(range(rank, n_views, world_size))


def _collective(t, fn, group):
    """Runs fn(tensor) on `t`.  RCCL ("nccl") works on device memory directly; the gloo rehearsal backend
    (CPU tests, single-GPU dry runs of the multi-rank control flow) is staged through host memory."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        fn(h)
        t.copy_(h)
    else:
        fn(t)


def broadcast_gaussians(buf, src=0, group=None):
    """Owner -> replicas: one collective on the flat parameter buffer."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        _collective(buf.flat if isinstance(buf, GaussianBuffer) else buf,
                    lambda x: dist.broadcast(x, src=src, group=group), group)
    return buf


def reduce_gradients(grad, dst=0, group=None, all_ranks=False):
    """Sum of per-view gradients (flat buffer of the same layout) onto the owner (or everywhere)."""
    t = grad.flat if isinstance(grad, GaussianBuffer) else grad
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if all_ranks:
            _collective(t, lambda x: dist.all_reduce(x, op=dist.ReduceOp.SUM, group=group), group)
        else:
            _collective(t, lambda x: dist.reduce(x, dst=dst, op=dist.ReduceOp.SUM, group=group), group)
    return grad


def launch_local_ranks(argv, n_ranks, extra_env=None, timeout=3600.0, out=None, err=None):
    """One process per GPU on this node WITHOUT an external launcher: starts `n_ranks` children running `argv`
    (rank r gets RANK = LOCAL_RANK = r, WORLD_SIZE, MASTER_ADDR = 127.0.0.1 and a free MASTER_PORT -- what
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` would set), relays rank 0's
    stdout to `out` (default: this process's stdout) and every rank's stderr to `err`, and returns the exit status:
    0 only when every rank exited 0.  When a rank fails the others are terminated (by pid) so a dead peer cannot
    leave them waiting in a collective.  The pipes are drained WHILE the ranks run (a reader thread each): a rank that
    writes more than a pipe buffer -- RCCL debug output, a long JSON line -- can never block on a parent that only reads
    at the end.  `timeout` (seconds, default 3600; None = none): ranks still running then are killed, status 124.
    The caller must not have touched the GPU: the children own the devices.
    """
    import os
    import socket
    import subprocess
    import sys
    import time

    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    with socket.socket() as s:  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this host driver
        env.update(extra_env or {})
        procs.append(subprocess.Popen(list(argv), env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=None if err is sys.stderr else subprocess.PIPE, text=True))
    import threading
    captured = {}

    def drain(key, pipe):
        captured[key] = pipe.read()  # (returns at EOF, i.e. when the child has closed its end)

    readers = []
    for r, p in enumerate(procs):
        for key, pipe in ((("out", r), p.stdout), (("err", r), p.stderr)):
            if pipe is not None:
                th = threading.Thread(target=drain, args=(key, pipe), daemon=True)
                th.start()
                readers.append(th)
    t0 = time.monotonic()
    status = 0
    pending = set(range(n_ranks))
    while pending:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 1
                print("[launch_local_ranks] rank %d exited with status %d; stopping the other ranks" % (r, rc), file=err)
                for q in pending:
                    procs[q].terminate()
        if pending and timeout is not None and time.monotonic() - t0 > timeout:
            status = status or 124
            print("[launch_local_ranks] timeout after %.0f s; stopping ranks %s" % (timeout, sorted(pending)), file=err)
            for q in pending:
                procs[q].kill()
            timeout = None
        if pending:
            time.sleep(0.05)
    for th in readers:
        th.join(timeout=10.0)
    text = captured.get(("out", 0)) or ""
    if text:
        out.write(text)
        out.flush()
    if err is not sys.stderr:
        for r in range(n_ranks):
            if captured.get(("err", r)):
                err.write(captured[("err", r)])
    return status


class ViewThreads:
    """The views of one optimiser iteration rendered from several host threads, each on a HIP stream of its own -- how
    the reference drives the rasterizer (four rendering threads; one process, one GPU).  Views are
    independent passes over the same Gaussians, and most of a forward is a chain of small latency-bound kernels (cull /
    project, depth sort, offset scan, tile sort): issued from different streams, one view's chain runs beside another
    view's blend kernels instead of in front of them (2 M Gaussians / 1080p, eight views per iteration: 1.01 -> 0.79 ms
    per view with four threads).  The library keeps its per-thread state (mailbox, counters, per-view histories) per host
    thread, so view j always goes to thread j % T and finds its own history there.

        vt = ViewThreads(4, device)
        outs = vt.render([lambda: rasterizer_j(means3D, means2D_j, opacity, shs=..., ...) for j in range(K)])
        loss = sum(... outs ...); loss.backward()      # every backward runs on the stream its forward ran on

    `render` makes each worker stream wait for the caller's current stream (the inputs are produced there), runs the
    callables, and makes the caller's stream wait for the workers before it returns; results come back in call order.
    """

    def __init__(self, n_threads, device):
        import queue
        import threading

        import torch
        self._torch = torch
        self.device = torch.device(device)
        self._workers = []
        for _ in range(max(1, int(n_threads))):
            qi, qo = queue.Queue(), queue.Queue()
            stream = torch.cuda.Stream(device=self.device)
            th = threading.Thread(target=self._loop, args=(qi, qo, stream), daemon=True)
            th.start()
            self._workers.append((qi, qo, stream, th))

    def _loop(self, qi, qo, stream):
        torch = self._torch
        torch.cuda.set_device(self.device)
        while True:
            job = qi.get()
            if job is None:
                return
            try:
                with torch.cuda.stream(stream):
                    qo.put([fn() for fn in job])
            except BaseException as e:  # noqa: BLE001 -- re-raised in the calling thread
                qo.put(e)

    def __len__(self):
        return len(self._workers)

    def render(self, calls):
        torch = self._torch
        calls = list(calls)
        main = torch.cuda.current_stream(self.device)
        T = len(self._workers)
        jobs = [calls[w::T] for w in range(T)]
        for (qi, qo, stream, th), job in zip(self._workers, jobs):
            stream.wait_stream(main)
            qi.put(job)
        res, err = [], None
        for qi, qo, stream, th in self._workers:
            r = qo.get()
            if isinstance(r, BaseException):
                err = err or r
                r = []
            res.append(r)
            main.wait_stream(stream)
        if err is not None:
            raise err
        return [res[j % T][j // T] for j in range(len(calls))]

    def close(self):
        for qi, qo, stream, th in self._workers:
            qi.put(None)
        for qi, qo, stream, th in self._workers:
            th.join(timeout=10.0)
        self._workers = []

