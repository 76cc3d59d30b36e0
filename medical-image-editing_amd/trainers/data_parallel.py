"""Data-parallel gradient exchange for one process per GPU (RCCL over xGMI via torch.distributed).

Replaces the Lightning DDPPlugin the reference launches with (run_vqwnet.py:112-121).  Parameters are
grouped into buckets in the order their gradients become ready in backward (decoder tail first); a
post-accumulate hook on each parameter counts its bucket down, and the moment a bucket is complete
its flat buffer is all-reduced asynchronously on a side stream while the remaining backward kernels
keep running.  `finish()` waits for the outstanding collectives and scatters the averaged gradients back.

MI355X sizing: xGMI is point-to-point (7 links per GPU), so a ring all-reduce is per-link bound and the
61.8 MB of fp32 gradients of the R-cfg model cost ~1 ms in total; few large buckets (16 MiB in the overlapped schedule)
keep launch/latency overhead negligible against a >100 ms step.

Two schedules.  `overlap=False` (default, VQW_DP_OVERLAP=0): the buckets are exchanged in finish(), after the backward pass
has been enqueued - as ONE all-reduce of the whole 61.8 MB (round 4; four 16 MiB ones before: the same step time on the one-rank
RCCL path, one collective's fixed cost instead of four on a real ring).  On this model the whole exchange is ~1 ms
of a 98 ms step, while launching buckets from inside the backward pass (`overlap=True`) costs more than it hides: the
stream that completes a bucket has to wait for every other producer stream of that bucket, which ties the two view streams
and the weight-gradient lanes together (measured on one GPU with a one-rank RCCL group: +8.5 ms per step overlapped,
profiles/r03_dp_one_gpu.txt).  The overlapped schedule stays for models whose gradient volume is worth hiding.
Works unchanged on the gloo backend (CPU tests, world_size 2).
"""
import os

import torch
import torch.distributed as dist


# Tensors handed to a collective are kept alive by the work object instead of being recorded on RCCL's stream: a recorded
# block cannot be reused by the caching allocator until that stream has passed it (pool growth = hipMalloc stalls inside the
# step, profiles/r03_dp_one_gpu.txt).  ProcessGroupNCCL reads the variable when the group is created, so it is set here, at
# import of the trainers package (before any trainer can create a group), for training runs and bench.py alike; an explicit
# setting in the environment wins.
os.environ.setdefault("TORCH_NCCL_AVOID_RECORD_STREAMS", "1")

_SKIP_COLLECTIVE = os.environ.get("VQW_DP_DEBUG", "") == "noar"
_TINY = os.environ.get("VQW_DP_DEBUG", "") == "tiny"
_SYNC_AR = os.environ.get("VQW_DP_SYNC_AR", "1") != "0"     # after-backward schedule: the blocking form of the call
_HOST_TIMING = os.environ.get("VQW_DP_HOST_TIMING", "0") == "1"     # measurement aid: host time spent inside dist.all_reduce


class GradientAllReducer:
    host_ms_in_all_reduce = 0.0

    def __init__(self, params, bucket_bytes=None, process_group=None, overlap=None):
        self.overlap = (os.environ.get("VQW_DP_OVERLAP", "0") != "0") if overlap is None else bool(overlap)
        if bucket_bytes is None:
            # buckets exist to start exchanging early: the overlapped schedule takes 16 MiB ones; exchanged after the backward pass
            # the gradients go in ONE message (the largest the ring can get, and one collective's fixed cost instead of four).
            # VQW_DP_BUCKET_MB overrides (0 = one bucket).
            mb = os.environ.get("VQW_DP_BUCKET_MB")
            mb = (16 if self.overlap else 0) if mb is None else int(mb)
            bucket_bytes = (mb << 20) if mb > 0 else (1 << 62)
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.buckets = []       # list of lists of params
        cur, size = [], 0
        for p in params:
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                self._bucket_of[p] = bi
        self._pending = None
        self._work = []
        self._flat = [None] * len(self.buckets)
        self._flat_buf = [None] * len(self.buckets)
        # gradients that flow through autograd announce themselves via the hook; conv weight gradients written
        # out-of-band on the side stream announce themselves via hipops.ops.grad_ready_listeners (when importable)
        self._hooks = []
        self._events = {}
        self._armed = False
        self.launches = 0          # gradient-bucket collectives issued (tools/dp_probe)
        self._sync_lanes = None
        if self.overlap:
            self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]
            for p in params:       # hipops.ops.wgrad_through_autograd: these hooks do not force the autograd route
                p.__dict__["_vqw_own_hooks"] = p.__dict__.get("_vqw_own_hooks", 0) + 1
        try:
            from hipops import ops as _ops
            if self.overlap:
                _ops.grad_ready_listeners.append(self._on_grad_listener)
            self._sync_lanes = _ops.sync_wgrad_lanes
        except Exception:       # plain torch modules (CPU tests)
            pass

    def describe(self):
        """What a log line needs to be read without the source: schedule, bucket count and sizes."""
        sizes = [sum(p.numel() * p.element_size() for p in b) for b in self.buckets]
        return dict(dp_schedule="overlap" if self.overlap else "after_backward", gradient_buckets=len(self.buckets),
                    gradient_bucket_bytes=sizes, gradient_bytes=sum(sizes))

    def close(self):
        """Detach from the parameters: remove the hooks, give back the hook allowance, stop listening to the side stream."""
        for h in self._hooks:
            h.remove()
        if self._hooks:
            for b in self.buckets:
                for p in b:
                    n = p.__dict__.get("_vqw_own_hooks", 0)
                    if n > 0:
                        p.__dict__["_vqw_own_hooks"] = n - 1
        self._hooks = []
        try:
            from hipops import ops as _ops
            if self._on_grad_listener in _ops.grad_ready_listeners:
                _ops.grad_ready_listeners.remove(self._on_grad_listener)
        except Exception:
            pass
        self.buckets, self._bucket_of = [], {}

    def prepare(self):
        """Arm the hooks for one backward pass."""
        self._pending = [len(b) for b in self.buckets]
        self._seen = set()
        self._work = []
        self._armed = self.world > 1 or self._forced()

    @staticmethod
    def _forced():
        """hipops.ops.FORCE_COLLECTIVES: run the bucketed all-reduce in a one-rank group too (the RCCL path on one GPU)."""
        try:
            from hipops import ops as _ops
            return bool(_ops.FORCE_COLLECTIVES) and dist.is_initialized()
        except Exception:
            return False

    def _on_grad_listener(self, p):
        if p in self._bucket_of:
            self._on_grad(p)

    def _on_grad(self, p):
        if not self.overlap or not self._armed or id(p) in self._seen:
            return          # a parameter may be announced by both the autograd hook and the side-stream listener
        if p.grad is None:
            return          # hook fired for a gradient that is written out-of-band: wait for the listener
        self._seen.add(id(p))
        if p.grad.is_cuda:
            # the gradient was produced on the stream that announces it (a view's stream through autograd, a
            # weight-gradient lane out-of-band): the stream that flattens the bucket orders itself after every one of them
            self._events[id(p)] = torch.cuda.current_stream(p.grad.device).record_event()
        bi = self._bucket_of[p]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi):
        grads = [p.grad for p in self.buckets[bi]]
        if self.overlap and grads[0].is_cuda:
            cur = torch.cuda.current_stream(grads[0].device)
            for p in self.buckets[bi]:
                ev = self._events.pop(id(p), None)
                if ev is not None:
                    cur.wait_event(ev)
        # flatten in MEMORY order (grads may be channels_last): view each as its dense storage
        views = [_dense_1d(g) for g in grads]
        # One persistent flat buffer per bucket: a fresh 16 MiB tensor per step that ProcessGroupNCCL records on its own
        # stream cannot be reused by the caching allocator until that stream has passed it, so the pool grew by a device
        # segment (hipMalloc, a device-wide stall) in most steps: 17 segments / +7 GB inside ten timed steps against 5 of
        # the plain step, which was most of the data-parallel step's single-GPU cost (profiles/r03_dp_one_gpu.txt).
        flat = self._flat_buf[bi]
        n = sum(v.numel() for v in views)
        if flat is None or flat.numel() != n or flat.device != views[0].device or flat.dtype != views[0].dtype:
            flat = self._flat_buf[bi] = torch.empty(n, dtype=views[0].dtype, device=views[0].device)
        torch.cat(views, out=flat)
        self._flat[bi] = (flat, views)
        # async: on RCCL the collective runs on the process group's own stream, ordered after the producing
        # kernels by the event torch.distributed records, so it overlaps with the rest of backward
        self.launches += 1
        if _HOST_TIMING:
            import time
            t0 = time.perf_counter()
        if _SKIP_COLLECTIVE:       # measurement aid: flatten / scale / scatter without the collective (wrong on > 1 rank)
            self._work.append((bi, None))
            return
        # (VQW_DP_DEBUG=tiny: the collective on 1 KiB of the bucket only - a measurement aid, wrong on > 1 rank: is the one-rank
        # group's cost per call or per byte?)
        # The after-backward schedule has nothing to overlap the exchange with and takes the BLOCKING form of the call (it still
        # only enqueues: the host does not wait for the GPU): on the one-rank RCCL path the asynchronous form + work.wait() costs
        # 6-8 ms per step whatever the message size or count (95.6-98.1 against 89.4 ms; the plain step: 88.9), the blocking
        # form nothing.  VQW_DP_SYNC_AR=0 restores the asynchronous form there (A/B).
        use_async = self.overlap or not _SYNC_AR
        work = dist.all_reduce(flat[:256] if _TINY else flat, op=dist.ReduceOp.SUM, group=self.group, async_op=use_async)
        if not use_async:
            work = None
        if _HOST_TIMING:
            self.host_ms_in_all_reduce += (time.perf_counter() - t0) * 1e3
        self._work.append((bi, work))

    def finish(self):
        """Wait for every bucket, write the rank-mean back into p.grad."""
        if not self._armed:
            return
        if not self.overlap:
            # every gradient has been enqueued and the caller's stream is ordered after all producer streams
            # (FirstStepTrainer joins the view stream and the weight-gradient lanes first): exchange the buckets now
            for bi, b in enumerate(self.buckets):
                if any(p.grad is None for p in b):
                    raise RuntimeError("gradient bucket %d incomplete: a parameter received no gradient" % bi)
                self._launch(bi)
        for bi, pend in enumerate(self._pending):
            if self.overlap and pend != 0:      # parameters without gradient this step
                raise RuntimeError("gradient bucket %d incomplete: a parameter received no gradient" % bi)
        inv = 1.0 / self.world
        for bi, work in self._work:
            if work is not None:
                work.wait()
            flat, views = self._flat[bi]
            flat.mul_(inv)
            pieces, off = [], 0
            for v in views:
                n = v.numel()
                pieces.append(flat[off:off + n])
                off += n
            torch._foreach_copy_(views, pieces)        # one multi-tensor launch per bucket, not two per parameter
            self._flat[bi] = None
        self._armed = False


def _dense_1d(t):
    """1-D view over a dense tensor's storage in memory order (contiguous or channels_last)."""
    if t.is_contiguous():
        return t.view(-1)
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return t.permute(0, 2, 3, 1).reshape(-1)      # a view: permute makes it contiguous in memory order
    raise RuntimeError("gradient tensor is not dense")
