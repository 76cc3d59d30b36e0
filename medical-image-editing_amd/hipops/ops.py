"""Differentiable operators over the C ABI of libvqwnet_hip.so.

Every function takes / returns torch tensors that live on a ROCm device; PyTorch is
used only for device memory (caching allocator), the current stream and the
autograd tape.  All arithmetic happens in the hand-written HIP kernels; a missing
library or a CPU tensor raises (no eager fallback).

4-D activations are logically NCHW and physically NHWC (torch.channels_last).
Conv weights are logically OIHW and physically OHWI (channels_last too).
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import _lib

CL = torch.channels_last


# Kernels are reached through the PyTorch dispatcher: every entry point of the C ABI is an operator torch.ops.vqw.<name>
# with a schema derived from its prototype (hipops/library.py; mutation annotations from the `const` qualifiers).
# VQW_DISPATCH=0 calls the C functions directly through ctypes instead (A/B of the host-side cost).
DISPATCH = os.environ.get("VQW_DISPATCH", "1") != "0"
_dispatch = None


def _L():
    global _dispatch
    if not DISPATCH:
        return _lib.load()
    if _dispatch is None:
        from . import library
        _dispatch = library.Dispatch()
    return _dispatch


if DISPATCH:
    from .library import Dispatch as _D

    def _st():
        return _D.STREAM

    def _p(t):
        return t
else:
    def _st():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _p(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _raw(t):
    """Raw device pointer for the few host-side entry points that take pointer arrays by value."""
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _raw_stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("hipops operators need tensors on a ROCm device (got %s); "
                               "there is no CPU fallback" % t.device)


def nhwc(x):
    """Dense fp32 NHWC view of a 4-D tensor (no copy when it already is)."""
    if x.dtype != torch.float32:
        raise RuntimeError("hipops operators are fp32 (got %s)" % x.dtype)
    if x.dim() != 4:
        raise RuntimeError("expected a 4-D (N,C,H,W) tensor, got shape %s" % (tuple(x.shape),))
    if not x.is_contiguous(memory_format=CL):
        x = x.contiguous(memory_format=CL)
    # size-1 dims make PyTorch's stride check ambiguous; normalise the strides we rely on
    N, C, H, W = x.shape
    if x.stride() != (H * W * C, 1, W * C, C):
        x = x.as_strided((N, C, H, W), (H * W * C, 1, W * C, C))
    return x


def empty_nhwc(N, C, H, W, like):
    return torch.empty((N, C, H, W), dtype=torch.float32, device=like.device, memory_format=CL)


def _ws(nbytes, like):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=like.device)


def _flat(t):
    if t.dtype != torch.float32:
        raise RuntimeError("expected fp32")
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------------------------
# weight-gradient side stream
# ----------------------------------------------------------------------------------------------
# In backward the weight gradient of a conv is a leaf of the dependency chain: nothing downstream needs it before the
# optimiser.  It is therefore enqueued on a side HIP stream, where the MFMA-bound wgrad kernels overlap with the
# HBM-bound normalisation / element-wise backward kernels of the main chain, and written straight into `param.grad`
# (overwrite on the first use of a step, in-kernel accumulate afterwards — both views of a step share one buffer, no
# autograd add pass).  The main stream re-joins the side stream at the end of the backward pass (engine callback).
# VQW_WGRAD_STREAM=0 disables it (weight gradients then flow through autograd as usual).
WGRAD_ASYNC = os.environ.get("VQW_WGRAD_STREAM", "1") != "0"
_side_streams = {}
_join_queued_for = None        # id of the backward pass (autograd graph task) whose end-of-pass lane join is queued
grad_ready_listeners = []      # callables(param): the param's gradient is final and enqueued on the side stream


# The DDP detection below reads a private class attribute of torch.  On a torch build without it the check would silently
# say "no DDP" and the reducer would never see the conv weight gradients (hang or unsynchronised training), so its absence
# switches the out-of-band route off for good instead: every weight gradient then flows through autograd, which is always
# correct.  set_wgrad_route("autograd") is the explicit form a wrapper can call.
_DDP = torch.nn.parallel.DistributedDataParallel
_WGRAD_ROUTE = "auto" if hasattr(_DDP, "_active_ddp_module") else "autograd"


def set_wgrad_route(route):
    """"auto": conv weight gradients of leaf parameters are written out-of-band on the side stream unless a DDP forward is
    active or the parameter carries foreign hooks; "autograd": always through autograd.  Returns the previous route."""
    global _WGRAD_ROUTE
    if route not in ("auto", "autograd"):
        raise ValueError("wgrad route must be 'auto' or 'autograd'")
    old, _WGRAD_ROUTE = _WGRAD_ROUTE, route
    return old


def wgrad_through_autograd(*params):
    """True when weight gradients must flow through autograd instead of being written out-of-band on a side stream:
    inside a torch DistributedDataParallel forward (its reducer learns about a gradient from the AccumulateGrad hook of
    the parameter; the reference launches under Lightning's DDPPlugin, run_vqwnet.py:112-121) or when somebody
    registered a hook on the parameter."""
    if _WGRAD_ROUTE == "autograd" or getattr(_DDP, "_active_ddp_module", None) is not None:
        return True
    for p in params:
        if p is None:
            continue
        # (the data-parallel reducer of this package listens on both routes and marks its own hooks)
        if p._backward_hooks or len(getattr(p, "_post_accumulate_grad_hooks", None) or ()) > p.__dict__.get("_vqw_own_hooks", 0):
            return True
    return False


def _order_begin(token):
    """In-place updates of a module buffer (BN running stats, VQ codebook) keep host program order even when the two
    views of a step run on different streams: wait for the event the previous updater left on the buffer."""
    prev = token.__dict__.get("_vqw_order")
    cur = torch.cuda.current_stream()
    if prev is not None and prev[1] != cur:
        cur.wait_event(prev[0])
    return cur


def _order_end(token, cur):
    token.__dict__["_vqw_order"] = (cur.record_event(), cur)


def reset_pending(params):
    """Forget forward passes that were never back-propagated (call before the forwards of a training step)."""
    for p in params:
        if hasattr(p, "_vqw_pending"):
            p._vqw_pending = 0


# Weight gradients alternate between WGRAD_LANES side streams (a parameter keeps its lane: the second view accumulates
# into the first view's result).  One layer's slab reduction (a short, latency-bound launch that depends on the layer's
# wgrad kernel) then overlaps the next layer's wgrad kernel; on a single stream those gaps were exposed, most of all
# in the last tenth of the step when only weight gradients are left to run.  A gradient-ready listener that reads
# gradients of several parameters (data parallel: a bucket is flattened on the lane that announces its last gradient)
# orders its lane after the others first: sync_wgrad_lanes().
WGRAD_LANES = max(1, int(os.environ.get("VQW_WGRAD_LANES", "2")))
_lane_counter = 0


def wgrad_stream(device, lane=0):
    dev = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    key = (dev, lane)
    st = _side_streams.get(key)
    if st is None:
        st = torch.cuda.Stream(device=dev)
        _side_streams[key] = st
    return st


def sync_wgrad_lanes():
    """Order the current stream after everything enqueued so far on the weight-gradient lanes."""
    if not _side_streams:
        return
    cur = torch.cuda.current_stream()
    for st in _side_streams.values():
        if st.device == cur.device and st != cur:
            cur.wait_stream(st)


def _wgrad_lane(param):
    global _lane_counter
    if WGRAD_LANES == 1:
        return 0
    lane = param.__dict__.get("_vqw_lane")
    if lane is None:
        lane = param.__dict__["_vqw_lane"] = _lane_counter % WGRAD_LANES
        _lane_counter += 1
    return lane


def _pass_id():
    """Identity of the running backward pass (autograd graph task); -1 outside of one."""
    return torch._C._current_graph_task_id()


def _join_side_stream():
    """Order the current stream after the weight-gradient lanes (idempotent), then fold the split-K slabs the lanes' weight
    gradient kernels have left (one launch for the whole pass, see _fold_flush)."""
    global _join_queued_for
    _join_queued_for = None
    for st in _side_streams.values():
        torch.cuda.current_stream(st.device).wait_stream(st)
    _fold_flush()


# Deferred slab folds (VQW_FOLD_DEFER=0: every weight-gradient call folds its own slabs, two short launches per layer).
# A weight-gradient kernel leaves split-K slabs; folding them is a short, dependent launch behind every one of the ~110
# weight-gradient kernels of a step (dW and dbias: ~250 launches).  On the lanes the library only records the folds
# (vqw_fold_defer) and the end-of-pass lane join folds them all in ONE launch; the two views' folds into one gradient are
# summed in recording order (overwrite, then accumulate) - the same values in a fixed order.  The slab buffers are kept alive
# here until that launch is enqueued.  Off while a gradient-ready listener is registered (the overlapped data-parallel
# schedule needs a parameter's gradient final when it is announced).
FOLD_DEFER = os.environ.get("VQW_FOLD_DEFER", "1") != "0"
_fold_keep = []                # slab workspaces of the recorded folds
_fold_slots = []               # ring of [pinned host table, device table, event of the launch that read them]
_fold_next = 0
fold_flushes = 0               # batched fold launches since import (tests)


def _fold_active():
    return FOLD_DEFER and not grad_ready_listeners


def _fold_flush():
    """Fold every recorded slab set in one launch on the current stream (which the caller has ordered after the lanes)."""
    global _fold_next, fold_flushes
    lib = _lib.load()
    if not _fold_keep and lib.vqw_fold_pending() == 0:
        return
    need = int(lib.vqw_fold_table_bytes())
    cur = torch.cuda.current_stream()
    if len(_fold_slots) < 4:
        _fold_slots.append(None)
        _fold_next = len(_fold_slots) - 1
    slot = _fold_slots[_fold_next]
    if slot is not None:
        slot[2].synchronize()          # the launch that read this slot's tables has run (only ever waits when > 4 passes are in flight)
    if slot is None or slot[0].numel() < need or slot[1].device != cur.device:
        size = max(need, 1 << 16)
        slot = [torch.empty(size, dtype=torch.uint8, pin_memory=True), torch.empty(size, dtype=torch.uint8, device=cur.device), None]
    _lib.check(lib.vqw_fold_flush_host(ctypes.c_void_p(slot[0].data_ptr()), ctypes.c_void_p(slot[1].data_ptr()), slot[0].numel(),
                                       ctypes.c_void_p(cur.cuda_stream)), "vqw_fold_flush_host")
    slot[2] = cur.record_event()
    _fold_slots[_fold_next] = slot
    _fold_next = (_fold_next + 1) % 4
    fold_flushes += 1
    # the slab buffers belong to the lanes' memory pools: order the lanes after this launch before the pools may reuse them
    for st in _side_streams.values():
        if st.device == cur.device:
            st.wait_event(slot[2])
    _fold_keep.clear()


def _queue_lane_join():
    """Queue the lane join as an end-of-pass callback, once per backward pass.  The flag holds the pass's id, not a bool: a
    pass that ended in an exception (the engine skips every callback queued behind one that raises) cannot latch it."""
    global _join_queued_for
    tid = _pass_id()
    if _join_queued_for != tid:
        _join_queued_for = tid
        torch.autograd.Variable._execution_engine.queue_callback(_join_side_stream)


# ----------------------------------------------------------------------------------------------
# branch streams: independent sub-graphs of one forward pass on a second HIP stream
# ----------------------------------------------------------------------------------------------
# The SPADE modulation maps of the decoder depend only on the skip features, not on the up-path trunk; run on a branch
# stream their MFMA-bound convolutions overlap the trunk's HBM-bound normalisation kernels.  Autograd replays each
# node on the stream its forward ran on and orders the streams itself, so the backward pass forks the same way.
# Off by default: measured at B=32, 256x256 it gains 2.5 % when the two views run one after the other (196 vs 191
# img/s) but LOSES 8 % on top of the two-view streams (183 vs 200 img/s: five streams of persistent, LDS-filling conv
# kernels starve each other), and two concurrent views alone are the fastest arrangement.  VQW_BRANCH_STREAMS=1 enables.
BRANCH_STREAMS = os.environ.get("VQW_BRANCH_STREAMS", "0") != "0"
_branch_streams = {}


class Branch:
    """with Branch(inputs...) as b: outs = f(...)   # on the branch stream of the current stream
    b.join(outs...)                                  # current stream waits; tensors become usable on it"""

    def __init__(self, *inputs):
        self.inputs = [t for t in inputs if t is not None]
        self.active = BRANCH_STREAMS and bool(self.inputs) and self.inputs[0].is_cuda

    def __enter__(self):
        if not self.active:
            return self
        self.cur = torch.cuda.current_stream()
        key = (self.cur.device.index, self.cur.cuda_stream)
        br = _branch_streams.get(key)
        if br is None:
            br = torch.cuda.Stream(device=self.cur.device, priority=self.cur.priority)
            _branch_streams[key] = br
        self.br = br
        br.wait_event(self.cur.record_event())
        for t in self.inputs:
            t.record_stream(br)
        self.ctx = torch.cuda.stream(br)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.active:
            self.ev = self.br.record_event()
            self.ctx.__exit__(*exc)
        return False

    def join(self, *outs):
        if self.active:
            self.cur.wait_event(self.ev)
            for t in outs:
                if t is not None:
                    t.record_stream(self.cur)


def join_streams():
    """Order the current stream after every branch stream (end of a training step / before reading results elsewhere)."""
    cur = torch.cuda.current_stream()
    for br in _branch_streams.values():
        if br.device == cur.device:
            cur.wait_stream(br)
    global _defer_counters
    _defer_counters = False
    flush_counters()


# Derived weight layouts (dgrad-packed, tap-collapsed) are cached on the parameter until it changes: both views of a
# step reuse them.  A parameter "changes" when torch bumps its version counter or when hipops.Adam (which updates
# through raw pointers) advances the epoch below.
_weight_epoch = 0


def bump_weight_epoch():
    global _weight_epoch
    _weight_epoch += 1


_CACHE_ON = os.environ.get("VQW_WEIGHT_CACHE", "1") != "0"


def _cached(weight, key, build, deps=()):
    """`deps`: further tensors the derived value is built from (concatenated convs)."""
    if not _CACHE_ON:
        return build()
    cache = weight.__dict__.setdefault("_vqw_cache", {})
    tag = (weight._version, _weight_epoch, weight.data_ptr()) + tuple((d._version, d.data_ptr()) for d in deps)
    cur = torch.cuda.current_stream()
    hit = cache.get(key)
    if hit is not None and hit[0] == tag:
        if hit[3] != cur:            # built on another stream (the other view): order this stream after the build
            cur.wait_event(hit[2])
        return hit[1]
    val = build()
    cache[key] = (tag, val, cur.record_event(), cur)
    return val


def _run_wgrad(L, x0, x1, gy, gw, gb, up0, ks, dilation, N, H, W, Cout, acc, collapsed, defer_fold=False):
    """dW / db on the CURRENT stream; `collapsed` selects the low-resolution form for 3x3-over-upsampled layers.
    defer_fold: the slab folds are only recorded (the caller has queued the end-of-pass lane join, which runs them)."""
    C0 = x0.shape[1]
    C1 = 0 if x1 is None else x1.shape[1]
    lib = _lib.load()
    defer_fold = defer_fold and _fold_active()
    for attempt in (0, 1):
        old = lib.vqw_fold_defer(1) if defer_fold else 0
        try:
            if collapsed and L.vqw_conv3x3_up2_wgrad_supported(C0, Cout, N, H // 2, W // 2):
                ws = _ws(L.vqw_conv3x3_up2_wgrad_ws_bytes(C0, Cout, N, H // 2, W // 2), gy)
                _lib.check(L.vqw_conv3x3_up2_wgrad(_p(x0), _p(gy), _p(gw), _p(gb), _p(ws), ws.numel(), N, H // 2, W // 2, C0, Cout,
                                                   int(acc), _st()), "vqw_conv3x3_up2_wgrad")
            else:
                ws = _ws(L.vqw_conv2d_wgrad_ws_bytes(C0, C1, N, H, W, Cout, ks), gy)
                _lib.check(L.vqw_conv2d_wgrad(_p(x0), C0, int(up0), _p(x1), C1, _p(gy), _p(gw), _p(gb), _p(ws), ws.numel(),
                                              N, H, W, Cout, ks, dilation, int(acc), _st()), "vqw_conv2d_wgrad")
        except RuntimeError as e:
            if defer_fold and attempt == 0 and "flush first" in str(e):
                # a third use of one weight inside a pass: fold what is recorded (on this lane, after the other lanes), retry
                lib.vqw_fold_defer(old)
                sync_wgrad_lanes()
                _fold_flush()
                continue
            raise
        finally:
            if defer_fold:
                lib.vqw_fold_defer(old)
        break
    if defer_fold:
        _fold_keep.append(ws)


def _deferred_wgrad(weight, bias, x0, x1, gy, up0, ks, dilation, N, H, W, Cout, collapsed=False):
    """Enqueue dW (and db) on the side stream, writing into weight.grad / bias.grad."""
    L = _L()
    main = torch.cuda.current_stream()
    side = wgrad_stream(gy.device, _wgrad_lane(weight))
    ev = main.record_event()
    C0 = x0.shape[1]
    C1 = 0 if x1 is None else x1.shape[1]
    for t in (x0, x1, gy):
        if t is not None:
            t.record_stream(side)
    with torch.cuda.stream(side):
        side.wait_event(ev)
        acc = weight.grad is not None
        if not acc:
            # (the parameter's own strides: channels_last for a 3 x 3 weight, the default ones for a 1 x 1 weight - the same memory order)
            weight.grad = torch.empty_strided((Cout, C0 + C1, ks, ks), weight.stride(), dtype=torch.float32, device=gy.device)
            if bias is not None:
                bias.grad = torch.empty(Cout, dtype=torch.float32, device=gy.device)
        gw, gb = weight.grad, (bias.grad if bias is not None else None)
        if gw.stride() != weight.stride():
            raise RuntimeError("conv2d: existing weight.grad layout does not match the parameter layout")
        _run_wgrad(L, x0, x1, gy, gw, gb, up0, ks, dilation, N, H, W, Cout, acc, collapsed, defer_fold=True)
        weight._vqw_pending = getattr(weight, "_vqw_pending", 1) - 1
        if weight._vqw_pending <= 0:
            weight._vqw_pending = 0
            for fn in grad_ready_listeners:
                fn(weight)
                if bias is not None:
                    fn(bias)
    _queue_lane_join()


# ----------------------------------------------------------------------------------------------
# convolution
# ----------------------------------------------------------------------------------------------
def _wino_weights(L, w_ohwi, Cin, Cout):
    """U = G w G^T [16][Cout][Cin] of a 3x3 layer (vqw_conv3x3_wino_prepare) in a fresh buffer."""
    buf = _ws(L.vqw_conv3x3_wino_ws_bytes(Cin, Cout), w_ohwi)
    _lib.check(L.vqw_conv3x3_wino_prepare(_p(w_ohwi), _p(buf), buf.numel(), Cin, Cout, _st()), "vqw_conv3x3_wino_prepare")
    return buf


def _wino_weights_dgrad(L, w_ohwi, Cin, Cout):
    """U of a 3x3 layer's INPUT-GRADIENT convolution (Cin = the layer's couts, Cout = its input channels), straight from the
    layer's weight: what _wino_weights gives on the packed input-gradient weights, without the pack launch."""
    buf = _ws(L.vqw_conv3x3_wino_ws_bytes(Cin, Cout), w_ohwi)
    _lib.check(L.vqw_conv3x3_wino_prepare_dgrad(_p(w_ohwi), _p(buf), buf.numel(), Cin, Cout, _st()), "vqw_conv3x3_wino_prepare_dgrad")
    return buf


def _conv_fwd_raw(x0, up0, x1, w, bias, N, H, W, Cout, ks, dil, relu=False):
    y = empty_nhwc(N, Cout, H, W, x0)
    _lib.check(_L().vqw_conv2d_fwd(_p(x0), x0.shape[1], int(up0), _p(x1), 0 if x1 is None else x1.shape[1],
                                   _p(w), _p(bias), _p(y), N, H, W, Cout, ks, dil, int(relu), _st()), "vqw_conv2d_fwd")
    return y


CONV_STATS = os.environ.get("VQW_CONV_STATS", "1") != "0"      # 0: InstanceNorm always reduces itself (A/B timing)
# Winograd F(2x2, 3x3) form of plain 3x3 layers.  The INPUT and WEIGHT GRADIENTS take it whenever the library serves the
# shape (VQW_WINOGRAD=0 turns the kernels off altogether): they are linear in dy given the forward's masks, so the form's
# rounding difference (a few ulps of the accumulated magnitude) reaches the parameter gradients unamplified.  The training
# FORWARD does not by default: the Winograd form computes the four pixels of a tile by four different formulas, so equal
# inputs no longer give bit-equal outputs, and behind the quantised (piecewise constant) map the reference's max-pools sit
# on exact ties - on the config-4 step fixture the broken ties moved decoder gradients 3-5x the reference's own fp32 spread
# away from its fp64 gradient.  VQW_WINOGRAD_FWD=1 opts in (throughput runs).
WINOGRAD_FWD = os.environ.get("VQW_WINOGRAD_FWD", "0") == "1"
# Forward-only uses (torch.no_grad(): validation forward, run_recon) have no gradient to disturb and take the Winograd
# forward by themselves (VQW_WINOGRAD_EVAL=0: direct form there too).
WINOGRAD_EVAL = os.environ.get("VQW_WINOGRAD_EVAL", "1") != "0"
_in_custom_op = False          # set by hipops.functional around its no_grad() calls: those are training forwards


WINOGRAD_FWD_ENCODER = os.environ.get("VQW_WINOGRAD_FWD_ENCODER", "0") == "1"
WINOGRAD_FWD_POOLFREE = os.environ.get("VQW_WINOGRAD_FWD_POOLFREE", "1") != "0"      # decoder layers past its last max-pool
_wino_fwd_scope = 0            # > 0 inside `with winograd_forward():`


class winograd_forward:
    """Context: plain 3x3 layers called inside take the Winograd forward also in training.  For sub-networks whose
    activations are not piecewise constant (the encoder: its input is an image).  The Winograd form computes the four pixels
    of a 2x2 tile by four different formulas, so mathematically equal outputs are no longer bit-equal; behind the quantised
    (piecewise constant) map the reference's max-pools sit on exact ties and a broken tie re-routes gradients."""

    def __enter__(self):
        global _wino_fwd_scope
        _wino_fwd_scope += 1

    def __exit__(self, *exc):
        global _wino_fwd_scope
        _wino_fwd_scope -= 1


def _decide_wino_fwd():
    """The per-call decision, handed to the autograd Functions as an explicit (non-differentiable) argument: a caller that
    reaches them without the wrapper, or a second host thread, cannot inherit another call's decision."""
    return bool(WINOGRAD_FWD or _wino_fwd_scope > 0 or (WINOGRAD_EVAL and not torch.is_grad_enabled() and not _in_custom_op))


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, x1, weight, bias, dilation, up0, relu, want_stats=False, wino_fwd=False, grad_group=None, in_src=None):
        _dev(x0, x1, weight, bias)
        ctx.in_src = None
        if in_src is not None and x1 is None and not up0 and grad_group is None and dilation == 1 and weight.shape[2] == 3 \
                and ctx.needs_input_grad[0] and tuple(in_src[0].shape) == tuple(x0.shape) \
                and _L().vqw_conv3x3_wino_fwd_inbwd_parts(weight.shape[0], weight.shape[1], x0.shape[0], x0.shape[2], x0.shape[3]) > 0:
            ctx.in_src = in_src            # (raw input of the InstanceNorm whose output x0 is, its (mean, rstd), its ReLU flag)
        x0 = nhwc(x0)
        x1 = nhwc(x1) if x1 is not None else None
        w = nhwc(weight)
        Cout, Cin, ks, _ = weight.shape
        if w is not weight and ks == 1 and weight.is_contiguous() and w.data_ptr() == weight.data_ptr():
            # a 1 x 1 weight is dense in both memory formats and keeps the default strides: nhwc() re-strides the same memory.  The
            # parameter itself serves (the kernels take pointers): its derived layouts stay cached on it from step to step and its
            # gradient takes the side-lane route like every other layer's
            w = weight
        N = x0.shape[0]
        H, W = (x0.shape[2] * 2, x0.shape[3] * 2) if up0 else (x0.shape[2], x0.shape[3])
        c1 = 0 if x1 is None else x1.shape[1]
        part = None
        if x0.shape[1] + c1 != Cin:
            raise RuntimeError("conv2d: input channels %d+%d do not match weight %s" % (x0.shape[1], c1, tuple(weight.shape)))
        if x1 is not None and (x1.shape[0] != N or x1.shape[2] != H or x1.shape[3] != W):
            raise RuntimeError("conv2d: concat source shape %s does not match %s" % (tuple(x1.shape), (N, c1, H, W)))
        if bias is not None:
            bias = _flat(bias)
        # 3x3 over a nearest x2 up-sampled single source: collapsed onto the low-res grid (4/9 of the FLOPs)
        up_ws = None
        if up0 and x1 is None and ks == 3 and dilation == 1 and \
                _L().vqw_conv3x3_up2_supported(Cin, Cout, N, H // 2, W // 2):
            L = _L()

            def _collapse():
                buf = _ws(L.vqw_conv3x3_up2_ws_bytes(Cin, Cout), x0)
                _lib.check(L.vqw_conv3x3_up2_prepare(_p(w), _p(buf), buf.numel(), Cin, Cout, _st()), "vqw_conv3x3_up2_prepare")
                return buf
            up_ws = _cached(weight, "up2", _collapse)
            y = empty_nhwc(N, Cout, H, W, x0)
            nparts = L.vqw_conv3x3_up2_fwd_stats_parts(Cin, Cout, N, H // 2, W // 2) if (want_stats and not relu) else 0
            if nparts > 0:
                part = torch.empty(N * nparts * Cout * 2, dtype=torch.float32, device=x0.device)
                _lib.check(L.vqw_conv3x3_up2_fwd_stats(_p(x0), _p(up_ws), _p(bias), _p(y), _p(part), N, H // 2, W // 2, Cin, Cout, _st()),
                           "vqw_conv3x3_up2_fwd_stats")
            else:
                _lib.check(L.vqw_conv3x3_up2_fwd(_p(x0), _p(up_ws), _p(bias), _p(y), N, H // 2, W // 2, Cin, Cout, int(relu), _st()),
                           "vqw_conv3x3_up2_fwd")
        elif wino_fwd and not up0 and x1 is None and ks == 3 and dilation == 1 and \
                _L().vqw_conv3x3_wino_supported(Cin, Cout, N, H, W) and \
                (not (want_stats and not relu) or _L().vqw_conv3x3_wino_fwd_stats_parts(Cin, Cout, N, H, W) > 0
                 or _L().vqw_conv2d_fwd_stats_parts(Cin, 0, 0, N, H, W, Cout, ks, dilation) == 0):
            # (wanted statistics that only the direct form can leave for this height keep the direct form)
            # plain 3x3 layer: Winograd F(2x2, 3x3), 4/9 of the matrix work; U = G w G^T is kept while w is unchanged
            L = _L()
            u = _cached(weight, "wino", lambda: _wino_weights(L, w, Cin, Cout))
            y = empty_nhwc(N, Cout, H, W, x0)
            nparts = L.vqw_conv3x3_wino_fwd_stats_parts(Cin, Cout, N, H, W) if (want_stats and not relu) else 0
            if nparts > 0:
                part = torch.empty(N * nparts * Cout * 2, dtype=torch.float32, device=x0.device)
                _lib.check(L.vqw_conv3x3_wino_fwd_stats(_p(x0), _p(u), _p(bias), _p(y), _p(part), N, H, W, Cin, Cout, _st()),
                           "vqw_conv3x3_wino_fwd_stats")
            else:
                _lib.check(L.vqw_conv3x3_wino_fwd(_p(x0), _p(u), _p(bias), _p(y), N, H, W, Cin, Cout, int(relu), _st()),
                           "vqw_conv3x3_wino_fwd")
        elif wino_fwd and not up0 and x1 is None and ks == 3 and dilation == 2 and \
                _L().vqw_conv3x3_wino_dil2_supported(Cin, Cout, N, H, W) and \
                (not (want_stats and not relu) or _L().vqw_conv3x3_wino_dil2_stats_parts(Cin, Cout, N, H, W) > 0):
            # dilation 2: the plain Winograd kernel on the four phase images of the tensors (same U as the plain layer)
            L = _L()
            u = _cached(weight, "wino", lambda: _wino_weights(L, w, Cin, Cout))
            y = empty_nhwc(N, Cout, H, W, x0)
            nparts = L.vqw_conv3x3_wino_dil2_stats_parts(Cin, Cout, N, H, W) if (want_stats and not relu) else 0
            if nparts > 0:
                part = torch.empty(N * nparts * Cout * 2, dtype=torch.float32, device=x0.device)
            _lib.check(L.vqw_conv3x3_wino_dil2_fwd(_p(x0), _p(u), _p(bias), _p(y), _p(part) if nparts > 0 else None, 0, N, H, W, Cin, Cout,
                                                   int(relu), _st()), "vqw_conv3x3_wino_dil2_fwd")
        else:
            nparts = 0
            if want_stats and not relu:
                nparts = _L().vqw_conv2d_fwd_stats_parts(x0.shape[1], c1, int(up0), N, H, W, Cout, ks, dilation)
            if nparts > 0:      # the epilogue leaves the following InstanceNorm's statistics (per-tile partial sums)
                y = empty_nhwc(N, Cout, H, W, x0)
                part = torch.empty(N * nparts * Cout * 2, dtype=torch.float32, device=x0.device)
                _lib.check(_L().vqw_conv2d_fwd_stats(_p(x0), x0.shape[1], int(up0), _p(x1), c1, _p(w), _p(bias), _p(y), _p(part),
                                                     N, H, W, Cout, ks, dilation, _st()), "vqw_conv2d_fwd_stats")
            else:
                y = _conv_fwd_raw(x0, up0, x1, w, bias, N, H, W, Cout, ks, dilation, relu)
        ctx.up_ws = up_ws
        ctx.group = grad_group
        ctx.save_for_backward(x0, x1, w, y if relu else None)
        ctx.cfg = (dilation, up0, ks, N, H, W, Cout, bias is not None)
        # leaf parameters get their gradient written out-of-band on the side stream (see _deferred_wgrad)
        ctx.defer = (WGRAD_ASYNC and ctx.needs_input_grad[2] and weight.is_leaf and w is weight and not wgrad_through_autograd(weight, bias)
                     and (bias is None or (bias.is_leaf and bias.is_contiguous())))
        if ctx.defer:
            ctx.params = (weight, bias)
            weight._vqw_pending = getattr(weight, "_vqw_pending", 0) + 1
        if want_stats:
            if part is not None:
                ctx.mark_non_differentiable(part)
            # (without this autograd hands backward a zero-filled "gradient" of the statistics partials: one fill launch per
            # convolution and step, 108 of them)
            ctx.set_materialize_grads(False)
            return y, part
        return y

    @staticmethod
    def backward(ctx, gy, *_):
        if gy is None:             # y itself was not used (only possible with gradient materialisation off)
            return (None,) * 11
        x0, x1, w, y_relu = ctx.saved_tensors
        dilation, up0, ks, N, H, W, Cout, has_bias = ctx.cfg
        need0, need1, needw, needb = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        defer = ctx.defer and (needw or (needb and has_bias))
        g0, g1, gw, gb, gy = conv2d_backward_impl(gy, x0, x1, w, y_relu, dilation, up0, has_bias, ctx.up_ws, need0, need1,
                                                  needw and not defer, needb and not defer, group=ctx.group, in_src=ctx.in_src)
        if defer:
            weight, bias = ctx.params
            _deferred_wgrad(weight, bias if (has_bias and bias.requires_grad) else None, x0, x1, gy, up0, ks, dilation,
                            N, H, W, Cout, collapsed=ctx.up_ws is not None)
        return g0, g1, gw, gb, None, None, None, None, None, None, None


class _GradNotes:
    """Notes that a consumer's input-gradient launch leaves for the producer's backward of the SAME backward pass, keyed by the
    gradient tensor it hands to autograd.  A note is honoured only when the tensor that reaches the producer is that very
    gradient, untouched: same address, same version counter (autograd's InputBuffer sums the gradients of a tensor with
    several consumers IN PLACE when it owns the first one - `old.add_(new)` keeps the address and bumps the version), same
    backward pass (addresses recur from step to step under the caching allocator).  Whatever is left at the end of a pass is
    dropped by an end-of-pass callback, and begin_step() drops it again (the engine skips callbacks behind one that raises)."""

    def __init__(self):
        self.notes = {}
        self._armed_for = None

    def put(self, grad, payload):
        tid = _pass_id()
        if tid >= 0 and self._armed_for != tid:      # (outside a backward pass nobody will take the note: begin_step() drops it)
            self._armed_for = tid
            torch.autograd.Variable._execution_engine.queue_callback(self.clear)
        self.notes[grad.data_ptr()] = (grad._version, tid, payload)

    def take(self, grad):
        ent = self.notes.pop(grad.data_ptr(), None)
        if ent is None or ent[0] != grad._version or ent[1] != _pass_id():
            return None
        return ent[2]

    def clear(self):
        self.notes.clear()
        self._armed_for = None

    def __len__(self):
        return len(self.notes)


# ----------------------------------------------------------------------------------------------
# two 32-cout 3x3 layers of one up-sampled input as ONE 64-cout launch (StyledResUpBlock's shortcut `conv` and `conv1`)
# ----------------------------------------------------------------------------------------------
UP_PAIR = os.environ.get("VQW_UP_PAIR", "1") != "0"      # 0: the two layers run one by one (A/B timing)
up_pair_calls = 0


class _ConvUpPair(torch.autograd.Function):
    """(y_a, part_a, y_b, part_b) = the forward of conv2d(x, w_a, b_a, up2x=True, want_stats=True) and of the same with (w_b, b_b),
    computed by one launch of the nine-product kernel on the concatenated weights (a 32-cout layer alone falls back to the
    collapsed 4-tap form at a third of that kernel's rate).  The backward is the two layers' own: each input gradient through
    its layer's own collapsed / nine-product weights (the second added to the first in its kernel's epilogue when `group` is
    given), each weight gradient on the side lanes."""

    @staticmethod
    def forward(ctx, x, wa, ba, wb, bb, group):
        global up_pair_calls
        _dev(x, wa, ba, wb, bb)
        x = nhwc(x)
        Ca, Cin, ks, _ = wa.shape
        N, _, h, w = x.shape
        L = _L()
        nparts = L.vqw_conv3x3_up2_fwd_pair_supported(Cin, Ca, N, h, w)
        if nparts <= 0 or tuple(wb.shape) != tuple(wa.shape) or ks != 3:
            raise RuntimeError("conv2d_up_pair: shape not served (query vqw_conv3x3_up2_fwd_pair_supported)")

        def _prep():
            wc = torch.empty((2 * Ca, Cin, 3, 3), dtype=torch.float32, device=wa.device, memory_format=CL)
            wc[:Ca].copy_(wa.detach())
            wc[Ca:].copy_(wb.detach())
            buf = _ws(L.vqw_conv3x3_up2_ws_bytes(Cin, 2 * Ca), wa)
            _lib.check(L.vqw_conv3x3_up2_prepare(_p(wc), _p(buf), buf.numel(), Cin, 2 * Ca, _st()), "vqw_conv3x3_up2_prepare")
            bc = None
            if ba is not None and bb is not None:
                bc = torch.cat([ba.detach().reshape(-1), bb.detach().reshape(-1)])
            return buf, bc
        deps = (wb,) + tuple(t for t in (ba, bb) if t is not None)
        up_ws, bias_cat = _cached(wa, "up2pair", _prep, deps=deps)
        ya = empty_nhwc(N, Ca, 2 * h, 2 * w, x)
        yb = empty_nhwc(N, Ca, 2 * h, 2 * w, x)
        pa = torch.empty(N * nparts * Ca * 2, dtype=torch.float32, device=x.device)
        pb = torch.empty(N * nparts * Ca * 2, dtype=torch.float32, device=x.device)
        _lib.check(L.vqw_conv3x3_up2_fwd_pair(_p(x), _p(up_ws), _p(bias_cat), _p(ya), _p(yb), _p(pa), _p(pb), N, h, w, Cin, Ca, _st()),
                   "vqw_conv3x3_up2_fwd_pair")
        up_pair_calls += 1
        ctx.save_for_backward(x, nhwc(wa), nhwc(wb))
        ctx.group = group
        ctx.cfg = (N, 2 * h, 2 * w, Ca, Cin)
        ctx.params = ((wa, ba), (wb, bb))
        ctx.defer = []
        for i, (wgt, bias) in enumerate(ctx.params):
            d = (WGRAD_ASYNC and ctx.needs_input_grad[1 + 2 * i] and wgt.is_leaf and nhwc(wgt) is wgt
                 and not wgrad_through_autograd(wgt, bias) and (bias is None or (bias.is_leaf and bias.is_contiguous())))
            ctx.defer.append(d)
            if d:
                wgt._vqw_pending = getattr(wgt, "_vqw_pending", 0) + 1
        ctx.mark_non_differentiable(pa, pb)
        ctx.set_materialize_grads(False)
        return ya, pa, yb, pb

    @staticmethod
    def backward(ctx, ga, _gpa, gb, _gpb):
        x, wa_n, wb_n = ctx.saved_tensors
        N, H, W, Cout, Cin = ctx.cfg
        L = _L()
        out = [None] * 6
        gx_total = None
        # (the second layer's input gradient runs first, like the second of two separate nodes would)
        for i in (1, 0):
            gy = (ga, gb)[i]
            if gy is None:
                if ctx.group is not None:
                    raise RuntimeError("conv2d_up_pair: both outputs must take part in the backward pass of a gradient group")
                continue
            wgt, bias = ctx.params[i]
            w_n = (wa_n, wb_n)[i]

            def _collapse(w_n=w_n):
                buf = _ws(L.vqw_conv3x3_up2_ws_bytes(Cin, Cout), w_n)
                _lib.check(L.vqw_conv3x3_up2_prepare(_p(w_n), _p(buf), buf.numel(), Cin, Cout, _st()), "vqw_conv3x3_up2_prepare")
                return buf
            up_ws = _cached(wgt, "up2", _collapse)
            need0 = ctx.needs_input_grad[0]
            needw, needb = ctx.needs_input_grad[1 + 2 * i], (bias is not None and ctx.needs_input_grad[2 + 2 * i])
            defer = ctx.defer[i] and (needw or needb)
            g0, _, gw, gbias, gy_n = conv2d_backward_impl(gy, x, None, w_n, None, 1, True, bias is not None, up_ws, need0, False,
                                                          needw and not defer, needb and not defer, group=ctx.group)
            if defer:
                _deferred_wgrad(wgt, bias if (bias is not None and bias.requires_grad) else None, x, None, gy_n, True, 3, 1,
                                N, H, W, Cout, collapsed=True)
            out[1 + 2 * i], out[2 + 2 * i] = gw, gbias
            if g0 is not None:
                gx_total = g0 if gx_total is None else gx_total.add_(g0)
        out[0] = gx_total
        return tuple(out)


def conv2d_up_pair_supported(x, weight_a, weight_b):
    """True when conv2d_up_pair serves these two layers (both 3x3, 32 couts, same shape, the nine-product kernel's geometry)."""
    if not (UP_PAIR and x.is_cuda and weight_a.shape == weight_b.shape and weight_a.shape[2] == 3 and not _in_custom_op):
        return False
    N, _, h, w = x.shape
    return _L().vqw_conv3x3_up2_fwd_pair_supported(weight_a.shape[1], weight_a.shape[0], N, h, w) > 0


def conv2d_up_pair(x, weight_a, bias_a, weight_b, bias_b, grad_group=None):
    """-> ((y_a, part_a), (y_b, part_b)): conv2d(x, w, b, up2x=True, want_stats=True) of two layers of one input, one launch."""
    ya, pa, yb, pb = _ConvUpPair.apply(x, weight_a, bias_a, weight_b, bias_b, grad_group if GRAD_GROUPS else None)
    return (ya, pa), (yb, pb)


# ----------------------------------------------------------------------------------------------
# two plain 3x3 layers of one input as ONE launch on concatenated weights (StyledResUpBlock: the mlp_shared convolutions of its two
# StyledDenorms read the same style tensor, blocks.py:72-75 / 100-134)
# ----------------------------------------------------------------------------------------------
CONV_PAIR = os.environ.get("VQW_CONV_PAIR", "1") != "0"      # 0: the two layers run one by one (A/B timing)
conv_pair_calls = 0


class _ConvPair(torch.autograd.Function):
    """(y_a, y_b) = conv2d(x, w_a, b_a, relu=relu), conv2d(x, w_b, b_b, relu=relu) in Winograd form, one launch of the 64-cout
    kernel with a two-tensor epilogue (vqw_conv3x3_wino_fwd_split): the input is read and transformed once, and two 32-cout
    layers leave the 128-tile x 32-cout workgroup shape for the 64 x 64 one.  The backward is the two layers' own."""

    @staticmethod
    def forward(ctx, x, wa, ba, wb, bb, relu, group):
        global conv_pair_calls
        _dev(x, wa, ba, wb, bb)
        x = nhwc(x)
        Ca, Cin, ks, _ = wa.shape
        N, _, H, W = x.shape
        L = _L()
        if tuple(wb.shape) != tuple(wa.shape) or ks != 3 or (ba is None) != (bb is None) \
                or not L.vqw_conv3x3_wino_split_supported(Cin, 2 * Ca, Ca, 0, N, H, W):
            raise RuntimeError("conv2d_pair: shape not served (query vqw_conv3x3_wino_split_supported)")

        def _prep():
            wc = torch.empty((2 * Ca, Cin, 3, 3), dtype=torch.float32, device=wa.device, memory_format=CL)
            wc[:Ca].copy_(wa.detach())
            wc[Ca:].copy_(wb.detach())
            u = _wino_weights(L, wc, Cin, 2 * Ca)
            bc = torch.cat([ba.detach().reshape(-1), bb.detach().reshape(-1)]) if ba is not None else None
            return u, bc
        deps = (wb,) + tuple(t for t in (ba, bb) if t is not None)
        u, bias_cat = _cached(wa, "pair_wino", _prep, deps=deps)
        ya = empty_nhwc(N, Ca, H, W, x)
        yb = empty_nhwc(N, Ca, H, W, x)
        _lib.check(L.vqw_conv3x3_wino_fwd_split(_p(x), _p(u), _p(bias_cat), _p(ya), _p(yb), N, H, W, Cin, 2 * Ca, Ca, 0, int(relu), _st()),
                   "vqw_conv3x3_wino_fwd_split")
        conv_pair_calls += 1
        ctx.save_for_backward(x, nhwc(wa), nhwc(wb), ya if relu else None, yb if relu else None)
        ctx.group = group
        ctx.cfg = (N, H, W, Ca, Cin)
        ctx.params = ((wa, ba), (wb, bb))
        ctx.defer = []
        for i, (wgt, bias) in enumerate(ctx.params):
            d = (WGRAD_ASYNC and ctx.needs_input_grad[1 + 2 * i] and wgt.is_leaf and nhwc(wgt) is wgt
                 and not wgrad_through_autograd(wgt, bias) and (bias is None or (bias.is_leaf and bias.is_contiguous())))
            ctx.defer.append(d)
            if d:
                wgt._vqw_pending = getattr(wgt, "_vqw_pending", 0) + 1
        ctx.set_materialize_grads(False)
        return ya, yb

    @staticmethod
    def backward(ctx, ga, gb):
        x, wa_n, wb_n, ya, yb = ctx.saved_tensors
        N, H, W, Cout, Cin = ctx.cfg
        out = [None] * 7
        gx_total = None
        # (the second layer's input gradient runs first, like the second of two separate nodes would)
        for i in (1, 0):
            gy = (ga, gb)[i]
            if gy is None:
                if ctx.group is not None:
                    raise RuntimeError("conv2d_pair: both outputs must take part in the backward pass of a gradient group")
                continue
            wgt, bias = ctx.params[i]
            w_n = (wa_n, wb_n)[i]
            need0 = ctx.needs_input_grad[0]
            needw, needb = ctx.needs_input_grad[1 + 2 * i], (bias is not None and ctx.needs_input_grad[2 + 2 * i])
            defer = ctx.defer[i] and (needw or needb)
            g0, _, gw, gbias, gy_n = conv2d_backward_impl(gy, x, None, w_n, (ya, yb)[i], 1, False, bias is not None, None, need0, False,
                                                          needw and not defer, needb and not defer, group=ctx.group)
            if defer:
                _deferred_wgrad(wgt, bias if (bias is not None and bias.requires_grad) else None, x, None, gy_n, False, 3, 1,
                                N, H, W, Cout)
            out[1 + 2 * i], out[2 + 2 * i] = gw, gbias
            if g0 is not None:
                gx_total = g0 if gx_total is None else gx_total.add_(g0)
        out[0] = gx_total
        return tuple(out)


def conv2d_pair_supported(x, weight_a, weight_b):
    """True when conv2d_pair serves these two layers here: both 3x3 of one shape, the Winograd forward admitted at this point of the
    graph (ops.winograd_forward() / evaluation), the 64-cout kernel's geometry."""
    if not (CONV_PAIR and x.is_cuda and weight_a.shape == weight_b.shape and weight_a.shape[2] == 3 and not _in_custom_op
            and _decide_wino_fwd()):
        return False
    N, _, H, W = x.shape
    Ca, Cin = weight_a.shape[0], weight_a.shape[1]
    L = _L()
    return bool(L.vqw_conv3x3_wino_supported(Cin, 2 * Ca, N, H, W) and L.vqw_conv3x3_wino_split_supported(Cin, 2 * Ca, Ca, 0, N, H, W))


def conv2d_pair(x, weight_a, bias_a, weight_b, bias_b, relu=False, grad_group=None):
    """-> (y_a, y_b): conv2d(x, w, b, relu=relu) of two 3x3 layers of one input, one launch."""
    return _ConvPair.apply(x, weight_a, bias_a, weight_b, bias_b, bool(relu), grad_group if GRAD_GROUPS else None)


# Gradients that arrive already multiplied by a fused ReLU's mask: payload = data_ptr of the ReLU output the gradient was masked
# with.  Written by a consumer whose input-gradient kernel applies the mask in its epilogue (vqw_conv3x3_wino_fwd_masked),
# taken by the producer's backward, which then skips its own mask pass.
_MASKED_GRADS = _GradNotes()
FUSE_RELU_MASK = os.environ.get("VQW_FUSE_RELU_MASK", "1") != "0"
# Norm-backward sums that a consumer convolution's input-gradient launch has left per region: payload = (partials, regions per
# image, data_ptr of the norm's raw input) - written by conv2d_backward_impl (vqw_conv3x3_wino_fwd_inbwd), taken by
# _InstanceNorm.backward, which then skips its reduction pass over the activation and the gradient.
_IN_BWD_PARTS = _GradNotes()
FUSE_IN_BWD = os.environ.get("VQW_FUSE_IN_BWD", "1") != "0"


# BatchNorm's num_batches_tracked counters (read by nobody on this path: momentum is a number, not None) are bumped by ONE
# multi-tensor launch at the end of a trainer's step instead of one launch per normalisation call; outside a trainer step
# (begin_step() ... join_streams()) every call bumps its counter at once.
_defer_counters = False
_pending_counters = []


def _bump_counter(t):
    if _defer_counters:
        _pending_counters.append(t)
    else:
        t.add_(1)


def flush_counters():
    global _pending_counters
    if _pending_counters:
        pend, _pending_counters = _pending_counters, []
        uniq, times = [], {}
        for t in pend:
            k = id(t)
            if k not in times:
                uniq.append(t)
            times[k] = times.get(k, 0) + 1
        by = {}
        for t in uniq:
            by.setdefault(times[id(t)], []).append(t)
        for n, ts in by.items():
            torch._foreach_add_(ts, n)


def begin_step():
    """Call before the forwards of a training step: drops fusion notes and a lane join that a failed backward pass left."""
    global _defer_counters
    flush_counters()               # (whatever an aborted step left)
    _defer_counters = True         # until join_streams()
    _MASKED_GRADS.clear()
    _IN_BWD_PARTS.clear()
    if _join_queued_for is not None:
        _join_side_stream()
    elif _fold_keep or _lib.load().vqw_fold_pending():      # a pass that never reached its lane join: its slabs are still alive
        sync_wgrad_lanes()
        _fold_flush()


in_bwd_fused_calls = 0         # InstanceNorm backward calls that took their sums from a convolution's epilogue (tests)
masked_dgrad_calls = 0         # input-gradient launches that applied a ReLU mask in their epilogue (tests)
group_acc_calls = 0            # Winograd input-gradient launches that added to a gradient group's buffer in their epilogue (tests)
split_dgrad_calls = 0          # two-source input-gradient launches whose epilogue wrote both sources' gradients (tests)
SPLIT_DGRAD = os.environ.get("VQW_SPLIT_DGRAD", "1") != "0"      # 0: concatenated gradient + two gather passes (A/B)


def conv2d_backward_impl(gy, x0, x1, w, y_relu, dilation, up0, has_bias, up_ws, need0, need1, needw, needb, group=None, in_src=None):
    """Input / weight / bias gradients of conv2d on the current stream -> (g0, g1, gw, gb, masked gy).  x0 / x1 / w are the
    NHWC tensors the forward saw, y_relu its output when the ReLU was fused (the incoming gradient is masked first),
    up_ws the collapsed-weight buffer when the forward took the low-resolution form."""
    global group_acc_calls
    L = _L()
    Cout, Cin, ks, _ = w.shape
    N = x0.shape[0]
    H, W = (x0.shape[2] * 2, x0.shape[3] * 2) if up0 else (x0.shape[2], x0.shape[3])
    masked_with = _MASKED_GRADS.take(gy) if y_relu is not None else None
    gy = nhwc(gy)
    if y_relu is not None:   # fused ReLU epilogue: mask the incoming gradient first
        if masked_with is not None and masked_with == y_relu.data_ptr():
            pass             # the consumer's input-gradient kernel has applied this very mask in its epilogue (_ConvCat.backward)
        else:
            gm = torch.empty_like(y_relu, memory_format=CL)
            _lib.check(L.vqw_relu_bwd(_p(y_relu), _p(gy), _p(gm), gy.numel(), _st()), "vqw_relu_bwd")
            gy = gm
    C0 = x0.shape[1]
    C1 = 0 if x1 is None else x1.shape[1]
    g0 = g1 = gw = gb = None
    if need0 and up_ws is not None:
        if group is not None and group.buf is not None and L.vqw_conv3x3_up2_dgrad_acc_supported(Cin, Cout, N, H // 2, W // 2):
            # the other up-sampled convolution of this input has run: add to its gradient in this kernel's epilogue
            _lib.check(L.vqw_conv3x3_up2_dgrad_acc(_p(gy), _p(up_ws), _p(group.buf), N, H // 2, W // 2, Cin, Cout, _st()),
                       "vqw_conv3x3_up2_dgrad_acc")
            group_acc_calls += 1
            g0 = group.member_done(None)
        else:
            g0 = torch.empty_like(x0, memory_format=CL)
            _lib.check(L.vqw_conv3x3_up2_dgrad(_p(gy), _p(up_ws), _p(g0), N, H // 2, W // 2, Cin, Cout, _st()),
                       "vqw_conv3x3_up2_dgrad")
            if group is not None:
                g0 = group.member_done(g0)
    elif need0 or (need1 and x1 is not None):
        def _pack():
            buf = torch.empty(Cin * ks * ks * Cout, dtype=torch.float32, device=gy.device)
            _lib.check(L.vqw_pack_dgrad_weights(_p(w), _p(buf), Cout, Cin, ks, _st()), "vqw_pack_dgrad_weights")
            return buf
        wt_of = lambda: _cached(w, "dgrad", _pack)          # noqa: E731  (only the non-Winograd routes read the packed weights)
        if SPLIT_DGRAD and x1 is not None and need0 and need1 and group is None and ks == 3 and dilation == 1 \
                and L.vqw_conv3x3_wino_supported(Cout, Cin, N, H, W) \
                and L.vqw_conv3x3_wino_split_supported(Cout, Cin, C0, int(up0), N, H, W):
            # two sources [up2x(x0) | x1]: both gradients leave the input-gradient kernel's epilogue (x0's summed over each 2 x 2
            # tile when x0 was up-sampled: a Winograd tile IS one low-resolution pixel) instead of two gather passes over the
            # concatenated gradient
            global split_dgrad_calls
            ut = _cached(w, "wino_dgrad", lambda: _wino_weights_dgrad(L, w, Cout, Cin))
            g0 = torch.empty_like(x0, memory_format=CL)
            g1 = torch.empty_like(x1, memory_format=CL)
            _lib.check(L.vqw_conv3x3_wino_fwd_split(_p(gy), _p(ut), None, _p(g0), _p(g1), N, H, W, Cout, Cin, C0, int(up0), 0, _st()),
                       "vqw_conv3x3_wino_fwd_split(dgrad)")
            split_dgrad_calls += 1
            if needw or (needb and has_bias):
                gw = torch.empty((Cout, Cin, ks, ks), dtype=torch.float32, device=gy.device, memory_format=CL)
                gb = torch.empty(Cout, dtype=torch.float32, device=gy.device) if has_bias else None
                _run_wgrad(L, x0, x1, gy, gw, gb, up0, ks, dilation, N, H, W, Cout, False, up_ws is not None)
            return g0, g1, gw, gb, gy
        if SPLIT_DGRAD and x1 is not None and need0 and need1 and group is None and ks == 3 and dilation == 1 and Cin % 64 != 0 \
                and C0 % 16 == 0 and C1 % 16 == 0:
            # ... and for a channel total that is not a multiple of the kernel's cout tile (48 at the encoder's full-resolution
            # level): the layer is widened to 64 input channels with zero weights, the padding couts of the launch are not stored
            Cp = (Cin + 63) // 64 * 64
            if L.vqw_conv3x3_wino_supported(Cout, Cp, N, H, W) and L.vqw_conv3x3_wino_split_padded_supported(Cout, Cp, C0, C1, int(up0), N, H, W):
                def _padded():
                    wp = torch.zeros((Cout, Cp, 3, 3), dtype=torch.float32, device=w.device).contiguous(memory_format=CL)
                    wp[:, :Cin].copy_(w.detach())
                    return _wino_weights_dgrad(L, wp, Cout, Cp)
                ut = _cached(w, "wino_dgrad_pad", _padded)
                g0 = torch.empty_like(x0, memory_format=CL)
                g1 = torch.empty_like(x1, memory_format=CL)
                _lib.check(L.vqw_conv3x3_wino_fwd_split_padded(_p(gy), _p(ut), None, _p(g0), _p(g1), N, H, W, Cout, Cp, C0, C1, int(up0), 0,
                                                               _st()), "vqw_conv3x3_wino_fwd_split_padded(dgrad)")
                split_dgrad_calls += 1
                if needw or (needb and has_bias):
                    gw = torch.empty((Cout, Cin, ks, ks), dtype=torch.float32, device=gy.device, memory_format=CL)
                    gb = torch.empty(Cout, dtype=torch.float32, device=gy.device) if has_bias else None
                    _run_wgrad(L, x0, x1, gy, gw, gb, up0, ks, dilation, N, H, W, Cout, False, up_ws is not None)
                return g0, g1, gw, gb, gy
        if group is not None and need0 and group.buf is not None:
            # a later member of a gradient group (single full-resolution source): add into the shared buffer
            if ks == 3 and dilation == 1 and L.vqw_conv3x3_wino_supported(Cout, Cin, N, H, W) \
                    and L.vqw_conv3x3_wino_masked_supported(Cout, Cin, N, H, W):
                # Winograd form, the shared buffer read and added in the kernel's epilogue
                ut = _cached(w, "wino_dgrad", lambda: _wino_weights_dgrad(L, w, Cout, Cin))
                _lib.check(L.vqw_conv3x3_wino_fwd_acc(_p(gy), _p(ut), _p(group.buf), N, H, W, Cout, Cin, _st()),
                           "vqw_conv3x3_wino_fwd_acc(dgrad)")
                group_acc_calls += 1
                g_full = None
            elif ks == 3 and dilation == 2 and L.vqw_conv3x3_wino_dil2_supported(Cout, Cin, N, H, W):
                # dilation 2: the Winograd kernel on the phase images, adding to the shared buffer in its epilogue
                ut = _cached(w, "wino_dgrad", lambda: _wino_weights_dgrad(L, w, Cout, Cin))
                _lib.check(L.vqw_conv3x3_wino_dil2_fwd(_p(gy), _p(ut), None, _p(group.buf), None, 1, N, H, W, Cout, Cin, 0, _st()),
                           "vqw_conv3x3_wino_dil2_fwd(dgrad, acc)")
                g_full = None
            elif L.vqw_conv2d_fwd_acc_supported(Cout, N, H, W, Cin, ks, dilation):
                # row-chain kernel (dilated 3x3) or the implicit-GEMM kernel (1x1): y += conv in the epilogue
                _lib.check(L.vqw_conv2d_fwd_acc(_p(gy), Cout, _p(wt_of()), _p(group.buf), N, H, W, Cin, ks, dilation, _st()),
                           "vqw_conv2d_fwd_acc")
                if ks == 1:
                    group_acc_calls += 1
                g_full = None
            else:
                g_full = empty_nhwc(N, Cin, H, W, gy)
        else:
            g_full = empty_nhwc(N, Cin, H, W, gy)
        if g_full is None:
            pass
        elif in_src is not None and need0 and group is None and FUSE_IN_BWD \
                and L.vqw_conv3x3_wino_fwd_inbwd_parts(Cout, Cin, N, H, W) > 0:
            # x0 is the output of an InstanceNorm (+ReLU) and feeds this layer only: the norm's backward sums ride in this launch
            global in_bwd_fused_calls
            nparts = L.vqw_conv3x3_wino_fwd_inbwd_parts(Cout, Cin, N, H, W)
            ut = _cached(w, "wino_dgrad", lambda: _wino_weights_dgrad(L, w, Cout, Cin))
            bpart = torch.empty(N * nparts * Cin * 2, dtype=torch.float32, device=gy.device)
            xraw, mr, nrelu = in_src
            _lib.check(L.vqw_conv3x3_wino_fwd_inbwd(_p(gy), _p(ut), _p(xraw), _p(mr), int(nrelu), _p(g_full), _p(bpart),
                                                    N, H, W, Cout, Cin, _st()), "vqw_conv3x3_wino_fwd_inbwd(dgrad)")
            _IN_BWD_PARTS.put(g_full, (bpart, nparts, xraw.data_ptr()))
        elif ks == 3 and dilation == 1 and L.vqw_conv3x3_wino_supported(Cout, Cin, N, H, W):
            ut = _cached(w, "wino_dgrad", lambda: _wino_weights_dgrad(L, w, Cout, Cin))
            _lib.check(L.vqw_conv3x3_wino_fwd(_p(gy), _p(ut), None, _p(g_full), N, H, W, Cout, Cin, 0, _st()),
                       "vqw_conv3x3_wino_fwd(dgrad)")
        elif ks == 3 and dilation == 2 and x1 is None and not up0 and L.vqw_conv3x3_wino_dil2_supported(Cout, Cin, N, H, W):
            ut = _cached(w, "wino_dgrad", lambda: _wino_weights_dgrad(L, w, Cout, Cin))
            _lib.check(L.vqw_conv3x3_wino_dil2_fwd(_p(gy), _p(ut), None, _p(g_full), None, 0, N, H, W, Cout, Cin, 0, _st()),
                       "vqw_conv3x3_wino_dil2_fwd(dgrad)")
        else:
            _lib.check(L.vqw_conv2d_fwd(_p(gy), Cout, 0, None, 0, _p(wt_of()), None, _p(g_full), N, H, W, Cin, ks, dilation, 0, _st()),
                       "vqw_conv2d_fwd(dgrad)")
        if group is not None and need0:
            g0 = group.member_done(g_full)
        elif not up0 and x1 is None:
            g0 = g_full
        else:
            if need0:
                g0 = torch.empty_like(x0, memory_format=CL)
                _lib.check(L.vqw_input_grad_gather(_p(g_full), Cin, 0, C0, int(up0), _p(g0), 0, N, H, W, _st()),
                           "vqw_input_grad_gather")
            if need1 and x1 is not None:
                g1 = torch.empty_like(x1, memory_format=CL)
                _lib.check(L.vqw_input_grad_gather(_p(g_full), Cin, C0, C1, 0, _p(g1), 0, N, H, W, _st()),
                           "vqw_input_grad_gather")
    if needw or (needb and has_bias):
        gw = torch.empty((Cout, Cin, ks, ks), dtype=torch.float32, device=gy.device, memory_format=CL)
        gb = torch.empty(Cout, dtype=torch.float32, device=gy.device) if has_bias else None
        _run_wgrad(L, x0, x1, gy, gw, gb, up0, ks, dilation, N, H, W, Cout, False, up_ws is not None)
    return g0, g1, gw, gb, gy


class GradGroup:
    """Several convolutions of ONE input tensor whose input gradients are summed in place instead of by autograd
    (aspp.py:44-47: the pyramid's five branches).  Every member's backward adds its input gradient to one shared buffer
    and returns None for the input, except the last one to run, which returns the buffer: autograd sees a single gradient
    and launches no add kernels (three passes over the tensor each).  All members must take part in every backward pass
    over the graph (their outputs are all used) and run on one stream.  The group re-arms itself when its last member has
    run, so a second pass over a retained graph works; a pass in which only SOME members ran (torch.autograd.grad over part
    of the branch outputs) cannot hand the input its gradient and raises at the end of that pass instead of dropping it."""

    def __init__(self, members):
        self.members = self.remaining = int(members)
        self.buf = None
        self._watching = False

    def opt_out(self):
        """A member whose input gradient goes through autograd after all (no accumulating route for its form)."""
        self.members -= 1
        self.remaining -= 1

    def member_done(self, g_full):
        """Called by a member's backward with its input gradient (None: already added to the buffer by the kernel).
        Returns what the member hands to autograd: the summed buffer from the last member to run, None from the others."""
        if not self._watching:             # first member of this pass: check completeness when the pass ends
            self._watching = True
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_pass)
        if self.buf is None:
            self.buf = g_full                     # first member to run: its gradient is the buffer
        elif g_full is not None:
            self.buf.add_(g_full)                 # no accumulating kernel for this member's shape
        self.remaining -= 1
        if self.remaining < 0:
            raise RuntimeError("GradGroup: more backward calls than members")
        if self.remaining > 0:
            return None
        out, self.buf, self.remaining = self.buf, None, self.members          # complete: re-armed for another pass
        return out

    def _end_of_pass(self):
        self._watching = False
        if self.buf is not None or self.remaining != self.members:
            ran = self.members - self.remaining
            self.buf, self.remaining = None, self.members
            # the engine skips every callback queued behind one that raises: do the lane join and drop the fusion notes here
            _join_side_stream()
            _MASKED_GRADS.clear()
            _IN_BWD_PARTS.clear()
            raise RuntimeError("GradGroup: only %d of %d members took part in this backward pass - their shared input "
                               "gradient was not delivered (partial backward over grouped branches: set VQW_GRAD_GROUPS=0)"
                               % (ran, self.members))


GRAD_GROUPS = os.environ.get("VQW_GRAD_GROUPS", "1") != "0"      # 0: autograd sums the branch gradients (A/B timing)


def conv2d(x, weight, bias=None, dilation=1, up2x=False, skip=None, relu=False, want_stats=False, grad_group=None, norm_input=False):
    """'same' conv (k in {1,3}, stride 1) of the virtual input [up2x(x) | skip] (channel concat);
    relu=True fuses nn.ReLU into the epilogue.  want_stats=True returns (y, part): `part` (or None when the shape is not
    served) holds the statistics of y for the InstanceNorm that follows: instance_norm(y, ..., part=part)."""
    wino = _decide_wino_fwd()
    group = grad_group if (GRAD_GROUPS and skip is None and not up2x) else None
    if grad_group is not None and group is None and GRAD_GROUPS and up2x and skip is None and weight.shape[2] == 3 and dilation == 1:
        # an up-sampled single source in its collapsed form (the branch _Conv2d.forward takes for these shapes)
        Cout_, Cin_ = weight.shape[0], weight.shape[1]
        if _L().vqw_conv3x3_up2_supported(Cin_, Cout_, x.shape[0], x.shape[2], x.shape[3]):
            group = grad_group
    if grad_group is not None and group is None:
        grad_group.opt_out()               # this member's gradient goes through autograd: the others must not wait for it
    # norm_input=True: x is the output of instance_norm(...) and feeds this layer ONLY - its input-gradient launch then also
    # leaves that norm's backward sums (a gradient that autograd had to sum with another consumer's would miss them)
    in_src = getattr(x, "_vqw_in_src", None) if (norm_input and FUSE_IN_BWD) else None
    if want_stats and CONV_STATS:
        return _Conv2d.apply(x, skip, weight, bias, int(dilation), bool(up2x), bool(relu), True, wino, group, in_src)
    y = _Conv2d.apply(x, skip, weight, bias, int(dilation), bool(up2x), bool(relu), False, wino, group, in_src)
    return (y, None) if want_stats else y


# ----------------------------------------------------------------------------------------------
# two 3x3 convs of the same input as ONE conv with concatenated output channels (StyledDenorm's mlp_gamma | mlp_beta):
# one pass over the input and twice the N-width per tile in forward, one K=2C dgrad whose accumulator sums the two
# input gradients (no add pass), one wgrad.  The parameters stay the two reference tensors; their .grad are the two
# halves of one buffer.
# ----------------------------------------------------------------------------------------------
def _cat_weights(wa, ba, wb, bb):
    def _build():
        Ca, Cin, ks, _ = wa.shape
        Cb = wb.shape[0]
        w = torch.empty((Ca + Cb, Cin, ks, ks), dtype=torch.float32, device=wa.device, memory_format=CL)
        w[:Ca].copy_(wa.detach())
        w[Ca:].copy_(wb.detach())
        b = torch.cat([ba.detach().reshape(-1), bb.detach().reshape(-1)])
        return w, b
    return _cached(wa, "cat", _build, deps=(wb, ba, bb))


def _grad_halves_adjacent(pa, pb, ga, gb_):
    """True when pa.grad / pb.grad are the two halves of the buffer this module allocated."""
    buf = pa.__dict__.get("_vqw_gcat")
    return (buf is not None and ga is not None and gb_ is not None and ga.data_ptr() == buf.data_ptr()
            and gb_.data_ptr() == buf.data_ptr() + ga.numel() * 4 and ga.numel() + gb_.numel() == buf.numel())


def _deferred_wgrad_cat(wa, ba, wb, bb, x0, gy, ks, N, H, W):
    L = _L()
    main = torch.cuda.current_stream()
    side = wgrad_stream(gy.device, _wgrad_lane(wa))
    ev = main.record_event()
    Ca, Cin = wa.shape[0], wa.shape[1]
    Ct = Ca + wb.shape[0]
    x0.record_stream(side)
    gy.record_stream(side)
    with torch.cuda.stream(side):
        side.wait_event(ev)
        fresh = wa.grad is None and wb.grad is None and ba.grad is None and bb.grad is None
        if fresh:
            gw = torch.empty((Ct, Cin, ks, ks), dtype=torch.float32, device=gy.device, memory_format=CL)
            gb_ = torch.empty(Ct, dtype=torch.float32, device=gy.device)
            wa.grad, wb.grad, ba.grad, bb.grad = gw[:Ca], gw[Ca:], gb_[:Ca], gb_[Ca:]
            wa._vqw_gcat, ba._vqw_gcat = gw, gb_
            _run_wgrad(L, x0, None, gy, gw, gb_, False, ks, 1, N, H, W, Ct, False, False, defer_fold=True)
        elif _grad_halves_adjacent(wa, wb, wa.grad, wb.grad) and _grad_halves_adjacent(ba, bb, ba.grad, bb.grad):
            _run_wgrad(L, x0, None, gy, wa._vqw_gcat, ba._vqw_gcat, False, ks, 1, N, H, W, Ct, True, False, defer_fold=True)
        else:    # gradients someone else allocated: compute once, then accumulate the halves
            gw = torch.empty((Ct, Cin, ks, ks), dtype=torch.float32, device=gy.device, memory_format=CL)
            gb_ = torch.empty(Ct, dtype=torch.float32, device=gy.device)
            _run_wgrad(L, x0, None, gy, gw, gb_, False, ks, 1, N, H, W, Ct, False, False)
            for p, g in ((wa, gw[:Ca]), (wb, gw[Ca:]), (ba, gb_[:Ca]), (bb, gb_[Ca:])):
                if p.grad is None:
                    p.grad = g
                else:
                    p.grad.add_(g)
        wa._vqw_pending = getattr(wa, "_vqw_pending", 1) - 1
        if wa._vqw_pending <= 0:
            wa._vqw_pending = 0
            for fn in grad_ready_listeners:
                for p in (wa, ba, wb, bb):
                    fn(p)
    _queue_lane_join()


class _ConvCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, wa, ba, wb, bb, wino_fwd=False, relu_in=False):
        _dev(x, wa, ba, wb, bb)
        ctx.relu_in = bool(relu_in)       # x is the output of a fused ReLU and has no other consumer
        x = nhwc(x)
        Ca, Cin, ks, _ = wa.shape
        Cb = wb.shape[0]
        N, _, H, W = x.shape
        if x.shape[1] != Cin or tuple(wb.shape[1:]) != (Cin, ks, ks):
            raise RuntimeError("conv2d_cat: shapes %s / %s / %s do not match" % (tuple(x.shape), tuple(wa.shape), tuple(wb.shape)))
        w, b = _cat_weights(wa, ba, wb, bb)
        if wino_fwd and ks == 3 and _L().vqw_conv3x3_wino_supported(Cin, Ca + Cb, N, H, W):
            L = _L()
            u = _cached(wa, "cat_wino", lambda: _wino_weights(L, w, Cin, Ca + Cb), deps=(wb,))
            y = empty_nhwc(N, Ca + Cb, H, W, x)
            _lib.check(L.vqw_conv3x3_wino_fwd(_p(x), _p(u), _p(b), _p(y), N, H, W, Cin, Ca + Cb, 0, _st()), "vqw_conv3x3_wino_fwd")
        else:
            y = _conv_fwd_raw(x, False, None, w, b, N, H, W, Ca + Cb, ks, 1, False)
        ctx.save_for_backward(x, w)
        ctx.cfg = (ks, N, H, W, Ca, Cb)
        ctx.params = (wa, ba, wb, bb)
        ctx.defer = (WGRAD_ASYNC and all(ctx.needs_input_grad[1:5]) and all(p.is_leaf for p in (wa, ba, wb, bb)) and not wgrad_through_autograd(wa, ba, wb, bb)
                     and nhwc(wa) is wa and nhwc(wb) is wb and ba.is_contiguous() and bb.is_contiguous())
        if ctx.defer:
            wa._vqw_pending = getattr(wa, "_vqw_pending", 0) + 1
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        ks, N, H, W, Ca, Cb = ctx.cfg
        wa, ba, wb, bb = ctx.params
        Ct, Cin = Ca + Cb, x.shape[1]
        L = _L()
        gy = nhwc(gy)
        gx = gwa = gba = gwb = gbb = None
        if ctx.needs_input_grad[0]:
            def _pack():
                buf = torch.empty(Cin * ks * ks * Ct, dtype=torch.float32, device=gy.device)
                _lib.check(L.vqw_pack_dgrad_weights(_p(w), _p(buf), Ct, Cin, ks, _st()), "vqw_pack_dgrad_weights")
                return buf
            gx = empty_nhwc(N, Cin, H, W, gy)
            if ks == 3 and L.vqw_conv3x3_wino_supported(Ct, Cin, N, H, W):
                ut = _cached(wa, "cat_wino_dgrad", lambda: _wino_weights_dgrad(L, w, Ct, Cin), deps=(wb,))
                if ctx.relu_in and FUSE_RELU_MASK and L.vqw_conv3x3_wino_masked_supported(Ct, Cin, N, H, W):
                    # the gradient in FRONT of the producer's ReLU: its mask (x > 0) applied in this kernel's epilogue
                    _lib.check(L.vqw_conv3x3_wino_fwd_masked(_p(gy), _p(ut), _p(x), _p(gx), N, H, W, Ct, Cin, _st()),
                               "vqw_conv3x3_wino_fwd_masked(dgrad)")
                    _MASKED_GRADS.put(gx, x.data_ptr())
                    global masked_dgrad_calls
                    masked_dgrad_calls += 1
                else:
                    _lib.check(L.vqw_conv3x3_wino_fwd(_p(gy), _p(ut), None, _p(gx), N, H, W, Ct, Cin, 0, _st()),
                               "vqw_conv3x3_wino_fwd(dgrad)")
            else:
                wt = _cached(wa, "cat_dgrad", _pack, deps=(wb,))
                _lib.check(L.vqw_conv2d_fwd(_p(gy), Ct, 0, None, 0, _p(wt), None, _p(gx), N, H, W, Cin, ks, 1, 0, _st()),
                           "vqw_conv2d_fwd(dgrad)")
        if ctx.defer:
            _deferred_wgrad_cat(wa, ba, wb, bb, x, gy, ks, N, H, W)
        elif any(ctx.needs_input_grad[1:5]):
            gw = torch.empty((Ct, Cin, ks, ks), dtype=torch.float32, device=gy.device, memory_format=CL)
            gb_ = torch.empty(Ct, dtype=torch.float32, device=gy.device)
            _run_wgrad(L, x, None, gy, gw, gb_, False, ks, 1, N, H, W, Ct, False, False)
            gwa, gwb, gba, gbb = gw[:Ca], gw[Ca:], gb_[:Ca], gb_[Ca:]
        return gx, gwa, gba, gwb, gbb, None, None


def conv2d_cat(x, weight_a, bias_a, weight_b, bias_b, relu_input=False):
    """[conv(x, weight_a, bias_a) | conv(x, weight_b, bias_b)] along channels (3x3 / 1x1, stride 1, 'same').
    relu_input=True: x is the output of conv2d(..., relu=True) and feeds nothing else - the input gradient then leaves this
    node already masked by that ReLU (one kernel epilogue instead of a separate pass over the gradient)."""
    return _ConvCat.apply(x, weight_a, bias_a, weight_b, bias_b, _decide_wino_fwd(), bool(relu_input))


# ----------------------------------------------------------------------------------------------
# InstanceNorm (+ReLU)
# ----------------------------------------------------------------------------------------------
class _InstanceNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, relu, eps, part=None):
        _dev(x)
        x = nhwc(x)
        N, C, H, W = x.shape
        L = _L()
        y = torch.empty_like(x, memory_format=CL)
        mr = torch.empty(N * C * 2, dtype=torch.float32, device=x.device)
        if part is not None:        # statistics left by the producing convolution's epilogue
            nparts = part.numel() // (N * C * 2)
            _lib.check(L.vqw_inorm_fwd_parts(_p(x), _p(y), C, 0, _p(mr), _p(part), nparts, N, H * W, C, eps, int(relu), _st()),
                       "vqw_inorm_fwd_parts")
        else:
            ws = _ws(L.vqw_plane_ws_bytes(N, C, H * W), x)
            _lib.check(L.vqw_inorm_fwd(_p(x), _p(y), C, 0, _p(mr), _p(ws), ws.numel(), N, H * W, C, eps, int(relu), _st()), "vqw_inorm_fwd")
        ctx.save_for_backward(x, mr)
        ctx.relu = relu
        ctx.mark_non_differentiable(mr)
        ctx.set_materialize_grads(False)
        return y, mr               # mr = (mean, rstd) per (image, channel): for a consumer that fuses this norm's backward sums

    @staticmethod
    def backward(ctx, gy, _gmr=None):
        if gy is None:
            return None, None, None, None
        x, mr = ctx.saved_tensors
        N, C, H, W = x.shape
        L = _L()
        ent = _IN_BWD_PARTS.take(gy)
        gy = nhwc(gy)
        gx = torch.empty_like(x, memory_format=CL)
        if ent is not None and ent[2] == x.data_ptr():
            # the only consumer's input-gradient launch has left (sum gm, sum gm * xhat) per region: no reduction pass
            global in_bwd_fused_calls
            in_bwd_fused_calls += 1
            bpart, nparts, _ = ent
            means = torch.empty(N * C * 2, dtype=torch.float32, device=x.device)
            _lib.check(L.vqw_inorm_bwd_parts(_p(x), _p(mr), _p(gy), _p(bpart), nparts, _p(means), _p(gx), N, H * W, C,
                                             int(ctx.relu), _st()), "vqw_inorm_bwd_parts")
            return gx, None, None, None
        ws = _ws(L.vqw_plane_ws_bytes(N, C, H * W), x)
        _lib.check(L.vqw_inorm_bwd(_p(x), _p(mr), _p(gy), C, 0, _p(gx), _p(ws), ws.numel(), N, H * W, C, int(ctx.relu), _st()),
                   "vqw_inorm_bwd")
        return gx, None, None, None


def instance_norm(x, relu=False, eps=1e-5, part=None):
    """part: statistics partials from conv2d(..., want_stats=True) of the SAME tensor (skips the reduction pass).
    The result carries `_vqw_in_src` = (raw input, relu): a convolution that is this tensor's ONLY consumer can be told so
    (conv2d(..., norm_input=True)) and then leaves the norm's backward sums in its input-gradient epilogue."""
    y, mr = _InstanceNorm.apply(x, bool(relu), float(eps), part)
    if FUSE_IN_BWD and torch.is_grad_enabled() and y.requires_grad:
        y._vqw_in_src = (nhwc(x), mr, bool(relu))
    return y


class _InstanceNormCat(torch.autograd.Function):
    """InstanceNorm(+ReLU) of several tensors written straight into the channel slices of ONE output
    (the torch.cat of aspp.py:47 is never materialised as a separate pass)."""

    @staticmethod
    def forward(ctx, relu, eps, parts, *xs):
        _dev(*xs)
        xs = [nhwc(x) for x in xs]
        N, _, H, W = xs[0].shape
        Ct = sum(x.shape[1] for x in xs)
        L = _L()
        y = empty_nhwc(N, Ct, H, W, xs[0])
        mrs, off = [], 0
        for x in xs:
            C = x.shape[1]
            if x.shape[0] != N or x.shape[2] != H or x.shape[3] != W:
                raise RuntimeError("instance_norm_cat: shape mismatch")
            mr = torch.empty(N * C * 2, dtype=torch.float32, device=x.device)
            part = parts[len(mrs)] if parts is not None else None
            if part is not None:        # statistics left by the producing convolution's epilogue
                _lib.check(L.vqw_inorm_fwd_parts(_p(x), _p(y), Ct, off, _p(mr), _p(part), part.numel() // (N * C * 2), N, H * W, C,
                                                 eps, int(relu), _st()), "vqw_inorm_fwd_parts")
            else:
                ws = _ws(L.vqw_plane_ws_bytes(N, C, H * W), x)
                _lib.check(L.vqw_inorm_fwd(_p(x), _p(y), Ct, off, _p(mr), _p(ws), ws.numel(), N, H * W, C, eps, int(relu), _st()),
                           "vqw_inorm_fwd")
            mrs.append(mr)
            off += C
        ctx.save_for_backward(*xs, *mrs)
        ctx.relu, ctx.n = relu, len(xs)
        return y

    @staticmethod
    def backward(ctx, gy):
        saved = ctx.saved_tensors
        xs, mrs = saved[:ctx.n], saved[ctx.n:]
        gy = nhwc(gy)
        N, Ct, H, W = gy.shape
        L = _L()
        outs, off = [], 0
        for x, mr in zip(xs, mrs):
            C = x.shape[1]
            gx = torch.empty_like(x, memory_format=CL)
            ws = _ws(L.vqw_plane_ws_bytes(N, C, H * W), x)
            _lib.check(L.vqw_inorm_bwd(_p(x), _p(mr), _p(gy), Ct, off, _p(gx), _p(ws), ws.numel(), N, H * W, C, int(ctx.relu), _st()),
                       "vqw_inorm_bwd")
            outs.append(gx)
            off += C
        return (None, None, None, *outs)


def instance_norm_cat(xs, relu=True, eps=1e-5, parts=None):
    """parts: per input, the statistics partials of conv2d(..., want_stats=True) or None."""
    return _InstanceNormCat.apply(bool(relu), float(eps), tuple(parts) if parts is not None else None, *xs)


# ----------------------------------------------------------------------------------------------
# StyledDenorm core: BatchNorm2d(affine=False)(x) * (1 + gamma) + beta (+ReLU)
# ----------------------------------------------------------------------------------------------
# VQW_DP_FORCE=1 (or force_collectives(True)): issue every data-parallel collective - SyncBN statistics, VQ statistics, the
# gradient buckets of trainers.GradientAllReducer - also in a process group of ONE rank.  A one-GPU box can then run the
# real RCCL code path (ProcessGroupNCCL's stream / event ordering against the two view streams and the weight-gradient
# lanes); with one rank every all-reduce is the identity, so the step must equal the non-distributed step bit for bit
# (tests/test_gpu_dp.py::test_rccl_world_size_one_equals_plain_step).
FORCE_COLLECTIVES = os.environ.get("VQW_DP_FORCE", "0") == "1"
collective_calls = 0           # small collectives issued from this module since import (tools/dp_probe prints it per step)


def force_collectives(on=True):
    global FORCE_COLLECTIVES
    old, FORCE_COLLECTIVES = FORCE_COLLECTIVES, bool(on)
    return old


_FORCE_STATS = os.environ.get("VQW_DP_FORCE_STATS", "1") != "0"      # measurement aid: 0 = a forced one-rank run skips the statistics collectives


def _dist_on():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or (FORCE_COLLECTIVES and _FORCE_STATS))


def _all_reduce(t):
    global collective_calls
    collective_calls += 1
    dist.all_reduce(t)


class _Spade(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, relu, sync, nbt, res=None, part=None,
                res_norm=None):
        """res_norm = (statistics partials of res or None, relu, eps): `res` is the RAW input of an InstanceNorm(+ReLU) whose output is
        the residual; it is normalised inside the modulation kernel (vqw_spade_fwd_res_norm) and never materialised."""
        _dev(x, gamma, beta, res)
        x, gamma = nhwc(x), nhwc(gamma)
        N, C, H, W = x.shape
        # beta=None: `gamma` holds [gamma | beta] as 2C channels (conv2d_cat)
        fused = beta is None
        if fused:
            if gamma.shape[1] != 2 * C:
                raise RuntimeError("spade_norm: fused gamma|beta map needs %d channels, got %d" % (2 * C, gamma.shape[1]))
        else:
            beta = nhwc(beta)
        gbs = 2 * C if fused else C
        gptr = _p(gamma)
        bptr = _p(gamma.narrow(1, C, C) if fused else beta)      # fused: beta = channels C..2C of the same NHWC map
        L = _L()
        mr = torch.empty(2 * C, dtype=torch.float32, device=x.device)
        count = float(N * H * W)
        if training:
            sums = torch.empty(2 * C, dtype=torch.float64, device=x.device)
            fused_finalize = part is not None and not (sync and _dist_on())
            if fused_finalize:      # no collective between the sums and the statistics: one launch for both (below)
                pass
            elif part is not None:    # per-tile sums left by the producing convolution's epilogue
                rows = part.numel() // (2 * C)
                _lib.check(L.vqw_bn_stats_from_parts(_p(part), _p(sums), rows, C, count / rows, _st()), "vqw_bn_stats_from_parts")
            else:
                ws = _ws(L.vqw_plane_ws_bytes(N, C, H * W), x)
                _lib.check(L.vqw_bn_partial_stats(_p(x), _p(sums), _p(ws), ws.numel(), N, H * W, C, _st()), "vqw_bn_partial_stats")
            if sync and _dist_on():
                # SyncBatchNorm semantics (run_vqwnet.py:121): global-batch statistics, one small all-reduce.
                # Every rank holds the same per-rank batch (weak scaling), so the count needs no exchange.
                _all_reduce(sums)
                count *= dist.get_world_size()
            cur = _order_begin(running_mean)
            if fused_finalize:
                rows = part.numel() // (2 * C)
                _lib.check(L.vqw_bn_finalize_parts(_p(part), rows, count / rows, _p(sums), count, _p(mr), _p(running_mean), _p(running_var),
                                                   momentum, eps, C, _st()), "vqw_bn_finalize_parts")
            else:
                _lib.check(L.vqw_bn_finalize(_p(sums), count, _p(mr), _p(running_mean), _p(running_var), momentum, eps, C, _st()),
                           "vqw_bn_finalize")
            if nbt is not None:
                _bump_counter(nbt)
            _order_end(running_mean, cur)
        else:
            _lib.check(L.vqw_bn_eval_stats(_p(running_mean), _p(running_var), _p(mr), eps, C, _st()), "vqw_bn_eval_stats")
        y = torch.empty_like(x, memory_format=CL)
        rmr = None
        if res is not None:         # y = act(...) + res: the block's `shortcut + main` inside this kernel
            res = nhwc(res)
            if res.shape != x.shape:
                raise RuntimeError("spade_norm: residual shape %s does not match %s" % (tuple(res.shape), tuple(x.shape)))
        if res is not None and res_norm is not None:
            rpart, rrelu, reps = res_norm
            rmr = torch.empty(N * C * 2, dtype=torch.float32, device=x.device)
            if rpart is not None:
                _lib.check(L.vqw_inorm_stats_parts(_p(rpart), rpart.numel() // (N * C * 2), _p(rmr), N, H * W, C, reps, _st()),
                           "vqw_inorm_stats_parts")
            else:
                ws = _ws(L.vqw_plane_ws_bytes(N, C, H * W), res)
                _lib.check(L.vqw_inorm_stats(_p(res), _p(rmr), _p(ws), ws.numel(), N, H * W, C, reps, _st()), "vqw_inorm_stats")
            _lib.check(L.vqw_spade_fwd_res_norm(_p(x), _p(mr), gptr, bptr, gbs, _p(res), _p(rmr), int(rrelu), _p(y), N, H * W, C,
                                                int(relu), _st()), "vqw_spade_fwd_res_norm")
            ctx.res_relu = bool(rrelu)
        elif res is not None:
            _lib.check(L.vqw_spade_fwd_res(_p(x), _p(mr), gptr, bptr, gbs, _p(res), _p(y), N * H * W, C, int(relu), _st()),
                       "vqw_spade_fwd_res")
        else:
            _lib.check(L.vqw_spade_fwd(_p(x), _p(mr), gptr, bptr, gbs, _p(y), N * H * W, C, int(relu), _st()), "vqw_spade_fwd")
        ctx.save_for_backward(x, gamma, beta, mr, res if rmr is not None else None, rmr)
        ctx.cfg = (training, relu, count, sync, fused)
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, beta, mr, res_raw, rmr = ctx.saved_tensors
        training, relu, count, sync, fused = ctx.cfg
        N, C, H, W = x.shape
        L = _L()
        gy = nhwc(gy)
        if fused:
            dgamma, dbeta, gbs = torch.empty_like(gamma, memory_format=CL), None, 2 * C
            gptr, dgptr = _p(gamma), _p(dgamma)
            bptr, dbptr = _p(gamma.narrow(1, C, C)), _p(dgamma.narrow(1, C, C))
        else:
            dgamma = torch.empty_like(x, memory_format=CL)
            dbeta = torch.empty_like(x, memory_format=CL)
            gbs, gptr, bptr, dgptr, dbptr = C, _p(gamma), _p(beta), _p(dgamma), _p(dbeta)
        gx = torch.empty_like(x, memory_format=CL)
        sums = torch.empty(2 * C, dtype=torch.float64, device=x.device)
        ws = _ws(L.vqw_plane_ws_bytes(N, C, H * W), x)
        _lib.check(L.vqw_spade_bwd_reduce(_p(x), _p(mr), gptr, bptr, _p(gy), dgptr, dbptr, gbs, _p(sums), _p(ws),
                                          ws.numel(), N, H * W, C, int(relu), _st()), "vqw_spade_bwd_reduce")
        if training and sync and _dist_on():
            _all_reduce(sums)
        _lib.check(L.vqw_spade_bwd_apply(_p(x), _p(mr), gptr, bptr, gbs, _p(gy), _p(sums), count, _p(gx), N * H * W, C,
                                         int(relu), int(training), _st()), "vqw_spade_bwd_apply")
        gres = gy if ctx.has_res else None
        if rmr is not None and ctx.needs_input_grad[11]:
            # the residual was normalised in the forward kernel: its gradient goes back through that InstanceNorm(+ReLU) here
            gres = torch.empty_like(res_raw, memory_format=CL)
            ws2 = _ws(L.vqw_plane_ws_bytes(N, C, H * W), res_raw)
            _lib.check(L.vqw_inorm_bwd(_p(res_raw), _p(rmr), _p(gy), C, 0, _p(gres), _p(ws2), ws2.numel(), N, H * W, C,
                                       int(ctx.res_relu), _st()), "vqw_inorm_bwd(residual)")
        return gx, dgamma, dbeta, None, None, None, None, None, None, None, None, gres, None, None


RES_NORM_FUSED = os.environ.get("VQW_RES_NORM_FUSED", "1") != "0"      # 0: the shortcut's norm writes its tensor first (A/B)


def spade_norm(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, relu=False, sync=True,
               num_batches_tracked=None, residual=None, part=None, residual_norm=None):
    """residual: added AFTER the activation (y = act(spade(x)) + residual); needs C % 4 == 0.
    part: statistics partials of x from conv2d(..., want_stats=True) (training mode skips its reduction pass).
    residual_norm = (partials or None, relu, eps): `residual` is the RAW input of an InstanceNorm(+ReLU) - the block's shortcut
    branch - and is normalised inside the modulation kernel where the shape allows, by a separate pass otherwise."""
    if residual is not None and residual_norm is not None:
        N, C, H, W = x.shape
        if not (RES_NORM_FUSED and x.is_cuda and not (C & 3) and not (gamma.shape[1] & 3) and tuple(residual.shape) == tuple(x.shape)
                and _L().vqw_spade_fwd_res_norm_supported(H * W, C)):
            rpart, rrelu, reps = residual_norm
            residual = instance_norm(residual, relu=rrelu, eps=reps, part=rpart)
            residual_norm = None
    if residual_norm is not None:
        return _Spade.apply(x, gamma, beta, running_mean, running_var, bool(training), float(momentum), float(eps), bool(relu),
                            bool(sync), num_batches_tracked, residual, part if training else None,
                            (residual_norm[0], bool(residual_norm[1]), float(residual_norm[2])))
    if residual is not None and (x.shape[1] & 3 or gamma.shape[1] & 3):
        return add(spade_norm(x, gamma, beta, running_mean, running_var, training, momentum, eps, relu, sync,
                              num_batches_tracked, part=part), residual)
    return _Spade.apply(x, gamma, beta, running_mean, running_var, bool(training), float(momentum), float(eps), bool(relu),
                        bool(sync), num_batches_tracked, residual, part if training else None)


# ----------------------------------------------------------------------------------------------
# element-wise / pooling
# ----------------------------------------------------------------------------------------------
class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, relu, a_group=None):
        _dev(a, b)
        a, b = nhwc(a), nhwc(b)
        if a.shape != b.shape:
            raise RuntimeError("add: shape mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        y = torch.empty_like(a, memory_format=CL)
        _lib.check(_L().vqw_add(_p(a), _p(b), _p(y), a.numel(), int(relu), _st()), "vqw_add")
        ctx.relu = relu
        ctx.a_group = a_group
        if relu:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        if ctx.relu:
            (y,) = ctx.saved_tensors
            gy = nhwc(gy)
            gx = torch.empty_like(y, memory_format=CL)
            _lib.check(_L().vqw_relu_bwd(_p(y), _p(gy), _p(gx), y.numel(), _st()), "vqw_relu_bwd")
            gy = gx
        if ctx.a_group is not None:
            # `a` has further consumers that form a gradient group (their backward runs AFTER everything upstream of `b`, which
            # reads this gradient, has run): this gradient is the group's member - as the first to arrive it becomes the buffer
            # the others add to in their kernels' epilogues; the last member hands the sum to autograd
            return ctx.a_group.member_done(nhwc(gy)), gy, None, None
        return gy, gy, None, None


ADD_NORM_FUSED = os.environ.get("VQW_ADD_NORM_FUSED", "1") != "0"      # 0: the norm writes its tensor, then the add reads it (A/B)


class _AddNorm(torch.autograd.Function):
    """y = a + InstanceNorm(+ReLU)(x) with x the RAW convolution output and `part` its statistics partials (or None): the norm is
    applied inside the add's kernel (vqw_inorm_add_fwd), its output never written.  Backward: the gradient of `a` is gy (handed to
    `a_group` like _Add does), the gradient of x is the norm's backward of gy."""

    @staticmethod
    def forward(ctx, a, x, part, relu, eps, a_group):
        _dev(a, x)
        a, x = nhwc(a), nhwc(x)
        if a.shape != x.shape:
            raise RuntimeError("add_norm: shape mismatch %s vs %s" % (tuple(a.shape), tuple(x.shape)))
        N, C, H, W = x.shape
        L = _L()
        mr = torch.empty(N * C * 2, dtype=torch.float32, device=x.device)
        if part is not None:
            _lib.check(L.vqw_inorm_stats_parts(_p(part), part.numel() // (N * C * 2), _p(mr), N, H * W, C, eps, _st()), "vqw_inorm_stats_parts")
        else:
            ws = _ws(L.vqw_plane_ws_bytes(N, C, H * W), x)
            _lib.check(L.vqw_inorm_stats(_p(x), _p(mr), _p(ws), ws.numel(), N, H * W, C, eps, _st()), "vqw_inorm_stats")
        y = torch.empty_like(x, memory_format=CL)
        _lib.check(L.vqw_inorm_add_fwd(_p(x), _p(mr), _p(a), _p(y), N, H * W, C, int(relu), _st()), "vqw_inorm_add_fwd")
        ctx.save_for_backward(x, mr)
        ctx.relu, ctx.a_group = bool(relu), a_group
        return y

    @staticmethod
    def backward(ctx, gy):
        x, mr = ctx.saved_tensors
        N, C, H, W = x.shape
        L = _L()
        gy = nhwc(gy)
        gx = None
        if ctx.needs_input_grad[1]:
            gx = torch.empty_like(x, memory_format=CL)
            ws = _ws(L.vqw_plane_ws_bytes(N, C, H * W), x)
            _lib.check(L.vqw_inorm_bwd(_p(x), _p(mr), _p(gy), C, 0, _p(gx), _p(ws), ws.numel(), N, H * W, C, int(ctx.relu), _st()),
                       "vqw_inorm_bwd(add_norm)")
        ga = gy
        if ctx.a_group is not None:
            ga = ctx.a_group.member_done(gy)
        return ga, gx, None, None, None, None


def add_norm_supported(a, x):
    return bool(ADD_NORM_FUSED and x.is_cuda and x.dim() == 4 and tuple(a.shape) == tuple(x.shape) and _L().vqw_inorm_add_supported(x.shape[1]))


def add_norm(a, x, part=None, relu=False, eps=1e-5, a_group=None):
    """a + instance_norm(x, relu, eps, part) in one kernel (see _AddNorm); query add_norm_supported first."""
    return _AddNorm.apply(a, x, part, bool(relu), float(eps), a_group)


def add(a, b, relu=False, a_group=None):
    """a + b (+ReLU).  a_group: an ops.GradGroup that the OTHER consumers of `a` belong to, sized for them plus this op; `b`
    must be computed from `a` through those consumers (x + f(x)), so that their backward runs after b's whole chain."""
    return _Add.apply(a, b, bool(relu), a_group)


class _MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _dev(x)
        x = nhwc(x)
        N, C, H, W = x.shape
        y = empty_nhwc(N, C, H // 2, W // 2, x)
        _lib.check(_L().vqw_maxpool2_fwd(_p(x), _p(y), N, H, W, C, _st()), "vqw_maxpool2_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        N, C, H, W = x.shape
        gy = nhwc(gy)
        gx = torch.empty_like(x, memory_format=CL)
        _lib.check(_L().vqw_maxpool2_bwd(_p(x), _p(gy), None, _p(gx), N, H, W, C, _st()), "vqw_maxpool2_bwd")
        return gx


def maxpool2(x):
    return _MaxPool2.apply(x)


class _ResTail(torch.autograd.Function):
    """out = ReLU(a + b); pooled = MaxPool2d(2)(out) as ONE autograd node (blocks.py:29-36).  `out` has two consumers (the
    pooling and the skip connection): as separate nodes their gradients meet in autograd's add, then pass the ReLU mask —
    three kernels over full-resolution tensors; here one (vqw_res_tail_bwd)."""

    @staticmethod
    def forward(ctx, a, b):
        _dev(a, b)
        a, b = nhwc(a), nhwc(b)
        if a.shape != b.shape:
            raise RuntimeError("res_tail: shape mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        N, C, H, W = a.shape
        L = _L()
        out = torch.empty_like(a, memory_format=CL)
        pooled = empty_nhwc(N, C, H // 2, W // 2, a)
        _lib.check(L.vqw_res_tail_fwd(_p(a), _p(b), _p(out), _p(pooled), N, H, W, C, _st()), "vqw_res_tail_fwd")
        ctx.save_for_backward(out)
        ctx.set_materialize_grads(False)       # an unused output's gradient arrives as None (the kernel takes a null pointer), not as a zero fill
        return pooled, out

    @staticmethod
    def backward(ctx, g_pooled, g_out):
        if g_pooled is None and g_out is None:
            return None, None
        (out,) = ctx.saved_tensors
        N, C, H, W = out.shape
        gp = nhwc(g_pooled) if g_pooled is not None else None
        go = nhwc(g_out) if g_out is not None else None
        gx = torch.empty_like(out, memory_format=CL)
        _lib.check(_L().vqw_res_tail_bwd(_p(out), _p(gp), _p(go), _p(gx), N, H, W, C, _st()), "vqw_res_tail_bwd")
        return gx, gx


IN_BWD_PAIR = os.environ.get("VQW_IN_BWD_PAIR", "1") != "0"      # 0: two separate InstanceNorm backward calls (A/B timing)
RES_TAIL_BWD_FUSED = os.environ.get("VQW_RES_TAIL_BWD_FUSED", "1") != "0"      # 0: vqw_res_tail_bwd, then vqw_inorm_bwd_pair (A/B)


class _ResTailNorm(torch.autograd.Function):
    """ResBlock tail on the RAW outputs of its two conv branches (blocks.py:25-36):
    a = ReLU(InstanceNorm(x2)), b = InstanceNorm(xid), out = ReLU(a + b), pooled = MaxPool2d(2)(out).
    The kernel normalises while it reads, so the two apply passes of the norms disappear; x2's statistics come from its
    convolution's epilogue partials when available.  Backward = the tail's single kernel, then the two norms' backward."""

    @staticmethod
    def forward(ctx, x2, xid, eps, part2, partid=None):
        _dev(x2, xid)
        x2, xid = nhwc(x2), nhwc(xid)
        if x2.shape != xid.shape:
            raise RuntimeError("res_tail_norm: shape mismatch %s vs %s" % (tuple(x2.shape), tuple(xid.shape)))
        N, C, H, W = x2.shape
        L = _L()
        mr2 = torch.empty(N * C * 2, dtype=torch.float32, device=x2.device)
        mrid = torch.empty(N * C * 2, dtype=torch.float32, device=x2.device)
        if part2 is not None and partid is not None:      # both norms' statistics from their convolutions' partials: one launch
            _lib.check(L.vqw_inorm_stats_parts2(_p(part2), part2.numel() // (N * C * 2), _p(mr2), _p(partid), partid.numel() // (N * C * 2),
                                                _p(mrid), N, H * W, C, eps, _st()), "vqw_inorm_stats_parts2")
        elif part2 is not None:
            _lib.check(L.vqw_inorm_stats_parts(_p(part2), part2.numel() // (N * C * 2), _p(mr2), N, H * W, C, eps, _st()),
                       "vqw_inorm_stats_parts")
        else:
            ws = _ws(L.vqw_plane_ws_bytes(N, C, H * W), x2)
            _lib.check(L.vqw_inorm_stats(_p(x2), _p(mr2), _p(ws), ws.numel(), N, H * W, C, eps, _st()), "vqw_inorm_stats")
        if part2 is not None and partid is not None:
            pass
        elif partid is not None:
            _lib.check(L.vqw_inorm_stats_parts(_p(partid), partid.numel() // (N * C * 2), _p(mrid), N, H * W, C, eps, _st()),
                       "vqw_inorm_stats_parts")
        else:
            ws = _ws(L.vqw_plane_ws_bytes(N, C, H * W), xid)
            _lib.check(L.vqw_inorm_stats(_p(xid), _p(mrid), _p(ws), ws.numel(), N, H * W, C, eps, _st()), "vqw_inorm_stats")
        out = torch.empty_like(x2, memory_format=CL)
        pooled = empty_nhwc(N, C, H // 2, W // 2, x2)
        _lib.check(L.vqw_res_tail_norm_fwd(_p(x2), _p(mr2), _p(xid), _p(mrid), _p(out), _p(pooled), N, H, W, C, _st()),
                   "vqw_res_tail_norm_fwd")
        ctx.save_for_backward(x2, xid, mr2, mrid, out)
        ctx.set_materialize_grads(False)
        return pooled, out

    @staticmethod
    def backward(ctx, g_pooled, g_out):
        if g_pooled is None and g_out is None:
            return None, None, None, None, None
        x2, xid, mr2, mrid, out = ctx.saved_tensors
        N, C, H, W = out.shape
        L = _L()
        gp = nhwc(g_pooled) if g_pooled is not None else None
        go = nhwc(g_out) if g_out is not None else None
        g = torch.empty_like(out, memory_format=CL)
        if RES_TAIL_BWD_FUSED and IN_BWD_PAIR and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and not (C & 3):
            # the tail's backward and both norms' backward sums in one pass over (out, gradients, x2, xid); g is written for the
            # apply pass only
            gx2 = torch.empty_like(x2, memory_format=CL)
            gxid = torch.empty_like(xid, memory_format=CL)
            ws = _ws(2 * L.vqw_plane_ws_bytes(N, C, H * W), x2)
            _lib.check(L.vqw_res_tail_bwd_pair(_p(out), _p(gp), _p(go), _p(x2), _p(mr2), _p(xid), _p(mrid), _p(g), _p(gx2), _p(gxid),
                                               _p(ws), ws.numel(), N, H, W, C, _st()), "vqw_res_tail_bwd_pair")
            return gx2, gxid, None, None, None
        _lib.check(L.vqw_res_tail_bwd(_p(out), _p(gp), _p(go), _p(g), N, H, W, C, _st()), "vqw_res_tail_bwd")
        gx2 = gxid = None
        if IN_BWD_PAIR and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:      # both norms' backward, common gradient read once
            gx2 = torch.empty_like(x2, memory_format=CL)
            gxid = torch.empty_like(xid, memory_format=CL)
            ws = _ws(2 * L.vqw_plane_ws_bytes(N, C, H * W), x2)
            _lib.check(L.vqw_inorm_bwd_pair(_p(x2), _p(mr2), _p(xid), _p(mrid), _p(g), _p(gx2), _p(gxid), _p(ws), ws.numel(),
                                            N, H * W, C, _st()), "vqw_inorm_bwd_pair")
            return gx2, gxid, None, None, None
        if ctx.needs_input_grad[0]:
            gx2 = torch.empty_like(x2, memory_format=CL)
            ws = _ws(L.vqw_plane_ws_bytes(N, C, H * W), x2)
            _lib.check(L.vqw_inorm_bwd(_p(x2), _p(mr2), _p(g), C, 0, _p(gx2), _p(ws), ws.numel(), N, H * W, C, 1, _st()), "vqw_inorm_bwd")
        if ctx.needs_input_grad[1]:
            gxid = torch.empty_like(xid, memory_format=CL)
            ws = _ws(L.vqw_plane_ws_bytes(N, C, H * W), xid)
            _lib.check(L.vqw_inorm_bwd(_p(xid), _p(mrid), _p(g), C, 0, _p(gxid), _p(ws), ws.numel(), N, H * W, C, 0, _st()), "vqw_inorm_bwd")
        return gx2, gxid, None, None, None


def res_tail_norm(x2, xid, eps=1e-5, part2=None, partid=None):
    """(pooled, out) of the ResBlock tail from the raw conv outputs x2 (main branch, norm + ReLU) and xid (1x1 branch,
    norm only); None when the shape needs the separate operators (odd sizes / channel counts)."""
    N, C, H, W = x2.shape
    if (H | W) & 1 or C & 3:
        return None
    return _ResTailNorm.apply(x2, xid, float(eps), part2, partid)


def res_tail(a, b):
    """(MaxPool2d(2)(ReLU(a + b)), ReLU(a + b)); falls back to the separate operators for odd sizes / channel counts."""
    N, C, H, W = a.shape
    if (H | W) & 1 or C & 3:
        out = add(a, b, relu=True)
        return maxpool2(out), out
    return _ResTail.apply(a, b)


class _Tanh(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _dev(x)
        x = nhwc(x)
        y = torch.empty_like(x, memory_format=CL)
        _lib.check(_L().vqw_tanh_fwd(_p(x), _p(y), x.numel(), _st()), "vqw_tanh_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gy = nhwc(gy)
        gx = torch.empty_like(y, memory_format=CL)
        _lib.check(_L().vqw_tanh_bwd(_p(y), _p(gy), _p(gx), y.numel(), _st()), "vqw_tanh_bwd")
        return gx


def tanh(x):
    return _Tanh.apply(x)


def affine_(x, scale, shift):
    """In-place x*scale+shift (utils norm/denorm, utils/__init__.py:81-92)."""
    _dev(x)
    if not (x.is_contiguous() or x.is_contiguous(memory_format=CL)):
        raise RuntimeError("affine_: tensor must be dense")
    _lib.check(_L().vqw_affine(_p(x), _p(x), float(scale), float(shift), x.numel(), _st()), "vqw_affine")
    return x


class _Mse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _dev(a, b)
        a, b = nhwc(a), nhwc(b)
        if a.shape != b.shape:
            raise RuntimeError("mse_loss: shape mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        L = _L()
        out = torch.empty((), dtype=torch.float32, device=a.device)
        ws = _ws(L.vqw_reduce_ws_bytes(a.numel()), a)
        _lib.check(L.vqw_mse_fwd(_p(a), _p(b), _p(out), _p(ws), ws.numel(), a.numel(), _st()), "vqw_mse_fwd")
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        ga = torch.empty_like(a, memory_format=CL)
        _lib.check(_L().vqw_mse_bwd(_p(a), _p(b), _p(g), _p(ga), a.numel(), _st()), "vqw_mse_bwd")
        return ga, None


def mse_loss(a, b):
    """F.mse_loss(a, b, reduction='mean'); gradient flows to `a` only (targets are data)."""
    return _Mse.apply(a, b.detach())


class _WindowMse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, alpha, beta, lo, hi):
        _dev(a, b)
        a, b = nhwc(a), nhwc(b)
        if a.shape != b.shape:
            raise RuntimeError("window_mse_loss: shape mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        L = _L()
        out = torch.empty((), dtype=torch.float32, device=a.device)
        ws = _ws(L.vqw_reduce_ws_bytes(a.numel()), a)
        _lib.check(L.vqw_window_mse_fwd(_p(a), _p(b), _p(out), _p(ws), ws.numel(), a.numel(), alpha, beta, lo, hi, _st()),
                   "vqw_window_mse_fwd")
        ctx.save_for_backward(a, b)
        ctx.win = (alpha, beta, lo, hi)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = torch.empty_like(a, memory_format=CL)
        _lib.check(_L().vqw_window_mse_bwd(_p(a), _p(b), _p(g.contiguous()), _p(ga), a.numel(), *ctx.win, _st()), "vqw_window_mse_bwd")
        return ga, None, None, None, None, None


def window_map(dataset_window, target_window):
    """(alpha, beta, lo, hi) of w(x) = normalize(denormalize(x, dataset window), target window) for x in the dataset's
    normalised units (utils/__init__.py:17-51 as used by trainers/base.py:290-314); windows are (width, center, scale)."""
    w0, c0, s0 = dataset_window
    w1, c1, s1 = target_window
    vmax0, vmin0 = c0 + w0 // 2, c0 - w0 // 2
    vmax1, vmin1 = c1 + w1 // 2, c1 - w1 // 2
    # hu = (x / s0 + 0.5) * (vmax0 - vmin0) + vmin0 ;  y = ((clip(hu) - vmin1) / (vmax1 - vmin1) - 0.5) * s1
    a_hu, b_hu = (vmax0 - vmin0) / s0, 0.5 * (vmax0 - vmin0) + vmin0
    k = s1 / (vmax1 - vmin1)
    alpha, beta = a_hu * k, (b_hu - vmin1) * k - 0.5 * s1
    return float(alpha), float(beta), float(-0.5 * s1), float(0.5 * s1)


def window_mse_loss(a, b, dataset_window, target_window):
    """F.mse_loss(to_window(a), to_window(b)): the lung / mediastinal terms of the multi-window reconstruction loss."""
    return _WindowMse.apply(a, b.detach(), *window_map(dataset_window, target_window))


class _WeightedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, *terms):
        dev = terms[0].device
        _dev(*terms)
        terms = [t.reshape(()).contiguous() for t in terms]
        out = torch.empty((), dtype=torch.float32, device=dev)
        if len(terms) <= 16:
            # pointers and weights travel as kernel arguments: no device-side table, no host-to-device copy (a blocking
            # copy synchronises the stream — the host would wait for the whole forward pass before it can enqueue the
            # backward pass —, a pinned staging buffer per call makes the host allocator wait for the device now and then)
            tp = (ctypes.c_void_p * len(terms))(*[t.data_ptr() for t in terms])
            tw = (ctypes.c_float * len(terms))(*[float(x) for x in weights])
            _lib.check(_lib.load().vqw_weighted_sum_host(tp, tw, len(terms), _raw(out), _raw_stream()), "vqw_weighted_sum_host")
        else:
            ptrs = torch.tensor([t.data_ptr() for t in terms], dtype=torch.int64).to(dev)
            w = torch.tensor(list(weights), dtype=torch.float32).to(dev)
            _lib.check(_L().vqw_weighted_sum(_p(ptrs), _p(w), len(terms), _p(out), _st()), "vqw_weighted_sum")
        ctx.weights = list(weights)
        ctx.keep = terms
        return out

    @staticmethod
    def backward(ctx, g):
        gs = []
        for wgt, need in zip(ctx.weights, ctx.needs_input_grad[1:]):
            if not need:
                gs.append(None)
                continue
            o = torch.empty((), dtype=torch.float32, device=g.device)
            gc = g.contiguous()
            _lib.check(_L().vqw_affine(_p(gc), _p(o), float(wgt), 0.0, 1, _st()), "vqw_affine")
            gs.append(o)
        return (None, *gs)


def weighted_sum(terms, weights):
    """sum_i weights[i] * terms[i] for 0-dim device tensors (single_window_trainer.py:131-137)."""
    return _WeightedSum.apply(tuple(float(w) for w in weights), *terms)


# ----------------------------------------------------------------------------------------------
# vector quantisation
# ----------------------------------------------------------------------------------------------
class _VQ(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, embed, cluster_size, embed_avg, training, momentum, eps, dist_mode, id_base):
        _dev(x, embed, cluster_size, embed_avg)
        x = nhwc(x)
        N, D, H, W = x.shape
        K = embed.shape[0]
        if embed.shape[1] != D:
            raise RuntimeError("VQ: emb_dim %d does not match input channels %d" % (embed.shape[1], D))
        L = _L()
        npix = N * H * W
        ids = torch.empty((N, H, W), dtype=torch.int64, device=x.device)
        q = torch.empty_like(x, memory_format=CL)
        commit = torch.empty((), dtype=torch.float32, device=x.device)
        stats = torch.empty(K + D * K, dtype=torch.float64, device=x.device) if training else None
        ws = _ws(L.vqw_vq_ws_bytes(npix, D, K), x)
        if not (embed.is_contiguous() and cluster_size.is_contiguous() and embed_avg.is_contiguous()):
            raise RuntimeError("VQ buffers must be contiguous")
        cur = _order_begin(embed)        # the codebook read AND its EMA update stay in program order across streams
        _lib.check(L.vqw_vq_fwd(_p(x), _p(embed), _p(ids), int(id_base), _p(q), _p(commit), _p(stats), _p(ws), ws.numel(), npix, D, K, _st()),
                   "vqw_vq_fwd")
        if training:
            scale = 1.0
            if _dist_on() and dist_mode != "local":
                if dist_mode == "global":
                    _all_reduce(stats)                  # counts and sums over the global batch
                elif dist_mode == "reference":
                    # vq_module.py:187-193: embed_sum rank-averaged, counts local (C3 quirk; C2's dead
                    # N x K all-reduce is never reproduced)
                    _all_reduce(stats[K:])
                    scale = 1.0 / dist.get_world_size()
                else:
                    raise RuntimeError("unknown VQ dist_mode %r" % dist_mode)
            _lib.check(L.vqw_vq_ema_update(_p(stats), _p(embed), _p(cluster_size), _p(embed_avg), momentum, eps, scale, D, K, _st()),
                       "vqw_vq_ema_update")
        _order_end(embed, cur)
        ctx.save_for_backward(x, q)
        ctx.mark_non_differentiable(ids)
        ctx.set_materialize_grads(False)       # no zero-filled gradient for the ids; an unused q / commit arrives as None
        return q, commit, ids

    @staticmethod
    def backward(ctx, gq, gcommit, _gids):
        if gq is None and gcommit is None:
            return (None,) * 9
        x, q = ctx.saved_tensors
        gq = nhwc(gq) if gq is not None else None
        gc = gcommit.contiguous() if gcommit is not None else None
        gx = torch.empty_like(x, memory_format=CL)
        _lib.check(_L().vqw_vq_bwd(_p(x), _p(q), _p(gq), _p(gc), _p(gx), x.numel(), _st()), "vqw_vq_bwd")
        return gx, None, None, None, None, None, None, None, None


def vq_quantize(x, embed, cluster_size, embed_avg, training, momentum, eps, dist_mode="global", id_base=0):
    """-> (quantized with straight-through gradient, commit loss, ids (N,H,W) int64 = code + id_base, per pixel)."""
    return _VQ.apply(x, embed, cluster_size, embed_avg, bool(training), float(momentum), float(eps), dist_mode, int(id_base))


def kmeans_codebook(features, dict_size, seed=0, tol=1e-4, max_iter=100, return_ids=False):
    """Lloyd's k-means over pixel features (P, D) -> centres (K, D): what kmeans_pytorch.kmeans (the reference's codebook
    initialisation, unet_encoder.py:77-82) computes.  Own semantics (the dependency is absent, parity unpinned): centres
    start from K distinct random rows (seeded), an iteration = nearest-centre assignment + per-centre mean (the VQ search /
    statistics kernels and vqw_kmeans_update), empty clusters keep their centre, stop when (sum_k |delta_k|)^2 < tol as
    kmeans_pytorch does, or after max_iter.  Returns (centres, list of (mean squared distance per row, shift, empty codes)
    per iteration) and, with return_ids, the assignment of the last search (oracle/kmeans_ref.py restates this on the CPU)."""
    _dev(features)
    P, D = features.shape
    if P < dict_size:
        raise RuntimeError("k-means needs at least dict_size = %d feature rows, got %d" % (dict_size, P))
    L = _L()
    x = features.detach().contiguous().float()
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    centres = x[torch.randperm(P, generator=g)[:dict_size].to(x.device)].clone()
    ids = torch.empty(P, dtype=torch.int64, device=x.device)
    q = torch.empty_like(x)
    commit = torch.empty((), dtype=torch.float32, device=x.device)
    stats = torch.empty(dict_size + D * dict_size, dtype=torch.float64, device=x.device)
    shift = torch.empty(2, dtype=torch.float64, device=x.device)
    ws = _ws(L.vqw_vq_ws_bytes(P, D, dict_size), x)
    ws2 = _ws(16 * dict_size, x)
    history = []
    for _ in range(int(max_iter)):
        _lib.check(L.vqw_vq_fwd(_p(x), _p(centres), _p(ids), 0, _p(q), _p(commit), _p(stats), _p(ws), ws.numel(), P, D, dict_size, _st()),
                   "vqw_vq_fwd")
        _lib.check(L.vqw_kmeans_update(_p(stats), _p(centres), _p(shift), _p(ws2), ws2.numel(), D, dict_size, _st()), "vqw_kmeans_update")
        sh = shift.tolist()                      # one host sync per iteration of a one-off initialisation
        history.append((float(commit) * D, sh[0], int(sh[1])))       # commit = mean squared distance per element -> per row
        if sh[0] ** 2 < tol:
            break
    return (centres, history, ids) if return_ids else (centres, history)


def vq_lookup(ids, embed, mask=None, scale=None):
    """embed[ids] as (N,D,H,W) NHWC; optional mask (uint8 (N,H,W)) and device scalar scale."""
    _dev(ids, embed)
    if ids.dtype != torch.int64:
        ids = ids.long()
    ids = ids.contiguous()
    N, H, W = ids.shape
    K, D = embed.shape
    out = empty_nhwc(N, D, H, W, embed)
    _lib.check(_L().vqw_vq_lookup(_p(ids), _p(embed.contiguous()), _p(mask), _p(scale), _p(out), N * H * W, D, K, _st()),
               "vqw_vq_lookup")
    return out


def mask_scale(label_map):
    """run_recon.py:179-192: -> (mask uint8, ids0 int64, scale (1,) float32 = numel/sum(mask))."""
    _dev(label_map)
    lab = label_map.long().contiguous()
    mask = torch.empty(lab.shape, dtype=torch.uint8, device=lab.device)
    ids0 = torch.empty_like(lab)
    scale = torch.empty(1, dtype=torch.float32, device=lab.device)
    _lib.check(_L().vqw_mask_scale(_p(lab), _p(mask), _p(ids0), _p(scale), lab.numel(), _st()), "vqw_mask_scale")
    return mask, ids0, scale


# ----------------------------------------------------------------------------------------------
# embedding loss
# ----------------------------------------------------------------------------------------------
class _CrossLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, embed, labels_or_r, codebook_kd, dense):
        _dev(embed, labels_or_r, codebook_kd)
        e = nhwc(embed)
        B, D, H, W = e.shape
        K = codebook_kd.shape[0]
        cb = codebook_kd.detach().contiguous()
        L = _L()
        loss = torch.empty((), dtype=torch.float32, device=e.device)
        coef = torch.empty(B * K, dtype=torch.float32, device=e.device)
        ws = _ws(L.vqw_cross_ws_bytes(B, K, H * W), e)
        if dense:
            r = labels_or_r.contiguous()
            if tuple(r.shape) != (B, K, H, W) or r.dtype != torch.float32:
                raise RuntimeError("cross loss: r_ids must be float (B,K,H,W) = %s, got %s" % ((B, K, H, W), tuple(r.shape)))
            _lib.check(L.vqw_cross_loss_dense_fwd(_p(e), _p(r), _p(cb), _p(loss), _p(coef), _p(ws), ws.numel(), B, H * W, D, K, _st()),
                       "vqw_cross_loss_dense_fwd")
        else:
            r = labels_or_r.contiguous()
            if tuple(r.shape) != (B, H, W) or r.dtype != torch.int32:
                raise RuntimeError("cross loss: labels must be int32 (B,H,W)")
            _lib.check(L.vqw_cross_loss_fwd(_p(e), _p(r), _p(cb), _p(loss), _p(coef), _p(ws), ws.numel(), B, H * W, D, K, _st()),
                       "vqw_cross_loss_fwd")
        ctx.save_for_backward(e, r, cb, coef)
        ctx.dense = dense
        return loss

    @staticmethod
    def backward(ctx, g):
        e, r, cb, coef = ctx.saved_tensors
        B, D, H, W = e.shape
        K = cb.shape[0]
        g = g.contiguous()
        ge = torch.empty_like(e, memory_format=CL)
        fn = _L().vqw_cross_loss_dense_bwd if ctx.dense else _L().vqw_cross_loss_bwd
        _lib.check(fn(_p(e), _p(r), _p(cb), _p(coef), _p(g), _p(ge), B, H * W, D, K, _st()), "vqw_cross_loss_bwd")
        return ge, None, None, None


def cross_loss_labels(embed, labels, codebook_kd):
    return _CrossLoss.apply(embed, labels, codebook_kd, False)


def cross_loss_dense(embed, r, codebook_kd):
    return _CrossLoss.apply(embed, r, codebook_kd, True)


def codebook_losses(codebook_kd, margin):
    """(l_dist, l_reg) of embed_loss.py:68-88; the codebook is a buffer -> no gradient."""
    _dev(codebook_kd)
    cb = codebook_kd.detach().contiguous()
    K, D = cb.shape
    ld = torch.empty((), dtype=torch.float32, device=cb.device)
    lr = torch.empty((), dtype=torch.float32, device=cb.device)
    ws = _ws(16 * K, cb)
    _lib.check(_L().vqw_codebook_losses(_p(cb), float(margin), _p(ld), _p(lr), _p(ws), ws.numel(), D, K, _st()), "vqw_codebook_losses")
    return ld, lr


def onehot(labels, n_classes):
    _dev(labels)
    lab = labels.to(torch.int32).contiguous()
    B = lab.shape[0]
    hw = lab.numel() // B
    out = torch.empty((B, n_classes) + tuple(lab.shape[1:]), dtype=torch.float32, device=lab.device)
    _lib.check(_L().vqw_onehot(_p(lab), _p(out), B, hw, n_classes, _st()), "vqw_onehot")
    return out


def flip_labels(ids, border=0):
    """Cross-view id map for identity / h-flip views: int32 (B,H,W), 0 inside `border`."""
    _dev(ids)
    ids = ids.long().contiguous()
    B, H, W = ids.shape
    out = torch.empty((B, H, W), dtype=torch.int32, device=ids.device)
    _lib.check(_L().vqw_flip_labels(_p(ids), _p(out), int(border), B, H, W, _st()), "vqw_flip_labels")
    return out


# ----------------------------------------------------------------------------------------------
# two-view augmentation + id-map warps (data path: no autograd)
# ----------------------------------------------------------------------------------------------
def warp_image(x, minv):
    """x (B, C, H, W) fp32, minv (B, 3, 3) fp32 destination->source pixel matrices: bilinear resample, zero padding."""
    _dev(x, minv)
    x, minv = _flat(x), _flat(minv)
    B, C, H, W = x.shape
    if tuple(minv.shape) != (B, 3, 3):
        raise RuntimeError("warp_image: expected %s matrices, got %s" % ((B, 3, 3), tuple(minv.shape)))
    y = torch.empty_like(x)
    _lib.check(_L().vqw_warp_image(_p(x), _p(minv), _p(y), B, C, H, W, _st()), "vqw_warp_image")
    return y


def warp_labels(ids, minv):
    """ids (B, H, W) int64 or int32 -> int32 map sampled with nearest neighbour through minv; 0 = out of frame."""
    _dev(ids, minv)
    if ids.dtype not in (torch.int64, torch.int32):
        raise RuntimeError("warp_labels: ids must be int64 or int32 (got %s)" % ids.dtype)
    ids = ids.contiguous()
    minv = _flat(minv)
    B, H, W = ids.shape
    if tuple(minv.shape) != (B, 3, 3):
        raise RuntimeError("warp_labels: expected %s matrices, got %s" % ((B, 3, 3), tuple(minv.shape)))
    out = torch.empty((B, H, W), dtype=torch.int32, device=ids.device)
    _lib.check(_L().vqw_warp_labels(_p(ids), int(ids.dtype == torch.int64), _p(minv), _p(out), B, H, W, _st()), "vqw_warp_labels")
    return out


def photometric(x, params, noise=None):
    """Per-sample brightness add, contrast multiply (clamped to [0,1]), posterize, + std * noise; params (B, 4)."""
    _dev(x, params, noise)
    x, params = _flat(x), _flat(params)
    B = x.shape[0]
    if tuple(params.shape) != (B, 4):
        raise RuntimeError("photometric: params must be (%d, 4)" % B)
    if noise is not None:
        noise = _flat(noise)
        if noise.shape != x.shape:
            raise RuntimeError("photometric: noise shape %s != %s" % (tuple(noise.shape), tuple(x.shape)))
    y = torch.empty_like(x)
    _lib.check(_L().vqw_photometric(_p(x), _p(params), _p(noise), _p(y), B, x.numel() // B, _st()), "vqw_photometric")
    return y


def gauss_blur(x, taps, apply=None):
    """Separable Gaussian blur of (B, C, H, W) with 1-D `taps`, reflect border; apply (B,) uint8 selects samples."""
    _dev(x, taps, apply)
    x, taps = _flat(x), _flat(taps)
    B, C, H, W = x.shape
    if apply is not None:
        apply = apply.to(torch.uint8).contiguous()
        if apply.numel() != B:
            raise RuntimeError("gauss_blur: apply must have one entry per sample")
    tmp, y = torch.empty_like(x), torch.empty_like(x)
    _lib.check(_L().vqw_gauss_blur(_p(x), _p(taps), _p(apply), _p(tmp), _p(y), B, C, H, W, taps.numel(), _st()), "vqw_gauss_blur")
    return y


# ----------------------------------------------------------------------------------------------
# second training step: PatchGAN discriminator pieces (strided conv, BatchNorm(affine)+LeakyReLU, hinge losses)
# ----------------------------------------------------------------------------------------------
class _SConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, slope):
        _dev(x, weight, bias)
        x, w = nhwc(x), nhwc(weight)
        Cout, Cin, ks, _ = weight.shape
        N, _, H, W = x.shape
        if x.shape[1] != Cin:
            raise RuntimeError("sconv2d: input has %d channels, weight expects %d" % (x.shape[1], Cin))
        Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        if bias is not None:
            bias = _flat(bias)
        y = empty_nhwc(N, Cout, Ho, Wo, x)
        L = _L()
        ws = _ws(L.vqw_sconv_fwd_ws_bytes(N, H, W, Cin, Cout, ks, stride, pad), x)
        _lib.check(L.vqw_sconv_fwd(_p(x), _p(w), _p(bias), _p(y), _p(ws), ws.numel(), N, H, W, Cin, Cout, ks, stride, pad,
                                   float(slope), _st()), "vqw_sconv_fwd")
        ctx.save_for_backward(x, w, y if slope != 1.0 else None)
        ctx.cfg = (stride, pad, float(slope), bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        stride, pad, slope, has_bias = ctx.cfg
        L = _L()
        gy = nhwc(gy)
        Cout, Cin, ks, _ = w.shape
        N, _, H, W = x.shape
        if y is not None:
            gm = torch.empty_like(y, memory_format=CL)
            _lib.check(L.vqw_leaky_relu_bwd(_p(y), _p(gy), _p(gm), slope, gy.numel(), _st()), "vqw_leaky_relu_bwd")
            gy = gm
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x, memory_format=CL)
            ws = _ws(L.vqw_sconv_dgrad_ws_bytes(N, H, W, Cin, Cout, ks, stride, pad), gy)
            _lib.check(L.vqw_sconv_dgrad(_p(gy), _p(w), _p(gx), _p(ws), ws.numel(), N, H, W, Cin, Cout, ks, stride, pad, _st()),
                       "vqw_sconv_dgrad")
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            gw = torch.empty((Cout, Cin, ks, ks), dtype=torch.float32, device=gy.device, memory_format=CL)
            gb = torch.empty(Cout, dtype=torch.float32, device=gy.device) if has_bias else None
            ws = _ws(L.vqw_sconv_wgrad_ws_bytes(Cin, Cout, ks, N, H, W, stride, pad), gy)
            _lib.check(L.vqw_sconv_wgrad(_p(x), _p(gy), _p(gw), _p(gb), _p(ws), ws.numel(), N, H, W, Cin, Cout, ks, stride, pad, 0,
                                         _st()), "vqw_sconv_wgrad")
        return gx, gw, gb, None, None, None


def sconv2d(x, weight, bias=None, stride=1, padding=0, slope=1.0):
    """nn.Conv2d(k, stride in {1,2}, padding) with an optional LeakyReLU(slope) epilogue."""
    return _SConv.apply(x, weight, bias, int(stride), int(padding), float(slope))


class _BnLrelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, slope, sync):
        _dev(x, gamma, beta)
        x = nhwc(x)
        gamma, beta = _flat(gamma), _flat(beta)
        N, C, H, W = x.shape
        L = _L()
        mr = torch.empty(2 * C, dtype=torch.float32, device=x.device)
        count = float(N * H * W)
        if training:
            sums = torch.empty(2 * C, dtype=torch.float64, device=x.device)
            ws = _ws(L.vqw_plane_ws_bytes(N, C, H * W), x)
            _lib.check(L.vqw_bn_partial_stats(_p(x), _p(sums), _p(ws), ws.numel(), N, H * W, C, _st()), "vqw_bn_partial_stats")
            if sync and _dist_on():
                _all_reduce(sums)
                count *= dist.get_world_size()
            cur = _order_begin(running_mean)
            _lib.check(L.vqw_bn_finalize(_p(sums), count, _p(mr), _p(running_mean), _p(running_var), momentum, eps, C, _st()),
                       "vqw_bn_finalize")
            if nbt is not None:
                _bump_counter(nbt)
            _order_end(running_mean, cur)
        else:
            _lib.check(L.vqw_bn_eval_stats(_p(running_mean), _p(running_var), _p(mr), eps, C, _st()), "vqw_bn_eval_stats")
        y = torch.empty_like(x, memory_format=CL)
        _lib.check(L.vqw_bn_affine_fwd(_p(x), _p(mr), _p(gamma), _p(beta), _p(y), N * H * W, C, float(slope), _st()), "vqw_bn_affine_fwd")
        ctx.save_for_backward(x, gamma, beta, mr)
        ctx.cfg = (training, float(slope), count, sync)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, beta, mr = ctx.saved_tensors
        training, slope, count, sync = ctx.cfg
        N, C, H, W = x.shape
        L = _L()
        gy = nhwc(gy)
        sums = torch.empty(2 * C, dtype=torch.float64, device=x.device)
        ws = _ws(L.vqw_plane_ws_bytes(N, C, H * W), x)
        _lib.check(L.vqw_bn_affine_bwd_reduce(_p(x), _p(mr), _p(gamma), _p(beta), _p(gy), _p(sums), _p(ws), ws.numel(), N, H * W, C,
                                              slope, _st()), "vqw_bn_affine_bwd_reduce")
        # dgamma / dbeta are this rank's sums (DDP averages parameter gradients); dx needs the global-batch sums
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        gx = torch.empty_like(x, memory_format=CL)
        if training and sync and _dist_on():
            local = sums.clone()
            _all_reduce(sums)
            _lib.check(L.vqw_bn_affine_bwd_apply(_p(x), _p(mr), _p(gamma), _p(beta), _p(gy), _p(sums), count, _p(gx), None, None,
                                                 N * H * W, C, slope, 1, 0, _st()), "vqw_bn_affine_bwd_apply")
            dbeta.copy_(local[0::2])
            dgamma.copy_(local[1::2])
        else:
            _lib.check(L.vqw_bn_affine_bwd_apply(_p(x), _p(mr), _p(gamma), _p(beta), _p(gy), _p(sums), count, _p(gx), _p(dgamma),
                                                 _p(dbeta), N * H * W, C, slope, int(training), 0, _st()), "vqw_bn_affine_bwd_apply")
        return gx, dgamma, dbeta, None, None, None, None, None, None, None, None


def batch_norm_lrelu(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, slope=0.2, sync=True,
                     num_batches_tracked=None):
    """nn.BatchNorm2d (affine) followed by nn.LeakyReLU(slope) (slope = 1: plain BatchNorm)."""
    return _BnLrelu.apply(x, gamma, beta, running_mean, running_var, num_batches_tracked, bool(training), float(momentum),
                          float(eps), float(slope), bool(sync))


class _Hinge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        _dev(x)
        x = x.contiguous() if not (x.is_contiguous() or x.is_contiguous(memory_format=CL)) else x
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        _lib.check(_L().vqw_hinge_fwd(_p(x), x.numel(), mode, _p(loss), _st()), "vqw_hinge_fwd")
        ctx.save_for_backward(x)
        ctx.mode = mode
        return loss

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        gx = torch.empty_like(x)
        g = g.contiguous().float()
        _lib.check(_L().vqw_hinge_bwd(_p(x), x.numel(), ctx.mode, _p(g), _p(gx), _st()), "vqw_hinge_bwd")
        return gx, None


def hinge_real(logits):
    """mean(relu(1 - logits)) (gan_loss.py:7)"""
    return _Hinge.apply(logits, 0)


def hinge_fake(logits):
    """mean(relu(1 + logits)) (gan_loss.py:8)"""
    return _Hinge.apply(logits, 1)


def neg_mean(logits):
    """-mean(logits): the generator loss (single_window_trainer.py:463)"""
    return _Hinge.apply(logits, 2)


def set_conv_backend(mode):
    """0 = auto (MFMA kernels where shapes allow), 1 = generic VALU kernels only (testing), 2 = no LDS-resident tile kernels,
    3 = auto without any Winograd-form kernel (F(2x2, 3x3) forward / input / weight gradients and the nine-product forms of the
    up-sampled layers: plain direct-form arithmetic everywhere).  Modes 2 and 3 are the A/B references of the tests.  Returns
    the previous mode."""
    return _L().vqw_set_conv_backend(int(mode))


# ----------------------------------------------------------------------------------------------
# optional paths: PixelShuffle(2), DropBlock, SoftDice / Focal
# ----------------------------------------------------------------------------------------------
class _PixelShuffle2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _dev(x)
        x = nhwc(x)
        N, C4, h, w = x.shape
        if C4 % 4:
            raise RuntimeError("pixel_shuffle(2): channels must be a multiple of 4")
        y = empty_nhwc(N, C4 // 4, 2 * h, 2 * w, x)
        _lib.check(_L().vqw_pixel_shuffle2(_p(x), _p(y), N, 2 * h, 2 * w, C4 // 4, 0, _st()), "vqw_pixel_shuffle2")
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = nhwc(gy)
        N, C, H, W = gy.shape
        gx = empty_nhwc(N, 4 * C, H // 2, W // 2, gy)
        _lib.check(_L().vqw_pixel_shuffle2(_p(gy), _p(gx), N, H, W, C, 1, _st()), "vqw_pixel_shuffle2")
        return gx


def pixel_shuffle2(x):
    return _PixelShuffle2.apply(x)


def dropblock_mask(seed_mask, block_size):
    """seed (B,H,W) float {0,1} on device -> (keep (B,H,W), scale (1,) = numel/sum(keep))."""
    _dev(seed_mask)
    seed = seed_mask.float().contiguous()
    B, H, W = seed.shape
    keep = torch.empty_like(seed)
    scale = torch.empty(1, dtype=torch.float32, device=seed.device)
    _lib.check(_L().vqw_dropblock_mask(_p(seed), _p(keep), _p(scale), B, H, W, int(block_size), _st()), "vqw_dropblock_mask")
    return keep, scale


class _DropBlockApply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, keep, scale):
        _dev(x, keep, scale)
        x = nhwc(x)
        N, C, H, W = x.shape
        y = torch.empty_like(x, memory_format=CL)
        _lib.check(_L().vqw_dropblock_apply(_p(x), _p(keep), _p(scale), _p(y), N * H * W, C, _st()), "vqw_dropblock_apply")
        ctx.save_for_backward(keep, scale)
        return y

    @staticmethod
    def backward(ctx, gy):
        keep, scale = ctx.saved_tensors
        gy = nhwc(gy)
        N, C, H, W = gy.shape
        gx = torch.empty_like(gy, memory_format=CL)
        _lib.check(_L().vqw_dropblock_apply(_p(gy), _p(keep), _p(scale), _p(gx), N * H * W, C, _st()), "vqw_dropblock_apply")
        return gx, None, None


def dropblock_apply(x, keep, scale):
    return _DropBlockApply.apply(x, keep, scale)


class _SegLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index, smooth, gamma, eps):
        _dev(logits, target)
        z = logits.contiguous()
        t = target.float().contiguous()
        if z.dtype != torch.float32 or z.shape != t.shape or z.dim() < 2:
            raise RuntimeError("seg losses: fp32 logits and same-shape one-hot targets (B,C,...) expected")
        B, C = z.shape[0], z.shape[1]
        HW = z.numel() // (B * C)
        L = _L()
        out = torch.empty(2, dtype=torch.float32, device=z.device)
        sums = torch.empty(2 * C + 2, dtype=torch.float64, device=z.device)
        ws = _ws(L.vqw_seg_ws_bytes(C), z)
        _lib.check(L.vqw_seg_losses_fwd(_p(z), _p(t), _p(out), _p(sums), _p(ws), ws.numel(), B, HW, C, ignore_index, smooth, gamma,
                                        eps, _st()), "vqw_seg_losses_fwd")
        ctx.save_for_backward(z, t, sums)
        ctx.cfg = (ignore_index, smooth, gamma, eps)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_dice, g_focal):
        z, t, sums = ctx.saved_tensors
        ignore_index, smooth, gamma, eps = ctx.cfg
        B, C = z.shape[0], z.shape[1]
        HW = z.numel() // (B * C)
        gz = torch.empty_like(z)
        gd = g_dice.contiguous() if g_dice is not None else None
        gf = g_focal.contiguous() if g_focal is not None else None
        _lib.check(_L().vqw_seg_losses_bwd(_p(z), _p(t), _p(sums), _p(gd), _p(gf), _p(gz), B, HW, C, ignore_index, smooth, gamma,
                                           eps, _st()), "vqw_seg_losses_bwd")
        return gz, None, None, None, None, None


def seg_losses(logits, target, ignore_index=-1, smooth=1e-6, gamma=2.0, eps=1e-6):
    """-> (soft dice loss, focal loss) of functions/seg_loss.py for NCHW logits and one-hot targets."""
    return _SegLosses.apply(logits, target, int(ignore_index), float(smooth), float(gamma), float(eps))
