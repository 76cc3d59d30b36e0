"""Adam over the fused HIP kernel (vqw_adam_step).

Drop-in for the `torch.optim.Adam(params, lr, betas, weight_decay)` instances the
reference builds in trainers/base.py:165-175: same constructor arguments, same update
rule, same state_dict layout ('step', 'exp_avg', 'exp_avg_sq' per parameter).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from . import ops as _ops


# One multi-tensor launch per parameter group (VQW_ADAM_MULTI, default on since round 3) needs a pointer table uploaded every
# step (gradient tensors are new every step; a small ring of pinned buffers).  Measured on the 95 ms step of round 3:
# 95.36 / 95.13 ms per-tensor (147 launches of ~3 us at the very end of the step, where nothing overlaps them) against
# 95.04 / 94.88 ms multi-tensor.  VQW_ADAM_MULTI=0 restores the per-tensor launches; a group whose tensors do not share a
# step count or a layout falls back to them by itself.
MULTI_TENSOR = os.environ.get("VQW_ADAM_MULTI", "1") != "0"
# elements per workgroup of the multi-tensor launch (VQW_ADAM_CHUNK): ~950 workgroups for the R-cfg model; 8 192 ... 65 536 measure
# the same step time - the optimiser's launches overlap the next step's first kernels
CHUNK = int(os.environ.get("VQW_ADAM_CHUNK", 1 << 14))


def _same_layout(a, b):
    """Same element order in memory: equal strides on every dimension of size > 1 (a size-1 dimension's stride is
    arbitrary: a (16, 1, 3, 3) weight is 'channels_last' with NCHW strides, its freshly allocated gradient is not)."""
    return a.shape == b.shape and all(sa == sb for sa, sb, n in zip(a.stride(), b.stride(), a.shape) if n > 1)


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._pin_rings = {}          # (group, table size) -> [[pinned int64 buffer, event of its last upload], ...]

    def load_state_dict(self, state_dict):
        """Accepts the state a stock torch.optim.Adam saved for the reference (Lightning checkpoint `optimizer_states`):
        its moments keep the NCHW strides they were saved with and `step` is a tensor; both are brought to this
        optimiser's conventions (moments in the parameter's own - channels_last - layout, integer step)."""
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if not st:
                    continue
                if torch.is_tensor(st.get("step")):
                    st["step"] = int(st["step"].item())
                for k in ("exp_avg", "exp_avg_sq"):
                    t = st.get(k)
                    if t is not None and not _same_layout(t, p):
                        st[k] = torch.empty_strided(p.size(), p.stride(), dtype=p.dtype, device=p.device).copy_(t)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L, st = _ops._L(), _ops._st()         # torch.ops.vqw.adam_step / adam_multi (hipops/library.py)
        for group in self.param_groups:
            b1, b2 = group["betas"]
            if MULTI_TENSOR and self._step_multi(L, st, group):
                continue
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("hipops.Adam needs parameters on a ROCm device")
                g = p.grad
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                t = state["step"]
                # element-wise update: any dense layout works as long as p, g, m, v share it
                if not _same_layout(g, p):
                    g = torch.empty_strided(p.size(), p.stride(), dtype=p.dtype, device=p.device).copy_(p.grad)
                m, v = state["exp_avg"], state["exp_avg_sq"]
                if not (_same_layout(m, p) and _same_layout(v, p)):
                    raise RuntimeError("Adam state layout does not match the parameter layout")
                _lib.check(L.vqw_adam_step(_ops._p(p), _ops._p(g), _ops._p(m), _ops._p(v), p.numel(),
                                           group["lr"], b1, b2, group["eps"], group["weight_decay"],
                                           1.0 - b1 ** t, 1.0 - b2 ** t, st), "vqw_adam_step")
        _ops.bump_weight_epoch()      # parameters changed through raw pointers: invalidate derived weight layouts
        return loss

    def _step_multi(self, L, st, group):
        """All tensors of the group in one launch.  Falls back (returns False) when the tensors do not share a step
        count or a gradient needs a layout copy; the per-tensor path then handles the group."""
        params = [p for p in group["params"] if p.grad is not None]
        if not params:
            return True
        dev = params[0].device
        steps = set()
        for p in params:
            if not p.is_cuda or p.device != dev or not _same_layout(p.grad, p):
                return False
            state = self.state[p]
            if len(state) == 0:
                state["step"] = 0
                state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if not (_same_layout(state["exp_avg"], p) and _same_layout(state["exp_avg_sq"], p)):
                return False
            steps.add(state["step"])
        if len(steps) != 1:
            return False
        t = steps.pop() + 1
        rows = []
        for p in params:
            state = self.state[p]
            state["step"] = t
            ptrs = (p.data_ptr(), p.grad.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr())
            n = p.numel()
            for off in range(0, n, CHUNK):
                rows.append((ptrs[0] + 4 * off, ptrs[1] + 4 * off, ptrs[2] + 4 * off, ptrs[3] + 4 * off, min(CHUNK, n - off)))
        # the pointer table goes up through a small ring of persistent pinned buffers (a fresh pinned allocation per
        # step can make the host allocator wait for the device); a slot is reused only after its copy has run
        arr = np.asarray(rows, dtype=np.int64)
        ring = self._pin_rings.setdefault((id(group), arr.size), [])
        slot = None
        for cand in ring:
            if cand[1].query():
                slot = cand
                break
        if slot is None:
            if len(ring) >= 8:
                slot = ring[0]
                slot[1].synchronize()
            else:
                slot = [torch.empty(arr.size, dtype=torch.int64).pin_memory(), torch.cuda.Event()]
                ring.append(slot)
        slot[0].numpy()[:] = arr.reshape(-1)
        table = slot[0].to(dev, non_blocking=True).view(arr.shape)
        slot[1].record(torch.cuda.current_stream())
        b1, b2 = group["betas"]
        _lib.check(L.vqw_adam_multi(_ops._p(table), len(rows), group["lr"], b1, b2, group["eps"],
                                    group["weight_decay"], 1.0 - b1 ** t, 1.0 - b2 ** t, st), "vqw_adam_multi")
        table.record_stream(torch.cuda.current_stream())
        return True
