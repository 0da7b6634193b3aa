"""Adam with the interface and numerics of torch.optim.Adam (the optimizer train.py:443-469 of the reference
builds), executed as ONE kernel launch for all parameter tensors (nmf_adam_step, csrc/adam.hip) instead of
~13 foreach launches per param group.  param_groups / state_dict / LambdaLR work as with torch.optim.Adam
(state keys: step, exp_avg, exp_avg_sq); amsgrad / maximize / capturable are not supported (the reference does not
use them)."""
import ctypes as C
import math

import numpy as np
import torch

from . import hip


# callables run behind every FusedAdam.step() (not step_unhooked: the Trainer prefetches itself).  nmf_amd/fast_step.py registers
# the table prefetch of a pass that serves TensorNeRF.forward under a foreign training loop (weak references: a dead pass drops out)
AFTER_STEP = []


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        self._slots = None
        self._plan = None
        self._last_grads = None
        # 0-d fp32 device tensor or None: a non-finite value makes the next step() a no-op on the device (Trainer: the summed
        # loss of the step's chunks -- train.py:704-705 without the host read-back)
        self.guard = None
        self._gidx = None
        self._touched = None
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise hip.NmfHipError("FusedAdam.step does not take a closure")
        if not self._step_planned():
            self._step_checked()
        for cb in list(AFTER_STEP):     # (a copy: a callback of a dead pass removes itself)  e.g. a fused training pass queues the next step's derived tables on its side stream
            cb()

    def step_unhooked(self):
        """The same update without torch.optim.Optimizer's per-call wrapper around `step` (profiler record + the pre / post
        hook dispatch: ~50 us on a path where the host is the critical resource).  For callers that registered no optimizer
        hooks -- the Trainer; the kernels touch raw pointers, so no grad mode is involved."""
        if not self._step_planned():
            with torch.no_grad():
                self._step_checked()

    def zero_grad(self, set_to_none=True):
        """torch.optim.Optimizer.zero_grad without its per-call profiler / hook plumbing (50 us per step for 31 tensors)"""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group["params"]:
                p.grad = None

    def _step_planned(self):
        """Steady state: the same tensors carry gradients of the same layout as in the previous step, so the slot table is
        already filled except for the gradient pointers and the step-dependent factors (the full validation of
        _step_checked costs ~9 us per tensor, i.e. 270 us of host time per step on a path where the GPU is waiting)."""
        plan = self._plan
        if plan is None:
            return False
        slots, entries, skipped, hyper0, col_u64, col_f64 = plan
        hyper = [(g["betas"][0], g["betas"][1], float(g["eps"]), float(g["weight_decay"])) for g in self.param_groups]
        if hyper != hyper0:
            return False
        for p in skipped:
            if p.grad is not None:
                return False
        last = self._last_grads
        same = last is not None and len(last) == len(entries)
        if same:                                        # the training pass hands over the SAME gradient tensors every step
            for e, g0 in zip(entries, last):
                if e[0].grad is not g0 or e[0].data_ptr() != e[3]:
                    same = False
                    break
        if same:
            gptr = None                                 # validated in the previous step: pointers and layouts stand
        else:
            gptr, grads = [], []
            for p, st, gi, pptr, gstride in entries:    # validate everything before touching any state
                g = p.grad
                if g is None or g.dtype != p.dtype or g.stride() != gstride or p.data_ptr() != pptr or g.is_sparse:
                    self._last_grads = None
                    return False
                gptr.append(g.data_ptr())
                grads.append(g)
            self._last_grads = grads
        lrs = [float(g["lr"]) for g in self.param_groups]
        n = len(entries)
        steps = [e[1]["step"] for e in entries]
        if n and steps.count(steps[0]) == n:
            # all tensors have taken the same number of steps (always, unless a tensor joined later): one pair of factors per
            # param group, spread over the tensors with the plan's group-index array
            t = steps[0] + 1
            fs = np.empty(len(lrs), dtype=np.float64)
            fb = np.empty(len(lrs), dtype=np.float64)
            for gi, lr in enumerate(lrs):
                beta1, beta2 = hyper[gi][0], hyper[gi][1]
                fs[gi] = lr / (1 - beta1 ** t)
                fb[gi] = math.sqrt(1 - beta2 ** t)
            gidx = self._gidx
            if gidx is None or gidx.shape[0] != n:
                gidx = self._gidx = np.asarray([e[2] for e in entries], dtype=np.int64)
            step_size, bc2 = fs[gidx], fb[gidx]
            for e in entries:
                e[1]["step"] = t
        else:
            memo, step_size, bc2 = {}, [], []
            for p, st, gi, pptr, gstride in entries:
                st["step"] = t = st["step"] + 1
                f = memo.get((gi, t))
                if f is None:
                    beta1, beta2 = hyper[gi][0], hyper[gi][1]
                    f = memo[(gi, t)] = (lrs[gi] / (1 - beta1 ** t), math.sqrt(1 - beta2 ** t))
                step_size.append(f[0])
                bc2.append(f[1])
        if n:
            # the slot table is a dense array of 12 eight-byte words per tensor: write the three per-step columns at once
            if gptr is not None:
                col_u64[:n, 1] = gptr
            col_f64[:n, 9] = step_size
            col_f64[:n, 10] = bc2
            hip.adam_step(slots, n, self.guard)
            torch.autograd.graph.increment_version(self._touched if self._touched is not None and len(self._touched) == n
                                                   else [e[0] for e in entries])
        return True

    def _step_checked(self):
        self._last_grads = None
        n_params = sum(len(g["params"]) for g in self.param_groups)
        if self._slots is None or len(self._slots) < n_params:
            self._slots = (hip.AdamSlot * max(n_params, 1))()
        slots, n, keep, touched = self._slots, 0, [], []
        entries, skipped, plannable = [], [], True
        for gi, group in enumerate(self.param_groups):
            beta1, beta2 = group["betas"]
            lr, eps, wd = float(group["lr"]), float(group["eps"]), float(group["weight_decay"])
            for p in group["params"]:
                g = p.grad
                if g is None:
                    skipped.append(p)
                    continue
                if g.is_sparse:
                    raise hip.NmfHipError("FusedAdam does not support sparse gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = t = int(st["step"]) + 1
                if p.dtype not in (torch.float32, torch.float64) or g.dtype != p.dtype:
                    raise hip.NmfHipError(f"FusedAdam: unsupported dtype {p.dtype}/{g.dtype}")
                if not _same_layout(g, p) or not _dense(p):
                    if not _dense(p):
                        raise hip.NmfHipError("FusedAdam: parameter storage must be dense")
                    g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
                    keep.append(g)
                    plannable = False                      # this gradient needs a re-layout copy every step
                s = slots[n]
                s.param, s.grad = p.data_ptr(), g.data_ptr()
                s.exp_avg, s.exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                s.numel = p.numel()
                s.beta1, s.beta2, s.eps, s.weight_decay = beta1, beta2, eps, wd
                s.step_size = lr / (1 - beta1 ** t)
                s.bc2_sqrt = math.sqrt(1 - beta2 ** t)
                s.is_f64 = 1 if p.dtype == torch.float64 else 0
                n += 1
                touched.append(p)
                entries.append((p, st, gi, p.data_ptr(), p.grad.stride()))
        if n:
            hip.adam_step(slots, n, self.guard)
            # the kernel writes through raw pointers: tell autograd (and every cache keyed on Tensor._version -- packed
            # density tables, SAT, stacked head weights, host mirrors of scalars) that the parameters changed
            torch.autograd.graph.increment_version(touched)
        if plannable:
            hyper = [(g["betas"][0], g["betas"][1], float(g["eps"]), float(g["weight_decay"])) for g in self.param_groups]
            words = C.sizeof(hip.AdamSlot) // 8
            assert words == 12
            self._plan = (slots, entries, skipped, hyper, np.frombuffer(slots, dtype=np.uint64).reshape(-1, words),
                          np.frombuffer(slots, dtype=np.float64).reshape(-1, words))
            self._gidx, self._touched = None, list(touched)
        else:
            self._plan = None
        return None

    def add_param_group(self, group):
        self._plan = None
        self._last_grads = None
        return super().add_param_group(group)

    def load_state_dict(self, state_dict):
        self._plan = None
        self._last_grads = None
        return super().load_state_dict(state_dict)


def _same_layout(a, b):
    """same memory order: strides agree on every dimension of extent > 1 (size-1 dimensions carry arbitrary strides)"""
    return a.shape == b.shape and all(sa == sb for sa, sb, n in zip(a.stride(), b.stride(), a.shape) if n > 1)


def _dense(t):
    if t.numel() == 0:
        return True
    sz = sorted(((st, s) for st, s in zip(t.stride(), t.shape) if s > 1))
    expect = 1
    for st, s in sz:
        if st != expect:
            return False
        expect *= s
    return True
