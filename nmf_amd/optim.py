"""Adam with the interface and numerics of torch.optim.Adam (the optimizer train.py:443-469 of the reference
builds), executed as ONE kernel launch for all parameter tensors (nmf_adam_step, csrc/adam.hip) instead of
~13 foreach launches per param group.  param_groups / state_dict / LambdaLR work as with torch.optim.Adam
(state keys: step, exp_avg, exp_avg_sq); amsgrad / maximize / capturable are not supported (the reference does not
use them)."""
import math

import torch

from . import hip


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._slots = None

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise hip.NmfHipError("FusedAdam.step does not take a closure")
        n_params = sum(len(g["params"]) for g in self.param_groups)
        if self._slots is None or len(self._slots) < n_params:
            self._slots = (hip.AdamSlot * max(n_params, 1))()
        slots, n, keep, touched = self._slots, 0, [], []
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            lr, eps, wd = float(group["lr"]), float(group["eps"]), float(group["weight_decay"])
            for p in group["params"]:
                g = p.grad
                if g is None:
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
        if n:
            hip.adam_step(slots, n)
            # the kernel writes through raw pointers: tell autograd (and every cache keyed on Tensor._version -- packed
            # density tables, SAT, stacked head weights, host mirrors of scalars) that the parameters changed
            torch.autograd.graph.increment_version(touched)
        return None


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
