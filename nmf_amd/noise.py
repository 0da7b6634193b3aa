"""Random-number plumbing of the hot path.  The reference consumes torch's global RNG at fixed call sites
(RNG); the operators here take a noise source object instead so that parity tests can
replay the reference's draws:

  DeviceNoise  production: draws on the GPU (torch Philox) / inside the kernels (march jitter);
               draws the reference multiplies by zero (start_std=0, mipnoise=0) are skipped.
  ReplayNoise  tests: replays a recorded tape, or re-creates the reference's draws from torch's seeded CPU
               generator (same call order, same shapes, including the unused draws).
"""
import math
import random

import torch


class DeviceNoise:
    def __init__(self, device, seed=0):
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)
        self.seed = seed
        self.calls = 0

    # march jitter is generated inside the kernel (Philox keyed by seed/offset)
    def jitter(self, B, N):
        self.calls += 1
        return None, (self.seed, self.calls)

    def normal(self, shape):
        return torch.randn(shape, device=self.device, generator=self.gen)

    def normal_deferred(self, shape):
        """an i.i.d. normal [M, C] matrix of which only some rows will be used: nothing is drawn until rows() asks"""
        return tuple(shape)

    def rows(self, deferred, rows):
        return torch.randn((rows.shape[0],) + deferred[1:], device=self.device, generator=self.gen)

    def uniform(self, shape):
        return torch.rand(shape, device=self.device, generator=self.gen)

    def skip(self, kind, shape):
        return None

    def select_dense(self, b, N, ray_id, step_id):
        """U at the kept entries of the dense [b,N] matrix + sum of U over ALL b*N entries
        (modules/pt_selectors.py:24 perturbs the dense weight matrix).  The dropped entries only enter
        through their sum, which is drawn from its exact large-n normal limit."""
        M = ray_id.shape[0]
        u = self.uniform((M,))
        n_drop = b * N - M
        extra = 0.5 * n_drop + math.sqrt(max(n_drop, 0) / 12.0) * random.Random(self.calls * 1000003 + self.seed).gauss(0.0, 1.0)
        self.calls += 1
        return u, u.sum(dtype=torch.float64) + extra


class ReplayNoise:
    """tape: list of (kind, tensor) in the reference's call order, or None to draw from torch's global CPU RNG
    (seed it with torch.manual_seed(s) first) in that same order."""

    def __init__(self, device, tape=None):
        self.device = torch.device(device)
        self.tape = list(tape) if tape is not None else None
        self.pos = 0

    def _next(self, kind, shape):
        shape = tuple(int(s) for s in shape)
        if self.tape is not None:
            k, t = self.tape[self.pos]
            self.pos += 1
            assert k.startswith("randn") == (kind == "randn"), (k, kind, self.pos)
            assert tuple(t.shape) == shape, (k, tuple(t.shape), shape, self.pos)
            return t
        return torch.randn(shape) if kind == "randn" else torch.rand(shape)

    def jitter(self, B, N):
        return self._next("rand", (B, N)).to(self.device).contiguous(), (0, 0)

    def normal(self, shape):
        return self._next("randn", shape).to(self.device)

    def normal_deferred(self, shape):
        return self._next("randn", shape).to(self.device)          # consumed at the reference's position in the stream

    def rows(self, deferred, rows):
        return deferred[rows.long()].contiguous()

    def uniform(self, shape):
        return self._next("rand", shape).to(self.device)

    def skip(self, kind, shape):
        self._next(kind, shape)
        return None

    def select_dense(self, b, N, ray_id, step_id):
        U = self._next("rand", (b, N))
        total = U.sum(dtype=torch.float64)
        u = U.to(self.device)[ray_id.long(), step_id.long()].contiguous()
        return u, total.to(self.device)
