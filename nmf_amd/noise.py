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


class Pins:
    """TESTS ONLY: replayed bookkeeping decisions of a reference run, carried by a ReplayNoise next to the replayed random
    draws (both are "what the reference decided / drew at this call site").  The product modules keep no test state: the
    sampler, the shading model and the tape-free training pass read `noise.pins` (None on DeviceNoise).

      counts[level]         int32 [M]   secondary rays per kept sample (output of select_bounces, pt_selectors.py:5-60)
      retrace_order[level]  int64 [R]   the reference's argsort of the re-trace scores (models/microfacet.py:522)
      valid[level]          bool [R,N]  occupancy decisions of the candidate steps of that level's rays (alphagrid.py:341-346)
      exact_retrace_order   sort the scores even when every secondary ray is re-traced, as models/microfacet.py:506-509 does
      trace                 dict that receives intermediate tensors (own counts, scores, order, rgb_map ...), or None
      trace_scores          with a trace: take the argsort branch at every level so that scores / order are recorded
      valid_flips           out: 64-step words the marcher itself decided differently from `valid`"""

    def __init__(self, counts=None, retrace_order=None, valid=None, exact_retrace_order=False, trace=True, trace_scores=True):
        self.counts = dict(counts or {})
        self.retrace_order = dict(retrace_order or {})
        self.valid = dict(valid or {})
        self.exact_retrace_order = bool(exact_retrace_order)
        self.trace = {} if trace else None
        self.trace_scores = bool(trace_scores)
        self.valid_flips = None

    def counts_for(self, level, own):
        """own = what the kernel decided; -> the pinned counts when they exist for this level and size"""
        if self.trace is not None:
            self.trace[f"counts_own{level}"] = own
        c = self.counts.get(level)
        if c is not None and c.shape[0] == own.shape[0]:
            own = c.to(own.device).int().contiguous()
        if self.trace is not None:
            self.trace[f"counts{level}"] = own
        return own

    def sorts(self, level):
        """does this level take the argsort branch even when every ray is re-traced?"""
        return level in self.retrace_order or self.exact_retrace_order or (self.trace is not None and self.trace_scores)


class DeviceNoise:
    """Draws of one pass come out of two pools (uniform / normal) filled with ONE torch.rand / torch.randn each at the
    start of the pass (begin_pass, sized by the previous pass's consumption + 25 %): six small generator launches per
    training step become two that are issued while the GPU still works on the previous step.  A draw that does not fit
    (first pass, growing scene) falls back to its own launch.  The numbers are i.i.d. either way; only the assignment of
    generator outputs to call sites differs from draw-per-call."""

    pins = None
    capacity_draws = True       # i.i.d. draws with no call order to reproduce: a caller may draw for a bound instead of the exact count

    def __init__(self, device, seed=0, pooled=True):
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)
        self.seed = seed
        self.calls = 0
        self.pooled = pooled and self.device.type == "cuda"
        self._pool = {"u": None, "n": None}
        self._off = {"u": 0, "n": 0}
        self._want = {"u": 0, "n": 0}

    def begin_pass(self):
        """called by TensorNeRF.forward at recursion 0, before anything of the pass is drawn"""
        if not self.pooled:
            return
        for kind, fn in (("u", torch.rand), ("n", torch.randn)):
            need = (int(self._want[kind] * 1.25) + 1024 + 3) & ~3
            self._pool[kind] = fn(need, device=self.device, generator=self.gen) if self._want[kind] else None
            self._off[kind], self._want[kind] = 0, 0

    def _take(self, kind, shape, fn):
        n = 1
        for d in shape:
            n *= int(d)
        self._want[kind] += n
        pool, off = self._pool[kind], self._off[kind]
        if pool is not None and off + n <= pool.shape[0]:
            self._off[kind] = (off + n + 3) & ~3          # every slice starts 16-byte aligned (kernels read float4 runs)
            return pool[off:off + n].view(shape)
        return fn(shape, device=self.device, generator=self.gen)

    # march jitter is generated inside the kernel (Philox keyed by seed/offset)
    def jitter(self, B, N):
        self.calls += 1
        return None, (self.seed, self.calls)

    def normal(self, shape):
        return self._take("n", tuple(shape), torch.randn)

    def normal_deferred(self, shape):
        """an i.i.d. normal [M, C] matrix of which only some rows will be used: nothing is drawn until rows() asks"""
        return tuple(shape)

    def rows(self, deferred, rows):
        return self._take("n", (rows.shape[0],) + tuple(deferred[1:]), torch.randn)

    def uniform(self, shape):
        return self._take("u", tuple(shape), torch.rand)

    def skip(self, kind, shape):
        return None

    def select_dense_parts(self, b, N, M):
        """U at the M kept entries of the dense [b,N] matrix + the SUM of U over the b*N - M dropped entries as a host
        float (modules/pt_selectors.py:24 perturbs the dense weight matrix).  The dropped entries only enter through their
        sum, which is drawn from its exact large-n normal limit."""
        u = self.uniform((M,))
        n_drop = b * N - M
        extra = 0.5 * n_drop + math.sqrt(max(n_drop, 0) / 12.0) * random.Random(self.calls * 1000003 + self.seed).gauss(0.0, 1.0)
        self.calls += 1
        return u, extra

    def select_dense(self, b, N, ray_id, step_id):
        """-> (u [M], sum of U over ALL b*N entries as a float64 device scalar)"""
        u, extra = self.select_dense_parts(b, N, ray_id.shape[0])
        return u, u.sum(dtype=torch.float64) + extra


class ReplayNoise:
    """tape: list of (kind, tensor) in the reference's call order, or None to draw from torch's global CPU RNG
    (seed it with torch.manual_seed(s) first) in that same order."""

    def __init__(self, device, tape=None, pins=None):
        self.device = torch.device(device)
        self.tape = list(tape) if tape is not None else None
        self.pos = 0
        self.pins = pins

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
