"""Training step of `model=microfacet_tensorf2` -- counterpart of the reference's train.py:443-469 (optimizer
and LambdaLR), :497-747 (loss assembly, dynamic ray batch, gradient accumulation, step) and :806-813 (schedule),
plus the data-parallel extension the reference lacks: every rank renders its own slice of the ray
batch and ONE RCCL all-reduce (sum) of the flat fp32 gradient precedes optimizer.step().
"""
import gc
import math
import os

import numpy as np
import torch
import torch.distributed as dist

from .fast_step import Unsupported
from .functional import LossMix, SquaredError
from .optim import FusedAdam

_CONST = {}


def _ones(shape, device):
    """cached constant tensors of ones (backward seed, white background): no fill launch per step"""
    key = (tuple(shape), device)
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = torch.ones(shape, dtype=torch.float32, device=device)
    return t


def _one(like):
    return _ones((), like.device)


def _zero_scalar(device):
    key = ("zero", device)
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = torch.zeros((), dtype=torch.float32, device=device)
    return t


from .controllers import RayBatchController, learning_rate_decay  # noqa: E402,F401


class DecayLR:
    """lr_g = initial_lr_g * f(step) for every param group: what torch.optim.lr_scheduler.LambdaLR does with one lambda
    (train.py:468-469), evaluated once per step instead of once per group and without LambdaLR's per-call bookkeeping
    (100 us per step for 11 groups).  Same constructor side effect (lr set to initial_lr * f(0)) and state layout."""

    def __init__(self, optimizer, lr_lambda):
        self.optimizer, self.lr_lambda, self.last_epoch = optimizer, lr_lambda, 0
        self.base_lrs = [g.setdefault("initial_lr", g["lr"]) for g in optimizer.param_groups]
        self._apply()

    def _apply(self):
        f = self.lr_lambda(self.last_epoch)
        for g, b in zip(self.optimizer.param_groups, self.base_lrs):
            g["lr"] = b * f

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def state_dict(self):
        return {"last_epoch": self.last_epoch, "base_lrs": list(self.base_lrs)}

    def load_state_dict(self, sd):
        self.last_epoch, self.base_lrs = int(sd["last_epoch"]), list(sd["base_lrs"])
        self._apply()


from .renderer import psnr_8bit  # noqa: E402,F401  (re-exported: renderer.py:399-401)


def rank_slice(n_total, world_size, rank):
    """Contiguous, disjoint, complete partition of a global ray batch over ranks (first ranks take the remainder)."""
    base, rem = divmod(n_total, world_size)
    start = rank * base + min(rank, rem)
    return slice(start, start + base + (1 if rank < rem else 0))


class FlatGradAllReduce:
    """Gradients are packed into flat fp32 buffers (14 MB at 128^3, 50 MB at 300^3), summed over ranks (RCCL over xGMI when
    backend='nccl', gloo on CPU) and unpacked.  TWO buckets since round 5 (at a 1.4 ms step a 0.1-0.6 ms collective on
    the serial tail is 7-40 % exposed): `early()` takes the gradients that are final before the field walks of the last chunk --
    BRDF MLP, material heads, environment map: 6.4 of the 14.3 MB -- and sums them on a communication stream NEXT TO those walks;
    `__call__` sums the field tables behind them.  Both buckets are entered exactly once per step by every rank (a rank whose last
    chunk did not run the fused pass enters the early one behind its chunks): the collectives of the ranks always pair up.

    The buffer layout is FIXED: [gradients | one has-gradient flag per parameter | the step's guard value].  Every parameter
    owns its slot whether or not this rank produced a gradient for it in this step (a rank whose chunk spawned no bounce
    rows, or whose chunks were all empty, still has to enter the collective with the same element count as its peers); a
    missing gradient travels as zeros.  After the sum, a parameter NO rank produced a gradient for keeps `grad = None` --
    as in a single-process step, where Adam then leaves its moments and step count alone (materialising zeros instead would
    move it on momentum at world > 1 and not at world = 1: two different trajectories).  The flags are only read back (one
    host synchronisation) by a rank that itself lacks a gradient; in the steady state every rank has them all.

    `guard`: 0-d device value that gates the optimizer step (the step's summed loss, train.py:704-705).  Its SUM over the
    ranks comes back, so every replica takes the same decision (a NaN on one rank is a NaN everywhere); None = 0."""

    def __init__(self, params):
        self.params = [p for p in params]
        self.numel = sum(p.numel() for p in self.params)
        self.buf = None
        self._slots = None
        self._consts = None
        self.last_comm_ms = None          # (start, end) device events of the last collective, or host seconds on CPU
        self.mask_reads = 0               # how often the has-gradient flags had to be read back (tests)
        # NMF_ALLREDUCE_SINGLE_RANK=1: enter the collective even when the group has ONE rank (a sum over one rank is the
        # identity): exercises RCCL, the pack / unpack launches and their ordering against the training pass's side streams
        # on a 1-GPU box, and gives a first comm_ms_per_step (bench.py with NMF_BENCH_BACKEND=nccl)
        self.single_rank = os.environ.get("NMF_ALLREDUCE_SINGLE_RANK") == "1"
        self._early = None                # the early bucket of the running step: dict(pairs, buf, work, stream, events)
        self._early_key = None
        self._late_key = None
        self.buf_early = None
        self._slots_early = None
        self._exposed = None              # (main stream ready, collectives done) events of the last step

    # ---- the early bucket -----------------------------------------------------------------------------------------------------
    def begin_step(self):
        self._early = None

    def early(self, pairs, comm_stream, group=None):
        """pairs [(parameter, gradient tensor)]: packed and all-reduced on `comm_stream` NOW (the stream already waits for the
        producers of the tensors); the sums are written back into the same tensors by finish_early().  A gradient some chunk left in
        .grad outside the accumulators (a chunk that went through the operator graph) is folded in first."""
        if self._early is not None or not self.active(group) or not pairs:
            return
        from . import hip
        main = torch.cuda.current_stream()
        torch.cuda.set_stream(comm_stream)
        try:
            for prm, g in pairs:
                if prm.grad is not None and prm.grad.data_ptr() != g.data_ptr():
                    g.add_(prm.grad.reshape(g.shape).to(g.dtype))
                    prm.grad = None
            key = tuple(g.data_ptr() for _, g in pairs)
            if self._early_key != key:          # (the accumulator tensors are persistent: the slot tables stand from step to step)
                n = sum(g.numel() for _, g in pairs)
                if self.buf_early is None or self.buf_early.numel() != n or self.buf_early.device != pairs[0][1].device:
                    self.buf_early = torch.empty(n, dtype=torch.float32, device=pairs[0][1].device)
                self._slots_early = ((hip.CopySlot * len(pairs))(), (hip.CopySlot * len(pairs))())
                base, off = self.buf_early.data_ptr(), 0
                for i, (_, g) in enumerate(pairs):
                    if not g.is_contiguous() or g.dtype != torch.float32:
                        raise hip.NmfHipError("early gradient bucket: dense fp32 tensors")
                    a, b = self._slots_early[0][i], self._slots_early[1][i]
                    a.src, a.dst, a.numel, a.src_is_f64, a.dst_is_f64 = g.data_ptr(), base + 4 * off, g.numel(), 0, 0
                    b.src, b.dst, b.numel, b.src_is_f64, b.dst_is_f64 = base + 4 * off, g.data_ptr(), g.numel(), 0, 0
                    off += g.numel()
                self._early_key, self._early_n = key, n
                self._early_ids = {id(p) for p, _ in pairs}
            n = self._early_n
            hip.multi_copy(self._slots_early[0], len(pairs))
            ev = self._early_events = getattr(self, "_early_events", None) or tuple(torch.cuda.Event(enable_timing=True) for _ in range(3))
            ev[0].record()
            work = dist.all_reduce(self.buf_early, op=dist.ReduceOp.SUM, group=group, async_op=True)
            self._early = dict(pairs=pairs, work=work, stream=comm_stream, ev=ev, bytes=4 * n, ids=self._early_ids)
        finally:
            torch.cuda.set_stream(main)

    def early_inplace(self, region, pairs, comm_stream, group=None):
        """the early bucket WITHOUT packing: `region` is the contiguous slice of the pass's flat accumulator that holds every early
        gradient (fast_step.TrainPass.comm_regions), summed in place on `comm_stream` (which already waits for its producers).
        pairs [(parameter, tensor inside region)]: a gradient a chunk left in .grad outside the accumulators is folded in first."""
        if self._early is not None or not self.active(group):
            return
        main = torch.cuda.current_stream()
        torch.cuda.set_stream(comm_stream)
        try:
            for prm, g in pairs:
                if prm.grad is not None and prm.grad.data_ptr() != g.data_ptr():
                    g.add_(prm.grad.reshape(g.shape).to(g.dtype))
                    prm.grad = None
            ev = self._early_events = getattr(self, "_early_events", None) or tuple(torch.cuda.Event(enable_timing=True) for _ in range(3))
            ev[0].record()
            work = dist.all_reduce(region, op=dist.ReduceOp.SUM, group=group, async_op=True)
            self._early = dict(inplace=True, work=work, stream=comm_stream, ev=ev, bytes=4 * region.numel())
        finally:
            torch.cuda.set_stream(main)

    def late_inplace(self, region, tail, has_grad, has_env, guard, group=None):
        """the late bucket in place: `region` = the field's gradients + `tail` = [has-gradient flag, has-env flag, guard, pad] (its
        last four floats).  -> (bytes exchanged, any rank had field gradients, any rank had env gradients); self.guard = the summed
        guard.  The flags are read back (a host synchronisation) only by a rank that lacks one of the two itself."""
        from . import hip
        self.finish_early()
        early = self._early
        if self._consts is None or self._consts.device != region.device:
            self._consts = torch.tensor([0.0, 1.0], dtype=torch.float32, device=region.device)
        zero, one = self._consts.data_ptr(), self._consts.data_ptr() + 4
        key = (tail.data_ptr(), bool(has_grad), bool(has_env), None if guard is None else guard.data_ptr())
        if self._late_key != key:
            self._tail_slots = (hip.CopySlot * 3)()
            srcs = (one if has_grad else zero, one if has_env else zero, zero if guard is None else guard.data_ptr())
            for i, src in enumerate(srcs):
                a = self._tail_slots[i]
                a.src, a.dst, a.numel, a.src_is_f64, a.dst_is_f64 = src, tail.data_ptr() + 4 * i, 1, \
                    (1 if (i == 2 and guard is not None and guard.dtype == torch.float64) else 0), 0
            self._late_key = key
        self._keep_guard = guard
        hip.multi_copy(self._tail_slots, 3)
        ev = self._events = getattr(self, "_events", None) or (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        dist.all_reduce(region, op=dist.ReduceOp.SUM, group=group)
        ev[1].record()
        self.last_comm_ms = ev
        if early is not None:
            self._exposed = (early["ready"], early["ev"][1], ev)
        any_grad, any_env = bool(has_grad), bool(has_env)
        if not (has_grad and has_env):
            self.mask_reads += 1
            f = tail[:2].tolist()
            any_grad, any_env = f[0] > 0.0, f[1] > 0.0
        self.guard = tail[2]
        self._early = None
        return 4 * region.numel() + (early["bytes"] if early is not None else 0), any_grad, any_env

    def finish_early(self):
        """the early sums land in the tensors they were packed from (the communication stream unpacks; the current stream then waits
        for it): call before anything reads those tensors -- end_step's conversion into .grad"""
        e = self._early
        if e is None or e.get("done"):
            return
        if e.get("inplace"):                   # nothing to unpack: the current stream waits for the collective itself
            ready = e["ev"][2]
            ready.record()
            e["work"].wait()
            e["ev"][1].record()
            e["done"], e["ready"] = True, ready
            return
        from . import hip
        main = torch.cuda.current_stream()
        ready = e["ev"][2]
        ready.record(main)                     # from here on the main stream would idle if the collective were not finished
        torch.cuda.set_stream(e["stream"])
        try:
            e["work"].wait()
            hip.multi_copy(self._slots_early[1], len(e["pairs"]))
            e["ev"][1].record()
        finally:
            torch.cuda.set_stream(main)
        main.wait_event(e["ev"][1])
        e["done"], e["ready"] = True, ready

    def active(self, group=None):
        return (dist.is_available() and dist.is_initialized() and self.numel > 0
                and (dist.get_world_size(group) > 1 or self.single_rank))

    def __call__(self, group=None, guard=None):
        """-> bytes of gradient exchanged (0: no collective).  With `guard`, self.guard is its sum over the ranks (a 0-d view
        of the buffer) after the call."""
        self.guard = guard
        if not self.active(group):
            return 0
        early = self._early
        if early is not None:
            return self._call_late(group, guard, early)
        ps, n = self.params, self.numel
        dev = ps[0].device
        total = n + len(ps) + 1
        if self.buf is None or self.buf.device != dev or self.buf.numel() != total:
            self.buf = torch.empty(total, dtype=torch.float32, device=dev)
        have = [p.grad is not None for p in ps]
        if self.buf.is_cuda:
            self._pack_device(ps, n, have, guard)
            ev = self._events = getattr(self, "_events", None) or (torch.cuda.Event(enable_timing=True),
                                                                   torch.cuda.Event(enable_timing=True))
            ev[0].record()
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=group)
            ev[1].record()
            self.last_comm_ms = ev
        else:                                     # host tensors (gloo tests of the sharding logic): plain torch copies
            off = 0
            for p, h in zip(ps, have):
                if h:
                    self.buf[off:off + p.numel()].copy_(p.grad.reshape(-1).float())
                else:
                    self.buf[off:off + p.numel()].zero_()
                off += p.numel()
            self.buf[n:n + len(ps)] = torch.tensor([1.0 if h else 0.0 for h in have])
            self.buf[total - 1] = 0.0 if guard is None else float(guard)
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=group)
        # parameters without a local gradient: did any rank produce one?
        anyone = have
        if not all(have):
            self.mask_reads += 1
            flags = self.buf[n:n + len(ps)].tolist()          # (device: waits for the collective)
            anyone = [f > 0.0 for f in flags]
        if self.buf.is_cuda:
            self._unpack_device(ps, n, have, anyone)
        else:
            off = 0
            for p, h, a in zip(ps, have, anyone):
                if a:
                    if not h:
                        p.grad = torch.zeros_like(p, memory_format=torch.preserve_format)
                    p.grad.copy_(self.buf[off:off + p.numel()].reshape(p.grad.shape).to(p.grad.dtype))
                off += p.numel()
        if guard is not None:
            self.guard = self.buf[total - 1]
        return n * 4

    def _call_late(self, group, guard, early):
        """the second bucket of a step whose early bucket is under way: [gradients of the remaining parameters | one has-gradient flag
        per parameter (all of them) | guard].  An early parameter has a gradient on a rank iff .grad is set there (end_step sets it
        from the accumulator tensor the early sum was written into); where no rank had one it stays None."""
        from . import hip
        self.finish_early()
        ps = self.params
        late = [p for p in ps if id(p) not in early["ids"]]
        n = sum(p.numel() for p in late)
        dev = ps[0].device
        total = n + len(ps) + 1
        if self.buf is None or self.buf.device != dev or self.buf.numel() != total:
            self.buf = torch.empty(total, dtype=torch.float32, device=dev)
        have_all = [p.grad is not None for p in ps]
        have = [p.grad is not None for p in late]
        # the fused pass hands over the SAME gradient tensors every step: the slot tables of pack and unpack stand until a pointer moves
        key = (tuple(p.grad.data_ptr() if p.grad is not None else 0 for p in late), tuple(have_all),
               None if guard is None else guard.data_ptr(), self.buf.data_ptr())
        planned = all(have_all) and key == self._late_key
        if planned:
            hip.multi_copy(self._slots[0], self._late_counts[0])
        else:
            self._pack_device(late, n, have, guard, flags=have_all)
            self._late_key = None
        ev = self._events = getattr(self, "_events", None) or (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=group)
        ev[1].record()
        self.last_comm_ms = ev
        self._exposed = (early["ready"], early["ev"][1], ev)
        anyone_all = have_all
        if not all(have_all):
            self.mask_reads += 1
            anyone_all = [f > 0.0 for f in self.buf[n:n + len(ps)].tolist()]
        by_early = {id(p): g for p, g in early["pairs"]}
        anyone = []
        for p, h, a in zip(ps, have_all, anyone_all):
            if id(p) in by_early:
                if a and not h:                # another rank produced it: the sum is in the early tensor
                    g = by_early[id(p)]
                    p.grad = g.reshape(p.shape).to(p.dtype) if (g.shape != p.shape or g.dtype != p.dtype) else g
            else:
                anyone.append(a)
        if planned:
            hip.multi_copy(self._slots[1], self._late_counts[1])
        else:
            self._unpack_device(late, n, have, anyone)
            if all(have_all) and guard is not None:
                self._late_key, self._late_counts = key, (self._pack_count, self._unpack_count)
        if guard is not None:
            self.guard = self.buf[total - 1]
        self._early = None
        return n * 4 + early["bytes"]

    def exposed_ms(self):
        """time the main stream waited for the step's collectives: the early bucket's remainder behind the walks + the late bucket
        (blocks until they have finished); None without an overlapped step"""
        x = self._exposed
        if x is None:
            return None
        ready, early_done, late = x
        late[1].synchronize()
        return max(ready.elapsed_time(early_done), 0.0) + late[0].elapsed_time(late[1])

    def _pack_device(self, ps, n, have, guard, flags=None):
        """one launch: nmf_multi_copy over all gradients (each in its own memory order), the flags and the guard"""
        from . import hip
        from .optim import _dense
        k = len(ps)
        nf = len(flags) if flags is not None else k
        if self._slots is None or len(self._slots[0]) < k + nf + 1:
            self._slots = ((hip.CopySlot * (len(self.params) * 2 + 1))(), (hip.CopySlot * len(self.params))())
        if self._consts is None or self._consts.device != self.buf.device:
            self._consts = torch.tensor([0.0, 1.0], dtype=torch.float32, device=self.buf.device)
        pack = self._slots[0]
        base, off, m = self.buf.data_ptr(), 0, 0
        zero, one = self._consts.data_ptr(), self._consts.data_ptr() + 4
        missing = []
        for i, (p, h) in enumerate(zip(ps, have)):
            if h:
                g = p.grad
                if not _dense(g) or g.dtype not in (torch.float32, torch.float64):
                    raise hip.NmfHipError("gradient all-reduce needs dense fp32 / fp64 gradients")
                a = pack[m]
                a.src, a.dst, a.numel, a.src_is_f64, a.dst_is_f64 = g.data_ptr(), base + 4 * off, g.numel(), \
                    (1 if g.dtype == torch.float64 else 0), 0
                m += 1
            else:
                missing.append((off, p.numel()))
            if flags is None:
                a = pack[m]
                a.src, a.dst, a.numel, a.src_is_f64, a.dst_is_f64 = (one if h else zero), base + 4 * (n + i), 1, 0, 0
                m += 1
            off += p.numel()
        if flags is not None:                    # flags of ALL parameters behind the gradients of the packed ones
            for i, h in enumerate(flags):
                a = pack[m]
                a.src, a.dst, a.numel, a.src_is_f64, a.dst_is_f64 = (one if h else zero), base + 4 * (n + i), 1, 0, 0
                m += 1
        k = nf
        if guard is not None:
            if guard.dtype not in (torch.float32, torch.float64) or guard.numel() != 1:
                raise hip.NmfHipError("the guard is a single fp32 / fp64 device value")
            a = pack[m]
            a.src, a.dst, a.numel, a.src_is_f64, a.dst_is_f64 = guard.data_ptr(), base + 4 * (n + k), 1, \
                (1 if guard.dtype == torch.float64 else 0), 0
            m += 1
            self._keep_guard = guard
        else:
            self.buf[n + k:].zero_()
        for off_, cnt in missing:                # slots of gradients this rank does not have travel as zeros
            self.buf[off_:off_ + cnt].zero_()
        hip.multi_copy(pack, m)
        self._pack_count = m

    def _unpack_device(self, ps, n, have, anyone):
        from . import hip
        unpack = self._slots[1]
        base, off, m = self.buf.data_ptr(), 0, 0
        for p, h, a in zip(ps, have, anyone):
            if a:
                if not h:
                    p.grad = torch.empty_like(p, memory_format=torch.preserve_format)
                g = p.grad
                b = unpack[m]
                b.src, b.dst, b.numel, b.src_is_f64, b.dst_is_f64 = base + 4 * off, g.data_ptr(), g.numel(), 0, \
                    (1 if g.dtype == torch.float64 else 0)
                m += 1
            off += p.numel()
        if m:
            hip.multi_copy(unpack, m)
        self._unpack_count = m

    def comm_ms(self):
        """duration of the last device collective (blocks until it has finished); None if there was none"""
        ev = self.last_comm_ms
        if ev is None:
            return None
        ev[1].synchronize()
        return ev[0].elapsed_time(ev[1])


def agree(value, op="min", group=None, device=None):
    """One scalar agreed over the ranks (min / max / sum of a python number): controller inputs that must be identical on
    every replica (global batch size, the decision to re-permute).  Identity without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return value
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op={"min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX, "sum": dist.ReduceOp.SUM}[op], group=group)
    v = float(t[0])
    return int(round(v)) if isinstance(value, int) else v


# ---- data parallel: identical replicas at start-up, and a check that they stay identical -------------------------------------------
_CALIBRATED = (("model.brdf", "bias"), ("model.diffuse_module", "diffuse_bias"), ("model.diffuse_module", "roughness_bias"))


def _attr_path(obj, path):
    for k in path.split("."):
        obj = getattr(obj, k)
    return obj


def _replica_group(group):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def broadcast_replica(nerf, src=0, group=None):
    """after construction and calibration every rank takes rank `src`'s replica -- every parameter and buffer of the
    module (field tables, BRDF MLP, heads, env map, the Sobol table `angs`, the alpha volume) and the three CALIBRATED biases, which
    are Python floats computed from random points (models/microfacet.py:79-96, train.py:429-437) and therefore the one piece of
    state that "same seed on every rank" does not pin across devices.  A buffer whose shape differs on a rank (an alpha volume
    rebuilt at another size) is an error, not something to paper over.  -> number of bytes broadcast (0 without a process group)."""
    if not _replica_group(group):
        return 0
    import zlib
    dev = nerf.get_device()
    n = 0
    with torch.no_grad():
        names = sorted(nerf.state_dict().keys())        # the same tensors on every rank, or the broadcasts below would not pair up
        sig = torch.tensor([len(names), zlib.crc32("\n".join(names).encode())], dtype=torch.int64, device=dev)
        ref = torch.cat([sig, -sig])                    # (max, -min) in ONE collective: EVERY rank learns of a mismatch and raises
        dist.all_reduce(ref, op=dist.ReduceOp.MAX, group=group)
        if ref[:2].tolist() != (-ref[2:]).tolist():
            raise RuntimeError(f"broadcast_replica: the ranks' modules do not hold the same set of state tensors ({len(names)} here; "
                               "an alpha mask built on one rank only?)")
        for name, t in sorted(nerf.state_dict().items()):
            shape = torch.tensor(list(t.shape) + [-1] * (8 - t.dim()), dtype=torch.int64, device=dev)
            ref = shape.clone()
            dist.broadcast(ref, src=src, group=group)
            if not torch.equal(ref.cpu(), shape.cpu()):
                raise RuntimeError(f"broadcast_replica: {name} has shape {tuple(t.shape)} here and {ref.tolist()} on rank {src}")
            if t.numel() == 0:
                continue
            buf = t.detach().clone().contiguous()
            dist.broadcast(buf, src=src, group=group)
            if dist.get_rank(group) != src:
                t.detach().copy_(buf)          # (copy_ moves the version counter: caches keyed on it -- packed alpha bits, derived tables -- rebuild)
            n += t.numel() * t.element_size()
        bias = torch.tensor([float(getattr(_attr_path(nerf, m), k)) for m, k in _CALIBRATED], dtype=torch.float64, device=dev)
        dist.broadcast(bias, src=src, group=group)
        for (m, k), v in zip(_CALIBRATED, bias.tolist()):
            setattr(_attr_path(nerf, m), k, v)
    if hasattr(nerf.sampler, "update"):          # derived sampler state (bit-packed alpha mask, step size) follows the broadcast volume
        nerf.sampler.update(nerf.rf, init=True)
    return n + 24


def replica_checksum(nerf):
    """-> float64 device tensor [2]: (sum of the fp64 sums of every parameter, buffer and calibrated bias; the same with position-dependent weights, so that two
    replicas whose differences cancel in the plain sum still differ).  Deterministic on equal bits: equal replicas give equal
    numbers, and the replicas of a data-parallel run ARE equal bit for bit (the same summed gradient into the same Adam)."""
    dev = nerf.get_device()
    parts = []
    with torch.no_grad():
        for i, (name, t) in enumerate(sorted(nerf.state_dict().items())):
            if t.numel() == 0 or not (t.is_floating_point() or t.dtype in (torch.int32, torch.int64, torch.uint8, torch.bool)):
                continue
            s = t.detach().to(torch.float64).sum()
            parts.append(torch.stack([s, s * (1.0 + 0.001 * (i + 1))]))
        b = torch.tensor([float(getattr(_attr_path(nerf, m), k)) for m, k in _CALIBRATED], dtype=torch.float64, device=dev)
        parts.append(torch.stack([b.sum(), (b * torch.tensor([3.0, 5.0, 7.0], dtype=torch.float64, device=dev)).sum()]))
    return torch.stack(parts).sum(0)


class ReplicaDivergence(RuntimeError):
    pass


def check_replicas(nerf, group=None, what=""):
    """all-reduce (min, max) of replica_checksum: any difference between the ranks' replicas raises ReplicaDivergence on EVERY rank
    (one collective of four doubles + one host read-back: run every K steps, Trainer(check_every=K)).  -> the checksum pair."""
    cs = replica_checksum(nerf)
    if not _replica_group(group):
        return cs.tolist()
    v = torch.cat([cs, -cs])                   # max of (x, -x) = (max, -min): ONE collective
    dist.all_reduce(v, op=dist.ReduceOp.MAX, group=group)
    hi, lo = v[:2].tolist(), (-v[2:]).tolist()
    if hi != lo:
        raise ReplicaDivergence(f"data-parallel replicas differ{' ' + what if what else ''}: checksum min {lo} max {hi} "
                                f"(this rank {cs.tolist()}); parameters, buffers or calibrated biases are no longer identical")
    return hi


class Trainer:
    def __init__(self, nerf, params, world_size=1, rank=0, tape_free=True, check_every=None):
        self.nerf = nerf
        self.p = params
        self.world_size, self.rank = world_size, rank
        # The loss of microfacet_tensorf2.yaml:196-232: photometric + density_L1 + orientation + accumulated opacity.  The terms that
        # file gives weight 0 are not assembled here -- a run that switches one on must not train silently without it.
        on = [k for k in ("TV_weight_density", "TV_weight_app", "TV_weight_bg", "envmap_lambda", "diffuse_lambda", "brdf_lambda",
                          "normal_err_lambda", "distortion_lambda", "visibility_lambda", "ortho_weight") if params.get(k)]
        if on or params.get("charbonier_loss"):
            raise NotImplementedError(f"Trainer.step does not assemble the loss terms {on or ['charbonier_loss']} (weight 0 in "
                                      "microfacet_tensorf2.yaml); the reference-style loop over TensorNeRF.forward with "
                                      "nerf.regulariser_stats = True evaluates envmap_reg / brdf_reg / diffuse_reg")
        self.batch = RayBatchController(params)          # train.py:504-507,618-626 (num_rays / lbatch_size)
        self.iteration = 0
        self.reduce = None
        # data parallel: every `check_every` steps (and before the first one) the replicas' checksums are compared over the ranks and a
        # mismatch raises ReplicaDivergence everywhere.  Default 500 (NMF_REPLICA_CHECK_EVERY; 0: never): 30 000 iterations = 60
        # checks of ~35 small launches + one read-back each.
        self.check_every = int(os.environ.get("NMF_REPLICA_CHECK_EVERY", "500")) if check_every is None else int(check_every)
        self.replica_checks = 0
        # train.py:470-481,748-749: exponential decay of the two regulariser weights towards their final values
        self.ori_lambda, self.pred_lambda = float(params["ori_lambda"]), float(params["pred_lambda"])
        n_it = params["n_iters"]
        fo, fp = params.get("final_ori_lambda"), params.get("final_pred_lambda")
        self.ori_decay = math.exp(math.log(fo / self.ori_lambda) / n_it) if self.ori_lambda > 0 and fo is not None else 1.0
        self.pred_decay = math.exp(math.log(fp / self.pred_lambda) / n_it) if self.pred_lambda > 0 and fp is not None else 1.0
        self._make_optimizer()
        # tape-free training pass (nmf_amd/fast_step.py over csrc/step_core.inc): forward + loss head + backward of a chunk in one C++
        # call.  tape_free=False: every chunk goes through TensorNeRF.forward + backward() instead -- the loop of the reference's
        # train.py, which enters the same C++ pass through ONE autograd node per chunk (or, with nerf.fused_training_pass = False, the
        # operator graph of nmf_amd/functional.py)
        self.fast = None
        if tape_free:
            from .fast_step import TrainPass
            self.fast = TrainPass(nerf)
        # The step allocates a few hundred short-lived Python containers; a full (generation-2) collection walks every
        # tracked object of the process (~270 k after importing torch: 70 ms measured, i.e. 13 steps).  Park what exists now
        # in the permanent generation so collections only look at what the steps create.
        gc.collect()
        gc.freeze()

    def _make_optimizer(self):
        # train.py:443-469: Adam over the per-module param groups, LambdaLR(learning_rate_decay) from step 0 (DecayLR)
        p = self.p
        groups = self.nerf.get_optparam_groups()
        self.optimizer = FusedAdam(groups, betas=tuple(p["betas"]), eps=p["eps"], weight_decay=p["weight_decay"])
        lam = lambda s: float(learning_rate_decay(s, p["lr_init"], p["lr_final"], p["n_iters"], p["lr_delay_steps"],  # noqa: E731
                                                  p["lr_delay_mult"]))
        self.scheduler = DecayLR(self.optimizer, lam)
        # The all-reduce covers the parameters that can move: a group with learning rate 0 (env-map brightness / mul,
        # microfacet_tensorf2.yaml:150-151) stays where it is whatever its gradient, and dbasis_mat is not part of the model with
        # dbasis = False (fields/tensoRF.py:117).  They never have a gradient on any rank -- and a parameter WITHOUT a local gradient
        # makes the reducer read the ranks' has-gradient flags back, a host synchronisation per step (rounds 3-4 paid it every step:
        # +0.39 ms at one rank).
        dead = {id(q) for q in getattr(self.nerf.rf, "dbasis_mat", torch.nn.Identity()).parameters()} \
            if not getattr(self.nerf.rf, "dbasis", False) else set()
        self.reduce = FlatGradAllReduce([q for g in self.optimizer.param_groups if g.get("initial_lr", g["lr"]) != 0
                                         for q in g["params"] if id(q) not in dead])

    @property
    def num_rays(self):
        return self.batch.num_rays

    @num_rays.setter
    def num_rays(self, v):
        self.batch.num_rays = v

    def lbatch_size(self):
        return self.batch.lbatch_size()

    def step(self, rays, rgb_gt, focal, noise=None, update_controllers=True, fixed_chunk=None, global_rays=None,
             fetch=None, trace=None):
        """One optimizer step over this rank's rays (train.py:497-747).  rays [n,6], rgb_gt [n,3] (already blended
        onto the background colour, train.py:525-530).  `global_rays`: the loss normaliser `lbatch_size` of train.py:703,
        i.e. the number of rays ALL ranks process in this step (default: n * world_size, equal shards).
        `fetch(n) -> (rays [n,6], rgb [n,3])` instead of rays / rgb_gt: the step pulls its chunks one by one exactly like
        train.py:509-512 (`trainingSampler.nextids(lnum_rays)` per chunk, so a re-permutation can fall inside a step);
        `n` = this rank's share of the step (rays.shape[0] when rays is given, else lbatch_size()).
        `trace`: list that receives one record per chunk (num_rays, rays in / kept, n_samples, loss, max_retrace_rays).
        Returns a stats dict (python scalars)."""
        p = self.p
        nerf = self.nerf
        if self.world_size > 1 and self.check_every > 0 and self.iteration % self.check_every == 0 and _replica_group(None):
            check_replicas(nerf, what=f"before iteration {self.iteration}")
            self.replica_checks += 1
        self.optimizer.zero_grad(set_to_none=True)
        n_total = rays.shape[0] if rays is not None else self.lbatch_size()
        lbatch = global_rays if global_rays is not None else n_total * self.world_size
        pos, used_rays, losses, n_samples_last, n_chunks = 0, 0, [], None, 0
        went_graph = False
        bg = None
        fast = self.fast if (self.fast is not None and self.fast.supported()) else None
        if fast is not None:
            fast.begin_step()
        # data parallel: the gradients no field walk writes (BRDF MLP, material heads, environment map) are summed over the ranks on
        # a communication stream from inside the last chunk's backward, next to the walks that end it (FlatGradAllReduce.early)
        self.reduce.begin_step()
        early = None
        if fast is not None and self.reduce.active() and rays is not None and rays.is_cuda:
            comm = fast._side.get("comm")
            if comm is None:
                comm = fast._side["comm"] = torch.cuda.Stream()
            dev_ = rays.device
            def start_early(behind_chunks=False):
                # What the collective reads must be complete and ordered in front of it: the zero fill of accumulators opened here (a rank
                # whose chunks never reached the backward) and the env-map table gradient when the last chunk's backward did not queue
                # it (that chunk kept no sample / left the fused pass / NMF_OVERLAP=0) while earlier chunks looked the map up -- both are
                # queued on the current stream by prepare_early, and the communication stream then waits for it.
                if fast.prepare_early(dev_, behind_chunks) or behind_chunks:
                    comm.wait_stream(torch.cuda.current_stream())
                pairs = fast.early_pairs(dev_)
                self.reduce.early_inplace(fast.comm_regions()[0], pairs, comm)
            early = (start_early, comm.cuda_stream)
        while pos < n_total:
            chunk = fixed_chunk if fixed_chunk is not None else max(int(self.num_rays), 1)
            if fetch is not None:
                r, gt = fetch(min(chunk, n_total - pos))
            else:
                r = rays[pos:pos + chunk]
                gt = rgb_gt[pos:pos + chunk]
            if bg is None:
                bg = _ones((3,), r.device)
            pos += r.shape[0]
            n_chunks += 1
            if trace is not None:
                trace.append(dict(num_rays=chunk, rays_in=int(r.shape[0]), max_retrace=list(nerf.model.max_retrace_rays)))
            if fast is not None:
                try:
                    out = fast.chunk(r, gt, focal, noise, 1.0 / lbatch,
                                     (1.0, p["L1_weight_initial"], self.ori_lambda, 2.0 * self.pred_lambda),
                                     want_total=trace is not None, last=pos >= n_total, early=early,
                                     ctx=(n_chunks - 1) % fast.n_contexts)      # chunk k + 1's forward next to chunk k's backward
                except Unsupported:
                    out = None                      # this chunk goes through the autograd path below
                if out is not None:
                    n_samples = out["n_samples"]
                    if trace is not None:
                        trace[-1].update(n_samples=list(n_samples), kept=int(out["kept"]))
                    if out["loss"] is None:
                        continue
                    if trace is not None:
                        trace[-1]["total"] = out["total"]
                    used_rays += out["kept"]
                    losses.append(out["loss"])
                    n_samples_last = n_samples
                    if update_controllers:
                        self.batch.update(out["kept"], n_samples[0])
                        nerf.model.update_n_samples(n_samples[1:])
                    continue
            went_graph = True                        # this chunk's gradients land in .grad outside the pass's accumulators
            ims, st = nerf(r, focal, bg_col=bg, is_train=True, ndc_ray=False, noise=noise)
            n_samples = st["n_samples"]
            if trace is not None:
                trace[-1].update(n_samples=list(n_samples), kept=int(ims["rgb_map"].shape[0]))
            if n_samples[0] == 0:
                continue
            rgb_map = ims["rgb_map"]                 # valid rays are a prefix of the chunk (alphagrid.py:353-364)
            loss = SquaredError.apply(rgb_map, gt[: rgb_map.shape[0]])                           # train.py:598-601
            l1 = nerf.rf.density_L1(with_pass=True) if hasattr(nerf.rf, "flush_pending_l1") else nerf.rf.density_L1()
            terms, wts = [loss, l1], [1.0, p["L1_weight_initial"]]                              # train.py:640-677
            if st.get("ori_terms") is not None:
                terms.append(st["ori_terms"]); wts.append(self.ori_lambda)
            if "acc_terms" in st:
                terms.append(st["acc_terms"]); wts.append(2.0 * self.pred_lambda)
            else:
                terms.append(st["prediction_loss"]); wts.append(self.pred_lambda)
            total = LossMix.apply(1.0 / lbatch, wts, *terms)
            # train.py:704-705 skips a chunk whose loss is NaN (a host read-back per chunk); here the chunk is
            # back-propagated regardless and the optimizer launch is gated by the step's summed loss on the device (below)
            total.backward(_one(total))
            if trace is not None:
                trace[-1]["total"] = total.detach()
            if hasattr(nerf.rf, "flush_pending_l1"):
                nerf.rf.flush_pending_l1()
            kept = rgb_map.shape[0]                  # = number of valid rays (no device read-back)
            used_rays += kept
            losses.append(loss.detach())             # read back after the optimizer step has been queued
            n_samples_last = n_samples
            if update_controllers:                                                               # train.py:618-627
                self.batch.update(kept, n_samples[0])
                nerf.model.update_n_samples(n_samples[1:])
        if early is not None:
            if self.reduce._early is None:        # the last chunk did not run the fused backward: the early bucket behind the chunks
                early[0](True)
            self.reduce.finish_early()            # the sums are in the accumulator tensors before end_step turns them into .grad
        if fast is not None:
            fast.end_step()
        # NaN guard (train.py:704-705 reads the loss back and skips a NaN chunk): the summed loss of the step stays on the
        # device and gates the fused Adam launch -- a non-finite loss leaves parameters and moments untouched, no host sync.
        # With several ranks the value that gates is the SUM of the ranks' guards (it rides in the gradient all-reduce): a
        # rank-local decision would let one replica skip a step its peers take, and nothing re-synchronises parameters.
        guard = None if not losses else (losses[0] if len(losses) == 1 else torch.stack(losses).sum())
        if self.reduce.active() and guard is None and len(self.reduce.params):
            guard = _zero_scalar(self.reduce.params[0].device)          # this rank's chunks were all empty: a finite contribution
        if early is not None:
            # the fused pass keeps every gradient in two contiguous regions of its flat buffer: summed in place, nothing is packed
            if went_graph:
                fast.fold_foreign()
            _e, late, tail = fast.comm_regions()
            has_grad = any(q.grad is not None for q in self.reduce.params)
            has_env = nerf.bg_module.bg_mat.grad is not None
            comm_bytes, any_grad, any_env = self.reduce.late_inplace(late, tail, has_grad, has_env, guard)
            if (any_grad and not has_grad) or (any_env and not has_env):
                fast.assign_reduced(any_grad and not has_grad, any_env and not has_env)
        else:
            comm_bytes = self.reduce(guard=guard)
        if comm_bytes:
            guard = self.reduce.guard
        if p.get("clip_grad") is not None:                                                       # train.py:744-745
            torch.nn.utils.clip_grad_norm_([q for q in nerf.parameters() if q.grad is not None], p["clip_grad"])
        if hasattr(self.optimizer, "guard"):
            self.optimizer.guard = guard
        (getattr(self.optimizer, "step_unhooked", None) or self.optimizer.step)()
        self.scheduler.step()
        self.ori_lambda *= self.ori_decay                                                        # train.py:748-749
        self.pred_lambda *= self.pred_decay
        if nerf.check_schedule(self.iteration, 1):                                               # train.py:806-813
            self._make_optimizer()
            self.batch.reset()
            nerf.model.reset_counter()
        self.iteration += 1
        pre = fast if fast is not None else getattr(nerf, "_fused_pass", None)
        if pre is not None:
            pre.prefetch()            # next step's derived tables, on a side stream behind this optimizer update
        return StepStats(losses, rays=used_rays, n_samples=n_samples_last, comm_bytes=comm_bytes, chunks=n_chunks,
                         reduce=self.reduce)


class StepStats(dict):
    """What Trainer.step returns: rays / n_samples / comm_bytes are host values; `loss` (sum of squared errors over the
    step's rays) and `psnr` (train-PSNR proxy, train.py:609-613) live on the device until first read, so a training loop that
    does not log every step never waits for the GPU at the end of a step."""

    def __init__(self, losses, **kw):
        super().__init__(**kw)
        self._losses = losses

    def __missing__(self, key):
        if key == "comm_ms":             # duration of this step's (late) gradient all-reduce (device events; waits for it)
            self[key] = self["reduce"].comm_ms() if self.get("reduce") is not None and self["comm_bytes"] else None
            return self[key]
        if key == "comm_exposed_ms":     # time the main stream waited for the step's collectives (early remainder + late bucket)
            self[key] = self["reduce"].exposed_ms() if self.get("reduce") is not None and self["comm_bytes"] else None
            return self[key]
        if key not in ("loss", "psnr"):
            raise KeyError(key)
        vals = torch.stack(self._losses).tolist() if self._losses else []
        good = [v for v in vals if math.isfinite(v)]           # a chunk whose loss is not finite is left out of the statistics
        self["nan_chunks"] = len(vals) - len(good)
        loss_sum = float(sum(good))
        self["loss"] = loss_sum
        self["psnr"] = -10.0 * math.log10(max(loss_sum / max(self["rays"] * 3, 1), 1e-12))
        return self[key]
