"""The three host-side feedback loops of the reference's training run, as plain Python (no device, no kernels) so that they
can be replayed against a recorded reference run (tests/golden/train_trace.npz, tests/test_train_trace_cpu.py):

  RayBatchController   train.py:504-507,618-626,813   rays per chunk from the kept-rays / primary-samples ratio
  RetraceController    models/microfacet.py:236-269    how many secondary rays are re-traced, from rays / secondary samples
  learning_rate_decay  utils.py:327-359               log-linear decay with a sine warm-up, restarted at every upsample
  SimpleSampler        train.py:34-51                 which rays a chunk gets (the cursor moves BEFORE the slice is taken)
"""
import math


def learning_rate_decay(step, lr_init, lr_final, max_steps, lr_delay_steps=0, lr_delay_mult=1):
    # utils.py:327-359 (float64 arithmetic like the reference's numpy scalars, without numpy's per-call overhead)
    if lr_delay_steps > 0:
        delay_rate = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
    else:
        delay_rate = 1.0
    t = min(max(step / max_steps, 0.0), 1.0)
    return delay_rate * math.exp(t * (math.log(lr_final) - math.log(lr_init)) + math.log(lr_init))


class RayBatchController:
    """num_rays: rays per forward/backward chunk, steered so that a chunk holds ~target_num_samples primary samples."""

    def __init__(self, params):
        self.min_batch, self.max_batch = params["min_batch_size"], params["max_batch_size"]
        self.start, self.target = params["starting_batch_size"], params["target_num_samples"]
        self.reset()

    def reset(self):                                   # train.py:486-487,811-813
        self.num_rays = self.start
        self.prev_ratio = None

    def lbatch_size(self):                             # train.py:504-507
        return min(self.min_batch if self.num_rays < self.min_batch else self.num_rays, self.max_batch)

    def update(self, kept, n_primary):                 # train.py:618-626
        ratio = kept / n_primary
        mean_ratio = ratio if self.prev_ratio is None else min(0.1 * ratio + 0.9 * self.prev_ratio, ratio)
        self.prev_ratio = mean_ratio
        self.num_rays = int(mean_ratio * self.target + 1)
        return self.num_rays


class RetraceController:
    """max_retrace_rays per recursion level: the minimum over the last 20 chunks of (rays re-traced / secondary samples they
    produced) times the sample target, capped by max_brdf_rays (1000 -> all secondary rays within ~20 chunks)."""

    def __init__(self, max_retrace_rays, target_num_samples, max_brdf_rays):
        self.start = list(max_retrace_rays)
        self.target, self.max_brdf_rays = list(target_num_samples), list(max_brdf_rays)
        self.reset()

    def reset(self):                                   # models/microfacet.py:236-239
        self.max_retrace_rays = list(self.start)
        self.mean_ratios = None
        self.ratio_list = None

    def update(self, n_samples):                       # models/microfacet.py:241-269
        if len(n_samples) != len(self.max_retrace_rays):
            return self.max_retrace_rays
        ratios = [(n_rays / n) if n > 0 else 1e-3 for n_rays, n in zip(self.max_retrace_rays, n_samples)]
        if self.ratio_list is None:
            self.ratio_list = [[r, 1e-3] for r in ratios]
        else:
            self.ratio_list = [([ratio] + rl)[:20] for ratio, rl in zip(ratios, self.ratio_list)]
        self.mean_ratios = [min(rl) if len(rl) > 0 else None for rl in self.ratio_list]
        self.max_retrace_rays = [min(int(t * r + 1), mx) if r is not None else prev for t, r, mx, prev in
                                 zip(self.target, self.mean_ratios, self.max_brdf_rays[:-1], self.max_retrace_rays)]
        return self.max_retrace_rays


class SimpleSampler:
    """train.py:34-51, statement for statement: the ray ids of the next `batch` rays of a permutation of `total` rays that is redrawn
    when it runs out.  The reference advances its cursor BEFORE it takes the slice -- `curr += batch; ids[curr : curr + batch]` -- and
    calls nextids once per CHUNK with that chunk's size (train.py:509-512), so that consecutive chunks of different sizes overlap: a
    chunk of b2 rays behind one of b1 > b2 rays starts b1 - b2 rays INSIDE the previous chunk's range.  With the steady chunk sizes of a
    run (e.g. 471 + 471 + 82 of a 1024-ray batch) the small tail chunk of every iteration repeats rays of the chunk before it, and the
    chunk after a small one skips rays: 8 % of a batch are duplicates.  Kept as it is -- it decides which rays a training sees, i.e. the
    PSNR after equal iterations.  `randperm(n) -> permutation` is the caller's generator (device or CPU)."""

    def __init__(self, total, batch, randperm):
        self.total, self.batch, self.randperm = total, batch, randperm
        self.curr = total
        self.ids = None

    def nextids(self, batch=None):
        batch = self.batch if batch is None else batch
        self.curr += batch
        if self.curr + batch > self.total:
            self.ids = self.randperm(self.total)
            self.curr = 0
        return self.ids[self.curr:self.curr + batch]
