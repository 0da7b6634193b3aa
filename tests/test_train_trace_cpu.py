"""the host-side control logic of the training run -- dynamic ray batch (train.py:504-507,618-626), re-trace
controller (models/microfacet.py:236-269), per-group learning rates with the restart at the upsample (train.py:443-469,
806-813, utils.py:327-359) -- replayed against a run of the REFERENCE's own `reconstruction()` (40 iterations on a small
scene with a forced upsample at iteration 15; tests/golden/make_train_trace.py -> train_trace.npz).  The controllers are
fed the reference's recorded per-chunk statistics and must reproduce its decisions exactly."""
import numpy as np

from conftest import Golden
from nmf_amd.controllers import RayBatchController, RetraceController, learning_rate_decay


def _params(g):
    mn, mx, start, target = (int(v) for v in g.np("params_params"))
    return dict(min_batch_size=mn, max_batch_size=mx, starting_batch_size=start, target_num_samples=target)


def test_ray_batch_and_retrace_controllers_replay_the_reference_run():
    g = Golden("train_trace")
    ctl = RayBatchController(_params(g))
    rt = RetraceController([200], [40000], [40000, 20000])
    it_of, num_rays, rays_in = g.np("chunk_iter"), g.np("chunk_num_rays"), g.np("chunk_rays_in")
    kept, ns, mr = g.np("chunk_kept"), g.np("chunk_n_samples"), g.np("chunk_max_retrace")
    n_iters, up = int(g["n_iters"]), int(g["upsample_at"])
    c = 0
    for it in range(n_iters):
        lbatch = ctl.lbatch_size()
        assert lbatch == int(g.np("iter_lbatch")[it])
        remaining = lbatch
        while remaining > 0:                                            # train.py:508-512
            assert int(it_of[c]) == it
            assert ctl.num_rays == int(num_rays[c]), (it, c)
            n = min(ctl.num_rays, remaining)
            assert n == int(rays_in[c])
            remaining -= n
            assert rt.max_retrace_rays[0] == int(mr[c]), (it, c, rt.max_retrace_rays, int(mr[c]))
            if int(ns[c][0]) > 0:                                        # train.py:567-568
                ctl.update(int(kept[c]), int(ns[c][0]))
                rt.update([int(ns[c][1])])
            c += 1
        assert rt.max_retrace_rays[0] == int(g.np("iter_max_retrace")[it])
        if it == up:                                                     # check_schedule -> train.py:806-813
            ctl.reset()
            rt.reset()
    assert c == len(it_of)
    assert int(g.np("iter_grid")[up]) == int(g["grid0"]) and int(g.np("iter_grid")[up + 1]) == int(g["grid1"])
    opt = g.np("iter_optimizer")
    assert list(opt[: up + 1]) == [0] * (up + 1) and list(opt[up + 1:]) == [1] * (n_iters - up - 1)
    assert list(g.np("iter_detach_N")) == [True] + [False] * (n_iters - 1) or not g.np("iter_detach_N")[2:].any()


def test_learning_rates_follow_the_reference_run():
    g = Golden("train_trace")
    lrs = g.np("iter_lr")                                               # [iteration, param group]
    n_iters, up = int(g["n_iters"]), int(g["upsample_at"])
    # fields/tensoRF.py:298-313, models/microfacet.py:98-110, modules/integral_equirect.py:232-257 (group order of
    # TensorNeRF.get_optparam_groups, modules/tensor_nerf.py:105-118)
    base = [1e-3, 1e-3, 0.02, 0.02, 0.02, 0.02, 1e-3, 1e-3, 0.02, 0.0, 0.0, 1e-4]
    assert lrs.shape == (n_iters, len(base))
    step = 0
    for it in range(n_iters):
        f = learning_rate_decay(step, 1, 1e-3, n_iters, 100, 0.1)
        assert np.allclose(lrs[it], np.asarray(base) * f, rtol=1e-12, atol=0), (it, lrs[it], f)
        step += 1
        if it == up:
            step = 0                                                    # optimizer + LambdaLR re-created (train.py:806-809)


def test_simple_sampler_is_the_references_incl_its_overlapping_chunks():
    """train.py:34-51 (tests/golden/make_golden.py simple_sampler: the reference's own class, torch seeded): nmf_amd.controllers.
    SimpleSampler returns the same ids call by call -- the cursor moves BEFORE the slice is taken, so a chunk that is smaller than the
    one before it starts inside that chunk's range: with the steady chunk sizes of a run (471 + 471 + 82) the tail chunk of every
    iteration repeats 82 rays of the chunk before it.  Round 6: this is what separated the two sides' PSNR trajectories."""
    import os
    import torch
    from nmf_amd.controllers import SimpleSampler
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "simple_sampler.npz"))
    torch.manual_seed(int(g["seed"]))
    smp = SimpleSampler(int(g["total"]), int(g["batch"]), lambda n: torch.randperm(n, dtype=torch.long))
    got = [smp.nextids(int(b)) for b in g["sizes"]]
    assert torch.equal(torch.cat(got), torch.as_tensor(g["ids"])) and smp.curr == int(g["curr"])
    # the overlap, spelled out on the first full iteration (calls 3, 4, 5 = 471 + 471 + 82 rays): the 82-ray chunk lies inside the second
    a, b, c = got[3], got[4], got[5]
    assert len(set(a.tolist()) & set(b.tolist())) == 0
    assert set(c.tolist()) <= set(b.tolist()) and c.shape[0] == 82
