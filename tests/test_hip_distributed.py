"""Data parallelism of the training step on hardware, as far as a 1-GPU box allows: two ranks share cuda:0 and
exchange through gloo (production: one rank per GPU, backend nccl = RCCL over xGMI; same Trainer / FlatGradAllReduce code).

The reference has no data-parallel path; the statement checked here is the one that makes the sharded step THE SAME
estimator as the single-process step of train.py:497-747: an optimizer step over 2 x 4096 rays as two chunks of one process
accumulates the gradient that two ranks with one chunk each obtain from their all-reduce, given the same noise per chunk
(a split of the rays alone is not comparable: bounce budgets and normalisers are per chunk, models/microfacet.py:318-331)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

GRID, BG, CHUNK = 48, 64, 2048


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _PerChunkNoise:
    """chunk k of a step draws from source k: the single process replays what rank k draws for its one chunk"""

    pins = None

    def __init__(self, sources):
        self.sources, self.k = sources, -1

    def begin_pass(self):
        self.k = (self.k + 1) % len(self.sources)
        self.sources[self.k].begin_pass()

    def __getattr__(self, name):
        return getattr(self.sources[self.k], name)


def _build(dev):
    from nmf_amd import synthetic
    from nmf_amd.config import build_model, resolved_config
    torch.manual_seed(0)
    over = {"sampler.max_samples": 60000, "model.max_brdf_rays": [120000, 80000], "model.rays_per_ray": 32}
    nerf, _ = build_model(grid=GRID, bg_resolution=BG, device=dev, overrides=over)
    nerf.load_state_dict(synthetic.state_dict_s1(grid=GRID, bg_resolution=BG, seed=0), strict=False)
    nerf.train()
    nerf.sampler.update(nerf.rf, init=False)
    nerf.sampler.update(nerf.rf, init=True)
    nerf.model.detach_N = False
    nerf.model.max_retrace_rays = [nerf.model.max_brdf_rays[0]]
    return nerf, resolved_config()["params"]


def _flat_grad(tr):
    ps = tr.reduce.params
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in ps])


def _checksum(tr):
    return [float(p.detach().double().sum()) for p in tr.reduce.params]


def _data(dev):
    from nmf_amd import synthetic
    rays, focal = synthetic.camera_rays(2 * CHUNK, seed=77)
    gt = torch.rand(2 * CHUNK, 3, generator=torch.Generator().manual_seed(5)) * 0.6 + 0.2
    return rays.to(dev), gt.to(dev), focal


def _worker(rank, world, port, out, nan_rank):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nmf_amd.noise import DeviceNoise
    from nmf_amd.trainer import Trainer, rank_slice
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    nerf, params = _build(dev)
    tr = Trainer(nerf, params, world_size=world, rank=rank)
    rays, gt, focal = _data(dev)
    sl = rank_slice(2 * CHUNK, world, rank)
    res = {"sums": [], "skipped": None}
    # step 1 of the comparison: this rank's chunk with ITS noise source; the reduced gradient is left in p.grad
    p0 = [p.detach().clone() for p in tr.reduce.params]
    tr.step(rays[sl], gt[sl], focal, noise=DeviceNoise(dev, seed=100 + rank), update_controllers=False, fixed_chunk=CHUNK,
            global_rays=2 * CHUNK)
    res["grad"] = _flat_grad(tr).cpu()
    res["sums"].append(_checksum(tr))
    # replica consistency over further steps (each rank its own rays and noise): bit-identical parameters on both ranks
    noise = DeviceNoise(dev, seed=200 + rank)
    for it in range(4):
        r2, g2, _ = _data(dev)
        perm = torch.randperm(2 * CHUNK, generator=torch.Generator().manual_seed(it)).to(dev)
        ids = perm[sl]
        tr.step(r2[ids], g2[ids], focal, noise=noise, update_controllers=False, fixed_chunk=CHUNK, global_rays=2 * CHUNK)
        res["sums"].append(_checksum(tr))
    # a non-finite loss on ONE rank: the step is skipped on EVERY rank (the guard rides in the all-reduce)
    # (the gradients stay finite: only the step-level guard can stop the update, and only if it is the reduced one)
    before = [p.detach().clone() for p in tr.reduce.params]
    if rank == nan_rank:
        chunk = tr.fast.chunk

        def nan_loss(*a, **k):
            o = chunk(*a, **k)
            if o.get("loss") is not None:
                o["loss"] = o["loss"] * float("nan")
            return o
        tr.fast.chunk = nan_loss
    tr.step(rays[sl], gt[sl], focal, noise=noise, update_controllers=False, fixed_chunk=CHUNK, global_rays=2 * CHUNK)
    res["skipped"] = all(bool(torch.equal(a, b.detach())) for a, b in zip(before, tr.reduce.params))
    res["moved"] = any(not torch.equal(a, b.detach()) for a, b in zip(p0, before))
    out[rank] = res
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_one_process_with_two_chunks():
    from nmf_amd.noise import DeviceNoise
    from nmf_amd.trainer import Trainer
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        out = m.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, out, 1)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(600)
            assert p.exitcode == 0
        res = dict(out)
    a, b = res[0], res[1]
    # every replica holds the same reduced gradient and, step after step, bit-identical parameters
    assert torch.equal(a["grad"], b["grad"])
    assert a["sums"] == b["sums"] and len(a["sums"]) == 5
    assert a["moved"] and b["moved"]
    assert a["skipped"] and b["skipped"], "a NaN loss on one rank must gate the optimizer step on every rank"
    # ---- the single process: the same 2 x CHUNK rays as two chunks of one step, chunk k with rank k's noise
    dev = torch.device("cuda", 0)
    nerf, params = _build(dev)
    tr = Trainer(nerf, params)
    rays, gt, focal = _data(dev)
    noise = _PerChunkNoise([DeviceNoise(dev, seed=100), DeviceNoise(dev, seed=101)])
    tr.step(rays, gt, focal, noise=noise, update_controllers=False, fixed_chunk=CHUNK)
    one = _flat_grad(tr).cpu()
    two = a["grad"]
    assert float(one.abs().max()) > 0
    # float atomics accumulate in another order, nothing else differs
    err = float((one - two).abs().max())
    assert err <= 2e-5 * float(one.abs().max()) + 1e-9, (err, float(one.abs().max()))
    rel = float((one - two).norm() / one.norm())
    assert rel < 1e-5, rel


def _bucket_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nmf_amd.trainer import FlatGradAllReduce
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    g = torch.Generator().manual_seed(11)
    shapes = [(1, 16, 24, 24), (1, 16, 24, 1), (64, 66), (64,), (11, 24), (3, 32, 64), (1,), (24, 72)]
    params = [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]
    early_ids = (2, 3, 4, 5, 6)                      # "BRDF MLP, heads, environment map, mip bias": final before the field walks
    gr = torch.Generator().manual_seed(100 + rank)

    def fresh():
        gs = [torch.randn(s, generator=gr).to(dev) for s in shapes]
        if rank == 1:
            gs[7] = None                              # a parameter only rank 0 has a gradient for
        gs[1] = None                                  # ... and one nobody has
        return gs
    grads = fresh()
    comm = torch.cuda.Stream()
    res = {}
    for mode in ("flat", "bucketed"):
        red = FlatGradAllReduce(params)
        red.begin_step()
        acc = [None if t is None else t.clone() for t in grads]
        for p in params:
            p.grad = None
        guard = torch.tensor(1.5 + rank, device=dev)
        if mode == "bucketed":
            comm.wait_stream(torch.cuda.current_stream())
            red.early([(params[i], acc[i]) for i in early_ids], comm)
            red.finish_early()
        for p, t in zip(params, acc):                 # "end_step": the accumulator tensors become .grad
            if t is not None:
                p.grad = t
        n = red(guard=guard)
        torch.cuda.synchronize()
        res[mode] = dict(bytes=n, guard=float(red.guard), grads=[None if p.grad is None else p.grad.detach().cpu().clone() for p in params],
                         exposed=red.exposed_ms())
    out[rank] = res
    dist.destroy_process_group()


def test_bucketed_all_reduce_is_bit_identical_to_the_flat_one():
    """FlatGradAllReduce with the early bucket (the gradients that are final before the field walks, summed on a communication stream
    next to them) against the single flat collective of rounds 1-4, two ranks: the same bytes travel, every gradient and the
    step's guard come back bit for bit, a parameter only one rank has a gradient for gets the sum on both, a parameter no rank has one
    for keeps .grad = None, and both replicas agree."""
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        out = m.dict()
        port = _free_port()
        procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, out)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        res = dict(out)
    for rank in (0, 1):
        flat, buck = res[rank]["flat"], res[rank]["bucketed"]
        assert flat["bytes"] == buck["bytes"] and flat["guard"] == buck["guard"] == 4.0
        assert flat["exposed"] is None and buck["exposed"] is not None and buck["exposed"] >= 0.0
        for a, b in zip(flat["grads"], buck["grads"]):
            assert (a is None) == (b is None)
            if a is not None:
                assert torch.equal(a, b)
        assert flat["grads"][1] is None and flat["grads"][7] is not None
    for a, b in zip(res[0]["bucketed"]["grads"], res[1]["bucketed"]["grads"]):
        assert (a is None) == (b is None) and (a is None or torch.equal(a, b))


def _miss_rays(n, dev):
    o = torch.tensor([[4.0, 4.0, 4.0]]).expand(n, 3)
    d = torch.nn.functional.normalize(torch.tensor([[1.0, 0.2, 0.1]]), dim=-1).expand(n, 3)      # pointing away from the box
    return torch.cat([o, d], -1).to(dev).contiguous()


def _empty_last_worker(rank, world, port, out, overlap):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0", NMF_OVERLAP=overlap)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nmf_amd.noise import DeviceNoise
    from nmf_amd.trainer import Trainer, check_replicas
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    nerf, params = _build(dev)
    tr = Trainer(nerf, params, world_size=world, rank=rank, check_every=1)
    rays, gt, focal = _data(dev)
    half = CHUNK // 2
    mine = rays[rank * CHUNK:(rank + 1) * CHUNK].clone()
    if rank == 1:
        mine[half:] = _miss_rays(CHUNK - half, dev)        # this rank's LAST chunk keeps no sample; its first one looks the env map up
    res = {}
    noise = _PerChunkNoise([DeviceNoise(dev, seed=300 + 2 * rank), DeviceNoise(dev, seed=301 + 2 * rank)])
    st = tr.step(mine, gt[rank * CHUNK:(rank + 1) * CHUNK], focal, noise=noise, update_controllers=False, fixed_chunk=half,
                 global_rays=2 * CHUNK)
    res["chunks"] = st["chunks"]
    res["bg"] = nerf.bg_module.bg_mat.grad.detach().float().reshape(-1).cpu()
    res["mlp"] = nerf.model.brdf.mlp[0].weight.grad.detach().float().reshape(-1).cpu()
    res["grad"] = _flat_grad(tr).cpu()
    for it in range(2):                                     # the replicas stay identical (check_every=1 compares them before each step)
        tr.step(mine, gt[rank * CHUNK:(rank + 1) * CHUNK], focal, noise=noise, update_controllers=False, fixed_chunk=half,
                global_rays=2 * CHUNK)
    check_replicas(nerf, what="after the steps with an empty last chunk")
    res["checks"] = tr.replica_checks
    out[rank] = res
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_env_gradient_is_reduced_when_the_last_chunk_of_a_rank_is_empty(overlap, monkeypatch):
    """the env-map table gradient is produced by the LAST chunk's backward; a rank whose last chunk kept no
    sample (or NMF_OVERLAP=0) used to enter the early in-place all-reduce with a zero d_bg and write its LOCAL gradient over the sum
    afterwards -- bg_mat.grad then differed between the ranks and the env maps drifted apart silently.  Two ranks x two chunks, rank
    1's last chunk empty: both ranks hold the same bg_mat gradient, it is the one-process sum over the three non-empty chunks,
    and the replicas' checksums agree step after step."""
    from nmf_amd.noise import DeviceNoise
    from nmf_amd.trainer import Trainer
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        out = m.dict()
        port = _free_port()
        procs = [ctx.Process(target=_empty_last_worker, args=(r, 2, port, out, overlap)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(600)
            assert p.exitcode == 0
        res = dict(out)
    a, b = res[0], res[1]
    assert a["chunks"] == b["chunks"] == 2 and a["checks"] == b["checks"] == 3
    assert float(a["bg"].abs().max()) > 0
    assert torch.equal(a["bg"], b["bg"]) and torch.equal(a["mlp"], b["mlp"]) and torch.equal(a["grad"], b["grad"])
    # ---- one process: the same four chunks in one step (chunk k of rank r with that chunk's noise source)
    monkeypatch.setenv("NMF_OVERLAP", overlap)
    dev = torch.device("cuda", 0)
    nerf, params = _build(dev)
    tr = Trainer(nerf, params)
    rays, gt, focal = _data(dev)
    half = CHUNK // 2
    allr = rays.clone()
    allr[CHUNK + half:] = _miss_rays(CHUNK - half, dev)
    noise = _PerChunkNoise([DeviceNoise(dev, seed=300 + k) for k in range(4)])
    tr.step(allr, gt, focal, noise=noise, update_controllers=False, fixed_chunk=half)
    one_bg = nerf.bg_module.bg_mat.grad.detach().float().reshape(-1).cpu()
    one = _flat_grad(tr).cpu()
    err = float((one_bg - a["bg"]).abs().max())
    assert err <= 2e-5 * float(one_bg.abs().max()) + 1e-9, (err, float(one_bg.abs().max()))
    rel = float((one - a["grad"]).norm() / one.norm())
    assert rel < 1e-5, rel


def _rccl_one_rank_worker(port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
    import nmf_amd.trainer as T
    nerf, _ = _build(dev)
    T._replica_group = lambda group: True          # a group of one rank still issues every collective
    before = T.replica_checksum(nerf).tolist()
    dtypes = sorted({str(t.dtype) for t in nerf.state_dict().values()})
    nbytes = T.broadcast_replica(nerf, src=0)
    after = T.check_replicas(nerf)
    torch.cuda.synchronize()
    out.update(before=before, after=after, nbytes=nbytes, dtypes=dtypes)
    dist.destroy_process_group()


def test_replica_broadcast_and_check_over_rccl():
    """What a 1-GPU box can show of the start-up broadcast over the production backend: a process group of ONE rank over RCCL runs
    every collective of broadcast_replica / check_replicas (int64 max all-reduce of the key-set signature, int64 shape broadcasts,
    one broadcast per parameter / buffer in its own dtype, the float64 biases, the float64 checksum all-reduce) and leaves the
    replica as it was.  (Two ranks: tests/test_distributed_cpu.py over gloo; RCCL refuses two ranks on one device.)"""
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        out = m.dict()
        p = ctx.Process(target=_rccl_one_rank_worker, args=(_free_port(), out))
        p.start()
        p.join(600)
        assert p.exitcode == 0
        res = dict(out)
    assert res["before"] == res["after"]
    assert res["nbytes"] > 1 << 20 and "torch.float32" in res["dtypes"]
