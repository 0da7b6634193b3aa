"""Generates tests/golden/*.npz by running the REFERENCE (/root/reference, read-only) in this
container.  Only arrays (inputs, recorded noise, outputs) are written; no reference source or
bytecode leaves the container.  Re-run with:  python tests/golden/make_golden.py [case ...]

Each case stores everything needed to replay it: the inputs, the random draws the reference made
(in call order) and the reference's outputs / parameter gradients.
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_harness as rh  # noqa: E402
from nmf_amd import synthetic  # noqa: E402


def to_np(v):
    if isinstance(v, torch.Tensor):
        v = v.detach().cpu()
        if v.dtype == torch.bool:
            return v.numpy()
        return v.numpy()
    return np.asarray(v)


def save(name, d):
    flat = {}
    for k, v in d.items():
        if isinstance(v, (list, tuple)) and len(v) > 0 and isinstance(v[0], tuple):   # noise tape
            flat[k + "/n"] = np.asarray(len(v))
            for i, (kind, t) in enumerate(v):
                flat[f"{k}/{i}/{kind}"] = to_np(t)
        else:
            flat[k] = to_np(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **flat)
    print(f"wrote {path}  ({os.path.getsize(path) / 1e6:.2f} MB, {len(flat)} arrays)")


def small_reference(grid, bg_res, seed=0, **kw):
    nerf = rh.build_reference(grid=grid, bg_resolution=bg_res, seed=seed, **kw)
    sd = synthetic.state_dict_s1(grid=grid, bg_resolution=bg_res, seed=seed)
    nerf.load_state_dict(sd, strict=False)
    return nerf, sd


def random_field_state(nerf, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in list(nerf.rf.density_rf.app_plane) + list(nerf.rf.density_rf.app_line):
            p.copy_(0.3 * torch.randn(p.shape, generator=g))
        for p in list(nerf.rf.app_rf.app_plane) + list(nerf.rf.app_rf.app_line):
            p.copy_(0.3 * torch.randn(p.shape, generator=g))


# ------------------------------------------------------------------------------------------
def case_sampler():
    G = 32
    nerf, _ = small_reference(G, 16)
    g = torch.Generator().manual_seed(1)
    vol = (torch.rand(1, 1, G, G, G, generator=g) < 0.2).float()
    from samplers.alphagrid import AlphaGridMask
    nerf.sampler.alphaMask = AlphaGridMask(nerf.rf.aabb, vol[0, 0])
    rays, focal = synthetic.camera_rays(96, seed=3)
    out = dict(grid=G, alpha_volume=np.packbits(vol.bool().numpy().reshape(-1)), rays=rays, focal=focal)
    # (a) eval, deterministic
    xyz, rv, N, z, dists, wv = nerf.sampler.sample(rays, focal, rf=nerf.rf, is_train=False)
    out.update(eval_xyz=xyz, eval_ray_valid=np.packbits(rv.numpy().reshape(-1)), eval_z=z, eval_dists=dists,
               eval_whole_valid=wv, N=N)
    # (b) train with budget truncation
    nerf.sampler.max_samples = 1500
    with rh.NoiseTape() as tape:
        xyz, rv, N, z, dists, wv = nerf.sampler.sample(rays, focal, rf=nerf.rf, is_train=True)
    assert not bool(wv.all()) and bool(wv.any())
    out.update(train_jitter=tape.draws[0][1], train_xyz=xyz, train_ray_valid=np.packbits(rv.numpy().reshape(-1)),
               train_z=z, train_dists=dists, train_whole_valid=wv, train_max_samples=1500)
    # (c) secondary rays: origins inside the box, random directions, override_near
    g = torch.Generator().manual_seed(5)
    o = (torch.rand(64, 3, generator=g) * 2 - 1) * 1.2
    dd = torch.randn(64, 3, generator=g)
    dd = dd / dd.norm(dim=-1, keepdim=True)
    dd[0] = torch.tensor([0.0, 0.0, 1.0])      # exercises the d==0 -> 1e-6 replacement
    dd[1] = torch.tensor([1.0, 0.0, 0.0])
    srays = torch.cat([o, dd], -1)
    near = 3 * nerf.sampler.stepsize
    with rh.NoiseTape() as tape:
        xyz, rv, N, z, dists, wv = nerf.sampler.sample(srays, focal, rf=nerf.rf, is_train=True,
                                                       override_near=near, dynamic_batch_size=False)
    out.update(sec_rays=srays, sec_near=near, sec_jitter=tape.draws[0][1], sec_xyz=xyz,
               sec_ray_valid=np.packbits(rv.numpy().reshape(-1)), sec_z=z, sec_dists=dists)
    save("sampler", out)


def case_field():
    G = 24
    nerf, _ = small_reference(G, 16)
    random_field_state(nerf, 7)
    g = torch.Generator().manual_seed(11)
    xyz = (torch.rand(400, 4, generator=g) * 2 - 1) * 1.5
    xyz[:8, :3] = torch.tensor([[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5], [0, 0, 0], [1.5, 0, -1.5],
                                [0.0652174, 0.3, 1.5], [-1.5, 1.5, 0.75], [1.4999, -1.4999, 0.1], [0.5, 0.5, 0.5]])
    lat = torch.linspace(-1.5, 1.5, G)
    xyz[8:16, 0] = lat[3:11]                    # samples exactly on lattice planes
    rf = nerf.rf
    sf = rf.compute_densityfeature(xyz, activate=False)
    sg = rf.compute_densityfeature(xyz)
    app = rf.compute_appfeature(xyz)
    nrm = rf.compute_normals(xyz)
    ca, cb, cc, cd = (torch.randn(s, generator=g) for s in (sg.shape, app.shape, nrm.shape, sf.shape))
    loss = (sg * ca).sum() + (app * cb).sum() + (nrm * cc).sum() + (sf * cd).sum()
    params = dict(nerf.rf.named_parameters())
    grads = torch.autograd.grad(loss, [p for n, p in params.items() if "dbasis" not in n])
    out = dict(grid=G, xyz=xyz, sigma_feat=sf, sigma=sg, app=app, normals=nrm, ca=ca, cb=cb, cc=cc, cd=cd)
    for (n, p), gr in zip([(n, p) for n, p in params.items() if "dbasis" not in n], grads):
        out["param/" + n] = p
        out["grad/" + n] = gr
    save("field", out)


def case_alpha_mask():
    G = 20
    nerf, sd = small_reference(G, 16)
    nerf.rf.density_shift = -8.0          # coarse grid => long steps; keep empty space below the 1e-3 threshold
    nerf.sampler.update(nerf.rf, init=False)
    vol1 = nerf.sampler.alphaMask.alpha_volume.clone()
    # second rebuild goes through the existing mask (compute_alpha's alphaMask branch)
    with torch.no_grad():
        nerf.rf.density_rf.app_plane[0][0, 0] *= 0.5
    nerf.sampler.update(nerf.rf, init=False)
    vol2 = nerf.sampler.alphaMask.alpha_volume.clone()
    save("alpha_mask", dict(grid=G, density_shift=-8.0, vol1=np.packbits(vol1.bool().numpy().reshape(-1)),
                            vol2=np.packbits(vol2.bool().numpy().reshape(-1)),
                            n1=int(vol1.sum()), n2=int(vol2.sum())))


def case_env():
    from modules.integral_equirect import IntegralEquirect
    H = 32
    g = torch.Generator().manual_seed(21)
    env = IntegralEquirect(bg_resolution=H, mipbias=1, activation="exp", lr=0.02, init_val=-0.6, mul_lr=0,
                           brightness_lr=0, mipbias_lr=1e-4, mipnoise=0.0)
    with torch.no_grad():
        env.bg_mat.copy_(-0.6 + 0.7 * torch.randn(1, 3, H, 2 * H, generator=g))
    n = 700
    dirs = torch.randn(n, 3, generator=g)
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    dirs[0] = torch.tensor([0.0, 0.0, 1.0])
    dirs[1] = torch.tensor([0.0, 0.0, -1.0])
    dirs[2] = torch.tensor([1.0, 0.0, 0.0])
    dirs[3] = torch.tensor([-1.0, 1e-4, 0.0])
    dirs[4] = torch.tensor([-1.0, -1e-4, 0.0])
    dirs[5] = torch.nn.functional.normalize(torch.tensor([0.05, 0.02, 0.99]), dim=0)
    dirs[6] = torch.nn.functional.normalize(torch.tensor([-0.3, 1e-3, -0.95]), dim=0)
    dirs[7:40, 2] = torch.linspace(-0.999, 0.999, 33)
    dirs[7:40] = dirs[7:40] / dirs[7:40].norm(dim=-1, keepdim=True)
    sa = torch.rand(n, generator=g) * 14 - 12
    sa[40:60] = -100.0
    sa[60:80] = 2.0
    dirs.requires_grad_(True)
    vals = env(dirs, sa)
    c = torch.randn(vals.shape, generator=g)
    gb, gm, gd = torch.autograd.grad((vals * c).sum(), [env.bg_mat, env.mipbias, dirs])
    coeffs, conv = env.get_spherical_harmonics(100)
    save("env", dict(H=H, bg_mat=env.bg_mat, dirs=dirs, sa=sa, vals=vals, c=c, grad_bg=gb, grad_mipbias=gm,
                     grad_dirs=gd, sh_coeffs=coeffs, sh_conv=conv, mean_color=env.mean_color()))


def case_shading_parts():
    from brdf_samplers.ggx import GGXSampler
    from modules.pt_selectors import select_bounces
    nerf, sd = small_reference(16, 16)
    g = torch.Generator().manual_seed(31)
    out = {}
    # --- GGX sampling -----------------------------------------------------------------------
    Mb, m = 60, 24
    V = torch.nn.functional.normalize(torch.randn(Mb, 3, generator=g), dim=-1)
    N = torch.nn.functional.normalize(V + 0.8 * torch.randn(Mb, 3, generator=g), dim=-1)
    N[0] = torch.tensor([0.0, 0.0, 1.0])
    N[1] = torch.tensor([0.0, 0.0, -1.0])
    N[2] = torch.nn.functional.normalize(torch.tensor([0.01, 0.0, 0.9999]), dim=0)
    N = N * (V * N).sum(-1, keepdim=True).sign()
    r = torch.rand(Mb, 1, generator=g) * 0.49 + 0.01
    r[3] = 0.01
    r[4] = 0.5
    counts = torch.randint(1, m + 1, (Mb,), generator=g)
    ray_mask = torch.arange(m)[None] < counts[:, None]
    u = torch.rand(Mb, m, 2, generator=g)
    N.requires_grad_(True)
    r.requires_grad_(True)
    smp = nerf.model.brdf_sampler
    L, basisT, logp = smp.sample(u[..., 0], u[..., 1], V, N, r, r, ray_mask)
    c = torch.randn(L.shape, generator=g)
    gN, gr = torch.autograd.grad((L * c).sum(), [N, r])
    out.update(ggx_V=V, ggx_N=N, ggx_r=r, ggx_u=u, ggx_ray_mask=ray_mask, ggx_L=L, ggx_basisT=basisT,
               ggx_logp=logp, ggx_c=c, ggx_gN=gN, ggx_gr=gr)
    # --- Sobol draw ---------------------------------------------------------------------------
    with rh.NoiseTape() as tape:
        angs = smp.draw(Mb, m)
    out.update(sobol_table=smp.angs, sobol_offset=tape.draws[0][1], sobol_out=angs)
    # --- BRDF MLP -----------------------------------------------------------------------------
    R = 257
    hv = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    dv = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    feat = torch.randn(R, 24, generator=g)
    rough = torch.rand(R, generator=g) * 0.49 + 0.01
    brdf = nerf.model.brdf
    brdf.bias = 0.37
    feat.requires_grad_(True)
    w = brdf(hv, hv, hv, hv, hv, hv, dv, feat, rough, rough)
    c2 = torch.randn(w.shape, generator=g)
    names = [n for n, _ in brdf.named_parameters()]
    gs = torch.autograd.grad((w * c2).sum(), [feat] + [p for _, p in brdf.named_parameters()])
    out.update(brdf_half=hv, brdf_diff=dv, brdf_feat=feat, brdf_rough=rough, brdf_bias=0.37, brdf_out=w,
               brdf_c=c2, brdf_gfeat=gs[0])
    for n, p, gq in zip(names, [p for _, p in brdf.named_parameters()], gs[1:]):
        out["brdf_param/" + n] = p
        out["brdf_grad/" + n] = gq
    # --- material heads -----------------------------------------------------------------------
    dm = nerf.model.diffuse_module
    f2 = torch.randn(100, 24, generator=g) * 3
    albedo, tint, mp = dm(torch.zeros(100, 4), torch.zeros(100, 3), f2, std=0)
    out.update(heads_feat=f2, heads_albedo=albedo, heads_tint=tint, heads_f0=mp["f0"], heads_r1=mp["r1"],
               heads_r2=mp["r2"])
    for n, p in dm.named_parameters():
        out["heads_param/" + n] = p
    # --- select_bounces ------------------------------------------------------------------------
    b, Ns = 40, 30
    app_mask = torch.rand(b, Ns, generator=g) < 0.3
    weights = torch.rand(b, Ns, generator=g) ** 4 * app_mask
    with rh.NoiseTape() as tape:
        bm0, rm0 = select_bounces(weights, app_mask, 650000, 0.0, 128)
    out.update(sel_weights=weights, sel_app_mask=app_mask, sel0_u=tape.draws[0][1], sel0_bounce=bm0, sel0_ray_mask=rm0)
    with rh.NoiseTape() as tape:
        bm1, rm1 = select_bounces(weights, app_mask, 4000, 0.0, None)
    out.update(sel1_u=tape.draws[0][1], sel1_bounce=bm1, sel1_ray_mask=rm1, sel1_num=4000)
    with rh.NoiseTape() as tape:
        bm2, rm2 = select_bounces(weights, app_mask, 100, 0.0, None)     # N <= 0 branch
    out.update(sel2_u=tape.draws[0][1], sel2_bounce=bm2, sel2_ray_mask=rm2, sel2_num=100)
    # --- compositing ---------------------------------------------------------------------------
    from modules.row_mask_sum import row_mask_sum
    from modules.tensor_nerf import raw2alpha
    from modules.tonemap import SRGBTonemap
    sigma = torch.rand(b, Ns, generator=g) * 40 * app_mask
    dists = torch.rand(b, Ns, generator=g) * 0.02
    dists[:, -1] = 0
    wgt = raw2alpha(sigma, dists * 25)
    rgbm = torch.rand(int(app_mask.sum()), 3, generator=g)
    comp = row_mask_sum(wgt[app_mask][..., None] * rgbm, app_mask)
    tm = SRGBTonemap()
    x = torch.linspace(-0.1, 1.3, 50)
    out.update(comp_sigma=sigma, comp_dists=dists, comp_weight=wgt, comp_rgb=rgbm, comp_out=comp,
               tm_in=x, tm_clip=tm(x), tm_noclip=tm(x, noclip=True))
    save("shading_parts", out)


def _run_e2e(nerf, rays, focal, is_train, tape_store=True):
    with rh.NoiseTape() as tape:
        ims, stats = nerf(rays, focal, bg_col=torch.ones(3), is_train=is_train, ndc_ray=False)
    return ims, stats, tape.draws


def case_e2e_small():
    G, BG, B = 48, 32, 160
    for tag, is_train, detach in (("train", True, False), ("train_detachN", True, True), ("eval", False, True)):
        nerf, sd = small_reference(G, BG, max_retrace_rays=(300,), max_samples=2600)
        nerf.sampler.update(nerf.rf, init=False)
        nerf.sampler.update(nerf.rf, init=True)
        nerf.model.detach_N = detach
        nerf.model.brdf.bias = 0.21
        nerf.model.diffuse_module.diffuse_bias = -0.4
        nerf.model.diffuse_module.roughness_bias = -0.7
        rays, focal = synthetic.camera_rays(B, seed=2)
        ims, stats, draws = _run_e2e(nerf, rays, focal, is_train)
        out = dict(grid=G, bg_res=BG, rays=rays, focal=focal, max_retrace=300, max_samples=2600,
                   detach_N=detach, brdf_bias=0.21, diffuse_bias=-0.4, roughness_bias=-0.7,
                   alpha_volume=np.packbits(nerf.sampler.alphaMask.alpha_volume.bool().numpy().reshape(-1)),
                   rgb_map=ims["rgb_map"], acc_map=ims["acc_map"], whole_valid=stats["whole_valid"],
                   n_samples=np.asarray(stats["n_samples"]), noise=draws)
        if is_train:
            g = torch.Generator().manual_seed(9)
            gt = torch.rand(B, 3, generator=g)
            wv = stats["whole_valid"]
            loss = ((ims["rgb_map"].clip(max=1).clip(0, 1) - gt[wv].clip(0, 1)) ** 2).sum()
            total = (loss + 0.1 * stats["ori_loss"] + 3e-4 * stats["prediction_loss"]
                     + 8e-5 * nerf.rf.density_L1()) / 4096
            total.backward()
            out.update(gt=gt, loss=loss, total=total, ori_loss=stats["ori_loss"],
                       prediction_loss=stats["prediction_loss"], diffuse_reg=stats["diffuse_reg"],
                       brdf_reg=stats["brdf_reg"], envmap_reg=stats["envmap_reg"])
            for n, p in nerf.named_parameters():
                if p.grad is not None:
                    out["gradnorm/" + n] = p.grad.norm()
                    if p.numel() <= 5000:
                        out["grad/" + n] = p.grad
            out["grad_slice/bg_mat"] = nerf.bg_module.bg_mat.grad[0, :, ::4, ::4]
            out["grad_slice/density_plane0"] = nerf.rf.density_rf.app_plane[0].grad[0, :, ::3, ::3]
            out["grad_slice/app_plane1"] = nerf.rf.app_rf.app_plane[1].grad[0, :, ::3, ::3]
            for k in ("diffuse", "tint", "roughness", "spec", "albedo"):
                out["debug/" + k] = ims[k]
        else:
            out.update(depth=ims["depth"], world_normal=ims["world_normal"])
            for k in ("diffuse", "tint", "roughness", "spec", "albedo"):
                out["debug/" + k] = ims[k]
        save("e2e_small_" + tag, out)


class BookkeepingTap:
    """Records the reference's own bookkeeping decisions while it runs: the per-sample secondary-ray counts of
    select_bounces (per recursion level, in call order) and every argsort the shading model takes (the re-trace order,
    models/microfacet.py:506-509).  Outputs of reference functions, no reference code."""

    def __init__(self):
        self.counts, self.orders, self.valid = [], [], []

    def __enter__(self):
        import models.microfacet as mm
        from samplers.alphagrid import AlphaGridSampler
        self._mm, self._sel, self._argsort = mm, mm.select_bounces, torch.Tensor.argsort
        self._smp_cls, self._smp = AlphaGridSampler, AlphaGridSampler.sample
        tap = self

        def sample(smp, *a, **k):
            out = tap._smp(smp, *a, **k)
            tap.valid.append(out[1].clone())                # ray_valid [b, N] of this recursion level
            return out

        AlphaGridSampler.sample = sample

        def sel(weights, app_mask, *a, **k):
            bm, rm = tap._sel(weights, app_mask, *a, **k)
            c = torch.zeros(bm.shape[0], dtype=torch.int32)
            c[bm] = rm.sum(1).int()
            tap.counts.append(c)
            return bm, rm

        def argsort(t, *a, **k):
            o = tap._argsort(t, *a, **k)
            if t.dim() == 1 and t.is_floating_point():
                tap.orders.append(o.clone())
            return o

        mm.select_bounces = sel
        torch.Tensor.argsort = argsort
        return self

    def __exit__(self, *exc):
        self._mm.select_bounces = self._sel
        torch.Tensor.argsort = self._argsort
        self._smp_cls.sample = self._smp
        return False


def _store_grads(out, nerf, full_bytes=1 << 20):
    """norm of every parameter gradient; the full gradient of every tensor <= 1 MiB; strided slices of the larger ones"""
    for n, p in nerf.named_parameters():
        if p.grad is None:
            continue
        out["gradnorm/" + n] = p.grad.norm()
        if p.grad.numel() * p.grad.element_size() <= full_bytes:
            out["grad/" + n] = p.grad
        elif p.grad.dim() == 4:
            out["grad_slice4/" + n] = p.grad[0, :, ::4, ::4]


def _full_size_case(name, max_retrace, G=128, BG=512, B=4096, ray_seed=0, noise_seed=1234, near_far=(2.5, 7.0), aabb_half=1.5,
                    roughness_bias=None, eye=None):
    """Full BASELINE size (4096 rays, 128^3, 512x1024 env): noise is replayed BY SEED (the global torch CPU generator);
    per-ray outputs, losses, parameter gradients and the reference's bookkeeping decisions are stored."""
    nerf, sd = small_reference(G, BG, max_retrace_rays=(max_retrace,), near_far=tuple(near_far), aabb_half=aabb_half)
    if roughness_bias is not None:            # (a calibrated attribute of the reference's diffuse module, train.py:429-437)
        nerf.model.diffuse_module.roughness_bias = roughness_bias
    nerf.sampler.update(nerf.rf, init=False)
    nerf.sampler.update(nerf.rf, init=True)
    nerf.model.detach_N = False
    rays, focal = synthetic.camera_rays(B, seed=ray_seed) if eye is None else synthetic.camera_rays(B, seed=ray_seed, eye=tuple(eye))
    torch.manual_seed(noise_seed)
    with BookkeepingTap() as tap:
        ims, stats = nerf(rays, focal, bg_col=torch.ones(3), is_train=True, ndc_ray=False)
    g = torch.Generator().manual_seed(9)
    gt = torch.rand(B, 3, generator=g)
    wv = stats["whole_valid"]
    loss = ((ims["rgb_map"].clip(max=1).clip(0, 1) - gt[wv].clip(0, 1)) ** 2).sum()
    total = (loss + 0.1 * stats["ori_loss"] + 3e-4 * stats["prediction_loss"] + 8e-5 * nerf.rf.density_L1()) / 4096
    total.backward()
    out = dict(grid=G, bg_res=BG, n_rays=B, ray_seed=ray_seed, noise_seed=noise_seed, max_retrace=max_retrace,
               near_far=np.asarray(near_far, dtype=np.float64), aabb_half=float(aabb_half),
               roughness_bias=float(nerf.model.diffuse_module.roughness_bias), eye=np.asarray(eye if eye is not None else (2.4, -2.8, 1.6)),
               rgb_map=ims["rgb_map"], acc_map=ims["acc_map"],
               whole_valid=wv, n_samples=np.asarray(stats["n_samples"]), loss=loss, total=total,
               ori_loss=stats["ori_loss"], prediction_loss=stats["prediction_loss"],
               n_alpha=int(nerf.sampler.alphaMask.alpha_volume.sum()))
    for lvl, c in enumerate(tap.counts):
        assert int(c.max()) < 32768
        out[f"counts{lvl}"] = c.to(torch.int16)
    # which candidate steps of the SECONDARY rays survive the occupancy test (samplers/alphagrid.py:341-346): their
    # origins / directions come out of a float chain (normals -> GGX), so an implementation that differs in the last bit of a
    # direction flips a sample that sits on a voxel boundary -- recorded so that a replay can pin them
    assert len(tap.valid) == 2
    out["valid1"] = np.packbits(tap.valid[1].numpy().reshape(-1))
    out["valid1_shape"] = np.asarray(tap.valid[1].shape)
    assert len(tap.orders) == 1
    order = tap.orders[0]
    R = order.shape[0]
    out["n_secondary"] = R
    if max_retrace >= R:
        out["retrace_order0"] = order.int()                 # every ray re-traced: the order pairs rays with jitter rows
    else:
        out["retrace_idx0"] = order[R - max_retrace:].int()  # the re-traced rays, in the order they are traced
    _store_grads(out, nerf)
    save(name, out)


def case_e2e_full():
    """early phase (first 19 chunks after any (re)start): 1000 of ~246 k secondary rays are re-traced"""
    _full_size_case("e2e_full_seeded", 1000)


def case_e2e_full_steady():
    """steady state (models/microfacet.py:241-268): max_retrace_rays has ramped to max_brdf_rays[0], every
    secondary ray is re-traced -- the regime bench.py times"""
    _full_size_case("e2e_full_steady", 650000)


def case_e2e_g300():
    """final grid of the schedule (300^3, 1036 steps per ray, 41 MB of factor tables), steady state, small ray batch"""
    _full_size_case("e2e_g300_steady", 650000, G=300, B=192, ray_seed=4, noise_seed=77)


def case_e2e_g300_1k():
    """the same at a realistic load: 1024 rays at 300^3, steady state -- 57 k secondary rays and 0.3 M
    level-1 samples through the walks of the final grid, against 10 k in e2e_g300_steady"""
    _full_size_case("e2e_g300_steady_1k", 650000, G=300, B=1024, ray_seed=5, noise_seed=99)


def case_init_step():
    """the same single chunk from an INITIAL state: the freshly constructed and calibrated model of the reference's own S2 run 0
    (tests/golden/psnr_trace.npz s0/init/*, s0/biases), max_retrace_rays 1000 as at the start of a run.  Round 6: the seed-mean gradient
    norm of the density planes differs between the two sides already in iterations 0-7 (profiles/r06_psnr_trajectory.txt)."""
    here = os.path.dirname(os.path.abspath(__file__))
    tr = np.load(os.path.join(here, "psnr_trace.npz"))
    st = {"sd/" + k[len("s0/init/"):]: tr[k] for k in tr.files if k.startswith("s0/init/")}
    st.update(biases=tr["s0/biases"], max_retrace=np.asarray(1000), num_rays=np.asarray(471), min_rough=np.asarray(0.0),
              ori_lambda=np.asarray(0.1), pred_lambda=np.asarray(3e-4))
    _state_step("init_step", st, noise_seed=6161)


def case_trained_step():
    """Round 6: ONE training chunk of the reference from a TRAINED state -- the model of this build's own S2 training after 100
    iterations (tests/golden/trained_state_it100.npz, written on the GPU by tools/trained_state_dump.py; 48^3, 32 x 64 env map, the
    configuration of the PSNR runs: max_samples 40 000, max_brdf_rays [80 000, 40 000], partial re-trace).  Every fixture before this one
    sits at scene S1's synthetic state; the question here is whether the two sides' single-step gradients also agree where training
    takes the model (sharp roughness, a learnt env map, density factors that are no longer an indicator function)."""
    st = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "trained_state_it100.npz"))
    _state_step("trained_step", {k: st[k] for k in st.files}, noise_seed=5150)


def _state_step(name, st, noise_seed):
    tr = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "psnr_trace.npz"))
    G, BG, B = 48, 32, int(st["num_rays"])
    max_retrace = int(st["max_retrace"])
    nerf = rh.build_reference(grid=G, bg_resolution=BG, max_samples=40000, max_brdf_rays=(80000, 40000),
                              max_retrace_rays=(max_retrace,), target_num_samples=(80000,))
    sd = {k[3:]: torch.as_tensor(st[k]) for k in st if k.startswith("sd/")}
    missing, unexpected = nerf.load_state_dict(sd, strict=False)
    assert not [k for k in missing if "alphaMask" not in k], missing
    nerf.model.brdf.bias, nerf.model.diffuse_module.diffuse_bias, nerf.model.diffuse_module.roughness_bias = (float(v) for v in st["biases"])
    nerf.sampler.update(nerf.rf, init=True)                     # no alpha mask before iteration 2000 (sampler.update_list)
    nerf.model.detach_N = False
    nerf.model.min_rough = float(st["min_rough"])
    rays, gt, focal = torch.as_tensor(tr["rays_train"][:B]), torch.as_tensor(tr["rgb_train"][:B]), float(tr["focal"])
    ori_lambda, pred_lambda = float(st["ori_lambda"]), float(st["pred_lambda"])
    torch.manual_seed(noise_seed)
    with BookkeepingTap() as tap:
        ims, stats = nerf(rays, focal, bg_col=torch.ones(3), is_train=True, ndc_ray=False)
    wv = stats["whole_valid"]
    loss = ((ims["rgb_map"].clip(0, 1) - gt[wv].clip(0, 1)) ** 2).sum()                       # train.py:598-601
    total = (loss + ori_lambda * stats["ori_loss"] + pred_lambda * stats["prediction_loss"] + 8e-5 * nerf.rf.density_L1()) / 1024
    total.backward()
    out = dict(grid=G, bg_res=BG, n_rays=B, noise_seed=noise_seed, max_retrace=max_retrace, focal=focal, rays=rays, gt=gt,
               ori_lambda=ori_lambda, pred_lambda=pred_lambda, min_rough=float(st["min_rough"]), biases=st["biases"],
               rgb_map=ims["rgb_map"], acc_map=ims["acc_map"], whole_valid=wv, n_samples=np.asarray(stats["n_samples"]),
               loss=loss, total=total, ori_loss=stats["ori_loss"], prediction_loss=stats["prediction_loss"])
    for lvl, c in enumerate(tap.counts):
        out[f"counts{lvl}"] = c.to(torch.int16)
    assert len(tap.valid) == 2 and len(tap.orders) == 1
    out["valid1"] = np.packbits(tap.valid[1].numpy().reshape(-1))
    out["valid1_shape"] = np.asarray(tap.valid[1].shape)
    order = tap.orders[0]
    R = order.shape[0]
    out["n_secondary"] = R
    out["retrace_order0"] = order.int()          # the full argsort: [not re-traced | re-traced], both in the reference's order
    for n, p in nerf.named_parameters():
        if p.grad is not None:
            out["gradnorm/" + n] = p.grad.norm()
            out["grad/" + n] = p.grad                                  # 48^3: every gradient in full
    for k, v in sd.items():
        out["sd/" + k] = v
    print(name, ": n_samples", stats["n_samples"], "R", R, "kept", int(wv.sum()), "of", B, "max_retrace", max_retrace)
    save(name, out)


def _step_stats(name, st, K, chunks):
    """`chunks` = [(first ray, last ray)] of tests/golden/psnr_trace.npz's training rays: ONE optimizer step's worth of chunks, their
    gradients accumulated as train.py:509-712 does (each chunk's loss / 1024), K times with the reference's OWN noise (torch's global
    generator, seeds 1000 .. 1000 + K - 1): per run the loss terms and sample counts of the first chunk and the norm of every parameter's
    accumulated gradient -- the DISTRIBUTION of a step's gradient at a given state (tools/trained_step_stats.py draws the same from this
    build with its production noise source)"""
    here = os.path.dirname(os.path.abspath(__file__))
    tr = np.load(os.path.join(here, "psnr_trace.npz"))
    G, BG = 48, 32
    max_retrace = int(st["max_retrace"])
    nerf = rh.build_reference(grid=G, bg_resolution=BG, max_samples=40000, max_brdf_rays=(80000, 40000),
                              max_retrace_rays=(max_retrace,), target_num_samples=(80000,))
    sd = {k[3:]: torch.as_tensor(st[k]) for k in st if k.startswith("sd/")}
    nerf.load_state_dict(sd, strict=False)
    nerf.model.brdf.bias, nerf.model.diffuse_module.diffuse_bias, nerf.model.diffuse_module.roughness_bias = (float(v) for v in st["biases"])
    nerf.sampler.update(nerf.rf, init=True)
    nerf.model.detach_N = False
    nerf.model.min_rough = float(st["min_rough"])
    focal = float(tr["focal"])
    ori_lambda, pred_lambda = float(st["ori_lambda"]), float(st["pred_lambda"])
    names = [n for n, _ in nerf.named_parameters()]
    rows, losses, ns = [], [], []
    for k in range(K):
        for p in nerf.parameters():
            p.grad = None
        torch.manual_seed(1000 + k)
        for ci, (lo, hi) in enumerate(chunks):
            rays, gt = torch.as_tensor(tr["rays_train"][lo:hi]), torch.as_tensor(tr["rgb_train"][lo:hi])
            nerf.model.max_retrace_rays = [max_retrace]
            ims, stats = nerf(rays, focal, bg_col=torch.ones(3), is_train=True, ndc_ray=False)
            wv = stats["whole_valid"]
            loss = ((ims["rgb_map"].clip(0, 1) - gt[wv].clip(0, 1)) ** 2).sum()
            total = (loss + ori_lambda * stats["ori_loss"] + pred_lambda * stats["prediction_loss"] + 8e-5 * nerf.rf.density_L1()) / 1024
            total.backward()
            if ci == 0:
                losses.append([float(loss), float(stats["ori_loss"]), float(stats["prediction_loss"]), float(total)])
                ns.append([int(v) for v in stats["n_samples"]] + [int(wv.sum())])
        g = dict(nerf.named_parameters())
        rows.append([float(g[n].grad.norm()) if g[n].grad is not None else np.nan for n in names])
        print(k, losses[-1], ns[-1], flush=True)
    save(name, dict(names="\n".join(names), gradnorm=np.asarray(rows), losses=np.asarray(losses), n_samples=np.asarray(ns),
                    max_retrace=max_retrace, chunks=np.asarray(chunks)))


def case_trained_step_stats(K=40):
    """the chunk of case_trained_step K times with the reference's own noise"""
    here = os.path.dirname(os.path.abspath(__file__))
    st = np.load(os.path.join(here, "trained_state_it100.npz"))
    _step_stats("trained_step_stats", {k: st[k] for k in st.files}, K, [(0, int(st["num_rays"]))])


def case_init_step3_stats(K=40):
    """a whole optimizer step (three chunks: 471 + 471 + 82 of the first 1024 training rays) from the INITIAL state of the reference's
    run 0, K times: the gradient a first iteration hands to Adam"""
    here = os.path.dirname(os.path.abspath(__file__))
    tr = np.load(os.path.join(here, "psnr_trace.npz"))
    st = {"sd/" + k[len("s0/init/"):]: tr[k] for k in tr.files if k.startswith("s0/init/")}
    st.update(biases=tr["s0/biases"], max_retrace=np.asarray(1000), min_rough=np.asarray(0.0), ori_lambda=np.asarray(0.1),
              pred_lambda=np.asarray(3e-4))
    _step_stats("init_step3_stats", st, K, [(0, 471), (471, 942), (942, 1024)])


def case_simple_sampler():
    """train.py:34-51 as it is: the ids of a sequence of nextids() calls with the chunk sizes of a run (100, then 471 + 453, then
    471 + 471 + 82 per iteration ...) over 5000 rays, torch seeded -- incl. the overlap of consecutive chunks of different sizes and a
    re-permutation"""
    rh.install_stubs()
    import sys as _sys
    _sys.modules["hydra"].utils = _sys.modules["hydra.utils"]
    import types as _types
    tb = _types.ModuleType("torch.utils.tensorboard")          # (train.py imports SummaryWriter: absent here, no arithmetic in it)
    tb.SummaryWriter = rh._Anything
    _sys.modules["torch.utils.tensorboard"] = tb
    import train as ref_train
    torch.manual_seed(77)
    smp = ref_train.SimpleSampler(5000, 1024)
    sizes = [100, 471, 453] + [471, 471, 82] * 5 + [1024, 7, 300]
    ids = [smp.nextids(b)[0].clone() for b in sizes]
    save("simple_sampler", dict(total=5000, batch=1024, seed=77, sizes=np.asarray(sizes), ids=torch.cat(ids), curr=smp.curr))


def case_e2e_variant():
    """the scene variations of the reference's dataset configs at full size: near_far [2, 6]
    (configs/dataset/materials.yaml), aabb_scale 2 (helmet.yaml:8: the box of the field is twice the scene box), a high-specular
    material (roughness_bias -2.5: sharp GGX lobes, small env-map footprints), another camera; steady state"""
    _full_size_case("e2e_variant_steady", 650000, ray_seed=3, noise_seed=4321, near_far=(2.0, 6.0), aabb_half=3.0,
                    roughness_bias=-2.5, eye=(-3.1, 2.2, 2.9))


def _full_size_eval_case(name, G=128, BG=512, B=4096, ray_seed=0, noise_seed=2468, max_retrace=650000):
    """the reference's `is_train=False` forward at full size (modules/tensor_nerf.py:210-674 with the evaluation
    branches :480-566 -- depth, world_normal, debug maps; renderer.py:56-106 calls it chunk by chunk), steady re-trace state.  Stored:
    rgb_map, acc_map, depth, world_normal per ray, n_samples, whole_valid, and the run's bookkeeping decisions (bounce counts,
    occupancy bits of the secondary rays, the re-trace argsort) like the training fixtures; noise replayed BY SEED."""
    nerf, sd = small_reference(G, BG, max_retrace_rays=(max_retrace,))
    nerf.sampler.update(nerf.rf, init=False)
    nerf.sampler.update(nerf.rf, init=True)
    nerf.model.detach_N = False
    nerf.eval()
    rays, focal = synthetic.camera_rays(B, seed=ray_seed)
    torch.manual_seed(noise_seed)
    with torch.no_grad(), BookkeepingTap() as tap:
        ims, stats = nerf(rays, focal, bg_col=torch.ones(3), is_train=False, ndc_ray=False)
    wv = stats["whole_valid"]
    out = dict(grid=G, bg_res=BG, n_rays=B, ray_seed=ray_seed, noise_seed=noise_seed, max_retrace=max_retrace,
               near_far=np.asarray((2.5, 7.0), dtype=np.float64), aabb_half=1.5,
               roughness_bias=float(nerf.model.diffuse_module.roughness_bias), eye=np.asarray((2.4, -2.8, 1.6)),
               rgb_map=ims["rgb_map"], acc_map=ims["acc_map"], depth=ims["depth"], world_normal=ims["world_normal"],
               whole_valid=wv, n_samples=np.asarray(stats["n_samples"]),
               n_alpha=int(nerf.sampler.alphaMask.alpha_volume.sum()))
    for k in ("diffuse", "roughness", "albedo"):          # three of the per-ray debug maps of the evaluation branch
        out["debug/" + k] = ims[k]
    for lvl, c in enumerate(tap.counts):
        assert int(c.max()) < 32768
        out[f"counts{lvl}"] = c.to(torch.int16)
    assert len(tap.valid) == 2
    out["valid1"] = np.packbits(tap.valid[1].numpy().reshape(-1))
    out["valid1_shape"] = np.asarray(tap.valid[1].shape)
    assert len(tap.orders) == 1
    order = tap.orders[0]
    R = order.shape[0]
    out["n_secondary"] = R
    assert max_retrace >= R
    out["retrace_order0"] = order.int()
    print(name, "n_samples", stats["n_samples"], "R", R, "kept", int(wv.sum()))
    save(name, out)


def case_e2e_full_eval():
    """4096 rays, 128^3: the chunk `extras.inference` of bench.py renders 157 times per 800 x 800 frame"""
    _full_size_eval_case("e2e_full_eval")


def case_e2e_g300_eval():
    """300^3, 1024 rays: the level-1 walks / queries of the final grid under a realistic load (e2e_g300_steady has 192 rays)"""
    _full_size_eval_case("e2e_g300_eval", G=300, B=1024, ray_seed=6, noise_seed=1357)


def case_upsample():
    """a25: TensoRF.upsample + update_stepSize + the voxel schedule (fields/tensoRF.py:207-227,
    fields/tensor_base.py:194-243, utils.py:55-58): factors before / after one scheduled upsample."""
    nerf, _ = small_reference(grid=16, bg_res=16)
    random_field_state(nerf, 5)
    rf = nerf.rf
    out = dict(N_voxel_list=np.asarray(rf.N_voxel_list), upsamp_list=np.asarray(rf.upsamp_list),
               grid0=rf.grid_size, stepsize0=rf.stepsize, nSamples0=np.asarray(rf.nSamples), units0=rf.units)
    for i in range(3):
        out[f"d_plane{i}_0"], out[f"d_line{i}_0"] = rf.density_rf.app_plane[i], rf.density_rf.app_line[i]
        out[f"a_plane{i}_0"], out[f"a_line{i}_0"] = rf.app_rf.app_plane[i], rf.app_rf.app_line[i]
    target = [21, 21, 21]
    rf.upsample_volume_grid(target)
    out.update(target=np.asarray(target), grid1=rf.grid_size, stepsize1=rf.stepsize, nSamples1=np.asarray(rf.nSamples),
               units1=rf.units)
    for i in range(3):
        out[f"d_plane{i}_1"], out[f"d_line{i}_1"] = rf.density_rf.app_plane[i], rf.density_rf.app_line[i]
        out[f"a_plane{i}_1"], out[f"a_line{i}_1"] = rf.app_rf.app_plane[i], rf.app_rf.app_line[i]
    # the production schedule (128^3 -> 300^3): resolutions the reference derives for each upsample iteration
    import utils as ref_utils
    aabb = torch.tensor([[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]])
    nv = (torch.round(torch.linspace(2097156 ** (1 / 3), 27000000 ** (1 / 3), 6) ** 3).long()).tolist()
    out["sched_voxels"] = np.asarray(nv)
    out["sched_reso"] = np.asarray([ref_utils.N_to_reso(n, aabb) for n in nv])
    save("upsample", out)


def case_blender_rays():
    """(f) Blender loader: the reference's pinhole ray construction (dataLoader/ray_utils.py:23-41,65-85 as used by
    dataLoader/blender.py:97-173) for a small image; kornia.create_meshgrid (pixel lattice, absent here) is restated."""
    import types
    kornia = sys.modules.get("kornia") or types.ModuleType("kornia")

    def create_meshgrid(H, W, normalized_coordinates=False):
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        return torch.stack([xs, ys], -1)[None]

    kornia.create_meshgrid = create_meshgrid
    sys.modules["kornia"] = kornia
    rh.install_stubs()
    sys.path.insert(0, "/root/reference")
    from dataLoader import ray_utils as ru
    ru.create_meshgrid = create_meshgrid
    H, W = 5, 7
    angle = 0.6911112
    fx = 0.5 * W / np.tan(0.5 * angle)
    g = torch.Generator().manual_seed(2)
    A = torch.randn(3, 3, generator=g)
    Q, _ = torch.linalg.qr(A)
    tm = torch.eye(4)
    tm[:3, :3] = Q
    tm[:3, 3] = torch.tensor([2.0, -1.5, 3.0])
    blender2opencv = np.array([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]])
    pose = np.array(tm.numpy().tolist()) @ blender2opencv
    c2w = torch.FloatTensor(pose)
    dirs = ru.get_ray_directions(H, W, [fx, fx])
    dirs = dirs / torch.norm(dirs, dim=-1, keepdim=True)
    ro, rd = ru.get_rays(dirs, c2w)
    save("blender_rays", dict(H=np.asarray(H), W=np.asarray(W), camera_angle_x=np.asarray(angle), fx=np.asarray(fx),
                              transform_matrix=tm, rays=torch.cat([ro, rd], 1)))


CASES = dict(blender_rays=case_blender_rays, upsample=case_upsample, sampler=case_sampler, field=case_field, alpha_mask=case_alpha_mask, env=case_env,
             shading_parts=case_shading_parts, e2e_small=case_e2e_small, e2e_full=case_e2e_full,
             e2e_full_steady=case_e2e_full_steady, e2e_g300=case_e2e_g300, e2e_variant=case_e2e_variant,
             e2e_full_eval=case_e2e_full_eval, e2e_g300_eval=case_e2e_g300_eval, e2e_g300_1k=case_e2e_g300_1k, trained_step=case_trained_step, init_step=case_init_step, trained_step_stats=case_trained_step_stats, init_step3_stats=case_init_step3_stats, simple_sampler=case_simple_sampler)

if __name__ == "__main__":
    torch.set_num_threads(8)
    names = sys.argv[1:] or list(CASES)
    for n in names:
        print("==", n)
        CASES[n]()
