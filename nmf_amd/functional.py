"""Autograd glue between torch tensors and the HIP kernels (nmf_amd.hip).  Each Function is a thin
forward/backward pair of C-ABI calls; there is no Python/torch re-implementation behind them."""
import math

import torch

from . import hip


class GradPass:
    """Gradient accumulators shared by every operator call of ONE forward/backward pass over a parameter set (the
    primary and the re-traced secondary rays query the same tables).  The per-call backward kernels
    accumulate into `bufs`; the pass's graph node (FieldGrads / SatBuild) converts them to parameter gradients once."""

    def __init__(self):
        self.bufs = None
        self.token_sent = False
        self.pending = []          # deferred field walks: (adjoint-set key, segment tuple), see defer_field_walk
        self.l1 = None             # (density factors, d_out) of a density_L1 term that joins the table gradients of this pass
        self.used = False          # a field query of the pass holds the token: FieldGrads.backward will run
        self.done = False

    def token_grad(self, like):
        """Gradient for the pass token: the node behind the token only has to be scheduled, so exactly one consumer
        hands it a (zero) gradient and the others return None -- no fill kernel per consumer, no accumulation adds."""
        if self.token_sent:
            return None
        self.token_sent = True
        return zero_token(like)


_ZERO = {}


def zero_token(like):
    """A 0-d zero for pass tokens and their gradients.  Token VALUES are never read (the nodes behind them only have to be
    scheduled), so every token aliases one cached per-device scalar instead of paying a fill launch each."""
    z = _ZERO.get(like.device)
    if z is None:
        z = _ZERO[like.device] = torch.zeros((), dtype=torch.float32, device=like.device)
    return z.detach()


class ParamGrads(torch.autograd.Function):
    """Pass node for a plain parameter list (BRDF MLP, material heads): the operator backwards of the pass accumulate into
    one zero-initialised flat buffer (grad_views) and this node returns its per-parameter views once."""

    @staticmethod
    def forward(ctx, holder, *params):
        ctx.holder, ctx.n = holder, len(params)
        return zero_token(params[0])

    @staticmethod
    def backward(ctx, _d_token):
        holder = ctx.holder
        if holder.bufs is None:
            return (None,) * (1 + ctx.n)
        views = holder.bufs[1]
        holder.bufs = None
        return (None,) + tuple(views)


def grad_views(holder, params):
    """fp32 gradient accumulators shaped like `params`, carved from one flat zero buffer owned by the pass."""
    if holder.bufs is None:
        sizes = [p.numel() for p in params]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=params[0].device)
        holder.bufs = (flat, [v.view(p.shape) for v, p in zip(flat.split(sizes), params)])
    return holder.bufs[1]


class FastPrivateAttrs:
    """torch.nn.Module.__setattr__ costs ~3 us (parameter / buffer / submodule bookkeeping); the operator modules rebind a
    few private caches and pass tokens ~40 times per step.  A private name that already lives in the instance dict is a
    plain attribute by construction, so it is rebound directly."""

    def __setattr__(self, name, value):
        d = self.__dict__
        if name[0] == "_" and name in d:
            d[name] = value
        else:
            super().__setattr__(name, value)


class PassMixin(FastPrivateAttrs):
    """begin_pass()/end_pass() bracket one forward/backward pass; every operator call in between shares one
    (GradPass, token) pair, i.e. one gradient node per parameter set (TensorNeRF.forward opens it at recursion 0)."""
    _pass = None
    _pass_open = False
    _memo = None        # per-pass memo of derived operands (stacked / detached weights): parameters cannot change inside a pass

    def begin_pass(self):
        self._pass, self._pass_open, self._memo = None, True, {}

    def end_pass(self):
        self._pass, self._pass_open, self._memo = None, False, None

    def _param_pass(self, params):
        if not (torch.is_grad_enabled() and any(p.requires_grad for p in params)):
            return None, None
        if self._pass_open and self._pass is not None:
            return self._pass
        holder = GradPass()
        token = ParamGrads.apply(holder, *params)
        if self._pass_open:
            self._pass = (holder, token)
        return holder, token


def defer_field_walk(holder, field, seg):
    """Queue one backward walk of the field (seg = (xyzt, sigma_feat, grad, d_sigma, d_sigma_feat, d_normal, d_app)).  The
    walks of one pass -- primary and re-traced samples -- accumulate into the same tables, so they are run together when
    the pass's FieldGrads node fires: one binning, and every brick both sample sets touch is flushed once."""
    key = tuple(t is not None for t in seg[3:])
    holder.pending.append((key, seg))
    if sum(1 for k, _ in holder.pending if k == key) >= hip.VM_MAX_SEGMENTS:
        flush_field_walks(holder, field)


def flush_field_walks(holder, field):
    if not holder.pending:
        return
    pending, holder.pending = holder.pending, []
    p, dpk, dlk, apl, ali, basis = field._tables()
    g_dpk, g_dlk, g_apl, g_ali, g_basis = field_grad_buffers(holder, p.grid, pending[0][1][0].device)
    for key in dict.fromkeys(k for k, _ in pending):
        segs = [sg for k, sg in pending if k == key]
        hip.vm_query_bwd_segments(p, segs, dpk, dlk, apl, ali, basis, g_dpk, g_dlk, g_apl, g_ali,
                                  g_basis if key[3] else None)


def field_grad_buffers(holder, G, dev):
    """packed gradient tables of the pass (one zero fill for all 13), created by the first backward that needs them"""
    if holder.bufs is None:
        shapes = [(G, G, 48)] * 3 + [(G, 32)] * 3 + [(G, G, 24)] * 3 + [(G, 24)] * 3 + [(24, 72)]
        sizes = [math.prod(sh) for sh in shapes]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        v = [t.view(sh) for t, sh in zip(flat.split(sizes), shapes)]
        holder.bufs = (v[0:3], v[3:6], v[6:9], v[9:12], v[12])
    return holder.bufs


class FieldGrads(torch.autograd.Function):
    """Graph node that owns the table gradients of a pass: forward hands out a scalar token every VMQuery of the pass
    takes as an input, so autograd runs this backward exactly once, after the last VMQuery backward."""

    @staticmethod
    def forward(ctx, holder, field, *params):
        ctx.holder, ctx.field = holder, field
        return zero_token(params[0])

    @staticmethod
    def backward(ctx, _d_token):
        holder, field = ctx.holder, ctx.field
        flush_field_walks(holder, field)
        holder.done = True
        l1, holder.l1 = holder.l1, None
        if holder.bufs is None:
            out = [None] * len(field._param_list())
            if l1 is not None:                       # no walk happened: the regulariser's own gradients
                out[:len(l1[0])] = hip.l1_mean_bwd(l1[0], l1[1])
            return (None, None) + tuple(out)
        g_dpk, g_dlk, g_apl, g_ali, g_basis = holder.bufs
        holder.bufs = None
        p = field._tables()[0]
        gp, gl = hip.vm_unpack_density_grad(p, g_dpk, g_dlk)
        if l1 is not None:
            # density_L1 of the same step (train.py:670-677): its sign(x)/n term is added to the unpacked gradients in one
            # launch ([G,G,16] / [G,16] are the factors' own storage order) instead of six autograd accumulation kernels
            hip.l1_mean_bwd(l1[0], l1[1], out=gp + gl)
        return (None, None) + tuple(field._grads_to_param_layout(gp, gl, g_apl, g_ali, g_basis))


class VMQuery(torch.autograd.Function):
    """sigma, sigma_feat, app, normal = field(xyzt).  Gradients flow to the 12 factor tables and basis_mat
    (incl. the second-order path through the normals) through the pass's FieldGrads node; sample positions carry
    no gradient, exactly like the reference (fields/tensor_base.py:109 detaches xyz)."""

    @staticmethod
    def forward(ctx, field, xyzt, want_app, want_normal, holder, token):
        p, dpk, dlk, apl, ali, basis = field._fwd_tables()
        sf, sg, gr, nr, ap, cf = hip.vm_query_fwd(p, xyzt, dpk, dlk, apl, ali, basis, want_density=True,
                                                  want_normal=want_normal, want_app=want_app, want_coef=False)
        ctx.field, ctx.holder = field, holder
        if holder is not None:
            holder.used = True
        ctx.flags = (want_app, want_normal)
        ctx.save_for_backward(xyzt, sf, gr, cf)
        ctx.mark_non_differentiable(sf)
        ctx.set_materialize_grads(False)
        outs = [sg, sf,
                ap if want_app else xyzt.new_empty((0, 24)),          # branch not evaluated: empty placeholder
                nr if want_normal else xyzt.new_empty((0, 3))]
        return tuple(outs)

    @staticmethod
    def backward(ctx, d_sigma, _d_sf, d_app, d_normal):
        field, holder = ctx.field, ctx.holder
        want_app, want_normal = ctx.flags
        xyzt, sf, gr, cf = ctx.saved_tensors
        d_sigma = d_sigma.contiguous() if d_sigma is not None else None
        d_app_c = d_app.contiguous() if (want_app and d_app is not None) else None
        d_nrm_c = d_normal.contiguous() if (want_normal and d_normal is not None) else None
        if d_sigma is not None or d_app_c is not None or d_nrm_c is not None:
            defer_field_walk(holder, field, (xyzt, sf, gr, d_sigma, None, d_nrm_c, d_app_c))
        return None, None, None, None, None, holder.token_grad(xyzt)


class VMQueryWeights(torch.autograd.Function):
    """VMQuery followed by Composite as ONE graph node (sigma only feeds raw2alpha, tensor_nerf.py:366): weights, sigma_feat,
    app, normals = composite(field(xyzt)).  One autograd node less per level in each direction (~12 us of host time each on a
    path where the GPU is waiting)."""

    @staticmethod
    def forward(ctx, field, xyzt, want_app, want_normal, holder, token, dist, offsets, b, scale):
        p, dpk, dlk, apl, ali, basis = field._fwd_tables()
        sf, sg, gr, nr, ap, cf = hip.vm_query_fwd(p, xyzt, dpk, dlk, apl, ali, basis, want_density=True,
                                                  want_normal=want_normal, want_app=want_app, want_coef=False)
        w, _acc = hip.composite_fwd(sg, dist, offsets, b, scale)
        ctx.meta = (field, holder, want_app, want_normal, b, scale)
        if holder is not None:
            holder.used = True
        ctx.save_for_backward(xyzt, sf, gr, sg, dist, w, offsets)
        ctx.mark_non_differentiable(sf)
        ctx.set_materialize_grads(False)
        return (w, sf, ap if want_app else xyzt.new_empty((0, 24)), nr if want_normal else xyzt.new_empty((0, 3)))

    @staticmethod
    def backward(ctx, d_w, _d_sf, d_app, d_normal):
        field, holder, want_app, want_normal, b, scale = ctx.meta
        xyzt, sf, gr, sg, dist, w, offsets = ctx.saved_tensors
        d_sigma = hip.composite_bwd(sg, dist, w, offsets, b, scale, d_w) if d_w is not None else None
        d_app_c = d_app.contiguous() if (want_app and d_app is not None) else None
        d_nrm_c = d_normal.contiguous() if (want_normal and d_normal is not None) else None
        if holder is not None and (d_sigma is not None or d_app_c is not None or d_nrm_c is not None):
            defer_field_walk(holder, field, (xyzt, sf, gr, d_sigma, None, d_nrm_c, d_app_c))
        return (None, None, None, None, None, holder.token_grad(xyzt) if holder is not None else None, None, None, None, None)


class VMAppQuery(torch.autograd.Function):
    """app = field appearance features at xyzt, nothing else (fields/tensoRF.py:402-405).  Used on the bounce rows only:
    in training the appearance branch is needed where a secondary ray starts, not at every kept sample."""

    @staticmethod
    def forward(ctx, field, xyzt, holder, token):
        p, dpk, dlk, apl, ali, basis = field._fwd_tables()
        ap = hip.vm_query_fwd(p, xyzt, dpk, dlk, apl, ali, basis, want_density=False, want_normal=False, want_app=True)[4]
        ctx.field, ctx.holder = field, holder
        ctx.save_for_backward(xyzt)
        return ap

    @staticmethod
    def backward(ctx, d_app):
        field, holder = ctx.field, ctx.holder
        (xyzt,) = ctx.saved_tensors
        defer_field_walk(holder, field, (xyzt, None, None, None, None, None, d_app.contiguous()))
        return None, None, None, holder.token_grad(xyzt)


class Composite(torch.autograd.Function):
    """weights = raw2alpha(sigma, dist * distance_scale) over ray segments (modules/tensor_nerf.py:19-35)."""

    @staticmethod
    def forward(ctx, sigma, dist, offsets, b, scale):
        w, _acc = hip.composite_fwd(sigma.contiguous(), dist, offsets, b, scale)
        ctx.save_for_backward(sigma, dist, w, offsets)
        ctx.meta = (b, scale)
        return w

    @staticmethod
    def backward(ctx, d_w):
        sigma, dist, w, offsets = ctx.saved_tensors
        b, scale = ctx.meta
        return hip.composite_bwd(sigma.contiguous(), dist, w, offsets, b, scale, d_w.contiguous()), None, None, None, None


class SegmentSum(torch.autograd.Function):
    """out[r] = sum of vals rows in segment r, added in index order (modules/row_mask_sum.py:15-22)."""

    @staticmethod
    def forward(ctx, vals, offsets, seg_id, n_seg):
        ctx.save_for_backward(seg_id)
        return hip.segment_sum(vals.contiguous(), None, offsets, n_seg)

    @staticmethod
    def backward(ctx, d_out):
        (seg_id,) = ctx.saved_tensors
        return d_out.index_select(0, seg_id.long()), None, None, None


def segment_sum(vals, offsets, seg_id, n_seg):
    if vals.dim() == 1:
        return SegmentSum.apply(vals[:, None], offsets, seg_id, n_seg)[:, 0]
    return SegmentSum.apply(vals, offsets, seg_id, n_seg)


class SatBuild(torch.autograd.Function):
    """Graph node for the cached summed-area table: sat = cumsum_W(cumsum_H(exp(brightness + mul*bg_mat)/1000))
    (modules/integral_equirect.py:431-433) and for the mip bias.  Hands out a scalar token; all lookups of the pass
    accumulate their adjoints into holder.bufs = (d_sat [H,W,4], d_pole [2,3], d_mipbias [1]) -- one zero fill for the three --
    and the two reverse prefix sums run once here."""

    @staticmethod
    def forward(ctx, holder, env, bg_mat, brightness, mul, mipbias):
        ctx.env, ctx.holder = env, holder
        return zero_token(bg_mat)

    @staticmethod
    def backward(ctx, _d_token):
        env, holder = ctx.env, ctx.holder
        if holder.bufs is None:
            return None, None, None, None, None, None
        d_sat, d_pole, d_mip = holder.bufs
        holder.bufs = None
        act, sat, pole = env._tables()
        sc = env._dev_scalars()
        d_bg = hip.sat_build_bwd(d_sat, env.bg_mat.detach(), act, d_pole, sc=sc)    # d_sat is consumed in place
        d_br = d_mul = None
        if ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
            d_pre = d_bg / sc[2]                             # adjoint of (brightness + mul * bg_mat)
            d_br = d_pre.sum(dtype=torch.float64)
            d_mul = (d_pre * env.bg_mat.detach().reshape(d_pre.shape)).sum(dtype=torch.float64)
        return None, None, d_bg.reshape(env.bg_mat.shape), d_br, d_mul, d_mip.to(torch.float64).reshape(())


def env_grad_buffers(holder, sat):
    if holder.bufs is None:
        H, W, _ = hip._sat_layout(sat)
        flat = torch.zeros(H * W * 4 + 8, dtype=torch.float32, device=sat.device)
        holder.bufs = (flat[: H * W * 4].view(H, W, 4), flat[H * W * 4: H * W * 4 + 6].view(2, 3),
                       flat[H * W * 4 + 6: H * W * 4 + 7])
    return holder.bufs


class EnvLookup(torch.autograd.Function):
    """IntegralEquirect.forward (modules/integral_equirect.py:409-504) on the cached SAT.  The adjoints of the table, of
    the pole rows and of the mip bias go to the pass's SatBuild node; the direction adjoint is returned here."""

    @staticmethod
    def forward(ctx, env, dirs, sa, holder, token):
        act, _sat, pole = env._tables()
        sat = env._lookup_table()                    # [H,W,4]
        dirs_c = dirs.contiguous()
        sa_c = sa.reshape(-1).contiguous()
        sc = env._dev_scalars()
        out = hip.sat_lookup_fwd(sat, dirs_c, sa_c, 0.0, pole, sc=sc)
        ctx.env, ctx.holder = env, holder
        ctx.save_for_backward(dirs_c, sa_c, sat, sc)
        return out

    @staticmethod
    def backward(ctx, d_out):
        holder = ctx.holder
        dirs, sa, sat, sc = ctx.saved_tensors
        want_tab = holder is not None and ctx.needs_input_grad[4]
        if want_tab:
            d_sat, d_pole, d_mip = env_grad_buffers(holder, sat)
        else:
            d_sat, d_mip = None, None
            d_pole = torch.zeros((2, 3), dtype=torch.float32, device=sat.device)
        d_dirs = hip.sat_lookup_bwd(sat, dirs, sa, 0.0, d_out.contiguous(), d_sat, d_pole, d_mip,
                                    want_dirs=ctx.needs_input_grad[1], sc=sc)
        return None, d_dirs, None, None, holder.token_grad(d_out) if want_tab else None


class BrdfMLP(torch.autograd.Function):
    """sigmoid(MLP([feat | ISH(half) | half | ISH(diff) | diff])[:3] + bias) in one fused MFMA kernel
    (modules/brdf.py:177-261).  Differentiable wrt the per-bounce-point feature rows and, through the pass's ParamGrads
    node, the six MLP tensors."""

    @staticmethod
    def forward(ctx, half_vec, diff_vec, feat_rows, rough_rows, row_of_ray, row_offsets, out_bias, holder, token,
                *weights):
        hv, dv = half_vec.contiguous(), diff_vec.contiguous()
        fr, rr = feat_rows.contiguous(), rough_rows.contiguous()
        ws = [w.detach().contiguous() for w in weights]
        out, mask = hip.brdf_mlp_fwd(ws, hv, dv, fr, rr, row_of_ray, out_bias, with_mask=True)
        ctx.save_for_backward(hv, dv, fr, rr, row_of_ray, row_offsets, out, mask, *ws)
        ctx.holder = holder
        return out

    @staticmethod
    def backward(ctx, d_out):
        hv, dv, fr, rr, row_of_ray, row_offsets, out, mask, *ws = ctx.saved_tensors
        grads = grad_views(ctx.holder, ws) if ctx.holder is not None else [torch.zeros_like(w) for w in ws]
        d_feat = hip.brdf_mlp_bwd(ws, hv, dv, fr, rr, row_of_ray, out, mask, d_out, grads)
        return (None, None, d_feat, None, None, None, None, None,
                ctx.holder.token_grad(d_out) if ctx.holder is not None else None) + (None,) * len(ws)


class MaterialHeads(torch.autograd.Function):
    """(albedo | tint | f0 | roughness) [M,11] from the app features (modules/render_modules.py:519-574).
    W [11,24] / b [11] are the four Linear layers stacked (cached per parameter version by the module); their
    gradients go through the pass's ParamGrads node in the same stacked layout."""

    @staticmethod
    def forward(ctx, feat, hp, W, b, holder, token):
        feat_c = feat.contiguous()
        out = hip.heads_fwd(feat_c, W, b, hp)
        ctx.save_for_backward(feat_c, W, b)
        ctx.hp, ctx.holder = hp, holder
        return out

    @staticmethod
    def backward(ctx, d_out):
        feat, W, b = ctx.saved_tensors
        gW, gb = grad_views(ctx.holder, [W, b]) if ctx.holder is not None else (torch.zeros_like(W), torch.zeros_like(b))
        d_feat = hip.heads_bwd(feat, W, b, ctx.hp, d_out, gW, gb)
        return d_feat, None, None, None, None, ctx.holder.token_grad(d_out) if ctx.holder is not None else None


class StackedHeadGrads(torch.autograd.Function):
    """ParamGrads for the stacked head weights: returns the row blocks of gW [11,24] / gb [11] to the four Linear layers."""

    @staticmethod
    def forward(ctx, holder, wd, bd, wt, bt, wf, bf, wr, br):
        ctx.holder = holder
        return zero_token(wd)

    @staticmethod
    def backward(ctx, _d_token):
        holder = ctx.holder
        if holder.bufs is None:
            return (None,) * 9
        gW, gb = holder.bufs[1]
        holder.bufs = None
        return (None, gW[0:3], gb[0:3], gW[3:6], gb[3:6], gW[6:9], gb[6:9], gW[9:11], gb[9:11])


class GgxRays(torch.autograd.Function):
    """Secondary rays from GGX visible-normal sampling (brdf_samplers/ggx.py:61-268 + models/microfacet.py:377-456).
    Differentiable wrt the row normals and roughness through L (and through rays[:, 3:6] = L, rays[:, :3] = x + 5e-3 L)."""

    @staticmethod
    def forward(ctx, V, N, r, x, off, cnt, sobol, row_of_ray, j_of_ray, row_off):
        r_shape = r.shape
        V, N, r, x, off = V.contiguous(), N.contiguous(), r.reshape(-1).contiguous(), x.contiguous(), off.contiguous()
        L, hl, dl, lpdf, mip, rays = hip.ggx_rays_fwd(V, N, r, x, off, cnt, sobol, row_of_ray, j_of_ray)
        ctx.save_for_backward(V, N, r, off, sobol, row_of_ray, j_of_ray, row_off)
        ctx.r_shape = r_shape
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(hl, dl, lpdf, mip)
        return L, hl, dl, lpdf, mip, rays

    @staticmethod
    def backward(ctx, dL, _hl, _dl, _lp, _mip, d_rays):
        V, N, r, off, sobol, row_of_ray, j_of_ray, row_off = ctx.saved_tensors
        if dL is None and d_rays is None:
            return (None,) * 10
        d_nr = hip.ggx_rays_bwd(V, N, r, off, sobol, row_of_ray, j_of_ray,
                                dL.contiguous() if dL is not None else None,
                                d_rays.contiguous() if d_rays is not None else None)
        rows = hip.segment_sum(d_nr, None, row_off, V.shape[0], lanes=8)
        return None, rows[:, 0:3], rows[:, 3].reshape(ctx.r_shape), None, None, None, None, None, None, None


class ShadeMix(torch.autograd.Function):
    """reflect_rgb rows = sum over the row's rays of (F Li brdf + (1-F) diffuse) / count (microfacet.py:595-613)."""

    @staticmethod
    def forward(ctx, V, f0, diff, cnt, row_of_ray, row_off, L, inc, brdf):
        V, f0, diff = V.contiguous(), f0.contiguous(), diff.contiguous()
        L, inc, brdf = L.contiguous(), inc.contiguous(), brdf.contiguous()
        contrib = hip.shade_mix_fwd(V, f0, diff, cnt, row_of_ray, L, inc, brdf)
        ctx.save_for_backward(V, f0, diff, cnt, row_of_ray, row_off, L, inc, brdf)
        return hip.segment_sum(contrib, None, row_off, V.shape[0], lanes=8)

    @staticmethod
    def backward(ctx, d_rows):
        V, f0, diff, cnt, row_of_ray, row_off, L, inc, brdf = ctx.saved_tensors
        d_inc, d_brdf, dL, d_fd = hip.shade_mix_bwd(V, f0, diff, cnt, row_of_ray, L, inc, brdf, d_rows.contiguous())
        rows = hip.segment_sum_wide(d_fd, 6, row_off, V.shape[0])
        return None, rows[:, 0:3], rows[:, 3:6], None, None, None, dL, d_inc, d_brdf


class BouncePrep(torch.autograd.Function):
    """Per bounce row, one pass (models/microfacet.py:297,304-316,352-361): V, facing N, clipped roughness, f0,
    diffuse = albedo * SH irradiance, noised feature, position.  The adjoint is written for all M samples through
    the inverse map (zeros elsewhere), so no index_add / zero fill is needed."""

    @staticmethod
    def forward(ctx, normals, app, heads, bidx, inv, xyzt, ray_id, rays, conv, feat_noise, anoise, min_rough, detach_n,
                row_inputs=False):
        normals, app, heads = normals.contiguous(), app.contiguous(), heads.contiguous()
        outs = hip.bounce_prep_fwd(bidx, normals, app, heads, xyzt, ray_id, rays, conv, feat_noise, anoise, min_rough,
                                   row_inputs)
        ctx.save_for_backward(normals, heads, inv, ray_id, rays, conv, bidx)
        ctx.cfg = (min_rough, detach_n, row_inputs)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(outs[0], outs[6])
        return outs

    @staticmethod
    def backward(ctx, _dV, dN, dr1, df0, ddiff, dfeat, _dxyz):
        normals, heads, inv, ray_id, rays, conv, bidx = ctx.saved_tensors
        min_rough, detach_n, row_inputs = ctx.cfg
        c = lambda t: None if t is None else t.contiguous()  # noqa: E731
        d_normals, d_heads, d_app = hip.bounce_prep_bwd(inv, normals, heads, ray_id, rays, conv, min_rough, detach_n,
                                                        dN, dr1, df0, ddiff, c(dfeat), bidx=bidx,
                                                        row_inputs=row_inputs)          # column slices read in place
        return (None if detach_n else d_normals, d_app, d_heads) + (None,) * 11


class RayCompose(torch.autograd.Function):
    """weights x row radiance -> pixels in one pass per ray: acc / rgb sums (modules/tensor_nerf.py:448-452), the
    orientation-loss term (:583-587), tonemap (modules/tonemap.py:34-55) and background blend (:658-659).
    Returns (rgb_map [B,3], acc [B], ori [B] or None)."""

    @staticmethod
    def forward(ctx, weight, refl_rows, normals, bg, inv, offsets, ray_id, rays, B, bg_per_ray, tonemap, noclip,
                want_ori):
        weight = weight.contiguous()
        refl = refl_rows.contiguous() if refl_rows is not None else None
        normals = normals.contiguous() if normals is not None else None
        bg = bg.contiguous()
        rgb_map, acc, rgb_lin, ori = hip.ray_compose_fwd(weight, refl, inv if refl is not None else None,
                                                         normals if want_ori else None, rays, offsets, B, bg,
                                                         bg_per_ray, tonemap, noclip, want_ori)
        ctx.save_for_backward(weight, refl, normals, bg, inv, ray_id, rays, rgb_lin, acc)
        ctx.cfg = (bg_per_ray, tonemap, noclip, want_ori)
        ctx.set_materialize_grads(False)
        return rgb_map, acc, ori

    @staticmethod
    def backward(ctx, d_rgb, d_acc, d_ori):
        weight, refl, normals, bg, inv, ray_id, rays, rgb_lin, acc = ctx.saved_tensors
        bg_per_ray, tonemap, noclip, want_ori = ctx.cfg
        c = lambda t: None if t is None else t.contiguous()  # noqa: E731
        d_rgb, d_acc, d_ori = c(d_rgb), c(d_acc), c(d_ori) if want_ori else None
        want_dn = d_ori is not None and ctx.needs_input_grad[2]
        d_weight, d_refl, d_normals = hip.ray_compose_bwd(weight, refl, inv if refl is not None else None,
                                                          normals if d_ori is not None else None, rays, ray_id, bg,
                                                          bg_per_ray, tonemap, noclip, rgb_lin, d_rgb, d_acc, d_ori,
                                                          want_dn)
        d_bg = None
        if ctx.needs_input_grad[3] and d_rgb is not None:
            d_bg = (1 - acc)[:, None] * d_rgb
            if not bg_per_ray:
                d_bg = d_bg.sum(0).reshape(bg.shape)
        return (d_weight, d_refl if ctx.needs_input_grad[1] else None, d_normals, d_bg) + (None,) * 9


class ShadeCompose(torch.autograd.Function):
    """ShadeMix followed by RayCompose as ONE graph node: the row radiance (microfacet.py:595-613) only feeds the per-ray
    sums of tensor_nerf.py:448-452, so the pair costs one autograd node per level and direction instead of two.
    Returns (rgb_map [B,3], acc [B], ori [B] or None, refl_rows [Mb,3] detached for the debug maps)."""

    @staticmethod
    def forward(ctx, weight, normals, bg, inv, offsets, ray_id, rays, B, bg_per_ray, tonemap, noclip, want_ori,
                V, f0, diff, cnt, row_of_ray, row_off, L, inc, brdf):
        V, f0, diff = V.contiguous(), f0.contiguous(), diff.contiguous()
        L, inc, brdf = L.contiguous(), inc.contiguous(), brdf.contiguous()
        contrib = hip.shade_mix_fwd(V, f0, diff, cnt, row_of_ray, L, inc, brdf)
        refl = hip.segment_sum(contrib, None, row_off, V.shape[0], lanes=8)
        weight = weight.contiguous()
        normals = normals.contiguous() if normals is not None else None
        bg = bg.contiguous()
        rgb_map, acc, rgb_lin, ori = hip.ray_compose_fwd(weight, refl, inv, normals if want_ori else None, rays, offsets, B,
                                                         bg, bg_per_ray, tonemap, noclip, want_ori)
        ctx.save_for_backward(weight, refl, normals, bg, inv, ray_id, rays, rgb_lin, acc, V, f0, diff, cnt, row_of_ray,
                              row_off, L, inc, brdf)
        ctx.cfg = (bg_per_ray, tonemap, noclip, want_ori)
        ctx.mark_non_differentiable(refl)
        ctx.set_materialize_grads(False)
        return rgb_map, acc, ori, refl

    @staticmethod
    def backward(ctx, d_rgb, d_acc, d_ori, _d_refl):
        (weight, refl, normals, bg, inv, ray_id, rays, rgb_lin, acc, V, f0, diff, cnt, row_of_ray, row_off, L, inc,
         brdf) = ctx.saved_tensors
        bg_per_ray, tonemap, noclip, want_ori = ctx.cfg
        c = lambda t: None if t is None else t.contiguous()  # noqa: E731
        d_rgb, d_acc, d_ori = c(d_rgb), c(d_acc), c(d_ori) if want_ori else None
        want_dn = d_ori is not None and ctx.needs_input_grad[1]
        d_weight, d_refl, d_normals = hip.ray_compose_bwd(weight, refl, inv, normals if d_ori is not None else None, rays,
                                                          ray_id, bg, bg_per_ray, tonemap, noclip, rgb_lin, d_rgb, d_acc, d_ori,
                                                          want_dn)
        d_bg = None
        if ctx.needs_input_grad[2] and d_rgb is not None:
            d_bg = (1 - acc)[:, None] * d_rgb
            if not bg_per_ray:
                d_bg = d_bg.sum(0).reshape(bg.shape)
        dV_rows = None
        if ctx.needs_input_grad[12]:
            # recursion level >= 1: the view direction of these rows is the direction the level above sampled
            # (bV = -viewdirs, models/microfacet.py:354) and the Fresnel term depends on it
            d_inc, d_brdf, dL, d_fd, dV = hip.shade_mix_bwd_view(V, f0, diff, cnt, row_of_ray, L, inc, brdf, d_refl)
            dV_rows = hip.segment_sum(dV, None, row_off, V.shape[0], lanes=8)
        else:
            d_inc, d_brdf, dL, d_fd = hip.shade_mix_bwd(V, f0, diff, cnt, row_of_ray, L, inc, brdf, d_refl)
        rows = hip.segment_sum_wide(d_fd, 6, row_off, V.shape[0])
        return (d_weight, d_normals, d_bg) + (None,) * 9 + (dV_rows, rows[:, 0:3], rows[:, 3:6], None, None, None, dL, d_inc,
                                                               d_brdf)


class BounceRays(torch.autograd.Function):
    """Sparse training / inference path: everything between "these samples spawn secondary rays" and the secondary rays with
    their BRDF weights as ONE graph node -- appearance features of the bounce rows (VMAppQuery), material heads
    (MaterialHeads), row preparation (BouncePrep, row_inputs), GGX rays (GgxRays) and the BRDF MLP (BrdfMLP), i.e. the same
    seven C-ABI calls forward and backward, minus four nodes of autograd bookkeeping per recursion level.  `c` carries the
    no-grad inputs; parameter gradients leave through the three pass tokens.
    Returns L, half_local, diff_local, lpdf, mipval, bounce_rays, brdf_weight, V_rows, f0_rows, diffuse_rows, N_rows."""

    @staticmethod
    def forward(ctx, normals, c, tok_field, tok_heads, tok_mlp, rays=None):
        """rays: the [b,6] ray rows when they are part of the graph (recursion level >= 1: origin | direction sampled by the
        level above).  The rows' view vector V = -direction then carries a gradient (models/microfacet.py:354 does not
        detach it): through the GGX sample L(V, N, r) here and through the Fresnel term in ShadeCompose."""
        normals = normals.contiguous()
        ctx.view_grad = rays is not None and rays.requires_grad
        p, dpk, dlk, apl, ali, basis = c.field._fwd_tables()
        app = hip.vm_query_fwd(p, c.xyz_rows, dpk, dlk, apl, ali, basis, want_density=False, want_normal=False,
                               want_app=True)[4]
        heads = hip.heads_fwd(app, c.head_W, c.head_b, c.head_hp)
        V, N, r1, f0, diff, feat, xyz = hip.bounce_prep_fwd(c.bidx, normals, app, heads, c.xyzt, c.ray_id, c.rays, c.conv,
                                                           c.feat_noise, c.anoise, c.min_rough, True)
        L, hl, dl, lpdf, mip, brays = hip.ggx_rays_fwd(V, N, r1, xyz, c.off, c.cnt, c.sobol, c.row_of_ray, c.j_of_ray)
        brdf, brdf_mask = hip.brdf_mlp_fwd(c.mlp_ws, hl, dl, feat, r1, c.row_of_ray, c.mlp_bias, with_mask=True)
        ctx.c = c
        ctx.save_for_backward(normals, app, heads, V, N, r1, feat, hl, dl, brdf, brdf_mask)
        N_out = N.detach().clone()
        if ctx.view_grad:
            ctx.mark_non_differentiable(hl, dl, lpdf, mip, N_out)
        else:
            ctx.mark_non_differentiable(hl, dl, lpdf, mip, V, N_out)
        ctx.set_materialize_grads(False)
        return L, hl, dl, lpdf, mip, brays, brdf, V, f0, diff, N_out

    @staticmethod
    def backward(ctx, dL, _hl, _dl, _lp, _mip, d_brays, d_brdf, dV_in, d_f0, d_diff, _dN):
        c = ctx.c
        normals, app, heads, V, N, r1, feat, hl, dl, brdf, brdf_mask = ctx.saved_tensors
        Mb = V.shape[0]
        cc = lambda t: None if t is None else t.contiguous()  # noqa: E731
        d_feat = None
        if d_brdf is not None:
            grads = grad_views(c.mlp_holder, c.mlp_ws) if c.mlp_holder is not None else [torch.zeros_like(w) for w in c.mlp_ws]
            d_feat = hip.brdf_mlp_bwd(c.mlp_ws, hl, dl, feat, r1, c.row_of_ray, brdf, brdf_mask, cc(d_brdf), grads)
        dN = dr1 = dV_rows = None
        if dL is not None or d_brays is not None:
            if ctx.view_grad:
                d_nrv = hip.ggx_rays_bwd_view(V, N, r1, c.off, c.sobol, c.row_of_ray, c.j_of_ray, cc(dL), cc(d_brays))
                rows7 = hip.segment_sum_wide(d_nrv, 7, c.row_off, Mb)
                dN, dr1, dV_rows = rows7[:, 0:3], rows7[:, 3], rows7[:, 4:7]
            else:
                d_nr = hip.ggx_rays_bwd(V, N, r1, c.off, c.sobol, c.row_of_ray, c.j_of_ray, cc(dL), cc(d_brays))
                rows4 = hip.segment_sum(d_nr, None, c.row_off, Mb, lanes=8)
                dN, dr1 = rows4[:, 0:3], rows4[:, 3]
        d_rays = None
        if ctx.view_grad:
            if dV_in is not None:
                dV_rows = dV_in if dV_rows is None else dV_rows + dV_in
            if dV_rows is not None:           # V_row = -direction of the row's ray: scatter the rows back onto their rays
                ray_of_row = torch.index_select(c.ray_id, 0, c.bidx.long()).long()
                d_rays = torch.zeros_like(c.rays)
                d_rays[:, 3:6].index_add_(0, ray_of_row, -dV_rows)
        d_normals, d_heads, d_app = hip.bounce_prep_bwd(c.inv, normals, heads, c.ray_id, c.rays, c.conv, c.min_rough,
                                                        c.detach_n, dN, dr1, d_f0, d_diff, d_feat, bidx=c.bidx,
                                                        row_inputs=True)
        if c.head_holder is not None:
            gW, gb = grad_views(c.head_holder, [c.head_W, c.head_b])
        else:
            gW, gb = torch.zeros_like(c.head_W), torch.zeros_like(c.head_b)
        d_app.add_(hip.heads_bwd(app, c.head_W, c.head_b, c.head_hp, d_heads, gW, gb))
        tf = None
        if c.field_holder is not None:
            defer_field_walk(c.field_holder, c.field, (c.xyz_rows, None, None, None, None, None, d_app))
            tf = c.field_holder.token_grad(d_app)
        th = c.head_holder.token_grad(d_app) if c.head_holder is not None else None
        tm = c.mlp_holder.token_grad(d_app) if (c.mlp_holder is not None and d_brdf is not None) else None
        return (None if c.detach_n else d_normals), None, tf, th, tm, d_rays


def brdf_mlp(half_vec, diff_vec, feat_rows, rough_rows, row_of_ray, row_offsets, out_bias, weights, owner=None):
    """BrdfMLP with its gradient pass: `owner` (a PassMixin module) shares one ParamGrads node per forward/backward pass;
    without an owner the call gets a node of its own."""
    if owner is not None:
        holder, token = owner._param_pass(weights)
    elif torch.is_grad_enabled() and any(w.requires_grad for w in weights):
        holder = GradPass()
        token = ParamGrads.apply(holder, *weights)
    else:
        holder, token = None, None
    return BrdfMLP.apply(half_vec, diff_vec, feat_rows, rough_rows, row_of_ray, row_offsets, out_bias, holder, token,
                         *weights)


def material_heads(feat, hp, params, owner=None, stacked=None):
    """MaterialHeads over the eight head tensors (wd, bd, wt, bt, wf, bf, wr, br); `stacked` = cached (W [11,24], b [11])."""
    if stacked is None:
        stacked = (torch.cat([p.detach() for p in params[0::2]], 0).contiguous(),
                   torch.cat([p.detach() for p in params[1::2]], 0).contiguous())
    holder, token = None, None
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        if owner is not None and owner._pass_open and owner._pass is not None:
            holder, token = owner._pass
        else:
            holder = GradPass()
            token = StackedHeadGrads.apply(holder, *params)
            if owner is not None and owner._pass_open:
                owner._pass = (holder, token)
    return MaterialHeads.apply(feat, hp, stacked[0], stacked[1], holder, token)


class L1Mean(torch.autograd.Function):
    """sum_i mean(|x_i|) over a list of dense tensors in one launch (fields/tensoRF.py:332-340).  `holder`: the GradPass
    of a field pass whose FieldGrads node is part of the same backward (the trainer's total loss); the gradient is then
    handed to that node, which adds it to the table gradients it emits (see FieldGrads.backward)."""

    @staticmethod
    def forward(ctx, holder, *tensors):
        ctx.holder = holder
        ctx.save_for_backward(*tensors)
        return hip.l1_mean_fwd([t.detach() for t in tensors])

    @staticmethod
    def backward(ctx, d_out):
        h = ctx.holder
        if h is not None and h.used and not h.done and h.l1 is None:
            h.l1 = ([t.detach() for t in ctx.saved_tensors], d_out.contiguous())
            return (None,) * (1 + len(ctx.saved_tensors))
        return (None,) + tuple(hip.l1_mean_bwd(list(ctx.saved_tensors), d_out.contiguous()))


class LossMix(torch.autograd.Function):
    """scale * sum_i w_i * sum(x_i): the loss assembly of train.py:640-677 (photometric term, orientation and prediction
    regularisers as per-ray vectors, density L1) in one launch per direction instead of ~20 scalar kernels."""

    @staticmethod
    def forward(ctx, scale, weights, *tensors):
        ctx.cfg = (float(scale), tuple(float(w) for w in weights), [t.shape for t in tensors])
        return hip.loss_mix_fwd([t.detach().contiguous() for t in tensors], ctx.cfg[1], ctx.cfg[0])

    @staticmethod
    def backward(ctx, d_out):
        scale, weights, shapes = ctx.cfg
        return (None, None) + tuple(hip.loss_mix_bwd(shapes, weights, scale, d_out.contiguous()))


class SquaredError(torch.autograd.Function):
    """sum (clip(pred,0,1) - clip(gt,0,1))^2 (train.py:598-601); differentiable wrt pred."""

    @staticmethod
    def forward(ctx, pred, gt):
        pred, gt = pred.contiguous(), gt.contiguous()
        ctx.save_for_backward(pred, gt)
        return hip.sqerr_fwd(pred, gt)

    @staticmethod
    def backward(ctx, d_out):
        pred, gt = ctx.saved_tensors
        return hip.sqerr_bwd(pred, gt, d_out.contiguous()), None
