"""Microfacet shading model -- host-side mirror of the reference's models/microfacet.py (Microfacet :12-673)
for diffuse_mixing_mode='fresnel', no_emitters=True, russian_roulette=False.

Works on the compact sample list (weights [M] + ray offsets) instead of the dense [rays x N] matrices, and on
a compact secondary-ray list (row_of_ray, j_of_ray) instead of the padded [bounce points x m] ray_mask."""
import math
import types

import torch

from .. import hip
from ..functional import BouncePrep, BounceRays, GgxRays, ShadeMix, FastPrivateAttrs
from ..modules import sh
from ..brdf_samplers.ggx import mat3T_vec, normalize  # noqa: F401
from ..controllers import RetraceController


class Microfacet(FastPrivateAttrs, torch.nn.Module):
    def __init__(self, app_dim, diffuse_module, brdf, brdf_sampler, anoise, max_brdf_rays, target_num_samples,
                 russian_roulette, percent_bright, cold_start_bg_iters, detach_N_iters, min_rough_start=0,
                 min_rough_decay=1, start_std=0, std_decay=1, std_decay_interval=10, conserve_energy=True,
                 no_emitters=True, diffuse_mixing_mode="lambda", visibility_module=None, max_retrace_rays=(),
                 bright_sampler=None, freeze=False, rays_per_ray=512, test_rays_per_ray=512):
        super().__init__()
        if diffuse_mixing_mode != "fresnel" or not no_emitters or russian_roulette or visibility_module is not None \
                or start_std != 0 or percent_bright != 0:
            raise NotImplementedError("implements the microfacet_tensorf2.yaml:52-72 configuration")
        self.diffuse_module = diffuse_module(in_channels=app_dim)
        self.brdf = brdf(in_channels=app_dim)
        self.brdf_sampler = brdf_sampler(max_samples=1024)
        self.freeze = freeze
        self.needs_normals = lambda x: True
        self.conserve_energy = conserve_energy
        self.brdf.init_val = 0.5 if conserve_energy else 0.25
        self.min_rough, self.min_rough_decay = min_rough_start, min_rough_decay
        self.std, self.std_decay, self.std_decay_interval = start_std, std_decay, std_decay_interval
        self.anoise = anoise
        self.target_num_samples = list(target_num_samples)
        self.max_brdf_rays = list(max_brdf_rays)
        self.start_max_retrace_rays = list(max_retrace_rays)
        # models/microfacet.py:236-269 (reset_counter / update_n_samples): host-side feedback loop, nmf_amd/controllers.py
        self._retrace = RetraceController(max_retrace_rays, self.target_num_samples, self.max_brdf_rays)
        self.detach_N_iters = detach_N_iters
        self.detach_N = True
        self.rays_per_ray, self.test_rays_per_ray = rays_per_ray, test_rays_per_ray
        self.outputs = {"diffuse": 3, "roughness": 1, "tint": 3, "spec": 3, "albedo": 3}

    # ---- controllers / bookkeeping (models/microfacet.py:79-121,236-269) --------------------------------
    def calibrate(self, args, xyz, feat, bg_brightness, save_config=True):
        self.diffuse_module.calibrate(bg_brightness, self.conserve_energy, xyz, None, feat)
        self.brdf.calibrate(feat, bg_brightness)
        return args

    def get_optparam_groups(self, lr_scale=1):
        if self.freeze:
            return []
        return [{"params": self.diffuse_module.parameters(), "lr": self.diffuse_module.lr * lr_scale},
                {"params": self.brdf.parameters(), "lr": self.brdf.lr * lr_scale}]

    def check_schedule(self, iter, batch_mul, **kwargs):
        if iter % 10 == 0:
            self.min_rough *= self.min_rough_decay
        if iter > batch_mul * self.detach_N_iters:
            self.detach_N = False
        if iter % self.std_decay_interval == 0:
            self.std *= self.std_decay
        return False

    def reset_counter(self):
        self._retrace.reset()

    def update_n_samples(self, n_samples):
        self._retrace.update(n_samples)

    @property
    def max_retrace_rays(self):
        return self._retrace.max_retrace_rays

    @max_retrace_rays.setter
    def max_retrace_rays(self, v):
        self._retrace.max_retrace_rays = list(v)

    @property
    def mean_ratios(self):
        return self._retrace.mean_ratios

    @property
    def ratio_list(self):
        return self._retrace.ratio_list

    # ---- shading ----------------------------------------------------------------------------------------
    def forward(self, xyzs, xyzs_normed, app_features, viewdirs, normals, weights, app_mask, B, render_reflection,
                bg_module=None, is_train=False, recur=0, eps=torch.finfo(torch.float32).eps, noise=None):
        """Reference interface (models/microfacet.py:271-673): xyzs [M,4], app_features [M,C], viewdirs [M,3], normals
        [M,3] of the kept samples, the dense weights [b,N] and mask app_mask [b,N] they were taken from, and
        render_reflection(rays [R,6], mipval [R], retrace=bool) -> (radiance [R,3], visibility).  Returns rgb [M,3] and the
        debug dict {diffuse, tint, roughness, spec, albedo}.  The dense inputs are converted to the compact sample list and
        run through shade_compact(), i.e. the same HIP kernels as the hot path (TensorNeRF.forward calls shade_compact
        directly and never forms the [M,3] radiance or the padded ray mask)."""
        from ..noise import DeviceNoise
        from ..samplers.alphagrid import Samples
        dev = xyzs.device
        b, N = app_mask.shape
        ri, rj = torch.where(app_mask)
        M = int(ri.shape[0])
        offsets = torch.zeros(b + 1, dtype=torch.int64, device=dev)
        offsets[1:] = torch.cumsum(app_mask.sum(dim=1), 0)
        w = (weights[app_mask] if weights.dim() == 2 else weights.reshape(-1)).contiguous()
        rays = torch.zeros(b, 6, device=dev)                    # shade_compact reads a sample's view direction per RAY
        rays[ri, 3:6] = viewdirs.detach().float()
        S = Samples(xyzs.float().contiguous(), ri.int().contiguous(), rj.int().contiguous(), None, None, offsets, None, M, b, N,
                    rays=rays)
        if noise is None:
            noise = self.__dict__.get("_own_noise")
            if noise is None:
                noise = self.__dict__["_own_noise"] = DeviceNoise(dev, seed=20211200)

        def rr(brays, mipval, retrace):
            out = render_reflection(brays, mipval, retrace=retrace)
            return out[0] if isinstance(out, tuple) else out

        sh_ = self.shade_compact(S, app_features.float().contiguous(), normals, w, rr, bg_module, is_train, recur, noise)
        return sh_.rgb(), sh_.debug()

    def shade_compact(self, samples, app_features, normals, weights, render_reflection, bg_module, is_train, recur,
                      noise, app_fn=None):
        """samples: samplers.alphagrid.Samples; app_features [M,24] or None; normals [M,3]; weights [M].
        Returns a Shaded record: radiance per BOUNCE ROW (samples that spawned secondary rays; every other sample
        has zero radiance, models/microfacet.py:596-613) + the inverse map, with the debug maps computed on demand.

        app_features=None selects the sparse evaluation: the appearance branch of the field (app_fn(xyzt_rows) -> [Mb,24]),
        its noise and the material heads are evaluated on the bounce rows only -- 5-20 % of the samples -- which is all
        the radiance depends on; the per-sample outputs of the reference (albedo / roughness maps ...) are then produced
        on demand by Shaded.debug()."""
        M = samples.M
        dev = normals.device
        sparse = app_features is None
        feat_noise, heads, deferred = None, None, None
        if not sparse:
            feat_noise = noise.normal((M, app_features.shape[1]))                                   # :297
            if feat_noise is not None:
                feat_noise = feat_noise.contiguous()
        else:
            deferred = noise.normal_deferred((M, 24))                                               # :297, rows drawn later
        noise.skip("randn", (M, 3))
        noise.skip("randn", (M, 2))
        if not sparse:
            heads = self.diffuse_module.heads(app_features)                                          # :299
        noise.skip("rand", (5000,))                                                                 # :304-315
        noise.skip("rand", (5000,))
        _, conv = bg_module.get_spherical_harmonics(100)
        conv = conv.reshape(9, 3)

        # ---- how many secondary rays per sample (:327-333, pt_selectors.py)
        w_det = weights.detach().contiguous()
        if recur == 0:
            rpr = self.rays_per_ray if is_train else self.test_rays_per_ray
            counts = hip.select_bounces(w_det, noise.uniform((M,)).contiguous(), 0, float(rpr))
        else:
            if hasattr(noise, "select_dense_parts"):        # device noise: the normaliser in one launch (nmf_select_total)
                u, extra = noise.select_dense_parts(samples.b, samples.N, M)
                total = hip.select_total(w_det, u.contiguous(), extra)
            else:
                u, u_total = noise.select_dense(samples.b, samples.N, samples.ray_id, samples.step_id)
                total = (w_det.sum(dtype=torch.float64) + 1e-3 * u_total).float().clip(min=1e-3)
            Nbudget = self.max_brdf_rays[recur] - M
            if Nbudget > 0:
                counts = hip.select_bounces(w_det, u.contiguous(), 1, float(Nbudget), 1.0, total)
            else:
                counts = hip.select_bounces(w_det, u.contiguous(), 1, float(self.max_brdf_rays[recur]), 0.5, total)
        pins = getattr(noise, "pins", None)          # tests: replayed bookkeeping of a reference run (noise.Pins)
        trace = pins.trace if pins is not None else None
        if pins is not None:
            counts = pins.counts_for(recur, counts)
        bidx, row_off, cnt32, inv, tot = hip.bounce_index(counts)                                    # :333-350
        R, Mb = (int(v) for v in tot.cpu())
        out = Shaded(self, samples, heads, normals, conv, w_det, inv, M, app_fn)
        if R == 0:
            return out
        bidx, row_off, cnt32 = bidx[:Mb], row_off[:Mb + 1], cnt32[:Mb]
        row_of_ray, j_of_ray = hip.expand_segments(row_off, Mb, R)            # = torch.where(ray_mask)
        off = noise.uniform((Mb, 1, 2)).reshape(Mb, 2)                                              # base.py:18
        if sparse:      # appearance, its noise, the heads, GGX rays and BRDF weights of the bounce rows: one graph node
            c = types.SimpleNamespace()
            c.field = field = app_fn.__self__
            c.xyz_rows = torch.index_select(samples.xyzt, 0, bidx)
            c.field_holder, tok_field = field._pass_token()
            c.head_hp, c.head_W, c.head_b, c.head_holder, tok_heads = self.diffuse_module.head_pass()            # :299
            c.mlp_ws, c.mlp_bias, c.mlp_holder, tok_mlp = self.brdf.mlp_pass()
            c.bidx, c.inv, c.xyzt, c.ray_id, c.rays, c.conv = bidx, inv, samples.xyzt, samples.ray_id, samples.rays, conv
            c.feat_noise = noise.rows(deferred, bidx)                                                # :297
            c.anoise, c.min_rough = float(self.anoise), float(self.min_rough) if is_train else -1e30
            c.detach_n = bool(self.detach_N)
            c.off, c.cnt, c.sobol = off.contiguous(), cnt32, self.brdf_sampler.angs
            c.row_of_ray, c.j_of_ray, c.row_off = row_of_ray, j_of_ray, row_off
            # recursion level >= 1: the ray rows are outputs of the level above (origin x + 5e-3 L | direction L) and their
            # direction enters the shading as the view vector bV = -viewdirs (:354, not detached)
            graph_rays = samples.rays if (recur > 0 and torch.is_grad_enabled() and samples.rays.requires_grad) else None
            L, halfvec, diffvec, lpdf, mipval, bounce_rays, brdf_weight, bV, f0, diffuse, bN = BounceRays.apply(
                normals, c, tok_field, tok_heads, tok_mlp, graph_rays)                                # :352-472
        else:
            bV, bN, r1, f0, diffuse, feat, xyz = BouncePrep.apply(
                normals, app_features, heads, bidx, inv, samples.xyzt, samples.ray_id, samples.rays, conv, feat_noise,
                float(self.anoise), float(self.min_rough) if is_train else -1e30, bool(self.detach_N), False)   # :352-361
            L, halfvec, diffvec, lpdf, mipval, bounce_rays = GgxRays.apply(                         # :367-456
                bV, bN, r1, xyz, off, cnt32, self.brdf_sampler.angs, row_of_ray, j_of_ray, row_off)
            brdf_weight = self.brdf.forward_compact(halfvec, diffvec, feat, r1, row_of_ray, row_off)
        if trace is not None:
            trace.update({f"L{recur}": L, f"mipval{recur}": mipval, f"brdf_weight{recur}": brdf_weight,
                          f"halfvec{recur}": halfvec, f"diffvec{recur}": diffvec, f"lpdf{recur}": lpdf})
        if len(self.max_retrace_rays) > recur:                                                      # :475-559
            num_retrace = min(R, self.max_retrace_rays[recur])
            pinned = pins is not None and recur in pins.retrace_order
            if num_retrace >= R and not (pins is not None and pins.sorts(recur)):
                # steady state: every secondary ray is re-traced.  The reference still argsorts the
                # scores, which only permutes the rays before they meet their i.i.d. jitter rows; the draw is
                # consumed for stream parity and the identity order is used (same distribution, no 250 k-key sort).
                noise.skip("rand", (R,))
                incoming = render_reflection(bounce_rays, mipval, True)
            else:
                with torch.no_grad():
                    w_rows = torch.index_select(w_det, 0, bidx.long())
                    cc = hip.retrace_scores(brdf_weight.detach().contiguous(), bV, bN.detach().contiguous(),
                                            lpdf.contiguous(), w_rows, cnt32, row_of_ray)           # :480-500
                    cc = cc / cc.sum() * num_retrace
                    cc = cc + noise.uniform((R,))
                    order = hip.argsort_f32(cc.contiguous()).long()                                  # :522
                    if trace is not None:
                        trace[f"retrace_order_own{recur}"] = order
                    if pinned:
                        order = pins.retrace_order[recur].to(dev)
                    cut = max(R - num_retrace, 0)
                    idx_re, idx_no = order[cut:], order[:cut]
                    if trace is not None:
                        trace.update({f"retrace_score{recur}": cc, f"retrace_order{recur}": order,
                                      f"retrace_idx{recur}": idx_re})
                incoming = torch.zeros((R, 3), device=dev)
                if idx_re.shape[0] > 0:
                    inc = render_reflection(bounce_rays[idx_re], mipval[idx_re], True)
                    incoming = incoming.index_put((idx_re,), inc)
                if idx_no.shape[0] > 0:
                    inc = render_reflection(bounce_rays[idx_no], mipval[idx_no], False)
                    incoming = incoming.index_put((idx_no,), inc)
        else:
            incoming = render_reflection(bounce_rays, mipval, False)
        if trace is not None:
            trace[f"incoming{recur}"] = incoming
        # :596-613 -- evaluated together with the per-ray sums (functional.ShadeCompose) by TensorNeRF, or on first read
        out.mix_args = (bV, f0, diffuse, cnt32, row_of_ray, row_off, L, incoming, brdf_weight)
        out.rows = (bidx, row_off, cnt32, row_of_ray, incoming.detach(), brdf_weight.detach())
        return out


class Shaded:
    """Result of Microfacet.shade_compact: refl_rows [Mb,3] (radiance of the samples that spawned secondary rays,
    None when there are none), inv [M] (row of each sample or -1) and, on demand, the reference's per-sample
    outputs: rgb() [M,3] and debug() = {diffuse, tint, roughness, spec, albedo} (models/microfacet.py:642-673).
    Training only reads refl_rows; the on-demand tensors are plain torch expressions of the graph tensors, so a
    caller that puts them into a loss still gets their gradients."""

    def __init__(self, model, samples, heads, normals, conv, w_det, inv, M, app_fn=None):
        self.model, self.samples, self.heads, self.normals, self.conv = model, samples, heads, normals, conv
        self.w_det, self.inv, self.M, self.app_fn = w_det, inv, M, app_fn
        self._refl_rows = None
        self.mix_args = None          # inputs of the Fresnel mix when the row radiance has not been formed yet
        self.rows = None
        self._debug = None

    @property
    def refl_rows(self):
        if self._refl_rows is None and self.mix_args is not None:
            self._refl_rows = ShadeMix.apply(*self.mix_args)
        return self._refl_rows

    @refl_rows.setter
    def refl_rows(self, value):
        self._refl_rows = value

    def rgb(self):
        z = torch.zeros((self.M, 3), device=self.normals.device)
        if self.refl_rows is None:
            return z
        return z.index_put((self.rows[0].long(),), self.refl_rows)

    def debug(self):
        if self._debug is None:
            if self.heads is None:          # sparse evaluation: the per-sample material maps were not needed so far
                self.heads = self.model.diffuse_module.heads(self.app_fn(self.samples.xyzt))
            S, h, n = self.samples, self.heads, self.normals
            albedo, f0, r1 = h[:, 0:3], h[:, 6:9], h[:, 9:10]
            viewdirs = torch.index_select(S.rays[:, 3:6], 0, S.ray_id.long())
            with torch.no_grad():
                E = (self.conv.reshape(1, -1, 3) * sh.eval_sh_bases(9, n.detach()).reshape(self.M, -1, 1)).sum(dim=1)
            diffuse = albedo * E                                                                    # :316
            z = torch.zeros_like(diffuse)
            spec, brdf_rgb = z, z
            if self.rows is not None:
                bidx, row_off, cnt32, row_of_ray, incoming, brdf_weight = self.rows
                with torch.no_grad():
                    ec = cnt32.float().clip(min=1)[row_of_ray.long()][:, None]
                    Mb = bidx.shape[0]
                    spec = z.index_put((bidx.long(),), hip.segment_sum((incoming / ec).contiguous(), None, row_off, Mb))
                    brdf_rgb = z.index_put((bidx.long(),),
                                           hip.segment_sum((brdf_weight / ec).contiguous(), None, row_off, Mb))
            cos_t = (-viewdirs * n).sum(dim=-1, keepdim=True).abs()                                 # :642
            Fr = f0 + (1 - f0) * (1 - cos_t).clip(min=0, max=1) ** 5
            self._debug = dict(diffuse=(1 - Fr) * diffuse, tint=Fr * brdf_rgb, roughness=r1, spec=spec, albedo=albedo)
        return self._debug
