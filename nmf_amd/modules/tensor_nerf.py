"""Scene module -- host-side mirror of the reference's modules/tensor_nerf.py (TensorNeRF :38-674):
sample -> density -> weights -> appearance + normals -> microfacet shading (recursive) -> composite ->
tonemap, with the same constructor keywords, __call__ signature, returned image / statistics keys and
state_dict layout.  Every stage runs on the compact sample list produced by the HIP sampler."""
import torch

from .. import hip
from ..functional import Composite, RayCompose, ShadeCompose, segment_sum, FastPrivateAttrs
from ..noise import DeviceNoise
from .tonemap import SRGBTonemap


class TensorNeRF(FastPrivateAttrs, torch.nn.Module):
    _warned_fallback = False            # the one-time warning of a training forward that left the fused pass (process-wide)

    def __init__(self, rf, model, aabb, near_far, sampler, tonemap=None, bg_module=None, normal_module=None,
                 alphaMask=None, infinity_border=False, recur_stepmul=1, recur_alpha_thres=1e-3, detach_inter=False,
                 hdr=False, bg_noise=0, bg_noise_decay=0.999, use_predicted_normals=True, orient_world_normals=False,
                 align_pred_norms=True, eval_batch_size=512, geonorm_iters=-1, geonorm_interp_iters=1, lr_scale=1,
                 contraction="AABB", **kwargs):
        super().__init__()
        if normal_module is not None or hdr or detach_inter or geonorm_iters > 0:
            raise NotImplementedError("implements the microfacet_tensorf2.yaml:1-27 configuration")
        self.rf = rf(aabb=aabb)
        self.normal_module = None
        self.sampler = sampler(near_far=near_far, aabb=aabb)
        self.model = model(self.rf.app_dim)
        self.bg_module = bg_module
        self.tonemap = SRGBTonemap() if tonemap is None else tonemap
        if not isinstance(self.tonemap, SRGBTonemap):
            raise NotImplementedError("nmf_ray_compose implements modules.tonemap.SRGBTonemap (microfacet_tensorf2.yaml:26)")
        self.lr_scale = lr_scale
        self.hdr = hdr
        self.bg_noise, self.bg_noise_decay = bg_noise, bg_noise_decay
        self.recur_alpha_thres = recur_alpha_thres
        self.eval_batch_size = eval_batch_size
        self.recur_stepmul = recur_stepmul
        self.use_predicted_normals = False
        self.align_pred_norms = use_predicted_normals | align_pred_norms
        self.orient_world_normals = orient_world_normals | (not self.align_pred_norms)
        self._noise = None
        # One training chunk as ONE autograd node over the C++ pass (nmf_amd/fast_step.py: ChunkPass over csrc/step_core.inc) whenever
        # the call is the training forward of microfacet_tensorf2.yaml; False: always the operator graph of nmf_amd/functional.py.
        self.fused_training_pass = True
        # The regulariser statistics envmap_reg / brdf_reg / diffuse_reg (modules/tensor_nerf.py:600-649) need the material heads on
        # EVERY sample; the fused pass evaluates them on the bounce rows only and reports 0 for the three (their weights are 0 in
        # microfacet_tensorf2.yaml:208-214).  True: the operator graph, which evaluates them (differentiably) on first read.
        self.regulariser_stats = False
        self.fused_eval_pass = True                 # renderer.render_images: evaluation chunks as one C++ call each
        self._fused_pass = None
        self.operator_graph_forwards = 0        # training forwards (recur 0, CUDA, grad mode) that built the operator graph
        self.regulariser_weights = None         # optional {stat name: weight} of the caller's loss: a fused forward refuses a non-zero one

    def get_device(self):
        return self.rf.units.device

    def get_optparam_groups(self):
        g = self.rf.get_optparam_groups(self.lr_scale) + self.model.get_optparam_groups(self.lr_scale)
        if isinstance(self.bg_module, torch.nn.Module):
            g += self.bg_module.get_optparam_groups(self.lr_scale)
        return g

    def save(self, path, config):
        config["use_predicted_normals"] = self.use_predicted_normals
        sd = {k: (v.contiguous() if v.dim() == 4 else v) for k, v in self.state_dict().items()}
        torch.save({"config": config, "state_dict": sd}, path)

    @staticmethod
    def load(ckpt, config=None, near_far=None, device="cuda", **kwargs):
        """modules/tensor_nerf.py:136-175: rebuild the scene module from a checkpoint written by save() -- {"config":
        arch config (the `model.arch` node, `_target_` strings of the reference), "state_dict": ...} -- with the grid size
        and AABB stored in the state_dict, the calibrated biases taken from the checkpoint's config, then load the weights.
        Checkpoints written by the reference itself (config = an OmegaConf DictConfig) are read by
        nmf_amd.checkpoint.load_checkpoint without omegaconf and without executing pickled code."""
        from ..checkpoint import load_checkpoint, to_plain
        from ..yaml_config import instantiate
        if not isinstance(ckpt, dict):
            ckpt = load_checkpoint(ckpt)
        else:
            ckpt = dict(ckpt, config=to_plain(ckpt["config"]))
        saved = ckpt["config"]
        if config is not None:
            config = dict(config)
            config["model"]["brdf"]["bias"] = saved["model"]["brdf"]["bias"]
            for k in ("diffuse_bias", "roughness_bias"):
                config["model"]["diffuse_module"][k] = saved["model"]["diffuse_module"][k]
        else:
            config = saved
        sd = dict(ckpt["state_dict"])
        aabb = sd["rf.aabb"]
        near_far = near_far if near_far is not None else [1, 6]
        grid_size = sd["rf.grid_size"].tolist() if "rf.grid_size" in sd else [300, 300, 300]
        cfg = {k: v for k, v in config.items() if k != "use_predicted_normals"}
        cfg["rf"] = dict(cfg["rf"], grid_size=grid_size)
        nerf = instantiate(cfg)(aabb=aabb, near_far=list(near_far)).to(device)
        if "sampler.alphaMask.alpha_volume" in sd:            # give the mask module the stored shape before loading
            from ..samplers.alphagrid import AlphaGridMask
            nerf.sampler.alphaMask = AlphaGridMask(sd["sampler.alphaMask.aabb"].to(device),
                                                   sd["sampler.alphaMask.alpha_volume"].to(device))
        # the reference drops the stored Sobol table and keeps the freshly scrambled one (:151); here the stored table is
        # kept when it fits, so that a reloaded model reproduces the renders of the saved one
        own = nerf.model.brdf_sampler.angs
        if "model.brdf_sampler.angs" not in sd or sd["model.brdf_sampler.angs"].shape != own.shape:
            sd["model.brdf_sampler.angs"] = own
        nerf.load_state_dict(sd, **kwargs)
        nerf.sampler.update(nerf.rf, init=True)               # step size / sample count of the loaded grid
        return nerf

    def check_schedule(self, iter, batch_mul):
        # modules/tensor_nerf.py:177-195 (mask rebuild BEFORE the field upsamples on the same iteration)
        req = self.model.check_schedule(iter, batch_mul)
        req |= self.sampler.check_schedule(iter, batch_mul, self.rf)
        req |= self.rf.check_schedule(iter, batch_mul)
        if req:
            self.sampler.update(self.rf, init=True)
        self.bg_noise *= self.bg_noise_decay
        return req

    def render_just_bg(self, viewdirs, roughness):
        if viewdirs.shape[0] == 0:
            return torch.empty((0, 3), device=viewdirs.device)
        return self.bg_module(viewdirs, roughness).reshape(-1, 3)

    def forward(self, rays, focal, start_mipval=None, bg_col=torch.tensor([1, 1, 1]), stepmul=1, recur=0,
                override_near=None, output_alpha=None, dynamic_batch_size=True, gt_normals=None,
                override_alpha_thres=None, is_train=False, ndc_ray=False, N_samples=-1, tonemap=True, draw_debug=True,
                max_weight_N=-1, noise=None):
        if (recur == 0 and is_train and self.fused_training_pass and not self.regulariser_stats and rays.is_cuda
                and torch.is_grad_enabled() and start_mipval is None and stepmul == 1 and override_near is None
                and dynamic_batch_size and gt_normals is None and override_alpha_thres is None and not ndc_ray and N_samples == -1
                and tonemap and max_weight_N == -1 and bg_col is not None):
            # (`output_alpha` -- train.py:525-534 passes it for every RGBA dataset -- is ignored by the reference module, and here)
            out = self._forward_fused(rays, focal, bg_col, noise)
            if out is not None:
                return out
        if recur == 0 and is_train and rays.is_cuda and torch.is_grad_enabled():
            # A training forward that leaves the fused pass builds the operator graph of nmf_amd/functional.py (~3.5 x the cost):
            # counted, and said once, so that a caller knows (bench.py asserts 0 in its fused legs)
            self.operator_graph_forwards += 1
            if self.fused_training_pass and not self.regulariser_stats and not TensorNeRF._warned_fallback:
                TensorNeRF._warned_fallback = True
                import warnings
                warnings.warn("TensorNeRF.forward(is_train=True) left the fused training pass for the operator graph (a call outside "
                              "the pass: non-default arguments, no host extension, a chunk without bounce rows, or a second pending "
                              "forward); `nerf.operator_graph_forwards` counts these calls", RuntimeWarning, stacklevel=3)
        if recur == 0:          # one gradient pass: primary and re-traced rays share the table-gradient nodes
            passes = [m for m in (self.rf, self.bg_module, getattr(self.model, "brdf", None),
                                  getattr(self.model, "diffuse_module", None)) if hasattr(m, "begin_pass")]
            for m in passes:
                m.begin_pass()
            nz = noise if noise is not None else self._noise
            if nz is not None and hasattr(nz, "begin_pass"):
                nz.begin_pass()                         # the pass's uniform / normal pools, one generator launch each
            # derived tables of the field (packed density planes) depend only on the parameters: queue their rebuild now, so
            # that it is issued while the GPU may still be busy with the previous step and before the sampler's read-back
            if rays.is_cuda:
                self.rf._fwd_tables() if hasattr(self.rf, "_fwd_tables") else self.rf._tables()
                if hasattr(self.bg_module, "_tables"):          # summed-area table + SH projection of the environment
                    self.bg_module._tables()
                    if hasattr(self.model, "brdf"):             # the microfacet model's diffuse irradiance (G=100)
                        self.bg_module.get_spherical_harmonics(100)
                if is_train and torch.is_grad_enabled():            # pass tokens / stacked weights of the shading modules
                    for m, fn in ((getattr(self.model, "diffuse_module", None), "head_pass"),
                                  (getattr(self.model, "brdf", None), "mlp_pass")):
                        if m is not None and hasattr(m, fn) and getattr(m, "fused", True):
                            getattr(m, fn)()
                    self.rf._pass_token()
            try:
                return self._render(rays, focal, start_mipval, bg_col, stepmul, recur, override_near, output_alpha,
                                    dynamic_batch_size, gt_normals, override_alpha_thres, is_train, ndc_ray, N_samples,
                                    tonemap, draw_debug, max_weight_N, noise)
            finally:
                for m in passes:
                    m.end_pass()
        return self._render(rays, focal, start_mipval, bg_col, stepmul, recur, override_near, output_alpha,
                            dynamic_batch_size, gt_normals, override_alpha_thres, is_train, ndc_ray, N_samples, tonemap,
                            draw_debug, max_weight_N, noise)

    def _forward_fused(self, rays, focal, bg_col, noise):
        """the training forward as ONE autograd node (fast_step.ChunkPass): -> (images, stats) like _render, or None when the pass
        does not cover this call (configuration, no host extension, a chunk without a bounce row): the caller then builds the
        operator graph.  What a training loop reads of the result (train.py:541-577): rgb_map, whole_valid, n_samples, ori_loss,
        prediction_loss -- and the three zero-weight regularisers, see regulariser_stats."""
        from ..fast_step import TrainPass, Unsupported
        tp = self._fused_pass
        if tp is None:
            tp = self._fused_pass = TrainPass(self)
        if not tp.supported():
            return None
        dev = rays.device
        if noise is None:
            if self._noise is None:
                self._noise = DeviceNoise(dev, seed=20211200)
            noise = self._noise
        c = tp.core()
        bg = bg_col.detach().to(device=dev, dtype=torch.float32).reshape(1, 3)
        c.white = bg                                # (the background colour of the primary rays; the Trainer passes white)
        try:
            res = tp.forward_autograd(rays if rays.is_contiguous() else rays.contiguous(), focal, noise)
        except Unsupported:
            return None
        if res is None:                             # no sample kept (train.py:567-568 skips the chunk): background only, the module
            return None
        rgb_map, acc_map, ori, out = res
        stats = dict(recur=0, whole_valid=out["whole_valid"], n_samples=list(out["n_samples"]), rays_kept=int(out["kept"]),
                     ori_terms=ori, acc_terms=acc_map)
        stats = LazyStats(stats, self, None, None, int(out["n_samples"][0]))
        images = LazyImages(None)
        images["rgb_map"] = rgb_map
        images["acc_map"] = acc_map.detach()
        return images, stats

    def _render(self, rays, focal, start_mipval, bg_col, stepmul, recur, override_near, output_alpha,
                dynamic_batch_size, gt_normals, override_alpha_thres, is_train, ndc_ray, N_samples, tonemap, draw_debug,
                max_weight_N, noise):
        dev = rays.device
        if noise is None:
            if self._noise is None:
                self._noise = DeviceNoise(dev, seed=20211200)
            noise = self._noise
        S = self.sampler.sample_compact(rays, focal, rf=self.rf, override_near=override_near, is_train=is_train,
                                        dynamic_batch_size=dynamic_batch_size, noise=noise)
        B, M = S.b, S.M
        n_samples = [M]
        wv = S.whole_valid
        offsets = S.offsets[: B + 1]
        # valid rays are a prefix (alphagrid.py:359); the env-map lookup takes whole ray rows and reads columns 3..5
        ray_dirs = rays if B == rays.shape[0] else rays[:B]

        # Sparse appearance: the radiance only depends on the appearance features of the samples that spawn secondary
        # rays, so unless the per-sample debug maps are wanted (eval with draw_debug) the field's appearance branch and the
        # material heads run on those rows only (Microfacet.shade_compact).
        dense_app = (not is_train) and draw_debug
        if M > 0 and hasattr(self.rf, "query_weights"):      # field query + raw2alpha as one graph node (:286,366,386,393)
            weight, _sf, app, world_normal = self.rf.query_weights(S.xyzt, S.dist, offsets, B, want_app=dense_app,
                                                                   want_normal=True)
        else:
            sigma, _sf, app, world_normal = self.rf.query(S.xyzt, want_app=dense_app, want_normal=True)
            weight = Composite.apply(sigma, S.dist, offsets, B, float(self.rf.distance_scale))
        if not dense_app:
            app = None

        def render_reflection(brays, mipval, retrace):                                               # :291-317
            if retrace:
                ims, st = self(brays, focal, recur=recur + 1, bg_col=None, dynamic_batch_size=False,
                               stepmul=self.recur_stepmul, start_mipval=mipval.reshape(-1),
                               override_near=3 * float(hip.host(self.sampler.stepsize)), is_train=is_train,
                               ndc_ray=False, tonemap=False, draw_debug=False, noise=noise)
                n_samples.extend(st["n_samples"])
                return ims["rgb_map"]
            noise.skip("rand", (brays.shape[0],))
            noise.skip("rand", (brays.shape[0],))
            return self.render_just_bg(brays, mipval.reshape(-1))

        shaded = None
        if M > 0:
            shaded = self.model.shade_compact(S, app, world_normal, weight, render_reflection, self.bg_module,
                                              is_train, recur, noise, app_fn=self.rf.compute_appfeature)

        images = {}
        stats = dict(recur=recur, whole_valid=wv, n_samples=n_samples, rays_kept=B)
        per_ray_bg = self.bg_module is not None and bg_col is None
        if per_ray_bg:                                                                               # :460-468
            rough = -100 * torch.ones(B, device=dev) if start_mipval is None else start_mipval[:B]
            noise.skip("rand", (B,))
            noise.skip("rand", (B,))
            bg = self.render_just_bg(ray_dirs, rough).reshape(-1, 3)
            if tonemap:
                bg = self.tonemap(bg, noclip=True)
        else:
            bg = bg_col.to(device=dev, dtype=torch.float32).reshape(1, 3)

        want_stats = recur == 0 and (is_train or not draw_debug)
        # one pass per ray: acc (:448), rgb (:452), orientation term (:583-587), tonemap (:658), background (:659)
        if shaded is not None and shaded.mix_args is not None and shaded._refl_rows is None:
            # Fresnel mix of the bounce rows + the per-ray sums as one graph node
            rgb_map, acc_map, ori, refl = ShadeCompose.apply(
                weight, world_normal, bg, shaded.inv, offsets, S.ray_id, S.rays, B, per_ray_bg, bool(tonemap), bool(self.hdr),
                bool(want_stats and M > 0), *shaded.mix_args)
            shaded.refl_rows = refl              # detached: the debug maps of this pass read values only
        else:
            rgb_map, acc_map, ori = RayCompose.apply(
                weight, shaded.refl_rows if shaded is not None else None, world_normal if M > 0 else None, bg,
                shaded.inv if shaded is not None else None, offsets, S.ray_id, S.rays, B, per_ray_bg, bool(tonemap),
                bool(self.hdr), bool(want_stats and M > 0))

        if not is_train and draw_debug:                                                              # :480-566
            with torch.no_grad():
                acc_d = acc_map.detach()
                images["depth"] = segment_sum(weight * S.z, offsets, S.ray_id, B)
                wn = segment_sum(world_normal * weight[:, None], offsets, S.ray_id, B)
                images["world_normal"] = acc_d[..., None] * wn + (1 - acc_d[..., None])
                images["surf_width"] = (offsets[1:] - offsets[:-1])
                debug = shaded.debug() if shaded is not None else \
                    {k: torch.empty((0, v), device=dev) for k, v in self.model.outputs.items()}
                for k, v in debug.items():
                    images[k] = segment_sum(v * weight[:, None], offsets, S.ray_id, B) + (1 - acc_d[..., None]) * bg
        elif recur == 0:                                                                             # :567-649
            # per-ray terms; the sums (ori_loss :583-587, prediction_loss :598-602 -- normal_module is None, so
            # pred_norms == 0 and align_world_loss == 2) are formed on first read, the trainer mixes the
            # vectors directly (functional.LossMix)
            stats["ori_terms"], stats["acc_terms"] = ori, acc_map
            stats = LazyStats(stats, self, shaded, weight, M)
            images = LazyImages(shaded)
        images["rgb_map"] = rgb_map
        images["acc_map"] = acc_map.detach()
        return images, stats


class LazyStats(dict):
    """Statistics dict of TensorNeRF.forward whose regulariser entries (envmap_reg, brdf_reg, diffuse_reg,
    modules/tensor_nerf.py:600-649) are evaluated when first read: the default params give them zero weight
    (train.py:650-654), so the training step never pays for them, while a caller that does read them gets the same
    differentiable tensors as before."""

    _LAZY = ("envmap_reg", "brdf_reg", "diffuse_reg", "ori_loss", "prediction_loss", "distortion_loss")

    def __init__(self, base, nerf, shaded, weight, M):
        super().__init__(base)
        self._src = (nerf, shaded, weight, M)

    def __missing__(self, key):
        if key not in self._LAZY:
            raise KeyError(key)
        nerf, shaded, weight, M = self._src
        dev = self["acc_terms"].device if weight is None else weight.device
        if key == "ori_loss":
            ori = self["ori_terms"]
            v = ori.sum() if ori is not None else torch.zeros((), device=dev)
        elif key == "prediction_loss":
            v = 2.0 * self["acc_terms"].sum()
        elif key == "distortion_loss":
            v = torch.zeros((), device=dev)
        elif shaded is None:        # the fused training pass (TensorNeRF.regulariser_stats): the three zero-weight regularisers
            # (a loop that gives one of them a non-zero weight says so: nerf.regulariser_weights = dict(envmap_reg=..., ...) makes the
            #  read fail instead of training without the term)
            if (getattr(nerf, "regulariser_weights", None) or {}).get(key):
                raise RuntimeError(f"stats['{key}'] has a non-zero weight but the fused training pass does not evaluate it: set "
                                   "nerf.regulariser_stats = True (operator graph)")
            v = torch.zeros((), device=dev)
        elif key == "envmap_reg":
            v = (nerf.bg_module.mean_color().mean() - 0.05).clip(min=0)
        elif key == "brdf_reg":
            v = shaded.debug()["tint"].mean().clip(min=0) if M > 0 else torch.tensor(0.0, device=dev)
        else:
            v = ((weight.detach().reshape(-1, 1) * shaded.debug()["diffuse"]).sum() / 3
                 if M > 0 else torch.tensor(0.0, device=dev))
        self[key] = v
        return v

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._LAZY

    def keys(self):
        return list(dict.keys(self)) + [k for k in self._LAZY if not dict.__contains__(self, k)]


class LazyImages(dict):
    """images dict of a training forward: rgb_map / acc_map plus the per-sample debug maps of the shading model
    (modules/tensor_nerf.py:646-648), the latter built on first access."""

    def __init__(self, shaded):
        super().__init__()
        self._shaded = shaded

    def __missing__(self, key):
        if self._shaded is None:
            raise KeyError(f"{key}: the fused training pass returns rgb_map / acc_map only (what train.py:541-577 reads); set "
                           "nerf.fused_training_pass = False for the per-sample debug maps of a training forward")
        d = self._shaded.debug()
        if key not in d:
            raise KeyError(key)
        self[key] = d[key]
        return d[key]
