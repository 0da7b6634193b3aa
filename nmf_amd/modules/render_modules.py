"""Material heads -- host-side mirror of the reference's RandHydraMLPDiffuse
(modules/render_modules.py:447-574) for pospe=-1, feape=0, num_layers=1: four Linear(24 -> 3,3,3,2) heads.
Evaluated as one 24 -> 11 product per sample by nmf_heads_fwd/bwd (csrc/heads.hip); there is no CPU path."""
import torch

from ..functional import PassMixin, material_heads
from .util import create_mlp


def inv_sigmoid(v):
    return torch.log(v / (1 - v))


class RandHydraMLPDiffuse(PassMixin, torch.nn.Module):
    def __init__(self, in_channels, pospe=12, view_encoder=None, roughness_view_encoder=None, roughness_cfg=None,
                 feape=6, allocation=0, unlit_tint=False, lr=1e-4, tint_bias=-1, diffuse_bias=-2, diffuse_mul=1,
                 roughness_bias=1, start_roughness=0.35, f0_bias=0, **kwargs):
        super().__init__()
        if pospe >= 0 or feape != 0 or view_encoder is not None or roughness_view_encoder is not None or allocation > 0:
            raise NotImplementedError("implements the microfacet_tensorf2.yaml:106-131 configuration")
        self.in_mlpC = in_channels
        self.tint_bias, self.diffuse_bias, self.roughness_bias = tint_bias, diffuse_bias, roughness_bias
        self.lr = lr
        self.diffuse_mul = diffuse_mul
        self.start_roughness = start_roughness
        self.f0_bias = f0_bias
        self.diffuse_mlp = create_mlp(self.in_mlpC, 3, **kwargs)
        self.tint_mlp = create_mlp(self.in_mlpC, 3, **kwargs)
        self.f0_mlp = create_mlp(self.in_mlpC, 3, **kwargs)
        self.roughness_mlp = create_mlp(self.in_mlpC, 2, **(roughness_cfg if roughness_cfg is not None else kwargs))
        self._stacked = None

    def _head_params(self):
        return (self.diffuse_mlp[0].weight, self.diffuse_mlp[0].bias, self.tint_mlp[0].weight, self.tint_mlp[0].bias,
                self.f0_mlp[0].weight, self.f0_mlp[0].bias, self.roughness_mlp[0].weight, self.roughness_mlp[0].bias)

    def head_pass(self):
        """(hp, W [11,24], b [11], holder, token) for fused callers (functional.BounceRays): the stacked head weights of the
        current parameter version and the gradient pass they accumulate into"""
        from ..functional import GradPass, StackedHeadGrads
        if self._memo is not None and "head_pass" in self._memo:
            return self._memo["head_pass"]
        hp = (float(self.diffuse_mul), float(self.diffuse_bias), float(self.tint_bias), float(self.f0_bias),
              float(self.roughness_bias))
        ps = self._head_params()
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if self._stacked is None or self._stacked[0] != key:
            self._stacked = (key, self._restack(ps))
        W, b = self._stacked[1]
        holder, token = None, None
        if torch.is_grad_enabled() and any(p.requires_grad for p in ps):
            if self._pass_open and self._pass is not None:
                holder, token = self._pass
            else:
                holder = GradPass()
                token = StackedHeadGrads.apply(holder, *ps)
                if self._pass_open:
                    self._pass = (holder, token)
        if self._memo is not None:
            self._memo["head_pass"] = (hp, W, b, holder, token)
        return hp, W, b, holder, token

    def _restack(self, ps):
        """the four Linear layers stacked as W [11,24], b [11]: persistent buffers refreshed with ONE copy launch
        (nmf_multi_copy) per parameter update instead of two torch.cat"""
        from .. import hip
        st = getattr(self, "_stack_state", None)
        ptrs = tuple(p.data_ptr() for p in ps)
        if ps[0].is_cuda and all(p.dtype == torch.float32 and p.is_contiguous() for p in ps):
            if st is None or st[0] != ptrs:
                W = torch.empty((sum(p.shape[0] for p in ps[0::2]), ps[0].shape[1]), dtype=torch.float32, device=ps[0].device)
                b = torch.empty((W.shape[0],), dtype=torch.float32, device=ps[0].device)
                slots = (hip.CopySlot * len(ps))()
                ow = ob = 0
                for i in range(0, len(ps), 2):
                    w_, b_ = ps[i], ps[i + 1]
                    slots[i] = hip.CopySlot(w_.data_ptr(), W.data_ptr() + 4 * ow, w_.numel(), 0, 0)
                    slots[i + 1] = hip.CopySlot(b_.data_ptr(), b.data_ptr() + 4 * ob, b_.numel(), 0, 0)
                    ow += w_.numel()
                    ob += b_.numel()
                st = self._stack_state = (ptrs, W, b, slots)
            hip.multi_copy(st[3], len(ps))
            return st[1], st[2]
        return (torch.cat([p.detach() for p in ps[0::2]], 0).contiguous(), torch.cat([p.detach() for p in ps[1::2]], 0).contiguous())

    def heads(self, features):
        """[M,11] = (albedo 3 | tint 3 | f0 3 | roughness 2) with the activations applied (nmf_heads_fwd)."""
        if features.shape[0] == 0:
            return features.new_zeros((0, 11))
        hp = (float(self.diffuse_mul), float(self.diffuse_bias), float(self.tint_bias), float(self.f0_bias),
              float(self.roughness_bias))
        ps = self._head_params()
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if self._stacked is None or self._stacked[0] != key:      # the four Linear layers stacked, once per update
            self._stacked = (key, self._restack(ps))
        return material_heads(features, hp, ps, owner=self, stacked=self._stacked[1])

    def forward(self, pts, viewdirs, features, std=0, **kwargs):
        o = self.heads(features)
        diffuse, tint, f0, r = o[:, 0:3], o[:, 3:6], o[:, 6:9], o[:, 9:11]
        return diffuse, tint, dict(diffuse=diffuse, r1=r[:, 0:1], r2=r[:, 1:2], f0=f0, tint=tint)

    def calibrate(self, mean_brightness, conserve_energy, *args, **kwargs):
        # modules/render_modules.py:505-515
        with torch.no_grad():
            diffuse, tint, extra = self(*args, **kwargs)
            v = (0.25 if not conserve_energy else 0.5) / float(mean_brightness)
            self.diffuse_bias += float(inv_sigmoid(torch.tensor(v))) - float(inv_sigmoid(diffuse).mean())
            rough = (extra["r1"] + extra["r2"]) / 2 / 2
            self.roughness_bias += float(inv_sigmoid(torch.tensor(self.start_roughness))) - float(inv_sigmoid(rough).mean())
