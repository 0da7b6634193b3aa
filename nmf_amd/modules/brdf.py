"""BRDF MLP -- host-side mirror of the reference's modules/brdf.py (MLPBRDF :72-261) for the
microfacet_tensorf2 configuration (feape=0, dotpe=-1, h/d encoders = ListISH([0,1,2,4]), 66 -> 64 -> 64 -> 4).
Feature build (gather + two ISH encodings) and the three dense layers are ONE fused kernel (nmf_brdf_mlp_fwd / _bwd:
v_mfma_f32_32x32x16_bf16 tiles on split-bf16 operands, fp32-class accuracy: csrc/brdf_mlp.hip); `forward` keeps the reference's per-ray signature, `forward_compact` is what the hot path
calls (per-row features / roughness gathered inside the kernel)."""
import torch

from ..functional import PassMixin, brdf_mlp
from .util import create_mlp


class ListISH(torch.nn.Module):
    """modules/ish.py:94-105 -- only its identity (degrees) matters here: the encoding is inside the kernel"""

    def __init__(self, degs=(0, 1, 2, 4)):
        super().__init__()
        self.degs = list(degs)

    def dim(self):
        return sum(2 * d + 1 for d in self.degs)


class MLPBRDF(PassMixin, torch.nn.Module):
    def __init__(self, in_channels, h_encoder=None, d_encoder=None, v_encoder=None, n_encoder=None, l_encoder=None,
                 feape=6, dotpe=0, activation="sigmoid", mul_LdotN=True, bias=0, lr=1e-4, shift=0, **kwargs):
        super().__init__()
        ok = (feape == 0 and dotpe < 0 and activation == "sigmoid" and not mul_LdotN and v_encoder is None
              and n_encoder is None and l_encoder is None and h_encoder is not None and d_encoder is not None
              and list(h_encoder.degs) == [0, 1, 2, 4] and list(d_encoder.degs) == [0, 1, 2, 4] and in_channels == 24)
        if not ok:
            raise NotImplementedError("HIP BRDF features implement the microfacet_tensorf2.yaml:86-104 configuration")
        self.in_channels = in_channels
        self.bias = bias
        self.lr = lr
        self.activation_name = activation
        self.in_mlpC = in_channels + 2 * (h_encoder.dim() + 3)
        self.h_encoder, self.d_encoder = h_encoder, d_encoder
        self.mlp = create_mlp(self.in_mlpC, 4, **kwargs)
        self.init_val = 0.25
        self.fused = kwargs.get("hidden_w", 128) == 64 and kwargs.get("num_layers", 0) == 3

    def mlp_pass(self):
        """(weights, bias, holder, token) for fused callers (functional.BounceRays)"""
        if not self.fused:
            raise NotImplementedError("the fused path implements hidden_w=64, num_layers=3 (microfacet_tensorf2.yaml:86-104)")
        if self._memo is not None and "mlp_pass" in self._memo:
            return self._memo["mlp_pass"]
        ws = self._weights()
        holder, token = self._param_pass(ws)
        out = ([w.detach().contiguous() for w in ws], float(self.bias), holder, token)
        if self._memo is not None:
            self._memo["mlp_pass"] = out
        return out

    def _weights(self):
        m = self.mlp
        return (m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[4].weight, m[4].bias)

    def forward_compact(self, half_vec, diff_vec, feat_rows, rough_rows, row_of_ray, row_offsets):
        """half_vec / diff_vec [R,3] (local frame), feat_rows [Mb,24], rough_rows [Mb], row_of_ray [R] -> weights [R,3]"""
        if not self.fused:
            raise NotImplementedError("nmf_brdf_mlp implements hidden_w=64, num_layers=3 (microfacet_tensorf2.yaml:86-104)")
        return brdf_mlp(half_vec.detach(), diff_vec.detach(), feat_rows, rough_rows.detach().reshape(-1),
                        row_of_ray, row_offsets, float(self.bias), self._weights(), owner=self)

    def forward(self, V, L, N, H, local_v, half_vec, diff_vec, efeatures, eax, eay):
        """Reference signature (modules/brdf.py:177-261): one row per secondary ray.  With feape=0, dotpe=-1 and the h / d
        encoders only half_vec, diff_vec [R,3] (local frame), efeatures [R,24] and eax [R] (roughness) enter the network;
        V, L, N, H, local_v, eay are accepted and unused, as in the reference for this configuration.  Returns [R,3].
        Same fused kernel as forward_compact with the identity ray -> row map."""
        R = half_vec.shape[0]
        dev = half_vec.device
        idx = torch.arange(R, device=dev, dtype=torch.int32)
        off = torch.arange(R + 1, device=dev, dtype=torch.int64)
        return self.forward_compact(half_vec.reshape(R, 3), diff_vec.reshape(R, 3), efeatures.reshape(R, -1).contiguous(),
                                    eax.reshape(R), idx, off)

    def calibrate(self, efeatures, bg_brightness):
        # modules/brdf.py:141-175 (random unit vectors; only the mean of the output matters)
        N = efeatures.shape[0]
        dev = efeatures.device

        def rv():
            v = 2 * torch.rand((N, 3), device=dev) - 1
            return v / v.norm(dim=-1, keepdim=True).clip(min=1e-8)

        with torch.no_grad():
            hv, dv = rv(), rv()
            w = self(hv, hv, hv, hv, hv, hv, dv, efeatures, torch.rand(N, device=dev), None)
        target = self.init_val / float(bg_brightness)
        inv = lambda v: torch.log(v / (1 - v))  # noqa: E731
        self.bias += float(torch.log(torch.tensor(target / (1 - target)))) - float(inv(w).mean())
