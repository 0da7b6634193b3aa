"""Learnable equirectangular environment map with summed-area-table prefiltering on HIP kernels --
host-side mirror of the reference's modules/integral_equirect.py (IntegralEquirect :176-504).
The SAT is rebuilt when bg_mat changes (once per optimiser step), not on every call (:431-433)."""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import hip
from ..functional import EnvLookup, GradPass, SatBuild, FastPrivateAttrs
from . import sh


class IntegralEquirect(FastPrivateAttrs, torch.nn.Module):
    def __init__(self, bg_resolution, init_val, activation="identity", mipbias=0, mipnoise=0, lr=0.15, mipbias_lr=1e-3,
                 brightness_lr=0.01, mul_lr=0.01, mul_betas=(0.9, 0.999), betas=(0.9, 0.99)):
        super().__init__()
        if activation != "exp":
            raise NotImplementedError("HIP env map implements activation='exp' (microfacet_tensorf2.yaml:147)")
        if mipnoise != 0:
            raise NotImplementedError("mipnoise != 0 is not used by this config")
        self.bg_mat = nn.Parameter(init_val * torch.ones((1, 3, bg_resolution, 2 * bg_resolution)))
        self.register_parameter("mipbias", nn.Parameter(torch.tensor(mipbias, dtype=float)))
        self.register_parameter("brightness", nn.Parameter(torch.tensor(0.0, dtype=float)))
        self.register_parameter("mul", nn.Parameter(torch.tensor(1.0, dtype=float)))
        self.mipnoise = mipnoise
        self.lr, self.mul_lr, self.mipbias_lr, self.brightness_lr = lr, mul_lr, mipbias_lr, brightness_lr
        self.mul_betas, self.betas = list(mul_betas), list(betas)
        self.activation = activation
        self.register_buffer("sh_A", torch.tensor(sum([[sh.Al2(l)] * (2 * l + 1) for l in range(16)], []),
                                                  dtype=torch.float32))
        self._cache = None
        self._sh_cache = None
        self._sh_const = None
        self._scalars = None
        self._memo = None
        self._pass, self._pass_open = None, False   # (GradPass, token) shared by the lookups of a forward/backward pass

    def get_optparam_groups(self, lr_scale=1):
        # modules/integral_equirect.py:232-257
        return [{"params": self.bg_mat, "betas": self.betas, "lr": self.lr * lr_scale, "name": "bg"},
                {"params": self.brightness, "lr": self.brightness_lr * lr_scale, "name": "bg"},
                {"params": self.mul, "lr": self.mul_lr * lr_scale, "betas": self.mul_betas, "name": "bg"},
                {"params": [self.mipbias], "lr": self.mipbias_lr * lr_scale, "name": "mipbias"}]

    def hw(self):
        return self.bg_mat.shape[-2], self.bg_mat.shape[-1]

    @property
    def bg_resolution(self):
        return self.hw()[0]

    def get_device(self):
        return self.bg_mat.device

    def _dev_scalars(self):
        """float32 [3] = (mipbias, brightness, mul) on the device, refreshed when a parameter changes: the kernels read the
        three learnable scalars from here, so an optimizer update never forces a host read-back"""
        memo = self._memo
        if memo is not None and "sc" in memo:
            return memo["sc"]
        sc = self._dev_scalars_checked()
        if memo is not None:
            memo["sc"] = sc
        return sc

    def _dev_scalars_checked(self):
        ps = (self.mipbias, self.brightness, self.mul)
        key = (ps[0]._version, ps[1]._version, ps[2]._version)
        ptrs = (ps[0].data_ptr(), ps[1].data_ptr(), ps[2].data_ptr())
        c = self._scalars
        if c is None or c[2] != ptrs:
            # persistent float32 [3] + the slot table of the converting copy (0-d float64 parameters, :199-207)
            sc = torch.empty(3, dtype=torch.float32, device=self.bg_mat.device)
            slots = (hip.CopySlot * 3)()
            for i, p in enumerate(ps):
                if p.dtype not in (torch.float32, torch.float64):
                    raise hip.NmfHipError("env-map scalars must be float32 or float64")
                slots[i] = hip.CopySlot(p.data_ptr(), sc.data_ptr() + 4 * i, 1, 1 if p.dtype == torch.float64 else 0, 0)
            c = self._scalars = (None, sc, ptrs, slots)
        if c[0] != key:
            hip.multi_copy(c[3], 3)                      # one launch, no host read-back, no temporaries
            self._scalars = (key, c[1], c[2], c[3])
        return c[1]

    def _host_scalars(self):
        """(mipbias, brightness, mul) as python floats (one read-back; used by tests / tools, not by the render path)"""
        return tuple(float(v) for v in self._dev_scalars().tolist())

    def _tables(self):
        memo = self._memo
        if memo is not None and "tab" in memo:
            return memo["tab"]
        tab = self._tables_checked()
        if memo is not None:
            memo["tab"] = tab
        return tab

    def _tables_checked(self):
        key = (self.bg_mat._version, self.brightness._version, self.mul._version)
        ptr = self.bg_mat.data_ptr()
        c = self._cache
        if c is None or c[2] != ptr:
            tab = hip.sat_build(self.bg_mat.detach(), sc=self._dev_scalars(), pole=True, interleaved=True)
            self._cache = (key, tab, ptr)
        elif c[0] != key:       # same storage, new values (an optimizer step): rebuild the tables in place
            hip.sat_build(self.bg_mat.detach(), sc=self._dev_scalars(), out=c[1], pole=True, interleaved=True)
            self._cache = (key, c[1], ptr)
        return self._cache[1][:3]

    def _lookup_table(self):
        """the channel-interleaved copy [H,W,4] of the summed-area table the lookup kernels read (same values as sat)"""
        self._tables()
        return self._cache[1][3]

    def activation_fn(self, x):
        return torch.exp((self.brightness + self.mul * x).clip(max=20))

    def mean_color(self):
        return self.activation_fn(self.bg_mat).reshape(-1, 3).mean(dim=0)

    def save(self, path, prefix="", tonemap=None):
        """modules/integral_equirect.py:363-371: the activated map as `<prefix>pano.exr` (float RGB, equirectangular)"""
        import os
        from .. import exr
        with torch.no_grad():
            im = self.activation_fn(self.bg_mat.detach())
            if tonemap is not None:
                im = tonemap(im)
        exr.imwrite(os.path.join(str(path), f"{prefix}pano.exr"), im.permute(0, 2, 3, 1).squeeze(0).cpu().numpy())

    @torch.no_grad()
    def calc_envmap_psnr(self, gt_im, fH=500):
        """modules/integral_equirect.py:289-322: PSNR of the learned map against a ground-truth panorama [H,W,3] (numpy)
        after flipping / rolling it into this module's parameterisation, resampling both to fH x 2fH (nearest) and fitting
        an affine colour transform pred -> gt by least squares (the reference uses sklearn's LinearRegression: 3x3 matrix +
        intercept), squared error clipped to [0,1]."""
        import numpy as np
        import torch.nn.functional as F
        fW = 2 * fH
        gW = gt_im.shape[1]
        gt_im = gt_im[:, ::-1]
        gt_im = np.concatenate([gt_im[:, gW // 2:], gt_im[:, : gW // 2]], axis=1)
        gt = torch.as_tensor(np.ascontiguousarray(gt_im), dtype=torch.float32)
        gt = F.interpolate(gt.permute(2, 0, 1).unsqueeze(0), (fH, fW)).squeeze(0).permute(1, 2, 0)
        pred = self.activation_fn(self.bg_mat[0]).permute(1, 2, 0).detach().float().cpu()
        pred = F.interpolate(pred.permute(2, 0, 1).unsqueeze(0), (fH, fW)).squeeze(0).permute(1, 2, 0)
        Y = gt.reshape(-1, 3).double()
        X = torch.cat([pred.reshape(-1, 3).double(), torch.ones(fH * fW, 1, dtype=torch.float64)], dim=1)
        coef = torch.linalg.lstsq(X, Y).solution
        err = ((X @ coef - Y) ** 2).clip(min=0, max=1)
        return float(-10.0 * torch.log10(err.mean()))

    def forward(self, viewdirs, saSample, max_level=None):
        """viewdirs [R,3]; the build's callers may also hand over [R,6] ray rows (origin | direction): they are looked up
        along columns 3..5 in place, and the gradient comes back with the rays' own shape (no slice / pad kernels)."""
        if viewdirs.shape[0] == 0:
            return viewdirs.new_zeros((0, 3))
        sa = saSample.reshape(-1).detach().float()
        holder, token = self._pass_token()
        return EnvLookup.apply(self, viewdirs.float(), sa, holder, token)

    # ---- gradient pass: every lookup between begin_pass() and end_pass() shares one SatBuild node ---------------
    def begin_pass(self):
        # parameters cannot change inside a pass: the derived tables are looked up once and memoised until end_pass()
        # (the version / pointer comparison of every cache costs ~15 us of host time per call on the critical path)
        self._pass, self._pass_open, self._memo = None, True, {}

    def end_pass(self):
        self._pass, self._pass_open, self._memo = None, False, None

    def _pass_token(self):
        if not (torch.is_grad_enabled() and (self.bg_mat.requires_grad or self.brightness.requires_grad
                                             or self.mul.requires_grad or self.mipbias.requires_grad)):
            return None, None
        if self._pass_open and self._pass is not None:
            return self._pass
        holder = GradPass()
        token = SatBuild.apply(holder, self, self.bg_mat, self.brightness, self.mul, self.mipbias)
        if self._pass_open:
            self._pass = (holder, token)
        return holder, token

    @torch.no_grad()
    def get_spherical_harmonics(self, G, mipval=-5):
        """modules/integral_equirect.py:324-360; cached per bg_mat version (the reference recomputes it)."""
        memo = self._memo
        if memo is not None and ("sh", G, mipval) in memo:
            return memo[("sh", G, mipval)]
        out = self._spherical_harmonics_checked(G, mipval)
        if memo is not None:
            memo[("sh", G, mipval)] = out
        return out

    def _spherical_harmonics_checked(self, G, mipval):
        key = (self.bg_mat.data_ptr(), self.bg_mat._version, G, mipval, self.mipbias._version, self.brightness._version,
               self.mul._version)
        if self._sh_cache is None or self._sh_cache[0] != key:
            dev = self.get_device()
            ck = (G, float(mipval), str(dev))
            if self._sh_const is None or self._sh_const[0] != ck:
                # everything that does not depend on the map: the direction lattice, its log-solid-angle argument and the
                # quadrature weights 2 pi^2 sin(theta) Y_k / SB (so that the projection is one weighted sum per update)
                theta, phi = torch.meshgrid(torch.linspace(0, np.pi, G // 2, device=dev),
                                            torch.linspace(0, 2 * np.pi, G, device=dev), indexing="ij")
                dirs = torch.stack([torch.sin(theta) * torch.cos(phi), torch.sin(theta) * torch.sin(phi),
                                    torch.cos(theta)], dim=-1).reshape(-1, 3).contiguous()
                SB = dirs.shape[0]
                wq = (2 * np.pi ** 2 / SB) * sh.eval_sh_bases(9, dirs) * torch.sin(theta.reshape(SB, 1))     # [SB, 9]
                self._sh_const = (ck, dirs, torch.full((SB,), float(mipval), device=dev), wq.float().contiguous(),
                                  self.sh_A.reshape(-1)[:9].float().contiguous())
            _, dirs, mips, wq, shA = self._sh_const
            act, sat, pole = self._tables()
            bg = hip.sat_lookup_fwd(self._lookup_table(), dirs, mips, 0.0, pole, sc=self._dev_scalars())
            self._sh_cache = (key, hip.sh_project(bg, wq, shA))            # (coeffs, conv) [9,3] each
        return self._sh_cache[1]

    def _load_from_state_dict(self, *a, **k):
        self._cache = self._sh_cache = self._scalars = self._pass = None
        super()._load_from_state_dict(*a, **k)
