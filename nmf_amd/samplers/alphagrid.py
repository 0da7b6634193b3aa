"""Alpha-grid ray sampler on HIP kernels -- host-side mirror of the reference's samplers/alphagrid.py
(AlphaGridMask :6-60, AlphaGridSampler :63-370).  `sample()` keeps the reference's signature and return
tuple; `sample_compact()` is what the rest of this package uses: the kept samples in (ray, step) order plus
CSR ray offsets, never the dense [rays x N] tensors."""
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
from ..functional import FastPrivateAttrs
import torch.nn.functional as F

from .. import hip


@dataclass
class Samples:
    xyzt: torch.Tensor          # [M,4] world xyz, t/focal
    ray_id: torch.Tensor        # [M] int32, index into the b kept rays
    step_id: torch.Tensor       # [M] int32, step index k
    z: torch.Tensor             # [M]
    dist: torch.Tensor          # [M]
    offsets: torch.Tensor       # [B+1] int64 (entries past b clamped to M)
    whole_valid: torch.Tensor   # [B] bool
    M: int
    b: int
    N: int
    params: object = None
    rays: Optional[torch.Tensor] = None
    jitter: Optional[torch.Tensor] = None
    valid_bits: Optional[torch.Tensor] = None

    def dense(self):
        """ray_valid [b,N] bool and z_vals [b,N] -- API compatibility / parity tests only"""
        return hip.march_dense(self.params, self.rays, self.b, self.jitter, self.valid_bits)


class AlphaGridMask(FastPrivateAttrs, torch.nn.Module):
    def __init__(self, aabb, alpha_volume):
        super().__init__()
        self.register_buffer("aabb", aabb)
        aabbSize = self.aabb[1] - self.aabb[0]
        self.register_buffer("invgrid_size", 1.0 / aabbSize * 2)
        self.register_buffer("grid_size", torch.LongTensor(
            [alpha_volume.shape[-1], alpha_volume.shape[-2], alpha_volume.shape[-3]]))
        self.register_buffer("alpha_volume", alpha_volume.view(1, 1, *alpha_volume.shape[-3:]))
        self._bits = None

    def _packed(self):
        if self._bits is None or self._bits[0] != (self.alpha_volume.data_ptr(), self.alpha_volume._version):
            bits = hip.alpha_pack(self.alpha_volume.reshape(-1).float())
            gz, gy, gx = self.alpha_volume.shape[-3:]
            self._bits = ((self.alpha_volume.data_ptr(), self.alpha_volume._version), bits,
                          hip.alpha_coarse(bits, (gx, gy, gz)), self._occupied_box())
        return self._bits

    @torch.no_grad()
    def _occupied_box(self):
        """World-space box outside of which sample_alpha (:23-45) cannot be positive: a set voxel i reaches points whose
        texel coordinate lies in (i-1, i+1) (trilinear footprint, align_corners=True), so the box of the set voxels grown
        by 1.5 voxels.  Evaluated once per mask (one read-back), handed to the marcher as nmf_march_params.occ_min/max."""
        vol = self.alpha_volume[0, 0] > 0
        if not bool(vol.any()):
            return None
        a = hip.host(self.aabb).astype(np.float64)
        lo, hi = [], []
        for axis, dim in ((0, 2), (1, 1), (2, 0)):            # world x, y, z <-> volume dims 2, 1, 0
            keep = [d for d in range(3) if d != dim]
            occ = vol.any(dim=keep[1]).any(dim=keep[0]) if keep[1] > keep[0] else vol.any(dim=keep[0]).any(dim=keep[1])
            idx = torch.nonzero(occ).reshape(-1)
            g = vol.shape[dim]
            i0, i1 = float(idx.min()) - 1.5, float(idx.max()) + 1.5
            size = a[1][axis] - a[0][axis]
            den = max(g - 1, 1)
            lo.append(a[0][axis] + size * i0 / den)
            hi.append(a[0][axis] + size * i1 / den)
        return lo, hi

    def occupied_box(self):
        return self._packed()[3]

    def bits(self):
        """occupancy bit per voxel (nmf_alpha_pack)"""
        return self._packed()[1]

    def coarse_bits(self):
        """occupancy bit per 8^3 voxels (nmf_alpha_coarse): lets the marcher skip the 8-corner test in empty space"""
        return self._packed()[2]


class AlphaGridSampler(FastPrivateAttrs, torch.nn.Module):
    def __init__(self, aabb, enable_alpha_mask=False, threshold=1e-4, multiplier=1, near_far=(2, 6), nEnvSamples=0,
                 alphaMask_thres=0.001, update_list=(), max_samples=-1):
        super().__init__()
        self.aabb = aabb
        self.enable_alpha_mask = enable_alpha_mask
        self.alphaMask = None
        self._params_cache = {}
        self.multiplier = int(multiplier)
        self.near_far = list(near_far)
        self.update_list = list(update_list)
        self.grid_size = [128, 128, 128]
        self.alphaMask_thres = alphaMask_thres
        self.max_samples = max_samples
        self._calls = 0

    def check_schedule(self, iteration, batch_mul, rf):
        if iteration in self.update_list:
            self.update(rf)
        return False

    @torch.no_grad()
    def update(self, rf, init=False):
        # samplers/alphagrid.py:96-111
        self.aabb = rf.aabb
        self.units = rf.units
        self.contract_space = rf.contract_space
        self.nSamples = rf.nSamples * self.multiplier
        self.stepsize = rf.stepsize / self.multiplier
        if not init:
            self.updateAlphaMask(rf, rf.grid_size)
            self.grid_size = rf.grid_size

    @torch.no_grad()
    def getDenseAlpha(self, rf, grid_size):
        # samplers/alphagrid.py:226-247: density on the G^3 lattice, alpha = 1 - exp(-sigma * stepsize)
        gs = [int(g) for g in grid_size]
        dev = rf.get_device()
        lin = [torch.linspace(0, 1, g, device=dev) for g in gs]
        alpha = torch.zeros(gs, device=dev)
        for i in range(gs[0]):
            s = torch.stack(torch.meshgrid(lin[0][i:i + 1], lin[1], lin[2], indexing="ij"), -1).reshape(-1, 3)
            xyz = self.aabb[0] * (1 - s) + self.aabb[1] * s
            if self.alphaMask is not None:
                m = F.grid_sample(self.alphaMask.alpha_volume,
                                  ((xyz - self.alphaMask.aabb[0]) * self.alphaMask.invgrid_size - 1).view(1, -1, 1, 1, 3),
                                  align_corners=True).view(-1) > 0
            else:
                m = torch.ones(xyz.shape[0], dtype=torch.bool, device=dev)
            sigma = torch.zeros(xyz.shape[0], device=dev)
            if m.any():
                sigma[m] = rf.compute_densityfeature(xyz[m])
            alpha[i] = (1 - torch.exp(-sigma * self.stepsize)).view(gs[1], gs[2])
        return alpha

    @torch.no_grad()
    def updateAlphaMask(self, rf, grid_size=(200, 200, 200)):
        # samplers/alphagrid.py:249-276
        gs = [int(g) for g in grid_size]
        alpha = self.getDenseAlpha(rf, gs)
        alpha = alpha.clamp(0, 1).transpose(0, 2).contiguous()[None, None]
        alpha = F.max_pool3d(alpha, kernel_size=3, padding=1, stride=1).view(gs[::-1])
        vol = (alpha >= self.alphaMask_thres).float()
        self.alphaMask = AlphaGridMask(self.aabb, vol).to(rf.get_device())
        return self.aabb

    # ---- the hot path ---------------------------------------------------------------------------------
    @torch.no_grad()
    def sample_compact(self, rays_chunk, focal, rf=None, override_near=None, is_train=False, dynamic_batch_size=True,
                       noise=None, **_):
        return self.sample_finish(self.sample_begin(rays_chunk, focal, override_near, is_train, dynamic_batch_size, noise))

    def sample_begin(self, rays_chunk, focal, override_near=None, is_train=False, dynamic_batch_size=True, noise=None):
        """First half of sample_compact: the counting pass, the budget scan and the START of the size read-back (into pinned
        memory, behind an event).  Whatever the caller queues between this and sample_finish() runs while the two sizes
        travel to the host -- the sampler needs nothing of the field, so the per-step table rebuilds (level 0) or the BRDF
        MLP of the level above (level 1) go there instead of the device idling through the round trip."""
        dev = rays_chunk.device
        B = rays_chunk.shape[0]
        N = int(self.nSamples)
        near, far = self.near_far
        if override_near is not None:
            near = float(override_near)
        use_mask = self.alphaMask is not None and self.enable_alpha_mask
        jitter, (seed, off) = (None, (0, 0))
        if is_train:
            if noise is None:
                self._calls += 1
                seed, off = 0x9E3779B9, self._calls
            else:
                jitter, (seed, off) = noise.jitter(B, N)
        packed, hit = self.params_block(focal, near, is_train)
        p = hip.MarchParams.from_buffer_copy(hit[1])
        p.seed, p.offset = int(seed), int(off)
        rays = rays_chunk.contiguous()
        valid, counts = hip.march_count(p, rays, jitter, packed[1] if use_mask else None, packed[2] if use_mask else None)
        pins = getattr(noise, "pins", None)
        if pins is not None and override_near is not None and 1 in pins.valid and tuple(pins.valid[1].shape) == (B, N):
            # tests (noise.Pins): the reference's recorded occupancy decisions of the SECONDARY rays' candidate steps replace
            # the marcher's own; `valid_flips` counts the 64-step words the marcher itself decided otherwise
            fv = pins.valid[1].to(dev)
            W = valid.shape[1]
            bits = torch.zeros((B, W * 64), dtype=torch.int64, device=dev)
            bits[:, :N] = fv
            words = (bits.view(B, W, 64) << torch.arange(64, device=dev)).sum(-1)          # bit k of word j = step 64 j + k
            pins.valid_flips = int(((words ^ valid) != 0).sum())
            valid, counts = words.contiguous(), fv.sum(dim=1).int().contiguous()
        budget = self.max_samples if (self.max_samples > 0 and is_train and dynamic_batch_size) else -1
        offsets, wv, totals = hip.march_scan(counts, budget)
        rb = hip.Readback.of(dev)
        rb.start(totals)
        return (rb, p, rays, jitter, valid, offsets, wv, N)

    def params_block(self, focal, near=None, is_train=False):
        """-> (packed alpha mask or None, (owners, nmf_march_params block)) for rays that start at `near` (default: the scene's
        near plane).  The geometry half of the parameter block only changes with the mask / the step size: kept per (near, far,
        focal, ...) and copied by the caller (filling the ctypes struct field by field costs 10 us per call); seed / offset of the
        jitter are the caller's.  The C++ training pass (csrc/step_core.inc) reads the block by address."""
        N = int(self.nSamples)
        far = self.near_far[1]
        near = self.near_far[0] if near is None else float(near)
        use_mask = self.alphaMask is not None and self.enable_alpha_mask
        packed = self.alphaMask._packed() if use_mask else None
        stepsize = float(hip.host(self.stepsize))
        ck = (near, far, float(focal), N, bool(is_train), stepsize, use_mask)
        owner = (self.aabb, self.aabb._version, self.alphaMask if use_mask else None, packed[0] if use_mask else None)
        hit = self._params_cache.get(ck)
        ho = hit[0] if hit is not None else None         # the entry holds its tensors: identities cannot be reused
        if ho is None or ho[0] is not owner[0] or ho[1] != owner[1] or ho[2] is not owner[2] or ho[3] != owner[3]:
            if len(self._params_cache) > 8:
                self._params_cache.clear()
            hit = self._params_cache[ck] = (owner, hip.march_params(
                self.aabb, hip.host(self.alphaMask.invgrid_size) if use_mask else None, stepsize, near, far, focal, N,
                [int(g) for g in hip.host(self.alphaMask.grid_size)] if use_mask else None, is_train, 0, 0,
                occ_box=packed[3] if use_mask else None))
        return packed, hit

    def sample_finish(self, pending):
        rb, p, rays, jitter, valid, offsets, wv, N = pending
        M, b = rb.get()                                  # the one host sync of the sampler
        xyzt, ray_id, step_id, z, dist = hip.march_fill(p, rays, b, M, jitter, valid, offsets)
        return Samples(xyzt, ray_id, step_id, z, dist, offsets, wv.view(torch.bool), M, b, N, p, rays, jitter, valid)   # 0 / 1 bytes: no conversion launch

    @torch.no_grad()
    def sample(self, rays_chunk, focal, rf=None, override_near=None, is_train=False, dynamic_batch_size=True,
               override_alpha_thres=None, stepmul=1, ndc_ray=False, noise=None, **args):
        """Reference signature (samplers/alphagrid.py:279-370): returns
        (xyzs[M,4], ray_valid[b,N] bool, N, z_vals[b,N], dists[b,N], whole_valid[B])."""
        if ndc_ray:
            raise NotImplementedError("ndc rays are not used by the blender datasets of this config")
        s = self.sample_compact(rays_chunk, focal, rf, override_near, is_train, dynamic_batch_size, noise)
        ray_valid, z_vals = s.dense()
        dists = torch.cat((z_vals[:, 1:] - z_vals[:, :-1], torch.zeros_like(z_vals[:, :1])), dim=-1)
        return s.xyzt, ray_valid, s.N, z_vals, dists, s.whole_valid
