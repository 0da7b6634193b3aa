"""CPU oracle for the microfacet_tensorf2 render/train hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement (plain PyTorch on the CPU, explicit noise arguments, no
classes from the reference) of the algorithm the reference implements in
modules/tensor_nerf.py, samplers/alphagrid.py, fields/tensoRF.py, fields/tensor_base.py,
modules/grid_sample_Cinf.py, models/microfacet.py, brdf_samplers/ggx.py, brdf_samplers/base.py,
modules/brdf.py, modules/ish.py, modules/sh.py, modules/render_modules.py,
modules/integral_equirect.py, modules/pt_selectors.py, modules/row_mask_sum.py,
modules/safemath.py, modules/tonemap.py (all paths relative to /root/reference; every function
below cites the lines it follows).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it,
and only as the checker / the timed CPU baseline; the product path (nmf_amd/) never does and fails
loudly without its HIP library.

Pinning: the oracle is pinned against the reference itself.  `tests/golden/make_golden.py` (run in
the build container, where /root/reference is mounted) executes the reference with recorded
noise and stores inputs/outputs under tests/golden/*.npz; `tests/test_oracle_golden.py` replays
them through this file (bit-exact for masks / indices / counts, <=1e-5 rel for floats).

Arithmetic lives in torch (pinned 2.10.0): F.grid_sample, F.conv2d, cumsum/cumprod (float64
accumulation on CPU, rounded per element -- SURVEY F14), scatter_add_, argsort.
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional

import torch
import torch.nn.functional as F

EPS = torch.finfo(torch.float32).eps


# ----------------------------------------------------------------------------------------------
# configuration (resolved values of configs/model/microfacet_tensorf2.yaml + field/tensorf_og.yaml)
# ----------------------------------------------------------------------------------------------
@dataclass
class Cfg:
    aabb: torch.Tensor = field(default_factory=lambda: torch.tensor([[-1.5] * 3, [1.5] * 3]))
    grid: int = 128
    near_far: tuple = (2.5, 7.0)
    step_ratio: float = 0.5
    distance_scale: float = 25.0
    density_shift: float = -4.0
    max_samples: int = 200000
    rays_per_ray: int = 128
    max_brdf_rays: tuple = (650000, 450000)
    max_retrace_rays: tuple = (1000,)
    anoise: float = 0.25
    min_rough: float = 0.0
    detach_N: bool = True
    brdf_bias: float = 0.0
    diffuse_bias: float = -0.619
    diffuse_mul: float = 1.5
    roughness_bias: float = -1.0
    tint_bias: float = 0.0
    f0_bias: float = 0.0
    ish_degs: tuple = (0, 1, 2, 4)

    # derived exactly like TensorVoxelBase.update_stepSize (fields/tensor_base.py:219-232)
    def derived(self):
        aabb = self.aabb.float()
        size = aabb[1] - aabb[0]
        gs = torch.LongTensor([self.grid] * 3)
        units = size / (gs - 1)
        stepsize = torch.min(units) * self.step_ratio
        diag = torch.sqrt(torch.sum(torch.square(size)))
        n_samples = int((diag / stepsize).item()) + 1
        return dict(aabb=aabb, size=size, inv=2.0 / size, units=units, stepsize=stepsize,
                    n_samples=n_samples)


class Noise:
    """Noise source.  `tape` = list of recorded (kind, tensor) draws of the reference in call order
    (tests/golden/ref_harness.NoiseTape); without a tape fresh torch draws are produced.  Call
    sites flagged unused=True exist in the reference but are multiplied by zero in this config
    (start_std=0, mipnoise=0); they are consumed from a tape to stay aligned and skipped otherwise."""

    def __init__(self, tape=None, generator=None, draw_unused=False):
        self.tape = list(tape) if tape is not None else None
        self.pos = 0
        self.gen = generator
        self.draw_unused = draw_unused   # True: replay-by-seed (same global RNG call sequence as the reference)

    def draw(self, kind, shape, unused=False):
        shape = tuple(int(s) for s in shape)
        if self.tape is not None:
            k, t = self.tape[self.pos]
            self.pos += 1
            want = "randn" if kind == "randn" else "rand"
            assert k.startswith(want) and (want == "randn") == k.startswith("randn"), (k, kind, self.pos)
            assert tuple(t.shape) == shape, (k, tuple(t.shape), shape, self.pos)
            return t
        if unused and not self.draw_unused:
            return None
        if kind == "randn":
            return torch.randn(shape, generator=self.gen)
        return torch.rand(shape, generator=self.gen)


def normalize(v):
    # mutils.py:8-12 / fields/tensor_base.py:25-29
    return v / (v ** 2).sum(dim=-1, keepdim=True).clip(min=EPS).sqrt()


# ----------------------------------------------------------------------------------------------
# a1-a3: sampler
# ----------------------------------------------------------------------------------------------
def sample_ray(rays_o, rays_d, focal, aabb, stepsize, n_samples, near, far, jitter=None):
    """samplers/alphagrid.py:131-207.  jitter: [B,N] uniform draws (train) or None (eval).
    Returns pts [B,N,4], z [B,N], inbox [B,N] bool."""
    vec = torch.where(rays_d == 0, torch.full_like(rays_d, 1e-6), rays_d)          # :149
    rate_a = (aabb[1] - rays_o) / vec
    rate_b = (aabb[0] - rays_o) / vec
    t_min = torch.minimum(rate_a, rate_b).amax(-1).clamp(min=near, max=far)        # :152
    if jitter is not None:
        steps = jitter * stepsize + stepsize / 2                                    # :169-172
        step = torch.cumsum(steps, dim=1)                                           # :173 (F14)
    else:
        rng = torch.arange(n_samples)[None].float()
        step = stepsize * rng                                                       # :190
    z = t_min[..., None] + step                                                     # :192
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., None]                # :194
    outside = ((aabb[0] > pts) | (pts > aabb[1])).any(dim=-1)                       # :195
    pts = torch.cat([pts, z.unsqueeze(-1) / focal], dim=-1)                         # :200
    return pts, z, ~outside


def alpha_query(alpha_volume, aabb, xyz):
    """AlphaGridMask.sample_alpha, samplers/alphagrid.py:23-45,47-50.  alpha_volume [1,1,D,H,W]."""
    inv = 1.0 / (aabb[1] - aabb[0]) * 2
    coords = (xyz[..., :3] - aabb[0]) * inv - 1
    return F.grid_sample(alpha_volume, coords.view(1, -1, 1, 1, 3), align_corners=True).view(-1)


@torch.no_grad()                                                                     # :278 -- sample positions, z and dists
def sample(rays, focal, cfg: Cfg, alpha_volume, noise: Noise, is_train, override_near=None,   # carry NO gradient to the rays
           dynamic_batch_size=True):
    """AlphaGridSampler.sample, samplers/alphagrid.py:278-370 (N_samples is forced to nSamples,
    SURVEY F7).  Returns xyzs [M,4], ray_valid [b,N], N, z_vals, dists, whole_valid [B].
    The reference decorates it with @torch.no_grad() (:278): the secondary rays' direction L depends on the normals and
    the roughness, but their sample positions / step lengths do not back-propagate into them (only the environment lookup
    and the BRDF inputs do).  Matters in the steady state, where every secondary ray is marched (e2e_full_steady)."""
    d = cfg.derived()
    near, far = cfg.near_far
    if override_near is not None:
        near = override_near
    n = d["n_samples"]
    jitter = noise.draw("rand", (rays.shape[0], n)) if is_train else None
    pts, z, ray_valid = sample_ray(rays[:, :3], rays[:, 3:6], focal, d["aabb"], d["stepsize"], n,
                                   near, far, jitter)
    if alpha_volume is not None:                                                    # :341-346
        alphas = alpha_query(alpha_volume, d["aabb"], pts[ray_valid])
        keep = alphas > 0
        invalid = ~ray_valid
        invalid[ray_valid] |= ~keep
        ray_valid = ~invalid
    dists = torch.cat((z[:, 1:] - z[:, :-1], torch.zeros_like(z[:, :1])), dim=-1)   # :348-350
    B = rays.shape[0]
    if cfg.max_samples > 0 and is_train and dynamic_batch_size and ray_valid.sum() > cfg.max_samples:
        whole_valid = torch.cumsum(ray_valid.sum(dim=1), dim=0) < cfg.max_samples   # :359
        ray_valid, pts, z, dists = ray_valid[whole_valid], pts[whole_valid], z[whole_valid], dists[whole_valid]
    else:
        whole_valid = torch.ones(B, dtype=torch.bool)
    return pts[ray_valid], ray_valid, n, z, dists, whole_valid


def dense_alpha_mask(sd, cfg: Cfg, prev_volume=None, thres=1e-3):
    """updateAlphaMask / getDenseAlpha / compute_alpha, samplers/alphagrid.py:209-276.
    Returns the 0/1 float volume [1,1,G,G,G] (z,y,x order)."""
    d = cfg.derived()
    G = cfg.grid
    lin = torch.linspace(0, 1, G)
    samples = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1)
    dense_xyz = d["aabb"][0] * (1 - samples) + d["aabb"][1] * samples               # :238
    alpha = torch.zeros(G, G, G)
    with torch.no_grad():
        for i in range(G):
            xyz = dense_xyz[i].view(-1, 3)
            if prev_volume is not None:
                m = alpha_query(prev_volume, d["aabb"], xyz) > 0
            else:
                m = torch.ones(xyz.shape[0], dtype=torch.bool)
            sigma = torch.zeros(xyz.shape[0])
            if m.any():
                sigma[m] = density(sd, cfg, xyz[m])
            alpha[i] = (1 - torch.exp(-sigma * d["stepsize"])).view(G, G)          # :222 (no distance_scale)
        alpha = alpha.clamp(0, 1).transpose(0, 2).contiguous()[None, None]          # :253
        alpha = F.max_pool3d(alpha, kernel_size=3, padding=1, stride=1).view(G, G, G)
        alpha[alpha >= thres] = 1
        alpha[alpha < thres] = 0
    return alpha.view(1, 1, G, G, G)


# ----------------------------------------------------------------------------------------------
# a5-a9: TensoRF VM field
# ----------------------------------------------------------------------------------------------
MAT_MODE = ((0, 1), (0, 2), (1, 2))   # fields/tensoRF.py:40
VEC_MODE = (2, 1, 0)                  # fields/tensoRF.py:41


# ----------------------------------------------------------------------------------------------
# a25: grid schedule, step size and factor upsampling
# ----------------------------------------------------------------------------------------------
def n_to_reso(n_voxels, aabb):
    """utils.py:55-58"""
    size = aabb[1] - aabb[0]
    voxel = (size.prod() / n_voxels).pow(1 / 3)
    return (size / voxel).long().tolist()


def voxel_schedule(n_init, n_final, n_upsamples):
    """fields/tensor_base.py:194-203: voxel counts after each scheduled upsample"""
    return (torch.round(torch.linspace(n_init ** (1 / 3), n_final ** (1 / 3), n_upsamples + 1) ** 3).long()).tolist()[1:]


def step_size(aabb, grid_size, step_ratio=0.5):
    """TensorVoxelBase.update_stepSize (fields/tensor_base.py:219-232) -> units [3], stepsize (0-d fp32), nSamples"""
    gs = torch.as_tensor(grid_size, dtype=torch.long)
    size = aabb[1] - aabb[0]
    units = size / (gs - 1)
    stepsize = torch.min(units) * step_ratio
    diag = torch.sqrt(torch.sum(torch.square(size)))
    return units, stepsize, int((diag / stepsize).item()) + 1


def upsample_factors(planes, lines, res_target):
    """TensoRF.upsample (fields/tensoRF.py:207-227): bilinear, align_corners=True; plane i spans
    (matMode[i][1], matMode[i][0]) = (H, W), line i runs along vecMode[i]."""
    mat, vec = [[0, 1], [0, 2], [1, 2]], [2, 1, 0]
    new_p = [F.interpolate(planes[i], size=(res_target[mat[i][1]], res_target[mat[i][0]]), mode="bilinear",
                           align_corners=True) for i in range(3)]
    new_l = [F.interpolate(lines[i], size=(res_target[vec[i]], 1), mode="bilinear", align_corners=True) for i in range(3)]
    return new_p, new_l


def normalize_coord(cfg: Cfg, xyz):
    # fields/tensor_base.py:66-69
    d = cfg.derived()
    coords = (xyz[..., :3] - d["aabb"][0]) * d["inv"] - 1
    return torch.cat((coords, xyz[..., 3:4]), dim=-1)


def _plane_line_coords(xn):
    # fields/tensoRF.py:161-179
    cp = torch.stack([xn[..., list(m)] for m in MAT_MODE]).view(3, -1, 1, 2)
    cl = torch.stack([xn[..., v] for v in VEC_MODE])
    cl = torch.stack((torch.zeros_like(cl), cl), dim=-1).view(3, -1, 1, 2)
    return cp, cl


def _gs(inp, grid):
    return F.grid_sample(inp, grid, mode="bilinear", padding_mode="zeros", align_corners=True)


def vm_products(planes, lines, xn):
    """TensoRF.forward, fields/tensoRF.py:181-205: list of 3 tensors [C,M] = plane_i * line_i."""
    cp, cl = _plane_line_coords(xn)
    M = xn.shape[0]
    out = []
    for i in range(3):
        pc = _gs(planes[i], cp[[i]]).view(-1, M)
        lc = _gs(lines[i], cl[[i]]).view(-1, M)
        out.append(pc * lc)
    return out


def _density_factors(sd):
    return ([sd[f"rf.density_rf.app_plane.{i}"] for i in range(3)],
            [sd[f"rf.density_rf.app_line.{i}"] for i in range(3)])


def _app_factors(sd):
    return ([sd[f"rf.app_rf.app_plane.{i}"] for i in range(3)],
            [sd[f"rf.app_rf.app_line.{i}"] for i in range(3)])


def feature2density(cfg: Cfg, f):
    # fields/tensor_base.py:83-85
    return F.softplus(f.clamp(-15, 1e3) + cfg.density_shift)


def density_feature(sd, cfg: Cfg, xyz):
    """_compute_densityfeature with dbasis=False, fields/tensoRF.py:392-400."""
    planes, lines = _density_factors(sd)
    feats = vm_products(planes, lines, normalize_coord(cfg, xyz))
    return sum(feats).sum(dim=0)


def density(sd, cfg: Cfg, xyz):
    # TensorBase.compute_densityfeature, fields/tensor_base.py:131-138
    if xyz.shape[0] == 0:
        return torch.empty(0)
    return feature2density(cfg, density_feature(sd, cfg, xyz)).reshape(-1)


def app_feature(sd, cfg: Cfg, xyz):
    """_compute_appfeature, fields/tensoRF.py:402-405."""
    planes, lines = _app_factors(sd)
    coefs = torch.cat(vm_products(planes, lines, normalize_coord(cfg, xyz)), dim=0).T
    return coefs @ sd["rf.basis_mat.weight"].T


def derivative_stencils():
    """The two 5x5 cross-correlation stencils of GridSampler2D.backward for smoothing=1
    (modules/grid_sample_Cinf.py:117-118,218-233,49-63): a normalised 3x3 Gaussian (std 1)
    combined with SIGN*[1,0,-1]/2 by `-conv2d(k1, k2, padding=2)`.  Returns (dx, dy) [1,1,5,5]."""
    f_blur = torch.tensor([0.0, 1.0, 0.0])
    f_edge = -1 * torch.tensor([1, 0.0, -1]) / 2
    dy_filter = (f_blur[None, :] * f_edge[:, None]).reshape(1, 1, 3, 3)
    dx_filter = dy_filter.permute(0, 1, 3, 2)
    n = torch.arange(0, 3) - 1.0
    g1 = torch.exp(-(n ** 2) / 2.0)
    smooth = torch.outer(g1, g1)
    smooth = smooth / smooth.sum()

    def combine(k1, k2):
        return -F.conv2d(k1.reshape(1, 1, 3, 3), k2.reshape(1, 1, 3, 3), stride=1, padding=2)

    return combine(smooth, dx_filter), combine(smooth, dy_filter)


def density_gradient(sd, cfg: Cfg, xyz):
    """d(sigma_feat)/d(xyz) as the reference's custom backward defines it
    (fields/tensor_base.py:107-129 + modules/grid_sample_Cinf.py:109-325): the derivative of a
    bilinear plane sample wrt its grid coordinate is the bilinear sample of the stencil-filtered
    plane (per-texel units); for a line only the centre stencil column overlaps.
    Written as explicit differentiable ops, so autograd of this function reproduces the
    reference's second-order path.  Returns [M,3] (world-space, includes 2/aabbSize)."""
    planes, lines = _density_factors(sd)
    xn = normalize_coord(cfg, xyz)
    cp, cl = _plane_line_coords(xn)
    M = xyz.shape[0]
    kx, ky = derivative_stencils()
    g = [0, 0, 0]
    for i in range(3):
        P, L = planes[i], lines[i]
        pc = _gs(P, cp[[i]]).view(-1, M)
        lc = _gs(L, cl[[i]]).view(-1, M)
        Pc = P.permute(1, 0, 2, 3)
        dxp = _gs(F.conv2d(Pc, kx, padding=2).permute(1, 0, 2, 3), cp[[i]]).view(-1, M)
        dyp = _gs(F.conv2d(Pc, ky, padding=2).permute(1, 0, 2, 3), cp[[i]]).view(-1, M)
        dyl = _gs(F.conv2d(L.permute(1, 0, 2, 3), ky, padding=2).permute(1, 0, 2, 3), cl[[i]]).view(-1, M)
        a, b = MAT_MODE[i]
        g[a] = g[a] + (lc * dxp).sum(0)
        g[b] = g[b] + (lc * dyp).sum(0)
        g[VEC_MODE[i]] = g[VEC_MODE[i]] + (pc * dyl).sum(0)
    return torch.stack(g, dim=-1) * cfg.derived()["inv"]


def normals(sd, cfg: Cfg, xyz):
    # fields/tensor_base.py:128
    return normalize(-density_gradient(sd, cfg, xyz))


# ----------------------------------------------------------------------------------------------
# a10-a11: compositing
# ----------------------------------------------------------------------------------------------
def raw2alpha(sigma, dist):
    # modules/tensor_nerf.py:19-35
    alpha = 1.0 - torch.exp(-sigma * dist)
    T = torch.cumprod(torch.cat([torch.ones(alpha.shape[0], 1), 1.0 - alpha + 1e-10], dim=-1), dim=-1)
    return alpha * T[:, :-1]


def row_mask_sum(mat, mask):
    # modules/row_mask_sum.py:15-22
    B = mask.shape[0]
    idx = torch.where(mask)[0]
    out = torch.zeros((B, mat.shape[1]), dtype=mat.dtype)
    out.scatter_add_(0, idx[:, None].expand(-1, mat.shape[1]), mat)
    return out


def srgb_tonemap(img, noclip=False):
    # modules/tonemap.py:38-49
    limit = 0.0031308
    out = torch.where(img > limit, 1.055 * (img.clip(min=limit) ** (1.0 / 2.4)) - 0.055, 12.92 * img)
    return out if noclip else out.clip(0, 1)


# ----------------------------------------------------------------------------------------------
# a12: material heads
# ----------------------------------------------------------------------------------------------
def material_heads(sd, cfg: Cfg, feat):
    """RandHydraMLPDiffuse.forward with pospe=-1, feape=0, std=0
    (modules/render_modules.py:519-574).  Returns albedo, tint, f0 [M,3], r [M,2]."""
    p = "model.diffuse_module."

    def lin(name):
        # nn.Linear = F.linear (one addmm with the bias as the accumulator's start value: rounds differently from
        # `x @ W.T + b`, and the re-trace ORDER of the steady state is sensitive to the last bit of every score input)
        return F.linear(feat, sd[p + name + "_mlp.0.weight"], sd[p + name + "_mlp.0.bias"])

    albedo = torch.sigmoid(cfg.diffuse_mul * lin("diffuse") + cfg.diffuse_bias).clip(min=0, max=1)
    r = (torch.sigmoid(lin("roughness") + cfg.roughness_bias) / 2).clip(min=1e-2, max=1)
    tint = torch.sigmoid(lin("tint") + cfg.tint_bias)
    f0 = torch.sigmoid(lin("f0") + cfg.f0_bias)
    return albedo, tint, f0, r


# ----------------------------------------------------------------------------------------------
# spherical harmonics (modules/sh.py)
# ----------------------------------------------------------------------------------------------
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
# modules/sh.py:67-73 -- the SH_C2 table that eval_sh_bases reads has all-positive entries
# (unlike the signed `C2` list at :10-16, which is unused on this path).
SH_C2 = (1.0925484305920792, 1.0925484305920792, 0.31539156525252005, 1.0925484305920792,
         0.5462742152960396)


def eval_sh9(dirs):
    """eval_sh_bases(9, dirs), modules/sh.py:97-142 (first 9 bases)."""
    x, y, z = dirs.unbind(-1)
    xx, yy, zz = x * x, y * y, z * z
    return torch.stack([
        torch.full_like(x, SH_C0), SH_C1 * y, SH_C1 * z, SH_C1 * x,
        SH_C2[0] * (x * y), SH_C2[1] * (y * z), SH_C2[2] * (3 * zz - 1), SH_C2[3] * (x * z),
        SH_C2[4] * (xx - yy)], dim=-1)


def al2(l):
    # modules/sh.py:149-157
    if l == 0:
        return math.pi
    if l == 1:
        return 2 * math.pi / 3
    if l % 2 == 1:
        return 0
    return 2 * math.pi * (-1) ** (l / 2 - 1) / ((l + 2) * (l - 1)) * (
        math.factorial(l) / (2 ** l * math.factorial(l // 2) ** 2))


def ish_basis(degs, dirs, kappa):
    """sh_basis(degs, dirs, kappa), modules/sh.py:251-308, degrees 0,1,2,4 only (the quirks are
    the reference's: degree-2 entry 3 is -x*y, degree 4 is not attenuated)."""
    kappa = kappa.reshape(-1)
    x, y, z = dirs.T[0], dirs.T[1], dirs.T[2]
    xx, yy, zz = x * x, y * y, z * z
    x4, y4, z4 = x ** 4, y ** 4, z ** 4
    vals = []
    for deg in degs:
        scale = torch.exp(-deg * (deg + 1) / 2 / (kappa + 1e-8))     # Al, modules/sh.py:145-147
        if deg == 0:
            vals.append(scale * 0.28209479177387814 * torch.ones_like(x))
        elif deg == 1:
            vals += [-scale * 0.488603 * x, scale * 0.488603 * z, -scale * 0.488603 * y]
        elif deg == 2:
            vals += [scale * 1.092548 * y * x, -scale * 1.092548 * y * z, scale * 0.315392 * (3 * zz - 1),
                     -scale * 1.092548 * x * y, scale * 0.546274 * (xx - yy)]
        elif deg == 4:
            vals += [2.50334 * x * y * (xx - yy), -1.77013 * y * z * (-3 * xx + yy),
                     0.946175 * x * y * (7 * zz - 1), 0.669047 * y * z * (7 * zz - 3),
                     3.70251 * z4 - 3.17358 * zz + 0.317358, 0.669047 * x * z * (7 * zz - 3),
                     (0.473087 * xx - 0.473087 * yy) * (7 * zz - 1), 1.77013 * x * z * (xx - 3 * yy),
                     0.625836 * x4 - 3.755016 * xx * yy + 0.625836 * y4]
        else:
            raise ValueError(deg)
    return torch.stack(vals, dim=-1)


# ----------------------------------------------------------------------------------------------
# a19: environment map (modules/integral_equirect.py)
# ----------------------------------------------------------------------------------------------
class _SafeAtan2(torch.autograd.Function):
    # modules/safemath.py:8-30
    @staticmethod
    def forward(ctx, x, y):
        ctx.save_for_backward(x, y)
        return torch.atan2(x, y)

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        den = x ** 2 + y ** 2 + 1e-5
        return g * y / den, g * -x / den


def env_activation(sd):
    # IntegralEquirect.activation_fn with activation='exp', modules/integral_equirect.py:263-273
    x = sd["bg_module.brightness"] + sd["bg_module.mul"] * sd["bg_module.bg_mat"]
    return torch.exp(x.clip(max=20))


def env_sat(sd):
    # modules/integral_equirect.py:431-433 (float64-accumulated prefix sums)
    return torch.cumsum(torch.cumsum(env_activation(sd) / 1000, dim=2), dim=3)


def _box(bl, br, tl, tr, size, sat):
    # integrate_area, modules/integral_equirect.py:18-39
    def s(p):
        return F.grid_sample(sat, p.clip(min=-1, max=1), mode="bilinear", align_corners=True,
                             padding_mode="zeros")
    return (s(tr) + s(bl) - s(tl) - s(br)).reshape(3, -1).T / size


def _box_lr(bl, br, tl, tr, size, sat):
    # integrate_area_wrap_lr, modules/integral_equirect.py:42-93 (corners are [1,1,n,2])
    vals = _box(bl, br, tl, tr, size, sat)
    ex = (tr[..., 0] > 1).reshape(-1)
    if ex.any():
        def pick(p):
            return p[0, 0, ex].clone().reshape(1, 1, -1, 2)
        bl_r, tl_r, tr_r, br_r = pick(bl), pick(tl), pick(tr), pick(br)
        bl_r[..., 0] = -1.0
        tl_r[..., 0] = -1.0
        tr_r[..., 0] = tr_r[..., 0] - 2
        br_r[..., 0] = br_r[..., 0] - 2
        add = _box(bl_r, br_r, tl_r, tr_r, size[ex], sat)
        vals = vals.index_add(0, torch.where(ex)[0], add)
    ex = (bl[..., 0] < -1).reshape(-1)
    if ex.any():
        def pick(p):
            return p[0, 0, ex].clone().reshape(1, 1, -1, 2)
        bl_l, tl_l, tr_l, br_l = pick(bl), pick(tl), pick(tr), pick(br)
        bl_l[..., 0] = bl_l[..., 0] + 2
        tl_l[..., 0] = tl_l[..., 0] + 2
        tr_l[..., 0] = 1.0
        br_l[..., 0] = 1.0
        add = _box(bl_l, br_l, tl_l, tr_l, size[ex], sat)
        vals = vals.index_add(0, torch.where(ex)[0], add)
    return vals


def _box_wrap(bl, br, tl, tr, size, sat):
    # integrate_area_wrap, modules/integral_equirect.py:96-173
    vals = _box_lr(bl, br, tl, tr, size, sat)
    ex = (tl[..., 1] > 1).reshape(-1)
    if ex.any():
        def pick(p):
            return p[0, 0, ex].clone().reshape(1, 1, -1, 2)
        tl_t, tr_t, bl_t, br_t = pick(tl), pick(tr), pick(bl), pick(br)
        rot = torch.where(tl_t[..., 0] > 0, -1.0, 1.0)
        over = (tl_t[..., 1] - 1).clip(max=0.5, min=0)
        tl_t = torch.stack([tl_t[..., 0] + rot, torch.ones_like(rot)], -1)
        tr_t = torch.stack([tr_t[..., 0] + rot, torch.ones_like(rot)], -1)
        bl_t = torch.stack([bl_t[..., 0] + rot, 1 - over], -1)
        br_t = torch.stack([br_t[..., 0] + rot, 1 - over], -1)
        add = _box_lr(bl_t, br_t, tl_t, tr_t, size[ex], sat)
        vals = vals.index_add(0, torch.where(ex)[0], add)
    ex = (bl[..., 1] < -1).reshape(-1)
    if ex.any():
        def pick(p):
            return p[0, 0, ex].clone().reshape(1, 1, -1, 2)
        tl_b, tr_b, bl_b, br_b = pick(tl), pick(tr), pick(bl), pick(br)
        rot = torch.where(tl_b[..., 0] > 0, -1.0, 1.0)
        over = (-1 - bl_b[..., 1]).clip(max=0.5, min=0)
        bl_b = torch.stack([bl_b[..., 0] + rot, -torch.ones_like(rot)], -1)
        br_b = torch.stack([br_b[..., 0] + rot, -torch.ones_like(rot)], -1)
        tl_b = torch.stack([tl_b[..., 0] + rot, -1 + over], -1)
        tr_b = torch.stack([tr_b[..., 0] + rot, -1 + over], -1)
        add = _box_lr(bl_b, br_b, tl_b, tr_b, size[ex], sat)
        vals = vals.index_add(0, torch.where(ex)[0], add)
    return vals


def env_mip_levels(sd, u, sa):
    # sa2mip, modules/integral_equirect.py:373-397 (mipnoise = 0)
    h, w = sd["bg_module.bg_mat"].shape[-2:]
    sa = sa.reshape(-1)
    cos = (1 - u[:, 2] ** 2).clip(min=EPS).sqrt()
    d = h * w / (2 * math.pi ** 2 * cos).clip(min=EPS)
    area = ((d / 2).log() + sa).exp()
    hh = (area.clip(min=EPS).sqrt() * cos).clip(min=EPS)
    ww = area / hh
    mw = ww.log() / math.log(2) + sd["bg_module.mipbias"]
    mh = hh.log() / math.log(2) + sd["bg_module.mipbias"]
    return mw.clip(0, 7), mh.clip(0, 7)


def env_lookup(sd, dirs, sa, sat=None):
    """IntegralEquirect.forward, modules/integral_equirect.py:409-504."""
    if dirs.shape[0] == 0:
        return torch.empty((0, 3))
    h, w = sd["bg_module.bg_mat"].shape[-2:]
    mw, mh = env_mip_levels(sd, dirs, sa)
    sw = 2 ** mw / h / 2
    sh = 2 ** mh / h
    offset = torch.stack([sw, sh], dim=-1).reshape(1, 1, -1, 2)
    activated = env_activation(sd)
    if sat is None:
        sat = torch.cumsum(torch.cumsum(activated / 1000, dim=2), dim=3)
    size = (offset / 2 * torch.tensor([w, h]).reshape(1, 1, 1, 2)).prod(dim=-1).reshape(-1, 1)
    a, b, c = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    norm2d = torch.sqrt(a ** 2 + b ** 2)
    phi = _SafeAtan2.apply(b, a)
    theta = _SafeAtan2.apply(c, norm2d)
    coords = torch.cat([(phi % (2 * math.pi) - math.pi) / math.pi, -theta / math.pi * 2], dim=1)
    x = coords.reshape(1, 1, -1, 2)
    bl = x - offset / 2
    tr = x + offset / 2
    br = x + torch.stack([sw, -sh], dim=-1).reshape(1, 1, -1, 2) / 2
    tl = x + torch.stack([-sw, sh], dim=-1).reshape(1, 1, -1, 2) / 2
    vals = _box_wrap(bl, br, tl, tr, size, sat) * 1000
    cutoff = 1 - 2 / h * 3
    top_row = activated[..., 0, :].mean(dim=-1)
    bot_row = activated[..., -1, :].mean(dim=-1)
    vals = torch.where((coords[:, 1] > cutoff)[:, None], bot_row.expand_as(vals), vals)
    vals = torch.where((coords[:, 1] < -cutoff)[:, None], top_row.expand_as(vals), vals)
    return vals


def env_sh_irradiance(sd, noise: Noise, G=100, mipval=-5):
    """get_spherical_harmonics + the clamped-cosine convolution
    (modules/integral_equirect.py:324-360, sh_A :229-230).  Returns conv_coeffs/pi [9,3]."""
    _theta = torch.linspace(0, math.pi, G // 2)
    _phi = torch.linspace(0, 2 * math.pi, G)
    theta, phi = torch.meshgrid(_theta, _phi, indexing="ij")
    dirs = torch.stack([torch.sin(theta) * torch.cos(phi), torch.sin(theta) * torch.sin(phi),
                        torch.cos(theta)], dim=-1).reshape(-1, 3)
    SB = dirs.shape[0]
    noise.draw("rand", (SB,), unused=True)
    noise.draw("rand", (SB,), unused=True)
    bg = env_lookup(sd, dirs, mipval * torch.ones(SB, 1))
    ev = eval_sh9(dirs)
    coeffs = 2 * math.pi ** 2 * (bg.reshape(SB, 1, 3) * ev.reshape(SB, -1, 1)
                                 * torch.sin(theta.reshape(SB, 1, 1))).mean(dim=0)
    sh_A = torch.tensor(sum([[al2(l)] * (2 * l + 1) for l in range(3)], []), dtype=torch.float32)
    return coeffs, sh_A.reshape(-1, 1) * coeffs / math.pi


# ----------------------------------------------------------------------------------------------
# a14: bounce selection
# ----------------------------------------------------------------------------------------------
def select_bounces(weights, app_mask, num_rays, rays_per_ray, noise: Noise):
    """modules/pt_selectors.py:5-60.  rays_per_ray=None selects the recursion>=1 branch."""
    with torch.no_grad():
        if rays_per_ray is not None:
            pt = weights[app_mask]
            pt = pt * rays_per_ray + noise.draw("rand", pt.shape) - 0.5
        else:
            w = weights + 1e-3 * noise.draw("rand", weights.shape)
            N = num_rays - app_mask.sum()
            if N > 0:
                pt = w / (w.sum().clip(min=1e-3)) * N + 1
            else:
                pt = w / (w.sum().clip(min=1e-3)) * num_rays + 0.5
            pt = pt[app_mask]
        num = pt.floor().max().clip(min=0, max=400).int()
        ray_mask = torch.arange(num).reshape(1, -1) < pt.reshape(-1, 1).floor()
        bounce_mask = ray_mask.sum(dim=-1) > 0
        return bounce_mask, ray_mask[bounce_mask]


# ----------------------------------------------------------------------------------------------
# a15-a17: GGX visible-normal sampling (brdf_samplers/ggx.py, base.py)
# ----------------------------------------------------------------------------------------------
def sobol_draw(angs, n_rows, m, noise: Noise):
    # PseudoRandomSampler.draw, brdf_samplers/base.py:11-20
    a = angs.reshape(1, -1, 2)[:, :m, :].expand(n_rows, m, 2)
    offset = noise.draw("rand", (n_rows, 1, 2)) * 0.25
    return (a + offset) % 1.0


def _safe_trig(x, fn):
    return fn(x % (100 * math.pi))    # modules/safemath.py:34-46


def ggx_prob(l_in, l_out, half, r):
    # GGXSampler.compute_prob, brdf_samplers/ggx.py:228-268 (isotropic: r2 = r1)
    r2 = r.reshape(-1).clip(min=EPS)
    r1 = (r.reshape(-1) + r2).clip(min=EPS) / 2
    lam = (-1 + (1 + ((l_in[:, 0] * r1) ** 2 + (l_in[:, 1] * r2) ** 2)
                 / (l_in[:, 2] ** 2).clip(min=1e-6)).clip(min=EPS).sqrt()) / 2
    invG = 1 + lam
    invD = math.pi * r1 * r2 * (half[:, 0] ** 2 / r1 ** 2 + half[:, 1] ** 2 / r2 ** 2 + half[:, 2] ** 2) ** 2
    logD = -(invG * invD).clip(min=EPS).log() - (4 * l_out[..., 2]).clip(min=EPS).log()
    prob = logD.exp().reshape(-1, 1)
    return torch.where(l_in[:, 2:3] > 0, prob, torch.zeros_like(prob))


def ggx_sample(u1, u2, V, N, r, ray_mask):
    """GGXSampler.sample, brdf_samplers/ggx.py:61-226.  V,N [Mb,3], r [Mb,1], u [Mb,m].
    Returns L [R,3], basisT [R,3,3] (columns = tangent, bitangent, normal), logpdf [R]."""
    Mb, m = u1.shape
    z_up = torch.tensor([0.0, 0.0, 1.0]).reshape(1, 3).expand(Mb, 3)
    x_up = torch.tensor([-1.0, 0.0, 0.0]).reshape(1, 3).expand(Mb, 3)
    up = torch.where(N[:, 2:3].abs() < 0.999, z_up, x_up)
    tangent = normalize(torch.linalg.cross(up, N))
    bitangent = normalize(torch.linalg.cross(N, tangent))
    basis = torch.stack([tangent, bitangent, N], dim=1)                            # rows
    V_l = torch.matmul(basis, V.unsqueeze(-1)).squeeze(-1)
    rc = r.squeeze(-1)
    Vs = normalize(torch.stack([rc * V_l[..., 0], rc * V_l[..., 1], V_l[..., 2]], dim=-1)).unsqueeze(1)
    T1 = torch.where(Vs[..., 2:3] < 0.999,
                     normalize(torch.linalg.cross(Vs, z_up.unsqueeze(1), dim=-1)), x_up.unsqueeze(1))
    T2 = normalize(torch.linalg.cross(T1, Vs, dim=-1))
    z = Vs[..., 2].reshape(-1, 1)
    a = (1 / (1 + z.detach()).clip(min=1e-8)).clip(max=1e4)

    def ex(t):
        return t.expand(Mb, m, *t.shape[2:])[ray_mask]

    a_m = a.expand(Mb, m)[ray_mask]
    r_m = rc.reshape(-1, 1).expand(Mb, m)[ray_mask]
    z_m = z.expand(Mb, m)[ray_mask]
    u1_m, u2_m = u1[ray_mask], u2[ray_mask]
    T1_m, T2_m, Vs_m = ex(T1), ex(T2), ex(Vs)
    basisT = basis.permute(0, 2, 1).reshape(Mb, 1, 3, 3).expand(Mb, m, 3, 3)[ray_mask]
    rr = torch.sqrt(u1_m)
    phi = torch.where(u2_m < a_m, u2_m / a_m * math.pi, (u2_m - a_m) / (1 - a_m) * math.pi + math.pi)
    P1 = (rr * _safe_trig(phi, torch.cos)).unsqueeze(-1)
    P2 = (rr * _safe_trig(phi, torch.sin) * torch.where(u2_m < a_m, torch.tensor(1.0), z_m)).unsqueeze(-1)
    Ns = P1 * T1_m + P2 * T2_m + (1 - P1 * P1 - P2 * P2).clip(min=EPS).sqrt() * Vs_m
    H_l = normalize(torch.stack([Ns[..., 0] * r_m, Ns[..., 1] * r_m, Ns[..., 2]], dim=-1))
    H = torch.matmul(basisT, H_l.unsqueeze(-1)).squeeze(-1)
    w_o = V.unsqueeze(1).expand(Mb, m, 3)[ray_mask]
    eN = N.unsqueeze(1).expand(Mb, m, 3)[ray_mask]
    w_i = normalize(2.0 * (w_o * H).sum(dim=-1, keepdim=True) * H - w_o)
    sign = torch.where((w_i * eN).sum(dim=-1, keepdim=True) > 0, 1, -1)
    w_i = w_i * sign
    lw_i = torch.matmul(basisT.permute(0, 2, 1), w_i.unsqueeze(-1)).squeeze(-1)
    lw_o = torch.matmul(basisT.permute(0, 2, 1), w_o.unsqueeze(-1)).squeeze(-1)
    with torch.no_grad():
        logp = ggx_prob(lw_i, lw_o, H_l, r_m).clip(min=EPS).log().reshape(-1)
    return w_i, basisT, logp


# ----------------------------------------------------------------------------------------------
# a18: BRDF MLP (modules/brdf.py:177-261 with feape=0, dotpe=-1, h/d encoders = ListISH)
# ----------------------------------------------------------------------------------------------
def brdf_mlp(sd, cfg: Cfg, half_vec, diff_vec, feat, rough):
    kappa = 1 / (rough + 1e-3)                                      # modules/ish.py:103
    x = torch.cat([feat, ish_basis(cfg.ish_degs, half_vec, kappa), half_vec,
                   ish_basis(cfg.ish_degs, diff_vec, kappa), diff_vec], dim=-1)
    p = "model.brdf.mlp."
    h = torch.relu(F.linear(x, sd[p + "0.weight"], sd[p + "0.bias"]))       # nn.Sequential(Linear, ReLU, ...) = F.linear
    h = torch.relu(F.linear(h, sd[p + "2.weight"], sd[p + "2.bias"]))
    o = F.linear(h, sd[p + "4.weight"], sd[p + "4.bias"])
    return torch.sigmoid(o[..., :3] + cfg.brdf_bias)


# ----------------------------------------------------------------------------------------------
# a20: retrace selection (models/microfacet.py:475-537)
# ----------------------------------------------------------------------------------------------
def retrace_select(brdf_weight, eV, eN, samp_prob, w_bounce, ray_count, ray_mask, num_retrace, noise: Noise,
                   forced_order=None):
    """forced_order: the argsort result recorded from the reference (tests/golden BookkeepingTap).  The scores are
    contribution + U(0,1) in fp32; their inputs (normals -> GGX directions -> BRDF weights) agree with the reference to
    ~1 ulp, not bit for bit, so ~0.1 % of neighbouring rays swap places in the sort -- harmless when a few rays are selected,
    but in the steady state (every ray re-traced) the order pairs each ray with a jitter row of the recursive render."""
    with torch.no_grad():
        ri, rj = torch.where(ray_mask)
        per_sample = w_bounce.reshape(-1, 1) / ray_count
        per_ray = (brdf_weight.max(dim=-1, keepdim=True).values
                   * ((eV * eN).sum(dim=-1, keepdim=True) > 0) * samp_prob)
        cc = per_ray.reshape(-1) * per_sample.expand(ray_mask.shape)[ri, rj]
        cc = cc / cc.sum() * num_retrace
        cc = cc + noise.draw("rand", cc.shape)
        order = cc.argsort()
        if forced_order is not None:
            assert forced_order.shape == order.shape
            own, order = order, forced_order.long()
            cc = (cc, own)
        M = max(order.shape[0] - num_retrace, 0)
        if isinstance(cc, tuple):
            return order[M:], order[:M], (cc[0], order, cc[1])
        return order[M:], order[:M], (cc, order, order)


# ----------------------------------------------------------------------------------------------
# a21: shading (models/microfacet.py:271-673, diffuse_mixing_mode='fresnel')
# ----------------------------------------------------------------------------------------------
def shade(sd, cfg: Cfg, xyzs, app_features, viewdirs, nrm, weights, app_mask, render_reflection,
          noise: Noise, is_train, recur, trace=None, forced=None):
    M = xyzs.shape[0]
    noise_feat = app_features + noise.draw("randn", app_features.shape) * cfg.anoise       # :297
    noise.draw("randn", (M, 3), unused=True)
    noise.draw("randn", (M, 2), unused=True)
    albedo, tint, f0, r = material_heads(sd, cfg, app_features)                             # :299
    with torch.no_grad():                                                                   # :304-315
        _, conv = env_sh_irradiance(sd, noise)
        E = (conv.reshape(1, -1, 3) * eval_sh9(nrm).reshape(M, -1, 1)).sum(dim=1).detach()
    diffuse = albedo * E
    bounce_mask, ray_mask = select_bounces(weights, app_mask, cfg.max_brdf_rays[recur],
                                           cfg.rays_per_ray if recur == 0 else None, noise)  # :327
    reflect_rgb = torch.zeros_like(diffuse)
    brdf_rgb = torch.zeros_like(diffuse)
    spec = torch.zeros_like(diffuse)
    if trace is not None:
        trace.update({f"bounce_mask{recur}": bounce_mask, f"ray_mask{recur}": ray_mask})
    if bounce_mask.any() and ray_mask.any():
        ri, rj = torch.where(ray_mask)
        n_b, m = ray_mask.shape
        bN = nrm[bounce_mask]
        if cfg.detach_N:
            bN = bN.detach()
        bV = -viewdirs[bounce_mask]
        bN = bN * (bV * bN).sum(dim=-1, keepdim=True).sign()                                # :356
        r1 = r[:, 0:1][bounce_mask]
        if is_train:
            r1 = r1.clip(min=cfg.min_rough)
        angs = sobol_draw(sd["model.brdf_sampler.angs"], n_b, m, noise)                     # :367
        L, basisT, lpdf = ggx_sample(angs[..., 0], angs[..., 1], bV, bN, r1, ray_mask)      # :370
        eV = bV.reshape(-1, 1, 3).expand(-1, m, 3)[ri, rj]
        eN = bN.reshape(-1, 1, 3).expand(-1, m, 3)[ri, rj]
        ea = r1.expand(ray_mask.shape)[ri, rj]
        efeat = noise_feat[bounce_mask].reshape(n_b, 1, -1).expand(n_b, m, -1)[ri, rj]
        exyz = xyzs[bounce_mask][..., :3].reshape(-1, 1, 3).expand(-1, m, 3)[ri, rj]
        H = normalize((eV + L) / 2)                                                         # :388
        basis_rows = basisT.permute(0, 2, 1)
        diffvec = torch.matmul(basis_rows, L.unsqueeze(-1)).squeeze(-1)                     # :406
        halfvec = torch.matmul(basis_rows, H.unsqueeze(-1)).squeeze(-1)                     # :426
        samp_prob = lpdf.exp().reshape(-1, 1)
        counts = ray_mask.sum(dim=1, keepdim=True).expand(ray_mask.shape)[ray_mask]
        mipval = -torch.log(counts.clip(min=1)) - lpdf                                      # :448
        bounce_rays = torch.cat([exyz + L * 5e-3, L], dim=-1)                               # :450
        brdf_weight = brdf_mlp(sd, cfg, halfvec.detach(), diffvec.detach(), efeat, ea.detach())  # :461
        ray_count = (ray_mask.sum(dim=1) + 1e-8)[..., None]
        if len(cfg.max_retrace_rays) > recur:
            num_retrace = min(brdf_weight.shape[0], cfg.max_retrace_rays[recur])
            idx_re, idx_no, cc = retrace_select(brdf_weight, eV, eN, samp_prob,
                                                weights[app_mask][bounce_mask], ray_count, ray_mask,
                                                num_retrace, noise,
                                                None if forced is None else forced.get(f"retrace_order{recur}"))
            if trace is not None:
                trace.update({f"retrace_idx{recur}": idx_re, f"retrace_score{recur}": cc[0],
                              f"retrace_order{recur}": cc[1], f"retrace_order_own{recur}": cc[2]})
            incoming = torch.zeros((bounce_rays.shape[0], 3))
            if len(idx_re) > 0:
                inc = render_reflection(bounce_rays[idx_re], mipval[idx_re], True)
                incoming = incoming.index_put((idx_re,), inc)
            if len(idx_no) > 0:
                inc = render_reflection(bounce_rays[idx_no], mipval[idx_no], False)
                incoming = incoming.index_put((idx_no,), inc)
        else:
            incoming = render_reflection(bounce_rays, mipval, False)
        ecount = ray_count.reshape(-1, 1).expand(ray_mask.shape)[ray_mask].reshape(-1, 1).clip(min=1)
        spec = spec.index_put((torch.where(bounce_mask)[0],), row_mask_sum(incoming / ecount, ray_mask))
        brdf_rgb = brdf_rgb.index_put((torch.where(bounce_mask)[0],), row_mask_sum(brdf_weight / ecount, ray_mask))
        R0 = f0[bounce_mask].reshape(-1, 1, 3).expand(-1, m, 3)[ri, rj]                     # :596
        ediff = diffuse[bounce_mask].reshape(-1, 1, 3).expand(-1, m, 3)[ri, rj]
        cos_t = (-eV * H).sum(dim=-1, keepdim=True).abs()
        Fr = R0 + (1 - R0) * (1 - cos_t).clip(min=0, max=1) ** 5
        comb = Fr * incoming * brdf_weight + (1 - Fr) * ediff                               # :609
        reflect_rgb = reflect_rgb.index_put((torch.where(bounce_mask)[0],), row_mask_sum(comb / ecount, ray_mask))
        if trace is not None:
            trace.update({f"L{recur}": L, f"mipval{recur}": mipval, f"brdf_weight{recur}": brdf_weight,
                          f"incoming{recur}": incoming})
    cos_t = (-viewdirs * nrm).sum(dim=-1, keepdim=True).abs()                               # :642
    Fr = f0 + (1 - f0) * (1 - cos_t).clip(min=0, max=1) ** 5
    debug = dict(diffuse=(1 - Fr) * diffuse, tint=Fr * brdf_rgb, roughness=r[:, 0:1], spec=spec,
                 albedo=albedo)
    return reflect_rgb, debug


# ----------------------------------------------------------------------------------------------
# a22: the scene module forward (modules/tensor_nerf.py:210-674)
# ----------------------------------------------------------------------------------------------
def render(sd, cfg: Cfg, rays, focal, alpha_volume, noise: Noise, is_train=True, recur=0,
           bg_col=None, start_mipval=None, override_near=None, dynamic_batch_size=True, tonemap=True,
           trace=None, forced=None):
    """Returns (images, stats) with the reference's key names.  forced: {"retrace_order0": ...} see retrace_select."""
    d = cfg.derived()
    xyz, ray_valid, N, z_vals, dists, whole_valid = sample(
        rays, focal, cfg, alpha_volume, noise, is_train, override_near, dynamic_batch_size)
    B = ray_valid.shape[0]
    M = xyz.shape[0]
    n_samples = [M]
    viewdirs = rays[whole_valid, 3:6].view(-1, 1, 3).expand(B, N, 3)
    sigma = torch.zeros(B, N)
    world_normal = torch.zeros(M, 3)
    if M > 0:
        sigma = _scatter(sigma, ray_valid, density(sd, cfg, xyz))                          # :289
    weight = raw2alpha(sigma, dists * cfg.distance_scale)                                   # :366
    if trace is not None:
        trace.update({f"xyz{recur}": xyz, f"ray_valid{recur}": ray_valid, f"whole_valid{recur}": whole_valid,
                      f"z_vals{recur}": z_vals, f"weight{recur}": weight})

    def render_reflection(brays, mipval, retrace):                                          # :291-317
        if retrace:
            ims, st = render(sd, cfg, brays, focal, alpha_volume, noise, is_train=is_train,
                             recur=recur + 1, bg_col=None, start_mipval=mipval.reshape(-1),
                             override_near=3 * d["stepsize"], dynamic_batch_size=False,
                             tonemap=False, trace=trace, forced=forced)
            n_samples.extend(st["n_samples"])
            return ims["rgb_map"]
        noise.draw("rand", (brays.shape[0],), unused=True)
        noise.draw("rand", (brays.shape[0],), unused=True)
        return env_lookup(sd, brays[..., 3:6], mipval.reshape(-1)).reshape(-1, 3)

    if M > 0:
        app = app_feature(sd, cfg, xyz)                                                     # :386
        world_normal = normals(sd, cfg, xyz)                                                # :393
        rgb, debug = shade(sd, cfg, xyz, app, viewdirs[ray_valid], world_normal, weight, ray_valid,
                           render_reflection, noise, is_train, recur, trace, forced)
    else:
        rgb = torch.empty((0, 3))
        debug = {}
    acc_map = torch.sum(weight, 1)
    eweight = weight[ray_valid][..., None]
    rgb_map = row_mask_sum(eweight * rgb, ray_valid)                                        # :452
    stats = dict(recur=recur, whole_valid=whole_valid, n_samples=n_samples)
    images = {}
    if bg_col is None:                                                                      # :460-468
        rough = -100 * torch.ones(B, 1) if start_mipval is None else start_mipval
        noise.draw("rand", (B,), unused=True)
        noise.draw("rand", (B,), unused=True)
        bg = env_lookup(sd, viewdirs[:, 0, :], rough).reshape(-1, 3)
        if tonemap:
            bg = srgb_tonemap(bg, noclip=True)
    else:
        bg = bg_col.reshape(1, 3)
    if not is_train and recur == 0:
        with torch.no_grad():                                                               # :480-537
            images["depth"] = torch.sum(weight * z_vals, 1)
            wn = row_mask_sum(world_normal * eweight, ray_valid)
            images["world_normal"] = acc_map[..., None] * wn + (1 - acc_map[..., None])
        for k, v in debug.items():
            images[k] = row_mask_sum(v * eweight, ray_valid) + (1 - acc_map[..., None]) * bg
    elif recur == 0:
        aweight = weight[ray_valid]
        ndv = (-viewdirs[ray_valid].reshape(-1, 3).detach() * world_normal.reshape(-1, 3)).sum(dim=-1)
        stats["ori_loss"] = (aweight * (ndv.clamp(max=0) ** 2)).sum()                       # :583
        # normal_module is None -> pred_norms == 0 -> align_world_loss == 2    # :598-602
        stats["prediction_loss"] = (aweight * (2 * (1 - (torch.zeros(M, 3) * world_normal).sum(dim=-1)))).sum()
        stats["envmap_reg"] = (env_activation(sd).reshape(-1, 3).mean(dim=0).mean() - 0.05).clip(min=0)
        stats["brdf_reg"] = debug["tint"].mean().clip(min=0) if "tint" in debug else torch.tensor(0.0)
        stats["diffuse_reg"] = ((aweight.detach().reshape(-1, 1) * debug["diffuse"]).sum() / 3
                                if "diffuse" in debug else torch.tensor(0.0))
        for k, v in debug.items():
            images[k] = v
    if tonemap:
        rgb_map = srgb_tonemap(rgb_map, noclip=False)                                       # :658 (hdr False)
    rgb_map = rgb_map + (1 - acc_map[..., None]) * bg                                       # :659
    images["rgb_map"] = rgb_map
    images["acc_map"] = acc_map.detach()
    return images, stats


def _scatter(dense, mask, vals):
    out = dense.clone()
    out[mask] = vals
    return out


def training_loss(images, stats, rgb_gt, lbatch_size, sd, ori_lambda=0.1, pred_lambda=3e-4, l1_weight=8e-5):
    """Loss assembly of train.py:598-708 for the microfacet_tensorf2 params block."""
    wv = stats["whole_valid"]
    loss = ((images["rgb_map"].clip(max=1).clip(0, 1) - rgb_gt[wv].clip(0, 1)) ** 2).sum()
    total = loss + ori_lambda * stats["ori_loss"] + pred_lambda * stats["prediction_loss"]
    l1 = 0
    for i in range(3):                                                # density_L1, fields/tensoRF.py:332-340
        l1 = l1 + sd[f"rf.density_rf.app_plane.{i}"].abs().mean() + sd[f"rf.density_rf.app_line.{i}"].abs().mean()
    total = total + l1_weight * l1
    return total / lbatch_size, loss


def psnr_8bit(pred, gt):
    # renderer.py:399-401 (per image)
    q = torch.floor(pred.clip(0, 1) * 255) / 255
    return -10.0 * torch.log10(((q - gt.clip(0, 1)) ** 2).mean())
