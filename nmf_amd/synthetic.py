"""Deterministic synthetic workload "S1 box"  used by the parity tests, the golden
generator, smoke() and bench.py.  nerf_synthetic is not available offline, so the benchmark scene
is defined procedurally: a solid cube of half-size 0.75 inside the +-1.5 AABB, written straight
into the TensoRF density factors, random appearance factors, a smooth environment map and an
800x800 pinhole camera at (2.4,-2.8,1.6) looking at the origin (blender convention:
fx = 400 / tan(camera_angle_x / 2), unit-norm directions, dataLoader/blender.py:97-120,170-173
of the reference).

Everything is generated on the CPU with a seeded torch.Generator so that this container and the
GPU box see bit-identical inputs.
"""
import math

import torch

CAMERA_ANGLE_X = 0.6911112
IMG_WH = 800


def state_dict_s1(grid=128, n_density=16, n_app=24, app_dim=24, bg_resolution=512, seed=0,
                  env_pattern=True):
    """Parameter tensors of scene S1, keyed with the reference's state_dict names (Appendix C)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    lin = torch.linspace(-1, 1, grid)
    inside = (lin.abs() < 0.5).float()
    for i in range(3):
        plane = 0.001 * torch.randn(1, n_density, grid, grid, generator=g)
        line = 0.001 * torch.randn(1, n_density, grid, 1, generator=g)
        plane[0, 0] = 10.0 * inside[:, None] * inside[None, :]
        line[0, 0, :, 0] = inside
        sd[f"rf.density_rf.app_plane.{i}"] = plane
        sd[f"rf.density_rf.app_line.{i}"] = line
    for i in range(3):
        sd[f"rf.app_rf.app_plane.{i}"] = 0.1 * torch.randn(1, n_app, grid, grid, generator=g)
        sd[f"rf.app_rf.app_line.{i}"] = 0.1 * torch.randn(1, n_app, grid, 1, generator=g)
    bound = 1.0 / math.sqrt(3 * n_app)
    sd["rf.basis_mat.weight"] = (torch.rand(app_dim, 3 * n_app, generator=g) * 2 - 1) * bound

    def xavier(o, i):
        a = math.sqrt(6.0 / (i + o))
        return (torch.rand(o, i, generator=g) * 2 - 1) * a

    for head, o in (("diffuse", 3), ("tint", 3), ("f0", 3)):
        sd[f"model.diffuse_module.{head}_mlp.0.weight"] = xavier(o, app_dim)
        sd[f"model.diffuse_module.{head}_mlp.0.bias"] = torch.zeros(o)
    b = 1.0 / math.sqrt(app_dim)
    sd["model.diffuse_module.roughness_mlp.0.weight"] = (torch.rand(2, app_dim, generator=g) * 2 - 1) * b
    sd["model.diffuse_module.roughness_mlp.0.bias"] = (torch.rand(2, generator=g) * 2 - 1) * b

    def kaiming(o, i):
        a = math.sqrt(6.0 / i)
        return (torch.rand(o, i, generator=g) * 2 - 1) * a

    for k, (o, i) in {0: (64, 66), 2: (64, 64), 4: (4, 64)}.items():
        sd[f"model.brdf.mlp.{k}.weight"] = kaiming(o, i)
        sd[f"model.brdf.mlp.{k}.bias"] = 0.01 * torch.randn(o, generator=g)
    h, w = bg_resolution, 2 * bg_resolution
    bg = torch.full((1, 3, h, w), -0.6)
    if env_pattern:
        yy = torch.linspace(0, math.pi, h)[:, None]
        xx = torch.linspace(0, 2 * math.pi, w)[None, :]
        for c in range(3):
            bg[0, c] += 0.8 * torch.sin(yy * (c + 1)) * torch.cos(xx * (c + 2) + c) + 0.3 * torch.cos(3 * yy)
        bg += 0.05 * torch.randn(1, 3, h, w, generator=g)
    sd["bg_module.bg_mat"] = bg
    sd["bg_module.mipbias"] = torch.tensor(1.0, dtype=torch.float64)
    sd["bg_module.brightness"] = torch.tensor(0.0, dtype=torch.float64)
    sd["bg_module.mul"] = torch.tensor(1.0, dtype=torch.float64)
    sd["model.brdf_sampler.angs"] = torch.quasirandom.SobolEngine(2, scramble=True, seed=seed).draw(1024)
    return sd


def camera_rays(n_rays, seed=0, eye=(2.4, -2.8, 1.6), wh=IMG_WH, all_pixels=False):
    """[n,6] rays (origin, unit direction) through uniformly random pixels of the S1 camera."""
    focal = 0.5 * wh / math.tan(0.5 * CAMERA_ANGLE_X)
    eye_t = torch.tensor(eye, dtype=torch.float32)
    fwd = -eye_t / eye_t.norm()
    up = torch.tensor([0.0, 0.0, 1.0])
    right = torch.linalg.cross(fwd, up)
    right = right / right.norm()
    true_up = torch.linalg.cross(right, fwd)
    if all_pixels:
        idx = torch.arange(wh * wh)
    else:
        g = torch.Generator().manual_seed(seed)
        idx = torch.randint(0, wh * wh, (n_rays,), generator=g)
    px = (idx % wh).float() + 0.5
    py = (idx // wh).float() + 0.5
    x = (px - wh / 2) / focal
    y = -(py - wh / 2) / focal
    d = fwd[None] + x[:, None] * right[None] + y[:, None] * true_up[None]
    d = d / d.norm(dim=-1, keepdim=True)
    o = eye_t[None].expand_as(d)
    return torch.cat([o, d], dim=-1).contiguous(), focal


def orbit_cameras(n_views, radius=4.0, seed=0):
    """S2 "orbit": camera positions on the upper hemisphere of a sphere around the origin."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n_views, generator=g)
    phi = 2 * math.pi * torch.rand(n_views, generator=g)
    z = 0.15 + 0.8 * u                                  # elevation: avoid the horizon and the pole
    rxy = torch.sqrt(1 - z * z)
    return torch.stack([radius * rxy * torch.cos(phi), radius * rxy * torch.sin(phi), radius * z], -1)


def orbit_rays(n_views, wh, seed=0, radius=4.0):
    """All pixel rays of `n_views` wh x wh pinhole cameras looking at the origin -> [n_views*wh*wh, 6], focal."""
    eyes = orbit_cameras(n_views, radius, seed)
    out, focal = [], None
    for e in eyes:
        r, focal = camera_rays(0, eye=tuple(e.tolist()), wh=wh, all_pixels=True)
        out.append(r)
    return torch.cat(out, 0), focal
