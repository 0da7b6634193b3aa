"""GGX visible-normal importance sampling -- the reference's operator interface (brdf_samplers/ggx.py GGXSampler :60-268,
brdf_samplers/base.py PseudoRandomSampler :3-23) on top of the HIP kernels.

The hot path (models/microfacet.py `Microfacet.shade_compact`) calls nmf_ggx_rays_fwd / _bwd directly on the compact ray
list; the methods below keep the reference's dense signatures -- `draw(B, m)`, `sample(u1, u2, V, N, r1, r2, ray_mask)`,
`compute_prob(dir_in, dir_out, halfvec, r1, r2)` -- for callers written against the reference: they convert the padded
[bounce points x m] mask to the compact list and run the same kernels (differentiable wrt the normal and the roughness)."""
import torch

from .. import hip
from ..functional import GgxRays

EPS = torch.finfo(torch.float32).eps


def normalize(v):
    return v / (v ** 2).sum(dim=-1, keepdim=True).clip(min=EPS).sqrt()


def mat3_vec(m, v):
    """[n,3,3] x [n,3] -> [n,3] as elementwise ops (a batched 3x3 GEMM through rocBLAS costs 1.6 ms per call at
    n = 250 k -- profiles/r01_c)"""
    return (m * v.unsqueeze(-2)).sum(-1)


def mat3T_vec(m, v):
    return (m * v.unsqueeze(-1)).sum(-2)


def compact_from_mask(ray_mask):
    """[Mb, m] bool -> (cnt [Mb] i32, row_of_ray [R] i32, j_of_ray [R] i32, row_off [Mb+1] i64): the rays of
    `torch.where(ray_mask)` in its (row-major) order"""
    ri, rj = torch.where(ray_mask)
    cnt = ray_mask.sum(dim=1)
    row_off = torch.zeros(ray_mask.shape[0] + 1, dtype=torch.int64, device=ray_mask.device)
    row_off[1:] = torch.cumsum(cnt, 0)
    return cnt.int().contiguous(), ri.int().contiguous(), rj.int().contiguous(), row_off


def world_basis(N):
    """row_world_basis of ggx.py:83-91 transposed: columns (tangent, bitangent, normal) per row [Mb,3,3]"""
    Mb = N.shape[0]
    z_up = torch.tensor([0.0, 0.0, 1.0], device=N.device).expand(Mb, 3)
    x_up = torch.tensor([-1.0, 0.0, 0.0], device=N.device).expand(Mb, 3)
    up = torch.where(N[:, 2:3].abs() < 0.999, z_up, x_up)
    tangent = normalize(torch.linalg.cross(up, N))
    bitangent = normalize(torch.linalg.cross(N, tangent))
    return torch.stack([tangent, bitangent, N], dim=1).permute(0, 2, 1)


class PseudoRandomSampler(torch.nn.Module):
    def __init__(self, max_samples):
        super().__init__()
        self.max_samples = max_samples
        self.register_buffer("angs", torch.quasirandom.SobolEngine(dimension=2, scramble=True).draw(max_samples))

    def draw(self, B, num_samples, noise=None):
        """base.py:11-20: (Sobol[:m] + 0.25 * U[B,1,2]) mod 1 -> [B, m, 2].  (The hot path never materialises this
        matrix: nmf_ggx_rays_fwd forms the same numbers per ray from the table and the row offsets.)"""
        if noise is not None:
            offset = noise.uniform((B, 1, 2)).reshape(B, 1, 2)
        else:
            offset = torch.rand(B, 1, 2, device=self.angs.device)
        angs = self.angs.reshape(1, self.max_samples, 2)[:, :num_samples, :].expand(B, num_samples, 2)
        return (angs + offset * 0.25) % 1.0

    def update(self, *args, **kwargs):
        pass


class GGXSampler(PseudoRandomSampler):
    def sample(self, u1, u2, dir_out, normal, r1, r2, ray_mask, eps=EPS, **kwargs):
        """ggx.py:61-226.  u1, u2 [B,m] uniforms, dir_out V [B,3], normal [B,3], r1 [B,1] (r2 is overwritten by r1, :75),
        ray_mask [B,m] bool -> (L [R,3], row_world_basis [R,3,3], logpdf [R]) for the R rays of torch.where(ray_mask).
        One launch of nmf_ggx_rays_fwd: the kernel's per-ray Sobol lookup is pointed at the caller's (u1,u2) matrix."""
        B, m = ray_mask.shape
        cnt, row_of_ray, rj, row_off = compact_from_mask(ray_mask)
        R = int(row_of_ray.shape[0])
        dev = normal.device
        if R == 0:
            return normal.new_zeros((0, 3)), normal.new_zeros((0, 3, 3)), normal.new_zeros((0,))
        table = torch.stack([u1, u2], dim=-1).reshape(B * m, 2).float().contiguous()       # "sobol" table = the given draws
        j_of_ray = (row_of_ray * m + rj).contiguous()
        zero_off = torch.zeros(B, 2, device=dev)
        L, _hl, _dl, lpdf, _mip, _rays = GgxRays.apply(dir_out.float(), normal.float(), r1.float().reshape(B, 1),
                                                        torch.zeros(B, 3, device=dev), zero_off, cnt, table, row_of_ray,
                                                        j_of_ray, row_off)
        basisT = torch.index_select(world_basis(normal.float()), 0, row_of_ray.long())
        return L, basisT, lpdf

    @torch.no_grad()
    def compute_prob(self, dir_in, dir_out, halfvec, r1, r2, **kwargs):
        """ggx.py:228-268 (isotropic: r2 = r1): pdf of the sampled direction, [R,1]; zero below the horizon.  Local-frame
        inputs.  Evaluated by the kernel behind nmf_retrace_scores' sibling entry nmf_ggx_prob."""
        return hip.ggx_prob(dir_in.float().contiguous(), dir_out.float().contiguous(), halfvec.float().contiguous(),
                            r1.float().reshape(-1).contiguous()).reshape(-1, 1)
