"""GGX visible-normal importance sampling -- host-side mirror of the reference's brdf_samplers/ggx.py
(GGXSampler :60-268) and brdf_samplers/base.py (PseudoRandomSampler :3-23).

Round-1 status: this operator is differentiable wrt the normal and the roughness (second-order effects of
the reference flow through it), so it is expressed with torch tensor ops ON THE DEVICE and torch autograd
provides its backward; it works on the compact ray list (row_of_ray) instead of the reference's padded
[bounce points x m] mask.  A fused HIP forward/backward is the next step for this row."""
import math

import torch

EPS = torch.finfo(torch.float32).eps


def normalize(v):
    return v / (v ** 2).sum(dim=-1, keepdim=True).clip(min=EPS).sqrt()


def mat3_vec(m, v):
    """[n,3,3] x [n,3] -> [n,3] as elementwise ops (a batched 3x3 GEMM through rocBLAS costs 1.6 ms per call at
    n = 250 k -- profiles/r01_c)"""
    return (m * v.unsqueeze(-2)).sum(-1)


def mat3T_vec(m, v):
    return (m * v.unsqueeze(-1)).sum(-2)


def _safe_trig(x, fn):
    return fn(x % (100 * math.pi))


class PseudoRandomSampler(torch.nn.Module):
    def __init__(self, max_samples):
        super().__init__()
        self.max_samples = max_samples
        self.register_buffer("angs", torch.quasirandom.SobolEngine(dimension=2, scramble=True).draw(max_samples))

    def draw_compact(self, n_rows, row_of_ray, j_of_ray, noise):
        """(Sobol[j] + 0.25*U[row]) mod 1 for every ray (base.py:11-20)"""
        offset = noise.uniform((n_rows, 1, 2)).reshape(n_rows, 2) * 0.25
        return (self.angs[j_of_ray.long()] + offset[row_of_ray.long()]) % 1.0

    def update(self, *args, **kwargs):
        pass


class GGXSampler(PseudoRandomSampler):
    def sample_compact(self, u1, u2, V, N, r, row_of_ray):
        """V, N [Mb,3], r [Mb,1], u1/u2/row_of_ray [R] -> L [R,3], basisT [R,3,3], logpdf [R] (ggx.py:61-226)."""
        Mb = V.shape[0]
        dev = V.device
        rows = row_of_ray.long()
        z_up = torch.tensor([0.0, 0.0, 1.0], device=dev).expand(Mb, 3)
        x_up = torch.tensor([-1.0, 0.0, 0.0], device=dev).expand(Mb, 3)
        up = torch.where(N[:, 2:3].abs() < 0.999, z_up, x_up)
        tangent = normalize(torch.linalg.cross(up, N))
        bitangent = normalize(torch.linalg.cross(N, tangent))
        basis = torch.stack([tangent, bitangent, N], dim=1)
        V_l = mat3_vec(basis, V)
        rc = r.reshape(-1)
        Vs = normalize(torch.stack([rc * V_l[..., 0], rc * V_l[..., 1], V_l[..., 2]], dim=-1))
        T1 = torch.where(Vs[..., 2:3] < 0.999, normalize(torch.linalg.cross(Vs, z_up, dim=-1)), x_up)
        T2 = normalize(torch.linalg.cross(T1, Vs, dim=-1))
        z = Vs[..., 2]
        a = (1 / (1 + z.detach()).clip(min=1e-8)).clip(max=1e4)
        a_m, r_m, z_m = a[rows], rc[rows], z[rows]
        T1_m, T2_m, Vs_m = T1[rows], T2[rows], Vs[rows]
        basisT = basis.permute(0, 2, 1)[rows]
        rr = torch.sqrt(u1)
        phi = torch.where(u2 < a_m, u2 / a_m * math.pi, (u2 - a_m) / (1 - a_m) * math.pi + math.pi)
        P1 = (rr * _safe_trig(phi, torch.cos)).unsqueeze(-1)
        P2 = (rr * _safe_trig(phi, torch.sin) * torch.where(u2 < a_m, torch.ones_like(z_m), z_m)).unsqueeze(-1)
        Ns = P1 * T1_m + P2 * T2_m + (1 - P1 * P1 - P2 * P2).clip(min=EPS).sqrt() * Vs_m
        H_l = normalize(torch.stack([Ns[..., 0] * r_m, Ns[..., 1] * r_m, Ns[..., 2]], dim=-1))
        H = mat3_vec(basisT, H_l)
        w_o, eN = V[rows], N[rows]
        w_i = normalize(2.0 * (w_o * H).sum(dim=-1, keepdim=True) * H - w_o)
        w_i = w_i * torch.where((w_i * eN).sum(dim=-1, keepdim=True) > 0, 1.0, -1.0)
        with torch.no_grad():
            lw_i = mat3T_vec(basisT, w_i)
            lw_o = mat3T_vec(basisT, w_o)
            logp = self.compute_prob(lw_i, lw_o, H_l, r_m, r_m).clip(min=EPS).log().reshape(-1)
        return w_i, basisT, logp

    def compute_prob(self, dir_in, dir_out, halfvec, r1, r2, **kwargs):
        # ggx.py:228-268 (isotropic: r2 = r1)
        r2 = r1.reshape(-1).clip(min=EPS)
        r1 = (r1.reshape(-1) + r2).clip(min=EPS) / 2
        lam = (-1 + (1 + ((dir_in[:, 0] * r1) ** 2 + (dir_in[:, 1] * r2) ** 2)
                     / (dir_in[:, 2] ** 2).clip(min=1e-6)).clip(min=EPS).sqrt()) / 2
        invD = math.pi * r1 * r2 * (halfvec[:, 0] ** 2 / r1 ** 2 + halfvec[:, 1] ** 2 / r2 ** 2 + halfvec[:, 2] ** 2) ** 2
        logD = -((1 + lam) * invD).clip(min=EPS).log() - (4 * dir_out[..., 2]).clip(min=EPS).log()
        prob = logD.exp().reshape(-1, 1)
        return torch.where(dir_in[:, 2:3] > 0, prob, torch.zeros_like(prob))
