"""Spherical-harmonic helpers used by the diffuse irradiance term -- mirrors the functions of the
reference's modules/sh.py that the microfacet_tensorf2 path reaches: eval_sh_bases (:97-142, first 9
bases, with the all-positive SH_C2 table of :67-73) and Al2 (:149-157)."""
import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, 1.0925484305920792, 0.31539156525252005, 1.0925484305920792, 0.5462742152960396)


def eval_sh_bases(basis_dim, dirs):
    if basis_dim != 9:
        raise NotImplementedError("only the 9 bases of the irradiance term are used on this path")
    x, y, z = dirs.unbind(-1)
    xx, yy, zz = x * x, y * y, z * z
    return torch.stack([torch.full_like(x, SH_C0), SH_C1 * y, SH_C1 * z, SH_C1 * x, SH_C2[0] * (x * y),
                        SH_C2[1] * (y * z), SH_C2[2] * (3 * zz - 1), SH_C2[3] * (x * z), SH_C2[4] * (xx - yy)], dim=-1)


def Al2(l):
    if l == 0:
        return math.pi
    if l == 1:
        return 2 * math.pi / 3
    if l % 2 == 1:
        return 0
    return 2 * math.pi * (-1) ** (l / 2 - 1) / ((l + 2) * (l - 1)) * (
        math.factorial(l) / (2 ** l * math.factorial(l // 2) ** 2))
