"""Micro-benchmark + error measurement of nmf_brdf_mlp_fwd / _bwd (csrc/brdf_mlp.hip) on random inputs of a step's shape.

    python tools/mlp_bench.py [R ...]       (default: 242000 45000 8000)

Prints per R: time per launch (HIP events, 20 launches), and the largest deviation of the outputs / gradients from a
float64 evaluation of the same network on the GPU (torch), next to what a float32 torch evaluation deviates by.
"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmf_amd import hip  # noqa: E402

DEV = "cuda"


def ish(v, kappa):
    x, y, z = v.unbind(-1)
    k = kappa + 1e-8
    a1, a2 = torch.exp(-1 / k), torch.exp(-3 / k)
    xx, yy, zz = x * x, y * y, z * z
    o = [torch.full_like(x, 0.28209479177387814), -a1 * 0.488603 * x, a1 * 0.488603 * z, -a1 * 0.488603 * y,
         a2 * 1.092548 * y * x, -a2 * 1.092548 * y * z, a2 * 0.315392 * (3 * zz - 1), -a2 * 1.092548 * x * y,
         a2 * 0.546274 * (xx - yy), 2.50334 * x * y * (xx - yy), -1.77013 * y * z * (-3 * xx + yy),
         0.946175 * x * y * (7 * zz - 1), 0.669047 * y * z * (7 * zz - 3), 3.70251 * zz * zz - 3.17358 * zz + 0.317358,
         0.669047 * x * z * (7 * zz - 3), (0.473087 * xx - 0.473087 * yy) * (7 * zz - 1), 1.77013 * x * z * (xx - 3 * yy),
         0.625836 * xx * xx - 3.755016 * xx * yy + 0.625836 * yy * yy]
    return torch.stack(o, -1)


def reference(ws, hv, dv, feat, rough, idx, bias, c, dt):
    ws = [w.detach().to(dt).requires_grad_(True) for w in ws]
    feat = feat.detach().to(dt).requires_grad_(True)
    hv, dv, rough = hv.to(dt), dv.to(dt), rough.to(dt)
    f = feat[idx.long()]
    kappa = 1 / (rough[idx.long()] + 1e-3)
    X = torch.cat([f, ish(hv, kappa), hv, ish(dv, kappa), dv], -1)
    h = torch.relu(X @ ws[0].T + ws[1])
    h = torch.relu(h @ ws[2].T + ws[3])
    o = torch.sigmoid((h @ ws[4].T + ws[5])[:, :3] + bias)
    g = torch.autograd.grad((o * c.to(dt)).sum(), [feat] + ws)
    return o.detach(), g


def main():
    Rs = [int(a) for a in sys.argv[1:]] or [242000, 45000, 8000]
    gen = torch.Generator(device="cpu").manual_seed(0)
    shapes = [(64, 66), (64,), (64, 64), (64,), (4, 64), (4,)]
    ws = [(torch.randn(s, generator=gen) * (0.25 if len(s) == 2 else 0.1)).to(DEV) for s in shapes]
    for R in Rs:
        Mb = max(R // int(os.environ.get("NMF_MLP_RUN", "13")), 1)      # rays per bounce row
        idx = torch.sort(torch.randint(0, Mb, (R,), generator=gen)).values.int().to(DEV)
        hv = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1).to(DEV)
        dv = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1).to(DEV)
        feat = torch.randn(Mb, 24, generator=gen).to(DEV)
        rough = (torch.rand(Mb, generator=gen) * 0.49 + 0.01).to(DEV)
        c = torch.randn(R, 3, generator=gen).to(DEV)
        out, mask = hip.brdf_mlp_fwd(ws, hv, dv, feat, rough, idx, 0.37, with_mask=True)
        grads = [torch.zeros_like(w) for w in ws]
        dfeat = hip.brdf_mlp_bwd(ws, hv, dv, feat, rough, idx, out, mask, c, grads)
        o64, g64 = reference(ws, hv, dv, feat, rough, idx, 0.37, c, torch.float64)
        o32, g32 = reference(ws, hv, dv, feat, rough, idx, 0.37, c, torch.float32)

        def err(a, b):
            return float((a.double() - b).abs().max() / (b.abs().max() + 1e-30))
        names = ["d_feat", "dW0", "db0", "dW2", "db2", "dW4", "db4"]
        print(f"R={R}: out max abs err {float((out.double() - o64).abs().max()):.2e} "
              f"(torch f32: {float((o32.double() - o64).abs().max()):.2e})")
        for n, a, b32, b64 in zip(names, [dfeat] + grads, g32, g64):
            print(f"   {n:7s} rel-to-max err {err(a, b64):.2e}   (torch f32: {err(b32, b64):.2e})")
        img = hip.brdf_mlp_pack(ws)
        for name, fn in (("fwd", lambda: hip.brdf_mlp_fwd(ws, hv, dv, feat, rough, idx, 0.37, with_mask=True)),
                         ("bwd", lambda: hip.brdf_mlp_bwd(ws, hv, dv, feat, rough, idx, out, mask, c, grads)),
                         ("fwd, packed weights", lambda: hip.brdf_mlp_fwd(None, hv, dv, feat, rough, idx, 0.37, with_mask=True, image=img)),
                         ("bwd, packed weights", lambda: hip.brdf_mlp_bwd(None, hv, dv, feat, rough, idx, out, mask, c, grads, image=img)),
                         ("pack", lambda: hip.brdf_mlp_pack(ws, img))):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print(f"   {name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch")


if __name__ == "__main__":
    main()
