import sys, os, torch
sys.path.insert(0, '/root/repo/scratch/old'); sys.path.insert(0, '/root/repo/scratch/old/tests')
os.environ["NMF_STEP_CORE"] = "0"
from nmf_amd import hip
import test_hip_e2e as T
orig = hip.brdf_mlp_fwd
new_outs = torch.load("/tmp/outs.pt")
res = {}
for tag in ("A", "B"):
    it = iter(new_outs)
    def spy(*a, **k):
        o = orig(*a, **k)
        if tag == "B":
            n = next(it)
            if n.shape == o.shape:
                dd = (n - o)
                print("  call", tuple(o.shape), "diff max", float(dd.abs().max()), "mean signed", float(dd.mean()), "frac nonzero", float((dd != 0).float().mean()))
                o = n
            else:
                print("  shape mismatch", n.shape, o.shape)
        return o
    hip.brdf_mlp_fwd = spy
    g = T.Golden("e2e_full_steady")
    nerf = T._full_size_model(g)
    pins = T._pin_reference_bookkeeping(g, order=False, valid=False, exact=True)
    with torch.no_grad():
        ims, st = T._seeded_render(nerf, g, pins)
    tr = pins.trace
    res[tag] = dict(rgb=ims["rgb_map"].cpu(), ns=list(st["n_samples"]), tr={k: (v.cpu() if torch.is_tensor(v) else v) for k, v in tr.items()}, gold=g["rgb_map"])
    print(tag, "n_samples", res[tag]["ns"])
a, b = res["A"], res["B"]
for t in (a, b):
    tr = t["tr"]
    print("M1", tr["counts1"].shape[0], "sum counts1", int(tr["counts1"].sum()), "sum counts_own1", int(tr["counts_own1"].sum()), "L1 rows", tr["L1"].shape[0],
          "nonzero own", int((tr["counts_own1"] > 0).sum()), "max own", int(tr["counts_own1"].max()), "incoming1 sum", float(tr["incoming1"].double().sum()))
