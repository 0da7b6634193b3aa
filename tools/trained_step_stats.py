"""GPU: the training chunk of tests/golden/trained_step.npz (a trained S2 state, 471 rays) K times with this build's PRODUCTION noise
source (DeviceNoise), through the operator graph -- loss terms, sample counts and the norm of every parameter gradient per run --
against the same statistics of the reference's own runs (tests/golden/trained_step_stats.npz, torch's generator).  Equal single-step
gradients under equal noise are tests/test_hip_e2e.py::test_trained_state_single_step_vs_reference; this compares the DISTRIBUTIONS
the two noise sources induce (round 6: is the build's gradient noise the reference's?).
    python tools/trained_step_stats.py [K]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import Golden  # noqa: E402
from nmf_amd.config import build_model  # noqa: E402
from nmf_amd.noise import DeviceNoise  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
STATE = sys.argv[2] if len(sys.argv) > 2 else "trained_step"            # fixture that holds the state: trained_step | init_step
STATS = sys.argv[3] if len(sys.argv) > 3 else "trained_step_stats"      # the reference's statistics: trained_step_stats | init_step3_stats
FUSED = len(sys.argv) > 4 and sys.argv[4] == "fused"                    # chunks through the fused pass (ChunkPass) instead of the operator graph
DEV = "cuda"
g = Golden(STATE)
ref = np.load(os.path.join(ROOT, "tests", "golden", STATS + ".npz"))
trace = np.load(os.path.join(ROOT, "tests", "golden", "psnr_trace.npz"))
chunks = [tuple(int(v) for v in c) for c in ref["chunks"]] if "chunks" in ref.files else [(0, int(g["n_rays"]))]
G, BG = g["grid"], g["bg_res"]
over = {"sampler.update_list": [10 ** 9], "rf.upsamp_list": [10 ** 9], "rf.N_voxel_init": G ** 3, "rf.N_voxel_final": G ** 3,
        "sampler.max_samples": 40000, "model.max_brdf_rays": [80000, 40000], "model.target_num_samples": [80000],
        "model.max_retrace_rays": [g["max_retrace"]], "model.rays_per_ray": 128}
nerf, _ = build_model(grid=G, bg_resolution=BG, device=DEV, overrides=over)
sd = {k[3:]: g[k] if g.np(k).shape != () else torch.as_tensor(g.np(k)) for k in g.keys("sd/")}
nerf.load_state_dict(sd, strict=False)
nerf.model.brdf.bias, nerf.model.diffuse_module.diffuse_bias, nerf.model.diffuse_module.roughness_bias = (float(v) for v in g.np("biases"))
nerf.train()
nerf.sampler.update(nerf.rf, init=True)
nerf.model.detach_N = False
nerf.model.min_rough = float(g["min_rough"])
nerf.fused_training_pass = FUSED
rays_all, gt_all = torch.as_tensor(trace["rays_train"][:chunks[-1][1]]).to(DEV), torch.as_tensor(trace["rgb_train"][:chunks[-1][1]]).to(DEV)
names = str(ref["names"]).split("\n")
params = dict(nerf.named_parameters())
noise = DeviceNoise(torch.device(DEV), seed=77)
rows, losses, ns = [], [], []
for k in range(K):
    for p in nerf.parameters():
        p.grad = None
    for ci, (lo, hi) in enumerate(chunks):
        rays, gt = rays_all[lo:hi], gt_all[lo:hi]
        nerf.model.max_retrace_rays = [int(g["max_retrace"])]
        ims, st = nerf(rays, float(g["focal"]), bg_col=torch.ones(3, device=DEV), is_train=True, ndc_ray=False, noise=noise)
        wv = st["whole_valid"]
        loss = ((ims["rgb_map"].clip(0, 1) - gt[wv].clip(0, 1)) ** 2).sum()
        total = (loss + float(g["ori_lambda"]) * st["ori_loss"] + float(g["pred_lambda"]) * st["prediction_loss"]
                 + 8e-5 * nerf.rf.density_L1()) / 1024
        total.backward()
        if ci == 0:
            losses.append([float(loss.detach()), float(st["ori_loss"].detach()), float(st["prediction_loss"].detach()), float(total.detach())])
            ns.append([int(v) for v in st["n_samples"]] + [int(wv.sum())])
    rows.append([float(params[n].grad.norm()) if params[n].grad is not None else np.nan for n in names])
H, R = np.asarray(rows), ref["gradnorm"]
HL, RL = np.asarray(losses), ref["losses"]
HN, RN = np.asarray(ns, dtype=np.float64), ref["n_samples"].astype(np.float64)
se = lambda x: x.std(0, ddof=1) / np.sqrt(x.shape[0])  # noqa: E731
print(f"state {STATE}, chunks {chunks} accumulated ({'fused pass' if FUSED else 'operator graph'}): reference {R.shape[0]} runs (its generator), this build {K} runs (DeviceNoise)")
print(f"{'quantity':48s} {'reference mean':>15s} {'here mean':>13s} {'rel diff %':>11s} {'z':>7s}   {'ref std/mean':>12s} {'here std/mean':>13s}")
def row(name, r, h):
    if not (np.isfinite(r).all() and np.isfinite(h).all()) or r.mean() == 0:
        return
    d = h.mean() - r.mean()
    s = np.sqrt(se(r) ** 2 + se(h) ** 2)
    print(f"{name[:48]:48s} {r.mean():15.6g} {h.mean():13.6g} {100 * d / abs(r.mean()):11.2f} {d / s if s > 0 else 0:7.1f}   "
          f"{r.std(ddof=1) / abs(r.mean()):12.3f} {h.std(ddof=1) / abs(h.mean()):13.3f}")
for j, n in enumerate(("photometric loss", "ori_loss", "prediction_loss", "total")):
    row(n, RL[:, j], HL[:, j])
for j, n in enumerate(("n_samples0", "n_samples1", "rays kept")):
    row(n, RN[:, j], HN[:, j])
for j, n in enumerate(names):
    row("|grad| " + n, R[:, j], H[:, j])
