"""which factor moves the PSNR after equal iterations: the initial state (reference's fixture states vs this build's constructors +
calibration) or the noise source (torch's CPU generator in the reference's call order vs the device generator)

    python tools/psnr_factors.py [runs per variant, default 24] [variants, default "ref+cpu,ref+dev,own+cpu,own+dev"]
A run with the CPU generator takes ~40 s (host-side draws of every noise tensor), one with the device generator ~1 s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from nmf_amd.config import build_model, resolved_config
from nmf_amd.noise import DeviceNoise, ReplayNoise
from nmf_amd.renderer import psnr_8bit, render_images
from nmf_amd.trainer import Trainer
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
g = np.load(os.path.join(bench.ROOT, "tests", "golden", "psnr_trace.npz"))
G0, BG, res = int(g["grid0"]), int(g["bg_res"]), int(g["res"])
ov = dict(line.split("=", 1) for line in str(g["overrides"]).split("\n"))
ints = lambda k: [int(v) for v in ov[k].strip("[]").split(",")]
over = {"sampler.update_list": [10 ** 9], "rf.upsamp_list": [10 ** 9], "rf.N_voxel_init": G0 ** 3, "rf.N_voxel_final": G0 ** 3,
        "sampler.max_samples": ints("model.arch.sampler.max_samples")[0], "model.max_brdf_rays": ints("model.arch.model.max_brdf_rays"),
        "model.target_num_samples": ints("model.arch.model.target_num_samples"),
        "model.max_retrace_rays": ints("model.arch.model.max_retrace_rays"), "model.rays_per_ray": ints("model.arch.model.rays_per_ray")[0]}
mn, mx, start, target = (int(v) for v in g["params_params"])
params = dict(resolved_config()["params"], n_iters=int(ov["model.params.n_iters"]), batch_size=mn, min_batch_size=mn,
              max_batch_size=mx, starting_batch_size=start, target_num_samples=target)
rays_tr, rgb_tr = torch.as_tensor(g["rays_train"]).to(dev), torch.as_tensor(g["rgb_train"]).to(dev)
rays_te, rgb_te = torch.as_tensor(g["rays_test"]).to(dev), torch.as_tensor(g["rgb_test"]).to(dev)
focal, n_views = float(g["focal"]), rays_te.shape[0] // (res * res)
at = [int(v) for v in g["psnr_at"]]
n_total = rays_tr.shape[0]


def run(seed, ref_init, cpu_noise):
    torch.manual_seed(20211200 + 1000 + seed)
    nerf, _ = build_model(grid=G0, bg_resolution=BG, device=dev, overrides=over)
    if ref_init:
        s = seed % 6
        sd = {k[len(f"s{s}/init/"):]: torch.as_tensor(g[k]) for k in g.files if k.startswith(f"s{s}/init/")}
        nerf.load_state_dict(sd, strict=False)
        nerf.model.brdf.bias, nerf.model.diffuse_module.diffuse_bias, nerf.model.diffuse_module.roughness_bias = (float(v) for v in g[f"s{s}/biases"])
    nerf.train()
    nerf.sampler.update(nerf.rf, init=True)
    if not ref_init:
        with torch.no_grad():
            xyz = torch.rand(100000, 4, device=dev) * 2 - 1
            xyz[:, 3] *= 0
            nerf.model.calibrate(None, xyz, nerf.rf.compute_appfeature(xyz), nerf.bg_module.mean_color().mean())
    tr = Trainer(nerf, params)
    torch.manual_seed(777 + seed)
    noise = ReplayNoise(dev, None) if cpu_noise else DeviceNoise(dev, seed=5000 + seed)
    gen = torch.Generator(device=dev).manual_seed(seed)
    perm, cur, row = torch.randperm(n_total, device=dev, generator=gen), 0, []
    for it in range(at[-1]):
        nb = tr.lbatch_size()
        if cur + nb > n_total:
            perm, cur = torch.randperm(n_total, device=dev, generator=gen), 0
        ids = perm[cur:cur + nb]
        cur += nb
        tr.step(rays_tr[ids], rgb_tr[ids], focal, noise=noise, global_rays=nb)
        if it + 1 in at:
            nerf.eval()
            pred = render_images(nerf, rays_te, focal, 4096, noise, draw_debug=True)
            nerf.train()
            pv, gv = pred.reshape(n_views, -1, 3), rgb_te.reshape(n_views, -1, 3)
            row.append(float(torch.stack([psnr_8bit(pv[i], gv[i]) for i in range(n_views)]).mean()))
    return row


N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
se = lambda x: x.std(0, ddof=1) / np.sqrt(x.shape[0])
ref, _ = bench.reference_psnr_seeds()
print("reference", ref.shape[0], ref.mean(0).round(3), se(ref).round(3), flush=True)
want = (sys.argv[2] if len(sys.argv) > 2 else "ref+cpu,ref+dev,own+cpu,own+dev").split(",")
for ref_init in (True, False):
    for cpu_noise in (True, False):
        if ("ref" if ref_init else "own") + "+" + ("cpu" if cpu_noise else "dev") not in want:
            continue
        t0 = time.time()
        rows = np.asarray([run(s, ref_init, cpu_noise) for s in range(N)])
        print(f"init {'reference states' if ref_init else 'own constructors '}  noise {'CPU generator' if cpu_noise else 'device       '}  n {N}: "
              f"{rows.mean(0).round(3)} +- {se(rows).round(3)}   ({time.time() - t0:.0f} s)", flush=True)
