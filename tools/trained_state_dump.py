"""GPU: one S2 training of bench.psnr_runs, the model's state after iteration N written to an npz (state_dict, the three calibrated
biases, the controllers' values) -- the input of tests/golden/make_golden.py trained_step, which runs ONE training chunk of the
reference from that state (round 6: do the two sides' single-step gradients agree at a TRAINED state, not only at scene S1?).
    python tools/trained_state_dump.py [iteration [seed [out]]]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", f"trained_state_it{N}.npz")
got = {}


def on_iter(seed_, it, nerf, tr):
    if it + 1 == N and not got:
        for k, v in nerf.state_dict().items():
            got["sd/" + k] = v.detach().cpu().numpy()
        got["biases"] = np.asarray([nerf.model.brdf.bias, nerf.model.diffuse_module.diffuse_bias, nerf.model.diffuse_module.roughness_bias], dtype=np.float64)
        got["max_retrace"] = np.asarray(int(nerf.model.max_retrace_rays[0]))
        got["num_rays"] = np.asarray(int(tr.num_rays))
        got["min_rough"] = np.asarray(float(nerf.model.min_rough))
        got["iteration"] = np.asarray(N)
        got["ori_lambda"], got["pred_lambda"] = np.asarray(tr.ori_lambda), np.asarray(tr.pred_lambda)


hip, _, _ = bench.psnr_runs(torch.device("cuda", 0), [seed], on_iter=on_iter)
os.makedirs(os.path.dirname(out), exist_ok=True)
np.savez_compressed(out, **got)
print("wrote", out, "max_retrace", int(got["max_retrace"]), "num_rays", int(got["num_rays"]), "psnr", hip)
