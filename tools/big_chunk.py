"""One optimizer step over 32 768 rays (BASELINE configs[3]: rays per GPU): eight chunks of 4096 under the reference's budgets
(sampler.max_samples 200 000, model.max_brdf_rays [650 000, 450 000]) against fewer, larger chunks with the budgets scaled
by the same factor -- 288 GB of HBM do not need the reference's 200 k-sample cap.
    python tools/big_chunk.py [factor ...]      (chunk = 4096 * factor)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from nmf_amd.noise import DeviceNoise  # noqa: E402
from nmf_amd.trainer import Trainer  # noqa: E402
import nmf_amd.fast_step as fast_step  # noqa: E402

if os.environ.get("NMF_MLP_SIDE_WGS"):          # workgroup cap of the side-stream BRDF-MLP backward (tuned at 4096-ray chunks: 96)
    fast_step.MLP_SIDE_WGS = fast_step.MLP_SIDE_WGS_ENV = int(os.environ["NMF_MLP_SIDE_WGS"])

RAYS = 32768
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for f in [int(a) for a in sys.argv[1:]] or (1, 2, 4, 8):
    nerf, params = bench.build(dev)
    nerf.sampler.max_samples = 200000 * f
    nerf.model.max_brdf_rays = [650000 * f, 450000 * f]
    nerf.model.max_retrace_rays = [nerf.model.max_brdf_rays[0]]
    tr = Trainer(nerf, params)
    batches, focal = bench.make_batches(nerf, 12, RAYS, 0, dev, distinct=6)
    dt, rays_done, last, _ = bench.time_train(tr, batches, focal, DeviceNoise(dev, seed=5), 3, 8, bench.CHUNK * f,
                                              torch.cuda.synchronize)
    print(f"chunk {bench.CHUNK * f:6d} (budgets x{f}): {1e3 * dt / 8:7.2f} ms per 32 768-ray step, {rays_done / dt / 1e6:5.2f} M rays/s, "
          f"samples per chunk {last['n_samples']}, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    del tr, nerf, batches
    torch.cuda.empty_cache()
