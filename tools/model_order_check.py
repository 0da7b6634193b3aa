"""Does the history of a process change the step time of a model?  Builds (and discards) another model first -- a 300^3 grid,
the PSNR run, or a large-chunk leg -- and then times a bf16-table and an fp32-table trainer interleaved.
    python tools/model_order_check.py none|grid|psnr|big"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from nmf_amd.noise import DeviceNoise  # noqa: E402
from nmf_amd.trainer import Trainer  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
what = sys.argv[1] if len(sys.argv) > 1 else "none"
sync = torch.cuda.synchronize
nerf0, params = bench.build(dev)
if what == "grid":
    for G in (196, 300):
        n_, _ = bench.build(dev, grid=G)
        tr = Trainer(n_, params)
        b, f = bench.make_batches(n_, 8, bench.CHUNK, 0, dev, distinct=8)
        bench.time_train(tr, b, f, DeviceNoise(dev, seed=5), 3, 5, bench.CHUNK, sync)
        del n_, tr, b
        if os.environ.get("NMF_CHECK_NO_EMPTY") != "1":
            torch.cuda.empty_cache()
elif what == "psnr":
    bench.psnr_at_iter(dev)
elif what == "big":
    bench.scale_budgets(nerf0, 4)
    tr = Trainer(nerf0, params)
    b, f = bench.make_batches(nerf0, 6, 32768, 0, dev, distinct=6)
    bench.time_train(tr, b, f, DeviceNoise(dev, seed=5), 2, 4, 4 * bench.CHUNK, sync)
    del tr, b
    torch.cuda.empty_cache()
del nerf0
legs = {}
for name in ("bf16", "f32"):
    nerf, params = bench.build(dev, table_dtype=name)
    tr = Trainer(nerf, params)
    batches, f = bench.make_batches(nerf, 16, bench.CHUNK, 0, dev, distinct=12)
    nz = DeviceNoise(dev, seed=5)
    bench.time_train(tr, batches, f, nz, 15, 1, bench.CHUNK, sync)
    legs[name] = (tr, batches, f, nz)
for r in range(2):
    for name, (tr, batches, f, nz) in legs.items():
        dt, _, last, _ = bench.time_train(tr, batches, f, nz, 0, 20, bench.CHUNK, sync)
        print(what, name, round(1e3 * dt / 20, 4), "ms", last["n_samples"], flush=True)
