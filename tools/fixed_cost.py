"""Per-kernel FIXED cost of the training step: the same launch sequence on a tiny chunk under rocprofv3 -- what every kernel takes
when it has (almost) nothing to do: launch ramp, prologue (weight staging, table scans), cold instruction cache, epilogue.

    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python tools/fixed_cost.py [rays_per_step=64] [steps=40] [bounce budget=0]
    python tools/fixed_cost.py --table OUT_small OUT_full [steps]      -> kernel, launches/step, us at the tiny size, us at full size
"""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def stats(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0]
        c, t = out.get(name, (0, 0.0))
        out[name] = (c + int(r["Calls"]), t + float(r["TotalDurationNs"]))
    return out


def table(small, full, steps):
    a, b = stats(small), stats(full)
    rows = []
    for k in sorted(set(a) | set(b)):
        ca, ta = a.get(k, (0, 0.0))
        cb, tb = b.get(k, (0, 0.0))
        rows.append((tb / 1e3 / steps, k, cb / steps, ta / max(ca, 1) / 1e3, tb / max(cb, 1) / 1e3, ta / 1e3 / steps))
    rows.sort(reverse=True)
    print(f"{'kernel':44s} {'n/step':>6s} {'tiny us':>8s} {'full us':>8s} {'tiny us/step':>12s} {'full us/step':>12s}")
    for tot, k, n, ua, ub, sa in rows:
        if n >= 0.5:
            print(f"{k[:44]:44s} {n:6.1f} {ua:8.1f} {ub:8.1f} {sa:12.1f} {tot:12.1f}")
    print(f"sum per step: tiny {sum(r[5] for r in rows if r[2] >= 0.5):.0f} us, full {sum(r[0] for r in rows if r[2] >= 0.5):.0f} us")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--table":
        return table(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 40)
    import torch
    import bench
    from nmf_amd.noise import DeviceNoise
    from nmf_amd.trainer import Trainer
    rays_n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    budget = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # secondary-ray budgets (0: the configured 250 k / 50 k ones)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    nerf, params = bench.build(dev)
    if budget:      # the bounce budgets do not shrink with the chunk: 64 rays still spawn 4 k + 64 k secondary rays
        nerf.model.max_brdf_rays = [budget for _ in nerf.model.max_brdf_rays]
        nerf.model.max_retrace_rays = [budget]
    tr = Trainer(nerf, params)
    noise = DeviceNoise(dev, seed=1)
    batches, focal = bench.make_batches(nerf, 8, rays_n, 0, dev, distinct=8)
    for i in range(steps):
        tr.step(*batches[i % 8], focal, noise=noise, update_controllers=False, fixed_chunk=rays_n)
    torch.cuda.synchronize()
    print("sizes", tr.fast.last_sizes)


if __name__ == "__main__":
    main()
