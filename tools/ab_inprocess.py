"""In-process A/B of the training step: ONE trainer, the variant switched every BLOCK steps for ROUNDS rounds, so clock /
thermal drift and box-to-box differences (4 % between `python bench.py` runs on different boxes) cancel; resolves ~0.5 %.

    python tools/ab_inprocess.py <what> <v0,v1,...> [rounds [max_retrace_rays]]

<what>:
    NAME                 os.environ[NAME] = value            (knobs that are read at call time)
    obj:ATTR             TrainPass attribute, as bool        (obj:overlap 0,1   obj:sparse_normals 0,1)
    core:ATTR            StepCore attribute, as bool         (core:env_split 0,1)
    attr:NAME            nmf_amd.fast_step module constant   (attr:MLP_SIDE_WGS 64,128,256)
    hip:NAME             nmf_amd.hip module constant         (hip:ENV_BINNED_MIN_LOOKUPS 16384,4611686018427387904)
    calldelay:NAME       busy-wait of <value> us on the host in front of every call of the C++ wrapper NAME (csrc/host_ext.cpp)
    prio                 0,1: the training pass on torch's default stream / on a high-priority stream (side streams stay normal)
    delay:METHOD         busy-wait of <value> us on the host in front of TrainPass.METHOD (delay:_flush_walks 0,100) or,
                         with delay:hip.FUNC, in front of a wrapper of nmf_amd.hip (delay:hip.march_fill 0,50): shows whether
                         the host or the device bounds that stretch of the step
"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import nmf_amd.fast_step as fast_step  # noqa: E402
from nmf_amd import hip  # noqa: E402
from nmf_amd.noise import DeviceNoise  # noqa: E402
from nmf_amd.trainer import Trainer  # noqa: E402

BLOCK = 40


def main():
    var, vals = sys.argv[1], sys.argv[2].split(",")
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    retrace = int(sys.argv[4]) if len(sys.argv) > 4 else None      # partial re-trace instead of bench.py's steady state
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    grid = int(os.environ.get("NMF_AB_GRID", "128"))          # NMF_AB_GRID=300: the final grid of the schedule
    nerf, params = bench.build(dev, grid=grid)
    if retrace is not None:
        nerf.model.max_retrace_rays = [retrace]
    tr = Trainer(nerf, params)
    noise = DeviceNoise(dev, seed=1)
    batches, focal = bench.make_batches(nerf, 16, bench.CHUNK, 0, dev, distinct=16)
    delay_us = [0.0]

    def install_delay(where):
        tgt, name = (hip, where[4:]) if where.startswith("hip.") else (tr.fast, where)
        f = getattr(tgt, name)

        def delayed(*a, **k):
            if delay_us[0] > 0:
                end = time.perf_counter() + delay_us[0] * 1e-6
                while time.perf_counter() < end:
                    pass
            return f(*a, **k)
        setattr(tgt, name, delayed)

    if var.startswith("delay:"):
        install_delay(var[6:])

    def set_variant(v):
        if var.startswith("calldelay:"):       # calldelay:march_fill 0,50 -- host busy-wait in front of every call of that C++ wrapper
            hip.HOST_EXT.set_call_delay(var[10:], float(v))
        elif var == "prio":                      # prio 0,1: torch's default stream / a high-priority stream as the pass's main stream
            torch.cuda.synchronize()
            use_hi[0] = bool(int(v))
        elif var.startswith("delay:"):
            delay_us[0] = float(v)
        elif var.startswith("obj:"):
            setattr(tr.fast, var[4:], bool(int(v)))
        elif var.startswith("core:"):
            torch.cuda.synchronize()
            tr.fast.set_switch(var[5:], bool(int(v)))
        elif var.startswith("attr:"):
            setattr(fast_step, var[5:], int(v))
        elif var.startswith("hip:"):
            setattr(hip, var[4:], int(v))
        else:
            os.environ[var] = v

    hi_stream = torch.cuda.Stream(priority=-1)
    use_hi = [os.environ.get("NMF_MAIN_PRIO") == "1"]     # the pass's main stream at high priority, its side streams stay normal

    def run(n):
        hi = hi_stream if use_hi[0] else None
        if hi is not None:
            with torch.cuda.stream(hi):
                for i in range(n):
                    tr.step(*batches[i % 16], focal, noise=noise, update_controllers=False, fixed_chunk=bench.CHUNK)
            return
        for i in range(n):
            tr.step(*batches[i % 16], focal, noise=noise, update_controllers=False, fixed_chunk=bench.CHUNK)

    for v in vals:
        set_variant(v)
        run(30)
    res = {v: [] for v in vals}
    for r in range(rounds):
        for v in (vals if r % 2 == 0 else vals[::-1]):
            set_variant(v)
            run(5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(BLOCK)
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t0) / BLOCK * 1e3)
    for v in vals:
        print(f"{var}={v}: median {statistics.median(res[v]):.4f} ms  mean {statistics.mean(res[v]):.4f}  min {min(res[v]):.4f}  "
              + " ".join(f"{x:.3f}" for x in res[v]))


if __name__ == "__main__":
    main()
