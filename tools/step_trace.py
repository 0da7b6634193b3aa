"""Multi-stream timeline of one steady-state training step WITHOUT a profiler: every C-ABI wrapper of lib/_nmf_host.so
records a HIP event in front of and behind its call on the stream it launches on (CallTimer); this tool prints, for the last
of a few instrumented steps, each call with its stream, device start / end relative to the step's first call, the gap to the
previous call on the same stream, and when the HOST issued it (so that host-bound stretches show as device start ~ host issue).
rocprofv3 --kernel-trace serialises dependent dispatches ~10 us apart (profiles/*_steady_state_per_step.csv: span 2.2-2.6 ms for
a 1.6 ms step); this view costs two event records per call.

    python tools/step_trace.py [steps_before] [ENV=VALUE ...]      -> stdout + gpurun_out/step_trace.txt
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for a in sys.argv[2:]:
    if "=" in a:
        k, v = a.split("=", 1)
        os.environ[k] = v
import torch  # noqa: E402

import bench  # noqa: E402
from nmf_amd import hip  # noqa: E402
from nmf_amd.noise import DeviceNoise  # noqa: E402
from nmf_amd.trainer import Trainer  # noqa: E402


def main():
    warm = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    nerf, params = bench.build(dev, grid=int(os.environ.get("NMF_AB_GRID", "128")))
    tr = Trainer(nerf, params)
    noise = DeviceNoise(dev, seed=1)
    batches, focal = bench.make_batches(nerf, 16, bench.CHUNK, 0, dev, distinct=16)

    def run(n):
        for i in range(n):
            tr.step(*batches[i % 16], focal, noise=noise, update_controllers=False, fixed_chunk=bench.CHUNK)

    run(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(40)
    torch.cuda.synchronize()
    plain = (time.perf_counter() - t0) / 40 * 1e3
    fx = hip.HOST_EXT
    fx.call_timing_begin(os.environ.get("NMF_TRACE_ONLY", ""))      # e.g. "march_count,loss_head,adam_step": a few marks, no distortion
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(4)
    torch.cuda.synchronize()
    timed = (time.perf_counter() - t0) / 4 * 1e3
    tl = fx.call_timing_timeline()
    # steps end with adam_step; take the calls between the last two
    ends = [i for i, r in enumerate(tl) if r[0] == "adam_step"]
    lo, hi = (ends[-2] + 1, ends[-1] + 1) if len(ends) >= 2 else (0, len(tl))
    rows = tl[lo:hi]
    streams = {}
    for r in rows:
        streams.setdefault(r[1], len(streams))
    t_ref, h_ref = min(r[2] for r in rows), min(r[4] for r in rows)
    last_end = {}
    out = [f"# step {plain:.3f} ms uninstrumented, {timed:.3f} ms with events; {len(rows)} calls on {len(streams)} streams",
           f"{'start':>8s} {'end':>8s} {'dur':>7s} {'gap':>7s} {'issued':>8s} s  call"]
    busy = 0.0
    for name, st, a, b, h in sorted(rows, key=lambda r: r[2]):
        k = streams[st]
        gap = a - last_end.get(k, a)
        last_end[k] = b
        busy += b - a
        out.append(f"{a - t_ref:8.1f} {b - t_ref:8.1f} {b - a:7.1f} {gap:7.1f} {h - h_ref:8.1f} {k}  {'  ' * k}{name}")
    span = max(r[3] for r in rows) - t_ref
    out.append(f"# span {span:.1f} us, summed call time {busy:.1f} us")
    text = "\n".join(out)
    print(text)
    os.makedirs("gpurun_out", exist_ok=True)
    open(os.path.join("gpurun_out", os.environ.get("NMF_TRACE_OUT", "step_trace.txt")), "w").write(text + "\n")


if __name__ == "__main__":
    main()
