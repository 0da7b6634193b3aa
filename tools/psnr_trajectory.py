"""where do the reference's and this build's 300-iteration S2 trainings part?  Seed-MEAN trajectories of both sides,
300 points of statistical power instead of the 3 PSNR evaluations: per iteration the loss the loop back-propagates, the global batch
and the chunk count (ray controller, train.py:618-626), max_retrace_rays (models/microfacet.py:241-268), the sample counts of both
levels, every group's learning rate, gradient norms and parameter norms.

  reference side: tests/golden/psnr_ref_traj_*.npz (tests/golden/make_psnr_traj.py, the reference's own loop on CPU)
  this build:     bench.psnr_runs(..., traj=[]) on the GPU, `--seeds` fresh initialisations

Prints, per quantity, the seed means at a few iterations, the largest |z| = |mean difference| / standard error of the difference over
all iterations and the iteration ranges where |z| > 3, and writes the table to --out (profiles/)."""
import argparse
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_iter(T, n_it):
    """chunk records -> per-iteration arrays"""
    it = np.asarray(T["chunk_iter"]).astype(int)
    out = {}
    ns = np.asarray(T["chunk_n_samples"]).reshape(len(it), -1)

    def acc(v, how="sum"):
        r = np.zeros(n_it)
        cnt = np.zeros(n_it)
        np.add.at(r, it, np.asarray(v, dtype=np.float64))
        np.add.at(cnt, it, 1.0)
        return r if how == "sum" else r / np.maximum(cnt, 1)
    out["loss"] = acc(T["chunk_loss"])
    out["rays_kept"] = acc(T["chunk_kept"])
    out["rays_in"] = acc(T["chunk_rays_in"])
    out["num_rays(chunk)"] = acc(T["chunk_num_rays"], "mean")
    out["n_samples0"] = acc(ns[:, 0])
    out["n_samples1"] = acc(ns[:, 1])
    out["max_retrace(chunk)"] = acc(T["chunk_max_retrace"], "mean")
    out["chunks"] = np.asarray(T["iter_num_chunks"], dtype=np.float64)[:n_it]
    out["lbatch"] = np.asarray(T["iter_lbatch"], dtype=np.float64)[:n_it]
    lr = np.asarray(T["iter_lr"], dtype=np.float64)[:n_it]
    for g in range(lr.shape[1]):
        out[f"lr[{g}]"] = lr[:, g]
    return out


def load_reference(pattern="psnr_ref_traj_*.npz"):
    runs, names, pnames = [], None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", pattern))):
        with np.load(f) as z:
            seeds = sorted({k.split("/")[0] for k in z.files if "/" in k})
            names = str(z["gradnorm_names"]).split("\n") if "gradnorm_names" in z.files else names
            pnames = str(z["param_names"]).split("\n") if "param_names" in z.files else pnames
            for s in seeds:
                if f"{s}/iter_gradnorm" not in z.files:
                    continue
                runs.append({k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(s + "/")})
    return runs, names, pnames


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=32)
    ap.add_argument("--window", type=int, default=25)
    ap.add_argument("--save", default=None, help="npz that receives this build's trajectories")
    ap.add_argument("--ref-glob", default="psnr_ref_traj_*.npz", help="reference files under tests/golden")
    ap.add_argument("--sampler", default="reference", choices=("disjoint", "reference"),
                    help="which rays a chunk gets: disjoint slices of a permutation, or train.py:34-51 as it is (overlapping chunks)")
    ap.add_argument("--param", action="append", default=[], metavar="KEY=VALUE", help="model.params key changed for this build's runs")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "psnr_trajectory.txt"))
    a = ap.parse_args()
    import torch
    import bench
    ref, gnames, pnames = load_reference(a.ref_glob)
    if not ref:
        raise SystemExit(f"no tests/golden/{a.ref_glob}")
    over = {}
    for kv in a.param:                     # a factor experiment: the same key changed on both sides (make_psnr_traj.py --override)
        k, v = kv.split("=")
        over[k] = float(v)
    n_it = 300
    mine = []
    hip, _, _ = bench.psnr_runs(torch.device("cuda", 0), range(a.seeds), traj=mine, params_over=over, sampler=a.sampler)
    if a.save:          # the build's own runs, for offline analysis next to the reference's files
        flat = {}
        for i, t in enumerate(mine):
            for k, v in t.items():
                if k != "names":
                    flat[f"s{i}/{k}"] = np.asarray(v)
        flat["names"] = np.asarray("\n".join(mine[0]["names"]))
        np.savez_compressed(a.save, **flat)
    R = [per_iter(t, n_it) for t in ref]
    H = [per_iter(t, n_it) for t in mine]
    # gradient / parameter norms by parameter name (the reference's names = this build's: the state_dict contract)
    for side, runs, names_g, names_p in ((R, ref, gnames, pnames), (H, mine, None, None)):
        for d, t in zip(side, runs):
            ng = names_g if names_g is not None else list(t["names"])
            npar = names_p if names_p is not None else list(t["names"])
            gn = np.asarray(t["iter_gradnorm"], dtype=np.float64)[:n_it]
            pn = np.asarray(t["iter_param_norm"], dtype=np.float64)[:n_it]
            for j, n in enumerate(ng):
                d["gradnorm/" + n] = gn[:, j]
            for j, n in enumerate(npar):
                d["paramnorm/" + n] = pn[:, j]
    keys = [k for k in R[0] if k in H[0]]
    lines = [f"# seed-mean trajectories of the 300-iteration S2 training: reference (its own loop, CPU) {len(R)} runs, this build {len(H)} runs",
             f"# test PSNR here {np.round(hip.mean(0), 3).tolist()} +- {np.round(hip.std(0, ddof=1) / np.sqrt(len(H)), 3).tolist()}; "
             f"reference runs of this file {np.round(np.mean([t['test_psnr'].mean(-1) for t in ref], 0), 3).tolist()}",
             "# z = (mean here - mean reference) / standard error of that difference, per iteration; `>3` = iterations with |z| > 3",
             f"{'quantity':44s} {'it':>4s} " + " ".join(f"{'ref@' + str(i):>11s} {'here@' + str(i):>11s}" for i in (10, 50, 100, 150, 200, 299)) + "   max|z| at   >3"]
    at = (10, 50, 100, 150, 200, 299)
    W = a.window
    def blocks(x):                       # [runs, 300] -> means over windows of W iterations (per run: the unit of the statistics is the run)
        n = (x.shape[1] // W) * W
        return x[:, :n].reshape(x.shape[0], -1, W).mean(-1)
    block_rows = []
    for k in keys:
        r = np.stack([d[k] for d in R])
        h = np.stack([d[k] for d in H])
        ok = np.isfinite(r).all(0) & np.isfinite(h).all(0)
        se = np.sqrt(r.var(0, ddof=1) / len(R) + h.var(0, ddof=1) / len(H))
        diff = h.mean(0) - r.mean(0)
        same = np.abs(diff) <= 1e-5 * np.maximum(np.abs(r.mean(0)), 1e-30)          # equal up to float rounding (learning rates, batch sizes)
        dz = np.where(ok & (se > 0) & ~same, diff / np.where(se > 0, se, 1), 0.0)
        if np.isfinite(r).all() and np.isfinite(h).all():
            rb, hb = blocks(r), blocks(h)
            seb = np.sqrt(rb.var(0, ddof=1) / len(R) + hb.var(0, ddof=1) / len(H))
            db = hb.mean(0) - rb.mean(0)
            sameb = np.abs(db) <= 1e-5 * np.maximum(np.abs(rb.mean(0)), 1e-30)
            zb = np.where((seb > 0) & ~sameb, db / np.where(seb > 0, seb, 1), 0.0)
            rel = db / np.maximum(np.abs(rb.mean(0)), 1e-30)
            block_rows.append((k, zb, rel))
        big = np.nonzero(np.abs(dz) > 3)[0]
        rng = ""
        if big.size:
            runs_, start = [], big[0]
            for x, y in zip(big, list(big[1:]) + [None]):
                if y is None or y != x + 1:
                    runs_.append(f"{start}-{x}" if x != start else f"{x}")
                    start = y
            rng = ",".join(runs_[:6]) + (" ..." if len(runs_) > 6 else "")
        imax = int(np.abs(dz).argmax())
        lines.append(f"{k[:44]:44s} {'':4s} " + " ".join(f"{r.mean(0)[i]:11.5g} {h.mean(0)[i]:11.5g}" for i in at)
                     + f"   {abs(dz[imax]):5.1f} @{imax:<4d} {len(big):3d} {rng}")
    lines.append("")
    lines.append(f"# the same as means over windows of {W} iterations (per run first): z per window, and below it the relative difference "
                 "(here - reference) / |reference| in per cent")
    lines.append(f"{'quantity':44s} " + " ".join(f"{i * W:>6d}" for i in range(n_it // W)))
    for k, zb, rel in block_rows:
        if np.abs(zb).max() == 0:
            continue
        lines.append(f"{k[:44]:44s} " + " ".join(f"{v:6.1f}" for v in zb))
        lines.append(f"{'   rel %':44s} " + " ".join(f"{100 * v:6.2f}" for v in rel))
    text = "\n".join(lines)
    print(text)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    open(a.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
