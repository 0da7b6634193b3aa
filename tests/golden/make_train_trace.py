"""Container-only: runs the REFERENCE's training loop -- `reconstruction()` of /root/reference/train.py, unmodified -- for a
few dozen iterations on a small synthetic scene and records what SURVEY 8(f1) lists: per chunk `num_rays`, rays kept and
`n_samples`, `max_retrace_rays`; per iteration `lbatch_size`, the loss of every chunk, per-group learning rates,
per-tensor gradient norms and post-step parameter checksums; the forced upsample + optimizer restart; train / test PSNR.
Writes tests/golden/train_trace.npz (arrays only).

    python tests/golden/make_train_trace.py

How the reference is driven without its absent dependencies (none of which carries arithmetic):
  * hydra / omegaconf: the YAML files are composed by nmf_amd.yaml_config (pinned to the reference's own resolved example,
    tests/test_config.py) into plain dicts wrapped in an attribute-dict; `hydra.utils.instantiate` is a 15-line
    `_target_` / `_partial_` interpreter that imports the REFERENCE's classes;
  * dataset: `dataLoader.dataset_dict["synthetic"]` = a tiny class with the attributes train.py reads, serving rays of the
    S2 orbit cameras and colours rendered by the reference model itself from the S1 scene (eval mode);
  * tensorboard / loguru / tqdm / imageio / cv2: inert stubs (tests/golden/ref_harness.py).
Observation only: Adam.step, Tensor.backward and train.renderer are wrapped to RECORD; nothing the loop computes is changed.
"""
import functools
import importlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_harness as rh  # noqa: E402
from nmf_amd import synthetic, yaml_config  # noqa: E402

# the 40-iteration trace (train_trace.npz); make_psnr_trace.py calls run() with a longer, larger configuration
KNOBS = dict(grid0=32, grid1=40, teacher_grid=32, bg=32, upsample_at=(15,), n_iters=40, psnr_at=(10, 20, 30, 40), res=20,
             train_views=10, test_views=2, seed=20211200, batch=512, max_batch=1000, max_samples=20000,
             max_brdf_rays=(40000, 20000), target_num_samples=40000, max_retrace=200, rays_per_ray=32, light=False,
             threads=8, stop_at=None)       # stop_at: leave the loop after that many iterations (n_iters keeps sizing the lr decay)


class Cfg(dict):
    """attribute + item access like an OmegaConf node (hasattr() is False for missing keys)"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v


def wrap(x):
    if isinstance(x, dict):
        return Cfg({k: wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [wrap(v) for v in x]
    return x


def instantiate(node, *a, **kw):
    """hydra.utils.instantiate for plain dicts, resolving `_target_` inside the reference tree"""
    if isinstance(node, dict):
        if "_target_" in node:
            mod, _, name = node["_target_"].rpartition(".")
            cls = getattr(importlib.import_module(mod), name)
            kwargs = {k: instantiate(v) for k, v in node.items() if k not in ("_target_", "_partial_")}
            return functools.partial(cls, **kwargs) if node.get("_partial_", False) else cls(**kwargs)
        return Cfg({k: instantiate(v) for k, v in node.items()})
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    return node


_DATA = {}


def run(**knobs):
    """-> dict of arrays (what main() writes).  light=True keeps the data set, the initial state and the test PSNR only."""
    K = dict(KNOBS, **knobs)
    GRID0, GRID1, BG, N_ITERS = K["grid0"], K["grid1"], K["bg"], K["n_iters"]
    UPSAMPLE_AT, PSNR_AT, RES, SEED = list(K["upsample_at"]), tuple(K["psnr_at"]), K["res"], K["seed"]
    N_TRAIN_VIEWS, N_TEST_VIEWS = K["train_views"], K["test_views"]
    torch.set_num_threads(K["threads"])
    rh.install_stubs()
    sys.modules["hydra"].utils = sys.modules["hydra.utils"]
    sys.modules["hydra.utils"].instantiate = instantiate
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = rh._Anything
    sys.modules["torch.utils.tensorboard"] = tb
    kornia = sys.modules["kornia"]
    kornia.create_meshgrid = lambda H, W, normalized_coordinates=False: torch.stack(
        torch.meshgrid(torch.arange(W, dtype=torch.float32), torch.arange(H, dtype=torch.float32), indexing="xy"), -1)[None]
    import train as ref_train                                       # /root/reference/train.py
    from dataLoader import dataset_dict

    # ---- ground truth: the reference model itself renders the S1 scene from the orbit cameras (eval mode)
    dkey = (K["teacher_grid"], BG, RES, N_TRAIN_VIEWS, N_TEST_VIEWS)
    if dkey not in _DATA:                                            # several seeds train on one data set
        TG = K["teacher_grid"]
        teacher = rh.build_reference(grid=TG, bg_resolution=BG, seed=0, max_samples=20000, max_brdf_rays=(40000, 20000),
                                     max_retrace_rays=(40000,), target_num_samples=(40000,))
        teacher.load_state_dict(synthetic.state_dict_s1(grid=TG, bg_resolution=BG, seed=0), strict=False)
        teacher.sampler.update(teacher.rf, init=False)
        teacher.sampler.update(teacher.rf, init=True)
        teacher.eval()
        rays_tr, focal = synthetic.orbit_rays(N_TRAIN_VIEWS, RES, seed=1)
        rays_te, _ = synthetic.orbit_rays(N_TEST_VIEWS, RES, seed=2)
        torch.manual_seed(7)
        with torch.no_grad():
            rgb_tr = torch.cat([teacher(rays_tr[i:i + 800], focal, bg_col=torch.ones(3), is_train=False, ndc_ray=False)[0]["rgb_map"]
                                for i in range(0, rays_tr.shape[0], 800)])
            rgb_te = torch.cat([teacher(rays_te[i:i + 800], focal, bg_col=torch.ones(3), is_train=False, ndc_ray=False)[0]["rgb_map"]
                                for i in range(0, rays_te.shape[0], 800)])
        del teacher
        _DATA[dkey] = (rays_tr, rgb_tr, rays_te, rgb_te, focal)
    rays_tr, rgb_tr, rays_te, rgb_te, focal = _DATA[dkey]

    class SyntheticDataset:
        def __init__(self, datadir, split="train", downsample=1.0, is_stack=False, stack_norms=False, white_bg=True,
                     is_testing=False, **kw):
            r, c = (rays_tr, rgb_tr) if split == "train" else (rays_te, rgb_te)
            self.all_rays = r if not is_stack else r.reshape(-1, RES * RES, 6)
            self.all_rgbs = c if not is_stack else c.reshape(-1, RES, RES, 3)
            self.scene_bbox = torch.tensor([[-1.5] * 3, [1.5] * 3])
            self.near_far, self.white_bg, self.hdr, self.stack_norms = [2.5, 7.0], True, False, False
            self.fx, self.focal, self.img_wh = focal, [focal, focal], (RES, RES)

    dataset_dict["synthetic"] = SyntheticDataset

    # ---- configuration: the reference's YAML, shrunk to a scene a CPU trains in a minute
    small = [
        "model=microfacet_tensorf2", "field=tensorf_og", "dataset=lego", "expname=trace", "render_test=false", "N_vis=0",
        "vis_every=1000000000", "progress_refresh_rate=1000000000", f"seed={SEED}",
        "dataset.dataset_name=synthetic", "dataset.scenedir=synthetic/s2", "dataset.gt_bg=null",
        f"field.grid_size=[{GRID0},{GRID0},{GRID0}]", f"field.N_voxel_init={GRID0 ** 3}", f"field.N_voxel_final={GRID1 ** 3}",
        f"field.upsamp_list={UPSAMPLE_AT}".replace(" ", ""),
        f"model.arch.sampler.update_list={UPSAMPLE_AT}".replace(" ", ""), f"model.arch.sampler.max_samples={K['max_samples']}",
        f"model.arch.model.max_brdf_rays={list(K['max_brdf_rays'])}".replace(" ", ""),
        f"model.arch.model.target_num_samples=[{K['target_num_samples']}]",
        f"model.arch.model.max_retrace_rays=[{K['max_retrace']}]", f"model.arch.model.rays_per_ray={K['rays_per_ray']}",
        f"model.arch.bg_module.bg_resolution={BG}",
        f"model.params.n_iters={N_ITERS}", f"model.params.batch_size={K['batch']}", f"model.params.min_batch_size={K['batch']}",
        f"model.params.max_batch_size={K['max_batch']}", "model.params.starting_batch_size=100",
        f"model.params.target_num_samples={K['max_samples']}",
    ] + list(K.get("extra_overrides", ()))          # (e.g. "model.params.ori_lambda=0": make_psnr_traj.py --override)
    cfg = yaml_config.compose(os.path.join(rh.REF, "configs"), small)
    args = wrap(cfg)
    tmp = tempfile.mkdtemp(prefix="nmf_trace_")
    args.basedir, args.datadir = tmp, tmp

    # ---- observers ---------------------------------------------------------------------------------------------------
    T = dict(chunk_num_rays=[], chunk_rays_in=[], chunk_kept=[], chunk_n_samples=[], chunk_iter=[], chunk_loss=[],
             chunk_max_retrace=[], iter_lbatch=[], iter_lr=[], iter_gradnorm=[], iter_checksum=[], iter_optimizer=[],
             iter_grid=[], iter_detach_N=[], iter_max_retrace=[], iter_num_chunks=[])
    state = dict(tensorf=None, names=None, it=0, pending_loss=[], rng_at_loop=None, init_sd=None, biases=None)
    orig_instantiate = instantiate

    def spy_instantiate(node, *a, **kw):
        out = orig_instantiate(node, *a, **kw)
        if isinstance(node, dict) and node.get("_target_", "").endswith("TensorNeRF"):
            def build(*aa, **kk):
                state["tensorf"] = out(*aa, **kk)
                return state["tensorf"]
            return build
        return out

    sys.modules["hydra.utils"].instantiate = spy_instantiate
    orig_renderer = ref_train.renderer

    def spy_renderer(rays, tensorf, *a, **kw):
        if state["rng_at_loop"] is None:          # first chunk: everything before the loop (init, calibration) is done
            state["rng_at_loop"] = None           # (the generator state is taken in SimpleSampler.nextids, see below)
        ims, stats = orig_renderer(rays, tensorf, *a, **kw)
        if kw.get("is_train", False):
            T["chunk_num_rays"].append(int(kw["chunk"]))
            T["chunk_rays_in"].append(int(rays.shape[0]))
            T["chunk_kept"].append(int(stats["whole_valid"].sum()))
            ns = [int(v) for v in stats["n_samples"]]
            T["chunk_n_samples"].append(ns + [0] * (2 - len(ns)))
            T["chunk_iter"].append(state["it"])
            T["chunk_max_retrace"].append(int(tensorf.model.max_retrace_rays[0]))
        return ims, stats

    ref_train.renderer = spy_renderer
    orig_backward = torch.Tensor.backward

    def spy_backward(t, *a, **kw):
        if t.dim() == 0:
            T["chunk_loss"].append(float(t.detach()))
        return orig_backward(t, *a, **kw)

    torch.Tensor.backward = spy_backward
    orig_nextids = ref_train.SimpleSampler.nextids

    def spy_nextids(smp, batch=None):
        if state["init_sd"] is None:              # the very first draw of the loop: snapshot model + generator
            nerf = state["tensorf"]
            state["init_sd"] = {k: v.detach().clone() for k, v in nerf.state_dict().items()}
            state["biases"] = (float(nerf.model.brdf.bias), float(nerf.model.diffuse_module.diffuse_bias),
                               float(nerf.model.diffuse_module.roughness_bias))
            state["rng_at_loop"] = torch.get_rng_state().clone()
        return orig_nextids(smp, batch)

    ref_train.SimpleSampler.nextids = spy_nextids
    orig_step = torch.optim.Adam.step

    def spy_step(opt, *a, **kw):
        nerf = state["tensorf"]
        names = {id(p): n for n, p in nerf.named_parameters()}
        T["iter_lr"].append([float(g["lr"]) for g in opt.param_groups])
        gn = {}
        for g in opt.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    gn[names[id(p)]] = float(p.grad.norm())
        T["iter_gradnorm"].append(gn)
        out = orig_step(opt, *a, **kw)
        T["iter_checksum"].append({n: (float(p.detach().double().sum()), float(p.detach().double().norm()))
                                   for n, p in nerf.named_parameters()})
        T["iter_optimizer"].append(id(opt))
        T["iter_grid"].append(int(nerf.rf.density_rf.app_plane[0].shape[-1]))
        T["iter_detach_N"].append(bool(nerf.model.detach_N))
        T["iter_max_retrace"].append(int(nerf.model.max_retrace_rays[0]))
        n_chunks = sum(1 for i in T["chunk_iter"] if i == state["it"])
        T["iter_num_chunks"].append(n_chunks)
        T["iter_lbatch"].append(sum(r for r, i in zip(T["chunk_rays_in"], T["chunk_iter"]) if i == state["it"]))
        state["it"] += 1
        return out

    def test_psnr(nerf):
        """renderer.py:399-401,511-513 on the held-out views, eval mode; the generator is forked so that the observation
        does not move the training run's random stream"""
        with torch.random.fork_rng():
            torch.manual_seed(11)
            was = nerf.training
            nerf.eval()
            with torch.no_grad():
                pred = torch.cat([nerf(rays_te[i:i + 800], focal, bg_col=torch.ones(3), is_train=False, ndc_ray=False)[0]["rgb_map"]
                                  for i in range(0, rays_te.shape[0], 800)])
            nerf.train(was)
        q = (torch.floor(pred.clip(0, 1) * 255) / 255).reshape(N_TEST_VIEWS, -1, 3)
        gt = rgb_te.reshape(N_TEST_VIEWS, -1, 3).clip(0, 1)
        return [-10.0 * float(torch.log10(((q[i] - gt[i]) ** 2).mean())) for i in range(N_TEST_VIEWS)]

    T["test_psnr"] = []
    inner_step = spy_step

    class _Stop(Exception):
        pass

    def spy_step_psnr(opt, *a, **kw):
        out = inner_step(opt, *a, **kw)
        if state["it"] in PSNR_AT:
            T["test_psnr"].append(test_psnr(state["tensorf"]))
            print(f"iteration {state['it']}: test PSNR {np.round(T['test_psnr'][-1], 3).tolist()}", flush=True)
        if K["stop_at"] is not None and state["it"] >= K["stop_at"]:
            raise _Stop()                 # the loop is left as it is; only its length is cut
        return out

    torch.optim.Adam.step = spy_step_psnr

    torch.manual_seed(SEED)                        # train.py:906-908
    np.random.seed(SEED)
    args.model.arch.rf = args.field                # train.py:911
    try:
        ref_train.reconstruction(args)
    except _Stop:
        pass
    finally:
        torch.Tensor.backward = orig_backward
        torch.optim.Adam.step = orig_step
        ref_train.SimpleSampler.nextids = orig_nextids
        ref_train.renderer = orig_renderer
    nerf = state["tensorf"]
    assert state["it"] == (K["stop_at"] or N_ITERS), state["it"]
    per_img = T["test_psnr"][-1]

    out = dict(grid0=GRID0, grid1=GRID1, bg_res=BG, upsample_at=UPSAMPLE_AT[0] if len(UPSAMPLE_AT) == 1 else np.asarray(UPSAMPLE_AT),
               n_iters=N_ITERS, res=RES, focal=focal, seed=SEED,
               rays_train=rays_tr, rgb_train=rgb_tr, rays_test=rays_te, rgb_test=rgb_te,
               rng_state_at_loop=state["rng_at_loop"], biases=np.asarray(state["biases"]),
               test_psnr=np.asarray(T["test_psnr"]), psnr_at=np.asarray(PSNR_AT), overrides="\n".join(small))
    for k, v in state["init_sd"].items():
        out["init/" + k] = v
    out["params_params"] = np.asarray([args.model.params[k] for k in ("min_batch_size", "max_batch_size", "starting_batch_size",
                                                                      "target_num_samples")])
    if K["light"] == "traj":
        # the run's TRAJECTORY and nothing heavy (make_psnr_traj.py): per chunk the loss / ray controller / re-trace controller /
        # sample counts, per iteration the global batch and the learning-rate factor, the test PSNR at the evaluations
        traj = dict(seed=SEED, test_psnr=np.asarray(T["test_psnr"]), psnr_at=np.asarray(PSNR_AT))
        for k in ("chunk_num_rays", "chunk_rays_in", "chunk_kept", "chunk_n_samples", "chunk_iter", "chunk_loss", "chunk_max_retrace",
                  "iter_lbatch", "iter_lr", "iter_max_retrace", "iter_num_chunks"):
            traj[k] = np.asarray(T[k])
        names = sorted({n for d in T["iter_gradnorm"] for n in d})
        traj["gradnorm_names"] = "\n".join(names)
        traj["iter_gradnorm"] = np.asarray([[d.get(n, np.nan) for n in names] for d in T["iter_gradnorm"]], dtype=np.float32)
        pnames = sorted(T["iter_checksum"][0])
        traj["param_names"] = "\n".join(pnames)
        traj["iter_param_norm"] = np.asarray([[d[n][1] for n in pnames] for d in T["iter_checksum"]], dtype=np.float32)
        return traj
    if K["light"]:
        print("test psnr per image at", PSNR_AT, np.round(np.asarray(T["test_psnr"]), 3).tolist(), flush=True)
        return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()}
    for k in ("chunk_num_rays", "chunk_rays_in", "chunk_kept", "chunk_n_samples", "chunk_iter", "chunk_loss",
              "chunk_max_retrace", "iter_lbatch", "iter_lr", "iter_grid", "iter_detach_N", "iter_max_retrace",
              "iter_num_chunks"):
        out[k] = np.asarray(T[k])
    ids = {}
    out["iter_optimizer"] = np.asarray([ids.setdefault(i, len(ids)) for i in T["iter_optimizer"]])
    names = sorted({n for d in T["iter_gradnorm"] for n in d})
    out["gradnorm_names"] = "\n".join(names)
    out["iter_gradnorm"] = np.asarray([[d.get(n, np.nan) for n in names] for d in T["iter_gradnorm"]])
    pnames = sorted(T["iter_checksum"][0])
    out["param_names"] = "\n".join(pnames)
    out["iter_param_norm"] = np.asarray([[d[n][1] for n in pnames] for d in T["iter_checksum"]])
    out["iter_param_sum"] = np.asarray([[d[n][0] for n in pnames] for d in T["iter_checksum"]])
    flat = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()}
    print("lbatch", out["iter_lbatch"].tolist())
    print("num_rays per chunk", out["chunk_num_rays"].tolist())
    print("max_retrace", out["iter_max_retrace"].tolist())
    print("grid", out["iter_grid"].tolist(), "optimizer", out["iter_optimizer"].tolist())
    print("loss", [round(v, 5) for v in out["chunk_loss"].tolist()][:40])
    print("test psnr per image at", PSNR_AT, T["test_psnr"])
    return flat


def main():
    flat = run()
    path = os.path.join(HERE, "train_trace.npz")
    np.savez_compressed(path, **flat)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


if __name__ == "__main__":
    main()
