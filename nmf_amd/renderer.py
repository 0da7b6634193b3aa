"""Chunked rendering -- counterpart of the reference's renderer.py: `chunk_renderer` (:56-106, the callable train.py:541
and BundleRender use) and the evaluation PSNR (:399-401, :511-513)."""
from collections import defaultdict

import torch


def _stack(d):
    out = {}
    for k, v in d.items():
        if isinstance(v[0], torch.Tensor):
            out[k] = torch.cat([t if t.dim() > 0 else t.reshape(1) for t in v], 0)
        else:
            out[k] = v
    return out


def chunk_renderer(rays, tensorf, focal, keys=("rgb_map",), chunk=4096, render2completion=False, **kwargs):
    """renderer.py:56-106.  Rays are rendered `chunk` at a time; with render2completion every chunk is re-submitted with
    the rays the sampler's sample budget cut off (`~whole_valid`, samplers/alphagrid.py:353-364) until none is left, so the
    concatenated outputs cover every ray exactly once, in submission order per round.  Returns (images, stats): tensors of
    the requested keys concatenated over the calls (keys=None: everything the module returns)."""
    ims, stats = defaultdict(list), defaultdict(list)
    n_all = rays.shape[0]
    for start in range(0, n_all, chunk):
        rays_chunk = rays[start:start + chunk]
        if rays_chunk.numel() == 0:
            continue
        pending = rays_chunk
        while pending.shape[0] > 0:
            cims, cstats = tensorf(pending, focal, **kwargs)
            for src, dst in ((cims, ims), (cstats, stats)):
                for key in (keys if keys is not None else list(src.keys())):
                    if key in src:
                        dst[key].append(src[key])
            if not render2completion:
                break
            kept = cstats.get("rays_kept")
            if kept is None:                                   # a module without the host-side count: one read-back
                wv = cstats["whole_valid"]
                pending = pending[~wv]
            elif kept >= pending.shape[0]:
                break
            else:                                              # valid rays are a prefix (cumsum < budget)
                if kept == 0:
                    raise RuntimeError("render2completion: the sample budget admits no ray of this chunk")
                pending = pending[kept:]
    return _stack(ims), _stack(stats)


def psnr_8bit(pred, gt):
    """renderer.py:399-401: the prediction is quantised to 8 bits (floor) before the error is taken"""
    q = torch.floor(pred.clip(0, 1) * 255) / 255
    return -10.0 * torch.log10(((q - gt.clip(0, 1)) ** 2).mean())


@torch.no_grad()
def render_images(nerf, rays, focal, chunk=None, noise=None, keys=("rgb_map",), **kw):
    """evaluation render (renderer.py:119-170 without the random permutation): eval_batch_size rays per chunk, rendered
    to completion, is_train=False"""
    chunk = chunk or nerf.eval_batch_size
    kw.setdefault("draw_debug", False)
    module_kw = dict(bg_col=torch.ones(3, device=rays.device), is_train=False, ndc_ray=False, noise=noise, **kw)
    tensorf = nerf
    maps = bool(set(keys) & {"depth", "world_normal"})
    fast = _eval_pass(nerf) if (rays.is_cuda and not kw["draw_debug"] and set(keys) <= {"rgb_map", "acc_map", "depth", "world_normal"}
                                and len(kw) == 1) else None
    if fast is not None:
        from .fast_step import Unsupported

        def tensorf(pending, focal_, **_kw):          # noqa: F811 -- the straight-line forward, the module as its fallback
            try:
                nz = noise
                if nz is None:                        # the module's own generator (tensor_nerf.py: _render)
                    if nerf._noise is None:
                        from .noise import DeviceNoise
                        nerf._noise = DeviceNoise(pending.device, seed=20211200)
                    nz = nerf._noise
                out = fast.render_chunk(pending, focal_, nz, want_maps=maps)
            except Unsupported:
                return nerf(pending, focal_, **(dict(module_kw, draw_debug=True) if maps else module_kw))
            ims_ = dict(rgb_map=out[0], acc_map=out[1])
            if maps:
                ims_.update(depth=out[4], world_normal=out[5])
            return ims_, dict(rays_kept=out[2], n_samples=out[3])
    ims, _ = chunk_renderer(rays, tensorf, focal, keys=keys, chunk=chunk, render2completion=True, **module_kw)
    return ims["rgb_map"] if tuple(keys) == ("rgb_map",) else ims


def _eval_pass(nerf):
    """the C++ pass of nmf_amd/fast_step.py, forward only (nerf.fused_eval_pass = False: always the module path)"""
    if not getattr(nerf, "fused_eval_pass", True):
        return None
    fp = getattr(nerf, "_fused_pass", None)
    if fp is None:
        from .fast_step import TrainPass
        fp = TrainPass(nerf)
        if hasattr(nerf, "_fused_pass"):
            nerf._fused_pass = fp
        else:
            object.__setattr__(nerf, "_fused_pass", fp)
    return fp if fp.supported() else None
