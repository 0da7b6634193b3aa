"""Host side of the tape-free training / evaluation pass.

The pass itself -- forward of recursion level 0 and 1 keeping the intermediates, the adjoint calls in reverse order, side streams,
size read-backs -- is ONE C++ call per chunk and direction (`StepCore` in csrc/step_core.inc, built into lib/_nmf_host.so: the C-ABI
calls of include/nmf_hip.h issued from C++).  `TrainPass` owns what happens once per optimizer step around it: the derived tables
(packed density planes, stacked head / MLP weights, env-map SAT, SH irradiance) rebuilt on a side stream, the flat gradient
accumulator, and the conversion of the accumulators into parameter gradients.  It has two callers:

  * `nmf_amd.trainer.Trainer.step` -> `chunk()`: forward + loss head + backward in one call, gradients handed over by `end_step()`;
  * `TensorNeRF.forward(is_train=True)` under autograd -> `forward_autograd()`: the SAME forward, returned as the outputs of ONE
    autograd node (`ChunkPass`); the node's backward runs the SAME backward and leaves the gradients in `.grad`.  This is the path of
    the reference's own training loop (train.py:509-747: forward, torch loss, `total_loss.backward()`, `optimizer.step()`).

Scope: is_train=True (or the forward-only evaluation render), the sparse-appearance path (no debug maps), recursion depth
len(max_retrace_rays) <= 1, the C++ host extension present.  Anything else raises Unsupported BEFORE touching an accumulator and the
caller runs that chunk through the autograd operator graph of nmf_amd/functional.py (tests/test_hip_e2e.py compares the two).
Replayed bookkeeping of a reference run (noise.Pins on a ReplayNoise: bounce counts, re-trace order, occupancy decisions) is honoured
by the C++ pass, so the reference's full-size fixtures are checked on THIS path (tests/test_hip_timed_path.py).
Reference spans are the ones cited in functional.py for each call.  (Rounds 2-4 kept a Python twin of the C++ pass in this file; it was
removed in round 5 -- three orchestrations of one step were one too many.)"""
import ctypes
import os
import types
import weakref

import torch

from . import hip


class Unsupported(Exception):
    pass


MLP_SIDE_MIN_RAYS = 100000   # the capped launch runs at about half speed: it pays only next to enough other work, i.e. when the
                             # level below re-traces this many rays (steady state: all 0.24 M).  With 20 k re-traced rays
                             # under 0.1 M+ MLP rows the main stream ran dry and waited for it (-4 %), and at the deepest level
                             # the 46 k-row launch next to the env adjoint was no faster than the two one after the other
MLP_SIDE_MIN_ENV_RAYS = 200000   # ... or when this many of the level's own bounce rays go to the env map (partial re-trace)
MLP_SIDE_WGS_ENV = 96            # workgroups next to that env-map adjoint alone.  Round 3 (split-bf16 kernel): early phase, in-process
                                 # A/B 64: 1.661 ms, 96: 1.664, 128: 1.669, 256: 1.742 (round 2, fp32 kernel: 256; 128 had made the
                                 # MLP the long pole at 0.5 M rays)
WALK_SIDE_MIN_SAMPLES = 200000
MLP_SIDE_WGS = 192      # persistent workgroups of a BRDF-MLP backward that shares the chip.  R4: 192 since the partial sums of the
                        # workgroups cost one atomic per gradient element and call (k_brdf_mlp_reduce): in-process A/B 96: 1.521 ms,
                        # 128: 1.510, 160: 1.518, 192: 1.497-1.503, 224: 1.506, 256: 1.505.  Round 3 (split-bf16 kernel, one
                        # workgroup of 4 waves and 150 KB of LDS per CU): in-process A/B 32: 1.812 ms, 64: 1.628, 96: 1.608,
                        # 128: 1.625, 192: 1.678, 256: 1.696 -- the launch is short now, and every CU it occupies is a CU whose
                        # LDS the kernels of the main stream cannot use (round 2, fp32 kernel: 192 / 256 best)


# Chunk contexts: the chunks of ONE optimizer step are independent given the parameters (the reference runs them one after another,
# train.py:509-712, and sums their gradients); with two contexts -- a StepCore, a main stream and a set of side streams each -- chunk
# k + 1's forward (latency bound: four size read-backs) is queued and runs while chunk k's backward fills the chip.  A step of one
# chunk only ever uses context 0 = torch's current stream, exactly as before.  VERDICT r05 item 2; NMF_CHUNK_CONTEXTS=1 switches it off.
N_CONTEXTS = max(1, int(os.environ.get("NMF_CHUNK_CONTEXTS", "2")))


def _ns(**kw):
    return types.SimpleNamespace(**kw)


class _ChunkHolder:
    """what functional.L1Mean needs of a pass to hand its gradient over (GradPass's l1 / used / done): the density_L1 term of a chunk
    rides in the unpack launch of that chunk's ChunkPass node"""
    __slots__ = ("l1", "used", "done", "chunk_pass", "serial", "__weakref__")

    def __init__(self, serial=0):
        self.l1, self.used, self.done, self.chunk_pass = None, True, False, True
        self.serial = serial            # which training forward of the pass this node belongs to (TrainPass._pending)


class ChunkPass(torch.autograd.Function):
    """One training chunk of TensorNeRF.forward as ONE graph node: forward = StepCore.train_forward (already run by
    TrainPass.forward_autograd, which hands its outputs in), backward = StepCore.train_backward + the parameter gradients.  The
    token is a leaf that requires grad, so that the node is part of any graph built on its outputs; parameters are NOT inputs of
    the node -- their gradients are written to .grad by the pass (an input per parameter would cost autograd a clone or an add
    launch per parameter and chunk: 30 launches on a path that is latency bound)."""

    @staticmethod
    def forward(ctx, tp, holder, token, rgb, acc, ori):
        ctx.tp, ctx.holder = tp, holder
        ctx.set_materialize_grads(False)
        return rgb.detach(), acc.detach(), ori.detach()

    @staticmethod
    def backward(ctx, d_rgb, d_acc, d_ori):
        tp, holder = ctx.tp, ctx.holder
        if d_rgb is None and d_acc is None and d_ori is None:
            return None, None, None, None, None, None
        if d_rgb is None:
            d_rgb = torch.zeros((tp.last_sizes["rays"], 3), dtype=torch.float32, device=(d_acc if d_acc is not None else d_ori).device)
        tp.backward_autograd(holder, d_rgb.float().contiguous(), None if d_acc is None else d_acc.float().contiguous(),
                             None if d_ori is None else d_ori.float().contiguous())
        return None, None, None, None, None, None


class TrainPass:
    def __init__(self, nerf):
        self.nerf = nerf
        self.acc = None
        self.n_loss_chunks = 0
        # The BRDF-MLP backward of a level needs only d_brdf; it runs on a side stream next to the adjoint of that level's
        # bounce rays (level 0: the whole backward of level 1), capped to MLP_SIDE_WGS persistent workgroups so that it
        # leaves registers and LDS on every CU to the main stream -- uncapped it holds both and nothing overlaps.  The
        # kernels it runs next to are latency- or atomic-bound (docs/DESIGN_rounds_1-5.md section 5.1): 2.19 - 2.25 -> 2.12 - 2.18 ms per step.
        # The environment adjoint of a level's own rays (atomic-bound scatter, result first used when the level returns) runs
        # on another side stream next to the rest of that level's backward: 2.12 - 2.18 -> 2.07 - 2.11 ms.
        # Measured and dropped: the appearance walk next to the density walk and the level-1 density walk started early (no
        # gain), the composite backward on a side stream (2.25 ms); anything of the FORWARD on a side stream (the MLP next to the
        # level-1 sampler: 2.36 ms, the background lookup of the secondary rays next to it: 2.28 ms).  NMF_OVERLAP=0 keeps everything on one stream.
        self.overlap = os.environ.get("NMF_OVERLAP", "1") != "0"
        self._last_chunk = False
        self._acc_cache = None
        self._owner_slots = None
        self.sparse_normals = os.environ.get("NMF_SPARSE_NORMALS", "1") != "0"
        self._early_env = None
        # Side streams are PROCESS-WIDE, one per role and device: HIP maps streams onto a few hardware queues in creation
        # order (GPU_MAX_HW_QUEUES, 4 by default), and which roles end up sharing a queue decides what can overlap.  With
        # streams of its own, the second, third, ... TrainPass of a process got another role -> queue assignment than the
        # first and ran up to 0.25 ms per step slower (tools/model_order_check.py); passes of one process run one after the
        # other, so they can share the streams.
        self._side = _SIDE_STREAMS.setdefault(torch.cuda.current_device() if torch.cuda.is_available() else -1, {})
        self._main = None
        # the same pass as ONE C++ call per chunk (csrc/step_core.inc, in lib/_nmf_host.so): the methods below stay the
        # specification and the path for bf16 tables / NMF_STEP_CORE=0 / a missing host extension
        self._core = None               # context 0's StepCore (None: not created yet, False: no host extension)
        self._ctxs = []                 # chunk contexts: _ns(index, core, main (torch Stream; None = torch's current stream), ...)
        self._ctx_used = []             # contexts with chunks of the running optimizer step in flight
        self._switches = {}
        self.n_contexts = N_CONTEXTS if self.overlap else 1
        self._mlp_image = None          # hip.brdf_mlp_pack of the MLP weights, rewritten with the per-step tables
        self._token_params = None
        self._token_plist = None
        self._tables_token = None
        self._tables_cur = None         # what the last _core_tables built: bound to each context's StepCore (_bind_tables)
        self._tables_serial = 0
        self._table_events = None
        self.last_sizes = None            # sizes of the last chunk the C++ pass ran (reports)
        self._delivered = None            # [(parameter, gradient tensor, its version)] the autograd node left in .grad
        self._acc_empty = False
        self._prefetch_registered = False
        self._autograd_chunks = 0         # ChunkPass backwards since the last optimizer step
        self._fwd_serial = 0              # training forwards handed out as autograd nodes
        self._pending = None              # (serial, weakref to its holder) of the forward whose tape the C++ pass keeps

    # ------------------------------------------------------------------------------------------------------------
    def supported(self):
        n = self.nerf
        m = n.model
        return (len(m.max_retrace_rays) <= 1 and n.bg_module is not None and not n.hdr and getattr(m.brdf, "fused", False)
                and self.core() is not None)

    # ---- the C++ pass ----------------------------------------------------------------------------------------------
    _CORE_STREAMS = (("mlp", 0), ("mlp", 1), ("env", 0), ("env", 1), ("walk", 1), "sat_bwd", "env_table")

    def context(self, i=0):
        """-> chunk context i (created on first use), or None without the host extension.  Context 0 works on torch's current
        stream and the process-wide side streams; context i > 0 on a main stream and side streams of its own."""
        if self._core is False:
            return None
        while len(self._ctxs) <= i:
            fx = hip.HOST_EXT
            if fx is None or not hasattr(fx, "StepCore"):
                self._core = False
                return None
            k = len(self._ctxs)
            main = None
            if k > 0:
                main = self._side.get(("ctx", k, "main"))
                if main is None:
                    main = self._side[("ctx", k, "main")] = torch.cuda.Stream()
            self._ctxs.append(_ns(index=k, core=fx.StepCore(), main=main, key=None, keep=None, static=None, march_blocks=None,
                                  retrace=None, acc=None, bound=-1))
            for name, v in self._switches.items():
                setattr(self._ctxs[-1].core, name, v)
            if k == 0:
                self._core = self._ctxs[0].core
        return self._ctxs[i]

    def core(self, i=0):
        """-> lib/_nmf_host.so's StepCore of chunk context i, configured for this model, or None (no host extension)"""
        cx = self.context(i)
        return None if cx is None else cx.core

    def cores(self):
        return [cx.core for cx in self._ctxs]

    def set_switch(self, name, value):
        """a boolean switch of the C++ pass (env_split, ...: A/B runs and tests) on every chunk context, present and future"""
        self._switches[name] = value
        if self.context(0) is not None:
            for cx in self._ctxs:
                setattr(cx.core, name, value)

    def _param_token(self):
        """(version, storage) of every parameter the derived tables are built from: equal token = equal tables"""
        ps = self._token_params
        plist = self.nerf.rf._param_list()        # a NEW list object after an upsample / a load_state_dict that replaced the tables
        if ps is None or plist is not self._token_plist:
            n = self.nerf
            self._token_plist = plist
            ps = self._token_params = (list(plist) + list(n.model.diffuse_module._head_params())
                                       + list(n.model.brdf._weights()) + [n.bg_module.bg_mat, n.bg_module.mipbias,
                                                                          n.bg_module.brightness, n.bg_module.mul])
        return tuple([p._version for p in ps]), tuple([p.data_ptr() for p in ps])

    def _core_tables(self, dev):
        """The per-step rebuilds (packed density tables, stacked head / MLP weights, env-map SAT, SH irradiance) queued on a
        SIDE stream, with two events the C++ pass makes the main stream wait for where the results are first read (field query;
        first env-map use).  Called by prefetch() right after the optimizer step -- the device then rebuilds while the host
        prepares the next step and the sampler of the next step starts without the ~130 us of rebuild kernels in front of it --
        or by the first chunk that finds its tables stale."""
        n = self.nerf
        rf, model, bgm, smp = n.rf, n.model, n.bg_module, n.sampler
        main = torch.cuda.current_stream()
        if self.overlap:
            tb = self._side.get("tables")
            if tb is None:
                tb = self._side["tables"] = torch.cuda.Stream()
            te = tb                                   # (a stream each for the field and the env chain measured the same: R4)
            if self._table_events is None:
                self._table_events = (torch.cuda.Event(), torch.cuda.Event())
            tb.wait_stream(main)                      # the optimizer update of the parameters was queued on the main stream
        def field_tables():
            if self.overlap:
                torch.cuda.set_stream(tb)
            r = (rf._fwd_tables(), model.diffuse_module.head_pass(), model.brdf.mlp_pass())
            # the BRDF MLP's weights as the image its kernels copy into LDS (one small launch here instead of a conversion in
            # front of the first tile of each of the step's four MLP launches)
            self._mlp_image = hip.brdf_mlp_pack(r[2][0], self._mlp_image)
            if self.overlap:
                self._table_events[0].record(tb)
            return r

        def env_tables():
            if self.overlap:
                torch.cuda.set_stream(te)
            r = (bgm._tables(), bgm._dev_scalars(), bgm.get_spherical_harmonics(100)[1].reshape(9, 3))
            if self.overlap:
                self._table_events[1].record(te)
            return r

        try:
            tab, (hp, hW, hb, _, _), (mlp_ws, mlp_bias, _, _) = field_tables()
            env, sc, conv = env_tables()
        finally:
            if self.overlap:
                torch.cuda.set_stream(main)
        self._tables_cur = (tab, hp, hW, hb, mlp_ws, mlp_bias, env, sc, conv)
        self._tables_serial += 1
        self._tables_token = self._param_token()
        if not self._ctxs:
            self.context(0)
        for cx in self._ctxs:
            self._bind_tables(cx, dev)

    def _bind_tables(self, cx, dev):
        """what _core_tables built -> the attributes of chunk context `cx`'s StepCore: everything when a table moved (a new grid, a
        new stream), else the few per-step values"""
        tab, hp, hW, hb, mlp_ws, mlp_bias, env, sc, conv = self._tables_cur
        c, n = cx.core, self.nerf
        rf, model, bgm, smp = n.rf, n.model, n.bg_module, n.sampler
        main = cx.main if cx.main is not None else torch.cuda.current_stream()
        key = (ctypes.addressof(tab[0]), tab[1][0].data_ptr(), tab[3][0].data_ptr(), tab[5].data_ptr(), bgm._cache[1][3].data_ptr(),
               hW.data_ptr(), mlp_ws[0].data_ptr(), model.brdf_sampler.angs.data_ptr(), main.cuda_stream, self.overlap, id(smp))
        if cx.key != key:
            cx.key = key
            p, dpk, dlk, apl, ali, basis = rf._tables()          # fp32 masters: the backward walks
            c.vm_p, c.dpk, c.dlk, c.apl, c.ali, c.basis = ctypes.addressof(p), list(dpk), list(dlk), list(apl), list(ali), basis
            if rf.table_dtype == "f32":
                c.f_dpk, c.f_dlk, c.f_apl, c.f_ali = [], [], [], []
            else:                                                # bf16 copies for the forward queries
                c.f_dpk, c.f_dlk, c.f_apl, c.f_ali = list(tab[1]), list(tab[2]), list(tab[3]), list(tab[4])
            vt = rf._value_tables()
            c.dpl, c.dli = (list(vt[0]), list(vt[1])) if vt is not None else ([], [])
            c.head_W, c.head_b, c.mlp_ws = hW, hb, list(mlp_ws)
            c.sobol = model.brdf_sampler.angs
            c.env_table, c.env_pole, c.env_act = bgm._cache[1][3], env[2], env[0]
            c.env_bg = bgm.bg_mat.detach().reshape(3, bgm.bg_mat.shape[-2], bgm.bg_mat.shape[-1])
            c.white, c.one = _white(dev), _one(dev)
            c.select_ws = hip.select_total_workspace(dev, main)          # (one per stream: chunks of two contexts overlap)
            c.scale = float(rf.distance_scale)
            c.max_brdf_rays = [int(v) for v in model.max_brdf_rays]
            c.rays_per_ray, c.test_rays_per_ray = float(model.rays_per_ray), float(model.test_rays_per_ray)
            c.sampler = smp
            c.overlap = bool(self.overlap)
            c.main_stream, c.main_stream_obj, c.set_stream = main.cuda_stream, main, torch.cuda.set_stream
            if self.overlap:
                roles = [k if cx.index == 0 else ("ctx", cx.index, k) for k in self._CORE_STREAMS]
                for k in roles:
                    if k not in self._side:
                        self._side[k] = torch.cuda.Stream()
                objs = [self._side[k] for k in roles]
                c.side_stream_objs, c.side_streams = objs, [o.cuda_stream for o in objs]
            else:
                c.side_stream_objs, c.side_streams = [], []
            cx.keep = (tab, env, hW, hb, mlp_ws, main)
        c.head_p, c.mlp_bias = [float(v) for v in hp], float(mlp_bias)
        c.mlp_image = self._mlp_image
        c.env_sc, c.sh_conv = sc, conv
        if self.overlap:
            c.wait_tables, c.wait_env = self._table_events[0].cuda_event, self._table_events[1].cuda_event
        else:
            c.wait_tables, c.wait_env = 0, 0
        cx.bound = self._tables_serial

    @torch.no_grad()
    def prefetch(self):
        """Trainer.step calls this after the optimizer step (and the schedule): the next step's derived tables"""
        if self.core() is None or self._acc_cache is None or not self.supported():
            return
        self._core_tables(self._acc_cache.flat.device)

    def _core_sync(self, dev, focal, is_train, cx=None):
        """hands the C++ pass of chunk context `cx` what it reads: the tables (rebuilt here only when no prefetch() left them
        current), then the few per-chunk values"""
        cx = cx if cx is not None else self.context(0)
        c, n = cx.core, self.nerf
        model, smp = n.model, n.sampler
        if self._tables_token is None or self._tables_token != self._param_token():
            if cx.main is not None:                   # (the rebuild is queued behind the optimizer update: the caller's stream)
                raise hip.NmfHipError("derived tables are rebuilt from context 0 (a chunk context > 0 never runs the first chunk of a step)")
            self._core_tables(dev)
        if cx.bound != self._tables_serial:
            self._bind_tables(cx, dev)
        st = (self.sparse_normals, MLP_SIDE_MIN_RAYS, MLP_SIDE_MIN_ENV_RAYS, MLP_SIDE_WGS_ENV, WALK_SIDE_MIN_SAMPLES, MLP_SIDE_WGS,
              int(hip.ENV_BINNED_MIN_LOOKUPS), int(smp.max_samples), float(model.anoise))
        if st != cx.static:
            cx.static = st
            c.sparse_normals = bool(self.sparse_normals)
            c.mlp_side_min_rays, c.mlp_side_min_env_rays, c.mlp_side_wgs_env = MLP_SIDE_MIN_RAYS, MLP_SIDE_MIN_ENV_RAYS, MLP_SIDE_WGS_ENV
            c.walk_side_min_samples, c.mlp_side_wgs, c.env_binned_from = WALK_SIDE_MIN_SAMPLES, MLP_SIDE_WGS, int(hip.ENV_BINNED_MIN_LOOKUPS)
            c.max_samples, c.anoise = int(smp.max_samples), float(model.anoise)
        packed, blk0 = smp.params_block(focal, None, is_train)
        _, blk1 = smp.params_block(focal, 3 * float(hip.host(smp.stepsize)), is_train)
        if cx.march_blocks is None or cx.march_blocks[0] is not blk0 or cx.march_blocks[1] is not blk1:
            cx.march_blocks = (blk0, blk1, packed)
            c.march_p0, c.march_p1 = ctypes.addressof(blk0[1]), ctypes.addressof(blk1[1])
            c.alpha_bits, c.alpha_coarse = (packed[1], packed[2]) if packed is not None else (None, None)
        c.min_rough, c.detach_n = float(model.min_rough), bool(model.detach_N)
        mr = [int(v) for v in model.max_retrace_rays]
        if mr != cx.retrace:
            cx.retrace = mr
            c.max_retrace_rays = mr
        return c

    def _core_chunk(self, cx, rays, gt, focal, noise, inv_lbatch, wts, want_total, last):
        c = cx.core
        rf = self.nerf.rf
        dev = rays.device
        mods = [m for m in (rf, self.nerf.bg_module, self.nerf.model.brdf, self.nerf.model.diffuse_module) if hasattr(m, "begin_pass")]
        for m in mods:
            m.begin_pass()
        try:
            self._core_sync(dev, focal, True, cx)
            c.white = _white(dev)
            a = self._accumulators(dev)
            self._bind_accumulators(cx, a)
            c.used_env = bool(a.used_env)

            def total_of(loss, ori, acc):
                dens = list(rf.density_rf.app_plane) + list(rf.density_rf.app_line)
                l1 = hip.l1_mean_fwd([x.detach() for x in dens])
                return hip.loss_mix_fwd([loss, l1, ori, acc], wts, inv_lbatch)

            try:
                out = c.chunk(rays, gt, float(focal), noise, float(inv_lbatch), [float(w) for w in wts], bool(want_total), bool(last),
                              total_of)
            except RuntimeError as e:
                if "Unsupported" in str(e):
                    raise Unsupported(str(e)) from None
                raise
            a.used_env = bool(a.used_env) or bool(c.env_was_used())
            if c.env_table_backward_queued():
                self._early_env = (c, a.d_bg)
            if out["loss"] is None:
                return dict(loss=None, kept=out["kept"], n_samples=[0])
            pins = getattr(noise, "pins", None)
            if pins is not None and pins.trace is not None:
                pass                                  # (the C++ pass wrote rgb_map0 / acc_map0 / whole_valid0 ... itself)
            self.n_loss_chunks += 1
            self.l1_scale += float(wts[1]) * float(inv_lbatch)
            self.last_sizes = dict(rays=int(out["kept"]), n_samples=list(out["n_samples"]), n_rays=list(out["n_rays"]),
                                   n_rows=list(out["n_rows"]))
            return dict(loss=out["loss"], total=out["total"], kept=out["kept"], n_samples=list(out["n_samples"]))
        finally:
            for m in mods:
                m.end_pass()

    # ---- accumulators of one optimizer step ------------------------------------------------------------------------
    def begin_step(self):
        self.acc = None
        self._acc_empty = False
        self._early_env = None
        self.n_loss_chunks = 0
        self.l1_scale = 0.0
        self._ctx_used = []
        for cx in self._ctxs:
            cx.core.begin_step()

    def _accumulators(self, dev, zero=True):
        """The gradient state of an optimizer step.  Allocated ONCE per (grid, env size): the flat accumulator buffer, its
        views, the parameter-shaped gradient tensors and the (parameter, gradient) pairs end_step hands over -- per step only
        one zero fill remains (the host side of a step is on the critical path behind the last size read-back).
        zero=False (the autograd node): the buffers as they are -- that caller decides in its backward whether a sum continues."""
        if self.acc is not None and zero:
            return self.acc
        n = self.nerf
        G = int(n.rf.density_rf.grid_size)
        H, W = n.bg_module.hw()
        # new Parameter objects (upsample, load, a replaced module): new state.  The identity of every owner, read through
        # the modules' parameter dicts (nn.Module.__getattr__ costs 1.5 us per hop: 30 us for the 22 owners)
        mods = (n.rf, n.model, n.model.brdf, n.model.diffuse_module, n.bg_module)
        sl = self._owner_slots
        if sl is None or any(a_ is not b_ for a_, b_ in zip(sl[0], mods)):
            m, dm = n.model.brdf.mlp, n.model.diffuse_module
            lin = [m[0], m[2], m[4], dm.diffuse_mlp[0], dm.tint_mlp[0], dm.f0_mlp[0], dm.roughness_mlp[0]]
            sl = self._owner_slots = (mods, [(x._parameters, k) for x in lin for k in ("weight", "bias")]
                                      + [(n.bg_module._parameters, "bg_mat"), (n.bg_module._parameters, "mipbias")], lin)
        key = (dev, G, H, W) + tuple(id(q) for q in n.rf._param_list()) + tuple(id(d[k]) for d, k in sl[1])
        c = self._acc_cache
        if c is None or c.key != key:
            # Layout of the flat buffer.  What the optimizer reads as .grad lies in two CONTIGUOUS regions, so that the data-parallel
            # trainer sums each over the ranks in place, without packing: `late` = the field's gradients in parameter layout (the
            # unpacked density gradients, appearance tables, basis matrix) + a tail of [has-gradient flag, has-env flag, guard, pad];
            # `early` = everything no field walk writes (BRDF MLP, stacked heads, mip bias, the env-map table gradient).
            shapes = ([(G, G, 48)] * 3 + [(G, 32)] * 3                                                        # 0-5   packed density accumulators
                      + [(G, G, 16)] * 3 + [(G, 16)] * 3 + [(G, G, 24)] * 3 + [(G, 24)] * 3 + [(24, 72)] + [(4,)]      # 6-19  late region
                      + [(64, 66), (64,), (64, 64), (64,), (4, 64), (4,)] + [(11, 24), (11,)] + [(1,)] + [(3, H, W)]   # 20-29 early region
                      + [(H, W, 4), (2, 3)])                                                                  # 30-31 d_sat, d_pole
            sizes = [int(torch.Size(s_).numel()) for s_ in shapes]
            pad = [(s_ + 3) & ~3 for s_ in sizes]                      # every view 16-byte aligned
            flat = torch.zeros(sum(pad), dtype=torch.float32, device=dev)
            v, o, offs = [], 0, []
            for s_, sh, p_ in zip(sizes, shapes, pad):
                offs.append(o)
                v.append(flat[o:o + s_].view(sh))
                o += p_
            offs.append(o)
            gp = [t.view(1, G, G, 16).permute(0, 3, 1, 2) for t in v[6:9]]      # [1,16,G,G] over [G][G][16] storage (channels last)
            gl = [t.view(1, G, 1, 16).permute(0, 3, 1, 2) for t in v[9:12]]     # [1,16,G,1] over [G][16]
            c = _ns(key=key, flat=flat, g_dpk=v[0:3], g_dlk=v[3:6], gp=gp, gl=gl, g_apl=v[12:15], g_ali=v[15:18], g_basis=v[18],
                    tail=v[19], g_mlp=v[20:26], g_hW=v[26], g_hb=v[27], d_mip=v[28], d_bg=v[29], d_sat=v[30], d_pole=v[31],
                    late=flat[offs[6]:offs[20]], early=flat[offs[20]:offs[30]], used_env=False,
                    pairs=None, d_bg_view=None, l1=None, l1_dev=None, early_pairs=None)
            self._acc_cache = c
            self._delivered = None
        if zero:
            c.flat.zero_()
            c.used_env = False
            self.acc = c
        return c

    # ---- evaluation: forward only ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def render_chunk(self, rays, focal, noise, want_maps=False):
        """`TensorNeRF.forward(rays, focal, bg_col=white, is_train=False, draw_debug=False)` as one C++ call:
        -> (rgb_map [b,3], acc_map [b], b = rays the sampler kept, n_samples) + (depth [b], world_normal [b,3]) with want_maps (the two
        maps of the reference's evaluation branch that need no dense appearance pass, modules/tensor_nerf.py:480-501).  Raises
        Unsupported (configuration, no sample, no bounce row): the caller renders that chunk through the module."""
        nerf = self.nerf
        if not self.supported():
            raise Unsupported("configuration")
        mods = [m for m in (nerf.rf, nerf.bg_module, nerf.model.brdf, nerf.model.diffuse_module) if hasattr(m, "begin_pass")]
        for m in mods:
            m.begin_pass()
        try:
            core = self.core()
            self._core_sync(rays.device, focal, False)
            try:
                out = core.render(rays, float(focal), noise, bool(want_maps))
            except RuntimeError as e:
                if "Unsupported" in str(e):
                    raise Unsupported(str(e)) from None
                raise
            if out is None:
                raise Unsupported("no sample")
            return (out[0], out[1], out[2], list(out[3])) + ((out[4], out[5]) if want_maps else ())
        finally:
            for m in mods:
                m.end_pass()

    # ---- one chunk: forward, loss, backward (nmf_amd.trainer.Trainer) ------------------------------------------------------
    @torch.no_grad()
    def chunk(self, rays, gt, focal, noise, inv_lbatch, wts, want_total=False, last=False, early=None, ctx=0):
        """wts = (w_photo, w_l1, w_ori, w_acc).  Returns dict(loss 0-d tensor, kept, n_samples) -- loss None when the chunk
        had no sample (train.py:567-568 skips it).  want_total: also evaluate the chunk's total loss value (the gradients do
        not need it: every term enters linearly with a constant weight)."""
        if not self.supported():
            raise Unsupported("configuration")
        cx = self.context(ctx if ctx < self.n_contexts else 0)
        c = cx.core
        # early = (callable, raw comm stream): called from the last chunk's backward when the non-field gradients are final
        c.early_cb, c.comm_stream = (early[0], int(early[1])) if (early is not None and last) else (None, 0)
        cur = torch.cuda.current_stream()
        first_use = cx not in self._ctx_used
        if first_use:
            self._ctx_used.append(cx)
        # what ends the step in the last chunk's backward (env-map table backward, early all-reduce) waits for the other contexts too
        c.peer_streams = [(o.main.cuda_stream if o.main is not None else cur.cuda_stream) for o in self._ctx_used if o is not cx] if last else []
        if cx.main is not None:
            # the caller's stream carries what this chunk reads: the zero fill of the step's accumulators and the optimizer update before
            # it (first use), the chunk's rays when the caller gathers them per chunk (Trainer.step(fetch=...), train.py:509-512)
            cx.main.wait_stream(cur)
            rays.record_stream(cx.main)           # (made on the caller's stream, read by this context's kernels until its backward ends)
            gt.record_stream(cx.main)
            torch.cuda.set_stream(cx.main)
        try:
            return self._core_chunk(cx, rays, gt, focal, noise, inv_lbatch, wts, want_total, last)
        finally:
            c.early_cb = None
            if cx.main is not None:
                torch.cuda.set_stream(cur)

    def join_contexts(self):
        """torch's current stream waits for everything the chunk contexts of this step have queued on streams of their own"""
        cur = None
        for cx in self._ctx_used:
            if cx.main is not None:
                cur = cur or torch.cuda.current_stream()
                cur.wait_stream(cx.main)

    # ---- data parallel: the gradients that are final before the walks of the last chunk ------------------------------------------
    @torch.no_grad()
    def prepare_early(self, dev, behind_chunks=False):
        """Data parallel, in front of the early collective: everything it sums exists and is queued on the CURRENT stream.
        (1) A rank none of whose chunks reached the fused backward opens zeroed accumulators (zeros travel).  (2) The env-map table
        gradient d_bg: the last chunk's backward queues the table backward itself (StepCore: last chunk, env map used, overlap); when
        it did not -- that chunk kept no sample, left the fused pass, NMF_OVERLAP=0 -- while a chunk of the step did look the map up,
        it is run here, so that the collective sums it and end_step does not write it AFTER the sum.  -> True when anything was queued."""
        queued = False
        if behind_chunks:                   # (not from inside the last chunk's backward: there StepCore waits for its peers itself)
            self.join_contexts()
        a = self.acc
        if a is None:                       # no chunk of this step reached the fused backward: zeros travel, .grad stays None here
            a = self._accumulators(dev)
            self._acc_empty = True
            queued = True
        cores = self._cores_in_use()
        used = any(bool(c.env_was_used()) for c in cores) or bool(a.used_env)
        # (called from inside the last chunk's backward, `_early_env` is not set yet: the cores themselves know what they queued)
        if used and self._early_env is None and not any(c.env_table_backward_queued() for c in cores):
            bgm = self.nerf.bg_module
            act, _sat, _pole = bgm._tables()
            hip.sat_build_bwd(a.d_sat, bgm.bg_mat.detach(), act, a.d_pole, sc=bgm._dev_scalars(), out=a.d_bg)
            a.used_env = True
            self._early_env = ("done", a.d_bg)
            queued = True
        return queued

    def _cores_in_use(self):
        return [cx.core for cx in self._ctxs]

    def early_pairs(self, dev):
        """[(parameter, accumulator tensor)] of the BRDF MLP, the material heads and the environment map: what no field walk writes.
        The tensors are the ones end_step hands to the optimizer as .grad (views of the flat buffer, the env-map table gradient, the
        fp32 mip-bias adjoint), so a sum over the ranks written into them in place is what Adam reads.  A step whose chunks never
        looked the environment map up has a zero table gradient."""
        a = self.acc
        if a is None:                       # (prepare_early opens them in front of the collective's stream dependency)
            a = self._accumulators(dev)
            self._acc_empty = True
        n = self.nerf
        bgm = n.bg_module
        if a.early_pairs is not None:
            return a.early_pairs
        m = n.model.brdf.mlp
        pairs = list(zip((m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[4].weight, m[4].bias), a.g_mlp))
        hps = n.model.diffuse_module._head_params()
        for i, (lo, hi) in enumerate(((0, 3), (3, 6), (6, 9), (9, 11))):
            pairs += [(hps[2 * i], a.g_hW[lo:hi]), (hps[2 * i + 1], a.g_hb[lo:hi])]
        pairs += [(bgm.bg_mat, a.d_bg), (bgm.mipbias, a.d_mip)]     # (a step that never looked the env map up: the zeros of the step's fill)
        a.early_pairs = [(prm, g) for prm, g in pairs if prm.requires_grad]
        return a.early_pairs

    def _register_prefetch(self):
        """a training loop that is not the Trainer (the reference's train.py) steps its optimizer itself: behind FusedAdam.step()
        the next step's derived tables are queued on their side stream, as Trainer.step does after its own optimizer call"""
        if self._prefetch_registered:
            return
        self._prefetch_registered = True
        from . import optim
        ref = weakref.ref(self)

        def cb():
            tp = ref()
            if tp is None:
                optim.AFTER_STEP.remove(cb)
            elif tp._delivered is not None or tp._autograd_chunks:
                tp._autograd_chunks = 0
                tp.prefetch()
        optim.AFTER_STEP.append(cb)

    # ---- one chunk as ONE autograd node: what TensorNeRF.forward(is_train=True) returns to a caller that forms its own loss -----
    def forward_autograd(self, rays, focal, noise):
        """-> (rgb_map [b,3], acc [b], ori [b], out dict of StepCore.train_forward) with rgb_map / acc / ori attached to a ChunkPass
        node, or None when the chunk kept no sample.  Raises Unsupported before anything is recorded."""
        if not self.supported():
            raise Unsupported("configuration")
        c = self.core()
        dev = rays.device
        rf = self.nerf.rf
        # ONE training forward may be pending per model (the C++ pass keeps one tape).  A second forward while the first one's graph is
        # still alive and has not run its backward -- a multi-chunk `chunk_renderer(..., is_train=True)` whose loss is formed over all
        # chunks, as the reference's renderer allows -- goes through the operator graph instead of superseding the first;
        # a forward whose outputs were dropped without a backward (its holder is gone) is simply replaced.
        if self._pending is not None:
            h = self._pending[1]()
            if h is not None and not h.done and c.has_pending():
                raise Unsupported("another training forward of this model is pending (one tape per model)")
            self._pending = None
        if hasattr(rf, "flush_pending_l1"):
            rf.flush_pending_l1()               # a density_L1 term whose pass never ran its backward
        mods = [m for m in (rf, self.nerf.bg_module, self.nerf.model.brdf, self.nerf.model.diffuse_module) if hasattr(m, "begin_pass")]
        for m in mods:
            m.begin_pass()
        try:
            self._core_sync(dev, focal, True)
            self._bind_accumulators(self.context(0), self._accumulators(dev, zero=False))
            self._fwd_serial += 1
            holder = _ChunkHolder(self._fwd_serial)
            try:
                out = c.train_forward(rays.detach(), float(focal), noise)
            except RuntimeError as e:
                if "Unsupported" in str(e):
                    raise Unsupported(str(e)) from None
                raise
            if list(out["n_samples"]) == [0]:
                return None
            self.last_sizes = dict(rays=int(out["kept"]), n_samples=list(out["n_samples"]), n_rays=list(out["n_rays"]),
                                   n_rows=list(out["n_rows"]))
            rgb, acc, ori = ChunkPass.apply(self, holder, _token(dev), out["rgb_map"], out["acc"], out["ori"])
            self._pending = (holder.serial, weakref.ref(holder))
            rf._last_holder = holder            # density_L1() of this chunk rides on the node (fields/tensoRF.py)
            return rgb, acc, ori, out
        finally:
            for m in mods:
                m.end_pass()

    def _bind_accumulators(self, cx, a):
        c = cx.core
        if cx.acc is not a:
            cx.acc = a
            c.g_dpk, c.g_dlk, c.g_apl, c.g_ali, c.g_mlp = list(a.g_dpk), list(a.g_dlk), list(a.g_apl), list(a.g_ali), list(a.g_mlp)
            c.g_basis, c.g_hW, c.g_hb, c.d_sat, c.d_pole, c.d_mip = a.g_basis, a.g_hW, a.g_hb, a.d_sat, a.d_pole, a.d_mip
            c.d_bg_out = a.d_bg if self.overlap else None

    @torch.no_grad()
    def backward_autograd(self, holder, d_rgb, d_acc, d_ori):
        """the backward of the pending chunk + its parameter gradients.  Gradients ACCUMULATE like autograd's: a parameter whose
        .grad is None (zero_grad(set_to_none=True), train.py:498) starts a new sum; one whose .grad is still the tensor this pass
        left there continues the sum in the accumulators (no copy, no add launch per parameter); anything else -- somebody wrote
        into .grad in between -- is added to out of place."""
        c = self.core()
        if not c.has_pending() or self._pending is None or self._pending[0] != holder.serial:
            raise RuntimeError("backward of a TensorNeRF training forward whose chunk was superseded by a later forward (one "
                               "training forward may be pending per model)")
        self._pending = None
        a = self._acc_cache
        pairs = self._delivered
        state = "fresh"
        if pairs is not None:
            mine = [prm.grad is g and g._version == v for prm, g, v in pairs]
            none = [prm.grad is None for prm, g, v in pairs]
            if all(mine):
                state = "continue"
            elif not all(none):
                state = "detach"
        elif any(prm.grad is not None for prm in self._grad_params()):
            state = "detach"
        if state == "detach":
            # what this pass had summed so far becomes an ordinary gradient tensor: every .grad that still ALIASES the flat buffer (by
            # address -- a delivered tensor somebody modified in place fails the identity / version test above and is still a view of it)
            lo = a.flat.data_ptr()
            hi = lo + 4 * a.flat.numel()
            for prm in self._grad_params():
                g = prm.grad
                if g is not None and lo <= g.data_ptr() < hi:
                    prm.grad = g.clone()
        if state != "continue":
            a.flat.zero_()
            a.used_env = False
            a.l1_dev = None
            c.begin_step()
        self.acc = a
        c.used_env = bool(a.used_env)
        if holder.l1 is not None:               # the chunk's density_L1 term: scale of mean|x| (fields/tensoRF.py:332-340)
            d_l1 = holder.l1[1].reshape(()).float()
            a.l1_dev = d_l1 if a.l1_dev is None else a.l1_dev + d_l1
            holder.l1 = None
        holder.done = True
        self._autograd_chunks += 1
        self._register_prefetch()
        c.env_keep_sat = True                   # (more chunks of this step may follow: the env-map adjoint table stays a sum)
        try:
            c.train_backward(d_rgb, d_acc, d_ori, True)
        finally:
            c.env_keep_sat = False
        a.used_env = bool(c.env_was_used())
        if c.env_table_backward_queued():
            self._early_env = (c, a.d_bg)
        grads = self._finish_grads(a, (None if a.l1_dev is None else a.l1_dev), keep_sat=True)
        if state == "detach":
            for prm, g in grads:
                prm.grad = g.clone() if prm.grad is None else prm.grad + g
            self._delivered = None
        else:
            for prm, g in grads:
                prm.grad = g
            self._delivered = [(prm, g, g._version) for prm, g in grads]
        self.acc = None

    def _grad_params(self):
        n = self.nerf
        return (list(n.rf._param_list()) + list(n.model.brdf._weights()) + list(n.model.diffuse_module._head_params())
                + [n.bg_module.bg_mat, n.bg_module.mipbias])

    # ---- accumulators -> parameter gradients ----------------------------------------------------------------------------
    @torch.no_grad()
    def _finish_grads(self, a, l1_scale, keep_sat=False):
        """accumulators -> [(parameter, gradient tensor)]: the density gradients unpacked from the packed value + derivative layout
        (with the density_L1 term of scale `l1_scale`, a 0-d device tensor or None, added in the same launch), views of the flat
        buffer for everything else, the env-map table backward (joined if it was queued on its side stream)"""
        nerf = self.nerf
        rf, model, bgm = nerf.rf, nerf.model, nerf.bg_module
        early_env = self._early_env
        self._early_env = None
        if early_env is not None and early_env[0] != "done":       # queued by a StepCore on its side stream: its main stream waits for it,
            early_env[0].join_early_env()
        self.join_contexts()                                        # ... and this stream for every chunk context's main stream
        p = rf._tables()[0]
        l1 = None
        if l1_scale is not None:          # the L1 term's gradient rides in the unpack launch (one launch less on the step's tail)
            l1 = ([x.detach() for x in list(rf.density_rf.app_plane) + list(rf.density_rf.app_line)], l1_scale)
        hip.vm_unpack_density_grad(p, a.g_dpk, a.g_dlk, out=(a.gp, a.gl), l1=l1)
        gp, gl = a.gp, a.gl
        if a.pairs is None:       # (parameter, gradient tensor) of everything that lives in the persistent buffers
            pairs = list(zip(rf._param_list(), rf._grads_to_param_layout(gp, gl, a.g_apl, a.g_ali, a.g_basis)))
            m = model.brdf.mlp
            pairs += list(zip((m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[4].weight, m[4].bias), a.g_mlp))
            hps = model.diffuse_module._head_params()
            for i, (lo, hi) in enumerate(((0, 3), (3, 6), (6, 9), (9, 11))):
                pairs += [(hps[2 * i], a.g_hW[lo:hi]), (hps[2 * i + 1], a.g_hb[lo:hi])]
            a.pairs = [(prm, g) for prm, g in pairs if prm.requires_grad]
        grads = list(a.pairs)
        if a.used_env:
            act, sat, pole = bgm._tables()
            sc = bgm._dev_scalars()
            if early_env is not None:
                d_bg = early_env[1]
            else:
                d_bg = a.d_bg = hip.sat_build_bwd(a.d_sat.clone() if keep_sat else a.d_sat, bgm.bg_mat.detach(), act, a.d_pole, sc=sc,
                                                  out=a.d_bg)
            if bgm.bg_mat.requires_grad:
                if a.d_bg_view is None or a.d_bg_view[0] is not d_bg:
                    a.d_bg_view = (d_bg, d_bg.reshape(bgm.bg_mat.shape))
                grads.append((bgm.bg_mat, a.d_bg_view[1]))
            if bgm.brightness_lr != 0 or bgm.mul_lr != 0:           # lr 0 (microfacet_tensorf2.yaml:150-151): no update anyway
                d_pre = d_bg / sc[2]
                grads.append((bgm.brightness, d_pre.sum(dtype=torch.float64)))
                grads.append((bgm.mul, (d_pre * bgm.bg_mat.detach().reshape(d_pre.shape)).sum(dtype=torch.float64)))
            if bgm.mipbias.requires_grad:
                grads.append((bgm.mipbias, a.d_mip.to(torch.float64).reshape(())))
        return [(prm, g) for prm, g in grads if prm.requires_grad]

    @torch.no_grad()
    def end_step(self):
        """Trainer.step, after the last chunk: the accumulators of the step become (or are added to) the parameters' .grad"""
        a = self.acc
        if a is None:
            return
        if self._acc_empty:                 # accumulators that only exist to enter the ranks' early collective with zeros
            self.acc, self._acc_empty = None, False
            return
        l1 = None
        if self.l1_scale != 0.0:
            if a.l1 is None or a.l1[0] != self.l1_scale:
                a.l1 = (self.l1_scale, torch.full((), self.l1_scale, dtype=torch.float32, device=a.flat.device))
            l1 = a.l1[1]
        for prm, g in self._finish_grads(a, l1):
            if prm.grad is not None and prm.grad is not g:      # a chunk of this step went through the operator graph: folded INTO the
                g.add_(prm.grad.reshape(g.shape).to(g.dtype))   # pass's tensor (the data-parallel trainer sums these tensors in place)
            prm.grad = g
        self._delivered = None
        self.acc = None

    def comm_regions(self):
        """(early, late, tail) of the accumulators of the step that end_step has just closed (or that early_pairs opened): the two
        contiguous slices of the flat buffer the data-parallel trainer sums over the ranks in place, and the late slice's tail
        [has-gradient flag, has-env flag, guard, pad]"""
        a = self._acc_cache
        return a.early, a.late, a.tail

    def fold_foreign(self):
        """a gradient a chunk left in .grad OUTSIDE the accumulators (a chunk that went through the operator graph) on a rank whose
        fused chunks produced none: moved into the accumulator tensor, which is what the ranks sum"""
        a = self._acc_cache
        self.assign_reduced(False, False)          # (builds the pairs)
        bgm = self.nerf.bg_module
        for prm, g in a.pairs + [(bgm.bg_mat, a.d_bg)]:
            if prm.grad is not None and prm.grad.data_ptr() != g.data_ptr():
                g.add_(prm.grad.reshape(g.shape).to(g.dtype))
                prm.grad = g.reshape(prm.shape) if g.shape != prm.shape else g

    def assign_reduced(self, field, env):
        """a rank that had no gradient of its own (no chunk with a sample / no env-map lookup) takes the ranks' sums: .grad of the
        field + shading parameters (`field`) / of the env map (`env`) from the accumulator tensors the collectives summed into"""
        a = self._acc_cache
        if a.pairs is None:
            n = self.nerf
            rf, model = n.rf, n.model
            pairs = list(zip(rf._param_list(), rf._grads_to_param_layout(a.gp, a.gl, a.g_apl, a.g_ali, a.g_basis)))
            m = model.brdf.mlp
            pairs += list(zip((m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[4].weight, m[4].bias), a.g_mlp))
            hps = model.diffuse_module._head_params()
            for i, (lo, hi) in enumerate(((0, 3), (3, 6), (6, 9), (9, 11))):
                pairs += [(hps[2 * i], a.g_hW[lo:hi]), (hps[2 * i + 1], a.g_hb[lo:hi])]
            a.pairs = [(prm, g) for prm, g in pairs if prm.requires_grad]
        if field:
            for prm, g in a.pairs:
                prm.grad = g
        if env:
            bgm = self.nerf.bg_module
            if bgm.bg_mat.requires_grad:
                bgm.bg_mat.grad = a.d_bg.reshape(bgm.bg_mat.shape)
            if bgm.mipbias.requires_grad:
                bgm.mipbias.grad = a.d_mip.to(torch.float64).reshape(())


_CONST = {}
_SIDE_STREAMS = {}          # device index -> {role: torch.cuda.Stream}, shared by every TrainPass of the process


def _white(dev):
    k = ("white", dev)
    if k not in _CONST:
        _CONST[k] = torch.ones((1, 3), dtype=torch.float32, device=dev)
    return _CONST[k]


def _zeros(like):
    """a read-only zero tensor of `like`'s shape (a view of one buffer that only grows: no fill per step)"""
    k = ("zeros", like.device)
    n = like.numel()
    if k not in _CONST or _CONST[k].numel() < n:
        _CONST[k] = torch.zeros(max(n, 1 << 16) * 5 // 4, dtype=torch.float32, device=like.device)
    return _CONST[k][:n].view(like.shape)


def _one(dev):
    k = ("one", dev)
    if k not in _CONST:
        _CONST[k] = torch.ones((), dtype=torch.float32, device=dev)
    return _CONST[k]


def _token(dev):
    """the leaf every ChunkPass node hangs on (never receives a gradient)"""
    k = ("token", dev)
    if k not in _CONST:
        _CONST[k] = torch.zeros((), dtype=torch.float32, device=dev, requires_grad=True)
    return _CONST[k]
