"""The training pass of `TensorNeRF.forward` + `backward()` without the autograd engine.

`modules/tensor_nerf.py` / `models/microfacet.py` / `functional.py` express the hot path as ~12 autograd nodes per direction
around the C-ABI kernels; that keeps the reference's operator API differentiable for any caller, but in the training loop
the graph is the same every step and the engine's bookkeeping (node construction, the backward dispatch through Python,
pass tokens) is host time on a path where host time and kernel time are the same ~3 ms.  TrainPass runs the SAME kernel
sequence as straight-line code: forward for recursion level 0 and 1 keeping the intermediates on a tape, then the adjoint
calls in reverse order, with the table / MLP / head / env-map gradients accumulated in persistent buffers over all chunks of
an optimizer step and converted to parameter gradients once (`end_step`).

Scope: is_train=True, the sparse-appearance path (no debug maps), recursion depth len(max_retrace_rays) <= 1.  Anything
else raises Unsupported BEFORE touching an accumulator and the Trainer runs that chunk through the autograd path instead
(tests/test_hip_e2e.py compares the two paths).  Replayed bookkeeping of a reference run (noise.Pins on a ReplayNoise: bounce
counts, re-trace order, occupancy decisions) is honoured here exactly as in the modules, so the reference's full-size fixtures
are checked on THIS path (tests/test_hip_timed_path.py).
Reference spans are the ones cited in functional.py for each call."""
import ctypes
import os
import types

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
LAUNCH_DIET = 1         # 0: the separate loss / background-adjoint / head-adjoint launches of round 2 (A/B knob)
WALK_LATE = 0           # 1: that walk is queued after the levels below instead of before them (A/B knob)
MLP_SIDE_WGS = 192      # persistent workgroups of a BRDF-MLP backward that shares the chip.  R4: 192 since the partial sums of the
                        # workgroups cost one atomic per gradient element and call (k_brdf_mlp_reduce): in-process A/B 96: 1.521 ms,
                        # 128: 1.510, 160: 1.518, 192: 1.497-1.503, 224: 1.506, 256: 1.505.  Round 3 (split-bf16 kernel, one
                        # workgroup of 4 waves and 150 KB of LDS per CU): in-process A/B 32: 1.812 ms, 64: 1.628, 96: 1.608,
                        # 128: 1.625, 192: 1.678, 256: 1.696 -- the launch is short now, and every CU it occupies is a CU whose
                        # LDS the kernels of the main stream cannot use (round 2, fp32 kernel: 192 / 256 best)


def _ns(**kw):
    return types.SimpleNamespace(**kw)


class _on:
    """`with _on((main, side)):` -- the side stream is torch's current stream inside the block.  set_stream on entry and exit
    (0.4 us each) instead of torch.cuda.stream()'s context manager (6 us): four of these sit on the host-critical backward."""
    __slots__ = ("fork",)

    def __init__(self, fork):
        self.fork = fork

    def __enter__(self):
        torch.cuda.set_stream(self.fork[1])

    def __exit__(self, *exc):
        torch.cuda.set_stream(self.fork[0])
        return False


class TrainPass:
    def __init__(self, nerf):
        self.nerf = nerf
        self.acc = None
        self.n_loss_chunks = 0
        # The BRDF-MLP backward of a level needs only d_brdf; it runs on a side stream next to the adjoint of that level's
        # bounce rays (level 0: the whole backward of level 1), capped to MLP_SIDE_WGS persistent workgroups so that it
        # leaves registers and LDS on every CU to the main stream -- uncapped it holds both and nothing overlaps.  The
        # kernels it runs next to are latency- or atomic-bound: 2.19 - 2.25 -> 2.12 - 2.18 ms per step.
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
        self._core = None if os.environ.get("NMF_STEP_CORE", "1") != "0" else False
        self._core_key = None
        self._core_keep = None
        self._mlp_image = None          # hip.brdf_mlp_pack of the MLP weights, rewritten with the per-step tables
        self._core_acc = None
        self._march_blocks = None
        self._token_params = None
        self._token_plist = None
        self._tables_token = None
        self._table_events = None
        self._core_static = None
        self._core_retrace = None
        self.last_sizes = None            # sizes of the last chunk the C++ pass ran (reports)

    # ------------------------------------------------------------------------------------------------------------
    def supported(self):
        n = self.nerf
        m = n.model
        return (len(m.max_retrace_rays) <= 1 and n.bg_module is not None and not n.hdr and getattr(m.brdf, "fused", False))

    # ---- the C++ pass ----------------------------------------------------------------------------------------------
    _CORE_STREAMS = (("mlp", 0), ("mlp", 1), ("env", 0), ("env", 1), ("walk", 1), "sat_bwd")

    def core(self):
        """-> lib/_nmf_host.so's StepCore configured for this model, or None (Python pass)"""
        if self._core is False:
            return None
        if self._core is None:
            fx = hip.HOST_EXT
            if fx is None or not hasattr(fx, "StepCore"):
                self._core = False
                return None
            self._core = fx.StepCore()
        return self._core

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
        c, n = self._core, self.nerf
        rf, model, bgm, smp = n.rf, n.model, n.bg_module, n.sampler
        main = torch.cuda.current_stream()
        # NMF_TABLES_2STREAMS=1 (R4 experiment): the two rebuild chains on a stream each, the env chain (SAT columns -> rows -> poles
        # -> 5000 lookups -> SH projection, ~90 us, read by level 0's row preparation) first.  Measured: 1.542 -> 1.540 ms, nothing;
        # one stream stays the default (a ninth stream has to share one of the eight hardware queues).
        two = self.overlap and os.environ.get("NMF_TABLES_2STREAMS", "0") == "1"
        if self.overlap:
            tb = self._side.get("tables")
            if tb is None:
                tb = self._side["tables"] = torch.cuda.Stream()
            te = tb
            if two:
                te = self._side.get("env_tables")
                if te is None:
                    te = self._side["env_tables"] = torch.cuda.Stream()
            if self._table_events is None:
                self._table_events = (torch.cuda.Event(), torch.cuda.Event())
            tb.wait_stream(main)                      # the optimizer update of the parameters was queued on the main stream
            if te is not tb:
                te.wait_stream(main)
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
            if two:                                   # the env chain is the longer one: issued first
                env, sc, conv = env_tables()
                tab, (hp, hW, hb, _, _), (mlp_ws, mlp_bias, _, _) = field_tables()
            else:
                tab, (hp, hW, hb, _, _), (mlp_ws, mlp_bias, _, _) = field_tables()
                env, sc, conv = env_tables()
        finally:
            if self.overlap:
                torch.cuda.set_stream(main)
        key = (ctypes.addressof(tab[0]), tab[1][0].data_ptr(), tab[3][0].data_ptr(), tab[5].data_ptr(), bgm._cache[1][3].data_ptr(),
               hW.data_ptr(), mlp_ws[0].data_ptr(), model.brdf_sampler.angs.data_ptr(), main.cuda_stream, self.overlap, id(smp))
        if self._core_key != key:
            self._core_key = key
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
            c.select_ws = hip.select_total_workspace(dev)
            c.scale = float(rf.distance_scale)
            c.max_brdf_rays = [int(v) for v in model.max_brdf_rays]
            c.rays_per_ray, c.test_rays_per_ray = float(model.rays_per_ray), float(model.test_rays_per_ray)
            c.sampler = smp
            c.overlap = bool(self.overlap)
            c.main_stream, c.main_stream_obj, c.set_stream = main.cuda_stream, main, torch.cuda.set_stream
            if self.overlap:
                for k in self._CORE_STREAMS:
                    if k not in self._side:
                        self._side[k] = torch.cuda.Stream()
                objs = [self._side[k] for k in self._CORE_STREAMS]
                c.side_stream_objs, c.side_streams = objs, [o.cuda_stream for o in objs]
            else:
                c.side_stream_objs, c.side_streams = [], []
            self._core_keep = (tab, env, hW, hb, mlp_ws, main)
        c.head_p, c.mlp_bias = [float(v) for v in hp], float(mlp_bias)
        c.mlp_image = self._mlp_image
        c.env_sc, c.sh_conv = sc, conv
        if self.overlap:
            c.wait_tables, c.wait_env = self._table_events[0].cuda_event, self._table_events[1].cuda_event
        else:
            c.wait_tables, c.wait_env = 0, 0
        self._tables_token = self._param_token()

    @torch.no_grad()
    def prefetch(self):
        """Trainer.step calls this after the optimizer step (and the schedule): the next step's derived tables"""
        if self.core() is None or self._acc_cache is None or not self.supported():
            return
        self._core_tables(self._acc_cache.flat.device)

    def _core_sync(self, dev, focal, is_train):
        """hands the C++ pass what it reads: the tables (rebuilt here only when no prefetch() left them current), then the few
        per-chunk values"""
        c, n = self._core, self.nerf
        model, smp = n.model, n.sampler
        if self._tables_token is None or self._tables_token != self._param_token():
            self._core_tables(dev)
        st = (self.sparse_normals, WALK_LATE, LAUNCH_DIET, MLP_SIDE_MIN_RAYS, MLP_SIDE_MIN_ENV_RAYS, MLP_SIDE_WGS_ENV, WALK_SIDE_MIN_SAMPLES, MLP_SIDE_WGS,
              int(hip.ENV_BINNED_MIN_LOOKUPS), int(smp.max_samples), float(model.anoise))
        if st != self._core_static:
            self._core_static = st
            c.sparse_normals = bool(self.sparse_normals)
            c.mlp_side_min_rays, c.mlp_side_min_env_rays, c.mlp_side_wgs_env = MLP_SIDE_MIN_RAYS, MLP_SIDE_MIN_ENV_RAYS, MLP_SIDE_WGS_ENV
            c.walk_side_min_samples, c.mlp_side_wgs, c.env_binned_from = WALK_SIDE_MIN_SAMPLES, MLP_SIDE_WGS, int(hip.ENV_BINNED_MIN_LOOKUPS)
            c.max_samples, c.anoise = int(smp.max_samples), float(model.anoise)
            c.walk_late, c.launch_diet = int(WALK_LATE), int(LAUNCH_DIET)
        packed, blk0 = smp.params_block(focal, None, is_train)
        _, blk1 = smp.params_block(focal, 3 * float(hip.host(smp.stepsize)), is_train)
        if self._march_blocks is None or self._march_blocks[0] is not blk0 or self._march_blocks[1] is not blk1:
            self._march_blocks = (blk0, blk1, packed)
            c.march_p0, c.march_p1 = ctypes.addressof(blk0[1]), ctypes.addressof(blk1[1])
            c.alpha_bits, c.alpha_coarse = (packed[1], packed[2]) if packed is not None else (None, None)
        c.min_rough, c.detach_n = float(model.min_rough), bool(model.detach_N)
        mr = [int(v) for v in model.max_retrace_rays]
        if mr != self._core_retrace:
            self._core_retrace = mr
            c.max_retrace_rays = mr
        return c

    def _core_chunk(self, c, rays, gt, focal, noise, inv_lbatch, wts, want_total, last, next_rays=None):
        rf = self.nerf.rf
        dev = rays.device
        mods = [m for m in (rf, self.nerf.bg_module, self.nerf.model.brdf, self.nerf.model.diffuse_module) if hasattr(m, "begin_pass")]
        for m in mods:
            m.begin_pass()
        try:
            self._core_sync(dev, focal, True)
            a = self._accumulators(dev)
            if self._core_acc is not a:
                self._core_acc = a
                c.g_dpk, c.g_dlk, c.g_apl, c.g_ali, c.g_mlp = list(a.g_dpk), list(a.g_dlk), list(a.g_apl), list(a.g_ali), list(a.g_mlp)
                c.g_basis, c.g_hW, c.g_hb, c.d_sat, c.d_pole, c.d_mip = a.g_basis, a.g_hW, a.g_hb, a.d_sat, a.d_pole, a.d_mip
                bgm = self.nerf.bg_module
                if a.d_bg is None:
                    a.d_bg = torch.empty_like(bgm.bg_mat.detach().reshape(3, bgm.bg_mat.shape[-2], bgm.bg_mat.shape[-1]))
                c.d_bg_out = a.d_bg if self.overlap else None
            c.used_env = bool(a.used_env)

            def total_of(loss, ori, acc):
                dens = list(rf.density_rf.app_plane) + list(rf.density_rf.app_line)
                l1 = hip.l1_mean_fwd([x.detach() for x in dens])
                return hip.loss_mix_fwd([loss, l1, ori, acc], wts, inv_lbatch)

            c.next_rays = next_rays if (last and next_rays is not None and next_rays.is_contiguous()) else None
            try:
                out = c.chunk(rays, gt, float(focal), noise, float(inv_lbatch), [float(w) for w in wts], bool(want_total), bool(last),
                              total_of)
            except RuntimeError as e:
                if "Unsupported" in str(e):
                    raise Unsupported(str(e)) from None
                raise
            a.used_env = bool(c.env_was_used())
            if c.env_table_backward_queued():
                self._early_env = ("core", a.d_bg)
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
        self._early_env = None
        self.n_loss_chunks = 0
        self.l1_scale = 0.0
        if self._core:
            self._core.begin_step()

    def _accumulators(self, dev):
        """The gradient state of an optimizer step.  Allocated ONCE per (grid, env size): the flat accumulator buffer, its
        views, the parameter-shaped gradient tensors and the (parameter, gradient) pairs end_step hands over -- per step only
        one zero fill remains (the host side of a step is on the critical path behind the last size read-back)."""
        if self.acc is None:
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
                shapes = ([(G, G, 48)] * 3 + [(G, 32)] * 3 + [(G, G, 24)] * 3 + [(G, 24)] * 3 + [(24, 72)]      # field (packed)
                          + [(64, 66), (64,), (64, 64), (64,), (4, 64), (4,)]                                    # BRDF MLP
                          + [(11, 24), (11,)]                                                                     # stacked heads
                          + [(H, W, 4), (2, 3), (1,)])                                                            # d_sat, d_pole, d_mip
                sizes = [int(torch.Size(s).numel()) for s in shapes]
                pad = [(s + 3) & ~3 for s in sizes]                      # every view 16-byte aligned
                flat = torch.empty(sum(pad), dtype=torch.float32, device=dev)
                v, o = [], 0
                for s, sh, p_ in zip(sizes, shapes, pad):
                    v.append(flat[o:o + s].view(sh))
                    o += p_
                c = _ns(key=key, flat=flat, g_dpk=v[0:3], g_dlk=v[3:6], g_apl=v[6:9], g_ali=v[9:12], g_basis=v[12],
                        g_mlp=v[13:19], g_hW=v[19], g_hb=v[20], d_sat=v[21], d_pole=v[22], d_mip=v[23], used_env=False,
                        pairs=None, gp=None, gl=None, d_bg=None, l1=None)
                self._acc_cache = c
            c.flat.zero_()
            c.used_env = False
            self.acc = c
        return self.acc

    # ---- forward of one recursion level ------------------------------------------------------------------------------
    def _env_fwd(self, rows, sa):
        bgm = self.nerf.bg_module
        act, sat, pole = bgm._tables()
        return hip.sat_lookup_fwd(bgm._lookup_table(), rows, sa, 0.0, pole, sc=bgm._dev_scalars())

    def _env_bwd(self, rows, sa, d_out):
        bgm = self.nerf.bg_module
        act, sat, pole = bgm._tables()
        a = self.acc
        a.used_env = True
        return hip.sat_lookup_bwd(bgm._lookup_table(), rows, sa, 0.0, d_out, a.d_sat, a.d_pole, a.d_mip, want_dirs=True,
                                  sc=bgm._dev_scalars())

    def _fwd(self, lvl, rays, focal, start_mip, noise, is_train=True, filler=None):
        """filler: work that does not depend on this level's samples; it is queued between the sampler's counting pass and
        its size read-back so that the device runs it while the host waits for the two numbers (level 0: the per-step table
        rebuilds, level 1: the BRDF MLP of the level above).  The second read-back of a level (bounce rows) is covered the
        same way by the env-map rebuild + SH projection (level 0) and the background lookup of the level's rays (level 1)."""
        nerf = self.nerf
        rf, model, smp = nerf.rf, nerf.model, nerf.sampler
        pending = smp.sample_begin(rays, focal, override_near=None if lvl == 0 else self.near1, is_train=is_train,
                                   dynamic_batch_size=(lvl == 0), noise=noise)
        if filler is not None:
            filler()
        S = smp.sample_finish(pending)
        B, M = S.b, S.M
        t = _ns(lvl=lvl, S=S, B=B, M=M, n_samples=[M])
        if M == 0:
            return t
        offsets = S.offsets[: B + 1]
        p, dpk, dlk, apl, ali, basis = rf._fwd_tables()
        # Sparse normals: below the first level a normal is only needed where secondary rays start (the orientation term is a
        # level-0 statistic, tensor_nerf.py:583-587), so the re-traced samples get the density VALUE alone (a third of the
        # table bytes and of the products) and the bounce rows are queried for value + gradient + appearance afterwards;
        # the backward mirrors it: a value-only walk over all samples, the normal adjoint walked with the rows.
        sparse_n = lvl > 0 and self.sparse_normals
        vt = rf._value_tables() if sparse_n else None
        if vt is not None:          # the density factors themselves: a third of the cache lines of the packed tables
            (sf, sg), gr, nr = hip.vm_query_sigma(p, S.xyzt, vt[0], vt[1]), None, None
        else:
            sf, sg, gr, nr, _, _ = hip.vm_query_fwd(p, S.xyzt, dpk, dlk, apl, ali, basis, want_density=True,
                                                    want_normal=not sparse_n, want_app=False, want_coef=False)
        w, _acc = hip.composite_fwd(sg, S.dist, offsets, B, self.scale)
        # ---- Microfacet.shade_compact, sparse appearance (same draw order as the autograd path)
        deferred = noise.normal_deferred((M, 24))
        noise.skip("randn", (M, 3))
        noise.skip("randn", (M, 2))
        noise.skip("rand", (5000,))
        noise.skip("rand", (5000,))
        if lvl == 0:
            counts = hip.select_bounces(w, noise.uniform((M,)).contiguous(), 0,
                                        float(model.rays_per_ray if is_train else model.test_rays_per_ray))
        else:
            if hasattr(noise, "select_dense_parts"):        # device noise: the normaliser in one launch (nmf_select_total)
                u, extra = noise.select_dense_parts(S.b, S.N, M)
                total = hip.select_total(w, u.contiguous(), extra)
            else:
                u, u_total = noise.select_dense(S.b, S.N, S.ray_id, S.step_id)
                total = (w.sum(dtype=torch.float64) + 1e-3 * u_total).float().clip(min=1e-3)
            nb = model.max_brdf_rays[lvl] - M
            if nb > 0:
                counts = hip.select_bounces(w, u.contiguous(), 1, float(nb), 1.0, total)
            else:
                counts = hip.select_bounces(w, u.contiguous(), 1, float(model.max_brdf_rays[lvl]), 0.5, total)
        pins = getattr(noise, "pins", None)          # tests: replayed bookkeeping of a reference run (noise.Pins)
        trace = pins.trace if pins is not None else None
        if pins is not None:
            counts = pins.counts_for(lvl, counts)
        bidx, row_off, cnt32, inv, tot, xyz_rows = hip.bounce_index(counts, S.xyzt)
        rb = hip.Readback.of(tot.device).start(tot)
        conv = nerf.bg_module.get_spherical_harmonics(100)[1].reshape(9, 3)      # first call of a pass: SAT + SH rebuild
        per_ray_bg = lvl > 0
        if per_ray_bg:
            t.rough = start_mip[:B].contiguous()
            bg = self._env_fwd(S.rays if B == S.rays.shape[0] else S.rays[:B], t.rough)
        else:
            bg = self.white
        R, Mb = rb.get()
        if R == 0:
            raise Unsupported("no bounce rows")
        bidx, row_off, cnt32 = bidx[:Mb], row_off[: Mb + 1], cnt32[:Mb]
        row_of_ray, j_of_ray = hip.expand_segments(row_off, Mb, R)
        off = noise.uniform((Mb, 1, 2)).reshape(Mb, 2).contiguous()
        xyz_rows = xyz_rows[:Mb]
        feat_noise = noise.rows(deferred, bidx)
        sf_rows = gr_rows = None
        if sparse_n:
            sf_rows, gr_rows, nr = hip.vm_query_rows(p, xyz_rows, dpk, dlk)
        app = hip.vm_query_fwd(p, xyz_rows, dpk, dlk, apl, ali, basis, want_density=False, want_normal=False, want_app=True)[4]
        hp, hW, hb = self.heads
        heads = hip.heads_fwd(app, hW, hb, hp)
        V, N, r1, f0, diff, feat, xyz = hip.bounce_prep_fwd(bidx, nr, app, heads, S.xyzt, S.ray_id, S.rays, conv, feat_noise,
                                                           self.anoise, self.min_rough if is_train else -1e30,
                                                           2 if sparse_n else 1)
        sobol = model.brdf_sampler.angs
        L, hl, dl, lpdf, mip, brays = hip.ggx_rays_fwd(V, N, r1, xyz, off, cnt32, sobol, row_of_ray, j_of_ray)
        sorts = pins is not None and pins.sorts(lvl)
        full_retrace = lvl < len(model.max_retrace_rays) and min(R, model.max_retrace_rays[lvl]) >= R and not sorts
        brdf = brdf_mask = None
        if not full_retrace:
            brdf, brdf_mask = hip.brdf_mlp_fwd(self.mlp_ws, hl, dl, feat, r1, row_of_ray, self.mlp_bias, with_mask=True)
        t.__dict__.update(offsets=offsets, sf=sf, sg=sg, gr=gr, nr=nr, w=w, conv=conv, bidx=bidx, row_off=row_off, cnt32=cnt32,
                          inv=inv, R=R, Mb=Mb, row_of_ray=row_of_ray, j_of_ray=j_of_ray, off=off, xyz_rows=xyz_rows, app=app,
                          heads=heads, V=V, N=N, r1=r1, f0=f0, diff=diff, feat=feat, L=L, hl=hl, dl=dl, mip=mip, brays=brays,
                          brdf=brdf, brdf_mask=brdf_mask, child=None, idx_re=None, idx_no=None, sparse_n=sparse_n, sf_rows=sf_rows, gr_rows=gr_rows)
        # ---- incoming radiance of the secondary rays (models/microfacet.py:475-563)
        if lvl < len(model.max_retrace_rays):
            num_retrace = min(R, model.max_retrace_rays[lvl])
            if num_retrace >= R and not sorts:
                noise.skip("rand", (R,))

                def mlp():          # needs nothing of the level below: runs under its sampler's read-back
                    t.brdf, t.brdf_mask = hip.brdf_mlp_fwd(self.mlp_ws, hl, dl, feat, r1, row_of_ray, self.mlp_bias,
                                                           with_mask=True)
                t.child = self._fwd(lvl + 1, brays, focal, mip, noise, is_train, filler=mlp)
                brdf = t.brdf
                if t.child.M == 0:
                    raise Unsupported("no secondary sample")
                t.n_samples += t.child.n_samples
                incoming = t.child.rgb_map
            else:
                w_rows = torch.index_select(w, 0, bidx)
                cc = hip.retrace_scores(brdf, V, N, lpdf, w_rows, cnt32, row_of_ray)
                cc = cc / cc.sum() * num_retrace
                cc = cc + noise.uniform((R,))
                order = hip.argsort_f32(cc.contiguous()).long()
                if trace is not None:
                    trace[f"retrace_order_own{lvl}"] = order
                if pins is not None and lvl in pins.retrace_order:
                    order = pins.retrace_order[lvl].to(order.device)
                cut = max(R - num_retrace, 0)
                t.idx_re, t.idx_no = order[cut:], order[:cut]
                if trace is not None:
                    trace.update({f"retrace_score{lvl}": cc, f"retrace_order{lvl}": order, f"retrace_idx{lvl}": t.idx_re})
                # idx_re and idx_no partition the rays: both index_copy_ together write every row
                incoming = torch.empty((R, 3), dtype=torch.float32, device=rays.device)
                sel = torch.index_select
                if t.idx_re.shape[0] > 0:
                    t.brays_re, t.mip_re = sel(brays, 0, t.idx_re), sel(mip, 0, t.idx_re)
                    t.child = self._fwd(lvl + 1, t.brays_re, focal, t.mip_re, noise, is_train)
                    if t.child.M == 0:
                        raise Unsupported("no secondary sample")
                    t.n_samples += t.child.n_samples
                    incoming.index_copy_(0, t.idx_re, t.child.rgb_map)
                if t.idx_no.shape[0] > 0:
                    t.brays_no, t.mip_no = sel(brays, 0, t.idx_no), sel(mip, 0, t.idx_no)
                    noise.skip("rand", (t.idx_no.shape[0],))
                    noise.skip("rand", (t.idx_no.shape[0],))
                    incoming.index_copy_(0, t.idx_no, self._env_fwd(t.brays_no, t.mip_no))
        else:
            noise.skip("rand", (R,))
            noise.skip("rand", (R,))
            incoming = self._env_fwd(brays, mip)
        # ---- Fresnel mix + per-ray sums, tonemap, background (ShadeCompose)
        if per_ray_bg:
            noise.skip("rand", (B,))
            noise.skip("rand", (B,))
        contrib = hip.shade_mix_fwd(V, f0, diff, cnt32, row_of_ray, L, incoming, brdf)
        refl = hip.segment_sum(contrib, None, row_off, Mb, lanes=8)
        rgb_map, acc, rgb_lin, ori = hip.ray_compose_fwd(w, refl, inv, nr if lvl == 0 else None, S.rays, offsets, B, bg,
                                                         per_ray_bg, lvl == 0, False, lvl == 0)
        t.__dict__.update(incoming=incoming, bg=bg, refl=refl, rgb_map=rgb_map, acc=acc, rgb_lin=rgb_lin, ori=ori,
                          per_ray_bg=per_ray_bg)
        if trace is not None:
            trace.update({f"rgb_map{lvl}": rgb_map, f"acc_map{lvl}": acc, f"whole_valid{lvl}": S.whole_valid,
                          f"incoming{lvl}": incoming, f"ori{lvl}": ori})
        return t

    # ---- backward of one recursion level -----------------------------------------------------------------------------
    def _bwd(self, t, d_rgb, d_acc, d_ori):
        a = self.acc
        S, lvl = t.S, t.lvl
        view = lvl > 0                  # the rows' view vector is the direction the level above sampled: keep its adjoint
        d_w, d_refl, d_nrm = hip.ray_compose_bwd(t.w, t.refl, t.inv, t.nr if d_ori is not None else None, S.rays, S.ray_id,
                                                 t.bg, t.per_ray_bg, lvl == 0, False, t.rgb_lin, d_rgb, d_acc, d_ori,
                                                 d_ori is not None)
        # needs only d_w and is first used by the field walk: issued here it fills time in which this stream would wait for the
        # side streams below, at the end of the level it would sit on the critical path (45 us for the re-traced rays)
        d_sigma = hip.composite_bwd(t.sg, S.dist, t.w, t.offsets, t.B, self.scale, d_w)
        early_walk = False
        if t.sparse_n and t.M >= WALK_SIDE_MIN_SAMPLES:
            # the value-only walk of a re-traced level needs nothing but d_sigma: on a side stream from here on, next to the
            # whole shading backward, instead of at the end of the pass in front of the other walks
            wfork = self._fork(("walk", lvl))
            if wfork is not None:
                p_, dpk, dlk, apl, ali, basis = self.nerf.rf._tables()
                with _on(wfork):
                    hip.vm_query_bwd_segments(p_, [(S.xyzt, t.sf, None, d_sigma, None, None, None)], dpk, dlk, apl, ali, basis,
                                              a.g_dpk, a.g_dlk, a.g_apl, a.g_ali, None)
                self._walk_forks.append((wfork, d_sigma))
                early_walk = True
        d_rays, env_fork = None, None
        if t.per_ray_bg:
            d_bg = (1 - t.acc)[:, None] * d_rgb
            env_rows = S.rays if t.B == S.rays.shape[0] else S.rays[:t.B]
            # atomic-bound scatter: next to the rest of this level's backward (when it is long enough to be worth a fork)
            env_fork = self._fork(("env", lvl)) if t.B >= MLP_SIDE_MIN_RAYS else None
            if env_fork is not None:
                with _on(env_fork):
                    d_rays = self._env_bwd(env_rows, t.rough, d_bg)
            else:
                d_rays = self._env_bwd(env_rows, t.rough, d_bg)
        dV_rows = None
        if view:
            d_inc, d_brdf, dL, d_fd, dV = hip.shade_mix_bwd_view(t.V, t.f0, t.diff, t.cnt32, t.row_of_ray, t.L, t.incoming,
                                                                 t.brdf, d_refl)
            dV_rows = hip.segment_sum(dV, None, t.row_off, t.Mb, lanes=8)
        else:
            d_inc, d_brdf, dL, d_fd = hip.shade_mix_bwd(t.V, t.f0, t.diff, t.cnt32, t.row_of_ray, t.L, t.incoming, t.brdf,
                                                        d_refl)
        rows6 = hip.segment_sum_wide(d_fd, 6, t.row_off, t.Mb)
        # ---- BRDF MLP backward: on a side stream, next to the adjoint of the bounce rays below
        # (or next to the env-map adjoint of this level's own bounce rays when there are many of them: with a partial re-trace
        # the level below gets the rest of the ray budget, microfacet.py:318-331 -- 0.5 M rays at half the re-trace count)
        n_env = t.R - (t.idx_re.shape[0] if t.idx_re is not None else (t.R if t.child is not None else 0))
        below = t.child is not None and t.child.B >= MLP_SIDE_MIN_RAYS
        fork = self._fork(("mlp", lvl)) if (below or n_env >= MLP_SIDE_MIN_ENV_RAYS) else None
        if fork is not None:
            with _on(fork):
                d_feat = hip.brdf_mlp_bwd(self.mlp_ws, t.hl, t.dl, t.feat, t.r1, t.row_of_ray, t.brdf, t.brdf_mask, d_brdf,
                                           a.g_mlp, max_workgroups=MLP_SIDE_WGS if below else MLP_SIDE_WGS_ENV)
        # ---- adjoint of the incoming radiance -> adjoint of the bounce rays [R,6]
        if t.idx_re is None and t.child is not None:
            d_brays = self._bwd(t.child, d_inc, None, None)
        elif t.idx_re is None:
            d_brays = self._env_bwd(t.brays, t.mip, d_inc)
        else:
            d_brays = torch.empty_like(t.brays)      # the two index sets partition the rays: every row is written below
            sel = torch.index_select
            if t.idx_re.shape[0] > 0:
                d_brays.index_copy_(0, t.idx_re, self._bwd(t.child, sel(d_inc, 0, t.idx_re), None, None))
            if t.idx_no.shape[0] > 0:
                d_brays.index_copy_(0, t.idx_no, self._env_bwd(t.brays_no, t.mip_no, sel(d_inc, 0, t.idx_no)))
        if lvl == 0 and self._last_chunk and a.used_env:
            # every environment adjoint of the optimizer step has been queued (level 0 has no background lookup of its own):
            # the two reverse prefix sums of the env-map table run on a side stream from here, next to the rest of the pass,
            # instead of after the field walks (end_step picks the result up)
            sfork = self._fork("sat_bwd")
            if sfork is not None:
                bgm = self.nerf.bg_module
                act, _sat, _pole = bgm._tables()
                if a.d_bg is None:
                    a.d_bg = torch.empty_like(bgm.bg_mat.detach().reshape(3, bgm.bg_mat.shape[-2], bgm.bg_mat.shape[-1]))
                with _on(sfork):
                    d_bg = hip.sat_build_bwd(a.d_sat, bgm.bg_mat.detach(), act, a.d_pole, sc=bgm._dev_scalars(), out=a.d_bg)
                self._early_env = (sfork, d_bg)
        # ---- BounceRays backward: BRDF MLP, GGX rays, row preparation, heads, appearance rows
        if fork is not None:
            self._join(fork, d_feat)
        else:
            d_feat = hip.brdf_mlp_bwd(self.mlp_ws, t.hl, t.dl, t.feat, t.r1, t.row_of_ray, t.brdf, t.brdf_mask, d_brdf, a.g_mlp)
        sobol = self.nerf.model.brdf_sampler.angs
        if view:
            d_nrv = hip.ggx_rays_bwd_view(t.V, t.N, t.r1, t.off, sobol, t.row_of_ray, t.j_of_ray, dL, d_brays)
            rows7 = hip.segment_sum_wide(d_nrv, 7, t.row_off, t.Mb)
            dN, dr1, dV_ggx = rows7[:, 0:3], rows7[:, 3], rows7[:, 4:7]
        else:
            d_nr = hip.ggx_rays_bwd(t.V, t.N, t.r1, t.off, sobol, t.row_of_ray, t.j_of_ray, dL, d_brays)
            rows4 = hip.segment_sum(d_nr, None, t.row_off, t.Mb, lanes=8)
            dN, dr1 = rows4[:, 0:3], rows4[:, 3]
        d_normals, d_heads, d_app = hip.bounce_prep_bwd(None if t.sparse_n else t.inv, t.nr, t.heads, S.ray_id, S.rays, t.conv,
                                                        self.min_rough, self.detach_n, dN, dr1, rows6[:, 0:3], rows6[:, 3:6],
                                                        d_feat, bidx=t.bidx, row_inputs=2 if t.sparse_n else 1)
        hp, hW, hb = self.heads
        d_app = hip.heads_bwd(t.app, hW, hb, hp, d_heads, a.g_hW, a.g_hb, add_into=d_app)
        self.app_segs.append((t.xyz_rows, None, None, None, None, None, d_app))
        if self.detach_n:
            d_normal = d_nrm
        elif d_nrm is not None:
            d_normal = d_normals.add_(d_nrm)
        else:
            d_normal = d_normals
        if t.sparse_n:      # value-only walk over the level's samples; the rows carry the normal adjoint (with a zero d_sigma so
            # that they share a walk with the level-0 samples)
            if not early_walk:
                self.dens_segs.append((S.xyzt, t.sf, None, d_sigma, None, None, None))
            if d_normal is not None:
                self.dens_segs.append((t.xyz_rows, t.sf_rows, t.gr_rows, _zeros(t.sf_rows), None, d_normal, None))
        else:
            self.dens_segs.append((S.xyzt, t.sf, t.gr, d_sigma, None, d_normal, None))
        if env_fork is not None:
            self._join(env_fork, d_rays)
        if view:        # V_row = -direction of the row's ray: both view adjoints back onto the rays, one launch
            if d_rays is None:
                d_rays = torch.zeros_like(S.rays)
            hip.view_adjoint_to_rays(S.ray_id, t.bidx, dV_rows, dV_ggx, d_rays)
        return d_rays

    # ---- side streams ---------------------------------------------------------------------------------------------------
    def _fork(self, key):
        """-> (main, side) with the side stream waiting for everything queued on the current one, or None"""
        if not self.overlap:
            return None
        if key not in self._side:
            self._side[key] = torch.cuda.Stream()
        main, side = self._main, self._side[key]
        side.wait_stream(main)
        return main, side

    @staticmethod
    def _join(fork, *made_on_side):
        """the current stream waits for the side stream; tensors allocated under the side stream and consumed (and released)
        on this one are registered with the allocator"""
        main, side = fork
        main.wait_stream(side)
        for x in made_on_side:
            x.record_stream(main)

    def _flush_walks(self):
        rf, a = self.nerf.rf, self.acc
        p, dpk, dlk, apl, ali, basis = rf._tables()

        def walk(all_segs, g_basis):
            # one walk takes sample sets that carry the same adjoints (with detached normals the re-traced samples have no
            # normal adjoint while the primary ones still have the orientation-loss term)
            for key in dict.fromkeys(tuple(x is not None for x in sg[3:]) for sg in all_segs):
                segs = [sg for sg in all_segs if tuple(x is not None for x in sg[3:]) == key]
                for i in range(0, len(segs), hip.VM_MAX_SEGMENTS):
                    hip.vm_query_bwd_segments(p, segs[i:i + hip.VM_MAX_SEGMENTS], dpk, dlk, apl, ali, basis, a.g_dpk,
                                              a.g_dlk, a.g_apl, a.g_ali, g_basis)

        walk(self.dens_segs, None)       # (the appearance walk next to the density walk on a second stream: no gain)
        walk(self.app_segs, a.g_basis)
        for wfork, _keep in self._walk_forks:
            self._join(wfork)
        self._walk_forks = []
        self.dens_segs, self.app_segs = [], []

    # ---- evaluation: forward only ---------------------------------------------------------------------------------------
    def _begin(self, dev, noise):
        nerf = self.nerf
        rf, model, bgm = nerf.rf, nerf.model, nerf.bg_module
        if hasattr(noise, "begin_pass"):
            noise.begin_pass()
        self.scale = float(rf.distance_scale)
        self.anoise, self.min_rough, self.detach_n = float(model.anoise), float(model.min_rough), bool(model.detach_N)
        self.near1 = 3 * float(hip.host(nerf.sampler.stepsize))
        self.white = _white(dev)

    def _begin_tables(self):
        """the per-step rebuilds of the field side (packed tables, stacked heads, MLP weights); the env-map side follows at
        its first use (_fwd).  Passed to the level-0 _fwd as its filler."""
        nerf = self.nerf
        nerf.rf._fwd_tables()
        hp, hW, hb, _, _ = nerf.model.diffuse_module.head_pass()
        self.heads = (hp, hW, hb)
        self.mlp_ws, self.mlp_bias, _, _ = nerf.model.brdf.mlp_pass()

    @torch.no_grad()
    def render_chunk(self, rays, focal, noise):
        """`TensorNeRF.forward(rays, focal, bg_col=white, is_train=False, draw_debug=False)` as the same straight-line kernel
        sequence as the training forward: -> (rgb_map [b,3], acc_map [b], b = rays the sampler kept, n_samples).  Raises
        Unsupported (configuration, no sample, no bounce row): the caller renders that chunk through the module."""
        nerf = self.nerf
        if not self.supported():
            raise Unsupported("configuration")
        mods = [m for m in (nerf.rf, nerf.bg_module, nerf.model.brdf, nerf.model.diffuse_module) if hasattr(m, "begin_pass")]
        for m in mods:
            m.begin_pass()
        try:
            core = self.core()
            if core is not None:
                self._core_sync(rays.device, focal, False)
                try:
                    out = core.render(rays, float(focal), noise)
                except RuntimeError as e:
                    if "Unsupported" in str(e):
                        raise Unsupported(str(e)) from None
                    raise
                if out is None:
                    raise Unsupported("no sample")
                return out[0], out[1], out[2], list(out[3])
            self._begin(rays.device, noise)
            t = self._fwd(0, rays, focal, None, noise, is_train=False, filler=self._begin_tables)
            if t.M == 0:
                raise Unsupported("no sample")
            return t.rgb_map, t.acc, t.B, t.n_samples
        finally:
            for m in mods:
                m.end_pass()

    # ---- one chunk: forward, loss, backward ---------------------------------------------------------------------------
    @torch.no_grad()
    def chunk(self, rays, gt, focal, noise, inv_lbatch, wts, want_total=False, last=False, next_rays=None):
        """wts = (w_photo, w_l1, w_ori, w_acc).  Returns dict(loss 0-d tensor, kept, n_samples) -- loss None when the chunk
        had no sample (train.py:567-568 skips it).  want_total: also evaluate the chunk's total loss value (the gradients do
        not need it: every term enters linearly with a constant weight)."""
        nerf = self.nerf
        if not self.supported():
            raise Unsupported("configuration")
        core = self.core()
        if core is not None:
            return self._core_chunk(core, rays, gt, focal, noise, inv_lbatch, wts, want_total, last, next_rays)
        rf, model, bgm = nerf.rf, nerf.model, nerf.bg_module
        dev = rays.device
        mods = [m for m in (rf, bgm, model.brdf, model.diffuse_module) if hasattr(m, "begin_pass")]
        for m in mods:
            m.begin_pass()
        try:
            self._begin(dev, noise)
            self._main = torch.cuda.current_stream() if self.overlap else None
            self.dens_segs, self.app_segs, self._walk_forks = [], [], []
            t = self._fwd(0, rays, focal, None, noise, filler=self._begin_tables)
            if t.M == 0:
                return dict(loss=None, kept=t.B, n_samples=[0])
            a = self._accumulators(dev)
            b = t.rgb_map.shape[0]
            gt_b = gt[:b].contiguous()
            loss = hip.sqerr_fwd(t.rgb_map, gt_b)
            total = None
            if want_total:
                dens = list(rf.density_rf.app_plane) + list(rf.density_rf.app_line)
                l1 = hip.l1_mean_fwd([x.detach() for x in dens])
                total = hip.loss_mix_fwd([loss, l1, t.ori, t.acc], wts, inv_lbatch)
            d_loss, _d_l1, d_ori, d_acc = hip.loss_mix_bwd([loss.shape, loss.shape, t.ori.shape, t.acc.shape], wts, inv_lbatch,
                                                           _one(dev))
            d_rgb = hip.sqerr_bwd(t.rgb_map, gt_b, d_loss)
            self._last_chunk = bool(last)
            self._bwd(t, d_rgb, d_acc, d_ori)
            self._flush_walks()
            self.n_loss_chunks += 1
            self.l1_scale += float(wts[1]) * float(inv_lbatch)
            return dict(loss=loss, total=total, kept=b, n_samples=t.n_samples)
        finally:
            for m in mods:
                m.end_pass()

    # ---- accumulators -> parameter gradients ----------------------------------------------------------------------------
    @torch.no_grad()
    def end_step(self):
        a = self.acc
        if a is None:
            return
        nerf = self.nerf
        rf, model, bgm = nerf.rf, nerf.model, nerf.bg_module
        p = rf._tables()[0]
        l1 = None
        if self.l1_scale != 0.0:          # the L1 term's gradient rides in the unpack launch (one launch less on the step's tail)
            if a.l1 is None or a.l1[0] != self.l1_scale:
                a.l1 = (self.l1_scale, torch.full((), self.l1_scale, dtype=torch.float32, device=a.flat.device))
            l1 = ([x.detach() for x in list(rf.density_rf.app_plane) + list(rf.density_rf.app_line)], a.l1[1])
        if a.gp is None:
            a.gp, a.gl = hip.vm_unpack_density_grad(p, a.g_dpk, a.g_dlk, l1=l1)
        else:
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
            if self._early_env is not None:
                fork, d_bg = self._early_env
                if fork == "core":
                    self._core.join_early_env()
                else:
                    self._join(fork)
                self._early_env = None
            else:
                d_bg = a.d_bg = hip.sat_build_bwd(a.d_sat, bgm.bg_mat.detach(), act, a.d_pole, sc=sc, out=a.d_bg)
            if bgm.bg_mat.requires_grad:
                grads.append((bgm.bg_mat, d_bg.reshape(bgm.bg_mat.shape)))
            if bgm.brightness_lr != 0 or bgm.mul_lr != 0:           # lr 0 (microfacet_tensorf2.yaml:150-151): no update anyway
                d_pre = d_bg / sc[2]
                grads.append((bgm.brightness, d_pre.sum(dtype=torch.float64)))
                grads.append((bgm.mul, (d_pre * bgm.bg_mat.detach().reshape(d_pre.shape)).sum(dtype=torch.float64)))
            if bgm.mipbias.requires_grad:
                grads.append((bgm.mipbias, a.d_mip.to(torch.float64).reshape(())))
        for prm, g in grads:
            if not prm.requires_grad:
                continue
            prm.grad = g if prm.grad is None else prm.grad.add_(g)
        self.acc = None


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
