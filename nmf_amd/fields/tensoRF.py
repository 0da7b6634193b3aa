"""TensoRF vector-matrix field on HIP kernels -- host-side mirror of the reference's
fields/tensoRF.py (TensoRF :25-243, TensorVMSplit :246-445) and fields/tensor_base.py
(TensorBase :32-168, TensorVoxelBase :171-252): same constructor keywords (configs/field/tensorf_og.yaml),
method names, return values and state_dict keys.

MI355X-first differences (documented in DESIGN.md):
  * factor tables are stored channel-last (torch.channels_last for [1,C,G,G]; state_dict shapes unchanged),
  * density / appearance / normals of a sample come out of ONE kernel launch (`query`); the three
    reference entry points are thin views of it,
  * the stencil-filtered derivative tables are rebuilt once per parameter update, not per call.
"""
import math

import torch
import torch.nn.functional as F

from .. import hip
from ..functional import FieldGrads, GradPass, L1Mean, VMAppQuery, VMQuery, VMQueryWeights, FastPrivateAttrs


def N_to_reso(n_voxels, bbox):
    # utils.py:55-58
    xyz_min, xyz_max = bbox
    voxel_size = ((xyz_max - xyz_min).prod() / n_voxels).pow(1 / 3)
    return ((xyz_max - xyz_min) / voxel_size).long().tolist()


class TensoRF(torch.nn.Module):
    """One VM-decomposed grid: 3 planes [1,C,G,G] x 3 lines [1,C,G,1] (fields/tensoRF.py:25-63)."""

    def __init__(self, grid_size, dim, init_mode, interp_mode, init_val, lr, smoothing=0.5, numer_grad=True):
        super().__init__()
        if interp_mode != "bilinear" or not numer_grad:
            raise NotImplementedError("HIP field supports interp_mode=bilinear with numer_grad=True (tensorf_og.yaml)")
        self.matMode = [[0, 1], [0, 2], [1, 2]]
        self.vecMode = [2, 1, 0]
        self.grid_size = int(grid_size)
        self.lr = lr
        self.smoothing = smoothing
        self._dim = dim
        planes, lines = [], []
        for _ in range(3):   # init_mode 'rand' -> default branch of init_one_svd (:151-155)
            planes.append(torch.nn.Parameter(self._cl(init_val * torch.randn(1, dim, grid_size, grid_size))))
            lines.append(torch.nn.Parameter(self._cl(init_val * torch.randn(1, dim, grid_size, 1))))
        self.app_plane = torch.nn.ParameterList(planes)
        self.app_line = torch.nn.ParameterList(lines)

    @staticmethod
    def _cl(t):
        """dense channel-last storage ([H][W][C]) behind the reference's [1,C,H,W] shape"""
        _, C, H, W = t.shape
        buf = t.permute(0, 2, 3, 1).contiguous()
        return buf.permute(0, 3, 1, 2)

    def dim(self):
        return self._dim * 3

    def get_optparam_groups(self, lr_scale=1):
        return [{"params": self.app_plane.parameters(), "lr": self.lr * lr_scale},
                {"params": self.app_line.parameters(), "lr": self.lr * lr_scale}]

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # accept reference checkpoints of any resolution / memory format
        for name, plist in (("app_plane", self.app_plane), ("app_line", self.app_line)):
            for i in range(3):
                k = f"{prefix}{name}.{i}"
                if k in state_dict:
                    src = state_dict[k]
                    if src.shape != plist[i].shape:
                        plist[i] = torch.nn.Parameter(self._cl(torch.empty_like(src, device=plist[i].device)))
                    state_dict[k] = self._cl(src.to(plist[i].device))
                    self.grid_size = int(src.shape[2])
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        with torch.no_grad():
            for plist in (self.app_plane, self.app_line):
                for i in range(3):
                    if not hip.channels_last_ptr_ok(plist[i]):
                        plist[i].data = self._cl(plist[i].data)

    def tables(self):
        """kernel views: planes [G,G,C], lines [G,C] (no copy when the storage is channel-last)"""
        G, C = self.grid_size, self._dim
        pl, li = [], []
        for i in range(3):
            p = self.app_plane[i]
            if not hip.channels_last_ptr_ok(p):
                p.data = self._cl(p.data)
            pl.append(p.detach().permute(0, 2, 3, 1).reshape(G, G, C))
            q = self.app_line[i]
            if not hip.channels_last_ptr_ok(q):
                q.data = self._cl(q.data)
            li.append(q.detach().permute(0, 2, 3, 1).reshape(G, C))
        return pl, li

    @torch.no_grad()
    def upsample(self, res_target):
        # fields/tensoRF.py:207-227
        for i in range(3):
            m0, m1 = self.matMode[i]
            self.app_plane[i] = torch.nn.Parameter(self._cl(F.interpolate(
                self.app_plane[i].data.contiguous(), size=(res_target[m1], res_target[m0]), mode="bilinear",
                align_corners=True)))
            self.app_line[i] = torch.nn.Parameter(self._cl(F.interpolate(
                self.app_line[i].data.contiguous(), size=(res_target[self.vecMode[i]], 1), mode="bilinear",
                align_corners=True)))
        self.grid_size = int(res_target[0])


class TensorVMSplit(FastPrivateAttrs, torch.nn.Module):
    def __init__(self, aabb, smoothing=1, interp_mode="bilinear", calibrate=True, dbasis=True, triplanar=False,
                 init_mode="trig", d_init_val=0.1, app_init_val=0.1, numer_grad=True, density_n_comp=16,
                 appearance_n_comp=24, step_ratio=0.5, app_dim=24, density_res_multi=1, N_voxel_init=2097156,
                 N_voxel_final=27000000, upsamp_list=(2000, 3000, 4000, 5500, 7000), grid_size=None,
                 density_shift=-4, activation="softplus", lr=2e-2, lr_net=1e-3, contract_space=False,
                 distance_scale=25, num_pretrain=0, **kwargs):
        super().__init__()
        if dbasis or triplanar or contract_space or activation != "softplus" or smoothing != 1:
            raise NotImplementedError("HIP field implements the tensorf_og.yaml configuration "
                                      "(dbasis=False, softplus, smoothing=1, AABB space)")
        if density_n_comp != hip_const("NMF_DENSITY_C") or appearance_n_comp != hip_const("NMF_APP_C") or app_dim != 24:
            raise NotImplementedError("kernels are compiled for density_n_comp=16, appearance_n_comp=24, app_dim=24")
        self.lr, self.lr_net = lr, lr_net
        self.activation = activation
        self.num_pretrain = num_pretrain
        self.density_shift = density_shift
        self.contract_space = contract_space
        self.distance_scale = distance_scale
        self.separate_appgrid = True
        self.calibrate = calibrate
        self.dbasis = dbasis
        self.init_mode, self.interp_mode = init_mode, interp_mode
        self.density_n_comp = [density_n_comp] * 3
        self.app_n_comp = [appearance_n_comp] * 3
        self.density_res_multi = density_res_multi
        self.app_dim = app_dim
        self.step_ratio = step_ratio
        self.smoothing = smoothing
        self.upsamp_list = list(upsamp_list)
        self.matMode = [[0, 1], [0, 2], [1, 2]]
        self.vecMode = [2, 1, 0]
        self.set_aabb(torch.as_tensor(aabb, dtype=torch.float32))
        self.N_voxel_list = torch.round(torch.linspace(N_voxel_init ** (1 / 3), N_voxel_final ** (1 / 3),
                                                       len(self.upsamp_list) + 1) ** 3).long().tolist()[1:]
        gs = torch.tensor(N_to_reso(N_voxel_init, self.aabb)) if grid_size is None else grid_size
        self.update_stepSize(gs)
        G = int(self.grid_size[0])
        self.density_rf = TensoRF(G, density_n_comp, init_mode, interp_mode, d_init_val, lr, smoothing)
        self.app_rf = TensoRF(G, appearance_n_comp, init_mode, interp_mode, app_init_val, lr, smoothing)
        self.basis_mat = torch.nn.Linear(self.app_rf.dim(), app_dim, bias=False)
        self.dbasis_mat = torch.nn.Linear(self.density_rf.dim(), 1, bias=False)
        self._cache = None
        self._plist = None
        self._pass, self._pass_open, self._tables_memo = None, False, None
        self._last_holder = None
        self.table_dtype = "f32"
        self._bf16 = None
        self._fwd_memo = None

    # ---- geometry bookkeeping (fields/tensor_base.py:55-64,219-232) ---------------------------------
    def set_register(self, name, val):
        if hasattr(self, name):
            setattr(self, name, val.type_as(getattr(self, name)))
        else:
            self.register_buffer(name, val)

    def set_aabb(self, aabb):
        self.set_register("aabb", aabb)
        self.set_register("aabbSize", aabb[1] - aabb[0])
        self.set_register("invaabbSize", 2.0 / self.aabbSize)
        self.set_register("aabbDiag", torch.sqrt(torch.sum(torch.square(self.aabbSize))))

    def update_stepSize(self, grid_size):
        grid_size = torch.as_tensor(grid_size, dtype=torch.long, device=self.aabb.device)
        self.set_register("grid_size", grid_size)
        self.set_register("units", self.aabbSize / (self.grid_size - 1))
        self.set_register("stepsize", torch.min(self.units) * self.step_ratio)
        self.nSamples = int((self.aabbDiag / self.stepsize).item()) + 1

    def get_device(self):
        return self.aabbSize.device

    def normalize_coord(self, xyz_sampled):
        coords = (xyz_sampled[..., :3] - self.aabb[0]) * self.invaabbSize - 1
        return torch.cat((coords, xyz_sampled[..., 3:4]), dim=-1)

    # ---- kernel tables ---------------------------------------------------------------------------------
    def _param_list(self):
        """the 13 tensors of the field (cached: the Parameter objects only change on upsample / load_state_dict)"""
        if self._plist is None:
            self._plist = (list(self.density_rf.app_plane) + list(self.density_rf.app_line)
                           + list(self.app_rf.app_plane) + list(self.app_rf.app_line) + [self.basis_mat.weight])
        return self._plist

    def _tables(self):
        # parameters cannot change inside a pass: one version / pointer comparison per pass, then the memo
        if self._pass_open and self._tables_memo is not None:
            return self._tables_memo
        tab = self._tables_checked()
        if self._pass_open:
            self._tables_memo = tab
        return tab

    def _tables_checked(self):
        ps = self._param_list()
        key = [p._version for p in ps]
        ptrs = [p.data_ptr() for p in ps]
        c = self._cache
        if c is not None and c[2] == ptrs:
            if c[0] != key:
                # same storages, new values (an optimizer step): the views of the factors stay valid, only the packed
                # density tables (value + derivative planes) are stale -- re-pack them in place
                vp, dpk, dlk = c[1][0], c[1][1], c[1][2]
                hip.vm_pack_density(vp, c[3][0], c[3][1], out=(dpk, dlk))
                self._cache = (key, c[1], ptrs, c[3])
            return c[1]
        G = int(hip.host(self.grid_size)[0])
        vp = hip.vm_params(self.aabb, self.invaabbSize, self.density_shift, G)
        dpl, dli = self.density_rf.tables()
        apl, ali = self.app_rf.tables()
        dpk, dlk = hip.vm_pack_density(vp, dpl, dli)
        basis = self.basis_mat.weight.detach().contiguous()
        # the pack may re-layout a parameter in place (channel-last): take versions / pointers afterwards
        self._cache = ([p._version for p in ps], (vp, dpk, dlk, apl, ali, basis), [p.data_ptr() for p in ps], (dpl, dli))
        return self._cache[1]

    # ---- bf16-table variant (BASELINE configs[1]) -----------------------------------------------------------------
    def set_table_dtype(self, dtype):
        """'f32' (default) or 'bf16': the FORWARD queries read bfloat16 copies of the factor tables (packed density planes /
        lines, appearance planes / lines) -- half the bytes per tap, fp32 arithmetic.  The parameters, the optimizer and the
        backward walk keep the fp32 tables (the copies are refreshed once per parameter update, one launch), i.e. the usual
        mixed-precision arrangement: low-precision operands in the forward pass, fp32 master weights."""
        if dtype not in ("f32", "bf16"):
            raise ValueError(dtype)
        self.table_dtype = dtype
        self._bf16 = None
        self._fwd_memo = None

    def _value_tables(self):
        """(planes [G,G,16] x 3, lines [G,16] x 3): the density factors themselves, for value-only queries (hip.vm_query_sigma);
        with bf16 tables their bfloat16 copies (refreshed with the other copies, _fwd_tables)"""
        if self.table_dtype != "f32":
            self._fwd_tables()
            c = self._bf16[1]
            return c[12:15], c[15:18]
        self._tables()
        return self._cache[3]

    def _fwd_tables(self):
        """(p, dpk, dlk, apl, ali, basis) as the forward kernel should read them"""
        if self.table_dtype == "f32":
            return self._tables()
        if self._pass_open and self._fwd_memo is not None:
            return self._fwd_memo
        tab = self._tables()
        p, dpk, dlk, apl, ali, basis = tab
        key = (self._cache[0], self._cache[2])
        srcs = list(dpk) + list(dlk) + list(apl) + list(ali) + list(self._cache[3][0]) + list(self._cache[3][1])
        if self._bf16 is None or self._bf16[0][1] != key[1]:
            copies = hip.to_bf16_tables(srcs)
            self._bf16 = (key, copies)
        elif self._bf16[0][0] != key[0]:
            hip.to_bf16_tables(srcs, self._bf16[1])
            self._bf16 = (key, self._bf16[1])
        c = self._bf16[1]
        out = (p, c[0:3], c[3:6], c[6:9], c[9:12], basis)
        if self._pass_open:
            self._fwd_memo = out
        return out

    def _grads_to_param_layout(self, gp, gl, g_apl, g_ali, g_basis):
        """kernel layouts ([G,G,C] / [G,C]) -> views shaped like the parameters (no copies)"""
        out = []
        for t in gp:
            out.append(t if t.dim() == 4 else t.permute(2, 0, 1)[None])     # vm_unpack_density_grad returns parameter shapes
        for t in gl:
            out.append(t if t.dim() == 4 else t.t()[None, :, :, None])
        for t in g_apl:
            out.append(t.permute(2, 0, 1)[None])
        for t in g_ali:
            out.append(t.t()[None, :, :, None])
        out.append(g_basis)
        return out

    # ---- the fused query and the reference's three entry points ------------------------------------
    def query(self, xyz_sampled, want_app=True, want_normal=True):
        """-> sigma [M], sigma_feat [M], app [M,24], normals [M,3] from one launch"""
        if xyz_sampled.shape[0] == 0:
            z = xyz_sampled.new_zeros
            return z(0), z(0), z((0, self.app_dim)), z((0, 3))
        xyz = xyz_sampled.detach()
        if xyz.shape[-1] == 3:
            xyz = torch.cat([xyz, torch.zeros_like(xyz[:, :1])], -1)
        holder, token = self._pass_token()
        return VMQuery.apply(self, xyz.contiguous(), want_app, want_normal, holder, token)

    def query_weights(self, xyzt, dist, offsets, b, want_app=True, want_normal=True):
        """-> weights [M] (raw2alpha over the ray segments `offsets`, tensor_nerf.py:19-35,366), sigma_feat [M], app [M,24],
        normals [M,3]: query() and the compositing as one graph node.  xyzt [M,4] with M > 0."""
        holder, token = self._pass_token()
        return VMQueryWeights.apply(self, xyzt.detach().contiguous(), want_app, want_normal, holder, token, dist, offsets, b,
                                    float(self.distance_scale))

    # ---- gradient pass: all queries between begin_pass() and end_pass() share one FieldGrads node -----------
    def begin_pass(self):
        self._pass, self._pass_open, self._tables_memo, self._fwd_memo = None, True, None, None

    def end_pass(self):
        self._pass, self._pass_open, self._tables_memo, self._fwd_memo = None, False, None, None

    def _pass_token(self):
        ps = self._param_list()
        if not (torch.is_grad_enabled() and any(p.requires_grad for p in ps)):
            return None, None
        if self._pass_open and self._pass is not None:
            return self._pass
        holder = GradPass()
        token = FieldGrads.apply(holder, self, *ps)
        if self._pass_open:
            self._pass = (holder, token)
            self._last_holder = holder
        return holder, token

    def compute_densityfeature(self, xyz_sampled, activate=True):
        sg, sf, _, _ = self.query(xyz_sampled, want_app=False, want_normal=False)
        return sg if activate else sf.detach()

    def compute_appfeature(self, xyz_sampled):
        """[M,24] appearance features only (one launch without the density branch)"""
        if xyz_sampled.shape[0] == 0:
            return xyz_sampled.new_zeros((0, self.app_dim))
        xyz = xyz_sampled.detach()
        if xyz.shape[-1] == 3:
            xyz = torch.cat([xyz, torch.zeros_like(xyz[:, :1])], -1)
        holder, token = self._pass_token()
        return VMAppQuery.apply(self, xyz.contiguous(), holder, token)

    def compute_normals(self, xyz):
        return self.query(xyz, want_app=False, want_normal=True)[3]

    def feature2density(self, f):
        return F.softplus(f.clamp(-15, 1e3) + self.density_shift)

    # ---- optimiser / regularisers / schedule ----------------------------------------------------------
    def get_optparam_groups(self, lr_scale=1):
        # fields/tensoRF.py:298-313
        return [{"params": self.basis_mat.parameters(), "lr": lr_scale * self.lr_net, "betas": [0.9, 0.99]},
                {"params": self.dbasis_mat.parameters(), "lr": lr_scale * self.lr_net, "betas": [0.9, 0.99]},
                *self.density_rf.get_optparam_groups(lr_scale), *self.app_rf.get_optparam_groups(lr_scale)]

    def density_L1(self, with_pass=False):
        """fields/tensoRF.py:332-340.  with_pass: the caller adds this term to the loss of the rendering pass that has just
        been evaluated (the trainer does), so its gradient can ride on that pass's table-gradient node.  After a training forward
        through the fused pass (TensorNeRF.fused_training_pass) the term always rides on that chunk's node -- the reference's loop
        adds it to the chunk's loss (train.py:672-677), and the node writes .grad itself."""
        h = self._last_holder
        holder = h if (with_pass or (h is not None and getattr(h, "chunk_pass", False) and not h.done and h.l1 is None)) else None
        return L1Mean.apply(holder, *self.density_rf.app_plane, *self.density_rf.app_line)

    @torch.no_grad()
    def flush_pending_l1(self):
        """Safety net for density_L1(with_pass=True): if the pass's gradient node did not run in the backward that consumed
        the term (no field query was connected to that loss), the term's gradient is applied here."""
        h = self._last_holder
        if h is None or h.l1 is None:
            return
        tensors, d_out = h.l1
        h.l1 = None
        grads = hip.l1_mean_bwd(tensors, d_out)
        for prm, g in zip(list(self.density_rf.app_plane) + list(self.density_rf.app_line), grads):
            prm.grad = g if prm.grad is None else prm.grad + g

    def vector_comp_diffs(self):
        total = 0
        for lines in (self.density_rf.app_line, self.app_rf.app_line):
            for v in lines:
                n_comp, n_size = v.shape[1:-1]
                dotp = torch.matmul(v.view(n_comp, n_size), v.view(n_comp, n_size).transpose(-1, -2))
                total = total + torch.mean(torch.abs(dotp.view(-1)[1:].view(n_comp - 1, n_comp + 1)[..., :-1]))
        return total

    def TV_loss_density(self, reg):
        return sum(reg(self.density_rf.app_plane[i]) * 1e-2 + reg(self.density_rf.app_line[i]) * 1e-3 for i in range(3))

    def TV_loss_app(self, reg, start_ind=0, end_ind=-1):
        return sum(reg(self.app_rf.app_plane[i]) * 1e-2 + reg(self.app_rf.app_line[i]) * 1e-3 for i in range(3))

    @torch.no_grad()
    def upsample_volume_grid(self, res_target):
        # fields/tensoRF.py:407-413
        self.app_rf.upsample(res_target)
        self.density_rf.upsample(res_target)
        self.update_stepSize(res_target)
        self._cache = None
        self._plist = None
        self._bf16 = None

    def check_schedule(self, iter, batch_mul):
        # fields/tensor_base.py:234-243
        ups = [i * batch_mul for i in self.upsamp_list]
        if iter in ups:
            n_voxels = self.N_voxel_list[ups.index(iter)]
            self.upsample_volume_grid(N_to_reso(n_voxels, self.aabb))
            return True
        return False

    def shrink(self, new_aabb, voxel_size):
        raise NotImplementedError("shrink is disabled in the reference for this config (alphagrid.py:109)")

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        k = prefix + "grid_size"
        if k in state_dict:
            self.update_stepSize(state_dict[k])
        self._cache = None
        self._plist = None
        self._bf16 = None
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


def hip_const(name):
    return {"NMF_DENSITY_C": 16, "NMF_APP_C": 24}[name]
