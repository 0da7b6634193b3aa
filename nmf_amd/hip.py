"""ctypes binding of libnmf_hip.so (include/nmf_hip.h).  This is the ONLY way the host-side operator
classes reach the GPU kernels; there is no CPU or PyTorch fallback: if the library is missing the
import of any product operator raises.

torch is used for what it is here for: device memory (tensors), the current HIP stream and
torch.distributed.  Every wrapper takes torch tensors, checks dtype / contiguity / device and
passes raw pointers + sizes + the current stream to the C ABI.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NMF_HIP_LIB") or os.path.join(_HERE, "lib", "libnmf_hip.so")      # (NMF_HIP_LIB: kernel experiments, tools/)


class NmfHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise NmfHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or nmf_amd/csrc/build.sh).  nmf_amd has no CPU fallback.")
    return C.CDLL(LIB_PATH)


_lib = _load()

c_f32p = C.POINTER(C.c_float)
c_vp = C.c_void_p


class MarchParams(C.Structure):
    _fields_ = [("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3), ("alpha_inv", C.c_float * 3),
                ("stepsize", C.c_float), ("half_step", C.c_float), ("near_t", C.c_float), ("far_t", C.c_float),
                ("focal", C.c_float), ("n_steps", C.c_int32), ("grid", C.c_int32 * 3), ("is_train", C.c_int32),
                ("seed", C.c_uint64), ("offset", C.c_uint64), ("occ_min", C.c_float * 3), ("occ_max", C.c_float * 3)]


class VmParams(C.Structure):
    _fields_ = [("aabb_min", C.c_float * 3), ("inv_size", C.c_float * 3), ("density_shift", C.c_float),
                ("grid", C.c_int32), ("stencil", C.c_float * 5), ("stencil_off", C.c_float * 5)]


class AdamSlot(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("numel", C.c_int64), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("weight_decay", C.c_double), ("step_size", C.c_double), ("bc2_sqrt", C.c_double),
                ("is_f64", C.c_int32), ("reserved", C.c_int32)]


class CopySlot(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("numel", C.c_int64), ("src_is_f64", C.c_int32),
                ("dst_is_f64", C.c_int32)]


EXPORTS = [
    "nmf_version", "nmf_last_error_string", "nmf_event_create", "nmf_event_create_timed", "nmf_event_elapsed_ms", "nmf_event_destroy", "nmf_event_record", "nmf_event_synchronize",
    "nmf_stream_wait_event", "nmf_memcpy_d2h_async", "nmf_host_alloc_mapped", "nmf_host_free_mapped", "nmf_publish_i64x2", "nmf_wait_seq", "nmf_set_launch_probe", "nmf_alpha_pack", "nmf_march_count", "nmf_march_scan", "nmf_march_scan_workspace_bytes", "nmf_march_scan_publish", "nmf_bounce_index_publish", "nmf_bounce_index_live", "nmf_vm_query_fwd_live",
    "nmf_march_fill", "nmf_march_dense", "nmf_vm_pack_density", "nmf_vm_query_fwd", "nmf_vm_query_fwd_bf16", "nmf_vm_query_rows", "nmf_vm_query_sigma", "nmf_sat_lookup_bwd_dirs", "nmf_vm_query_bwd",
    "nmf_vm_unpack_density_grad", "nmf_vm_bwd_workspace_bytes", "nmf_composite_fwd", "nmf_composite_bwd", "nmf_segment_sum",
    "nmf_sat_build", "nmf_sat_build_bwd", "nmf_sat_lookup_fwd", "nmf_sat_lookup_bwd", "nmf_sat_lookup_bwd_binned",
    "nmf_sat_lookup_bwd_workspace_bytes",
    "nmf_select_bounces", "nmf_select_total", "nmf_view_adjoint_to_rays", "nmf_expand_segments", "nmf_segment_sum_wide",
    "nmf_brdf_mlp_fwd", "nmf_brdf_mlp_bwd", "nmf_brdf_mlp_bwd_workspace_bytes", "nmf_brdf_mlp_image_bytes", "nmf_brdf_mlp_pack",
    "nmf_brdf_mlp_fwd_packed", "nmf_brdf_mlp_bwd_packed", "nmf_brdf_mlp_bwd_segments", "nmf_brdf_mlp_bwd_segments_workspace_bytes", "nmf_heads_fwd", "nmf_heads_bwd", "nmf_ggx_rays_fwd", "nmf_ggx_rays_bwd", "nmf_ggx_rays_bwd_view", "nmf_ggx_prob", "nmf_shade_mix_fwd", "nmf_shade_mix_bwd", "nmf_shade_mix_bwd_view",
    "nmf_adam_step", "nmf_adam_step_guarded", "nmf_bounce_index", "nmf_bounce_index_workspace_bytes", "nmf_bounce_prep_fwd", "nmf_bounce_prep_bwd",
    "nmf_ray_compose_fwd", "nmf_ray_compose_bwd", "nmf_l1_mean_fwd", "nmf_l1_mean_bwd", "nmf_sqerr_fwd", "nmf_sqerr_bwd",
    "nmf_loss_mix_fwd", "nmf_loss_mix_bwd", "nmf_loss_head", "nmf_loss_head_workspace_bytes", "nmf_bg_adjoint", "nmf_vm_query_bwd_segments", "nmf_vm_query_bwd_segments_clean", "nmf_vm_bwd_clean_bytes", "nmf_vm_unpack_density_grad_l1", "nmf_vm_bin_plan", "nmf_vm_bin_plan_bytes", "nmf_vm_walk_workspace_bytes", "nmf_vm_query_bwd_planned", "nmf_sh_project",
    "nmf_retrace_scores", "nmf_argsort_f32", "nmf_argsort_workspace_bytes", "nmf_topk_select", "nmf_topk_select_workspace_bytes", "nmf_alpha_coarse", "nmf_alpha_coarse_words", "nmf_bounce_index_select", "nmf_bounce_prep_fwd_heads", "nmf_bounce_prep_heads_bwd", "nmf_multi_copy",
]
for _n in EXPORTS:
    if not hasattr(_lib, _n):
        raise NmfHipError(f"libnmf_hip.so does not export {_n}")


def _declare_from_header():
    """argtypes / restype of every entry point, derived from include/nmf_hip.h (pointers and arrays -> void*, so wrappers
    pass Tensor.data_ptr() integers straight through; scalars are converted by ctypes)."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "nmf_hip.h")
    if not os.path.exists(hdr):
        # without prototypes ctypes would pass Tensor.data_ptr() integers as 32-bit C ints (truncated device pointers)
        raise NmfHipError(f"{hdr} is missing: the C-ABI prototypes of libnmf_hip.so are read from it (keep include/ next "
                          "to the nmf_amd package)")
    text = re.sub(r"/\*.*?\*/", " ", open(hdr).read(), flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    scal = {"int64_t": C.c_int64, "int32_t": C.c_int32, "int": C.c_int32, "uint64_t": C.c_uint64, "uint32_t": C.c_uint32,
            "float": C.c_float, "double": C.c_double}
    for ret, name, args in re.findall(r"\b(int64_t|int|const char\s*\*)\s+(nmf_\w+)\s*\(([^;{}]*?)\)\s*;", text):
        fn = getattr(_lib, name, None)
        if fn is None:
            continue
        fn.restype = C.c_int64 if ret == "int64_t" else (C.c_char_p if "char" in ret else C.c_int)
        types = []
        args = re.sub(r"\(\s*\*\s*(\w+)\s*\)\s*\([^)]*\)", r"*\1", args)      # a function-pointer parameter is a pointer
        for a in [x.strip() for x in args.split(",")]:
            if a in ("void", ""):
                continue
            if "*" in a or "[" in a:
                types.append(C.c_void_p)
            else:
                tok = [t for t in a.replace("const", " ").split() if t in scal]
                if not tok:
                    raise NmfHipError(f"cannot derive the ctypes type of '{a}' in {name}")
                types.append(scal[tok[0]])
        fn.argtypes = types


_declare_from_header()
_lib.nmf_last_error_string.restype = C.c_char_p
_lib.nmf_version.restype = C.c_int
_lib.nmf_vm_bwd_workspace_bytes.restype = C.c_int64
_lib.nmf_vm_bin_plan_bytes.restype = C.c_int64
_lib.nmf_vm_walk_workspace_bytes.restype = C.c_int64
_lib.nmf_march_scan_workspace_bytes.restype = C.c_int64
_lib.nmf_bounce_index_workspace_bytes.restype = C.c_int64
_lib.nmf_argsort_workspace_bytes.restype = C.c_int64
_lib.nmf_topk_select_workspace_bytes.restype = C.c_int64
_lib.nmf_alpha_coarse_words.restype = C.c_int64
_lib.nmf_sat_lookup_bwd_workspace_bytes.restype = C.c_int64


def version():
    return _lib.nmf_version()


def _check(code, what):
    if code != 0:
        raise NmfHipError(f"{what} failed: {_lib.nmf_last_error_string().decode()} [{code}]")


def _stream():
    # raw handle of torch's current stream on the current device (what torch.cuda.current_stream().cuda_stream
    # returns, without building the Stream object: this runs ~100 times per training step)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _p(t, dtype=None):
    """device pointer (integer) of a contiguous tensor; None -> NULL"""
    if t is None:
        return None
    if dtype is not None and t.dtype != dtype:
        raise NmfHipError(f"expected {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise NmfHipError("nmf_amd operators need device tensors (no CPU path)")
    if not t.is_contiguous():
        raise NmfHipError("tensor must be contiguous")
    return t.data_ptr() or None


def _p3(ts, dtype=torch.float32):
    arr = (C.c_void_p * 3)()
    if ts is None:
        return None
    for i in range(3):
        arr[i] = _p(ts[i], dtype)
    return arr


def channels_last_ptr_ok(t):
    """True if a [1,C,H,W] tensor is stored [H][W][C] densely (what the kernels index)."""
    _, Cn, H, W = t.shape
    return t.stride(1) == 1 and t.stride(3) == Cn and t.stride(2) == W * Cn


_host_mirror = {}


def host(t):
    """numpy copy of a small tensor, cached per (storage, version): geometry constants (aabb, grid size, step size)
    live in module buffers on the device; reading them back costs a stream sync each, so it is done once per value."""
    if not isinstance(t, torch.Tensor):
        return np.asarray(t)
    if not t.is_cuda:
        return t.detach().numpy()
    key = (t.data_ptr(), t._version, tuple(t.shape), t.dtype)
    v = _host_mirror.get(id(t))
    if v is None or v[0] != key:
        if len(_host_mirror) > 256:
            _host_mirror.clear()
        v = (key, t.detach().cpu().numpy(), t)          # keeps t alive so id(t) stays unique
        _host_mirror[id(t)] = v
    return v[1]


class Readback:
    """A few int64 sizes from the device to the host, split into start() and get() so that work queued in between runs
    while the numbers travel.  Pinned staging buffers from a small ring per device (at most two are ever in flight)."""
    _ring = {}

    def __init__(self):
        self.pin = torch.empty(4, dtype=torch.int64).pin_memory()
        self.ev = torch.cuda.Event()
        self.n = 0

    @classmethod
    def of(cls, dev):
        r = cls._ring.setdefault(str(dev), [[cls() for _ in range(4)], 0])
        r[1] = (r[1] + 1) & 3
        return r[0][r[1]]

    def start(self, t):
        self.n = t.numel()
        self.pin[: self.n].copy_(t, non_blocking=True)
        self.ev.record()
        return self

    def get(self):
        self.ev.synchronize()
        return self.pin[: self.n].tolist()


# ---- sampler ----------------------------------------------------------------------------------
def march_params(aabb, alpha_inv, stepsize, near, far, focal, n_steps, grid, is_train, seed=0, offset=0, occ_box=None):
    """occ_box: optional ((x0,y0,z0), (x1,y1,z1)) world box outside of which the alpha mask cannot keep a step"""
    p = MarchParams()
    if occ_box is not None:
        p.occ_min[:] = [float(v) for v in occ_box[0]]
        p.occ_max[:] = [float(v) for v in occ_box[1]]
    else:
        p.occ_min[:] = [1.0, 1.0, 1.0]
        p.occ_max[:] = [-1.0, -1.0, -1.0]
    a = host(aabb).astype(np.float32)
    p.aabb_min[:] = a[0].tolist()
    p.aabb_max[:] = a[1].tolist()
    p.alpha_inv[:] = np.asarray(alpha_inv, dtype=np.float32).tolist() if alpha_inv is not None else [0, 0, 0]
    st = np.float32(stepsize)
    p.stepsize = float(st)
    p.half_step = float(np.float32(st / np.float32(2)))
    p.near_t = float(np.float32(near))
    p.far_t = float(np.float32(far))
    p.focal = float(np.float32(focal))
    p.n_steps = int(n_steps)
    p.grid[:] = [int(g) for g in grid] if grid is not None else [0, 0, 0]
    p.is_train = 1 if is_train else 0
    p.seed = int(seed)
    p.offset = int(offset)
    return p


def alpha_pack(volume):
    n = volume.numel()
    bits = torch.zeros((n + 31) // 32, dtype=torch.int32, device=volume.device)
    _check(_lib.nmf_alpha_pack(_p(volume.contiguous(), torch.float32), C.c_int64(n), _p(bits), _stream()), "nmf_alpha_pack")
    return bits


def alpha_coarse(bits, grid):
    """coarse occupancy mask (one bit per 8^3 voxels) for nmf_march_count; grid = (gx, gy, gz) of the alpha volume"""
    g = (C.c_int32 * 3)(*[int(v) for v in grid])
    words = _lib.nmf_alpha_coarse_words(g)
    coarse = torch.zeros(max(words, 1), dtype=torch.int32, device=bits.device)
    _check(_lib.nmf_alpha_coarse(_p(bits, torch.int32), g, _p(coarse), _stream()), "nmf_alpha_coarse")
    return coarse


def march_count(p, rays, jitter, alpha_bits, alpha_coarse=None):
    B = rays.shape[0]
    W = (p.n_steps + 63) // 64
    valid = torch.empty((B, W), dtype=torch.int64, device=rays.device)
    counts = torch.empty(B, dtype=torch.int32, device=rays.device)
    _check(_lib.nmf_march_count(C.byref(p), _p(rays, torch.float32), C.c_int64(B), _p(jitter), _p(alpha_bits),
                                _p(alpha_coarse if alpha_bits is not None else None), _p(valid), _p(counts), _stream()),
           "nmf_march_count")
    return valid, counts


def march_scan(counts, max_samples):
    B = counts.shape[0]
    offsets = torch.empty(B + 1, dtype=torch.int64, device=counts.device)
    whole_valid = torch.empty(B, dtype=torch.uint8, device=counts.device)
    totals = torch.empty(2, dtype=torch.int64, device=counts.device)
    nbytes = _lib.nmf_march_scan_workspace_bytes(C.c_int64(B))
    ws = torch.empty(nbytes // 8, dtype=torch.int64, device=counts.device)
    _check(_lib.nmf_march_scan(_p(counts, torch.int32), C.c_int64(B), C.c_int64(max_samples), _p(offsets),
                               _p(whole_valid), _p(totals), _p(ws), C.c_int64(nbytes), _stream()), "nmf_march_scan")
    return offsets, whole_valid, totals


def march_fill(p, rays, b, M, jitter, valid, offsets, want_z=True):
    dev = rays.device
    xyzt = torch.empty((M, 4), dtype=torch.float32, device=dev)
    ray_id = torch.empty(M, dtype=torch.int32, device=dev)
    step_id = torch.empty(M, dtype=torch.int32, device=dev)
    z = torch.empty(M, dtype=torch.float32, device=dev) if want_z else None
    dist = torch.empty(M, dtype=torch.float32, device=dev)
    _check(_lib.nmf_march_fill(C.byref(p), _p(rays, torch.float32), C.c_int64(b), _p(jitter), _p(valid), _p(offsets),
                               _p(xyzt), _p(ray_id), _p(step_id), _p(z), _p(dist), _stream()), "nmf_march_fill")
    return xyzt, ray_id, step_id, z, dist


def march_dense(p, rays, b, jitter, valid):
    ray_valid = torch.empty((b, p.n_steps), dtype=torch.uint8, device=rays.device)
    z_vals = torch.empty((b, p.n_steps), dtype=torch.float32, device=rays.device)
    _check(_lib.nmf_march_dense(C.byref(p), _p(rays, torch.float32), C.c_int64(b), _p(jitter), _p(valid),
                                _p(ray_valid), _p(z_vals), _stream()), "nmf_march_dense")
    return ray_valid.bool(), z_vals


# ---- VM field ----------------------------------------------------------------------------------
def derivative_stencil_rows():
    """Rows of the 5x5 x-derivative stencil of GridSampler2D.backward (smoothing=1): a normalised 3x3
    Gaussian (std 1) correlated with [-0.5, 0, 0.5]; computed in fp32 exactly like the reference does
    (modules/grid_sample_Cinf.py:16-29,49-63,218-233): kx[i][j] = 0.5*(S[i-1][j-2] - S[i-1][j])."""
    n = np.arange(3, dtype=np.float32) - np.float32(1.0)
    g1 = np.exp(-(n ** 2) / np.float32(2.0)).astype(np.float32)
    S = np.outer(g1, g1).astype(np.float32)
    S = (S / S.sum(dtype=np.float32)).astype(np.float32)
    rows = []
    for i in (1, 2):          # S row 0 (== row 2) and S row 1
        r = S[i - 1]
        rows.append([-0.5 * r[0], -0.5 * r[1], 0.0, 0.5 * r[1], 0.5 * r[2]])
    return np.asarray(rows[1], np.float32), np.asarray(rows[0], np.float32)   # centre row, off-centre rows


def vm_params(aabb, inv_size, density_shift, grid):
    p = VmParams()
    p.aabb_min[:] = host(aabb).astype(np.float32)[0].tolist()
    p.inv_size[:] = host(inv_size).astype(np.float32).tolist()
    p.density_shift = float(density_shift)
    p.grid = int(grid)
    c, o = derivative_stencil_rows()
    p.stencil[:] = c.tolist()
    p.stencil_off[:] = o.tolist()
    return p


def vm_pack_density(p, planes, lines, out=None):
    """planes[i]: [G,G,16] channel-last storage, lines[i]: [G,16].  out = (dpk, dlk) of an earlier call re-packs in place."""
    G = p.grid
    dev = planes[0].device
    if out is not None:
        dpk, dlk = out
    else:
        dpk = [torch.empty((G, G, 48), dtype=torch.float32, device=dev) for _ in range(3)]
        dlk = [torch.empty((G, 32), dtype=torch.float32, device=dev) for _ in range(3)]
    _check(_lib.nmf_vm_pack_density(C.byref(p), _p3(planes), _p3(lines), _p3(dpk), _p3(dlk), _stream()),
           "nmf_vm_pack_density")
    return dpk, dlk


def vm_query_fwd(p, xyzt, dpk, dlk, app_planes, app_lines, basis, want_density=True, want_normal=True,
                 want_app=True, want_coef=False):
    M = xyzt.shape[0]
    dev = xyzt.device
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
    sf = f(M) if want_density else None
    sg = f(M) if want_density else None
    gr = f(M, 3) if want_normal else None
    nr = f(M, 3) if want_normal else None
    ap = f(M, 24) if want_app else None
    cf = f(M, 72) if want_coef else None
    need_d = want_density or want_normal
    need_a = want_app or want_coef
    probe = dpk[0] if (need_d and dpk is not None) else app_planes[0]
    if probe.dtype == torch.bfloat16:          # bf16-table variant: same kernel, half the bytes per tap
        fn, name, td = _lib.nmf_vm_query_fwd_bf16, "nmf_vm_query_fwd_bf16", torch.bfloat16
    else:
        fn, name, td = _lib.nmf_vm_query_fwd, "nmf_vm_query_fwd", torch.float32
    _check(fn(C.byref(p), _p(xyzt, torch.float32), C.c_int64(M),
              _p3(dpk, td) if need_d else None, _p3(dlk, td) if need_d else None,
              _p3(app_planes, td) if need_a else None, _p3(app_lines, td) if need_a else None,
              _p(basis) if need_a else None, _p(sf), _p(sg), _p(gr), _p(nr), _p(ap), _p(cf), _stream()), name)
    return sf, sg, gr, nr, ap, cf


def vm_query_rows(p, xyzt, dpk, dlk):
    """value + gradient + normal of a few rows (16 lanes per row): -> (sigma_feat [M], grad [M,3], normal [M,3])"""
    M = xyzt.shape[0]
    dev = xyzt.device
    sf = torch.empty(M, dtype=torch.float32, device=dev)
    gr = torch.empty((M, 3), dtype=torch.float32, device=dev)
    nr = torch.empty((M, 3), dtype=torch.float32, device=dev)
    td = dpk[0].dtype
    _check(_lib.nmf_vm_query_rows(C.byref(p), _p(xyzt, torch.float32), C.c_int64(M), _p3(dpk, td), _p3(dlk, td),
                                  C.c_int32(1 if td == torch.bfloat16 else 0), _p(sf), None, _p(gr), _p(nr), _stream()),
           "nmf_vm_query_rows")
    return sf, gr, nr


def vm_query_sigma(p, xyzt, planes, lines):
    """density value of all samples from the density factors themselves (planes [G,G,16], lines [G,16], fp32 or bfloat16):
    -> (sigma_feat [M], sigma [M]), the bits of vm_query_fwd"""
    M = xyzt.shape[0]
    sf = torch.empty(M, dtype=torch.float32, device=xyzt.device)
    sg = torch.empty(M, dtype=torch.float32, device=xyzt.device)
    td = planes[0].dtype
    for t in list(planes) + list(lines):
        if t.shape[-1] != 16:
            raise NmfHipError("vm_query_sigma: density factors [G,G,16] / [G,16] expected")
    _check(_lib.nmf_vm_query_sigma(C.byref(p), _p(xyzt, torch.float32), C.c_int64(M), _p3(planes, td), _p3(lines, td),
                                   C.c_int32(1 if td == torch.bfloat16 else 0), _p(sf), _p(sg), _stream()),
           "nmf_vm_query_sigma")
    return sf, sg


def to_bf16_tables(srcs, dsts=None):
    """fp32 tables -> bfloat16 copies in ONE launch (nmf_multi_copy, round to nearest even); dsts = the list of an earlier
    call refreshes the copies in place."""
    if dsts is None:
        dsts = [torch.empty(t.shape, dtype=torch.bfloat16, device=t.device) for t in srcs]
    n = len(srcs)
    slots = (CopySlot * n)()
    for i, (a, b) in enumerate(zip(srcs, dsts)):
        if a.dtype != torch.float32 or b.dtype != torch.bfloat16 or a.numel() != b.numel():
            raise NmfHipError("to_bf16_tables: fp32 sources and bf16 destinations of equal size")
        _p(a), _p(b)                                  # device / contiguity checks
        slots[i] = CopySlot(a.data_ptr(), b.data_ptr(), a.numel(), 0, 2)
    multi_copy(slots, n)
    return dsts


def vm_query_bwd(p, xyzt, dpk, dlk, app_planes, app_lines, basis, sigma_feat, grad, d_sigma, d_sigma_feat,
                 d_normal, d_app, g_dpk, g_dlk, g_app_planes, g_app_lines, g_basis=None):
    M = xyzt.shape[0]
    want_d = d_sigma is not None or d_sigma_feat is not None or d_normal is not None
    want_a = d_app is not None
    nbytes = _lib.nmf_vm_bwd_workspace_bytes(C.c_int64(M), C.c_int32(p.grid))
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=xyzt.device)
    _check(_lib.nmf_vm_query_bwd(C.byref(p), _p(xyzt, torch.float32), C.c_int64(M),
                                 _p3(dpk) if want_d else None, _p3(dlk) if want_d else None,
                                 _p3(app_planes) if want_a else None, _p3(app_lines) if want_a else None,
                                 _p(basis) if want_a else None, _p(sigma_feat), _p(grad), _p(d_sigma),
                                 _p(d_sigma_feat), _p(d_normal), _p(d_app),
                                 _p3(g_dpk) if want_d else None, _p3(g_dlk) if want_d else None,
                                 _p3(g_app_planes) if want_a else None, _p3(g_app_lines) if want_a else None,
                                 _p(g_basis if want_a else None), _p(ws), C.c_int64(nbytes), _stream()), "nmf_vm_query_bwd")


class VmBwdSegment(C.Structure):
    _fields_ = [("xyzt", C.c_void_p), ("M", C.c_int64), ("sigma_feat", C.c_void_p), ("grad", C.c_void_p),
                ("d_sigma", C.c_void_p), ("d_sigma_feat", C.c_void_p), ("d_normal", C.c_void_p), ("d_app", C.c_void_p)]


VM_MAX_SEGMENTS = 4


def vm_bin_plan(p, xyzts):
    """The brick sort of a backward walk over the sample sets `xyzts` ([M_i,4] each), from the positions alone -> opaque plan
    tensor for vm_query_bwd_segments(..., plan=) over the same sets in the same order (a training pass builds it under its
    forward; the backward then only permutes the adjoints)."""
    n = len(xyzts)
    if n == 0 or n > VM_MAX_SEGMENTS:
        raise NmfHipError(f"1..{VM_MAX_SEGMENTS} segments per plan")
    ptrs = (C.c_void_p * n)(*[_p(x, torch.float32) for x in xyzts])
    Ms = (C.c_int64 * n)(*[int(x.shape[0]) for x in xyzts])
    M = sum(int(x.shape[0]) for x in xyzts)
    nbytes = _lib.nmf_vm_bin_plan_bytes(C.c_int64(M), C.c_int32(p.grid))
    plan = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=xyzts[0].device)
    _check(_lib.nmf_vm_bin_plan(C.byref(p), ptrs, Ms, C.c_int32(n), _p(plan), C.c_int64(nbytes), _stream()), "nmf_vm_bin_plan")
    return plan


def vm_bwd_clean_scratch(p, device):
    """the zeroed scratch a caller keeps between walks (vm_query_bwd_segments(..., clean=...)): counters that every walk hands back zero"""
    return torch.zeros(int(_lib.nmf_vm_bwd_clean_bytes(C.c_int32(p.grid))), dtype=torch.uint8, device=device)


def vm_query_bwd_segments(p, segs, dpk, dlk, app_planes, app_lines, basis, g_dpk, g_dlk, g_app_planes, g_app_lines,
                          g_basis=None, plan=None, clean=None):
    """One backward walk over several sample sets (no concatenation).  segs: list of tuples
    (xyzt, sigma_feat, grad, d_sigma, d_sigma_feat, d_normal, d_app) -- the argument order of vm_query_bwd.
    plan: vm_bin_plan of the same sample sets (the sort is then not redone).  clean: vm_bwd_clean_scratch (no memset launches)."""
    n = len(segs)
    if n > VM_MAX_SEGMENTS:
        raise NmfHipError(f"at most {VM_MAX_SEGMENTS} segments per walk")
    arr = (VmBwdSegment * max(n, 1))()
    M, want_d, want_a = 0, False, False
    for i, (xyzt, sf, gr, ds, dsf, dn, da) in enumerate(segs):
        arr[i] = VmBwdSegment(_p(xyzt, torch.float32), xyzt.shape[0], _p(sf), _p(gr), _p(ds), _p(dsf), _p(dn), _p(da))
        M += xyzt.shape[0]
        want_d = want_d or ds is not None or dsf is not None or dn is not None
        want_a = want_a or da is not None
    if M == 0:
        return
    if plan is not None:
        nbytes = _lib.nmf_vm_walk_workspace_bytes(C.c_int64(M))
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=segs[0][0].device)
        _check(_lib.nmf_vm_query_bwd_planned(C.byref(p), arr, C.c_int32(n),
                                             _p3(dpk) if want_d else None, _p3(dlk) if want_d else None,
                                             _p3(app_planes) if want_a else None, _p3(app_lines) if want_a else None,
                                             _p(basis) if want_a else None,
                                             _p3(g_dpk) if want_d else None, _p3(g_dlk) if want_d else None,
                                             _p3(g_app_planes) if want_a else None, _p3(g_app_lines) if want_a else None,
                                             _p(g_basis if want_a else None), _p(plan), C.c_int64(plan.numel() * 4), _p(ws),
                                             C.c_int64(nbytes), _stream()),
               "nmf_vm_query_bwd_planned")
        return
    nbytes = _lib.nmf_vm_bwd_workspace_bytes(C.c_int64(M), C.c_int32(p.grid))
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=segs[0][0].device)
    if clean is not None:
        _check(_lib.nmf_vm_query_bwd_segments_clean(C.byref(p), arr, C.c_int32(n),
                                                    _p3(dpk) if want_d else None, _p3(dlk) if want_d else None,
                                                    _p3(app_planes) if want_a else None, _p3(app_lines) if want_a else None,
                                                    _p(basis) if want_a else None,
                                                    _p3(g_dpk) if want_d else None, _p3(g_dlk) if want_d else None,
                                                    _p3(g_app_planes) if want_a else None, _p3(g_app_lines) if want_a else None,
                                                    _p(g_basis if want_a else None), _p(clean), C.c_int64(clean.numel()), _p(ws),
                                                    C.c_int64(nbytes), _stream()),
               "nmf_vm_query_bwd_segments_clean")
        return
    _check(_lib.nmf_vm_query_bwd_segments(C.byref(p), arr, C.c_int32(n),
                                          _p3(dpk) if want_d else None, _p3(dlk) if want_d else None,
                                          _p3(app_planes) if want_a else None, _p3(app_lines) if want_a else None,
                                          _p(basis) if want_a else None,
                                          _p3(g_dpk) if want_d else None, _p3(g_dlk) if want_d else None,
                                          _p3(g_app_planes) if want_a else None, _p3(g_app_lines) if want_a else None,
                                          _p(g_basis if want_a else None), _p(ws), C.c_int64(nbytes), _stream()),
           "nmf_vm_query_bwd_segments")


def vm_unpack_density_grad(p, g_dpk, g_dlk, out=None, l1=None):
    """out = (gp, gl) of an earlier call: the same tensors are overwritten (a training pass keeps its gradient tensors).
    l1 = (the six density parameters [planes + lines] in the gradients' storage order, 0-d device scale): the gradient of
    scale * sum_i mean |x_i| is added in the same launch (same bits as l1_mean_bwd(..., out=gp + gl) behind the unpack)"""
    G = p.grid
    dev = g_dpk[0].device
    if out is not None:
        gp, gl = out
    else:
        # parameter-shaped outputs with channel-last strides: the storage is the [G][G][16] / [G][16] the kernel writes
        gp = [torch.empty((1, 16, G, G), dtype=torch.float32, device=dev).contiguous(memory_format=torch.channels_last)
              for _ in range(3)]
        gl = [torch.empty((1, 16, G, 1), dtype=torch.float32, device=dev).contiguous(memory_format=torch.channels_last)
              for _ in range(3)]
    arr_p, arr_l = (C.c_void_p * 3)(*[t.data_ptr() for t in gp]), (C.c_void_p * 3)(*[t.data_ptr() for t in gl])
    if l1 is not None:
        xs, scale = l1
        for x, g in zip(xs, gp + gl):
            if x.numel() != g.numel():
                raise NmfHipError("vm_unpack_density_grad: parameter / gradient size mismatch")
        xp = (C.c_void_p * 3)(*[_dense_f32(t) for t in xs[:3]])
        xl = (C.c_void_p * 3)(*[_dense_f32(t) for t in xs[3:]])
        _check(_lib.nmf_vm_unpack_density_grad_l1(C.byref(p), _p3(g_dpk), _p3(g_dlk), arr_p, arr_l, xp, xl, _p(scale, torch.float32),
                                                  _stream()), "nmf_vm_unpack_density_grad_l1")
        return gp, gl
    _check(_lib.nmf_vm_unpack_density_grad(C.byref(p), _p3(g_dpk), _p3(g_dlk), arr_p, arr_l, _stream()),
           "nmf_vm_unpack_density_grad")
    return gp, gl


# ---- compositing --------------------------------------------------------------------------------
def composite_fwd(sigma, dist, offsets, b, distance_scale):
    M = sigma.shape[0]
    weight = torch.empty(M, dtype=torch.float32, device=sigma.device)
    acc = torch.empty(b, dtype=torch.float32, device=sigma.device)
    if M == 0:                      # no kept sample at all: nothing to launch (empty tensors have no storage)
        return weight, acc.zero_()
    _check(_lib.nmf_composite_fwd(_p(sigma, torch.float32), _p(dist, torch.float32), _p(offsets, torch.int64),
                                  C.c_int64(b), C.c_float(distance_scale), _p(weight), _p(acc), _stream()),
           "nmf_composite_fwd")
    return weight, acc


def composite_bwd(sigma, dist, weight, offsets, b, distance_scale, d_weight):
    d_sigma = torch.empty_like(sigma)
    if sigma.shape[0] == 0:
        return d_sigma
    _check(_lib.nmf_composite_bwd(_p(sigma, torch.float32), _p(dist, torch.float32), _p(weight, torch.float32),
                                  _p(offsets, torch.int64), C.c_int64(b), C.c_float(distance_scale),
                                  _p(d_weight.contiguous(), torch.float32), _p(d_sigma), _stream()),
           "nmf_composite_bwd")
    return d_sigma


def segment_sum(vals, scale, offsets, n_seg, lanes=1):
    """lanes=1: index-order sums (bit-reproducible); lanes=8: eight lanes per segment (tree sum)"""
    D = vals.shape[1]
    out = torch.empty((n_seg, D), dtype=torch.float32, device=vals.device)
    if vals.shape[0] == 0:
        return out.zero_()
    _check(_lib.nmf_segment_sum(_p(vals, torch.float32), _p(scale), _p(offsets, torch.int64), C.c_int64(n_seg),
                                C.c_int32(D), C.c_int32(lanes), _p(out), _stream()), "nmf_segment_sum")
    return out


# ---- environment map ---------------------------------------------------------------------------
def sat_build(bg_mat, brightness=0.0, mul=1.0, sc=None, out=None, pole=False, interleaved=False):
    """bg_mat [1,3,H,W] or [3,H,W] -> (activated, sat) [3,H,W] (+ pole-row means [2,3] with pole=True, + the channel-
    interleaved copy of sat [H,W,4] with interleaved=True: what the lookups read fastest).  sc: optional device float32 [3] =
    (mipbias, brightness, mul) read by the kernels instead of the by-value scalars (no host read-back of the parameters).
    out = the tuple of an earlier call (same flags) rebuilds the tables in place."""
    bg = bg_mat.reshape(3, bg_mat.shape[-2], bg_mat.shape[-1])
    H, W = bg.shape[-2:]
    if out is not None:
        act, sat = out[0], out[1]
        pl = out[2] if pole else None
        s4 = out[-1] if interleaved else None
    else:
        act = torch.empty_like(bg)
        sat = torch.empty_like(bg)
        pl = torch.empty((2, 3), dtype=torch.float32, device=bg.device) if pole else None
        s4 = torch.zeros((H, W, 4), dtype=torch.float32, device=bg.device) if interleaved else None
    _check(_lib.nmf_sat_build(_p(bg.contiguous(), torch.float32), C.c_int32(H), C.c_int32(W), C.c_float(brightness),
                              C.c_float(mul), _p(sc), _p(act), _p(sat), _p(pl), _p(s4), _stream()), "nmf_sat_build")
    return (act, sat) + ((pl,) if pole else ()) + ((s4,) if interleaved else ())


def _sat_layout(sat):
    """-> (H, W, layout) of a summed-area table: planar [3,H,W] (0) or channel-interleaved [H,W,4] (1)"""
    if sat.dim() == 3 and sat.shape[-1] == 4 and sat.shape[0] != 3:
        return sat.shape[0], sat.shape[1], 1
    return sat.shape[-2], sat.shape[-1], 0


def sh_project(vals, wq, sh_A, out=None):
    """-> (coeffs [K,3], conv [K,3]); wq [n,K] contiguous, sh_A [>=K]"""
    n, K = wq.shape[0], wq.shape[1]
    if out is None:
        out = (torch.empty((K, 3), dtype=torch.float32, device=vals.device),
               torch.empty((K, 3), dtype=torch.float32, device=vals.device))
    _check(_lib.nmf_sh_project(_p(vals, torch.float32), _p(wq, torch.float32), C.c_int64(n), C.c_int32(K),
                               _p(sh_A, torch.float32), _p(out[0]), _p(out[1]), _stream()), "nmf_sh_project")
    return out


def sat_build_bwd(d_sat, bg_mat, act, d_pole, brightness=0.0, mul=1.0, sc=None, out=None):
    bg = bg_mat.reshape(3, bg_mat.shape[-2], bg_mat.shape[-1])
    H, W = bg.shape[-2:]
    d_bg = out if out is not None else torch.empty_like(bg)
    _check(_lib.nmf_sat_build_bwd(_p(d_sat, torch.float32), _p(bg.contiguous(), torch.float32), _p(act), C.c_int32(H),
                                  C.c_int32(W), C.c_float(brightness), C.c_float(mul), _p(sc), _p(d_pole), _p(d_bg), _stream()),
           "nmf_sat_build_bwd")
    return d_bg


def sat_lookup_fwd(sat, dirs, sa, mipbias, pole_rows, sc=None):
    """dirs: [R,3] directions, or [R,6] ray rows (origin | direction) looked up along their columns 3..5; sat: the planar
    table [3,H,W] or sat_build's interleaved copy [H,W,4] (same results)"""
    R, ld = dirs.shape[0], dirs.shape[1]
    H, W, layout = _sat_layout(sat)
    out = torch.empty((R, 3), dtype=torch.float32, device=dirs.device)
    _check(_lib.nmf_sat_lookup_fwd(_p(sat, torch.float32), C.c_int32(H), C.c_int32(W), _p(dirs, torch.float32),
                                   C.c_int32(ld), _p(sa, torch.float32), C.c_int64(R), C.c_float(mipbias), _p(sc), _p(pole_rows),
                                   C.c_int32(layout), _p(out), _stream()), "nmf_sat_lookup_fwd")
    return out


# lookups per call from which the table adjoint is binned (three more launches than the direct scatter; measured with
# tools/env_bwd_bench.py: 47 k lookups 53-73 us against 51 direct, 247 k lookups 138 against 176; in the training step the call
# sits on a side stream and the step time is the same either way, tools/ab_inprocess.py hip:ENV_BINNED_MIN_LOOKUPS).
# (a module constant: tests and tools/env_bwd_bench.py set it to compare the two forms)
# (from 30 k lookups: alone the two forms take the same time at 47 k -- 51 / 52 us -- but in the backward of a training step the leaf
# level's adjoint runs next to the binned adjoint of the level above, the value walk and the MLP backward, and the direct form's
# float atomics queue behind theirs at the memory side: 1.375 -> 1.361 ms per step)
ENV_BINNED_MIN_LOOKUPS = 30000


def _cdiv(a, b):
    return -(-a // b)


def sat_lookup_bwd(sat, dirs, sa, mipbias, d_out, d_sat, d_pole, d_mip=None, want_dirs=True, want_mipbias=None, sc=None):
    """d_sat [H,W,4] / d_pole [2,3] / d_mip [1] are ACCUMULATED into (any may be None except d_pole).  Returns d_dirs; with
    want_mipbias=True (legacy form) a fresh d_mip accumulator is allocated and (d_dirs, d_mip) is returned."""
    R, ld = dirs.shape[0], dirs.shape[1]
    H, W, layout = _sat_layout(sat)
    d_dirs = torch.empty((R, ld), dtype=torch.float32, device=dirs.device) if want_dirs else None   # shaped like dirs
    legacy = want_mipbias is not None
    if legacy and want_mipbias and d_mip is None:
        d_mip = torch.zeros(1, dtype=torch.float32, device=dirs.device)
    if d_sat is not None and R >= ENV_BINNED_MIN_LOOKUPS and _cdiv(H, 32) * _cdiv(W, 64) <= 1024:
        # many lookups: the binned table adjoint (csrc/env.hip).  The record pool comes from torch's stream-ordered caching
        # allocator per call: two of these calls may be in flight on different streams (fast_step's side streams)
        nbytes = int(_lib.nmf_sat_lookup_bwd_workspace_bytes(C.c_int64(R)))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dirs.device)
        _check(_lib.nmf_sat_lookup_bwd_binned(_p(sat, torch.float32), C.c_int32(H), C.c_int32(W), _p(dirs, torch.float32),
                                              C.c_int32(ld), _p(sa, torch.float32), C.c_int64(R), C.c_float(mipbias), _p(sc),
                                              C.c_int32(layout), _p(d_out.contiguous(), torch.float32), _p(d_sat), _p(d_pole),
                                              _p(d_dirs), _p(d_mip), _p(ws), C.c_int64(nbytes), _stream()),
               "nmf_sat_lookup_bwd_binned")
        return (d_dirs, d_mip) if legacy else d_dirs
    _check(_lib.nmf_sat_lookup_bwd(_p(sat, torch.float32), C.c_int32(H), C.c_int32(W), _p(dirs, torch.float32),
                                   C.c_int32(ld), _p(sa, torch.float32), C.c_int64(R), C.c_float(mipbias), _p(sc), C.c_int32(layout),
                                   _p(d_out.contiguous(), torch.float32), _p(d_sat), _p(d_pole), _p(d_dirs), _p(d_mip),
                                   _stream()), "nmf_sat_lookup_bwd")
    return (d_dirs, d_mip) if legacy else d_dirs


# ---- shading helpers -------------------------------------------------------------------------------
def select_bounces(weights, u, mode, mul, add=0.0, sum_w=1.0):
    """sum_w: python float, or a 0-d fp32 DEVICE tensor (read by the kernel: no host sync)."""
    M = weights.shape[0]
    counts = torch.empty(M, dtype=torch.int32, device=weights.device)
    dev_sum = sum_w if isinstance(sum_w, torch.Tensor) else None
    _check(_lib.nmf_select_bounces(_p(weights, torch.float32), _p(u, torch.float32), C.c_int64(M), C.c_int32(mode),
                                   C.c_float(mul), C.c_float(add), C.c_float(1.0 if dev_sum is not None else sum_w),
                                   _p(dev_sum, torch.float32), _p(counts), _stream()),
           "nmf_select_bounces")
    return counts


_select_ws = {}


def select_total_workspace(dev, stream=None):
    """the partial-sum / ticket workspace of nmf_select_total: one per (device, stream) -- launches on different streams (the
    chunk contexts of an optimizer step, nmf_amd/fast_step.py) must not share it"""
    key = (dev, (stream if stream is not None else torch.cuda.current_stream(dev)).cuda_stream)
    ws = _select_ws.get(key)
    if ws is None:
        ws = _select_ws[key] = torch.zeros(258, dtype=torch.float64, device=dev)       # zeroed once; the kernel resets it
    return ws


def select_total(weights, u, extra):
    """-> 0-d fp32 device tensor clip(float(sum(w) + 1e-3 * (sum(u) + extra)), 1e-3) (one launch, float64 sums)"""
    dev = weights.device
    ws = select_total_workspace(dev)
    total = torch.empty((), dtype=torch.float32, device=dev)
    _check(_lib.nmf_select_total(_p(weights, torch.float32), _p(u, torch.float32), C.c_int64(weights.shape[0]),
                                 C.c_double(float(extra)), _p(ws), _p(total), _stream()), "nmf_select_total")
    return total


def view_adjoint_to_rays(ray_id, bidx, dv_a, dv_b, d_rays):
    """d_rays [B,6] (columns 3..5) -= per-row view adjoints dv_a [Mb,>=3] (+ dv_b), rows may be column slices"""
    Mb = bidx.shape[0]
    (pa, la) = _rows(dv_a, 3)
    (pb, lb) = _rows(dv_b, 3) if dv_b is not None else (None, 3)
    _check(_lib.nmf_view_adjoint_to_rays(_p(ray_id, torch.int32), _p(bidx, torch.int32), pa, C.c_int32(la), pb, C.c_int32(lb),
                                         C.c_int64(Mb), _p(d_rays, torch.float32), _stream()), "nmf_view_adjoint_to_rays")


def expand_segments(offsets, n_seg, total):
    seg = torch.empty(total, dtype=torch.int32, device=offsets.device)
    loc = torch.empty(total, dtype=torch.int32, device=offsets.device)
    _check(_lib.nmf_expand_segments(_p(offsets, torch.int64), C.c_int64(n_seg), _p(seg), _p(loc), _stream()),
           "nmf_expand_segments")
    return seg, loc


def segment_sum_wide(vals, D, offsets, n_seg):
    out = torch.empty((n_seg, D), dtype=torch.float32, device=vals.device)
    if vals.shape[0] == 0:
        return out.zero_()
    _check(_lib.nmf_segment_sum_wide(_p(vals, torch.float32), C.c_int64(vals.shape[1]), C.c_int32(D),
                                     _p(offsets, torch.int64), C.c_int64(n_seg), _p(out), _stream()),
           "nmf_segment_sum_wide")
    return out


def brdf_mlp_pack(weights, into=None):
    """The six weight tensors as the packed image the fused MLP kernels copy into LDS (nmf_brdf_mlp_pack): built once per weight
    update, passed as `image=` to brdf_mlp_fwd / brdf_mlp_bwd.  into: a uint8 device tensor of nmf_brdf_mlp_image_bytes() to reuse."""
    n = int(_lib.nmf_brdf_mlp_image_bytes())
    img = into if into is not None else torch.empty(n, dtype=torch.uint8, device=weights[0].device)
    if img.dtype != torch.uint8 or not img.is_contiguous() or img.numel() < n:
        raise ValueError("brdf_mlp_pack: `into` must be a contiguous uint8 tensor of nmf_brdf_mlp_image_bytes() bytes")
    _check(_lib.nmf_brdf_mlp_pack(*[_p(w, torch.float32) for w in weights], _p(img), C.c_int64(img.numel()), _stream()),
           "nmf_brdf_mlp_pack")
    return img


def brdf_mlp_fwd(weights, half_vec, diff_vec, feat_src, rough_src, src_idx, out_bias, max_workgroups=0, with_mask=False, image=None):
    """weights = (W0 [64,66], b0, W2 [64,64], b2, W4 [4,64], b4); max_workgroups: see brdf_mlp_bwd.  with_mask: also return
    the ReLU masks [R, 4] (int32 storage) that brdf_mlp_bwd takes together with the output.  image: brdf_mlp_pack(weights) --
    the same bits, a shorter launch (`weights` is then not read)."""
    R = half_vec.shape[0]
    out = torch.empty((R, 3), dtype=torch.float32, device=half_vec.device)
    mask = torch.empty((R, 4), dtype=torch.int32, device=half_vec.device) if with_mask else None
    tail = (_p(half_vec, torch.float32), _p(diff_vec, torch.float32), _p(feat_src, torch.float32), _p(rough_src, torch.float32),
            _p(src_idx, torch.int32), C.c_int64(R), C.c_float(out_bias), _p(out), _p(mask, torch.int32), C.c_int32(max_workgroups),
            _stream())
    if image is not None:
        _check(_lib.nmf_brdf_mlp_fwd_packed(_p(image), *tail), "nmf_brdf_mlp_fwd_packed")
    else:
        _check(_lib.nmf_brdf_mlp_fwd(*[_p(w, torch.float32) for w in weights], *tail), "nmf_brdf_mlp_fwd")
    return (out, mask) if with_mask else out


def brdf_mlp_bwd(weights, half_vec, diff_vec, feat_src, rough_src, src_idx, fwd_out, act_mask, d_out, grads, max_workgroups=0,
                 image=None):
    """fwd_out, act_mask: what brdf_mlp_fwd(..., with_mask=True) returned for the same inputs.  grads: six fp32 tensors
    shaped like `weights`, ACCUMULATED into (caller zeroes them once per pass).  max_workgroups > 0 caps the persistent
    workgroups (a launch that shares the chip with kernels of another stream).  -> d_feat [rows of feat_src, 24]: the adjoint of
    feat_src, summed over the rays that gathered each row (src_idx must be non-decreasing)."""
    R = half_vec.shape[0]
    dev = half_vec.device
    d_feat = torch.zeros((feat_src.shape[0], 24), dtype=torch.float32, device=dev)
    nws = int(_lib.nmf_brdf_mlp_bwd_workspace_bytes(C.c_int64(R), C.c_int32(max_workgroups)))
    ws = torch.empty(max(nws, 4) // 4, dtype=torch.float32, device=dev)
    tail = (_p(half_vec, torch.float32), _p(diff_vec, torch.float32), _p(feat_src, torch.float32), _p(rough_src, torch.float32),
            _p(src_idx, torch.int32), C.c_int64(R), _p(fwd_out, torch.float32), _p(act_mask, torch.int32),
            _p(d_out.contiguous(), torch.float32), _p(d_feat), *[_p(g) for g in grads], C.c_int32(max_workgroups), _p(ws),
            C.c_int64(nws), _stream())
    if image is not None:
        _check(_lib.nmf_brdf_mlp_bwd_packed(_p(image), *tail), "nmf_brdf_mlp_bwd_packed")
    else:
        _check(_lib.nmf_brdf_mlp_bwd(*[_p(w, torch.float32) for w in weights], *tail), "nmf_brdf_mlp_bwd")
    return d_feat


class MlpBwdSegment(C.Structure):
    _fields_ = [("half_vec", C.c_void_p), ("diff_vec", C.c_void_p), ("feat_src", C.c_void_p), ("rough_src", C.c_void_p),
                ("src_idx", C.c_void_p), ("R", C.c_int64), ("fwd_out", C.c_void_p), ("act_mask", C.c_void_p), ("d_out", C.c_void_p),
                ("d_feat", C.c_void_p)]


def brdf_mlp_bwd_segments(weights, sets, grads, max_workgroups=0, image=None):
    """brdf_mlp_bwd over one or two ray sets in ONE launch (the evaluations of a level and of the level below share the weights).
    sets: tuples (half_vec, diff_vec, feat_src, rough_src, src_idx, fwd_out, act_mask, d_out) -> list of d_feat, one per set."""
    if not 1 <= len(sets) <= 2:
        raise ValueError("brdf_mlp_bwd_segments: one or two ray sets")
    dev = sets[0][0].device
    arr = (MlpBwdSegment * len(sets))()
    keep, outs = [], []
    for i, (hv, dv, feat, rough, idx, out, mask, d_out) in enumerate(sets):
        d_feat = torch.zeros((feat.shape[0], 24), dtype=torch.float32, device=dev)
        go = d_out.contiguous()
        keep.append(go)
        outs.append(d_feat)
        arr[i] = MlpBwdSegment(_p(hv, torch.float32), _p(dv, torch.float32), _p(feat, torch.float32), _p(rough, torch.float32),
                               _p(idx, torch.int32), hv.shape[0], _p(out, torch.float32), _p(mask, torch.int32), _p(go, torch.float32),
                               _p(d_feat))
    Rs = (C.c_int64 * len(sets))(*[s[0].shape[0] for s in sets])
    nws = int(_lib.nmf_brdf_mlp_bwd_segments_workspace_bytes(Rs, C.c_int32(len(sets)), C.c_int32(max_workgroups)))
    ws = torch.empty(max(nws, 4) // 4, dtype=torch.float32, device=dev)
    wp = [None] * 6 if image is not None else [_p(w, torch.float32) for w in weights]
    _check(_lib.nmf_brdf_mlp_bwd_segments(_p(image) if image is not None else None, *wp, arr, C.c_int32(len(sets)),
                                          *[_p(g) for g in grads], C.c_int32(max_workgroups), _p(ws), C.c_int64(nws), _stream()),
           "nmf_brdf_mlp_bwd_segments")
    return outs


def heads_fwd(feat, W, b, hp):
    """hp = (diffuse_mul, diffuse_bias, tint_bias, f0_bias, rough_bias)"""
    M = feat.shape[0]
    out = torch.empty((M, 11), dtype=torch.float32, device=feat.device)
    _check(_lib.nmf_heads_fwd(_p(feat, torch.float32), C.c_int64(M), _p(W, torch.float32), _p(b, torch.float32),
                              *[C.c_float(v) for v in hp], _p(out), _stream()), "nmf_heads_fwd")
    return out


def heads_bwd(feat, W, b, hp, d_out, gW, gb, add_into=None):
    """gW [11,24] / gb [11] are ACCUMULATED into.  add_into: another adjoint of the same rows (dense fp32 [M,24]); the result is
    added to it in place and it is returned (one launch less than `add_into += heads_bwd(...)`, the same bits)."""
    M = feat.shape[0]
    if add_into is not None and (add_into.dtype != torch.float32 or not add_into.is_contiguous() or add_into.numel() != feat.numel()):
        raise ValueError("heads_bwd: add_into must be a dense float32 [M,24] tensor")
    d_feat = torch.empty_like(feat) if add_into is None else add_into
    _check(_lib.nmf_heads_bwd(_p(feat, torch.float32), C.c_int64(M), _p(W, torch.float32), _p(b, torch.float32),
                              *[C.c_float(v) for v in hp], _p(d_out.contiguous(), torch.float32),
                              None if add_into is None else _p(add_into, torch.float32), _p(d_feat), _p(gW),
                              _p(gb), _stream()), "nmf_heads_bwd")
    return d_feat


def ggx_rays_fwd(V, N, r, x, off, cnt, sobol, row_of_ray, j_of_ray):
    R = row_of_ray.shape[0]
    dev = V.device
    f = lambda *s_: torch.empty(s_, dtype=torch.float32, device=dev)  # noqa: E731
    L, hl, dl, lpdf, mip, rays = f(R, 3), f(R, 3), f(R, 3), f(R), f(R), f(R, 6)
    _check(_lib.nmf_ggx_rays_fwd(_p(V, torch.float32), _p(N, torch.float32), _p(r, torch.float32), _p(x, torch.float32),
                                 _p(off, torch.float32), _p(cnt, torch.int32), _p(sobol, torch.float32),
                                 _p(row_of_ray, torch.int32), _p(j_of_ray, torch.int32), C.c_int64(R), _p(L), _p(hl),
                                 _p(dl), _p(lpdf), _p(mip), _p(rays), _stream()), "nmf_ggx_rays_fwd")
    return L, hl, dl, lpdf, mip, rays


def ggx_prob(dir_in, dir_out, half, rough):
    R = dir_in.shape[0]
    out = torch.empty((R,), dtype=torch.float32, device=dir_in.device)
    _check(_lib.nmf_ggx_prob(_p(dir_in, torch.float32), _p(dir_out, torch.float32), _p(half, torch.float32),
                             _p(rough, torch.float32), C.c_int64(R), _p(out), _stream()), "nmf_ggx_prob")
    return out


def ggx_rays_bwd(V, N, r, off, sobol, row_of_ray, j_of_ray, dL, d_rays=None):
    R = row_of_ray.shape[0]
    d_nr = torch.empty((R, 4), dtype=torch.float32, device=V.device)
    _check(_lib.nmf_ggx_rays_bwd(_p(V, torch.float32), _p(N, torch.float32), _p(r, torch.float32), _p(off, torch.float32),
                                 _p(sobol, torch.float32), _p(row_of_ray, torch.int32), _p(j_of_ray, torch.int32),
                                 C.c_int64(R), _p(dL), _p(d_rays), _p(d_nr), _stream()), "nmf_ggx_rays_bwd")
    return d_nr


def ggx_rays_bwd_view(V, N, r, off, sobol, row_of_ray, j_of_ray, dL, d_rays=None):
    """-> d_nrv [R,7] = per-ray adjoints of (normal | roughness | view direction)"""
    R = row_of_ray.shape[0]
    d = torch.empty((R, 7), dtype=torch.float32, device=V.device)
    _check(_lib.nmf_ggx_rays_bwd_view(_p(V, torch.float32), _p(N, torch.float32), _p(r, torch.float32), _p(off, torch.float32),
                                      _p(sobol, torch.float32), _p(row_of_ray, torch.int32), _p(j_of_ray, torch.int32),
                                      C.c_int64(R), _p(dL), _p(d_rays), _p(d), _stream()), "nmf_ggx_rays_bwd_view")
    return d


def shade_mix_bwd_view(V, f0, diff, cnt, row_of_ray, L, inc, brdf, d_rows):
    """shade_mix_bwd plus dV [R,3]"""
    R = row_of_ray.shape[0]
    dev = V.device
    f = lambda *s_: torch.empty(s_, dtype=torch.float32, device=dev)  # noqa: E731
    d_inc, d_brdf, dL, d_fd, dV = f(R, 3), f(R, 3), f(R, 3), f(R, 6), f(R, 3)
    _check(_lib.nmf_shade_mix_bwd_view(_p(V, torch.float32), _p(f0, torch.float32), _p(diff, torch.float32),
                                       _p(cnt, torch.int32), _p(row_of_ray, torch.int32), C.c_int64(R), _p(L, torch.float32),
                                       _p(inc, torch.float32), _p(brdf, torch.float32), _p(d_rows, torch.float32), _p(d_inc),
                                       _p(d_brdf), _p(dL), _p(d_fd), _p(dV), _stream()), "nmf_shade_mix_bwd_view")
    return d_inc, d_brdf, dL, d_fd, dV


def shade_mix_fwd(V, f0, diff, cnt, row_of_ray, L, inc, brdf):
    R = row_of_ray.shape[0]
    contrib = torch.empty((R, 3), dtype=torch.float32, device=V.device)
    _check(_lib.nmf_shade_mix_fwd(_p(V, torch.float32), _p(f0, torch.float32), _p(diff, torch.float32), _p(cnt, torch.int32),
                                  _p(row_of_ray, torch.int32), C.c_int64(R), _p(L, torch.float32), _p(inc, torch.float32),
                                  _p(brdf, torch.float32), _p(contrib), _stream()), "nmf_shade_mix_fwd")
    return contrib


def shade_mix_bwd(V, f0, diff, cnt, row_of_ray, L, inc, brdf, d_rows):
    R = row_of_ray.shape[0]
    dev = V.device
    d_inc = torch.empty((R, 3), dtype=torch.float32, device=dev)
    d_brdf = torch.empty((R, 3), dtype=torch.float32, device=dev)
    dL = torch.empty((R, 3), dtype=torch.float32, device=dev)
    d_fd = torch.empty((R, 6), dtype=torch.float32, device=dev)
    _check(_lib.nmf_shade_mix_bwd(_p(V, torch.float32), _p(f0, torch.float32), _p(diff, torch.float32), _p(cnt, torch.int32),
                                  _p(row_of_ray, torch.int32), C.c_int64(R), _p(L, torch.float32), _p(inc, torch.float32),
                                  _p(brdf, torch.float32), _p(d_rows, torch.float32), _p(d_inc), _p(d_brdf), _p(dL),
                                  _p(d_fd), _stream()), "nmf_shade_mix_bwd")
    return d_inc, d_brdf, dL, d_fd


# ---- optimizer ----------------------------------------------------------------------------------
def adam_step(slots, n, guard=None):
    """slots: (AdamSlot * k) host array, the first n entries are applied in one launch (nmf_adam_step); guard: optional 0-d fp32
    device tensor, a non-finite value turns the launch into a no-op"""
    _check(_lib.nmf_adam_step_guarded(slots, C.c_int32(n), _p(guard, torch.float32), _stream()), "nmf_adam_step")


# ---- shading glue --------------------------------------------------------------------------------
def bounce_index(counts, xyzt=None):
    """counts [M] int32 -> (bidx [M] int32, row_off [M+1] int64, cnt_rows [M] int32, inv [M] int32,
    totals [2] int64 = (R, Mb)); the caller slices bidx[:Mb] / row_off[:Mb+1] / cnt_rows[:Mb] once it has read totals.
    With xyzt [M,4]: a sixth output xyzt_rows [M,4] whose first Mb rows are xyzt[bidx]."""
    M = counts.shape[0]
    dev = counts.device
    rows = torch.empty((max(M, 1), 4), dtype=torch.float32, device=dev) if xyzt is not None else None
    bidx = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
    row_off = torch.empty(M + 1, dtype=torch.int64, device=dev)
    inv = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
    cnt_rows = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
    totals = torch.empty(2, dtype=torch.int64, device=dev)
    nbytes = _lib.nmf_bounce_index_workspace_bytes(C.c_int64(M))
    ws = torch.empty(nbytes // 8, dtype=torch.int64, device=dev)
    _check(_lib.nmf_bounce_index(_p(counts, torch.int32) if M else C.c_void_p(0), C.c_int64(M), _p(bidx), _p(row_off),
                                 _p(cnt_rows), _p(inv), _p(totals), _p(xyzt, torch.float32) if (xyzt is not None and M) else None,
                                 _p(rows), _p(ws), C.c_int64(nbytes), _stream()), "nmf_bounce_index")
    if xyzt is not None:
        return bidx, row_off, cnt_rows, inv[:M], totals, rows
    return bidx, row_off, cnt_rows, inv[:M], totals


def bounce_prep_fwd_heads(bidx, normals, app, head_W, head_b, hp, xyzt, ray_id, rays, conv, feat_noise, anoise, min_rough, row_inputs=1):
    """heads_fwd(app, head_W, head_b, hp) + bounce_prep_fwd(..., heads, ...) in one launch -> (heads, V, N, r1, f0, diffuse, feat, xyz)"""
    Mb = bidx.shape[0]
    dev = normals.device
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
    heads, V, N, r1, f0, diff, feat, xyz = f(Mb, 11), f(Mb, 3), f(Mb, 3), f(Mb), f(Mb, 3), f(Mb, 3), f(Mb, 24), f(Mb, 3)
    if Mb:
        _check(_lib.nmf_bounce_prep_fwd_heads(_p(bidx, torch.int32), C.c_int64(Mb), _p(normals, torch.float32), _p(app, torch.float32),
                                              _p(head_W, torch.float32), _p(head_b, torch.float32), *[C.c_float(v) for v in hp],
                                              _p(xyzt, torch.float32), _p(ray_id, torch.int32), _p(rays, torch.float32),
                                              _p(conv, torch.float32), _p(feat_noise, torch.float32), C.c_float(anoise),
                                              C.c_float(min_rough), C.c_int32(int(row_inputs)), _p(heads), _p(V), _p(N), _p(r1), _p(f0),
                                              _p(diff), _p(feat), _p(xyz), _stream()), "nmf_bounce_prep_fwd_heads")
    return heads, V, N, r1, f0, diff, feat, xyz


def bounce_index_select(weights, u, mode, mul, add=0.0, sum_w=1.0, xyzt=None):
    """select_bounces(weights, u, mode, mul, add, sum_w) + bounce_index(counts, xyzt) without materialising the counts: the count of a
    sample is evaluated inside the two launches of the index (nmf_bounce_index_select).  sum_w: float or 0-d device tensor."""
    M = weights.shape[0]
    dev = weights.device
    rows = torch.empty((max(M, 1), 4), dtype=torch.float32, device=dev) if xyzt is not None else None
    bidx = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
    row_off = torch.empty(M + 1, dtype=torch.int64, device=dev)
    inv = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
    cnt_rows = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
    totals = torch.empty(2, dtype=torch.int64, device=dev)
    nbytes = _lib.nmf_bounce_index_workspace_bytes(C.c_int64(M))
    ws = torch.empty(nbytes // 8, dtype=torch.int64, device=dev)
    dev_sum = sum_w if isinstance(sum_w, torch.Tensor) else None
    _check(_lib.nmf_bounce_index_select(_p(weights, torch.float32) if M else None, _p(u, torch.float32) if M else None, C.c_int32(mode),
                                        C.c_float(mul), C.c_float(add), C.c_float(1.0 if dev_sum is not None else sum_w),
                                        _p(dev_sum, torch.float32), C.c_int64(M), None, _p(bidx), _p(row_off), _p(cnt_rows), _p(inv),
                                        _p(totals), _p(xyzt, torch.float32) if (xyzt is not None and M) else None, _p(rows), _p(ws),
                                        C.c_int64(nbytes), None, C.c_int64(0), _stream()), "nmf_bounce_index_select")
    if xyzt is not None:
        return bidx, row_off, cnt_rows, inv[:M], totals, rows
    return bidx, row_off, cnt_rows, inv[:M], totals


def bounce_prep_fwd(bidx, normals, app, heads, xyzt, ray_id, rays, conv, feat_noise, anoise, min_rough, row_inputs=False):
    Mb = bidx.shape[0]
    dev = normals.device
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
    V, N, r1, f0, diff, feat, xyz = f(Mb, 3), f(Mb, 3), f(Mb), f(Mb, 3), f(Mb, 3), f(Mb, 24), f(Mb, 3)
    if Mb:
        _check(_lib.nmf_bounce_prep_fwd(_p(bidx, torch.int32), C.c_int64(Mb), _p(normals, torch.float32),
                                        _p(app, torch.float32), _p(heads, torch.float32), _p(xyzt, torch.float32),
                                        _p(ray_id, torch.int32), _p(rays, torch.float32), _p(conv, torch.float32),
                                        _p(feat_noise), C.c_float(anoise), C.c_float(min_rough), C.c_int32(int(row_inputs)),
                                        _p(V), _p(N), _p(r1),
                                        _p(f0), _p(diff), _p(feat), _p(xyz), _stream()), "nmf_bounce_prep_fwd")
    return V, N, r1, f0, diff, feat, xyz


def _rows(t, width):
    """(pointer, row pitch in floats) of a [n,width] / [n] fp32 tensor whose rows are dense but may be a column slice of
    a wider tensor; anything else is made contiguous first."""
    if t is None:
        return C.c_void_p(0), width
    if t.dtype != torch.float32 or not t.is_cuda:
        raise NmfHipError("expected a float32 device tensor")
    ok = (t.dim() == 2 and t.shape[1] == width and t.stride(1) == 1) or (t.dim() == 1 and width == 1)
    if not ok or t.stride(0) < width:
        t = t.contiguous()
    return C.c_void_p(t.data_ptr()), (t.stride(0) if t.shape[0] > 1 else width)


def bounce_prep_bwd(inv, normals, heads, ray_id, rays, conv, min_rough, detach_n, dN, dr1, df0, ddiff, dfeat,
                    bidx=None, row_inputs=False):
    """row_inputs: heads is [Mb,11] and d_heads / d_app come back per bounce row ([Mb,11], [Mb,24]); row_inputs == 2: normals
    [Mb,3] and d_normals [Mb,3] are per bounce row too (inv may be None, M is taken from ray_id)"""
    M = inv.shape[0] if inv is not None else ray_id.shape[0]
    Mb = bidx.shape[0] if bidx is not None else 0
    n_out = Mb if row_inputs else M
    dev = normals.device
    d_normals = torch.empty((Mb if int(row_inputs) == 2 else M, 3), dtype=torch.float32, device=dev)
    d_heads = torch.empty((n_out, 11), dtype=torch.float32, device=dev)
    d_app = torch.empty((n_out, 24), dtype=torch.float32, device=dev)
    if M:
        (pN, sN), (pr, sr), (pf, sf), (pd, sd) = _rows(dN, 3), _rows(dr1, 1), _rows(df0, 3), _rows(ddiff, 3)
        strides = (C.c_int32 * 4)(sN, sr, sf, sd)
        _check(_lib.nmf_bounce_prep_bwd(_p(inv, torch.int32) if inv is not None else None, C.c_int64(M), _p(bidx), C.c_int64(Mb),
                                        _p(normals, torch.float32), _p(heads if heads.shape[0] else None),
                                        _p(ray_id, torch.int32), _p(rays, torch.float32), _p(conv, torch.float32),
                                        C.c_float(min_rough), C.c_int32(1 if detach_n else 0),
                                        C.c_int32(int(row_inputs)), pN, pr, pf, pd, strides, _p(dfeat), _p(d_normals), _p(d_heads), _p(d_app),
                                        _stream()), "nmf_bounce_prep_bwd")
    return d_normals, d_heads, d_app


def ray_compose_fwd(weight, refl_rows, inv, normals, rays, offsets, B, bg, bg_per_ray, tonemap, noclip, want_ori):
    dev = weight.device
    rgb_map = torch.empty((B, 3), dtype=torch.float32, device=dev)
    acc = torch.empty(B, dtype=torch.float32, device=dev)
    rgb_lin = torch.empty((B, 3), dtype=torch.float32, device=dev)
    ori = torch.empty(B, dtype=torch.float32, device=dev) if want_ori else None
    if B:
        _check(_lib.nmf_ray_compose_fwd(_p(weight, torch.float32), _p(refl_rows), _p(inv), _p(normals),
                                        _p(rays, torch.float32), _p(offsets, torch.int64), C.c_int64(B),
                                        _p(bg, torch.float32), C.c_int32(1 if bg_per_ray else 0),
                                        C.c_int32(1 if tonemap else 0), C.c_int32(1 if noclip else 0), _p(rgb_map),
                                        _p(acc), _p(rgb_lin), _p(ori), _stream()), "nmf_ray_compose_fwd")
    return rgb_map, acc, rgb_lin, ori


def ray_compose_bwd(weight, refl_rows, inv, normals, rays, ray_id, bg, bg_per_ray, tonemap, noclip, rgb_lin, d_rgb_map,
                    d_acc, d_ori, want_d_normals):
    M = weight.shape[0]
    dev = weight.device
    d_weight = torch.empty(M, dtype=torch.float32, device=dev)
    d_refl = torch.empty_like(refl_rows) if refl_rows is not None else None
    d_normals = torch.empty((M, 3), dtype=torch.float32, device=dev) if want_d_normals else None
    if M:
        _check(_lib.nmf_ray_compose_bwd(_p(weight, torch.float32), _p(refl_rows), _p(inv), _p(normals),
                                        _p(rays, torch.float32), _p(ray_id, torch.int32), C.c_int64(M),
                                        _p(bg, torch.float32), C.c_int32(1 if bg_per_ray else 0),
                                        C.c_int32(1 if tonemap else 0), C.c_int32(1 if noclip else 0), _p(rgb_lin),
                                        _p(d_rgb_map), _p(d_acc), _p(d_ori), _p(d_weight), _p(d_refl), _p(d_normals),
                                        _stream()), "nmf_ray_compose_bwd")
    return d_weight, d_refl, d_normals


# ---- loss terms ------------------------------------------------------------------------------------
def _dense_f32(t):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise NmfHipError("expected a float32 device tensor")
    if not (t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last)):
        raise NmfHipError("tensor storage must be dense")
    return t.data_ptr()


def l1_mean_fwd(tensors):
    n = len(tensors)
    out = torch.zeros((), dtype=torch.float32, device=tensors[0].device)
    ptrs = (C.c_void_p * n)(*[_dense_f32(t) for t in tensors])
    numel = (C.c_int64 * n)(*[t.numel() for t in tensors])
    _check(_lib.nmf_l1_mean_fwd(ptrs, numel, C.c_int32(n), _p(out), _stream()), "nmf_l1_mean_fwd")
    return out


def l1_mean_bwd(tensors, d_out, out=None):
    """-> gradients in the tensors' own memory order; out = existing gradient tensors of the same storage order to ADD into"""
    n = len(tensors)
    grads = out if out is not None else [torch.empty_like(t, memory_format=torch.preserve_format) for t in tensors]
    ptrs = (C.c_void_p * n)(*[_dense_f32(t) for t in tensors])
    gptrs = (C.c_void_p * n)(*[_dense_f32(g) for g in grads])
    numel = (C.c_int64 * n)(*[t.numel() for t in tensors])
    for t, g in zip(tensors, grads):
        if g.numel() != t.numel():
            raise NmfHipError("l1_mean_bwd: gradient / tensor size mismatch")
    _check(_lib.nmf_l1_mean_bwd(ptrs, numel, C.c_int32(n), _p(d_out, torch.float32), gptrs, C.c_int32(0 if out is None else 1),
                                _stream()), "nmf_l1_mean_bwd")
    return grads


def loss_mix_fwd(tensors, weights, scale):
    """scale * sum_i w_i * sum(x_i) -> 0-d tensor (one launch)"""
    n = len(tensors)
    out = torch.zeros((), dtype=torch.float32, device=tensors[0].device)
    ptrs = (C.c_void_p * n)(*[_dense_f32(t) for t in tensors])
    numel = (C.c_int64 * n)(*[t.numel() for t in tensors])
    w = (C.c_float * n)(*[float(v) for v in weights])
    _check(_lib.nmf_loss_mix_fwd(ptrs, numel, w, C.c_int32(n), C.c_float(scale), _p(out), _stream()), "nmf_loss_mix_fwd")
    return out


def loss_mix_bwd(shapes, weights, scale, d_out):
    """constant gradients d_out * scale * w_i shaped like the inputs (one launch)"""
    n = len(shapes)
    grads = [torch.empty(s, dtype=torch.float32, device=d_out.device) for s in shapes]
    numel = (C.c_int64 * n)(*[g.numel() for g in grads])
    w = (C.c_float * n)(*[float(v) for v in weights])
    gptrs = (C.c_void_p * n)(*[g.data_ptr() for g in grads])
    _check(_lib.nmf_loss_mix_bwd(numel, w, C.c_int32(n), C.c_float(scale), _p(d_out, torch.float32), gptrs, _stream()),
           "nmf_loss_mix_bwd")
    return grads


_loss_ws = {}


def loss_head_workspace(device, n_rays):
    """zeroed workspace of nmf_loss_head, one per (device, current stream), grown on demand (its ticket counter is zero between
    launches)"""
    key = (device, _stream())
    need = int(_lib.nmf_loss_head_workspace_bytes(C.c_int64(n_rays)))
    ws = _loss_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _loss_ws[key] = torch.zeros(max(2 * need, 4096), dtype=torch.uint8, device=device)
    return ws


def loss_head(pred, gt, d_out, scale, w_pred, w_a, w_b):
    """sqerr_fwd + loss_mix_bwd + sqerr_bwd of one chunk in one launch -> (loss 0-d, d_pred [B,3], g_a [B], g_b [B]):
    loss = sum (clip(pred) - clip(gt))^2 (summed in a fixed order, written: no zero fill), d_pred = 2 (pred - clip(gt)) *
    (d_out scale w_pred) inside [0,1], g_a / g_b filled with d_out scale w_a / w_b."""
    B = pred.shape[0]
    loss = torch.empty((), dtype=torch.float32, device=pred.device)
    d_pred = torch.empty_like(pred)
    g_a = torch.empty(B, dtype=torch.float32, device=pred.device)
    g_b = torch.empty(B, dtype=torch.float32, device=pred.device)
    ws = loss_head_workspace(pred.device, B)
    _check(_lib.nmf_loss_head(_p(pred, torch.float32), _p(gt, torch.float32), C.c_int64(B), _p(d_out, torch.float32),
                              C.c_float(scale), C.c_float(w_pred), C.c_float(w_a), C.c_float(w_b), _p(loss), _p(d_pred),
                              _p(g_a), _p(g_b), C.c_void_p(ws.data_ptr()), C.c_int64(ws.numel()), _stream()), "nmf_loss_head")
    return loss, d_pred, g_a, g_b


def bg_adjoint(acc, d_rgb):
    """(1 - acc)[:, None] * d_rgb in one launch (the background adjoint nmf_ray_compose_bwd leaves to the caller)"""
    d_bg = torch.empty_like(d_rgb)
    _check(_lib.nmf_bg_adjoint(_p(acc, torch.float32), _p(d_rgb.contiguous(), torch.float32), C.c_int64(acc.shape[0]), _p(d_bg),
                               _stream()), "nmf_bg_adjoint")
    return d_bg


def sqerr_fwd(pred, gt):
    out = torch.zeros((), dtype=torch.float32, device=pred.device)
    if pred.numel():
        _check(_lib.nmf_sqerr_fwd(_p(pred, torch.float32), _p(gt, torch.float32), C.c_int64(pred.numel()), _p(out),
                                  _stream()), "nmf_sqerr_fwd")
    return out


def sqerr_bwd(pred, gt, d_out):
    d_pred = torch.empty_like(pred)
    if pred.numel():
        _check(_lib.nmf_sqerr_bwd(_p(pred, torch.float32), _p(gt, torch.float32), C.c_int64(pred.numel()),
                                  _p(d_out, torch.float32), _p(d_pred), _stream()), "nmf_sqerr_bwd")
    return d_pred


# ---- retrace selection ------------------------------------------------------------------------------
def retrace_scores(brdf, V_rows, N_rows, lpdf, w_rows, cnt_rows, row_of_ray):
    R = row_of_ray.shape[0]
    score = torch.empty(R, dtype=torch.float32, device=brdf.device)
    if R:
        _check(_lib.nmf_retrace_scores(_p(brdf, torch.float32), _p(V_rows, torch.float32), _p(N_rows, torch.float32),
                                       _p(lpdf, torch.float32), _p(w_rows, torch.float32), _p(cnt_rows, torch.int32),
                                       _p(row_of_ray, torch.int32), C.c_int64(R), _p(score), _stream()),
               "nmf_retrace_scores")
    return score


def argsort_f32(keys):
    """ascending argsort of a 1-D fp32 device tensor -> int32 indices"""
    n = keys.shape[0]
    order = torch.empty(n, dtype=torch.int32, device=keys.device)
    if n:
        nbytes = _lib.nmf_argsort_workspace_bytes(C.c_int64(n))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=keys.device)
        _check(_lib.nmf_argsort_f32(_p(keys, torch.float32), C.c_int64(n), _p(order), _p(ws), C.c_int64(nbytes),
                                    _stream()), "nmf_argsort_f32")
    return order


def topk_select(keys, k):
    """The partition models/microfacet.py:506-537 takes from `color_contribution.argsort()`, by radix select (no full sort):
    -> (idx_top [k] int32 = argsort(keys)[n-k:] in that order, idx_rest [n-k] int32 = the other indices in index order)"""
    n = keys.shape[0]
    k = int(k)
    top = torch.empty(k, dtype=torch.int32, device=keys.device)
    rest = torch.empty(n - k, dtype=torch.int32, device=keys.device)
    if n:
        nbytes = _lib.nmf_topk_select_workspace_bytes(C.c_int64(n))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=keys.device)
        _check(_lib.nmf_topk_select(_p(keys, torch.float32), C.c_int64(n), C.c_int64(k), _p(top) if k else None,
                                    _p(rest) if n - k else None, _p(ws), C.c_int64(nbytes), _stream()), "nmf_topk_select")
    return top, rest


def multi_copy(slots, n):
    """slots: (CopySlot * k) host array; copies the first n (src -> dst, with fp32 <-> fp64 conversion) in one launch"""
    _check(_lib.nmf_multi_copy(slots, n, _stream()), "nmf_multi_copy")


# ---- host-side fast path ---------------------------------------------------------------------------------------------
# lib/_nmf_host.so (csrc/host_ext.cpp) implements the forward wrappers above in C++ -- same argument order, same outputs,
# same checks, same C-ABI entry points -- at ~3 us per call instead of 10-25 us of Python (output allocation, checked
# pointers, ctypes marshalling).  When the module is present its functions replace the Python definitions; the latter
# stay reachable as PY_WRAPPERS[name] (tests compare both) and NMF_HOST_EXT=0 disables the replacement.
PY_WRAPPERS = {}
HOST_EXT = None


def _load_host_ext():
    if os.environ.get("NMF_HOST_EXT", "1") == "0":
        return None
    path = os.path.join(_HERE, "lib", "_nmf_host.so")
    if not os.path.exists(path):
        return None
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("_nmf_host", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if mod.abi_version() != version():
            return None
        mod.set_error_class(NmfHipError)
        return mod
    except Exception:                   # stale build, missing torch symbols, ...: the Python wrappers do the same job
        return None


def _install_host_ext():
    global HOST_EXT
    fx = HOST_EXT = _load_host_ext()
    if fx is None:
        return
    g = globals()
    addr = C.addressof

    def march_count(p, rays, jitter, alpha_bits, alpha_coarse=None):
        return fx.march_count(addr(p), rays, jitter, alpha_bits, alpha_coarse, _stream())

    def march_scan(counts, max_samples):
        return fx.march_scan(counts, int(max_samples), _stream())

    def march_fill(p, rays, b, M, jitter, valid, offsets, want_z=True):
        return fx.march_fill(addr(p), rays, b, M, jitter, valid, offsets, want_z, _stream())

    py_vm_query_fwd = g["vm_query_fwd"]

    def vm_query_fwd(p, xyzt, dpk, dlk, app_planes, app_lines, basis, want_density=True, want_normal=True,
                     want_app=True, want_coef=False):
        if app_planes[0].dtype != torch.float32:       # bf16 tables: the ctypes wrapper dispatches on the table dtype
            return py_vm_query_fwd(p, xyzt, dpk, dlk, app_planes, app_lines, basis, want_density, want_normal, want_app,
                                   want_coef)
        return fx.vm_query_fwd(addr(p), xyzt, dpk, dlk, app_planes, app_lines, basis, want_density, want_normal,
                               want_app, want_coef, _stream())

    def composite_fwd(sigma, dist, offsets, b, distance_scale):
        return fx.composite_fwd(sigma, dist, offsets, b, distance_scale, _stream())

    def segment_sum(vals, scale, offsets, n_seg, lanes=1):
        return fx.segment_sum(vals, scale, offsets, n_seg, lanes, _stream())

    def sat_lookup_fwd(sat, dirs, sa, mipbias, pole_rows, sc=None):
        return fx.sat_lookup_fwd(sat, dirs, sa, mipbias, pole_rows, sc, _stream())

    def select_bounces(weights, u, mode, mul, add=0.0, sum_w=1.0):
        dev_sum = sum_w if isinstance(sum_w, torch.Tensor) else None
        return fx.select_bounces(weights, u, mode, mul, add, 1.0 if dev_sum is not None else sum_w, dev_sum, _stream())

    def expand_segments(offsets, n_seg, total):
        return fx.expand_segments(offsets, n_seg, total, _stream())

    def brdf_mlp_fwd(weights, half_vec, diff_vec, feat_src, rough_src, src_idx, out_bias, max_workgroups=0, with_mask=False, image=None):
        r = fx.brdf_mlp_fwd(list(weights or ()), half_vec, diff_vec, feat_src, rough_src, src_idx, out_bias, bool(with_mask),
                            int(max_workgroups), _stream(), image)
        return r if with_mask else r[0]

    def brdf_mlp_pack(weights, into=None):
        return fx.brdf_mlp_pack(list(weights), into, _stream())

    def heads_fwd(feat, W, b, hp):
        return fx.heads_fwd(feat, W, b, list(hp), _stream())

    def ggx_rays_fwd(V, N, r, x, off, cnt, sobol, row_of_ray, j_of_ray):
        return fx.ggx_rays_fwd(V, N, r, x, off, cnt, sobol, row_of_ray, j_of_ray, _stream())

    def shade_mix_fwd(V, f0, diff, cnt, row_of_ray, L, inc, brdf):
        return fx.shade_mix_fwd(V, f0, diff, cnt, row_of_ray, L, inc, brdf, _stream())

    def bounce_index(counts, xyzt=None):
        return fx.bounce_index(counts, xyzt, _stream())

    def bounce_prep_fwd(bidx, normals, app, heads, xyzt, ray_id, rays, conv, feat_noise, anoise, min_rough, row_inputs=False):
        return fx.bounce_prep_fwd(bidx, normals, app, heads, xyzt, ray_id, rays, conv, feat_noise, anoise, min_rough,
                                  int(row_inputs), _stream())

    def ray_compose_fwd(weight, refl_rows, inv, normals, rays, offsets, B, bg, bg_per_ray, tonemap, noclip, want_ori):
        return fx.ray_compose_fwd(weight, refl_rows, inv, normals, rays, offsets, B, bg, bool(bg_per_ray), bool(tonemap),
                                  bool(noclip), bool(want_ori), _stream())

    def composite_bwd(sigma, dist, weight, offsets, b, distance_scale, d_weight):
        return fx.composite_bwd(sigma, dist, weight, offsets, b, distance_scale, d_weight, _stream())

    def segment_sum_wide(vals, D, offsets, n_seg):
        return fx.segment_sum_wide(vals, D, offsets, n_seg, _stream())

    py_sat_lookup_bwd = g["sat_lookup_bwd"]

    def sat_lookup_bwd(sat, dirs, sa, mipbias, d_out, d_sat, d_pole, d_mip=None, want_dirs=True, want_mipbias=None, sc=None):
        if want_mipbias is not None:      # legacy return form (tests): the Python wrapper
            return py_sat_lookup_bwd(sat, dirs, sa, mipbias, d_out, d_sat, d_pole, d_mip, want_dirs, want_mipbias, sc)
        return fx.sat_lookup_bwd(sat, dirs, sa, mipbias, d_out, d_sat, d_pole, d_mip, want_dirs, sc,
                                 int(g["ENV_BINNED_MIN_LOOKUPS"]), _stream())

    def brdf_mlp_bwd(weights, half_vec, diff_vec, feat_src, rough_src, src_idx, fwd_out, act_mask, d_out, grads,
                     max_workgroups=0, image=None):
        return fx.brdf_mlp_bwd(list(weights or ()), half_vec, diff_vec, feat_src, rough_src, src_idx, fwd_out, act_mask, d_out,
                               list(grads), int(max_workgroups), _stream(), image)

    def heads_bwd(feat, W, b, hp, d_out, gW, gb, add_into=None):
        return fx.heads_bwd(feat, W, b, list(hp), d_out, gW, gb, add_into, _stream())

    def ggx_rays_bwd(V, N, r, off, sobol, row_of_ray, j_of_ray, dL, d_rays=None):
        return fx.ggx_rays_bwd(V, N, r, off, sobol, row_of_ray, j_of_ray, dL, d_rays, _stream())

    def shade_mix_bwd(V, f0, diff, cnt, row_of_ray, L, inc, brdf, d_rows):
        return fx.shade_mix_bwd(V, f0, diff, cnt, row_of_ray, L, inc, brdf, d_rows, _stream())

    def ray_compose_bwd(weight, refl_rows, inv, normals, rays, ray_id, bg, bg_per_ray, tonemap, noclip, rgb_lin, d_rgb_map,
                        d_acc, d_ori, want_d_normals):
        return fx.ray_compose_bwd(weight, refl_rows, inv, normals, rays, ray_id, bg, bool(bg_per_ray), bool(tonemap),
                                  bool(noclip), rgb_lin, d_rgb_map, d_acc, d_ori, bool(want_d_normals), _stream())

    def vm_query_bwd_segments(p, segs, dpk, dlk, app_planes, app_lines, basis, g_dpk, g_dlk, g_app_planes, g_app_lines,
                              g_basis=None, plan=None, clean=None):
        if clean is not None:
            return fx.vm_query_bwd_clean(addr(p), list(segs), dpk, dlk, app_planes, app_lines, basis, g_dpk, g_dlk, g_app_planes,
                                         g_app_lines, g_basis, clean, _stream())
        if plan is not None:
            return fx.vm_query_bwd_planned(addr(p), list(segs), dpk, dlk, app_planes, app_lines, basis, g_dpk, g_dlk, g_app_planes,
                                           g_app_lines, g_basis, plan, _stream())
        return fx.vm_query_bwd_segments(addr(p), list(segs), dpk, dlk, app_planes, app_lines, basis, g_dpk, g_dlk, g_app_planes,
                                        g_app_lines, g_basis, _stream())

    def vm_query_rows(p, xyzt, dpk, dlk):
        return fx.vm_query_rows(addr(p), xyzt, list(dpk), list(dlk), _stream())

    def vm_query_sigma(p, xyzt, planes, lines):
        return fx.vm_query_sigma(addr(p), xyzt, list(planes), list(lines), _stream())

    def sqerr_fwd(pred, gt):
        return fx.sqerr_fwd(pred, gt, _stream())

    def sqerr_bwd(pred, gt, d_out):
        return fx.sqerr_bwd(pred, gt, d_out, _stream())

    def shade_mix_bwd_view(V, f0, diff, cnt, row_of_ray, L, inc, brdf, d_rows):
        return fx.shade_mix_bwd_view(V, f0, diff, cnt, row_of_ray, L, inc, brdf, d_rows, _stream())

    def ggx_rays_bwd_view(V, N, r, off, sobol, row_of_ray, j_of_ray, dL, d_rays=None):
        return fx.ggx_rays_bwd_view(V, N, r, off, sobol, row_of_ray, j_of_ray, dL, d_rays, _stream())

    def view_adjoint_to_rays(ray_id, bidx, dv_a, dv_b, d_rays):
        return fx.view_adjoint_to_rays(ray_id, bidx, dv_a, dv_b, d_rays, _stream())

    def select_total(weights, u, extra):
        dev = weights.device
        return fx.select_total(weights, u, float(extra), g["select_total_workspace"](dev), _stream())

    def bounce_prep_bwd(inv, normals, heads, ray_id, rays, conv, min_rough, detach_n, dN, dr1, df0, ddiff, dfeat,
                        bidx=None, row_inputs=False):
        return fx.bounce_prep_bwd(inv, normals, heads, ray_id, rays, conv, float(min_rough), bool(detach_n), dN, dr1, df0, ddiff,
                                  dfeat, bidx, int(row_inputs), _stream())

    def adam_step(slots, n, guard=None):
        return fx.adam_step(C.addressof(slots), int(n), guard, _stream())

    def multi_copy(slots, n):
        return fx.multi_copy(C.addressof(slots), int(n), _stream())

    def loss_mix_bwd(shapes, weights, scale, d_out):
        return fx.loss_mix_bwd([list(s_) for s_ in shapes], [float(v) for v in weights], float(scale), d_out, _stream())

    py_l1_bwd, py_sat_bwd, py_sat_build, py_sh, py_pack = (g["l1_mean_bwd"], g["sat_build_bwd"], g["sat_build"], g["sh_project"],
                                                          g["vm_pack_density"])

    def l1_mean_bwd(tensors, d_out, out=None):
        if out is None:
            return py_l1_bwd(tensors, d_out, out)
        fx.l1_mean_bwd_into(list(tensors), d_out, list(out), _stream())
        return out

    def sat_build_bwd(d_sat, bg_mat, act, d_pole, brightness=0.0, mul=1.0, sc=None, out=None):
        if out is None or not bg_mat.is_contiguous():
            return py_sat_bwd(d_sat, bg_mat, act, d_pole, brightness, mul, sc, out)
        fx.sat_build_bwd_into(d_sat, bg_mat, act, d_pole, float(brightness), float(mul), sc, out, _stream())
        return out

    def sat_build(bg_mat, brightness=0.0, mul=1.0, sc=None, out=None, pole=False, interleaved=False):
        if out is None or not bg_mat.is_contiguous():
            return py_sat_build(bg_mat, brightness, mul, sc, out, pole, interleaved)
        fx.sat_build_into(bg_mat, float(brightness), float(mul), sc, out[0], out[1], out[2] if pole else None,
                          out[-1] if interleaved else None, _stream())
        return (out[0], out[1]) + ((out[2],) if pole else ()) + ((out[-1],) if interleaved else ())

    def sh_project(vals, wq, sh_A, out=None):
        if out is None:
            return py_sh(vals, wq, sh_A, out)
        fx.sh_project_into(vals, wq, sh_A, out[0], out[1], _stream())
        return out

    def vm_pack_density(p, planes, lines, out=None):
        if out is None:
            return py_pack(p, planes, lines, out)
        fx.vm_pack_density_into(addr(p), list(planes), list(lines), list(out[0]), list(out[1]), _stream())
        return out

    py_unpack = g["vm_unpack_density_grad"]

    def vm_unpack_density_grad(p, g_dpk, g_dlk, out=None, l1=None):
        if out is not None or l1 is not None:
            return py_unpack(p, g_dpk, g_dlk, out, l1)
        return fx.vm_unpack_density_grad(addr(p), g_dpk, g_dlk, _stream())

    for name, fn in list(locals().items()):
        if callable(fn) and name in g and name not in ("fx", "addr", "g", "py_sat_lookup_bwd", "py_unpack", "py_l1_bwd", "py_sat_bwd", "py_sat_build", "py_sh",
                                                             "py_pack"):
            PY_WRAPPERS[name] = g[name]
            fn.__doc__ = g[name].__doc__
            g[name] = fn


_install_host_ext()
