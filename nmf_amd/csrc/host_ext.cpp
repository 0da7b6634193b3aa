// Host-side fast path for the forward wrappers of nmf_amd/hip.py (compiled with g++, no device code).
//
// The training step is host-bound where the GPU kernels are short (level-0 shading, the forward tail): a ctypes wrapper
// costs 10-25 us of Python per call -- one torch.empty per output (~2 us each), a checked data_ptr per argument, ctypes
// marshalling -- against 3-10 us of kernel time.  The functions here do exactly what the Python wrappers of the same
// name do (same argument order, same outputs, same checks, same C-ABI entry point of include/nmf_hip.h -- with the
// compiler checking the prototypes), in ~3 us.  hip.py installs them over its own definitions when this module is
// present (NMF_HOST_EXT=0 keeps the pure-Python wrappers); nothing else in the package knows about it.
#include <torch/extension.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "nmf_hip.h"

namespace py = pybind11;
using at::Tensor;
using OT = c10::optional<Tensor>;

namespace {

PyObject* g_error_class = nullptr;   // nmf_amd.hip.NmfHipError

[[noreturn]] void fail(const std::string& msg) {
    if (g_error_class) {
        PyErr_SetString(g_error_class, msg.c_str());
        throw py::error_already_set();
    }
    throw std::runtime_error(msg);
}

void check(int code, const char* what) {
    if (code != 0) fail(std::string(what) + " failed: " + nmf_last_error_string() + " [" + std::to_string(code) + "]");
}

// device pointer of a contiguous tensor of the expected dtype (mirrors hip._p); empty tensors give NULL
template <class T>
T* ptr(const Tensor& t, at::ScalarType st) {
    if (t.scalar_type() != st) fail(std::string("expected ") + c10::toString(st) + ", got " + c10::toString(t.scalar_type()));
    if (!t.is_cuda()) fail("nmf_amd operators need device tensors (no CPU path)");
    if (!t.is_contiguous()) fail("tensor must be contiguous");
    return static_cast<T*>(t.data_ptr());
}
template <class T>
T* optr(const OT& t, at::ScalarType st) {
    return t.has_value() ? ptr<T>(*t, st) : nullptr;
}
// any dtype (bit masks, workspaces): only device + contiguity are checked
void* vptr(const OT& t) {
    if (!t.has_value()) return nullptr;
    if (!t->is_cuda()) fail("nmf_amd operators need device tensors (no CPU path)");
    if (!t->is_contiguous()) fail("tensor must be contiguous");
    return t->data_ptr();
}
const float* f32(const Tensor& t) { return ptr<const float>(t, at::kFloat); }
const float* of32(const OT& t) { return optr<const float>(t, at::kFloat); }
const int32_t* i32(const Tensor& t) { return ptr<const int32_t>(t, at::kInt); }
const int64_t* i64(const Tensor& t) { return ptr<const int64_t>(t, at::kLong); }

Tensor fe(const Tensor& like, at::IntArrayRef shape) { return at::empty(shape, like.options().dtype(at::kFloat)); }
Tensor ie(const Tensor& like, at::IntArrayRef shape, at::ScalarType st) { return at::empty(shape, like.options().dtype(st)); }
float* out(Tensor& t) { return static_cast<float*>(t.data_ptr()); }
void* st(int64_t stream) { return reinterpret_cast<void*>(stream); }

// ---- optional per-call device timing (bench.py: per-kernel table of the step) --------------------------------------------
// When a CallTimer is active, every wrapper below records an event in front of and behind its C-ABI call on the stream it
// launches on; report() turns them into {call name: (summed ms, calls)}.  Inactive: one pointer test per call.
struct CallTimer {
    struct Rec { const char* name; void* a; void* b; void* stream; double host_us; };
    std::vector<void*> pool;
    size_t next = 0;
    std::vector<Rec> recs;
    void* ev() {
        if (next == pool.size()) {
            void* e = nullptr;
            check(nmf_event_create_timed(&e), "nmf_event_create_timed");
            pool.push_back(e);
        }
        return pool[next++];
    }
    ~CallTimer() { for (void* e : pool) nmf_event_destroy(e); }
};
CallTimer* g_call_timer = nullptr;
std::string g_call_filter;           // non-empty: only the wrapper of that name (or of a comma-separated list of names) is timed
// measurement knob (tools/ab_inprocess.py calldelay:NAME): a host busy-wait in front of every call of one wrapper -- a delay the
// step absorbs means the device bounds that stretch, a delay that shows 1:1 means the host's issue rate does
std::string g_delay_name;
double g_delay_us = 0.0;
struct TimedScope {
    const char* name; void* stream; void* a = nullptr;
    TimedScope(const char* n, int64_t s) : name(n), stream(reinterpret_cast<void*>(s)) {
        if (g_delay_us > 0.0 && g_delay_name == n) {
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < g_delay_us) {}
        }
        if (g_call_timer && (g_call_filter.empty() || g_call_filter == n ||
                             (g_call_filter.find(',') != std::string::npos && (',' + g_call_filter + ',').find(',' + std::string(n) + ',') != std::string::npos))) {
            a = g_call_timer->ev();
            nmf_event_record(a, stream);
        }
    }
    ~TimedScope() {
        if (a && g_call_timer) {
            void* b = g_call_timer->ev();
            nmf_event_record(b, stream);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
            g_call_timer->recs.push_back({name, a, b, stream, us});
        }
    }
};

// ---- per-KERNEL device timing (bench.py: roofline of the dominant kernel) ---------------------------------------------------
// libnmf_hip.so calls a probe in front of and behind every kernel launch (nmf_set_launch_probe); while a kernel timer is active the
// probe records a timed event on the launching stream on either side.  Names are the kernels' identifiers as rocprofv3 prints them.
CallTimer* g_kernel_timer = nullptr;
std::string g_kernel_filter;         // non-empty: only kernels whose name contains it
bool g_kernel_by_stream = false;
void* g_probe_open = nullptr;
void launch_probe(const char* name, void* stream, int phase) {
    CallTimer* t = g_kernel_timer;
    if (!t) return;
    if (phase == 0) {
        g_probe_open = nullptr;
        if (!g_kernel_filter.empty() && std::strstr(name, g_kernel_filter.c_str()) == nullptr) return;
        g_probe_open = t->ev();
        nmf_event_record(g_probe_open, stream);
    } else if (g_probe_open) {
        void* b = t->ev();
        nmf_event_record(b, stream);
        t->recs.push_back({name, g_probe_open, b, stream, 0.0});
        g_probe_open = nullptr;
    }
}
std::string kernel_name(const char* raw) {        // "(k_foo<1, 8>)" -> "k_foo<1, 8>"
    std::string s(raw);
    if (!s.empty() && s.front() == '(' && s.back() == ')') s = s.substr(1, s.size() - 2);
    return s;
}

struct P3 {
    const float* p[3];
};
P3 three(const std::vector<Tensor>& ts) {
    if (ts.size() != 3) fail("expected three tensors");
    P3 r;
    for (int i = 0; i < 3; ++i) r.p[i] = f32(ts[i]);
    return r;
}

// ---- sampler --------------------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor> march_count(int64_t p_addr, const Tensor& rays, const OT& jitter, const OT& alpha_bits,
                                       const OT& alpha_coarse, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const auto* p = reinterpret_cast<const nmf_march_params*>(p_addr);
    const int64_t B = rays.size(0), W = (p->n_steps + 63) / 64;
    Tensor valid = ie(rays, {B, W}, at::kLong), counts = ie(rays, {B}, at::kInt);
    check(nmf_march_count(p, f32(rays), B, of32(jitter), static_cast<const uint32_t*>(vptr(alpha_bits)),
                          alpha_bits.has_value() ? static_cast<const uint32_t*>(vptr(alpha_coarse)) : nullptr,
                          static_cast<uint64_t*>(valid.data_ptr()), static_cast<int32_t*>(counts.data_ptr()), st(stream)),
          "nmf_march_count");
    return {valid, counts};
}

// pub / pub_seq: mapped host memory (device pointer) the scan publishes its totals into, with that sequence number (0: none)
std::tuple<Tensor, Tensor, Tensor> march_scan(const Tensor& counts, int64_t max_samples, int64_t stream, int64_t pub = 0,
                                              int64_t pub_seq = 0) {
    TimedScope _ts(__func__, stream);
    const int64_t B = counts.size(0);
    Tensor offsets = ie(counts, {B + 1}, at::kLong), whole = ie(counts, {B}, at::kByte), totals = ie(counts, {2}, at::kLong);
    const int64_t nbytes = nmf_march_scan_workspace_bytes(B);
    Tensor ws = ie(counts, {nbytes / 8}, at::kLong);
    check(nmf_march_scan_publish(i32(counts), B, max_samples, static_cast<int64_t*>(offsets.data_ptr()),
                                 static_cast<uint8_t*>(whole.data_ptr()), static_cast<int64_t*>(totals.data_ptr()), ws.data_ptr(),
                                 nbytes, reinterpret_cast<void*>(pub), pub_seq, st(stream)),
          "nmf_march_scan");
    return {offsets, whole, totals};
}

std::tuple<Tensor, Tensor, Tensor, OT, Tensor> march_fill(int64_t p_addr, const Tensor& rays, int64_t b, int64_t M,
                                                         const OT& jitter, const Tensor& valid, const Tensor& offsets,
                                                         bool want_z, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const auto* p = reinterpret_cast<const nmf_march_params*>(p_addr);
    Tensor xyzt = fe(rays, {M, 4}), ray_id = ie(rays, {M}, at::kInt), step_id = ie(rays, {M}, at::kInt), dist = fe(rays, {M});
    OT z;
    if (want_z) z = fe(rays, {M});
    check(nmf_march_fill(p, f32(rays), b, of32(jitter), static_cast<const uint64_t*>(vptr(valid)), i64(offsets), out(xyzt),
                         static_cast<int32_t*>(ray_id.data_ptr()), static_cast<int32_t*>(step_id.data_ptr()),
                         z.has_value() ? out(*z) : nullptr, out(dist), st(stream)),
          "nmf_march_fill");
    return {xyzt, ray_id, step_id, z, dist};
}

// ---- field ----------------------------------------------------------------------------------------------------------
std::tuple<OT, OT, OT, OT, OT, OT> vm_query_fwd(int64_t p_addr, const Tensor& xyzt, const std::vector<Tensor>& dpk,
                                                const std::vector<Tensor>& dlk, const std::vector<Tensor>& apl,
                                                const std::vector<Tensor>& ali, const OT& basis, bool want_density,
                                                bool want_normal, bool want_app, bool want_coef, int64_t stream, int64_t live = 0) {
    // live: device address of the sample count when xyzt is sized by a bound (nmf_vm_query_fwd_live)
    TimedScope _ts(__func__, stream);
    const auto* p = reinterpret_cast<const nmf_vm_params*>(p_addr);
    const int64_t M = xyzt.size(0);
    OT sf, sg, gr, nr, ap, cf;
    if (want_density) { sf = fe(xyzt, {M}); sg = fe(xyzt, {M}); }
    if (want_normal) { gr = fe(xyzt, {M, 3}); nr = fe(xyzt, {M, 3}); }
    if (want_app) ap = fe(xyzt, {M, 24});
    if (want_coef) cf = fe(xyzt, {M, 72});
    const bool need_d = want_density || want_normal, need_a = want_app || want_coef;
    auto o = [](OT& t) { return t.has_value() ? static_cast<float*>(t->data_ptr()) : nullptr; };
    const bool bf16 = (need_d ? dpk : apl).at(0).scalar_type() == at::kBFloat16;
    if (live) {
        const void *a[3], *b[3], *c[3], *d[3];
        for (int i = 0; i < 3; ++i) {
            a[i] = need_d ? vptr(dpk.at(i)) : nullptr;
            b[i] = need_d ? vptr(dlk.at(i)) : nullptr;
            c[i] = need_a ? vptr(apl.at(i)) : nullptr;
            d[i] = need_a ? vptr(ali.at(i)) : nullptr;
        }
        check(nmf_vm_query_fwd_live(p, f32(xyzt), M, reinterpret_cast<const int64_t*>(live), need_d ? a : nullptr, need_d ? b : nullptr,
                                    need_a ? c : nullptr, need_a ? d : nullptr, bf16 ? 1 : 0, need_a ? of32(basis) : nullptr, o(sf),
                                    o(sg), o(gr), o(nr), o(ap), o(cf), st(stream)),
              "nmf_vm_query_fwd_live");
        return {sf, sg, gr, nr, ap, cf};
    }
    if (bf16) {        // bfloat16 copies of the tables (BASELINE configs[1]); fp32 arithmetic
        const uint16_t *a[3], *b[3], *c[3], *d[3];
        for (int i = 0; i < 3; ++i) {
            a[i] = need_d ? static_cast<const uint16_t*>(vptr(dpk.at(i))) : nullptr;
            b[i] = need_d ? static_cast<const uint16_t*>(vptr(dlk.at(i))) : nullptr;
            c[i] = need_a ? static_cast<const uint16_t*>(vptr(apl.at(i))) : nullptr;
            d[i] = need_a ? static_cast<const uint16_t*>(vptr(ali.at(i))) : nullptr;
        }
        check(nmf_vm_query_fwd_bf16(p, f32(xyzt), M, need_d ? a : nullptr, need_d ? b : nullptr, need_a ? c : nullptr,
                                    need_a ? d : nullptr, need_a ? of32(basis) : nullptr, o(sf), o(sg), o(gr), o(nr), o(ap), o(cf),
                                    st(stream)),
              "nmf_vm_query_fwd_bf16");
        return {sf, sg, gr, nr, ap, cf};
    }
    P3 a{}, b{}, c{}, d{};
    if (need_d) { a = three(dpk); b = three(dlk); }
    if (need_a) { c = three(apl); d = three(ali); }
    check(nmf_vm_query_fwd(p, f32(xyzt), M, need_d ? a.p : nullptr, need_d ? b.p : nullptr, need_a ? c.p : nullptr,
                           need_a ? d.p : nullptr, need_a ? of32(basis) : nullptr, o(sf), o(sg), o(gr), o(nr), o(ap), o(cf),
                           st(stream)),
          "nmf_vm_query_fwd");
    return {sf, sg, gr, nr, ap, cf};
}

// ---- compositing ------------------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor> composite_fwd(const Tensor& sigma, const Tensor& dist, const Tensor& offsets, int64_t b,
                                         double distance_scale, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const int64_t M = sigma.size(0);
    Tensor weight = fe(sigma, {M}), acc = fe(sigma, {b});
    if (M == 0) {
        acc.zero_();
        return {weight, acc};
    }
    check(nmf_composite_fwd(f32(sigma), f32(dist), i64(offsets), b, (float)distance_scale, out(weight), out(acc), st(stream)),
          "nmf_composite_fwd");
    return {weight, acc};
}

Tensor segment_sum(const Tensor& vals, const OT& scale, const Tensor& offsets, int64_t n_seg, int64_t lanes, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const int64_t D = vals.size(1);
    Tensor o = fe(vals, {n_seg, D});
    if (vals.size(0) == 0) {
        o.zero_();
        return o;
    }
    check(nmf_segment_sum(f32(vals), static_cast<const float*>(vptr(scale)), i64(offsets), n_seg, (int32_t)D, (int32_t)lanes,
                          out(o), st(stream)),
          "nmf_segment_sum");
    return o;
}

// ---- environment map --------------------------------------------------------------------------------------------------
Tensor sat_lookup_fwd(const Tensor& sat, const Tensor& dirs, const Tensor& sa, double mipbias, const OT& pole_rows,
                      const OT& sc, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const int64_t R = dirs.size(0), ld = dirs.size(1);
    const bool i4 = sat.dim() == 3 && sat.size(-1) == 4 && sat.size(0) != 3;          // [H][W][4] (nmf_sat_build's sat_i4)
    const int64_t H = i4 ? sat.size(0) : sat.size(-2), W = i4 ? sat.size(1) : sat.size(-1);
    Tensor o = fe(dirs, {R, 3});
    check(nmf_sat_lookup_fwd(f32(sat), (int32_t)H, (int32_t)W, f32(dirs), (int32_t)ld, f32(sa), R, (float)mipbias,
                             static_cast<const float*>(vptr(sc)), static_cast<const float*>(vptr(pole_rows)), i4 ? 1 : 0,
                             out(o), st(stream)),
          "nmf_sat_lookup_fwd");
    return o;
}

// ---- shading ------------------------------------------------------------------------------------------------------------
Tensor select_bounces(const Tensor& weights, const Tensor& u, int64_t mode, double mul, double add, double sum_w,
                      const OT& sum_w_dev, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const int64_t M = weights.size(0);
    Tensor counts = ie(weights, {M}, at::kInt);
    check(nmf_select_bounces(f32(weights), f32(u), M, (int32_t)mode, (float)mul, (float)add,
                             sum_w_dev.has_value() ? 1.0f : (float)sum_w, of32(sum_w_dev),
                             static_cast<int32_t*>(counts.data_ptr()), st(stream)),
          "nmf_select_bounces");
    return counts;
}

std::tuple<Tensor, Tensor> expand_segments(const Tensor& offsets, int64_t n_seg, int64_t total, int64_t stream) {
    TimedScope _ts(__func__, stream);
    Tensor seg = ie(offsets, {total}, at::kInt), loc = ie(offsets, {total}, at::kInt);
    check(nmf_expand_segments(i64(offsets), n_seg, static_cast<int32_t*>(seg.data_ptr()),
                              static_cast<int32_t*>(loc.data_ptr()), st(stream)),
          "nmf_expand_segments");
    return {seg, loc};
}

// returns {out [R,3], act_mask [R,4] int32 (undefined tensor without with_mask)}
std::tuple<Tensor, Tensor> brdf_mlp_fwd(const std::vector<Tensor>& w, const Tensor& half_vec, const Tensor& diff_vec,
                                        const Tensor& feat_src, const Tensor& rough_src, const OT& src_idx, double out_bias,
                                        bool with_mask, int64_t max_workgroups, int64_t stream, const OT& image = c10::nullopt) {
    // image: the packed weights (brdf_mlp_pack) -- the launch copies them instead of converting `w`, which may then be empty
    TimedScope _ts(__func__, stream);
    const bool packed = image.has_value() && image->defined();
    if (!packed && w.size() != 6) fail("brdf_mlp_fwd: six weight tensors expected");
    const int64_t R = half_vec.size(0);
    Tensor o = fe(half_vec, {R, 3});
    Tensor mask;
    if (with_mask) mask = ie(half_vec, {R, 4}, at::kInt);
    if (packed)
        check(nmf_brdf_mlp_fwd_packed(image->data_ptr(), f32(half_vec), f32(diff_vec), f32(feat_src), f32(rough_src),
                                      optr<const int32_t>(src_idx, at::kInt), R, (float)out_bias, out(o),
                                      with_mask ? static_cast<uint32_t*>(mask.data_ptr()) : nullptr, (int32_t)max_workgroups,
                                      st(stream)),
              "nmf_brdf_mlp_fwd_packed");
    else
        check(nmf_brdf_mlp_fwd(f32(w[0]), f32(w[1]), f32(w[2]), f32(w[3]), f32(w[4]), f32(w[5]), f32(half_vec), f32(diff_vec),
                               f32(feat_src), f32(rough_src), optr<const int32_t>(src_idx, at::kInt), R, (float)out_bias, out(o),
                               with_mask ? static_cast<uint32_t*>(mask.data_ptr()) : nullptr, (int32_t)max_workgroups,
                               st(stream)),
              "nmf_brdf_mlp_fwd");
    return {o, mask};
}

// the packed image of six weight tensors (into `into` when given: the training pass keeps one buffer)
Tensor brdf_mlp_pack(const std::vector<Tensor>& w, const OT& into, int64_t stream) {
    TimedScope _ts(__func__, stream);
    if (w.size() != 6) fail("brdf_mlp_pack: six weight tensors expected");
    const int64_t n = nmf_brdf_mlp_image_bytes();
    Tensor img = (into.has_value() && into->defined()) ? *into : at::empty({n}, w[0].options().dtype(at::kByte));
    if (img.scalar_type() != at::kByte || !img.is_contiguous() || img.numel() < n || !img.is_cuda())
        fail("brdf_mlp_pack: `into` must be a contiguous device uint8 tensor of nmf_brdf_mlp_image_bytes() bytes");
    check(nmf_brdf_mlp_pack(f32(w[0]), f32(w[1]), f32(w[2]), f32(w[3]), f32(w[4]), f32(w[5]), img.data_ptr(), img.numel(), st(stream)),
          "nmf_brdf_mlp_pack");
    return img;
}

Tensor heads_fwd(const Tensor& feat, const Tensor& W, const Tensor& b, const std::vector<double>& hp, int64_t stream) {
    TimedScope _ts(__func__, stream);
    if (hp.size() != 5) fail("heads_fwd: hp = (diffuse_mul, diffuse_bias, tint_bias, f0_bias, rough_bias)");
    const int64_t M = feat.size(0);
    Tensor o = fe(feat, {M, 11});
    check(nmf_heads_fwd(f32(feat), M, f32(W), f32(b), (float)hp[0], (float)hp[1], (float)hp[2], (float)hp[3], (float)hp[4],
                        out(o), st(stream)),
          "nmf_heads_fwd");
    return o;
}

std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> ggx_rays_fwd(const Tensor& V, const Tensor& N, const Tensor& r,
                                                                        const Tensor& x, const Tensor& off, const Tensor& cnt,
                                                                        const Tensor& sobol, const Tensor& row_of_ray,
                                                                        const Tensor& j_of_ray, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const int64_t R = row_of_ray.size(0);
    Tensor L = fe(V, {R, 3}), hl = fe(V, {R, 3}), dl = fe(V, {R, 3}), lpdf = fe(V, {R}), mip = fe(V, {R}), rays = fe(V, {R, 6});
    check(nmf_ggx_rays_fwd(f32(V), f32(N), f32(r), f32(x), f32(off), i32(cnt), f32(sobol), i32(row_of_ray), i32(j_of_ray), R,
                           out(L), out(hl), out(dl), out(lpdf), out(mip), out(rays), st(stream)),
          "nmf_ggx_rays_fwd");
    return {L, hl, dl, lpdf, mip, rays};
}

Tensor shade_mix_fwd(const Tensor& V, const Tensor& f0, const Tensor& diff, const Tensor& cnt, const Tensor& row_of_ray,
                     const Tensor& L, const Tensor& inc, const Tensor& brdf, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const int64_t R = row_of_ray.size(0);
    Tensor contrib = fe(V, {R, 3});
    check(nmf_shade_mix_fwd(f32(V), f32(f0), f32(diff), i32(cnt), i32(row_of_ray), R, f32(L), f32(inc), f32(brdf),
                            out(contrib), st(stream)),
          "nmf_shade_mix_fwd");
    return contrib;
}

py::tuple bounce_index(const Tensor& counts, const OT& xyzt, int64_t stream, int64_t pub = 0, int64_t pub_seq = 0, int64_t live = 0) {
    TimedScope _ts(__func__, stream);
    const int64_t M = counts.size(0), M1 = M > 0 ? M : 1;
    Tensor bidx = ie(counts, {M1}, at::kInt), row_off = ie(counts, {M + 1}, at::kLong), inv = ie(counts, {M1}, at::kInt);
    Tensor cnt_rows = ie(counts, {M1}, at::kInt), totals = ie(counts, {2}, at::kLong);
    const int64_t nbytes = nmf_bounce_index_workspace_bytes(M);
    Tensor ws = ie(counts, {nbytes / 8}, at::kLong);
    Tensor rows;
    if (xyzt.has_value()) rows = at::empty({M1, 4}, counts.options().dtype(at::kFloat));
    check(nmf_bounce_index_live(M ? i32(counts) : nullptr, M, reinterpret_cast<const int64_t*>(live), static_cast<int32_t*>(bidx.data_ptr()),
                                   static_cast<int64_t*>(row_off.data_ptr()), static_cast<int32_t*>(cnt_rows.data_ptr()),
                                   static_cast<int32_t*>(inv.data_ptr()), static_cast<int64_t*>(totals.data_ptr()),
                                   (xyzt.has_value() && M) ? f32(*xyzt) : nullptr, xyzt.has_value() ? out(rows) : nullptr,
                                   ws.data_ptr(), nbytes, reinterpret_cast<void*>(pub), pub_seq, st(stream)),
          "nmf_bounce_index");
    if (xyzt.has_value()) return py::make_tuple(bidx, row_off, cnt_rows, inv.narrow(0, 0, M), totals, rows);
    return py::make_tuple(bidx, row_off, cnt_rows, inv.narrow(0, 0, M), totals);
}

// select_bounces + bounce_index in the two launches of the latter (nmf_bounce_index_select); sum_w_dev: the device scalar of
// select_total (mode 1), absent for mode 0
py::tuple bounce_index_select(const Tensor& w, const Tensor& u, int64_t mode, double mul, double add, double sum_w, const OT& sum_w_dev,
                              const OT& xyzt, int64_t stream, int64_t pub = 0, int64_t pub_seq = 0, int64_t live = 0) {
    TimedScope _ts("bounce_index", stream);
    const int64_t M = w.size(0), M1 = M > 0 ? M : 1;
    Tensor bidx = ie(w, {M1}, at::kInt), row_off = ie(w, {M + 1}, at::kLong), inv = ie(w, {M1}, at::kInt);
    Tensor cnt_rows = ie(w, {M1}, at::kInt), totals = ie(w, {2}, at::kLong);
    const int64_t nbytes = nmf_bounce_index_workspace_bytes(M);
    Tensor ws = ie(w, {nbytes / 8}, at::kLong);
    Tensor rows;
    if (xyzt.has_value()) rows = at::empty({M1, 4}, w.options().dtype(at::kFloat));
    check(nmf_bounce_index_select(M ? f32(w) : nullptr, M ? f32(u) : nullptr, (int32_t)mode, (float)mul, (float)add, (float)sum_w,
                                  of32(sum_w_dev), M, reinterpret_cast<const int64_t*>(live), static_cast<int32_t*>(bidx.data_ptr()),
                                  static_cast<int64_t*>(row_off.data_ptr()), static_cast<int32_t*>(cnt_rows.data_ptr()),
                                  static_cast<int32_t*>(inv.data_ptr()), static_cast<int64_t*>(totals.data_ptr()),
                                  (xyzt.has_value() && M) ? f32(*xyzt) : nullptr, xyzt.has_value() ? out(rows) : nullptr, ws.data_ptr(),
                                  nbytes, reinterpret_cast<void*>(pub), pub_seq, st(stream)),
          "nmf_bounce_index_select");
    if (xyzt.has_value()) return py::make_tuple(bidx, row_off, cnt_rows, inv.narrow(0, 0, M), totals, rows);
    return py::make_tuple(bidx, row_off, cnt_rows, inv.narrow(0, 0, M), totals);
}

std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> bounce_prep_fwd(
    const Tensor& bidx, const Tensor& normals, const Tensor& app, const Tensor& heads, const Tensor& xyzt, const Tensor& ray_id,
    const Tensor& rays, const Tensor& conv, const OT& feat_noise, double anoise, double min_rough, int64_t row_inputs,
    int64_t stream) {
    TimedScope _ts(__func__, stream);
    const int64_t Mb = bidx.size(0);
    Tensor V = fe(normals, {Mb, 3}), N = fe(normals, {Mb, 3}), r1 = fe(normals, {Mb}), f0 = fe(normals, {Mb, 3});
    Tensor diff = fe(normals, {Mb, 3}), feat = fe(normals, {Mb, 24}), xyz = fe(normals, {Mb, 3});
    if (Mb)
        check(nmf_bounce_prep_fwd(i32(bidx), Mb, f32(normals), f32(app), f32(heads), f32(xyzt), i32(ray_id), f32(rays),
                                  f32(conv), static_cast<const float*>(vptr(feat_noise)), (float)anoise, (float)min_rough,
                                  (int32_t)row_inputs, out(V), out(N), out(r1), out(f0), out(diff), out(feat), out(xyz),
                                  st(stream)),
              "nmf_bounce_prep_fwd");
    return {V, N, r1, f0, diff, feat, xyz};
}

// heads_fwd + bounce_prep_fwd in one launch (nmf_bounce_prep_fwd_heads) -> {heads, V, N, r1, f0, diffuse, feat, xyz}
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> bounce_prep_fwd_heads(
    const Tensor& bidx, const Tensor& normals, const Tensor& app, const Tensor& head_W, const Tensor& head_b, const std::vector<double>& hp,
    const Tensor& xyzt, const Tensor& ray_id, const Tensor& rays, const Tensor& conv, const OT& feat_noise, double anoise, double min_rough,
    int64_t row_inputs, int64_t stream) {
    TimedScope _ts("bounce_prep_fwd", stream);
    if (hp.size() != 5) fail("bounce_prep_fwd_heads: hp = (diffuse_mul, diffuse_bias, tint_bias, f0_bias, rough_bias)");
    const int64_t Mb = bidx.size(0);
    Tensor heads = fe(normals, {Mb, 11});
    Tensor V = fe(normals, {Mb, 3}), N = fe(normals, {Mb, 3}), r1 = fe(normals, {Mb}), f0 = fe(normals, {Mb, 3});
    Tensor diff = fe(normals, {Mb, 3}), feat = fe(normals, {Mb, 24}), xyz = fe(normals, {Mb, 3});
    if (Mb)
        check(nmf_bounce_prep_fwd_heads(i32(bidx), Mb, f32(normals), f32(app), f32(head_W), f32(head_b), (float)hp[0], (float)hp[1],
                                        (float)hp[2], (float)hp[3], (float)hp[4], f32(xyzt), i32(ray_id), f32(rays), f32(conv),
                                        static_cast<const float*>(vptr(feat_noise)), (float)anoise, (float)min_rough, (int32_t)row_inputs,
                                        out(heads), out(V), out(N), out(r1), out(f0), out(diff), out(feat), out(xyz), st(stream)),
              "nmf_bounce_prep_fwd_heads");
    return {heads, V, N, r1, f0, diff, feat, xyz};
}

std::tuple<Tensor, Tensor, Tensor, OT> ray_compose_fwd(const Tensor& weight, const OT& refl_rows, const OT& inv,
                                                       const OT& normals, const Tensor& rays, const Tensor& offsets, int64_t B,
                                                       const Tensor& bg, bool bg_per_ray, bool tonemap, bool noclip,
                                                       bool want_ori, int64_t stream) {
    TimedScope _ts(__func__, stream);
    Tensor rgb_map = fe(weight, {B, 3}), acc = fe(weight, {B}), rgb_lin = fe(weight, {B, 3});
    OT ori;
    if (want_ori) ori = fe(weight, {B});
    if (B)
        check(nmf_ray_compose_fwd(f32(weight), static_cast<const float*>(vptr(refl_rows)),
                                  static_cast<const int32_t*>(vptr(inv)), static_cast<const float*>(vptr(normals)), f32(rays),
                                  i64(offsets), B, f32(bg), bg_per_ray ? 1 : 0, tonemap ? 1 : 0, noclip ? 1 : 0, out(rgb_map),
                                  out(acc), out(rgb_lin), ori.has_value() ? out(*ori) : nullptr, st(stream)),
              "nmf_ray_compose_fwd");
    return {rgb_map, acc, rgb_lin, ori};
}

// ---- backward wrappers (the step is GPU-bound there on a fast host, host-bound on a loaded one) ----------------------
Tensor composite_bwd(const Tensor& sigma, const Tensor& dist, const Tensor& weight, const Tensor& offsets, int64_t b,
                     double distance_scale, const Tensor& d_weight, int64_t stream) {
    TimedScope _ts(__func__, stream);
    Tensor d_sigma = at::empty_like(sigma);
    if (sigma.size(0) == 0) return d_sigma;
    Tensor dw = d_weight.contiguous();
    check(nmf_composite_bwd(f32(sigma), f32(dist), f32(weight), i64(offsets), b, (float)distance_scale, f32(dw),
                            out(d_sigma), st(stream)),
          "nmf_composite_bwd");
    return d_sigma;
}

Tensor segment_sum_wide(const Tensor& vals, int64_t D, const Tensor& offsets, int64_t n_seg, int64_t stream) {
    TimedScope _ts(__func__, stream);
    Tensor o = fe(vals, {n_seg, D});
    if (vals.size(0) == 0) {
        o.zero_();
        return o;
    }
    check(nmf_segment_sum_wide(f32(vals), vals.size(1), (int32_t)D, i64(offsets), n_seg, out(o), st(stream)),
          "nmf_segment_sum_wide");
    return o;
}

OT sat_lookup_bwd(const Tensor& sat, const Tensor& dirs, const Tensor& sa, double mipbias, const Tensor& d_out, const OT& d_sat,
                  const OT& d_pole, const OT& d_mip, bool want_dirs, const OT& sc, int64_t binned_from, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const int64_t R = dirs.size(0), ld = dirs.size(1);
    const bool i4 = sat.dim() == 3 && sat.size(-1) == 4 && sat.size(0) != 3;
    const int64_t H = i4 ? sat.size(0) : sat.size(-2), W = i4 ? sat.size(1) : sat.size(-1);
    OT d_dirs;
    if (want_dirs) d_dirs = fe(dirs, {R, ld});
    Tensor go = d_out.contiguous();
    if (d_sat.has_value() && R >= binned_from && ((H + 31) / 32) * ((W + 63) / 64) <= 1024) {
        // binned table adjoint; the record pool comes from the stream-ordered caching allocator per call
        const int64_t nbytes = nmf_sat_lookup_bwd_workspace_bytes(R);
        Tensor ws = torch::empty({nbytes}, dirs.options().dtype(torch::kUInt8));
        check(nmf_sat_lookup_bwd_binned(f32(sat), (int32_t)H, (int32_t)W, f32(dirs), (int32_t)ld, f32(sa), R, (float)mipbias,
                                        static_cast<const float*>(vptr(sc)), i4 ? 1 : 0, f32(go), static_cast<float*>(vptr(d_sat)),
                                        static_cast<float*>(vptr(d_pole)), d_dirs.has_value() ? out(*d_dirs) : nullptr,
                                        static_cast<float*>(vptr(d_mip)), ws.data_ptr(), nbytes, st(stream)),
              "nmf_sat_lookup_bwd_binned");
        return d_dirs;
    }
    check(nmf_sat_lookup_bwd(f32(sat), (int32_t)H, (int32_t)W, f32(dirs), (int32_t)ld, f32(sa), R, (float)mipbias,
                             static_cast<const float*>(vptr(sc)), i4 ? 1 : 0, f32(go), static_cast<float*>(vptr(d_sat)),
                             static_cast<float*>(vptr(d_pole)), d_dirs.has_value() ? out(*d_dirs) : nullptr,
                             static_cast<float*>(vptr(d_mip)), st(stream)),
          "nmf_sat_lookup_bwd");
    return d_dirs;
}

// the two halves of the binned adjoint on streams of their own (nmf_sat_lookup_bwd_dirs / nmf_sat_lookup_bwd_binned without d_pole)
Tensor sat_lookup_bwd_dirs(const Tensor& sat, const Tensor& dirs, const Tensor& sa, double mipbias, const Tensor& d_out,
                           const Tensor& d_pole, const OT& d_mip, const OT& sc, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const int64_t R = dirs.size(0), ld = dirs.size(1);
    const bool i4 = sat.dim() == 3 && sat.size(-1) == 4 && sat.size(0) != 3;
    const int64_t H = i4 ? sat.size(0) : sat.size(-2), W = i4 ? sat.size(1) : sat.size(-1);
    Tensor d_dirs = fe(dirs, {R, ld});
    if (!d_out.is_contiguous()) fail("sat_lookup_bwd_dirs: d_out must be contiguous");
    check(nmf_sat_lookup_bwd_dirs(f32(sat), (int32_t)H, (int32_t)W, f32(dirs), (int32_t)ld, f32(sa), R, (float)mipbias,
                                  static_cast<const float*>(vptr(sc)), i4 ? 1 : 0, f32(d_out), static_cast<float*>(d_pole.data_ptr()), out(d_dirs),
                                  static_cast<float*>(vptr(d_mip)), st(stream)),
          "nmf_sat_lookup_bwd_dirs");
    return d_dirs;
}
void sat_lookup_bwd_table(const Tensor& sat, const Tensor& dirs, const Tensor& sa, double mipbias, const Tensor& d_out,
                          const Tensor& d_sat, const OT& sc, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const int64_t R = dirs.size(0), ld = dirs.size(1);
    const bool i4 = sat.dim() == 3 && sat.size(-1) == 4 && sat.size(0) != 3;
    const int64_t H = i4 ? sat.size(0) : sat.size(-2), W = i4 ? sat.size(1) : sat.size(-1);
    if (!d_out.is_contiguous()) fail("sat_lookup_bwd_table: d_out must be contiguous");
    const int64_t nbytes = nmf_sat_lookup_bwd_workspace_bytes(R);
    Tensor ws = torch::empty({nbytes}, dirs.options().dtype(torch::kUInt8));      // (torch's current stream = `stream`: see StepCore::fork)
    check(nmf_sat_lookup_bwd_binned(f32(sat), (int32_t)H, (int32_t)W, f32(dirs), (int32_t)ld, f32(sa), R, (float)mipbias,
                                    static_cast<const float*>(vptr(sc)), i4 ? 1 : 0, f32(d_out), static_cast<float*>(d_sat.data_ptr()), nullptr, nullptr, nullptr,
                                    ws.data_ptr(), nbytes, st(stream)),
          "nmf_sat_lookup_bwd_binned");
}

Tensor brdf_mlp_bwd(const std::vector<Tensor>& w, const Tensor& half_vec, const Tensor& diff_vec, const Tensor& feat_src,
                    const Tensor& rough_src, const OT& src_idx, const Tensor& fwd_out, const Tensor& act_mask,
                    const Tensor& d_out, const std::vector<Tensor>& grads, int64_t max_workgroups, int64_t stream,
                    const OT& image = c10::nullopt) {
    TimedScope _ts(__func__, stream);
    const bool packed = image.has_value() && image->defined();
    if ((!packed && w.size() != 6) || grads.size() != 6) fail("brdf_mlp_bwd: six weight / gradient tensors expected");
    const int64_t R = half_vec.size(0);
    Tensor d_feat = at::zeros({feat_src.size(0), 24}, half_vec.options().dtype(at::kFloat));      // summed per row of feat_src
    Tensor go = d_out.contiguous();
    float* g[6];
    for (int i = 0; i < 6; ++i) g[i] = static_cast<float*>(vptr(grads[i]));
    const int64_t nws = nmf_brdf_mlp_bwd_workspace_bytes(R, (int32_t)max_workgroups);
    Tensor ws = fe(half_vec, {std::max<int64_t>(nws, 4) / 4});
    if (packed)
        check(nmf_brdf_mlp_bwd_packed(image->data_ptr(), f32(half_vec), f32(diff_vec), f32(feat_src), f32(rough_src),
                                      optr<const int32_t>(src_idx, at::kInt), R, f32(fwd_out),
                                      static_cast<const uint32_t*>(act_mask.data_ptr()), f32(go), out(d_feat), g[0], g[1], g[2], g[3],
                                      g[4], g[5], (int32_t)max_workgroups, out(ws), nws, st(stream)),
              "nmf_brdf_mlp_bwd_packed");
    else
        check(nmf_brdf_mlp_bwd(f32(w[0]), f32(w[1]), f32(w[2]), f32(w[3]), f32(w[4]), f32(w[5]), f32(half_vec), f32(diff_vec),
                               f32(feat_src), f32(rough_src), optr<const int32_t>(src_idx, at::kInt), R, f32(fwd_out),
                               static_cast<const uint32_t*>(act_mask.data_ptr()), f32(go), out(d_feat), g[0], g[1], g[2], g[3], g[4],
                               g[5], (int32_t)max_workgroups, out(ws), nws, st(stream)),
              "nmf_brdf_mlp_bwd");
    return d_feat;
}

// the same over ONE OR TWO ray sets in one launch (nmf_brdf_mlp_bwd_segments); a set = {half_vec, diff_vec, feat_src, rough_src,
// src_idx, fwd_out, act_mask, d_out}; -> d_feat per set
using MlpSet = std::tuple<Tensor, Tensor, Tensor, Tensor, OT, Tensor, Tensor, Tensor>;
std::vector<Tensor> brdf_mlp_bwd_sets(const std::vector<Tensor>& w, const std::vector<MlpSet>& sets, const std::vector<Tensor>& grads,
                                      int64_t max_workgroups, int64_t stream, const OT& image = c10::nullopt) {
    TimedScope _ts("brdf_mlp_bwd", stream);
    const bool packed = image.has_value() && image->defined();
    if ((!packed && w.size() != 6) || grads.size() != 6) fail("brdf_mlp_bwd_sets: six weight / gradient tensors expected");
    if (sets.empty() || sets.size() > 2) fail("brdf_mlp_bwd_sets: one or two ray sets");
    nmf_mlp_bwd_segment arr[2];
    int64_t Rs[2] = {0, 0};
    std::vector<Tensor> outs, keep;
    for (size_t i = 0; i < sets.size(); ++i) {
        const Tensor& hv = std::get<0>(sets[i]);
        const Tensor& feat = std::get<2>(sets[i]);
        Tensor d_feat = at::zeros({feat.size(0), 24}, hv.options().dtype(at::kFloat));
        Tensor go = std::get<7>(sets[i]).contiguous();
        keep.push_back(go);
        outs.push_back(d_feat);
        arr[i] = nmf_mlp_bwd_segment{f32(hv), f32(std::get<1>(sets[i])), f32(feat), f32(std::get<3>(sets[i])),
                                     optr<const int32_t>(std::get<4>(sets[i]), at::kInt), hv.size(0), f32(std::get<5>(sets[i])),
                                     static_cast<const uint32_t*>(std::get<6>(sets[i]).data_ptr()), f32(go), out(d_feat)};
        Rs[i] = hv.size(0);
    }
    float* g[6];
    for (int i = 0; i < 6; ++i) g[i] = static_cast<float*>(vptr(grads[i]));
    const int64_t nws = nmf_brdf_mlp_bwd_segments_workspace_bytes(Rs, (int32_t)sets.size(), (int32_t)max_workgroups);
    Tensor ws = fe(std::get<0>(sets[0]), {std::max<int64_t>(nws, 4) / 4});
    check(nmf_brdf_mlp_bwd_segments(packed ? image->data_ptr() : nullptr, packed ? nullptr : f32(w[0]), packed ? nullptr : f32(w[1]),
                                    packed ? nullptr : f32(w[2]), packed ? nullptr : f32(w[3]), packed ? nullptr : f32(w[4]),
                                    packed ? nullptr : f32(w[5]), arr, (int32_t)sets.size(), g[0], g[1], g[2], g[3], g[4], g[5],
                                    (int32_t)max_workgroups, out(ws), nws, st(stream)),
          "nmf_brdf_mlp_bwd_segments");
    return outs;
}

Tensor heads_bwd(const Tensor& feat, const Tensor& W, const Tensor& b, const std::vector<double>& hp, const Tensor& d_out,
                 const Tensor& gW, const Tensor& gb, const OT& add_into, int64_t stream) {
    // add_into: another adjoint of the same rows [M,24] (dense fp32); the result is added to it IN PLACE and it is returned
    TimedScope _ts(__func__, stream);
    if (hp.size() != 5) fail("heads_bwd: hp = (diffuse_mul, diffuse_bias, tint_bias, f0_bias, rough_bias)");
    const int64_t M = feat.size(0);
    if (add_into.has_value() && (add_into->scalar_type() != at::kFloat || !add_into->is_contiguous() || add_into->numel() != feat.numel()))
        fail("heads_bwd: add_into must be a dense float32 [M,24] tensor");
    Tensor d_feat = add_into.has_value() ? *add_into : at::empty_like(feat);
    Tensor go = d_out.contiguous();
    check(nmf_heads_bwd(f32(feat), M, f32(W), f32(b), (float)hp[0], (float)hp[1], (float)hp[2], (float)hp[3], (float)hp[4],
                        f32(go), add_into.has_value() ? f32(d_feat) : nullptr, out(d_feat), static_cast<float*>(vptr(gW)),
                        static_cast<float*>(vptr(gb)), st(stream)),
          "nmf_heads_bwd");
    return d_feat;
}

// photometric term + constant adjoints of a chunk in one launch -> (loss 0-d, d_pred [B,3], g_a [B], g_b [B]);
// ws: uint8 workspace of nmf_loss_head_workspace_bytes(B) bytes created with zeros (hip.loss_head_workspace)
std::tuple<Tensor, Tensor, Tensor, Tensor> loss_head(const Tensor& pred, const Tensor& gt, const Tensor& d_out, double scale,
                                                     double w_pred, double w_a, double w_b, const Tensor& ws, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const int64_t B = pred.size(0);
    Tensor loss = at::empty({}, pred.options().dtype(at::kFloat));
    Tensor d_pred = at::empty_like(pred), g_a = fe(pred, {B}), g_b = fe(pred, {B});
    check(nmf_loss_head(f32(pred), f32(gt), B, f32(d_out), (float)scale, (float)w_pred, (float)w_a, (float)w_b, out(loss),
                        out(d_pred), out(g_a), out(g_b), ws.data_ptr(), ws.numel() * ws.element_size(), st(stream)),
          "nmf_loss_head");
    return {loss, d_pred, g_a, g_b};
}

Tensor bg_adjoint(const Tensor& acc, const Tensor& d_rgb, int64_t stream) {
    TimedScope _ts(__func__, stream);
    Tensor d_bg = at::empty_like(d_rgb);
    Tensor go = d_rgb.contiguous();
    check(nmf_bg_adjoint(f32(acc), f32(go), acc.size(0), out(d_bg), st(stream)), "nmf_bg_adjoint");
    return d_bg;
}

Tensor ggx_rays_bwd(const Tensor& V, const Tensor& N, const Tensor& r, const Tensor& off, const Tensor& sobol,
                    const Tensor& row_of_ray, const Tensor& j_of_ray, const OT& dL, const OT& d_rays, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const int64_t R = row_of_ray.size(0);
    Tensor d_nr = fe(V, {R, 4});
    check(nmf_ggx_rays_bwd(f32(V), f32(N), f32(r), f32(off), f32(sobol), i32(row_of_ray), i32(j_of_ray), R,
                           static_cast<const float*>(vptr(dL)), static_cast<const float*>(vptr(d_rays)), out(d_nr), st(stream)),
          "nmf_ggx_rays_bwd");
    return d_nr;
}

std::tuple<Tensor, Tensor, Tensor, Tensor> shade_mix_bwd(const Tensor& V, const Tensor& f0, const Tensor& diff, const Tensor& cnt,
                                                         const Tensor& row_of_ray, const Tensor& L, const Tensor& inc,
                                                         const Tensor& brdf, const Tensor& d_rows, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const int64_t R = row_of_ray.size(0);
    Tensor d_inc = fe(V, {R, 3}), d_brdf = fe(V, {R, 3}), dL = fe(V, {R, 3}), d_fd = fe(V, {R, 6});
    check(nmf_shade_mix_bwd(f32(V), f32(f0), f32(diff), i32(cnt), i32(row_of_ray), R, f32(L), f32(inc), f32(brdf), f32(d_rows),
                            out(d_inc), out(d_brdf), out(dL), out(d_fd), st(stream)),
          "nmf_shade_mix_bwd");
    return {d_inc, d_brdf, dL, d_fd};
}

std::tuple<Tensor, OT, OT> ray_compose_bwd(const Tensor& weight, const OT& refl_rows, const OT& inv, const OT& normals,
                                           const Tensor& rays, const Tensor& ray_id, const Tensor& bg, bool bg_per_ray,
                                           bool tonemap, bool noclip, const OT& rgb_lin, const OT& d_rgb_map, const OT& d_acc,
                                           const OT& d_ori, bool want_d_normals, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const int64_t M = weight.size(0);
    Tensor d_weight = fe(weight, {M});
    OT d_refl, d_normals;
    if (refl_rows.has_value()) d_refl = at::empty_like(*refl_rows);
    if (want_d_normals) d_normals = fe(weight, {M, 3});
    if (M)
        check(nmf_ray_compose_bwd(f32(weight), static_cast<const float*>(vptr(refl_rows)), static_cast<const int32_t*>(vptr(inv)),
                                  static_cast<const float*>(vptr(normals)), f32(rays), i32(ray_id), M, f32(bg),
                                  bg_per_ray ? 1 : 0, tonemap ? 1 : 0, noclip ? 1 : 0, static_cast<const float*>(vptr(rgb_lin)),
                                  static_cast<const float*>(vptr(d_rgb_map)), static_cast<const float*>(vptr(d_acc)),
                                  static_cast<const float*>(vptr(d_ori)), out(d_weight),
                                  d_refl.has_value() ? out(*d_refl) : nullptr,
                                  d_normals.has_value() ? out(*d_normals) : nullptr, st(stream)),
              "nmf_ray_compose_bwd");
    return {d_weight, d_refl, d_normals};
}

// ---- field backward (FieldGrads.backward: two walks + the unpack, ~180 us of Python per step) -------------------------
using Seg = std::tuple<Tensor, OT, OT, OT, OT, OT, OT>;     // xyzt, sigma_feat, grad, d_sigma, d_sigma_feat, d_normal, d_app

// the brick sort of a walk, built from the positions alone (nmf_vm_bin_plan): -> opaque plan buffer
Tensor vm_bin_plan(int64_t p_addr, const std::vector<Tensor>& xyzts, int64_t stream, const OT& into = OT()) {
    TimedScope _ts(__func__, stream);
    const auto* p = reinterpret_cast<const nmf_vm_params*>(p_addr);
    if (xyzts.empty() || xyzts.size() > NMF_VM_MAX_SEGMENTS) fail("vm_bin_plan: 1.." + std::to_string(NMF_VM_MAX_SEGMENTS) + " segments");
    const float* ptrs[NMF_VM_MAX_SEGMENTS];
    int64_t Ms[NMF_VM_MAX_SEGMENTS], M = 0;
    for (size_t i = 0; i < xyzts.size(); ++i) {
        ptrs[i] = f32(xyzts[i]);
        Ms[i] = xyzts[i].size(0);
        M += Ms[i];
    }
    const int64_t nbytes = nmf_vm_bin_plan_bytes(M, p->grid);
    Tensor plan = into.has_value() ? *into : ie(xyzts[0], {(nbytes + 3) / 4}, at::kInt);
    if (plan.numel() * plan.element_size() < nbytes) fail("vm_bin_plan: the buffer passed in is too small");
    check(nmf_vm_bin_plan(p, ptrs, Ms, (int32_t)xyzts.size(), plan.data_ptr(), nbytes, st(stream)), "nmf_vm_bin_plan");
    return plan;
}

void vm_query_bwd_impl(const char* name, int64_t p_addr, const std::vector<Seg>& segs, const std::vector<Tensor>& dpk,
                       const std::vector<Tensor>& dlk, const std::vector<Tensor>& apl, const std::vector<Tensor>& ali,
                       const OT& basis, const std::vector<Tensor>& g_dpk, const std::vector<Tensor>& g_dlk,
                       const std::vector<Tensor>& g_apl, const std::vector<Tensor>& g_ali, const OT& g_basis, const OT& plan,
                       int64_t stream, const OT& clean = c10::nullopt) {
    // clean: the kept zero scratch of nmf_vm_query_bwd_segments_clean (a walk without a plan)
    TimedScope _ts(name, stream);
    const auto* p = reinterpret_cast<const nmf_vm_params*>(p_addr);
    if (segs.size() > NMF_VM_MAX_SEGMENTS) fail("at most " + std::to_string(NMF_VM_MAX_SEGMENTS) + " segments per walk");
    nmf_vm_bwd_segment arr[NMF_VM_MAX_SEGMENTS];
    int64_t M = 0;
    bool want_d = false, want_a = false;
    for (size_t i = 0; i < segs.size(); ++i) {
        const Tensor& x = std::get<0>(segs[i]);
        arr[i].xyzt = f32(x);
        arr[i].M = x.size(0);
        arr[i].sigma_feat = static_cast<const float*>(vptr(std::get<1>(segs[i])));
        arr[i].grad = static_cast<const float*>(vptr(std::get<2>(segs[i])));
        arr[i].d_sigma = static_cast<const float*>(vptr(std::get<3>(segs[i])));
        arr[i].d_sigma_feat = static_cast<const float*>(vptr(std::get<4>(segs[i])));
        arr[i].d_normal = static_cast<const float*>(vptr(std::get<5>(segs[i])));
        arr[i].d_app = static_cast<const float*>(vptr(std::get<6>(segs[i])));
        M += arr[i].M;
        want_d = want_d || std::get<3>(segs[i]).has_value() || std::get<4>(segs[i]).has_value() || std::get<5>(segs[i]).has_value();
        want_a = want_a || std::get<6>(segs[i]).has_value();
    }
    if (M == 0) return;
    const int64_t nbytes = plan.has_value() ? nmf_vm_walk_workspace_bytes(M) : nmf_vm_bwd_workspace_bytes(M, p->grid);
    Tensor ws = ie(std::get<0>(segs[0]), {(nbytes + 3) / 4}, at::kInt);
    P3 a{}, b{}, c{}, d{};
    float *ga[3] = {nullptr, nullptr, nullptr}, *gb[3] = {nullptr, nullptr, nullptr}, *gc[3] = {nullptr, nullptr, nullptr},
          *gd[3] = {nullptr, nullptr, nullptr};
    auto three_out = [](const std::vector<Tensor>& ts, float* (&o)[3]) {
        if (ts.size() != 3) fail("expected three tensors");
        for (int i = 0; i < 3; ++i) o[i] = const_cast<float*>(f32(ts[i]));
    };
    if (want_d) { a = three(dpk); b = three(dlk); three_out(g_dpk, ga); three_out(g_dlk, gb); }
    if (want_a) { c = three(apl); d = three(ali); three_out(g_apl, gc); three_out(g_ali, gd); }
    if (plan.has_value())
        check(nmf_vm_query_bwd_planned(p, arr, (int32_t)segs.size(), want_d ? a.p : nullptr, want_d ? b.p : nullptr,
                                       want_a ? c.p : nullptr, want_a ? d.p : nullptr, want_a ? of32(basis) : nullptr,
                                       want_d ? ga : nullptr, want_d ? gb : nullptr, want_a ? gc : nullptr, want_a ? gd : nullptr,
                                       want_a ? static_cast<float*>(vptr(g_basis)) : nullptr, vptr(plan), plan->numel() * 4,
                                       ws.data_ptr(), nbytes, st(stream)),
              "nmf_vm_query_bwd_planned");
    else if (clean.has_value() && clean->defined())
        check(nmf_vm_query_bwd_segments_clean(p, arr, (int32_t)segs.size(), want_d ? a.p : nullptr, want_d ? b.p : nullptr,
                                              want_a ? c.p : nullptr, want_a ? d.p : nullptr, want_a ? of32(basis) : nullptr,
                                              want_d ? ga : nullptr, want_d ? gb : nullptr, want_a ? gc : nullptr, want_a ? gd : nullptr,
                                              want_a ? static_cast<float*>(vptr(g_basis)) : nullptr, clean->data_ptr(), clean->numel(),
                                              ws.data_ptr(), nbytes, st(stream)),
              "nmf_vm_query_bwd_segments_clean");
    else
        check(nmf_vm_query_bwd_segments(p, arr, (int32_t)segs.size(), want_d ? a.p : nullptr, want_d ? b.p : nullptr,
                                        want_a ? c.p : nullptr, want_a ? d.p : nullptr, want_a ? of32(basis) : nullptr,
                                        want_d ? ga : nullptr, want_d ? gb : nullptr, want_a ? gc : nullptr, want_a ? gd : nullptr,
                                        want_a ? static_cast<float*>(vptr(g_basis)) : nullptr, ws.data_ptr(), nbytes, st(stream)),
              "nmf_vm_query_bwd_segments");
}

void vm_query_bwd_segments(int64_t p_addr, const std::vector<Seg>& segs, const std::vector<Tensor>& dpk,
                           const std::vector<Tensor>& dlk, const std::vector<Tensor>& apl, const std::vector<Tensor>& ali,
                           const OT& basis, const std::vector<Tensor>& g_dpk, const std::vector<Tensor>& g_dlk,
                           const std::vector<Tensor>& g_apl, const std::vector<Tensor>& g_ali, const OT& g_basis,
                           int64_t stream) {
    vm_query_bwd_impl(__func__, p_addr, segs, dpk, dlk, apl, ali, basis, g_dpk, g_dlk, g_apl, g_ali, g_basis, OT(), stream);
}

void vm_query_bwd_clean(int64_t p_addr, const std::vector<Seg>& segs, const std::vector<Tensor>& dpk,
                        const std::vector<Tensor>& dlk, const std::vector<Tensor>& apl, const std::vector<Tensor>& ali,
                        const OT& basis, const std::vector<Tensor>& g_dpk, const std::vector<Tensor>& g_dlk,
                        const std::vector<Tensor>& g_apl, const std::vector<Tensor>& g_ali, const OT& g_basis,
                        const Tensor& clean, int64_t stream) {
    if (clean.scalar_type() != at::kByte || !clean.is_contiguous() || !clean.is_cuda()) fail("vm_query_bwd_clean: scratch must be a device uint8 tensor");
    vm_query_bwd_impl("vm_query_bwd_segments", p_addr, segs, dpk, dlk, apl, ali, basis, g_dpk, g_dlk, g_apl, g_ali, g_basis, OT(), stream,
                      OT(clean));
}

void vm_query_bwd_planned(int64_t p_addr, const std::vector<Seg>& segs, const std::vector<Tensor>& dpk,
                          const std::vector<Tensor>& dlk, const std::vector<Tensor>& apl, const std::vector<Tensor>& ali,
                          const OT& basis, const std::vector<Tensor>& g_dpk, const std::vector<Tensor>& g_dlk,
                          const std::vector<Tensor>& g_apl, const std::vector<Tensor>& g_ali, const OT& g_basis,
                          const Tensor& plan, int64_t stream) {
    vm_query_bwd_impl(__func__, p_addr, segs, dpk, dlk, apl, ali, basis, g_dpk, g_dlk, g_apl, g_ali, g_basis, OT(plan), stream);
}

std::tuple<std::vector<Tensor>, std::vector<Tensor>> vm_unpack_density_grad(int64_t p_addr, const std::vector<Tensor>& g_dpk,
                                                                            const std::vector<Tensor>& g_dlk, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const auto* p = reinterpret_cast<const nmf_vm_params*>(p_addr);
    const int64_t G = p->grid;
    P3 a = three(g_dpk), b = three(g_dlk);
    std::vector<Tensor> gp, gl;
    float *op[3], *ol[3];
    for (int i = 0; i < 3; ++i) {
        // parameter-shaped ([1,16,G,G] / [1,16,G,1]) with channel-last strides: the storage is the [G][G][16] / [G][16] the
        // kernel writes, and autograd gets the gradient without a chain of permute / unsqueeze views
        gp.push_back(at::empty({1, 16, G, G}, g_dpk[0].options().dtype(at::kFloat).memory_format(at::MemoryFormat::ChannelsLast)));
        gl.push_back(at::empty({1, 16, G, 1}, g_dpk[0].options().dtype(at::kFloat).memory_format(at::MemoryFormat::ChannelsLast)));
        op[i] = out(gp[i]);
        ol[i] = out(gl[i]);
    }
    check(nmf_vm_unpack_density_grad(p, a.p, b.p, op, ol, st(stream)), "nmf_vm_unpack_density_grad");
    return {gp, gl};
}

// ---- round 2: the remaining per-step wrappers of the training pass (the host side of a step is on the critical path) ----
// (pointer, row pitch in floats) of a [n,width] / [n] fp32 tensor whose rows are dense but may be a column slice (hip._rows)
std::pair<const float*, int> rows_of(const OT& t, int width, std::vector<Tensor>& keep) {
    if (!t.has_value()) return {nullptr, width};
    Tensor x = *t;
    if (x.scalar_type() != at::kFloat || !x.is_cuda()) fail("expected a float32 device tensor");
    const bool ok = (x.dim() == 2 && x.size(1) == width && x.stride(1) == 1) || (x.dim() == 1 && width == 1);
    if (!ok || x.stride(0) < width) {
        x = x.contiguous();
        keep.push_back(x);
    }
    return {static_cast<const float*>(x.data_ptr()), (int)(x.size(0) > 1 ? x.stride(0) : width)};
}

std::tuple<Tensor, Tensor, Tensor> vm_query_rows(int64_t p_addr, const Tensor& xyzt, const std::vector<Tensor>& dpk,
                                                 const std::vector<Tensor>& dlk, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const auto* p = reinterpret_cast<const nmf_vm_params*>(p_addr);
    const int64_t M = xyzt.size(0);
    if (dpk.size() != 3 || dlk.size() != 3) fail("vm_query_rows: three planes / lines expected");
    const bool bf16 = dpk[0].scalar_type() == at::kBFloat16;
    const void *a[3], *b[3];
    for (int i = 0; i < 3; ++i) {
        if (dpk[i].scalar_type() != dpk[0].scalar_type() || dlk[i].scalar_type() != dpk[0].scalar_type())
            fail("vm_query_rows: mixed table dtypes");
        a[i] = vptr(dpk[i]);
        b[i] = vptr(dlk[i]);
    }
    Tensor sf = fe(xyzt, {M}), gr = fe(xyzt, {M, 3}), nr = fe(xyzt, {M, 3});
    check(nmf_vm_query_rows(p, f32(xyzt), M, a, b, bf16 ? 1 : 0, out(sf), nullptr, out(gr), out(nr), st(stream)),
          "nmf_vm_query_rows");
    return {sf, gr, nr};
}

// density value of all samples from the density factors themselves: -> (sigma_feat [M], sigma [M])
std::tuple<Tensor, Tensor> vm_query_sigma(int64_t p_addr, const Tensor& xyzt, const std::vector<Tensor>& planes,
                                          const std::vector<Tensor>& lines, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const auto* p = reinterpret_cast<const nmf_vm_params*>(p_addr);
    const int64_t M = xyzt.size(0);
    if (planes.size() != 3 || lines.size() != 3) fail("vm_query_sigma: three planes / lines expected");
    const bool bf16 = planes[0].scalar_type() == at::kBFloat16;
    const void *a[3], *b[3];
    for (int i = 0; i < 3; ++i) {
        if (planes[i].scalar_type() != planes[0].scalar_type() || lines[i].scalar_type() != planes[0].scalar_type())
            fail("vm_query_sigma: mixed table dtypes");
        if (planes[i].size(-1) != 16 || lines[i].size(-1) != 16 || !planes[i].is_contiguous() || !lines[i].is_contiguous())
            fail("vm_query_sigma: density factors [G,G,16] / [G,16] (channel-last views) expected");
        a[i] = vptr(planes[i]);
        b[i] = vptr(lines[i]);
    }
    Tensor sf = fe(xyzt, {M}), sg = fe(xyzt, {M});
    check(nmf_vm_query_sigma(p, f32(xyzt), M, a, b, bf16 ? 1 : 0, out(sf), out(sg), st(stream)), "nmf_vm_query_sigma");
    return {sf, sg};
}

Tensor sqerr_fwd(const Tensor& pred, const Tensor& gt, int64_t stream) {
    TimedScope _ts(__func__, stream);
    Tensor o = at::zeros({}, pred.options().dtype(at::kFloat));
    if (pred.numel()) check(nmf_sqerr_fwd(f32(pred), f32(gt), pred.numel(), out(o), st(stream)), "nmf_sqerr_fwd");
    return o;
}

Tensor sqerr_bwd(const Tensor& pred, const Tensor& gt, const Tensor& d_out, int64_t stream) {
    TimedScope _ts(__func__, stream);
    Tensor d = at::empty_like(pred);
    if (pred.numel()) check(nmf_sqerr_bwd(f32(pred), f32(gt), pred.numel(), f32(d_out), out(d), st(stream)), "nmf_sqerr_bwd");
    return d;
}

std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> shade_mix_bwd_view(const Tensor& V, const Tensor& f0, const Tensor& diff,
                                                                      const Tensor& cnt, const Tensor& row_of_ray,
                                                                      const Tensor& L, const Tensor& inc, const Tensor& brdf,
                                                                      const Tensor& d_rows, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const int64_t R = row_of_ray.size(0);
    Tensor d_inc = fe(V, {R, 3}), d_brdf = fe(V, {R, 3}), dL = fe(V, {R, 3}), d_fd = fe(V, {R, 6}), dV = fe(V, {R, 3});
    check(nmf_shade_mix_bwd_view(f32(V), f32(f0), f32(diff), i32(cnt), i32(row_of_ray), R, f32(L), f32(inc), f32(brdf),
                                 f32(d_rows), out(d_inc), out(d_brdf), out(dL), out(d_fd), out(dV), st(stream)),
          "nmf_shade_mix_bwd_view");
    return {d_inc, d_brdf, dL, d_fd, dV};
}

Tensor ggx_rays_bwd_view(const Tensor& V, const Tensor& N, const Tensor& r, const Tensor& off, const Tensor& sobol,
                         const Tensor& row_of_ray, const Tensor& j_of_ray, const Tensor& dL, const OT& d_rays,
                         int64_t stream) {
    TimedScope _ts(__func__, stream);
    const int64_t R = row_of_ray.size(0);
    Tensor d = fe(V, {R, 7});
    check(nmf_ggx_rays_bwd_view(f32(V), f32(N), f32(r), f32(off), f32(sobol), i32(row_of_ray), i32(j_of_ray), R,
                                static_cast<const float*>(vptr(dL)), static_cast<const float*>(vptr(d_rays)), out(d),
                                st(stream)),
          "nmf_ggx_rays_bwd_view");
    return d;
}

void view_adjoint_to_rays(const Tensor& ray_id, const Tensor& bidx, const Tensor& dv_a, const OT& dv_b, Tensor d_rays,
                          int64_t stream) {
    TimedScope _ts(__func__, stream);
    std::vector<Tensor> keep;
    const auto a = rows_of(dv_a, 3, keep), b = rows_of(dv_b, 3, keep);
    check(nmf_view_adjoint_to_rays(i32(ray_id), i32(bidx), a.first, a.second, b.first, b.second, bidx.size(0),
                                   ptr<float>(d_rays, at::kFloat), st(stream)),
          "nmf_view_adjoint_to_rays");
}

Tensor select_total(const Tensor& weights, const Tensor& u, double extra, Tensor ws, int64_t stream) {
    TimedScope _ts(__func__, stream);
    Tensor total = at::empty({}, weights.options().dtype(at::kFloat));
    check(nmf_select_total(f32(weights), f32(u), weights.size(0), extra, ptr<double>(ws, at::kDouble), out(total),
                           st(stream)),
          "nmf_select_total");
    return total;
}

std::tuple<Tensor, Tensor, Tensor> bounce_prep_bwd(const OT& inv, const Tensor& normals, const Tensor& heads, const Tensor& ray_id,
                                                   const Tensor& rays, const Tensor& conv, double min_rough, bool detach_n,
                                                   const OT& dN, const OT& dr1, const OT& df0, const OT& ddiff, const OT& dfeat,
                                                   const OT& bidx, int64_t row_inputs, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const int64_t M = inv.has_value() ? inv->size(0) : ray_id.size(0);
    const int64_t Mb = bidx.has_value() ? bidx->size(0) : 0;
    const int64_t n_out = row_inputs ? Mb : M;
    Tensor d_normals = fe(normals, {row_inputs == 2 ? Mb : M, 3}), d_heads = fe(normals, {n_out, 11});
    Tensor d_app = fe(normals, {n_out, 24});
    if (M) {
        std::vector<Tensor> keep;
        const auto a = rows_of(dN, 3, keep), b = rows_of(dr1, 1, keep), c = rows_of(df0, 3, keep), d = rows_of(ddiff, 3, keep);
        const int32_t strides[4] = {a.second, b.second, c.second, d.second};
        check(nmf_bounce_prep_bwd(static_cast<const int32_t*>(vptr(inv)), M, static_cast<const int32_t*>(vptr(bidx)), Mb,
                                  f32(normals), heads.size(0) ? f32(heads) : nullptr, i32(ray_id), f32(rays), f32(conv),
                                  (float)min_rough, detach_n ? 1 : 0, (int32_t)row_inputs, a.first, b.first, c.first, d.first,
                                  strides, static_cast<const float*>(vptr(dfeat)), out(d_normals), out(d_heads), out(d_app),
                                  st(stream)),
              "nmf_bounce_prep_bwd");
    }
    return {d_normals, d_heads, d_app};
}

// bounce_prep_bwd (row_inputs = 2) + heads_bwd in one launch (nmf_bounce_prep_heads_bwd) -> {d_normals [Mb,3], d_app [Mb,24]}
std::tuple<Tensor, Tensor> bounce_prep_heads_bwd(const Tensor& bidx, const Tensor& normals, const Tensor& heads, const Tensor& ray_id,
                                                 const Tensor& rays, const Tensor& conv, double min_rough, bool detach_n, const OT& dN,
                                                 const OT& dr1, const OT& df0, const OT& ddiff, const OT& dfeat, const Tensor& app,
                                                 const Tensor& W, const Tensor& b, const std::vector<double>& hp, const Tensor& gW,
                                                 const Tensor& gb, int64_t stream) {
    TimedScope _ts("heads_bwd", stream);
    if (hp.size() != 5) fail("bounce_prep_heads_bwd: hp = (diffuse_mul, diffuse_bias, tint_bias, f0_bias, rough_bias)");
    const int64_t Mb = bidx.size(0);
    Tensor d_normals = fe(normals, {Mb, 3}), d_app = fe(normals, {Mb, 24});
    if (Mb) {
        std::vector<Tensor> keep;
        const auto a = rows_of(dN, 3, keep), b2 = rows_of(dr1, 1, keep), c = rows_of(df0, 3, keep), d = rows_of(ddiff, 3, keep);
        const int32_t strides[4] = {a.second, b2.second, c.second, d.second};
        OT df;
        if (dfeat.has_value()) df = dfeat->contiguous();
        check(nmf_bounce_prep_heads_bwd(i32(bidx), Mb, f32(normals), f32(heads), i32(ray_id), f32(rays), f32(conv), (float)min_rough,
                                        detach_n ? 1 : 0, a.first, b2.first, c.first, d.first, strides, of32(df), f32(app), f32(W),
                                        f32(b), (float)hp[0], (float)hp[1], (float)hp[2], (float)hp[3], (float)hp[4], out(d_normals),
                                        out(d_app), static_cast<float*>(vptr(gW)), static_cast<float*>(vptr(gb)), st(stream)),
              "nmf_bounce_prep_heads_bwd");
    }
    return {d_normals, d_app};
}

// slot tables are ctypes arrays owned by the Python side: passed by address
void adam_step(int64_t slots_addr, int64_t n, const OT& guard, int64_t stream) {
    TimedScope _ts(__func__, stream);
    check(nmf_adam_step_guarded(reinterpret_cast<const nmf_adam_slot*>(slots_addr), (int32_t)n, of32(guard), st(stream)),
          "nmf_adam_step");
}
void multi_copy(int64_t slots_addr, int64_t n, int64_t stream) {
    TimedScope _ts(__func__, stream);
    check(nmf_multi_copy(reinterpret_cast<const nmf_copy_slot*>(slots_addr), (int32_t)n, st(stream)), "nmf_multi_copy");
}

const float* dense_f32(const Tensor& t) {       // hip._dense_f32: contiguous or channels-last storage
    if (t.scalar_type() != at::kFloat || !t.is_cuda()) fail("expected a float32 device tensor");
    if (!(t.is_contiguous() || t.is_contiguous(at::MemoryFormat::ChannelsLast))) fail("tensor storage must be dense");
    return static_cast<const float*>(t.data_ptr());
}

std::vector<Tensor> loss_mix_bwd(const std::vector<std::vector<int64_t>>& shapes, const std::vector<double>& weights,
                                 double scale, const Tensor& d_out, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const size_t n = shapes.size();
    if (weights.size() != n || n > 16) fail("loss_mix_bwd: shapes / weights");
    std::vector<Tensor> grads;
    int64_t numel[16];
    float w[16];
    float* g[16];
    for (size_t i = 0; i < n; ++i) {
        grads.push_back(at::empty(shapes[i], d_out.options().dtype(at::kFloat)));
        numel[i] = grads[i].numel();
        w[i] = (float)weights[i];
        g[i] = out(grads[i]);
    }
    check(nmf_loss_mix_bwd(numel, w, (int32_t)n, (float)scale, f32(d_out), g, st(stream)), "nmf_loss_mix_bwd");
    return grads;
}

void l1_mean_bwd_into(const std::vector<Tensor>& tensors, const Tensor& d_out, std::vector<Tensor> grads, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const size_t n = tensors.size();
    if (grads.size() != n || n > 32) fail("l1_mean_bwd: tensors / gradients");
    const float* x[32];
    float* g[32];
    int64_t numel[32];
    for (size_t i = 0; i < n; ++i) {
        if (grads[i].numel() != tensors[i].numel()) fail("l1_mean_bwd: gradient / tensor size mismatch");
        x[i] = dense_f32(tensors[i]);
        g[i] = const_cast<float*>(dense_f32(grads[i]));
        numel[i] = tensors[i].numel();
    }
    check(nmf_l1_mean_bwd(x, numel, (int32_t)n, f32(d_out), g, 1, st(stream)), "nmf_l1_mean_bwd");
}

void sat_build_bwd_into(Tensor d_sat, const Tensor& bg, const Tensor& act, const OT& d_pole, double brightness, double mul,
                        const OT& sc, Tensor d_bg, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const int64_t H = bg.size(-2), W = bg.size(-1);
    check(nmf_sat_build_bwd(ptr<float>(d_sat, at::kFloat), f32(bg), f32(act), (int32_t)H, (int32_t)W, (float)brightness,
                            (float)mul, static_cast<const float*>(vptr(sc)), static_cast<const float*>(vptr(d_pole)),
                            ptr<float>(d_bg, at::kFloat), st(stream)),
          "nmf_sat_build_bwd");
}

void sat_build_into(const Tensor& bg, double brightness, double mul, const OT& sc, Tensor act, Tensor sat, const OT& pole,
                    const OT& sat_i4, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const int64_t H = bg.size(-2), W = bg.size(-1);
    check(nmf_sat_build(f32(bg), (int32_t)H, (int32_t)W, (float)brightness, (float)mul, static_cast<const float*>(vptr(sc)),
                        ptr<float>(act, at::kFloat), ptr<float>(sat, at::kFloat), static_cast<float*>(vptr(pole)),
                        static_cast<float*>(vptr(sat_i4)), st(stream)),
          "nmf_sat_build");
}

void sh_project_into(const Tensor& vals, const Tensor& wq, const Tensor& sh_A, Tensor coeffs, Tensor conv, int64_t stream) {
    TimedScope _ts(__func__, stream);
    check(nmf_sh_project(f32(vals), f32(wq), wq.size(0), (int32_t)wq.size(1), f32(sh_A), ptr<float>(coeffs, at::kFloat),
                         ptr<float>(conv, at::kFloat), st(stream)),
          "nmf_sh_project");
}

void vm_pack_density_into(int64_t p_addr, const std::vector<Tensor>& planes, const std::vector<Tensor>& lines,
                          const std::vector<Tensor>& dpk, const std::vector<Tensor>& dlk, int64_t stream) {
    TimedScope _ts(__func__, stream);
    const auto* p = reinterpret_cast<const nmf_vm_params*>(p_addr);
    P3 a = three(planes), b = three(lines), c = three(dpk), d = three(dlk);
    float *oc[3], *od[3];
    for (int i = 0; i < 3; ++i) { oc[i] = const_cast<float*>(c.p[i]); od[i] = const_cast<float*>(d.p[i]); }
    check(nmf_vm_pack_density(p, a.p, b.p, oc, od, st(stream)), "nmf_vm_pack_density");
}

#include "step_core.inc"

}  // namespace

PYBIND11_MODULE(_nmf_host, m) {
    m.doc() = "host-side fast path of nmf_amd.hip (same C ABI underneath)";
    m.def("set_error_class", [](py::object cls) {
        g_error_class = cls.ptr();
        Py_XINCREF(g_error_class);
    });
    m.def("abi_version", []() { return (int)NMF_ABI_VERSION; });     // what THIS module was compiled against (hip.py compares)
    m.def("call_timing_begin", [](const std::string& only) {
        if (!g_call_timer) g_call_timer = new CallTimer();
        g_call_timer->next = 0;
        g_call_timer->recs.clear();
        g_call_filter = only;
    }, py::arg("only") = "");
    m.def("call_timing_end", []() {        // waits for the recorded work; -> {name: (ms, calls)}
        py::dict out;
        if (!g_call_timer) return out;
        CallTimer* t = g_call_timer;
        g_call_timer = nullptr;
        std::map<std::string, std::pair<double, int64_t>> acc;
        for (auto& r : t->recs) {
            float ms = 0.f;
            check(nmf_event_synchronize(r.b), "nmf_event_synchronize");
            check(nmf_event_elapsed_ms(r.a, r.b, &ms), "nmf_event_elapsed_ms");
            auto& e = acc[r.name];
            e.first += ms;
            e.second += 1;
        }
        for (auto& kv : acc) out[py::str(kv.first)] = py::make_tuple(kv.second.first, kv.second.second);
        delete t;
        return out;
    });
    m.def("kernel_timing_begin", [](const std::string& only, bool by_stream) {
        g_kernel_by_stream = by_stream;
        if (!g_kernel_timer) g_kernel_timer = new CallTimer();
        g_kernel_timer->next = 0;
        g_kernel_timer->recs.clear();
        g_kernel_filter = only;
        g_probe_open = nullptr;
        check(nmf_set_launch_probe(&launch_probe), "nmf_set_launch_probe");
    }, py::arg("only") = "", py::arg("by_stream") = false);
    m.def("kernel_timing_end", []() {      // waits for the recorded work; -> {kernel: (ms, launches)}
        py::dict out;
        check(nmf_set_launch_probe(nullptr), "nmf_set_launch_probe");
        if (!g_kernel_timer) return out;
        CallTimer* t = g_kernel_timer;
        g_kernel_timer = nullptr;
        std::map<std::string, std::pair<double, int64_t>> acc;
        for (auto& r : t->recs) {
            float ms = 0.f;
            check(nmf_event_synchronize(r.b), "nmf_event_synchronize");
            check(nmf_event_elapsed_ms(r.a, r.b, &ms), "nmf_event_elapsed_ms");
            auto& e = acc[kernel_name(r.name)];
            e.first += ms;
            e.second += 1;
            const std::string sid = "@" + std::to_string(reinterpret_cast<int64_t>(r.stream));
            auto& q = acc[sid];      // per-stream sums (the main stream = the chain) ...
            q.first += ms;
            q.second += 1;
            if (g_kernel_by_stream) {      // ... and per kernel and stream
                auto& w = acc[kernel_name(r.name) + sid];
                w.first += ms;
                w.second += 1;
            }
        }
        for (auto& kv : acc) out[py::str(kv.first)] = py::make_tuple(kv.second.first, kv.second.second);
        delete t;
        return out;
    });
    m.def("readback_wait_us", [](bool reset) { const double w = SizeReadback::waited_us(); if (reset) SizeReadback::waited_us() = 0.0; return w; });
    m.def("readback_wait_by_slot_us", [](bool reset) {
        std::vector<double> v(SizeReadback::waited_by_slot(), SizeReadback::waited_by_slot() + 4);
        if (reset) for (int i = 0; i < 4; ++i) SizeReadback::waited_by_slot()[i] = 0.0;
        return v;
    });
    m.def("set_call_delay", [](const std::string& name, double us) { g_delay_name = name; g_delay_us = us; });
    m.def("call_timing_timeline", []() {   // waits for the recorded work; -> [(name, stream, start_us, end_us, host_issue_us)], times
        py::list out;                      // relative to the first recorded call (device clock / host clock)
        if (!g_call_timer) return out;
        CallTimer* t = g_call_timer;
        g_call_timer = nullptr;
        if (!t->recs.empty()) {
            void* ref = t->recs[0].a;
            const double h0 = t->recs[0].host_us;
            for (auto& r : t->recs) {
                float s_ms = 0.f, e_ms = 0.f;
                check(nmf_event_synchronize(r.b), "nmf_event_synchronize");
                check(nmf_event_elapsed_ms(ref, r.a, &s_ms), "nmf_event_elapsed_ms");
                check(nmf_event_elapsed_ms(ref, r.b, &e_ms), "nmf_event_elapsed_ms");
                out.append(py::make_tuple(std::string(r.name), reinterpret_cast<int64_t>(r.stream), 1e3 * s_ms, 1e3 * e_ms, r.host_us - h0));
            }
        }
        delete t;
        return out;
    });
    m.def("march_count", &march_count);
    m.def("march_scan", &march_scan, py::arg("counts"), py::arg("max_samples"), py::arg("stream"), py::arg("pub") = 0, py::arg("pub_seq") = 0);
    m.def("march_fill", &march_fill);
    m.def("vm_query_fwd", [](int64_t p_addr, const Tensor& xyzt, const std::vector<Tensor>& dpk, const std::vector<Tensor>& dlk,
                             const std::vector<Tensor>& apl, const std::vector<Tensor>& ali, const OT& basis, bool want_density,
                             bool want_normal, bool want_app, bool want_coef, int64_t stream) {
        return vm_query_fwd(p_addr, xyzt, dpk, dlk, apl, ali, basis, want_density, want_normal, want_app, want_coef, stream, 0);
    });
    m.def("composite_fwd", &composite_fwd);
    m.def("segment_sum", &segment_sum);
    m.def("sat_lookup_fwd", &sat_lookup_fwd);
    m.def("select_bounces", &select_bounces);
    m.def("expand_segments", &expand_segments);
    m.def("brdf_mlp_fwd", &brdf_mlp_fwd, py::arg("w"), py::arg("half_vec"), py::arg("diff_vec"), py::arg("feat_src"), py::arg("rough_src"),
          py::arg("src_idx"), py::arg("out_bias"), py::arg("with_mask"), py::arg("max_workgroups"), py::arg("stream"),
          py::arg("image") = py::none());
    m.def("brdf_mlp_pack", &brdf_mlp_pack);
    m.def("bounce_prep_heads_bwd", &bounce_prep_heads_bwd);
    m.def("bounce_prep_fwd_heads", &bounce_prep_fwd_heads);
    m.def("brdf_mlp_bwd_sets", &brdf_mlp_bwd_sets, py::arg("w"), py::arg("sets"), py::arg("grads"), py::arg("max_workgroups"), py::arg("stream"),
          py::arg("image") = py::none());
    m.def("heads_fwd", &heads_fwd);
    m.def("ggx_rays_fwd", &ggx_rays_fwd);
    m.def("shade_mix_fwd", &shade_mix_fwd);
    m.def("bounce_index", &bounce_index, py::arg("counts"), py::arg("xyzt"), py::arg("stream"), py::arg("pub") = 0, py::arg("pub_seq") = 0,
          py::arg("live") = 0);
    m.def("bounce_index_select", [](const Tensor& w, const Tensor& u, int64_t mode, double mul, double add, double sum_w, const OT& sum_w_dev,
                                    const OT& xyzt, int64_t stream) {
        return bounce_index_select(w, u, mode, mul, add, sum_w, sum_w_dev, xyzt, stream);
    });
    m.def("bounce_prep_fwd", &bounce_prep_fwd);
    m.def("ray_compose_fwd", &ray_compose_fwd);
    m.def("composite_bwd", &composite_bwd);
    m.def("segment_sum_wide", &segment_sum_wide);
    m.def("sat_lookup_bwd", &sat_lookup_bwd);
    m.def("brdf_mlp_bwd", &brdf_mlp_bwd, py::arg("w"), py::arg("half_vec"), py::arg("diff_vec"), py::arg("feat_src"), py::arg("rough_src"),
          py::arg("src_idx"), py::arg("fwd_out"), py::arg("act_mask"), py::arg("d_out"), py::arg("grads"), py::arg("max_workgroups"),
          py::arg("stream"), py::arg("image") = py::none());
    m.def("heads_bwd", &heads_bwd);
    m.def("ggx_rays_bwd", &ggx_rays_bwd);
    m.def("shade_mix_bwd", &shade_mix_bwd);
    m.def("ray_compose_bwd", &ray_compose_bwd);
    m.def("vm_query_bwd_segments", &vm_query_bwd_segments);
    m.def("vm_query_bwd_planned", &vm_query_bwd_planned);
    m.def("vm_query_bwd_clean", &vm_query_bwd_clean);
    m.def("vm_bin_plan", &vm_bin_plan, py::arg("p_addr"), py::arg("xyzts"), py::arg("stream"), py::arg("into") = py::none());
    m.def("vm_unpack_density_grad", &vm_unpack_density_grad);
    m.def("adam_step", &adam_step);
    m.def("multi_copy", &multi_copy);
    m.def("loss_mix_bwd", &loss_mix_bwd);
    m.def("l1_mean_bwd_into", &l1_mean_bwd_into);
    m.def("sat_build_bwd_into", &sat_build_bwd_into);
    m.def("sat_build_into", &sat_build_into);
    m.def("sh_project_into", &sh_project_into);
    m.def("vm_pack_density_into", &vm_pack_density_into);
    m.def("vm_query_rows", &vm_query_rows);
    m.def("vm_query_sigma", &vm_query_sigma);
    m.def("sat_lookup_bwd_dirs", &sat_lookup_bwd_dirs);
    m.def("sat_lookup_bwd_table", &sat_lookup_bwd_table);
    m.def("loss_head", &loss_head);
    m.def("bg_adjoint", &bg_adjoint);
    m.def("sqerr_fwd", &sqerr_fwd);
    m.def("sqerr_bwd", &sqerr_bwd);
    m.def("shade_mix_bwd_view", &shade_mix_bwd_view);
    m.def("ggx_rays_bwd_view", &ggx_rays_bwd_view);
    m.def("view_adjoint_to_rays", &view_adjoint_to_rays);
    m.def("select_total", &select_total);
    m.def("bounce_prep_bwd", &bounce_prep_bwd);
    py::class_<StepCore>(m, "StepCore")
        .def(py::init<>())
        .def("chunk", &StepCore::chunk)
        .def("train_forward", &StepCore::train_forward)
        .def("train_backward", &StepCore::train_backward)
        .def("has_pending", &StepCore::has_pending)
        .def("drop_pending", &StepCore::drop_pending)
        .def("render", &StepCore::render)
        .def("begin_step", &StepCore::begin_step)
        .def("join_early_env", &StepCore::join_early_env)
        .def("env_was_used", &StepCore::env_was_used)
        .def("env_table_backward_queued", &StepCore::env_table_backward_queued)
#define RW(name) .def_readwrite(#name, &StepCore::name)
        RW(env_keep_sat) RW(env_split) RW(early_cb) RW(comm_stream) RW(peer_streams) RW(main_stream) RW(side_streams) RW(set_stream) RW(main_stream_obj) RW(side_stream_objs) RW(overlap) RW(sparse_normals)
        RW(mlp_side_min_rays) RW(mlp_side_min_env_rays) RW(mlp_side_wgs_env) RW(walk_side_min_samples) RW(mlp_side_wgs)
        RW(env_binned_from) RW(vm_p) RW(dpk) RW(dlk) RW(dpl) RW(dli) RW(f_dpk) RW(f_dlk) RW(f_apl) RW(f_ali) RW(apl) RW(ali) RW(basis) RW(head_p) RW(head_W) RW(head_b) RW(mlp_ws) RW(mlp_image)
        RW(mlp_bias) RW(sobol) RW(env_table) RW(env_pole) RW(env_sc) RW(env_act) RW(env_bg) RW(sh_conv) RW(march_p0) RW(march_p1)
        RW(max_samples) RW(alpha_bits) RW(alpha_coarse) RW(scale) RW(anoise) RW(min_rough) RW(rays_per_ray) RW(test_rays_per_ray)
        RW(detach_n) RW(max_brdf_rays) RW(max_retrace_rays) RW(white) RW(one) RW(select_ws) RW(g_dpk) RW(g_dlk) RW(g_apl) RW(g_ali)
        RW(g_mlp) RW(g_basis) RW(g_hW) RW(g_hb) RW(d_sat) RW(d_pole) RW(d_mip) RW(d_bg_out) RW(used_env) RW(sampler) RW(wait_tables) RW(wait_env)
#undef RW
        ;
}
