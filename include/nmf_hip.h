/*
 * nmf_hip.h -- C ABI of libnmf_hip.so, the MI355X (gfx950) kernels beneath the
 * microfacet_tensorf2 render/train hot path of half-potato/nmf.
 *
 * The reference has no FFI on this path (it is pure PyTorch); every entry point
 * below replaces a span of reference Python that the host-side mirror classes in nmf_amd/ call
 * instead (file:line of the reference given per function, relative to /root/reference).
 *
 * Conventions
 *   - extern "C", plain pointers, int64_t sizes, hipStream_t passed as void*.
 *   - The CALLER allocates every output (and zeroes the ones documented as "accumulated").
 *   - The library never allocates persistent memory, never frees and never synchronises the
 *     stream; one call = a fixed small number of kernel launches on the given stream.
 *   - Return value: 0 = ok, negative = bad argument (NMF_E*), positive = hipError_t.
 *     nmf_last_error_string() describes the last failure on the calling thread.
 *   - All device pointers must be valid on the current device (hipSetDevice by the caller).
 *   - fp32 throughout; tables are CHANNEL-LAST: plane [G_h][G_w][C], line [G][C]
 *     (= torch.channels_last storage of the reference's [1,C,G,G] / [1,C,G,1] parameters).
 */
#ifndef NMF_HIP_H
#define NMF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NMF_OK 0
#define NMF_EINVAL (-1)   /* null pointer / bad size */
#define NMF_ERANGE (-2)   /* size outside compiled limits (channels, steps ...) */

#define NMF_DENSITY_C 16  /* density_n_comp, configs/field/tensorf_og.yaml */
#define NMF_APP_C 24      /* appearance_n_comp */
#define NMF_APP_DIM 24    /* app_dim (basis_mat rows) */
#define NMF_MLP_IN 66     /* MLPBRDF input width, modules/brdf.py:73-120 */
#define NMF_MLP_HID 64

/* Bumped with every incompatible change of a prototype or of a workspace size.  nmf_version() returns the value the LIBRARY
 * was built with; a separately built caller (nmf_amd/lib/_nmf_host.so) compares it with the value it was compiled against. */
#define NMF_ABI_VERSION 116
int nmf_version(void);
const char* nmf_last_error_string(void);

/* Runtime plumbing for host-side drivers that are built without the HIP headers (nmf_amd/csrc/host_ext.cpp is plain g++ against
 * the torch headers): events without timing, stream ordering and the start of a size read-back into pinned host memory.  The
 * reference has no counterpart (its sizes come back through tensor.item() / .sum() host syncs, e.g. samplers/alphagrid.py:357).
 * nmf_event_synchronize is the one BLOCKING entry point of the library. */
int nmf_event_create(void** event);
int nmf_event_create_timed(void** event);                        /* with timestamps, for nmf_event_elapsed_ms */
int nmf_event_elapsed_ms(void* start, void* stop, float* ms);   /* both recorded and complete */
int nmf_event_destroy(void* event);
int nmf_event_record(void* event, void* stream);
int nmf_event_synchronize(void* event);
/* Size read-back without a copy command or an event (runtime plumbing like the calls above; the reference reads its sizes with
 * implicit .item() synchronisations, e.g. samplers/alphagrid.py:353-364, models/microfacet.py:318-331): the sizes are published by a
 * one-thread kernel into mapped, coherent host memory [value0, value1, seq] and the host spins on seq.  nmf_wait_seq is blocking. */
int nmf_host_alloc_mapped(void** host_ptr, void** dev_ptr, int64_t nbytes);
int nmf_host_free_mapped(void* host_ptr);
int nmf_publish_i64x2(const int64_t* src_dev, void* dst_mapped_dev, int64_t seq, void* stream);
int nmf_wait_seq(const void* host_ptr, int64_t seq, double timeout_s);
int nmf_stream_wait_event(void* stream, void* event);
int nmf_memcpy_d2h_async(void* dst_host, const void* src_dev, int64_t nbytes, void* stream);
/* Launch probe (measurement plumbing: bench.py's per-kernel HIP-event timing on the launching stream): when set, EVERY kernel launch
 * of the library calls probe(kernel_name, stream, 0) in front of and probe(kernel_name, stream, 1) behind the launch, on the calling
 * thread.  kernel_name is the kernel's identifier as rocprofv3 --kernel-trace prints it (without arguments).  NULL removes it. */
typedef void (*nmf_launch_probe_fn)(const char* kernel_name, void* stream, int phase);
int nmf_set_launch_probe(void (*probe)(const char* kernel_name, void* stream, int phase));

/* ------------------------------------------------------------------------------------------
 * Sampler: AlphaGridSampler.sample / sample_ray / AlphaGridMask.sample_alpha
 * (samplers/alphagrid.py:131-207, 23-45, 279-370).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    float aabb_min[3], aabb_max[3];
    float alpha_inv[3];   /* AlphaGridMask.invgrid_size = 1/aabbSize*2 (alphagrid.py:12)          */
    float stepsize;       /* rf.stepsize (fp32)                                                   */
    float half_step;      /* stepsize / 2 as torch computes it (alphagrid.py:171)                 */
    float near_t, far_t;  /* near_far, near replaced by override_near for secondary rays          */
    float focal;          /* 4th coordinate = t / focal (alphagrid.py:200)                        */
    int32_t n_steps;      /* N = nSamples (<= 4096)                                               */
    int32_t grid[3];      /* alpha volume size (x,y,z); 0 => no alpha mask                        */
    int32_t is_train;     /* 1: cumulative jitter (alphagrid.py:168-173); 0: stepsize*k (:190)    */
    uint64_t seed;        /* Philox seed used when jitter == NULL and is_train                    */
    uint64_t offset;      /* Philox stream offset (advance per call)                              */
    float occ_min[3], occ_max[3];  /* optional world-space box around every point whose 8-corner footprint can touch a
                                    * set alpha bit (tight box of the set voxels grown by one voxel + margin): the
                                    * marcher stops a ray once it has left this box (nothing can be kept beyond it).
                                    * occ_min > occ_max on any axis = not provided.  Results are unchanged.          */
} nmf_march_params;

/* float 0/1 volume [gz][gy][gx] -> bitfield, bit i of word i/32 = volume[i] > 0
 * (the ">0 after trilinear sampling" test of alphagrid.py:341-346 only needs occupancy bits). */
int nmf_alpha_pack(const float* volume, int64_t n_voxels, uint32_t* bits, void* stream);

/* Pass 1: per-ray validity bitmask ([B][W] uint64, W = ceil(N/64), bit k = step k kept) and
 * per-ray kept count.  jitter: [B][N] uniforms in [0,1) or NULL (Philox). */
int nmf_march_count(const nmf_march_params* p, const float* rays /*[B][6]*/, int64_t B,
                    const float* jitter, const uint32_t* alpha_bits, const uint32_t* alpha_coarse,
                    uint64_t* valid_bits, int32_t* counts, void* stream);
/* Optional accelerator of the occupancy test: one bit per 8^3-voxel cell = OR of the 9^3 fine bits any point of the
 * cell can touch (nmf_alpha_coarse_words(grid) uint32 words).  nmf_march_count stages it in LDS and skips the 8-corner
 * test where it is clear; results are unchanged.  alpha_coarse may be NULL. */
int nmf_alpha_coarse(const uint32_t* alpha_bits, const int32_t grid[3], uint32_t* coarse, void* stream);
int64_t nmf_alpha_coarse_words(const int32_t grid[3]);

/* Pass 2: exclusive scan of counts + the sample budget of alphagrid.py:353-364:
 * if max_samples > 0 and sum(counts) > max_samples then whole_valid[i] = cumsum(counts)[i] < max_samples
 * else all rays valid.  totals[0] = M (kept samples of valid rays), totals[1] = b (valid rays; they
 * are always a prefix).  offsets has B+1 entries (entries past b are clamped to M). */
int nmf_march_scan(const int32_t* counts, int64_t B, int64_t max_samples, int64_t* offsets,
                   uint8_t* whole_valid, int64_t* totals, void* workspace, int64_t workspace_bytes,
                   void* stream);
int64_t nmf_march_scan_workspace_bytes(int64_t B);
/* The same scan; the thread that writes totals also PUBLISHES them: [M, b, publish_seq] into mapped host memory
 * (nmf_host_alloc_mapped's device pointer; NULL = nmf_march_scan) for a host that waits with nmf_wait_seq -- the size
 * read-back of alphagrid.py:353-364 (`ray_valid.sum() > max_samples`, a host sync) without a launch of its own. */
int nmf_march_scan_publish(const int32_t* counts, int64_t B, int64_t max_samples, int64_t* offsets,
                           uint8_t* whole_valid, int64_t* totals, void* workspace, int64_t workspace_bytes,
                           void* publish_mapped_dev, int64_t publish_seq, void* stream);

/* Pass 3: emit the compacted samples of the first b rays: xyzt [M][4] (world xyz, t/focal),
 * ray_id [M], step_id [M], z [M], dist [M] (= z[k+1]-z[k] over ALL candidates, 0 for the last,
 * alphagrid.py:348-350).  Any output pointer may be NULL. */
int nmf_march_fill(const nmf_march_params* p, const float* rays, int64_t b, const float* jitter,
                   const uint64_t* valid_bits, const int64_t* offsets, float* xyzt, int32_t* ray_id,
                   int32_t* step_id, float* z, float* dist, void* stream);

/* Dense views for API compatibility / parity tests: ray_valid [b][N] bool and z_vals [b][N]. */
int nmf_march_dense(const nmf_march_params* p, const float* rays, int64_t b, const float* jitter,
                    const uint64_t* valid_bits, uint8_t* ray_valid, float* z_vals, void* stream);

/* ------------------------------------------------------------------------------------------
 * TensoRF VM field: TensorVMSplit._compute_densityfeature / _compute_appfeature /
 * TensorBase.compute_normals (fields/tensoRF.py:181-205,392-405; fields/tensor_base.py:66-129)
 * with the derivative stencil of GridSampler2D.backward (modules/grid_sample_Cinf.py:109-325).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    float aabb_min[3];
    float inv_size[3];     /* rf.invaabbSize = 2/aabbSize (tensor_base.py:58)                     */
    float density_shift;   /* tensor_base.py:85                                                    */
    int32_t grid;          /* G (cubic)                                                            */
    float stencil[5];      /* centre row of the 5x5 x-stencil; rows above/below via stencil_off   */
    float stencil_off[5];  /* rows i=1 and i=3 of the x-stencil (grid_sample_Cinf.py:218-233)     */
} nmf_vm_params;

/* Derived density tables, rebuilt whenever the density factors change:
 * dpk[i] [G][G][48] = (P | conv_x(P) | conv_y(P)) per texel, dlk[i] [G][32] = (L | conv(L)). */
int nmf_vm_pack_density(const nmf_vm_params* p, const float* const planes[3],
                        const float* const lines[3], float* const dpk[3], float* const dlk[3],
                        void* stream);

/* Forward query of M world-space samples.  Outputs (any may be NULL):
 * sigma_feat [M], sigma [M] (softplus(clamp(f,-15,1e3)+shift)), grad [M][3] (d sigma_feat/d xyz,
 * reference units), normal [M][3] = normalize(-grad), app [M][24], coef [M][72] (plane*line
 * products, saved for the basis_mat gradient).  Density outputs need dpk/dlk, appearance outputs
 * need app_planes/app_lines/basis ([24][72] row-major). */
int nmf_vm_query_fwd(const nmf_vm_params* p, const float* xyzt, int64_t M,
                     const float* const dpk[3], const float* const dlk[3],
                     const float* const app_planes[3], const float* const app_lines[3],
                     const float* basis, float* sigma_feat, float* sigma, float* grad, float* normal,
                     float* app, float* coef, void* stream);
/* bf16-table variant (BASELINE configs[1], train.py:540 `fp16` autocast is the reference's only precision hook): the same
 * query with every factor table stored as bfloat16 (the upper 16 bits of the fp32 pattern) -- half the bytes per tap --
 * and fp32 arithmetic.  The caller keeps fp32 master tables (Adam, backward walk) and refreshes the bf16 copies after each
 * optimizer step (nmf_multi_copy with dst_is_f64 = 2). */
int nmf_vm_query_fwd_bf16(const nmf_vm_params* p, const float* xyzt, int64_t M, const uint16_t* const dpk[3],
                          const uint16_t* const dlk[3], const uint16_t* const app_planes[3],
                          const uint16_t* const app_lines[3], const float* basis, float* sigma_feat, float* sigma,
                          float* grad, float* normal, float* app, float* coef, void* stream);
/* Density VALUE of M samples from the density factors THEMSELVES: planes[i] [G][G][16] (channel-last storage of
 * rf.density_rf.app_plane.i), lines[i] [G][16] -- what fields/tensoRF.py:181-190 + tensor_base.py:85 compute, for the levels
 * of a pass that need no normals (nmf_amd/fast_step.py "sparse normals").  Same taps, same order of the sums as
 * nmf_vm_query_fwd: identical bits; it touches a third of the cache lines the packed value + derivative tables spread the same
 * numbers over.  tables_bf16 != 0: bfloat16 copies of the factors. */
int nmf_vm_query_sigma(const nmf_vm_params* p, const float* xyzt, int64_t M, const void* const planes[3],
                       const void* const lines[3], int32_t tables_bf16, float* sigma_feat, float* sigma, void* stream);
/* The density part of the query (value, sigma, gradient, normal; any output may be NULL) for a FEW rows -- the bounce rows
 * of a re-traced level, where the training pass needs normals (fields/tensor_base.py:66-129 on xyz[bounce_mask]) -- with
 * 16 lanes per row (one per plane tap); the sums are combined in nmf_vm_query_fwd's order: identical bits.
 * dpk / dlk: the packed tables of nmf_vm_pack_density, fp32 or (tables_bf16 != 0) bfloat16. */
int nmf_vm_query_rows(const nmf_vm_params* p, const float* xyzt, int64_t M, const void* const dpk[3],
                      const void* const dlk[3], int32_t tables_bf16, float* sigma_feat, float* sigma, float* grad,
                      float* normal, void* stream);

/* Backward.  Upstream adjoints (any may be NULL): d_sigma [M] (wrt activated sigma), d_sigma_feat [M]
 * (wrt the raw feature, added to the former's contribution), d_normal [M][3], d_app [M][24].
 * sigma_feat / grad are the saved forward outputs.  Accumulates into g_dpk[i] [G][G][48],
 * g_dlk[i] [G][32], g_app_planes[i] [G][G][24], g_app_lines[i] [G][24], g_basis [24][72] (basis_mat, may be NULL)
 * -- caller zeroes them.
 * Samples are counting-sorted by 8^3-voxel brick and reduced in LDS per brick (see vm.hip). */
int nmf_vm_query_bwd(const nmf_vm_params* p, const float* xyzt, int64_t M,
                     const float* const dpk[3], const float* const dlk[3],
                     const float* const app_planes[3], const float* const app_lines[3],
                     const float* basis, const float* sigma_feat, const float* grad,
                     const float* d_sigma, const float* d_sigma_feat, const float* d_normal,
                     const float* d_app, float* const g_dpk[3], float* const g_dlk[3],
                     float* const g_app_planes[3], float* const g_app_lines[3], float* g_basis,
                     void* workspace, int64_t workspace_bytes, void* stream);
/* Scratch the backward needs (brick ids, bin counters, records in brick order); contents are undefined on return.
 * For nmf_vm_query_bwd_segments M is the total over the segments. */
int64_t nmf_vm_bwd_workspace_bytes(int64_t M, int32_t grid);

/* The same walk over the concatenation of up to NMF_VM_MAX_SEGMENTS sample sets (no copy): one training pass queries the
 * field for the primary and for the re-traced rays (tensor_nerf.py:286-393 at recur 0 and 1); their adjoints all land in
 * the same tables, so walking them together bins once and flushes every brick both sets touch once.  All non-empty
 * segments must provide the same set of adjoints (d_sigma / d_normal / d_app NULL-ness). */
#define NMF_VM_MAX_SEGMENTS 4
typedef struct nmf_vm_bwd_segment {
    const float* xyzt;          /* [M][4] */
    int64_t M;
    const float* sigma_feat;    /* [M]    saved forward outputs */
    const float* grad;          /* [M][3] */
    const float* d_sigma;       /* [M]    upstream adjoints, any may be NULL */
    const float* d_sigma_feat;  /* [M] */
    const float* d_normal;      /* [M][3] */
    const float* d_app;         /* [M][24] */
} nmf_vm_bwd_segment;
int nmf_vm_query_bwd_segments(const nmf_vm_params* p, const nmf_vm_bwd_segment* segs /*HOST array*/, int32_t n_segs,
                              const float* const dpk[3], const float* const dlk[3],
                              const float* const app_planes[3], const float* const app_lines[3],
                              const float* basis, float* const g_dpk[3], float* const g_dlk[3],
                              float* const g_app_planes[3], float* const g_app_lines[3], float* g_basis,
                              void* workspace, int64_t workspace_bytes, void* stream);
/* The same call for a caller that walks again and again (a training loop: train.py:497-747 calls backward() every iteration):
 * the counters of the brick sort, the chunk words of its scan and the scratch copies of the basis_mat gradient live in `clean`,
 * nmf_vm_bwd_clean_bytes(grid) bytes of device memory (16-byte aligned) that are ZERO on entry -- zeroed once, when allocated --
 * and zero again when the kernels of the call have run (each clears what it has consumed).  No memset launch per walk (two for
 * an appearance walk: 3.5 us + a launch gap each on the serial tail of a step).  One scratch per walk that may be in flight. */
int64_t nmf_vm_bwd_clean_bytes(int32_t grid);
int nmf_vm_query_bwd_segments_clean(const nmf_vm_params* p, const nmf_vm_bwd_segment* segs /*HOST array*/, int32_t n_segs,
                                    const float* const dpk[3], const float* const dlk[3],
                                    const float* const app_planes[3], const float* const app_lines[3],
                                    const float* basis, float* const g_dpk[3], float* const g_dlk[3],
                                    float* const g_app_planes[3], float* const g_app_lines[3], float* g_basis,
                                    void* clean, int64_t clean_bytes, void* workspace, int64_t workspace_bytes, void* stream);
/* The brick sort of a walk as a PLAN: it depends on the sample positions alone (not on the adjoints), so a training pass
 * builds it when the positions become known -- under its forward, on a side stream -- and the backward only permutes the
 * adjoints (autograd of fields/tensoRF.py:181-205 sees the same sample set twice: forward and backward).
 *   nmf_vm_bin_plan            positions of up to NMF_VM_MAX_SEGMENTS sample sets -> plan (opaque device buffer of
 *                              nmf_vm_bin_plan_bytes(total M, grid) bytes: slots, brick offsets, work items, sorted positions)
 *   nmf_vm_query_bwd_planned   nmf_vm_query_bwd_segments over the SAME segment sizes in the same order with that plan;
 *                              workspace >= nmf_vm_walk_workspace_bytes(total M).
 * nmf_vm_query_bwd_segments = plan + planned walk inside one workspace (nmf_vm_bwd_workspace_bytes = the two sizes added). */
int64_t nmf_vm_bin_plan_bytes(int64_t M, int32_t grid);
int64_t nmf_vm_walk_workspace_bytes(int64_t M);
int nmf_vm_bin_plan(const nmf_vm_params* p, const float* const* xyzt /*HOST array of device pointers*/,
                    const int64_t* Ms /*HOST array*/, int32_t n_segs, void* plan, int64_t plan_bytes, void* stream);
int nmf_vm_query_bwd_planned(const nmf_vm_params* p, const nmf_vm_bwd_segment* segs /*HOST array*/, int32_t n_segs,
                             const float* const dpk[3], const float* const dlk[3],
                             const float* const app_planes[3], const float* const app_lines[3],
                             const float* basis, float* const g_dpk[3], float* const g_dlk[3],
                             float* const g_app_planes[3], float* const g_app_lines[3], float* g_basis,
                             const void* plan, int64_t plan_bytes, void* workspace, int64_t workspace_bytes, void* stream);

/* Folds the packed gradient back onto the factors (transpose of nmf_vm_pack_density):
 * g_planes[i] [G][G][16], g_lines[i] [G][16] are OVERWRITTEN. */
int nmf_vm_unpack_density_grad(const nmf_vm_params* p, const float* const g_dpk[3],
                               const float* const g_dlk[3], float* const g_planes[3],
                               float* const g_lines[3], void* stream);
/* The same with the gradient of the density L1 term added in the same pass: g += l1_scale_dev[0] * sgn(x) / numel(x) for each of
 * the six density parameters x (fields/tensoRF.py:332-340 `density_L1`, weighted by train.py:640-677), given in the storage
 * order of the gradients.  Same bits as the unpack followed by nmf_l1_mean_bwd(accumulate), one launch less at the end of a step. */
int nmf_vm_unpack_density_grad_l1(const nmf_vm_params* p, const float* const g_dpk[3], const float* const g_dlk[3],
                                  float* const g_planes[3], float* const g_lines[3], const float* const x_planes[3],
                                  const float* const x_lines[3], const float* l1_scale_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Compositing: raw2alpha (modules/tensor_nerf.py:19-35) and row_mask_sum
 * (modules/row_mask_sum.py:15-22), segmented by the ray offsets of nmf_march_scan.
 * ---------------------------------------------------------------------------------------- */
int nmf_composite_fwd(const float* sigma, const float* dist, const int64_t* offsets, int64_t b,
                      float distance_scale, float* weight /*[M]*/, float* acc /*[b]*/, void* stream);
int nmf_composite_bwd(const float* sigma, const float* dist, const float* weight,
                      const int64_t* offsets, int64_t b, float distance_scale,
                      const float* d_weight /*[M]*/, float* d_sigma /*[M]*/, void* stream);
/* out[r][d] = sum_{k in segment r} (scale ? scale[k] : 1) * vals[k][d], D = 1..4.  lanes = 1: one lane per segment,
 * added in index order (fp32, reproduces scatter_add_ on the CPU bit for bit); lanes = 8: eight lanes per segment and a
 * shuffle tree (different fp32 rounding; for the adjoint reductions over the secondary rays of a bounce point). */
int nmf_segment_sum(const float* vals, const float* scale, const int64_t* offsets, int64_t n_seg,
                    int32_t D, int32_t lanes, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Environment map: IntegralEquirect (modules/integral_equirect.py:18-173,373-504), activation 'exp'.
 * bg_mat / activated / sat are [3][H][W] fp32.  The summed-area table reproduces torch's CPU
 * rounding (float64 running sums rounded per element, H then W, :433).
 * scalars_dev (optional, DEVICE float[3] = mipbias, brightness, mul) overrides the by-value scalars so that the
 * three learnable 0-d parameters never have to be read back to the host.
 * ---------------------------------------------------------------------------------------- */
/* activated = exp(min(brightness + mul*bg_mat, 20)); sat = cumsum_W(cumsum_H(activated/1000));
 * pole_rows [2][3] (optional) = mean of the first / last row of `activated` per channel (:499-502).
 * sat_i4 [H][W][4] (optional) = the same table with the channels interleaved (4th lane unwritten): the lookups read one
 * 16-byte texel per tap from it instead of three 4-byte values on three planes (layout = 1 below). */
int nmf_sat_build(const float* bg_mat, int32_t H, int32_t W, float brightness, float mul,
                  const float* scalars_dev, float* activated, float* sat, float* pole_rows, float* sat_i4, void* stream);
/* SH projection of prefiltered lookups (modules/integral_equirect.py:324-360, no gradient): coeffs[k][c] =
 * sum_i wq[i][k] * vals[i][c] over n lattice directions (vals [n][3] from nmf_sat_lookup_fwd, wq [n][K] = quadrature
 * weight x SH basis), conv[k][c] = sh_A[k] * coeffs[k][c] / pi (conv / sh_A may be NULL). */
int nmf_sh_project(const float* vals, const float* wq, int64_t n, int32_t K, const float* sh_A, float* coeffs,
                   float* conv, void* stream);
/* d_sat = channel-interleaved adjoint table [H][W][4] (4th lane unused; accumulated by nmf_sat_lookup_bwd;
 * DESTROYED here) -> d_bg [3][H][W] (overwritten).
 * d_pole [2][3]: adjoint of the (top,bottom) pole-row means, may be NULL. */
int nmf_sat_build_bwd(float* d_sat, const float* bg_mat, const float* activated, int32_t H, int32_t W,
                      float brightness, float mul, const float* scalars_dev, const float* d_pole, float* d_bg,
                      void* stream);
/* out[r] = prefiltered radiance along dirs[r] for log-solid-angle sa[r] (IntegralEquirect.forward).
 * pole_rows [2][3] = mean of the first / last row of `activated` (:499-502).
 * dirs_ld = row pitch of dirs in floats: 3 for [R][3] directions, 6 for [R][6] ray rows (origin | direction) whose
 * direction is columns 3..5 -- secondary rays are looked up without slicing them (tensor_nerf.py:302-317).
 * layout: 0 = sat is the planar table [3][H][W], 1 = sat is nmf_sat_build's interleaved sat_i4 [H][W][4]; the results
 * are bit-identical. */
int nmf_sat_lookup_fwd(const float* sat, int32_t H, int32_t W, const float* dirs, int32_t dirs_ld,
                       const float* sa /*[R]*/, int64_t R, float mipbias, const float* scalars_dev,
                       const float* pole_rows, int32_t layout, float* out /*[R][3]*/, void* stream);
/* Adjoints: d_sat [H][W][4] (channel-interleaved so that 8 lanes share one 32-byte atomic run, see csrc/env.hip) and
 * d_pole [2][3] are ACCUMULATED (caller zeroes), d_dirs [R][dirs_ld] is overwritten (with dirs_ld = 6 the origin
 * columns are written as zeros), d_mipbias [1] accumulated.  d_sat / d_dirs / d_mipbias may be NULL. */
int nmf_sat_lookup_bwd(const float* sat, int32_t H, int32_t W, const float* dirs, int32_t dirs_ld, const float* sa,
                       int64_t R, float mipbias, const float* scalars_dev, int32_t layout, const float* d_out /*[R][3]*/,
                       float* d_sat, float* d_pole, float* d_dirs, float* d_mipbias, void* stream);
/* The same adjoints for LARGE lookup counts (autograd of modules/integral_equirect.py:18-173,409-504): the table adjoint is
 * binned instead of scattered with float atomics -- corners counted per 32 x 64-texel SAT tile, written as records into the
 * caller's workspace, accumulated per tile in LDS as 64-bit fixed point (integer LDS atomics run ~8x faster than float
 * atomics on gfx950, and the sums do not depend on the order) and flushed once.  d_sat must be given.  workspace: 16-byte
 * aligned device memory of nmf_sat_lookup_bwd_workspace_bytes(R) bytes (content irrelevant, overwritten): a header, one
 * 32-bit slot per (workgroup of 256 lookups, tile) -- the counting pass hands every workgroup its place in each tile's range,
 * so the record pass walks the footprints once -- and the record pool.  A smaller pool still gives correct results (corners
 * that do not fit take the float atomics); a workspace without room for header + slots + one record is NMF_EINVAL.  Four
 * launches on `stream`. */
int64_t nmf_sat_lookup_bwd_workspace_bytes(int64_t R);
int nmf_sat_lookup_bwd_binned(const float* sat, int32_t H, int32_t W, const float* dirs, int32_t dirs_ld, const float* sa,
                              int64_t R, float mipbias, const float* scalars_dev, int32_t layout,
                              const float* d_out /*[R][3]*/, float* d_sat, float* d_pole, float* d_dirs, float* d_mipbias,
                              void* workspace, int64_t workspace_bytes, void* stream);
/* The two halves of that adjoint as calls of their own, for a caller whose dependency chain needs d_dirs only (the backward of the
 * bounce rays: the sampler's adjoint waits for d_dirs, nothing before the end of the step for d_sat):
 *   nmf_sat_lookup_bwd_dirs                       d_dirs, d_mipbias (both optional), d_pole -- the forward on dual numbers, one launch
 *   nmf_sat_lookup_bwd_binned with d_pole = NULL  d_sat alone (d_dirs, d_mipbias must be NULL too): the three passes without the
 *                                                 dual-number workgroups riding in them, on any other stream
 * Together they add exactly what one full call adds (the table role never touches the pole rows). */
int nmf_sat_lookup_bwd_dirs(const float* sat, int32_t H, int32_t W, const float* dirs, int32_t dirs_ld, const float* sa,
                            int64_t R, float mipbias, const float* scalars_dev, int32_t layout, const float* d_out /*[R][3]*/,
                            float* d_pole, float* d_dirs, float* d_mipbias, void* stream);

/* ------------------------------------------------------------------------------------------
 * Shading helpers.
 * ---------------------------------------------------------------------------------------- */
/* select_bounces (modules/pt_selectors.py:5-60): counts[i] = clamp(floor(pt_i), 0, 400) with
 *   mode 0 (recursion 0): pt = w*mul + u - 0.5                       (mul = rays_per_ray)
 *   mode 1 (recursion>=1): pt = (w + 1e-3*u) / sum_w * mul + add     (sum_w = clip(sum(w + 1e-3 u), 1e-3))
 * sum_w_dev (optional DEVICE scalar) overrides sum_w, so the host need not read the sum back.
 * The dense ray_mask of the reference is the per-row prefix [0, counts[i]). */
int nmf_select_bounces(const float* weights, const float* u, int64_t M, int32_t mode, float mul,
                       float add, float sum_w, const float* sum_w_dev, int32_t* counts, void* stream);
/* Normaliser of the level >= 1 selection above (pt_selectors.py:24-31), on the device in one launch:
 * total = clip(float(sum(weights) + 1e-3 * (sum(u) + extra)), 1e-3), float64 sums.  `extra` = the caller's value for the sum
 * of the uniforms at the culled entries of the dense matrix.  workspace: NMF_SELECT_TOTAL_WS doubles the caller zeroes ONCE
 * (a ticket + per-workgroup partial sums that the last workgroup adds in workgroup order: the result does not depend on the
 * scheduling; the kernel leaves the ticket zeroed); one workspace per stream.  The result is what nmf_select_bounces takes as
 * sum_w_dev. */
#define NMF_SELECT_TOTAL_WS 258
int nmf_select_total(const float* weights, const float* u, int64_t M, double extra, double* workspace, float* total,
                     void* stream);
/* Adjoint of the bounce rows' view vector (V = -ray direction: bV = -viewdirs, models/microfacet.py:354) scattered to the
 * rays: d_rays[ray_id[bidx[row]]][3..5] -= dv_a[row] (+ dv_b[row]); dv_* rows of pitch lda / ldb floats, dv_b nullable. */
int nmf_view_adjoint_to_rays(const int32_t* ray_id, const int32_t* bidx, const float* dv_a, int32_t lda, const float* dv_b,
                             int32_t ldb, int64_t Mb, float* d_rays, void* stream);
/* seg_id[r] / local[r] for r in [offsets[i], offsets[i+1]) = i / r - offsets[i]  (= torch.where(ray_mask)). */
int nmf_expand_segments(const int64_t* offsets, int64_t n_seg, int32_t* seg_id, int32_t* local,
                        void* stream);
/* Secondary rays of the compact ray list (ray i belongs to bounce row row_of_ray[i], is its j_of_ray[i]-th ray):
 * PseudoRandomSampler.draw + GGXSampler.sample/compute_prob + the per-ray glue of Microfacet.forward
 * (brdf_samplers/base.py:11-20, brdf_samplers/ggx.py:61-268, models/microfacet.py:377-456).
 * Rows: V (to viewer), N (flipped to the viewer), r (roughness), x (bounce point), off [Mb][2] (uniform draws, the
 * 0.25 scale is applied inside), cnt (rays per row); sobol [1024][2].  Outputs per ray: L [R][3], half/diff vectors in
 * the local frame [R][3], lpdf [R], mipval [R] = -log(cnt) - lpdf, rays [R][6] = (x + 5e-3 L, L). */
int nmf_ggx_rays_fwd(const float* V_rows, const float* N_rows, const float* r_rows, const float* x_rows,
                     const float* off_rows, const int32_t* cnt_rows, const float* sobol,
                     const int32_t* row_of_ray, const int32_t* j_of_ray, int64_t R, float* L,
                     float* half_local, float* diff_local, float* lpdf, float* mipval, float* rays, void* stream);
/* GGXSampler.compute_prob (brdf_samplers/ggx.py:228-268, isotropic r2 = r1): pdf of a sampled direction from the
 * local-frame incoming / outgoing / half vectors [R][3] and the roughness [R]; 0 below the horizon (dir_in.z <= 0). */
int nmf_ggx_prob(const float* dir_in_local, const float* dir_out_local, const float* half_local, const float* rough,
                 int64_t R, float* prob, void* stream);
/* d_nr [R][4] = (dL/dN)^T g | (dL/dr)^T g per ray (dual-number evaluation of the same sampler); reduce per row.
 * g = dL + d_rays[:,3:6] + 5e-3 d_rays[:,0:3]  (adjoints of L [R][3] and of the bounce rays [R][6]; either may be NULL). */
int nmf_ggx_rays_bwd(const float* V_rows, const float* N_rows, const float* r_rows, const float* off_rows,
                     const float* sobol, const int32_t* row_of_ray, const int32_t* j_of_ray, int64_t R,
                     const float* dL, const float* d_rays, float* d_nr, void* stream);
/* Same with the view direction as a third differentiable input: d_nrv [R][7] = (dN | dr | dV).  The rays of recursion
 * level >= 1 look along the direction the level above sampled (viewdirs = rays[:, 3:6], modules/tensor_nerf.py:262, is part
 * of the graph: bV = -viewdirs, models/microfacet.py:354), so their radiance back-propagates into that direction. */
int nmf_ggx_rays_bwd_view(const float* V_rows, const float* N_rows, const float* r_rows, const float* off_rows,
                          const float* sobol, const int32_t* row_of_ray, const int32_t* j_of_ray, int64_t R,
                          const float* dL, const float* d_rays, float* d_nrv, void* stream);
/* Fresnel-Schlick mix (models/microfacet.py:595-613): contrib [R][3] = (F Li brdf + (1-F) diffuse) / cnt with
 * F = f0 + (1-f0)(1-|V.H|)^5, H = normalize((V+L)/2); sum contrib per row for reflect_rgb. */
int nmf_shade_mix_fwd(const float* V_rows, const float* f0_rows, const float* diffuse_rows,
                      const int32_t* cnt_rows, const int32_t* row_of_ray, int64_t R, const float* L,
                      const float* incoming, const float* brdf, float* contrib, void* stream);
/* d_rows [Mb][3] = adjoint of the row sums.  d_incoming, d_brdf, dL [R][3] overwritten;
 * d_f0diff [R][6] = per-ray (d f0 | d diffuse), reduce per row. */
int nmf_shade_mix_bwd(const float* V_rows, const float* f0_rows, const float* diffuse_rows,
                      const int32_t* cnt_rows, const int32_t* row_of_ray, int64_t R, const float* L,
                      const float* incoming, const float* brdf, const float* d_rows, float* d_incoming,
                      float* d_brdf, float* dL, float* d_f0diff, void* stream);
/* Same, plus dV [R][3] (nullable): the adjoint of the view direction per ray (F depends on V.H, H = normalize((V+L)/2)). */
int nmf_shade_mix_bwd_view(const float* V_rows, const float* f0_rows, const float* diffuse_rows,
                           const int32_t* cnt_rows, const int32_t* row_of_ray, int64_t R, const float* L,
                           const float* incoming, const float* brdf, const float* d_rows, float* d_incoming,
                           float* d_brdf, float* dL, float* d_f0diff, float* dV, void* stream);
/* RandHydraMLPDiffuse heads (modules/render_modules.py:519-574; pospe=-1, feape=0, one Linear each, std=0):
 * out [M][11] = (albedo 3 | tint 3 | f0 3 | roughness 2) with the activations applied; W [11][24] / b [11] are
 * the four Linear layers stacked in that order. */
int nmf_heads_fwd(const float* feat, int64_t M, const float* W, const float* b, float diffuse_mul,
                  float diffuse_bias, float tint_bias, float f0_bias, float rough_bias, float* out, void* stream);
/* d_feat [M][24] overwritten; gW [11][24], gb [11] ACCUMULATED (caller zeroes).  d_feat_add (may be NULL, may be d_feat
 * itself): another adjoint of the same rows, d_feat = d_feat_add + (this adjoint) -- the sum the caller would otherwise
 * spend a launch on. */
int nmf_heads_bwd(const float* feat, int64_t M, const float* W, const float* b, float diffuse_mul,
                  float diffuse_bias, float tint_bias, float f0_bias, float rough_bias, const float* d_out,
                  const float* d_feat_add, float* d_feat, float* gW, float* gb, void* stream);
/* Fused MLPBRDF (modules/brdf.py:177-261): out[r] = sigmoid(MLP(X[r])[0:3] + out_bias) with X as above and
 * MLP = Linear(66,64) ReLU Linear(64,64) ReLU Linear(64,4) (weights row-major [out][in], torch layout).
 * The dense layers run on v_mfma_f32_32x32x16_bf16 with every fp32 operand split into three bf16 terms (six products
 * per K block, fp32 accumulation: fp32-class accuracy, csrc/brdf_mlp.hip).  Writes out [R][3] and, when act_mask is not
 * NULL, the ReLU masks of the two hidden layers for the backward: act_mask [R][4] uint32 = per ray {layer-1 mask,
 * layer-2 mask} of lane half 0, then of lane half 1 (bit 16 ub + q = unit 32 ub + (q & 3) + 8 (q >> 2) + 4 half). */
int nmf_brdf_mlp_fwd(const float* W0, const float* b0, const float* W2, const float* b2, const float* W4,
                     const float* b4, const float* half_vec, const float* diff_vec, const float* feat_src,
                     const float* rough_src, const int32_t* src_idx, int64_t R, float out_bias, float* out,
                     uint32_t* act_mask, int32_t max_workgroups /* 0 = default; see nmf_brdf_mlp_bwd */, void* stream);
/* Backward of the call above for the SAME inputs: fwd_out [R][3] and act_mask [R][4] are that call's outputs (the
 * sigmoid adjoint and the ReLU decisions come from them; the hidden activations are recomputed per 32-ray tile as
 * values, two bf16 terms per operand).  d_feat [rows of feat_src][24] = adjoint of feat_src, ACCUMULATED (caller zeroes):
 * the adjoints of the rays that gathered a row (src_idx non-decreasing: consecutive rays) are summed inside the kernel,
 * one atomic per (row, column) run of a 32-ray tile.  gW* / gb* are ACCUMULATED (caller zeroes): gW0 [64][66], gb0 [64],
 * gW2 [64][64], gb2 [64], gW4 [4][64], gb4 [4] (row 3 of gW4 / gb4 stays untouched: the fourth output is unused).
 * max_workgroups: 0 = the kernel's own choice (one persistent workgroup per CU, fewer for short launches); > 0 caps
 * them, for callers that run this launch on a second stream NEXT TO other kernels and want it to leave CUs free for
 * them (the training pass: nmf_amd/fast_step.py, csrc/step_core.inc).
 * workspace (device, nmf_brdf_mlp_bwd_workspace_bytes(R, max_workgroups) bytes, need not be initialised): one partial
 * sum of the weight gradients per workgroup; a second small launch adds them in workgroup order, so a gradient element
 * receives one atomic per call instead of one per workgroup. */
int nmf_brdf_mlp_bwd(const float* W0, const float* b0, const float* W2, const float* b2, const float* W4,
                     const float* b4, const float* half_vec, const float* diff_vec, const float* feat_src,
                     const float* rough_src, const int32_t* src_idx, int64_t R, const float* fwd_out,
                     const uint32_t* act_mask, const float* d_out, float* d_feat, float* gW0, float* gb0,
                     float* gW2, float* gb2, float* gW4, float* gb4, int32_t max_workgroups, void* workspace,
                     int64_t workspace_bytes, void* stream);
int64_t nmf_brdf_mlp_bwd_workspace_bytes(int64_t R, int32_t max_workgroups);
/* The backward over ONE OR TWO ray sets in one launch (R4): the BRDF evaluations of a recursion level and of the level below it
 * (models/microfacet.py:377-456 at recur 0 and 1) share the weights, and a launch costs ~25 us before and after its tiles whatever
 * its size.  image: nmf_brdf_mlp_pack's output or NULL (then W0..b4 are read); per set the arguments of nmf_brdf_mlp_bwd; sets
 * with R = 0 are skipped.  Gradients as in nmf_brdf_mlp_bwd (ACCUMULATED).  Workspace: nmf_brdf_mlp_bwd_segments_workspace_bytes. */
typedef struct nmf_mlp_bwd_segment {
    const float* half_vec;      /* [R][3] */
    const float* diff_vec;      /* [R][3] */
    const float* feat_src;      /* [rows][24] */
    const float* rough_src;     /* [rows] */
    const int32_t* src_idx;     /* [R] row of every ray, non-decreasing (NULL: ray r reads row r) */
    int64_t R;
    const float* fwd_out;       /* [R][3]  the forward's output */
    const uint32_t* act_mask;   /* [R][4]  the forward's ReLU masks */
    const float* d_out;         /* [R][3] */
    float* d_feat;              /* [rows][24], ACCUMULATED */
} nmf_mlp_bwd_segment;
int nmf_brdf_mlp_bwd_segments(const void* image, const float* W0, const float* b0, const float* W2, const float* b2,
                              const float* W4, const float* b4, const nmf_mlp_bwd_segment* segs /*HOST array*/, int32_t n_segs,
                              float* gW0, float* gb0, float* gW2, float* gb2, float* gW4, float* gb4, int32_t max_workgroups,
                              void* workspace, int64_t workspace_bytes, void* stream);
int64_t nmf_brdf_mlp_bwd_segments_workspace_bytes(const int64_t* Rs /*HOST*/, int32_t n_segs, int32_t max_workgroups);
/* The same two calls with the weights as a PACKED IMAGE (R4): what the kernels stage into LDS in front of their first tile --
 * the split-bf16 planes of W0 | b0, W2 (forward: three terms; backward: two terms plus W2^T and W0[:, :24]^T) and the fp32 rows
 * of the last layer -- written once per weight update by nmf_brdf_mlp_pack (one launch of two workgroups) into
 * nmf_brdf_mlp_image_bytes() bytes of device memory (16-byte aligned).  A launch then starts with a byte copy instead of the
 * conversion (the parameters are those of modules/brdf.py:177-261's `self.mlp`, which change only in the optimizer step:
 * train.py:733-735).  Results are bit-identical to the unpacked calls. */
int64_t nmf_brdf_mlp_image_bytes(void);
int nmf_brdf_mlp_pack(const float* W0, const float* b0, const float* W2, const float* b2, const float* W4, const float* b4,
                      void* image, int64_t image_bytes, void* stream);
int nmf_brdf_mlp_fwd_packed(const void* image, const float* half_vec, const float* diff_vec, const float* feat_src,
                            const float* rough_src, const int32_t* src_idx, int64_t R, float out_bias, float* out,
                            uint32_t* act_mask, int32_t max_workgroups, void* stream);
int nmf_brdf_mlp_bwd_packed(const void* image, const float* half_vec, const float* diff_vec, const float* feat_src,
                            const float* rough_src, const int32_t* src_idx, int64_t R, const float* fwd_out,
                            const uint32_t* act_mask, const float* d_out, float* d_feat, float* gW0, float* gb0,
                            float* gW2, float* gb2, float* gW4, float* gb4, int32_t max_workgroups, void* workspace,
                            int64_t workspace_bytes, void* stream);
/* out[s][0:D] = sum_{r in segment s} vals[r*row_stride + 0:D], D <= 64 (adjoint of the feature gather). */
int nmf_segment_sum_wide(const float* vals, int64_t row_stride, int32_t D, const int64_t* offsets,
                         int64_t n_seg, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Shading glue (csrc/shade.hip): the per-sample / per-ray spans between the big operators.
 * ---------------------------------------------------------------------------------------- */
/* models/microfacet.py:333-350.  counts [M] = secondary rays per sample (nmf_select_bounces).  Outputs:
 * bidx [>=Mb] samples with counts > 0 in order ("rows"), row_off [>=Mb+1] exclusive scan of their counts
 * (row_off[Mb] = R), cnt_rows [>=Mb] their counts, inv [M] row of each sample or -1, totals = {R, Mb}.
 * bidx / row_off / cnt_rows are sized by the caller for the worst case (M, M+1, M).
 * xyzt_rows (optional, [>=Mb][4], needs xyzt [M][4]): xyzt_rows[row] = xyzt[bidx[row]], the sample positions of the rows
 * (microfacet.py:352: `xyzs[bounce_mask]`), written by the same pass instead of a gather launch behind the size read-back. */
int nmf_bounce_index(const int32_t* counts, int64_t M, int32_t* bidx, int64_t* row_off, int32_t* cnt_rows,
                     int32_t* inv, int64_t* totals, const float* xyzt, float* xyzt_rows, void* workspace,
                     int64_t workspace_bytes, void* stream);
int64_t nmf_bounce_index_workspace_bytes(int64_t M);
/* The same queries / index with the live element count still ON THE DEVICE (M_live: e.g. totals[0] of nmf_march_scan): the launch
 * is sized by the bound M_cap the caller allocated for (sampler.max_samples, samplers/alphagrid.py:353-364), elements at or beyond
 * *M_live are neither read nor written.  Lets a caller queue the level-0 pipeline behind the scan without waiting for its sizes. */
int nmf_vm_query_fwd_live(const nmf_vm_params* p, const float* xyzt, int64_t M_cap, const int64_t* M_live,
                          const void* const dpk[3], const void* const dlk[3], const void* const app_planes[3],
                          const void* const app_lines[3], int32_t tables_bf16, const float* basis, float* sigma_feat,
                          float* sigma, float* grad, float* normal, float* app, float* coef, void* stream);
int nmf_bounce_index_live(const int32_t* counts, int64_t M_cap, const int64_t* M_live, int32_t* bidx, int64_t* row_off,
                          int32_t* cnt_rows, int32_t* inv, int64_t* totals, const float* xyzt, float* xyzt_rows, void* workspace,
                          int64_t workspace_bytes, void* publish_mapped_dev, int64_t publish_seq, void* stream);
/* nmf_select_bounces + nmf_bounce_index_live in the launches of the latter (R4): the bounce count of a sample is one expression
 * of (weights[i], u[i]) (modules/pt_selectors.py:5-60; arguments as nmf_select_bounces), evaluated inside the index kernels
 * instead of by a launch of its own on the forward's chain.  counts are not materialised.  M_cap <= 2^20; M_live may be NULL. */
int nmf_bounce_index_select(const float* weights, const float* u, int32_t mode, float mul, float add, float sum_w,
                            const float* sum_w_dev, int64_t M_cap, const int64_t* M_live, int32_t* bidx, int64_t* row_off,
                            int32_t* cnt_rows, int32_t* inv, int64_t* totals, const float* xyzt, float* xyzt_rows,
                            void* workspace, int64_t workspace_bytes, void* publish_mapped_dev, int64_t publish_seq,
                            void* stream);
/* nmf_bounce_index that also publishes [R, Mb, publish_seq] into mapped host memory (see nmf_march_scan_publish). */
int nmf_bounce_index_publish(const int32_t* counts, int64_t M, int32_t* bidx, int64_t* row_off, int32_t* cnt_rows,
                             int32_t* inv, int64_t* totals, const float* xyzt, float* xyzt_rows, void* workspace,
                             int64_t workspace_bytes, void* publish_mapped_dev, int64_t publish_seq, void* stream);
/* Per bounce row (models/microfacet.py:297,304-316,352-361): V = -ray direction, N = normal facing V
 * (n * sign(V.n)), r1 = max(roughness, min_rough), f0, diffuse = albedo * E(n) with E the 9-term SH irradiance
 * (conv [9][3] DEVICE pointer, modules/sh.py:97-142), feat = app + anoise * feat_noise (feat_noise may be NULL),
 * xyz.  heads is nmf_heads_fwd's output, rays [b][6], ray_id [M].  app / heads / feat_noise are indexed by SAMPLE
 * ([M][24], [M][11], [M][24]) or, with row_inputs != 0, by BOUNCE ROW ([Mb][.]: appearance evaluated only where it is
 * used -- in training that is ~5-20 % of the samples).  row_inputs == 2: `normals` is indexed by bounce row as well
 * ([Mb][3]: below the first recursion level normals are only needed where secondary rays start). */
int nmf_bounce_prep_fwd(const int32_t* bidx, int64_t Mb, const float* normals, const float* app, const float* heads,
                        const float* xyzt, const int32_t* ray_id, const float* rays, const float* conv,
                        const float* feat_noise, float anoise, float min_rough, int32_t row_inputs, float* V, float* N,
                        float* r1, float* f0, float* diffuse, float* feat, float* xyz, void* stream);
/* The same with the material heads evaluated inside the launch (R4): head_W [11][24], head_b [11] and the five activation
 * parameters of nmf_heads_fwd instead of its output; heads_out [Mb][11] receives what nmf_heads_fwd(app) would have written (the
 * backward reads it).  app must be given per bounce row (row_inputs 1 or 2).  One launch less per recursion level on the
 * forward's chain; the same bits as nmf_heads_fwd followed by nmf_bounce_prep_fwd. */
int nmf_bounce_prep_fwd_heads(const int32_t* bidx, int64_t Mb, const float* normals, const float* app, const float* head_W,
                              const float* head_b, float diffuse_mul, float diffuse_bias, float tint_bias, float f0_bias,
                              float rough_bias, const float* xyzt, const int32_t* ray_id, const float* rays, const float* conv,
                              const float* feat_noise, float anoise, float min_rough, int32_t row_inputs, float* heads_out,
                              float* V, float* N, float* r1, float* f0, float* diffuse, float* feat, float* xyz, void* stream);
/* Adjoint: d_normals [M][3] written for ALL samples (zeros where inv < 0 or detach_normals); d_heads / d_app written for
 * all M samples ([M][11], [M][24], zeros where inv < 0) or, with row_inputs, per bounce row ([Mb][11], [Mb][24]; bidx
 * required); row_inputs == 2: normals AND d_normals are per bounce row ([Mb][3]), inv is not read.  Row gradients may be
 * NULL (= zero).  row_strides = row pitch in floats of (dN, dr1, df0, ddiffuse), so
 * column slices of wider row tensors are read in place (NULL = dense 3,1,3,3). */
int nmf_bounce_prep_bwd(const int32_t* inv, int64_t M, const int32_t* bidx, int64_t Mb, const float* normals,
                        const float* heads, const int32_t* ray_id, const float* rays, const float* conv,
                        float min_rough, int32_t detach_normals, int32_t row_inputs, const float* dN,
                        const float* dr1, const float* df0,
                        const float* ddiffuse, const int32_t row_strides[4], const float* dfeat, float* d_normals,
                        float* d_heads, float* d_app, void* stream);
/* nmf_bounce_prep_bwd (everything per bounce row: row_inputs = 2) + nmf_heads_bwd in ONE launch (R4): the adjoint of the heads'
 * outputs and the feature-row adjoint that nmf_heads_bwd adds are computed per row from the inputs of nmf_bounce_prep_bwd instead
 * of written and read back.  d_normals [Mb][3], d_app [Mb][24] (the adjoint of `app`: heads' backward + dfeat); g_head_W / g_head_b
 * ACCUMULATED as in nmf_heads_bwd.  One launch less per recursion level on the backward's main chain. */
int nmf_bounce_prep_heads_bwd(const int32_t* bidx, int64_t Mb, const float* normals, const float* heads, const int32_t* ray_id,
                              const float* rays, const float* conv, float min_rough, int32_t detach_normals, const float* dN,
                              const float* dr1, const float* df0, const float* ddiffuse, const int32_t row_strides[4],
                              const float* dfeat, const float* app, const float* head_W, const float* head_b, float diffuse_mul,
                              float diffuse_bias, float tint_bias, float f0_bias, float rough_bias, float* d_normals, float* d_app,
                              float* g_head_W, float* g_head_b, void* stream);
/* modules/tensor_nerf.py:448-452,583-587,658-659 + modules/tonemap.py:34-55, one thread per ray in sample order:
 * acc = sum w, rgb_lin = sum w * refl_rows[inv], ori = sum w * min(-d.n, 0)^2 (ori / normals may be NULL),
 * rgb_map = (tonemap ? srgb(rgb_lin) : rgb_lin) + (1 - acc) * bg;  bg [3] or [B][3] (bg_per_ray). */
int nmf_ray_compose_fwd(const float* weight, const float* refl_rows, const int32_t* inv, const float* normals,
                        const float* rays, const int64_t* offsets, int64_t B, const float* bg, int32_t bg_per_ray,
                        int32_t tonemap, int32_t noclip, float* rgb_map, float* acc, float* rgb_lin, float* ori,
                        void* stream);
/* Adjoint, one thread per sample: d_weight [M], d_refl [Mb][3] (every row written), d_normals [M][3] (may be NULL).
 * d_rgb_map [B][3], d_acc [B], d_ori [B] may each be NULL.  The background gradient (1 - acc) * d_rgb_map is
 * left to the caller. */
int nmf_ray_compose_bwd(const float* weight, const float* refl_rows, const int32_t* inv, const float* normals,
                        const float* rays, const int32_t* ray_id, int64_t M, const float* bg, int32_t bg_per_ray,
                        int32_t tonemap, int32_t noclip, const float* rgb_lin, const float* d_rgb_map,
                        const float* d_acc, const float* d_ori, float* d_weight, float* d_refl, float* d_normals,
                        void* stream);
/* ... which is this: d_bg [B][3] = (1 - acc[B]) * d_rgb_map [B][3] (tensor_nerf.py:658-659 backward for a per-ray
 * background: the adjoint handed to nmf_sat_lookup_bwd). */
int nmf_bg_adjoint(const float* acc, const float* d_rgb_map, int64_t B, float* d_bg, void* stream);

/* Retrace selection (models/microfacet.py:475-537, csrc/retrace.hip).
 * score[r] = max_c(brdf[r][c]) * [V.N > 0 of the ray's row] * exp(lpdf[r]) * w_rows[row] / (cnt_rows[row] + 1e-8). */
int nmf_retrace_scores(const float* brdf /*[R][3]*/, const float* V_rows, const float* N_rows /*[Mb][3]*/,
                       const float* lpdf /*[R]*/, const float* w_rows /*[Mb]*/, const int32_t* cnt_rows,
                       const int32_t* row_of_ray, int64_t R, float* score, void* stream);
/* order = argsort(keys) ascending (= torch.argsort of microfacet.py:522; ties in unspecified order), radix sort of
 * (key, index) pairs on the caller's workspace. */
int nmf_argsort_f32(const float* keys, int64_t n, int32_t* order, void* workspace, int64_t workspace_bytes,
                    void* stream);
/* The partition the reference takes from that argsort (models/microfacet.py:506-537: retrace_ray_inds = cc_as[M:], notrace_ray_inds =
 * cc_as[:M]) WITHOUT sorting all n keys: a radix select (three histogram passes of 11 / 11 / 10 bits over the keys find the key of ascending rank
 * n - k; ties at the cut go the way a stable ascending sort places them).  idx_top [k] = argsort(keys)[n-k:] exactly, in that order
 * (the re-traced rays are processed in it); idx_rest [n-k] = the indices of argsort(keys)[:n-k] in INDEX order (their order is never
 * used).  Either output may be NULL when empty. */
int64_t nmf_topk_select_workspace_bytes(int64_t n);
int nmf_topk_select(const float* keys, int64_t n, int64_t k, int32_t* idx_top, int32_t* idx_rest, void* workspace,
                    int64_t workspace_bytes, void* stream);
int64_t nmf_argsort_workspace_bytes(int64_t n);

/* ------------------------------------------------------------------------------------------
 * Loss terms (csrc/loss.hip).  `out` scalars are ACCUMULATED into (caller zeroes).
 * ---------------------------------------------------------------------------------------- */
/* fields/tensoRF.py:332-340 density_L1: out += sum_i mean(|x_i|) over up to 8 dense fp32 tensors (x, numel: HOST arrays
 * of `count` entries); backward g_i = d_out * sgn(x_i) / numel_i, written in x_i's own memory order -- or, with
 * accumulate != 0, ADDED to g_i (which then already holds the rendering gradient of the same factor: one launch instead
 * of one accumulation kernel per tensor). */
int nmf_l1_mean_fwd(const float* const x[], const int64_t numel[], int32_t count, float* out, void* stream);
int nmf_l1_mean_bwd(const float* const x[], const int64_t numel[], int32_t count, const float* d_out,
                    float* const g[], int32_t accumulate, void* stream);
/* train.py:598-601 photometric term: out += sum (clip(pred,0,1) - clip(gt,0,1))^2 over n floats; backward
 * d_pred = 2 (pred - clip(gt)) d_out inside [0,1], 0 outside (d_out: device scalar). */
int nmf_sqerr_fwd(const float* pred, const float* gt, int64_t n, float* out, void* stream);
int nmf_sqerr_bwd(const float* pred, const float* gt, int64_t n, const float* d_out, float* d_pred, void* stream);
/* train.py:640-677 loss assembly: out += scale * sum_i w_i * sum(x_i) over up to 8 dense fp32 tensors (scalars or
 * per-ray vectors such as the orientation terms of tensor_nerf.py:583-587 and acc_map of :598-602); backward fills
 * g_i[:] = d_out * scale * w_i (d_out: device scalar).  x, numel, w, g: HOST arrays of `count` entries. */
int nmf_loss_mix_fwd(const float* const x[], const int64_t numel[], const float w[], int32_t count, float scale,
                     float* out, void* stream);
int nmf_loss_mix_bwd(const int64_t numel[], const float w[], int32_t count, float scale, const float* d_out,
                     float* const g[], void* stream);
/* The head of a training chunk's backward in ONE launch (train.py:598-601 + 640-677; the same values as nmf_sqerr_fwd +
 * nmf_loss_mix_bwd + nmf_sqerr_bwd, which stay for callers that need the pieces): loss[0] = sum (clip(pred,0,1) -
 * clip(gt,0,1))^2 over [n_rays][3], WRITTEN (per-workgroup sums added in workgroup order by the workgroup that finishes
 * last: no zero fill, no float atomics, run-to-run identical); d_pred = 2 (pred - clip(gt)) * (d_out scale w_pred) inside
 * [0,1], 0 outside; g_a / g_b [n_rays] (each may be NULL) filled with d_out scale w_a / w_b (the constant adjoints of the
 * per-ray acc / orientation terms).  d_out: device scalar.  workspace: nmf_loss_head_workspace_bytes(n_rays) bytes, 16-byte
 * aligned, whose first 4 bytes are ZERO before the first use (a ticket counter the launch leaves at zero again); one
 * workspace serves one stream at a time. */
int64_t nmf_loss_head_workspace_bytes(int64_t n_rays);
int nmf_loss_head(const float* pred, const float* gt, int64_t n_rays, const float* d_out, float scale, float w_pred,
                  float w_a, float w_b, float* loss, float* d_pred, float* g_a, float* g_b, void* workspace,
                  int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer: torch.optim.Adam over the per-module param groups (train.py:443-469), every tensor in one launch.
 * `slots` is a HOST array (copied into kernel arguments); hyper-parameters per slot are the group's, with the
 * bias corrections computed by the host in double precision exactly as torch/optim/adam.py does:
 *   step_size = lr / (1 - beta1^t),  bc2_sqrt = sqrt(1 - beta2^t).
 * param/grad/exp_avg/exp_avg_sq share one dense layout of `numel` elements (fp32, or fp64 when is_f64).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    void* param;
    const void* grad;
    void* exp_avg;
    void* exp_avg_sq;
    int64_t numel;
    double beta1, beta2, eps, weight_decay, step_size, bc2_sqrt;
    int32_t is_f64;
    int32_t reserved;
} nmf_adam_slot;
int nmf_adam_step(const nmf_adam_slot* slots, int32_t n_slots, void* stream);
/* The same update gated by a device float (e.g. the summed loss of the step): not finite -> no parameter and no moment changes.
 * train.py:704-705 skips a chunk whose loss is NaN after a host read-back; this keeps the decision on the device. */
int nmf_adam_step_guarded(const nmf_adam_slot* slots, int32_t n_slots, const float* guard, void* stream);

/* Multi-tensor copy with fp32 <-> fp64 conversion, all slots in one launch (slots: HOST array, passed by value): packs the
 * per-parameter gradients into the flat fp32 all-reduce buffer and unpacks the reduced sums (ONE collective
 * per optimizer step).  src / dst are dense runs of `numel` elements. */
typedef struct {
    const void* src;
    void* dst;
    int64_t numel;
    int32_t src_is_f64;      /* 0: fp32 source, 1: fp64 source */
    int32_t dst_is_f64;      /* 0: fp32, 1: fp64, 2: bfloat16 (fp32 source only, round to nearest even) */
} nmf_copy_slot;
int nmf_multi_copy(const nmf_copy_slot* slots, int32_t n_slots, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NMF_HIP_H */
