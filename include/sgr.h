/*
 * sgr.h -- C ABI of the MI355X-native differentiable Gaussian rasterizer (libsgr_hip.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types.  Every entry point mirrors a
 * native interface of the reference (zju3dv/street_gaussians); paths below are relative to
 * /root/reference/submodules/.  All array arguments are DEVICE pointers (HIP) unless stated otherwise; a
 * NULL pointer means "feature absent", exactly like an empty tensor does in the reference
 * (diff-gaussian-rasterization/cuda_rasterizer/rasterizer_impl.cu:324,452,482).  `stream` is a
 * hipStream_t (NULL = default stream); unlike the reference (legacy default stream everywhere) all
 * work is enqueued on it.
 *
 * Return convention: >= 0 success, < 0 = -SGR_E_*; sgr_last_error() gives the message (thread local).
 * INTEGRATION.md shows the binding a maintainer of the reference adds on top of this header.
 */
#ifndef SGR_H_INCLUDED
#define SGR_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGR_E_INVALID 1   /* bad argument (shape / size / unsupported channel count)          */
#define SGR_E_HIP 2       /* a HIP runtime call or kernel failed (message has hipGetErrorString) */
#define SGR_E_ALLOC 3     /* a scratch callback returned NULL                                   */
#define SGR_E_PREFILTER 4 /* prefiltered=1 but a Gaussian failed the frustum test
                             (reference: device printf + __trap(), cuda_rasterizer/auxiliary.h:156-161) */
#define SGR_E_LAZY 5      /* lazy mode (sgr_set_lazy): the PREVIOUS forward of this thread was invalid -- discard its outputs */

/* Growable device scratch supplied by the caller: must return a device pointer to at least `nbytes`
 * bytes that stays valid until the matching backward has run.  Replaces std::function<char*(size_t)>
 * of CudaRasterizer::Rasterizer::forward (diff-gaussian-rasterization/cuda_rasterizer/rasterizer.h:32-34;
 * the torch glue builds them in rasterize_points.cu:27-33). */
typedef char* (*sgr_alloc_fn)(size_t nbytes, void* user);

const char* sgr_last_error(void);
int sgr_version(void);

/* CudaRasterizer::Rasterizer::forward  (cuda_rasterizer/rasterizer.h:31-62, rasterizer_impl.cu:197-343).
 * P Gaussians, D = active SH degree, M = SH coefficients per Gaussian (row stride of shs), S = semantic
 * channels (<= 32).  Outputs: out_color[3,H,W], out_depth[1,H,W], out_alpha[1,H,W] (= sum alpha_i*T_i),
 * out_semantic[S,H,W], radii[P] (may be NULL).  Every output element is written (P == 0: zeros, as the
 * reference's torch::full(0) would leave them).  Returns num_rendered (R) = the number of (tile, Gaussian) instances this
 * call emitted; it sizes the binning buffer and is what sgr_backward must be given.  By default a Gaussian is emitted
 * only for the tiles in which it can reach alpha >= 1/255 (a subset of the reference's getRect square: same images, see
 * sgr_test_switches bits 10 / 11), so R is smaller than the reference's count for the same frame. */
int sgr_forward(sgr_alloc_fn geometry_buffer, void* geometry_user, sgr_alloc_fn binning_buffer, void* binning_user,
                sgr_alloc_fn image_buffer, void* image_user, int P, int D, int M, int S, const float* background,
                int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
                const float* semantics, const float* opacities, const float* scales, float scale_modifier,
                const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                float* out_depth, float* out_alpha, float* out_semantic, int* radii, int debug, void* stream);

/* CudaRasterizer::Rasterizer::backward  (cuda_rasterizer/rasterizer.h:64-102, rasterizer_impl.cu:396-506).
 * geom/binning/image buffers and R are the ones forward produced.  `scratch` provides temporary device
 * memory (one float4 per Gaussian + R partial rows of sgr_partial_row_floats(S) floats) that may be released when
 * the call returns.
 * Binning-buffer contract: the backward WRITES into the forward's binning buffer -- one "row written" byte per sorted
 * instance, cleared by the forward's tile-ranges launch.  The set of rows a backward writes is a pure function of the
 * forward's hit record and n_contrib (never of the upstream gradients), so any number of backward calls over ONE forward
 * re-mark the same bytes (tests: two backward passes with retain_graph, incl. all-zero upstream gradients).  Do not hand
 * the backward a binning buffer copied from, or shared with, another forward, even one with the same R: rows would be
 * summed that this backward never wrote.
 * All gradient outputs are
 * fully written (no zero-initialisation needed, cf. rasterize_points.cu:166-176):
 * dL_dmean2D[P,3] (z = sum |gx|+|gy|), dL_dopacity[P], dL_dcolor[P,3], dL_dmean3D[P,3], dL_dcov3D[P,6],
 * dL_dsh[P,M,3] (NULL if shs NULL), dL_dscale[P,3], dL_drot[P,4], dL_dsemantic[P,S].
 * (The reference's dL_dconic / dL_ddepth intermediates are internal here.) */
int sgr_backward(int P, int D, int M, int R, int S, const float* background, int width, int height,
                 const float* means3D, const float* shs, const float* colors_precomp, const float* semantics,
                 const float* alphas, const float* scales, float scale_modifier, const float* rotations,
                 const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
                 float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer,
                 char* image_buffer, const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dalphas,
                 const float* dL_dpix_semantic, float* dL_dmean2D, float* dL_dopacity, float* dL_dcolor,
                 float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                 float* dL_dsemantic, sgr_alloc_fn scratch, void* scratch_user, int debug, void* stream);

/* sgr_backward with optional extras (NULL = plain sgr_backward).
 * Densification statistics sink: the per-Gaussian epilogue of the backward holds dL/dmean2D and the radius in
 * registers, so it can apply this view's set_max_radii2D + add_densification_stats
 * (lib/models/street_gaussian_model.py:551-571) in place -- for every Gaussian with radii > 0:
 *   xyz_gradient_accum[g,0] += |dL/dmean2D[g,:2]|,  xyz_gradient_accum[g,1] += |dL/dmean2D[g,2]|,  denom[g] += 1,
 *   max_radii2D[g] = max(max_radii2D[g], radii[g])
 * -- instead of six masked torch ops per sub-model afterwards.  With segments == NULL the three arrays cover all P
 * Gaussians of the call in its order.  The reference rebuilds the list of rendered sub-models per frame
 * (street_gaussian_model.py:230-250: only the actors visible at that timestamp), so the rasterized set is in general a
 * SUBSET of the persistent per-model statistics: `segments` (host array, sorted by src_start, non-overlapping, at most
 * SGR_MAX_STAT_SEGMENTS) maps the call's Gaussians [src_start, src_start + count) to the persistent rows
 * [dst_offset, dst_offset + count); Gaussians in no segment leave no statistics.  The three arrays then have the
 * caller's persistent length (street_gaussians_amd.scene.FlatStats keeps the sub-models' statistics as views of them). */
#define SGR_MAX_STAT_SEGMENTS 128
typedef struct sgr_stat_segment {
    int src_start;  /* first Gaussian of the segment in this call's order */
    int count;
    int dst_offset; /* its first row in the persistent arrays */
} sgr_stat_segment;
typedef struct sgr_backward_extras {
    float* xyz_gradient_accum; /* [P,2] (or the persistent [total,2] with segments), or NULL */
    float* denom;              /* [P,1] */
    float* max_radii2D;        /* [P]   */
    const sgr_stat_segment* segments; /* host memory; NULL = identity map over the call's P Gaussians */
    int n_segments;
    void* color_ready_event; /* optional hipEvent_t, recorded on `stream` right after the row-sum stage: dL_dmean2D,
                              * dL_dopacity and dL_dcolor are FINAL from that event on, while the per-Gaussian stage (K12 + K13)
                              * is still to run -- a view-sharded trainer starts the all-gather of its dRGB behind it
                              * (street_gaussians_amd.multiview).  NULL: not recorded. */
    int rows; /* length of the three persistent arrays (rows); with segments every [dst_offset, dst_offset + count) must lie
               * inside [0, rows) and the destination ranges must be pairwise disjoint (two segments on the same rows would be
               * a racy read-modify-write) -- checked on the host, SGR_E_INVALID otherwise.  0 = unknown: not checked. */
    float* masked_color_out; /* optional [P,3]: dL_dcolor with the channels the forward clamped at zero set to 0
                              * (backward.cu:40-44) -- what sgr_masked_color_grad computes, written by the row-sum stage while
                              * dL_dcolor is in registers (final at color_ready_event): the view-sharded exchange hands a slot
                              * of its all-gather payload here and saves a launch and a 12 B/Gaussian copy.  NULL: not written. */
    int skip_sh_grad; /* != 0: dL_dsh is NOT written (the pointer may be NULL although shs is given) -- the factored exchange
                       * rebuilds the SH gradient of ALL views from the gathered dRGB (sgr_sh_grad_from_views), so this view's
                       * own 12*M B/Gaussian would be written only to be overwritten.  Every other output is unchanged. */
} sgr_backward_extras; /* layout of sgr_version() >= 102 (101 ended at rows, 100 at n_segments: a caller built against an
                        * older header must not be run against this library -- check sgr_version() before passing the struct) */
int sgr_backward_ex(int P, int D, int M, int R, int S, const float* background, int width, int height,
                    const float* means3D, const float* shs, const float* colors_precomp, const float* semantics,
                    const float* alphas, const float* scales, float scale_modifier, const float* rotations,
                    const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
                    float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer,
                    char* image_buffer, const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dalphas,
                    const float* dL_dpix_semantic, float* dL_dmean2D, float* dL_dopacity, float* dL_dcolor,
                    float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                    float* dL_dsemantic, sgr_alloc_fn scratch, void* scratch_user, int debug, void* stream,
                    const sgr_backward_extras* extras);

/* ---- view-sharded multi-GPU training (BASELINE.json north_star; no counterpart in the single-GPU reference) -------
 * The SH gradient of one view is rank-1: dL/dSH[k][c] = Y_k(dir) * dRGB[c] (cuda_rasterizer/backward.cu:46-105), so
 * ranks exchange the 3 floats of dRGB per Gaussian instead of the 3*M floats of dL/dSH and rebuild the sum locally.
 *
 * sgr_masked_color_grad: dL_drgb[P,3] = dL_dcolor[P,3] (as returned by sgr_backward) with the channels the forward
 * clamped at zero (forward.cu:64-70) set to 0 (backward.cu:40-44).  geom_buffer = the one sgr_forward filled. */
int sgr_masked_color_grad(int P, const char* geom_buffer, const float* dL_dcolor, float* dL_drgb, void* stream);
/* sgr_sh_grad_from_views: dL_dsh[P,M,3] = sum over v < V of Y(normalize(means3D - campos[v])) (x) dL_drgb[v]
 * (campos [V,3], dL_drgb [V,P,3]); coefficients k >= (D+1)^2 are written as zeros, like sgr_backward does. */
int sgr_sh_grad_from_views(int P, int D, int M, int V, const float* means3D, const float* campos, const float* dL_drgb,
                           float* dL_dsh, void* stream);
/* The same with explicit per-view strides (in floats): view v reads its camera centre at campos + v*campos_view_stride,
 * its dL_drgb rows at dL_drgb + v*drgb_view_stride, and -- means_view_stride > 0 -- its own positions at
 * means3D + v*means_view_stride (a posed sub-model, street_gaussian_model.py:287-330: actor positions differ per frame and
 * travel with the exchange); means_view_stride == 0: one set of positions for all views.  Lets the rebuild read straight
 * out of all-gathered payload rows. */
int sgr_sh_grad_from_views_ex(int P, int D, int M, int V, const float* means3D, size_t means_view_stride,
                              const float* campos, size_t campos_view_stride, const float* dL_drgb,
                              size_t drgb_view_stride, float* dL_dsh, void* stream);

/* CudaRasterizer::Rasterizer::markVisible  (cuda_rasterizer/rasterizer.h:24-29, rasterizer_impl.cu:141-153).
 * present[P] as bytes (0/1). */
int sgr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     void* stream);

/* CudaRasterizer::Rasterizer::visible_filter  (cuda_rasterizer/rasterizer.h:104-121, rasterizer_impl.cu:345-392).
 * radii[P], means2D[P,2]; both fully written. */
int sgr_visible_filter(int P, int width, int height, const float* means3D, const float* scales, float scale_modifier,
                       const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                       const float* projmatrix, float tan_fovx, float tan_fovy, int prefiltered, int* radii,
                       float* means2D, int debug, void* stream);

/* SimpleKNN::knn  (simple-knn/simple_knn.h:14-18, simple_knn.cu:185-220): mean squared distance to the three
 * nearest neighbours of each of the P points[P,3] -> meanDists[P]. */
int sgr_knn(int P, const float* points, float* meanDists, sgr_alloc_fn scratch, void* scratch_user, void* stream);

/* ---- sizes of the opaque buffers (so a caller may pre-allocate) ------------------------------------------- */
size_t sgr_geometry_bytes(int P);
size_t sgr_binning_bytes(int R);
size_t sgr_image_bytes(int width, int height);
int sgr_partial_row_floats(int S);

/* ---- per-stage timing with HIP events recorded on the caller's stream (bench.py's roofline leg) ---------------
 * stages: 0 preprocess 1 scan 2 duplicate 3 sort 4 tile_ranges 5 blend_fwd 6 partials memset 7 blend_bwd
 * 8 gauss_bwd.  sgr_profile_read sums the durations (ms) of the up-to-1024 stage executions recorded (by any host thread) since
 * sgr_profile_enable(1), writes 9 sums + 9 counts, resets the recorder and returns the number of records. */
int sgr_profile_enable(int on);
int sgr_profile_read(double* sum_ms, int* counts);
/* which stages are recorded (bit i = stage i; default all): a timed region that only needs the dominant kernel records
 * two events per step instead of eighteen.  stage_mask < 0 only queries; returns the previous mask. */
int sgr_profile_select(int stage_mask);
/* Records only every k-th occurrence of each selected stage (k >= 1; default 1): an event pair drains the queue
 * around the bracketed launch (~10 us each side), which a throughput measurement should not pay every step.
 * Returns the previous k; k < 1 only queries. */
int sgr_profile_sample(int every);

/* ---- introspection for parity tests: copies one internal array, densely packed, to dst (device). -------------
 * which: 0 depths f32[P] | 1 clamped u8[3P] | 2 means2D f32[2P] | 3 cov3D f32[6P] | 4 conic_opacity f32[4P]
 *        5 rgb f32[3P] | 6 tiles_touched u32[P] | 7 point_offsets u32[P] | 8 point_list u32[R]
 *        9 sorted keys u64[R] | 12 ranges u32[2T] | 13 n_contrib u32[H*W] | 14 extents f32[2P]
 *        15 hit record u8[R] (bit q: the forward blended the instance into quadrant q of its tile; entries behind the
 *           last batch a tile processed are undefined)
 *        16 tile rect u32[4P] {x0, y0, x1, y1}, upper corner exclusive (zeros for a culled Gaussian): the tiles the
 *           Gaussian was emitted for -- the reference's getRect square cut down to the tiles in which it can reach
 *           alpha >= 1/255 (switch bit 10: the reference's own rect); 6, 7, 8, 9, 12, 13 are the reference's arrays
 *           restricted to these rects
 *        17 u32[1]: what num_rendered is with the reference's rects (reporting)
 *        18 tile mask u64[P]: for rects of 2..64 tiles, bit j = tile (x0 + j % w, y0 + j / w) of the rect is emitted
 *           (inside the bounding box only the tiles the alpha >= 1/255 ellipse reaches); 0 = every tile of the rect
 *        19 compact hit list u32[R]: per tile, from its range's start, the list positions (relative to the range,
 *           ascending) of the instances whose hit record is non-zero -- what the blend backward walks; entries
 *           behind a tile's count (= the largest value of 20 over its pixels, at least) are undefined
 *        20 u32[H*W]: n_contrib counted in entries of 19 (a pixel's last contributor's index in its tile's list + 1) */
int sgr_export_internal(int which, int P, int R, int width, int height, char* geom_buffer, char* binning_buffer,
                        char* image_buffer, void* dst, void* stream);

/* ---- A/B switches of the blend kernels (tests, tools/gpu_ab.sh): bit 0 no quadrant cull, bit 1 no DPP wave
 * reduction, bit 2 no deterministic LDS combine, bit 3 the backward ignores the forward's hit record and redoes the
 * geometric cull, bit 4 the S = 0 backward runs the transposed-accumulation kernel (A/B design, slower),
 * bit 5 the radix sorts run in their one-sweep (decoupled look-back) form (A/B design, slower; SGR_ONESWEEP),
 * bit 7 (SGR_EXACT=1) PARITY MODE: the blend FORWARD evaluates the reference's own power expression, the device
 * library's expf and the unfused D += d * alpha * T of the depth, alpha and semantic sums -- alpha / depth / semantic images
 * and n_contrib bit-identical to the reference's kernels; the three COLOUR sums are formed as FMAs (their inputs, the SH
 * colours, already differ from the reference's in the last bit): the colour image is held to rel 1e-4, not bit-exact; the blend BACKWARD evaluates the reference's power expression and makes every blend / skip decision
 * exactly as the forward did (a visit with a pixel within 4e-6 of the 1/255 threshold falls back to the accurate expf), the
 * per-Gaussian backward runs without FP contraction -- gradients within rel 1e-4 end to end,
 * bit 8 (SGR_SW=1) the S = 0 blend backward runs its scalar-walk form (csrc/sgr_blend_bwd_sw.hip: A/B design, slower),
 * bit 9 (SGR_RS_WAVE=1) the per-Gaussian row sum runs its wave-cooperative form (A/B design, slower),
 * bit 6 (SGR_PRE_STAGE=1) the preprocess stages its SH rows through LDS whatever P is (default: from 3 M Gaussians),
 * bit 10 (SGR_REF_RECT=1) every Gaussian is emitted for the reference's whole tile rect (auxiliary.h getRect), so that
 * num_rendered, point_list, the sorted keys and the ranges are the reference's arrays entry for entry; default: the rect
 * cut down to the tiles where the Gaussian can pass the alpha >= 1/255 test -- fewer instances, bit-identical images (gradients: same terms, the row sum groups its additions differently),
 * bit 11 (SGR_NO_TILE_MASK=1) the cut-down rect is the bounding box of those tiles without the per-tile mask (A/B),
 * bit 13 (SGR_REF_RECT_PLAIN=1, with bit 10) the reference's rects without the marks described next (round-5 form, A/B).
 * Since round 6 bit 10 emits the reference's list with the instances that lie outside the cut-down rect / tile mask MARKED
 * (bit 31 of the internal list entry; export 8 strips it): they are counted, sorted and ranged exactly like the reference's
 * -- num_rendered, point_list, keys, ranges, n_contrib entry for entry -- but the blend kernels skip them without fetching
 * their record and they own no partial-gradient row.
 * bits 14 / 15: tile order of the two blend launches.  Default: decided per frame on the device -- one one-workgroup launch
 * per forward looks at the list lengths and, when the longest list is more than 2.5 x the mean (a street scene: empty sky
 * next to actors), sorts the tile ids longest list first; an even scene keeps the XCD-aware supertile order.  Same results
 * either way.  bit 14 (SGR_LPT=1) always longest-first, bit 15 (SGR_NO_LPT=1) never (and no extra launch).
 * bits 16 / 17: with the reference's rects (bit 10) the forward also writes the COMPACT list of the instances it blended
 * somewhere (exports 19 / 20) and the blend backward walks that list instead of the list positions (the marked-dead 40 % and
 * the instances behind saturated pixels are never staged); same gradients bit for bit.  bit 16 (SGR_NO_HLIST=1) never (the
 * round-5 walk, A/B), bit 17 (SGR_HLIST_ALWAYS=1) in every mode (with the cut-down rects it costs the forward more than it
 * saves the backward).  Read when the forward runs and recorded in the frame's buffers: the backward walks the list if its
 * frame has one (bit 16 at backward time forces the positional walk, which every frame supports).
 * bit 18 (SGR_KEY32=1) the instance list's tile keys stay 32-bit (default: 16-bit whenever the frame has fewer than 65535
 * tiles -- the tile sort then moves 6 instead of 8 bytes per pair and pass; same lists, A/B).
 * bit 12 (SGR_TILE_SORT=1) the binning chain runs in its per-tile form (csrc/sgr_tile_sort.hip: no depth pre-sort of the
 * Gaussians, emission in index order, stable tile sort, then every tile's list radix-sorted by depth in LDS) -- the same
 * lists entry for entry (tests/test_gpu_tile_sort.py); A/B design, measured in DESIGN.md section 3.
 * mask >= 0 sets them process-wide, mask < 0 only queries; returns the previous mask.  The initial
 * value comes from the environment (SGR_NO_CULL, SGR_NO_DPP, SGR_NO_DET, SGR_NO_HITS, SGR_V2), read once. */
int sgr_test_switches(int mask);
/* 1 when the library was built with -DSGR_WITH_VARIANTS=1 (tools/build_variant.py): it then also contains the designs that
 * were measured slower and are kept as A/Bs -- the transposed-accumulation and scalar-walk blend backward, the one-sweep
 * radix sorts, the wave-cooperative row sum (sgr_test_switches bits 4, 5, 8, 9).  The shipped library returns 0 and
 * ignores those bits. */
int sgr_has_variants(void);
/* ---- the forward without a host wait (extension; the reference blocks on num_rendered, rasterizer_impl.cu:284) ---------
 * sgr_set_lazy(1) (or SGR_LAZY=1 in the environment): from a thread's second forward after the call on (the first one
 * blocks and seeds the capacity from its own frame), the instance-list buffers get a
 * capacity derived from the previous frames' num_rendered (+ 1/16), duplicate / sort / tile ranges run over the whole
 * capacity, nothing waits for the read-back, and sgr_forward RETURNS THE CAPACITY (hand it to sgr_backward as R like any
 * num_rendered).  The checks the host would have made at the wait happen one call late: the next lazy sgr_forward of the
 * thread returns -SGR_E_LAZY when the previous frame had more instances than its capacity, a prefilter violation or a
 * depth beyond the 27-bit depth keys -- that frame's outputs are invalid -- and the thread then runs one blocking forward.
 * The only wait left on this path is that late check (the previous frame's read-back: it keeps a fast host one frame ahead);
 * inside a stream capture it is skipped and nothing synchronises, so forward + backward can be captured in a hipGraph (ask
 * sgr_lazy_status after a synchronisation of your own).  on < 0 only queries; returns the previous setting. */
int sgr_set_lazy(int on);
/* The thread's most recent lazy forward, once its stream has been synchronised: the frame's real num_rendered, the
 * capacity it ran with, flags (bit 0 overflow, bit 1 prefilter violation, bit 2 depth beyond the narrow depth sort). */
int sgr_lazy_status(int* num_rendered, int* capacity, int* flags);

/* Microseconds the process's threads have spent in sgr_forward's one host wait (the read-back of num_rendered,
 * rasterizer_impl.cu:284 in the reference) since the last reset; reset != 0 also clears the counter.  Measurement only: the
 * time a caller spends inside sgr_forward minus this is the host's own work. */
int sgr_profile_host_wait_us(int reset);

/* ---- primitive self-tests (used by tests/ on the GPU box) ---------------------------------------------------- */
int sgr_test_scan(const uint32_t* in, uint32_t* out, size_t n, int inclusive, uint32_t* tmp, void* stream);
/* keys0/vals0 hold the input; returns 0 or 1 = which of (keys0,vals0)/(keys1,vals1) holds the sorted result */
int sgr_test_sort(uint64_t* keys0, uint64_t* keys1, uint32_t* vals0, uint32_t* vals1, uint32_t n, int end_bit,
                  uint32_t* hist, uint32_t* scan_tmp, void* stream);
/* same with 32-bit keys (the tile sort and the depth pre-sort of the forward use this instantiation); max_bits = 8 or 9:
 * the widest digit (9 is what the depth pre-sort asks for below 750 k Gaussians: 27 key bits in three passes) */
int sgr_test_sort32(uint32_t* keys0, uint32_t* keys1, uint32_t* vals0, uint32_t* vals1, uint32_t n, int end_bit,
                    int max_bits, uint32_t* hist, uint32_t* scan_tmp, void* stream);
size_t sgr_test_sort_hist_words(uint32_t n);
size_t sgr_test_scan_tmp_words(size_t n);
int sgr_test_wave_sum(const float* in, float* out_dpp, float* out_shfl, int nwaves, void* stream);
/* the ordering property the per-tile LDS sort's ranking relies on (csrc/sgr_tile_sort.hip): trial t runs one wave64
 * ds_add_rtn_u32 in which lane l adds 1 to LDS counter pattern[64 t + l] % 64 (all counters zero before); out[64 t + l] = the
 * value lane l got back.  Lanes that share a counter must see 0, 1, 2, ... in ascending lane order. */
int sgr_test_lds_atomic_order(const uint32_t* pattern, uint32_t* out, int trials, void* stream);
/* parity-mode elementary functions against the toolchain's own (device arrays of n floats each): exp_lib[i] = expf(x[i]),
 * exp_ref[i] = the written-out sequence the blend kernels run in SGR_EXACT mode; div_lib[i] = a[i] / b[i] (hipcc's IEEE
 * expansion), div_ref[i] = the shared-reciprocal form.  The tests require bit equality over the kernels' operand ranges. */
int sgr_test_exact_math(int n, const float* x, float* exp_lib, float* exp_ref, const float* a, const float* b,
                        float* div_lib, float* div_ref, void* stream);

#ifdef __cplusplus
}
#endif
#endif
