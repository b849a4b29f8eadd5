/* sgr_densify.h -- C ABI of adaptive density control: GaussianModel.densify_and_prune of the
 * reference (lib/models/gaussian_model.py:522-553 with densify_and_clone :494-520, densify_and_split :448-492,
 * prune_points :409-427 and the optimiser-state surgery :363-407) as a PLAN + GATHER:
 *
 *   1. sgr_densify_plan    one pass over the per-Gaussian statistics decides clone / split / prune for every point and
 *                          for the points it would create, scans the masks and returns the sizes;
 *   2. sgr_densify_map     writes, for every row of the RESULT, its source row and kind -- in the reference's order:
 *                          surviving originals (ascending), clones (ascending), split children (copy-major, :468-476);
 *   3. sgr_densify_gather  builds any per-Gaussian array of the result (parameters, Adam moments, ...) with one
 *                          row gather (new rows optionally zero, like the zeros_like extension of :396-397);
 *   4. sgr_densify_split_children  overwrites position and log-scale of the split children (:470-473).
 *
 * The reference does the same with ~150 masked-index / cat / repeat ops and three rounds of optimiser surgery.
 * All pointers are DEVICE pointers except `p` and `counts`. */
#ifndef SGR_DENSIFY_H
#define SGR_DENSIFY_H
#include <stddef.h>
#include <stdint.h>
#include "sgr.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sgr_densify_params {
    float max_grad;       /* densify_grad_threshold */
    float min_opacity;    /* prune: sigmoid(opacity) < min_opacity (:533) */
    float extent;         /* scene extent */
    float percent_dense;  /* clone if max scale <= percent_dense * extent, split if > (:463-464, :498-499) */
    float percent_big_ws; /* prune: max scale > extent * percent_big_ws, when prune_big (:536-540) */
    int32_t prune_big;    /* the reference's max_screen_size / prune_big_points being truthy */
    int32_t grad_column;  /* column of xyz_gradient_accum: 0 (:523), 1 = the "abs" variants (gaussian_model_bkgd.py:76-79) */
    int32_t n_split;      /* N of densify_and_split (2) */
    int32_t defer_prune;  /* 1: sgr_densify_plan / _map lay out ALL candidates (kept originals, clones, split children);
                             pruning is then decided per candidate row by sgr_densify_prune_mask -- needed by the
                             background and actor models, whose prune rules look at the NEW points' positions */
} sgr_densify_params;

#define SGR_KIND_KEEP 0
#define SGR_KIND_CLONE 1
#define SGR_KIND_SPLIT_CHILD 2

size_t sgr_densify_work_bytes(int N);

/* counts (host, written before the call returns; the call synchronises the stream like the reference's .item()s):
 * [0] points_total = N  [1] points_clone  [2] points_split  [3] points_pruned  [4] rows of the result
 * [5] rows of `normals` the split needs = n_split * points_split. */
int sgr_densify_plan(int N, const sgr_densify_params* p, const float* xyz_gradient_accum /*[N,2]*/,
                     const float* denom /*[N,1]*/, const float* scaling /*[N,3] log*/, const float* opacity /*[N,1] logit*/,
                     char* work, int64_t counts[6], void* stream);

/* src[i] = source row of result row i, kind[i] = SGR_KIND_*, sample_row[i] = row of `normals` for split children
 * (copy * points_split + rank of the parent among the split points, i.e. the reference's repeat(N, 1) order), else -1. */
int sgr_densify_map(int N, const sgr_densify_params* p, const char* work, int32_t* src, uint8_t* kind, int32_t* sample_row,
                    void* stream);

/* out[i, :] = in[src[i], :] (rows of `width` floats); with zero_new, rows whose kind != KEEP are zero-filled instead
 * (Adam's exp_avg / exp_avg_sq of new points, :396-397). */
int sgr_densify_gather(int n_out, int width, const float* in, const int32_t* src, const uint8_t* kind, int zero_new,
                       float* out, void* stream);

/* For kind == SPLIT_CHILD rows: xyz_out = R(rotation_in[src]) (normals[sample_row] * exp(scaling_in[src])) + xyz_in[src]
 * and scaling_out = log(exp(scaling_in[src]) / (0.8 * n_split))   (:466-473; normals ~ N(0,1), [counts[5],3]).
 * Other rows are left untouched (sgr_densify_gather filled them). */
int sgr_densify_split_children(int n_out, int n_split, const int32_t* src, const uint8_t* kind, const int32_t* sample_row,
                               const float* xyz_in, const float* scaling_in, const float* rotation_in,
                               const float* normals, float* xyz_out, float* scaling_out, void* stream);

/* ---- the prune rules of the models street_gaussians instantiates, evaluated on the candidate set -----------------
 * (candidates = rows laid out with defer_prune = 1; xyz / scaling / rotation / opacity are the candidates' RAW
 * parameters, i.e. gathered rows with the split children's position and log-scale already written.)
 *   variant SGR_PRUNE_BASE   GaussianModel.densify_and_prune (gaussian_model.py:532-543):
 *                            sigmoid(opacity) < min_opacity, or (prune_big and max scale > extent * percent_big_ws)
 *   variant SGR_PRUNE_BKGD   GaussianModelBkgd (gaussian_model_bkgd.py:91-104): the same, but a big point farther than
 *                            2 * sphere_radius from sphere_center is exempt; sphere = {cx, cy, cz, radius} (host)
 *   variant SGR_PRUNE_ACTOR  GaussianModelActor (gaussian_model_actor.py:226-252): additionally prunes a point when
 *                            either of two samples xyz + R(q) (z * scale) leaves the tracking box [box_min, box_max]
 *                            (box = {min xyz, max xyz}, host; box_normals [n, 2, 3] = the standard normals z of
 *                            torch.normal(mean=0, std=scale), device).  Only when prune_big, like the reference.
 * prune[i] = 1 for rows to drop.  counts (host; the call synchronises the stream): [0] below min opacity,
 * [1] big in world space (after the exemption), [2] outside the tracking box, [3] pruned. */
#define SGR_PRUNE_BASE 0
#define SGR_PRUNE_BKGD 1
#define SGR_PRUNE_ACTOR 2
int sgr_densify_prune_mask(int n, const sgr_densify_params* p, int variant, const float* xyz, const float* scaling,
                           const float* rotation, const float* opacity, const float* sphere, const float* box,
                           const float* box_normals, uint8_t* prune, int64_t counts[4], void* stream);

/* sel[0 .. *n_out) = ascending indices of the rows with prune[i] == 0.  work: sgr_densify_work_bytes(n) bytes.
 * *n_out is written before the call returns (the call synchronises the stream). */
int sgr_densify_compact(int n, const uint8_t* prune, char* work, int32_t* sel, int64_t* n_out, void* stream);

/* GaussianModel.reset_opacity (gaussian_model.py:410-414 with reset_optimizer :344-361):
 * opacity = inverse_sigmoid(min(sigmoid(opacity), 0.01)) in place; the two Adam moments of the group (may be NULL)
 * are zero-filled. */
int sgr_reset_opacity(int N, float* opacity, float* exp_avg, float* exp_avg_sq, void* stream);

#ifdef __cplusplus
}
#endif
#endif
