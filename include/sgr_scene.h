/* sgr_scene.h -- C ABI of the scene-graph rows next to the rasterizer hot path.
 *
 * street_gaussians renders a scene graph: one static background model plus one rigid, per-frame posed Gaussian model
 * per visible actor.  Every iteration the reference flattens it with per-attribute torch.cat / einsum / quaternion
 * products (lib/models/street_gaussian_model.py:287-449) before calling the rasterizer, and scatters statistics back
 * per model afterwards (:551-571).  These entry points do each of the two in one pass over HBM.
 *
 * All pointers are DEVICE pointers unless noted; `segs` is a HOST array.  Plain C, no torch types. */
#ifndef SGR_SCENE_H
#define SGR_SCENE_H
#include <stddef.h>
#include <stdint.h>
#include "sgr.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SGR_SEG_STATIC 0 /* background: parameters used as they are            (gaussian_model.py:224-251)   */
#define SGR_SEG_ACTOR 1  /* rigid actor: local frame -> world by the frame pose (street_gaussian_model.py:305-364) */
#define SGR_SEM_LOGITS 0
#define SGR_SEM_PROBABILITIES 1

/* One sub-model of the scene graph, in the order the reference concatenates them (background first, then
 * graph_obj_list; street_gaussian_model.py:232-252). */
typedef struct sgr_scene_segment {
    int32_t count;       /* Gaussians of this model */
    int32_t kind;        /* SGR_SEG_STATIC / SGR_SEG_ACTOR */
    int32_t fourier_dim; /* DC coefficients per Gaussian: 1 for static models, the Fourier dimension for actors */
    int32_t class_label; /* actor: the semantic column its single value goes to (gaussian_model_actor.py:62-69) */
    int32_t sem_mode;    /* SGR_SEM_LOGITS / SGR_SEM_PROBABILITIES (gaussian_model.py:243-248) */
    int32_t flip_axis;   /* axis mirrored by a flip (1 in the reference, street_gaussian_model.py:58) */
    float flip_quat[4];  /* matrix_to_quaternion of the flip matrix (street_gaussian_model.py:59-61), real part first */
    const float* xyz;           /* [count,3] */
    const float* rotation;      /* [count,4] raw quaternion */
    const float* scaling;       /* [count,3] log-scale */
    const float* opacity;       /* [count,1] logit */
    const float* features_dc;   /* [count,fourier_dim,3] */
    const float* features_rest; /* [count,M-1,3] */
    const float* semantic;      /* static: [count,S]; actor: [count,1]; NULL with S == 0 */
    const uint8_t* flip_mask;   /* actor in training mode: [count] 0/1 (street_gaussian_model.py:276-285); NULL = none */
    const float* pose;          /* actor: 7 floats: obj_rot (w,x,y,z; not normalised) and obj_trans (:254-272) */
    const float* idft;          /* actor: fourier_dim floats, IDFT(time, fourier_dim) (gaussian_model_actor.py:71-80) */
} sgr_scene_segment;

/* Gradient outputs of one segment; NULL = not wanted.  Every non-NULL array is fully written (no zero-fill needed). */
typedef struct sgr_scene_segment_grads {
    float* xyz;
    float* rotation;
    float* scaling;
    float* opacity;
    float* features_dc;
    float* features_rest;
    float* semantic;
    float* pose; /* 7 floats */
} sgr_scene_segment_grads;

/* Flattens the scene graph into the rasterizer's inputs: means3D [N,3] (get_xyz, :335-364), rotations [N,4]
 * (get_rotation, :305-333), scales [N,3] (get_scaling, :287-303), opacities [N,1] (get_opacity, :433-449),
 * shs [N,M,3] (get_features, :366-381) and semantics [N,S] (get_semantic, :416-431), N = sum of counts.
 * scratch: a few KB of device memory for the segment tables.  Returns 0 or a negative SGR_E_* code. */
int sgr_scene_compose_forward(int K, const sgr_scene_segment* segs, int M, int S, float* means3D, float* rotations,
                              float* scales, float* opacities, float* shs, float* semantics, sgr_alloc_fn scratch,
                              void* scratch_user, void* stream);

/* Backward of the above: what autograd would propagate through the reference's cat / einsum / quaternion code. */
int sgr_scene_compose_backward(int K, const sgr_scene_segment* segs, const sgr_scene_segment_grads* grads, int M, int S,
                               const float* dL_dmeans3D, const float* dL_drotations, const float* dL_dscales,
                               const float* dL_dopacities, const float* dL_dshs, const float* dL_dsemantics,
                               sgr_alloc_fn scratch, void* scratch_user, void* stream);

/* Per-model densification statistics of one rendered view (street_gaussian_model.py:551-571): for every Gaussian
 * with radii > 0 (the visibility filter),  xyz_gradient_accum[:,0] += |dL/dmeans2D[:, :2]|,
 * xyz_gradient_accum[:,1] += |dL/dmeans2D[:, 2]|,  denom += 1,  max_radii2D = max(max_radii2D, radii). */
typedef struct sgr_scene_stats_segment {
    int32_t count;
    float* xyz_gradient_accum; /* [count,2] */
    float* denom;              /* [count,1] */
    float* max_radii2D;        /* [count]   */
} sgr_scene_stats_segment;
int sgr_scene_densification_stats(int K, const sgr_scene_stats_segment* segs, const float* dL_dmeans2D, const int* radii,
                                  sgr_alloc_fn scratch, void* scratch_user, void* stream);

#ifdef __cplusplus
}
#endif
#endif
