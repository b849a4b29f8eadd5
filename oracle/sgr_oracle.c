/*
 * sgr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C99 CPU restatement of the reference differentiable Gaussian rasterizer
 * (zju3dv/street_gaussians, submodules/diff-gaussian-rasterization) and of
 * simple-knn's distCUDA2.  It exists only to check the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (street_gaussians_amd/) never links, imports or calls anything in oracle/.
 *
 * Parity pin: the reference ships no golden vectors for this path, so
 * this restatement is pinned against the reference's own kernels compiled for gfx950
 * from the sources where they lie (oracle/ref_build.sh -> oracle/_ref/) and run on the
 * MI355X box; the resulting fixtures live in tests/golden/ (see tests/golden/README.md).
 *
 * Every function cites the reference file:line it restates.  Paths are relative to
 * /root/reference/submodules/diff-gaussian-rasterization/ (DGR) or
 * /root/reference/submodules/simple-knn/ (KNN).  GLM is column-major: mat3.c[col][row]
 * mirrors glm::mat3 m[col][row], and mat3_mul follows the evaluation order of
 * DGR/third_party/glm/glm/detail/type_mat3x3.inl:486-519 term by term so that, built
 * with -ffp-contract=off, the integer outputs (radii, tile rects, sort order) are
 * reproducible bit for bit.
 *
 * Accumulation order where the reference is unordered (float atomicAdd in the backward
 * blend): tiles ascending, pixels row-major inside the tile, instances in kernel order.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BLOCK_X 16 /* DGR/cuda_rasterizer/config.h:17 */
#define BLOCK_Y 16 /* DGR/cuda_rasterizer/config.h:18 */
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)
#define NUM_CHANNELS 3 /* DGR/cuda_rasterizer/config.h:15 */

/* DGR/cuda_rasterizer/auxiliary.h:22-39 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                              -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                              0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                              -0.5900435899266435f};

typedef struct { float x, y, z; } f3;
typedef struct { float x, y, z, w; } f4;
typedef struct { float x, y; } f2;
typedef struct { uint32_t x, y; } u2;
typedef struct { float c[3][3]; } mat3; /* c[col][row], as glm::mat3 */

static inline float fmaxf_(float a, float b) { return fmaxf(a, b); }
static inline float fminf_(float a, float b) { return fminf(a, b); }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* glm::mat3(x0,y0,z0, x1,y1,z1, x2,y2,z2): arguments fill columns */
static inline mat3 mat3_cols(float x0, float y0, float z0, float x1, float y1, float z1, float x2,
                             float y2, float z2) {
    mat3 m;
    m.c[0][0] = x0; m.c[0][1] = y0; m.c[0][2] = z0;
    m.c[1][0] = x1; m.c[1][1] = y1; m.c[1][2] = z1;
    m.c[2][0] = x2; m.c[2][1] = y2; m.c[2][2] = z2;
    return m;
}
/* DGR/third_party/glm/glm/detail/type_mat3x3.inl:486-519 */
static inline mat3 mat3_mul(mat3 a, mat3 b) {
    mat3 r;
    for (int c = 0; c < 3; c++)
        for (int row = 0; row < 3; row++)
            r.c[c][row] = a.c[0][row] * b.c[c][0] + a.c[1][row] * b.c[c][1] + a.c[2][row] * b.c[c][2];
    return r;
}
/* DGR/third_party/glm/glm/detail/func_matrix.inl (compute_transpose<3,3>) */
static inline mat3 mat3_transpose(mat3 m) {
    mat3 r;
    for (int c = 0; c < 3; c++)
        for (int row = 0; row < 3; row++) r.c[c][row] = m.c[row][c];
    return r;
}
static inline mat3 mat3_scale(float s, mat3 m) { /* type_mat3x3.inl:459-466 (scalar * m) */
    mat3 r;
    for (int c = 0; c < 3; c++)
        for (int row = 0; row < 3; row++) r.c[c][row] = m.c[c][row] * s;
    return r;
}
static inline float dot3(const float a[3], const float b[3]) { /* func_geometric.inl:48-55 */
    float t0 = a[0] * b[0], t1 = a[1] * b[1], t2 = a[2] * b[2];
    return t0 + t1 + t2;
}

/* DGR/cuda_rasterizer/auxiliary.h:41-44 -- double arithmetic, stored as float */
static inline float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

/* DGR/cuda_rasterizer/auxiliary.h:46-56 */
static inline void getRect(f2 p, int max_radius, u2* rect_min, u2* rect_max, uint32_t gx, uint32_t gy) {
    rect_min->x = (uint32_t)imin((int)gx, imax(0, (int)((p.x - max_radius) / BLOCK_X)));
    rect_min->y = (uint32_t)imin((int)gy, imax(0, (int)((p.y - max_radius) / BLOCK_Y)));
    rect_max->x = (uint32_t)imin((int)gx, imax(0, (int)((p.x + max_radius + BLOCK_X - 1) / BLOCK_X)));
    rect_max->y = (uint32_t)imin((int)gy, imax(0, (int)((p.y + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}
/* DGR/cuda_rasterizer/auxiliary.h:58-66 */
static inline f3 transformPoint4x3(f3 p, const float* m) {
    f3 t = {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
    return t;
}
/* DGR/cuda_rasterizer/auxiliary.h:68-77 */
static inline f4 transformPoint4x4(f3 p, const float* m) {
    f4 t = {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]};
    return t;
}
/* DGR/cuda_rasterizer/auxiliary.h:89-97 */
static inline f3 transformVec4x3Transpose(f3 p, const float* m) {
    f3 t = {m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
            m[8] * p.x + m[9] * p.y + m[10] * p.z};
    return t;
}
/* DGR/cuda_rasterizer/auxiliary.h:107-117 */
static inline f3 dnormvdv(f3 v, f3 dv) {
    float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    f3 r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
}
/* DGR/cuda_rasterizer/auxiliary.h:139-164 (prefiltered trap omitted: returns -1 to the caller) */
static inline int in_frustum(int idx, const float* orig_points, const float* viewmatrix, f3* p_view) {
    f3 p_orig = {orig_points[3 * idx], orig_points[3 * idx + 1], orig_points[3 * idx + 2]};
    *p_view = transformPoint4x3(p_orig, viewmatrix);
    if (p_view->z <= 0.2f) return 0;
    return 1;
}

/* DGR/cuda_rasterizer/rasterizer_impl.cu:35-50 */
static uint32_t getHigherMsb(uint32_t n) {
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step;
        else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}
uint32_t sgo_get_higher_msb(uint32_t n) { return getHigherMsb(n); }

/* ------------------------------------------------------------------------------------------ */
/* forward.cu:20-71  computeColorFromSH                                                       */
static void computeColorFromSH(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                               const float* shs, uint8_t* clamped, float out[3]) {
    float dir[3] = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
    float len = sqrtf(dot3(dir, dir));
    dir[0] = dir[0] / len; dir[1] = dir[1] / len; dir[2] = dir[2] / len;
    const float* sh = shs + (size_t)idx * max_coeffs * 3;
#define SHV(k, ch) sh[(k) * 3 + (ch)]
    float x = dir[0], y = dir[1], z = dir[2];
    for (int ch = 0; ch < 3; ch++) {
        float result = SH_C0 * SHV(0, ch);
        if (deg > 0) {
            result = result - SH_C1 * y * SHV(1, ch) + SH_C1 * z * SHV(2, ch) - SH_C1 * x * SHV(3, ch);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z;
                float xy = x * y, yz = y * z, xz = x * z;
                result = result + SH_C2[0] * xy * SHV(4, ch) + SH_C2[1] * yz * SHV(5, ch) +
                         SH_C2[2] * (2.0f * zz - xx - yy) * SHV(6, ch) + SH_C2[3] * xz * SHV(7, ch) +
                         SH_C2[4] * (xx - yy) * SHV(8, ch);
                if (deg > 2) {
                    result = result + SH_C3[0] * y * (3.0f * xx - yy) * SHV(9, ch) +
                             SH_C3[1] * xy * z * SHV(10, ch) +
                             SH_C3[2] * y * (4.0f * zz - xx - yy) * SHV(11, ch) +
                             SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SHV(12, ch) +
                             SH_C3[4] * x * (4.0f * zz - xx - yy) * SHV(13, ch) +
                             SH_C3[5] * z * (xx - yy) * SHV(14, ch) + SH_C3[6] * x * (xx - 3.0f * yy) * SHV(15, ch);
                }
            }
        }
        result += 0.5f;
        clamped[3 * idx + ch] = (result < 0);
        out[ch] = fmaxf_(result, 0.0f);
    }
#undef SHV
}

/* forward.cu:74-113  computeCov2D */
static f3 computeCov2D(f3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy, const float* cov3D,
                       const float* viewmatrix) {
    f3 t = transformPoint4x3(mean, viewmatrix);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = t.x / t.z;
    const float tytz = t.y / t.z;
    t.x = fminf_(limx, fmaxf_(-limx, txtz)) * t.z;
    t.y = fminf_(limy, fmaxf_(-limy, tytz)) * t.z;

    mat3 J = mat3_cols(focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z), 0.0f, focal_y / t.z,
                       -(focal_y * t.y) / (t.z * t.z), 0, 0, 0);
    mat3 W = mat3_cols(viewmatrix[0], viewmatrix[4], viewmatrix[8], viewmatrix[1], viewmatrix[5], viewmatrix[9],
                       viewmatrix[2], viewmatrix[6], viewmatrix[10]);
    mat3 T = mat3_mul(W, J);
    mat3 Vrk = mat3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    mat3 cov = mat3_mul(mat3_mul(mat3_transpose(T), mat3_transpose(Vrk)), T);
    cov.c[0][0] += 0.3f;
    cov.c[1][1] += 0.3f;
    f3 r = {cov.c[0][0], cov.c[0][1], cov.c[1][1]};
    return r;
}

/* forward.cu:118-152  computeCov3D */
static void computeCov3D(const float* scale, float mod, const float* rot, float* cov3D) {
    mat3 S = mat3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.c[0][0] = mod * scale[0];
    S.c[1][1] = mod * scale[1];
    S.c[2][2] = mod * scale[2];
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3]; /* NOT normalised (forward.cu:127) */
    mat3 R = mat3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                       2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                       2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    mat3 M = mat3_mul(S, R);
    mat3 Sigma = mat3_mul(mat3_transpose(M), M);
    cov3D[0] = Sigma.c[0][0];
    cov3D[1] = Sigma.c[0][1];
    cov3D[2] = Sigma.c[0][2];
    cov3D[3] = Sigma.c[1][1];
    cov3D[4] = Sigma.c[1][2];
    cov3D[5] = Sigma.c[2][2];
}

/* ------------------------------------------------------------------------------------------ */
/* State that the reference keeps in its three opaque byte buffers
 * (rasterizer_impl.h:29-64), exposed here for parity checks.                               */
typedef struct {
    int P, W, H, S, M, D;
    uint32_t gx, gy;
    int R;
    /* GeometryState */
    float* depths;        /* [P]   */
    uint8_t* clamped;     /* [3P]  */
    int* radii;           /* [P]   */
    f2* means2D;          /* [P]   */
    float* cov3D;         /* [6P]  */
    f4* conic_opacity;    /* [P]   */
    float* rgb;           /* [3P]  */
    uint32_t* tiles_touched; /* [P] */
    uint32_t* point_offsets; /* [P] inclusive scan */
    /* BinningState */
    uint64_t* keys_unsorted; /* [R] */
    uint32_t* vals_unsorted; /* [R] */
    uint64_t* keys;          /* [R] sorted */
    uint32_t* point_list;    /* [R] sorted */
    /* ImageState */
    u2* ranges;           /* [T] */
    uint32_t* n_contrib;  /* [N] */
} sgo_state;

void sgo_free(sgo_state* s) {
    if (!s) return;
    free(s->depths); free(s->clamped); free(s->radii); free(s->means2D); free(s->cov3D);
    free(s->conic_opacity); free(s->rgb); free(s->tiles_touched); free(s->point_offsets);
    free(s->keys_unsorted); free(s->vals_unsorted); free(s->keys); free(s->point_list);
    free(s->ranges); free(s->n_contrib);
    free(s);
}

/* accessors for ctypes */
int sgo_num_rendered(const sgo_state* s) { return s->R; }
const float* sgo_depths(const sgo_state* s) { return s->depths; }
const uint8_t* sgo_clamped(const sgo_state* s) { return s->clamped; }
const int* sgo_radii(const sgo_state* s) { return s->radii; }
const float* sgo_means2D(const sgo_state* s) { return (const float*)s->means2D; }
const float* sgo_cov3D(const sgo_state* s) { return s->cov3D; }
const float* sgo_conic_opacity(const sgo_state* s) { return (const float*)s->conic_opacity; }
const float* sgo_rgb(const sgo_state* s) { return s->rgb; }
const uint32_t* sgo_tiles_touched(const sgo_state* s) { return s->tiles_touched; }
const uint32_t* sgo_point_offsets(const sgo_state* s) { return s->point_offsets; }
const uint64_t* sgo_keys_unsorted(const sgo_state* s) { return s->keys_unsorted; }
const uint32_t* sgo_vals_unsorted(const sgo_state* s) { return s->vals_unsorted; }
const uint64_t* sgo_keys(const sgo_state* s) { return s->keys; }
const uint32_t* sgo_point_list(const sgo_state* s) { return s->point_list; }
const uint32_t* sgo_ranges(const sgo_state* s) { return (const uint32_t*)s->ranges; }
const uint32_t* sgo_n_contrib(const sgo_state* s) { return s->n_contrib; }

/* forward.cu:155-256  preprocessCUDA (one Gaussian).  filter==1 restates
 * filter_preprocessCUDA (forward.cu:259-334): no SH/colour/depth/conic outputs. */
static void preprocess_one(int idx, sgo_state* s, int D, int M, const float* orig_points, const float* scales,
                           float scale_modifier, const float* rotations, const float* opacities, const float* shs,
                           const float* cov3D_precomp, const float* colors_precomp, const float* viewmatrix,
                           const float* projmatrix, const float* cam_pos, int W, int H, float tan_fovx,
                           float tan_fovy, float focal_x, float focal_y, int* radii, float* filter_means2D,
                           int filter) {
    radii[idx] = 0;
    if (!filter) s->tiles_touched[idx] = 0;

    f3 p_view;
    if (!in_frustum(idx, orig_points, viewmatrix, &p_view)) return;

    f3 p_orig = {orig_points[3 * idx], orig_points[3 * idx + 1], orig_points[3 * idx + 2]};
    f4 p_hom = transformPoint4x4(p_orig, projmatrix);
    float p_w = 1.0f / (p_hom.w + 0.0000001f);
    f3 p_proj = {p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w};

    const float* cov3D;
    if (cov3D_precomp != NULL) {
        cov3D = cov3D_precomp + idx * 6;
    } else {
        computeCov3D(scales + 3 * idx, scale_modifier, rotations + 4 * idx, s->cov3D + idx * 6);
        cov3D = s->cov3D + idx * 6;
    }
    f3 cov = computeCov2D(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix);

    float det = (cov.x * cov.z - cov.y * cov.y);
    if (det == 0.0f) return;
    float det_inv = 1.f / det;
    f3 conic = {cov.z * det_inv, -cov.y * det_inv, cov.x * det_inv};

    float mid = 0.5f * (cov.x + cov.z);
    float lambda1 = mid + sqrtf(fmaxf_(0.1f, mid * mid - det));
    float lambda2 = mid - sqrtf(fmaxf_(0.1f, mid * mid - det));
    float my_radius = ceilf(3.f * sqrtf(fmaxf_(lambda1, lambda2)));
    f2 point_image = {ndc2Pix(p_proj.x, W), ndc2Pix(p_proj.y, H)};
    u2 rect_min, rect_max;
    getRect(point_image, (int)my_radius, &rect_min, &rect_max, s->gx, s->gy);
    if ((rect_max.x - rect_min.x) * (rect_max.y - rect_min.y) == 0) return;

    if (filter) {
        radii[idx] = (int)my_radius;
        filter_means2D[2 * idx] = point_image.x;
        filter_means2D[2 * idx + 1] = point_image.y;
        return;
    }
    if (colors_precomp == NULL) {
        float result[3];
        computeColorFromSH(idx, D, M, orig_points, cam_pos, shs, s->clamped, result);
        s->rgb[idx * 3 + 0] = result[0];
        s->rgb[idx * 3 + 1] = result[1];
        s->rgb[idx * 3 + 2] = result[2];
    }
    s->depths[idx] = p_view.z;
    radii[idx] = (int)my_radius;
    s->means2D[idx] = point_image;
    f4 co = {conic.x, conic.y, conic.z, opacities[idx]};
    s->conic_opacity[idx] = co;
    s->tiles_touched[idx] = (rect_max.y - rect_min.y) * (rect_max.x - rect_min.x);
}

/* stable LSD radix sort on key bits [0, end_bit): the result CUB's
 * DeviceRadixSort::SortPairs is specified to give (rasterizer_impl.cu:306-311). */
static void stable_sort_pairs(const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout, int n,
                              int end_bit) {
    if (n <= 0) return;
    uint64_t* ka = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
    uint64_t* kb = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
    uint32_t* va = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
    uint32_t* vb = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
    memcpy(ka, kin, sizeof(uint64_t) * (size_t)n);
    memcpy(va, vin, sizeof(uint32_t) * (size_t)n);
    for (int shift = 0; shift < end_bit; shift += 8) {
        int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        uint32_t mask = (1u << bits) - 1u;
        size_t count[257];
        memset(count, 0, sizeof(count));
        for (int i = 0; i < n; i++) count[((ka[i] >> shift) & mask) + 1]++;
        for (int d = 0; d < 256; d++) count[d + 1] += count[d];
        for (int i = 0; i < n; i++) {
            size_t pos = count[(ka[i] >> shift) & mask]++;
            kb[pos] = ka[i];
            vb[pos] = va[i];
        }
        uint64_t* tk = ka; ka = kb; kb = tk;
        uint32_t* tv = va; va = vb; vb = tv;
    }
    memcpy(kout, ka, sizeof(uint64_t) * (size_t)n);
    memcpy(vout, va, sizeof(uint32_t) * (size_t)n);
    free(ka); free(kb); free(va); free(vb);
}

/* forward.cu:340-467  renderCUDA for one tile */
static void render_tile(const sgo_state* s, uint32_t tx, uint32_t ty, int W, int H, int S, const float* features,
                        const float* semantics, const float* bg_color, float* out_color, float* out_depth,
                        float* out_alpha, float* out_semantic) {
    const uint32_t horizontal_blocks = (W + BLOCK_X - 1) / BLOCK_X;
    u2 range = s->ranges[ty * horizontal_blocks + tx];
    for (uint32_t ly = 0; ly < BLOCK_Y; ly++) {
        for (uint32_t lx = 0; lx < BLOCK_X; lx++) {
            uint32_t px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
            if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
            uint32_t pix_id = W * py + px;
            f2 pixf = {(float)px, (float)py};
            float T = 1.0f;
            uint32_t contributor = 0, last_contributor = 0;
            float C[NUM_CHANNELS] = {0};
            float weight = 0, Dp = 0;
            int done = 0;
            for (uint32_t k = range.x; k < range.y && !done; k++) {
                contributor++;
                uint32_t id = s->point_list[k];
                f2 xy = s->means2D[id];
                f2 d = {xy.x - pixf.x, xy.y - pixf.y};
                f4 con_o = s->conic_opacity[id];
                float power = -0.5f * (con_o.x * d.x * d.x + con_o.z * d.y * d.y) - con_o.y * d.x * d.y;
                if (power > 0.0f) continue;
                float alpha = fminf_(0.99f, con_o.w * expf(power));
                if (alpha < 1.0f / 255.0f) continue;
                float test_T = T * (1 - alpha);
                if (test_T < 0.0001f) { done = 1; continue; }
                for (int ch = 0; ch < NUM_CHANNELS; ch++) C[ch] += features[id * NUM_CHANNELS + ch] * alpha * T;
                for (int ch = 0; ch < S; ch++)
                    out_semantic[(size_t)ch * H * W + pix_id] += semantics[(size_t)id * S + ch] * alpha * T;
                weight += alpha * T;
                Dp += s->depths[id] * alpha * T;
                T = test_T;
                last_contributor = contributor;
            }
            s->n_contrib[pix_id] = last_contributor;
            for (int ch = 0; ch < NUM_CHANNELS; ch++) out_color[(size_t)ch * H * W + pix_id] = C[ch] + T * bg_color[ch];
            out_alpha[pix_id] = weight;
            out_depth[pix_id] = Dp;
        }
    }
}

/* rasterizer_impl.cu:197-343  Rasterizer::forward.
 * Outputs must be zero-initialised by the caller, as rasterize_points.cu:70-74 does.
 * radii may be NULL (internal radii are then used, rasterizer_impl.cu:232-235). */
sgo_state* sgo_forward(int P, int D, int M, int S, const float* background, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp, const float* semantics,
                       const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                       const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                       const float* cam_pos, float tan_fovx, float tan_fovy, float* out_color, float* out_depth,
                       float* out_alpha, float* out_semantic, int* radii_out) {
    sgo_state* s = (sgo_state*)calloc(1, sizeof(sgo_state));
    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);
    s->P = P; s->W = width; s->H = height; s->S = S; s->M = M; s->D = D;
    s->gx = (width + BLOCK_X - 1) / BLOCK_X;
    s->gy = (height + BLOCK_Y - 1) / BLOCK_Y;
    size_t Pn = P > 0 ? (size_t)P : 1;
    s->depths = (float*)calloc(Pn, sizeof(float));
    s->clamped = (uint8_t*)calloc(Pn * 3, 1);
    s->radii = (int*)calloc(Pn, sizeof(int));
    s->means2D = (f2*)calloc(Pn, sizeof(f2));
    s->cov3D = (float*)calloc(Pn * 6, sizeof(float));
    s->conic_opacity = (f4*)calloc(Pn, sizeof(f4));
    s->rgb = (float*)calloc(Pn * 3, sizeof(float));
    s->tiles_touched = (uint32_t*)calloc(Pn, sizeof(uint32_t));
    s->point_offsets = (uint32_t*)calloc(Pn, sizeof(uint32_t));
    size_t N = (size_t)width * height, T = (size_t)s->gx * s->gy;
    s->n_contrib = (uint32_t*)calloc(N ? N : 1, sizeof(uint32_t));
    s->ranges = (u2*)calloc(T ? T : 1, sizeof(u2));
    int* radii = radii_out ? radii_out : s->radii;

#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++)
        preprocess_one(idx, s, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, cov3D_precomp,
                       colors_precomp, viewmatrix, projmatrix, cam_pos, width, height, tan_fovx, tan_fovy, focal_x,
                       focal_y, radii, NULL, 0);
    if (radii_out) memcpy(s->radii, radii_out, sizeof(int) * (size_t)P);

    /* rasterizer_impl.cu:280  InclusiveSum */
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) { acc += s->tiles_touched[i]; s->point_offsets[i] = acc; }
    int R = P > 0 ? (int)s->point_offsets[P - 1] : 0;
    s->R = R;
    size_t Rn = R > 0 ? (size_t)R : 1;
    s->keys_unsorted = (uint64_t*)calloc(Rn, sizeof(uint64_t));
    s->vals_unsorted = (uint32_t*)calloc(Rn, sizeof(uint32_t));
    s->keys = (uint64_t*)calloc(Rn, sizeof(uint64_t));
    s->point_list = (uint32_t*)calloc(Rn, sizeof(uint32_t));

    /* rasterizer_impl.cu:70-111  duplicateWithKeys */
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : s->point_offsets[idx - 1];
            u2 rect_min, rect_max;
            getRect(s->means2D[idx], radii[idx], &rect_min, &rect_max, s->gx, s->gy);
            for (uint32_t y = rect_min.y; y < rect_max.y; y++)
                for (uint32_t x = rect_min.x; x < rect_max.x; x++) {
                    uint64_t key = (uint64_t)y * s->gx + x;
                    key <<= 32;
                    uint32_t dbits;
                    memcpy(&dbits, &s->depths[idx], 4);
                    key |= dbits;
                    s->keys_unsorted[off] = key;
                    s->vals_unsorted[off] = (uint32_t)idx;
                    off++;
                }
        }
    }
    /* rasterizer_impl.cu:303-311 */
    int bit = (int)getHigherMsb(s->gx * s->gy);
    stable_sort_pairs(s->keys_unsorted, s->vals_unsorted, s->keys, s->point_list, R, 32 + bit);

    /* rasterizer_impl.cu:116-138  identifyTileRanges (ranges zeroed by calloc == cudaMemset :313) */
    for (int idx = 0; idx < R; idx++) {
        uint32_t currtile = (uint32_t)(s->keys[idx] >> 32);
        if (idx == 0) s->ranges[currtile].x = 0;
        else {
            uint32_t prevtile = (uint32_t)(s->keys[idx - 1] >> 32);
            if (currtile != prevtile) { s->ranges[prevtile].y = idx; s->ranges[currtile].x = idx; }
        }
        if (idx == R - 1) s->ranges[currtile].y = R;
    }

    const float* feature_ptr = colors_precomp != NULL ? colors_precomp : s->rgb;
    int ntiles = (int)T;
#pragma omp parallel for schedule(dynamic, 8)
    for (int t = 0; t < ntiles; t++)
        render_tile(s, (uint32_t)t % s->gx, (uint32_t)t / s->gx, width, height, S, feature_ptr, semantics, background,
                    out_color, out_depth, out_alpha, out_semantic);
    return s;
}

/* rasterizer_impl.cu:345-392  Rasterizer::visible_filter */
void sgo_visible_filter(int P, int width, int height, const float* means3D, const float* scales,
                        float scale_modifier, const float* rotations, const float* cov3D_precomp,
                        const float* viewmatrix, const float* projmatrix, float tan_fovx, float tan_fovy, int* radii,
                        float* means2D) {
    sgo_state s;
    memset(&s, 0, sizeof(s));
    s.gx = (width + BLOCK_X - 1) / BLOCK_X;
    s.gy = (height + BLOCK_Y - 1) / BLOCK_Y;
    s.cov3D = (float*)calloc(P > 0 ? (size_t)P * 6 : 1, sizeof(float));
    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);
    for (int idx = 0; idx < P; idx++)
        preprocess_one(idx, &s, 0, 0, means3D, scales, scale_modifier, rotations, NULL, NULL, cov3D_precomp, NULL,
                       viewmatrix, projmatrix, NULL, width, height, tan_fovx, tan_fovy, focal_x, focal_y, radii,
                       means2D, 1);
    free(s.cov3D);
}

/* rasterizer_impl.cu:54-66,141-153  checkFrustum / markVisible */
void sgo_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                      uint8_t* present) {
    (void)projmatrix;
    for (int idx = 0; idx < P; idx++) {
        f3 pv;
        present[idx] = (uint8_t)in_frustum(idx, means3D, viewmatrix, &pv);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* backward.cu:415-641  renderCUDA (backward) for one tile; float adds replace atomicAdd.    */
#define S_MAX 20 /* NUM_CLASSES, config.h:16 */
/* double shadows of the six arrays the tile kernel accumulates into (members named like the parameters) */
typedef struct { double *dL_dmean2D, *dL_dconic2D, *dL_dopacity, *dL_dcolors, *dL_ddepths, *dL_dsemantics; } sgo_acc64;
static void render_backward_tile(const sgo_state* s, uint32_t tx, uint32_t ty, int W, int H, int S,
                                 const float* bg_color, const float* colors, const float* semantics,
                                 const float* alphas, const float* dL_dpixels, const float* dL_dpixel_depths,
                                 const float* dL_dalphas, const float* dL_dpixel_semantics, float* dL_dmean2D,
                                 float* dL_dconic2D, float* dL_dopacity, float* dL_dcolors, float* dL_ddepths,
                                 float* dL_dsemantics, int use_atomics, const sgo_acc64* a64) {
    const int C = NUM_CHANNELS;
    const uint32_t horizontal_blocks = (W + BLOCK_X - 1) / BLOCK_X;
    const u2 range = s->ranges[ty * horizontal_blocks + tx];
    const int toDo0 = (int)(range.y - range.x);
    const float ddelx_dx = (float)(0.5 * W);
    const float ddely_dy = (float)(0.5 * H);
    /* a64 != NULL: the float terms (computed exactly as the reference computes them) are summed in double and rounded
     * to float once at the end (sgo_backward), so the result is independent of the order of the adds to ~1e-16: an
     * order-free statement of what the reference's unordered float atomicAdd computes, usable with all threads at
     * the BASELINE sizes. */
#define ADDF(arr, index, val)                                    \
    do {                                                         \
        float v__ = (val);                                       \
        if (a64) {                                               \
            _Pragma("omp atomic") a64->arr[index] += (double)v__; \
        } else if (use_atomics) {                                \
            _Pragma("omp atomic") arr[index] += v__;             \
        } else {                                                 \
            arr[index] += v__;                                   \
        }                                                        \
    } while (0)
    for (uint32_t ly = 0; ly < BLOCK_Y; ly++) {
        for (uint32_t lx = 0; lx < BLOCK_X; lx++) {
            uint32_t px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
            if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
            const uint32_t pix_id = W * py + px;
            const f2 pixf = {(float)px, (float)py};
            const float T_final = 1 - alphas[pix_id];
            float T = T_final;
            uint32_t contributor = (uint32_t)toDo0;
            const int last_contributor = (int)s->n_contrib[pix_id];
            float accum_rec[NUM_CHANNELS] = {0};
            float dL_dpixel[NUM_CHANNELS];
            float accum_depth_rec = 0, accum_alpha_rec = 0;
            float accum_semantic_rec[S_MAX] = {0};
            float dL_dpixel_semantic[S_MAX];
            for (int i = 0; i < C; i++) dL_dpixel[i] = dL_dpixels[(size_t)i * H * W + pix_id];
            for (int i = 0; i < S; i++) dL_dpixel_semantic[i] = dL_dpixel_semantics[(size_t)i * H * W + pix_id];
            const float dL_dpixel_depth = dL_dpixel_depths[pix_id];
            const float dL_dalpha = dL_dalphas[pix_id];
            float last_alpha = 0;
            float last_color[NUM_CHANNELS] = {0};
            float last_depth = 0;
            float last_semantic[S_MAX] = {0};

            for (int k = (int)range.y - 1; k >= (int)range.x; k--) {
                contributor--;
                if ((int)contributor >= last_contributor) continue;
                const uint32_t global_id = s->point_list[k];
                const f2 xy = s->means2D[global_id];
                const f2 d = {xy.x - pixf.x, xy.y - pixf.y};
                const f4 con_o = s->conic_opacity[global_id];
                const float power = -0.5f * (con_o.x * d.x * d.x + con_o.z * d.y * d.y) - con_o.y * d.x * d.y;
                if (power > 0.0f) continue;
                const float G = expf(power);
                const float alpha = fminf_(0.99f, con_o.w * G);
                if (alpha < 1.0f / 255.0f) continue;

                T = T / (1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                const float dpixel_depth_ddepth = alpha * T;
                float dL_dopa = 0.0f;
                for (int ch = 0; ch < C; ch++) {
                    const float c = colors[global_id * C + ch];
                    accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                    last_color[ch] = c;
                    const float dL_dchannel = dL_dpixel[ch];
                    dL_dopa += (c - accum_rec[ch]) * dL_dchannel;
                    ADDF(dL_dcolors, global_id * C + ch, dchannel_dcolor * dL_dchannel);
                }
                const float dchannel_dsemantic = alpha * T;
                for (int ch = 0; ch < S; ch++) {
                    const float sv = semantics[(size_t)global_id * S + ch];
                    accum_semantic_rec[ch] = last_alpha * last_semantic[ch] + (1.f - last_alpha) * accum_semantic_rec[ch];
                    last_semantic[ch] = sv;
                    const float dL_dchannel = dL_dpixel_semantic[ch];
                    dL_dopa += (sv - accum_semantic_rec[ch]) * dL_dchannel;
                    ADDF(dL_dsemantics, (size_t)global_id * S + ch, dchannel_dsemantic * dL_dchannel);
                }
                const float c_d = s->depths[global_id];
                accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
                last_depth = c_d;
                dL_dopa += (c_d - accum_depth_rec) * dL_dpixel_depth;
                ADDF(dL_ddepths, global_id, dpixel_depth_ddepth * dL_dpixel_depth);

                accum_alpha_rec = last_alpha + (1.f - last_alpha) * accum_alpha_rec;
                dL_dopa += (1 - accum_alpha_rec) * dL_dalpha;
                dL_dopa *= T;
                last_alpha = alpha;

                float bg_dot_dpixel = 0;
                for (int i = 0; i < C; i++) bg_dot_dpixel += bg_color[i] * dL_dpixel[i];
                dL_dopa += (-T_final / (1.f - alpha)) * bg_dot_dpixel;

                const float dL_dG = con_o.w * dL_dopa;
                const float gdx = G * d.x;
                const float gdy = G * d.y;
                const float dG_ddelx = -gdx * con_o.x - gdy * con_o.y;
                const float dG_ddely = -gdy * con_o.z - gdx * con_o.y;

                ADDF(dL_dmean2D, 3 * global_id + 0, dL_dG * dG_ddelx * ddelx_dx);
                ADDF(dL_dmean2D, 3 * global_id + 1, dL_dG * dG_ddely * ddely_dy);
                const float abs_dL_dmean2D = fabsf(dL_dG * dG_ddelx * ddelx_dx) + fabsf(dL_dG * dG_ddely * ddely_dy);
                ADDF(dL_dmean2D, 3 * global_id + 2, abs_dL_dmean2D);

                ADDF(dL_dconic2D, 4 * global_id + 0, -0.5f * gdx * d.x * dL_dG);
                ADDF(dL_dconic2D, 4 * global_id + 1, -0.5f * gdx * d.y * dL_dG);
                ADDF(dL_dconic2D, 4 * global_id + 3, -0.5f * gdy * d.y * dL_dG);
                ADDF(dL_dopacity, global_id, G * dL_dopa);
            }
        }
    }
#undef ADDF
}

/* backward.cu:20-139  computeColorFromSH (backward) */
static void computeColorFromSH_bw(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                                  const float* shs, const uint8_t* clamped, const float* dL_dcolor, float* dL_dmeans,
                                  float* dL_dshs) {
    float dir_orig[3] = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
    float len = sqrtf(dot3(dir_orig, dir_orig));
    float dir[3] = {dir_orig[0] / len, dir_orig[1] / len, dir_orig[2] / len};
    const float* sh = shs + (size_t)idx * max_coeffs * 3;
    float dL_dRGB[3] = {dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]};
    dL_dRGB[0] *= clamped[3 * idx + 0] ? 0 : 1;
    dL_dRGB[1] *= clamped[3 * idx + 1] ? 0 : 1;
    dL_dRGB[2] *= clamped[3 * idx + 2] ? 0 : 1;
    float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
    float x = dir[0], y = dir[1], z = dir[2];
    float* dL_dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
#define SHV(k, ch) sh[(k) * 3 + (ch)]
#define DSH(k, ch) dL_dsh[(k) * 3 + (ch)]
    for (int ch = 0; ch < 3; ch++) {
        float g = dL_dRGB[ch];
        float dRGBdsh0 = SH_C0;
        DSH(0, ch) = dRGBdsh0 * g;
        if (deg > 0) {
            float dRGBdsh1 = -SH_C1 * y, dRGBdsh2 = SH_C1 * z, dRGBdsh3 = -SH_C1 * x;
            DSH(1, ch) = dRGBdsh1 * g;
            DSH(2, ch) = dRGBdsh2 * g;
            DSH(3, ch) = dRGBdsh3 * g;
            dRGBdx[ch] = -SH_C1 * SHV(3, ch);
            dRGBdy[ch] = -SH_C1 * SHV(1, ch);
            dRGBdz[ch] = SH_C1 * SHV(2, ch);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z;
                float xy = x * y, yz = y * z, xz = x * z;
                DSH(4, ch) = (SH_C2[0] * xy) * g;
                DSH(5, ch) = (SH_C2[1] * yz) * g;
                DSH(6, ch) = (SH_C2[2] * (2.f * zz - xx - yy)) * g;
                DSH(7, ch) = (SH_C2[3] * xz) * g;
                DSH(8, ch) = (SH_C2[4] * (xx - yy)) * g;
                dRGBdx[ch] += SH_C2[0] * y * SHV(4, ch) + SH_C2[2] * 2.f * -x * SHV(6, ch) + SH_C2[3] * z * SHV(7, ch) +
                              SH_C2[4] * 2.f * x * SHV(8, ch);
                dRGBdy[ch] += SH_C2[0] * x * SHV(4, ch) + SH_C2[1] * z * SHV(5, ch) + SH_C2[2] * 2.f * -y * SHV(6, ch) +
                              SH_C2[4] * 2.f * -y * SHV(8, ch);
                dRGBdz[ch] += SH_C2[1] * y * SHV(5, ch) + SH_C2[2] * 2.f * 2.f * z * SHV(6, ch) + SH_C2[3] * x * SHV(7, ch);
                if (deg > 2) {
                    DSH(9, ch) = (SH_C3[0] * y * (3.f * xx - yy)) * g;
                    DSH(10, ch) = (SH_C3[1] * xy * z) * g;
                    DSH(11, ch) = (SH_C3[2] * y * (4.f * zz - xx - yy)) * g;
                    DSH(12, ch) = (SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * g;
                    DSH(13, ch) = (SH_C3[4] * x * (4.f * zz - xx - yy)) * g;
                    DSH(14, ch) = (SH_C3[5] * z * (xx - yy)) * g;
                    DSH(15, ch) = (SH_C3[6] * x * (xx - 3.f * yy)) * g;
                    dRGBdx[ch] += (SH_C3[0] * SHV(9, ch) * 3.f * 2.f * xy + SH_C3[1] * SHV(10, ch) * yz +
                                   SH_C3[2] * SHV(11, ch) * -2.f * xy + SH_C3[3] * SHV(12, ch) * -3.f * 2.f * xz +
                                   SH_C3[4] * SHV(13, ch) * (-3.f * xx + 4.f * zz - yy) +
                                   SH_C3[5] * SHV(14, ch) * 2.f * xz + SH_C3[6] * SHV(15, ch) * 3.f * (xx - yy));
                    dRGBdy[ch] += (SH_C3[0] * SHV(9, ch) * 3.f * (xx - yy) + SH_C3[1] * SHV(10, ch) * xz +
                                   SH_C3[2] * SHV(11, ch) * (-3.f * yy + 4.f * zz - xx) +
                                   SH_C3[3] * SHV(12, ch) * -3.f * 2.f * yz + SH_C3[4] * SHV(13, ch) * -2.f * xy +
                                   SH_C3[5] * SHV(14, ch) * -2.f * yz + SH_C3[6] * SHV(15, ch) * -3.f * 2.f * xy);
                    dRGBdz[ch] += (SH_C3[1] * SHV(10, ch) * xy + SH_C3[2] * SHV(11, ch) * 4.f * 2.f * yz +
                                   SH_C3[3] * SHV(12, ch) * 3.f * (2.f * zz - xx - yy) +
                                   SH_C3[4] * SHV(13, ch) * 4.f * 2.f * xz + SH_C3[5] * SHV(14, ch) * (xx - yy));
                }
            }
        }
    }
#undef SHV
#undef DSH
    f3 dL_ddir = {dot3(dRGBdx, dL_dRGB), dot3(dRGBdy, dL_dRGB), dot3(dRGBdz, dL_dRGB)};
    f3 dorig = {dir_orig[0], dir_orig[1], dir_orig[2]};
    f3 dL_dmean = dnormvdv(dorig, dL_ddir);
    dL_dmeans[3 * idx + 0] += dL_dmean.x;
    dL_dmeans[3 * idx + 1] += dL_dmean.y;
    dL_dmeans[3 * idx + 2] += dL_dmean.z;
}

/* backward.cu:144-274  computeCov2DCUDA */
static void computeCov2D_bw(int idx, const float* means, const int* radii, const float* cov3Ds, float h_x, float h_y,
                            float tan_fovx, float tan_fovy, const float* view_matrix, const float* dL_dconics,
                            float* dL_dmeans, float* dL_dcov) {
    if (!(radii[idx] > 0)) return;
    const float* cov3D = cov3Ds + 6 * idx;
    f3 mean = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
    f3 dL_dconic = {dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]};
    f3 t = transformPoint4x3(mean, view_matrix);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = t.x / t.z;
    const float tytz = t.y / t.z;
    t.x = fminf_(limx, fmaxf_(-limx, txtz)) * t.z;
    t.y = fminf_(limy, fmaxf_(-limy, tytz)) * t.z;
    const float x_grad_mul = txtz < -limx || txtz > limx ? 0 : 1;
    const float y_grad_mul = tytz < -limy || tytz > limy ? 0 : 1;

    mat3 J = mat3_cols(h_x / t.z, 0.0f, -(h_x * t.x) / (t.z * t.z), 0.0f, h_y / t.z, -(h_y * t.y) / (t.z * t.z), 0, 0, 0);
    mat3 W = mat3_cols(view_matrix[0], view_matrix[4], view_matrix[8], view_matrix[1], view_matrix[5], view_matrix[9],
                       view_matrix[2], view_matrix[6], view_matrix[10]);
    mat3 Vrk = mat3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    mat3 T = mat3_mul(W, J);
    mat3 cov2D = mat3_mul(mat3_mul(mat3_transpose(T), mat3_transpose(Vrk)), T);
#define Tm(i, j) T.c[i][j]
#define Vm(i, j) Vrk.c[i][j]
#define Wm(i, j) W.c[i][j]
    float a = cov2D.c[0][0] += 0.3f;
    float b = cov2D.c[0][1];
    float c = cov2D.c[1][1] += 0.3f;
    float denom = a * c - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    if (denom2inv != 0) {
        dL_da = denom2inv * (-c * c * dL_dconic.x + 2 * b * c * dL_dconic.y + (denom - a * c) * dL_dconic.z);
        dL_dc = denom2inv * (-a * a * dL_dconic.z + 2 * a * b * dL_dconic.y + (denom - a * c) * dL_dconic.x);
        dL_db = denom2inv * 2 * (b * c * dL_dconic.x - (denom + 2 * b * b) * dL_dconic.y + a * b * dL_dconic.z);
        dL_dcov[6 * idx + 0] = (Tm(0, 0) * Tm(0, 0) * dL_da + Tm(0, 0) * Tm(1, 0) * dL_db + Tm(1, 0) * Tm(1, 0) * dL_dc);
        dL_dcov[6 * idx + 3] = (Tm(0, 1) * Tm(0, 1) * dL_da + Tm(0, 1) * Tm(1, 1) * dL_db + Tm(1, 1) * Tm(1, 1) * dL_dc);
        dL_dcov[6 * idx + 5] = (Tm(0, 2) * Tm(0, 2) * dL_da + Tm(0, 2) * Tm(1, 2) * dL_db + Tm(1, 2) * Tm(1, 2) * dL_dc);
        dL_dcov[6 * idx + 1] = 2 * Tm(0, 0) * Tm(0, 1) * dL_da + (Tm(0, 0) * Tm(1, 1) + Tm(0, 1) * Tm(1, 0)) * dL_db +
                               2 * Tm(1, 0) * Tm(1, 1) * dL_dc;
        dL_dcov[6 * idx + 2] = 2 * Tm(0, 0) * Tm(0, 2) * dL_da + (Tm(0, 0) * Tm(1, 2) + Tm(0, 2) * Tm(1, 0)) * dL_db +
                               2 * Tm(1, 0) * Tm(1, 2) * dL_dc;
        dL_dcov[6 * idx + 4] = 2 * Tm(0, 2) * Tm(0, 1) * dL_da + (Tm(0, 1) * Tm(1, 2) + Tm(0, 2) * Tm(1, 1)) * dL_db +
                               2 * Tm(1, 1) * Tm(1, 2) * dL_dc;
    } else {
        for (int i = 0; i < 6; i++) dL_dcov[6 * idx + i] = 0;
    }
    float dL_dT00 = 2 * (Tm(0, 0) * Vm(0, 0) + Tm(0, 1) * Vm(0, 1) + Tm(0, 2) * Vm(0, 2)) * dL_da +
                    (Tm(1, 0) * Vm(0, 0) + Tm(1, 1) * Vm(0, 1) + Tm(1, 2) * Vm(0, 2)) * dL_db;
    float dL_dT01 = 2 * (Tm(0, 0) * Vm(1, 0) + Tm(0, 1) * Vm(1, 1) + Tm(0, 2) * Vm(1, 2)) * dL_da +
                    (Tm(1, 0) * Vm(1, 0) + Tm(1, 1) * Vm(1, 1) + Tm(1, 2) * Vm(1, 2)) * dL_db;
    float dL_dT02 = 2 * (Tm(0, 0) * Vm(2, 0) + Tm(0, 1) * Vm(2, 1) + Tm(0, 2) * Vm(2, 2)) * dL_da +
                    (Tm(1, 0) * Vm(2, 0) + Tm(1, 1) * Vm(2, 1) + Tm(1, 2) * Vm(2, 2)) * dL_db;
    float dL_dT10 = 2 * (Tm(1, 0) * Vm(0, 0) + Tm(1, 1) * Vm(0, 1) + Tm(1, 2) * Vm(0, 2)) * dL_dc +
                    (Tm(0, 0) * Vm(0, 0) + Tm(0, 1) * Vm(0, 1) + Tm(0, 2) * Vm(0, 2)) * dL_db;
    float dL_dT11 = 2 * (Tm(1, 0) * Vm(1, 0) + Tm(1, 1) * Vm(1, 1) + Tm(1, 2) * Vm(1, 2)) * dL_dc +
                    (Tm(0, 0) * Vm(1, 0) + Tm(0, 1) * Vm(1, 1) + Tm(0, 2) * Vm(1, 2)) * dL_db;
    float dL_dT12 = 2 * (Tm(1, 0) * Vm(2, 0) + Tm(1, 1) * Vm(2, 1) + Tm(1, 2) * Vm(2, 2)) * dL_dc +
                    (Tm(0, 0) * Vm(2, 0) + Tm(0, 1) * Vm(2, 1) + Tm(0, 2) * Vm(2, 2)) * dL_db;
    float dL_dJ00 = Wm(0, 0) * dL_dT00 + Wm(0, 1) * dL_dT01 + Wm(0, 2) * dL_dT02;
    float dL_dJ02 = Wm(2, 0) * dL_dT00 + Wm(2, 1) * dL_dT01 + Wm(2, 2) * dL_dT02;
    float dL_dJ11 = Wm(1, 0) * dL_dT10 + Wm(1, 1) * dL_dT11 + Wm(1, 2) * dL_dT12;
    float dL_dJ12 = Wm(2, 0) * dL_dT10 + Wm(2, 1) * dL_dT11 + Wm(2, 2) * dL_dT12;
#undef Tm
#undef Vm
#undef Wm
    float tz = 1.f / t.z;
    float tz2 = tz * tz;
    float tz3 = tz2 * tz;
    float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 +
                   (2 * h_y * t.y) * tz3 * dL_dJ12;
    f3 dt = {dL_dtx, dL_dty, dL_dtz};
    f3 dL_dmean = transformVec4x3Transpose(dt, view_matrix);
    dL_dmeans[3 * idx + 0] = dL_dmean.x; /* ASSIGN, backward.cu:273 */
    dL_dmeans[3 * idx + 1] = dL_dmean.y;
    dL_dmeans[3 * idx + 2] = dL_dmean.z;
}

/* backward.cu:278-341  computeCov3D (backward) */
static void computeCov3D_bw(int idx, const float* scale, float mod, const float* rot, const float* dL_dcov3Ds,
                            float* dL_dscales, float* dL_drots) {
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    mat3 R = mat3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                       2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                       2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    mat3 S = mat3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    S.c[0][0] = s[0];
    S.c[1][1] = s[1];
    S.c[2][2] = s[2];
    mat3 M = mat3_mul(S, R);
    const float* dL_dcov3D = dL_dcov3Ds + 6 * idx;
    mat3 dL_dSigma = mat3_cols(dL_dcov3D[0], 0.5f * dL_dcov3D[1], 0.5f * dL_dcov3D[2], 0.5f * dL_dcov3D[1],
                               dL_dcov3D[3], 0.5f * dL_dcov3D[4], 0.5f * dL_dcov3D[2], 0.5f * dL_dcov3D[4],
                               dL_dcov3D[5]);
    mat3 dL_dM = mat3_mul(mat3_scale(2.0f, M), dL_dSigma);
    mat3 Rt = mat3_transpose(R);
    mat3 dL_dMt = mat3_transpose(dL_dM);
    float* dL_dscale = dL_dscales + 3 * idx;
    dL_dscale[0] = dot3(Rt.c[0], dL_dMt.c[0]);
    dL_dscale[1] = dot3(Rt.c[1], dL_dMt.c[1]);
    dL_dscale[2] = dot3(Rt.c[2], dL_dMt.c[2]);
    for (int k = 0; k < 3; k++) {
        dL_dMt.c[0][k] *= s[0];
        dL_dMt.c[1][k] *= s[1];
        dL_dMt.c[2][k] *= s[2];
    }
#define Mt(i, j) dL_dMt.c[i][j]
    float q0 = 2 * z * (Mt(0, 1) - Mt(1, 0)) + 2 * y * (Mt(2, 0) - Mt(0, 2)) + 2 * x * (Mt(1, 2) - Mt(2, 1));
    float q1 = 2 * y * (Mt(1, 0) + Mt(0, 1)) + 2 * z * (Mt(2, 0) + Mt(0, 2)) + 2 * r * (Mt(1, 2) - Mt(2, 1)) -
               4 * x * (Mt(2, 2) + Mt(1, 1));
    float q2 = 2 * x * (Mt(1, 0) + Mt(0, 1)) + 2 * r * (Mt(2, 0) - Mt(0, 2)) + 2 * z * (Mt(1, 2) + Mt(2, 1)) -
               4 * y * (Mt(2, 2) + Mt(0, 0));
    float q3 = 2 * r * (Mt(0, 1) - Mt(1, 0)) + 2 * x * (Mt(2, 0) + Mt(0, 2)) + 2 * y * (Mt(1, 2) + Mt(2, 1)) -
               4 * z * (Mt(1, 1) + Mt(0, 0));
#undef Mt
    float* dL_drot = dL_drots + 4 * idx;
    dL_drot[0] = q0; dL_drot[1] = q1; dL_drot[2] = q2; dL_drot[3] = q3;
}

/* backward.cu:346-412  preprocessCUDA (backward) */
static void preprocess_bw(int idx, int D, int M, const float* means, const int* radii, const float* shs,
                          const uint8_t* clamped, const float* scales, const float* rotations, float scale_modifier,
                          const float* view, const float* proj, const float* campos, const float* dL_dmean2D,
                          float* dL_dmeans, float* dL_dcolor, float* dL_ddepth, float* dL_dcov3D, float* dL_dsh,
                          float* dL_dscale, float* dL_drot) {
    if (!(radii[idx] > 0)) return;
    f3 m = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
    f4 m_hom = transformPoint4x4(m, proj);
    float m_w = 1.0f / (m_hom.w + 0.0000001f);
    float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
    float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
    float gx = dL_dmean2D[3 * idx], gy = dL_dmean2D[3 * idx + 1];
    float dmx = (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
    float dmy = (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
    float dmz = (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
    dL_dmeans[3 * idx + 0] += dmx;
    dL_dmeans[3 * idx + 1] += dmy;
    dL_dmeans[3 * idx + 2] += dmz;

    float mul3 = view[2] * m.x + view[6] * m.y + view[10] * m.z + view[14];
    float d2x = (view[2] - view[3] * mul3) * dL_ddepth[idx];
    float d2y = (view[6] - view[7] * mul3) * dL_ddepth[idx];
    float d2z = (view[10] - view[11] * mul3) * dL_ddepth[idx];
    dL_dmeans[3 * idx + 0] += d2x;
    dL_dmeans[3 * idx + 1] += d2y;
    dL_dmeans[3 * idx + 2] += d2z;

    if (shs) computeColorFromSH_bw(idx, D, M, means, campos, shs, clamped, dL_dcolor, dL_dmeans, dL_dsh);
    if (scales) computeCov3D_bw(idx, scales + 3 * idx, scale_modifier, rotations + 4 * idx, dL_dcov3D, dL_dscale, dL_drot);
}

/* rasterizer_impl.cu:396-506  Rasterizer::backward.  All dL_d* outputs must be
 * zero-initialised by the caller (rasterize_points.cu:166-176).
 * parallel == 1 uses unordered omp-atomic float adds like the reference's atomicAdd;
 * parallel == 0 gives the deterministic order stated at the top of this file;
 * parallel == 2 sums the same float terms in double (all threads), rounded to float once: order-free. */
void sgo_backward(const sgo_state* s, int D, int M, int S, const float* background, const float* means3D,
                  const float* shs, const float* colors_precomp, const float* semantics, const float* alphas,
                  const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                  const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                  float tan_fovy, const int* radii, const float* dL_dpix, const float* dL_dpix_depth,
                  const float* dL_dalphas, const float* dL_dpix_semantic, float* dL_dmean2D, float* dL_dconic,
                  float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D,
                  float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dsemantic, int parallel) {
    const int P = s->P, width = s->W, height = s->H;
    if (radii == NULL) radii = s->radii;
    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);
    const float* color_ptr = (colors_precomp != NULL) ? colors_precomp : s->rgb;
    int ntiles = (int)(s->gx * s->gy);
    if (parallel == 2) {
        const size_t n[6] = {3 * (size_t)P, 4 * (size_t)P, (size_t)P, 3 * (size_t)P, (size_t)P, (size_t)S * (size_t)P};
        float* f[6] = {dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_ddepth, dL_dsemantic};
        double* d[6];
        for (int i = 0; i < 6; i++) d[i] = (double*)calloc(n[i] ? n[i] : 1, sizeof(double));
        const sgo_acc64 a64 = {d[0], d[1], d[2], d[3], d[4], d[5]};
#pragma omp parallel for schedule(dynamic, 8)
        for (int t = 0; t < ntiles; t++)
            render_backward_tile(s, (uint32_t)t % s->gx, (uint32_t)t / s->gx, width, height, S, background, color_ptr,
                                 semantics, alphas, dL_dpix, dL_dpix_depth, dL_dalphas, dL_dpix_semantic, dL_dmean2D,
                                 dL_dconic, dL_dopacity, dL_dcolor, dL_ddepth, dL_dsemantic, 1, &a64);
        for (int i = 0; i < 6; i++) {
            for (size_t k = 0; k < n[i]; k++) f[i][k] += (float)d[i][k];
            free(d[i]);
        }
    } else if (parallel) {
#pragma omp parallel for schedule(dynamic, 8)
        for (int t = 0; t < ntiles; t++)
            render_backward_tile(s, (uint32_t)t % s->gx, (uint32_t)t / s->gx, width, height, S, background, color_ptr,
                                 semantics, alphas, dL_dpix, dL_dpix_depth, dL_dalphas, dL_dpix_semantic, dL_dmean2D,
                                 dL_dconic, dL_dopacity, dL_dcolor, dL_ddepth, dL_dsemantic, 1, NULL);
    } else {
        for (int t = 0; t < ntiles; t++)
            render_backward_tile(s, (uint32_t)t % s->gx, (uint32_t)t / s->gx, width, height, S, background, color_ptr,
                                 semantics, alphas, dL_dpix, dL_dpix_depth, dL_dalphas, dL_dpix_semantic, dL_dmean2D,
                                 dL_dconic, dL_dopacity, dL_dcolor, dL_ddepth, dL_dsemantic, 0, NULL);
    }
    const float* cov3D_ptr = (cov3D_precomp != NULL) ? cov3D_precomp : s->cov3D;
#pragma omp parallel for schedule(static) if (parallel)
    for (int idx = 0; idx < P; idx++)
        computeCov2D_bw(idx, means3D, radii, cov3D_ptr, focal_x, focal_y, tan_fovx, tan_fovy, viewmatrix, dL_dconic,
                        dL_dmean3D, dL_dcov3D);
#pragma omp parallel for schedule(static) if (parallel)
    for (int idx = 0; idx < P; idx++)
        preprocess_bw(idx, D, M, means3D, radii, shs, s->clamped, scales, rotations, scale_modifier, viewmatrix,
                      projmatrix, campos, dL_dmean2D, dL_dmean3D, dL_dcolor, dL_ddepth, dL_dcov3D, dL_dsh, dL_dscale,
                      dL_drot);
}

/* ========================================================================================== */
/* simple-knn: KNN/simple_knn.cu                                                              */
#define BOX_SIZE 1024 /* simple_knn.cu:12 */
typedef struct { f3 minn, maxx; } MinMax;

static uint32_t prepMorton(uint32_t x) { /* simple_knn.cu:45-52 */
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}
static uint32_t coord2Morton(f3 coord, f3 minn, f3 maxx) { /* simple_knn.cu:54-61 */
    uint32_t x = prepMorton((uint32_t)(((coord.x - minn.x) / (maxx.x - minn.x)) * ((1 << 10) - 1)));
    uint32_t y = prepMorton((uint32_t)(((coord.y - minn.y) / (maxx.y - minn.y)) * ((1 << 10) - 1)));
    uint32_t z = prepMorton((uint32_t)(((coord.z - minn.z) / (maxx.z - minn.z)) * ((1 << 10) - 1)));
    return x | (y << 1) | (z << 2);
}
static float distBoxPoint(const MinMax* box, f3 p) { /* simple_knn.cu:119-129 */
    f3 diff = {0, 0, 0};
    if (p.x < box->minn.x || p.x > box->maxx.x) diff.x = fminf_(fabsf(p.x - box->minn.x), fabsf(p.x - box->maxx.x));
    if (p.y < box->minn.y || p.y > box->maxx.y) diff.y = fminf_(fabsf(p.y - box->minn.y), fabsf(p.y - box->maxx.y));
    if (p.z < box->minn.z || p.z > box->maxx.z) diff.z = fminf_(fabsf(p.z - box->minn.z), fabsf(p.z - box->maxx.z));
    return diff.x * diff.x + diff.y * diff.y + diff.z * diff.z;
}
static void updateKBest3(f3 ref, f3 point, float* knn) { /* simple_knn.cu:131-145 */
    f3 d = {point.x - ref.x, point.y - ref.y, point.z - ref.z};
    float dist = d.x * d.x + d.y * d.y + d.z * d.z;
    for (int j = 0; j < 3; j++) {
        if (knn[j] > dist) {
            float t = knn[j];
            knn[j] = dist;
            dist = t;
        }
    }
}
static f3 ldp(const float* pts, uint32_t i) { f3 p = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}; return p; }

/* simple_knn.cu:185-220  SimpleKNN::knn.  morton_out / indices_out may be NULL. */
void sgo_knn(int P, const float* points, float* meanDists, uint32_t* morton_out, uint32_t* indices_out) {
    if (P <= 0) return;
    /* simple_knn.cu:191-200: Reduce with init {0,0,0} */
    f3 minn = {0, 0, 0}, maxx = {0, 0, 0};
    for (int i = 0; i < P; i++) {
        f3 p = ldp(points, (uint32_t)i);
        minn.x = fminf_(minn.x, p.x); minn.y = fminf_(minn.y, p.y); minn.z = fminf_(minn.z, p.z);
        maxx.x = fmaxf_(maxx.x, p.x); maxx.y = fmaxf_(maxx.y, p.y); maxx.z = fmaxf_(maxx.z, p.z);
    }
    uint64_t* mk = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)P);
    uint64_t* mks = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)P);
    uint32_t* idx = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)P);
    uint32_t* idxs = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)P);
    for (int i = 0; i < P; i++) {
        mk[i] = coord2Morton(ldp(points, (uint32_t)i), minn, maxx);
        idx[i] = (uint32_t)i; /* thrust::sequence, simple_knn.cu:207 */
    }
    stable_sort_pairs(mk, idx, mks, idxs, P, 32); /* simple_knn.cu:210-213 */
    if (morton_out) for (int i = 0; i < P; i++) morton_out[i] = (uint32_t)mks[i];
    if (indices_out) memcpy(indices_out, idxs, sizeof(uint32_t) * (size_t)P);

    uint32_t num_boxes = ((uint32_t)P + BOX_SIZE - 1) / BOX_SIZE;
    MinMax* boxes = (MinMax*)malloc(sizeof(MinMax) * num_boxes);
    /* simple_knn.cu:78-117  boxMinMax */
    for (uint32_t b = 0; b < num_boxes; b++) {
        MinMax me = {{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
        for (uint32_t i = b * BOX_SIZE; i < (uint32_t)P && i < (b + 1) * BOX_SIZE; i++) {
            f3 p = ldp(points, idxs[i]);
            me.minn.x = fminf_(me.minn.x, p.x); me.minn.y = fminf_(me.minn.y, p.y); me.minn.z = fminf_(me.minn.z, p.z);
            me.maxx.x = fmaxf_(me.maxx.x, p.x); me.maxx.y = fmaxf_(me.maxx.y, p.y); me.maxx.z = fmaxf_(me.maxx.z, p.z);
        }
        boxes[b] = me;
    }
    /* simple_knn.cu:147-183  boxMeanDist */
#pragma omp parallel for schedule(dynamic, 256)
    for (int ii = 0; ii < P; ii++) {
        f3 point = ldp(points, idxs[ii]);
        float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        for (int i = imax(0, ii - 3); i <= imin(P - 1, ii + 3); i++) {
            if (i == ii) continue;
            updateKBest3(point, ldp(points, idxs[i]), best);
        }
        float reject = best[2];
        best[0] = FLT_MAX; best[1] = FLT_MAX; best[2] = FLT_MAX;
        for (uint32_t b = 0; b < num_boxes; b++) {
            MinMax box = boxes[b];
            float dist = distBoxPoint(&box, point);
            if (dist > reject || dist > best[2]) continue;
            for (int i = (int)(b * BOX_SIZE); i < imin(P, (int)((b + 1) * BOX_SIZE)); i++) {
                if (i == ii) continue;
                updateKBest3(point, ldp(points, idxs[i]), best);
            }
        }
        meanDists[idxs[ii]] = (best[0] + best[1] + best[2]) / 3.0f;
    }
    free(mk); free(mks); free(idx); free(idxs); free(boxes);
}
