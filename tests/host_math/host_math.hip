// TEST INFRASTRUCTURE.  Host (x86) build of the __host__ __device__ maths in
// street_gaussians_amd/csrc/sgr_math.h, so the per-Gaussian formulas the gfx950 kernels run can be
// checked against the oracle without a GPU (tests/test_host_math.py).  Built with
// `hipcc -x hip --cuda-host-only`; never loaded by the product.
#include "../../street_gaussians_amd/csrc/sgr_math.h"

static SgrCam make_cam(const float* view, const float* proj, const float* campos, float tanx, float tany, int W, int H,
                       float mod) {
    SgrCam c;
    for (int i = 0; i < 16; i++) { c.view[i] = view[i]; c.proj[i] = proj[i]; }
    for (int i = 0; i < 3; i++) c.campos[i] = campos ? campos[i] : 0.f;
    c.tan_fovx = tanx; c.tan_fovy = tany;
    c.focal_y = H / (2.0f * tany);
    c.focal_x = W / (2.0f * tanx);
    c.W = W; c.H = H;
    c.gx = (W + 15) / 16; c.gy = (H + 15) / 16;
    c.scale_modifier = mod;
    return c;
}

extern "C" {

void hm_forward(int P, int D, int M, const float* means, const float* scales, const float* rots, const float* opac,
                const float* shs, const float* view, const float* proj, const float* campos, float tanx, float tany,
                int W, int H, float mod, int* radii, float* means2D, float* depths, float* conic, float* cov3D,
                unsigned* tiles, float* rgb, unsigned char* clamped, float* extents) {
    const SgrCam cam = make_cam(view, proj, campos, tanx, tany, W, H, mod);
    for (int i = 0; i < P; i++) {
        radii[i] = 0; tiles[i] = 0;
        sgr_cov3d(scales + 3 * i, mod, rots + 4 * i, cov3D + 6 * i);
        SgrProj pr = sgr_project(means + 3 * i, cov3D + 6 * i, cam);
        depths[i] = pr.depth;
        if (!pr.ok) continue;
        radii[i] = pr.radius;
        tiles[i] = (pr.rx1 - pr.rx0) * (pr.ry1 - pr.ry0);
        means2D[2 * i] = pr.px; means2D[2 * i + 1] = pr.py;
        conic[3 * i] = pr.con_x; conic[3 * i + 1] = pr.con_y; conic[3 * i + 2] = pr.con_z;
        sgr_extent(opac[i], pr.cov_a, pr.cov_c, pr.con_x, pr.con_y, pr.con_z, extents[2 * i], extents[2 * i + 1]);
        // SH colour exactly as sgr_preprocess_kernel does it
        float dx = means[3 * i] - cam.campos[0], dy = means[3 * i + 1] - cam.campos[1], dz = means[3 * i + 2] - cam.campos[2];
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        dx /= len; dy /= len; dz /= len;
        float Y[16];
        sgr_sh_basis(D, dx, dy, dz, Y);
        const float* sh = shs + (size_t)i * M * 3;
        float c[3] = {0, 0, 0};
        for (int k = 0; k < (D + 1) * (D + 1); k++)
            for (int ch = 0; ch < 3; ch++) c[ch] = (k == 0) ? Y[0] * sh[ch] : c[ch] + Y[k] * sh[3 * k + ch];
        for (int ch = 0; ch < 3; ch++) {
            c[ch] += 0.5f;
            clamped[3 * i + ch] = c[ch] < 0;
            rgb[3 * i + ch] = fmaxf(c[ch], 0.f);
        }
    }
}

// per-Gaussian backward from the blend-stage gradients, as sgr_gauss_bwd_kernel composes it
void hm_backward(int P, int D, int M, const float* means, const float* scales, const float* rots, const float* shs,
                 const float* cov3D, const unsigned char* clamped, const int* radii, const float* view, const float* proj,
                 const float* campos, float tanx, float tany, int W, int H, float mod, const float* dmean2D,
                 const float* dconic, const float* dcolor, const float* ddepth, float* dmean3D, float* dcov3D,
                 float* dscale, float* drot, float* dsh) {
    const SgrCam cam = make_cam(view, proj, campos, tanx, tany, W, H, mod);
    for (int i = 0; i < P; i++) {
        if (!(radii[i] > 0)) continue;
        float dmean[3], dcov[6];
        sgr_cov2d_backward(means + 3 * i, cov3D + 6 * i, cam, dconic[4 * i], dconic[4 * i + 1], dconic[4 * i + 3], dcov, dmean);
        sgr_proj_depth_backward(means + 3 * i, cam, dmean2D[3 * i], dmean2D[3 * i + 1], ddepth[i], dmean);
        sgr_cov3d_backward(scales + 3 * i, mod, rots + 4 * i, dcov, dscale + 3 * i, drot + 4 * i);
        float dRGB[3];
        for (int ch = 0; ch < 3; ch++) dRGB[ch] = clamped[3 * i + ch] ? 0.f : dcolor[3 * i + ch];
        float dor[3] = {means[3 * i] - cam.campos[0], means[3 * i + 1] - cam.campos[1], means[3 * i + 2] - cam.campos[2]};
        const float len = sqrtf(dor[0] * dor[0] + dor[1] * dor[1] + dor[2] * dor[2]);
        const float dir[3] = {dor[0] / len, dor[1] / len, dor[2] / len};
        float Y[16];
        sgr_sh_basis(D, dir[0], dir[1], dir[2], Y);
        const float* sh = shs + (size_t)i * M * 3;
        float t[16] = {0};
        for (int k = 0; k < (D + 1) * (D + 1); k++)
            for (int ch = 0; ch < 3; ch++) { t[k] += sh[3 * k + ch] * dRGB[ch]; dsh[((size_t)i * M + k) * 3 + ch] = Y[k] * dRGB[ch]; }
        float ddir[3], dm[3];
        sgr_sh_dir_backward(D, dir[0], dir[1], dir[2], t, ddir);
        sgr_dnormvdv(dor, ddir, dm);
        for (int k = 0; k < 3; k++) dmean3D[3 * i + k] = dmean[k] + dm[k];
        for (int k = 0; k < 6; k++) dcov3D[6 * i + k] = dcov[k];
    }
}

// quadrant masks of n splats for the tile whose first pixel is (tx0, ty0)
void hm_quadrant_masks(int n, const float* means2D, const float* conic_opacity, const float* extents, float tx0, float ty0,
                       unsigned* masks) {
    for (int i = 0; i < n; i++) {
        const float4 a = make_float4(means2D[2 * i], means2D[2 * i + 1], extents[2 * i], extents[2 * i + 1]);
        const float4 b = make_float4(conic_opacity[4 * i], conic_opacity[4 * i + 1], conic_opacity[4 * i + 2], conic_opacity[4 * i + 3]);
        masks[i] = sgr_quadrant_mask(a, b, tx0, ty0);
    }
}

// tile masks of n splats over their rects {x0, y0, x1, y1} (tiles), and the two bit utilities the mask needs
void hm_tile_masks(int n, const float* means2D, const float* conic_opacity, const unsigned* rects, unsigned long long* masks) {
    for (int i = 0; i < n; i++)
        masks[i] = sgr_tile_mask(means2D[2 * i], means2D[2 * i + 1], conic_opacity[4 * i], conic_opacity[4 * i + 1],
                                 conic_opacity[4 * i + 2], conic_opacity[4 * i + 3], rects[4 * i], rects[4 * i + 1], rects[4 * i + 2],
                                 rects[4 * i + 3]);
}
void hm_tile_masks_per_tile(int n, const float* means2D, const float* conic_opacity, const unsigned* rects, unsigned long long* masks) {
    for (int i = 0; i < n; i++)
        masks[i] = sgr_tile_mask_per_tile(means2D[2 * i], means2D[2 * i + 1], conic_opacity[4 * i], conic_opacity[4 * i + 1],
                                          conic_opacity[4 * i + 2], conic_opacity[4 * i + 3], rects[4 * i], rects[4 * i + 1],
                                          rects[4 * i + 2], rects[4 * i + 3]);
}
unsigned hm_select_bit(unsigned long long m, unsigned k) { return sgr_select_bit(m, k); }
unsigned hm_row_of(unsigned rect, unsigned tx, unsigned ty, const unsigned* u0, const unsigned long long* tmask, unsigned g) {
    return sgr_row_of(rect, tx, ty, u0, (const uint64_t*)tmask, g);
}

float hm_power2(float qa, float qb, float qc, float dx, float dy) { return sgr_power2(qa, qb, qc, dx, dy); }
// parity mode: the staged power expression against the reference's own, and the shared-reciprocal quotient for a given
// reciprocal seed (the test perturbs the seed by an ulp either way: the device's v_rcp_f32 is only accurate to 1 ulp)
void hm_power_ref_pair(int n, const float* c, const float* d, float* ref, float* staged) {
    for (int i = 0; i < n; i++) {
        ref[i] = sgr_power_ref(c[3 * i], c[3 * i + 1], c[3 * i + 2], d[2 * i], d[2 * i + 1]);
        staged[i] = sgr_power_ref_staged(-0.5f * c[3 * i], -c[3 * i + 1], -0.5f * c[3 * i + 2], d[2 * i], d[2 * i + 1]);
    }
}
void hm_div_by_seed(int n, const float* a, const float* b, const float* seed, float* q) {
    for (int i = 0; i < n; i++) {
        const float y0 = seed[i];
        const float y = fmaf(fmaf(-b[i], y0, 1.0f), y0, y0);
        q[i] = sgr_div_by(a[i], b[i], y);
    }
}
}
