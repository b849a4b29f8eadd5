// sgr_math.h -- per-Gaussian and per-(pixel,Gaussian) maths of the rasterizer, written as
// __host__ __device__ inline functions so the very same code that the gfx950 kernels run can be
// unit-tested on the host (tests/host_math, built by hipcc for x86) against the oracle.
//
// Everything that decides an INTEGER output (radius, tile rect, tiles_touched, depth sort key) is
// evaluated with FP contraction off and in the operation order the reference's GLM code has
// (forward.cu:74-152, auxiliary.h:41-77), with multiplications by literal zeros dropped (exact in
// IEEE arithmetic), so radii / tiles / keys are bit-reproducible against oracle/sgr_oracle.c.
#pragma once
#include "sgr_common.h"
#include <math.h>

#define SGR_HD __host__ __device__ __forceinline__

struct SgrCam {
    float view[16];  // viewmatrix, flat (transposed / row-vector form)
    float proj[16];  // full projection, flat
    float campos[3];
    float tan_fovx, tan_fovy, focal_x, focal_y;
    int W, H;
    int gx, gy;  // tile grid
    float scale_modifier;
};

// camera of the stand-alone filter kernel (K3): device pointers to the caller's matrices + host-side scalars
struct SgrCamArgs {
    const float* view = nullptr;
    const float* proj = nullptr;
    float tan_fovx = 0.f, tan_fovy = 0.f, focal_x = 0.f, focal_y = 0.f;
    int W = 0, H = 0, gx = 0, gy = 0;
    float scale_modifier = 1.f;
};

struct SgrProj {
    float depth;       // view-space z
    float px, py;      // pixel-space mean (ndc2Pix)
    float cov_a, cov_b, cov_c;  // 2D covariance incl. +0.3 low-pass
    float con_x, con_y, con_z;  // conic
    int radius;
    uint32_t rx0, ry0, rx1, ry1;  // tile rect
    bool ok;
};

SGR_HD float sgr_ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

SGR_HD void sgr_get_rect(float px, float py, int max_radius, int gx, int gy, uint32_t& x0, uint32_t& y0, uint32_t& x1,
                         uint32_t& y1) {
#pragma clang fp contract(off)
    // auxiliary.h:46-56 -- float divide then C truncation
    int a = (int)((px - max_radius) / SGR_BLOCK_X);
    int b = (int)((py - max_radius) / SGR_BLOCK_Y);
    int c = (int)((px + max_radius + SGR_BLOCK_X - 1) / SGR_BLOCK_X);
    int d = (int)((py + max_radius + SGR_BLOCK_Y - 1) / SGR_BLOCK_Y);
    a = a < 0 ? 0 : a; b = b < 0 ? 0 : b; c = c < 0 ? 0 : c; d = d < 0 ? 0 : d;
    x0 = (uint32_t)(a > gx ? gx : a);
    y0 = (uint32_t)(b > gy ? gy : b);
    x1 = (uint32_t)(c > gx ? gx : c);
    y1 = (uint32_t)(d > gy ? gy : d);
}

// forward.cu:118-152
SGR_HD void sgr_cov3d(const float* scale, float mod, const float* rot, float* cov3D) {
#pragma clang fp contract(off)
    const float s0 = mod * scale[0], s1 = mod * scale[1], s2 = mod * scale[2];
    const float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    // GLM column c of R, then M[c][row] = s[row] * R[c][row]
    const float M00 = s0 * (1.f - 2.f * (y * y + z * z)), M01 = s1 * (2.f * (x * y - r * z)), M02 = s2 * (2.f * (x * z + r * y));
    const float M10 = s0 * (2.f * (x * y + r * z)), M11 = s1 * (1.f - 2.f * (x * x + z * z)), M12 = s2 * (2.f * (y * z - r * x));
    const float M20 = s0 * (2.f * (x * z - r * y)), M21 = s1 * (2.f * (y * z + r * x)), M22 = s2 * (1.f - 2.f * (x * x + y * y));
    // Sigma[c][row] = M[row][0]*M[c][0] + M[row][1]*M[c][1] + M[row][2]*M[c][2]
    cov3D[0] = M00 * M00 + M01 * M01 + M02 * M02;  // Sigma[0][0]
    cov3D[1] = M10 * M00 + M11 * M01 + M12 * M02;  // Sigma[0][1]
    cov3D[2] = M20 * M00 + M21 * M01 + M22 * M02;  // Sigma[0][2]
    cov3D[3] = M10 * M10 + M11 * M11 + M12 * M12;  // Sigma[1][1]
    cov3D[4] = M20 * M10 + M21 * M11 + M22 * M12;  // Sigma[1][2]
    cov3D[5] = M20 * M20 + M21 * M21 + M22 * M22;  // Sigma[2][2]
}

// forward.cu:155-256 geometry part (everything except SH) for one Gaussian.
// Returns ok=false where the reference returns early (radius stays 0).
SGR_HD SgrProj sgr_project(const float* p, const float* cov3D, const SgrCam& cam) {
#pragma clang fp contract(off)
    SgrProj o;
    o.ok = false;
    o.radius = 0;
    const float* v = cam.view;
    const float* m = cam.proj;
    // auxiliary.h:58-66 / 139-164
    float tx = v[0] * p[0] + v[4] * p[1] + v[8] * p[2] + v[12];
    float ty = v[1] * p[0] + v[5] * p[1] + v[9] * p[2] + v[13];
    const float tz = v[2] * p[0] + v[6] * p[1] + v[10] * p[2] + v[14];
    o.depth = tz;
    if (tz <= 0.2f) return o;
    // auxiliary.h:68-77
    const float hx = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    const float hy = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    const float hw = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
    const float p_w = 1.0f / (hw + 0.0000001f);
    const float projx = hx * p_w, projy = hy * p_w;
    // forward.cu:74-113
    const float limx = 1.3f * cam.tan_fovx;
    const float limy = 1.3f * cam.tan_fovy;
    const float txtz = tx / tz;
    const float tytz = ty / tz;
    tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
    ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
    const float J00 = cam.focal_x / tz, J02 = -(cam.focal_x * tx) / (tz * tz);
    const float J11 = cam.focal_y / tz, J12 = -(cam.focal_y * ty) / (tz * tz);
    const float T00 = v[0] * J00 + v[2] * J02, T01 = v[4] * J00 + v[6] * J02, T02 = v[8] * J00 + v[10] * J02;
    const float T10 = v[1] * J11 + v[2] * J12, T11 = v[5] * J11 + v[6] * J12, T12 = v[9] * J11 + v[10] * J12;
    const float c0 = cov3D[0], c1 = cov3D[1], c2 = cov3D[2], c3 = cov3D[3], c4 = cov3D[4], c5 = cov3D[5];
    const float X00 = T00 * c0 + T01 * c1 + T02 * c2, X10 = T00 * c1 + T01 * c3 + T02 * c4, X20 = T00 * c2 + T01 * c4 + T02 * c5;
    const float X01 = T10 * c0 + T11 * c1 + T12 * c2, X11 = T10 * c1 + T11 * c3 + T12 * c4, X21 = T10 * c2 + T11 * c4 + T12 * c5;
    float a = X00 * T00 + X10 * T01 + X20 * T02;
    const float b = X01 * T00 + X11 * T01 + X21 * T02;
    float c = X01 * T10 + X11 * T11 + X21 * T12;
    a += 0.3f;
    c += 0.3f;
    o.cov_a = a; o.cov_b = b; o.cov_c = c;
    const float det = (a * c - b * b);
    if (det == 0.0f) return o;
    const float det_inv = 1.f / det;
    o.con_x = c * det_inv;
    o.con_y = -b * det_inv;
    o.con_z = a * det_inv;
    const float mid = 0.5f * (a + c);
    const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
    o.px = sgr_ndc2pix(projx, cam.W);
    o.py = sgr_ndc2pix(projy, cam.H);
    sgr_get_rect(o.px, o.py, (int)my_radius, cam.gx, cam.gy, o.rx0, o.ry0, o.rx1, o.ry1);
    if ((o.rx1 - o.rx0) * (o.ry1 - o.ry0) == 0) return o;
    o.radius = (int)my_radius;
    o.ok = true;
    return o;
}

// SH constants, auxiliary.h:22-39
#define SGR_SH_C0 0.28209479177387814f
#define SGR_SH_C1 0.4886025119029199f
#define SGR_SH_C2_0 1.0925484305920792f
#define SGR_SH_C2_1 -1.0925484305920792f
#define SGR_SH_C2_2 0.31539156525252005f
#define SGR_SH_C2_3 -1.0925484305920792f
#define SGR_SH_C2_4 0.5462742152960396f
#define SGR_SH_C3_0 -0.5900435899266435f
#define SGR_SH_C3_1 2.890611442640554f
#define SGR_SH_C3_2 -0.4570457994644658f
#define SGR_SH_C3_3 0.3731763325901154f
#define SGR_SH_C3_4 -0.4570457994644658f
#define SGR_SH_C3_5 1.445305721320277f
#define SGR_SH_C3_6 -0.5900435899266435f

// SH basis Y_k(dir) for k < (deg+1)^2, forward.cu:30-59 (basis[0] = C0 etc.; signs folded in)
SGR_HD void sgr_sh_basis(int deg, float x, float y, float z, float* Y) {
    Y[0] = SGR_SH_C0;
    if (deg > 0) {
        Y[1] = -SGR_SH_C1 * y;
        Y[2] = SGR_SH_C1 * z;
        Y[3] = -SGR_SH_C1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            Y[4] = SGR_SH_C2_0 * xy;
            Y[5] = SGR_SH_C2_1 * yz;
            Y[6] = SGR_SH_C2_2 * (2.0f * zz - xx - yy);
            Y[7] = SGR_SH_C2_3 * xz;
            Y[8] = SGR_SH_C2_4 * (xx - yy);
            if (deg > 2) {
                Y[9] = SGR_SH_C3_0 * y * (3.0f * xx - yy);
                Y[10] = SGR_SH_C3_1 * xy * z;
                Y[11] = SGR_SH_C3_2 * y * (4.0f * zz - xx - yy);
                Y[12] = SGR_SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                Y[13] = SGR_SH_C3_4 * x * (4.0f * zz - xx - yy);
                Y[14] = SGR_SH_C3_5 * z * (xx - yy);
                Y[15] = SGR_SH_C3_6 * x * (xx - 3.0f * yy);
            }
        }
    }
}

// Conservative half extents (in pixels) of the region where a Gaussian can reach
// alpha >= 1/255, i.e. power >= -ln(255*opacity) (forward.cu:428-430).  The region is the ellipse
// d^T conic d <= 2*tau whose axis-aligned bounding box has half widths sqrt(2*tau*cov_xx),
// sqrt(2*tau*cov_yy) (conic = cov^-1).  Inflated (tau*1.02+0.05, then +1% +0.25 px) so that fp32
// rounding of `power` in the blend kernels can never accept a pair outside the box; a negative
// extent means "can never contribute".  NaN opacity yields NaN extents = "never culled".
// The blend kernels evaluate alpha from the CONIC (cov / det with det an fp32 difference a*c - b*b): for huge,
// nearly degenerate splats cancellation in det can move the conic by more than the slack above, so the box is also
// derived from the conic itself (half width sqrt(2*tau*conic.z / (conic.x*conic.z - conic.y^2))) and the larger of
// the two is kept; a conic that is not positive definite in fp32 gives an infinite box (never culled).
SGR_HD void sgr_extent(float opacity, float cov_a, float cov_c, float con_x, float con_y, float con_z, float& hx,
                       float& hy) {
    if (opacity < 0.0039f) {  // < 1/255 (with slack): alpha = min(.99, o*G) <= o can never pass
        hx = -1.0f;
        hy = -1.0f;
        return;
    }
    float tau = logf(255.0f * opacity);
    tau = fmaxf(tau, 0.0f) * 1.02f + 0.05f;
    const float den = con_x * con_z - con_y * con_y;
    float ex = cov_a, ey = cov_c;
    if (den > 0.0f && con_x > 0.0f && con_z > 0.0f) {
        ex = fmaxf(ex, con_z / den);
        ey = fmaxf(ey, con_x / den);
    } else {
        ex = ey = INFINITY;
    }
    hx = sqrtf(2.0f * tau * ex) * 1.01f + 0.25f;
    hy = sqrtf(2.0f * tau * ey) * 1.01f + 0.25f;
}

#ifndef SGR_CULL_FAST
#define SGR_CULL_FAST 1  // 0: the round-2 form (a division per edge, library logf) -- A/B only (tools/build_variant.py)
#endif
// Exact part of the quadrant cull.  tau2 = 2*tau' (same inflation as sgr_extent; negative = never visible).
SGR_HD float sgr_tau2(float opacity) {
    if (opacity < 0.0039f) return -1.0f;
#if defined(__HIP_DEVICE_COMPILE__) && SGR_CULL_FAST
    // v_log_f32 (log2, ~1 ulp) instead of the library's logf (~25 instructions per staged instance): the 2 % + 0.05
    // inflation below is five orders of magnitude above the difference
    const float lg = __builtin_amdgcn_logf(255.0f * opacity) * 0.6931471805599453f;
#else
    const float lg = logf(255.0f * opacity);
#endif
    const float tau = fmaxf(lg, 0.0f) * 1.02f + 0.05f;
    return 2.0f * tau;
}
// Minimum of Q(d) = A*dx^2 + 2*B*dx*dy + C*dy^2 over the rectangle dx in [dx0,dx1], dy in [dy0,dy1]
// (offsets of a pixel block from the splat centre).  Q is convex: 0 if the centre is inside, else the minimum
// lies on one of the four edges, where Q is a 1-D parabola whose vertex is clamped to the edge.
// nBiC = -B / C and nBiA = -B / A are formed ONCE per splat by the caller (the edge vertices are -B*d/C: written as a
// division per edge the compiler emitted 16 IEEE divisions -- ~190 instructions -- per staged instance, 13 % of the
// forward kernel's VALU work); a vertex that is off by an ulp moves Q by a second-order amount, far inside the 0.25 px +
// 2 % margins of the caller.
SGR_HD float sgr_min_quadform_rect(float A, float B, float C, float nBiA, float nBiC, float dx0, float dx1, float dy0,
                                   float dy1) {
    if (dx0 <= 0.0f && dx1 >= 0.0f && dy0 <= 0.0f && dy1 >= 0.0f) return 0.0f;
    float m;
    {
        const float dy = fminf(fmaxf(SGR_CULL_FAST ? nBiC * dx0 : -B * dx0 / C, dy0), dy1);
        m = A * dx0 * dx0 + 2.0f * B * dx0 * dy + C * dy * dy;
    }
    {
        const float dy = fminf(fmaxf(SGR_CULL_FAST ? nBiC * dx1 : -B * dx1 / C, dy0), dy1);
        m = fminf(m, A * dx1 * dx1 + 2.0f * B * dx1 * dy + C * dy * dy);
    }
    {
        const float dx = fminf(fmaxf(SGR_CULL_FAST ? nBiA * dy0 : -B * dy0 / A, dx0), dx1);
        m = fminf(m, A * dx * dx + 2.0f * B * dx * dy0 + C * dy0 * dy0);
    }
    {
        const float dx = fminf(fmaxf(SGR_CULL_FAST ? nBiA * dy1 : -B * dy1 / A, dx0), dx1);
        m = fminf(m, A * dx * dx + 2.0f * B * dx * dy1 + C * dy1 * dy1);
    }
    return m;
}
// 4-bit mask of the 8x8 quadrants of tile (tx0,ty0) a splat can touch: conservative box first (rec[0].zw), then
// the exact ellipse-vs-rectangle test with a 0.25 px margin.  Comparisons are written so NaN never culls.
SGR_HD uint32_t sgr_quadrant_mask(const float4& a, const float4& b, float tx0, float ty0) {
    uint32_t mask4 = 0;
    const float tau2 = sgr_tau2(b.w);
    const float nBiA = -b.y / b.x, nBiC = -b.y / b.z;  // two divisions per splat instead of sixteen (SGR_CULL_FAST)
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float qx0 = tx0 + (float)((q & 1) * 8), qy0 = ty0 + (float)((q >> 1) * 8);
        bool miss = (a.x + a.z < qx0) || (a.x - a.z > qx0 + 7.0f) || (a.y + a.w < qy0) || (a.y - a.w > qy0 + 7.0f);
        if (!miss) {
            const float mq = sgr_min_quadform_rect(b.x, b.y, b.z, nBiA, nBiC, qx0 - 0.25f - a.x, qx0 + 7.25f - a.x,
                                                   qy0 - 0.25f - a.y, qy0 + 7.25f - a.y);
            miss = mq > tau2;
        }
        mask4 |= miss ? 0u : (1u << q);
    }
    return mask4;
}

// ---- tile masks (round 4) ----------------------------------------------------------------------
// A Gaussian whose (cut-down) tile rect has 2..64 tiles carries a 64-bit mask of the tiles in which it can reach
// alpha >= 1/255 at all: bit j = tile (x0 + j % w, y0 + j / w).  The test per tile is the one the blend kernels apply per
// 8x8 quadrant (sgr_quadrant_mask: minimum of the conic's quadratic form over the pixel box, inflated by 0.25 px, against
// 2 tau', tau' = 1.02 ln(255 o) + 0.05) on the 16x16 box of the tile -- a box that contains its four quadrants, so a tile
// is only dropped when every quadrant of it would be.  NaN never drops a tile.
SGR_HD uint64_t sgr_tile_mask_per_tile(float px, float py, float con_x, float con_y, float con_z, float opacity, uint32_t x0,
                                       uint32_t y0, uint32_t x1, uint32_t y1) {  // the definition: one box test per tile
    const float tau2 = sgr_tau2(opacity);
    const float nBiA = -con_y / con_x, nBiC = -con_y / con_z;
    uint64_t tm = 0;
    uint32_t j = 0;
    for (uint32_t yy = y0; yy < y1; yy++) {
        const float dy0 = (float)(yy * 16u) - 0.25f - py, dy1 = (float)(yy * 16u) + 15.25f - py;
        for (uint32_t xx = x0; xx < x1; xx++, j++) {
            const float dx0 = (float)(xx * 16u) - 0.25f - px, dx1 = (float)(xx * 16u) + 15.25f - px;
            const float mq = sgr_min_quadform_rect(con_x, con_y, con_z, nBiA, nBiC, dx0, dx1, dy0, dy1);
            if (!(mq > tau2)) tm |= 1ull << j;
        }
    }
    return tm;
}
// What the preprocess runs: the same set, one tile ROW at a time.  Inside the band dy in [dy0, dy1] of a tile row the region
// Q(dx, dy) = A dx^2 + 2 B dx dy + C dy^2 <= t2 spans dx in [g(dy), f(dy)] / A with f, g = -B dy +- sqrt(A t2 - det dy^2):
// f is concave with its maximum at dy = -B s, g convex with its minimum at dy = +B s, s = sqrt(t2 / (det C)); both clamped to
// the band (and to |dy| <= sqrt(A t2 / det), outside of which the row is empty) give the row's exact x extent; the tiles
// that overlap it (0.26 px added on either side: the pixel box's 0.25 + rounding) get their bits.  O(rows) instead of O(tiles):
// the per-tile loop cost the preprocess 0.065 ms at 5 M Gaussians.  A conic that is not positive definite (or NaN) keeps
// every tile.
SGR_HD uint64_t sgr_tile_mask(float px, float py, float A, float B, float C, float opacity, uint32_t x0, uint32_t y0,
                              uint32_t x1, uint32_t y1) {
    const uint32_t w = x1 - x0, h = y1 - y0, n = w * h;
    const uint64_t full = n >= 64u ? ~0ull : ((1ull << n) - 1ull);
    const float t2 = sgr_tau2(opacity);
    if (t2 < 0.0f) return 0ull;
    const float det = A * C - B * B;
    if (!(det > 0.0f && A > 0.0f && C > 0.0f)) return full;
    const float ymax = sqrtf(A * t2 / det);
    const float s = sqrtf(t2 / (det * C));
    // (an overflowing s or ymax -- det cancelled away for a needle clipped by the screen -- would put NaN into the row extents
    // below, where fmin / fmax silently pick the other operand: keep every tile instead)
    if (!(s <= 3.0e38f && ymax <= 3.0e38f)) return full;
    const float inva = 1.0f / A;
    const float dyf = (B == 0.0f) ? 0.0f : -B * s, dyg = (B == 0.0f) ? 0.0f : B * s;  // where f peaks / g bottoms out
    uint64_t tm = 0;
    for (uint32_t rr = 0; rr < h; rr++) {
        const float dy0 = (float)((y0 + rr) * 16u) - 0.25f - py, dy1 = dy0 + 15.5f;
        const float lo = fmaxf(dy0, -ymax), hi = fminf(dy1, ymax);
        if (!(lo <= hi)) continue;  // the ellipse does not reach this row (NaN: det / t2 are finite and positive here)
        const float yf = fminf(fmaxf(dyf, lo), hi), yg = fminf(fmaxf(dyg, lo), hi);
        const float xr = (-B * yf + sqrtf(fmaxf(A * t2 - det * yf * yf, 0.0f))) * inva;
        const float xl = (-B * yg - sqrtf(fmaxf(A * t2 - det * yg * yg, 0.0f))) * inva;
        // tiles whose pixel columns [16 t, 16 t + 15] meet [px + xl - 0.26, px + xr + 0.26]
        const float ta = fmaxf(floorf((px + xl - 0.26f) * 0.0625f), (float)x0);
        const float tb = fminf(floorf((px + xr + 0.26f) * 0.0625f), (float)(x1 - 1u));
        if (!(ta <= tb)) continue;
        const uint32_t ca = (uint32_t)ta - x0, cnt = (uint32_t)tb - (uint32_t)ta + 1u;
        const uint64_t run = cnt >= 64u ? ~0ull : ((1ull << cnt) - 1ull);
        tm |= run << (rr * w + ca);
    }
    return tm & full;
}
// position of the k-th set bit of m (k < popcount(m)): the half first, then five halvings of a 32-bit word
SGR_HD uint32_t sgr_select_bit(uint64_t m, uint32_t k) {
    uint32_t word = (uint32_t)m, pos = 0;
    const uint32_t c0 = (uint32_t)__builtin_popcount(word);
    if (k >= c0) { k -= c0; word = (uint32_t)(m >> 32); pos = 32; }
#pragma unroll
    for (int sh = 16; sh >= 1; sh >>= 1) {
        const uint32_t c = (uint32_t)__builtin_popcount(word & ((1u << sh) - 1u));
        if (k >= c) { k -= c; word >>= sh; pos += (uint32_t)sh; }
    }
    return pos;
}
// Partial-gradient row of instance (Gaussian g, tile (tx, ty)): the Gaussian's first row + the rank of the tile among the
// tiles it is emitted for (its index inside the rect when there is no mask).
SGR_HD uint32_t sgr_row_of(uint32_t rect, uint32_t tx, uint32_t ty, const uint32_t* __restrict__ u0,
                           const uint64_t* __restrict__ tmask, uint32_t g) {
    const uint32_t rx0 = rect & 1023u, ry0 = (rect >> 10) & 1023u, rw = (rect >> 20) & 1023u;
    uint32_t j = (ty - ry0) * rw + (tx - rx0);
    if (rect & 0x80000000u) j = (uint32_t)__builtin_popcountll(tmask[g] & ((1ull << j) - 1ull));
    return u0[g] + j;
}

// ---------------------------------------------------------------------------------------------
// Per-(pixel, Gaussian) evaluation shared by the forward and backward blend kernels.  The conic is
// pre-scaled when a tile list is staged into LDS:  qa = -0.5*log2e*conic.x, qb = -log2e*conic.y,
// qc = -0.5*log2e*conic.z, so that  power*log2e = qa*dx*dx + qb*dx*dy + qc*dy*dy  and
// G = exp(power) = exp2(power*log2e) is a single v_exp_f32.  Explicit fmaf + contraction off: the
// forward and backward kernels must evaluate bit-identical alpha for the same pair
// (forward.cu:423-430 vs backward.cu:536-545).
// The pre-scaled conic of a record {conic.x, conic.y, conic.z, opacity}: ONE definition, because the forward (which
// stages it), the LDS-staged backward (which stages it again) and the preprocess (which files it in rec[3].xzw for the
// scalar-walk backward) must produce the same bits.
SGR_HD float4 sgr_stage_conic(const float4 b) {
    return make_float4(-0.5f * SGR_LOG2E * b.x, -SGR_LOG2E * b.y, -0.5f * SGR_LOG2E * b.z, b.w);
}
SGR_HD float sgr_power2(float qa, float qb, float qc, float dx, float dy) {
#pragma clang fp contract(off)
    const float u = fmaf(qb, dy, qa * dx);  // qa*dx + qb*dy
    return fmaf(qc * dy, dy, u * dx);
}

// "Exact" (parity) mode of the blend kernels: the reference's own expression, operation by operation and without
// contraction (forward.cu:420, backward.cu:536), followed by the library's accurate expf -- the arithmetic of oracle/_ref's
// strict build and of the C oracle, so that alpha_out (and with it T_final = 1 - alpha_out, the amplifier of every
// end-to-end gradient difference) is reproduced bit for bit.  ~13 VALU instructions more per pixel
// and visit than sgr_power2 + v_exp_f32: opt-in (sgr_test_switches bit 7 / SGR_EXACT=1), not the default.
SGR_HD float sgr_power_ref(float cx, float cy, float cz, float dx, float dy) {
#pragma clang fp contract(off)
    return -0.5f * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
}
// The same value from a conic staged as hx = -0.5*cx, ny = -cy, hz = -0.5*cz: scaling by -0.5 and negation commute with
// every rounding of the expression above (powers of two; the operands are far from the subnormal range), so
//   RN(RN(hx*dx)*dx) = -0.5*RN(RN(cx*dx)*dx),  RN(A' + B') = -0.5*RN(A + B),  X - RN(RN(cy*dx)*dy) = X + RN(RN(ny*dx)*dy)
// bit for bit -- 8 instead of 9 VALU instructions, and the staged triple is the one the backward's gradient terms use.
// tests/test_host_math.py holds the identity against sgr_power_ref.
SGR_HD float sgr_power_ref_staged(float hx, float ny, float hz, float dx, float dy) {
#pragma clang fp contract(off)
    return (hx * dx * dx + hz * dy * dy) + ny * dx * dy;
}

// ---- parity-mode elementary functions ------------------------------------------------------------------------------
// expf as the device library evaluates it on gfx950 (ocml expF / LLVM's f32 exp lowering, read off the ISA hipcc emits
// for `expf`: ph = x*c, pl = fma(x, cc, fma(x, c, -ph)), e = rint(ph), v_exp_f32(ph - e + pl), ldexp by e), operation by
// operation, for the arguments the blend kernels USE G for -- power <= 0 and alpha = opacity * G >= 1/255, i.e. x >= -5.6 --
// with three shortcuts that change no bit there (round 6; round 4 had dropped the library's two range selects):
//   * e = rint(ph) as (ph + 1.5 * 2^23) - 1.5 * 2^23: two full-rate adds instead of v_rndne_f32 (4.2 cycles per wave
//     instruction on gfx950, profiles/r5/valu_rates2.jsonl); exact for |ph| < 2^22;
//   * the scaling by 2^e as an integer add into the exponent field -- bits(v) + (bits(ph + 1.5 * 2^23) << 23), the low bits of
//     that sum ARE e in two's complement -- one v_lshl_add_u32 instead of v_cvt_i32_f32 + v_ldexp_f32 (4 + 4.2 cycles);
//     exact whenever the result is a normal number, which the clamp below guarantees (v in [0.7, 1.42], e >= -124);
//   * the ARGUMENT clamped at -86 (one v_max, where round 4 clamped ph at -200): exp(-86) = 4.5e-38 is as much "alpha <
//     1/255" as the 0 expf returns further down, and with it every intermediate stays in the range the two shortcuts above
//     are exact in (clamping ph instead leaves `a` unbounded below, and an exponent-field add on a tiny v can land on a NaN
//     pattern -- a NaN alpha PASSES the kernels' !(alpha < thr) test).
// Bit-identical to expf for every -86 <= x <= 87 (sgr_test_exact_math, tests/test_gpu_primitives.py); expf(-86) below; NaN
// -> expf(-86) as well (the reference blends a NaN power with alpha 0.99: fminf(0.99f, NaN)); x > 88 is not meaningful (the
// kernels never use G for a positive power).  10 -> 8 instructions, 34 -> 28 issue cycles.
SGR_HD float sgr_expf_ref(float x) {
#pragma clang fp contract(off)
    const float c = 0x1.715476p+0f, cc = 0x1.4ae0bep-26f;  // c + cc = 49 bits of log2(e)
    const float xc = fmaxf(x, -86.0f);
    const float ph = xc * c;
    const float pl = fmaf(xc, cc, fmaf(xc, c, -ph));
    const float t = ph + 12582912.0f;  // 1.5 * 2^23: the sum's low mantissa bits hold rint(ph)
    const float e = t - 12582912.0f;
    const float a = (ph - e) + pl;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(__float_as_uint(__builtin_amdgcn_exp2f(a)) + (__float_as_uint(t) << 23));
#else
    return ldexpf(exp2f(a), (int)e);
#endif
}

// a / b and a2 / b as hipcc's IEEE-754 expansion of `/` computes them (v_div_scale, v_rcp_f32, a Newton step on the
// reciprocal, two residual corrections of the quotient, v_div_fmas, v_div_fixup: 11 instructions per quotient), for
// operands that need neither scaling nor fix-up -- every (T, 1 - alpha) and (T_final, 1 - alpha) of the blend backward:
// 0 <= a <= 1, 0.01 <= b <= 1 -- where v_div_scale / v_div_fmas / v_div_fixup are the identity.  The refined reciprocal
// is shared between the two quotients: 3 + 5 + 5 instead of 22 instructions, same bits (sgr_test_exact_math sweeps the
// operand ranges against `/` on the GPU; tests/test_host_math.py shows that the result does not depend on the last
// bit of the v_rcp_f32 seed).
struct SgrRcp { float b, y; };
SGR_HD SgrRcp sgr_rcp_refined(float b) {
#pragma clang fp contract(off)
#if defined(__HIP_DEVICE_COMPILE__)
    const float y0 = __builtin_amdgcn_rcpf(b);
#else
    const float y0 = 1.0f / b;
#endif
    return SgrRcp{b, fmaf(fmaf(-b, y0, 1.0f), y0, y0)};
}
SGR_HD float sgr_div_by(float a, float b, float y) {  // y = refined reciprocal of b
#pragma clang fp contract(off)
    float q = a * y;
    q = fmaf(fmaf(-b, q, a), y, q);
    return fmaf(fmaf(-b, q, a), y, q);
}

// ---------------------------------------------------------------------------------------------
// Per-Gaussian backward maths (float tolerance only, so contraction is left to the compiler).

// backward.cu:144-274 (computeCov2DCUDA): dL/dconic -> dL/dcov3D[6] and the covariance part of
// dL/dmean (ASSIGNED).  T/Vrk/W index pairs below are the reference's glm [col][row] pairs.
SGR_HD void sgr_cov2d_backward(const float* p, const float* cov3D, const SgrCam& cam, float dcx, float dcy, float dcw,
                               float* dL_dcov, float* dL_dmean) {
    const float* v = cam.view;
    float tx = v[0] * p[0] + v[4] * p[1] + v[8] * p[2] + v[12];
    float ty = v[1] * p[0] + v[5] * p[1] + v[9] * p[2] + v[13];
    const float tz = v[2] * p[0] + v[6] * p[1] + v[10] * p[2] + v[14];
    const float limx = 1.3f * cam.tan_fovx, limy = 1.3f * cam.tan_fovy;
    const float txtz = tx / tz, tytz = ty / tz;
    tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
    ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
    const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    const float h_x = cam.focal_x, h_y = cam.focal_y;
    const float J00 = h_x / tz, J02 = -(h_x * tx) / (tz * tz), J11 = h_y / tz, J12 = -(h_y * ty) / (tz * tz);
    const float T00 = v[0] * J00 + v[2] * J02, T01 = v[4] * J00 + v[6] * J02, T02 = v[8] * J00 + v[10] * J02;
    const float T10 = v[1] * J11 + v[2] * J12, T11 = v[5] * J11 + v[6] * J12, T12 = v[9] * J11 + v[10] * J12;
    const float V00 = cov3D[0], V01 = cov3D[1], V02 = cov3D[2], V11 = cov3D[3], V12 = cov3D[4], V22 = cov3D[5];
    // rows of T*Vrk (reused by cov2D and by dL/dT)
    const float A0x = T00 * V00 + T01 * V01 + T02 * V02, A0y = T00 * V01 + T01 * V11 + T02 * V12, A0z = T00 * V02 + T01 * V12 + T02 * V22;
    const float A1x = T10 * V00 + T11 * V01 + T12 * V02, A1y = T10 * V01 + T11 * V11 + T12 * V12, A1z = T10 * V02 + T11 * V12 + T12 * V22;
    const float a = A0x * T00 + A0y * T01 + A0z * T02 + 0.3f;
    const float b = A1x * T00 + A1y * T01 + A1z * T02;
    const float c = A1x * T10 + A1y * T11 + A1z * T12 + 0.3f;
    const float denom = a * c - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    if (denom2inv != 0) {
        dL_da = denom2inv * (-c * c * dcx + 2 * b * c * dcy + (denom - a * c) * dcw);
        dL_dc = denom2inv * (-a * a * dcw + 2 * a * b * dcy + (denom - a * c) * dcx);
        dL_db = denom2inv * 2 * (b * c * dcx - (denom + 2 * b * b) * dcy + a * b * dcw);
        dL_dcov[0] = (T00 * T00 * dL_da + T00 * T10 * dL_db + T10 * T10 * dL_dc);
        dL_dcov[3] = (T01 * T01 * dL_da + T01 * T11 * dL_db + T11 * T11 * dL_dc);
        dL_dcov[5] = (T02 * T02 * dL_da + T02 * T12 * dL_db + T12 * T12 * dL_dc);
        dL_dcov[1] = 2 * T00 * T01 * dL_da + (T00 * T11 + T01 * T10) * dL_db + 2 * T10 * T11 * dL_dc;
        dL_dcov[2] = 2 * T00 * T02 * dL_da + (T00 * T12 + T02 * T10) * dL_db + 2 * T10 * T12 * dL_dc;
        dL_dcov[4] = 2 * T02 * T01 * dL_da + (T01 * T12 + T02 * T11) * dL_db + 2 * T11 * T12 * dL_dc;
    } else {
        for (int i = 0; i < 6; i++) dL_dcov[i] = 0;
    }
    const float dL_dT00 = 2 * A0x * dL_da + A1x * dL_db;
    const float dL_dT01 = 2 * A0y * dL_da + A1y * dL_db;
    const float dL_dT02 = 2 * A0z * dL_da + A1z * dL_db;
    const float dL_dT10 = 2 * A1x * dL_dc + A0x * dL_db;
    const float dL_dT11 = 2 * A1y * dL_dc + A0y * dL_db;
    const float dL_dT12 = 2 * A1z * dL_dc + A0z * dL_db;
    const float dL_dJ00 = v[0] * dL_dT00 + v[4] * dL_dT01 + v[8] * dL_dT02;
    const float dL_dJ02 = v[2] * dL_dT00 + v[6] * dL_dT01 + v[10] * dL_dT02;
    const float dL_dJ11 = v[1] * dL_dT10 + v[5] * dL_dT11 + v[9] * dL_dT12;
    const float dL_dJ12 = v[2] * dL_dT10 + v[6] * dL_dT11 + v[10] * dL_dT12;
    const float itz = 1.f / tz, itz2 = itz * itz, itz3 = itz2 * itz;
    const float dL_dtx = x_grad_mul * -h_x * itz2 * dL_dJ02;
    const float dL_dty = y_grad_mul * -h_y * itz2 * dL_dJ12;
    const float dL_dtz = -h_x * itz2 * dL_dJ00 - h_y * itz2 * dL_dJ11 + (2 * h_x * tx) * itz3 * dL_dJ02 +
                         (2 * h_y * ty) * itz3 * dL_dJ12;
    // transformVec4x3Transpose (auxiliary.h:89-97)
    dL_dmean[0] = v[0] * dL_dtx + v[1] * dL_dty + v[2] * dL_dtz;
    dL_dmean[1] = v[4] * dL_dtx + v[5] * dL_dty + v[6] * dL_dtz;
    dL_dmean[2] = v[8] * dL_dtx + v[9] * dL_dty + v[10] * dL_dtz;
}

// backward.cu:371-403: projection and depth paths, ADDED to dL_dmean
SGR_HD void sgr_proj_depth_backward(const float* m, const SgrCam& cam, float g2x, float g2y, float ddepth,
                                    float* dL_dmean) {
    const float* proj = cam.proj;
    const float* view = cam.view;
    const float hw = proj[3] * m[0] + proj[7] * m[1] + proj[11] * m[2] + proj[15];
    const float m_w = 1.0f / (hw + 0.0000001f);
    const float mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
    const float mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
    dL_dmean[0] += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
    dL_dmean[1] += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
    dL_dmean[2] += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
    const float mul3 = view[2] * m[0] + view[6] * m[1] + view[10] * m[2] + view[14];
    dL_dmean[0] += (view[2] - view[3] * mul3) * ddepth;
    dL_dmean[1] += (view[6] - view[7] * mul3) * ddepth;
    dL_dmean[2] += (view[10] - view[11] * mul3) * ddepth;
}

// backward.cu:20-139: given normalised dir (x,y,z) and the SH row already contracted with the (clamp-masked) dL/dRGB,
// t[k] = sum_ch sh[3k + ch] * dL/dRGB[ch], returns dL/ddir = sum_k dY_k/d{x,y,z} * t[k]; the caller applies dnormvdv.
// The reference sums dRGB/d{x,y,z} per channel first and contracts with dL/dRGB last; contracting first is the same sum
// reassociated and needs 16 live values per Gaussian instead of 48 (the per-Gaussian backward's register budget).
SGR_HD void sgr_sh_dir_backward(int deg, float x, float y, float z, const float* t, float* dL_ddir) {
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (deg > 0) {
        dx = -SGR_SH_C1 * t[3];
        dy = -SGR_SH_C1 * t[1];
        dz = SGR_SH_C1 * t[2];
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            dx += SGR_SH_C2_0 * y * t[4] + SGR_SH_C2_2 * 2.f * -x * t[6] + SGR_SH_C2_3 * z * t[7] + SGR_SH_C2_4 * 2.f * x * t[8];
            dy += SGR_SH_C2_0 * x * t[4] + SGR_SH_C2_1 * z * t[5] + SGR_SH_C2_2 * 2.f * -y * t[6] + SGR_SH_C2_4 * 2.f * -y * t[8];
            dz += SGR_SH_C2_1 * y * t[5] + SGR_SH_C2_2 * 2.f * 2.f * z * t[6] + SGR_SH_C2_3 * x * t[7];
            if (deg > 2) {
                dx += (SGR_SH_C3_0 * t[9] * 3.f * 2.f * xy + SGR_SH_C3_1 * t[10] * yz + SGR_SH_C3_2 * t[11] * -2.f * xy +
                       SGR_SH_C3_3 * t[12] * -3.f * 2.f * xz + SGR_SH_C3_4 * t[13] * (-3.f * xx + 4.f * zz - yy) +
                       SGR_SH_C3_5 * t[14] * 2.f * xz + SGR_SH_C3_6 * t[15] * 3.f * (xx - yy));
                dy += (SGR_SH_C3_0 * t[9] * 3.f * (xx - yy) + SGR_SH_C3_1 * t[10] * xz +
                       SGR_SH_C3_2 * t[11] * (-3.f * yy + 4.f * zz - xx) + SGR_SH_C3_3 * t[12] * -3.f * 2.f * yz +
                       SGR_SH_C3_4 * t[13] * -2.f * xy + SGR_SH_C3_5 * t[14] * -2.f * yz + SGR_SH_C3_6 * t[15] * -3.f * 2.f * xy);
                dz += (SGR_SH_C3_1 * t[10] * xy + SGR_SH_C3_2 * t[11] * 4.f * 2.f * yz +
                       SGR_SH_C3_3 * t[12] * 3.f * (2.f * zz - xx - yy) + SGR_SH_C3_4 * t[13] * 4.f * 2.f * xz +
                       SGR_SH_C3_5 * t[14] * (xx - yy));
            }
        }
    }
    dL_ddir[0] = dx;
    dL_ddir[1] = dy;
    dL_ddir[2] = dz;
}

// auxiliary.h:107-117
SGR_HD void sgr_dnormvdv(const float* v, const float* dv, float* out) {
    const float sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    out[0] = ((+sum2 - v[0] * v[0]) * dv[0] - v[1] * v[0] * dv[1] - v[2] * v[0] * dv[2]) * invsum32;
    out[1] = (-v[0] * v[1] * dv[0] + (sum2 - v[1] * v[1]) * dv[1] - v[2] * v[1] * dv[2]) * invsum32;
    out[2] = (-v[0] * v[2] * dv[0] - v[1] * v[2] * dv[1] + (sum2 - v[2] * v[2]) * dv[2]) * invsum32;
}

// backward.cu:278-341 (computeCov3D backward).  R/M/dM are indexed [col][row] like the glm code.
SGR_HD void sgr_cov3d_backward(const float* scale, float mod, const float* rot, const float* dL_dcov3D, float* dL_dscale,
                               float* dL_drot) {
    const float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    const float R[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                           {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                           {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
    const float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    float M2[3][3];  // 2*M, M[c][row] = s[row]*R[c][row]
    for (int c = 0; c < 3; c++)
        for (int k = 0; k < 3; k++) M2[c][k] = 2.0f * (s[k] * R[c][k]);
    const float dS[3][3] = {{dL_dcov3D[0], 0.5f * dL_dcov3D[1], 0.5f * dL_dcov3D[2]},
                            {0.5f * dL_dcov3D[1], dL_dcov3D[3], 0.5f * dL_dcov3D[4]},
                            {0.5f * dL_dcov3D[2], 0.5f * dL_dcov3D[4], dL_dcov3D[5]}};
    float dM[3][3];  // dM[c][row] = sum_k M2[k][row] * dS[c][k]
    for (int c = 0; c < 3; c++)
        for (int k = 0; k < 3; k++) dM[c][k] = M2[0][k] * dS[c][0] + M2[1][k] * dS[c][1] + M2[2][k] * dS[c][2];
    // Rt[i][k] = R[k][i], dMt[i][k] = dM[k][i]; dL_dscale[i] = dot(Rt[i], dMt[i])
    for (int i = 0; i < 3; i++) dL_dscale[i] = R[0][i] * dM[0][i] + R[1][i] * dM[1][i] + R[2][i] * dM[2][i];
#define SGR_MT(i, j) (s[i] * dM[j][i])
    dL_drot[0] = 2 * z * (SGR_MT(0, 1) - SGR_MT(1, 0)) + 2 * y * (SGR_MT(2, 0) - SGR_MT(0, 2)) + 2 * x * (SGR_MT(1, 2) - SGR_MT(2, 1));
    dL_drot[1] = 2 * y * (SGR_MT(1, 0) + SGR_MT(0, 1)) + 2 * z * (SGR_MT(2, 0) + SGR_MT(0, 2)) + 2 * r * (SGR_MT(1, 2) - SGR_MT(2, 1)) -
                 4 * x * (SGR_MT(2, 2) + SGR_MT(1, 1));
    dL_drot[2] = 2 * x * (SGR_MT(1, 0) + SGR_MT(0, 1)) + 2 * r * (SGR_MT(2, 0) - SGR_MT(0, 2)) + 2 * z * (SGR_MT(1, 2) + SGR_MT(2, 1)) -
                 4 * y * (SGR_MT(2, 2) + SGR_MT(0, 0));
    dL_drot[3] = 2 * r * (SGR_MT(0, 1) - SGR_MT(1, 0)) + 2 * x * (SGR_MT(2, 0) + SGR_MT(0, 2)) + 2 * y * (SGR_MT(1, 2) + SGR_MT(2, 1)) -
                 4 * z * (SGR_MT(1, 1) + SGR_MT(0, 0));
#undef SGR_MT
}
