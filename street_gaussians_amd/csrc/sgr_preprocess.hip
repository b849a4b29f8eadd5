// sgr_preprocess.hip -- per-Gaussian forward stage for gfx950:
//   K1 mark_visible, K2 preprocess (cull + EWA projection + SH->RGB), K3 visible_filter,
//   K6 duplicate-with-keys (in depth order), K9 tile ranges.
// Replaces checkFrustum / preprocessCUDA / filter_preprocessCUDA / duplicateWithKeys of the reference
// (rasterizer_impl.cu:54-111, forward.cu:155-334).  One Gaussian per lane, 256-lane workgroups; the
// outputs are packed into three float4 records per Gaussian so that the tile kernels gather each
// instance with three 16-byte loads.
#include "sgr_math.h"

#include <cstdlib>

#ifndef SGR_PRE_THREADS
#define SGR_PRE_THREADS 256
#endif

__device__ __forceinline__ uint32_t sgr_pack_rect(uint32_t x0, uint32_t y0, uint32_t w) {
    return x0 | (y0 << 10) | (w << 20);
}

// ---- K1 -------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SGR_PRE_THREADS)
sgr_mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ v, uint8_t* __restrict__ present) {
#pragma clang fp contract(off)
    // v = the caller's viewmatrix (device): in_frustum only needs its third column (auxiliary.h:139-164)
    const int idx = blockIdx.x * SGR_PRE_THREADS + threadIdx.x;
    if (idx >= P) return;
    const float x = means3D[3 * idx], y = means3D[3 * idx + 1], z = means3D[3 * idx + 2];
    const float tz = v[2] * x + v[6] * y + v[10] * z + v[14];
    present[idx] = tz <= 0.2f ? 0 : 1;
}

// ---- K2 / K3 ----------------------------------------------------------------------------------
// Geometry of one Gaussian (cull, covariance, EWA projection): the part K2 and K3 share.
__device__ __forceinline__ SgrProj
sgr_preprocess_geom(const int idx, const float (&p)[3], const float* __restrict__ scales, const float* __restrict__ rotations,
                    const float* __restrict__ cov3D_precomp, const SgrCam& cam, uint32_t* __restrict__ header,
                    int prefiltered) {
    float tz;
    {
#pragma clang fp contract(off)
        const float* v = cam.view;
        tz = v[2] * p[0] + v[6] * p[1] + v[10] * p[2] + v[14];
    }
    SgrProj pr;
    pr.ok = false;
    if (tz > 0.2f) {
        float cov3D[6];
        if (cov3D_precomp != nullptr) {
#pragma unroll
            for (int i = 0; i < 6; i++) cov3D[i] = cov3D_precomp[6 * (size_t)idx + i];
        } else {
            const float sc[3] = {scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]};
            const float4 q = *reinterpret_cast<const float4*>(rotations + 4 * (size_t)idx);
            const float rot[4] = {q.x, q.y, q.z, q.w};
            sgr_cov3d(sc, cam.scale_modifier, rot, cov3D);  // not stored: the backward recomputes it (same function)
        }
        pr = sgr_project(p, cov3D, cam);
    } else if (prefiltered) {
        if (header) atomicOr(&header[0], 1u);  // reference: printf + __trap() (auxiliary.h:156-161); we raise on the host
    }
    return pr;
}

// K3 (visible_filter): radii + means2D only.  The matrices are read from the caller's device arrays (wave-uniform
// addresses: scalar loads), the scalars travel as kernel arguments -- no allocation, no sync.
__global__ void __launch_bounds__(SGR_PRE_THREADS)
sgr_filter_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ scales,
                  const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp, uint32_t* __restrict__ header,
                  int* __restrict__ radii, float* __restrict__ filter_means2D, int prefiltered, SgrCamArgs ca) {
    const int idx = blockIdx.x * SGR_PRE_THREADS + threadIdx.x;
    SgrCam cam;
#pragma unroll
    for (int i = 0; i < 16; i++) { cam.view[i] = ca.view[i]; cam.proj[i] = ca.proj[i]; }
    cam.campos[0] = cam.campos[1] = cam.campos[2] = 0.f;
    cam.tan_fovx = ca.tan_fovx; cam.tan_fovy = ca.tan_fovy; cam.focal_x = ca.focal_x; cam.focal_y = ca.focal_y;
    cam.W = ca.W; cam.H = ca.H; cam.gx = ca.gx; cam.gy = ca.gy; cam.scale_modifier = ca.scale_modifier;
    if (idx >= P) return;
    const float p[3] = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    const SgrProj pr = sgr_preprocess_geom(idx, p, scales, rotations, cov3D_precomp, cam, header, prefiltered);
    if (!pr.ok) { radii[idx] = 0; return; }
    radii[idx] = pr.radius;
    filter_means2D[2 * idx] = pr.px;
    filter_means2D[2 * idx + 1] = pr.py;
}

// K2.  One Gaussian per lane.  The SH rows of a wave's 64 Gaussians are one contiguous 12 KB block (M = 16) which every
// lane reads with per-lane float4 loads at a 192-byte stride.  A/B form behind switch bit 6 (SGR_PRE_STAGE=1): once the
// geometry has decided which Gaussians survive the cull, the wave copies the survivors' rows through LDS with
// coalesced 1 KB transfers, half of its rows at a time -- measured on MI355X 0.099 vs 0.090 ms at 1 M Gaussians (the
// extra phases lengthen a launch that is only three rounds of workgroups deep) and 0.346 vs 0.360 ms at 5 M: not the
// default.
// num_rendered = sum of tiles_touched does not depend on the depth order, so it is accumulated here (one atomic per
// workgroup into header[4]) and the host can read it back while the depth sort and the offset scan are still running.
__global__ void __launch_bounds__(SGR_PRE_THREADS)
sgr_preprocess_kernel(int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ scales,
                      const float* __restrict__ rotations, const float* __restrict__ opacities,
                      const float* __restrict__ shs, const float* __restrict__ cov3D_precomp,
                      const float* __restrict__ colors_precomp, const SgrCam* __restrict__ camp, SgrGeomView gv,
                      int* __restrict__ radii, int prefiltered, int stage_sh, int tight) {
    // rows padded to 13 float4 (52 dwords): the per-lane float4 reads of 16 consecutive rows then fall on 16 disjoint
    // 4-bank groups (a 48-dword stride puts them on 4).  Dynamic LDS: none when the rows are read directly.
    extern __shared__ float4 sSHdyn[];
    float4 (*sSH)[32 * 13] = reinterpret_cast<float4 (*)[32 * 13]>(sSHdyn);
    __shared__ uint32_t wave_sum[2 * (SGR_PRE_THREADS / 64)];
    const SgrCam& cam = *camp;
    const int gidx = blockIdx.x * SGR_PRE_THREADS + threadIdx.x;
    const bool live = gidx < P;
    const int idx = live ? gidx : P - 1;  // lanes past P help with the cooperative copy; their stores are masked
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float p[3] = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    SgrProj pr = sgr_preprocess_geom(idx, p, scales, rotations, cov3D_precomp, cam, live ? gv.header : nullptr, prefiltered);
    const bool ok = live && pr.ok;

    // A/B form (switch bit 6): the rows of the wave's surviving Gaussians through LDS, HALF the wave's rows at a time
    // (6.5 KB per wave), each lane folding its row into r, g, b as it reads it -- the same operations in the same order
    // as the direct form below.
    const bool stage = stage_sh && shs != nullptr && colors_precomp == nullptr && M == 16;
    float rgb_s[3] = {0.f, 0.f, 0.f};
    if (stage) {
#pragma clang fp contract(off)
        float Y[16];
        {
            float dx = p[0] - cam.campos[0], dy = p[1] - cam.campos[1], dz = p[2] - cam.campos[2];
            const float t0 = dx * dx, t1 = dy * dy, t2 = dz * dz;
            const float len = sqrtf(t0 + t1 + t2);
            dx = dx / len; dy = dy / len; dz = dz / len;
            sgr_sh_basis(D, dx, dy, dz, Y);
        }
        const int ncoef = (D + 1) * (D + 1);
        const int n4 = (ncoef * 3 + 3) >> 2;
        const uint64_t vis = __ballot(ok);
        const int g0 = blockIdx.x * SGR_PRE_THREADS + wave * 64;
        const int nrow4 = max(0, min(64, P - g0)) * 12;  // float4s of this wave's rows
        const float4* src = reinterpret_cast<const float4*>(shs) + (size_t)g0 * 12;
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int it = 0; it < 6; it++) {
                const int f = h * 384 + it * 64 + lane;
                const int row = f / 12;
                if (f < nrow4 && ((vis >> row) & 1ull)) sSH[wave][f - h * 384 + (row - 32 * h)] = src[f];
            }
            __builtin_amdgcn_wave_barrier();
            if ((lane >> 5) == h && ok) {
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    if (i < n4) {
                        const float4 t = sSH[wave][(lane & 31) * 13 + i];
                        const float e[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int k = (4 * i + j) / 3, ch = (4 * i + j) % 3;
                            if (k < ncoef) rgb_s[ch] = (k == 0) ? Y[0] * e[j] : rgb_s[ch] + Y[k] * e[j];
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    uint32_t n = 0, nref = 0;
    if (live && !ok) {
        radii[idx] = 0;
        gv.aux[idx] = make_uint2(0u, 0u);
        if (tight == 3) gv.aux_ref[idx] = make_uint4(0u, 0u, 0u, 0u);
        gv.dkeys[0][idx] = 0xffffffffu;  // sorts behind every real depth (> 0.2 => sign bit clear)
    }
    if (ok) {
        // colour: precomputed, or SH -> RGB (forward.cu:20-71)
        float rgb[3];
        uint32_t clamped = 0;
        if (colors_precomp != nullptr) {
            rgb[0] = colors_precomp[3 * idx];
            rgb[1] = colors_precomp[3 * idx + 1];
            rgb[2] = colors_precomp[3 * idx + 2];
        } else {
#pragma clang fp contract(off)
            float dx = p[0] - cam.campos[0], dy = p[1] - cam.campos[1], dz = p[2] - cam.campos[2];
            const float t0 = dx * dx, t1 = dy * dy, t2 = dz * dz;
            const float len = sqrtf(t0 + t1 + t2);
            dx = dx / len; dy = dy / len; dz = dz / len;
            float Y[16];
            sgr_sh_basis(D, dx, dy, dz, Y);
            const int ncoef = (D + 1) * (D + 1);
            const float* sh = shs + (size_t)idx * M * 3;
            float r = 0.f, g = 0.f, b = 0.f;
            if (stage) {
                r = rgb_s[0]; g = rgb_s[1]; b = rgb_s[2];
            } else if (((M * 3) & 3) == 0 && M <= 16) {
                // row stride is a multiple of 16 B: the row as float4 (12 at SH degree 3)
                const float4* sh4 = reinterpret_cast<const float4*>(sh);
                const int n4 = (ncoef * 3 + 3) >> 2;
                float buf[48];
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    if (i < n4) {
                        const float4 t = sh4[i];
                        buf[4 * i] = t.x; buf[4 * i + 1] = t.y; buf[4 * i + 2] = t.z; buf[4 * i + 3] = t.w;
                    }
                }
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    if (k < ncoef) {
                        if (k == 0) { r = Y[0] * buf[0]; g = Y[0] * buf[1]; b = Y[0] * buf[2]; }
                        else { r = r + Y[k] * buf[3 * k]; g = g + Y[k] * buf[3 * k + 1]; b = b + Y[k] * buf[3 * k + 2]; }
                    }
                }
            } else {
                for (int k = 0; k < ncoef; k++) {
                    if (k == 0) { r = Y[0] * sh[0]; g = Y[0] * sh[1]; b = Y[0] * sh[2]; }
                    else { r = r + Y[k] * sh[3 * k]; g = g + Y[k] * sh[3 * k + 1]; b = b + Y[k] * sh[3 * k + 2]; }
                }
            }
            r += 0.5f; g += 0.5f; b += 0.5f;
            clamped = (r < 0 ? 1u : 0u) | (g < 0 ? 2u : 0u) | (b < 0 ? 4u : 0u);
            rgb[0] = fmaxf(r, 0.0f); rgb[1] = fmaxf(g, 0.0f); rgb[2] = fmaxf(b, 0.0f);
        }

        const float opacity = opacities[idx];
        float hx, hy;
        sgr_extent(opacity, pr.cov_a, pr.cov_c, pr.con_x, pr.con_y, pr.con_z, hx, hy);
        // Tile rect of the Gaussian.  The reference takes the square of ceil(3 sigma_max) pixels around the centre
        // (auxiliary.h:46-57 getRect) and so emits a (tile, Gaussian) instance for many tiles in which the Gaussian cannot pass
        // the alpha >= 1/255 test of the blend (forward.cu:428-430): low opacity, anisotropy.  Those instances change nothing in
        // any output (the blend skips them) but are duplicated, sorted, staged and given a partial-gradient row.  `tight`
        // intersects the reference rect with the tiles of the box [px -+ hx] x [py -+ hy] outside of which alpha < 1/255
        // (sgr_extent: conservative, the box the quadrant cull of the blend kernels has used since round 1): 0.72 of the
        // reference's instances on the benchmark scene; every image bit-identical, the gradients equal up to the grouping of the
        // row sum's additions (tests).  A Gaussian keeps at
        // least one tile, so that "radius > 0" still implies "owns a row".  rn = the reference's count (reported, header[5]).
        uint32_t x0 = pr.rx0, x1 = pr.rx1, y0 = pr.ry0, y1 = pr.ry1;
        const uint32_t rn = (x1 - x0) * (y1 - y0);
        if (tight) {
            // pixel x of tile t: 16 t .. 16 t + 15; pixels that can pass lie in [px - hx, px + hx]  (NaN extents: fmaxf / fminf
            // return the other operand = the reference rect)
            const float inv = 1.0f / SGR_BLOCK_X;
            static_assert(SGR_BLOCK_X == SGR_BLOCK_Y, "square tiles");
            float fx0 = fmaxf(floorf((pr.px - hx) * inv), (float)x0), fx1 = fminf(floorf((pr.px + hx) * inv) + 1.0f, (float)x1);
            float fy0 = fmaxf(floorf((pr.py - hy) * inv), (float)y0), fy1 = fminf(floorf((pr.py + hy) * inv) + 1.0f, (float)y1);
            if (!(fx0 < fx1)) { fx0 = fminf(fmaxf(floorf(pr.px * inv), (float)x0), (float)(x1 - 1)); fx1 = fx0 + 1.0f; }
            if (!(fy0 < fy1)) { fy0 = fminf(fmaxf(floorf(pr.py * inv), (float)y0), (float)(y1 - 1)); fy1 = fy0 + 1.0f; }
            x0 = (uint32_t)fx0; x1 = (uint32_t)fx1; y0 = (uint32_t)fy0; y1 = (uint32_t)fy1;
        }
        // tight == 3 (marked-list mode): the reference's rect is what gets emitted (aux_ref); the cut-down rect + mask below
        // say which of those instances are live and number the Gaussian's partial-gradient rows
        const uint32_t w = x1 - x0, h = y1 - y0;
        uint32_t rect = sgr_pack_rect(x0, y0, w);
        uint32_t nemit = w * h;
        if (tight > 1 && w >= 2u && h >= 2u && nemit <= 64u) {  // (a single row or column of tiles IS its bounding box)
            // ... and inside that rect only the tiles the alpha >= 1/255 ellipse reaches (the rect is its bounding box: the
            // corners go): a 64-bit tile mask for rects of up to 64 tiles (sgr_math.h: sgr_tile_mask), bit 31 of the rect word
            // says "masked".  duplicate emits the set tiles, the backward numbers a Gaussian's rows by the rank of the tile in
            // the mask.  0.61 instead of 0.72 of the reference's instances on the benchmark scene.
            uint64_t tm = sgr_tile_mask(pr.px, pr.py, pr.con_x, pr.con_y, pr.con_z, opacity, x0, y0, x1, y1);
            if (tm == 0) {  // reaches no tile (opacity below 1/255 ...): the centre's tile, so that the Gaussian owns a row
                const uint32_t cx = (uint32_t)fminf(fmaxf(floorf(pr.px * (1.0f / SGR_BLOCK_X)), (float)x0), (float)(x1 - 1));
                const uint32_t cy = (uint32_t)fminf(fmaxf(floorf(pr.py * (1.0f / SGR_BLOCK_Y)), (float)y0), (float)(y1 - 1));
                tm = 1ull << ((cy - y0) * w + (cx - x0));
            }
            const uint32_t cnt = (uint32_t)__builtin_popcountll(tm);
            if (cnt != nemit) {
                rect |= SGR_RECT_MASKED;
                gv.tmask[idx] = tm;
                nemit = cnt;
            }
        }
        nref = rn;
        float4* rec = gv.rec + 4 * (size_t)idx;
        rec[0] = make_float4(pr.px, pr.py, hx, hy);
        rec[1] = make_float4(pr.con_x, pr.con_y, pr.con_z, opacity);
        rec[2] = make_float4(rgb[0], rgb[1], rgb[2], pr.depth);
        // rec[3] = {qa, packed tile rect, qb, qc}: the pre-scaled conic rides in the line's spare words, so that the
        // scalar-walk backward gets everything a visit needs with one s_load_dwordx16
        const float4 st = sgr_stage_conic(make_float4(pr.con_x, pr.con_y, pr.con_z, opacity));
        rec[3] = make_float4(st.x, __uint_as_float(rect), st.y, st.z);
        gv.clamped[idx] = clamped;
        gv.aux[idx] = make_uint2(nemit, rect);
        // tight == 3 (marked-list mode): the reference's rect is what gets emitted; the cut-down rect + mask say which of those
        // instances are live and number the Gaussian's partial-gradient rows
        if (tight == 3) gv.aux_ref[idx] = make_uint4(rn, sgr_pack_rect(pr.rx0, pr.ry0, pr.rx1 - pr.rx0), nemit, rect);
        // depth-sort key: the bits of the view depth minus the bits of 0.2 (every Gaussian that gets here has depth > 0.2):
        // monotone in the depth, and below 2^27 for depths under 13 107 -- sixteen octaves -- so that the sort takes THREE 9-bit
        // passes instead of four 8-bit ones.  A depth beyond that raises header[2]; the host reads it back together with
        // num_rendered and repeats the forward's front end with a sort on all 32 bits (sgr_api.hip).  Culled: all ones.
        const uint32_t dkey = __float_as_uint(pr.depth) - SGR_DEPTH_KEY_BIAS;
        if (dkey >> SGR_DEPTH_KEY_BITS) atomicOr(&gv.header[2], 1u);
        gv.dkeys[0][idx] = dkey;
        radii[idx] = pr.radius;
        n = tight == 3 ? rn : nemit;  // the length of the list
    }
    // device-scope atomics on one address are resolved beyond the per-XCD L2s (~6 ns each, serialised): one per
    // workgroup, not one per wave (16k of them cost 0.1 ms at P = 1M)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { n += __shfl_xor(n, m, 64); nref += __shfl_xor(nref, m, 64); }
    if (lane == 0) { wave_sum[wave] = n; wave_sum[SGR_PRE_THREADS / 64 + wave] = nref; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0, tr = 0;
#pragma unroll
        for (int i = 0; i < SGR_PRE_THREADS / 64; i++) { t += wave_sum[i]; tr += wave_sum[SGR_PRE_THREADS / 64 + i]; }
        // ONE 64-bit atomic for both sums (header[4] = num_rendered, header[5] = what it is with the reference's rects, reported
        // only): a second 32-bit atomic per workgroup cost 0.03 ms at 1 M Gaussians and 0.18 ms at 5 M (measured)
        if (tr) atomicAdd(reinterpret_cast<unsigned long long*>(gv.header + 4), (unsigned long long)t | ((unsigned long long)tr << 32));
    }
}

// ---- K6: one (tile key, Gaussian id) per overlapped tile (rasterizer_impl.cu:70-111).  Lane i handles the i-th
// Gaussian in (depth, id) order, so the instance array is already ordered by the low 32 bits of the reference's
// 64-bit key and a STABLE sort on the tile id alone reproduces the reference's order, ties included.
// Wave-cooperative emission: the 64 Gaussians of a wave own ONE contiguous range of slots [start, end).  Lane l
// writes slots start + 64*it + l (fully coalesced 256-byte stores; a lane-per-Gaussian loop writes 64 scattered
// 4-byte runs per instruction and was store-issue bound: 0.08 ms for 62 MB) and finds the slot's owner with a binary
// search over the wave's 64 exclusive offsets in LDS, then its tile from the owner's packed rect.
// Round 5: the kernel also does the LAST step of the forward's two scans itself (sgr_launch_scan_head left the block
// prefixes and the sub-block offsets): the slot range of every Gaussian in depth order = prefix of its 2048-block + offset of
// its 256-element sub-block + an exclusive scan over the workgroup's 256 counts -- and, in the same way over the counts in
// INDEX order, every Gaussian's first partial-gradient row u0 (SgrGeomView::u0).  One launch and one scanned array
// (4 bytes / Gaussian written and read back, 8 more read) less per forward: scan + duplicate 0.045 -> 0.036 ms at 1 M
// Gaussians, 0.16 -> 0.12 ms at 5 M.
static_assert(SGR_PRE_THREADS == 256 && SGR_SCAN_ITEMS == 8 * SGR_PRE_THREADS, "one workgroup = one sub-block of the scan");
// K = the tile keys' type: uint16_t whenever the frame has fewer than 65535 tiles (the sort's key traffic halves), else uint32_t
template <typename K>
__global__ void __launch_bounds__(SGR_PRE_THREADS)
sgr_duplicate_kernel(int P, SgrGeomView gv, const uint32_t* __restrict__ order, const uint32_t* __restrict__ bsum,
                     const uint32_t* __restrict__ sub, uint32_t nb, K* __restrict__ keys, uint32_t* __restrict__ vals,
                     int gx, uint32_t cap, int marks) {
    // cap != 0 (the forward without a host wait, sgr_set_lazy): the list buffers hold `cap` slots whatever the frame's
    // instance count R turns out to be (bsum[nb], known to the device only) -- nothing is written past them, and the slots
    // [R, cap) get a key above every tile id, so that the sort over all `cap` slots leaves them at the end
    if (cap != 0) {
        const uint32_t R = bsum[nb];
        for (uint32_t s = R + blockIdx.x * SGR_PRE_THREADS + threadIdx.x; s < cap; s += gridDim.x * SGR_PRE_THREADS) {
            keys[s] = (K)~(K)0;
            vals[s] = 0u;
        }
    }
    __shared__ uint32_t sOff[SGR_PRE_THREADS / 64][64];
    __shared__ uint32_t sRect[SGR_PRE_THREADS / 64][64];
    __shared__ uint32_t sIdx[SGR_PRE_THREADS / 64][64];
    __shared__ uint64_t sMask[SGR_PRE_THREADS / 64][64];
    __shared__ uint32_t sLive[SGR_PRE_THREADS / 64][64];   // marked-list mode: the owner's cut-down rect word
    __shared__ uint32_t sLiveN[SGR_PRE_THREADS / 64][64];  // ... and its number of live tiles
    __shared__ uint32_t lds4[4];
    const int i = blockIdx.x * SGR_PRE_THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t off = 0xffffffffu, incl = 0, idx = 0, rect = 0, live_rect = 0, live_n = 0;
    uint64_t tmask = 0;
    // everything in depth order and coalesced: the id, and {tiles_touched, tile rect} the depth sort's last pass carried
    // along (aux_sorted) -- no gather of the Gaussian's record
    uint2 as = make_uint2(0u, 0u), lv = make_uint2(0u, 0u);
    if (i < P) {
        if (marks) {  // 16-byte records {reference count, reference rect, live tiles, cut-down rect}
            const uint4 t = reinterpret_cast<const uint4*>(gv.aux_sorted)[i];
            as = make_uint2(t.x, t.y);
            lv = make_uint2(t.z, t.w);
        } else as = gv.aux_sorted[i];
    }
    const uint32_t t2 = (i < P) ? gv.aux[i].x : 0u;  // the count of rows in index order (marked-list mode: of the LIVE tiles)
    uint32_t total;
    const uint32_t ex1 = sgr_block_excl_scan256(as.x, lds4, total) + bsum[blockIdx.x >> 3] + sub[blockIdx.x];
    const uint32_t ex2 = sgr_block_excl_scan256(t2, lds4, total) + bsum[nb + 1 + (blockIdx.x >> 3)] + sub[8 * nb + blockIdx.x];
    if (i < P) {
        gv.u0[i] = ex2;
        idx = order ? order[i] : (uint32_t)i;  // order == nullptr: emission in index order (per-tile sort form, sgr_tile_sort.hip)
        off = ex1;
        incl = ex1 + as.x;
        if (incl != off) {  // tiles_touched > 0
            rect = as.y;
            if (marks) {  // the cut-down rect and its mask: which tiles of the reference's rect are live
                live_rect = lv.y;
                live_n = lv.x;
                if (live_rect & SGR_RECT_MASKED) tmask = gv.tmask[idx];
            } else if (rect & SGR_RECT_MASKED) tmask = gv.tmask[idx];  // (by id: 8 bytes, masked Gaussians only)
        }
    }
    sOff[wave][lane] = off;  // lanes past P: 0xffffffff, never <= a slot
    sRect[wave][lane] = rect;
    sIdx[wave][lane] = idx;
    sMask[wave][lane] = tmask;
    if (marks) { sLive[wave][lane] = live_rect; sLiveN[wave][lane] = live_n; }
    // wave-uniform slot range: exclusive offset of lane 0, inclusive offset of the last lane below P
    const uint32_t start = __builtin_amdgcn_readfirstlane(off);
    const int last = min(63, P - 1 - (blockIdx.x * SGR_PRE_THREADS + wave * 64));
    if (last < 0) return;  // whole wave past P
    uint32_t end = __builtin_amdgcn_readlane(incl, last);
    if (cap != 0 && end > cap) end = cap;  // (an overflowing frame is reported by the host once R has landed)
    __builtin_amdgcn_wave_barrier();  // LDS of this wave only: program order + s_waitcnt is enough
    for (uint32_t s = start + lane; s < end; s += 64) {
        // owner = largest l with sOff[l] <= s (offsets are non-decreasing; Gaussians without tiles share the next
        // one's offset, so "largest" skips them)
        int o = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1)
            if (sOff[wave][o + step] <= s) o += step;
        const uint32_t r = sRect[wave][o];
        const uint32_t x0 = r & 1023u, y0 = (r >> 10) & 1023u, w = (r >> 20) & 1023u;
        uint32_t k = s - sOff[wave][o];
        if (!marks && (r & SGR_RECT_MASKED)) k = sgr_select_bit(sMask[wave][o], k);  // the k-th tile of the mask, as an index into the rect
        // k / w with one v_rcp_f32 and an exact fix-up (k < 2^20)
        uint32_t q = (uint32_t)((float)k * __builtin_amdgcn_rcpf((float)w));
        int rem = (int)k - (int)(q * w);
        if (rem < 0) { q--; rem += (int)w; }
        if (rem >= (int)w) { q++; rem -= (int)w; }
        const uint32_t tx = x0 + (uint32_t)rem, ty = y0 + q;
        keys[s] = (K)(ty * (uint32_t)gx + tx);
        uint32_t v = sIdx[wave][o];
        if (marks) {
            // live = inside the cut-down rect and, with a mask, one of its set tiles (index j inside that rect: below the
            // number of its tiles without a mask -- which bounds the row -- or a set bit with one)
            const uint32_t lr = sLive[wave][o];
            const uint32_t lx0 = lr & 1023u, ly0 = (lr >> 10) & 1023u, lw = (lr >> 20) & 1023u;
            const uint32_t dxl = tx - lx0, dyl = ty - ly0;  // (wrap around for tiles left of / above the rect: >= lw / huge)
            const uint32_t j = dyl * lw + dxl;
            bool live = dxl < lw && ty >= ly0;
            if (lr & SGR_RECT_MASKED) live = live && j < 64u && ((sMask[wave][o] >> (j & 63u)) & 1ull);
            else live = live && j < sLiveN[wave][o];
            if (!live) v |= SGR_DEAD;
        }
        vals[s] = v;
    }
}

// ---- K9: tile ranges from the sorted tile keys (rasterizer_impl.cu:116-138) -------------------
// Eight consecutive entries per thread: one 16- / 32-byte load of the keys (+ the key in front of them), one 8-byte store of
// row flags -- the launch is a stream of 6 (10) bytes per entry, and one entry per thread left it latency-bound (1.7 TB/s).
#define SGR_RANGES_PER_THREAD 8
template <typename K>
__global__ void __launch_bounds__(256)
sgr_tile_ranges_kernel(int L, const K* __restrict__ keys, uint2* __restrict__ ranges, uint8_t* __restrict__ touched,
                       uint32_t T) {
    constexpr int E = SGR_RANGES_PER_THREAD;
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * E;
    if (i0 >= L) return;
    K k[E];
    if (i0 + E <= L) {  // (the key / flag arrays start 256-byte aligned and i0 is a multiple of 8)
        struct alignas(sizeof(K) * E) Pack { K v[E]; };
        const Pack p = *reinterpret_cast<const Pack*>(keys + i0);
#pragma unroll
        for (int e = 0; e < E; e++) k[e] = p.v[e];
        // one byte per partial-gradient row of the backward ("row written"), cleared here instead of by a memset dispatch
        // in front of the backward's dominant kernel (the index spaces coincide: one row per instance)
        *reinterpret_cast<uint2*>(touched + i0) = make_uint2(0u, 0u);
    } else {
#pragma unroll
        for (int e = 0; e < E; e++) {
            k[e] = i0 + e < L ? keys[i0 + e] : (K)0;
            if (i0 + e < L) touched[i0 + e] = 0;
        }
    }
    // keys >= T: the padding behind the frame's instances when the list has a fixed capacity (sgr_duplicate_kernel, cap)
    uint32_t prevtile = i0 > 0 ? (uint32_t)keys[i0 - 1] : 0u;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int idx = i0 + e;
        if (idx >= L) break;
        const uint32_t currtile = k[e];
        if (idx == 0) {
            if (currtile < T) ranges[currtile].x = 0;
        } else if (currtile != prevtile) {
            if (prevtile < T) ranges[prevtile].y = idx;
            if (currtile < T) ranges[currtile].x = idx;
        }
        if (idx == L - 1 && currtile < T) ranges[currtile].y = L;
        prevtile = currtile;
    }
}

// ---- tile order of the blend launches (sgr_wg_tile): decides per frame whether the lists are unequal enough for a
// LONGEST-FIRST walk -- the longest list more than `ratio_x16` / 16 times the mean (force: always) -- and if so sorts the T
// tile ids by descending list length into the T words behind the ranges; the word behind those is the flag.  One workgroup:
// maximum + sum, a 1024-bucket histogram of length * 1023 / max (descending), scan, scatter; the order inside a bucket is
// whatever the LDS atomics give (tiles are independent: no result depends on it).
__global__ void __launch_bounds__(1024)
sgr_tile_order_kernel(const uint2* __restrict__ ranges, uint32_t T, uint32_t* __restrict__ order, uint32_t ratio_x16, int force) {
    __shared__ uint32_t hist[1024];
    __shared__ uint32_t smax;
    __shared__ unsigned long long ssum;
    __shared__ uint32_t sw[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    hist[tid] = 0u;
    if (tid == 0) { smax = 0u; ssum = 0ull; }
    __syncthreads();
    uint32_t mx = 0;
    unsigned long long sm = 0;
    for (uint32_t t = tid; t < T; t += 1024) {
        const uint32_t len = ranges[t].y - ranges[t].x;
        mx = max(mx, len);
        sm += len;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
        sm += (unsigned long long)__shfl_xor((long long)sm, o, 64);
    }
    if (lane == 0) { atomicMax(&smax, mx); atomicAdd(&ssum, sm); }
    __syncthreads();
    // longest list vs the mean list: max * T * 16 > ratio_x16 * sum
    const bool lpt = force || (unsigned long long)smax * T * 16ull > (unsigned long long)ratio_x16 * ssum;
    if (tid == 0) order[T] = lpt ? 1u : 0u;
    if (!lpt) return;
    const float sc = 1023.0f / (float)max(smax, 1u);
    for (uint32_t t = tid; t < T; t += 1024) {
        const uint32_t len = ranges[t].y - ranges[t].x;
        atomicAdd(&hist[1023u - min(1023u, (uint32_t)((float)len * sc))], 1u);
    }
    __syncthreads();
    const uint32_t c = hist[tid];
    const uint32_t inc = sgr_wave_incl_scan(c, lane);
    if (lane == 63) sw[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; w++) base += sw[w];
    __syncthreads();
    hist[tid] = base + inc - c;
    __syncthreads();
    for (uint32_t t = tid; t < T; t += 1024) {
        const uint32_t len = ranges[t].y - ranges[t].x;
        order[atomicAdd(&hist[1023u - min(1023u, (uint32_t)((float)len * sc))], 1u)] = t;
    }
}
void sgr_launch_tile_order(const uint2* ranges, int T, int force, hipStream_t s) {
    if (T <= 0) return;
    // threshold: the longest list > 2.5 x the mean list (the benchmark's uniform scenes: 1.4-1.7 x; a street scene: 5 x)
    static const uint32_t ratio_x16 = [] { const char* e = getenv("SGR_LPT_RATIO_X16"); return e ? (uint32_t)atoi(e) : 40u; }();
    sgr_tile_order_kernel<<<1, 1024, 0, s>>>(ranges, (uint32_t)T, reinterpret_cast<uint32_t*>(const_cast<uint2*>(ranges) + T), ratio_x16, force);
}

// parity introspection: the reference's 64-bit sorted keys, recomposed from tile id and depth bits
template <typename K>
__global__ void __launch_bounds__(256)
sgr_compose_keys_kernel(int L, const K* __restrict__ tile_keys, const uint32_t* __restrict__ point_list,
                        const float4* __restrict__ rec, uint64_t* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L) return;
    out[i] = ((uint64_t)tile_keys[i] << 32) | (uint64_t)__float_as_uint(rec[4 * (size_t)(point_list[i] & ~SGR_DEAD) + 2].w);
}

// ---- host launchers ---------------------------------------------------------------------------
void sgr_launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s) {
    if (P <= 0) return;
    sgr_mark_visible_kernel<<<(P + SGR_PRE_THREADS - 1) / SGR_PRE_THREADS, SGR_PRE_THREADS, 0, s>>>(P, means3D, viewmatrix,
                                                                                                 present);
}

void sgr_launch_preprocess(int P, int D, int M, const float* means3D, const float* scales, const float* rotations,
                           const float* opacities, const float* shs, const float* cov3D_precomp,
                           const float* colors_precomp, const SgrCam* cam, const SgrGeomView& gv, int* radii,
                           int prefiltered, bool stage_sh, int tight, hipStream_t s) {
    if (P <= 0) return;
    const bool stage = stage_sh && shs != nullptr && colors_precomp == nullptr && M == 16;
    const size_t lds = stage ? (size_t)(SGR_PRE_THREADS / 64) * 32 * 13 * sizeof(float4) : 0;
    sgr_preprocess_kernel<<<(P + SGR_PRE_THREADS - 1) / SGR_PRE_THREADS, SGR_PRE_THREADS, lds, s>>>(
        P, D, M, means3D, scales, rotations, opacities, shs, cov3D_precomp, colors_precomp, cam, gv, radii, prefiltered,
        stage ? 1 : 0, tight);
}

void sgr_launch_filter(int P, const float* means3D, const float* scales, const float* rotations,
                       const float* cov3D_precomp, const SgrCamArgs& ca, const SgrGeomView& gv, int* radii,
                       float* means2D, int prefiltered, hipStream_t s) {
    if (P <= 0) return;
    sgr_filter_kernel<<<(P + SGR_PRE_THREADS - 1) / SGR_PRE_THREADS, SGR_PRE_THREADS, 0, s>>>(
        P, means3D, scales, rotations, cov3D_precomp, gv.header, radii, means2D, prefiltered, ca);
}

// bsum / sub: what sgr_launch_scan_head left for the two count sequences (aux_sorted in depth order, aux in index order)
// key16: the tile keys are uint16_t (sgr_api.hip: tile_key16) -- `keys` points at the same buffer either way
void sgr_launch_duplicate(int P, const SgrGeomView& gv, const uint32_t* order, const uint32_t* bsum, const uint32_t* sub,
                          void* keys, int key16, uint32_t* vals, int gx, uint32_t cap, int marks, hipStream_t s) {
    if (P <= 0) return;
    const uint32_t nb = (uint32_t)(((size_t)P + SGR_SCAN_ITEMS - 1) / SGR_SCAN_ITEMS);
    const unsigned grid = (P + SGR_PRE_THREADS - 1) / SGR_PRE_THREADS;
    if (key16) sgr_duplicate_kernel<uint16_t><<<grid, SGR_PRE_THREADS, 0, s>>>(P, gv, order, bsum, sub, nb, (uint16_t*)keys, vals, gx, cap, marks);
    else sgr_duplicate_kernel<uint32_t><<<grid, SGR_PRE_THREADS, 0, s>>>(P, gv, order, bsum, sub, nb, (uint32_t*)keys, vals, gx, cap, marks);
}

void sgr_launch_tile_ranges(int L, const void* keys, int key16, uint2* ranges, uint8_t* touched, uint32_t T, hipStream_t s) {
    if (L <= 0) return;
    const unsigned grid = (unsigned)((L + 256 * SGR_RANGES_PER_THREAD - 1) / (256 * SGR_RANGES_PER_THREAD));
    if (key16) sgr_tile_ranges_kernel<uint16_t><<<grid, 256, 0, s>>>(L, (const uint16_t*)keys, ranges, touched, T);
    else sgr_tile_ranges_kernel<uint32_t><<<grid, 256, 0, s>>>(L, (const uint32_t*)keys, ranges, touched, T);
}

void sgr_launch_compose_keys(int L, const void* tile_keys, int key16, const uint32_t* point_list, const float4* rec, uint64_t* out,
                             hipStream_t s) {
    if (L <= 0) return;
    if (key16) sgr_compose_keys_kernel<uint16_t><<<(L + 255) / 256, 256, 0, s>>>(L, (const uint16_t*)tile_keys, point_list, rec, out);
    else sgr_compose_keys_kernel<uint32_t><<<(L + 255) / 256, 256, 0, s>>>(L, (const uint32_t*)tile_keys, point_list, rec, out);
}
