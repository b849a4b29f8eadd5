// sgr_preprocess.hip -- per-Gaussian forward stage for gfx950:
//   K1 mark_visible, K2 preprocess (cull + EWA projection + SH->RGB), K3 visible_filter,
//   K6 duplicate-with-keys (in depth order), K9 tile ranges.
// Replaces checkFrustum / preprocessCUDA / filter_preprocessCUDA / duplicateWithKeys of the reference
// (rasterizer_impl.cu:54-111, forward.cu:155-334).  One Gaussian per lane, 256-lane workgroups; the
// outputs are packed into three float4 records per Gaussian so that the tile kernels gather each
// instance with three 16-byte loads.
#include "sgr_math.h"

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
// One Gaussian; returns its tiles_touched (0 when culled).
template <bool FILTER>
__device__ __forceinline__ uint32_t
sgr_preprocess_one(const int idx, int D, int M, const float* __restrict__ means3D, const float* __restrict__ scales,
                   const float* __restrict__ rotations, const float* __restrict__ opacities,
                   const float* __restrict__ shs, const float* __restrict__ cov3D_precomp,
                   const float* __restrict__ colors_precomp, const SgrCam& cam, const SgrGeomView& gv,
                   int* __restrict__ radii, float* __restrict__ filter_means2D, int prefiltered) {

    const float p[3] = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
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
            sgr_cov3d(sc, cam.scale_modifier, rot, cov3D);
            if (!FILTER) {
#pragma unroll
                for (int i = 0; i < 6; i++) gv.cov3D[6 * (size_t)idx + i] = cov3D[i];
            }
        }
        pr = sgr_project(p, cov3D, cam);
    } else if (prefiltered) {
        if (gv.header) atomicOr(&gv.header[0], 1u);  // reference: printf + __trap() (auxiliary.h:156-161); we raise on the host
    }

    if (!FILTER) gv.dvals[0][idx] = (uint32_t)idx;
    if (!pr.ok) {
        radii[idx] = 0;
        if (!FILTER) {
            gv.tiles_touched[idx] = 0;
            gv.dkeys[0][idx] = 0xffffffffu;  // sorts behind every real depth (> 0.2 => sign bit clear)
        }
        return 0;
    }
    if (FILTER) {
        radii[idx] = pr.radius;
        filter_means2D[2 * idx] = pr.px;
        filter_means2D[2 * idx + 1] = pr.py;
        return 0;
    }

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
        if (((M * 3) & 3) == 0) {
            // row stride is a multiple of 16 B: stream the row as float4 (12 loads at SH degree 3)
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
    const uint32_t w = pr.rx1 - pr.rx0, h = pr.ry1 - pr.ry0;
    float4* rec = gv.rec + 4 * (size_t)idx;
    rec[0] = make_float4(pr.px, pr.py, hx, hy);
    rec[1] = make_float4(pr.con_x, pr.con_y, pr.con_z, opacity);
    rec[2] = make_float4(rgb[0], rgb[1], rgb[2], pr.depth);
    rec[3] = make_float4(0.f, __uint_as_float(sgr_pack_rect(pr.rx0, pr.ry0, w)), 0.f, 0.f);
    gv.clamped[idx] = clamped;
    gv.tiles_touched[idx] = w * h;
    gv.dkeys[0][idx] = __float_as_uint(pr.depth);
    radii[idx] = pr.radius;
    return w * h;
}

// num_rendered = sum of tiles_touched does not depend on the depth order, so it is accumulated here (one atomic per
// workgroup into header[1]) and the host can read it back while the depth sort and the offset scan are still running.
template <bool FILTER>
__global__ void __launch_bounds__(SGR_PRE_THREADS)
sgr_preprocess_kernel(int P, int D, int M, const float* __restrict__ means3D, const float* __restrict__ scales,
                      const float* __restrict__ rotations, const float* __restrict__ opacities,
                      const float* __restrict__ shs, const float* __restrict__ cov3D_precomp,
                      const float* __restrict__ colors_precomp, const SgrCam* __restrict__ camp, SgrGeomView gv,
                      int* __restrict__ radii, float* __restrict__ filter_means2D, int prefiltered, SgrCamArgs ca) {
    const int idx = blockIdx.x * SGR_PRE_THREADS + threadIdx.x;
    uint32_t n = 0;
    if (FILTER) {
        // K3 has no geometry buffer to keep a packed camera in: the matrices are read from the caller's device arrays
        // (wave-uniform addresses: scalar loads), the scalars travel as kernel arguments -- no allocation, no sync
        SgrCam cam;
#pragma unroll
        for (int i = 0; i < 16; i++) { cam.view[i] = ca.view[i]; cam.proj[i] = ca.proj[i]; }
        cam.campos[0] = cam.campos[1] = cam.campos[2] = 0.f;
        cam.tan_fovx = ca.tan_fovx; cam.tan_fovy = ca.tan_fovy; cam.focal_x = ca.focal_x; cam.focal_y = ca.focal_y;
        cam.W = ca.W; cam.H = ca.H; cam.gx = ca.gx; cam.gy = ca.gy; cam.scale_modifier = ca.scale_modifier;
        if (idx < P)
            sgr_preprocess_one<true>(idx, D, M, means3D, scales, rotations, opacities, shs, cov3D_precomp, colors_precomp,
                                     cam, gv, radii, filter_means2D, prefiltered);
        return;
    }
    if (idx < P)
        n = sgr_preprocess_one<FILTER>(idx, D, M, means3D, scales, rotations, opacities, shs, cov3D_precomp, colors_precomp,
                                       *camp, gv, radii, filter_means2D, prefiltered);
    // device-scope atomics on one address are resolved beyond the per-XCD L2s (~6 ns each, serialised): one per
    // workgroup, not one per wave (16k of them cost 0.1 ms at P = 1M)
    __shared__ uint32_t wave_sum[SGR_PRE_THREADS / 64];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) n += __shfl_xor(n, m, 64);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
#pragma unroll
        for (int i = 0; i < SGR_PRE_THREADS / 64; i++) t += wave_sum[i];
        if (t) atomicAdd(&gv.header[1], t);
    }
}

// ---- tiles_touched gathered in (depth, id) order, ready for the scan that gives every Gaussian its slot ----
__global__ void __launch_bounds__(SGR_PRE_THREADS)
sgr_gather_tiles_kernel(int P, const uint32_t* __restrict__ order, const uint32_t* __restrict__ tiles_touched,
                        uint32_t* __restrict__ tt_sorted) {
    const int i = blockIdx.x * SGR_PRE_THREADS + threadIdx.x;
    if (i >= P) return;
    tt_sorted[i] = tiles_touched[order[i]];
}

// ---- K6: one (tile key, Gaussian id) per overlapped tile (rasterizer_impl.cu:70-111).  Lane i handles the i-th
// Gaussian in (depth, id) order, so the instance array is already ordered by the low 32 bits of the reference's
// 64-bit key and a STABLE sort on the tile id alone reproduces the reference's order, ties included.  The
// Gaussian's first slot is also its first partial-gradient row in the backward (rec[3].x).
// Wave-cooperative emission: the 64 Gaussians of a wave own ONE contiguous range of slots [start, end).  Lane l
// writes slots start + 64*it + l (fully coalesced 256-byte stores; a lane-per-Gaussian loop writes 64 scattered
// 4-byte runs per instruction and was store-issue bound: 0.08 ms for 62 MB) and finds the slot's owner with a binary
// search over the wave's 64 exclusive offsets in LDS, then its tile from the owner's packed rect.
__global__ void __launch_bounds__(SGR_PRE_THREADS)
sgr_duplicate_kernel(int P, SgrGeomView gv, const uint32_t* __restrict__ order, const uint32_t* __restrict__ offs_incl,
                     uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, int gx) {
    __shared__ uint32_t sOff[SGR_PRE_THREADS / 64][64];
    __shared__ uint32_t sRect[SGR_PRE_THREADS / 64][64];
    __shared__ uint32_t sIdx[SGR_PRE_THREADS / 64][64];
    const int i = blockIdx.x * SGR_PRE_THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t off = 0xffffffffu, incl = 0, idx = 0, rect = 0;
    if (i < P) {
        idx = order[i];
        off = (i == 0) ? 0u : offs_incl[i - 1];
        incl = offs_incl[i];
        if (incl != off) {  // tiles_touched > 0
            float4* rec = gv.rec + 4 * (size_t)idx;
            float4 d4 = rec[3];
            rect = __float_as_uint(d4.y);
            d4.x = __uint_as_float(off);
            rec[3] = d4;
        }
    }
    sOff[wave][lane] = off;  // lanes past P: 0xffffffff, never <= a slot
    sRect[wave][lane] = rect;
    sIdx[wave][lane] = idx;
    // wave-uniform slot range: exclusive offset of lane 0, inclusive offset of the last lane below P
    const uint32_t start = __builtin_amdgcn_readfirstlane(off);
    const int last = min(63, P - 1 - (blockIdx.x * SGR_PRE_THREADS + wave * 64));
    if (last < 0) return;  // whole wave past P
    const uint32_t end = __builtin_amdgcn_readlane(incl, last);
    __builtin_amdgcn_wave_barrier();  // LDS of this wave only: program order + s_waitcnt is enough
    for (uint32_t s = start + lane; s < end; s += 64) {
        // owner = largest l with sOff[l] <= s (offsets are non-decreasing; Gaussians without tiles share the next
        // one's offset, so "largest" skips them)
        int o = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1)
            if (sOff[wave][o + step] <= s) o += step;
        const uint32_t r = sRect[wave][o];
        const uint32_t x0 = r & 1023u, y0 = (r >> 10) & 1023u, w = r >> 20;
        const uint32_t k = s - sOff[wave][o];
        // k / w with one v_rcp_f32 and an exact fix-up (k < 2^20)
        uint32_t q = (uint32_t)((float)k * __builtin_amdgcn_rcpf((float)w));
        int rem = (int)k - (int)(q * w);
        if (rem < 0) { q--; rem += (int)w; }
        if (rem >= (int)w) { q++; rem -= (int)w; }
        keys[s] = (y0 + q) * (uint32_t)gx + x0 + (uint32_t)rem;
        vals[s] = sIdx[wave][o];
    }
}

// ---- K9: tile ranges from the sorted tile keys (rasterizer_impl.cu:116-138) -------------------
__global__ void __launch_bounds__(256)
sgr_tile_ranges_kernel(int L, const uint32_t* __restrict__ keys, uint2* __restrict__ ranges, uint8_t* __restrict__ touched) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= L) return;
    // one byte per partial-gradient row of the backward ("row written"), cleared here instead of by a memset dispatch
    // in front of the backward's dominant kernel (the index spaces coincide: one row per instance)
    touched[idx] = 0;
    const uint32_t currtile = keys[idx];
    if (idx == 0) {
        ranges[currtile].x = 0;
    } else {
        const uint32_t prevtile = keys[idx - 1];
        if (currtile != prevtile) {
            ranges[prevtile].y = idx;
            ranges[currtile].x = idx;
        }
    }
    if (idx == L - 1) ranges[currtile].y = L;
}

// parity introspection: the reference's 64-bit sorted keys, recomposed from tile id and depth bits
__global__ void __launch_bounds__(256)
sgr_compose_keys_kernel(int L, const uint32_t* __restrict__ tile_keys, const uint32_t* __restrict__ point_list,
                        const float4* __restrict__ rec, uint64_t* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L) return;
    out[i] = ((uint64_t)tile_keys[i] << 32) | (uint64_t)__float_as_uint(rec[4 * (size_t)point_list[i] + 2].w);
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
                           int prefiltered, hipStream_t s) {
    if (P <= 0) return;
    sgr_preprocess_kernel<false><<<(P + SGR_PRE_THREADS - 1) / SGR_PRE_THREADS, SGR_PRE_THREADS, 0, s>>>(
        P, D, M, means3D, scales, rotations, opacities, shs, cov3D_precomp, colors_precomp, cam, gv, radii, nullptr,
        prefiltered, SgrCamArgs{});
}

void sgr_launch_filter(int P, const float* means3D, const float* scales, const float* rotations,
                       const float* cov3D_precomp, const SgrCamArgs& ca, const SgrGeomView& gv, int* radii,
                       float* means2D, int prefiltered, hipStream_t s) {
    if (P <= 0) return;
    sgr_preprocess_kernel<true><<<(P + SGR_PRE_THREADS - 1) / SGR_PRE_THREADS, SGR_PRE_THREADS, 0, s>>>(
        P, 0, 0, means3D, scales, rotations, nullptr, nullptr, cov3D_precomp, nullptr, nullptr, gv, radii, means2D,
        prefiltered, ca);
}

void sgr_launch_gather_tiles(int P, const uint32_t* order, const uint32_t* tiles_touched, uint32_t* tt_sorted,
                             hipStream_t s) {
    if (P <= 0) return;
    sgr_gather_tiles_kernel<<<(P + SGR_PRE_THREADS - 1) / SGR_PRE_THREADS, SGR_PRE_THREADS, 0, s>>>(P, order, tiles_touched,
                                                                                                tt_sorted);
}

void sgr_launch_duplicate(int P, const SgrGeomView& gv, const uint32_t* order, const uint32_t* offs_incl, uint32_t* keys,
                          uint32_t* vals, int gx, hipStream_t s) {
    if (P <= 0) return;
    sgr_duplicate_kernel<<<(P + SGR_PRE_THREADS - 1) / SGR_PRE_THREADS, SGR_PRE_THREADS, 0, s>>>(P, gv, order, offs_incl, keys,
                                                                                             vals, gx);
}

void sgr_launch_tile_ranges(int L, const uint32_t* keys, uint2* ranges, uint8_t* touched, hipStream_t s) {
    if (L <= 0) return;
    sgr_tile_ranges_kernel<<<(L + 255) / 256, 256, 0, s>>>(L, keys, ranges, touched);
}

void sgr_launch_compose_keys(int L, const uint32_t* tile_keys, const uint32_t* point_list, const float4* rec, uint64_t* out,
                             hipStream_t s) {
    if (L <= 0) return;
    sgr_compose_keys_kernel<<<(L + 255) / 256, 256, 0, s>>>(L, tile_keys, point_list, rec, out);
}
