// sgr_gauss_bwd.hip -- per-Gaussian backward on gfx950: K12 (computeCov2DCUDA) + K13
// (preprocessCUDA backward) of the reference (backward.cu:144-274, 346-412) fused with the
// reduction of the per-(tile, instance) partial rows produced by sgr_blend_bwd.hip.
//
// One Gaussian per lane.  The rows of Gaussian g are contiguous: partials[offs_excl .. +tiles_touched),
// 64-B aligned, summed in ascending tile order -> deterministic gradients.  Every output element is
// written exactly once (zeros for culled Gaussians and for SH coefficients above the active degree),
// so the caller allocates the gradient tensors uninitialised instead of the reference's eleven
// torch::zeros (rasterize_points.cu:166-176).
#include "sgr_math.h"

#include <cstdlib>

#ifndef SGR_WITH_VARIANTS
#define SGR_WITH_VARIANTS 0  // 1: also build the designs that were measured slower and kept as A/Bs (tools/build_variant.py)
#endif

#ifndef SGR_GB_THREADS
#define SGR_GB_THREADS 256
#endif
#ifndef SGR_RS_LANES
#define SGR_RS_LANES 4  // lanes that share one Gaussian's partial rows in the row-sum stage (4 or 8)
#endif

// ---- stage 1: sum the partial rows ---------------------------------------------------------------------------
// A latency-bound gather (flag byte -> 48..176-byte row; n rows per Gaussian, n between 1 and hundreds), so it lives
// in its own kernel with few registers (8 waves / SIMD; the fused kernel had 4 because of the SH arrays) and FOUR
// LANES PER GAUSSIAN: lane q of a quad sums rows q, q+4, q+8, ... (two in flight), then the quad's four partial sums
// are combined with two quad_perm DPP steps.  The order of the additions is fixed -> deterministic.
// Writes the outputs that come straight from the blend backward (backward.cu:568-638) and hands the conic / depth
// terms to stage 2 through `cd`.
// QUAD: the rows of the scalar-walk blend backward (sgr_blend_bwd_sw.hip) -- FOUR rows per (tile, instance), one per
// quadrant, `touched` = mask of the ones that exist, each holding the raw moments [S gx, S gy, S abs, S gxx, S gxy, S gyy,
// S Gd, r, g, b, depth]; summed in (tile, quadrant) order, and the moments -> gradient step of the LDS kernel's flush
// (a product with the Gaussian's staged conic and opacity, backward.cu:616-635) applied ONCE, to the sums.
// kx, ky = 0.5 * W / lsc, 0.5 * H / lsc with lsc = log2(e), or 1 in parity mode (`exact`: the staged conic is then
// (-0.5 cx, -cy, -0.5 cz) instead of the log2(e)-scaled one in rec[3]).
template <int SMAX, bool QUAD>
__global__ void __launch_bounds__(SGR_GB_THREADS)
sgr_row_sum_kernel(int P, int S, const int* __restrict__ radii, SgrGeomView gv, const float* __restrict__ partials,
                   int row_stride, const uint8_t* __restrict__ touched, float* __restrict__ dL_dmean2D,
                   float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolor, float* __restrict__ dL_dsemantic,
                   float4* __restrict__ cd, SgrStatSink sink, float kx, float ky, int exact, uint32_t row_limit,
                   float* __restrict__ masked_out) {
    constexpr int NV = (SGR_ROW_BASE_N + SMAX + 3) / 4;
    const int gtid = blockIdx.x * SGR_GB_THREADS + threadIdx.x;
    const int idx = gtid / SGR_RS_LANES, q = gtid % SGR_RS_LANES;
    float acc[4 * NV];
#pragma unroll
    for (int k = 0; k < 4 * NV; k++) acc[k] = 0.f;
    const bool live = idx < P;  // P need not be a multiple of 16: keep whole quads alive for the DPP steps
    // tiles_touched is 0 for a culled Gaussian, so the row walk needs no look at the radius: the two loads below go out
    // together and the dependent chain is {u0, n} -> flag -> row
    uint32_t n = live ? gv.aux[idx].x : 0u;
    const uint32_t u0 = live ? gv.u0[idx] : 0u;  // defined for every Gaussian (exclusive scan)
    // rows past the array's end exist only after a lazy forward that overflowed its list capacity (sgr_set_lazy): the
    // blend backward did not write them and they are not read (that frame's results are discarded, one call later)
    n = min(n, row_limit - min(u0, row_limit));
    if (n && QUAD) {
        const uint8_t* flag = touched + u0;
        const float* rows = partials + (size_t)u0 * 4u * row_stride;
        for (uint32_t i = (uint32_t)q; i < n; i += SGR_RS_LANES) {
            const uint32_t h = flag[i];
            const float4* r = reinterpret_cast<const float4*>(rows + (size_t)i * 4u * row_stride);
            float4 t[4][NV];
#pragma unroll
            for (int b = 0; b < 4; b++)  // the (up to four) rows of the instance go out together
                if ((h >> b) & 1u) {
#pragma unroll
                    for (int k4 = 0; k4 < NV; k4++) t[b][k4] = r[b * (row_stride / 4) + k4];
                }
#pragma unroll
            for (int b = 0; b < 4; b++)
                if ((h >> b) & 1u) {
#pragma unroll
                    for (int k4 = 0; k4 < NV; k4++) {
                        acc[4 * k4] += t[b][k4].x; acc[4 * k4 + 1] += t[b][k4].y; acc[4 * k4 + 2] += t[b][k4].z; acc[4 * k4 + 3] += t[b][k4].w;
                    }
                }
        }
    } else if (n) {
        const uint8_t* flag = touched + u0;
        const float* rows = partials + (size_t)u0 * row_stride;
        auto add_row = [&](const float4 (&t)[NV]) __attribute__((always_inline)) {
#pragma unroll
            for (int k4 = 0; k4 < NV; k4++) {
                acc[4 * k4] += t[k4].x; acc[4 * k4 + 1] += t[k4].y; acc[4 * k4 + 2] += t[k4].z; acc[4 * k4 + 3] += t[k4].w;
            }
        };
        uint32_t i = (uint32_t)q;
        for (; i + SGR_RS_LANES < n; i += 2 * SGR_RS_LANES) {
            const uint8_t f0 = flag[i], f1 = flag[i + SGR_RS_LANES];
            float4 t0[NV], t1[NV];
            // rows never written by the blend backward hold garbage: load under the flag
            if (f0) { const float4* r = reinterpret_cast<const float4*>(rows + (size_t)i * row_stride);
#pragma unroll
                for (int k4 = 0; k4 < NV; k4++) t0[k4] = r[k4]; }
            if (f1) { const float4* r = reinterpret_cast<const float4*>(rows + (size_t)(i + SGR_RS_LANES) * row_stride);
#pragma unroll
                for (int k4 = 0; k4 < NV; k4++) t1[k4] = r[k4]; }
            if (f0) add_row(t0);
            if (f1) add_row(t1);
        }
        if (i < n && flag[i]) {
            const float4* r = reinterpret_cast<const float4*>(rows + (size_t)i * row_stride);
            float4 t[NV];
#pragma unroll
            for (int k4 = 0; k4 < NV; k4++) t[k4] = r[k4];
            add_row(t);
        }
    }
    // (q0 + q1) + (q2 + q3) in every lane of the quad
#pragma unroll
    for (int k = 0; k < 4 * NV; k++) {
        acc[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc[k]), 0xB1, 0xF, 0xF, false));
        acc[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc[k]), 0x4E, 0xF, 0xF, false));
        if (SGR_RS_LANES == 8)  // row_half_mirror: lane i <-> 7 - i, i.e. the other quad of the 8-lane group
            acc[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc[k]), 0x141, 0xF, 0xF, false));
    }
    if (QUAD && n) {
        // moments -> gradients (sgr_blend_bwd.hip's flush, per Gaussian instead of per (tile, instance))
#pragma clang fp contract(off)
        const float4 c1 = gv.rec[4 * (size_t)idx + 1], c3 = gv.rec[4 * (size_t)idx + 3];
        const float qx = exact ? -0.5f * c1.x : c3.x, qy = exact ? -c1.y : c3.z, qz = exact ? -0.5f * c1.z : c3.w;
        const float qw = c1.w, hq = -0.5f * c1.w;
        const float sgx = acc[0], sgy = acc[1];
        acc[0] = qw * kx * fmaf(qx + qx, sgx, qy * sgy);  // dL/dmean2D.x
        acc[1] = qw * ky * fmaf(qz + qz, sgy, qy * sgx);  // dL/dmean2D.y
        acc[2] = qw * acc[2];                             // sum |gx| + |gy|
        acc[3] = hq * acc[3];                             // dL/dconic.x
        acc[4] = hq * acc[4];                             // dL/dconic.y
        acc[5] = hq * acc[5];                             // dL/dconic.w
    }
    if (!live) return;
    // the quad shares the stores
    if (q == 0) {
        dL_dmean2D[3 * idx + 0] = acc[0];
        dL_dmean2D[3 * idx + 1] = acc[1];
        dL_dmean2D[3 * idx + 2] = acc[2];
        // densification statistics of this view (set_max_radii2D + add_densification_stats,
        // street_gaussian_model.py:551-571) while dL/dmean2D and the radius are in registers: visibility = radii > 0
        const int r = radii[idx];
        if (sink.accum != nullptr && r > 0) {
#pragma clang fp contract(off)
            // persistent row of this Gaussian: identity, or through the frame's segment map (binary search over the
            // segment starts, kernel arguments -> scalar loads)
            long row = idx;
            if (sink.nseg > 0) {
                int lo = 0, hi = sink.nseg - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (sink.start[mid] <= idx) lo = mid; else hi = mid - 1;
                }
                row = (idx >= sink.start[lo] && idx < sink.start[lo] + sink.count[lo]) ? (long)idx + sink.shift[lo] : -1;
            }
            if (row >= 0) {
                sink.accum[2 * (size_t)row] += sqrtf(acc[0] * acc[0] + acc[1] * acc[1]);  // norm of grad[:, :2]
                sink.accum[2 * (size_t)row + 1] += fabsf(acc[2]);                          // norm of grad[:, 2:]
                sink.denom[row] += 1.0f;
                sink.max_radii[row] = fmaxf(sink.max_radii[row], (float)r);
            }
        }
    } else if (q == 1) {
        dL_dopacity[idx] = acc[6];
        dL_dcolor[3 * idx + 0] = acc[7];
        dL_dcolor[3 * idx + 1] = acc[8];
        dL_dcolor[3 * idx + 2] = acc[9];
        if (masked_out != nullptr) {
            // sgr_backward_extras.masked_color_out: the colour gradient that reaches the SH coefficients (backward.cu:40-44:
            // channels the forward clamped at zero get none) -- the payload of the view-sharded exchange, written here
            // instead of by a launch of its own (sgr_masked_color_grad)
            const uint32_t cl = n ? gv.clamped[idx] : 0u;  // (culled Gaussians: acc = 0 anyway, `clamped` is not written for them)
            masked_out[3 * idx + 0] = (cl & 1u) ? 0.f : acc[7];
            masked_out[3 * idx + 1] = (cl & 2u) ? 0.f : acc[8];
            masked_out[3 * idx + 2] = (cl & 4u) ? 0.f : acc[9];
        }
    } else if (q == 2) {
        cd[idx] = make_float4(acc[3], acc[4], acc[5], acc[10]);
    }
    if (SMAX > 0) {
#pragma unroll
        for (int ch = 0; ch < SMAX; ch++)
            if (ch < S && (ch & 3) == (q & 3) && q < 4) dL_dsemantic[(size_t)idx * S + ch] = acc[SGR_ROW_BASE_N + ch];
    }
}

#if SGR_WITH_VARIANTS  // rejected A/B design, built only by tools/build_variant.py (-DSGR_WITH_VARIANTS=1)
// ---- stage 1, wave-cooperative form (round 4; A/B behind SGR_RS_WAVE=1: measured slower, see the launcher) ----------------
// The rows are in INDEX order (u0 = exclusive scan of tiles_touched over the Gaussians, culled ones contributing none), so
// the 64 Gaussians of a wave own ONE contiguous row range [ua, ub).  The wave streams it 64 rows at a time -- lane l takes
// row cbase + l: one coalesced flag-byte load, then (flag set) its row as float4s, 48..176 contiguous bytes per lane and a
// contiguous run for the wave -- instead of every quad of lanes chasing its own Gaussian's {u0, n} -> flag -> row chain.
// Which Gaussian a row belongs to: the wave's non-empty Gaussians are ranked (ballot), a Gaussian whose first row lies in
// the chunk marks that position, and an inclusive max-scan over the lanes (carried from chunk to chunk) spreads the ranks.
// The rows of a chunk are then summed per owner with a segmented Hillis-Steele scan across the lanes (six steps, a FIXED
// tree: deterministic), and the last lane of every segment adds the segment's sum to its Gaussian's accumulator in LDS
// (one writer per accumulator and chunk; chunks in order).  Same outputs as sgr_row_sum_kernel; the additions of a
// Gaussian's rows happen in another (fixed) order.
#ifndef SGR_RSW_WAVES
#define SGR_RSW_WAVES 4
#endif
template <int SMAX>
__global__ void __launch_bounds__(64 * SGR_RSW_WAVES)
sgr_row_sum_wave_kernel(int P, int S, const int* __restrict__ radii, SgrGeomView gv, const float* __restrict__ partials,
                        int row_stride, const uint8_t* __restrict__ touched, float* __restrict__ dL_dmean2D,
                        float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolor, float* __restrict__ dL_dsemantic,
                        float4* __restrict__ cd, SgrStatSink sink) {
    constexpr int NV = (SGR_ROW_BASE_N + SMAX + 3) / 4, NF = 4 * NV;
    constexpr int NVP = NV + ((NV & 1) ? 0 : 1);  // odd number of float4s per accumulator: rows spread over the banks
    __shared__ float4 sAcc[SGR_RSW_WAVES][64][NVP];
    __shared__ int sMark[SGR_RSW_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = (blockIdx.x * SGR_RSW_WAVES + wave) * 64 + lane;
    const bool live = g < P;
    const uint32_t n = live ? gv.aux[g].x : 0u;
    const uint32_t u0 = live ? gv.u0[g] : 0u;
    const uint64_t nz = sgr_uniform_u64(__ballot(n != 0u));
    const int rank = __popcll(nz & ((1ull << lane) - 1ull));
    float acc[NF];
#pragma unroll
    for (int k = 0; k < NF; k++) acc[k] = 0.f;
    if (nz != 0ull) {  // wave-uniform
        const int first = __builtin_ctzll(nz), last = 63 - __builtin_clzll(nz);
        const uint32_t ua = (uint32_t)__builtin_amdgcn_readlane((int)u0, first);
        const uint32_t ub = (uint32_t)__builtin_amdgcn_readlane((int)(u0 + n), last);
#pragma unroll
        for (int k4 = 0; k4 < NV; k4++) sAcc[wave][lane][k4] = make_float4(0.f, 0.f, 0.f, 0.f);
        int kprev = 0;
        for (uint32_t cbase = ua; cbase < ub; cbase += 64u) {
            // owner rank of every row of the chunk
            sMark[wave][lane] = -1;
            __builtin_amdgcn_wave_barrier();
            if (n != 0u && u0 >= cbase && u0 - cbase < 64u) sMark[wave][u0 - cbase] = rank;
            __builtin_amdgcn_wave_barrier();
            int k = sMark[wave][lane];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(k, d, 64);
                if (lane >= d) k = max(k, t);
            }
            k = max(k, kprev);
            kprev = __builtin_amdgcn_readlane(k, 63);
            // this lane's row
            const uint32_t u = cbase + (uint32_t)lane;
            const bool valid = u < ub;
            float val[NF];
#pragma unroll
            for (int i = 0; i < NF; i++) val[i] = 0.f;
            if (valid && touched[u]) {  // rows the blend backward did not write hold garbage
                const float4* r = reinterpret_cast<const float4*>(partials + (size_t)u * row_stride);
#pragma unroll
                for (int k4 = 0; k4 < NV; k4++) {
                    const float4 t = r[k4];
                    val[4 * k4] = t.x; val[4 * k4 + 1] = t.y; val[4 * k4 + 2] = t.z; val[4 * k4 + 3] = t.w;
                }
            }
            // segmented inclusive scan over the lanes (segments = runs of equal owner rank)
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int ko = __shfl_up(k, d, 64);
                const bool same = lane >= d && ko == k;
#pragma unroll
                for (int i = 0; i < NF; i++) {
                    const float t = __shfl_up(val[i], d, 64);
                    val[i] += same ? t : 0.f;
                }
            }
            const int knext = __shfl_down(k, 1, 64);
            const bool tail = valid && (lane == 63 || knext != k || u + 1u >= ub);
            if (tail) {
                float4* a = sAcc[wave][k];
#pragma unroll
                for (int k4 = 0; k4 < NV; k4++) {
                    float4 t = a[k4];
                    t.x += val[4 * k4]; t.y += val[4 * k4 + 1]; t.z += val[4 * k4 + 2]; t.w += val[4 * k4 + 3];
                    a[k4] = t;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (n != 0u) {
#pragma unroll
            for (int k4 = 0; k4 < NV; k4++) {
                const float4 t = sAcc[wave][rank][k4];
                acc[4 * k4] = t.x; acc[4 * k4 + 1] = t.y; acc[4 * k4 + 2] = t.z; acc[4 * k4 + 3] = t.w;
            }
        }
    }
    if (!live) return;
    const int idx = g;
    dL_dmean2D[3 * idx + 0] = acc[0];
    dL_dmean2D[3 * idx + 1] = acc[1];
    dL_dmean2D[3 * idx + 2] = acc[2];
    dL_dopacity[idx] = acc[6];
    dL_dcolor[3 * idx + 0] = acc[7];
    dL_dcolor[3 * idx + 1] = acc[8];
    dL_dcolor[3 * idx + 2] = acc[9];
    cd[idx] = make_float4(acc[3], acc[4], acc[5], acc[10]);
    if (SMAX > 0) {
#pragma unroll
        for (int ch = 0; ch < SMAX; ch++)
            if (ch < S) dL_dsemantic[(size_t)idx * S + ch] = acc[SGR_ROW_BASE_N + ch];
    }
    // densification statistics of this view (see sgr_row_sum_kernel)
    const int r = radii[idx];
    if (sink.accum != nullptr && r > 0) {
#pragma clang fp contract(off)
        long row = idx;
        if (sink.nseg > 0) {
            int lo = 0, hi = sink.nseg - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (sink.start[mid] <= idx) lo = mid; else hi = mid - 1;
            }
            row = (idx >= sink.start[lo] && idx < sink.start[lo] + sink.count[lo]) ? (long)idx + sink.shift[lo] : -1;
        }
        if (row >= 0) {
            sink.accum[2 * (size_t)row] += sqrtf(acc[0] * acc[0] + acc[1] * acc[1]);
            sink.accum[2 * (size_t)row + 1] += fabsf(acc[2]);
            sink.denom[row] += 1.0f;
            sink.max_radii[row] = fmaxf(sink.max_radii[row], (float)r);
        }
    }
}

#endif  // SGR_WITH_VARIANTS

// ---- stage 2: K12 + K13 ----------------------------------------------------------------------------------------
#ifndef SGR_GB_WAVES
#define SGR_GB_WAVES 1
#endif
__global__ void __launch_bounds__(SGR_GB_THREADS) __attribute__((amdgpu_waves_per_eu(SGR_GB_WAVES)))
sgr_gauss_bwd_kernel(int P, int D, int M, const float* __restrict__ means3D, const int* __restrict__ radii,
                     const float* __restrict__ shs, const float* __restrict__ scales,
                     const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,
                     const SgrCam* __restrict__ camp, SgrGeomView gv, const float4* __restrict__ cd,
                     const float* __restrict__ dL_dmean2D, const float* __restrict__ dL_dcolor,
                     float* __restrict__ dL_dmean3D, float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh,
                     float* __restrict__ dL_dscale, float* __restrict__ dL_drot, int skip_sh) {
    const SgrCam& cam = *camp;
    // lanes past P stay alive (they help with the cooperative SH copies) on a clamped index; their stores are masked
    const int gidx = blockIdx.x * SGR_GB_THREADS + threadIdx.x;
    const bool live = gidx < P;
    const int idx = live ? gidx : P - 1;
    const bool visible = live && radii[idx] > 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // SH rows of 48 floats (M = 16): the 64 rows of a wave are one contiguous 12 KB block.  Per-lane float4 loads at
    // a 192-byte stride touch 64 cache lines per instruction and were issue-bound (SQ_WAIT_INST_ANY 70 % of the
    // wave-cycles); instead the wave copies the block with coalesced 1 KB transfers through LDS, both ways.
    // Half of the wave's rows at a time (6 KB per wave): with all 64 rows resident the 48 KB per workgroup allowed 3
    // workgroups = 3 waves / SIMD on a kernel that streams 0.5 KB per Gaussian; the lanes whose rows are in LDS read
    // them into registers before the other half arrives.
    __shared__ float4 sSH[SGR_GB_THREADS / 64][32 * 12];
    const bool stage = shs != nullptr && M == 16;
    const int g0 = blockIdx.x * SGR_GB_THREADS + wave * 64;
    const int nrow4 = max(0, min(64, P - g0)) * 12;  // float4s of this wave's rows
    const int ncoef = (D + 1) * (D + 1);
    float acc[SGR_ROW_BASE_N];
#pragma unroll
    for (int k = 0; k < SGR_ROW_BASE_N; k++) acc[k] = 0.f;
    if (visible) {
        const float4 c = cd[idx];
        acc[0] = dL_dmean2D[3 * idx];
        acc[1] = dL_dmean2D[3 * idx + 1];
        acc[3] = c.x; acc[4] = c.y; acc[5] = c.z; acc[10] = c.w;
        acc[7] = dL_dcolor[3 * idx];
        acc[8] = dL_dcolor[3 * idx + 1];
        acc[9] = dL_dcolor[3 * idx + 2];
    }

    float dmean[3] = {0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dscale[3] = {0.f, 0.f, 0.f};
    float drot[4] = {0.f, 0.f, 0.f, 0.f};
    const float p[3] = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    float dRGB[3] = {0.f, 0.f, 0.f};
    float dir[3] = {0.f, 0.f, 1.f}, dir_orig[3] = {0.f, 0.f, 1.f};

    if (visible) {
        // K12: conic -> cov3D and the covariance part of dL/dmean (assigned)
        float cov3D[6];
        float sc[3] = {0.f, 0.f, 0.f}, rot[4] = {0.f, 0.f, 0.f, 0.f};
        if (scales != nullptr) {
            sc[0] = scales[3 * idx]; sc[1] = scales[3 * idx + 1]; sc[2] = scales[3 * idx + 2];
            const float4 q = *reinterpret_cast<const float4*>(rotations + 4 * (size_t)idx);
            rot[0] = q.x; rot[1] = q.y; rot[2] = q.z; rot[3] = q.w;
        }
        if (cov3D_precomp) {
#pragma unroll
            for (int i = 0; i < 6; i++) cov3D[i] = cov3D_precomp[6 * (size_t)idx + i];
        } else {
            // the forward does not store cov3D (24 B/Gaussian written there and read here): same function, same
            // inputs, FP contraction off inside it -> the forward's value bit for bit (forward.cu:115-150)
            sgr_cov3d(sc, cam.scale_modifier, rot, cov3D);
        }
        sgr_cov2d_backward(p, cov3D, cam, acc[3], acc[4], acc[5], dcov, dmean);
        // K13: projection + depth paths (added)
        sgr_proj_depth_backward(p, cam, acc[0], acc[1], acc[10], dmean);
        if (scales != nullptr) sgr_cov3d_backward(sc, cam.scale_modifier, rot, dcov, dscale, drot);
        if (shs != nullptr) {
            const uint32_t cl = gv.clamped[idx];
            dRGB[0] = (cl & 1u) ? 0.f : acc[7];
            dRGB[1] = (cl & 2u) ? 0.f : acc[8];
            dRGB[2] = (cl & 4u) ? 0.f : acc[9];
            dir_orig[0] = p[0] - cam.campos[0];
            dir_orig[1] = p[1] - cam.campos[1];
            dir_orig[2] = p[2] - cam.campos[2];
            const float len = sqrtf(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
            dir[0] = dir_orig[0] / len;
            dir[1] = dir_orig[1] / len;
            dir[2] = dir_orig[2] / len;
        }
    }

    if (shs != nullptr) {
        // dL/dSH[k] = Y_k(dir) * dL/dRGB for k < (D+1)^2, zero above (backward.cu:46-105); the
        // view-direction path adds dnormvdv(dir_orig, dL/ddir) to dL/dmean (backward.cu:131-138).
        float* dsh = dL_dsh + (size_t)idx * M * 3;
        const bool vec = ((M * 3) & 3) == 0 && M <= 16;  // 16-byte aligned rows: stream them as float4
        const int n4 = (ncoef * 3 + 3) >> 2;
        // t[k] = sum_ch sh[3k + ch] * dL/dRGB[ch]: the row contracted with the colour gradient as it arrives, 16 live
        // values instead of the 48 of the row (sgr_sh_dir_backward)
        float t[16];
#pragma unroll
        for (int k = 0; k < 16; k++) t[k] = 0.f;
        auto take4 = [&](int i, const float4& v) __attribute__((always_inline)) {
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; j++)
                if ((4 * i + j) / 3 < ncoef) t[(4 * i + j) / 3] += e[j] * dRGB[(4 * i + j) % 3];
        };
        if (stage) {
            const uint64_t vis = __ballot(visible);
            const float4* src = reinterpret_cast<const float4*>(shs) + (size_t)g0 * 12;
#pragma unroll
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int it = 0; it < 6; it++) {
                    const int f = h * 384 + it * 64 + lane;
                    if (f < nrow4 && ((vis >> (f / 12)) & 1ull)) sSH[wave][f - h * 384] = src[f];
                }
                __builtin_amdgcn_wave_barrier();
                if ((lane >> 5) == h && visible) {
#pragma unroll
                    for (int i = 0; i < 12; i++)
                        if (i < n4) take4(i, sSH[wave][(lane & 31) * 12 + i]);
                }
                __builtin_amdgcn_wave_barrier();  // consumed: the buffer may be overwritten
            }
        }
        float Y[16];
#pragma unroll
        for (int k = 0; k < 16; k++) Y[k] = 0.f;
        if (visible) {
            sgr_sh_basis(D, dir[0], dir[1], dir[2], Y);
            const float* sh = shs + (size_t)idx * M * 3;
            if (stage) {
                // contracted above
            } else if (vec) {
                const float4* sh4 = reinterpret_cast<const float4*>(sh);
#pragma unroll
                for (int i = 0; i < 12; i++)
                    if (i < n4) take4(i, sh4[i]);
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++)
                    if (k < ncoef) t[k] = sh[3 * k] * dRGB[0] + sh[3 * k + 1] * dRGB[1] + sh[3 * k + 2] * dRGB[2];
            }
            float ddir[3], dm[3];
            sgr_sh_dir_backward(D, dir[0], dir[1], dir[2], t, ddir);
            sgr_dnormvdv(dir_orig, ddir, dm);
            dmean[0] += dm[0]; dmean[1] += dm[1]; dmean[2] += dm[2];
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (k >= ncoef) Y[k] = 0.f;
        }
        // dL/dSH row: Y_k * dL/dRGB below the active degree, zeros above and for culled Gaussians (Y = 0, dRGB = 0) --
        // element e of the row is Y[e / 3] * dRGB[e % 3], formed where it is stored instead of held in 48 registers
        auto elem4 = [&](int i) __attribute__((always_inline)) {
            return make_float4(Y[(4 * i) / 3] * dRGB[(4 * i) % 3], Y[(4 * i + 1) / 3] * dRGB[(4 * i + 1) % 3],
                               Y[(4 * i + 2) / 3] * dRGB[(4 * i + 2) % 3], Y[(4 * i + 3) / 3] * dRGB[(4 * i + 3) % 3]);
        };
        if (skip_sh) {
            // sgr_backward_extras.skip_sh_grad: the factored exchange rebuilds dL/dSH of all views (sgr_sh_grad_from_views)
        } else if (stage) {
            float4* dst = reinterpret_cast<float4*>(dL_dsh) + (size_t)g0 * 12;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if ((lane >> 5) == h) {
#pragma unroll
                    for (int i = 0; i < 12; i++) sSH[wave][(lane & 31) * 12 + i] = elem4(i);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 6; it++) {
                    const int f = h * 384 + it * 64 + lane;
                    if (f < nrow4) dst[f] = sSH[wave][f - h * 384];
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else if (!live) {
        } else if (vec) {
            float4* dsh4 = reinterpret_cast<float4*>(dsh);
            const int n4 = (M * 3) >> 2;
#pragma unroll
            for (int i = 0; i < 12; i++)
                if (i < n4) dsh4[i] = elem4(i);
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (k < M) { dsh[3 * k] = Y[k] * dRGB[0]; dsh[3 * k + 1] = Y[k] * dRGB[1]; dsh[3 * k + 2] = Y[k] * dRGB[2]; }
            for (int k = 48; k < M * 3; k++) dsh[k] = 0.f;
        }
    }

    if (!live) return;
    dL_dmean3D[3 * idx + 0] = dmean[0];
    dL_dmean3D[3 * idx + 1] = dmean[1];
    dL_dmean3D[3 * idx + 2] = dmean[2];
#pragma unroll
    for (int i = 0; i < 6; i++) dL_dcov3D[6 * (size_t)idx + i] = dcov[i];
    dL_dscale[3 * idx + 0] = dscale[0];
    dL_dscale[3 * idx + 1] = dscale[1];
    dL_dscale[3 * idx + 2] = dscale[2];
    *reinterpret_cast<float4*>(dL_drot + 4 * (size_t)idx) = make_float4(drot[0], drot[1], drot[2], drot[3]);
}

// returns non-zero when `after_rows` could not be recorded (everything is launched regardless)
int sgr_launch_gauss_bwd(int P, int D, int M, int S, const float* means3D, const int* radii, const float* shs,
                          const float* scales, const float* rotations, const float* cov3D_precomp, const SgrCam* cam,
                          const SgrGeomView& gv, const float* partials, int row_stride, const uint8_t* touched,
                          float4* cd, float* dL_dmean2D, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                          float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dsemantic,
                          const SgrStatSink& sink, int quad, int exact, int W, int H, hipEvent_t after_rows, int rs_wave,
                          uint32_t row_limit, float* masked_color_out, int skip_sh, hipStream_t s) {
    if (P <= 0) return 0;
    const float lsc = exact ? 1.0f : SGR_LOG2E;
    const float kx = (0.5f * (float)W) / lsc, ky = (0.5f * (float)H) / lsc;
    const unsigned nb = (P + SGR_GB_THREADS - 1) / SGR_GB_THREADS;
    const unsigned nb4 = (unsigned)(((size_t)P * SGR_RS_LANES + SGR_GB_THREADS - 1) / SGR_GB_THREADS);
#define SGR_RS(N)                                                                                                    \
    sgr_row_sum_kernel<N, false><<<nb4, SGR_GB_THREADS, 0, s>>>(P, S, radii, gv, partials, row_stride, touched, dL_dmean2D, \
                                                               dL_dopacity, dL_dcolor, dL_dsemantic, cd, sink, kx, ky, exact, row_limit, \
                                                               masked_color_out)
    // switch bit 9 / SGR_RS_WAVE=1: the wave-cooperative row sum instead of the four-lanes-per-Gaussian one (A/B: measured SLOWER on MI355X --
    // per-Gaussian backward stage 0.258 vs 0.213 ms at 1 M Gaussians, 1.14 vs 0.76 ms at 5 M, 0.83 vs 0.57 ms at 2 M + 19
    // channels: its segmented scan is 13 ds_bpermute per step and row chunk, more than the gather chains it removes)
    // (both A/B forms exist only in a -DSGR_WITH_VARIANTS=1 build: tools/build_variant.py)
#if SGR_WITH_VARIANTS
    const bool quads = !rs_wave;  // (sgr_test_switches bit 9)
    const unsigned nbw = (unsigned)((P + 64 * SGR_RSW_WAVES - 1) / (64 * SGR_RSW_WAVES));
#define SGR_RSW(N)                                                                                                   \
    sgr_row_sum_wave_kernel<N><<<nbw, 64 * SGR_RSW_WAVES, 0, s>>>(P, S, radii, gv, partials, row_stride, touched, dL_dmean2D, \
                                                                 dL_dopacity, dL_dcolor, dL_dsemantic, cd, sink)
    if (quad) {  // the scalar-walk blend backward's rows (S = 0 only)
        sgr_row_sum_kernel<0, true><<<nb4, SGR_GB_THREADS, 0, s>>>(P, S, radii, gv, partials, row_stride, touched, dL_dmean2D,
                                                                  dL_dopacity, dL_dcolor, dL_dsemantic, cd, sink, kx, ky, exact, 0xffffffffu, masked_color_out);
    } else if (!quads) {
        if (S == 0) SGR_RSW(0);
        else if (S <= 4) SGR_RSW(4);
        else if (S <= 8) SGR_RSW(8);
        else if (S <= 12) SGR_RSW(12);
        else if (S <= 16) SGR_RSW(16);
        else if (S <= 20) SGR_RSW(20);
        else if (S <= 24) SGR_RSW(24);
        else SGR_RSW(32);
    } else
#undef SGR_RSW
#else
    (void)quad; (void)rs_wave;
#endif
    if (S == 0) SGR_RS(0);
    else if (S <= 4) SGR_RS(4);
    else if (S <= 8) SGR_RS(8);
    else if (S <= 12) SGR_RS(12);
    else if (S <= 16) SGR_RS(16);
    else if (S <= 20) SGR_RS(20);
    else if (S <= 24) SGR_RS(24);
    else SGR_RS(32);
#undef SGR_RS
    // dL/dmean2D, dL/dopacity, dL/dcolour are final from here on.  A failed record must not pass silently: the reducer's side
    // stream would wait on a stale record and read the colour gradient unsynchronised -- the error is picked up by the
    // failure is returned to sgr_backward_ex, which reports SGR_E_HIP
    const bool ev_failed = after_rows && hipEventRecord(after_rows, s) != hipSuccess;
    sgr_gauss_bwd_kernel<<<nb, SGR_GB_THREADS, 0, s>>>(P, D, M, means3D, radii, shs, scales, rotations, cov3D_precomp, cam,
                                                       gv, cd, dL_dmean2D, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh,
                                                       dL_dscale, dL_drot, skip_sh);
    return ev_failed ? 1 : 0;
}
