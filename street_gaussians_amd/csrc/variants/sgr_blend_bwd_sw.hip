// sgr_blend_bwd_sw.hip -- K11 (backward.cu:415-641), second walk design: the SCALAR WALK (S = 0).
//
// sgr_blend_bwd.hip stages a tile's list through LDS 128 entries at a time: four waves share the staged records and the
// LDS combine rows, so every round ends at a barrier -- the counters put 44 % of the wave-cycles at s_waitcnt / s_barrier
// (a round ends when the tile's busiest quadrant is done: 21 % skew; every record read is an LDS round trip).  Here a
// wave owns its 8x8 quadrant END TO END and never meets another wave:
//   * the forward's hit record says which list entries the quadrant blended; the wave scans it 64 entries at a time (one
//     coalesced byte + dword load per lane, prefetched one chunk ahead) and a ballot turns it into a scalar bit mask;
//   * for every set bit the Gaussian's 64-byte record is fetched with ONE s_load_dwordx16 through the scalar cache into
//     SGPRs -- wave-uniform operands cost no VGPR, no LDS and no VALU issue slot -- double-buffered, so the next
//     record's fetch is in flight under the current visit (SMEM returns out of order: the two register sets are waited
//     for alternately with an explicit s_waitcnt tied to the destination registers);
//   * the per-pixel mathematics is the LDS kernel's (factored channel recurrence, moments of G * dL/dalpha, reduce-scatter
//     over the wave with the folded DPP row stage); the 11 sums of a visit leave the wave as ONE 12-lane store into the
//     row of (instance, QUADRANT) -- four rows per instance, written at most once each: no LDS combine, no atomics, no
//     barrier, and the moments -> gradient step (a product with the instance's conic and opacity) moves to the
//     per-Gaussian row sum, once per Gaussian instead of once per (tile, instance);
//   * which rows exist is the hit record's business: `touched[u]` becomes a 4-bit mask, set with a fire-and-forget OR
//     (order-free) by the lanes that scanned the entry.
// Bit-reproducible like the LDS kernel: every float sum has a fixed order (lanes of a wave by the reduction tree, rows of
// a Gaussian by (tile, quadrant) in the row sum).
#include "../sgr_math.h"
#include "../sgr_reduce.h"

typedef int sgr_i16 __attribute__((ext_vector_type(16)));
// element of a record as float.  By VALUE: __builtin_bit_cast applied to a vector-element lvalue (R[i]) reads element 0
// whatever i is (hipcc, ROCm 7.2)
__device__ __forceinline__ float sgr_sf(const int x) { return __builtin_bit_cast(float, x); }

// lane-mask selects (m = all ones or zero): a & m, and (a & m) | (b & ~m) = v_bfi_b32
__device__ __forceinline__ float sgr_and(const float a, const uint32_t m) {
    return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, a) & m);
}
__device__ __forceinline__ float sgr_bfi(const uint32_t m, const float a, const float b) {
    return __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, a) & m) | (__builtin_bit_cast(uint32_t, b) & ~m));
}

// One Gaussian record (64 bytes) into 16 SGPRs: `p` is wave-uniform and `rec` is read-only for the whole kernel, so
// hipcc emits ONE s_load_dwordx16 for this load and keeps track of it itself (the s_waitcnt lgkmcnt(0) goes in front of
// the first use).  (Issuing the load from an inline-asm statement and waiting in a second one does NOT work: the compiler
// takes an asm output as available at once and is free to copy the registers in between -- seen in the ISA.)
__device__ __forceinline__ sgr_i16 sgr_sload_rec(const float4* __restrict__ p) {
    return *reinterpret_cast<const sgr_i16*>(__builtin_assume_aligned(p, 64));
}

// "these values exist now": an empty statement that takes them in and hands them back (nothing upstream of them can be
// scheduled below it) and may touch memory as far as the compiler knows (no load can be hoisted above it)
__device__ __forceinline__ void sgr_pin12(float (&v)[12]) {
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                      "+v"(v[9]), "+v"(v[10]) :: "memory");
}
__device__ __forceinline__ void sgr_pin12a(float (&v)[24]) {
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                      "+v"(v[9]), "+v"(v[10]) :: "memory");
}
__device__ __forceinline__ void sgr_pin12b(float (&v)[24]) {
    asm volatile("" : "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]), "+v"(v[16]), "+v"(v[17]), "+v"(v[18]), "+v"(v[19]),
                      "+v"(v[20]), "+v"(v[21]), "+v"(v[22]) :: "memory");
}
// Keeps ALL sixteen registers of a record allocated up to this point.  The kernel never reads some of its words (the
// cull extents, in the default mode the raw conic): hipcc hands those registers to other values as soon as the load has
// landed -- and has to WAIT for the load in order to do so, in the middle of the work the load was meant to run under.
__device__ __forceinline__ void sgr_keep(const sgr_i16& r) { asm volatile("" ::"s"(r)); }
#ifndef SGR_SW_PREFETCH
#define SGR_SW_PREFETCH 1
#endif
#ifndef SGR_SW_WAVES
#define SGR_SW_WAVES 8
#endif

template <bool EXACT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SGR_SW_WAVES, SGR_SW_WAVES)))
sgr_blend_bwd_sw_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int gx, int gy,
                        const float* __restrict__ bg_color, const float4* __restrict__ rec, const uint32_t* __restrict__ u0, const uint64_t* __restrict__ tmask,
                        const float* __restrict__ alphas, const uint32_t* __restrict__ n_contrib,
                        const uint8_t* __restrict__ hit4, const float* __restrict__ dL_dpixels,
                        const float* __restrict__ dL_dpixel_depths, const float* __restrict__ dL_dalphas,
                        float* __restrict__ partials, int row_stride, uint32_t* __restrict__ touched32) {
    // every fused multiply-add is written out, as in sgr_blend_bwd.hip
#pragma clang fp contract(off)
    const int lane = threadIdx.x;
    // workgroup = wave = one quadrant.  Workgroup b runs on XCD b % 8: the four quadrants of a tile are b = 32 k + x + 8 q,
    // consecutive dispatches on one XCD, and k -> tile is the supertile order of the other blend kernels
    const uint32_t b = blockIdx.x;
    const uint32_t q = (b >> 3) & 3u;
    uint32_t tx, ty;
    if (!sgr_xcd_tile(((b >> 5) << 3) | (b & 7u), (uint32_t)gx, (uint32_t)gy, tx, ty)) return;
    const uint32_t tile = ty * (uint32_t)gx + tx;
    const uint32_t px = tx * SGR_BLOCK_X + (q & 1u) * 8 + (lane & 7);
    const uint32_t py = ty * SGR_BLOCK_Y + (q >> 1) * 8 + (lane >> 3);
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const float pxf = (float)px, pyf = (float)py;
    const size_t pix_id = (size_t)W * py + px;
    const size_t plane = (size_t)H * W;
    const uint2 range = ranges[tile];

    // backward.cu:466-500
    const float T_final = inside ? (1.0f - alphas[pix_id]) : 0.0f;
    float T = T_final;
    const int lastc = inside ? (int)n_contrib[pix_id] : 0;
    float dLdC0 = 0.f, dLdC1 = 0.f, dLdC2 = 0.f, dLdD = 0.f, dLdA = 0.f;
    if (inside) {
        dLdC0 = dL_dpixels[pix_id];
        dLdC1 = dL_dpixels[plane + pix_id];
        dLdC2 = dL_dpixels[2 * plane + pix_id];
        dLdD = dL_dpixel_depths[pix_id];
        dLdA = dL_dalphas[pix_id];
    }
    const float bgdot = bg_color[0] * dLdC0 + bg_color[1] * dLdC1 + bg_color[2] * dLdC2;
    const bool bg_zero = bg_color[0] == 0.0f && bg_color[1] == 0.0f && bg_color[2] == 0.0f;
    constexpr float LSC = EXACT ? 1.0f : SGR_LOG2E;
    const float kx = (0.5f * (float)W) / LSC, ky = (0.5f * (float)H) / LSC;
    float Arec = 0.f, u_last = 0.f, last_alpha = 0.f;

    // highest list position any pixel of the QUADRANT blended (the LDS kernel takes the tile's)
    int mx = lastc;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
    const int maxc = __builtin_amdgcn_readfirstlane(mx);
    if (maxc == 0) return;

    // Where the sums of a visit leave the wave.  Reduce-scatter layout (sgr_reduce.h): with bank = (lane >> 2) & 3,
    // k = lane >> 4 and perm = {0,2,1,3}, every lane of bank b of g0 holds the wave total of value 4 perm[b] + perm[k]
    // (single visit: values 0-11 are the row; pair: 0-11 first visit, 12-15 the first four values of the second), and
    // banks 0, 2 of g1 hold value 16 + 4 perm[b] + perm[k] of a pair.  One lane per bank stores.
    const int pb = (0x3120 >> (((lane >> 2) & 3) * 4)) & 3, pk = (0x3120 >> ((lane >> 4) * 4)) & 3;
    const bool leader = (lane & 3) == 0;
    const uint32_t vidx = (uint32_t)(4 * pb + pk);                  // value index held in g0
    const bool st_a = leader && vidx < 12;                          // g0 -> row of the first visit, float vidx
    const bool st_b0 = leader && vidx >= 12;                        // g0 -> row of the second visit, float vidx - 12
    const bool st_b1 = leader && ((lane >> 2) & 1) == 0;            // g1 -> row of the second visit, float 4 + vidx
    const uint32_t qbit = 1u << q;

    // per-pixel part of one visit: record R (SGPRs), list position posj (wave-uniform); the 11 per-pixel terms -> v[0..11]
    auto pixel = [&](const sgr_i16& R, const int posj, float* v) __attribute__((always_inline)) {
        const float sx = sgr_sf(R[0]), sy = sgr_sf(R[1]);
        const float opac = sgr_sf(R[7]);
        const float dx = sx - pxf, dy = sy - pyf;
        float qx, qy, qz, pw, G;  // staged conic (what the gradient terms use), power, G
        if (EXACT) {
            const float cx = sgr_sf(R[4]), cy = sgr_sf(R[5]), cz = sgr_sf(R[6]);
            qx = -0.5f * cx; qy = -cy; qz = -0.5f * cz;
            pw = sgr_power_ref_staged(qx, qy, qz, dx, dy);
            G = sgr_expf_ref(pw);
        } else {
            qx = sgr_sf(R[12]); qy = sgr_sf(R[14]); qz = sgr_sf(R[15]);
            pw = sgr_power2(qx, qy, qz, dx, dy);
            G = __builtin_amdgcn_exp2f(pw);
        }
        const float alpha = fminf(0.99f, opac * G);
        // backward.cu:527-545.  No wave-wide early out: the hit record lists the visits that blended (plus the rare one
        // whose every passing pixel had finished on it; its row is flagged by the scan below and written as zeros here)
        const bool hit = (posj < lastc) && !(pw > 0.0f) && !(alpha < SGR_ALPHA_MIN);
        // BRANCH-FREE: every lane runs the update and the per-pixel state is kept with selects.  An `if (hit)` costs
        // nothing less on a SIMD (the other lanes idle) but splits the unit into basic blocks, and then the two visits of a
        // unit cannot be interleaved: measured on MI355X (tools/ubench/valu_rates.hip) a wave's DEPENDENT plain VALU
        // instructions issue every 4.3 cycles even with eight waves per SIMD, independent ones every 2.7.
        float Gd, wm;
        {
            const float oma = 1.0f - alpha;
            float inv1ma = __builtin_amdgcn_rcpf(oma);
            inv1ma = fmaf(fmaf(-oma, inv1ma, 1.0f), inv1ma, inv1ma);
            const float Tn = EXACT ? sgr_div_by(T, oma, inv1ma) : T * inv1ma;  // backward.cu:547
            const float one_m_la = 1.0f - last_alpha;
            const float An = fmaf(last_alpha, u_last, one_m_la * Arec);
            float u = fmaf(sgr_sf(R[8]), dLdC0, dLdA);
            u = fmaf(sgr_sf(R[9]), dLdC1, u);
            u = fmaf(sgr_sf(R[10]), dLdC2, u);
            u = fmaf(sgr_sf(R[11]), dLdD, u);
            const float d = (u - An) * Tn;
            float gd;
            if (EXACT) gd = bg_zero ? G * d : G * (d + sgr_div_by(-T_final, oma, inv1ma) * bgdot);
            else gd = G * fmaf(-T_final * inv1ma, bgdot, d);  // backward.cu:611-614
            // selects as bit operations on an opaque lane mask: written as `hit ? a : b`, hipcc turns the six selects
            // back into one branch over the whole block
            uint32_t hm = hit ? 0xffffffffu : 0u;
            asm volatile("" : "+v"(hm));
            Gd = sgr_and(gd, hm);
            wm = sgr_and(alpha * Tn, hm);
            T = sgr_bfi(hm, Tn, T);
            Arec = sgr_bfi(hm, An, Arec);
            u_last = sgr_bfi(hm, u, u_last);
            last_alpha = sgr_bfi(hm, alpha, last_alpha);
        }
        const float gxm = Gd * dx, gym = Gd * dy;
        float ax, ay;
        {
            const float e1 = fmaf(qy, dy, qx * dx), e2 = qz * dy;
            ax = fmaf(qx, dx, e1);
            ay = fmaf(qy, dx, e2 + e2);
        }
        v[0] = gxm;
        v[1] = gym;
        v[2] = fabsf(Gd) * fmaf(fabsf(ax), kx, fabsf(ay) * ky);
        v[3] = gxm * dx;
        v[4] = gxm * dy;
        v[5] = gym * dy;
        v[6] = Gd;
        v[7] = wm * dLdC0;
        v[8] = wm * dLdC1;
        v[9] = wm * dLdC2;
        v[10] = wm * dLdD;
        v[11] = 0.0f;
    };

    // ---- scan of the hit record, 64 list entries per chunk, back to front; the next chunk's entries are in flight under
    // the walk of the current one
    int base = maxc - 1;
    uint32_t g_c = 0, h_c = 0;
    {
        const int pos = base - lane;
        if (pos >= 0) {
            g_c = point_list[range.x + (uint32_t)pos] & ~SGR_DEAD;
            h_c = hit4[range.x + (uint32_t)pos];
        }
    }
    const uint64_t row_bytes = (uint64_t)row_stride * 4u;
    for (; base >= 0; base -= 64) {
        uint32_t g_n = 0, h_n = 0;
        {
            const int posn = base - 64 - lane;
            if (posn >= 0) {
                g_n = point_list[range.x + (uint32_t)posn] & ~SGR_DEAD;
                h_n = hit4[range.x + (uint32_t)posn];
            }
        }
        const bool mine = (h_c & qbit) != 0;
        uint64_t ro_c = 0;  // byte offset of the row of (instance, quadrant): row 4 u + q
        if (mine) {
            // u = first row of the Gaussian + index of this tile inside its rect (as in sgr_blend_bwd.hip)
            const uint32_t rc = __float_as_uint(reinterpret_cast<const float*>(rec + 4 * (size_t)g_c + 3)[1]);
            const uint32_t u = sgr_row_of(rc, tx, ty, u0, tmask, g_c);
            ro_c = ((uint64_t)u * 4u + q) * row_bytes;
            // "row 4 u + q will be written": an order-free OR into the instance's flag byte (no return value)
            __hip_atomic_fetch_or(&touched32[u >> 2], qbit << ((u & 3u) * 8u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint64_t m = sgr_uniform_u64(__builtin_amdgcn_ballot_w64(mine));
        int left = __builtin_popcountll(m);
        if (left != 0) {
            sgr_i16 RA, RB;
            auto issue = [&](sgr_i16& R, const int j) __attribute__((always_inline)) {
                const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)g_c, j);
                R = sgr_sload_rec(rec + 4 * (size_t)g);
            };
            auto rowptr = [&](const int j) __attribute__((always_inline)) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ro_c, j);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ro_c >> 32), j);
                return reinterpret_cast<float*>(reinterpret_cast<char*>(partials) + (((uint64_t)hi << 32) | lo));
            };
            // The odd visit of the chunk goes first and alone, then units of two: their 24 sums share one reduce-scatter.
            // The records of the NEXT unit are fetched while the current one is reduced (SGR_SW_PREFETCH).  For that the old
            // records must be dead when the new loads are issued -- otherwise both generations are live (64 SGPRs: hipcc
            // spills them through v_writelane) -- and hipcc sinks per-pixel arithmetic below the loads when left alone:
            // sgr_pin() makes the per-pixel results exist before it (and is a compiler barrier for memory operations),
            // __builtin_amdgcn_sched_barrier keeps the machine scheduler from undoing it.
#if SGR_SW_PREFETCH
            int jA = 0, jB = 0;
            if (left & 1) {
                const int j = sgr_pop_lowest(m);
                issue(RA, j);
                left--;
                float v[12], g[1];
                pixel(RA, base - j, v);
                float* const row = rowptr(j);
                sgr_pin12(v);
                __builtin_amdgcn_sched_barrier(0);
                if (left != 0) {
                    jA = sgr_pop_lowest(m);
                    jB = sgr_pop_lowest(m);
                    issue(RA, jA);
                    issue(RB, jB);
                }
                __builtin_amdgcn_sched_barrier(0);
                sgr_wave_reduce_fold<12>(v, g);
                if (st_a) row[vidx] = g[0];
            } else {
                jA = sgr_pop_lowest(m);
                jB = sgr_pop_lowest(m);
                issue(RA, jA);
                issue(RB, jB);
            }
            while (left != 0) {
                float v[24];
                left -= 2;
                // first visit, then its record's registers take the next unit's first record (in flight under the second
                // visit and the reduction); the same for the second
#if SGR_SW_PREFETCH == 2
                // both visits in ONE scheduling region (their arithmetic may interleave: independent instructions issue
                // faster than dependent ones), both fetches behind it
                pixel(RA, base - jA, v);
                pixel(RB, base - jB, v + 12);
                float* const rowA = rowptr(jA);
                float* const rowB = rowptr(jB);
                sgr_pin12a(v);
                sgr_pin12b(v);
                __builtin_amdgcn_sched_barrier(0);
                sgr_keep(RA);
                sgr_keep(RB);
                if (left != 0) {
                    jA = sgr_pop_lowest(m);
                    jB = sgr_pop_lowest(m);
                    issue(RA, jA);
                    issue(RB, jB);
                }
                __builtin_amdgcn_sched_barrier(0);
#else
                pixel(RA, base - jA, v);
                float* const rowA = rowptr(jA);
                sgr_pin12a(v);
                __builtin_amdgcn_sched_barrier(0);
                sgr_keep(RA);
                if (left != 0) {
                    jA = sgr_pop_lowest(m);
                    issue(RA, jA);
                }
                __builtin_amdgcn_sched_barrier(0);
                pixel(RB, base - jB, v + 12);
                float* const rowB = rowptr(jB);
                sgr_pin12b(v);
                __builtin_amdgcn_sched_barrier(0);
                sgr_keep(RB);
                if (left != 0) {
                    jB = sgr_pop_lowest(m);
                    issue(RB, jB);
                }
                __builtin_amdgcn_sched_barrier(0);
#endif
                float g0, g1;
                sgr_wave_reduce_fold24(v, g0, g1);
                if (st_a) rowA[vidx] = g0;
                if (st_b0) rowB[vidx - 12u] = g0;
                if (st_b1) rowB[4u + vidx] = g1;
            }
#else
            if (left & 1) {
                const int j = sgr_pop_lowest(m);
                issue(RA, j);
                left--;
                float v[12], g[1];
                pixel(RA, base - j, v);
                float* const row = rowptr(j);
                sgr_wave_reduce_fold<12>(v, g);
                if (st_a) row[vidx] = g[0];
            }
            while (left != 0) {
                const int jA = sgr_pop_lowest(m), jB = sgr_pop_lowest(m);
                issue(RA, jA);
                issue(RB, jB);
                float v[24];
                pixel(RA, base - jA, v);
                pixel(RB, base - jB, v + 12);
                float* const rowA = rowptr(jA);
                float* const rowB = rowptr(jB);
                left -= 2;
                float g0, g1;
                sgr_wave_reduce_fold24(v, g0, g1);
                if (st_a) rowA[vidx] = g0;
                if (st_b0) rowB[vidx - 12u] = g0;
                if (st_b1) rowB[4u + vidx] = g1;
            }
#endif
        }
        g_c = g_n;
        h_c = h_n;
    }
}

// rows per instance / flag semantics of this kernel: four rows (one per quadrant) of `sgr_partial_row_stride(0)` floats,
// holding the raw moments in natural order [S gx, S gy, S abs, S gxx, S gxy, S gyy, S Gd, r, g, b, depth, -];
// touched[u] = mask of the quadrant rows that were written (sgr_row_sum_kernel<0, true> consumes both)
void sgr_launch_blend_bwd_sw(bool exact, int gx, int gy, const uint2* ranges, const uint32_t* point_list, int W, int H,
                             const float* bg, const float4* rec, const uint32_t* u0, const uint64_t* tmask, const float* alphas,
                             const uint32_t* n_contrib, const uint8_t* hit4, const float* dL_dpix, const float* dL_ddepth,
                             const float* dL_dalpha, float* partials, int row_stride, uint8_t* touched, hipStream_t s) {
    if (gx <= 0 || gy <= 0) return;
    const unsigned waves = 4u * sgr_xcd_grid_blocks(gx, gy);
    if (exact)
        sgr_blend_bwd_sw_kernel<true><<<waves, 64, 0, s>>>(ranges, point_list, W, H, gx, gy, bg, rec, u0, tmask, alphas, n_contrib, hit4,
                                                          dL_dpix, dL_ddepth, dL_dalpha, partials, row_stride,
                                                          reinterpret_cast<uint32_t*>(touched));
    else
        sgr_blend_bwd_sw_kernel<false><<<waves, 64, 0, s>>>(ranges, point_list, W, H, gx, gy, bg, rec, u0, tmask, alphas, n_contrib, hit4,
                                                           dL_dpix, dL_ddepth, dL_dalpha, partials, row_stride,
                                                           reinterpret_cast<uint32_t*>(touched));
}
