// sgr_blend_fwd.hip -- K10: front-to-back alpha compositing of one 16x16 tile per workgroup on
// gfx950.  Replaces renderCUDA<3> of the reference (forward.cu:340-467).
//
// CDNA4 mapping: 256 lanes = 4 wave64; each wave owns one 8x8 pixel QUADRANT of the tile (compact
// footprint, so a whole wave can skip a splat).  The tile's depth-sorted instance list is staged
// through LDS 256 instances at a time: every lane gathers one instance (three 16-byte loads),
// pre-scales the conic (so exp() is one v_exp_f32), and tests the splat's conservative alpha>=1/255
// bounding box (recA.zw) against the four quadrants.  Four wave ballots turn those tests into one
// 64-bit survivor mask per (quadrant, 64-instance chunk); each wave then walks only the set bits of
// its own masks (s_ff1 loop) and reads the survivor's record from LDS as a broadcast.  Semantic
// channels accumulate in registers (the reference does a global read-modify-write per pair,
// forward.cu:442-444).  Results are identical to walking the full list: a culled instance can only
// fail the alpha test for every pixel of the quadrant.
#include "sgr_math.h"

#define SGR_TILE_THREADS 256
#ifndef SGR_FWD_SEL
#define SGR_FWD_SEL 1  // selects of the blend step on an SGPR-pair lane mask (0: plain ?: -- the compiler's v_cndmask on VCC)
#endif
// lane in m ? a : b, the lane mask in a scalar register pair
__device__ __forceinline__ float sgr_sel_mask(uint64_t m, float a, float b) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
#ifndef SGR_EXACT_COLOR_FUSED
#define SGR_EXACT_COLOR_FUSED 1  // parity mode: the three colour sums as FMAs (see blend_one); 0: unfused like depth (A/B)
#endif
#ifndef SGR_EXACT_TRIM
#define SGR_EXACT_TRIM 1  // parity mode: sgr_expf_ref instead of the library's expf (same bits, sgr_math.h)
#endif

// Register budget (tools/occupancy_audit.py).  Left alone the allocator lands a few registers past an occupancy step
// in most instantiations -- S = 0: 65 VGPRs (the allocation granule is 8, so that is 72 and 7 waves / SIMD instead of
// 8), 1-4 channels 71 (7 instead of 8), 5-8: 89 (5 instead of 6), 9-12: 101 (4 instead of 5), 21-24: 138 (3 instead of
// 4) -- and fits the step without a spill when asked to; 17-20 channels: 136 (3 instead of the 4 workgroups the LDS allows: capped
// in round 6, forward 1.15 -> 0.98 ms at 2 M Gaussians + 19 channels, profiles/r6/ab_fwd20_waves.jsonl), 25-32: 172 (2 instead
// of 3).  13-16 channels are limited by LDS, not registers.
#ifndef SGR_FWD_WAVES
#define SGR_FWD_WAVES(SMAX) ((SMAX) <= 4 ? 8 : (SMAX) == 8 ? 6 : (SMAX) == 12 ? 5 : ((SMAX) == 20 || (SMAX) == 24) ? 4 : (SMAX) == 32 ? 3 : 1)
#endif
template <int SMAX, bool CULL, bool EXACT>
__global__ void __launch_bounds__(SGR_TILE_THREADS) __attribute__((amdgpu_waves_per_eu(SGR_FWD_WAVES(SMAX))))
sgr_blend_fwd_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int S,
                     int gx, int gy, const float4* __restrict__ rec, const float* __restrict__ semantics,
                     const float* __restrict__ bg_color, float* __restrict__ out_color, float* __restrict__ out_depth,
                     float* __restrict__ out_alpha, float* __restrict__ out_semantic, uint32_t* __restrict__ n_contrib,
                     uint8_t* __restrict__ hit4, uint32_t* __restrict__ hlist, uint32_t* __restrict__ n_contrib_k) {
    // fused multiply-adds are written out (fmaf): the CULL / !CULL instantiations must produce bit-identical images
#pragma clang fp contract(off)
    __shared__ float4 sA[SGR_TILE_THREADS];  // {x, y, -, -}
    __shared__ float4 sB[SGR_TILE_THREADS];  // {qa, qb, qc, opacity}
    __shared__ float4 sC[SGR_TILE_THREADS];  // {r, g, b, depth}
    __shared__ uint64_t sBits[4][4];         // [quadrant][chunk of 64 instances]
    __shared__ uint32_t sDone[4];
    // [quadrant][chunk]: the instances of the current batch that were blended into >= 1 pixel of the quadrant.  They go
    // to hit4[] (one byte per sorted instance, bit q = quadrant q); the backward kernel walks exactly those.
    __shared__ uint64_t sHit[4][4];
    __shared__ __attribute__((aligned(16))) uint64_t sAny[2][4];  // [batch parity][chunk]: OR of the four waves' masks (LDS atomics)
    __shared__ uint32_t sCnt[SGR_TILE_THREADS];  // per slot of the batch store_hits last saw: its index in the compact hit list + 1
    __shared__ __attribute__((aligned(16))) float sSem[SMAX > 0 ? SGR_TILE_THREADS * SMAX : 4];  // zero-padded to SMAX

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t tx, ty;
    if (!sgr_wg_tile(blockIdx.x, gx, gy, ranges, tx, ty)) return;  // whole workgroup: padding block
    const uint32_t tile = ty * (uint32_t)gx + tx;
    const uint32_t px = tx * SGR_BLOCK_X + (wave & 1) * 8 + (lane & 7);
    const uint32_t py = ty * SGR_BLOCK_Y + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, Wt = 0.f;
    float sem[SMAX > 0 ? SMAX : 1];
#pragma unroll
    for (int i = 0; i < (SMAX > 0 ? SMAX : 1); i++) sem[i] = 0.f;
    uint32_t last = 0;
    // Finished pixels (outside the image, or T exhausted) are tracked twice: `done_mask` is the wave's scalar lane mask
    // (tile-/wave-wide exits are scalar compares), and `thr`, the per-lane alpha threshold, becomes +inf so that a
    // finished lane fails the alpha test without a separate predicate (hipcc turns ballot(compound bool) and
    // __any/__all into v_cndmask + v_cmp pairs; direct compares combined with scalar ANDs cost no VALU).
    uint64_t done_mask = sgr_uniform_u64(__builtin_amdgcn_ballot_w64(!inside));
    float thr = inside ? SGR_ALPHA_MIN : __builtin_inff();

    // quadrant bounds (pixel centres) used by the staging lanes for the cull test
    const float tx0 = (float)(tx * SGR_BLOCK_X), ty0 = (float)(ty * SGR_BLOCK_Y);

    // One byte per instance of the batch that starts at list index b0 (after the barrier that follows its walk) ... and the
    // COMPACT list of the instances that have a hit byte at all (SgrBinView::hlist: positions relative to the tile's range,
    // ascending): what the backward walks.  A thread's slot is (chunk = its wave, bit = its lane), so its place in the list =
    // hit instances of earlier batches (`nhit`, scalar) + of earlier chunks of this batch + of lower lanes of its own
    // (v_mbcnt on the OR of the four waves' masks).  The same number + 1 goes to sCnt[slot]: each thread also owns a PIXEL,
    // and once the counts of the batch that holds the pixel's last contributor are in LDS (behind the next barrier) it
    // converts its n_contrib (`last`, a list position + 1) into the same counting -- `lastk` = index of the last contributor
    // in the compact list + 1 (that instance has a hit bit: its wave blended it) -- so that the backward's per-pixel test
    // "this instance lies before my last contributor" works on compact indices.  The masks are wave-uniform: they are moved to
    // scalar registers (v_readfirstlane) and counted there; per lane it is four selects, two v_mbcnt and the stores.
    const bool hl = SGR_HLIST && hit4 != nullptr && hlist != nullptr;  // kernel argument: uniform
    uint32_t nhit = 0, lastk = 0, jl_wait = ~0u;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    auto store_hits = [&](const uint32_t b0, const int par) __attribute__((always_inline)) {
        if (hit4 == nullptr) return;
        const uint32_t idx = b0 + (uint32_t)tid;
        // the four waves' masks of this wave's chunk, in scalar registers: a lane's bit of each is one select on the mask
        const uint64_t m0 = sgr_uniform_u64(sHit[0][wave_s]), m1 = sgr_uniform_u64(sHit[1][wave_s]);
        const uint64_t m2 = sgr_uniform_u64(sHit[2][wave_s]), m3 = sgr_uniform_u64(sHit[3][wave_s]);
        const uint32_t h = __float_as_uint(sgr_sel_mask(m0, __uint_as_float(1u), 0.0f)) | __float_as_uint(sgr_sel_mask(m1, __uint_as_float(2u), 0.0f)) |
                           __float_as_uint(sgr_sel_mask(m2, __uint_as_float(4u), 0.0f)) | __float_as_uint(sgr_sel_mask(m3, __uint_as_float(8u), 0.0f));
        if (idx < range.y) hit4[idx] = (uint8_t)h;
        if (hl) {
            const uint64_t mine = m0 | m1 | m2 | m3;
            uint32_t before = 0, total = 0;  // scalar
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                const uint32_t pc = (uint32_t)__builtin_popcountll(sgr_uniform_u64(sAny[par][cc]));
                before += cc < wave_s ? pc : 0u;
                total += pc;
            }
            const uint32_t k = nhit + before + __builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
            if (h) hlist[range.x + k] = idx - range.x;
            sCnt[tid] = k + 1u;
            jl_wait = last - 1u - (b0 - range.x);  // slot of this pixel's last contributor, if it lies in this batch (< 256)
            nhit += total;
        }
    };
    // ... behind a barrier after store_hits
    auto take_lastk = [&]() __attribute__((always_inline)) {
        if (jl_wait < (uint32_t)SGR_TILE_THREADS) lastk = sCnt[jl_wait];
        jl_wait = ~0u;
    };
    bool pending = false;
    uint32_t pbase = 0;
    int par = 0;  // parity of the batch being walked: its sAny[] half (the other half is being read by store_hits)
    for (uint32_t base = range.x; base < range.y; base += SGR_TILE_THREADS, par ^= 1) {
        // tile-wide early exit (forward.cu:394-396); also the barrier that protects LDS reuse
        // (each wave posts "all my pixels are finished"; hipcc's __syncthreads_and is a 20-instruction DPP reduction)
        if (lane == 0) sDone[wave] = (done_mask == ~0ull) ? 1u : 0u;
        __syncthreads();
        if (pending) store_hits(pbase, par ^ 1);  // hit masks of the previous batch (complete: every wave is past its walk)
        pending = false;
        if (sDone[0] & sDone[1] & sDone[2] & sDone[3]) break;

        const uint32_t idx = base + tid;
        uint32_t mask4 = 0;
        // (marked-list mode: an entry flagged SGR_DEAD lies outside its Gaussian's cut-down rect / tile mask -- the cull below
        // would reject it for all four quadrants; it is skipped without fetching its record)
        const uint32_t g = idx < range.y ? point_list[idx] : SGR_DEAD;
        if (!(g & SGR_DEAD)) {
            const float4* r = rec + 4 * (size_t)g;  // one 64-byte line
            const float4 a = r[0];
            const float4 b = r[1];
            sA[tid] = a;
            // EXACT (parity mode): exact power-of-two scalings only -- the walk evaluates the reference's own power
            // expression on the staged triple (sgr_power_ref_staged)
            sB[tid] = EXACT ? make_float4(-0.5f * b.x, -b.y, -0.5f * b.z, b.w) : sgr_stage_conic(b);
            sC[tid] = r[2];
            if (SMAX > 0) {  // channels S..SMAX-1 are staged as zeros so the walk needs no per-channel test
#pragma unroll
                for (int ch = 0; ch < SMAX; ch++) sSem[tid * SMAX + ch] = ch < S ? semantics[(size_t)g * S + ch] : 0.0f;
            }
            mask4 = CULL ? sgr_quadrant_mask(a, b, tx0, ty0) : 0xFu;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint64_t m = __ballot((mask4 >> q) & 1u);
            if (lane == 0) sBits[q][wave] = m;
        }
        if (hl && tid < 4) sAny[par][tid] = 0ull;  // last read one batch ago, two barriers back
        __syncthreads();
        if (hl) take_lastk();
        // this wave's hit masks of the new batch start empty (the reads of the previous batch's are behind the barrier)
        if (lane < 4) sHit[wave][lane] = 0ull;
        pending = true;
        pbase = base;

        if (done_mask != ~0ull) {
            const uint32_t pos0 = base - range.x;  // list position of slot 0 of this batch
            bool stop = false;
            for (int chunk = 0; chunk < 4 && !stop; chunk++) {
                uint64_t m = sBits[wave][chunk];
                m = sgr_uniform_u64(m);
                uint64_t hb = 0;  // wave-uniform (scalar registers)
                // One blend step (forward.cu:425-445) of list slot j = chunk * 64 + bit.  The walk's bookkeeping is scalar
                // (SALU issues once per four cycles per SIMD, and this kernel issues about as many scalar as vector
                // instructions): bits are taken with s_ff1 + s_bitset0, the hit record is one s_bitset1.
                auto blend_one = [&](const int j, const int bit, const float power2, const float alpha) __attribute__((always_inline)) {
                    // skip if power > 0 or alpha < 1/255 (or the pixel is finished: thr = inf)
                    const bool k1 = !(power2 > 0.0f), k2 = !(alpha < thr);
                    const uint64_t hm = __builtin_amdgcn_ballot_w64(k1) & __builtin_amdgcn_ballot_w64(k2);
                    if (hm == 0) return;
                    const float test_T = T * (1.0f - alpha);
                    const bool k3 = test_T < 0.0001f;  // forward.cu:431-436
#if SGR_FWD_SEL
                    // the three selects on "this lane blends" take their lane mask from a scalar register pair
                    // (v_cndmask_b32_e64 with an SGPR-pair mask: 4.6 cycles per wave instruction measured, against 23 for
                    // back-to-back selects on VCC, tools/ubench/valu_rates.hip): the mask is a by-product of the ballots
                    // the walk needs anyway
                    const uint64_t k3m = __builtin_amdgcn_ballot_w64(k3);
                    const uint64_t bm = hm & ~k3m;
#define SGR_BLEND_SEL(a, b) sgr_sel_mask(bm, (a), (b))
#else
                    const bool blend = k1 && k2 && !k3;
#define SGR_BLEND_SEL(a, b) (blend ? (a) : (b))
#endif
                    const float4 c = sC[j];
                    const float w = SGR_BLEND_SEL(alpha * T, 0.0f);
                    if (EXACT) {
                        // the reference's association, unfused: C[ch] += features[ch] * alpha * T (forward.cu:438-441) -- for the
                        // DEPTH (and semantic) sums, whose images the parity mode reproduces bit for bit.  The colour sums take the
                        // fused form (SGR_EXACT_COLOR_FUSED): their inputs already differ from the reference's in the last bit
                        // (the SH evaluation), the image is held to rel 1e-4, and no gradient depends on it
                        const float ae = SGR_BLEND_SEL(alpha, 0.0f);
                        if (SGR_EXACT_COLOR_FUSED) {
                            C0 = fmaf(c.x, w, C0);
                            C1 = fmaf(c.y, w, C1);
                            C2 = fmaf(c.z, w, C2);
                        } else {
                            C0 = C0 + (c.x * ae) * T;
                            C1 = C1 + (c.y * ae) * T;
                            C2 = C2 + (c.z * ae) * T;
                        }
                        Dp = Dp + (c.w * ae) * T;
                    } else {
                        C0 = fmaf(c.x, w, C0);
                        C1 = fmaf(c.y, w, C1);
                        C2 = fmaf(c.z, w, C2);
                        Dp = fmaf(c.w, w, Dp);
                    }
                    Wt += w;
                    if (SMAX > 0) {
                        const float4* sj = reinterpret_cast<const float4*>(&sSem[j * SMAX]);
                        const float ae = SGR_BLEND_SEL(alpha, 0.0f);
#pragma unroll
                        for (int c4 = 0; c4 < SMAX / 4; c4++) {
                            const float4 sv = sj[c4];
                            if (EXACT) {
                                sem[4 * c4] = sem[4 * c4] + (sv.x * ae) * T;
                                sem[4 * c4 + 1] = sem[4 * c4 + 1] + (sv.y * ae) * T;
                                sem[4 * c4 + 2] = sem[4 * c4 + 2] + (sv.z * ae) * T;
                                sem[4 * c4 + 3] = sem[4 * c4 + 3] + (sv.w * ae) * T;
                            } else {
                                sem[4 * c4] = fmaf(sv.x, w, sem[4 * c4]);
                                sem[4 * c4 + 1] = fmaf(sv.y, w, sem[4 * c4 + 1]);
                                sem[4 * c4 + 2] = fmaf(sv.z, w, sem[4 * c4 + 2]);
                                sem[4 * c4 + 3] = fmaf(sv.w, w, sem[4 * c4 + 3]);
                            }
                        }
                    }
                    T = SGR_BLEND_SEL(test_T, T);
                    last = __float_as_uint(SGR_BLEND_SEL(__uint_as_float(pos0 + (uint32_t)j + 1u), __uint_as_float(last)));
#undef SGR_BLEND_SEL
                    // Some lane passed the alpha test: the backward has to visit (quadrant, instance).  (A superset of
                    // the visits that blend: if every passing lane finishes on this very instance nothing is blended, and
                    // the backward's own per-pixel test -- list position < n_contrib -- skips it.)
                    hb = sgr_bitset1(hb, bit);
#if SGR_FWD_SEL
                    const uint64_t sm = hm & k3m;
#else
                    const uint64_t sm = hm & __builtin_amdgcn_ballot_w64(k3);
#endif
                    if (sm != 0) {
                        thr = (k1 && k2 && k3) ? __builtin_inff() : thr;
                        done_mask |= sm;
                        if (done_mask == ~0ull) { m = 0; stop = true; }
                    }
                };
                if (__builtin_popcountll(m) & 1) {  // odd one out first, so that the loop below is pairs only
                    const int b0 = sgr_pop_lowest(m);
                    const int j0 = chunk * 64 + b0;
                    const float4 a0 = sA[j0], q0 = sB[j0];
                    const float pw0 = EXACT ? sgr_power_ref_staged(q0.x, q0.y, q0.z, a0.x - pxf, a0.y - pyf)
                                            : sgr_power2(q0.x, q0.y, q0.z, a0.x - pxf, a0.y - pyf);
                    blend_one(j0, b0, pw0, fminf(0.99f, q0.w * (EXACT ? (SGR_EXACT_TRIM ? sgr_expf_ref(pw0) : expf(pw0)) : __builtin_amdgcn_exp2f(pw0))));
                }
                while (m) {
                    // two survivors per trip: their LDS reads and exp() are independent, only the blend is ordered
                    const int b0 = sgr_pop_lowest(m);
                    const int b1 = sgr_pop_lowest(m);
                    const int j0 = chunk * 64 + b0, j1 = chunk * 64 + b1;
                    const float4 a0 = sA[j0], q0 = sB[j0];
                    const float4 a1 = sA[j1], q1 = sB[j1];
                    const float pw0 = EXACT ? sgr_power_ref_staged(q0.x, q0.y, q0.z, a0.x - pxf, a0.y - pyf)
                                            : sgr_power2(q0.x, q0.y, q0.z, a0.x - pxf, a0.y - pyf);
                    const float pw1 = EXACT ? sgr_power_ref_staged(q1.x, q1.y, q1.z, a1.x - pxf, a1.y - pyf)
                                            : sgr_power2(q1.x, q1.y, q1.z, a1.x - pxf, a1.y - pyf);
                    const float al0 = fminf(0.99f, q0.w * (EXACT ? (SGR_EXACT_TRIM ? sgr_expf_ref(pw0) : expf(pw0)) : __builtin_amdgcn_exp2f(pw0)));
                    const float al1 = fminf(0.99f, q1.w * (EXACT ? (SGR_EXACT_TRIM ? sgr_expf_ref(pw1) : expf(pw1)) : __builtin_amdgcn_exp2f(pw1)));
                    blend_one(j0, b0, pw0, al0);
                    blend_one(j1, b1, pw1, al1);  // if the wave finished on j0 every lane's threshold is +inf: a no-op
                }
                if (lane == 0) {
                    sHit[wave][chunk] = hb;
                    if (hl && hb) __hip_atomic_fetch_or(&sAny[par][chunk], hb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    if (pending) {  // the list ended before the tile was finished: the last batch's hit masks are still in LDS
        __syncthreads();
        store_hits(pbase, par ^ 1);
    }
    if (hl) {  // the counts store_hits left last (uniform branch)
        __syncthreads();
        take_lastk();
    }

    if (inside) {
        const size_t pix_id = (size_t)W * py + px;
        const size_t plane = (size_t)H * W;
        n_contrib[pix_id] = last;
        if (hl) n_contrib_k[pix_id] = lastk;
        out_color[pix_id] = C0 + T * bg_color[0];
        out_color[plane + pix_id] = C1 + T * bg_color[1];
        out_color[2 * plane + pix_id] = C2 + T * bg_color[2];
        out_alpha[pix_id] = Wt;
        out_depth[pix_id] = Dp;
        if (SMAX > 0) {
#pragma unroll
            for (int ch = 0; ch < SMAX; ch++)
                if (ch < S) out_semantic[ch * plane + pix_id] = sem[ch];
        }
    }
}

template <int SMAX>
static void launch_fwd(bool cull, bool exact, unsigned tiles, hipStream_t s, const uint2* ranges, const uint32_t* point_list, int W,
                       int H, int S, int gx, int gy, const float4* rec, const float* semantics, const float* bg, float* out_color, float* out_depth, float* out_alpha,
                       float* out_semantic, uint32_t* n_contrib, uint8_t* hit4, uint32_t* hlist, uint32_t* n_contrib_k) {
#define SGR_FWD_GO(C, E)                                                                                              \
    sgr_blend_fwd_kernel<SMAX, C, E><<<tiles, SGR_TILE_THREADS, 0, s>>>(ranges, point_list, W, H, S, gx, gy, rec, semantics, \
                                                                        bg, out_color, out_depth, out_alpha, out_semantic,   \
                                                                        n_contrib, hit4, hlist, n_contrib_k)
    if (exact) SGR_FWD_GO(true, true);  // parity mode: with the cull (it is invisible in the results)
    else if (cull) SGR_FWD_GO(true, false);
    else SGR_FWD_GO(false, false);
#undef SGR_FWD_GO
}

// S must be <= SGR_SEM_MAX (checked by the caller).
void sgr_launch_blend_fwd(bool cull, bool exact, int gx, int gy, const uint2* ranges, const uint32_t* point_list, int W, int H,
                          int S, const float4* rec, const float* semantics,
                          const float* bg, float* out_color, float* out_depth, float* out_alpha, float* out_semantic,
                          uint32_t* n_contrib, uint8_t* hit4, uint32_t* hlist, uint32_t* n_contrib_k, hipStream_t s) {
    if (gx <= 0 || gy == 0) return;
    const unsigned tiles = sgr_xcd_grid_blocks(gx, gy < 0 ? -gy : gy);  // supertile-ordered grid incl. padding blocks
#define SGR_FWD(N) launch_fwd<N>(cull, exact, tiles, s, ranges, point_list, W, H, S, gx, gy, rec, semantics, bg, \
                                 out_color, out_depth, out_alpha, out_semantic, n_contrib, hit4, hlist, n_contrib_k)
    if (S == 0) SGR_FWD(0);
    else if (S <= 4) SGR_FWD(4);
    else if (S <= 8) SGR_FWD(8);
    else if (S <= 12) SGR_FWD(12);
    else if (S <= 16) SGR_FWD(16);
    else if (S <= 20) SGR_FWD(20);
    else if (S <= 24) SGR_FWD(24);
    else SGR_FWD(32);
#undef SGR_FWD
}
