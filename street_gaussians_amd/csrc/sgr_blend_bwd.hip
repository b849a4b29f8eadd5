// sgr_blend_bwd.hip -- K11: backward of the tile compositing on gfx950.  Replaces
// renderCUDA<3,20> of the reference (backward.cu:415-641), which issues 11+S float atomicAdd per
// contributing (pixel, Gaussian) pair.
//
// MI355X design -- no global atomics at all:
//   * same quadrant-per-wave / LDS-staged / ballot-culled walk as the forward kernel, back to front;
//   * the 11+S per-pair gradient terms are summed across the 64 pixels of a wave with a REDUCE-SCATTER
//     (v_permlane32_swap / v_permlane16_swap fold two values into one register per cross-half step, then four
//     v_add_f32_dpp row steps on a quarter of the registers: 30 instructions for 12 values), and the (up to four)
//     wave partials of an instance meet in LDS with ds_add_f32 on two zero-initialised rows (deterministic);
//   * each (tile, instance) partial row is written ONCE to `partials[u]`, where u is the instance's
//     index in Gaussian-major order (u = exclusive tile offset of the Gaussian + index of this tile
//     inside its rect).  Rows of one Gaussian are therefore contiguous, and the per-Gaussian kernel
//     (sgr_gauss_bwd.hip) reduces them in a fixed order: gradients are bit-reproducible run to run,
//     unlike the reference's unordered atomics.
// Row layout (stride = the instantiation's 12 + SMAX floats rounded up to float4s, 16-B aligned): [0..2] dL/dmean2D (x, y, |x|+|y|),
// [3..5] dL/dconic (x, y, w), [6] dL/dopacity, [7..9] dL/drgb, [10] dL/ddepth, [11..11+S) dL/dsemantic.
#include "sgr_math.h"
#include "sgr_reduce.h"

#include <type_traits>

#ifndef SGR_WITH_VARIANTS
#define SGR_WITH_VARIANTS 0  // 1: also build the designs that were measured slower and kept as A/Bs (tools/build_variant.py)
#endif

#define SGR_TILE_THREADS 256
typedef float sgr_f2 __attribute__((ext_vector_type(2)));
// list entries staged in LDS per round.  128 (not 256) keeps the S = 0 workgroup at 20 KB of LDS so that occupancy is
// set by registers instead of LDS: measured +11 % time at 3 workgroups / CU vs 4.  The instantiations with more than 8
// semantic channels have wide rows (two 128 x 32-float row sets = 33 KB at 20 channels): with the factored channel
// recurrence they need 95-144 VGPRs (3-5 waves / SIMD), so LDS would be the limiter at 128 entries -- they stage 64
// (2 M Gaussians + 19 channels: 2.40 ms vs 2.57 ms at 128).  Every instantiation has the deterministic two-row combine.
#ifndef SGR_BWD_BATCH
#define SGR_BWD_BATCH 128
#endif
#ifndef SGR_BWD_BATCH_WIDE
#define SGR_BWD_BATCH_WIDE 64
#endif
// Waves per SIMD the channel-carrying instantiations are compiled for (register cap = 512 / waves).  The wide kernels
// live on occupancy: uncapped, 12 / 16 / 20 / 24 channels take 101 / 111 / 120 / 128 VGPRs (4-5 waves); capped at 96
// (80 for 12 and 16 channels, no spills; 13 spilled dwords at 20) they measured 1.67 -> 1.41, 1.88 -> 1.65 (1 M
// Gaussians) and 2.14 -> 1.99 ms (2 M + 19 channels); 6 waves at 20 channels spills into the walk (2.07 ms).  1-8
// channels are LDS-limited at 128-entry rounds (64-entry rounds measured 5 % slower there).
#ifndef SGR_BWD_WAVES
#define SGR_BWD_WAVES(SMAX) ((SMAX) >= 12 && (SMAX) <= 16 ? 6 : ((SMAX) >= 20 && (SMAX) <= 24 ? 5 : ((SMAX) > 24 ? 4 : 1)))
#endif
template <int SMAX>
struct SgrBwdBatch { static constexpr int value = SMAX <= 8 ? SGR_BWD_BATCH : SGR_BWD_BATCH_WIDE; };
#define SGR_ROW_BASE 11
// 1: folded row stage of the wave reduction (7 DPP adds + 1 LDS add per hit at S = 0), 0: four row steps per register
#ifndef SGR_FACTORED
#define SGR_FACTORED 1
#endif
// parity mode: 1 = expf / division written out without their range handling (sgr_math.h: sgr_expf_ref, sgr_div_by; same
// bits for the operands of this kernel), 0 = the library's expf and the compiler's IEEE `/` (A/B: tools/build_variant.py)
// Parity mode, backward (round 5).  What north_star asks of the gradients is rel 1e-4, not the reference's bits -- those are
// asked of the tile / bin indices and met by the forward, whose alpha test decides them.  SGR_EXACT_BWD_FAST=1: the backward
// of the parity mode keeps the reference's own power expression (cancellation in it is what can cost digits) but takes
// G = exp(power) from v_exp_f32 and T / (1 - alpha) from the Newton-refined reciprocal, like the default mode -- EXCEPT that a
// visit with any pixel whose alpha lands within 4e-6 (relative) of the 1/255 threshold recomputes G with the accurate expf for
// the whole wave (a wave-uniform branch, taken about once in 10^4 visits): every blend / skip decision is therefore the
// forward's, and what differs from the all-exact backward is rounding-level (<= 4e-7 relative in G, <= 1 ulp per layer in T).
#ifndef SGR_EXACT_BWD_FAST
#define SGR_EXACT_BWD_FAST 1  // 0: every function of the backward with the reference's bits (A/B: tools/build_variant.py)
#endif
#ifndef SGR_BWD_NEWTON
#define SGR_BWD_NEWTON 0  // 1: a Newton step on v_rcp_f32(1 - alpha) before the T recovery (rounds 1-4; measured in round 5: no parity figure moves without it)
#endif
#ifndef SGR_EXACT_BWD_GUARD
#define SGR_EXACT_BWD_GUARD 1  // 0 only for tools/valu_model.py (a static count of the loop without its rare branch)
#endif
#ifndef SGR_EXACT_TRIM
#define SGR_EXACT_TRIM 1
#endif
#ifndef SGR_FOLD
#define SGR_FOLD 1
#endif
// VISIT ROWS (round 5).  Until round 4 the (up to four) wave partials of an instance met in LDS with ds_add_f32 on two
// zero-initialised rows per slot.  The LDS adds floats ONE LANE AT A TIME -- tools/ubench/valu_rates2.hip: a ds_add_f32 costs
// the CU's LDS 3 cycles per active lane, 36 for the twelve lanes of a row, against 4 for a ds_write_b32 -- and the counters
// had the LDS array 60 % busy in this kernel (profiles/pmc_blend_bwd.json: SQ_LDS_IDX_ACTIVE), most of it these adds.  Now
// every VISIT gets a row of its own in LDS: wave q's visits of a round take consecutive rows in the order it walks them
// (row = the rows of the quadrants before it + a running count: scalar bookkeeping), the sums are written with PLAIN
// stores -- one writer per row, no atomics, no zero fill -- and the flush adds the rows of an instance's visits in quadrant
// order (the rank of a slot among the set bits of a quadrant's survivor mask = its row: one v_mbcnt pair).  The row array
// keeps its size (2 rows per slot = 256 at 128-entry rounds); a round whose survivor masks hold more visits than that
// (big splats seen by three or four quadrants each) walks fewer slots -- a multiple of 32, at least a quarter of the round --
// and the next round re-stages the rest.  Deterministic as before (fixed order everywhere).  SGR_VROWS=0: the two-row
// ds_add_f32 combine (A/B: tools/build_variant.py); the !DET / !DPP test instantiations always use it.
#ifndef SGR_VROWS
#define SGR_VROWS 1
#endif
// SPARSE visits (round 5).  The reduce-scatter costs the same 9 permlane swaps + 7 DPP adds however few of the wave's 64
// pixels hit, and on the benchmark frame 27 % of the visits have at most 8 hitting lanes, 41 % at most 16
// (tools/lane_hist.py).  A visit with few hitting lanes skips the cross-lane reduction: the hit lanes file their 12 values
// in an LDS stage (entry = rank of the lane among the hit lanes: twelve ds_write_b32 under the hit lanes' EXEC), the twelve
// lanes that own the row positions read the entries back and add them in rank order (k - 1 plain adds), then store the sum
// into the visit's row exactly like the dense path does.  LDS operations of one wave execute in order: no wait between the
// writes and the reads.  The stage is SGR_SPARSE_K rows per wave behind the visit rows (9: what the LDS has left at eight
// workgroups per CU).  Which path a visit takes depends on its hit lanes alone -- not on what else is in the round -- so the
// gradients stay bit-identical with the cull / hit record on or off (tests).  Measured on the benchmark frame
// (profiles/r5/ab_sparse.txt): thresholds 8, 9, 12 within 0.5 % of each other, 16 (stage in the round's unused rows) 1 %
// behind -- the sum over many entries costs what the reduce-scatter does.  0 disables the path (A/B: tools/build_variant.py).
// S = 0 instantiations with visit rows only.
// (Tried first and dropped: the hit lanes ADDING into one per-wave entry with ds_add_f32 -- no VALU work at all, but the
// LDS serialises float adds at 3-4 cycles per lane-operation, tools/ubench/valu_rates2.hip: 0.97 ms instead of 0.72 at a
// threshold of 8, profiles/r5/ab_sparse.txt.)
#ifndef SGR_SPARSE_K
#define SGR_SPARSE_K 9
#endif

// self-test of the DPP reduction (sgr_selftest in sgr_api.hip)
__global__ void sgr_wave_sum_test_kernel(const float* in, float* out_dpp, float* out_shfl) {
    const float v = in[blockIdx.x * 64 + threadIdx.x];
    float a = v, a1 = 2.f * v, a2 = -v, a3 = v + 1.f;
    sgr_wave_sum4(a, a1, a2, a3);
    const float b = sgr_wave_sum_shfl(v);
    const float c = sgr_wave_sum_dpp(v);
    // reduce-scatter of 12 values x_i = (i+1)*v: row k of r[t] must hold (4t + {0,2,1,3}[k] + 1) * sum(v)
    float x[12], r[3];
#pragma unroll
    for (int i = 0; i < 12; i++) x[i] = (float)(i + 1) * v;
    sgr_wave_reduce_scatter<12>(x, r);
    const int k = threadIdx.x >> 4;
    const int perm = (k == 1) ? 2 : ((k == 2) ? 1 : k);
    bool rs_ok = true;
#pragma unroll
    for (int t = 0; t < 3; t++) rs_ok = rs_ok && (r[t] == (float)(4 * t + perm + 1) * b);
    // folded variant, 12 and 32 values: bank b of row k of g[i] must hold (4t + {0,2,1,3}[k] + 1) * sum(v) with
    // t = 4i + {0,2,1,3}[b], for every t < NVAL/4
    {
        float y[12], g[1];
#pragma unroll
        for (int i = 0; i < 12; i++) y[i] = (float)(i + 1) * v;
        sgr_wave_reduce_fold<12>(y, g);
        const int bank = (threadIdx.x >> 2) & 3, t0 = (0x3120 >> (bank * 4)) & 3;
        if (t0 < 3) rs_ok = rs_ok && (g[0] == (float)(4 * t0 + perm + 1) * b);
        float z[32], gz[2];
#pragma unroll
        for (int i = 0; i < 32; i++) z[i] = (float)(i + 1) * v;
        sgr_wave_reduce_fold<32>(z, gz);
#pragma unroll
        for (int i = 0; i < 2; i++) rs_ok = rs_ok && (gz[i] == (float)(4 * (4 * i + t0) + perm + 1) * b);
        float w[20], gw[2];  // N = 5: a left-over register at both fold levels
#pragma unroll
        for (int i = 0; i < 20; i++) w[i] = (float)(i + 1) * v;
        sgr_wave_reduce_fold<20>(w, gw);
#pragma unroll
        for (int i = 0; i < 2; i++)
            if (4 * i + t0 < 5) rs_ok = rs_ok && (gw[i] == (float)(4 * (4 * i + t0) + perm + 1) * b);
    }
    {   // two visits in one statement: bank b of g0 <-> value 4*perm[b] + perm[k]; banks 0, 2 of g1 <-> 16 + 4*perm[b] + perm[k]
        float z[24], g0, g1;
#pragma unroll
        for (int i = 0; i < 24; i++) z[i] = (float)(i + 1) * v;
        sgr_wave_reduce_fold24(z, g0, g1);
        const int bank = (threadIdx.x >> 2) & 3, pb = (0x3120 >> (bank * 4)) & 3;
        rs_ok = rs_ok && (g0 == (float)(4 * pb + perm + 1) * b);
        if ((bank & 1) == 0) rs_ok = rs_ok && (g1 == (float)(16 + 4 * pb + perm + 1) * b);
    }
    rs_ok = __all(rs_ok);
    if (threadIdx.x == 63) {
        // all four asm chains, the builtin version and the reduce-scatter must agree with the shuffle tree
        const bool ok = (a1 == 2.f * a) && (a2 == -a) && (a3 == a + 64.f) && (c == a) && rs_ok;
        out_dpp[blockIdx.x] = ok ? a : __builtin_nanf("");
        out_shfl[blockIdx.x] = b;
    }
}
void sgr_launch_wave_sum_test(const float* in, float* out_dpp, float* out_shfl, int nwaves, hipStream_t s) {
    sgr_wave_sum_test_kernel<<<nwaves, 64, 0, s>>>(in, out_dpp, out_shfl);
}

// __syncthreads() preceded by an explicit LDS drain.  hipcc (ROCm 7.2, gfx950) emits the post-walk barrier of this
// kernel as a bare s_barrier at a branch target: the last trip's ds_add_f32 / ds_write_b32 of a wave may still be in
// flight when the other waves are released and flush the rows (seen on MI355X as a rare wrong row with 64-entry
// rounds; every other barrier of the library has hipcc's own s_waitcnt lgkmcnt(0) in front of it).
__device__ __forceinline__ void sgr_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}

// byte address inside the workgroup's LDS allocation of a pointer into a __shared__ array (for ds_* written as asm)
__device__ __forceinline__ uint32_t sgr_lds_addr(const float* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)p;
}
// population count of a wave-uniform lane mask, on the scalar unit (hipcc compares the i64 ctpop with a VALU v_cmp_gt_u64)
__device__ __forceinline__ int sgr_popc64(uint64_t m) {
    int r;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(r) : "s"(m) : "scc");
    return r;
}

template <int SMAX, bool CULL, bool DPP, bool DET, int BATCH, bool EXACT = false>
__device__ __forceinline__ void
sgr_blend_bwd_body(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int S,
                     int gx, int gy, const float* __restrict__ bg_color, const float4* __restrict__ rec,
                     const uint32_t* __restrict__ u0, const uint64_t* __restrict__ tmask, const float* __restrict__ semantics, const float* __restrict__ alphas,
                     const uint32_t* __restrict__ n_contrib, const uint8_t* __restrict__ hit4,
                     const float* __restrict__ dL_dpixels, const float* __restrict__ dL_dpixel_depths,
                     const float* __restrict__ dL_dalphas, const float* __restrict__ dL_dpixel_semantics,
                     float* __restrict__ partials, int row_stride, uint8_t* __restrict__ touched, uint32_t row_limit,
                     const uint32_t* __restrict__ hlist, const uint32_t* __restrict__ n_contrib_k, const uint32_t* __restrict__ hl_flag) {
    // Every fused multiply-add below is written out (fmaf / __builtin_elementwise_fma): with contraction left to the
    // optimiser, the CULL / !CULL and DPP / shuffle instantiations of this body can fuse differently and the "culling
    // is invisible, bit for bit" property (tests) would depend on code-generation luck.
#pragma clang fp contract(off)
    constexpr int NS = SMAX > 0 ? SMAX : 1;
    constexpr int NVAL = (SGR_ROW_BASE + SMAX + 3) / 4 * 4;  // values per row, padded to float4s
    // LDS row stride: 16-B aligned rows, and an ODD number of float4s so that the float4 reads / writes of 16 consecutive
    // rows (flush, zero fill) spread over all 64 banks -- at 16 / 24 / 32 floats per row (4, 12, 20 channels) they fell
    // on 4-8 banks (SQ_LDS_BANK_CONFLICT: 102 M cycles per launch at 20 channels, none at S = 0)
    constexpr int ACCW = NVAL + (((NVAL / 4) & 1) ? 0 : 4);
    __shared__ float4 sA[BATCH];  // {x, y, bits of the instance's partial-row index u, -}
    __shared__ float4 sB[BATCH];  // {qa, qb, qc, opacity}
    __shared__ float4 sC[BATCH];  // {r, g, b, depth}
    __shared__ uint64_t sBits[4][4];
    __shared__ uint64_t sVis[4][4];  // VROWS: [quadrant][chunk] the slots whose visit had a hitting pixel (= owns a row)
    __shared__ int sMax[4];
    // The (up to four) wave partials of an instance meet in LDS with ds_add_f32.  DET: two zero-initialised
    // rows per slot, waves {0,1} add into row 0 and waves {2,3} into row 1, the flush adds row 0 + row 1.  Every
    // float add then has exactly two operands (x + 0 is exact), and two-operand addition commutes, so the result
    // does not depend on the order in which the waves arrive: bit-reproducible at half the LDS of per-wave rows.
    // !DET: a single row shared by all four waves (arrival order can change the last bit, like the reference's
    // atomicAdd).
    constexpr int NROW = DET ? 2 : 1;
    // one LDS row per visit, plain stores (see SGR_VROWS).  S = 0 only: the channel-carrying instantiations stage 64 entries per
    // round, the row bookkeeping is paid twice as often per visit and measured 5 % slower at 2 M + 19 channels
    constexpr bool VROWS = SGR_VROWS && DET && DPP && SGR_FOLD && SMAX == 0;
    constexpr int CAP = NROW * BATCH;                             // rows in LDS
    constexpr int NCH = BATCH / 64;                               // 64-slot chunks of a round
    // sparse visits (see SGR_SPARSE_K): the largest hit-lane count that may take the stage path, and the rows the array has
    // beyond the CAP a round's visits may fill
    constexpr int SPK = (VROWS && SMAX == 0) ? SGR_SPARSE_K : 0;
    constexpr int XROWS = 4 * SPK;  // the stage: SPK entries (rows) per wave behind the visit rows
    static_assert(SPK == 0 || ACCW == 12, "a stage entry is a row of 12 floats");
    static_assert(SPK <= 16, "add sparse cases");
    __shared__ __attribute__((aligned(16))) float sAcc[(CAP + XROWS) * ACCW];
    __shared__ __attribute__((aligned(16))) float sSem[SMAX > 0 ? BATCH * SMAX : 4];  // zero-padded to SMAX

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t tx, ty;
    if (!sgr_wg_tile(blockIdx.x, gx, gy, ranges, tx, ty)) return;  // whole workgroup: padding block
    const uint32_t tile = ty * (uint32_t)gx + tx;
    const uint32_t px = tx * SGR_BLOCK_X + (wave & 1) * 8 + (lane & 7);
    const uint32_t py = ty * SGR_BLOCK_Y + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const float pxf = (float)px, pyf = (float)py;
    const size_t pix_id = (size_t)W * py + px;
    const size_t plane = (size_t)H * W;
    const uint2 range = ranges[tile];

    // backward.cu:466-500
    const float T_final = inside ? (1.0f - alphas[pix_id]) : 0.0f;
    float T = T_final;
    // COMPACT walk (round 6).  With the forward's hit record the kernel does not step through the list positions
    // maxc - 1 ... 0 but through the forward's compact list of the instances that have a hit byte (SgrBinView::hlist,
    // ascending positions): `hi` and the slots then count ENTRIES OF THAT LIST, and a pixel's `lastc` is its last contributor's
    // index in that list + 1 (n_contrib_k, written by the forward next to n_contrib: entry e lies before the pixel's last
    // contributor in the list iff its compact index is below that) -- the walk's arithmetic is unchanged.  A round stages 128
    // instances that all have work: in the strict mode 40 % of a tile's list is marked dead, in every mode the instances
    // behind saturated pixels and the cull's margin have no hit, so a tile takes fewer rounds (barriers, staging).
    // Without a hit record (A/B switch bit 3) or without the list (bit 16) the walk is positional as before.
    // (hl_flag: word 7 of the geometry header -- whether THIS frame's forward wrote the list: a backward under other switches
    // than its forward then falls back to the positional walk instead of gathering through a list that is not there)
    const bool HL = SGR_HLIST && CULL && hit4 != nullptr && hlist != nullptr && hl_flag[0] != 0u;
    const int lastc = inside ? (int)(HL ? n_contrib_k : n_contrib)[pix_id] : 0;
    float dLdC0 = 0.f, dLdC1 = 0.f, dLdC2 = 0.f, dLdD = 0.f, dLdA = 0.f;
    float dLdS[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) dLdS[i] = 0.f;
    if (inside) {
        dLdC0 = dL_dpixels[pix_id];
        dLdC1 = dL_dpixels[plane + pix_id];
        dLdC2 = dL_dpixels[2 * plane + pix_id];
        dLdD = dL_dpixel_depths[pix_id];
        dLdA = dL_dalphas[pix_id];
        if (SMAX > 0) {
#pragma unroll
            for (int i = 0; i < SMAX; i++)
                if (i < S) dLdS[i] = dL_dpixel_semantics[i * plane + pix_id];
        }
    }
    const float bgdot = bg_color[0] * dLdC0 + bg_color[1] * dLdC1 + bg_color[2] * dLdC2;
    const bool bg_zero = bg_color[0] == 0.0f && bg_color[1] == 0.0f && bg_color[2] == 0.0f;  // scalar loads: wave-uniform
    // d(pixel)/d(ndc) (backward.cu:501-502) with the 1/log2(e) of the pre-scaled conic folded in
    // EXACT (parity mode): the conic is staged with exact power-of-two scalings only (scale constant 1 instead of log2 e),
    // G comes from the reference's own power expression + the accurate expf, T is recovered by a true division
    constexpr float LSC = EXACT ? 1.0f : SGR_LOG2E;
    const float kx = (0.5f * (float)W) / LSC, ky = (0.5f * (float)H) / LSC;

    // colour / depth recurrences as register pairs {C0, C1} and {C2, D}: the four channels follow the same recurrence,
    // and the float4 {r, g, b, depth} read from LDS is already laid out that way, so they run on v_pk_mul / v_pk_fma
    // (two values per instruction) without any shuffling
    sgr_f2 acc01 = {0.f, 0.f}, acc2D = {0.f, 0.f}, last01 = {0.f, 0.f}, last2D = {0.f, 0.f};
    const sgr_f2 dL01 = {dLdC0, dLdC1}, dL2D = {dLdC2, dLdD};
    float accA = 0.f, last_alpha = 0.f;
    constexpr int NREC = SGR_FACTORED ? 1 : NS;
    float accS[NREC], lastS[NREC];
#pragma unroll
    for (int i = 0; i < NREC; i++) { accS[i] = 0.f; lastS[i] = 0.f; }
    // SGR_FACTORED: the 5 + S per-channel recurrences of backward.cu:553-589 (accum_rec[ch] = last_alpha * last_c[ch] +
    // (1 - last_alpha) * accum_rec[ch];  dL_dalpha += (c[ch] - accum_rec[ch]) * dL_dpixel[ch]) are linear in the
    // channel values, and only their dL_dpixel-weighted sum is used: with u = sum_ch c[ch] * dL_dpixel[ch] (alpha's
    // "colour" is 1) the sum is  u - Arec,  Arec <- last_alpha * u_last + (1 - last_alpha) * Arec: ONE scalar
    // recurrence per pixel instead of 5 + S, same value up to rounding (the terms are summed before the subtraction
    // instead of after it).
    float Arec = 0.f, u_last = 0.f;

    // highest list position any pixel of the tile blended
    int mx = lastc;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) sMax[wave] = mx;
    sgr_lds_barrier();
    const int maxc = max(max(sMax[0], sMax[1]), max(sMax[2], sMax[3]));
    const float tx0 = (float)(tx * SGR_BLOCK_X), ty0 = (float)(ty * SGR_BLOCK_Y);
    // LDS row of this lane's 16-lane group inside a slot (see the reduce-scatter layout below)
    const int acc_lane_off = (DET ? (wave >> 1) : 0) * BATCH * ACCW + (lane >> 4);
    // folded reduction (sgr_wave_reduce_fold): the bank (4 lanes) this lane sits in holds, in result register i,
    // float4 number 4i + fold_t0 of the slot's row; its first lane adds it at component lane >> 4
    const int fold_t0 = (0x3120 >> (((lane >> 2) & 3) * 4)) & 3;  // {0, 2, 1, 3}[bank]
    const bool fold_leader = (lane & 3) == 0;
    const int acc_fold_off = (VROWS ? 0 : (DET ? (wave >> 1) : 0) * BATCH * ACCW) + 4 * fold_t0 + (lane >> 4);
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);  // wave-uniform copy: scalar registers

    // The staging of a round is a dependent pair of gathers (list entry -> record).  The first half is taken out of the
    // round: the entry (and its hit byte) of round n + 1 is loaded while round n walks -- 2 VGPRs held across the walk --
    // so that a round opens with the record gather itself (SGR_BWD_PREFETCH, A/B in DESIGN.md section 10).
#ifndef SGR_BWD_PREFETCH
#define SGR_BWD_PREFETCH 1
#endif
    // list position of entry k of the walk (HL: through the compact list; maxc is then the largest compact count of a pixel)
    auto entry = [&](const int k) __attribute__((always_inline)) -> uint32_t {
        return HL ? hlist[range.x + (uint32_t)k] : (uint32_t)k;
    };
    uint32_t g_pre = 0, h_pre = 0;
    if (SGR_BWD_PREFETCH) {
        const int pos0 = (tid < BATCH) ? (maxc - 1) - tid : -1;
        if (pos0 >= 0) {
            const uint32_t lp = entry(pos0);
            g_pre = point_list[range.x + lp];
            if (CULL && hit4 != nullptr) h_pre = hit4[range.x + lp];
        }
    }
    for (int hi = maxc - 1; hi >= 0;) {
        // slot t of this batch holds list position hi - t (descending: back to front)
        sgr_lds_barrier();  // previous batch fully consumed (rows written) before LDS is overwritten
        const bool stager = tid < BATCH;  // whole waves: the batch is a multiple of 64
        const int pos = stager ? hi - tid : -1;
        uint32_t mask4 = 0;
        if (!VROWS) {
#pragma unroll
            for (int row = tid; row < NROW * BATCH; row += SGR_TILE_THREADS) {
                float4* z = reinterpret_cast<float4*>(&sAcc[row * ACCW]);
#pragma unroll
                for (int k4 = 0; k4 < ACCW / 4; k4++) z[k4] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // An entry no quadrant has to visit is not staged at all -- no record gather, no row: with the forward's hit record
        // that is every instance the forward blended nowhere (hit byte 0), among them the ones the marked-list mode flags as
        // unable to blend (SGR_DEAD, bit 31 of the list entry: sgr_duplicate_kernel)
        const uint32_t g_raw = pos >= 0 ? (SGR_BWD_PREFETCH ? g_pre : point_list[range.x + entry(pos)]) : SGR_DEAD;
        const uint32_t h_raw = (pos >= 0 && CULL && hit4 != nullptr) ? (SGR_BWD_PREFETCH ? h_pre : (uint32_t)hit4[range.x + entry(pos)]) : 0xFu;
        if (pos >= 0 && !(g_raw & SGR_DEAD) && h_raw != 0u) {
            const uint32_t g = g_raw;
            const float4* r = rec + 4 * (size_t)g;  // one 64-byte line
            const float4 a = r[0];
            const float4 b = r[1];
            const float4 d4 = r[3];
            sB[tid] = EXACT ? make_float4(-0.5f * b.x, -b.y, -0.5f * b.z, b.w)
                            : sgr_stage_conic(b);
            sC[tid] = r[2];
            const uint32_t dy_ = __float_as_uint(d4.y);
            // first row of the Gaussian (compact array, L2-resident) + rank of this tile among the tiles it is emitted for:
            // rides in the spare word of the slot's position record (written and read back by this thread only)
            const uint32_t urow = sgr_row_of(dy_, tx, ty, u0, tmask, g);
            sA[tid] = make_float4(a.x, a.y, __uint_as_float(urow), 0.0f);
            if (SMAX > 0) {
#pragma unroll
                for (int ch = 0; ch < SMAX; ch++) sSem[tid * SMAX + ch] = ch < S ? semantics[(size_t)g * S + ch] : 0.0f;
            }
            // which quadrants to walk: the forward's record of the quadrants it blended this instance into (exactly the
            // visits that can contribute; nothing to compute), else the geometric cull the forward uses
            mask4 = CULL ? (hit4 != nullptr ? h_raw : sgr_quadrant_mask(a, b, tx0, ty0)) : 0xFu;
            // lazy forward (sgr_set_lazy) whose frame had more instances than the list capacity: the rows are numbered over ALL
            // instances (index-order scan), the row array holds `row_limit` of them -- rows past it are not visited (the frame
            // is reported invalid one call later; nothing may be written outside the buffers meanwhile)
            if (urow >= row_limit) mask4 = 0;
        }
        if (SGR_BWD_PREFETCH) {  // next round's list entries: in flight under this round's walk
            const int posn = stager ? (hi - BATCH) - tid : -1;
            if (posn >= 0) {
                const uint32_t lp = entry(posn);
                g_pre = point_list[range.x + lp];
                if (CULL && hit4 != nullptr) h_pre = hit4[range.x + lp];
            }
        }
        if (stager) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint64_t m = __ballot((mask4 >> q) & 1u);
                if (lane == 0) sBits[q][wave] = m;
            }
        }
        sgr_lds_barrier();

        // VROWS: the survivor masks of all four quadrants (wave-uniform: scalar registers), how many slots of this round are
        // walked (`lim`: all of them unless the masks hold more visits than there are rows) and the first row of every quadrant
        int lim = BATCH;
        uint64_t vm[4][NCH];
        int vfirst[4] = {0, 0, 0, 0};
        if constexpr (VROWS) {
            auto below = [](const uint64_t m, const int nb) { return nb >= 64 ? m : (nb <= 0 ? 0ull : (m & ((1ull << nb) - 1ull))); };
            uint64_t raw[4][NCH];
            int total = 0;
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    raw[q][c] = sgr_uniform_u64(sBits[q][c]);
                    total += sgr_popc64(raw[q][c]);
                }
            if (total > CAP) {  // rare: shrink the round to the largest multiple of 32 slots whose visits fit the rows
                for (lim = BATCH - 32; lim > 32; lim -= 32) {
                    int t = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++)
#pragma unroll
                        for (int c = 0; c < NCH; c++) t += sgr_popc64(below(raw[q][c], lim - 64 * c));
                    if (t <= CAP) break;
                }
            }
            int first = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                vfirst[q] = first;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    vm[q][c] = below(raw[q][c], lim - 64 * c);
                    first += sgr_popc64(vm[q][c]);
                }
            }
        }
        int rown = 0;  // VROWS: the row of this wave's next visit
        constexpr int kq = SPK;                              // sparse visits: the threshold ...
        const int stage_off = (CAP + wave_s * SPK) * ACCW;   // ... and this wave's stage (floats from sAcc; wave-uniform)
        if constexpr (VROWS) {
#pragma unroll
            for (int q = 0; q < 4; q++) rown = (wave_s == q) ? vfirst[q] : rown;
        }
        (void)kq; (void)stage_off;

        uint64_t vis = 0;  // VROWS: the slots of the current chunk this wave gave a row (scalar)
        // VROWS: a visit with at least one hitting pixel takes the wave's next row (`rown`, wave-uniform) and sets its bit in `vis`
        auto process = [&](const int j, const float4 q, const float dx, const float dy, const float power2, const float G,
                           const float alpha) __attribute__((always_inline)) {
                const int posj = hi - j;  // 0-based list position == `contributor` after its decrement
                // backward.cu:527-545
                // (pixels outside the image have lastc = 0).  The wave-wide "any hit" is taken from the three compare
                // masks with scalar ANDs: ballot(compound bool) costs hipcc a v_cndmask + v_cmp per survivor.
                const bool k0 = posj < lastc, k1 = !(power2 > 0.0f), k2 = !(alpha < SGR_ALPHA_MIN);
                const uint64_t hm = __builtin_amdgcn_ballot_w64(k0) & __builtin_amdgcn_ballot_w64(k1) &
                                    __builtin_amdgcn_ballot_w64(k2);
                // no pixel takes part (every passing pixel of the forward finished on this instance): no row, nothing to add
                if (hm == 0) return;
                const int rowi = rown;
                if constexpr (VROWS) {
                    rown++;
                    vis = sgr_bitset1(vis, j & 63);
                }
                const bool hit = k0 && k1 && k2;

                const float4 c = sC[j];
                float v[NVAL];
#pragma unroll
                for (int k = 0; k < NVAL; k++) v[k] = 0.0f;
                // Everything per-pixel runs under the hit mask and updates the recurrences in place; the other lanes
                // keep dopa = wm = 0, and every output below is a product with one of those two.
                float Gd = 0.0f, wm = 0.0f;  // Gd = G * dL/dalpha-term: zero off the hit lanes even if G overflowed there
                if (hit) {
                    const float oma = 1.0f - alpha;
                    float inv1ma = __builtin_amdgcn_rcpf(oma);
                    if (SGR_BWD_NEWTON || (EXACT && !SGR_EXACT_BWD_FAST))
                        inv1ma = fmaf(fmaf(-oma, inv1ma, 1.0f), inv1ma, inv1ma);  // Newton step: T recovery compounds per layer
                    // T = T / (1 - alpha) (backward.cu:547).  EXACT: the IEEE quotient, from the refined reciprocal above
                    // and two residual corrections (sgr_div_by: the bits of `/` for these operand ranges)
                    T = (EXACT && !SGR_EXACT_BWD_FAST) ? (SGR_EXACT_TRIM ? sgr_div_by(T, oma, inv1ma) : T / oma) : T * inv1ma;
                    wm = alpha * T;
                    const float one_m_la = 1.0f - last_alpha;
                    float d;
                    if (SGR_FACTORED) {
                        Arec = fmaf(last_alpha, u_last, one_m_la * Arec);  // before u: u can then be formed in u_last's register
                        float u = fmaf(c.x, dLdC0, dLdA);
                        u = fmaf(c.y, dLdC1, u);
                        u = fmaf(c.z, dLdC2, u);
                        u = fmaf(c.w, dLdD, u);
                        if (SMAX > 0) {  // padded channels carry zeros end to end (sSem, dLdS), so no per-channel test
                            const float4* sj = reinterpret_cast<const float4*>(&sSem[j * SMAX]);
#pragma unroll
                            for (int c4 = 0; c4 < SMAX / 4; c4++) {
                                const float4 s4 = sj[c4];
                                u = fmaf(s4.x, dLdS[4 * c4], u);
                                u = fmaf(s4.y, dLdS[4 * c4 + 1], u);
                                u = fmaf(s4.z, dLdS[4 * c4 + 2], u);
                                u = fmaf(s4.w, dLdS[4 * c4 + 3], u);
                            }
                        }
                        d = u - Arec;
                        u_last = u;
                    } else {
                    const sgr_f2 c01 = {c.x, c.y}, c2D = {c.z, c.w};
                    const sgr_f2 la2 = {last_alpha, last_alpha};
                    acc01 = __builtin_elementwise_fma(la2, last01, one_m_la * acc01);
                    acc2D = __builtin_elementwise_fma(la2, last2D, one_m_la * acc2D);
                    const sgr_f2 t = __builtin_elementwise_fma(c2D - acc2D, dL2D, (c01 - acc01) * dL01);
                    d = t.x + t.y;
                    accA = fmaf(one_m_la, accA, last_alpha);
                    d = fmaf(1.0f - accA, dLdA, d);
                    if (SMAX > 0) {  // padded channels carry zeros end to end (sSem, dLdS), so no per-channel test
                        const float4* sj = reinterpret_cast<const float4*>(&sSem[j * SMAX]);
#pragma unroll
                        for (int c4 = 0; c4 < SMAX / 4; c4++) {
                            const float4 s4 = sj[c4];
                            const float svv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const int ch = (4 * c4 + e) % NREC;
                                accS[ch] = fmaf(last_alpha, lastS[ch], one_m_la * accS[ch]);
                                d = fmaf(svv[e] - accS[ch], dLdS[4 * c4 + e], d);
                                lastS[ch] = svv[e];
                            }
                        }
                    }
                    last01 = c01;
                    last2D = c2D;
                    }
                    last_alpha = alpha;
                    d *= T;
                    // backward.cu:611-614.  EXACT with a black background (bg_zero, wave-uniform): (-T_final / oma) * 0 adds a
                    // signed zero -- the second quotient is skipped
                    if (EXACT && SGR_EXACT_BWD_FAST)
                        Gd = bg_zero ? G * d : G * (d + (-T_final * inv1ma) * bgdot);
                    else if (EXACT)
                        Gd = (SGR_EXACT_TRIM && bg_zero) ? G * d
                                                         : G * (d + (SGR_EXACT_TRIM ? sgr_div_by(-T_final, oma, inv1ma) : -T_final / oma) * bgdot);
                    else
                        Gd = G * fmaf(-T_final * inv1ma, bgdot, d);
                }
                if (SMAX > 0 && !(DPP && SGR_FOLD)) {
#pragma unroll
                    for (int ch = 0; ch < SMAX; ch++) v[SGR_ROW_BASE + ch] = wm * dLdS[ch];
                }
                // Per pixel only the MOMENTS of Gd = G * dopa are formed (Gd*dx, Gd*dy, Gd*dx^2, Gd*dx*dy, Gd*dy^2) plus the
                // "abs" term, which is not linear; the flush turns the tile's sums into dL/dmean2D and dL/dconic with the
                // instance's conic and opacity (dL_dG = opacity * dopa is constant per instance, backward.cu:616-635):
                // 17 instead of 26 instructions per visit, same sums up to rounding.
                const float gx = Gd * dx, gy = Gd * dy;
                // dG/ddelx / G = (2*qa*dx + qb*dy)/log2e  (qa = -0.5*log2e*A, qb = -log2e*B; 1/log2e is in kx, ky)
                // built on the two products sgr_power2 already formed (qa*dx + qb*dy and qc*dy: common subexpressions)
                float ax, ay;
                {
#pragma clang fp contract(off)
                    const float e1 = fmaf(q.y, dy, q.x * dx), e2 = q.z * dy;
                    ax = fmaf(q.x, dx, e1);
                    ay = fmaf(q.y, dx, e2 + e2);
                }
                v[0] = gx;
                v[1] = gy;
                v[2] = fabsf(Gd) * fmaf(fabsf(ax), kx, fabsf(ay) * ky);
                v[3] = gx * dx;
                v[4] = gx * dy;
                v[5] = gy * dy;
                v[6] = Gd;
                const sgr_f2 o01 = wm * dL01, o2D = wm * dL2D;
                v[7] = o01.x;
                v[8] = o01.y;
                v[9] = o2D.x;
                v[10] = o2D.y;
                // LDS row layout: float4 t = values (4t, 4t+2, 4t+1, 4t+3) -- the order the reduce-scatter leaves
                // them in rows 0..3 of register t; the flush below swaps the middle pair back.
                float r[NVAL / 4];
                if constexpr (SPK > 0) {
                    const int kc = sgr_popc64(hm);
                    if (kc <= kq) {
                        if (hit) {  // entry = rank among the hit lanes; an entry is laid out like a row (float4 t = values 4t, 4t+2, 4t+1, 4t+3)
                            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                            // 32-bit stores written as ds_write_b32 (the optimiser merges plain stores into ds_write_b128, which
                            // wants its four values in consecutive registers: the copies that takes landed in the dense path too)
                            const uint32_t ea = sgr_lds_addr(sAcc + stage_off) + rank * 48u;
#define SGR_STAGE_ST(OFF, VAL) asm volatile("ds_write_b32 %0, %1 offset:" #OFF : : "v"(ea), "v"(VAL) : "memory")
                            SGR_STAGE_ST(0, v[0]); SGR_STAGE_ST(4, v[2]); SGR_STAGE_ST(8, v[1]); SGR_STAGE_ST(12, v[3]);
                            SGR_STAGE_ST(16, v[4]); SGR_STAGE_ST(20, v[6]); SGR_STAGE_ST(24, v[5]); SGR_STAGE_ST(28, v[7]);
                            SGR_STAGE_ST(32, v[8]); SGR_STAGE_ST(36, v[10]); SGR_STAGE_ST(40, v[9]); SGR_STAGE_ST(44, v[11]);
#undef SGR_STAGE_ST
                        }
                        __builtin_amdgcn_wave_barrier();  // same wave, program order: the LDS serves the reads below after the writes above
                        if (fold_leader && fold_t0 < NVAL / 4) {
                            const float* src = sAcc + (acc_fold_off + stage_off);
                            // kc is wave-uniform: one case, all its loads in flight at once, adds in rank order
                            auto sum_first = [&](auto KC) __attribute__((always_inline)) {
                                constexpr int kk = decltype(KC)::value, k0 = kk < 8 ? kk : 8;
                                float a[k0];  // (at most eight loads in flight: the kernel lives on 64 registers)
#pragma unroll
                                for (int e = 0; e < k0; e++) a[e] = src[e * 12];
                                float t = a[0];
#pragma unroll
                                for (int e = 1; e < k0; e++) t += a[e];
                                if constexpr (kk > 8) {
                                    float b[kk - 8];
#pragma unroll
                                    for (int e = 8; e < kk; e++) b[e - 8] = src[e * 12];
#pragma unroll
                                    for (int e = 8; e < kk; e++) t += b[e - 8];
                                }
                                return t;
                            };
                            float t = 0.0f;
                            switch (kc) {
#define SGR_SPARSE_CASE(N) case N: if constexpr (N <= SPK) t = sum_first(std::integral_constant<int, (N <= SPK ? N : 1)>{}); break
                                SGR_SPARSE_CASE(1); SGR_SPARSE_CASE(2); SGR_SPARSE_CASE(3); SGR_SPARSE_CASE(4);
                                SGR_SPARSE_CASE(5); SGR_SPARSE_CASE(6); SGR_SPARSE_CASE(7); SGR_SPARSE_CASE(8);
                                SGR_SPARSE_CASE(9); SGR_SPARSE_CASE(10); SGR_SPARSE_CASE(11); SGR_SPARSE_CASE(12);
                                SGR_SPARSE_CASE(13); SGR_SPARSE_CASE(14); SGR_SPARSE_CASE(15); SGR_SPARSE_CASE(16);
#undef SGR_SPARSE_CASE
                                default: break;
                            }
                            sAcc[acc_fold_off + rowi * ACCW] = t;  // the visit's own row: one writer
                            // keeps the optimiser from merging this store with the dense path's: the merged tail made the
                            // twelve values live across this block (and the loads above serialise for want of registers)
                            asm volatile("; sparse visit done" ::: "memory");
                        }
                        return;
                    }
                }
                if (DPP && SGR_FOLD) {
                    // 16 values at a time (one result register, one ds_add_f32 each): the same instructions as one
                    // pass over all NVAL values, but the channel products w * dL/dsemantic are formed chunk by chunk,
                    // so a wide instantiation keeps 16 + 8 instead of NVAL + NVAL/2 reduction registers live
                    float* dst = sAcc + (acc_fold_off + (VROWS ? rowi : j) * ACCW);  // wave-uniform: scalar multiply
                    auto fold_chunk = [&](auto C0, auto CN) __attribute__((always_inline)) {
                        constexpr int c0 = decltype(C0)::value, cn = decltype(CN)::value;
                        float vv[cn], g[1];
#pragma unroll
                        for (int k = 0; k < cn; k++) {
                            const int idx = c0 + k;
                            vv[k] = idx < SGR_ROW_BASE ? v[idx < NVAL ? idx : 0]
                                                       : (idx - SGR_ROW_BASE < SMAX ? wm * dLdS[(idx - SGR_ROW_BASE) % NS] : 0.0f);
                        }
                        sgr_wave_reduce_fold<cn>(vv, g);
                        if (fold_leader && fold_t0 < cn / 4) {
                            if constexpr (VROWS) dst[c0] = g[0];  // the visit's own row: one writer
                            else atomicAdd(&dst[c0], g[0]);
                        }
                    };
                    fold_chunk(std::integral_constant<int, 0>{}, std::integral_constant<int, (NVAL < 16 ? NVAL : 16)>{});
                    if constexpr (NVAL > 16)
                        fold_chunk(std::integral_constant<int, 16>{}, std::integral_constant<int, (NVAL - 16 < 16 ? NVAL - 16 : 16)>{});
                    if constexpr (NVAL > 32)
                        fold_chunk(std::integral_constant<int, 32>{}, std::integral_constant<int, (NVAL - 32 < 16 ? NVAL - 32 : 16)>{});
                    static_assert(NVAL <= 48, "add a fold chunk");
                    return;
                } else if (DPP) {
                    sgr_wave_reduce_scatter<NVAL>(v, r);
                } else {
#pragma unroll
                    for (int t = 0; t < NVAL / 4; t++) {
                        const float s0 = sgr_wave_sum_shfl(v[4 * t]), s1 = sgr_wave_sum_shfl(v[4 * t + 1]);
                        const float s2 = sgr_wave_sum_shfl(v[4 * t + 2]), s3 = sgr_wave_sum_shfl(v[4 * t + 3]);
                        const int k = lane >> 4;
                        r[t] = k == 0 ? s0 : (k == 1 ? s2 : (k == 2 ? s1 : s3));
                    }
                }
                if ((lane & 15) == 0) {  // one lane per 16-lane row stores that row's values
                    const int k = lane >> 4;
                    float* dst = sAcc + (acc_lane_off + j * ACCW);  // j is wave-uniform: scalar multiply
                    (void)k;
#pragma unroll
                    for (int t = 0; t < NVAL / 4; t++) atomicAdd(&dst[4 * t], r[t]);
                }
        };
        // parity mode: G = exp(power), power in natural-log units (see SGR_EXACT_BWD_FAST)
        auto exactG = [&](const float pw, const float opac) __attribute__((always_inline)) -> float {
            if constexpr (SGR_EXACT_BWD_FAST != 0) {
                float g = __builtin_amdgcn_exp2f(pw * SGR_LOG2E);
                const bool near = fabsf(opac * g - SGR_ALPHA_MIN) <= 4.0e-6f * SGR_ALPHA_MIN;
                if (SGR_EXACT_BWD_GUARD && __builtin_amdgcn_ballot_w64(near) != 0) g = SGR_EXACT_TRIM ? sgr_expf_ref(pw) : expf(pw);
                return g;
            } else {
                return SGR_EXACT_TRIM ? sgr_expf_ref(pw) : expf(pw);
            }
        };
        for (int chunk = 0; chunk < BATCH / 64; chunk++) {
            uint64_t m;
            m = sBits[wave][chunk];
            m = sgr_uniform_u64(m);
            if constexpr (VROWS) {  // only the slots this round walks (`chunk` may be a run-time index: no register array here)
                const int nb = lim - 64 * chunk;
                m = nb >= 64 ? m : (nb <= 0 ? 0ull : (m & ((1ull << nb) - 1ull)));
            }
            // scalar bookkeeping kept short (SALU issues once per four cycles per SIMD): s_ff1 + s_bitset0 per survivor, and
            // the odd survivor is taken first so that the loop is pairs only
            if (__builtin_popcountll(m) & 1) {
                const int j0 = chunk * 64 + sgr_pop_lowest(m);
                const float4 a0 = sA[j0], q0 = sB[j0];
                const float dx0 = a0.x - pxf, dy0 = a0.y - pyf;
                const float pw0 = EXACT ? sgr_power_ref_staged(q0.x, q0.y, q0.z, dx0, dy0) : sgr_power2(q0.x, q0.y, q0.z, dx0, dy0);
                const float G0 = EXACT ? exactG(pw0, q0.w) : __builtin_amdgcn_exp2f(pw0);
                process(j0, q0, dx0, dy0, pw0, G0, fminf(0.99f, q0.w * G0));
            }
            while (m) {
                // two survivors per trip: LDS reads and exp() of both are independent of each other; only the
                // per-pixel recurrences (process) are ordered
                const int j0 = chunk * 64 + sgr_pop_lowest(m);
                const int j1 = chunk * 64 + sgr_pop_lowest(m);
                const float4 a0 = sA[j0], q0 = sB[j0];
                const float4 a1 = sA[j1], q1 = sB[j1];
                const float dx0 = a0.x - pxf, dy0 = a0.y - pyf, dx1 = a1.x - pxf, dy1 = a1.y - pyf;
                const float pw0 = EXACT ? sgr_power_ref_staged(q0.x, q0.y, q0.z, dx0, dy0) : sgr_power2(q0.x, q0.y, q0.z, dx0, dy0);
                const float pw1 = EXACT ? sgr_power_ref_staged(q1.x, q1.y, q1.z, dx1, dy1) : sgr_power2(q1.x, q1.y, q1.z, dx1, dy1);
                const float G0 = EXACT ? exactG(pw0, q0.w) : __builtin_amdgcn_exp2f(pw0), G1 = EXACT ? exactG(pw1, q1.w) : __builtin_amdgcn_exp2f(pw1);
                const float al0 = fminf(0.99f, q0.w * G0), al1 = fminf(0.99f, q1.w * G1);
                process(j0, q0, dx0, dy0, pw0, G0, al0);
                process(j1, q1, dx1, dy1, pw1, G1, al1);
            }
            if constexpr (VROWS) {
                if (lane == 0) sVis[wave][chunk] = vis;
                vis = 0;
            }
        }
        sgr_lds_barrier();
        // One row per (tile, instance) some quadrant was asked to visit: plain stores, written exactly once.  With the
        // forward's hit record that is the set of instances that blended into the tile (plus the rare one whose every
        // passing pixel finished on it: a row of zeros); flagging rows per visit instead cost an LDS store + exec
        // juggling in the walk.
        uint64_t vq[4][NCH];  // VROWS: the slots every quadrant's wave gave a row (wave-uniform)
        if constexpr (VROWS) {
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int c = 0; c < NCH; c++) vq[q][c] = sgr_uniform_u64(sVis[q][c]);
        }
        if (mask4 != 0 && tid < lim) {
            const uint32_t u = __float_as_uint(sA[tid].z);
            touched[u] = 1;  // the per-Gaussian reduction only reads rows that were written (no 64 B/instance memset)
            float4* row = reinterpret_cast<float4*>(partials + (size_t)u * row_stride);
            float4 r[NVAL / 4];
#pragma unroll
            for (int k4 = 0; k4 < NVAL / 4; k4++) r[k4] = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (VROWS) {
                // the rows of this slot's visits, in quadrant order: row of quadrant q = the quadrant's first row + the rank of
                // the slot among the slots the quadrant's wave gave a row (whole chunks before this wave's: scalar; inside: v_mbcnt)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    int before = vfirst[q];
#pragma unroll
                    for (int c = 0; c + 1 < NCH; c++) before += (c < wave_s) ? sgr_popc64(vq[q][c]) : 0;
                    uint64_t mine = vq[q][0];
#pragma unroll
                    for (int c = 1; c < NCH; c++) mine = (wave_s == c) ? vq[q][c] : mine;
                    if ((mine >> lane) & 1ull) {
                        const int rq = before + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
                        const float4* src = reinterpret_cast<const float4*>(&sAcc[rq * ACCW]);
#pragma unroll
                        for (int k4 = 0; k4 < NVAL / 4; k4++) {
                            const float4 t = src[k4];
                            r[k4].x += t.x; r[k4].y += t.y; r[k4].z += t.z; r[k4].w += t.w;
                        }
                    }
                }
            } else {
                const float4* src = reinterpret_cast<const float4*>(&sAcc[tid * ACCW]);
#pragma unroll
                for (int k4 = 0; k4 < NVAL / 4; k4++) r[k4] = src[k4];
                if (DET) {
                    const float4* src1 = reinterpret_cast<const float4*>(&sAcc[(BATCH + tid) * ACCW]);
#pragma unroll
                    for (int k4 = 0; k4 < NVAL / 4; k4++) {
                        const float4 t = src1[k4];
                        r[k4].x += t.x; r[k4].y += t.y; r[k4].z += t.z; r[k4].w += t.w;
                    }
                }
            }
            // moments -> gradients (see `process`): r[0] = {S gx, S abs, S gy, S gxx}, r[1] = {S gxy, S Gd, S gyy, colour r}
            {
                const float4 qq = sB[tid];
                const float sgx = r[0].x, sgy = r[0].z, qw = qq.w, hq = -0.5f * qq.w;
                r[0].x = qw * kx * fmaf(qq.x + qq.x, sgx, qq.y * sgy);  // dL/dmean2D.x
                r[0].z = qw * ky * fmaf(qq.z + qq.z, sgy, qq.y * sgx);  // dL/dmean2D.y
                r[0].y = qw * r[0].y;                                    // sum |gx| + |gy|
                r[0].w = hq * r[0].w;                                    // dL/dconic.x
                r[1].x = hq * r[1].x;                                    // dL/dconic.y
                r[1].z = hq * r[1].z;                                    // dL/dconic.w
            }
#pragma unroll
            for (int k4 = 0; k4 < NVAL / 4; k4++) row[k4] = make_float4(r[k4].x, r[k4].z, r[k4].y, r[k4].w);
        }
        // next round: the slots this one did not walk (VROWS with more visits than rows) are staged again
        hi -= lim;
        if (SGR_BWD_PREFETCH && lim != BATCH) {  // rare: the entries fetched ahead were those of hi - BATCH
            const int posn = stager ? hi - tid : -1;
            if (posn >= 0) {
                const uint32_t lp = entry(posn);
                g_pre = point_list[range.x + lp];
                if (CULL && hit4 != nullptr) h_pre = hit4[range.x + lp];
            }
        }
    }
}

#define SGR_BWD_ARGS                                                                                                  \
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list, int W, int H, int S, int gx, int gy,       \
        const float *__restrict__ bg_color, const float4 *__restrict__ rec, const uint32_t *__restrict__ u0, const uint64_t *__restrict__ tmask,          \
        const float *__restrict__ semantics,          \
        const float *__restrict__ alphas, const uint32_t *__restrict__ n_contrib, const uint8_t *__restrict__ hit4,        \
        const float *__restrict__ dL_dpixels, const float *__restrict__ dL_dpixel_depths,                                  \
        const float *__restrict__ dL_dalphas,                                                                              \
        const float *__restrict__ dL_dpixel_semantics, float *__restrict__ partials, int row_stride,                      \
        uint8_t *__restrict__ touched, uint32_t row_limit, const uint32_t *__restrict__ hlist,  \
        const uint32_t *__restrict__ n_contrib_k, const uint32_t *__restrict__ hl_flag
#define SGR_BWD_PASS                                                                                                  \
    ranges, point_list, W, H, S, gx, gy, bg_color, rec, u0, tmask, semantics, alphas, n_contrib, hit4, dL_dpixels,               \
        dL_dpixel_depths, dL_dalphas, dL_dpixel_semantics, partials, row_stride, touched, row_limit, hlist, n_contrib_k, hl_flag
template <int SMAX, bool CULL, bool DPP, bool DET>
__global__ void __launch_bounds__(SGR_TILE_THREADS) __attribute__((amdgpu_waves_per_eu(SGR_BWD_WAVES(SMAX))))
sgr_blend_bwd_kernel(SGR_BWD_ARGS) {
    sgr_blend_bwd_body<SMAX, CULL, DPP, DET, SgrBwdBatch<SMAX>::value>(SGR_BWD_PASS);
}
// S = 0 (the training configuration of the benchmark): 64 VGPRs fit without spilling, so ask for 8 waves / SIMD
// (hipcc settles at 80 VGPRs = 6 waves otherwise; measured 1.103 -> 1.087 ms).  With semantic channels the register
// budget is larger and the default heuristic is kept.
template <bool CULL, bool DPP, bool DET>
__global__ void __launch_bounds__(SGR_TILE_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8)))
sgr_blend_bwd_kernel_s0(SGR_BWD_ARGS) {
    sgr_blend_bwd_body<0, CULL, DPP, DET, SgrBwdBatch<0>::value>(SGR_BWD_PASS);
}
// parity mode (sgr_math.h: sgr_power_ref): the shipped configuration only (cull / hit record, DPP, deterministic)
#ifndef SGR_BWD_EXACT_WAVES
#define SGR_BWD_EXACT_WAVES(SMAX) ((SMAX) == 0 ? 8 : SGR_BWD_WAVES(SMAX))
#endif
template <int SMAX>
__global__ void __launch_bounds__(SGR_TILE_THREADS) __attribute__((amdgpu_waves_per_eu(SGR_BWD_EXACT_WAVES(SMAX))))
sgr_blend_bwd_kernel_exact(SGR_BWD_ARGS) {
    sgr_blend_bwd_body<SMAX, true, true, true, SgrBwdBatch<SMAX>::value, true>(SGR_BWD_PASS);
}

#if SGR_WITH_VARIANTS  // rejected A/B design, built only by tools/build_variant.py (-DSGR_WITH_VARIANTS=1): DESIGN.md section 10
// =====================================================================================================================
// S = 0, second design ("transposed accumulation").  The kernel above spends ~60 of its ~100 VALU instructions per
// (quadrant, instance) visit on turning 64 per-pixel terms into 11 sums: the gradient products are formed on every
// lane (44 % of them hit) and then reduced ACROSS lanes.  Here the reduction is turned into per-lane serial work:
//   pass A (lane = pixel, as before): walk the visit, update the per-pixel recurrences, and store only TWO scalars per
//          pixel in LDS -- Gd = G * dL/dalpha-term and wm = alpha * T (zero where the pixel is not hit);
//   pass B (after 8 visits): lane = (visit v, pixel column g).  Each lane reads the 8 pixels of its column of its
//          visit back from LDS (an XOR swizzle keeps both directions bank-conflict free), and accumulates the moments
//          sum Gd*{1, dx, dy, dx^2, dx*dy, dy^2}, sum |..| and sum wm*dL/d{r,g,b,depth} in registers -- every lane busy,
//          no cross-lane traffic; then three DPP levels inside each 8-lane group, and one lane per visit turns the
//          moments into the row (mean2D, conic, opacity, abs, colour, depth) and adds it to the slot's row in LDS.
// ~48 + ~25 instructions per visit instead of ~100, still free of global atomics and bit-reproducible.
// dL/dmean2D and dL/dconic are assembled from moments (sum first, multiply by the conic once per visit), so they
// differ from the per-pixel products of backward.cu:616-635 by rounding only.
#define SGR_V2_BATCH 64
#define SGR_V2_CH 8
#ifndef SGR_V2_WAVES
#define SGR_V2_WAVES 5  // waves per SIMD the register allocation aims at (30 KB of LDS allow 5 workgroups per CU)
#endif
template <bool CULL, bool DET>
__global__ void __launch_bounds__(SGR_TILE_THREADS) __attribute__((amdgpu_waves_per_eu(SGR_V2_WAVES, SGR_V2_WAVES)))
sgr_blend_bwd_kernel_v2(SGR_BWD_ARGS) {
#pragma clang fp contract(off)
    constexpr int BATCH = SGR_V2_BATCH, CH = SGR_V2_CH, ACCW = 12, NROW = DET ? 2 : 1;
    __shared__ float4 sA[BATCH];  // {x, y, -, -}
    __shared__ float4 sB[BATCH];  // {qa, qb, qc, opacity}
    __shared__ float4 sC[BATCH];  // {r, g, b, depth}
    __shared__ uint32_t sU[BATCH];
    __shared__ uint32_t sFlag[BATCH];
    __shared__ uint64_t sBits[4];
    __shared__ int sMax[4];
    __shared__ __attribute__((aligned(16))) float sAcc[NROW * BATCH * ACCW];
    __shared__ __attribute__((aligned(16))) float2 sW[4][CH][64];  // [wave][visit][pixel ^ swizzle] = {Gd, wm}
    __shared__ float4 sDL[4][64];                                    // [wave][pixel] = dL/d{r, g, b, depth}
    __shared__ int sVis[4][CH];                                      // slot of each visit of the chunk
    (void)S; (void)semantics; (void)dL_dpixel_semantics;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t tx, ty;
    if (!sgr_wg_tile(blockIdx.x, gx, gy, ranges, tx, ty)) return;  // whole workgroup: padding block
    const uint32_t tile = ty * (uint32_t)gx + tx;
    const uint32_t qx0 = tx * SGR_BLOCK_X + (wave & 1) * 8, qy0 = ty * SGR_BLOCK_Y + (wave >> 1) * 8;
    const uint32_t px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const float pxf = (float)px, pyf = (float)py;
    const size_t pix_id = (size_t)W * py + px;
    const size_t plane = (size_t)H * W;
    const uint2 range = ranges[tile];

    const float T_final = inside ? (1.0f - alphas[pix_id]) : 0.0f;  // backward.cu:466-500
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
    sDL[wave][lane] = make_float4(dLdC0, dLdC1, dLdC2, dLdD);
    const float bgdot = bg_color[0] * dLdC0 + bg_color[1] * dLdC1 + bg_color[2] * dLdC2;
    const float kx = (0.5f * (float)W) / SGR_LOG2E, ky = (0.5f * (float)H) / SGR_LOG2E;

    float Arec = 0.f, u_last = 0.f, last_alpha = 0.f;

    int mx = lastc;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) sMax[wave] = mx;
    sgr_lds_barrier();
    const int maxc = max(max(sMax[0], sMax[1]), max(sMax[2], sMax[3]));
    const float tx0 = (float)(tx * SGR_BLOCK_X), ty0 = (float)(ty * SGR_BLOCK_Y);
    // pass B coordinates: this lane handles column g of visit v
    const int pv = lane >> 3, pg = lane & 7;
    const float bpx = (float)(qx0 + (uint32_t)pg), bpy0 = (float)qy0;
    const int rowset = (DET ? (wave >> 1) : 0) * BATCH * ACCW;

    for (int hi = maxc - 1; hi >= 0; hi -= BATCH) {
        sgr_lds_barrier();  // previous batch fully consumed (rows written) before LDS is overwritten
        const bool stager = tid < BATCH;
        const int pos = stager ? hi - tid : -1;
        uint32_t mask4 = 0;
        if (stager) sFlag[tid] = 0;
        if (tid < NROW * BATCH) {
            float4* z = reinterpret_cast<float4*>(&sAcc[tid * ACCW]);
            z[0] = make_float4(0.f, 0.f, 0.f, 0.f);
            z[1] = make_float4(0.f, 0.f, 0.f, 0.f);
            z[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (pos >= 0) {
            const uint32_t g = point_list[range.x + (uint32_t)pos] & ~SGR_DEAD;
            const float4* r = rec + 4 * (size_t)g;  // one 64-byte line
            const float4 a = r[0];
            const float4 b = r[1];
            const float4 d4 = r[3];
            sA[tid] = a;
            sB[tid] = sgr_stage_conic(b);
            sC[tid] = r[2];
            const uint32_t dy_ = __float_as_uint(d4.y);
            // first row of the Gaussian (compact array, L2-resident) + rank of this tile among the tiles it is emitted for
            sU[tid] = sgr_row_of(dy_, tx, ty, u0, tmask, g);
            mask4 = CULL ? (hit4 != nullptr ? (uint32_t)hit4[range.x + (uint32_t)pos] : sgr_quadrant_mask(a, b, tx0, ty0))
                         : 0xFu;
        }
        if (stager) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint64_t m = __ballot((mask4 >> q) & 1u);
                if (lane == 0) sBits[q] = m;
            }
        }
        sgr_lds_barrier();

        // ---- pass A for one visit: slot j, chunk position c (both wave-uniform) ----
        auto pass_a = [&](const int j, const int c, const float4 q, const float dx, const float dy, const float power2,
                          const float G, const float alpha) __attribute__((always_inline)) {
            (void)dx; (void)dy;
            const int posj = hi - j;
            const bool hit = (posj < lastc) && !(power2 > 0.0f) && !(alpha < SGR_ALPHA_MIN);  // backward.cu:527-545
            float Gd = 0.0f, wm = 0.0f;
            if (hit) {
                const float4 cc = sC[j];
                const float oma = 1.0f - alpha;
                float inv1ma = __builtin_amdgcn_rcpf(oma);
                inv1ma = fmaf(fmaf(-oma, inv1ma, 1.0f), inv1ma, inv1ma);  // Newton step: T recovery compounds per layer
                T = T * inv1ma;  // T = T / (1 - alpha)
                wm = alpha * T;
                const float one_m_la = 1.0f - last_alpha;
                // the factored channel recurrence of the reduction kernel (see SGR_FACTORED above)
                float u = fmaf(cc.x, dLdC0, dLdA);
                u = fmaf(cc.y, dLdC1, u);
                u = fmaf(cc.z, dLdC2, u);
                u = fmaf(cc.w, dLdD, u);
                Arec = fmaf(last_alpha, u_last, one_m_la * Arec);
                float d = u - Arec;
                u_last = u;
                last_alpha = alpha;
                d *= T;
                const float dopa = fmaf(-T_final * inv1ma, bgdot, d);  // backward.cu:611-614
                Gd = G * dopa;
            }
            (void)q;
            sW[wave][c][lane ^ ((c & 3) << 3)] = make_float2(Gd, wm);
            if (lane == 0) sVis[wave][c] = j;
        };

        // ---- pass B for the nv (<= 8) visits of a chunk ----
        auto pass_b = [&](const int nv) __attribute__((always_inline)) {
            float m0 = 0.f, m1x = 0.f, m1y = 0.f, m2xx = 0.f, m2xy = 0.f, m2yy = 0.f, wabs = 0.f;
            float c0 = 0.f, c1 = 0.f, c2 = 0.f, cD = 0.f;
            const bool act = pv < nv;
            int j = 0;
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (act) {
                j = sVis[wave][pv];
                const float4 a = sA[j];
                q = sB[j];
                const float dx = a.x - bpx;
                const float qa2 = q.x + q.x, qc2 = q.z + q.z;
                const float qbdx = q.y * dx;
                const int swz = (pv & 3) << 3;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int p = 8 * i + pg;
                    const float2 wv = sW[wave][pv][p ^ swz];
                    const float4 dl = sDL[wave][p];
                    const float dy = a.y - (bpy0 + (float)i);
                    const float Gd = wv.x, wmv = wv.y;
                    const float gxx = Gd * dx, gyy = Gd * dy;
                    m0 += Gd;
                    m1x += gxx;
                    m1y += gyy;
                    m2xx = fmaf(gxx, dx, m2xx);
                    m2xy = fmaf(gxx, dy, m2xy);
                    m2yy = fmaf(gyy, dy, m2yy);
                    const float A = fmaf(qa2, dx, q.y * dy), B = fmaf(qc2, dy, qbdx);
                    wabs = fmaf(fabsf(Gd), fmaf(fabsf(A), kx, fabsf(B) * ky), wabs);
                    c0 = fmaf(wmv, dl.x, c0);
                    c1 = fmaf(wmv, dl.y, c1);
                    c2 = fmaf(wmv, dl.z, c2);
                    cD = fmaf(wmv, dl.w, cD);
                }
            }
            // 8 lanes -> 1: the first level pairs a moment with a colour-side value (even bank <- first, odd bank <-
            // second), the other two levels run inside the quads
            sgr_fold4(m0, wabs);
            sgr_fold4(m1x, c0);
            sgr_fold4(m1y, c1);
            sgr_fold4(m2xx, c2);
            sgr_fold4(m2xy, cD);
            sgr_half4(m2yy);
            sgr_quad_sum2(m0, m1x);
            sgr_quad_sum2(m1y, m2xx);
            sgr_quad_sum2(m2xy, m2yy);
            if (act && (lane & 3) == 0) {
                const bool odd = (lane >> 2) & 1;
                // even bank: moments -> mean2D.x, mean2D.y, conic x / y / w, opacity (row 0, 1, 3, 4, 5, 6);
                // odd bank: abs term, colour, depth (row 2, 7, 8, 9, 10)
                const float qw = q.w;
                const float e0 = qw * kx * fmaf(q.x + q.x, m1x, q.y * m1y);
                const float e1 = qw * ky * fmaf(q.z + q.z, m1y, q.y * m1x);
                const float h = -0.5f * qw;
                float* dst = sAcc + (rowset + j * ACCW);
                atomicAdd(&dst[odd ? 2 : 0], odd ? qw * m0 : e0);
                atomicAdd(&dst[odd ? 7 : 1], odd ? m1x : e1);
                atomicAdd(&dst[odd ? 8 : 3], odd ? m1y : h * m2xx);
                atomicAdd(&dst[odd ? 9 : 4], odd ? m2xx : h * m2xy);
                atomicAdd(&dst[odd ? 10 : 5], odd ? m2xy : h * m2yy);
                if (!odd) {
                    atomicAdd(&dst[6], m0);
                    sFlag[j] = 1u;
                }
            }
        };

        uint64_t m = sgr_uniform_u64(sBits[wave]);
        while (m) {
            int nv = 0;
            for (int trip = 0; trip < CH / 2 && m; trip++) {
                const int j0 = __ffsll((unsigned long long)m) - 1;
                m &= m - 1;
                const bool two = m != 0;
                const int j1 = two ? (__ffsll((unsigned long long)m) - 1) : j0;
                m &= m - 1;
                const float4 a0 = sA[j0], q0 = sB[j0];
                const float4 a1 = sA[j1], q1 = sB[j1];
                const float dx0 = a0.x - pxf, dy0 = a0.y - pyf, dx1 = a1.x - pxf, dy1 = a1.y - pyf;
                const float pw0 = sgr_power2(q0.x, q0.y, q0.z, dx0, dy0);
                const float pw1 = sgr_power2(q1.x, q1.y, q1.z, dx1, dy1);
                const float G0 = __builtin_amdgcn_exp2f(pw0), G1 = __builtin_amdgcn_exp2f(pw1);
                const float al0 = fminf(0.99f, q0.w * G0), al1 = fminf(0.99f, q1.w * G1);
                pass_a(j0, nv, q0, dx0, dy0, pw0, G0, al0);
                nv++;
                if (two) {
                    pass_a(j1, nv, q1, dx1, dy1, pw1, G1, al1);
                    nv++;
                }
            }
            __builtin_amdgcn_wave_barrier();  // this wave's own LDS writes above precede its reads below (program order)
            pass_b(nv);
            __builtin_amdgcn_wave_barrier();
        }
        sgr_lds_barrier();
        const uint32_t flags = stager ? sFlag[tid] : 0u;
        if (flags) {
            const uint32_t u = sU[tid];
            touched[u] = 1;
            float4* row = reinterpret_cast<float4*>(partials + (size_t)u * row_stride);
            const float4* src = reinterpret_cast<const float4*>(&sAcc[tid * ACCW]);
            float4 r0 = src[0], r1 = src[1], r2 = src[2];
            if (DET) {
                const float4* s1 = reinterpret_cast<const float4*>(&sAcc[(BATCH + tid) * ACCW]);
                const float4 t0 = s1[0], t1 = s1[1], t2 = s1[2];
                r0.x += t0.x; r0.y += t0.y; r0.z += t0.z; r0.w += t0.w;
                r1.x += t1.x; r1.y += t1.y; r1.z += t1.z; r1.w += t1.w;
                r2.x += t2.x; r2.y += t2.y; r2.z += t2.z; r2.w += t2.w;
            }
            row[0] = r0;
            row[1] = r1;
            row[2] = r2;
        }
    }
}

#endif  // SGR_WITH_VARIANTS

template <int SMAX>
static void launch_bwd(bool cull, bool dpp, bool det, bool v2, bool exact, unsigned tiles, hipStream_t s, const uint2* ranges, const uint32_t* point_list,
                       int W, int H, int S, int gx, int gy, const float* bg, const float4* rec, const uint32_t* u0, const uint64_t* tmask, const float* semantics,
                       const float* alphas,
                       const uint32_t* n_contrib, const uint8_t* hit4, const float* dL_dpix, const float* dL_ddepth,
                       const float* dL_dalpha, const float* dL_dsem, float* partials, int row_stride, uint8_t* touched, uint32_t row_limit,
                       const uint32_t* hlist, const uint32_t* n_contrib_k, const uint32_t* hl_flag) {
    constexpr bool kDet = true;  // every instantiation has the two-row deterministic combine (see SgrBwdBatch)
    if (exact) {
        sgr_blend_bwd_kernel_exact<SMAX><<<tiles, SGR_TILE_THREADS, 0, s>>>(
            ranges, point_list, W, H, S, gx, gy, bg, rec, u0, tmask, semantics, alphas, n_contrib, hit4, dL_dpix, dL_ddepth, dL_dalpha,
            dL_dsem, partials, row_stride, touched, row_limit, hlist, n_contrib_k, hl_flag);
        return;
    }
#if SGR_WITH_VARIANTS
    if constexpr (SMAX == 0) {
        if (v2 && dpp) {  // transposed accumulation (S = 0)
#define SGR_V2(C, D) sgr_blend_bwd_kernel_v2<C, D><<<tiles, SGR_TILE_THREADS, 0, s>>>(                                    \
            ranges, point_list, W, H, S, gx, gy, bg, rec, u0, tmask, semantics, alphas, n_contrib, hit4, dL_dpix, dL_ddepth,         \
            dL_dalpha, dL_dsem, partials, row_stride, touched, row_limit, hlist, n_contrib_k, hl_flag)
            if (cull) { if (det) SGR_V2(true, true); else SGR_V2(true, false); }
            else { if (det) SGR_V2(false, true); else SGR_V2(false, false); }
#undef SGR_V2
            return;
        }
    }
#else
    (void)v2;
#endif
    if (SMAX > 4) { cull = true; dpp = true; }  // the A/B switches (tests) exist for the small instantiations only
#define SGR_GO(C, D)                                                                                                 \
    do {                                                                                                             \
        if constexpr (SMAX == 0) {                                                                                   \
            if (det)                                                                                                 \
                sgr_blend_bwd_kernel_s0<C, D, true><<<tiles, SGR_TILE_THREADS, 0, s>>>(                               \
                    ranges, point_list, W, H, S, gx, gy, bg, rec, u0, tmask, semantics, alphas, n_contrib, hit4, dL_dpix, dL_ddepth,    \
                    dL_dalpha, dL_dsem, partials, row_stride, touched, row_limit, hlist, n_contrib_k, hl_flag);                                               \
            else                                                                                                     \
                sgr_blend_bwd_kernel_s0<C, D, false><<<tiles, SGR_TILE_THREADS, 0, s>>>(                              \
                    ranges, point_list, W, H, S, gx, gy, bg, rec, u0, tmask, semantics, alphas, n_contrib, hit4, dL_dpix, dL_ddepth,    \
                    dL_dalpha, dL_dsem, partials, row_stride, touched, row_limit, hlist, n_contrib_k, hl_flag);                                               \
        } else if (kDet && det)                                                                                      \
            sgr_blend_bwd_kernel<SMAX, C, D, kDet><<<tiles, SGR_TILE_THREADS, 0, s>>>(                                \
                ranges, point_list, W, H, S, gx, gy, bg, rec, u0, tmask, semantics, alphas, n_contrib, hit4, dL_dpix, dL_ddepth,        \
                dL_dalpha, dL_dsem, partials, row_stride, touched, row_limit, hlist, n_contrib_k, hl_flag);                                                   \
        else                                                                                                         \
            sgr_blend_bwd_kernel<SMAX, C, D, false><<<tiles, SGR_TILE_THREADS, 0, s>>>(                               \
                ranges, point_list, W, H, S, gx, gy, bg, rec, u0, tmask, semantics, alphas, n_contrib, hit4, dL_dpix, dL_ddepth,        \
                dL_dalpha, dL_dsem, partials, row_stride, touched, row_limit, hlist, n_contrib_k, hl_flag);                                                   \
    } while (0)
    if (cull && dpp) SGR_GO(true, true);
    else if constexpr (SMAX <= 4) {
        if (cull) SGR_GO(true, false);
        else if (dpp) SGR_GO(false, true);
        else SGR_GO(false, false);
    }
#undef SGR_GO
}

// floats per partial row for S semantic channels: the kernel's SMAX bucket writes ceil((11+SMAX)/4) float4
// = NVAL of the instantiation that serves S (12 floats = 48 bytes at S = 0: the rows are only 16-byte aligned; padding
// them to 64 bytes cost 25 % more row traffic in this kernel's stores and in the per-Gaussian row sum's loads)
int sgr_partial_row_stride(int S) {
    const int smax = S == 0 ? 0 : (S <= 4 ? 4 : (S <= 8 ? 8 : (S <= 12 ? 12 : (S <= 16 ? 16 : (S <= 20 ? 20 : (S <= 24 ? 24 : 32))))));
    return (SGR_ROW_BASE + smax + 3) / 4 * 4;
}

void sgr_launch_blend_bwd(bool cull, bool dpp, bool det, bool v2, bool exact, int gx, int gy, const uint2* ranges, const uint32_t* point_list, int W,
                          int H, int S, const float* bg, const float4* rec, const uint32_t* u0, const uint64_t* tmask, const float* semantics, const float* alphas, const uint32_t* n_contrib,
                          const uint8_t* hit4, const float* dL_dpix, const float* dL_ddepth, const float* dL_dalpha,
                          const float* dL_dsem, float* partials, uint8_t* touched, uint32_t row_limit, const uint32_t* hlist,
                          const uint32_t* n_contrib_k, const uint32_t* hl_flag, hipStream_t s) {
    if (gx <= 0 || gy == 0) return;
    const unsigned tiles = sgr_xcd_grid_blocks(gx, gy < 0 ? -gy : gy);  // supertile-ordered grid incl. padding blocks
    const int stride = sgr_partial_row_stride(S);
#define SGR_BWD(N) launch_bwd<N>(cull, dpp, det, v2, exact, tiles, s, ranges, point_list, W, H, S, gx, gy, bg, rec, u0, tmask, semantics, alphas, \
                                 n_contrib, hit4, dL_dpix, dL_ddepth, dL_dalpha, dL_dsem, partials, stride, touched, row_limit, hlist, n_contrib_k, hl_flag)
    if (S == 0) SGR_BWD(0);
    else if (S <= 4) SGR_BWD(4);
    else if (S <= 8) SGR_BWD(8);
    else if (S <= 12) SGR_BWD(12);
    else if (S <= 16) SGR_BWD(16);
    else if (S <= 20) SGR_BWD(20);
    else if (S <= 24) SGR_BWD(24);
    else SGR_BWD(32);
#undef SGR_BWD
}
