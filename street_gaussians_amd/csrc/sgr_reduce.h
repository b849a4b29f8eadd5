// sgr_reduce.h -- wave64 cross-lane reductions of the blend backward kernels (gfx950): DPP wave sums, the
// reduce-scatter built on v_permlane32_swap / v_permlane16_swap, and its folded row stage.  Shared by
// sgr_blend_bwd.hip (LDS-staged walk) and sgr_blend_bwd_sw.hip (scalar walk); sgr_test_wave_sum checks them.
#pragma once
#include <hip/hip_runtime.h>

// Sum over the 64 lanes of a wave; the result is valid in lanes 48..63 (read it from lane 63).
__device__ __forceinline__ float sgr_wave_sum_dpp(float v) {
#define SGR_DPP_ADD(ctrl, rmask)                                                                              \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xF, false))
    SGR_DPP_ADD(0xB1, 0xF);   // quad_perm [1,0,3,2]
    SGR_DPP_ADD(0x4E, 0xF);   // quad_perm [2,3,0,1]
    SGR_DPP_ADD(0x141, 0xF);  // row_half_mirror
    SGR_DPP_ADD(0x140, 0xF);  // row_mirror      -> every lane of a 16-lane row holds the row sum
    SGR_DPP_ADD(0x142, 0xA);  // row_bcast15     -> rows 1,3 += previous row
    SGR_DPP_ADD(0x143, 0xC);  // row_bcast31     -> rows 2,3 += rows 0+1
#undef SGR_DPP_ADD
    return v;
}

__device__ __forceinline__ float sgr_wave_sum_shfl(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <bool DPP>
__device__ __forceinline__ float sgr_wave_sum(float v) {
    return DPP ? sgr_wave_sum_dpp(v) : sgr_wave_sum_shfl(v);
}

// Four wave sums at once, written as v_add_f32_dpp so that each step is ONE instruction (hipcc lowers the
// update_dpp builtin to v_mov 0 + v_mov_dpp + (SLP-packed) v_pk_add_f32 = 3 instructions per step).  The four
// chains are interleaved, so every DPP read is >= 3 instructions behind the write of its register (the
// "VALU write -> DPP read" hazard needs 2 wait states and nothing pads the inside of an asm statement); the
// leading s_nop covers values produced just before the statement.  Results are valid in lanes 48..63.
#define SGR_DPP4(ctrl)                                                                             \
    "v_add_f32_dpp %0, %0, %0 " ctrl "\n\tv_add_f32_dpp %1, %1, %1 " ctrl "\n\tv_add_f32_dpp %2, %2, %2 " ctrl \
    "\n\tv_add_f32_dpp %3, %3, %3 " ctrl "\n\t"
__device__ __forceinline__ void sgr_wave_sum4(float& a, float& b, float& c, float& d) {
    asm volatile("s_nop 1\n\t" SGR_DPP4("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 SGR_DPP4("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 SGR_DPP4("row_half_mirror row_mask:0xf bank_mask:0xf")
                 SGR_DPP4("row_mirror row_mask:0xf bank_mask:0xf")
                 SGR_DPP4("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 SGR_DPP4("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 0"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
#undef SGR_DPP4

// ---- wave64 REDUCE-SCATTER of NVAL values (NVAL % 4 == 0) -----------------------------------------------------
// Summing NVAL values over 64 lanes one by one costs 6 DPP adds each.  Instead halve the number of live registers
// at every cross-half step: v_permlane32_swap (gfx950) exchanges the upper half of A with the lower half of B, so
// A+B leaves value A's pair sums in lanes 0-31 and value B's in lanes 32-63 -- two values, one register.
// v_permlane16_swap does the same across 16-lane rows.  After the two stages register t holds, in row k, partial
// sums of value 4t + {0,2,1,3}[k]; a 4-step DPP row reduction (on NVAL/4 registers only) finishes the job:
//   NVAL = 12:  6+3 swaps, 9 adds, 12 DPP adds = 30 instructions instead of 72.
__device__ __forceinline__ void sgr_swap32(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    a = __builtin_bit_cast(float, (unsigned)r[0]);
    b = __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ void sgr_swap16(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    a = __builtin_bit_cast(float, (unsigned)r[0]);
    b = __builtin_bit_cast(float, (unsigned)r[1]);
}
// row (16-lane) sums, every lane of the row ends up with the row total; chains interleaved as in sgr_wave_sum4
#define SGR_ROW4(ctrl)                                                                              \
    "v_add_f32_dpp %0, %0, %0 " ctrl "\n\tv_add_f32_dpp %1, %1, %1 " ctrl "\n\tv_add_f32_dpp %2, %2, %2 " ctrl \
    "\n\tv_add_f32_dpp %3, %3, %3 " ctrl "\n\t"
#define SGR_ROW3(ctrl)                                                                              \
    "v_add_f32_dpp %0, %0, %0 " ctrl "\n\tv_add_f32_dpp %1, %1, %1 " ctrl "\n\tv_add_f32_dpp %2, %2, %2 " ctrl "\n\t"
#define SGR_ROW2(ctrl) "v_add_f32_dpp %0, %0, %0 " ctrl "\n\tv_add_f32_dpp %1, %1, %1 " ctrl "\n\ts_nop 0\n\t"
#define SGR_ROWSTEPS(M)                                                                                        \
    "s_nop 1\n\t" M("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") M("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf") \
        M("row_half_mirror row_mask:0xf bank_mask:0xf") M("row_mirror row_mask:0xf bank_mask:0xf") "s_nop 0"
__device__ __forceinline__ void sgr_row_sum4(float& a, float& b, float& c, float& d) {
    asm volatile(SGR_ROWSTEPS(SGR_ROW4) : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void sgr_row_sum3(float& a, float& b, float& c) {
    asm volatile(SGR_ROWSTEPS(SGR_ROW3) : "+v"(a), "+v"(b), "+v"(c));
}
__device__ __forceinline__ void sgr_row_sum2(float& a, float& b) { asm volatile(SGR_ROWSTEPS(SGR_ROW2) : "+v"(a), "+v"(b)); }
#undef SGR_ROW4
#undef SGR_ROW3
#undef SGR_ROW2
#undef SGR_ROWSTEPS

// in: v[NVAL] per lane.  out: r[NVAL/4]; in row k (= lane >> 4) r[t] is the wave total of value 4t + {0,2,1,3}[k].
template <int NVAL>
__device__ __forceinline__ void sgr_wave_reduce_scatter(float (&v)[NVAL], float (&r)[NVAL / 4]) {
    static_assert(NVAL % 4 == 0, "pad the value count to a multiple of 4");
    float h[NVAL / 2];
#pragma unroll
    for (int p = 0; p < NVAL / 2; p++) {
        sgr_swap32(v[2 * p], v[2 * p + 1]);
        h[p] = v[2 * p] + v[2 * p + 1];
    }
#pragma unroll
    for (int t = 0; t < NVAL / 4; t++) {
        sgr_swap16(h[2 * t], h[2 * t + 1]);
        r[t] = h[2 * t] + h[2 * t + 1];
    }
    constexpr int N = NVAL / 4;
    int t = 0;
#pragma unroll
    for (; t + 4 <= N && (N - t) != 5; t += 4) sgr_row_sum4(r[t], r[t + 1], r[t + 2], r[t + 3]);
#pragma unroll
    for (; t + 3 <= N; t += 3) sgr_row_sum3(r[t], r[t + 1], r[t + 2]);
    if (t + 2 == N) sgr_row_sum2(r[t], r[t + 1]);
    static_assert(N != 1, "unsupported value count");
}

// ---- FOLDED row stage -----------------------------------------------------------------------------------------------
// The row steps above spend 4 DPP adds per register although each register's row holds ONE value (16 partials of it).
// Folding keeps every lane busy instead: a DPP add with a bank mask puts the half-row sums of register a into lanes
// 0-7 and those of register b into lanes 8-15 (two instructions, two registers -> one), then quarter rows the same
// way, and only the last two steps (inside a quad) run on ceil(N/4) registers:
//   NVAL = 12:  3 + 2 + 2 = 7 DPP adds instead of 12, and ONE ds_add_f32 (16 lanes) per hit instead of three.
// VALU write -> DPP read needs two wait states and nothing pads the inside of an asm statement: every statement
// opens with s_nop 1 (the wave idles, the SIMD does not: its other waves issue meanwhile).
//   after fold8(a, b):  lanes 0-7 of each row: a[i] + a[i^8];  lanes 8-15: b[i] + b[i^8]
__device__ __forceinline__ void sgr_fold8(float& a, const float b) {
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc"
                 : "+v"(a)
                 : "v"(b));
}
__device__ __forceinline__ void sgr_self8(float& a) {  // odd register left over: both halves get a[i] + a[i^8]
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a));
}
//   after fold4(a, b):  banks 0 and 2 (lanes 0-3, 8-11): a[i] + a[i+4];  banks 1 and 3: b[i-4] + b[i]
__device__ __forceinline__ void sgr_fold4(float& a, const float b) {
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa"
                 : "+v"(a)
                 : "v"(b));
}
__device__ __forceinline__ void sgr_half4(float& a) {  // odd register left over: banks 0 and 2 only
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5" : "+v"(a));
}
__device__ __forceinline__ void sgr_quad_sum(float& a) {  // every lane of a quad gets the quad's sum
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 0"
                 : "+v"(a));
}
__device__ __forceinline__ void sgr_quad_sum2(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 0"
                 : "+v"(a), "+v"(b));
}
// in: v[NVAL] per lane.  out: g[(NVAL/4 + 3) / 4].  With t = 4i + {0,2,1,3}[bank] (bank = (lane >> 2) & 3) and
// k = lane >> 4, every lane of that bank of g[i] holds the wave total of value 4t + {0,2,1,3}[k] when t < NVAL/4
// (the remaining banks hold duplicates or garbage and must be ignored).
template <int NVAL>
__device__ __forceinline__ void sgr_wave_reduce_fold(float (&v)[NVAL], float (&g)[(NVAL / 4 + 3) / 4]) {
    static_assert(NVAL % 4 == 0, "pad the value count to a multiple of 4");
    constexpr int N = NVAL / 4, NF = (N + 1) / 2, NG = (NF + 1) / 2;
    float h[NVAL / 2], r[N];
#pragma unroll
    for (int p = 0; p < NVAL / 2; p++) {
        sgr_swap32(v[2 * p], v[2 * p + 1]);
        h[p] = v[2 * p] + v[2 * p + 1];
    }
#pragma unroll
    for (int t = 0; t < N; t++) {
        sgr_swap16(h[2 * t], h[2 * t + 1]);
        r[t] = h[2 * t] + h[2 * t + 1];
    }
    // half rows: f[i] = r[2i] (lanes 0-7: t = 2i, lanes 8-15: t = 2i+1)
#pragma unroll
    for (int i = 0; i < N / 2; i++) sgr_fold8(r[2 * i], r[2 * i + 1]);
    if (N & 1) sgr_self8(r[N - 1]);
    // quarter rows: g[i] = f[2i] (banks 0, 2 <- f[2i]; banks 1, 3 <- f[2i+1])
#pragma unroll
    for (int i = 0; i < NG; i++) {
        if (2 * i + 1 < NF) sgr_fold4(r[4 * i], r[4 * i + 2 < N ? 4 * i + 2 : N - 1]);
        else sgr_half4(r[4 * i]);
        g[i] = r[4 * i];
    }
    int i = 0;
#pragma unroll
    for (; i + 2 <= NG; i += 2) sgr_quad_sum2(g[i], g[i + 1]);
    if (i < NG) sgr_quad_sum(g[i]);
}


// ---- TWO visits at once: 24 values, the whole row stage as ONE statement ------------------------------------------
// sgr_wave_reduce_fold<12> spends 9 wait states per visit around its four DPP statements (VALU write -> DPP read needs
// two, and nothing pads the inside of an asm statement); with the six registers of two visits in one statement the
// dependent instructions are two or more apart by construction: 13 v_add_f32_dpp and 3 s_nop for TWO visits.
// in: v[24] (values 0-11 of the first visit, 12-23 of the second).  out: g0, g1.  With bank = (lane >> 2) & 3,
// k = lane >> 4 and perm = {0,2,1,3}: every lane of bank b of g0 holds the wave total of value 4*perm[b] + perm[k];
// every lane of banks 0 and 2 of g1 holds value 16 + 4*perm[b] + perm[k] (banks 1, 3 of g1: garbage).
__device__ __forceinline__ void sgr_wave_reduce_fold24(float (&v)[24], float& g0, float& g1) {
    float h[12], r[6];
#pragma unroll
    for (int p = 0; p < 12; p++) {
        sgr_swap32(v[2 * p], v[2 * p + 1]);
        h[p] = v[2 * p] + v[2 * p + 1];
    }
#pragma unroll
    for (int t = 0; t < 6; t++) {
        sgr_swap16(h[2 * t], h[2 * t + 1]);
        r[t] = h[2 * t] + h[2 * t + 1];
    }
#define SGR_RM " row_mask:0xf bank_mask:"
    asm volatile(
        "s_nop 1\n\t"
        // half rows: r0 <- (r0 | r1), r2 <- (r2 | r3), r4 <- (r4 | r5)
        "v_add_f32_dpp %0, %0, %0 row_ror:8" SGR_RM "0x3\n\t"
        "v_add_f32_dpp %2, %2, %2 row_ror:8" SGR_RM "0x3\n\t"
        "v_add_f32_dpp %4, %4, %4 row_ror:8" SGR_RM "0x3\n\t"
        "v_add_f32_dpp %0, %1, %1 row_ror:8" SGR_RM "0xc\n\t"
        "v_add_f32_dpp %2, %3, %3 row_ror:8" SGR_RM "0xc\n\t"
        "v_add_f32_dpp %4, %5, %5 row_ror:8" SGR_RM "0xc\n\t"
        // quarter rows: r0 <- (r0 | r2) by banks, r4 alone (banks 0 and 2)
        "v_add_f32_dpp %0, %0, %0 row_shl:4" SGR_RM "0x5\n\t"
        "v_add_f32_dpp %0, %2, %2 row_shr:4" SGR_RM "0xa\n\t"
        "v_add_f32_dpp %4, %4, %4 row_shl:4" SGR_RM "0x5\n\t"
        "s_nop 0\n\t"
        // inside the quads
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1]" SGR_RM "0xf\n\t"
        "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1]" SGR_RM "0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2]" SGR_RM "0xf\n\t"
        "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2]" SGR_RM "0xf\n\t"
        "s_nop 0"
        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]));
#undef SGR_RM
    g0 = r[0];
    g1 = r[4];
}
