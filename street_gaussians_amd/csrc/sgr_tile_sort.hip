// sgr_tile_sort.hip -- per-tile radix sort of the instance lists by depth, in LDS (north_star: "tile binning + per-tile radix
// sort by depth ... LDS-staged per-tile Gaussian lists").  An A/B form of the binning chain behind sgr_test_switches bit 12
// (SGR_TILE_SORT=1); what the reference does with ONE global sort of 64-bit (tile | depth) keys
// (/root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer/rasterizer_impl.cu:286-321).
//
// Default chain: depth pre-sort of the P Gaussians (3-4 global passes), emission in depth order, stable sort on the tile id
// (2 global passes over R).  This form drops the depth pre-sort: instances are emitted in INDEX order, the same two stable
// tile passes bring every tile's instances together in ascending Gaussian id, and one more stage sorts each tile's list by
// its 27-bit depth key inside LDS -- a stable LSD radix sort (3 passes of 9 bits; 4 when a depth beyond the 27-bit keys was
// seen), so equal depths keep ascending id: exactly the order of the reference's 64-bit keys, ties included.
//
// A wave owns a CONTIGUOUS run of the list and holds it in registers during a pass (memory order = (wave, step, lane)).
// Ranking, once per key and pass: ONE returning LDS atomic on a zeroed per-(wave, digit) counter -- the value that comes back
// is the key's rank among the wave's keys of that digit so far, the counter ends as the digit's count; a scan of the counters
// gives the digit's first slot for that wave and position = first slot + rank.  Stable because a wave's steps are issued in
// order and, within ONE ds_add_rtn_u32, the lanes that hit the same counter are served in ascending lane order (checked on
// the device: sgr_test_lds_atomic_order, tests/test_gpu_primitives.py; end to end by the equal-depth cases of
// tests/test_gpu_tile_sort.py).  Measured on MI355X (profiles/r6/ab_tile_sort.md), the stage at 4.8 M / 7.8 M / 28 M
// instances: 65 / 88 / 295 us -- bound by the LDS atomic unit, ~2.5 clocks per lane: passes x R x 2.5 / 256 CUs.  The
// alternatives built and measured at 4.8 M: a wave64 ballot match instead of the atomic (-DSGR_TS_ATOMIC=0: 9 ballots leave
// the lanes sharing the digit, the group's lowest lane advances the counter) 76 us, 93 us with its select on VCC; atomics on
// even and ballots on odd steps (-DSGR_TS_MIX=1) 68 us; a histogram atomic + a ranking atomic per key 73 us; workgroup-wide
// two-sweep passes with barriers 137 us.
//   lists of up to 1024 entries   one WAVE per tile (four tiles per workgroup), no barriers;
//   up to 4096 / 8192 entries     8 / 16 waves per tile, three barriers per pass;
//   beyond                        the same passes over a ping-pong in HBM (the tile's own slices of the binning buffer).
#include "sgr_common.h"

#define SGR_TS_BITS 9
#define SGR_TS_NB (1 << SGR_TS_BITS)
#define SGR_TS_STEPS_W 16  // one wave per tile: up to 16 x 64 entries in registers
#define SGR_TS_WAVE_CAP (64 * SGR_TS_STEPS_W)
#define SGR_TS_STEPS 8     // workgroup per tile: 8 x 64 entries per wave
#define SGR_TS_MED_WAVES 8
#define SGR_TS_BIG_WAVES 16
#define SGR_TS_MED_CAP (SGR_TS_MED_WAVES * 64 * SGR_TS_STEPS)
#define SGR_TS_BIG_CAP (SGR_TS_BIG_WAVES * 64 * SGR_TS_STEPS)

#ifndef SGR_TS_ATOMIC
#define SGR_TS_ATOMIC 1  // 1: rank with one returning LDS atomic per key (the fastest form measured); 0: wave64 ballot match
#endif
// Rank of this lane's key among the wave's keys of digit d seen so far (earlier steps + lower lanes), and the counter moved on
// by the step's count.  Every lane of the wave calls it (valid = holds a key).  Wave64 ballot match: BITS ballots leave the set
// of lanes that share the digit; all of them read the counter, the lowest one writes it back advanced by the group's size.
#ifndef SGR_TS_MIX
#define SGR_TS_MIX 0  // 1 (with SGR_TS_ATOMIC=0): even steps rank with the returning LDS atomic, odd steps with the ballot match -- both advance the same
#endif                // counters, and the two forms run on different pipes of the CU (LDS atomic unit / VALU)
__device__ __forceinline__ uint32_t sgr_ts_count_rank(uint32_t* c, uint32_t d, bool valid, int lane, bool atomic_step) {
#if SGR_TS_ATOMIC
    (void)lane; (void)atomic_step;
    return valid ? atomicAdd(&c[d], 1u) : 0u;
#else
    if (SGR_TS_MIX && atomic_step) return valid ? atomicAdd(&c[d], 1u) : 0u;
    // the lanes that share this lane's digit: per bit, keep the lanes whose bit equals mine -- m &= ballot XNOR (my bit spread
    // over the word).  Written on 32-bit halves with v_bfe_i32 / v_xnor / v_and: a `bit ? bal : ~bal` select compiles to
    // v_cndmask on VCC, 23 cycles per wave instruction back to back on gfx950 (profiles/r5/valu_rates2.jsonl)
    const uint64_t vm = __builtin_amdgcn_ballot_w64(valid);
    uint32_t mlo = (uint32_t)vm, mhi = (uint32_t)(vm >> 32);
#pragma unroll
    for (int b = 0; b < SGR_TS_BITS; b++) {
        const int y = __builtin_amdgcn_sbfe((int)d, b, 1);  // 0 or -1
        const uint64_t bal = __builtin_amdgcn_ballot_w64(y != 0);  // (lanes without a key are not in vm: their bit does not matter)
        mlo &= ~((uint32_t)bal ^ (uint32_t)y);
        mhi &= ~((uint32_t)(bal >> 32) ^ (uint32_t)y);
    }
    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));  // set bits of m below this lane
    const uint64_t m = ((uint64_t)mhi << 32) | mlo;
    const uint32_t before = valid ? c[d] : 0u;
    __builtin_amdgcn_wave_barrier();
    if (valid && prefix == 0) c[d] = before + (uint32_t)__popcll(m);
    __builtin_amdgcn_wave_barrier();
    return before + prefix;
#endif
}

__device__ __forceinline__ int sgr_ts_npass(const uint32_t* header) {
    return header[2] != 0u ? (32 + SGR_TS_BITS - 1) / SGR_TS_BITS : (SGR_DEPTH_KEY_BITS + SGR_TS_BITS - 1) / SGR_TS_BITS;
}

// ---- one wave per tile -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
sgr_tile_sort_wave_kernel(int T, const uint2* __restrict__ ranges, const uint32_t* __restrict__ vals_in,
                          uint32_t* __restrict__ vals_out, const uint32_t* __restrict__ dkeys,
                          const uint32_t* __restrict__ header) {
    __shared__ uint2 sKV[4][SGR_TS_WAVE_CAP];
    __shared__ uint32_t cnt[4][SGR_TS_NB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= T) return;
    const uint2 rg = ranges[tile];
    const uint32_t n = rg.y - rg.x;
    if (n == 0u || n > (uint32_t)SGR_TS_WAVE_CAP) return;  // longer lists: sgr_tile_sort_block_kernel
    const uint32_t* in = vals_in + rg.x;
    uint32_t* out = vals_out + rg.x;
    if (n == 1u) {
        if (lane == 0) out[0] = in[0];
        return;
    }
    const int npass = sgr_ts_npass(header);
    uint2 (&kv)[SGR_TS_WAVE_CAP] = sKV[wave];
    uint32_t (&c)[SGR_TS_NB] = cnt[wave];
    uint32_t key[SGR_TS_STEPS_W], val[SGR_TS_STEPS_W], rnk[SGR_TS_STEPS_W];
#pragma unroll
    for (int s = 0; s < SGR_TS_STEPS_W; s++) {
        const uint32_t i = s * 64 + lane;
        val[s] = (s * 64u < n && i < n) ? in[i] : 0u;
    }
#pragma unroll
    for (int s = 0; s < SGR_TS_STEPS_W; s++) {
        const uint32_t i = s * 64 + lane;
        key[s] = (s * 64u < n && i < n) ? dkeys[val[s] & ~SGR_DEAD] : 0u;
    }
    for (int p = 0; p < npass; p++) {
        const int shift = p * SGR_TS_BITS;
        const bool last = p == npass - 1;
#pragma unroll
        for (int i = 0; i < SGR_TS_NB / 64; i++) c[i * 64 + lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        // rank inside the digit = what the returning atomic hands back (steps in order, lanes in order)
#pragma unroll
        for (int s = 0; s < SGR_TS_STEPS_W; s++) {
            rnk[s] = 0u;
            if (s * 64u < n)  // wave-uniform
                rnk[s] = sgr_ts_count_rank(c, (key[s] >> shift) & (SGR_TS_NB - 1), s * 64u + lane < n, lane, (s & 1) == 0);
        }
        __builtin_amdgcn_wave_barrier();
        {   // counts -> first slots: lane l owns bins [l * NB / 64, (l + 1) * NB / 64)
            constexpr int BPL = SGR_TS_NB / 64;
            uint32_t cc[BPL], sum = 0;
#pragma unroll
            for (int j = 0; j < BPL; j++) {
                cc[j] = c[lane * BPL + j];
                sum += cc[j];
            }
            uint32_t run = sgr_wave_incl_scan(sum, lane) - sum;
#pragma unroll
            for (int j = 0; j < BPL; j++) {
                c[lane * BPL + j] = run;
                run += cc[j];
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < SGR_TS_STEPS_W; s++) {
            if (s * 64u < n && s * 64u + lane < n) {
                const uint32_t pos = c[(key[s] >> shift) & (SGR_TS_NB - 1)] + rnk[s];
                if (last) out[pos] = val[s];
                else kv[pos] = make_uint2(key[s], val[s]);
            }
        }
        if (last) break;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < SGR_TS_STEPS_W; s++) {
            if (s * 64u < n && s * 64u + lane < n) {
                const uint2 t = kv[s * 64 + lane];
                key[s] = t.x;
                val[s] = t.y;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- one workgroup per tile --------------------------------------------------------------------------------------------
template <int WAVES>
__device__ __forceinline__ uint32_t sgr_ts_block_excl_scan(uint32_t v, uint32_t* sw, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = sgr_wave_incl_scan(v, lane);
    if (lane == 63) sw[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WAVES; w++) {
        const uint32_t t = sw[w];
        base += w < wave ? t : 0u;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

// cnt[w][d] (per-wave digit counts) -> the first slot of digit d for wave w: digits ascending, waves ascending inside a digit
template <int WAVES>
__device__ __forceinline__ void sgr_ts_starts(uint32_t (*cnt)[SGR_TS_NB], uint32_t* sw) {
    constexpr int THREADS = 64 * WAVES, BPT = SGR_TS_NB > THREADS ? SGR_TS_NB / THREADS : 1;
    const int tid = threadIdx.x;
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < BPT; j++) {
        const int bin = tid * BPT + j;
        if (bin < SGR_TS_NB) {
#pragma unroll
            for (int w = 0; w < WAVES; w++) sum += cnt[w][bin];
        }
    }
    uint32_t total;
    uint32_t run = sgr_ts_block_excl_scan<WAVES>(sum, sw, total);
#pragma unroll
    for (int j = 0; j < BPT; j++) {
        const int bin = tid * BPT + j;
        if (bin < SGR_TS_NB) {
#pragma unroll
            for (int w = 0; w < WAVES; w++) {
                const uint32_t t = cnt[w][bin];
                cnt[w][bin] = run;
                run += t;
            }
        }
    }
}

// BIG = false: 8 waves, lists of SGR_TS_WAVE_CAP + 1 .. SGR_TS_MED_CAP entries; BIG = true: 16 waves, everything longer (in LDS
// up to SGR_TS_BIG_CAP entries, beyond that on a ping-pong in HBM: (gk0, vals_in) <-> (gk1, vals_out), the tile's own slices;
// vals_in is overwritten for those tiles).
template <bool BIG>
__global__ void __launch_bounds__(BIG ? 64 * SGR_TS_BIG_WAVES : 64 * SGR_TS_MED_WAVES)
sgr_tile_sort_block_kernel(const uint2* __restrict__ ranges, uint32_t* vals_in, uint32_t* __restrict__ vals_out,
                           const uint32_t* __restrict__ dkeys, const uint32_t* __restrict__ header, uint32_t* __restrict__ gk0,
                           uint32_t* __restrict__ gk1) {
    constexpr int WAVES = BIG ? SGR_TS_BIG_WAVES : SGR_TS_MED_WAVES, THREADS = 64 * WAVES;
    constexpr int CAP = WAVES * 64 * SGR_TS_STEPS;
    __shared__ uint2 sKV[CAP];
    __shared__ uint32_t cnt[WAVES][SGR_TS_NB];
    __shared__ uint32_t sw[WAVES];
    const uint2 rg = ranges[blockIdx.x];
    const uint32_t n = rg.y - rg.x;
    if (n <= (uint32_t)(BIG ? SGR_TS_MED_CAP : SGR_TS_WAVE_CAP) || (!BIG && n > (uint32_t)SGR_TS_MED_CAP)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int npass = sgr_ts_npass(header);
    const uint32_t* in = vals_in + rg.x;
    uint32_t* out = vals_out + rg.x;
    // wave w owns entries [w * seg, (w + 1) * seg), seg a multiple of 64
    const uint32_t seg = ((n + THREADS - 1) / THREADS) * 64u;
    const uint32_t s0 = min(n, (uint32_t)wave * seg), s1 = min(n, s0 + seg);
    if (n <= (uint32_t)CAP) {  // seg <= 64 * SGR_TS_STEPS: a wave's run fits its registers
        uint32_t key[SGR_TS_STEPS], val[SGR_TS_STEPS], rnk[SGR_TS_STEPS];
#pragma unroll
        for (int s = 0; s < SGR_TS_STEPS; s++) {
            const uint32_t i = s0 + s * 64 + lane;
            val[s] = i < s1 ? in[i] : 0u;
        }
#pragma unroll
        for (int s = 0; s < SGR_TS_STEPS; s++) key[s] = (s0 + s * 64 + lane) < s1 ? dkeys[val[s] & ~SGR_DEAD] : 0u;
        for (int p = 0; p < npass; p++) {
            const int shift = p * SGR_TS_BITS;
            const bool last = p == npass - 1;
#pragma unroll
            for (int i = 0; i < SGR_TS_NB / 64; i++) cnt[wave][i * 64 + lane] = 0u;  // every wave clears its own counters
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < SGR_TS_STEPS; s++) {
                rnk[s] = 0u;
                if (s0 + s * 64 < s1)  // wave-uniform
                    rnk[s] = sgr_ts_count_rank(cnt[wave], (key[s] >> shift) & (SGR_TS_NB - 1), s0 + s * 64 + lane < s1, lane, (s & 1) == 0);
            }
            __syncthreads();
            sgr_ts_starts<WAVES>(cnt, sw);
            __syncthreads();
#pragma unroll
            for (int s = 0; s < SGR_TS_STEPS; s++) {
                if (s0 + s * 64 + lane < s1) {
                    const uint32_t pos = cnt[wave][(key[s] >> shift) & (SGR_TS_NB - 1)] + rnk[s];
                    if (last) out[pos] = val[s];
                    else sKV[pos] = make_uint2(key[s], val[s]);
                }
            }
            if (last) break;
            __syncthreads();
#pragma unroll
            for (int s = 0; s < SGR_TS_STEPS; s++) {
                const uint32_t i = s0 + s * 64 + lane;
                if (i < s1) {
                    const uint2 t = sKV[i];
                    key[s] = t.x;
                    val[s] = t.y;
                }
            }
            // (sKV is next written behind the two barriers of the next pass: every wave has reloaded by then)
        }
    } else if (BIG) {
        // longer than LDS holds: the same passes over HBM, any length; the rank of an entry waits in the output slice of the
        // OTHER array pair between the counting and the placing sweep (rk), which is free at that point
        uint32_t* ks = gk0 + rg.x;
        uint32_t* vs = vals_in + rg.x;
        uint32_t* kd = gk1 + rg.x;
        uint32_t* vd = out;
        for (uint32_t i = tid; i < n; i += THREADS) ks[i] = dkeys[vs[i] & ~SGR_DEAD];
        __syncthreads();
        for (int p = 0; p < npass; p++) {
            const int shift = p * SGR_TS_BITS;
#pragma unroll
            for (int i = 0; i < SGR_TS_NB / 64; i++) cnt[wave][i * 64 + lane] = 0u;
            __builtin_amdgcn_wave_barrier();
            uint32_t* rk = kd;  // ranks parked at the entry's OWN index in the destination key array (overwritten below, after use)
            for (uint32_t base = s0; base < s1; base += 64) {
                const uint32_t i = base + lane;
                const bool valid = i < s1;
                const uint32_t r = sgr_ts_count_rank(cnt[wave], ((valid ? ks[i] : 0u) >> shift) & (SGR_TS_NB - 1), valid, lane, true);
                if (valid) rk[i] = r;
            }
            __syncthreads();
            sgr_ts_starts<WAVES>(cnt, sw);
            __syncthreads();
            // placing: entry i goes to first slot + rank.  rk[i] and kd[pos] live in the same array: read every rank of the
            // wave's chunk before any wave stores (two sweeps with a barrier between would cost a pass over HBM; instead the
            // positions are computed into vd first -- vd[i] = pos -- and the move happens in a third sweep)
            for (uint32_t i = s0 + lane; i < s1; i += 64) vd[i] = cnt[wave][(ks[i] >> shift) & (SGR_TS_NB - 1)] + rk[i];
            __syncthreads();
            // now kd is free (ranks consumed); positions sit in vd[i]: move keys first, then values through a register
            for (uint32_t i = s0 + lane; i < s1; i += 64) kd[vd[i]] = ks[i];
            __syncthreads();
            // values: vd holds the positions AND is the destination -- stage the (position, value) pairs of the whole tile in
            // ks (free now: its keys have moved) before overwriting vd
            for (uint32_t i = s0 + lane; i < s1; i += 64) ks[i] = vd[i];
            __syncthreads();
            for (uint32_t i = s0 + lane; i < s1; i += 64) vd[ks[i]] = vs[i];
            __syncthreads();
            uint32_t* t = ks; ks = kd; kd = t;
            t = vs; vs = vd; vd = t;
        }
        if (vs != out) {  // an even number of passes ends in the input half
            for (uint32_t i = tid; i < n; i += THREADS) out[i] = vs[i];
        }
    }
}

// ranges[T] (from the stable tile sort of the index-order emission), vals_in -> vals_out: every tile's list in (depth, id) order
void sgr_launch_tile_sort(int T, const uint2* ranges, uint32_t* vals_in, uint32_t* vals_out, const uint32_t* dkeys,
                          const uint32_t* header, uint32_t* gk0, uint32_t* gk1, hipStream_t s) {
    if (T <= 0) return;
    sgr_tile_sort_wave_kernel<<<(T + 3) / 4, 256, 0, s>>>(T, ranges, vals_in, vals_out, dkeys, header);
    sgr_tile_sort_block_kernel<false><<<T, 64 * SGR_TS_MED_WAVES, 0, s>>>(ranges, vals_in, vals_out, dkeys, header, gk0, gk1);
    sgr_tile_sort_block_kernel<true><<<T, 64 * SGR_TS_BIG_WAVES, 0, s>>>(ranges, vals_in, vals_out, dkeys, header, gk0, gk1);
}

// ---- self-test of the ordering property the ranking relies on: within one ds_add_rtn_u32, lanes that hit the same address are
// served in ascending lane order.  pattern[i] = counter index of lane i % 64 in trial i / 64; out[i] = the value returned.
__global__ void __launch_bounds__(64) sgr_lds_atomic_order_kernel(const uint32_t* __restrict__ pattern, uint32_t* __restrict__ out,
                                                                   int trials) {
    __shared__ uint32_t c[64];
    const int lane = threadIdx.x;
    for (int t = blockIdx.x; t < trials; t += gridDim.x) {
        c[lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        out[t * 64 + lane] = atomicAdd(&c[pattern[t * 64 + lane] & 63u], 1u);
        __builtin_amdgcn_wave_barrier();
    }
}
void sgr_launch_lds_atomic_order_test(const uint32_t* pattern, uint32_t* out, int trials, hipStream_t s) {
    if (trials <= 0) return;
    sgr_lds_atomic_order_kernel<<<min(trials, 1024), 64, 0, s>>>(pattern, out, trials);
}
