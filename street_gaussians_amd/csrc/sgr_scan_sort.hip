// sgr_scan_sort.hip -- hand-written device-wide prefix scan (K4) and stable LSD radix sort of
// (u64 key, u32 value) pairs (K7) for gfx950, replacing cub::DeviceScan::InclusiveSum and
// cub::DeviceRadixSort::SortPairs (rasterizer_impl.cu:280,306-311) and, with 32-bit keys widened
// to 64, simple-knn's Morton sort (simple_knn.cu:210-213).
//
// Scan: reduce-then-scan over 2048-element blocks; wave64 shuffles inside a block.
// Sort: 8-bit digits.  The scatter ranks its 2048-key block with a wave64 ballot-match (8 ballots per key give
// the set of lanes sharing the digit; popcount of the lower lanes is the stable rank) and per-wave LDS digit
// counters -- no atomics on data, fully deterministic.  Where a block's keys of one digit start in the output is the
// digit's global base + the counts of all earlier blocks; two ways to get that:
//   * three launches per pass (default): an LDS-privatised per-block digit histogram, a scan of the [digit][block]
//     table in which workgroup d scans row d and emits the row total (the scatter turns the 256 totals into digit
//     bases itself), and the scatter -- 3 * npass launches of ~5 us or more each;
//   * one sweep (A/B, switch bit 5 / SGR_ONESWEEP=1): ONE histogram kernel per sort gives the global digit counts of
//     every pass, and each pass is a single kernel in which the blocks publish their digit counts and look back over
//     their predecessors' (decoupled look-back on a [block][digit] status table; block ids are handed out by an
//     atomic ticket so that every predecessor is already running) -- npass + 1 launches.  Measured SLOWER on MI355X
//     (tile sort of 7.8 M pairs 0.242 vs 0.138 ms, depth sort + scan of 1 M keys 0.161 vs 0.110 ms; 5 M Gaussians:
//     1.09 vs 0.69 ms): 8 workgroups x 256 CUs publish at the same moment and the look-back walks device-scope words
//     that live beyond the per-XCD L2s; the launches it removes are cheaper than that.  Kept for the A/B.
#include "sgr_common.h"

#include <atomic>
#include <cstdlib>

#ifndef SGR_WITH_VARIANTS
#define SGR_WITH_VARIANTS 0  // 1: also build the designs that were measured slower and kept as A/Bs (tools/build_variant.py)
#endif


// ------------------------------------------------------------------------------------------------
// scan kernels: ITEMS = 2048 per block = 256 threads x 8 consecutive elements
// gather != nullptr: element i of the scanned sequence is in[gather[i]] (the forward scans tiles_touched in depth
// order without materialising the permuted array)
// Second sequence (in2 != nullptr): the launch carries TWO scans of n elements each -- workgroups [0, nb) do the first,
// [nb, 2 nb) the second (no gather, exclusive, same stride) with their own block sums behind the first's.  The forward
// uses it to get the index-order offsets of the partial-gradient rows (SgrGeomView::u0) in the launches that scan
// tiles_touched in depth order anyway: these kernels are launch-latency bound, three more launches would cost 16 us.
__global__ void __launch_bounds__(256) sgr_scan_reduce_kernel(const uint32_t* __restrict__ in, size_t n,
                                                              uint32_t* __restrict__ block_sums,
                                                              const uint32_t* __restrict__ gather, int stride,
                                                              const uint32_t* __restrict__ in2, unsigned nb,
                                                              uint32_t* __restrict__ sub, int stride2) {
    __shared__ uint32_t lds4[4];
    unsigned b = blockIdx.x;
    if (b >= nb) { b -= nb; in = in2; gather = nullptr; block_sums += nb + 1; if (sub) sub += 8 * nb; stride = stride2; }
    const size_t base = (size_t)b * SGR_SCAN_ITEMS + (size_t)threadIdx.x * 8;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (base + i < n) s += gather ? in[(size_t)gather[base + i] * stride] : in[(base + i) * stride];
    uint32_t total;
    const uint32_t ex = sgr_block_excl_scan256(s, lds4, total);
    if (threadIdx.x == 0) block_sums[b] = total;
    // (sgr_launch_scan_head) where each 256-element sub-block -- 32 threads' elements -- starts inside this block
    if (sub != nullptr && (threadIdx.x & 31) == 0) sub[8 * b + (threadIdx.x >> 5)] = ex;
}

// one block per sequence: exclusive scan of block_sums[0..nb) in place; block_sums[nb] = grand total
__global__ void __launch_bounds__(256) sgr_scan_spine_kernel(uint32_t* __restrict__ block_sums, size_t nb,
                                                             uint32_t* __restrict__ total_out) {
    __shared__ uint32_t lds4[4];
    if (blockIdx.x) { block_sums += nb + 1; total_out = nullptr; }
    uint32_t carry = 0;
    for (size_t start = 0; start < nb; start += 256) {
        const size_t i = start + threadIdx.x;
        const uint32_t v = i < nb ? block_sums[i] : 0;
        uint32_t total;
        const uint32_t ex = sgr_block_excl_scan256(v, lds4, total);
        if (i < nb) block_sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) {
        block_sums[nb] = carry;
        if (total_out) *total_out = carry;
    }
}

__global__ void __launch_bounds__(256) sgr_scan_final_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                             size_t n, const uint32_t* __restrict__ block_sums,
                                                             const uint32_t* __restrict__ gather, int stride, bool inclusive,
                                                             const uint32_t* __restrict__ in2, uint32_t* __restrict__ out2,
                                                             unsigned nb) {
    __shared__ uint32_t lds4[4];
    unsigned b = blockIdx.x;
    if (b >= nb) { b -= nb; in = in2; out = out2; gather = nullptr; block_sums += nb + 1; inclusive = false; }
    const size_t base = (size_t)b * SGR_SCAN_ITEMS + (size_t)threadIdx.x * 8;
    uint32_t v[8];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        v[i] = (base + i < n) ? (gather ? in[(size_t)gather[base + i] * stride] : in[(base + i) * stride]) : 0;
        s += v[i];
    }
    uint32_t total;
    uint32_t run = sgr_block_excl_scan256(s, lds4, total) + block_sums[b];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (inclusive) run += v[i];
        if (base + i < n) out[base + i] = run;
        if (!inclusive) run += v[i];
    }
}

// out may alias in.  tmp needs sgr_scan_tmp_count(n) words (twice that with a second sequence); tmp[nblocks] receives
// the grand total, and so does *total_out when given.  in2 / out2: a second, exclusive scan of in2[i * in_stride] in the
// same three launches.
void sgr_launch_scan(const uint32_t* in, uint32_t* out, size_t n, uint32_t* tmp, bool inclusive, hipStream_t s,
                     uint32_t* total_out, const uint32_t* gather, int in_stride, const uint32_t* in2, uint32_t* out2) {
    if (n == 0) return;
    const size_t nb = (n + SGR_SCAN_ITEMS - 1) / SGR_SCAN_ITEMS;
    const unsigned seqs = in2 ? 2u : 1u;

    sgr_scan_reduce_kernel<<<(unsigned)nb * seqs, 256, 0, s>>>(in, n, tmp, gather, in_stride, in2, (unsigned)nb, nullptr, in_stride);
    sgr_scan_spine_kernel<<<seqs, 256, 0, s>>>(tmp, nb, total_out);
    sgr_scan_final_kernel<<<(unsigned)nb * seqs, 256, 0, s>>>(in, out, n, tmp, gather, in_stride, inclusive, in2, out2,
                                                             (unsigned)nb);
}

void sgr_launch_scan_head(const uint32_t* in, const uint32_t* in2, size_t n, int in_stride, uint32_t* tmp, uint32_t* sub,
                          hipStream_t s, int in2_stride) {
    if (in2_stride <= 0) in2_stride = in_stride;
    if (n == 0) return;
    const size_t nb = (n + SGR_SCAN_ITEMS - 1) / SGR_SCAN_ITEMS;
    sgr_scan_reduce_kernel<<<(unsigned)nb * 2u, 256, 0, s>>>(in, n, tmp, nullptr, in_stride, in2, (unsigned)nb, sub, in2_stride);
    sgr_scan_spine_kernel<<<2, 256, 0, s>>>(tmp, nb, nullptr);
}

// ------------------------------------------------------------------------------------------------
// radix sort
// block b owns keys [b*ITEMS, (b+1)*ITEMS); wave w of the block owns ITEMS/4 consecutive keys, read in IPT
// steps of 64 (lane l <-> key base + w*512 + step*64 + l), so memory order == (wave, step, lane).
template <typename K, int IPT>
__global__ void __launch_bounds__(256)
sgr_sort_hist_kernel(const K* __restrict__ keys, uint32_t n, int shift, uint32_t mask, uint32_t nblocks,
                     uint32_t* __restrict__ hist) {
    // one private histogram per wave (a quarter of the same-address LDS atomics), merged at the end
    __shared__ uint32_t h[4][512];  // up to 9-bit digits
#pragma unroll
    for (int w = 0; w < 4; w++) { h[w][threadIdx.x] = 0; h[w][256 + threadIdx.x] = 0; }
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const uint32_t base = blockIdx.x * (256u * IPT);
    // (a histogram does not care which thread counts which key: 16 bytes of consecutive keys per load -- a wave instruction
    // then moves 1 KB instead of the 128 / 256 bytes of 64 two- / four-byte loads; the key buffers are 256-byte aligned)
    constexpr int V = 16 / (int)sizeof(K);  // keys per 16-byte load
    static_assert(IPT % V == 0, "keys per thread must be a multiple of the vector width");
    struct alignas(16) Vec { K k[V]; };
#pragma unroll
    for (int s = 0; s < IPT / V; s++) {
        const uint32_t i = base + (s * 256 + threadIdx.x) * V;
        if (i + V <= n) {
            const Vec v = *reinterpret_cast<const Vec*>(keys + i);
#pragma unroll
            for (int e = 0; e < V; e++) atomicAdd(&h[wave][(uint32_t)(v.k[e] >> shift) & mask], 1u);
        } else {
#pragma unroll
            for (int e = 0; e < V; e++)
                if (i + e < n) atomicAdd(&h[wave][(uint32_t)(keys[i + e] >> shift) & mask], 1u);
        }
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d <= mask; d += 256)
        hist[(size_t)d * nblocks + blockIdx.x] = h[0][d] + h[1][d] + h[2][d] + h[3][d];  // [digit][block]
}

// workgroup d: exclusive scan of row d of the [digit][block] table in place; totals[d] = row sum
__global__ void __launch_bounds__(256)
sgr_sort_rowscan_kernel(uint32_t* __restrict__ hist, uint32_t nblocks, uint32_t* __restrict__ totals) {
    __shared__ uint32_t lds4[4];
    uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
    uint32_t carry = 0;
    for (uint32_t start = 0; start < nblocks; start += 256 * 8) {
        const uint32_t base = start + threadIdx.x * 8;
        uint32_t v[8], s = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            v[i] = (base + i < nblocks) ? row[base + i] : 0u;
            s += v[i];
        }
        uint32_t total;
        uint32_t run = carry + sgr_block_excl_scan256(s, lds4, total);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (base + i < nblocks) row[base + i] = run;
            run += v[i];
        }
        carry += total;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// Scatter with a block-local shuffle: the 2048 keys of the block are first placed in LDS in their block-local sorted
// order (digit-major, stable), then written out by consecutive threads -- every digit's keys of this block form one
// contiguous run in global memory (8 keys on average at 256 digits, 32 at the 64 digits of the tile sort's second
// pass), instead of 64 scattered 4-byte stores per wave instruction.
// ONE (one-sweep form): `totals` = the pass's global digit counts, `status` = the [block][digit] look-back table,
// `ticket` = the pass's block-id counter, `err` = set if a look-back gave up (never observed; it bounds the spin).
// Status word: bits 0-47 count, bits 48-55 state (1 = this block's count, 2 = inclusive prefix up to this block),
// bits 56-63 = pass + 1 (words of earlier passes read as empty, so the table is zeroed once per sort).
#define SGR_OS_AGG (1ull << 48)
#define SGR_OS_PREFIX (2ull << 48)
#define SGR_OS_VALUE ((1ull << 48) - 1ull)
// BITS = digit width (NB = 2^BITS bins), IPT = keys per thread (block = 256 * IPT keys).  Fewer bins and larger blocks
// make every digit's run of a block longer (256 * IPT / NB keys on average: 8 at 8 bits x 2048 keys = a 32-byte store
// run, 32 at 7 bits x 4096 keys = 128 bytes), which is what the scatter's write efficiency depends on.
// vin == nullptr: the value of key i is i itself (first pass of a sort whose values are the element ids).
// aux_in != nullptr (last pass): aux_out[pos] = aux_in[value] as well -- a fused gather of an 8-byte per-id record into
// sorted order (the depth sort hands tiles_touched + tile rect of every Gaussian to the scan / duplicate stage this way).
template <typename K, int BITS, int IPT, bool ONE>
__global__ void __launch_bounds__(256)
sgr_sort_scatter_kernel(const K* __restrict__ kin, const uint32_t* __restrict__ vin, K* __restrict__ kout,
                        uint32_t* __restrict__ vout, uint32_t n, int shift, uint32_t nblocks,
                        const uint32_t* __restrict__ hist_scanned, const uint32_t* __restrict__ totals,
                        unsigned long long* __restrict__ status, uint32_t* __restrict__ ticket, uint32_t* __restrict__ err,
                        uint32_t pass_tag, const uint2* __restrict__ aux_in, uint2* __restrict__ aux_out, int aux16) {
    constexpr int NB = 1 << BITS;
    constexpr int BPT = NB > 256 ? NB / 256 : 1;  // bins per thread in the prefix section (9-bit digits: 2)
    static_assert(!(ONE && NB > 256), "the one-sweep form has one status word per thread");
    constexpr uint32_t ITEMS = 256u * IPT;
    __shared__ uint32_t cnt[4][NB];    // per wave: count of each digit, then its block-local start for that wave
    __shared__ uint32_t lstart[NB];    // block-local start of each digit
    __shared__ uint32_t gbase[NB];     // global position of the digit's first key of this block
    __shared__ uint32_t lds4[4];
    __shared__ K sK[ITEMS];
    __shared__ uint32_t sV[ITEMS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (ONE && tid == 0) lds4[0] = atomicAdd(ticket, 1u);  // block id in start order: every lower id is already running
    for (int b = tid; b < NB; b += 256) {
#pragma unroll
        for (int w = 0; w < 4; w++) cnt[w][b] = 0;
    }
    __syncthreads();
    const uint32_t block = ONE ? lds4[0] : blockIdx.x;

    // Wave 0 turns the digit counts into positions after the ranking (below).  What it needs from global memory for that --
    // the digit totals and this block's row of the scanned [digit][block] table -- does not depend on the ranking: requested
    // here, so that the two round trips overlap the key loads and the ballots (lane l owns bins [l * BPL, (l + 1) * BPL)).
    constexpr int BPL = NB > 64 ? NB / 64 : 1;
    uint32_t pre_before[BPL], pre_total[BPL];
    if (!ONE && wave == 0) {
#pragma unroll
        for (int j = 0; j < BPL; j++) {
            const int bin = lane * BPL + j;
            pre_before[j] = bin < NB ? hist_scanned[(size_t)bin * nblocks + block] : 0u;
            pre_total[j] = bin < NB ? totals[bin] : 0u;
        }
    }

    const uint32_t base = block * ITEMS + wave * (64 * IPT);
    K key[IPT];
    uint32_t val[IPT], rnk[IPT];
#pragma unroll
    for (int s = 0; s < IPT; s++) {
        const uint32_t i = base + s * 64 + lane;
        const bool valid = i < n;
        key[s] = valid ? kin[i] : (K)0;
        val[s] = valid ? (vin ? vin[i] : i) : 0u;
    }
#pragma unroll
    for (int s = 0; s < IPT; s++) {
        const bool valid = base + s * 64 + lane < n;
        const uint32_t d = (uint32_t)(key[s] >> shift) & (uint32_t)(NB - 1);
        // the lanes that share this lane's digit: per bit, keep the lanes whose bit equals mine -- m &= ballot XNOR (my bit
        // spread over the word).  On 32-bit halves with v_bfe_i32 / v_xor / v_and: a `bit ? bal : ~bal` select compiles to
        // v_cndmask on VCC, 23 cycles per wave instruction back to back on gfx950 (profiles/r5/valu_rates2.jsonl)
        const uint64_t vm = __builtin_amdgcn_ballot_w64(valid);
        uint32_t mlo = (uint32_t)vm, mhi = (uint32_t)(vm >> 32);
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            const int y = __builtin_amdgcn_sbfe((int)d, b, 1);  // 0 or -1
            const uint64_t bal = __builtin_amdgcn_ballot_w64(y != 0);
            mlo &= ~((uint32_t)bal ^ (uint32_t)y);
            mhi &= ~((uint32_t)(bal >> 32) ^ (uint32_t)y);
        }
        const uint32_t prefix = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));  // set bits of m below this lane
        // every lane of the group reads the wave's counter of the digit, its lowest lane moves it on by the group's size
        const uint32_t prev = valid ? cnt[wave][d] : 0u;
        __builtin_amdgcn_wave_barrier();
        if (valid && prefix == 0) cnt[wave][d] = prev + (uint32_t)(__builtin_popcount(mlo) + __builtin_popcount(mhi));
        rnk[s] = prev + prefix;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if constexpr (!ONE) {
        // counts -> positions, by wave 0 alone (two wave-level scans; the block-wide form cost four more barriers): for its
        // BPL consecutive bins a lane sums the four waves' counts, the two exclusive scans run over the lanes' sums, and the
        // bins of a lane are prefixed in order.  gbase = where this block's keys of the digit start in the output
        // (keys of lower digits anywhere + keys of the digit in earlier blocks), lstart / cnt[w] = where they start in LDS.
        if (wave == 0) {
            uint32_t c[BPL][4], tot[BPL];
            uint32_t sum_t = 0, sum_g = 0;
#pragma unroll
            for (int j = 0; j < BPL; j++) {
                const int bin = lane * BPL + j;
#pragma unroll
                for (int w = 0; w < 4; w++) c[j][w] = bin < NB ? cnt[w][bin] : 0u;
                tot[j] = c[j][0] + c[j][1] + c[j][2] + c[j][3];
                sum_t += tot[j];
                sum_g += pre_total[j];
            }
            uint32_t g = sgr_wave_incl_scan(sum_g, lane) - sum_g;
            uint32_t ls = sgr_wave_incl_scan(sum_t, lane) - sum_t;
#pragma unroll
            for (int j = 0; j < BPL; j++) {
                const int bin = lane * BPL + j;
                if (bin < NB) {
                    gbase[bin] = pre_before[j] + g;
                    lstart[bin] = ls;
                    cnt[0][bin] = ls;
                    cnt[1][bin] = ls + c[j][0];
                    cnt[2][bin] = ls + c[j][0] + c[j][1];
                    cnt[3][bin] = ls + c[j][0] + c[j][1] + c[j][2];
                }
                g += pre_total[j];
                ls += tot[j];
            }
        }
    } else
    if constexpr (BPT == 1) {
        const bool bin = tid < NB;
        const uint32_t c0 = bin ? cnt[0][tid] : 0u, c1 = bin ? cnt[1][tid] : 0u, c2 = bin ? cnt[2][tid] : 0u,
                       c3 = bin ? cnt[3][tid] : 0u;
        uint32_t all;
        // digit base over the whole array (contains a barrier), then the block-local digit starts (another scan)
        uint32_t before = 0;  // keys of digit `tid` in the blocks before this one
        if (ONE) {
            const unsigned long long tag = (unsigned long long)pass_tag << 56;
            const uint32_t c = c0 + c1 + c2 + c3;
            unsigned long long* mine = status + (size_t)block * 256 + tid;
            __hip_atomic_store(mine, tag | (block == 0 ? SGR_OS_PREFIX : SGR_OS_AGG) | c, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long sum = 0;
            uint32_t spins = 0;
            for (uint32_t pb = block; pb > 0;) {
                const unsigned long long w = __hip_atomic_load(status + (size_t)(pb - 1) * 256 + tid, __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_AGENT);
                if ((w >> 56) != pass_tag) {  // not published yet
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) { atomicOr(err, 1u); break; }
                    continue;
                }
                sum += w & SGR_OS_VALUE;
                if (w & SGR_OS_PREFIX) break;
                pb--;
            }
            if (block > 0)
                __hip_atomic_store(mine, tag | SGR_OS_PREFIX | (sum + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            before = (uint32_t)sum;
        } else if (bin) {
            before = hist_scanned[(size_t)tid * nblocks + block];
        }
        const uint32_t g = before + sgr_block_excl_scan256(bin ? totals[tid] : 0u, lds4, all);
        const uint32_t ls = sgr_block_excl_scan256(c0 + c1 + c2 + c3, lds4, all);
        if (bin) {
            gbase[tid] = g;
            lstart[tid] = ls;
            cnt[0][tid] = ls;
            cnt[1][tid] = ls + c0;
            cnt[2][tid] = ls + c0 + c1;
            cnt[3][tid] = ls + c0 + c1 + c2;
        }
    } else {
        // more bins than threads (9-bit digits): every thread owns BPT CONSECUTIVE bins, the two block scans run over the
        // threads' sums and the bins of a thread are prefixed in order
        uint32_t c[BPT][4], tot[BPT], gt[BPT], before[BPT];
        uint32_t sum_t = 0, sum_g = 0, all;
#pragma unroll
        for (int j = 0; j < BPT; j++) {
            const int bin = tid * BPT + j;
#pragma unroll
            for (int w = 0; w < 4; w++) c[j][w] = cnt[w][bin];
            tot[j] = c[j][0] + c[j][1] + c[j][2] + c[j][3];
            gt[j] = totals[bin];
            before[j] = hist_scanned[(size_t)bin * nblocks + block];
            sum_t += tot[j];
            sum_g += gt[j];
        }
        uint32_t g = sgr_block_excl_scan256(sum_g, lds4, all);
        uint32_t ls = sgr_block_excl_scan256(sum_t, lds4, all);
#pragma unroll
        for (int j = 0; j < BPT; j++) {
            const int bin = tid * BPT + j;
            gbase[bin] = before[j] + g;
            lstart[bin] = ls;
            cnt[0][bin] = ls;
            cnt[1][bin] = ls + c[j][0];
            cnt[2][bin] = ls + c[j][0] + c[j][1];
            cnt[3][bin] = ls + c[j][0] + c[j][1] + c[j][2];
            g += gt[j];
            ls += tot[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < IPT; s++) {
        const uint32_t i = base + s * 64 + lane;
        if (i < n) {
            const uint32_t d = (uint32_t)(key[s] >> shift) & (uint32_t)(NB - 1);
            const uint32_t lp = cnt[wave][d] + rnk[s];
            sK[lp] = key[s];
            sV[lp] = val[s];
        }
    }
    __syncthreads();
    const uint32_t first = block * ITEMS;
    const uint32_t nloc = min(ITEMS, n - first);
#pragma unroll
    for (int s = 0; s < IPT; s++) {
        const uint32_t li = s * 256 + tid;
        if (li < nloc) {
            const K k = sK[li];
            const uint32_t d = (uint32_t)(k >> shift) & (uint32_t)(NB - 1);
            const uint32_t pos = gbase[d] + (li - lstart[d]);
            const uint32_t v = sV[li];
            kout[pos] = k;
            vout[pos] = v;
            if (aux_in != nullptr) {  // aux16: the per-id record is 16 bytes (marked-list mode of the forward: SgrGeomView::aux_ref)
                if (aux16) reinterpret_cast<uint4*>(aux_out)[pos] = reinterpret_cast<const uint4*>(aux_in)[v];
                else aux_out[pos] = aux_in[v];
            }
        }
    }
}

#if SGR_WITH_VARIANTS  // the one-sweep A/B form is built only by tools/build_variant.py (-DSGR_WITH_VARIANTS=1)
// One-sweep prologue: the global digit counts of every pass in one read of the keys (LDS-privatised, one atomic per
// non-empty (pass, digit) per workgroup), and the look-back table zeroed.  Workgroups stride over 4096-key chunks.
template <typename K>
__global__ void __launch_bounds__(256)
sgr_sort_hist_all_kernel(const K* __restrict__ keys, uint32_t n, int npass, uint32_t* __restrict__ ghist,
                         unsigned long long* __restrict__ status, size_t status_words) {
    __shared__ uint32_t h[SGR_SORT_MAX_PASS][256];
    for (int p = 0; p < npass; p++) h[p][threadIdx.x] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < status_words; i += (size_t)gridDim.x * 256) status[i] = 0ull;
    for (uint32_t chunk = blockIdx.x; (size_t)chunk * 4096 < n; chunk += gridDim.x) {
#pragma unroll 4
        for (int s = 0; s < 16; s++) {
            const uint32_t i = chunk * 4096u + s * 256 + threadIdx.x;
            if (i < n) {
                const K k = keys[i];
                for (int p = 0; p < npass; p++) atomicAdd(&h[p][(uint32_t)(k >> (8 * p)) & 255u], 1u);
            }
        }
    }
    __syncthreads();
    for (int p = 0; p < npass; p++) {
        const uint32_t c = h[p][threadIdx.x];
        if (c) atomicAdd(&ghist[p * 256 + threadIdx.x], c);
    }
}

// 0 = three launches per pass (default), 1 = one sweep; set through sgr_test_switches bit 5 / SGR_ONESWEEP
#endif  // SGR_WITH_VARIANTS

static std::atomic<int> g_sort_one_sweep{-1};
void sgr_sort_set_one_sweep(int on) { g_sort_one_sweep.store(on ? 1 : 0, std::memory_order_relaxed); }
int sgr_sort_get_one_sweep() {
    int v = g_sort_one_sweep.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("SGR_ONESWEEP");
        v = (e && *e && *e != '0') ? 1 : 0;
        g_sort_one_sweep.store(v, std::memory_order_relaxed);
    }
    return v;
}

// Digit schedule of a sort on key bits [0, end_bit) with digits of at most `max_bits` bits: npass = ceil(end_bit / max_bits)
// passes of ceil(end_bit / npass) bits each (14 tile bits -> 2 x 7 instead of 8 + 6: half the bins, twice the run length).
// max_bits is 8 everywhere (so that the number of buffer flips of the tile sort is the same in the three-launch and in the
// one-sweep form: sgr_sort_pass_count) except in the forward's depth sort, which may ask for 9 (512 bins, two per thread in
// the scatter's prefix section): its 27 key bits then take three passes.  Measured on MI355X (tools/gpu_r4_g.sh, depth sort +
// scan): 3 x 9 bits 0.079 / 0.109 / 0.402 ms at 500 k / 1 M / 5 M Gaussians, 4 x 7 bits 0.085 / 0.109 / 0.348 ms -- nine-bit
// digits leave 4-key store runs, which only pays while the launches are latency-bound: chosen by n (sgr_api.hip).
static inline int sort_passes(int end_bit, int max_bits) {
    static const int forced = [] { const char* e = getenv("SGR_SORT_BITS"); return e ? atoi(e) : 0; }();  // A/B only
    const int maxb = forced == 8 ? 8 : max_bits;
    return (end_bit + maxb - 1) / maxb;
}
static inline int sort_pass_bits(int end_bit, int max_bits) {
    const int npass = sort_passes(end_bit, max_bits);
    return npass ? (end_bit + npass - 1) / npass : 8;
}
// Keys per thread = block size (256 * IPT keys): a larger block gives every digit a longer store run (32 instead of 16 pairs at
// 7-bit digits) at the price of occupancy (registers, LDS).  Measured on MI355X, interleaved (profiles/r6/ab_sort_ipt_*.jsonl):
//   16-bit keys (the forward's tile sort; 2 KB less LDS and shorter key runs than 32-bit keys): 4096-key blocks win at every
//     size -- 0.103 -> 0.090 ms at 7.8 M pairs, 0.500 -> 0.403 ms at 39 M;
//   32-bit keys (the depth pre-sort of the Gaussians): 2048-key blocks at 1 M (0.096 vs 0.108 ms, sort + scan), 4096-key blocks
//     at 5 M (0.371 vs 0.353 ms); rounds 4 / 5 had measured the 32-bit tile sort the same way (40.7 / 221.6 vs 42.3 / 242.9 us
//     per pass at 7.8 / 38.9 M pairs) and kept 2048.
static inline int sort_ipt(uint32_t n, size_t key_bytes) {
    static const int forced = [] { const char* e = getenv("SGR_SORT_IPT"); return e ? atoi(e) : 0; }();  // A/B only
    if (forced == 8 || forced == 16) return forced;
    if (key_bytes == 2) return n >= 500000u ? 16 : 8;
    return n >= 3000000u ? 16 : 8;
}

int sgr_sort_pass_count(int end_bit) { return (end_bit + 7) / 8; }  // buffer flips of a sort with the default 8-bit cap

template <typename K, int BITS, int IPT>
static void sort_pass(const K* kin, const uint32_t* vin, K* kout, uint32_t* vout, uint32_t n, int shift, uint32_t* hist,
                      const uint2* aux_in, uint2* aux_out, int aux16, hipStream_t s) {
    const uint32_t nblocks = (n + 256u * IPT - 1) / (256u * IPT);
    constexpr int NB = 1 << BITS;
    sgr_sort_hist_kernel<K, IPT><<<nblocks, 256, 0, s>>>(kin, n, shift, (uint32_t)(NB - 1), nblocks, hist);
    uint32_t* totals = hist + (size_t)NB * nblocks;
    sgr_sort_rowscan_kernel<<<NB, 256, 0, s>>>(hist, nblocks, totals);
    sgr_sort_scatter_kernel<K, BITS, IPT, false><<<nblocks, 256, 0, s>>>(kin, vin, kout, vout, n, shift, nblocks, hist, totals,
                                                                         nullptr, nullptr, nullptr, 0u, aux_in, aux_out, aux16);
}

// Sorts n pairs on key bits [0, end_bit).  keys[0]/vals[0] hold the input; returns the index (0/1)
// of the pair of buffers that holds the sorted output.  hist: sgr_sort_hist_words(n) dwords (scan_tmp is unused):
// three launches -- the [digit][block] table + the digit totals, recomputed before every scatter pass (the per-block digit
// histogram depends on where the previous pass left the keys);
// one sweep -- [control: 8 x 256 digit counts, 8 tickets, error flag | status table, 256 x 64 bit per block].
// iota: vals[0] is not read, the value of input element i is i.  aux_in / aux_out: see the scatter kernel (last pass).
template <typename K>
static int sort_pairs_impl(K* const keys[2], uint32_t* const vals[2], uint32_t n, int end_bit, uint32_t* hist,
                           uint32_t* scan_tmp, hipStream_t s, bool iota = false, const uint2* aux_in = nullptr,
                           uint2* aux_out = nullptr, int max_bits = 8, int aux16 = 0) {
    if (n == 0) return 0;
    int npass = (end_bit + 7) / 8;
    int cur = 0;
#if SGR_WITH_VARIANTS
    if (sgr_sort_get_one_sweep() && npass <= SGR_SORT_MAX_PASS && !iota && !aux_in) {
        const uint32_t nblocks = (n + SGR_SORT_ITEMS - 1) / SGR_SORT_ITEMS;
        uint32_t* ghist = hist;
        uint32_t* tickets = hist + SGR_SORT_MAX_PASS * 256;
        uint32_t* err = tickets + SGR_SORT_MAX_PASS;
        unsigned long long* status = reinterpret_cast<unsigned long long*>(hist + SGR_SORT_CTRL_WORDS);
        (void)hipMemsetAsync(hist, 0, SGR_SORT_CTRL_WORDS * sizeof(uint32_t), s);
        const uint32_t chunks = (n + 4095u) / 4096u;
        sgr_sort_hist_all_kernel<K><<<chunks < 1024u ? chunks : 1024u, 256, 0, s>>>(keys[0], n, npass, ghist, status,
                                                                                   (size_t)nblocks * 256);
        for (int p = 0; p < npass; p++) {
            sgr_sort_scatter_kernel<K, 8, SGR_SORT_IPT, true><<<nblocks, 256, 0, s>>>(
                keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, 8 * p, nblocks, nullptr, ghist + p * 256, status,
                tickets + p, err, (uint32_t)p + 1u, nullptr, nullptr, 0);
            cur ^= 1;
        }
        return cur;
    }
#endif
    const int bits = sizeof(K) == 8 ? 8 : sort_pass_bits(end_bit, max_bits);
    const int ipt = sizeof(K) == 8 ? 8 : sort_ipt(n, sizeof(K));
    if (sizeof(K) <= 4) npass = sort_passes(end_bit, max_bits);
    for (int p = 0; p < npass; p++) {
        const uint32_t* vin = (iota && p == 0) ? nullptr : vals[cur];
        const bool last = p == npass - 1;
        const uint2* ai = last ? aux_in : nullptr;
        uint2* ao = last ? aux_out : nullptr;
        const int shift = bits * p;
#define SGR_PASS(B, I) sort_pass<K, B, I>(keys[cur], vin, keys[cur ^ 1], vals[cur ^ 1], n, shift, hist, ai, ao, aux16, s)
        if constexpr (sizeof(K) == 8) { SGR_PASS(8, 8); }
        else if (ipt == 16 && bits <= 8) {
            if (bits <= 5) SGR_PASS(5, 16); else if (bits == 6) SGR_PASS(6, 16); else if (bits == 7) SGR_PASS(7, 16); else SGR_PASS(8, 16);
        } else {
            if (bits <= 5) SGR_PASS(5, 8); else if (bits == 6) SGR_PASS(6, 8); else if (bits == 7) SGR_PASS(7, 8);
            else if (bits == 8) SGR_PASS(8, 8); else SGR_PASS(9, 8);
        }
#undef SGR_PASS
        cur ^= 1;
    }
    return cur;
}
int sgr_launch_sort_pairs(uint64_t* const keys[2], uint32_t* const vals[2], uint32_t n, int end_bit, uint32_t* hist,
                          uint32_t* scan_tmp, hipStream_t s) {
    return sort_pairs_impl<uint64_t>(keys, vals, n, end_bit, hist, scan_tmp, s);
}
int sgr_launch_sort_pairs32(uint32_t* const keys[2], uint32_t* const vals[2], uint32_t n, int end_bit, uint32_t* hist,
                            uint32_t* scan_tmp, hipStream_t s, bool iota, const uint2* aux_in, uint2* aux_out, int max_bits, int aux16) {
    return sort_pairs_impl<uint32_t>(keys, vals, n, end_bit, hist, scan_tmp, s, iota, aux_in, aux_out, max_bits, aux16);
}
// 16-bit keys (the forward's tile sort when the frame has fewer than 65535 tiles): the same kernels on half the key bytes
int sgr_launch_sort_pairs16(uint16_t* const keys[2], uint32_t* const vals[2], uint32_t n, int end_bit, uint32_t* hist,
                            uint32_t* scan_tmp, hipStream_t s) {
    return sort_pairs_impl<uint16_t>(keys, vals, n, end_bit > 16 ? 16 : end_bit, hist, scan_tmp, s);
}
