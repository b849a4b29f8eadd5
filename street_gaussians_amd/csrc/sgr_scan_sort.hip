// sgr_scan_sort.hip -- hand-written device-wide prefix scan (K4) and stable LSD radix sort of
// (u64 key, u32 value) pairs (K7) for gfx950, replacing cub::DeviceScan::InclusiveSum and
// cub::DeviceRadixSort::SortPairs (rasterizer_impl.cu:280,306-311) and, with 32-bit keys widened
// to 64, simple-knn's Morton sort (simple_knn.cu:210-213).
//
// Scan: reduce-then-scan over 2048-element blocks; wave64 shuffles inside a block.
// Sort: 8-bit digits.  Per pass, three launches: an LDS-privatised per-block digit histogram; a scan of
// the [digit][block] table in which workgroup d scans row d and emits the row total (the scatter turns the
// 256 totals into digit bases itself -- a generic device-wide scan would cost three launches here, and at
// these sizes every launch is ~5 us of GPU time); and a scatter that ranks its 2048-key block with a wave64
// ballot-match (8 ballots per key give the set of lanes sharing the digit; popcount of the lower
// lanes is the stable rank) and per-wave LDS digit counters -- no atomics, fully deterministic.
#include "sgr_common.h"


// ------------------------------------------------------------------------------------------------
// wave / block primitives
__device__ __forceinline__ uint32_t sgr_wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan of one value per thread over a 256-thread block; returns block total in `total`
__device__ __forceinline__ uint32_t sgr_block_excl_scan256(uint32_t v, uint32_t* lds4, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = sgr_wave_incl_scan(v, lane);
    if (lane == 63) lds4[wave] = inc;
    __syncthreads();
    const uint32_t w0 = lds4[0], w1 = lds4[1], w2 = lds4[2], w3 = lds4[3];
    __syncthreads();
    const uint32_t base = (wave > 0 ? w0 : 0) + (wave > 1 ? w1 : 0) + (wave > 2 ? w2 : 0);
    total = w0 + w1 + w2 + w3;
    return base + inc - v;
}

// ------------------------------------------------------------------------------------------------
// scan kernels: ITEMS = 2048 per block = 256 threads x 8 consecutive elements
// gather != nullptr: element i of the scanned sequence is in[gather[i]] (the forward scans tiles_touched in depth
// order without materialising the permuted array)
__global__ void __launch_bounds__(256) sgr_scan_reduce_kernel(const uint32_t* __restrict__ in, size_t n,
                                                              uint32_t* __restrict__ block_sums,
                                                              const uint32_t* __restrict__ gather) {
    __shared__ uint32_t lds4[4];
    const size_t base = (size_t)blockIdx.x * SGR_SCAN_ITEMS + (size_t)threadIdx.x * 8;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (base + i < n) s += gather ? in[gather[base + i]] : in[base + i];
    uint32_t total;
    sgr_block_excl_scan256(s, lds4, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// single block: exclusive scan of block_sums[0..nb) in place; block_sums[nb] = grand total
__global__ void __launch_bounds__(256) sgr_scan_spine_kernel(uint32_t* __restrict__ block_sums, size_t nb,
                                                             uint32_t* __restrict__ total_out) {
    __shared__ uint32_t lds4[4];
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

template <bool INCLUSIVE>
__global__ void __launch_bounds__(256) sgr_scan_final_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                             size_t n, const uint32_t* __restrict__ block_sums,
                                                             const uint32_t* __restrict__ gather) {
    __shared__ uint32_t lds4[4];
    const size_t base = (size_t)blockIdx.x * SGR_SCAN_ITEMS + (size_t)threadIdx.x * 8;
    uint32_t v[8];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        v[i] = (base + i < n) ? (gather ? in[gather[base + i]] : in[base + i]) : 0;
        s += v[i];
    }
    uint32_t total;
    uint32_t run = sgr_block_excl_scan256(s, lds4, total) + block_sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (INCLUSIVE) run += v[i];
        if (base + i < n) out[base + i] = run;
        if (!INCLUSIVE) run += v[i];
    }
}

// out may alias in.  tmp needs sgr_scan_tmp_count(n) words; tmp[nblocks] receives the grand total, and so does
// *total_out when given.
void sgr_launch_scan(const uint32_t* in, uint32_t* out, size_t n, uint32_t* tmp, bool inclusive, hipStream_t s,
                     uint32_t* total_out, const uint32_t* gather) {
    if (n == 0) return;
    const size_t nb = (n + SGR_SCAN_ITEMS - 1) / SGR_SCAN_ITEMS;

    sgr_scan_reduce_kernel<<<(unsigned)nb, 256, 0, s>>>(in, n, tmp, gather);
    sgr_scan_spine_kernel<<<1, 256, 0, s>>>(tmp, nb, total_out);
    if (inclusive) sgr_scan_final_kernel<true><<<(unsigned)nb, 256, 0, s>>>(in, out, n, tmp, gather);
    else sgr_scan_final_kernel<false><<<(unsigned)nb, 256, 0, s>>>(in, out, n, tmp, gather);
}

// ------------------------------------------------------------------------------------------------
// radix sort
// block b owns keys [b*ITEMS, (b+1)*ITEMS); wave w of the block owns ITEMS/4 consecutive keys, read in IPT
// steps of 64 (lane l <-> key base + w*512 + step*64 + l), so memory order == (wave, step, lane).
template <typename K>
__global__ void __launch_bounds__(256)
sgr_sort_hist_kernel(const K* __restrict__ keys, uint32_t n, int shift, uint32_t nblocks,
                     uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * SGR_SORT_ITEMS;
#pragma unroll
    for (int s = 0; s < SGR_SORT_IPT; s++) {
        const uint32_t i = base + s * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];  // [digit][block]
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
template <typename K>
__global__ void __launch_bounds__(256)
sgr_sort_scatter_kernel(const K* __restrict__ kin, const uint32_t* __restrict__ vin, K* __restrict__ kout,
                        uint32_t* __restrict__ vout, uint32_t n, int shift, uint32_t nblocks,
                        const uint32_t* __restrict__ hist_scanned, const uint32_t* __restrict__ totals) {
    __shared__ uint32_t cnt[4][256];    // per wave: count of each digit, then its block-local start for that wave
    __shared__ uint32_t lstart[256];    // block-local start of each digit
    __shared__ uint32_t gbase[256];     // global position of the digit's first key of this block
    __shared__ uint32_t lds4[4];
    __shared__ K sK[SGR_SORT_ITEMS];
    __shared__ uint32_t sV[SGR_SORT_ITEMS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int w = 0; w < 4; w++) cnt[w][tid] = 0;
    __syncthreads();

    const uint32_t base = blockIdx.x * SGR_SORT_ITEMS + wave * (64 * SGR_SORT_IPT);
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    K key[SGR_SORT_IPT];
    uint32_t val[SGR_SORT_IPT], rnk[SGR_SORT_IPT];
#pragma unroll
    for (int s = 0; s < SGR_SORT_IPT; s++) {
        const uint32_t i = base + s * 64 + lane;
        const bool valid = i < n;
        key[s] = valid ? kin[i] : (K)0;
        val[s] = valid ? vin[i] : 0u;
        const uint32_t d = (uint32_t)(key[s] >> shift) & 255u;
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(valid && bit);
            m &= bit ? bal : ~bal;
        }
        const uint32_t prefix = __popcll(m & lt_mask);
        const uint32_t count = __popcll(m);
        uint32_t prev = 0;
        if (valid && prefix == 0) {  // lowest lane of each digit group: bump this wave's counter
            prev = cnt[wave][d];
            cnt[wave][d] = prev + count;
        }
        const int leader = m ? (__ffsll((unsigned long long)m) - 1) : lane;
        prev = __shfl(prev, leader, 64);
        rnk[s] = prev + prefix;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {
        const uint32_t c0 = cnt[0][tid], c1 = cnt[1][tid], c2 = cnt[2][tid], c3 = cnt[3][tid];
        uint32_t all;
        // digit base over the whole array (contains a barrier), then the block-local digit starts (another scan)
        const uint32_t g = hist_scanned[(size_t)tid * nblocks + blockIdx.x] + sgr_block_excl_scan256(totals[tid], lds4, all);
        const uint32_t ls = sgr_block_excl_scan256(c0 + c1 + c2 + c3, lds4, all);
        gbase[tid] = g;
        lstart[tid] = ls;
        cnt[0][tid] = ls;
        cnt[1][tid] = ls + c0;
        cnt[2][tid] = ls + c0 + c1;
        cnt[3][tid] = ls + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < SGR_SORT_IPT; s++) {
        const uint32_t i = base + s * 64 + lane;
        if (i < n) {
            const uint32_t d = (uint32_t)(key[s] >> shift) & 255u;
            const uint32_t lp = cnt[wave][d] + rnk[s];
            sK[lp] = key[s];
            sV[lp] = val[s];
        }
    }
    __syncthreads();
    const uint32_t first = blockIdx.x * SGR_SORT_ITEMS;
    const uint32_t nloc = min((uint32_t)SGR_SORT_ITEMS, n - first);
#pragma unroll
    for (int s = 0; s < SGR_SORT_IPT; s++) {
        const uint32_t li = s * 256 + tid;
        if (li < nloc) {
            const K k = sK[li];
            const uint32_t d = (uint32_t)(k >> shift) & 255u;
            const uint32_t pos = gbase[d] + (li - lstart[d]);
            kout[pos] = k;
            vout[pos] = sV[li];
        }
    }
}

// Sorts n pairs on key bits [0, end_bit).  keys[0]/vals[0] hold the input; returns the index (0/1)
// of the pair of buffers that holds the sorted output.  hist: sgr_sort_hist_words(n) dwords (scan_tmp is unused).
// The per-block digit histogram depends on where the previous pass left the keys, so it is
// recomputed before every scatter pass.
template <typename K>
static int sort_pairs_impl(K* const keys[2], uint32_t* const vals[2], uint32_t n, int end_bit, uint32_t* hist,
                           uint32_t* scan_tmp, hipStream_t s) {
    if (n == 0) return 0;
    const int npass = (end_bit + 7) / 8;
    const uint32_t nblocks = (n + SGR_SORT_ITEMS - 1) / SGR_SORT_ITEMS;
    int cur = 0;
    for (int p = 0; p < npass; p++) {
        sgr_sort_hist_kernel<K><<<nblocks, 256, 0, s>>>(keys[cur], n, 8 * p, nblocks, hist);
        uint32_t* totals = hist + (size_t)256 * nblocks;
        sgr_sort_rowscan_kernel<<<256, 256, 0, s>>>(hist, nblocks, totals);
        sgr_sort_scatter_kernel<K><<<nblocks, 256, 0, s>>>(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, 8 * p,
                                                           nblocks, hist, totals);
        cur ^= 1;
    }
    return cur;
}
int sgr_launch_sort_pairs(uint64_t* const keys[2], uint32_t* const vals[2], uint32_t n, int end_bit, uint32_t* hist,
                          uint32_t* scan_tmp, hipStream_t s) {
    return sort_pairs_impl<uint64_t>(keys, vals, n, end_bit, hist, scan_tmp, s);
}
int sgr_launch_sort_pairs32(uint32_t* const keys[2], uint32_t* const vals[2], uint32_t n, int end_bit, uint32_t* hist,
                            uint32_t* scan_tmp, hipStream_t s) {
    return sort_pairs_impl<uint32_t>(keys, vals, n, end_bit, hist, scan_tmp, s);
}
