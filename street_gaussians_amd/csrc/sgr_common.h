// sgr_common.h -- shared definitions of the MI355X-native Gaussian rasterizer (gfx950 only).
//
// Data layout in HBM (all private to the native side; the three byte buffers are only
// round-tripped through the caller, as in the reference's GeometryState/BinningState/ImageState,
// /root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer/rasterizer_impl.h:29-64):
//
//   geometry buffer, per Gaussian g (struct-of-float4-arrays so every gather is one 16-B load):
//     rec[4g+0] = {pix.x, pix.y, hx, hy}      2D mean + conservative half extents of the alpha>=1/255 ellipse
//     rec[4g+1] = {conic.x, conic.y, conic.z, opacity}
//     rec[4g+2] = {r, g, b, view depth}
//     rec[4g+3] = {-, bits: packed tile rect, -, -}   (backward only)
//     -> ONE 64-byte, 64-byte-aligned record per Gaussian: a tile kernel's gather of an instance touches
//        a single cache line (three separate arrays cost three lines per instance: measured 2-4x over-fetch)
//     aux[g] = {tiles_touched, packed tile rect} (8 bytes; gathered into depth order by the depth sort's last pass),
//     clamped[g] (3-bit mask), radii (if not given); cov3D is NOT stored: the backward recomputes it from scale and
//     rotation with the same function (24 B/Gaussian less to write and to read back)
//   binning buffer: 32-bit tile keys and 32-bit Gaussian ids, ping-pong for the stable LSD radix sort by
//     tile (the instances are emitted in (depth, id) order, so the reference's 64-bit (tile|depth) key
//     order falls out of a stable sort on the tile bits alone), plus the per-block digit histograms and one
//     "hit" byte per sorted instance (which quadrants of its tile the forward blended it into: the backward
//     walks exactly those).
//   image buffer: n_contrib[H*W], ranges[tiles].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SGR_BLOCK_X 16
#define SGR_BLOCK_Y 16
#define SGR_WAVE 64
#define SGR_MAX_GRID_DIM 1023  // tiles per axis representable in the packed rect (10 bits)
#define SGR_SEM_MAX 32         // semantic channels supported by the blend kernels
#define SGR_ROW_BASE_N 11      // non-semantic floats of a partial-gradient row (see sgr_blend_bwd.hip)
#define SGR_DEAD 0x80000000u  // marked-list mode: bit 31 of a point_list entry = the instance lies outside the Gaussian's cut-down
                              // rect / tile mask: it cannot pass the alpha test anywhere in its tile (the blend kernels skip it
                              // without fetching its record; it owns no partial-gradient row)
#define SGR_RECT_MASKED 0x80000000u  // packed tile rect, bit 31: the Gaussian is emitted for the tiles of SgrGeomView::tmask only

#define SGR_ALPHA_MIN (1.0f / 255.0f)
#define SGR_DEPTH_KEY_BIAS 0x3E4CCCCDu  // bits of 0.2f: the near-plane test of the preprocess (auxiliary.h:152)
#define SGR_DEPTH_KEY_BITS 27           // key bits the depth sort looks at unless a depth >= 0.2 * 2^16 shows up
#define SGR_LOG2E 1.4426950408889634f

struct SgrGeomView {
    float4* rec;  // [4P]
    uint2* aux;               // per Gaussian {tiles_touched, packed tile rect}; {0, 0} when culled.  Rect word: x0 | y0 << 10 |
                              // w << 20, bit 31 = "only the tiles set in tmask[g] are emitted" (SGR_RECT_MASKED)
    uint64_t* tmask;          // per Gaussian with bit 31 of its rect word set: bit j = tile (x0 + j % w, y0 + j / w) of the rect
                              // is emitted (rects of at most 64 tiles; written for those Gaussians only)
    uint4* aux_ref;           // marked-list mode (preprocess `tight` = 3: the reference's list with the instances that cannot blend
                              // MARKED, switch bit 10): {tiles of the reference's rect, that rect, live tiles, cut-down rect} --
                              // .xy is what is scanned in depth order and emitted, .zw (= aux) says which of those instances
                              // are live; aux / tmask describe the cut-down rect, i.e. the rows
    uint2* aux_sorted;        // aux (marked-list mode: the 16-byte aux_ref records) in (depth, id) order (written by the last pass
                              // of the depth sort); 16 bytes per Gaussian are reserved
    uint32_t* u0;             // per Gaussian: its first partial-gradient row of the backward = EXCLUSIVE scan of
                              // tiles_touched in INDEX order (the reference's point_offsets minus tiles_touched), made by
                              // the second sequence of the forward's scan launches.  Rows in index order: the row sum
                              // streams them (in depth order it was a gather of ~190-byte runs: 1.67x the bytes)
    uint32_t* dkeys[2];       // depth bits per Gaussian (0xffffffff = culled), ping-pong for the depth sort
    uint32_t* dvals[2];       // Gaussian ids; after the sort dvals[cur] = ids in (depth, id) order (dvals[0] is never
                              // written by the preprocess: the first pass takes the element index as the value)
    uint32_t* sub_sums;       // the scan's 256-element sub-block offsets (both sequences): [2][ceil(P / 2048) * 8] -- where a
                              // 256-thread workgroup of sgr_duplicate_kernel starts inside its 2048-element scan block
    uint32_t* dhist;          // digit histogram of the depth sort
    uint32_t* clamped;  // 3-bit mask per Gaussian, one u32 each (keeps stores simple and aligned)
    int* internal_radii;
    uint32_t* scan_tmp;   // block sums of the device-wide scan
    uint32_t* header;     // [6]=tile-rect mode of the frame (0 reference rects, 1 bounding box, 2 box + mask, 3 marked list), [0]=error flag, [2]=depth beyond the 27-bit sort keys, [4]=num_rendered, [5]=num_rendered with the reference rects (one u64 counter), [16..]=SgrCam
};

// 1: the forward writes the compact hit list (SgrBinView::hlist, SgrImgView::n_contrib_k) and the blend backward walks it.
// 0 compiles it out of both kernels (tools/build_variant.py nohlist -DSGR_HLIST=0: what the list costs the forward, A/B).
#ifndef SGR_HLIST
#define SGR_HLIST 1
#endif

struct SgrBinView {
    uint32_t* keys[2];   // tile id per instance (the depth order is already in the emission order)
    uint32_t* vals[2];
    uint32_t* hist;      // [256][nblocks] per-pass digit histogram, exclusive-scanned in place
    uint32_t* scan_tmp;
    uint32_t* header;    // [0]=index (0/1) of the buffer pair holding the sorted result
    uint8_t* hit4;       // per sorted instance: bit q = the forward blended it into >= 1 pixel of quadrant q of its tile
    uint8_t* touched;    // per partial-gradient row: written by the backward (cleared by the forward's tile-ranges launch)
    uint32_t* hlist;     // per tile (at its range's start): the list positions, relative to the range, of the instances the forward
                         // blended into at least one quadrant, ascending -- the COMPACT list the blend backward walks (it has
                         // nothing to do for the others: dead instances of the marked-list mode, instances behind a saturated
                         // pixel block, culled ones); a pixel's last contributor's index in it + 1 is SgrImgView::n_contrib_k
    uint32_t* tkeys;     // per-tile LDS sort (sgr_tile_sort.hip, switch bit 12): depth-key scratch of the lists too long for LDS
};

// optional sink of the backward for this view's densification statistics (sgr_backward_ex); all three or none
#define SGR_STAT_SEG_MAX 128  // == SGR_MAX_STAT_SEGMENTS (include/sgr.h)
struct SgrStatSink {
    float* accum = nullptr;      // xyz_gradient_accum [P,2]
    float* denom = nullptr;      // [P,1]
    float* max_radii = nullptr;  // max_radii2D [P]
    // optional map from this call's Gaussians to the persistent rows (a frame renders a subset of the sub-models):
    // segment s covers [start[s], start[s] + count[s]) -> rows shift[s] + index; nseg == 0: identity
    int nseg = 0;
    int start[SGR_STAT_SEG_MAX];
    int count[SGR_STAT_SEG_MAX];
    int shift[SGR_STAT_SEG_MAX];  // dst_offset - src_start
};

struct SgrImgView {
    uint32_t* n_contrib;
    uint2* ranges;
    uint32_t* n_contrib_k;  // per pixel: n_contrib counted in entries of SgrBinView::hlist (last contributor's index in it + 1)
};

static inline size_t sgr_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

template <typename T>
static inline void sgr_carve(char*& p, T*& out, size_t count) {
    p = (char*)sgr_align_up((size_t)p, 256);
    out = (T*)p;
    p += count * sizeof(T);
}

#define SGR_SCAN_ITEMS 2048   // elements per block in the device-wide scan
#ifndef SGR_SORT_IPT
#define SGR_SORT_IPT 8         // keys per thread in the radix sort
#endif
#define SGR_SORT_ITEMS (256 * SGR_SORT_IPT)   // keys per block (256 threads)
#define SGR_SORT_MAX_PASS 8

static inline size_t sgr_scan_tmp_count(size_t n) { return (n + SGR_SCAN_ITEMS - 1) / SGR_SCAN_ITEMS + 1; }
// dwords of the radix sort's work area (sgr_scan_sort.hip): the [256 digits][nblocks] table + 256 totals; sized for
// the one-sweep A/B form, which needs more: a control block (digit counts of up to 8 passes, tickets, error flag) + a
// look-back table of 256 64-bit words per block
#define SGR_SORT_CTRL_WORDS (SGR_SORT_MAX_PASS * 256 + 64)
static inline size_t sgr_sort_hist_words(size_t n) {
    return (size_t)SGR_SORT_CTRL_WORDS + (size_t)512 * ((n + SGR_SORT_ITEMS - 1) / SGR_SORT_ITEMS);
}

// Carve the geometry buffer.  base may be (char*)256 to compute the required size: *end - base.
static inline SgrGeomView sgr_geom_carve(char* base, size_t P, char** end = nullptr) {
    SgrGeomView v;
    char* p = base;
    size_t Pn = P ? P : 1;
    sgr_carve(p, v.header, 64);
    sgr_carve(p, v.rec, Pn * 4);
    sgr_carve(p, v.aux, Pn);
    sgr_carve(p, v.aux_sorted, 2 * Pn);
    sgr_carve(p, v.aux_ref, Pn);
    sgr_carve(p, v.tmask, Pn);
    sgr_carve(p, v.u0, Pn);
    sgr_carve(p, v.clamped, Pn);
    sgr_carve(p, v.internal_radii, Pn);
    sgr_carve(p, v.dkeys[0], Pn);
    sgr_carve(p, v.dkeys[1], Pn);
    sgr_carve(p, v.dvals[0], Pn);
    sgr_carve(p, v.dvals[1], Pn);
    sgr_carve(p, v.sub_sums, 2 * 8 * ((Pn + SGR_SCAN_ITEMS - 1) / SGR_SCAN_ITEMS));
    {
        const size_t nh = sgr_sort_hist_words(Pn);
        sgr_carve(p, v.dhist, nh);
        sgr_carve(p, v.scan_tmp, 2 * sgr_scan_tmp_count(nh > Pn ? nh : Pn));  // two sequences per scan launch
    }
    if (end) *end = p;
    return v;
}

static inline size_t sgr_sort_blocks(size_t R) { return (R + SGR_SORT_ITEMS - 1) / SGR_SORT_ITEMS; }

static inline SgrBinView sgr_bin_carve(char* base, size_t R, char** end = nullptr) {
    SgrBinView v;
    char* p = base;
    size_t Rn = R ? R : 1;
    size_t nb = sgr_sort_blocks(Rn);
    size_t nh = sgr_sort_hist_words(Rn);
    (void)nb;
    sgr_carve(p, v.header, 64);
    sgr_carve(p, v.keys[0], Rn);
    sgr_carve(p, v.keys[1], Rn);
    sgr_carve(p, v.vals[0], Rn);
    sgr_carve(p, v.vals[1], Rn);
    sgr_carve(p, v.hist, nh);
    sgr_carve(p, v.scan_tmp, sgr_scan_tmp_count(nh));
    sgr_carve(p, v.hit4, Rn);
    sgr_carve(p, v.touched, Rn);
    sgr_carve(p, v.tkeys, Rn);
    sgr_carve(p, v.hlist, Rn);
    if (end) *end = p;
    return v;
}

static inline SgrImgView sgr_img_carve(char* base, size_t N, size_t T, char** end = nullptr) {
    SgrImgView v;
    char* p = base;
    sgr_carve(p, v.n_contrib, N ? N : 1);
    sgr_carve(p, v.ranges, (T ? T : 1) + ((T ? T : 1) + 2) / 2);  // + T + 1 words behind the ranges: the tile-order block (sgr_wg_tile)
    sgr_carve(p, v.n_contrib_k, N ? N : 1);
    if (end) *end = p;
    return v;
}

template <typename F>
static inline size_t sgr_required(F carve) {
    char* end = nullptr;
    char* base = (char*)4096;
    carve(base, &end);
    return (size_t)(end - base) + 256;  // +256: the caller's pointer is re-aligned up to 256 B
}

#ifdef __HIPCC__
#include <mutex>
// process-wide 256-byte device block of the current device, held (mutex) until the struct is destroyed (sgr_api.hip)
struct SgrFlagBlock {
    uint32_t* ptr = nullptr;
    std::unique_lock<std::mutex> lock;
};
SgrFlagBlock sgr_acquire_flag_block();
// ---- primitives shared by several translation units (defined in sgr_scan_sort.hip); declared HERE only, so a changed
// signature cannot leave a stale copy behind in another file --------------------------------------------------------
// device-wide scan: out may alias in; tmp needs sgr_scan_tmp_count(n) words; tmp[nblocks] (and *total_out) receive the
// grand total; gather != nullptr scans in[gather[i]] instead of in[i]
// in_stride: element i is in[i * in_stride] (scan of one field of an array of records)
// in2 / out2: a second sequence of n elements scanned (exclusive, no gather, same stride) in the same three launches; tmp
// then needs 2 * sgr_scan_tmp_count(n) words
void sgr_launch_scan(const uint32_t* in, uint32_t* out, size_t n, uint32_t* tmp, bool inclusive, hipStream_t s,
                     uint32_t* total_out = nullptr, const uint32_t* gather = nullptr, int in_stride = 1,
                     const uint32_t* in2 = nullptr, uint32_t* out2 = nullptr);
// The first two of the scan's three launches for TWO sequences of n elements (in[i * in_stride], in2[i * in_stride]): block
// sums reduced and scanned (tmp[b] = exclusive prefix of 2048-element block b, tmp[nb] = total; second sequence behind it at
// tmp + nb + 1), and sub[seq * 8 nb + 8 b + j] = the offset of 256-element sub-block j inside block b.  The consumer does
// the last step itself: element i = tmp[i / 2048] + sub[i / 256] + its exclusive prefix inside its 256-element sub-block
// (the forward's duplicate kernel: no third launch, no scanned array written and read back).
void sgr_launch_scan_head(const uint32_t* in, const uint32_t* in2, size_t n, int in_stride, uint32_t* tmp, uint32_t* sub,
                          hipStream_t s, int in2_stride = 0);  // in2_stride: stride of the second sequence (0: in_stride)
// stable LSD radix sorts on key bits [0, end_bit); return the index (0/1) of the buffer pair holding the result
int sgr_sort_get_one_sweep();
void sgr_sort_set_one_sweep(int on);  // A/B: 1 = the one-sweep form instead of histogram + row scan + scatter per pass
int sgr_launch_sort_pairs(uint64_t* const keys[2], uint32_t* const vals[2], uint32_t n, int end_bit, uint32_t* hist,
                          uint32_t* scan_tmp, hipStream_t s);
// iota: vals[0] is not read (value of element i = i); aux_in / aux_out: the last pass also gathers an 8-byte per-id record
// into sorted order, aux_out[sorted position] = aux_in[value]
int sgr_launch_sort_pairs32(uint32_t* const keys[2], uint32_t* const vals[2], uint32_t n, int end_bit, uint32_t* hist,
                            uint32_t* scan_tmp, hipStream_t s, bool iota = false, const uint2* aux_in = nullptr,
                            uint2* aux_out = nullptr, int max_bits = 8, int aux16 = 0);  // max_bits: digit width cap, 8 or 9;
                            // aux16: the aux records are 16 bytes (uint4) instead of 8
int sgr_launch_sort_pairs16(uint16_t* const keys[2], uint32_t* const vals[2], uint32_t n, int end_bit, uint32_t* hist,
                            uint32_t* scan_tmp, hipStream_t s);
int sgr_sort_pass_count(int end_bit);  // passes (= buffer flips) of a sort on key bits [0, end_bit)
// per-tile LDS sort by depth (sgr_tile_sort.hip): vals_in (tile-major, ascending id inside a tile) -> vals_out in (depth, id) order
void sgr_launch_tile_sort(int T, const uint2* ranges, uint32_t* vals_in, uint32_t* vals_out, const uint32_t* dkeys,
                          const uint32_t* header, uint32_t* gk0, uint32_t* gk1, hipStream_t s);

// ---- wave / block scan primitives (sgr_scan_sort.hip, sgr_preprocess.hip) ----
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

// XCD-aware workgroup -> tile map.  The dispatcher places workgroup b on XCD b % 8 (observed, used for speed only),
// and each XCD has a private 4 MiB L2.  A splat's instances live in neighbouring tiles, so neighbouring tiles
// should share an L2: workgroups are handed out in 8x8-tile supertiles (128x128 px), supertile k going to XCD
// k % 8.  Returns false for the padding workgroups past the last supertile / outside the grid.
#define SGR_ST 8
static inline unsigned sgr_xcd_grid_blocks(int gx, int gy) {
    const unsigned nst = (unsigned)((gx + SGR_ST - 1) / SGR_ST) * (unsigned)((gy + SGR_ST - 1) / SGR_ST);
    return ((nst + 7u) / 8u) * 8u * (SGR_ST * SGR_ST);
}
__device__ __forceinline__ bool sgr_xcd_tile(uint32_t b, uint32_t gx, uint32_t gy, uint32_t& tx, uint32_t& ty) {
    const uint32_t x = b & 7u, q = b >> 3;
    const uint32_t st = (q / (SGR_ST * SGR_ST)) * 8u + x, within = q % (SGR_ST * SGR_ST);
    const uint32_t sgx = (gx + SGR_ST - 1) / SGR_ST;
    tx = (st % sgx) * SGR_ST + (within % SGR_ST);
    ty = (st / sgx) * SGR_ST + (within / SGR_ST);
    return tx < gx && ty < gy;
}

// Tile of workgroup b.  gy >= 0: the XCD-aware supertile order above.  gy < 0: the grid is |gy| rows high and the frame's
// tile-order block sits behind ranges[T] -- T tile ids sorted by descending list length, then ONE flag word: 1 = walk the
// tiles in that LONGEST-FIRST order, 0 = the lists are even enough, use the supertile order (sgr_tile_order_kernel decides
// per frame).  Heaviest tiles first shortens the ragged end of a launch whose tiles are very unequal -- a street scene: empty
// sky next to actors, -10 % of the step -- at the price of the L2 locality of neighbouring tiles, which is what an even
// scene lives on (+1 %): measured, DESIGN.md section 3.
__device__ __forceinline__ bool sgr_wg_tile(uint32_t b, int gx, int gy, const uint2* __restrict__ ranges, uint32_t& tx, uint32_t& ty) {
    if (gy >= 0) return sgr_xcd_tile(b, (uint32_t)gx, (uint32_t)gy, tx, ty);
    const uint32_t T = (uint32_t)gx * (uint32_t)(-gy);
    const uint32_t* order = reinterpret_cast<const uint32_t*>(ranges + T);
    if (order[T] == 0u) return sgr_xcd_tile(b, (uint32_t)gx, (uint32_t)(-gy), tx, ty);
    if (b >= T) return false;
    const uint32_t t = order[b];
    tx = t % (uint32_t)gx;
    ty = t / (uint32_t)gx;
    return true;
}

// Move a wave-uniform 64-bit value into SGPRs.  __builtin_amdgcn_readfirstlane returns a SIGNED int:
// each half must go through uint32_t, otherwise bit 31 of the low word sign-extends over the high word.
__device__ __forceinline__ uint64_t sgr_uniform_u64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | (uint64_t)lo;
}
// Scalar bit bookkeeping for wave-uniform 64-bit masks (hipcc expands `m &= m - 1` into s_add_u32 / s_addc_u32 /
// s_and_b64 and `hb |= 1ull << b` into s_lshl_b64 / s_or_b64; the walks of the blend kernels are SALU-heavy).
__device__ __forceinline__ int sgr_pop_lowest(uint64_t& m) {  // m != 0: index of its lowest set bit, cleared in m
    const int b = __builtin_ctzll(m);
    asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(b));
    return b;
}
__device__ __forceinline__ uint64_t sgr_bitset1(uint64_t m, int b) {
    asm("s_bitset1_b64 %0, %1" : "+s"(m) : "s"(b));
    return m;
}
#endif
