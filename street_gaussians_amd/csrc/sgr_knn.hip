// sgr_knn.hip -- simple-knn's distCUDA2 on gfx950: mean squared distance to the 3 nearest neighbours.
// Replaces SimpleKNN::knn (/root/reference/submodules/simple-knn/simple_knn.cu:185-220): K14 min/max
// reduce (with the reference's {0,0,0} init), K15 Morton codes, K16 radix sort (our own sort from
// sgr_scan_sort.hip on the 32 code bits), K17 per-1024 box AABBs, K18 pruned brute force.  No host
// round trip: the bounding box stays in device memory.  FP contraction is off so the codes and the
// distances reproduce the oracle bit for bit.
//
// K17/K18 work on the points GATHERED INTO MORTON ORDER once (12 B per point of scratch), so a box is a contiguous
// 12 KB run.  K18 is wave-cooperative: the 64 lanes of a wave own 64 consecutive (spatially close) points; a candidate
// box is scanned when ANY lane needs it -- its points are loaded 64 at a time with one coalesced 768-byte read and
// handed to all lanes by v_readlane (scalar broadcast), every lane applying the reference's own per-point predicate and
// insertion order, so the result is the reference's bit for bit.  (The reference walks the boxes with one thread per
// point, a divergent loop of scattered 12-byte gathers, simple_knn.cu:147-183.)
#include <string>

#include "../../include/sgr.h"
#include "sgr_common.h"
#include <float.h>

#define SGR_KNN_BOX 1024  // simple_knn.cu:12


struct SgrBox { float mn[3], mx[3]; };

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide AABB of up to 256*ITEMS points; result valid in thread 0
template <typename LoadF>
__device__ __forceinline__ void block_aabb(LoadF load, uint32_t first, uint32_t count, float* mn, float* mx,
                                           float (*lds)[6]) {
    float lmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, lmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) {
        float p[3];
        load(first + i, p);
#pragma unroll
        for (int k = 0; k < 3; k++) { lmn[k] = fminf(lmn[k], p[k]); lmx[k] = fmaxf(lmx[k], p[k]); }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; k++) { lmn[k] = wave_min(lmn[k]); lmx[k] = wave_max(lmx[k]); }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { lds[wave][k] = lmn[k]; lds[wave][3 + k] = lmx[k]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
#pragma unroll
        for (int k = 0; k < 3; k++) { mn[k] = lds[0][k]; mx[k] = lds[0][3 + k]; }
        for (int w = 1; w < nw; w++) {
#pragma unroll
            for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], lds[w][k]); mx[k] = fmaxf(mx[k], lds[w][3 + k]); }
        }
    }
}

// K14 stage 1: per-block AABB
__global__ void __launch_bounds__(256) sgr_knn_minmax1_kernel(uint32_t P, const float* __restrict__ pts, SgrBox* out) {
    __shared__ float lds[4][6];
    const uint32_t first = blockIdx.x * 4096u;
    const uint32_t count = min(4096u, P - first);
    float mn[3], mx[3];
    block_aabb([&](uint32_t i, float* p) { p[0] = pts[3 * (size_t)i]; p[1] = pts[3 * (size_t)i + 1]; p[2] = pts[3 * (size_t)i + 2]; },
               first, count, mn, mx, lds);
    if (threadIdx.x == 0) {
        for (int k = 0; k < 3; k++) { out[blockIdx.x].mn[k] = mn[k]; out[blockIdx.x].mx[k] = mx[k]; }
    }
}
// K14 stage 2: one block; init {0,0,0} as cub::DeviceReduce::Reduce(..., init) does (simple_knn.cu:191-199)
__global__ void __launch_bounds__(256) sgr_knn_minmax2_kernel(uint32_t nb, const SgrBox* __restrict__ in, SgrBox* out) {
    __shared__ float lds[4][6];
    float lmn[3] = {0.f, 0.f, 0.f}, lmx[3] = {0.f, 0.f, 0.f};
    for (uint32_t i = threadIdx.x; i < nb; i += 256) {
#pragma unroll
        for (int k = 0; k < 3; k++) { lmn[k] = fminf(lmn[k], in[i].mn[k]); lmx[k] = fmaxf(lmx[k], in[i].mx[k]); }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; k++) { lmn[k] = wave_min(lmn[k]); lmx[k] = wave_max(lmx[k]); }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { lds[wave][k] = lmn[k]; lds[wave][3 + k] = lmx[k]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k < 3; k++) {
            out->mn[k] = fminf(fminf(lds[0][k], lds[1][k]), fminf(lds[2][k], lds[3][k]));
            out->mx[k] = fmaxf(fmaxf(lds[0][3 + k], lds[1][3 + k]), fmaxf(lds[2][3 + k], lds[3][3 + k]));
        }
    }
}

__device__ __forceinline__ uint32_t prepMorton(uint32_t x) {  // simple_knn.cu:45-52
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}

// K15 (simple_knn.cu:54-70) + thrust::sequence (simple_knn.cu:207)
__global__ void __launch_bounds__(256)
sgr_knn_morton_kernel(uint32_t P, const float* __restrict__ pts, const SgrBox* __restrict__ bb, uint64_t* keys, uint32_t* vals) {
#pragma clang fp contract(off)
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
    const uint32_t mx = prepMorton((uint32_t)(((x - bb->mn[0]) / (bb->mx[0] - bb->mn[0])) * ((1 << 10) - 1)));
    const uint32_t my = prepMorton((uint32_t)(((y - bb->mn[1]) / (bb->mx[1] - bb->mn[1])) * ((1 << 10) - 1)));
    const uint32_t mz = prepMorton((uint32_t)(((z - bb->mn[2]) / (bb->mx[2] - bb->mn[2])) * ((1 << 10) - 1)));
    keys[i] = (uint64_t)(mx | (my << 1) | (mz << 2));
    vals[i] = i;
}

// points in Morton order: spts[i] = pts[indices[i]]
__global__ void __launch_bounds__(256)
sgr_knn_gather_kernel(uint32_t P, const float* __restrict__ pts, const uint32_t* __restrict__ indices, float* __restrict__ spts) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const size_t j = indices[i];
    spts[3 * (size_t)i] = pts[3 * j];
    spts[3 * (size_t)i + 1] = pts[3 * j + 1];
    spts[3 * (size_t)i + 2] = pts[3 * j + 2];
}

// K17 (simple_knn.cu:78-117): AABB of each run of 1024 Morton-sorted points
__global__ void __launch_bounds__(256)
sgr_knn_boxes_kernel(uint32_t P, const float* __restrict__ spts, SgrBox* boxes) {
    __shared__ float lds[4][6];
    const uint32_t first = blockIdx.x * SGR_KNN_BOX;
    const uint32_t count = min((uint32_t)SGR_KNN_BOX, P - first);
    float mn[3], mx[3];
    block_aabb([&](uint32_t i, float* p) { p[0] = spts[3 * (size_t)i]; p[1] = spts[3 * (size_t)i + 1]; p[2] = spts[3 * (size_t)i + 2]; },
               first, count, mn, mx, lds);
    if (threadIdx.x == 0) {
        for (int k = 0; k < 3; k++) { boxes[blockIdx.x].mn[k] = mn[k]; boxes[blockIdx.x].mx[k] = mx[k]; }
    }
}

__device__ __forceinline__ float distBoxPoint(const SgrBox& box, const float* p) {  // simple_knn.cu:119-129
#pragma clang fp contract(off)
    float d[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (p[k] < box.mn[k] || p[k] > box.mx[k]) d[k] = fminf(fabsf(p[k] - box.mn[k]), fabsf(p[k] - box.mx[k]));
    return d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
}
__device__ __forceinline__ void updateKBest3(const float* ref, const float* q, float* knn) {  // simple_knn.cu:131-145
#pragma clang fp contract(off)
    const float dx = q[0] - ref[0], dy = q[1] - ref[1], dz = q[2] - ref[2];
    float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (knn[j] > dist) {
            const float t = knn[j];
            knn[j] = dist;
            dist = t;
        }
    }
}

__device__ __forceinline__ float sgr_lane_bcast(float v, int k) {  // value of lane k (k wave-uniform) as a scalar
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
}

// K18 (simple_knn.cu:147-183), wave-cooperative (see the header of this file)
__global__ void __launch_bounds__(256)
sgr_knn_meandist_kernel(uint32_t P, const float* __restrict__ spts, const uint32_t* __restrict__ indices,
                        const SgrBox* __restrict__ boxes, float* __restrict__ dists) {
#pragma clang fp contract(off)
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool live = idx < (int)P;
    const int ci = live ? idx : (int)P - 1;  // lanes past P stay in the wave for the cooperative loads
    const float point[3] = {spts[3 * (size_t)ci], spts[3 * (size_t)ci + 1], spts[3 * (size_t)ci + 2]};
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    if (live) {
        for (int i = max(0, idx - 3); i <= min((int)P - 1, idx + 3); i++) {  // :156-162
            if (i == idx) continue;
            const float q[3] = {spts[3 * (size_t)i], spts[3 * (size_t)i + 1], spts[3 * (size_t)i + 2]};
            updateKBest3(point, q, best);
        }
    }
    const float reject = best[2];
    best[0] = FLT_MAX; best[1] = FLT_MAX; best[2] = FLT_MAX;
    const int nboxes = (int)((P + SGR_KNN_BOX - 1) / SGR_KNN_BOX);
    for (int b = 0; b < nboxes; b++) {
        const SgrBox box = boxes[b];  // wave-uniform address: scalar loads
        const float dist = distBoxPoint(box, point);
        const bool need = live && !(dist > reject || dist > best[2]);  // the reference's per-point test (:170-172)
        if (__ballot(need) == 0ull) continue;                          // no lane of the wave wants this box
        const int lo = b * SGR_KNN_BOX, hi = min((int)P, lo + SGR_KNN_BOX);
        for (int c = lo; c < hi; c += 64) {
            const int i = c + lane;
            float q0 = 0.f, q1 = 0.f, q2 = 0.f;
            if (i < hi) { q0 = spts[3 * (size_t)i]; q1 = spts[3 * (size_t)i + 1]; q2 = spts[3 * (size_t)i + 2]; }
            const int n = min(64, hi - c);
            for (int k = 0; k < n; k++) {  // candidates in ascending order, exactly as the per-thread loop visits them
                const float q[3] = {sgr_lane_bcast(q0, k), sgr_lane_bcast(q1, k), sgr_lane_bcast(q2, k)};
                if (need && (c + k) != idx) updateKBest3(point, q, best);
            }
        }
    }
    if (live) dists[indices[idx]] = (best[0] + best[1] + best[2]) / 3.0f;
}

int sgr_knn_impl(int P, const float* points, float* meanDists, sgr_alloc_fn scratch, void* scratch_user, hipStream_t s,
                 std::string& err) {
    if (P <= 0) return 0;
    if (!points || !meanDists || !scratch) { err = "points, meanDists and scratch are required"; return -SGR_E_INVALID; }
    const uint32_t n = (uint32_t)P;
    const uint32_t nb1 = (n + 4095u) / 4096u;
    const uint32_t nboxes = (n + SGR_KNN_BOX - 1) / SGR_KNN_BOX;
    const size_t nh = sgr_sort_hist_words(n);
    // carve the scratch
    char* p = (char*)256;
    uint64_t* keys[2]; uint32_t* vals[2]; uint32_t *hist, *scan_tmp; SgrBox *part, *bb, *boxes; float* spts;
    auto carve_all = [&](char* base) {
        char* q = base;
        sgr_carve(q, keys[0], (size_t)n); sgr_carve(q, keys[1], (size_t)n);
        sgr_carve(q, vals[0], (size_t)n); sgr_carve(q, vals[1], (size_t)n);
        sgr_carve(q, hist, nh); sgr_carve(q, scan_tmp, sgr_scan_tmp_count(nh));
        sgr_carve(q, part, (size_t)nb1); sgr_carve(q, bb, (size_t)1); sgr_carve(q, boxes, (size_t)nboxes);
        sgr_carve(q, spts, (size_t)n * 3);
        return q;
    };
    const size_t bytes = (size_t)(carve_all(p) - p) + 256;
    char* base = scratch(bytes, scratch_user);
    if (!base) { err = "knn scratch allocation failed"; return -SGR_E_ALLOC; }
    carve_all(base);

    sgr_knn_minmax1_kernel<<<nb1, 256, 0, s>>>(n, points, part);
    sgr_knn_minmax2_kernel<<<1, 256, 0, s>>>(nb1, part, bb);
    sgr_knn_morton_kernel<<<(n + 255) / 256, 256, 0, s>>>(n, points, bb, keys[0], vals[0]);
    const int cur = sgr_launch_sort_pairs(keys, vals, n, 32, hist, scan_tmp, s);
    sgr_knn_gather_kernel<<<(n + 255) / 256, 256, 0, s>>>(n, points, vals[cur], spts);
    sgr_knn_boxes_kernel<<<nboxes, 256, 0, s>>>(n, spts, boxes);
    sgr_knn_meandist_kernel<<<(n + 255) / 256, 256, 0, s>>>(n, spts, vals[cur], boxes, meanDists);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("knn: ") + hipGetErrorString(e); return -SGR_E_HIP; }
    return 0;
}
