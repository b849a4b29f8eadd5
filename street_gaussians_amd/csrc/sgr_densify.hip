// sgr_densify.hip -- adaptive density control as plan + gather (include/sgr_densify.h).
//
// Reference order of operations (gaussian_model.py:522-553): clone (appends copies), split (appends N children per
// selected point -- only ORIGINAL points can be selected, the clones' padded gradient is 0, :455-460 -- and removes the
// selected points), prune (low opacity / too big, evaluated on the new set), reset statistics.  Every decision is a
// function of the ORIGINAL point's statistics: a clone has its parent's parameters, a split child its parent's opacity
// and its parent's scale divided by 0.8 N.  So one pass computes, per original point, four masks
//     A  survives as itself        = !split && !pruned(self)
//     B  leaves a surviving clone  = clone && !pruned(self)
//     S  is split                  (rank among S = the child's row in the repeat(N,1) layout)
//     C  leaves surviving children = split && !pruned(child)
// exclusive scans of the masks give every result row its position, in the reference's order.
#include <algorithm>
#include <cstdint>
#include <string>

#include "../../include/sgr_densify.h"
#include "sgr_common.h"

int sgr_set_error(int code, const std::string& msg);

#define DN_HIP(call)                                                                                       \
    do {                                                                                                   \
        hipError_t e__ = (call);                                                                           \
        if (e__ != hipSuccess) return sgr_set_error(SGR_E_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

struct DnWork {
    uint32_t *flags, *offA, *offB, *offS, *offC, *tmp, *totals;  // totals: nA nB nS nC nClone nPrunedCand
};
static DnWork dn_carve(char* base, size_t N) {
    DnWork w;
    char* p = base;
    const size_t n = N ? N : 1;
    sgr_carve(p, w.flags, n);
    sgr_carve(p, w.offA, n);
    sgr_carve(p, w.offB, n);
    sgr_carve(p, w.offS, n);
    sgr_carve(p, w.offC, n);
    sgr_carve(p, w.tmp, sgr_scan_tmp_count(n));
    sgr_carve(p, w.totals, 16);
    return w;
}

#define DN_CLONE 1u
#define DN_SPLIT 2u
#define DN_PRUNE_SELF 4u
#define DN_PRUNE_CHILD 8u

__global__ void __launch_bounds__(256)
sgr_densify_flags_kernel(int N, sgr_densify_params p, const float* __restrict__ accum, const float* __restrict__ denom,
                         const float* __restrict__ scaling, const float* __restrict__ opacity, DnWork w) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float g = accum[2 * i + p.grad_column] / denom[i];  // :523
    if (g != g) g = 0.0f;                                // grads[grads.isnan()] = 0.0 (:524)
    const float s0 = expf(scaling[3 * i]), s1 = expf(scaling[3 * i + 1]), s2 = expf(scaling[3 * i + 2]);
    const float smax = fmaxf(s0, fmaxf(s1, s2));
    const float dense = p.percent_dense * p.extent;
    const bool clone = (fabsf(g) >= p.max_grad) && (smax <= dense);  // :497-499 (norm of a 1-vector)
    const bool split = (g >= p.max_grad) && (smax > dense);          // :462-464
    const float op = 1.0f / (1.0f + expf(-opacity[i]));
    const bool low = op < p.min_opacity;                              // :533
    const float big = p.extent * p.percent_big_ws;
    const bool prune_self = !p.defer_prune && (low || (p.prune_big && smax > big));  // :536-540
    // children: log(scale / (0.8 N)) -> exp gives scale / (0.8 N) again (up to rounding, like the reference's log/exp)
    const float child = expf(logf(smax / (0.8f * (float)p.n_split)));
    const bool prune_child = !p.defer_prune && (low || (p.prune_big && child > big));
    const uint32_t f = (clone ? DN_CLONE : 0u) | (split ? DN_SPLIT : 0u) | (prune_self ? DN_PRUNE_SELF : 0u) |
                       (prune_child ? DN_PRUNE_CHILD : 0u);
    w.flags[i] = f;
    w.offA[i] = (!split && !prune_self) ? 1u : 0u;
    w.offB[i] = (clone && !prune_self) ? 1u : 0u;
    w.offS[i] = split ? 1u : 0u;
    w.offC[i] = (split && !prune_child) ? 1u : 0u;
}

// number of cloned points (not a scan total: clones that are pruned again still count, gaussian_model.py:500-502).  An
// integer count -- order-independent, so a ballot per wave and one atomic per workgroup; the single-wave loop it replaces
// took 10 ms at 5 M points (rocprofv3), 80 % of the whole densify_and_prune.
__global__ void __launch_bounds__(256)
sgr_densify_count_kernel(int N, DnWork w) {
    __shared__ uint32_t part[4];
    uint32_t c = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) c += (w.flags[i] & DN_CLONE) ? 1u : 0u;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(&w.totals[4], t);
    }
}

__global__ void __launch_bounds__(256)
sgr_densify_map_kernel(int N, int n_split, DnWork w, int32_t* __restrict__ src, uint8_t* __restrict__ kind,
                       int32_t* __restrict__ sample_row) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint32_t f = w.flags[i];
    const uint32_t nA = w.totals[0], nB = w.totals[1], nS = w.totals[2], nC = w.totals[3];
    const bool split = f & DN_SPLIT;
    if (!split && !(f & DN_PRUNE_SELF)) {
        const uint32_t o = w.offA[i];
        src[o] = i; kind[o] = SGR_KIND_KEEP; sample_row[o] = -1;
    }
    if ((f & DN_CLONE) && !(f & DN_PRUNE_SELF)) {
        const uint32_t o = nA + w.offB[i];
        src[o] = i; kind[o] = SGR_KIND_CLONE; sample_row[o] = -1;
    }
    if (split && !(f & DN_PRUNE_CHILD)) {
        for (int n = 0; n < n_split; n++) {  // repeat(N, 1): copy-major (:468-476)
            const uint32_t o = nA + nB + (uint32_t)n * nC + w.offC[i];
            src[o] = i; kind[o] = SGR_KIND_SPLIT_CHILD; sample_row[o] = (int32_t)((uint32_t)n * nS + w.offS[i]);
        }
    }
}

// One workgroup builds 128 consecutive result rows: their source rows / kinds are staged in LDS once, then one lane per
// float walks the 128 * width outputs (coalesced stores, row-wise contiguous loads, 32-bit index arithmetic).
#define SGR_DN_ROWS 128
// WT > 0: the row width as a compile-time constant (the per-element row / column split is a multiply-shift instead of a
// 32-bit division, which was most of the kernel's instructions: 1 TB/s with the runtime width); WT == 0: any width.
template <int WT>
__global__ void __launch_bounds__(256)
sgr_densify_gather_kernel(int n_out, int width, const float* __restrict__ in, const int32_t* __restrict__ src,
                          const uint8_t* __restrict__ kind, int zero_new, float* __restrict__ out) {
    __shared__ int32_t sSrc[SGR_DN_ROWS];
    __shared__ uint8_t sKind[SGR_DN_ROWS];
    const int r0 = blockIdx.x * SGR_DN_ROWS;
    const int rows = min(SGR_DN_ROWS, n_out - r0);
    if ((int)threadIdx.x < rows) {
        sSrc[threadIdx.x] = src[r0 + threadIdx.x];
        sKind[threadIdx.x] = kind[r0 + threadIdx.x];
    }
    __syncthreads();
    const uint32_t w = WT > 0 ? (uint32_t)WT : (uint32_t)width;
    float* o = out + (size_t)r0 * w;
    const uint32_t total = (uint32_t)rows * w;
    if (WT > 0 && (WT & 3) == 0) {
        // rows are whole float4s (and 16-byte aligned: hipMalloc'ed tensors, width a multiple of 4): 16-byte copies
        constexpr uint32_t W4 = WT > 0 ? (uint32_t)WT / 4u : 1u;
        const float4* in4 = reinterpret_cast<const float4*>(in);
        float4* o4 = reinterpret_cast<float4*>(o);
        for (uint32_t e = threadIdx.x; e < (uint32_t)rows * W4; e += 256) {
            const uint32_t r = e / W4, j = e - r * W4;
            o4[e] = (zero_new && sKind[r] != SGR_KIND_KEEP) ? make_float4(0.f, 0.f, 0.f, 0.f) : in4[(size_t)sSrc[r] * W4 + j];
        }
        return;
    }
    for (uint32_t e = threadIdx.x; e < total; e += 256) {
        const uint32_t r = e / w, j = e - r * w;
        o[e] = (zero_new && sKind[r] != SGR_KIND_KEEP) ? 0.0f : in[(size_t)sSrc[r] * w + j];
    }
}

__global__ void __launch_bounds__(256)
sgr_densify_children_kernel(int n_out, int n_split, const int32_t* __restrict__ src, const uint8_t* __restrict__ kind,
                            const int32_t* __restrict__ sample_row, const float* __restrict__ xyz,
                            const float* __restrict__ scaling, const float* __restrict__ rotation,
                            const float* __restrict__ normals, float* __restrict__ xyz_out, float* __restrict__ scaling_out) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out || kind[o] != SGR_KIND_SPLIT_CHILD) return;
    const size_t i = (size_t)src[o], z = (size_t)sample_row[o];
    const float s[3] = {expf(scaling[3 * i]), expf(scaling[3 * i + 1]), expf(scaling[3 * i + 2])};
    const float v[3] = {normals[3 * z] * s[0], normals[3 * z + 1] * s[1], normals[3 * z + 2] * s[2]};  // normal(0, std)
    // quaternion_to_matrix(self._rotation[sel]) (general_utils.py:125-146): raw quaternion divided by its norm
    float qw = rotation[4 * i], qx = rotation[4 * i + 1], qy = rotation[4 * i + 2], qz = rotation[4 * i + 3];
    const float nrm = sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
    qw /= nrm; qx /= nrm; qy /= nrm; qz /= nrm;
    const float R[9] = {1.f - 2.f * (qy * qy + qz * qz), 2.f * (qx * qy - qw * qz), 2.f * (qx * qz + qw * qy),
                        2.f * (qx * qy + qw * qz), 1.f - 2.f * (qx * qx + qz * qz), 2.f * (qy * qz - qw * qx),
                        2.f * (qx * qz - qw * qy), 2.f * (qy * qz + qw * qx), 1.f - 2.f * (qx * qx + qy * qy)};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        xyz_out[3 * (size_t)o + a] = R[3 * a] * v[0] + R[3 * a + 1] * v[1] + R[3 * a + 2] * v[2] + xyz[3 * i + a];  // :471
        scaling_out[3 * (size_t)o + a] = logf(s[a] / (0.8f * (float)n_split));                                          // :472
    }
}

// ---- prune rules on the candidate set (include/sgr_densify.h) ----------------------------------------------------
struct DnSphere { float cx, cy, cz, r; };
struct DnBox { float lo[3], hi[3]; };
__global__ void __launch_bounds__(256)
sgr_densify_prune_kernel(int n, sgr_densify_params p, int variant, const float* __restrict__ xyz,
                         const float* __restrict__ scaling, const float* __restrict__ rotation,
                         const float* __restrict__ opacity, DnSphere sph, DnBox box,
                         const float* __restrict__ box_normals, uint8_t* __restrict__ prune, uint32_t* __restrict__ cnt) {
#pragma clang fp contract(off)
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    bool low = false, big = false, outside = false;
    {
        const float op = 1.0f / (1.0f + expf(-opacity[i]));
        low = op < p.min_opacity;
        const float s[3] = {expf(scaling[3 * i]), expf(scaling[3 * i + 1]), expf(scaling[3 * i + 2])};
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        if (p.prune_big) {
            big = fmaxf(s[0], fmaxf(s[1], s[2])) > p.extent * p.percent_big_ws;
            if (variant == SGR_PRUNE_BKGD) {  // gaussian_model_bkgd.py:95-97
                const float dx = x - sph.cx, dy = y - sph.cy, dz = z - sph.cz;
                const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
                if (dist > 2.0f * sph.r) big = false;
            }
            if (variant == SGR_PRUNE_ACTOR) {  // gaussian_model_actor.py:231-249
                float qw = rotation[4 * i], qx = rotation[4 * i + 1], qy = rotation[4 * i + 2], qz = rotation[4 * i + 3];
                const float nrm = sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
                qw /= nrm; qx /= nrm; qy /= nrm; qz /= nrm;
                const float R[9] = {1.f - 2.f * (qy * qy + qz * qz), 2.f * (qx * qy - qw * qz), 2.f * (qx * qz + qw * qy),
                                    2.f * (qx * qy + qw * qz), 1.f - 2.f * (qx * qx + qz * qz), 2.f * (qy * qz - qw * qx),
                                    2.f * (qx * qz - qw * qy), 2.f * (qy * qz + qw * qx), 1.f - 2.f * (qx * qx + qy * qy)};
                const float c[3] = {x, y, z};
                bool inside = true;
                for (int m = 0; m < 2; m++) {
                    const float* zn = box_normals + (size_t)(2 * i + m) * 3;
                    const float v[3] = {zn[0] * s[0], zn[1] * s[1], zn[2] * s[2]};
#pragma unroll
                    for (int a = 0; a < 3; a++) {
                        const float w = R[3 * a] * v[0] + R[3 * a + 1] * v[1] + R[3 * a + 2] * v[2] + c[a];
                        inside = inside && (w >= box.lo[a]) && (w <= box.hi[a]);
                    }
                }
                outside = !inside;
            }
        }
        prune[i] = (low || big || outside) ? 1 : 0;
    }
    c0 += low; c1 += big; c2 += outside; c3 += (low || big || outside);
    }
    // four counters (integers: order-independent): per-thread over the grid-stride loop, then one atomic each per
    // WORKGROUP -- one per wave were 440 k device-scope atomics on four words at 7 M candidates, 3.7 ms (rocprofv3)
    __shared__ uint32_t part[4][4];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        c0 += __shfl_xor(c0, m, 64); c1 += __shfl_xor(c1, m, 64); c2 += __shfl_xor(c2, m, 64); c3 += __shfl_xor(c3, m, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        const int wv = threadIdx.x >> 6;
        part[wv][0] = c0; part[wv][1] = c1; part[wv][2] = c2; part[wv][3] = c3;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const uint32_t t = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (t) atomicAdd(&cnt[threadIdx.x], t);
    }
}

__global__ void __launch_bounds__(256)
sgr_densify_keep_flags_kernel(int n, const uint8_t* __restrict__ prune, uint32_t* __restrict__ keep) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) keep[i] = prune[i] ? 0u : 1u;
}
__global__ void __launch_bounds__(256)
sgr_densify_compact_kernel(int n, const uint8_t* __restrict__ prune, const uint32_t* __restrict__ off,
                           int32_t* __restrict__ sel) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && !prune[i]) sel[off[i]] = i;
}

__global__ void __launch_bounds__(256)
sgr_reset_opacity_kernel(int N, float* __restrict__ opacity, float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float op = 1.0f / (1.0f + expf(-opacity[i]));   // get_opacity
    const float x = fminf(op, 0.01f);                      // torch.min(opacity, 0.01)
    opacity[i] = logf(x / (1.0f - x));                     // inverse_sigmoid (general_utils.py:28-29)
    if (exp_avg) exp_avg[i] = 0.0f;
    if (exp_avg_sq) exp_avg_sq[i] = 0.0f;
}

extern "C" {

int sgr_densify_prune_mask(int n, const sgr_densify_params* p, int variant, const float* xyz, const float* scaling,
                           const float* rotation, const float* opacity, const float* sphere, const float* box,
                           const float* box_normals, uint8_t* prune, int64_t counts[4], void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!p || !counts) return sgr_set_error(SGR_E_INVALID, "p and counts are required");
    for (int k = 0; k < 4; k++) counts[k] = 0;
    if (n <= 0) return 0;
    if (!xyz || !scaling || !opacity || !prune) return sgr_set_error(SGR_E_INVALID, "xyz, scaling, opacity and prune are required");
    if (variant < SGR_PRUNE_BASE || variant > SGR_PRUNE_ACTOR) return sgr_set_error(SGR_E_INVALID, "unknown prune variant");
    DnSphere sph = {0.f, 0.f, 0.f, 0.f};
    DnBox bx = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    if (variant == SGR_PRUNE_BKGD) {
        if (!sphere) return sgr_set_error(SGR_E_INVALID, "the background rule needs sphere = {cx, cy, cz, radius}");
        sph = {sphere[0], sphere[1], sphere[2], sphere[3]};
    }
    if (variant == SGR_PRUNE_ACTOR && p->prune_big) {
        if (!box || !box_normals || !rotation)
            return sgr_set_error(SGR_E_INVALID, "the actor rule needs box, box_normals and rotation");
        for (int a = 0; a < 3; a++) { bx.lo[a] = box[a]; bx.hi[a] = box[3 + a]; }
    }
    // 16 bytes of counters in the process-wide per-device block (held until this call has read them back)
    SgrFlagBlock fb = sgr_acquire_flag_block();
    if (!fb.ptr) return sgr_set_error(SGR_E_HIP, "counter block allocation failed");
    uint32_t* cnt = fb.ptr + 16;  // words 16..19: the first words belong to sgr_visible_filter's flag
    DN_HIP(hipMemsetAsync(cnt, 0, 16, stream));
    sgr_densify_prune_kernel<<<std::min((n + 255) / 256, 2048), 256, 0, stream>>>(n, *p, variant, xyz, scaling, rotation, opacity, sph, bx,
                                                                 box_normals, prune, cnt);
    uint32_t h[4];
    DN_HIP(hipMemcpyAsync(h, cnt, 16, hipMemcpyDeviceToHost, stream));
    DN_HIP(hipStreamSynchronize(stream));
    for (int k = 0; k < 4; k++) counts[k] = h[k];
    return 0;
}

int sgr_densify_compact(int n, const uint8_t* prune, char* work, int32_t* sel, int64_t* n_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!n_out) return sgr_set_error(SGR_E_INVALID, "n_out is required");
    *n_out = 0;
    if (n <= 0) return 0;
    if (!prune || !work || !sel) return sgr_set_error(SGR_E_INVALID, "prune, work and sel are required");
    const DnWork w = dn_carve((char*)sgr_align_up((size_t)work, 256), (size_t)n);
    sgr_densify_keep_flags_kernel<<<(n + 255) / 256, 256, 0, stream>>>(n, prune, w.offA);
    sgr_launch_scan(w.offA, w.offA, (size_t)n, w.tmp, false, stream, w.totals + 0);
    sgr_densify_compact_kernel<<<(n + 255) / 256, 256, 0, stream>>>(n, prune, w.offA, sel);
    uint32_t t = 0;
    DN_HIP(hipMemcpyAsync(&t, w.totals, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    DN_HIP(hipStreamSynchronize(stream));
    *n_out = t;
    return 0;
}

int sgr_reset_opacity(int N, float* opacity, float* exp_avg, float* exp_avg_sq, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N <= 0) return 0;
    if (!opacity) return sgr_set_error(SGR_E_INVALID, "opacity is required");
    sgr_reset_opacity_kernel<<<(N + 255) / 256, 256, 0, stream>>>(N, opacity, exp_avg, exp_avg_sq);
    DN_HIP(hipGetLastError());
    return 0;
}

size_t sgr_densify_work_bytes(int N) {
    char* base = (char*)4096;
    DnWork w = dn_carve(base, (size_t)(N > 0 ? N : 1));
    return (size_t)((char*)(w.totals + 16) - base) + 512;
}

int sgr_densify_plan(int N, const sgr_densify_params* p, const float* xyz_gradient_accum, const float* denom,
                     const float* scaling, const float* opacity, char* work, int64_t counts[6], void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!p || !counts) return sgr_set_error(SGR_E_INVALID, "p and counts are required");
    for (int k = 0; k < 6; k++) counts[k] = 0;
    if (N <= 0) return 0;
    if (!xyz_gradient_accum || !denom || !scaling || !opacity || !work)
        return sgr_set_error(SGR_E_INVALID, "statistics, scaling, opacity and work are required");
    if (p->n_split < 1 || p->grad_column < 0 || p->grad_column > 1) return sgr_set_error(SGR_E_INVALID, "n_split >= 1, grad_column in {0,1}");
    const DnWork w = dn_carve((char*)sgr_align_up((size_t)work, 256), (size_t)N);
    sgr_densify_flags_kernel<<<(N + 255) / 256, 256, 0, stream>>>(N, *p, xyz_gradient_accum, denom, scaling, opacity, w);
    sgr_launch_scan(w.offA, w.offA, (size_t)N, w.tmp, false, stream, w.totals + 0);
    sgr_launch_scan(w.offB, w.offB, (size_t)N, w.tmp, false, stream, w.totals + 1);
    sgr_launch_scan(w.offS, w.offS, (size_t)N, w.tmp, false, stream, w.totals + 2);
    sgr_launch_scan(w.offC, w.offC, (size_t)N, w.tmp, false, stream, w.totals + 3);
    DN_HIP(hipMemsetAsync(w.totals + 4, 0, sizeof(uint32_t), stream));
    sgr_densify_count_kernel<<<std::min((N + 255) / 256, 1024), 256, 0, stream>>>(N, w);
    uint32_t t[8];
    DN_HIP(hipMemcpyAsync(t, w.totals, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    DN_HIP(hipStreamSynchronize(stream));
    const int64_t nA = t[0], nB = t[1], nS = t[2], nC = t[3], nClone = t[4];
    const int64_t n_out = nA + nB + (int64_t)p->n_split * nC;
    const int64_t candidates = ((int64_t)N - nS) + nClone + (int64_t)p->n_split * nS;  // the set prune_mask is evaluated on
    counts[0] = N; counts[1] = nClone; counts[2] = nS; counts[3] = candidates - n_out; counts[4] = n_out;
    counts[5] = (int64_t)p->n_split * nS;
    if (n_out > 0x7fffffff) return sgr_set_error(SGR_E_INVALID, "more than 2^31 points after densification");
    return 0;
}

int sgr_densify_map(int N, const sgr_densify_params* p, const char* work, int32_t* src, uint8_t* kind, int32_t* sample_row,
                    void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N <= 0) return 0;
    if (!p || !work || !src || !kind || !sample_row) return sgr_set_error(SGR_E_INVALID, "p, work, src, kind and sample_row are required");
    const DnWork w = dn_carve((char*)sgr_align_up((size_t)work, 256), (size_t)N);
    sgr_densify_map_kernel<<<(N + 255) / 256, 256, 0, stream>>>(N, p->n_split, w, src, kind, sample_row);
    DN_HIP(hipGetLastError());
    return 0;
}

int sgr_densify_gather(int n_out, int width, const float* in, const int32_t* src, const uint8_t* kind, int zero_new,
                       float* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_out <= 0 || width <= 0) return 0;
    if (!in || !src || !kind || !out) return sgr_set_error(SGR_E_INVALID, "in, src, kind and out are required");
    const unsigned nb = (unsigned)((n_out + SGR_DN_ROWS - 1) / SGR_DN_ROWS);
    const bool al16 = (((uintptr_t)in | (uintptr_t)out) & 15u) == 0;
#define DN_G(W) sgr_densify_gather_kernel<W><<<nb, 256, 0, stream>>>(n_out, width, in, src, kind, zero_new, out)
    // the widths of the reference's parameter groups: xyz / scaling / f_dc 3, opacity 1, rotation 4, f_rest 45 (SH degree 3;
    // 9 / 24 at degrees 1 / 2), semantic logits (16, 19, 20 classes), actor f_dc (fourier_dim x 3: 6, 9, 12, 15)
    switch (width) {
        case 1: DN_G(1); break;
        case 3: DN_G(3); break;
        case 4: if (al16) DN_G(4); else DN_G(0); break;
        case 6: DN_G(6); break;
        case 9: DN_G(9); break;
        case 12: if (al16) DN_G(12); else DN_G(0); break;
        case 15: DN_G(15); break;
        case 16: if (al16) DN_G(16); else DN_G(0); break;
        case 19: DN_G(19); break;
        case 20: if (al16) DN_G(20); else DN_G(0); break;
        case 24: if (al16) DN_G(24); else DN_G(0); break;
        case 45: DN_G(45); break;
        case 48: if (al16) DN_G(48); else DN_G(0); break;
        default: DN_G(0); break;
    }
#undef DN_G
    DN_HIP(hipGetLastError());
    return 0;
}

int sgr_densify_split_children(int n_out, int n_split, const int32_t* src, const uint8_t* kind, const int32_t* sample_row,
                               const float* xyz_in, const float* scaling_in, const float* rotation_in,
                               const float* normals, float* xyz_out, float* scaling_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_out <= 0) return 0;
    if (!src || !kind || !sample_row || !xyz_in || !scaling_in || !rotation_in || !xyz_out || !scaling_out)
        return sgr_set_error(SGR_E_INVALID, "all arrays are required");
    sgr_densify_children_kernel<<<(n_out + 255) / 256, 256, 0, stream>>>(n_out, n_split, src, kind, sample_row, xyz_in,
                                                                        scaling_in, rotation_in, normals, xyz_out,
                                                                        scaling_out);
    DN_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
