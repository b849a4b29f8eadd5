// sgr_scene.hip -- scene-graph rows next to the rasterizer (include/sgr_scene.h).
//
// n1  compose: background + posed actors -> the rasterizer's flat inputs, forward and backward.  The reference does
//     this with ~40 torch ops per iteration (per-attribute cat, clone, masked flips, einsum, quaternion products,
//     normalize; street_gaussian_model.py:287-449) that each stream the whole attribute through HBM.  Here every
//     parameter is read once and every output written once:
//       * one workgroup per CHUNK (<= 256 consecutive Gaussians of one model, table built on the host), one lane per
//         Gaussian for the small attributes;
//       * SH rows move with one lane per FLOAT (coalesced on both sides; a lane-per-Gaussian copy of 45-float rows is
//         issue-bound on CDNA4: 64 cache lines per load instruction);
//       * the pose gradient of an actor (rotation 4 + translation 3) is a reduction over its Gaussians: per-chunk
//         partial sums in a fixed order, then one small kernel per actor sums its chunks in order -> deterministic.
// n2  densification statistics scattered back per model in one pass (street_gaussian_model.py:551-571).
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sgr_scene.h"
#include "sgr_common.h"

int sgr_set_error(int code, const std::string& msg);

#define SGR_SC_THREADS 256
#define SGR_SC_NPART 16  // per-chunk partials of an actor: G[9] (sum dx' x^T), T[3] (sum dx'), Q[4] (rotation path)

struct SgrChunk {
    int32_t seg, start, n, out;  // segment, first Gaussian inside it, Gaussians, first output row
};
struct SgrSegDev {
    sgr_scene_segment s;
    sgr_scene_segment_grads g;
    int32_t first_chunk, nchunks;
};

// ---- quaternion helpers (real part first; general_utils.py:220-238) ----
struct Q4 { float w, x, y, z; };
__device__ __forceinline__ Q4 qmul(const Q4 a, const Q4 b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Q4 qconj(const Q4 a) { return {a.w, -a.x, -a.y, -a.z}; }
__device__ __forceinline__ float qdot(const Q4 a, const Q4 b) { return a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z; }
// torch.nn.functional.normalize: x / max(|x|, 1e-12)
__device__ __forceinline__ Q4 qnormalize(const Q4 a, float& n) {
    n = fmaxf(sqrtf(qdot(a, a)), 1e-12f);
    return {a.w / n, a.x / n, a.y / n, a.z / n};
}
// y = x / max(|x|, eps): dx = (dy - y (y . dy)) / n   (for |x| > eps; the clamped branch is dy / eps)
__device__ __forceinline__ Q4 qnormalize_bwd(const Q4 y, float n, const Q4 dy, bool clamped) {
    if (clamped) return {dy.w / n, dy.x / n, dy.y / n, dy.z / n};
    const float d = qdot(y, dy);
    return {(dy.w - y.w * d) / n, (dy.x - y.x * d) / n, (dy.y - y.y * d) / n, (dy.z - y.z * d) / n};
}
// quaternion_to_matrix (general_utils.py:125-146): normalises by |q| (no eps), row-major R
__device__ __forceinline__ void qtomat(const Q4 q, float (&R)[9], Q4& qn, float& norm) {
    norm = sqrtf(qdot(q, q));
    qn = {q.w / norm, q.x / norm, q.y / norm, q.z / norm};
    const float r = qn.w, x = qn.x, y = qn.y, z = qn.z;
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z); R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y); R[7] = 2.f * (y * z + r * x); R[8] = 1.f - 2.f * (x * x + y * y);
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- forward: small attributes, one lane per Gaussian ---------------------------------------------------------
__global__ void __launch_bounds__(SGR_SC_THREADS)
sgr_scene_fwd_kernel(const SgrChunk* __restrict__ chunks, const SgrSegDev* __restrict__ segs, int S,
                     float* __restrict__ means3D, float* __restrict__ rotations, float* __restrict__ scales,
                     float* __restrict__ opacities, float* __restrict__ semantics) {
    const SgrChunk c = chunks[blockIdx.x];
    const sgr_scene_segment& sg = segs[c.seg].s;
    if ((int)threadIdx.x >= c.n) return;
    const size_t i = (size_t)c.start + threadIdx.x, o = (size_t)c.out + threadIdx.x;
    float x[3] = {sg.xyz[3 * i], sg.xyz[3 * i + 1], sg.xyz[3 * i + 2]};
    const float4 qr = *reinterpret_cast<const float4*>(sg.rotation + 4 * i);
    float n;
    Q4 q = qnormalize({qr.x, qr.y, qr.z, qr.w}, n);  // gaussian_model.py:229-230
    if (sg.kind == SGR_SEG_ACTOR) {
        if (sg.flip_mask && sg.flip_mask[i]) {  // street_gaussian_model.py:324-327, 354-356
            x[sg.flip_axis] = -x[sg.flip_axis];
            q = qmul({sg.flip_quat[0], sg.flip_quat[1], sg.flip_quat[2], sg.flip_quat[3]}, q);
        }
        const Q4 po = {sg.pose[0], sg.pose[1], sg.pose[2], sg.pose[3]};
        float R[9], pn;
        Q4 pq;
        qtomat(po, R, pq, pn);  // :357
        const float y0 = R[0] * x[0] + R[1] * x[1] + R[2] * x[2] + sg.pose[4];  // :358
        const float y1 = R[3] * x[0] + R[4] * x[1] + R[5] * x[2] + sg.pose[5];
        const float y2 = R[6] * x[0] + R[7] * x[1] + R[8] * x[2] + sg.pose[6];
        x[0] = y0; x[1] = y1; x[2] = y2;
        float n2;
        q = qnormalize(qmul(po, q), n2);  // :328-329
    }
    means3D[3 * o] = x[0]; means3D[3 * o + 1] = x[1]; means3D[3 * o + 2] = x[2];
    *reinterpret_cast<float4*>(rotations + 4 * o) = make_float4(q.w, q.x, q.y, q.z);
#pragma unroll
    for (int k = 0; k < 3; k++) scales[3 * o + k] = expf(sg.scaling[3 * i + k]);  // gaussian_model.py:225-226
    opacities[o] = sigmoidf_(sg.opacity[i]);                                        // :250-251
    if (S > 0) {
        float* out = semantics + o * (size_t)S;
        if (sg.semantic == nullptr) {
            for (int k = 0; k < S; k++) out[k] = 0.f;
        } else if (sg.kind == SGR_SEG_ACTOR) {  // gaussian_model_actor.py:62-69
            const float v = sg.semantic[i];
            for (int k = 0; k < S; k++) out[k] = 0.f;
            if (sg.class_label >= 0 && sg.class_label < S)
                out[sg.class_label] = sg.sem_mode == SGR_SEM_PROBABILITIES ? sigmoidf_(v) : v;
        } else if (sg.sem_mode == SGR_SEM_PROBABILITIES) {  // gaussian_model.py:247-248: softmax over the classes
            const float* in = sg.semantic + i * (size_t)S;
            float m = in[0];
            for (int k = 1; k < S; k++) m = fmaxf(m, in[k]);
            float sum = 0.f;
            for (int k = 0; k < S; k++) sum += expf(in[k] - m);
            for (int k = 0; k < S; k++) out[k] = expf(in[k] - m) / sum;
        }  // static logits: copied by the SH kernel, one lane per float
    }
}

// ---- forward: SH rows, one lane per float (get_features :366-381, get_features_fourier actor.py:71-80) ----------
// MC = M when it is known at compile time (16: divisions by the row length become multiplies), 0 = runtime M.
// Static models with logit semantics (a plain copy, gaussian_model.py:243-246) are also moved here, one lane per float.
template <int MC>
__global__ void __launch_bounds__(SGR_SC_THREADS)
sgr_scene_sh_fwd_kernel(const SgrChunk* __restrict__ chunks, const SgrSegDev* __restrict__ segs, int M_, int S,
                        float* __restrict__ shs, float* __restrict__ semantics) {
    const SgrChunk c = chunks[blockIdx.x];
    const sgr_scene_segment& sg = segs[c.seg].s;
    const int M = MC ? MC : M_;
    const int row = 3 * M, rest = 3 * (M - 1), C = sg.fourier_dim;
    if (S > 0 && sg.semantic && sg.kind == SGR_SEG_STATIC && sg.sem_mode == SGR_SEM_LOGITS) {
        const float* in = sg.semantic + (size_t)c.start * S;
        float* so = semantics + (size_t)c.out * S;
        for (int e = threadIdx.x; e < c.n * S; e += SGR_SC_THREADS) so[e] = in[e];
    }
    float* out = shs + (size_t)c.out * row;
    for (int e = threadIdx.x; e < c.n * row; e += SGR_SC_THREADS) {
        const int g = e / row, j = e - g * row;
        const size_t i = (size_t)c.start + g;
        float v;
        if (j < 3) {
            const float* dc = sg.features_dc + i * (size_t)C * 3 + j;
            if (sg.kind == SGR_SEG_ACTOR && sg.idft) {
                v = 0.f;
                for (int k = 0; k < C; k++) v += dc[3 * k] * sg.idft[k];
            } else {
                v = dc[0];
            }
        } else {
            v = sg.features_rest[i * (size_t)rest + (j - 3)];
        }
        out[e] = v;
    }
}

// ---- backward: small attributes + per-chunk pose partials ------------------------------------------------------
__global__ void __launch_bounds__(SGR_SC_THREADS)
sgr_scene_bwd_kernel(const SgrChunk* __restrict__ chunks, const SgrSegDev* __restrict__ segs, int S,
                     const float* __restrict__ dmeans, const float* __restrict__ drot, const float* __restrict__ dscale,
                     const float* __restrict__ dopac, const float* __restrict__ dsem, float* __restrict__ partials) {
    __shared__ float red[SGR_SC_THREADS / 64][SGR_SC_NPART];
    const SgrChunk c = chunks[blockIdx.x];
    const SgrSegDev& sd = segs[c.seg];
    const sgr_scene_segment& sg = sd.s;
    const sgr_scene_segment_grads& gr = sd.g;
    const bool live = (int)threadIdx.x < c.n;
    const size_t i = (size_t)c.start + (live ? threadIdx.x : 0), o = (size_t)c.out + (live ? threadIdx.x : 0);
    float part[SGR_SC_NPART];
#pragma unroll
    for (int k = 0; k < SGR_SC_NPART; k++) part[k] = 0.f;
    if (live) {
        float dx[3] = {dmeans ? dmeans[3 * o] : 0.f, dmeans ? dmeans[3 * o + 1] : 0.f, dmeans ? dmeans[3 * o + 2] : 0.f};
        Q4 dq = {0.f, 0.f, 0.f, 0.f};
        if (drot) {
            const float4 t = *reinterpret_cast<const float4*>(drot + 4 * o);
            dq = {t.x, t.y, t.z, t.w};
        }
        const float4 qr = *reinterpret_cast<const float4*>(sg.rotation + 4 * i);
        const Q4 qraw = {qr.x, qr.y, qr.z, qr.w};
        float n;
        const Q4 ql0 = qnormalize(qraw, n);
        const bool clamped = sqrtf(qdot(qraw, qraw)) < 1e-12f;
        if (sg.kind == SGR_SEG_ACTOR) {
            const bool flip = sg.flip_mask && sg.flip_mask[i];
            const Q4 fq = {sg.flip_quat[0], sg.flip_quat[1], sg.flip_quat[2], sg.flip_quat[3]};
            float x[3] = {sg.xyz[3 * i], sg.xyz[3 * i + 1], sg.xyz[3 * i + 2]};
            if (flip) x[sg.flip_axis] = -x[sg.flip_axis];
            const Q4 ql = flip ? qmul(fq, ql0) : ql0;
            const Q4 po = {sg.pose[0], sg.pose[1], sg.pose[2], sg.pose[3]};
            float R[9], pn;
            Q4 pq;
            qtomat(po, R, pq, pn);
            // x' = R x + t
#pragma unroll
            for (int a = 0; a < 3; a++) {
#pragma unroll
                for (int b = 0; b < 3; b++) part[3 * a + b] = dx[a] * x[b];
                part[9 + a] = dx[a];
            }
            float dl[3] = {R[0] * dx[0] + R[3] * dx[1] + R[6] * dx[2], R[1] * dx[0] + R[4] * dx[1] + R[7] * dx[2],
                           R[2] * dx[0] + R[5] * dx[1] + R[8] * dx[2]};
            if (flip) dl[sg.flip_axis] = -dl[sg.flip_axis];
            dx[0] = dl[0]; dx[1] = dl[1]; dx[2] = dl[2];
            // r = normalize(po (x) ql)
            float n2;
            const Q4 u = qmul(po, ql);
            const Q4 r = qnormalize(u, n2);
            const Q4 du = qnormalize_bwd(r, n2, dq, sqrtf(qdot(u, u)) < 1e-12f);
            const Q4 dpo = qmul(du, qconj(ql));   // d(a b)/da . dO = dO b*
            part[12] = dpo.w; part[13] = dpo.x; part[14] = dpo.y; part[15] = dpo.z;
            Q4 dql = qmul(qconj(po), du);          // d(a b)/db . dO = a* dO
            if (flip) dql = qmul(qconj(fq), dql);
            dq = dql;
        }
        if (gr.xyz) { gr.xyz[3 * i] = dx[0]; gr.xyz[3 * i + 1] = dx[1]; gr.xyz[3 * i + 2] = dx[2]; }
        if (gr.rotation) {
            const Q4 d = qnormalize_bwd(ql0, n, dq, clamped);
            *reinterpret_cast<float4*>(gr.rotation + 4 * i) = make_float4(d.w, d.x, d.y, d.z);
        }
        if (gr.scaling) {
#pragma unroll
            for (int k = 0; k < 3; k++) gr.scaling[3 * i + k] = (dscale ? dscale[3 * o + k] : 0.f) * expf(sg.scaling[3 * i + k]);
        }
        if (gr.opacity) {
            const float sgm = sigmoidf_(sg.opacity[i]);
            gr.opacity[i] = (dopac ? dopac[o] : 0.f) * sgm * (1.f - sgm);
        }
        if (gr.semantic && sg.semantic) {
            if (S <= 0 || !dsem) {
                const int cols = sg.kind == SGR_SEG_ACTOR ? 1 : S;
                for (int k = 0; k < cols; k++) gr.semantic[i * (size_t)cols + k] = 0.f;
            } else if (sg.kind == SGR_SEG_ACTOR) {
                float d = (sg.class_label >= 0 && sg.class_label < S) ? dsem[o * (size_t)S + sg.class_label] : 0.f;
                if (sg.sem_mode == SGR_SEM_PROBABILITIES) {
                    const float sgm = sigmoidf_(sg.semantic[i]);
                    d *= sgm * (1.f - sgm);
                }
                gr.semantic[i] = d;
            } else if (sg.sem_mode == SGR_SEM_PROBABILITIES) {
                const float* in = sg.semantic + i * (size_t)S;
                const float* d = dsem + o * (size_t)S;
                float m = in[0];
                for (int k = 1; k < S; k++) m = fmaxf(m, in[k]);
                float sum = 0.f, pd = 0.f;
                for (int k = 0; k < S; k++) sum += expf(in[k] - m);
                for (int k = 0; k < S; k++) pd += expf(in[k] - m) / sum * d[k];
                for (int k = 0; k < S; k++) gr.semantic[i * (size_t)S + k] = expf(in[k] - m) / sum * (d[k] - pd);
            }  // static logits: copied by the SH kernel, one lane per float
        }
    }
    if (sg.kind != SGR_SEG_ACTOR || partials == nullptr) return;  // uniform per workgroup
    // chunk-wide sums in a fixed order: wave butterfly, then waves 0..3 in order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < SGR_SC_NPART; k++) {
        float v = part[k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < SGR_SC_NPART) {
        const int k = threadIdx.x;
        partials[(size_t)blockIdx.x * SGR_SC_NPART + k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    }
}

// one workgroup (one wave) per actor: sum its chunks in order, then chain into the 7 pose parameters
__global__ void __launch_bounds__(64)
sgr_scene_pose_bwd_kernel(const SgrSegDev* __restrict__ segs, const float* __restrict__ partials) {
    const SgrSegDev& sd = segs[blockIdx.x];
    if (sd.s.kind != SGR_SEG_ACTOR || sd.g.pose == nullptr) return;
    __shared__ float tot[SGR_SC_NPART];
    if (threadIdx.x < SGR_SC_NPART) {
        float a = 0.f;
        for (int c = 0; c < sd.nchunks; c++) a += partials[(size_t)(sd.first_chunk + c) * SGR_SC_NPART + threadIdx.x];
        tot[threadIdx.x] = a;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const float* G = tot;  // G[3a+b] = sum dx'_a x_b
    const Q4 po = {sd.s.pose[0], sd.s.pose[1], sd.s.pose[2], sd.s.pose[3]};
    float R[9], pn;
    Q4 q;
    qtomat(po, R, q, pn);
    const float r = q.w, x = q.x, y = q.y, z = q.z;
    // dL/d(normalised quaternion) through R(q) (general_utils.py:137-145)
    Q4 dqn;
    dqn.w = 2.f * (-z * G[1] + y * G[2] + z * G[3] - x * G[5] - y * G[6] + x * G[7]);
    dqn.x = 2.f * (y * G[1] + z * G[2] + y * G[3] - 2.f * x * G[4] - r * G[5] + z * G[6] + r * G[7] - 2.f * x * G[8]);
    dqn.y = 2.f * (-2.f * y * G[0] + x * G[1] + r * G[2] + x * G[3] + z * G[5] - r * G[6] + z * G[7] - 2.f * y * G[8]);
    dqn.z = 2.f * (-2.f * z * G[0] - r * G[1] + x * G[2] + r * G[3] - 2.f * z * G[4] + y * G[5] + x * G[6] + y * G[7]);
    const Q4 dpo = qnormalize_bwd(q, pn, dqn, false);
    sd.g.pose[0] = dpo.w + tot[12];
    sd.g.pose[1] = dpo.x + tot[13];
    sd.g.pose[2] = dpo.y + tot[14];
    sd.g.pose[3] = dpo.z + tot[15];
    sd.g.pose[4] = tot[9];
    sd.g.pose[5] = tot[10];
    sd.g.pose[6] = tot[11];
}

// ---- backward: SH rows, one lane per float of each gradient array ----------------------------------------------
template <int MC>
__global__ void __launch_bounds__(SGR_SC_THREADS)
sgr_scene_sh_bwd_kernel(const SgrChunk* __restrict__ chunks, const SgrSegDev* __restrict__ segs, int M_, int S,
                        const float* __restrict__ dshs, const float* __restrict__ dsem) {
    const SgrChunk c = chunks[blockIdx.x];
    const SgrSegDev& sd = segs[c.seg];
    const int M = MC ? MC : M_;
    const int row = 3 * M, rest = 3 * (M - 1), C = sd.s.fourier_dim;
    if (S > 0 && dsem && sd.g.semantic && sd.s.semantic && sd.s.kind == SGR_SEG_STATIC && sd.s.sem_mode == SGR_SEM_LOGITS) {
        const float* in = dsem + (size_t)c.out * S;
        float* so = sd.g.semantic + (size_t)c.start * S;
        for (int e = threadIdx.x; e < c.n * S; e += SGR_SC_THREADS) so[e] = in[e];
    }
    const float* in = dshs + (size_t)c.out * row;
    if (sd.g.features_rest) {
        float* out = sd.g.features_rest + (size_t)c.start * rest;
        for (int e = threadIdx.x; e < c.n * rest; e += SGR_SC_THREADS) {
            const int g = e / rest, j = e - g * rest;
            out[e] = dshs ? in[g * row + 3 + j] : 0.f;
        }
    }
    if (sd.g.features_dc) {
        float* out = sd.g.features_dc + (size_t)c.start * C * 3;
        const bool four = sd.s.kind == SGR_SEG_ACTOR && sd.s.idft;
        for (int e = threadIdx.x; e < c.n * C * 3; e += SGR_SC_THREADS) {
            const int g = e / (C * 3), j = e - g * (C * 3), k = j / 3, ch = j - 3 * k;
            const float d = dshs ? in[g * row + ch] : 0.f;
            out[e] = four ? d * sd.s.idft[k] : (k == 0 ? d : 0.f);
        }
    }
}

// ---- n2: densification statistics ---------------------------------------------------------------------------
struct SgrStatSegDev { sgr_scene_stats_segment s; };
__global__ void __launch_bounds__(SGR_SC_THREADS)
sgr_scene_stats_kernel(const SgrChunk* __restrict__ chunks, const SgrStatSegDev* __restrict__ segs,
                       const float* __restrict__ dmeans2D, const int* __restrict__ radii) {
    const SgrChunk c = chunks[blockIdx.x];
    if ((int)threadIdx.x >= c.n) return;
    const sgr_scene_stats_segment& sg = segs[c.seg].s;
    const size_t i = (size_t)c.start + threadIdx.x, o = (size_t)c.out + threadIdx.x;
    const int r = radii[o];
    if (r <= 0) return;  // visibility_filter = radii > 0 (street_gaussian_renderer.py)
    const float gx = dmeans2D[3 * o], gy = dmeans2D[3 * o + 1], gz = dmeans2D[3 * o + 2];
    if (sg.xyz_gradient_accum) {
        sg.xyz_gradient_accum[2 * i] += sqrtf(gx * gx + gy * gy);  // street_gaussian_model.py:567
        sg.xyz_gradient_accum[2 * i + 1] += fabsf(gz);              // :568 (norm of one component)
    }
    if (sg.denom) sg.denom[i] += 1.0f;                               // :569
    if (sg.max_radii2D) sg.max_radii2D[i] = fmaxf(sg.max_radii2D[i], (float)r);  // :551-560
}

// ---- host side ----------------------------------------------------------------------------------------------------
template <typename Seg>
static size_t build_chunks(int K, const Seg* segs, std::vector<SgrChunk>& chunks, std::vector<int>& first,
                           std::vector<int>& count) {
    size_t out = 0;
    first.assign(K, 0);
    count.assign(K, 0);
    for (int k = 0; k < K; k++) {
        first[k] = (int)chunks.size();
        for (int s = 0; s < segs[k].count; s += SGR_SC_THREADS)
            chunks.push_back({k, s, std::min(SGR_SC_THREADS, segs[k].count - s), (int32_t)(out + s)});
        count[k] = (int)chunks.size() - first[k];
        out += (size_t)segs[k].count;
    }
    return out;
}

static int check_segments(int K, const sgr_scene_segment* segs, int M, int S) {
    if (K < 0 || (K > 0 && !segs)) return sgr_set_error(SGR_E_INVALID, "segs is required");
    if (M < 1 || S < 0) return sgr_set_error(SGR_E_INVALID, "need M >= 1 and S >= 0");
    for (int k = 0; k < K; k++) {
        const sgr_scene_segment& s = segs[k];
        if (s.count < 0 || s.fourier_dim < 1) return sgr_set_error(SGR_E_INVALID, "segment: count >= 0 and fourier_dim >= 1");
        if (s.count == 0) continue;
        if (!s.xyz || !s.rotation || !s.scaling || !s.opacity || !s.features_dc || (M > 1 && !s.features_rest))
            return sgr_set_error(SGR_E_INVALID, "segment: xyz, rotation, scaling, opacity, features_dc, features_rest are required");
        if (s.kind == SGR_SEG_ACTOR && !s.pose) return sgr_set_error(SGR_E_INVALID, "actor segment without a pose");
        if (s.kind == SGR_SEG_ACTOR && s.fourier_dim > 1 && !s.idft)
            return sgr_set_error(SGR_E_INVALID, "actor segment with fourier_dim > 1 needs idft");
        if (s.kind != SGR_SEG_ACTOR && s.fourier_dim != 1) return sgr_set_error(SGR_E_INVALID, "static segments have fourier_dim 1");
        if (s.flip_axis < 0 || s.flip_axis > 2) return sgr_set_error(SGR_E_INVALID, "flip_axis must be 0, 1 or 2");
    }
    return 0;
}

#define SC_HIP(call)                                                                                       \
    do {                                                                                                   \
        hipError_t e__ = (call);                                                                           \
        if (e__ != hipSuccess) return sgr_set_error(SGR_E_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

// Host tables travel through a per-thread, grow-only PINNED staging buffer so that the copy is really asynchronous
// (a pageable source makes hipMemcpyAsync stage or block); an event guards the buffer against reuse while the previous
// call's copy is still in flight.
struct SgrPinnedStage {
    char* p = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;
};
static int stage_get(size_t bytes, SgrPinnedStage** out) {
    static thread_local SgrPinnedStage stages[64];  // the guarding event belongs to a device: one stage per (thread, device)
    int dev = 0;
    SC_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return sgr_set_error(SGR_E_INVALID, "device index out of range");
    SgrPinnedStage& st = stages[dev];
    if (st.pending) {
        SC_HIP(hipEventSynchronize(st.ev));
        st.pending = false;
    }
    if (bytes > st.cap) {
        if (st.p) (void)hipHostFree(st.p);
        st.p = nullptr;
        st.cap = 0;
        const size_t want = sgr_align_up(bytes + bytes / 2 + 4096, 4096);
        SC_HIP(hipHostMalloc((void**)&st.p, want, hipHostMallocPortable));
        st.cap = want;
    }
    if (!st.ev) SC_HIP(hipEventCreateWithFlags(&st.ev, hipEventDisableTiming));
    *out = &st;
    return 0;
}

// uploads [segments | chunks] (+ room for partials) into scratch; returns device pointers
template <typename SegDev>
static int upload(const std::vector<SegDev>& sd, const std::vector<SgrChunk>& chunks, size_t extra_floats,
                  sgr_alloc_fn scratch, void* scratch_user, hipStream_t stream, SegDev** dsegs, SgrChunk** dchunks,
                  float** dextra) {
    const size_t b0 = sgr_align_up(sd.size() * sizeof(SegDev), 256), b1 = sgr_align_up(chunks.size() * sizeof(SgrChunk), 256);
    char* base = scratch(b0 + b1 + extra_floats * sizeof(float) + 512, scratch_user);
    if (!base) return sgr_set_error(SGR_E_ALLOC, "scene scratch allocation failed");
    base = (char*)sgr_align_up((size_t)base, 256);
    SgrPinnedStage* st;
    int rc = stage_get(b0 + b1, &st);
    if (rc) return rc;
    memcpy(st->p, sd.data(), sd.size() * sizeof(SegDev));
    memcpy(st->p + b0, chunks.data(), chunks.size() * sizeof(SgrChunk));
    SC_HIP(hipMemcpyAsync(base, st->p, b0 + b1, hipMemcpyHostToDevice, stream));
    SC_HIP(hipEventRecord(st->ev, stream));
    st->pending = true;
    *dsegs = (SegDev*)base;
    *dchunks = (SgrChunk*)(base + b0);
    if (dextra) *dextra = (float*)(base + b0 + b1);
    return 0;
}

extern "C" {

int sgr_scene_compose_forward(int K, const sgr_scene_segment* segs, int M, int S, float* means3D, float* rotations,
                              float* scales, float* opacities, float* shs, float* semantics, sgr_alloc_fn scratch,
                              void* scratch_user, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = check_segments(K, segs, M, S);
    if (rc) return rc;
    if (!scratch) return sgr_set_error(SGR_E_INVALID, "scratch callback is required");
    std::vector<SgrChunk> chunks;
    std::vector<int> first, count;
    const size_t N = build_chunks(K, segs, chunks, first, count);
    if (N == 0) return 0;
    if (!means3D || !rotations || !scales || !opacities || !shs || (S > 0 && !semantics))
        return sgr_set_error(SGR_E_INVALID, "all output arrays are required");
    std::vector<SgrSegDev> sd(K);
    for (int k = 0; k < K; k++) { sd[k].s = segs[k]; sd[k].g = sgr_scene_segment_grads{}; sd[k].first_chunk = first[k]; sd[k].nchunks = count[k]; }
    SgrSegDev* dsegs; SgrChunk* dchunks;
    if ((rc = upload(sd, chunks, 0, scratch, scratch_user, stream, &dsegs, &dchunks, nullptr))) return rc;
    const unsigned nb = (unsigned)chunks.size();
    sgr_scene_fwd_kernel<<<nb, SGR_SC_THREADS, 0, stream>>>(dchunks, dsegs, S, means3D, rotations, scales, opacities, semantics);
    if (M == 16) sgr_scene_sh_fwd_kernel<16><<<nb, SGR_SC_THREADS, 0, stream>>>(dchunks, dsegs, M, S, shs, semantics);
    else sgr_scene_sh_fwd_kernel<0><<<nb, SGR_SC_THREADS, 0, stream>>>(dchunks, dsegs, M, S, shs, semantics);
    SC_HIP(hipGetLastError());
    return 0;
}

int sgr_scene_compose_backward(int K, const sgr_scene_segment* segs, const sgr_scene_segment_grads* grads, int M, int S,
                               const float* dL_dmeans3D, const float* dL_drotations, const float* dL_dscales,
                               const float* dL_dopacities, const float* dL_dshs, const float* dL_dsemantics,
                               sgr_alloc_fn scratch, void* scratch_user, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = check_segments(K, segs, M, S);
    if (rc) return rc;
    if (!grads || !scratch) return sgr_set_error(SGR_E_INVALID, "grads and scratch are required");
    std::vector<SgrChunk> chunks;
    std::vector<int> first, count;
    const size_t N = build_chunks(K, segs, chunks, first, count);
    if (N == 0) return 0;
    std::vector<SgrSegDev> sd(K);
    for (int k = 0; k < K; k++) { sd[k].s = segs[k]; sd[k].g = grads[k]; sd[k].first_chunk = first[k]; sd[k].nchunks = count[k]; }
    SgrSegDev* dsegs; SgrChunk* dchunks; float* partials;
    if ((rc = upload(sd, chunks, chunks.size() * SGR_SC_NPART, scratch, scratch_user, stream, &dsegs, &dchunks, &partials)))
        return rc;
    const unsigned nb = (unsigned)chunks.size();
    sgr_scene_bwd_kernel<<<nb, SGR_SC_THREADS, 0, stream>>>(dchunks, dsegs, S, dL_dmeans3D, dL_drotations, dL_dscales,
                                                            dL_dopacities, dL_dsemantics, partials);
    sgr_scene_pose_bwd_kernel<<<(unsigned)K, 64, 0, stream>>>(dsegs, partials);
    if (M == 16) sgr_scene_sh_bwd_kernel<16><<<nb, SGR_SC_THREADS, 0, stream>>>(dchunks, dsegs, M, S, dL_dshs, dL_dsemantics);
    else sgr_scene_sh_bwd_kernel<0><<<nb, SGR_SC_THREADS, 0, stream>>>(dchunks, dsegs, M, S, dL_dshs, dL_dsemantics);
    SC_HIP(hipGetLastError());
    return 0;
}

int sgr_scene_densification_stats(int K, const sgr_scene_stats_segment* segs, const float* dL_dmeans2D, const int* radii,
                                  sgr_alloc_fn scratch, void* scratch_user, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (K < 0 || (K > 0 && !segs) || !scratch) return sgr_set_error(SGR_E_INVALID, "segs and scratch are required");
    for (int k = 0; k < K; k++)
        if (segs[k].count < 0) return sgr_set_error(SGR_E_INVALID, "segment count must be >= 0");
    std::vector<SgrChunk> chunks;
    std::vector<int> first, count;
    const size_t N = build_chunks(K, segs, chunks, first, count);
    if (N == 0) return 0;
    if (!dL_dmeans2D || !radii) return sgr_set_error(SGR_E_INVALID, "dL_dmeans2D and radii are required");
    std::vector<SgrStatSegDev> sd(K);
    for (int k = 0; k < K; k++) sd[k].s = segs[k];
    SgrStatSegDev* dsegs; SgrChunk* dchunks;
    int rc = upload(sd, chunks, 0, scratch, scratch_user, stream, &dsegs, &dchunks, nullptr);
    if (rc) return rc;
    sgr_scene_stats_kernel<<<(unsigned)chunks.size(), SGR_SC_THREADS, 0, stream>>>(dchunks, dsegs, dL_dmeans2D, radii);
    SC_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
