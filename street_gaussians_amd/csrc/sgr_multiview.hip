// sgr_multiview.hip -- the exchange step of view-sharded training without moving the SH gradient.
//
// With one camera view per GPU the per-Gaussian gradients are summed over the ranks every step; 48 of the 59 floats
// per Gaussian are dL/dSH.  But the SH gradient of ONE view is rank-1 (backward.cu:46-105):
//     dL/dSH_v[k][c] = Y_k(dir_v) * dRGB_v[c],   dir_v = normalize(mean - campos_v),
// with dRGB_v = dL/dcolour of that view, zeroed where the forward clamped the channel (forward.cu:64-70,
// backward.cu:40-44).  Every rank knows all means and all camera centres, so it is enough to ALL-GATHER the 3 floats of
// dRGB_v per Gaussian and rebuild  sum_v Y(dir_v) (x) dRGB_v  locally, in view order (deterministic, identical on
// every rank): 12 B per Gaussian and view on the wire instead of a 192 B all-reduce.
#include "sgr_math.h"

#define SGR_MV_THREADS 256

// dRGB[g][c] = clamped(g, c) ? 0 : dL_dcolor[g][c]
__global__ void __launch_bounds__(SGR_MV_THREADS)
sgr_masked_color_grad_kernel(int P, const uint32_t* __restrict__ clamped, const float* __restrict__ dL_dcolor,
                             float* __restrict__ out) {
    const int i = blockIdx.x * SGR_MV_THREADS + threadIdx.x;  // one float per lane: coalesced
    if (i >= 3 * P) return;
    const int g = i / 3, c = i - 3 * g;
    out[i] = ((clamped[g] >> c) & 1u) ? 0.0f : dL_dcolor[i];
}

// dL_dsh[g][k][c] = sum_v Y_k(dir_v(g)) * drgb[v][g][c]   (k < (D+1)^2; zero above)
// means_stride / drgb_stride / campos_stride: floats between consecutive views (means_stride == 0: one set of
// positions for every view -- a static model; > 0: the positions of a posed model differ per view and travel with the
// exchange).  The strides let the kernel read straight out of the all-gathered payload rows.
__global__ void __launch_bounds__(SGR_MV_THREADS)
sgr_sh_grad_from_views_kernel(int P, int D, int M, int V, const float* __restrict__ means3D, size_t means_stride,
                              const float* __restrict__ campos, size_t campos_stride, const float* __restrict__ drgb,
                              size_t drgb_stride, float* __restrict__ dL_dsh) {
    const int gidx = blockIdx.x * SGR_MV_THREADS + threadIdx.x;
    const bool live = gidx < P;
    const int idx = live ? gidx : P - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ncoef = (D + 1) * (D + 1);
    float acc[48];
#pragma unroll
    for (int k = 0; k < 48; k++) acc[k] = 0.f;
    const float p0[3] = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    // Views four at a time: the 12 (24 with per-view positions) loads of a group go out together, then the arithmetic
    // runs on registers.  (One view per trip with an early `continue` made every view's loads wait for the previous
    // view's arithmetic: at V = 8 the kernel was a chain of eight dependent round trips to HBM.)  The sum runs in view
    // order whatever the grouping: deterministic and identical on every rank.
    for (int v0 = 0; v0 < V; v0 += 4) {
        float d[4][3], pp[4][3], cc[4][3];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int v = min(v0 + j, V - 1);  // clamped: the extra loads of a ragged last group are discarded
            const float* g = drgb + (size_t)v * drgb_stride + (size_t)idx * 3;
            d[j][0] = g[0]; d[j][1] = g[1]; d[j][2] = g[2];
            if (means_stride) {
                const float* mp = means3D + (size_t)v * means_stride + (size_t)idx * 3;
                pp[j][0] = mp[0]; pp[j][1] = mp[1]; pp[j][2] = mp[2];
            } else {
                pp[j][0] = p0[0]; pp[j][1] = p0[1]; pp[j][2] = p0[2];
            }
            const float* cp = campos + (size_t)v * campos_stride;  // wave-uniform: scalar loads
            cc[j][0] = cp[0]; cc[j][1] = cp[1]; cc[j][2] = cp[2];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (v0 + j >= V) break;
            const float d0 = d[j][0], d1 = d[j][1], d2 = d[j][2];
            if (d0 == 0.f && d1 == 0.f && d2 == 0.f) continue;  // culled / fully clamped / not rendered in this view
            // same expressions as the per-view backward (sgr_gauss_bwd.hip)
            const float ox = pp[j][0] - cc[j][0], oy = pp[j][1] - cc[j][1], oz = pp[j][2] - cc[j][2];
            const float len = sqrtf(ox * ox + oy * oy + oz * oz);
            float Y[16];
            sgr_sh_basis(D, ox / len, oy / len, oz / len, Y);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k < ncoef) {
                    acc[3 * k] += Y[k] * d0;
                    acc[3 * k + 1] += Y[k] * d1;
                    acc[3 * k + 2] += Y[k] * d2;
                }
            }
        }
    }
    if (M == 16) {
        // rows of 48 floats: the wave's 64 rows are one contiguous 12 KB block -> coalesced 1 KB stores through LDS
        __shared__ float4 sRow[SGR_MV_THREADS / 64][64 * 12];
        const int g0 = blockIdx.x * SGR_MV_THREADS + wave * 64;
        const int nrow4 = max(0, min(64, P - g0)) * 12;
#pragma unroll
        for (int i = 0; i < 12; i++)
            sRow[wave][lane * 12 + i] = make_float4(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]);
        __builtin_amdgcn_wave_barrier();
        float4* dst = reinterpret_cast<float4*>(dL_dsh) + (size_t)g0 * 12;
#pragma unroll
        for (int it = 0; it < 12; it++) {
            const int f = it * 64 + lane;
            if (f < nrow4) dst[f] = sRow[wave][f];
        }
    } else if (live) {
        float* dsh = dL_dsh + (size_t)idx * M * 3;
#pragma unroll
        for (int k = 0; k < 48; k++)
            if (k < M * 3) dsh[k] = acc[k];
        for (int k = 48; k < M * 3; k++) dsh[k] = 0.f;
    }
}

void sgr_launch_masked_color_grad(int P, const uint32_t* clamped, const float* dL_dcolor, float* out, hipStream_t s) {
    if (P <= 0) return;
    sgr_masked_color_grad_kernel<<<(3 * P + SGR_MV_THREADS - 1) / SGR_MV_THREADS, SGR_MV_THREADS, 0, s>>>(P, clamped,
                                                                                                        dL_dcolor, out);
}

void sgr_launch_sh_grad_from_views(int P, int D, int M, int V, const float* means3D, size_t means_stride,
                                   const float* campos, size_t campos_stride, const float* drgb, size_t drgb_stride,
                                   float* dL_dsh, hipStream_t s) {
    if (P <= 0) return;
    sgr_sh_grad_from_views_kernel<<<(P + SGR_MV_THREADS - 1) / SGR_MV_THREADS, SGR_MV_THREADS, 0, s>>>(
        P, D, M, V, means3D, means_stride, campos, campos_stride, drgb, drgb_stride, dL_dsh);
}
