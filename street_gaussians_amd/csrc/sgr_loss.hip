// sgr_loss.hip -- the colour loss right after the rasterizer (include/sgr_loss.h) on gfx950.
//
// SSIM: the reference's 11x11 window is the outer product of a normalised 1-D Gaussian (loss_utils.py:70-78), so the
// five windowed means (mu1, mu2, E[x^2], E[y^2], E[xy]) are computed separably inside one workgroup: a 16x16 output tile
// stages its 26x26 input halo in LDS, filters the 26 rows horizontally (5 quantities), then the 16 columns
// vertically, evaluates the SSIM expression and -- for the backward -- its three partial derivatives per position.
// The backward filters those three maps the same way: dL/dx = G*(dM/dmu1) + 2x G*(dM/ds11) + y G*(dM/ds12)
// (the window is symmetric and the padding is zero, so the adjoint of the correlation is the correlation).
// 2 kernels instead of ~30 (MIOpen depthwise convs + elementwise autograd graph); HBM traffic: two images in, three
// maps out / three maps + two images in, one gradient out.
#include <string>

#include "../../include/sgr_loss.h"
#include "sgr_common.h"

int sgr_set_error(int code, const std::string& msg);

#define SGR_LS_T 16           // output tile edge
#define SGR_LS_R 5            // window radius (window_size 11)
#define SGR_LS_IN (SGR_LS_T + 2 * SGR_LS_R)  // 26

struct SgrGauss11 { float w[11]; };
// gaussian(11, 1.5) of loss_utils.py:70-72, evaluated like the reference: exp() in double, float32 tensor, / sum
static SgrGauss11 make_window() {
    SgrGauss11 g;
    float v[11], s = 0.f;
    for (int x = 0; x < 11; x++) {
        v[x] = (float)exp(-(double)((x - 5) * (x - 5)) / (2.0 * 1.5 * 1.5));
        s += v[x];
    }
    for (int x = 0; x < 11; x++) g.w[x] = v[x] / s;
    return g;
}

__device__ __forceinline__ float block_sum_256(float v, float* lds4) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((lds4[0] + lds4[1]) + lds4[2]) + lds4[3];
}

template <bool WITH_PARTIALS>
__global__ void __launch_bounds__(256)
sgr_ssim_fwd_kernel(int H, int W, const float* __restrict__ img1, const float* __restrict__ img2,
                    const uint8_t* __restrict__ mask, SgrGauss11 g, float* __restrict__ partials,
                    float* __restrict__ block_sums) {
    __shared__ float sx[SGR_LS_IN][SGR_LS_IN + 1], sy[SGR_LS_IN][SGR_LS_IN + 1];
    __shared__ float hb[5][SGR_LS_IN][SGR_LS_T + 1];
    __shared__ float lds4[4];
    const int c = blockIdx.z, x0 = blockIdx.x * SGR_LS_T, y0 = blockIdx.y * SGR_LS_T;
    const size_t plane = (size_t)H * W;
    const float* p1 = img1 + c * plane;
    const float* p2 = img2 + c * plane;
    for (int e = threadIdx.x; e < SGR_LS_IN * SGR_LS_IN; e += 256) {
        const int r = e / SGR_LS_IN, q = e - r * SGR_LS_IN;
        const int y = y0 + r - SGR_LS_R, x = x0 + q - SGR_LS_R;
        float a = 0.f, b = 0.f;
        if (x >= 0 && x < W && y >= 0 && y < H) {  // zero padding (F.conv2d padding=5); masked pixels are zeros (:92-94)
            const size_t o = (size_t)y * W + x;
            if (!mask || mask[o]) { a = p1[o]; b = p2[o]; }
        }
        sx[r][q] = a;
        sy[r][q] = b;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < SGR_LS_IN * SGR_LS_T; e += 256) {  // horizontal pass: 26 rows x 16 columns
        const int r = e / SGR_LS_T, q = e - r * SGR_LS_T;
        float m1 = 0.f, m2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float a = sx[r][q + k], b = sy[r][q + k], w = g.w[k];
            m1 += w * a; m2 += w * b; s11 += w * a * a; s22 += w * b * b; s12 += w * a * b;
        }
        hb[0][r][q] = m1; hb[1][r][q] = m2; hb[2][r][q] = s11; hb[3][r][q] = s22; hb[4][r][q] = s12;
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float m1 = 0.f, m2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float w = g.w[k];
        m1 += w * hb[0][ty + k][tx]; m2 += w * hb[1][ty + k][tx]; s11 += w * hb[2][ty + k][tx];
        s22 += w * hb[3][ty + k][tx]; s12 += w * hb[4][ty + k][tx];
    }
    const int x = x0 + tx, y = y0 + ty;
    float val = 0.f;
    if (x < W && y < H) {
        // loss_utils.py:104-118
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float mu1_sq = m1 * m1, mu2_sq = m2 * m2, mu12 = m1 * m2;
        const float sg1 = s11 - mu1_sq, sg2 = s22 - mu2_sq, sg12 = s12 - mu12;
        const float A = 2.f * mu12 + C1, B = 2.f * sg12 + C2, Cc = mu1_sq + mu2_sq + C1, D = sg1 + sg2 + C2;
        val = (A * B) / (Cc * D);
        if (WITH_PARTIALS) {
            const size_t o = c * plane + (size_t)y * W + x, cp = (size_t)gridDim.z * plane;
            const float icd = 1.0f / (Cc * D);
            partials[o] = 2.f * m2 * (B - A) * icd - 2.f * m1 * val * (1.0f / Cc - 1.0f / D);  // dM/dmu1
            partials[cp + o] = -val / D;                                                            // dM/dsigma1^2
            partials[2 * cp + o] = 2.f * A * icd;                                                    // dM/dsigma12
        }
    }
    const float s = block_sum_256(val, lds4);
    if (threadIdx.x == 0) block_sums[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
}

// one workgroup: out[0] = scale * sum(v[0..n)) in a fixed order; out[1] = optional second quantity (counts)
__global__ void __launch_bounds__(256)
sgr_final_sum_kernel(const float* __restrict__ v, size_t n, float scale, const float* __restrict__ v2, float* __restrict__ out,
                     int mean_by_second) {
    __shared__ float lds4[4];
    float a = 0.f, b = 0.f;
    for (size_t i = threadIdx.x; i < n; i += 256) {
        a += v[i];
        if (v2) b += v2[i];
    }
    const float sa = block_sum_256(a, lds4);
    __syncthreads();
    const float sb = v2 ? block_sum_256(b, lds4) : 0.f;
    if (threadIdx.x == 0) {
        if (mean_by_second) { out[0] = sa / sb; out[1] = sb; }
        else out[0] = sa * scale;
    }
}

__global__ void __launch_bounds__(256)
sgr_ssim_bwd_kernel(int H, int W, int C, const float* __restrict__ img1, const float* __restrict__ img2,
                    const uint8_t* __restrict__ mask, SgrGauss11 g, const float* __restrict__ partials,
                    const float* __restrict__ upstream, float* __restrict__ dimg1, float w_ssim, float w_l1,
                    const float* __restrict__ l1_out) {
    __shared__ float sp[3][SGR_LS_IN][SGR_LS_IN + 1];
    __shared__ float hb[3][SGR_LS_IN][SGR_LS_T + 1];
    const int c = blockIdx.z, x0 = blockIdx.x * SGR_LS_T, y0 = blockIdx.y * SGR_LS_T;
    const size_t plane = (size_t)H * W, cp = (size_t)C * plane;
    for (int e = threadIdx.x; e < SGR_LS_IN * SGR_LS_IN; e += 256) {
        const int r = e / SGR_LS_IN, q = e - r * SGR_LS_IN;
        const int y = y0 + r - SGR_LS_R, x = x0 + q - SGR_LS_R;
        float a = 0.f, b = 0.f, d = 0.f;
        if (x >= 0 && x < W && y >= 0 && y < H) {
            const size_t o = c * plane + (size_t)y * W + x;
            a = partials[o]; b = partials[cp + o]; d = partials[2 * cp + o];
        }
        sp[0][r][q] = a; sp[1][r][q] = b; sp[2][r][q] = d;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < SGR_LS_IN * SGR_LS_T; e += 256) {
        const int r = e / SGR_LS_T, q = e - r * SGR_LS_T;
        float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = g.w[k];
            a += w * sp[0][r][q + k]; b += w * sp[1][r][q + k]; d += w * sp[2][r][q + k];
        }
        hb[0][r][q] = a; hb[1][r][q] = b; hb[2][r][q] = d;
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x = x0 + tx, y = y0 + ty;
    if (x >= W || y >= H) return;
    float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float w = g.w[k];
        a += w * hb[0][ty + k][tx]; b += w * hb[1][ty + k][tx]; d += w * hb[2][ty + k][tx];
    }
    const size_t o = (size_t)y * W + x;
    float out = 0.f;
    if (!mask || mask[o]) {
        const float xv = img1[c * plane + o], yv = img2[c * plane + o];
        out = w_ssim * ((a + 2.f * xv * b + yv * d) * (upstream[0] / (float)cp));
        if (l1_out != nullptr) {  // fused colour loss: + w_l1 * d l1_loss / d img1  (loss_utils.py:21-37)
            const float df = xv - yv;
            const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
            out += w_l1 * upstream[0] * sg / l1_out[1];
        }
    }
    dimg1[c * plane + o] = out;
}

// ---- L1 -----------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
sgr_l1_fwd_kernel(int C, size_t plane, const float* __restrict__ a, const float* __restrict__ b,
                  const uint8_t* __restrict__ mask, float* __restrict__ sums, float* __restrict__ counts) {
    __shared__ float lds4[4];
    float s = 0.f, n = 0.f;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < plane; p += (size_t)gridDim.x * 256) {
        if (mask && !mask[p]) continue;
        for (int c = 0; c < C; c++) s += fabsf(a[c * plane + p] - b[c * plane + p]);
        n += (float)C;
    }
    const float ss = block_sum_256(s, lds4);
    __syncthreads();
    const float nn = block_sum_256(n, lds4);
    if (threadIdx.x == 0) { sums[blockIdx.x] = ss; counts[blockIdx.x] = nn; }
}

__global__ void __launch_bounds__(256)
sgr_l1_bwd_kernel(int C, size_t plane, const float* __restrict__ a, const float* __restrict__ b,
                  const uint8_t* __restrict__ mask, const float* __restrict__ out, const float* __restrict__ upstream,
                  float* __restrict__ da) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)C * plane) return;
    const size_t p = i % plane;
    float g = 0.f;
    if (!mask || mask[p]) {
        const float d = a[i] - b[i];
        g = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * (upstream[0] / out[1]);  // torch: sign(0) = 0
    }
    da[i] = g;
}

// ---- accumulation terms (train.py:106-121) --------------------------------------------------------------------------
__device__ __forceinline__ float bce_term(int mode, float a, bool m) {
    if (mode == SGR_BCE_SKY) return m ? -logf(1.0f - a) : -logf(a);
    return m ? -(a * logf(a) + (1.0f - a) * logf(1.0f - a)) : -logf(1.0f - a);
}
__global__ void __launch_bounds__(256)
sgr_bce_fwd_kernel(int n, int mode, const float* __restrict__ acc, const uint8_t* __restrict__ mask,
                   float* __restrict__ sums) {
    __shared__ float lds4[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * 256) {
        const float a = fminf(fmaxf(acc[i], 1e-6f), 1.0f - 1e-6f);  // torch.clamp(acc, min=1e-6, max=1.-1e-6)
        s += bce_term(mode, a, mask && mask[i]);
    }
    const float t = block_sum_256(s, lds4);
    if (threadIdx.x == 0) sums[blockIdx.x] = t;
}
__global__ void __launch_bounds__(256)
sgr_bce_bwd_kernel(int n, int mode, const float* __restrict__ acc, const uint8_t* __restrict__ mask,
                   const float* __restrict__ upstream, float* __restrict__ dacc) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n) return;
    const float x = acc[i];
    float g = 0.f;
    if (x >= 1e-6f && x <= 1.0f - 1e-6f) {  // clamp passes the gradient on [min, max]
        const bool m = mask && mask[i];
        if (mode == SGR_BCE_SKY) g = m ? 1.0f / (1.0f - x) : -1.0f / x;
        else g = m ? (logf(1.0f - x) - logf(x)) : 1.0f / (1.0f - x);
        g *= upstream[0] / (float)n;
    }
    dacc[i] = g;
}

// ---- LiDAR depth term with top-95 % selection (train.py:124-131) ---------------------------------------------------
struct LdWork {
    uint32_t* key;    // [n] float bits of the error (>= 0, so unsigned order == float order); 0xffffffff = not selected
    uint32_t* hist;   // [4][256]
    uint32_t* state;  // [0] count  [1] k  [2] prefix  [3] k remaining  [4] count below threshold  [5] ties
    float* sums;      // [SGR_L1_BLOCKS]
};
#define SGR_L1_BLOCKS_ 1024
static LdWork ld_carve(char* base, size_t n) {
    LdWork w;
    char* p = base;
    sgr_carve(p, w.key, n ? n : 1);
    sgr_carve(p, w.hist, 4 * 256);
    sgr_carve(p, w.state, 16);
    sgr_carve(p, w.sums, SGR_L1_BLOCKS_);
    return w;
}
__global__ void __launch_bounds__(256)
sgr_lidar_err_kernel(int n, const float* __restrict__ depth, const float* __restrict__ acc, const float* __restrict__ lidar,
                     const uint8_t* __restrict__ mask, LdWork w) {
    __shared__ uint32_t cnt[4];
    uint32_t c = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * 256) {
        const float l = lidar[i];
        uint32_t key = 0xffffffffu;
        if (l > 0.f && (!mask || mask[i])) {  // depth_mask = (lidar_depth > 0) & mask
            const float e = fabsf(depth[i] / (acc[i] + 1e-10f) - l);
            key = (e == e) ? __float_as_uint(e) : 0x7fc00000u;  // NaN sorts last among the selected, like topk
            c++;
        }
        w.key[i] = key;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&w.state[0], cnt[0] + cnt[1] + cnt[2] + cnt[3]);  // integers: order-independent
}
__global__ void sgr_lidar_k_kernel(LdWork w, double keep) {
    const uint32_t count = w.state[0];
    const uint32_t k = (uint32_t)(keep * (double)count);  // int(0.95 * depth_error.size(0)): Python float = double
    w.state[1] = k;
    w.state[2] = 0;  // prefix
    w.state[3] = k;  // we look for the k-th smallest (1-based) -> rank k inside the current prefix class
}
// histogram of byte `pass` (3 = most significant) over the selected keys whose higher bytes equal the prefix
__global__ void __launch_bounds__(256)
sgr_lidar_hist_kernel(int n, int pass, LdWork w) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t prefix = w.state[2];
    const int shift = 8 * pass;
    const uint32_t himask = pass == 3 ? 0u : (0xffffffffu << (shift + 8));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * 256) {
        const uint32_t key = w.key[i];
        if (key != 0xffffffffu && (key & himask) == prefix) atomicAdd(&h[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&w.hist[pass * 256 + threadIdx.x], h[threadIdx.x]);
}
__global__ void sgr_lidar_select_kernel(int pass, LdWork w) {  // one lane
    uint32_t k = w.state[3], acc = 0;
    if (w.state[1] == 0) return;
    int b = 0;
    for (; b < 256; b++) {
        const uint32_t c = w.hist[pass * 256 + b];
        if (acc + c >= k) break;
        acc += c;
    }
    if (b > 255) b = 255;
    w.state[2] |= (uint32_t)b << (8 * pass);
    w.state[3] = k - acc;  // rank inside the chosen bin; after the last pass: how many of the tied errors are taken
    if (pass == 0) {
        w.state[5] = w.hist[b];                 // errors equal to the threshold
        w.state[4] = w.state[1] - w.state[3];   // errors strictly below it
    }
}
__global__ void __launch_bounds__(256)
sgr_lidar_sum_kernel(int n, LdWork w) {
    __shared__ float lds4[4];
    const uint32_t t = w.state[2];
    float s = 0.f;
    if (w.state[1] > 0)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * 256) {
            const uint32_t key = w.key[i];
            if (key < t) s += __uint_as_float(key);
        }
    const float r = block_sum_256(s, lds4);
    if (threadIdx.x == 0) w.sums[blockIdx.x] = r;
}
__global__ void __launch_bounds__(256)
sgr_lidar_final_kernel(LdWork w, float* __restrict__ out) {
    __shared__ float lds4[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < SGR_L1_BLOCKS_; i += 256) a += w.sums[i];
    const float below = block_sum_256(a, lds4);
    if (threadIdx.x == 0) {
        const uint32_t k = w.state[1];
        const float thr = __uint_as_float(w.state[2]);
        const uint32_t take = w.state[3];  // how many of the tied errors belong to the k smallest
        out[0] = k ? (below + (float)take * thr) / (float)k : __builtin_nanf("");  // mean of an empty tensor is NaN
        out[1] = (float)k;
        out[2] = thr;
        out[3] = (k && w.state[5]) ? (float)take / (float)w.state[5] : 0.f;
    }
}
__global__ void __launch_bounds__(256)
sgr_lidar_bwd_kernel(int n, const float* __restrict__ depth, const float* __restrict__ acc, const float* __restrict__ lidar,
                     const float* __restrict__ out, LdWork w, const float* __restrict__ upstream,
                     float* __restrict__ ddepth, float* __restrict__ dacc) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n) return;
    const uint32_t key = w.key[i], t = __float_as_uint(out[2]);
    float wgt = 0.f;
    if (out[1] > 0.f && key != 0xffffffffu) wgt = key < t ? 1.0f : (key == t ? out[3] : 0.f);
    float gd = 0.f, ga = 0.f;
    if (wgt > 0.f) {
        const float den = acc[i] + 1e-10f, diff = depth[i] / den - lidar[i];
        const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        const float g = upstream[0] * wgt * sgn / out[1];
        gd = g / den;
        ga = -g * depth[i] / (den * den);
    }
    ddepth[i] = gd;
    dacc[i] = ga;
}

#define LS_HIP(call)                                                                                       \
    do {                                                                                                   \
        hipError_t e__ = (call);                                                                           \
        if (e__ != hipSuccess) return sgr_set_error(SGR_E_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

static inline dim3 tile_grid(int C, int H, int W) {
    return dim3((W + SGR_LS_T - 1) / SGR_LS_T, (H + SGR_LS_T - 1) / SGR_LS_T, C);
}
#define SGR_L1_BLOCKS 1024

extern "C" {

size_t sgr_ssim_workspace_floats(int C, int H, int W) {
    const dim3 g = tile_grid(C, H, W);
    return (size_t)g.x * g.y * g.z;
}

int sgr_ssim_forward(int C, int H, int W, const float* img1, const float* img2, const uint8_t* mask, float* out_ssim,
                     float* partials, float* workspace, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (C <= 0 || H <= 0 || W <= 0) return sgr_set_error(SGR_E_INVALID, "C, H, W must be positive");
    if (!img1 || !img2 || !out_ssim || !workspace) return sgr_set_error(SGR_E_INVALID, "img1, img2, out_ssim and workspace are required");
    static const SgrGauss11 g = make_window();
    const dim3 grid = tile_grid(C, H, W);
    if (partials) sgr_ssim_fwd_kernel<true><<<grid, 256, 0, stream>>>(H, W, img1, img2, mask, g, partials, workspace);
    else sgr_ssim_fwd_kernel<false><<<grid, 256, 0, stream>>>(H, W, img1, img2, mask, g, nullptr, workspace);
    sgr_final_sum_kernel<<<1, 256, 0, stream>>>(workspace, (size_t)grid.x * grid.y * grid.z,
                                                1.0f / ((float)C * (float)H * (float)W), nullptr, out_ssim, 0);
    LS_HIP(hipGetLastError());
    return 0;
}

int sgr_ssim_backward(int C, int H, int W, const float* img1, const float* img2, const uint8_t* mask,
                      const float* partials, const float* upstream, float* dL_dimg1, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (C <= 0 || H <= 0 || W <= 0) return sgr_set_error(SGR_E_INVALID, "C, H, W must be positive");
    if (!img1 || !img2 || !partials || !upstream || !dL_dimg1)
        return sgr_set_error(SGR_E_INVALID, "img1, img2, partials, upstream and dL_dimg1 are required");
    static const SgrGauss11 g = make_window();
    sgr_ssim_bwd_kernel<<<tile_grid(C, H, W), 256, 0, stream>>>(H, W, C, img1, img2, mask, g, partials, upstream, dL_dimg1,
                                                                1.0f, 0.0f, nullptr);
    LS_HIP(hipGetLastError());
    return 0;
}

int sgr_color_loss_backward(int C, int H, int W, const float* img1, const float* img2, const uint8_t* mask,
                            const float* partials, const float* l1_out, float w_l1, float w_ssim, const float* upstream,
                            float* dL_dimg1, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (C <= 0 || H <= 0 || W <= 0) return sgr_set_error(SGR_E_INVALID, "C, H, W must be positive");
    if (!img1 || !img2 || !partials || !l1_out || !upstream || !dL_dimg1)
        return sgr_set_error(SGR_E_INVALID, "img1, img2, partials, l1_out, upstream and dL_dimg1 are required");
    static const SgrGauss11 g = make_window();
    sgr_ssim_bwd_kernel<<<tile_grid(C, H, W), 256, 0, stream>>>(H, W, C, img1, img2, mask, g, partials, upstream, dL_dimg1,
                                                                w_ssim, w_l1, l1_out);
    LS_HIP(hipGetLastError());
    return 0;
}

size_t sgr_l1_workspace_floats(int C, int H, int W) {
    (void)C; (void)H; (void)W;
    return 2 * SGR_L1_BLOCKS;
}

int sgr_l1_forward(int C, int H, int W, const float* a, const float* b, const uint8_t* mask, float* out, float* workspace,
                   void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (C <= 0 || H <= 0 || W <= 0) return sgr_set_error(SGR_E_INVALID, "C, H, W must be positive");
    if (!a || !b || !out || !workspace) return sgr_set_error(SGR_E_INVALID, "a, b, out and workspace are required");
    const size_t plane = (size_t)H * W;
    sgr_l1_fwd_kernel<<<SGR_L1_BLOCKS, 256, 0, stream>>>(C, plane, a, b, mask, workspace, workspace + SGR_L1_BLOCKS);
    sgr_final_sum_kernel<<<1, 256, 0, stream>>>(workspace, SGR_L1_BLOCKS, 1.0f, workspace + SGR_L1_BLOCKS, out, 1);
    LS_HIP(hipGetLastError());
    return 0;
}

int sgr_l1_backward(int C, int H, int W, const float* a, const float* b, const uint8_t* mask, const float* out,
                    const float* upstream, float* dL_da, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (C <= 0 || H <= 0 || W <= 0) return sgr_set_error(SGR_E_INVALID, "C, H, W must be positive");
    if (!a || !b || !out || !upstream || !dL_da) return sgr_set_error(SGR_E_INVALID, "a, b, out, upstream and dL_da are required");
    const size_t n = (size_t)C * H * W;
    sgr_l1_bwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(C, (size_t)H * W, a, b, mask, out, upstream, dL_da);
    LS_HIP(hipGetLastError());
    return 0;
}

int sgr_bce_forward(int n, int mode, const float* acc, const uint8_t* mask, float* out, float* workspace, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0 || (mode != SGR_BCE_SKY && mode != SGR_BCE_OBJECT)) return sgr_set_error(SGR_E_INVALID, "n > 0 and a valid mode are required");
    if (!acc || !out || !workspace) return sgr_set_error(SGR_E_INVALID, "acc, out and workspace are required");
    sgr_bce_fwd_kernel<<<SGR_L1_BLOCKS, 256, 0, stream>>>(n, mode, acc, mask, workspace);
    sgr_final_sum_kernel<<<1, 256, 0, stream>>>(workspace, SGR_L1_BLOCKS, 1.0f / (float)n, nullptr, out, 0);
    LS_HIP(hipGetLastError());
    return 0;
}

int sgr_bce_backward(int n, int mode, const float* acc, const uint8_t* mask, const float* upstream, float* dL_dacc,
                     void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0 || (mode != SGR_BCE_SKY && mode != SGR_BCE_OBJECT)) return sgr_set_error(SGR_E_INVALID, "n > 0 and a valid mode are required");
    if (!acc || !upstream || !dL_dacc) return sgr_set_error(SGR_E_INVALID, "acc, upstream and dL_dacc are required");
    sgr_bce_bwd_kernel<<<(n + 255) / 256, 256, 0, stream>>>(n, mode, acc, mask, upstream, dL_dacc);
    LS_HIP(hipGetLastError());
    return 0;
}

size_t sgr_lidar_work_bytes(int n) {
    char* base = (char*)4096;
    LdWork w = ld_carve(base, (size_t)(n > 0 ? n : 1));
    return (size_t)((char*)(w.sums + SGR_L1_BLOCKS_) - base) + 512;
}

int sgr_lidar_depth_forward(int n, const float* depth, const float* acc, const float* lidar_depth, const uint8_t* mask,
                            double keep, float* out, char* work, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0) return sgr_set_error(SGR_E_INVALID, "n must be positive");
    if (!depth || !acc || !lidar_depth || !out || !work) return sgr_set_error(SGR_E_INVALID, "depth, acc, lidar_depth, out and work are required");
    const LdWork w = ld_carve((char*)sgr_align_up((size_t)work, 256), (size_t)n);
    LS_HIP(hipMemsetAsync(w.hist, 0, 4 * 256 * sizeof(uint32_t), stream));
    LS_HIP(hipMemsetAsync(w.state, 0, 16 * sizeof(uint32_t), stream));
    sgr_lidar_err_kernel<<<SGR_L1_BLOCKS_, 256, 0, stream>>>(n, depth, acc, lidar_depth, mask, w);
    sgr_lidar_k_kernel<<<1, 1, 0, stream>>>(w, keep);
    for (int pass = 3; pass >= 0; pass--) {
        sgr_lidar_hist_kernel<<<SGR_L1_BLOCKS_, 256, 0, stream>>>(n, pass, w);
        sgr_lidar_select_kernel<<<1, 1, 0, stream>>>(pass, w);
    }
    sgr_lidar_sum_kernel<<<SGR_L1_BLOCKS_, 256, 0, stream>>>(n, w);
    sgr_lidar_final_kernel<<<1, 256, 0, stream>>>(w, out);
    LS_HIP(hipGetLastError());
    return 0;
}

int sgr_lidar_depth_backward(int n, const float* depth, const float* acc, const float* lidar_depth, const float* out,
                             const char* work, const float* upstream, float* dL_ddepth, float* dL_dacc, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0) return sgr_set_error(SGR_E_INVALID, "n must be positive");
    if (!depth || !acc || !lidar_depth || !out || !work || !upstream || !dL_ddepth || !dL_dacc)
        return sgr_set_error(SGR_E_INVALID, "all arrays are required");
    const LdWork w = ld_carve((char*)sgr_align_up((size_t)const_cast<char*>(work), 256), (size_t)n);
    sgr_lidar_bwd_kernel<<<(n + 255) / 256, 256, 0, stream>>>(n, depth, acc, lidar_depth, out, w, upstream, dL_ddepth, dL_dacc);
    LS_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
