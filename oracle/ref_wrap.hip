// TEST INFRASTRUCTURE, NOT PRODUCT CODE.
// C-ABI wrapper around the reference's own CudaRasterizer::Rasterizer and SimpleKNN classes
// (/root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer/rasterizer.h:20-114,
//  /root/reference/submodules/simple-knn/simple_knn.h:14-18), which oracle/ref_build.sh compiles
// UNMODIFIED for gfx950 from the sources where they lie.  It replaces the torch glue of
// rasterize_points.cu so tests can run the reference kernels on the MI355X box through ctypes
// ("oracle/_ref", the secondary oracle that pins oracle/sgr_oracle.c and the HIP path).
#include <cstdint>
#include <cstdio>
#include <functional>
#include <vector>
#include "cuda_runtime.h"
#include "rasterizer.h"
#include "rasterizer_impl.h"

namespace {
struct RefHandle {
    char* geom = nullptr;
    char* binning = nullptr;
    char* img = nullptr;
    int P = 0, W = 0, H = 0, R = 0;
};
// Pooled mode (ref_set_pooled(1), bench.py's "reference kernels on MI355X" timing): the three scratch buffers are kept
// and re-used across calls like torch's caching allocator would for the reference's own glue
// (rasterize_points.cu:27-33), so the timing does not include hipMalloc / hipFree.  One live handle at a time.
bool g_pooled = false;
char* g_pool[3] = {nullptr, nullptr, nullptr};
size_t g_pool_bytes[3] = {0, 0, 0};
std::function<char*(size_t)> grow(char** slot, int which) {
    return [slot, which](size_t n) {
        if (g_pooled) {
            if (g_pool_bytes[which] < n) {
                if (g_pool[which]) hipFree(g_pool[which]);
                hipMalloc((void**)&g_pool[which], n + n / 4 + 256);
                g_pool_bytes[which] = n + n / 4 + 256;
            }
            *slot = g_pool[which];
            return *slot;
        }
        if (*slot) hipFree(*slot);
        hipMalloc((void**)slot, n ? n : 1);
        return *slot;
    };
}
}  // namespace

class SimpleKNN {
public:
    static void knn(int P, float3* points, float* meanDists);
};

extern "C" {

void* ref_forward(int P, int D, int M, int S, const float* bg, int W, int H, const float* means3D, const float* shs,
                  const float* colors_precomp, const float* semantics, const float* opacities, const float* scales,
                  float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                  const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, int prefiltered,
                  float* out_color, float* out_depth, float* out_alpha, float* out_semantic, int* radii, int debug,
                  int* num_rendered) {
    RefHandle* h = new RefHandle();
    h->P = P; h->W = W; h->H = H;
    int R = CudaRasterizer::Rasterizer::forward(grow(&h->geom, 0), grow(&h->binning, 1), grow(&h->img, 2), P, D, M, S, bg, W, H,
                                                means3D, shs, colors_precomp, semantics, opacities, scales,
                                                scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix,
                                                campos, tan_fovx, tan_fovy, prefiltered != 0, out_color, out_depth,
                                                out_alpha, out_semantic, radii, debug != 0);
    hipDeviceSynchronize();
    h->R = R;
    *num_rendered = R;
    return h;
}

void ref_backward(void* handle, int D, int M, int S, const float* bg, const float* means3D, const float* shs,
                  const float* colors_precomp, const float* semantics, const float* alphas, const float* scales,
                  float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                  const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                  const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dalphas,
                  const float* dL_dpix_semantic, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                  float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                  float* dL_dscale, float* dL_drot, float* dL_dsemantic, int debug) {
    RefHandle* h = (RefHandle*)handle;
    CudaRasterizer::Rasterizer::backward(h->P, D, M, h->R, S, bg, h->W, h->H, means3D, shs, colors_precomp, semantics,
                                         alphas, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix,
                                         projmatrix, campos, tan_fovx, tan_fovy, radii, h->geom, h->binning, h->img,
                                         dL_dpix, dL_dpix_depth, dL_dalphas, dL_dpix_semantic, dL_dmean2D, dL_dconic,
                                         dL_dopacity, dL_dcolor, dL_ddepth, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale,
                                         dL_drot, dL_dsemantic, debug != 0);
    hipDeviceSynchronize();
}

// Device pointers into the reference's opaque buffers, recovered with the reference's own
// fromChunk carving (rasterizer_impl.cu:155-193).  which: 0 depths 1 clamped 2 means2D 3 cov3D
// 4 conic_opacity 5 rgb 6 tiles_touched 7 point_offsets 8 point_list 9 point_list_keys
// 10 point_list_unsorted 11 point_list_keys_unsorted 12 ranges 13 n_contrib
void* ref_internal(void* handle, int which) {
    RefHandle* h = (RefHandle*)handle;
    char* g = h->geom;
    char* b = h->binning;
    char* i = h->img;
    CudaRasterizer::GeometryState gs = CudaRasterizer::GeometryState::fromChunk(g, h->P);
    CudaRasterizer::BinningState bs = CudaRasterizer::BinningState::fromChunk(b, h->R);
    CudaRasterizer::ImageState is = CudaRasterizer::ImageState::fromChunk(i, (size_t)h->W * h->H);
    switch (which) {
        case 0: return gs.depths;
        case 1: return gs.clamped;
        case 2: return gs.means2D;
        case 3: return gs.cov3D;
        case 4: return gs.conic_opacity;
        case 5: return gs.rgb;
        case 6: return gs.tiles_touched;
        case 7: return gs.point_offsets;
        case 8: return bs.point_list;
        case 9: return bs.point_list_keys;
        case 10: return bs.point_list_unsorted;
        case 11: return bs.point_list_keys_unsorted;
        case 12: return is.ranges;
        case 13: return is.n_contrib;
    }
    return nullptr;
}

void ref_free(void* handle) {
    RefHandle* h = (RefHandle*)handle;
    if (!h) return;
    if (!g_pooled || h->geom != g_pool[0]) { if (h->geom) hipFree(h->geom); }
    if (!g_pooled || h->binning != g_pool[1]) { if (h->binning) hipFree(h->binning); }
    if (!g_pooled || h->img != g_pool[2]) { if (h->img) hipFree(h->img); }
    delete h;
}

void ref_set_pooled(int on) {
    g_pooled = on != 0;
    if (!g_pooled) {
        for (int i = 0; i < 3; i++) {
            if (g_pool[i]) hipFree(g_pool[i]);
            g_pool[i] = nullptr;
            g_pool_bytes[i] = 0;
        }
    }
}

void ref_mark_visible(int P, float* means3D, float* viewmatrix, float* projmatrix, bool* present) {
    CudaRasterizer::Rasterizer::markVisible(P, means3D, viewmatrix, projmatrix, present);
    hipDeviceSynchronize();
}

void ref_visible_filter(int P, int M, int W, int H, const float* means3D, const float* scales, float scale_modifier,
                        const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                        const float* projmatrix, float tan_fovx, float tan_fovy, int prefiltered, int* radii,
                        float* means2D, int debug) {
    char *g = nullptr, *b = nullptr, *i = nullptr;
    const bool was_pooled = g_pooled;
    g_pooled = false;
    CudaRasterizer::Rasterizer::visible_filter(grow(&g, 0), grow(&b, 1), grow(&i, 2), P, M, W, H, means3D, scales,
                                               scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix,
                                               tan_fovx, tan_fovy, prefiltered != 0, radii, means2D, debug != 0);
    hipDeviceSynchronize();
    g_pooled = was_pooled;
    if (g) hipFree(g);
    if (b) hipFree(b);
    if (i) hipFree(i);
}

void ref_knn(int P, float* points, float* meanDists) {
    SimpleKNN::knn(P, (float3*)points, meanDists);
    hipDeviceSynchronize();
}
}
