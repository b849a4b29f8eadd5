// ext.cpp -- the pybind module `_C` of the drop-in packages, as the reference builds it with torch's cpp_extension
// (/root/reference/submodules/diff-gaussian-rasterization/ext.cpp:15-20 + rasterize_points.h:18-88, setup.py:21-30;
//  /root/reference/submodules/simple-knn/ext.cpp:15-17): the same five entry points, the same argument lists and
// return tuples, over the C ABI of libsgr_hip.so (include/sgr.h).  Pure host code: tensors in, raw device pointers
// and the current HIP stream out; the three opaque buffers and the backward scratch are torch byte tensors grown
// through the C ABI's allocation callback -- i.e. torch's caching allocator, like resizeFunctional does in the
// reference (rasterize_points.cu:27-33).  street_gaussians_amd/_C.py keeps the ctypes binding of the same ABI
// (no compiler needed); `street_gaussians_amd._C.binding()` tells which one is active.
#include <torch/extension.h>

#include <c10/hip/HIPStream.h>

#include <functional>
#include <string>
#include <tuple>

#include "../../include/sgr.h"

namespace {

// Sizes that follow the number of Gaussians are rounded up to a ladder of eight steps per octave, so that a training loop
// whose P drifts (adaptive density control) keeps asking torch's caching allocator for block sizes it already holds
// (street_gaussians_amd/_alloc.py; at most 12.5 % of slack, exact below 1 MiB).
size_t ladder(size_t n) {
    if (n < (size_t(1) << 20)) return n;
    int bl = 0;
    for (size_t v = n; v; v >>= 1) bl++;
    const size_t step = size_t(1) << (bl - 4);
    return (n + step - 1) / step * step;
}

char* grow(size_t n, void* user) {  // sgr_alloc_fn over a torch byte tensor (a larger block than asked for is fine)
    auto* t = static_cast<torch::Tensor*>(user);
    t->resize_({static_cast<long long>(ladder(n))});
    return reinterpret_cast<char*>(t->contiguous().data_ptr());
}

void check(int rc) {
    if (rc < 0) throw std::runtime_error(sgr_last_error());
}

const float* fp(const torch::Tensor& t) {  // empty tensor = NULL = "feature absent" (rasterizer_impl.cu:324,452,482)
    if (!t.defined() || t.numel() == 0) return nullptr;
    TORCH_CHECK(t.is_cuda(), "street_gaussians_amd has no CPU path: tensors must be HIP (cuda) tensors");
    TORCH_CHECK(t.scalar_type() == torch::kFloat32, "float32 tensors expected");
    return t.data_ptr<float>();
}

void* stream_of(const torch::Tensor& t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

}  // namespace

// RasterizeGaussiansCUDA (rasterize_points.cu:35-124)
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
rasterize_gaussians(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                    const torch::Tensor& semantics, const torch::Tensor& opacity, const torch::Tensor& scales,
                    const torch::Tensor& rotations, const float scale_modifier, const torch::Tensor& cov3D_precomp,
                    const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
                    const float tan_fovy, const int image_height, const int image_width, const torch::Tensor& sh,
                    const int degree, const torch::Tensor& campos, const bool prefiltered, const bool debug) {
    if (means3D.ndimension() != 2 || means3D.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");
    TORCH_CHECK(means3D.is_cuda(), "means3D must be a HIP (cuda) tensor: street_gaussians_amd has no CPU path");
    c10::DeviceGuard guard(means3D.device());
    const int P = means3D.size(0), H = image_height, W = image_width;
    const int S = semantics.defined() && semantics.ndimension() == 2 ? semantics.size(1) : 0;
    auto fo = means3D.options().dtype(torch::kFloat32);
    auto bo = means3D.options().dtype(torch::kByte);
    torch::Tensor out_color = torch::empty({3, H, W}, fo), out_depth = torch::empty({1, H, W}, fo);
    torch::Tensor out_alpha = torch::empty({1, H, W}, fo), out_semantic = torch::empty({S, H, W}, fo);
    // the preprocess kernel writes every element (0 for culled Gaussians): no zero fill (rasterize_points.cu:74)
    torch::Tensor radii = P ? torch::empty({P}, means3D.options().dtype(torch::kInt32))
                            : torch::zeros({P}, means3D.options().dtype(torch::kInt32));
    torch::Tensor geom = torch::empty({0}, bo), binning = torch::empty({0}, bo), img = torch::empty({0}, bo);
    int M = 0;
    if (sh.defined() && sh.numel() != 0 && sh.size(0) != 0) M = sh.size(1);
    // inputs are made contiguous like rasterize_points.cu:99-120 does
    auto c = [](const torch::Tensor& t) { return t.defined() && t.numel() ? t.contiguous() : t; };
    const torch::Tensor bg = c(background), m3 = c(means3D), col = c(colors), sem = c(semantics), op = c(opacity),
                        sc = c(scales), rot = c(rotations), cov = c(cov3D_precomp), vm = c(viewmatrix),
                        pm = c(projmatrix), shc = c(sh), cp = c(campos);
    const int rendered = sgr_forward(grow, &geom, grow, &binning, grow, &img, P, degree, M, S, fp(bg), W, H, fp(m3), fp(shc),
                                     fp(col), fp(sem), fp(op), fp(sc), scale_modifier, fp(rot), fp(cov), fp(vm), fp(pm),
                                     fp(cp), tan_fovx, tan_fovy, prefiltered ? 1 : 0, out_color.data_ptr<float>(),
                                     out_depth.data_ptr<float>(), out_alpha.data_ptr<float>(),
                                     S ? out_semantic.data_ptr<float>() : nullptr, P ? radii.data_ptr<int>() : nullptr,
                                     debug ? 1 : 0, stream_of(means3D));
    check(rendered);
    return std::make_tuple(rendered, out_color, out_depth, out_alpha, out_semantic, radii, geom, binning, img);
}

// RasterizeGaussiansBackwardCUDA (rasterize_points.cu:126-220); every gradient element is written by the kernels, so the
// eleven torch::zeros of :166-176 are torch::empty here
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor, torch::Tensor>
rasterize_gaussians_backward(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                             const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                             const float scale_modifier, const torch::Tensor& cov3D_precomp,
                             const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
                             const float tan_fovy, const torch::Tensor& dL_dout_color, const torch::Tensor& dL_dout_depth,
                             const torch::Tensor& dL_dout_alpha, const torch::Tensor& dL_dout_semantic,
                             const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
                             const torch::Tensor& geomBuffer, const int R, const torch::Tensor& binningBuffer,
                             const torch::Tensor& imageBuffer, const torch::Tensor& alphas, const torch::Tensor& semantics,
                             const bool debug) {
    TORCH_CHECK(means3D.is_cuda(), "means3D must be a HIP (cuda) tensor: street_gaussians_amd has no CPU path");
    c10::DeviceGuard guard(means3D.device());
    const int P = means3D.size(0), H = dL_dout_color.size(1), W = dL_dout_color.size(2);
    const int S = dL_dout_semantic.defined() && dL_dout_semantic.numel() ? dL_dout_semantic.size(0) : 0;
    int M = 0;
    if (sh.defined() && sh.numel() != 0 && sh.size(0) != 0) M = sh.size(1);
    auto fo = means3D.options().dtype(torch::kFloat32);
    auto mk = [&](std::initializer_list<int64_t> shape) {  // exact shape, ladder-sized storage
        if (!P) return torch::zeros(shape, fo);
        int64_t n = 1;
        for (int64_t d : shape) n *= d;
        const size_t nb = size_t(n) * sizeof(float);
        if (nb < (size_t(1) << 20)) return torch::empty(shape, fo);
        torch::Tensor buf = torch::empty({static_cast<int64_t>(ladder(nb) / sizeof(float))}, fo);
        return buf.narrow(0, 0, n).view(shape);
    };
    torch::Tensor dL_dmeans3D = mk({P, 3}), dL_dmeans2D = mk({P, 3}), dL_dcolors = mk({P, 3}), dL_dopacity = mk({P, 1});
    torch::Tensor dL_dcov3D = mk({P, 6}), dL_dsh = mk({P, M, 3}), dL_dscales = mk({P, 3}), dL_drotations = mk({P, 4});
    torch::Tensor dL_dsemantic = mk({P, S});
    if (P != 0) {
        torch::Tensor scratch = torch::empty({0}, means3D.options().dtype(torch::kByte));
        auto c = [](const torch::Tensor& t) { return t.defined() && t.numel() ? t.contiguous() : t; };
        const torch::Tensor bg = c(background), m3 = c(means3D), col = c(colors), sem = c(semantics), al = c(alphas),
                            sc = c(scales), rot = c(rotations), cov = c(cov3D_precomp), vm = c(viewmatrix),
                            pm = c(projmatrix), shc = c(sh), cp = c(campos), gc = c(dL_dout_color), gd = c(dL_dout_depth),
                            ga = c(dL_dout_alpha), gs = c(dL_dout_semantic), rd = c(radii);
        auto bytes = [](const torch::Tensor& t) { return t.numel() ? reinterpret_cast<char*>(t.data_ptr()) : nullptr; };
        auto op = [](torch::Tensor& t) { return t.numel() ? t.data_ptr<float>() : nullptr; };
        check(sgr_backward(P, degree, M, R, S, fp(bg), W, H, fp(m3), fp(shc), fp(col), fp(sem), fp(al), fp(sc),
                           scale_modifier, fp(rot), fp(cov), fp(vm), fp(pm), fp(cp), tan_fovx, tan_fovy,
                           rd.numel() ? rd.data_ptr<int>() : nullptr, bytes(geomBuffer), bytes(binningBuffer),
                           bytes(imageBuffer), fp(gc), fp(gd), fp(ga), fp(gs), op(dL_dmeans2D), op(dL_dopacity),
                           op(dL_dcolors), op(dL_dmeans3D), op(dL_dcov3D), op(dL_dsh), op(dL_dscales), op(dL_drotations),
                           op(dL_dsemantic), grow, &scratch, debug ? 1 : 0, stream_of(means3D)));
    }
    return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations,
                           dL_dsemantic);
}

// markVisible (rasterize_points.cu:222-241)
torch::Tensor mark_visible(torch::Tensor& means3D, torch::Tensor& viewmatrix, torch::Tensor& projmatrix) {
    TORCH_CHECK(means3D.is_cuda(), "means3D must be a HIP (cuda) tensor: street_gaussians_amd has no CPU path");
    c10::DeviceGuard guard(means3D.device());
    const int P = means3D.size(0);
    torch::Tensor present = torch::zeros({P}, means3D.options().dtype(torch::kBool));
    if (P != 0) {
        const torch::Tensor m = means3D.contiguous(), v = viewmatrix.contiguous(), p = projmatrix.contiguous();
        check(sgr_mark_visible(P, fp(m), fp(v), fp(p), reinterpret_cast<uint8_t*>(present.data_ptr<bool>()), stream_of(means3D)));
    }
    return present;
}

// RasterizeGaussiansfilterCUDA (rasterize_points.cu:243-307)
std::tuple<torch::Tensor, torch::Tensor> rasterize_gaussians_filter(
    const torch::Tensor& means3D, const torch::Tensor& scales, const torch::Tensor& rotations, const float scale_modifier,
    const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix,
    const float tan_fovx, const float tan_fovy, const int image_height, const int image_width, const bool prefiltered,
    const bool debug) {
    if (means3D.ndimension() != 2 || means3D.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");
    TORCH_CHECK(means3D.is_cuda(), "means3D must be a HIP (cuda) tensor: street_gaussians_amd has no CPU path");
    c10::DeviceGuard guard(means3D.device());
    const int P = means3D.size(0);
    torch::Tensor radii = torch::zeros({P}, means3D.options().dtype(torch::kInt32));
    torch::Tensor means2D = torch::zeros({P, 2}, means3D.options().dtype(torch::kFloat32));
    if (P != 0) {
        auto c = [](const torch::Tensor& t) { return t.defined() && t.numel() ? t.contiguous() : t; };
        const torch::Tensor m = c(means3D), sc = c(scales), rot = c(rotations), cov = c(cov3D_precomp), vm = c(viewmatrix),
                            pm = c(projmatrix);
        check(sgr_visible_filter(P, image_width, image_height, fp(m), fp(sc), scale_modifier, fp(rot), fp(cov), fp(vm), fp(pm),
                                 tan_fovx, tan_fovy, prefiltered ? 1 : 0, radii.data_ptr<int>(), means2D.data_ptr<float>(),
                                 debug ? 1 : 0, stream_of(means3D)));
    }
    return std::make_tuple(radii, means2D);
}

// distCUDA2 (simple-knn/spatial.cu:16-26)
torch::Tensor distCUDA2(const torch::Tensor& points) {
    TORCH_CHECK(points.is_cuda(), "points must be a HIP (cuda) tensor: street_gaussians_amd has no CPU path");
    c10::DeviceGuard guard(points.device());
    const int P = points.size(0);
    torch::Tensor means = torch::zeros({P}, points.options().dtype(torch::kFloat32));
    if (P != 0) {
        const torch::Tensor pts = points.contiguous();
        TORCH_CHECK(pts.scalar_type() == torch::kFloat32, "points must be float32");
        torch::Tensor scratch = torch::empty({0}, points.options().dtype(torch::kByte));
        check(sgr_knn(P, pts.data_ptr<float>(), means.data_ptr<float>(), grow, &scratch, stream_of(points)));
        c10::hip::getCurrentHIPStream(points.device().index()).synchronize();  // scratch must outlive the kernels
    }
    return means;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("rasterize_gaussians", &rasterize_gaussians);
    m.def("rasterize_gaussians_backward", &rasterize_gaussians_backward);
    m.def("mark_visible", &mark_visible);
    m.def("rasterize_gaussians_filter", &rasterize_gaussians_filter);
    m.def("distCUDA2", &distCUDA2);
}
