"""Drop-in for the reference's ``diff_gaussian_rasterization`` package
(/root/reference/submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py):
``from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`` as used by
/root/reference/lib/utils/camera_utils.py:13, lib/models/gaussian_renderer.py:3 and
script/test_gaussian_rasterization.py:4 resolves to the MI355X-native implementation.
"""
from street_gaussians_amd import _C  # noqa: F401  (same attribute name as the reference's pybind module)
from street_gaussians_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                            _RasterizeGaussians, cpu_deep_copy_tuple, rasterize_gaussians)
