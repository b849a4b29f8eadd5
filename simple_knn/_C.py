"""``simple_knn._C.distCUDA2`` (/root/reference/submodules/simple-knn/ext.cpp:15-17) on the HIP kernels."""
from street_gaussians_amd._C import distCUDA2  # noqa: F401
