"""ctypes loader of libsgr_hip.so (the C ABI declared in include/sgr.h).

The product path has NO fallback: if the library is missing, or a tensor is not on a HIP device,
this module raises -- it never routes to a CPU implementation (the oracle lives in oracle/ and is
test infrastructure only).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SGR_LIB: another BUILD of the same library (tools/gpu_ab.sh variants built with other -D switches); never a fallback
LIB_PATH = os.environ.get("SGR_LIB") or os.path.join(_HERE, "libsgr_hip.so")

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p)

_lib = None

# every symbol include/sgr.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "sgr_last_error", "sgr_version", "sgr_forward", "sgr_backward", "sgr_backward_ex", "sgr_mark_visible", "sgr_visible_filter",
    "sgr_knn", "sgr_geometry_bytes", "sgr_binning_bytes", "sgr_image_bytes", "sgr_partial_row_floats",
    "sgr_export_internal", "sgr_test_scan", "sgr_test_sort", "sgr_test_sort32", "sgr_test_sort_hist_words", "sgr_test_scan_tmp_words",
    "sgr_test_wave_sum", "sgr_test_exact_math", "sgr_test_lds_atomic_order", "sgr_test_switches", "sgr_has_variants", "sgr_set_lazy", "sgr_lazy_status", "sgr_profile_host_wait_us", "sgr_profile_enable", "sgr_profile_select", "sgr_profile_sample", "sgr_profile_read", "sgr_masked_color_grad",
    "sgr_sh_grad_from_views", "sgr_sh_grad_from_views_ex", "sgr_scene_compose_forward", "sgr_scene_compose_backward",
    "sgr_scene_densification_stats", "sgr_ssim_workspace_floats", "sgr_ssim_forward", "sgr_ssim_backward",
    "sgr_l1_workspace_floats", "sgr_l1_forward", "sgr_l1_backward", "sgr_color_loss_backward", "sgr_bce_forward", "sgr_bce_backward",
    "sgr_lidar_work_bytes", "sgr_lidar_depth_forward", "sgr_lidar_depth_backward", "sgr_densify_work_bytes", "sgr_densify_plan",
    "sgr_densify_map", "sgr_densify_gather", "sgr_densify_split_children", "sgr_densify_prune_mask",
    "sgr_densify_compact", "sgr_reset_opacity",
]


class SgrError(RuntimeError):
    pass


class SgrLazyError(SgrError):
    """-SGR_E_LAZY (lazy mode, sgr_set_lazy): the PREVIOUS lazy forward of this thread was invalid (list capacity overflow,
    prefilter violation, a depth beyond the narrow depth sort) -- ITS outputs and the gradients computed from them must be
    discarded (redo that step; do not apply its optimiser update).  The call that raised this rendered nothing; the thread's
    next forward is a blocking one that re-seeds the capacity.  Check ``_C.lazy_status()`` after a synchronisation and BEFORE
    applying gradients when a one-step-late report is not acceptable."""


SGR_E_LAZY = 5


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SgrError(f"{LIB_PATH} not found: build it with `python -m street_gaussians_amd.build` "
                           "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.sgr_last_error.restype = C.c_char_p
        L.sgr_version.restype = C.c_int
        L.sgr_profile_host_wait_us.restype = C.c_int
        L.sgr_profile_host_wait_us.argtypes = [C.c_int]
        for n in ("sgr_geometry_bytes", "sgr_binning_bytes", "sgr_test_sort_hist_words", "sgr_test_scan_tmp_words"):
            getattr(L, n).restype = C.c_size_t
        L.sgr_geometry_bytes.argtypes = [C.c_int]
        L.sgr_binning_bytes.argtypes = [C.c_int]
        L.sgr_image_bytes.restype = C.c_size_t
        L.sgr_image_bytes.argtypes = [C.c_int, C.c_int]
        L.sgr_test_sort_hist_words.argtypes = [C.c_uint32]
        L.sgr_test_scan_tmp_words.argtypes = [C.c_size_t]
        vp, f, i = C.c_void_p, C.c_float, C.c_int
        L.sgr_forward.restype = i
        L.sgr_forward.argtypes = [ALLOC_FN, vp, ALLOC_FN, vp, ALLOC_FN, vp, i, i, i, i, vp, i, i, vp, vp, vp, vp, vp, vp,
                                  f, vp, vp, vp, vp, vp, f, f, i, vp, vp, vp, vp, vp, i, vp]
        L.sgr_backward.restype = i
        L.sgr_backward.argtypes = [i, i, i, i, i, vp, i, i, vp, vp, vp, vp, vp, vp, f, vp, vp, vp, vp, vp, f, f, vp, vp,
                                   vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ALLOC_FN, vp, i, vp]
        L.sgr_backward_ex.restype = i
        L.sgr_backward_ex.argtypes = L.sgr_backward.argtypes + [vp]
        L.sgr_mark_visible.restype = i
        L.sgr_mark_visible.argtypes = [i, vp, vp, vp, vp, vp]
        L.sgr_visible_filter.restype = i
        L.sgr_visible_filter.argtypes = [i, i, i, vp, vp, f, vp, vp, vp, vp, f, f, i, vp, vp, i, vp]
        L.sgr_masked_color_grad.restype = i
        L.sgr_masked_color_grad.argtypes = [i, vp, vp, vp, vp]
        L.sgr_sh_grad_from_views.restype = i
        L.sgr_sh_grad_from_views.argtypes = [i, i, i, i, vp, vp, vp, vp, vp]
        L.sgr_sh_grad_from_views_ex.restype = i
        L.sgr_sh_grad_from_views_ex.argtypes = [i, i, i, i, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, vp, vp]
        L.sgr_scene_compose_forward.restype = i
        L.sgr_scene_compose_forward.argtypes = [i, vp, i, i, vp, vp, vp, vp, vp, vp, ALLOC_FN, vp, vp]
        L.sgr_scene_compose_backward.restype = i
        L.sgr_scene_compose_backward.argtypes = [i, vp, vp, i, i, vp, vp, vp, vp, vp, vp, ALLOC_FN, vp, vp]
        L.sgr_scene_densification_stats.restype = i
        L.sgr_scene_densification_stats.argtypes = [i, vp, vp, vp, ALLOC_FN, vp, vp]
        for n in ("sgr_ssim_workspace_floats", "sgr_l1_workspace_floats"):
            getattr(L, n).restype = C.c_size_t
            getattr(L, n).argtypes = [i, i, i]
        L.sgr_ssim_forward.restype = i
        L.sgr_ssim_forward.argtypes = [i, i, i, vp, vp, vp, vp, vp, vp, vp]
        L.sgr_ssim_backward.restype = i
        L.sgr_ssim_backward.argtypes = [i, i, i, vp, vp, vp, vp, vp, vp, vp]
        L.sgr_color_loss_backward.restype = i
        L.sgr_color_loss_backward.argtypes = [i, i, i, vp, vp, vp, vp, vp, f, f, vp, vp, vp]
        L.sgr_l1_forward.restype = i
        L.sgr_l1_forward.argtypes = [i, i, i, vp, vp, vp, vp, vp, vp]
        L.sgr_l1_backward.restype = i
        L.sgr_l1_backward.argtypes = [i, i, i, vp, vp, vp, vp, vp, vp, vp]
        L.sgr_bce_forward.restype = i
        L.sgr_bce_forward.argtypes = [i, i, vp, vp, vp, vp, vp]
        L.sgr_bce_backward.restype = i
        L.sgr_bce_backward.argtypes = [i, i, vp, vp, vp, vp, vp]
        L.sgr_lidar_work_bytes.restype = C.c_size_t
        L.sgr_lidar_work_bytes.argtypes = [i]
        L.sgr_lidar_depth_forward.restype = i
        L.sgr_lidar_depth_forward.argtypes = [i, vp, vp, vp, vp, C.c_double, vp, vp, vp]
        L.sgr_lidar_depth_backward.restype = i
        L.sgr_lidar_depth_backward.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.sgr_densify_work_bytes.restype = C.c_size_t
        L.sgr_densify_work_bytes.argtypes = [i]
        L.sgr_densify_plan.restype = i
        L.sgr_densify_plan.argtypes = [i, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_int64), vp]
        L.sgr_densify_map.restype = i
        L.sgr_densify_map.argtypes = [i, vp, vp, vp, vp, vp, vp]
        L.sgr_densify_gather.restype = i
        L.sgr_densify_gather.argtypes = [i, i, vp, vp, vp, i, vp, vp]
        L.sgr_densify_split_children.restype = i
        L.sgr_densify_split_children.argtypes = [i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.sgr_densify_prune_mask.restype = i
        L.sgr_densify_prune_mask.argtypes = [i, vp, i, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_int64), vp]
        L.sgr_densify_compact.restype = i
        L.sgr_densify_compact.argtypes = [i, vp, vp, vp, C.POINTER(C.c_int64), vp]
        L.sgr_reset_opacity.restype = i
        L.sgr_reset_opacity.argtypes = [i, vp, vp, vp, vp]
        L.sgr_knn.restype = i
        L.sgr_knn.argtypes = [i, vp, vp, ALLOC_FN, vp, vp]
        L.sgr_export_internal.restype = i
        L.sgr_export_internal.argtypes = [i, i, i, i, i, vp, vp, vp, vp, vp]
        L.sgr_test_scan.restype = i
        L.sgr_test_scan.argtypes = [vp, vp, C.c_size_t, i, vp, vp]
        L.sgr_test_sort.restype = i
        L.sgr_test_sort.argtypes = [vp, vp, vp, vp, C.c_uint32, i, vp, vp, vp]
        L.sgr_test_sort32.restype = i
        L.sgr_test_sort32.argtypes = [vp, vp, vp, vp, C.c_uint32, i, i, vp, vp, vp]
        L.sgr_test_wave_sum.restype = i
        L.sgr_test_wave_sum.argtypes = [vp, vp, vp, i, vp]
        L.sgr_test_lds_atomic_order.restype = i
        L.sgr_test_lds_atomic_order.argtypes = [vp, vp, i, vp]
        L.sgr_test_exact_math.restype = i
        L.sgr_test_exact_math.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, vp]
        L.sgr_test_switches.restype = i
        L.sgr_test_switches.argtypes = [i]
        L.sgr_profile_enable.restype = i
        L.sgr_profile_enable.argtypes = [i]
        L.sgr_profile_select.restype = i
        L.sgr_profile_select.argtypes = [i]
        L.sgr_profile_sample.restype = i
        L.sgr_profile_sample.argtypes = [i]
        L.sgr_profile_read.restype = i
        L.sgr_profile_read.argtypes = [vp, vp]
        L.sgr_partial_row_floats.restype = i
        L.sgr_partial_row_floats.argtypes = [i]
        _lib = L
    return _lib


def check(rc: int) -> int:
    if rc < 0:
        raise (SgrLazyError if rc == -SGR_E_LAZY else SgrError)(lib().sgr_last_error().decode())
    return rc
