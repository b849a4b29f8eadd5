// sgr_gauss_bwd_strict.hip -- the per-Gaussian backward (K12 + K13, sgr_gauss_bwd.hip) compiled a second time with FP
// contraction OFF, for the parity mode of the library (sgr_test_switches bit 7 / SGR_EXACT=1): the reference's strict
// build (oracle/_ref, -ffp-contract=off) rounds every product of computeCov2DCUDA / preprocessCUDA / computeCov3D
// (backward.cu:144-412) before it is added, and with the default contraction 13-37 of the 7 M dL/dscale / dL/drot elements
// of the full-size scenes sat outside rel 1e-4 of it.  Same source, same kernels under other names; the default mode keeps
// the contracted (faster) instantiation.
#pragma clang fp contract(off)
#define SGR_GB_STRICT 1
#define sgr_row_sum_kernel sgr_row_sum_kernel_strict
#define sgr_row_sum_wave_kernel sgr_row_sum_wave_kernel_strict  // (exists in -DSGR_WITH_VARIANTS=1 builds only)
#define sgr_gauss_bwd_kernel sgr_gauss_bwd_kernel_strict
#define sgr_launch_gauss_bwd sgr_launch_gauss_bwd_strict
#include "sgr_gauss_bwd.hip"
