// TEST INFRASTRUCTURE. Shim so the reference's untouched CUDA sources compile with hipcc for gfx950
// (oracle/ref_build.sh).  Maps the handful of CUDA runtime names the reference uses onto HIP.
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <hip/hip_cooperative_groups.h>
#define cudaMemcpy hipMemcpy
#define cudaMemset hipMemset
#define cudaMalloc hipMalloc
#define cudaFree hipFree
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#ifndef __trap
#define __trap() __builtin_trap()
#endif
