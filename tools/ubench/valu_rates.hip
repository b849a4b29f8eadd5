// valu_rates.hip -- issue cost of the VALU instructions the blend backward is made of, on gfx950.
// Each kernel runs a long unrolled chain of ONE instruction type in every wave (8 waves / SIMD resident, all CUs), with 4
// independent chains per wave so that the wave's own dependency latency is not what is measured; the reported figure is
// SIMD cycles per wave-instruction = elapsed * clock * (CUs * 4 SIMDs) / (waves * instructions per wave).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

#define KERNEL(name, body)                                                                               \
    __global__ void __launch_bounds__(256) name(float* out, int iters) {                                \
        float a = threadIdx.x * 1e-3f + 1.0f, b = a + 0.5f, c = a + 0.25f, d = a + 0.125f;              \
        for (int i = 0; i < iters; i++) { asm volatile(REP16(body) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc", "scc"); } \
        out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;                                            \
    }

// 4 independent instructions per repetition -> 64 instructions per asm statement
KERNEL(k_fma, "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n")
KERNEL(k_mul, "v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3\n")
KERNEL(k_exp, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")
KERNEL(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n")
KERNEL(k_swap32, "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %1, %2\n v_permlane32_swap_b32 %3, %0\n")
KERNEL(k_swap16, "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %1, %2\n v_permlane16_swap_b32 %3, %0\n")
KERNEL(k_dpp_quad, "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_dpp_ror, "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_dpp_bcast, "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf\n")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0\n")
// packed f32 (two floats per lane in a VGPR pair): does one v_pk_* cost what one plain f32 instruction costs?
typedef float f2 __attribute__((ext_vector_type(2)));
#define KERNEL2(name, body)                                                                              \
    __global__ void __launch_bounds__(256) name(float* out, int iters) {                                \
        f2 a = {threadIdx.x * 1e-3f + 1.0f, 0.5f}, b = a + 0.5f, c = a + 0.25f, d = a + 0.125f;         \
        for (int i = 0; i < iters; i++) { asm volatile(REP16(body) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc", "scc"); } \
        out[blockIdx.x * 256 + threadIdx.x] = a.x + b.x + c.x + d.x + a.y + b.y + c.y + d.y;           \
    }
KERNEL2(k_pk_fma, "v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n")
KERNEL2(k_pk_mul, "v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3\n")
KERNEL2(k_pk_add, "v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3\n")
// the same cndmask without a register chain between the four (the first version's 23 cycles: chain or VCC read?)
KERNEL(k_cndmask2, "v_cndmask_b32 %0, %0, %0, vcc\n v_cndmask_b32 %1, %1, %1, vcc\n v_cndmask_b32 %2, %2, %2, vcc\n v_cndmask_b32 %3, %3, %3, vcc\n")
// cndmask with an initialised, lane-varying mask: in VCC (e32) and in another SGPR pair (e64)
__global__ void __launch_bounds__(256) k_cndmask_vcc(float* out, int iters) {
    float a = threadIdx.x * 1e-3f + 1.0f, b = a + 0.5f, c = a + 0.25f, d = a + 0.125f;
    asm volatile("v_cmp_gt_f32 vcc, 1.1, %0\n" : : "v"(a) : "vcc");
    for (int i = 0; i < iters; i++) {
        asm volatile(REP16("v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %2, %3, %0, vcc\n v_cndmask_b32 %3, %0, %1, vcc\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "scc");
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}
__global__ void __launch_bounds__(256) k_cndmask_sgpr(float* out, int iters) {
    float a = threadIdx.x * 1e-3f + 1.0f, b = a + 0.5f, c = a + 0.25f, d = a + 0.125f;
    unsigned long long m = __ballot(a < 1.1f);
    for (int i = 0; i < iters; i++) {
        asm volatile(REP16("v_cndmask_b32_e64 %0, %1, %2, %4\n v_cndmask_b32_e64 %1, %2, %3, %4\n v_cndmask_b32_e64 %2, %3, %0, %4\n v_cndmask_b32_e64 %3, %0, %1, %4\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(m) : "scc");
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}
KERNEL(k_add, "v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
// scalar side: s_nop and simple SALU, to see what a scalar instruction costs next to nothing else
__global__ void __launch_bounds__(256) k_snop(float* out, int iters) {
    for (int i = 0; i < iters; i++) asm volatile(REP64("s_nop 0\n") ::: "scc");
    out[blockIdx.x * 256 + threadIdx.x] = 1.f;
}
__global__ void __launch_bounds__(256) k_salu(float* out, int iters) {
    int s = iters;
    for (int i = 0; i < iters; i++) asm volatile(REP64("s_add_u32 %0, %0, 1\n") : "+s"(s) : : "scc");
    out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}
// dependent chains: ONE chain per wave (latency of back-to-back dependent issue)
__global__ void __launch_bounds__(256) k_fma_dep(float* out, int iters) {
    float a = threadIdx.x * 1e-3f + 1.0f;
    for (int i = 0; i < iters; i++) asm volatile(REP64("v_fma_f32 %0, %0, %0, %0\n") : "+v"(a));
    out[blockIdx.x * 256 + threadIdx.x] = a;
}
__global__ void __launch_bounds__(256) k_swap32_dep(float* out, int iters) {
    float a = threadIdx.x * 1e-3f + 1.0f, b = a + 1.f;
    for (int i = 0; i < iters; i++) asm volatile(REP64("v_permlane32_swap_b32 %0, %1\n") : "+v"(a), "+v"(b));
    out[blockIdx.x * 256 + threadIdx.x] = a + b;
}

template <typename F>
static void run(const char* name, F kernel, float* out, int waves_per_simd, int insts_per_iter, double clock_ghz) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
    const int iters = 400;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)waves_per_simd * iters * insts_per_iter;
    const double cyc = ms * 1e-3 * clock_ghz * 1e9 / insts_per_simd;
    printf("{\"inst\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"simd_cycles_per_wave_inst_at_%.1fGHz\": %.2f}\n", name,
           waves_per_simd, ms, clock_ghz, cyc);
    fflush(stdout);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 256 * 64 * sizeof(float));
    const double ghz = 2.4;
    for (int w : {8, 1}) {
        run("v_fma_f32", k_fma, out, w, 64, ghz);
        run("v_mul_f32", k_mul, out, w, 64, ghz);
        run("v_exp_f32", k_exp, out, w, 64, ghz);
        run("v_rcp_f32", k_rcp, out, w, 64, ghz);
        run("v_permlane32_swap", k_swap32, out, w, 64, ghz);
        run("v_permlane16_swap", k_swap16, out, w, 64, ghz);
        run("v_add_f32_dpp quad_perm", k_dpp_quad, out, w, 64, ghz);
        run("v_add_f32_dpp row_ror", k_dpp_ror, out, w, 64, ghz);
        run("v_add_f32_dpp row_bcast", k_dpp_bcast, out, w, 64, ghz);
        run("v_cndmask_b32", k_cndmask, out, w, 64, ghz);
        run("v_cmp_lt_f32", k_cmp, out, w, 64, ghz);
        run("v_pk_fma_f32", k_pk_fma, out, w, 64, ghz);
        run("v_pk_mul_f32", k_pk_mul, out, w, 64, ghz);
        run("v_pk_add_f32", k_pk_add, out, w, 64, ghz);
        run("v_cndmask_b32 no chain", k_cndmask2, out, w, 64, ghz);
        run("v_cndmask_b32 vcc initialised", k_cndmask_vcc, out, w, 64, ghz);
        run("v_cndmask_b32_e64 sgpr mask", k_cndmask_sgpr, out, w, 64, ghz);
        run("v_add_f32", k_add, out, w, 64, ghz);
        run("v_mov_b32_dpp quad_perm", k_mov_dpp, out, w, 64, ghz);
        run("s_nop 0", k_snop, out, w, 64, ghz);
        run("s_add_u32", k_salu, out, w, 64, ghz);
        run("v_fma_f32 dependent chain", k_fma_dep, out, w, 64, ghz);
        run("v_permlane32_swap dependent chain", k_swap32_dep, out, w, 64, ghz);
    }
    return 0;
}
