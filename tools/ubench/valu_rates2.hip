// valu_rates2.hip -- issue cost of the instruction classes the blend kernels are made of, on gfx950, in REAL cycles.
// Round-5 rewrite of valu_rates.hip (which timed 0.07-0.7 ms launches with HIP events and divided by a nominal 2.4 GHz):
//   * every wave brackets its loop with s_memtime (tick = shader cycle, MI355X_MICROARCH.md) -> cycles come from the GPU's own
//     counter, the clock of the run is reported next to them (memtime ticks / wall_clock64 ticks * 100 MHz);
//   * launches are sized to >= 20 ms (a first short launch calibrates the iteration count);
//   * 1, 2, 4 and 8 waves per SIMD;
//   * extra questions of round 5: does a wave64 VALU instruction get cheaper when one 32-lane half of EXEC is empty
//     (lanes = 32 / 16 / 1)?  what does a ds_add_f32 cost when k lanes hit the same address, and in which order does
//     the LDS add them (the sparse-visit path of sgr_blend_bwd.hip relies on a fixed lane order)?
// Reported: SIMD cycles per wave-instruction = median over waves of (t1 - t0) / (waves per SIMD * instructions per wave).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates2.hip -o /tmp/valu_rates2 && /tmp/valu_rates2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

struct Stamp { unsigned long long t0, t1, w0, w1; };
__device__ __forceinline__ unsigned long long memtime() {
    unsigned long long t;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
#define PROLOGUE                                                      \
    const unsigned long long w0 = wall_clock64(), t0 = memtime();
#define EPILOGUE                                                                                          \
    const unsigned long long t1 = memtime(), w1 = wall_clock64();                                         \
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t0, t1, w0, w1};

// `lanes` < 64: only the first `lanes` lanes of each wave run the loop (EXEC = their mask)
#define KERNEL(name, body)                                                                               \
    __global__ void __launch_bounds__(256) name(float* out, Stamp* st, int iters, int lanes) {          \
        float a = threadIdx.x * 1e-3f + 1.0f, b = a + 0.5f, c = a + 0.25f, d = a + 0.125f;              \
        PROLOGUE                                                                                         \
        if ((int)(threadIdx.x & 63) < lanes)                                                             \
            for (int i = 0; i < iters; i++) { asm volatile(REP16(body) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc", "scc"); } \
        EPILOGUE                                                                                         \
        out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;                                            \
    }

KERNEL(k_fma, "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n")
KERNEL(k_mul, "v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3\n")
KERNEL(k_add, "v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n")
KERNEL(k_exp, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")
KERNEL(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n")
KERNEL(k_ldexp, "v_ldexp_f32 %0, %0, 1\n v_ldexp_f32 %1, %1, 1\n v_ldexp_f32 %2, %2, 1\n v_ldexp_f32 %3, %3, 1\n")
KERNEL(k_rndne, "v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3\n")
KERNEL(k_swap32, "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %1, %2\n v_permlane32_swap_b32 %3, %0\n")
KERNEL(k_swap16, "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %1, %2\n v_permlane16_swap_b32 %3, %0\n")
KERNEL(k_dpp_quad, "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_dpp_ror, "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0\n")
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, -1, %0\n v_mbcnt_lo_u32_b32 %1, -1, %1\n v_mbcnt_lo_u32_b32 %2, -1, %2\n v_mbcnt_lo_u32_b32 %3, -1, %3\n")
typedef float f2 __attribute__((ext_vector_type(2)));
#define KERNEL2(name, body)                                                                              \
    __global__ void __launch_bounds__(256) name(float* out, Stamp* st, int iters, int lanes) {          \
        f2 a = {threadIdx.x * 1e-3f + 1.0f, 0.5f}, b = a + 0.5f, c = a + 0.25f, d = a + 0.125f;         \
        PROLOGUE                                                                                         \
        if ((int)(threadIdx.x & 63) < lanes)                                                             \
            for (int i = 0; i < iters; i++) { asm volatile(REP16(body) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc", "scc"); } \
        EPILOGUE                                                                                         \
        out[blockIdx.x * 256 + threadIdx.x] = a.x + b.x + c.x + d.x + a.y + b.y + c.y + d.y;           \
    }
KERNEL2(k_pk_fma, "v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n")
KERNEL2(k_pk_mul, "v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3\n")
__global__ void __launch_bounds__(256) k_cndmask_sgpr(float* out, Stamp* st, int iters, int lanes) {
    float a = threadIdx.x * 1e-3f + 1.0f, b = a + 0.5f, c = a + 0.25f, d = a + 0.125f;
    unsigned long long m = __ballot(a < 1.1f);
    PROLOGUE
    if ((int)(threadIdx.x & 63) < lanes)
        for (int i = 0; i < iters; i++) {
            asm volatile(REP16("v_cndmask_b32_e64 %0, %1, %2, %4\n v_cndmask_b32_e64 %1, %2, %3, %4\n v_cndmask_b32_e64 %2, %3, %0, %4\n v_cndmask_b32_e64 %3, %0, %1, %4\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(m) : "scc");
        }
    EPILOGUE
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}
__global__ void __launch_bounds__(256) k_cndmask_vcc(float* out, Stamp* st, int iters, int lanes) {
    float a = threadIdx.x * 1e-3f + 1.0f, b = a + 0.5f, c = a + 0.25f, d = a + 0.125f;
    PROLOGUE
    if ((int)(threadIdx.x & 63) < lanes) {
        asm volatile("v_cmp_gt_f32 vcc, 1.1, %0\n" : : "v"(a) : "vcc");
        for (int i = 0; i < iters; i++) {
            asm volatile(REP16("v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %2, %3, %0, vcc\n v_cndmask_b32 %3, %0, %1, vcc\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "scc");
        }
    }
    EPILOGUE
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}
__global__ void __launch_bounds__(256) k_snop(float* out, Stamp* st, int iters, int lanes) {
    PROLOGUE
    for (int i = 0; i < iters; i++) asm volatile(REP64("s_nop 0\n") ::: "scc");
    EPILOGUE
    out[blockIdx.x * 256 + threadIdx.x] = (float)lanes;
}
__global__ void __launch_bounds__(256) k_salu(float* out, Stamp* st, int iters, int lanes) {
    int s = iters;
    PROLOGUE
    for (int i = 0; i < iters; i++) asm volatile(REP64("s_add_u32 %0, %0, 1\n") : "+s"(s) : : "scc");
    EPILOGUE
    out[blockIdx.x * 256 + threadIdx.x] = (float)(s + lanes);
}
__global__ void __launch_bounds__(256) k_fma_dep(float* out, Stamp* st, int iters, int lanes) {
    float a = threadIdx.x * 1e-3f + 1.0f;
    PROLOGUE
    if ((int)(threadIdx.x & 63) < lanes)
        for (int i = 0; i < iters; i++) asm volatile(REP64("v_fma_f32 %0, %0, %0, %0\n") : "+v"(a));
    EPILOGUE
    out[blockIdx.x * 256 + threadIdx.x] = a;
}
// ds_add_f32 with `lanes` = number of DISTINCT addresses among the 64 lanes of a wave (64: conflict free, 1: all lanes on
// one word); every wave has its own 64 words
__global__ void __launch_bounds__(256) k_ds_add(float* out, Stamp* st, int iters, int distinct) {
    __shared__ float acc[256];
    acc[threadIdx.x] = 0.f;
    __syncthreads();
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)acc +
                          4u * ((threadIdx.x & ~63u) + ((threadIdx.x & 63u) % (unsigned)distinct));
    const float v = 1.0f;
    PROLOGUE
    for (int i = 0; i < iters; i++) asm volatile(REP64("ds_add_f32 %0, %1\n") : : "v"(addr), "v"(v) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    EPILOGUE
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = acc[threadIdx.x];
}
// LDS instructions with only the first `lanes` lanes of every wave active (distinct, consecutive addresses): what a 12-lane
// ds_add_f32 / ds_write_b32 of the blend backward costs the CU's LDS
#define LDS_KERNEL(name, body)                                                                           \
    __global__ void __launch_bounds__(256) name(float* out, Stamp* st, int iters, int lanes) {          \
        __shared__ float acc[256 * 4];                                                                   \
        for (int i = threadIdx.x; i < 1024; i += 256) acc[i] = 0.f;                                      \
        __syncthreads();                                                                                 \
        const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)acc + 16u * threadIdx.x; \
        float v = 1.0f, v1 = 2.0f, v2 = 3.0f, v3 = 4.0f;                                                 \
        PROLOGUE                                                                                         \
        if ((int)(threadIdx.x & 63) < lanes)                                                             \
            for (int i = 0; i < iters; i++) asm volatile(REP64(body) : "+v"(v), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(addr) : "memory"); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
        EPILOGUE                                                                                         \
        __syncthreads();                                                                                 \
        out[blockIdx.x * 256 + threadIdx.x] = acc[threadIdx.x] + v + v1 + v2 + v3;                      \
    }
LDS_KERNEL(k_lds_add, "ds_add_f32 %4, %0\n")
LDS_KERNEL(k_lds_w32, "ds_write_b32 %4, %0\n")
LDS_KERNEL(k_lds_r32, "ds_read_b32 %0, %4\n")
__global__ void __launch_bounds__(256) k_lds_w128(float* out, Stamp* st, int iters, int lanes) {
    __shared__ float acc[256 * 4];
    for (int i = threadIdx.x; i < 1024; i += 256) acc[i] = 0.f;
    __syncthreads();
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)acc + 16u * threadIdx.x;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v = {1.f, 2.f, 3.f, 4.f};
    PROLOGUE
    if ((int)(threadIdx.x & 63) < lanes)
        for (int i = 0; i < iters; i++) asm volatile(REP64("ds_write_b128 %0, %1\n") : : "v"(addr), "v"(v) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    EPILOGUE
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = acc[threadIdx.x];
}
// the sparse visit's LDS traffic as the kernel issues it: 11 ds_add_f32 of k lanes on the wave's entry, one ds_read_b32 +
// ds_write_b32 + ds_add_f32 by 12 lanes; lanes = k
__global__ void __launch_bounds__(256) k_sparse_visit(float* out, Stamp* st, int iters, int k) {
    __shared__ float acc[4 * 16 + 4 * 12 * 8];
    for (int i = threadIdx.x; i < 4 * 16 + 4 * 12 * 8; i += 256) acc[i] = 0.f;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)acc;
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned ea = base + 4u * (16u * wave), row = base + 4u * (64u + 96u * wave + (lane % 12u)), ra = ea + 4u * (lane % 12u);
    float v = 1.0f + lane, t = 0.f, z = 0.f;
    PROLOGUE
    for (int i = 0; i < iters; i++) {
        for (int r = 0; r < 16; r++) {
            if ((int)lane < k)
                asm volatile("ds_add_f32 %0, %1\n ds_add_f32 %0, %1 offset:4\n ds_add_f32 %0, %1 offset:8\n ds_add_f32 %0, %1 offset:12\n"
                             "ds_add_f32 %0, %1 offset:16\n ds_add_f32 %0, %1 offset:20\n ds_add_f32 %0, %1 offset:24\n ds_add_f32 %0, %1 offset:28\n"
                             "ds_add_f32 %0, %1 offset:32\n ds_add_f32 %0, %1 offset:36\n ds_add_f32 %0, %1 offset:40\n" : : "v"(ea), "v"(v) : "memory");
            if (lane < 12)
                asm volatile("ds_read_b32 %0, %1\n ds_write_b32 %1, %2\n s_waitcnt lgkmcnt(1)\n ds_add_f32 %3, %0\n" : "=&v"(t) : "v"(ra), "v"(z), "v"(row) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    EPILOGUE
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = acc[threadIdx.x % (4 * 16)] + t;
}
// order of the LDS's same-address additions inside ONE ds_add_f32: lane i adds x_i (magnitudes mixed so that the order
// shows in the rounding); out[wave] = the word.  The host compares with the ascending-lane sequential sum.
__global__ void __launch_bounds__(64) k_ds_order(const float* x, float* out, int k) {
    __shared__ float acc[1];
    if (threadIdx.x == 0) acc[0] = 0.f;
    __syncthreads();
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)acc;
    const float v = x[blockIdx.x * 64 + threadIdx.x];
    if ((int)threadIdx.x < k) asm volatile("ds_add_f32 %0, %1\n s_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(v) : "memory");
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[0];
}

static int g_cus = 256;
template <typename F>
static void run(const char* name, F kernel, float* out, Stamp* st, int waves_per_simd, int insts_per_iter, int lanes = 64,
                const char* unit = "simd_cycles_per_wave_inst", int per_cu = 0) {
    const int blocks = g_cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
    int iters = 200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int pass = 0; pass < 2; pass++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, st, iters, lanes);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (pass == 0) iters = std::max(200, (int)(iters * 22.0 / std::max(ms, 0.01f)));  // >= 20 ms
    }
    std::vector<Stamp> h(blocks * 4);
    hipMemcpy(h.data(), st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> cyc, ghz;
    for (auto& s : h) {
        cyc.push_back((double)(s.t1 - s.t0));
        if (s.w1 > s.w0) ghz.push_back((double)(s.t1 - s.t0) / (double)(s.w1 - s.w0) * 0.1);  // wall_clock64: 100 MHz
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(ghz.begin(), ghz.end());
    const double med = cyc[cyc.size() / 2];
    const double n_inst = (double)iters * insts_per_iter;
    // per SIMD: its waves_per_simd waves share the pipe; per CU (LDS): 4 * waves_per_simd waves share the LDS
    const double per = med / ((per_cu ? 4.0 : 1.0) * waves_per_simd * n_inst);
    // the same from the launch's wall time at the clock the waves measured: total pipe cycles / total wave-instructions (does
    // not assume that all resident waves ran for the whole launch)
    const double clk = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
    const double per_wall = ms * 1e-3 * clk * 1e9 / ((per_cu ? 4.0 : 1.0) * waves_per_simd * n_inst);
    printf("{\"inst\": \"%s\", \"waves_per_simd\": %d, \"lanes\": %d, \"ms\": %.2f, \"iters\": %d, \"median_wave_cycles\": %.0f, "
           "\"%s\": %.3f, \"%s_from_wall_time\": %.3f, \"clock_ghz\": %.3f, \"wave_active_frac_of_launch\": %.3f}\n",
           name, waves_per_simd, lanes, ms, iters, med, unit, per, unit, per_wall, clk, clk > 0 ? med / (ms * 1e6 * clk) : 0.0);
    fflush(stdout);
}

int main(int argc, char** argv) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    g_cus = p.multiProcessorCount;
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    float* out;
    Stamp* st;
    hipMalloc(&out, 256 * 256 * 64 * sizeof(float));
    hipMalloc(&st, 256 * 8 * 4 * sizeof(Stamp));
    for (int w : {8, 4, 2, 1}) {
        if (quick && w != 8 && w != 1) continue;
        run("v_fma_f32", k_fma, out, st, w, 64);
        run("v_mul_f32", k_mul, out, st, w, 64);
        run("v_add_f32", k_add, out, st, w, 64);
        run("v_exp_f32", k_exp, out, st, w, 64);
        run("v_rcp_f32", k_rcp, out, st, w, 64);
        run("v_ldexp_f32", k_ldexp, out, st, w, 64);
        run("v_rndne_f32", k_rndne, out, st, w, 64);
        run("v_permlane32_swap", k_swap32, out, st, w, 64);
        run("v_permlane16_swap", k_swap16, out, st, w, 64);
        run("v_add_f32_dpp quad_perm", k_dpp_quad, out, st, w, 64);
        run("v_add_f32_dpp row_ror", k_dpp_ror, out, st, w, 64);
        run("v_cmp_lt_f32", k_cmp, out, st, w, 64);
        run("v_mbcnt_lo_u32_b32", k_mbcnt, out, st, w, 64);
        run("v_pk_fma_f32", k_pk_fma, out, st, w, 64);
        run("v_pk_mul_f32", k_pk_mul, out, st, w, 64);
        run("v_cndmask_b32 vcc", k_cndmask_vcc, out, st, w, 64);
        run("v_cndmask_b32_e64 sgpr mask", k_cndmask_sgpr, out, st, w, 64);
        run("s_nop 0", k_snop, out, st, w, 64);
        run("s_add_u32", k_salu, out, st, w, 64);
        run("v_fma_f32 dependent chain", k_fma_dep, out, st, w, 64);
        if (w == 8 || w == 1) {
            // EXEC with an empty 32-lane half: is the instruction cheaper?
            for (int lanes : {32, 16, 1}) {
                run("v_fma_f32 partial EXEC", k_fma, out, st, w, 64, lanes);
                run("v_exp_f32 partial EXEC", k_exp, out, st, w, 64, lanes);
                run("v_permlane32_swap partial EXEC", k_swap32, out, st, w, 64, lanes);
                run("v_add_f32_dpp quad_perm partial EXEC", k_dpp_quad, out, st, w, 64, lanes);
            }
            // LDS atomics: `lanes` = distinct addresses among the wave's 64 lanes; cycles per wave-instruction of the CU's LDS
            for (int distinct : {64, 32, 16, 8, 4, 2, 1})
                run("ds_add_f32 (lanes = distinct addresses)", k_ds_add, out, st, w, 64, distinct, "cu_lds_cycles_per_wave_inst", 1);
            for (int lanes : {64, 16, 12, 4, 1}) {
                run("ds_add_f32, `lanes` active lanes", k_lds_add, out, st, w, 64, lanes, "cu_lds_cycles_per_wave_inst", 1);
                run("ds_write_b32, `lanes` active lanes", k_lds_w32, out, st, w, 64, lanes, "cu_lds_cycles_per_wave_inst", 1);
                run("ds_write_b128, `lanes` active lanes", k_lds_w128, out, st, w, 64, lanes, "cu_lds_cycles_per_wave_inst", 1);
                run("ds_read_b32, `lanes` active lanes", k_lds_r32, out, st, w, 64, lanes, "cu_lds_cycles_per_wave_inst", 1);
            }
            for (int k : {1, 2, 4, 8, 16, 32})
                run("sparse visit LDS sequence (lanes = hit lanes k; 14 DS instructions each)", k_sparse_visit, out, st, w, 16, k,
                    "cu_lds_cycles_per_visit", 1);
        }
    }
    // order of same-address additions inside one ds_add_f32
    {
        const int nb = 4096;
        std::vector<float> x(nb * 64), ref(nb);
        unsigned s = 12345u;
        for (auto& v : x) {
            s = s * 1664525u + 1013904223u;
            const float m = (float)((s >> 8) & 0xffff) / 65536.0f + 0.5f;
            v = ((s >> 28) & 1) ? m * 1.0e6f : m;
            if ((s >> 30) & 1) v = -v;
        }
        float *dx, *dout;
        hipMalloc(&dx, x.size() * 4);
        hipMalloc(&dout, nb * 4);
        hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
        for (int k : {2, 4, 8, 16, 64}) {
            std::vector<float> first(nb), got(nb);
            int asc = 0, desc = 0, stable = 1;
            for (int rep = 0; rep < 50; rep++) {
                hipLaunchKernelGGL(k_ds_order, dim3(nb), dim3(64), 0, 0, dx, dout, k);
                hipMemcpy(got.data(), dout, nb * 4, hipMemcpyDeviceToHost);
                if (rep == 0) first = got;
                else if (memcmp(first.data(), got.data(), nb * 4)) stable = 0;
            }
            for (int b = 0; b < nb; b++) {
                volatile float a = 0.f, d = 0.f;
                for (int i = 0; i < k; i++) a = a + x[b * 64 + i];
                for (int i = k - 1; i >= 0; i--) d = d + x[b * 64 + i];
                asc += (memcmp((const void*)&a, &first[b], 4) == 0);
                desc += (memcmp((const void*)&d, &first[b], 4) == 0);
            }
            printf("{\"test\": \"ds_add_f32 same-address order\", \"lanes\": %d, \"words\": %d, \"matches_ascending_lane_order\": %d, "
                   "\"matches_descending\": %d, \"identical_over_50_runs\": %s}\n", k, nb, asc, desc, stable ? "true" : "false");
        }
    }
    return 0;
}
