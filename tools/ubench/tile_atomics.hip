// tile_atomics.hip -- what device-scope atomics on per-tile counters cost on MI355X, for the tile-binned front end
// (count per tile + slot inside the tile by a returning atomicAdd, DESIGN.md section 3 "binning").
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/tile_atomics tools/ubench/tile_atomics.hip && tools/ubench/tile_atomics
//
// Instances are laid out Gaussian-major like the rows of the backward (u = u0[g] + j): the instances of one Gaussian are
// the tiles of a small rect around a random centre, so neighbouring lanes hit neighbouring counters.  One JSON line per
// (kernel, R): microseconds per launch (HIP events, mean of 20 after 3 warm-ups) and atomics per microsecond.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_copy(const uint32_t* __restrict__ tile, uint32_t* __restrict__ slot, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) slot[i] = tile[i];
}
__global__ void __launch_bounds__(256) k_add_noret(const uint32_t* __restrict__ tile, uint32_t* __restrict__ cnt, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) atomicAdd(&cnt[tile[i]], 1u);
}
__global__ void __launch_bounds__(256) k_add_ret(const uint32_t* __restrict__ tile, uint32_t* __restrict__ cnt,
                                                 uint32_t* __restrict__ slot, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) slot[i] = atomicAdd(&cnt[tile[i]], 1u);
}
// counters sharded by XCD (workgroup b runs on XCD b % 8): 8 copies of the table, no cross-XCD contention on a word
__global__ void __launch_bounds__(256) k_add_ret_xcd(const uint32_t* __restrict__ tile, uint32_t* __restrict__ cnt,
                                                     uint32_t* __restrict__ slot, uint32_t n, uint32_t T) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) slot[i] = atomicAdd(&cnt[(blockIdx.x & 7u) * T + tile[i]], 1u);
}
// scatter of 8-byte pairs to start[tile] + slot (the second half of the binning)
__global__ void __launch_bounds__(256) k_scatter(const uint32_t* __restrict__ tile, const uint32_t* __restrict__ slot,
                                                 const uint32_t* __restrict__ start, uint2* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[start[tile[i]] + slot[i]] = make_uint2(i, i ^ 0x5555u);
}

int main() {
    const uint32_t T = 9600, gx = 120, gy = 80;
    for (uint32_t R_target : {4800000u, 7800000u, 28000000u}) {
        std::mt19937 rng(1);
        std::vector<uint32_t> tile;
        tile.reserve(R_target + 4096);
        std::uniform_int_distribution<int> cx(0, gx - 1), cy(0, gy - 1), ext(1, 3);
        while (tile.size() < R_target) {
            const int x0 = cx(rng), y0 = cy(rng), w = ext(rng), h = ext(rng);
            for (int y = y0; y < y0 + h && y < (int)gy; y++)
                for (int x = x0; x < x0 + w && x < (int)gx; x++) tile.push_back((uint32_t)(y * gx + x));
        }
        const uint32_t n = (uint32_t)tile.size();
        std::vector<uint32_t> count(T, 0), start(T + 1, 0);
        for (uint32_t t : tile) count[t]++;
        for (uint32_t t = 0; t < T; t++) start[t + 1] = start[t] + count[t];
        uint32_t *d_tile, *d_cnt, *d_slot, *d_start;
        uint2* d_out;
        CHECK(hipMalloc(&d_tile, n * 4));
        CHECK(hipMalloc(&d_cnt, 8 * T * 4));
        CHECK(hipMalloc(&d_slot, n * 4));
        CHECK(hipMalloc(&d_start, (T + 1) * 4));
        CHECK(hipMalloc(&d_out, (size_t)n * 8));
        CHECK(hipMemcpy(d_tile, tile.data(), n * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_start, start.data(), (T + 1) * 4, hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        const uint32_t nb = (n + 255) / 256;
        auto run = [&](const char* name, int which) {
            float total = 0.f;
            for (int rep = 0; rep < 23; rep++) {
                CHECK(hipMemsetAsync(d_cnt, 0, 8 * T * 4, 0));
                CHECK(hipEventRecord(e0, 0));
                switch (which) {
                    case 0: k_copy<<<nb, 256>>>(d_tile, d_slot, n); break;
                    case 1: k_add_noret<<<nb, 256>>>(d_tile, d_cnt, n); break;
                    case 2: k_add_ret<<<nb, 256>>>(d_tile, d_cnt, d_slot, n); break;
                    case 3: k_add_ret_xcd<<<nb, 256>>>(d_tile, d_cnt, d_slot, n, T); break;
                    case 4: k_scatter<<<nb, 256>>>(d_tile, d_slot, d_start, d_out, n); break;
                }
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 3) total += ms;
            }
            const double us = total / 20.0 * 1000.0;
            printf("{\"kernel\": \"%s\", \"R\": %u, \"us\": %.2f, \"per_us\": %.0f}\n", name, n, us, n / us);
            fflush(stdout);
        };
        run("copy_4B", 0);
        run("atomic_add_noret", 1);
        run("atomic_add_ret", 2);
        // leave valid slots for the scatter: one clean returning pass
        CHECK(hipMemset(d_cnt, 0, 8 * T * 4));
        k_add_ret<<<nb, 256>>>(d_tile, d_cnt, d_slot, n);
        CHECK(hipDeviceSynchronize());
        run("scatter_8B", 4);
        run("atomic_add_ret_xcd_sharded", 3);
        CHECK(hipFree(d_tile)); CHECK(hipFree(d_cnt)); CHECK(hipFree(d_slot)); CHECK(hipFree(d_start)); CHECK(hipFree(d_out));
    }
    return 0;
}
