// sgr_api.hip -- host orchestration + C ABI (include/sgr.h) of the MI355X-native rasterizer.
// Mirrors CudaRasterizer::Rasterizer::{forward,backward,markVisible,visible_filter}
// (/root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer/rasterizer_impl.cu:141-506).
// No torch types; all launches go to the caller's HIP stream.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>

#include "../../include/sgr.h"
#include "sgr_math.h"

// launchers implemented in the kernel translation units
void sgr_launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s);
void sgr_launch_preprocess(int P, int D, int M, const float* means3D, const float* scales, const float* rotations,
                           const float* opacities, const float* shs, const float* cov3D_precomp,
                           const float* colors_precomp, const SgrCam* cam, const SgrGeomView& gv, int* radii,
                           int prefiltered, bool stage_sh, int tight, hipStream_t s);
void sgr_launch_filter(int P, const float* means3D, const float* scales, const float* rotations,
                       const float* cov3D_precomp, const SgrCamArgs& ca, const SgrGeomView& gv, int* radii,
                       float* means2D, int prefiltered, hipStream_t s);
void sgr_launch_duplicate(int P, const SgrGeomView& gv, const uint32_t* order, const uint32_t* bsum, const uint32_t* sub, void* keys,
                          int key16, uint32_t* vals, int gx, uint32_t cap, int marks, hipStream_t s);
void sgr_launch_tile_ranges(int L, const void* keys, int key16, uint2* ranges, uint8_t* touched, uint32_t T, hipStream_t s);
void sgr_launch_tile_order(const uint2* ranges, int T, int force, hipStream_t s);
void sgr_launch_compose_keys(int L, const void* tile_keys, int key16, const uint32_t* point_list, const float4* rec, uint64_t* out,
                             hipStream_t s);
void sgr_launch_blend_fwd(bool cull, bool exact, int gx, int gy, const uint2* ranges, const uint32_t* point_list, int W, int H,
                          int S, const float4* rec, const float* semantics, const float* bg, float* out_color,
                          float* out_depth, float* out_alpha, float* out_semantic, uint32_t* n_contrib, uint8_t* hit4,
                          uint32_t* hlist, uint32_t* n_contrib_k, hipStream_t s);
int sgr_partial_row_stride(int S);
void sgr_launch_blend_bwd(bool cull, bool dpp, bool det, bool v2, bool exact, int gx, int gy, const uint2* ranges, const uint32_t* point_list, int W,
                          int H, int S, const float* bg, const float4* rec, const uint32_t* u0, const uint64_t* tmask, const float* semantics, const float* alphas,
                          const uint32_t* n_contrib, const uint8_t* hit4, const float* dL_dpix, const float* dL_ddepth,
                          const float* dL_dalpha, const float* dL_dsem, float* partials, uint8_t* touched, uint32_t row_limit,
                          const uint32_t* hlist, const uint32_t* n_contrib_k, const uint32_t* hl_flag, hipStream_t s);
int sgr_launch_gauss_bwd(int P, int D, int M, int S, const float* means3D, const int* radii, const float* shs,
                          const float* scales, const float* rotations, const float* cov3D_precomp, const SgrCam* cam,
                          const SgrGeomView& gv, const float* partials, int row_stride, const uint8_t* touched,
                          float4* cd, float* dL_dmean2D, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                          float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dsemantic,
                          const SgrStatSink& sink, int quad, int exact, int W, int H, hipEvent_t after_rows, int rs_wave,
                          uint32_t row_limit, float* masked_color_out, int skip_sh, hipStream_t s);
void sgr_launch_blend_bwd_sw(bool exact, int gx, int gy, const uint2* ranges, const uint32_t* point_list, int W, int H,
                             const float* bg, const float4* rec, const uint32_t* u0, const uint64_t* tmask, const float* alphas,
                             const uint32_t* n_contrib, const uint8_t* hit4, const float* dL_dpix, const float* dL_ddepth,
                             const float* dL_dalpha, float* partials, int row_stride, uint8_t* touched, hipStream_t s);
// the same compiled with FP contraction off (sgr_gauss_bwd_strict.hip): parity mode
int sgr_launch_gauss_bwd_strict(int P, int D, int M, int S, const float* means3D, const int* radii, const float* shs,
                                 const float* scales, const float* rotations, const float* cov3D_precomp, const SgrCam* cam,
                                 const SgrGeomView& gv, const float* partials, int row_stride, const uint8_t* touched,
                                 float4* cd, float* dL_dmean2D, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                                 float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, float* dL_dsemantic,
                                 const SgrStatSink& sink, int quad, int exact, int W, int H, hipEvent_t after_rows, int rs_wave,
                                 uint32_t row_limit, float* masked_color_out, int skip_sh, hipStream_t s);
void sgr_launch_masked_color_grad(int P, const uint32_t* clamped, const float* dL_dcolor, float* out, hipStream_t s);
void sgr_launch_sh_grad_from_views(int P, int D, int M, int V, const float* means3D, size_t means_stride,
                                   const float* campos, size_t campos_stride, const float* drgb, size_t drgb_stride,
                                   float* dL_dsh, hipStream_t s);
void sgr_launch_wave_sum_test(const float* in, float* out_dpp, float* out_shfl, int nwaves, hipStream_t s);
void sgr_launch_lds_atomic_order_test(const uint32_t* pattern, uint32_t* out, int trials, hipStream_t s);
int sgr_knn_impl(int P, const float* points, float* meanDists, sgr_alloc_fn scratch, void* scratch_user, hipStream_t s,
                 std::string& err);

static thread_local std::string g_err;

static bool env_flag(const char* name) {
    const char* v = getenv(name);
    return v && v[0] && v[0] != '0';
}

// (bit 6: the preprocess stages its SH rows through LDS -- an A/B that measured slower; bit 7: EXACT parity mode of the
// blend kernels, sgr_math.h sgr_power_ref: the reference's power expression + accurate expf + true division)
// A/B switches of the blend kernels (tests and tools/gpu_ab.sh): bit 0 no quadrant cull, 1 no DPP reduction,
// 2 no deterministic LDS combine, 3 backward ignores the forward's hit record, 4 the S = 0 backward runs the
// transposed-accumulation kernel (an A/B design that measured slower than the cross-lane reduction), 5 the
// radix sorts run in their one-sweep form (an A/B design that measured slower, sgr_scan_sort.hip; SGR_ONESWEEP).  Process-wide, set through
// sgr_test_switches(); the environment (SGR_NO_CULL / SGR_NO_DPP / SGR_NO_DET / SGR_NO_HITS / SGR_V2) only provides the
// initial value, read ONCE -- the per-step path is one relaxed atomic load, no getenv.
static std::atomic<int> g_switches{-1};
#ifndef SGR_WITH_VARIANTS
#define SGR_WITH_VARIANTS 0  // 1: the library also holds the designs that were measured slower (tools/build_variant.py)
#endif
// switch bits that select one of those designs: transposed backward (4), one-sweep sorts (5), scalar-walk backward (8),
// wave-cooperative row sum (9) -- ignored by a build that does not contain them
#define SGR_VARIANT_BITS (16 | 32 | 256 | 512)
static int switches() {
    int v = g_switches.load(std::memory_order_relaxed);
    if (v < 0) {
        v = (env_flag("SGR_NO_CULL") ? 1 : 0) | (env_flag("SGR_NO_DPP") ? 2 : 0) | (env_flag("SGR_NO_DET") ? 4 : 0) |
            (env_flag("SGR_NO_HITS") ? 8 : 0) | (env_flag("SGR_V2") ? 16 : 0) | (env_flag("SGR_PRE_STAGE") ? 64 : 0) |
            (env_flag("SGR_EXACT") ? 128 : 0) | ((env_flag("SGR_SW8") || env_flag("SGR_SW")) ? 256 : 0) | ((env_flag("SGR_SW9") || env_flag("SGR_RS_WAVE")) ? 512 : 0) |
            (env_flag("SGR_REF_RECT") ? 1024 : 0) | (env_flag("SGR_NO_TILE_MASK") ? 2048 : 0) | (env_flag("SGR_TILE_SORT") ? 4096 : 0) |
            (env_flag("SGR_REF_RECT_PLAIN") ? 8192 : 0) | (env_flag("SGR_LPT") ? 16384 : 0) |
            (env_flag("SGR_NO_LPT") ? 32768 : 0) | (env_flag("SGR_NO_HLIST") ? 65536 : 0) | (env_flag("SGR_HLIST_ALWAYS") ? 131072 : 0) |
            (env_flag("SGR_KEY32") ? 262144 : 0);
        if (!SGR_WITH_VARIANTS) v &= ~SGR_VARIANT_BITS;
        g_switches.store(v, std::memory_order_relaxed);
    }
    return v;
}

// ---- optional per-stage timing with HIP events on the caller's stream (sgr_profile_*) -------------------
// stages: 0 preprocess(+camera pack, memsets) 1 depth sort + scan 2 duplicate 3 tile sort 4 tile_ranges 5 blend_fwd
//         6 partials memset 7 blend_bwd 8 gauss_bwd
#define SGR_PROF_STAGES 9
#define SGR_PROF_SLOTS 1024
// One process-wide recorder: the forward runs on the caller's thread and the backward on autograd's, and both must land
// in the same recording.  Slots are claimed with an atomic counter (each thread records only into slots it claimed);
// enable / read are called by the measuring thread while no step is in flight.
struct SgrProf {
    std::atomic<bool> on{false};
    std::atomic<int> mask{(1 << SGR_PROF_STAGES) - 1};  // stages that are recorded (sgr_profile_select)
    std::atomic<int> n{0};  // claimed (begin, end) slots
    std::atomic<int> every{1};  // record every k-th occurrence of a stage (sgr_profile_sample)
    std::atomic<unsigned> seen[SGR_PROF_STAGES];
    hipEvent_t ev[SGR_PROF_SLOTS][2];
    int stage[SGR_PROF_SLOTS];
    bool created = false;
};
static SgrProf g_prof;
static thread_local int t_prof_slot = -1;  // slot opened by prof_begin on this thread
static void prof_begin(int stage, hipStream_t s) {
    t_prof_slot = -1;
    if (!g_prof.on.load(std::memory_order_relaxed)) return;
    if (!((g_prof.mask.load(std::memory_order_relaxed) >> stage) & 1)) return;
    const int every = g_prof.every.load(std::memory_order_relaxed);
    if (every > 1 && g_prof.seen[stage].fetch_add(1u, std::memory_order_relaxed) % (unsigned)every != 0) return;
    const int i = g_prof.n.fetch_add(1, std::memory_order_relaxed);
    if (i >= SGR_PROF_SLOTS) return;
    g_prof.stage[i] = stage;
    (void)hipEventRecord(g_prof.ev[i][0], s);
    t_prof_slot = i;
}
static void prof_end(hipStream_t s) {
    if (t_prof_slot < 0) return;
    (void)hipEventRecord(g_prof.ev[t_prof_slot][1], s);
    t_prof_slot = -1;
}

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return -code;
}
int sgr_set_error(int code, const std::string& msg) { return fail(code, msg); }  // for the other translation units

#define SGR_HIP(call)                                                                            \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess)                                                                   \
            return fail(SGR_E_HIP, std::string(#call) + ": " + hipGetErrorString(e__));          \
    } while (0)

// reference CHECK_CUDA (auxiliary.h:166-173): with debug, synchronise after the stage and report
#define SGR_STAGE(name)                                                                                   \
    do {                                                                                                  \
        if (debug && env_flag("SGR_TRACE")) { fprintf(stderr, "[sgr] stage %s launched\n", name); fflush(stderr); } \
        hipError_t e__ = hipGetLastError();                                                               \
        if (e__ == hipSuccess && debug) e__ = hipStreamSynchronize(stream);                               \
        if (e__ != hipSuccess) return fail(SGR_E_HIP, std::string("stage ") + name + ": " + hipGetErrorString(e__)); \
    } while (0)

// per-thread pinned landing zones for the forward's one device->host readback.  Slot 0: the blocking forward and
// sgr_visible_filter(prefiltered = 1), both of which wait for their copy before they return.  Slot 1: the LAZY forward, whose
// copy is looked at one call later -- its own slot, so that a filter call or a blocking forward in between cannot overwrite
// words a pending late check still has to read.
static uint32_t* pinned_pair(int slot = 0) {
    static thread_local uint32_t* p[2] = {nullptr, nullptr};
    // coherent (uncached on the device side): the host watches these words while the copy is in flight
    if (!p[slot] && hipHostMalloc((void**)&p[slot], 64, hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess) p[slot] = nullptr;
    return p[slot];
}
// ... and the event that marks "the readback has landed" while later kernels are already queued behind it
static hipEvent_t readback_event(int slot = 0) {
    static thread_local hipEvent_t ev[2][64] = {};  // events belong to a device: one per (thread, slot, device)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!ev[slot][dev] && hipEventCreateWithFlags(&ev[slot][dev], hipEventDisableTiming) != hipSuccess) ev[slot][dev] = nullptr;
    return ev[slot][dev];
}

// The forward's one host wait.  hipEventSynchronize() parks the thread on an interrupt, and how long it takes to come
// back is a property of the host (tens of microseconds on some boxes) -- time in which the GPU works off the ~0.12 ms of
// depth sort + scan queued behind the read-back and then idles until the rest of the forward is queued.  So the thread
// watches the pinned landing zone itself: its words were set to a value the device never writes (flags are 0 / 1,
// num_rendered < 2^31).  Bounded: after SGR_SPIN_US microseconds (default 2 ms -- the copy lands within ~0.2 ms of being
// queued; 0 = never spin) it falls back to the
// event, which is also what orders everything else behind the copy.
#define SGR_READBACK_PENDING 0xffffffffu
// nanoseconds the calling threads have spent in the forward's host wait (process-wide, sgr_profile_host_wait_us): what a
// measurement of "host time per step" has to subtract -- the wait is the GPU's time, not the host's
static std::atomic<unsigned long long> g_wait_ns{0};
static hipError_t wait_for_readback_(uint32_t* host_vals, hipEvent_t landed);
static hipError_t wait_for_readback(uint32_t* host_vals, hipEvent_t landed) {
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = wait_for_readback_(host_vals, landed);
    g_wait_ns.fetch_add((unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(),
                        std::memory_order_relaxed);
    return e;
}
static hipError_t wait_for_readback_(uint32_t* host_vals, hipEvent_t landed) {
    static const long spin_us = [] { const char* e = getenv("SGR_SPIN_US"); return e ? atol(e) : 2000L; }();
    if (spin_us > 0) {
        volatile uint32_t* hv = host_vals;
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            for (int i = 0; i < 256; i++) {
                if (hv[0] != SGR_READBACK_PENDING && hv[2] != SGR_READBACK_PENDING && hv[4] != SGR_READBACK_PENDING &&
                    hv[5] != SGR_READBACK_PENDING) {
                    std::atomic_thread_fence(std::memory_order_acquire);
                    return hipSuccess;
                }
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#else
                std::this_thread::yield();
#endif
            }
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us)
                break;
        }
    }
    return hipEventSynchronize(landed);
}

static bool tile_sort_on() { return (switches() & 4096) != 0; }
// Tile rects of a frame (sgr_preprocess.hip `tight`): 2 = the reference's rect cut down to the bounding box of the tiles where
// alpha >= 1/255 is possible + a tile mask inside it (default), 1 = the box alone (switch bit 11), 3 = switch bit 10
// (SGR_REF_RECT): the reference's rects, i.e. its lists entry for entry, with the instances outside box / mask MARKED dead
// (they are sorted and counted like the reference's, but the blend kernels skip them and they own no gradient row), 0 = bits
// 10 + 13 (SGR_REF_RECT_PLAIN): the reference's rects without marks (round-5 form of the strict mode, A/B).
static int rect_mode() {
    const int sw = switches();
    if (sw & 1024) return (sw & 8192) ? 0 : 3;
    return (sw & 2048) ? 1 : 2;
}
// The compact hit list (SgrBinView::hlist) is written by the forward and walked by the blend backward where it pays: with the
// reference's rects (switch bit 10) 40 % of a list cannot blend and the backward takes that many fewer rounds (-24 us at the
// bench size against +10 us in the forward); with the cut-down rects the lists hold next to nothing to skip and the forward's
// 10 us buy nothing (measured).  Bit 16 turns it off, bit 17 on in every mode (A/B and tests).
static bool hit_list_on() {
    const int sw = switches();
    return SGR_HLIST && !(sw & 65536) && ((sw & 1024) || (sw & 131072));
}
// The tile keys of the instance list are 16-bit whenever the frame has fewer than 65535 tiles (the all-ones key is the padding of
// the lazy mode): the tile sort moves 6 instead of 8 bytes per pair and pass and its histogram reads half.  Bit 18 (SGR_KEY32=1)
// keeps 32-bit keys (A/B; frames with more tiles -- beyond 4096 x 4080 pixels -- take them anyway).  Same buffers either way.
static int tile_key16(size_t T) { return (T < 65535u && !(switches() & 262144)) ? 1 : 0; }
// depth pre-sort: three passes of 9-bit digits below this many Gaussians, four of 7 bits from it on (sgr_scan_sort.hip)
static int depth9_max_p() {
    static const int v = [] { const char* e = getenv("SGR_DEPTH9_MAX_P"); return e ? atoi(e) : 750000; }();
    return v;
}
static int pre_stage_min_p() {
    static const int v = [] { const char* e = getenv("SGR_PRE_STAGE_MIN_P"); return e ? atoi(e) : 3000000; }();
    return v;
}

// rasterizer_impl.cu:35-50
static uint32_t getHigherMsb(uint32_t n) {
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step;
        else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

// The camera arrives as three device arrays; a small kernel packs it (plus the host-side scalars) into a
// device-resident SgrCam so no device->host copy is needed.  The same launch clears what the forward needs cleared before
// its first real kernel -- the 16 header words (flags, num_rendered) and the T tile ranges (rasterizer_impl.cu:313) --
// which were two memset dispatches of ~5 us each.
__global__ void __launch_bounds__(256)
sgr_pack_camera_kernel(SgrCam* cam, const float* view, const float* proj, const float* campos, float tan_fovx,
                       float tan_fovy, float focal_x, float focal_y, int W, int H, int gx, int gy, float scale_modifier,
                       uint32_t* header, uint2* ranges, int T, uint32_t rect_mode) {
    const int t = threadIdx.x;
    for (int i = blockIdx.x * 256 + t; i < T; i += gridDim.x * 256) ranges[i] = make_uint2(0u, 0u);
    if (blockIdx.x != 0) return;
    if (t < 16) {
        // word 6: the frame's tile-rect mode; word 7: whether its forward writes the compact hit list (the blend backward
        // follows the FRAME's flag, whatever the switches say by the time it runs)
        header[t] = t == 6 ? (rect_mode & 0xffu) : (t == 7 ? (rect_mode >> 8) : 0u);
        cam->view[t] = view[t];
        cam->proj[t] = proj ? proj[t] : 0.f;
    }
    if (t < 3) cam->campos[t] = campos ? campos[t] : 0.f;
    if (t == 0) {
        cam->tan_fovx = tan_fovx; cam->tan_fovy = tan_fovy;
        cam->focal_x = focal_x; cam->focal_y = focal_y;
        cam->W = W; cam->H = H; cam->gx = gx; cam->gy = gy;
        cam->scale_modifier = scale_modifier;
    }
}

// the SgrCam lives in the header block of the geometry buffer (words 16.. of the 64-word header)
static SgrCam* cam_slot(const SgrGeomView& gv) { return reinterpret_cast<SgrCam*>(gv.header + 16); }
static_assert(sizeof(SgrCam) <= 48 * 4, "SgrCam must fit the geometry header");
static_assert(SGR_STAT_SEG_MAX == SGR_MAX_STAT_SEGMENTS, "sgr_common.h and include/sgr.h disagree");

static void pack_camera(const SgrGeomView& gv, const float* view, const float* proj, const float* campos,
                        float tan_fovx, float tan_fovy, int W, int H, float scale_modifier, uint2* ranges, int T,
                        int rect_mode, hipStream_t s) {
    const float focal_y = H / (2.0f * tan_fovy);  // rasterizer_impl.cu:225-226
    const float focal_x = W / (2.0f * tan_fovx);
    const int gx = (W + SGR_BLOCK_X - 1) / SGR_BLOCK_X, gy = (H + SGR_BLOCK_Y - 1) / SGR_BLOCK_Y;
    const int nb = std::max(1, std::min(64, (T + 255) / 256));
    sgr_pack_camera_kernel<<<nb, 256, 0, s>>>(cam_slot(gv), view, proj, campos, tan_fovx, tan_fovy, focal_x, focal_y, W, H,
                                             gx, gy, scale_modifier, gv.header, ranges, T,
                                             (uint32_t)rect_mode | (hit_list_on() ? 0x100u : 0u));
}

// ---- the forward without a host wait (sgr_set_lazy) ---------------------------------------------------------------
// The reference reads num_rendered back before it can size the binning buffer and launch the sort (rasterizer_impl.cu:284);
// so does sgr_forward by default -- one host wait per step.  In lazy mode the list buffers get a CAPACITY derived from the
// previous frames' R (+ 1/16 + 1024) instead: duplicate, sort and tile ranges run over all `cap` slots (the slots behind
// the frame's instances carry a key above every tile id and sort to the end), the read-back is still queued but nobody
// waits for it, and the value sgr_forward returns -- and sgr_backward takes -- is the capacity.  What the host would have
// checked at the wait is checked ONE CALL LATE, at the next lazy forward of the thread (or by sgr_lazy_status after a
// synchronisation): R > capacity, a Gaussian failing the frustum test under prefiltered = 1, a depth beyond the 27-bit
// sort keys.  Such a frame's outputs are invalid; the call that finds out returns SGR_E_LAZY and the thread goes through
// one blocking forward to re-seed the capacity.  The first forward of a thread is always blocking.  Nothing on this path
// synchronises, so a step (forward + backward) can be captured in a hipGraph and replayed (tests/test_gpu_graph.py); the
// checks are then the caller's (sgr_lazy_status).  Opt-in because of the late error: right for a fixed camera rig or a
// captured step, wrong for a trainer that draws a new view with a very different R every iteration.
static std::atomic<int> g_lazy{-1};
static std::atomic<int> g_lazy_epoch{0};  // bumped by sgr_set_lazy: every thread's next forward is a blocking one
static bool lazy_on() {
    int v = g_lazy.load(std::memory_order_relaxed);
    if (v < 0) {
        v = env_flag("SGR_LAZY") ? 1 : 0;
        g_lazy.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}
struct LazyPending {
    bool pending = false;       // a lazy forward's read-back has not been looked at yet
    uint32_t cap = 0;           // the capacity that forward ran with
    uint32_t* host_vals = nullptr;
    hipEvent_t landed = nullptr;
    bool wide = false;          // it sorted the depths on all 32 bits
};
static thread_local LazyPending t_lazy;
// 0 = fine, else a bit set: 1 R > capacity, 2 prefilter violation, 4 far depth with the narrow depth sort
static int lazy_flags(const LazyPending& lp) {
    const uint32_t* hv = lp.host_vals;
    return (hv[4] > lp.cap ? 1 : 0) | ((hv[0] & 1u) ? 2 : 0) | (((hv[2] & 1u) && !lp.wide) ? 4 : 0);
}

// One 256-byte device block per DEVICE for the whole process (allocated on first use, kept): the flag word of
// sgr_visible_filter's `prefiltered` check and the counters of sgr_densify_prune_mask.  Both users finish with a stream
// synchronisation, so the block is handed out under a mutex that is held until they return -- no per-thread allocations
// that are never freed, and two host threads cannot interleave on the same words.
SgrFlagBlock sgr_acquire_flag_block() {
    static std::mutex mu;
    static uint32_t* blk[64] = {};
    SgrFlagBlock b;
    b.lock = std::unique_lock<std::mutex>(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return b;
    if (!blk[dev] && hipMalloc((void**)&blk[dev], 256) != hipSuccess) blk[dev] = nullptr;
    b.ptr = blk[dev];
    return b;
}

extern "C" {

const char* sgr_last_error(void) { return g_err.c_str(); }
int sgr_version(void) { return 102; }  // 102: sgr_backward_extras gained masked_color_out + skip_sh_grad (101: color_ready_event + rows)

size_t sgr_geometry_bytes(int P) {
    return sgr_required([&](char* b, char** e) { sgr_geom_carve(b, (size_t)P, e); });
}
size_t sgr_binning_bytes(int R) {
    return sgr_required([&](char* b, char** e) { sgr_bin_carve(b, (size_t)R, e); });
}
size_t sgr_image_bytes(int width, int height) {
    const size_t T = (size_t)((width + SGR_BLOCK_X - 1) / SGR_BLOCK_X) * ((height + SGR_BLOCK_Y - 1) / SGR_BLOCK_Y);
    return sgr_required([&](char* b, char** e) { sgr_img_carve(b, (size_t)width * height, T, e); });
}
int sgr_partial_row_floats(int S) { return sgr_partial_row_stride(S); }

int sgr_forward(sgr_alloc_fn geometry_buffer, void* geometry_user, sgr_alloc_fn binning_buffer, void* binning_user,
                sgr_alloc_fn image_buffer, void* image_user, int P, int D, int M, int S, const float* background,
                int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
                const float* semantics, const float* opacities, const float* scales, float scale_modifier,
                const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                float* out_depth, float* out_alpha, float* out_semantic, int* radii, int debug, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int W = width, H = height;
    if (P < 0 || W <= 0 || H <= 0) return fail(SGR_E_INVALID, "P, width and height must be positive");
    const int gx = (W + SGR_BLOCK_X - 1) / SGR_BLOCK_X, gy = (H + SGR_BLOCK_Y - 1) / SGR_BLOCK_Y;
    if (gx > SGR_MAX_GRID_DIM || gy > SGR_MAX_GRID_DIM)
        return fail(SGR_E_INVALID, "image larger than 16368 px per side is not supported");
    if (S < 0 || S > SGR_SEM_MAX) return fail(SGR_E_INVALID, "semantic channels must be in [0, 32]");
    if (D < 0 || D > 3) return fail(SGR_E_INVALID, "SH degree must be in [0, 3]");
    const size_t N = (size_t)W * H, T = (size_t)gx * gy;

    if (P == 0) {  // rasterize_points.cu:86: nothing runs, outputs stay at their zero fill
        SGR_HIP(hipMemsetAsync(out_color, 0, 3 * N * sizeof(float), stream));
        SGR_HIP(hipMemsetAsync(out_depth, 0, N * sizeof(float), stream));
        SGR_HIP(hipMemsetAsync(out_alpha, 0, N * sizeof(float), stream));
        if (S) SGR_HIP(hipMemsetAsync(out_semantic, 0, (size_t)S * N * sizeof(float), stream));
        return 0;
    }
    if (!means3D || !opacities || !viewmatrix || !projmatrix || !background)
        return fail(SGR_E_INVALID, "means3D, opacities, viewmatrix, projmatrix and background are required");
    if ((shs == nullptr) == (colors_precomp == nullptr))
        return fail(SGR_E_INVALID, "provide exactly one of shs / colors_precomp");
    if (((scales == nullptr) || (rotations == nullptr)) == (cov3D_precomp == nullptr))
        return fail(SGR_E_INVALID, "provide exactly one of scales+rotations / cov3D_precomp");
    if (shs && (!cam_pos || M < (D + 1) * (D + 1))) return fail(SGR_E_INVALID, "shs need campos and M >= (D+1)^2");
    if (S > 0 && !semantics) return fail(SGR_E_INVALID, "S > 0 but semantics is NULL");

    char* gbase = geometry_buffer(sgr_geometry_bytes(P), geometry_user);
    if (!gbase) return fail(SGR_E_ALLOC, "geometry buffer allocation failed");
    const SgrGeomView gv = sgr_geom_carve(gbase, (size_t)P);
    char* ibase = image_buffer(sgr_image_bytes(W, H), image_user);
    if (!ibase) return fail(SGR_E_ALLOC, "image buffer allocation failed");
    const SgrImgView iv = sgr_img_carve(ibase, N, T);
    int* radii_ptr = radii ? radii : gv.internal_radii;  // rasterizer_impl.cu:232-235

    // The front end (camera pack, preprocess, depth sort, scan) runs once -- or twice, the second time with a depth sort on
    // all 32 key bits, when the preprocess reports a view depth beyond what the 27-bit keys order (>= 13 107: never seen in a
    // street scene; the host thread then keeps the wide sort for its next 64 forwards before it tries the narrow one again,
    // so a scene that really has such depths pays the repeat on one frame in 64).
    static thread_local int wide_left = 0;
    bool wide_depth = wide_left > 0;
    if (wide_depth) wide_left--;
    uint32_t* host_vals = nullptr;
    const uint32_t* order = nullptr;
    char* bbase = nullptr;
    size_t have_bytes = 0;
    static thread_local size_t r_hint = 0;

    // ---- lazy mode (sgr_set_lazy): what the previous lazy forward of this thread left to check
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cap_status);
    const bool capturing = cap_status != hipStreamCaptureStatusNone;
    // (also inside a stream capture: the pending copy and its event were queued BEFORE the capture began, waiting for them
    // on the host is legal, and the frame they belong to must not lose its check)
    if (t_lazy.pending) {
        t_lazy.pending = false;
        SGR_HIP(wait_for_readback(t_lazy.host_vals, t_lazy.landed));  // queued a whole step ago: landed long since
        const int fl = lazy_flags(t_lazy);
        if (t_lazy.host_vals[2] & 1u) wide_left = 64;
        if (fl) {
            r_hint = 0;  // the next forward of this thread is a blocking one: it re-seeds the capacity
            return fail(SGR_E_LAZY, std::string("the PREVIOUS lazy forward of this thread was invalid (") +
                        ((fl & 1) ? "more tile instances than the list capacity; " : "") +
                        ((fl & 2) ? "a Gaussian failed the frustum test although prefiltered is set; " : "") +
                        ((fl & 4) ? "a view depth beyond the 27-bit depth keys; " : "") + "its outputs must be discarded)");
        }
        // high-water mark with a slow decay, as in the blocking path (lazy: 1/16 of head-room -- the sort runs over it)
        const size_t Rp = t_lazy.host_vals[4];
        r_hint = std::max(Rp + Rp / 16 + 1024, r_hint - r_hint / 64);
    }
    {   // a change of the mode forgets the thread's capacity: its next forward blocks and seeds it from THAT frame (a
        // high-water mark left by much larger frames would make the lazy sort run over mostly padding)
        static thread_local int t_epoch = 0;
        const int e = g_lazy_epoch.load(std::memory_order_relaxed);
        if (t_epoch != e) {
            t_epoch = e;
            r_hint = 0;
        }
    }
    const bool lazy = lazy_on() && r_hint > 0;
    const bool tile_sort = tile_sort_on() && !lazy_on();  // (the lazy mode keeps the default chain: list_index())
    const int rmode = rect_mode();
    // what the list is made of, in index order: {count, rect} records of 8 bytes, or (marked-list mode) 16-byte records whose
    // first half is the reference's count + rect and whose second half is the live part
    const int aux16 = rmode == 3 ? 1 : 0;
    const uint2* aux_emit = aux16 ? reinterpret_cast<const uint2*>(gv.aux_ref) : gv.aux;

    int R = 0;
    uint32_t cap = 0;  // lazy: slots of the instance list
    if (lazy) {
        prof_begin(0, stream);
        pack_camera(gv, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, W, H, scale_modifier, iv.ranges, (int)T, rmode, stream);
        SGR_STAGE("pack_camera");
        sgr_launch_preprocess(P, D, M, means3D, scales, rotations, opacities, shs, cov3D_precomp, colors_precomp,
                              cam_slot(gv), gv, radii_ptr, prefiltered, (switches() & 64) != 0 || P >= pre_stage_min_p(),
                              rmode, stream);
        SGR_STAGE("preprocess");
        prof_end(stream);
        host_vals = pinned_pair(1);
        hipEvent_t landed = readback_event(1);
        if (!host_vals || !landed) return fail(SGR_E_HIP, "pinned readback slot / event creation failed");
        if (!capturing) host_vals[0] = host_vals[2] = host_vals[4] = host_vals[5] = SGR_READBACK_PENDING;
        SGR_HIP(hipMemcpyAsync(host_vals, gv.header, 6 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        if (!capturing) SGR_HIP(hipEventRecord(landed, stream));
        // from here on a copy is in flight: the late check is owed whatever happens to the rest of this call (an error
        // return below must not leave a copy nobody waits for -- the next forward would take its late arrival for its own)
        t_lazy.pending = !capturing;
        t_lazy.host_vals = host_vals;
        t_lazy.landed = landed;
        t_lazy.wide = wide_depth;
        t_lazy.cap = 0x7fffffffu;  // (set below; until then "no overflow")
        prof_begin(1, stream);
        const int dcur = sgr_launch_sort_pairs32(gv.dkeys, gv.dvals, (uint32_t)P, wide_depth ? 32 : SGR_DEPTH_KEY_BITS, gv.dhist,
                                                 gv.scan_tmp, stream, true, aux_emit, gv.aux_sorted, P < depth9_max_p() ? 9 : 8, aux16);
        order = gv.dvals[dcur];
        sgr_launch_scan_head(reinterpret_cast<const uint32_t*>(gv.aux_sorted), reinterpret_cast<const uint32_t*>(gv.aux), (size_t)P,
                             aux16 ? 4 : 2, gv.scan_tmp, gv.sub_sums, stream, 2);
        SGR_STAGE("depth_sort+scan");
        prof_end(stream);
        {   // capacity on the coarse ladder of the blocking path (consecutive calls ask for the same block size)
            size_t hq = r_hint, step = 1;
            while ((step << 1) <= hq) step <<= 1;
            step = std::max<size_t>(step >> 4, 1024);
            hq = (hq + step - 1) / step * step;
            cap = (uint32_t)std::min<size_t>(hq, 0x7fffffffu);
        }
        bbase = binning_buffer(sgr_binning_bytes((int)cap), binning_user);
        if (!bbase) return fail(SGR_E_ALLOC, "binning buffer allocation failed");
        R = (int)cap;  // what the caller hands to sgr_backward: it sizes the same carving there
        t_lazy.cap = cap;
    }
    for (int attempt = 0; attempt < 2 && !lazy; attempt++) {
        prof_begin(0, stream);
        pack_camera(gv, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, W, H, scale_modifier, iv.ranges, (int)T, rmode, stream);
        SGR_STAGE("pack_camera");

        sgr_launch_preprocess(P, D, M, means3D, scales, rotations, opacities, shs, cov3D_precomp, colors_precomp,
                              cam_slot(gv), gv, radii_ptr, prefiltered,
                              // SH rows staged through LDS (half a wave's rows at a time) or read per lane: bit-identical forms
                              // (tests); the staged one wins once the launch is deep enough to be throughput-bound -- measured
                              // 346 vs 360 us at 5 M Gaussians, 98.8 vs 90.2 us at 1 M -- so it is chosen by P (switch bit 6
                              // forces it, SGR_PRE_STAGE_MIN_P moves the threshold)
                              (switches() & 64) != 0 || P >= pre_stage_min_p(),
                              // tile rects: the reference's 3-sigma squares cut down to the tiles the Gaussian can reach
                              // alpha >= 1/255 in (sgr_preprocess.hip: 2 = bounding box + tile mask, 1 = bounding box only,
                              // switch bit 11); switch bit 10 keeps the reference's rects
                              rmode, stream);
        SGR_STAGE("preprocess");
        prof_end(stream);

        // K5 first: num_rendered (header[4], summed by the preprocess kernel) and the prefilter flag (header[0]) go to
        // pinned host memory in ONE 24-byte copy.  The host only waits for THAT copy (an event), after the depth sort and
        // the scan have been queued behind it: while it wakes up, allocates the binning buffer and queues the dozen short
        // binning kernels, the GPU is busy with the ~0.15 ms of sort + scan instead of idling (rocprofv3 kernel trace:
        // ~0.2 ms of gaps per forward with the wait placed after the scan, as rasterizer_impl.cu:281 has it).
        host_vals = pinned_pair();
        hipEvent_t landed = readback_event();
        if (!host_vals || !landed) return fail(SGR_E_HIP, "pinned readback slot / event creation failed");
        // the words carry a value the device never writes, so that the host can see them arrive (wait_for_readback);
        // header[2] = "a depth beyond the 27-bit sort keys" rides along
        host_vals[0] = host_vals[2] = host_vals[4] = host_vals[5] = SGR_READBACK_PENDING;
        SGR_HIP(hipMemcpyAsync(host_vals, gv.header, 6 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        SGR_HIP(hipEventRecord(landed, stream));
        // an error return between here and the wait must not leave this copy in flight: the next forward of this thread would
        // take its late arrival for its own read-back
        struct ReadbackDrain {
            hipEvent_t ev;
            ~ReadbackDrain() { if (ev) (void)hipEventSynchronize(ev); }
        } drain{landed};

        // Depth pre-sort of the P Gaussians (27 key bits: 3 x 9 or 4 x 7 bits over P elements; 32 bits = 4 passes of 8 when a
        // depth beyond 13 107 has been seen), then K4: scan of tiles_touched in that order.
        prof_begin(1, stream);
        // (the ids are not materialised before the sort: its first pass takes the element index as the value; its last
        // pass also carries every Gaussian's {tiles_touched, tile rect} into depth order -- ONE fused 8-byte gather instead
        // of three per-stage gathers through `order`: at 5 M Gaussians those read 0.6 GB each, rocprofv3 FETCH_SIZE)
        // (27 bits: 3 passes of 9 bits while the launches are latency-bound, 4 passes of 7 bits -- longer store runs -- from
        // 750 k Gaussians on: sgr_scan_sort.hip)
        if (tile_sort) {
            // per-tile sort form (switch bit 12, sgr_tile_sort.hip): no depth pre-sort -- the instances are emitted in index
            // order and every tile's list is sorted by depth in LDS after the tile sort; both scan sequences are index order
            order = nullptr;
            sgr_launch_scan_head(reinterpret_cast<const uint32_t*>(aux_emit), reinterpret_cast<const uint32_t*>(gv.aux), (size_t)P,
                                 aux16 ? 4 : 2, gv.scan_tmp, gv.sub_sums, stream, 2);
        } else {
        const int dcur = sgr_launch_sort_pairs32(gv.dkeys, gv.dvals, (uint32_t)P, wide_depth ? 32 : SGR_DEPTH_KEY_BITS, gv.dhist,
                                                 gv.scan_tmp, stream, true, aux_emit, gv.aux_sorted, P < depth9_max_p() ? 9 : 8, aux16);
        order = gv.dvals[dcur];
        // (second sequence of the same launches: the exclusive scan in index order = every Gaussian's first partial-gradient
        // row of the backward, SgrGeomView::u0)
        // (the scan's last step -- offsets of the individual Gaussians -- is done by the duplicate kernel, which needs them:
        // round 4 ran a third launch that wrote them to an array)
        sgr_launch_scan_head(reinterpret_cast<const uint32_t*>(gv.aux_sorted), reinterpret_cast<const uint32_t*>(gv.aux), (size_t)P,
                             aux16 ? 4 : 2, gv.scan_tmp, gv.sub_sums, stream, 2);
        }
        SGR_STAGE("depth_sort+scan");
        prof_end(stream);
        // The window between "R is known" and "the GPU runs out of queued work" is only the ~0.12 ms of sort + scan
        // above, and the allocation callback (a Python call into the torch allocator in the shipped binding) is the
        // slowest thing in it.  So the binning buffer is requested BEFORE the wait, sized for the previous forward's R
        // + 25 % (per host thread); only if that turns out too small is it requested again with the exact size.  The
        // carving below depends on R alone, so a larger buffer is simply partly unused.
        have_bytes = 0;
        bbase = nullptr;
        if (r_hint > 0) {
            // the hint decays a little every call; request sizes on a coarse ladder (steps of 1/16 of the next lower power of
            // two) so that consecutive calls ask the caller's caching allocator for the SAME block size instead of a new,
            // slightly smaller one each time (every new size is a device allocation: tens of ms on some hosts)
            size_t hq = r_hint;
            {
                size_t step = 1;
                while ((step << 1) <= hq) step <<= 1;
                step = std::max<size_t>(step >> 4, 1024);
                hq = (hq + step - 1) / step * step;
            }
            have_bytes = sgr_binning_bytes((int)std::min<size_t>(hq, 0x7fffffffu));
            bbase = binning_buffer(have_bytes, binning_user);
            if (!bbase) return fail(SGR_E_ALLOC, "binning buffer allocation failed");
        }
        SGR_HIP(wait_for_readback(host_vals, landed));  // the one host wait of the forward
        drain.ev = nullptr;

        if (!(host_vals[2] & 1u) || wide_depth || tile_sort) break;  // (the per-tile sort reads the far-depth flag on the device)
        wide_depth = true;
        wide_left = 64;
    }
    if (!lazy) {
        if (host_vals[0] & 1u)
            return fail(SGR_E_PREFILTER, "Point is filtered although prefiltered is set. This shouldn't happen!");
        // (host_vals[5], what num_rendered would be with the reference's rects, is reporting only -- export 17, best effort: it
        // shares a 64-bit atomic with the emitted count and is not a reason to refuse a frame whose emitted list fits)
        if (host_vals[4] > 0x7fffffffu) return fail(SGR_E_INVALID, "more than 2^31 tile instances");
        R = (int)host_vals[4];

        if (!bbase || sgr_binning_bytes(R) > have_bytes) {
            bbase = binning_buffer(sgr_binning_bytes(R), binning_user);
            if (!bbase) return fail(SGR_E_ALLOC, "binning buffer allocation failed");
        }
        // high-water mark with a slow decay: the views of one scene differ, and a miss only costs the late allocation
        r_hint = std::max((size_t)R + (size_t)R / 4 + 1024, r_hint - r_hint / 16);
    }
    const SgrBinView bv = sgr_bin_carve(bbase, (size_t)R);

    int cur = 0;
    // (also with R == 0: the kernel finishes the index-order scan, SgrGeomView::u0, which the exports read)
    prof_begin(2, stream);
    SgrGeomView gv_dup = gv;
    if (tile_sort) gv_dup.aux_sorted = const_cast<uint2*>(aux_emit);  // index-order emission: "depth order" is the identity
    const int key16 = tile_key16(T);
    sgr_launch_duplicate(P, gv_dup, order, gv.scan_tmp, gv.sub_sums, bv.keys[0], key16, bv.vals[0], gx, cap, rmode == 3 ? 1 : 0, stream);
    SGR_STAGE("duplicate");
    prof_end(stream);
    if (R > 0) {
        prof_begin(3, stream);
        const int bit = (int)getHigherMsb((uint32_t)T);  // rasterizer_impl.cu:303
        // (lazy: over all `cap` slots; the padding's keys are all ones in every sorted bit and stay behind the instances)
        if (key16) {
            uint16_t* const k16[2] = {reinterpret_cast<uint16_t*>(bv.keys[0]), reinterpret_cast<uint16_t*>(bv.keys[1])};
            cur = sgr_launch_sort_pairs16(k16, bv.vals, (uint32_t)R, bit, bv.hist, bv.scan_tmp, stream);
        } else {
            cur = sgr_launch_sort_pairs32(bv.keys, bv.vals, (uint32_t)R, bit, bv.hist, bv.scan_tmp, stream);
        }
        SGR_STAGE("sort");
        prof_end(stream);
        prof_begin(4, stream);
        sgr_launch_tile_ranges(R, bv.keys[cur], key16, iv.ranges, bv.touched, lazy ? (uint32_t)T : 0xffffffffu, stream);
        SGR_STAGE("tile_ranges");
        if (tile_sort) {
            // every tile's list (ascending id so far) into (depth, id) order: a stable radix sort on the depth keys in LDS
            sgr_launch_tile_sort((int)T, iv.ranges, bv.vals[cur], bv.vals[cur ^ 1], gv.dkeys[0], gv.header, bv.keys[cur ^ 1], bv.tkeys, stream);
            SGR_STAGE("tile_sort");
        }
        prof_end(stream);
    }
    const int lcur = tile_sort ? (cur ^ 1) : cur;  // which vals[] holds the final list (list_index())
    prof_begin(5, stream);
    const bool cull = !(switches() & 1);
    // Tile order of the two blend launches: decided per frame on the device (sgr_tile_order_kernel: longest list first when
    // the longest list is more than 2.5 x the mean, else the XCD-aware supertile order); switch bit 14 forces longest-first,
    // bit 15 keeps the supertile order without looking (no extra launch: the round-5 behaviour)
    const bool lpt = (switches() & 32768) == 0;
    if (lpt) {
        sgr_launch_tile_order(iv.ranges, (int)T, (switches() & 16384) ? 1 : 0, stream);
        SGR_STAGE("tile_order");
    }
    sgr_launch_blend_fwd(cull, (switches() & 128) != 0, gx, lpt ? -gy : gy, iv.ranges, bv.vals[lcur], W, H, S, gv.rec, semantics,
                         background, out_color, out_depth, out_alpha, out_semantic, iv.n_contrib, bv.hit4, hit_list_on() ? bv.hlist : nullptr, iv.n_contrib_k, stream);
    SGR_STAGE("blend_fwd");
    prof_end(stream);
    return R;
}

// which of the two ping-pong pairs holds the sorted tile keys: one flip per 8-bit pass
static int sorted_index(int width, int height) {
    const int gx = (width + SGR_BLOCK_X - 1) / SGR_BLOCK_X, gy = (height + SGR_BLOCK_Y - 1) / SGR_BLOCK_Y;
    const int end_bit = (int)getHigherMsb((uint32_t)(gx * gy));  // the tile sort only; depth order comes from the emission order
    return sgr_sort_pass_count(end_bit) & 1;
}
// ... and which vals[] holds the final instance list: the per-tile sort form (switch bit 12) writes it to the other one.
// (A forward and the backward / exports over its buffers must run under the same switch, like every other switch.)
static int list_index(int width, int height) { return sorted_index(width, height) ^ ((tile_sort_on() && !lazy_on()) ? 1 : 0); }

int sgr_backward(int P, int D, int M, int R, int S, const float* background, int width, int height,
                 const float* means3D, const float* shs, const float* colors_precomp, const float* semantics,
                 const float* alphas, const float* scales, float scale_modifier, const float* rotations,
                 const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
                 float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer,
                 char* image_buffer, const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dalphas,
                 const float* dL_dpix_semantic, float* dL_dmean2D, float* dL_dopacity, float* dL_dcolor,
                 float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                 float* dL_dsemantic, sgr_alloc_fn scratch, void* scratch_user, int debug, void* stream_) {
    return sgr_backward_ex(P, D, M, R, S, background, width, height, means3D, shs, colors_precomp, semantics, alphas, scales,
                           scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy,
                           radii, geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_dpix_depth, dL_dalphas,
                           dL_dpix_semantic, dL_dmean2D, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale,
                           dL_drot, dL_dsemantic, scratch, scratch_user, debug, stream_, nullptr);
}

int sgr_backward_ex(int P, int D, int M, int R, int S, const float* background, int width, int height,
                    const float* means3D, const float* shs, const float* colors_precomp, const float* semantics,
                    const float* alphas, const float* scales, float scale_modifier, const float* rotations,
                    const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
                    float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer,
                    char* image_buffer, const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dalphas,
                    const float* dL_dpix_semantic, float* dL_dmean2D, float* dL_dopacity, float* dL_dcolor,
                    float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                    float* dL_dsemantic, sgr_alloc_fn scratch, void* scratch_user, int debug, void* stream_,
                    const sgr_backward_extras* extras) {
    hipStream_t stream = (hipStream_t)stream_;
    SgrStatSink sink;
    if (extras && (extras->xyz_gradient_accum || extras->denom || extras->max_radii2D)) {
        if (!extras->xyz_gradient_accum || !extras->denom || !extras->max_radii2D)
            return fail(SGR_E_INVALID, "the statistics sink needs xyz_gradient_accum, denom and max_radii2D together");
        sink.accum = extras->xyz_gradient_accum;
        sink.denom = extras->denom;
        sink.max_radii = extras->max_radii2D;
        if (extras->segments != nullptr) {
            const int n = extras->n_segments;
            if (n < 0 || n > SGR_STAT_SEG_MAX)
                return fail(SGR_E_INVALID, "statistics sink: at most 128 segments per call");
            int end = 0;
            for (int i = 0; i < n; i++) {
                const sgr_stat_segment& sg = extras->segments[i];
                if (sg.src_start < end || sg.count < 0 || sg.dst_offset < 0 || (long)sg.src_start + sg.count > (long)P)
                    return fail(SGR_E_INVALID, "statistics sink: segments must be sorted, non-overlapping and inside [0, P)");
                end = sg.src_start + sg.count;
                sink.start[i] = sg.src_start;
                sink.count[i] = sg.count;
                sink.shift[i] = sg.dst_offset - sg.src_start;
            }
            // destinations: inside the persistent arrays and pairwise disjoint (n <= 128: the quadratic check is nothing)
            for (int i = 0; i < n; i++) {
                const sgr_stat_segment& a = extras->segments[i];
                if (extras->rows > 0 && (long)a.dst_offset + a.count > (long)extras->rows)
                    return fail(SGR_E_INVALID, "statistics sink: a segment's destination lies outside the persistent arrays");
                for (int j = 0; j < i; j++) {
                    const sgr_stat_segment& b = extras->segments[j];
                    if (a.count > 0 && b.count > 0 && a.dst_offset < b.dst_offset + b.count && b.dst_offset < a.dst_offset + a.count)
                        return fail(SGR_E_INVALID, "statistics sink: two segments write the same persistent rows");
                }
            }
            sink.nseg = n;
            if (n == 0) sink.accum = sink.denom = sink.max_radii = nullptr;  // nothing of this frame is tracked
        }
    }
    (void)colors_precomp; (void)scale_modifier; (void)viewmatrix; (void)projmatrix; (void)campos;
    (void)tan_fovx; (void)tan_fovy;  // already resident in the geometry buffer's camera block
    if (P <= 0) return 0;
    const int W = width, H = height;
    if (S < 0 || S > SGR_SEM_MAX) return fail(SGR_E_INVALID, "semantic channels must be in [0, 32]");
    if (!geom_buffer || !image_buffer || (R > 0 && !binning_buffer))
        return fail(SGR_E_INVALID, "backward needs the buffers produced by forward");
    // every array the kernels dereference unconditionally (a NULL here would be a GPU memory fault, not an error code)
    if (!means3D || !background || !alphas || !dL_dpix || !dL_dpix_depth || !dL_dalphas)
        return fail(SGR_E_INVALID, "means3D, background, alphas, dL_dpix, dL_dpix_depth and dL_dalphas are required");
    if (S > 0 && (!semantics || !dL_dpix_semantic || !dL_dsemantic))
        return fail(SGR_E_INVALID, "S > 0 needs semantics, dL_dpix_semantic and dL_dsemantic");
    if (!dL_dmean2D || !dL_dopacity || !dL_dcolor || !dL_dmean3D || !dL_dcov3D || !dL_dscale || !dL_drot)
        return fail(SGR_E_INVALID, "all gradient outputs except dL_dsh / dL_dsemantic are required");
    const int skip_sh = (extras && extras->skip_sh_grad) ? 1 : 0;
    if (shs && !dL_dsh && !skip_sh) return fail(SGR_E_INVALID, "shs given but dL_dsh is NULL");
    const int gx = (W + SGR_BLOCK_X - 1) / SGR_BLOCK_X, gy = (H + SGR_BLOCK_Y - 1) / SGR_BLOCK_Y;
    const size_t N = (size_t)W * H, T = (size_t)gx * gy;
    const SgrGeomView gv = sgr_geom_carve(geom_buffer, (size_t)P);
    const SgrImgView iv = sgr_img_carve(image_buffer, N, T);
    const int* radii_ptr = radii ? radii : gv.internal_radii;
    const int stride = sgr_partial_row_stride(S);
    // Which blend backward runs.  Default: the LDS-staged kernel (sgr_blend_bwd.hip).  Switch bit 8 (SGR_SW=1) selects the
    // scalar walk (sgr_blend_bwd_sw.hip; S = 0 with the hit record, cull, DPP reduction and deterministic combine on): it
    // writes FOUR rows per (tile, instance), one per quadrant.  A second design that was built, is parity-tested and
    // measured 11 % slower per step: both kernels sit at the VALU issue bound of the same per-visit
    // arithmetic, and the per-quadrant rows cost the row sum more than the barriers cost the LDS kernel.
    const int sw_all = switches();
    const bool quad = SGR_WITH_VARIANTS && S == 0 && R > 0 && (sw_all & 256) != 0 && (sw_all & (1 | 2 | 4 | 8 | 16)) == 0;
    // scratch = [P float4: conic + depth terms between the two per-Gaussian stages][R (or 4 R) partial rows]
    const size_t cd_bytes = sgr_align_up((size_t)P * sizeof(float4), 256);
    const size_t bytes = sgr_align_up((size_t)R * (quad ? 4u : 1u) * stride * sizeof(float), 256);
    char* sbase = scratch(cd_bytes + bytes, scratch_user);
    if (!sbase) return fail(SGR_E_ALLOC, "backward scratch allocation failed");
    float4* cd = reinterpret_cast<float4*>(sbase);
    float* partials = reinterpret_cast<float*>(sbase + cd_bytes);
    // One byte per (tile, instance) row: "written by this backward".  It lives in the binning buffer and arrives zeroed
    // by the forward's tile-ranges launch; the rows a backward writes are a function of the forward's hit record alone,
    // so repeated backwards over one forward re-mark the same bytes.  The A/B switches that change the visited set
    // (no cull / no hit record) clear it explicitly, before and after.
    uint8_t* touched = nullptr;
    if (R > 0) {
        const SgrBinView bv = sgr_bin_carve(binning_buffer, (size_t)R);
        touched = bv.touched;
        const int cur = list_index(W, H);
        // (the scalar walk leaves quadrant MASKS in these bytes and the LDS kernel ones: a scalar-walk backward clears them
        // before and after itself, so that the two kernels can follow each other over one forward)
        const bool odd_set = (switches() & (1 | 8)) != 0 || quad;
        prof_begin(6, stream);
        if (odd_set) SGR_HIP(hipMemsetAsync(touched, 0, (size_t)R, stream));
        prof_end(stream);
        prof_begin(7, stream);
        const int sw = switches();
        const bool cull = !(sw & 1), dpp = !(sw & 2), det = !(sw & 4);
        // the forward's record of which (quadrant, instance) pairs blended at all; switch 8: the kernel redoes the
        // geometric cull instead (A/B and tests: the two walks must give bit-identical gradients)
        const uint8_t* hits = (sw & 8) ? nullptr : bv.hit4;
#if SGR_WITH_VARIANTS
        if (quad)
            sgr_launch_blend_bwd_sw((sw & 128) != 0, gx, gy, iv.ranges, bv.vals[cur], W, H, background, gv.rec, gv.u0, gv.tmask, alphas,
                                    iv.n_contrib, bv.hit4, dL_dpix, dL_dpix_depth, dL_dalphas, partials, stride, touched, stream);
        else
#endif
            sgr_launch_blend_bwd(cull, dpp, det, (sw & 16) != 0, (sw & 128) != 0, gx, (sw & 32768) ? gy : -gy, iv.ranges, bv.vals[cur], W, H, S, background, gv.rec, gv.u0, gv.tmask, semantics,
                                 alphas, iv.n_contrib, hits, dL_dpix, dL_dpix_depth, dL_dalphas, dL_dpix_semantic, partials,
                                 touched, (uint32_t)R, (SGR_HLIST && !(sw & 65536)) ? bv.hlist : nullptr, iv.n_contrib_k,
                                 gv.header + 7, stream);
        SGR_STAGE("blend_bwd");
        prof_end(stream);
    }
    prof_begin(8, stream);
    const int ev_failed = ((switches() & 128) ? sgr_launch_gauss_bwd_strict : sgr_launch_gauss_bwd)(
        P, D, M, S, means3D, radii_ptr, shs, scales, rotations, cov3D_precomp, cam_slot(gv), gv, partials, stride, touched, cd,
        dL_dmean2D, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, dL_dsemantic, sink, quad ? 1 : 0,
        (switches() & 128) ? 1 : 0, W, H, extras ? (hipEvent_t)extras->color_ready_event : nullptr, (switches() & 512) ? 1 : 0,
        (uint32_t)R, extras ? extras->masked_color_out : nullptr, skip_sh, stream);
    SGR_STAGE("gauss_bwd");
    prof_end(stream);
    if (ev_failed) return fail(SGR_E_HIP, "hipEventRecord(color_ready_event) failed: is it a valid event of this device?");
    if (touched && ((switches() & (1 | 8)) != 0 || quad)) SGR_HIP(hipMemsetAsync(touched, 0, (size_t)R, stream));
    return 0;
}

// ---- view-sharded training: factored exchange of the SH gradient (sgr_multiview.hip) ----
int sgr_masked_color_grad(int P, const char* geom_buffer, const float* dL_dcolor, float* dL_drgb, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 0;
    if (P <= 0) return 0;
    if (!geom_buffer || !dL_dcolor || !dL_drgb) return fail(SGR_E_INVALID, "geom_buffer, dL_dcolor and dL_drgb are required");
    const SgrGeomView gv = sgr_geom_carve(const_cast<char*>(geom_buffer), (size_t)P);
    sgr_launch_masked_color_grad(P, gv.clamped, dL_dcolor, dL_drgb, stream);
    SGR_STAGE("masked_color_grad");
    return 0;
}

int sgr_sh_grad_from_views(int P, int D, int M, int V, const float* means3D, const float* campos, const float* dL_drgb,
                           float* dL_dsh, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 0;
    if (P <= 0) return 0;
    if (D < 0 || D > 3 || M < (D + 1) * (D + 1) || V < 0) return fail(SGR_E_INVALID, "need 0 <= D <= 3, M >= (D+1)^2, V >= 0");
    if (!means3D || !campos || !dL_drgb || !dL_dsh) return fail(SGR_E_INVALID, "means3D, campos, dL_drgb and dL_dsh are required");
    sgr_launch_sh_grad_from_views(P, D, M, V, means3D, 0, campos, 3, dL_drgb, (size_t)3 * P, dL_dsh, stream);
    SGR_STAGE("sh_grad_from_views");
    return 0;
}

int sgr_sh_grad_from_views_ex(int P, int D, int M, int V, const float* means3D, size_t means_view_stride,
                              const float* campos, size_t campos_view_stride, const float* dL_drgb,
                              size_t drgb_view_stride, float* dL_dsh, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 0;
    if (P <= 0) return 0;
    if (D < 0 || D > 3 || M < (D + 1) * (D + 1) || V < 0) return fail(SGR_E_INVALID, "need 0 <= D <= 3, M >= (D+1)^2, V >= 0");
    if (!means3D || !campos || !dL_drgb || !dL_dsh) return fail(SGR_E_INVALID, "means3D, campos, dL_drgb and dL_dsh are required");
    sgr_launch_sh_grad_from_views(P, D, M, V, means3D, means_view_stride, campos, campos_view_stride, dL_drgb,
                                  drgb_view_stride, dL_dsh, stream);
    SGR_STAGE("sh_grad_from_views_ex");
    return 0;
}

int sgr_set_lazy(int on) {
    const int prev = lazy_on() ? 1 : 0;
    if (on >= 0) {
        g_lazy.store(on ? 1 : 0, std::memory_order_relaxed);
        g_lazy_epoch.fetch_add(1, std::memory_order_relaxed);
    }
    return prev;
}

int sgr_lazy_status(int* num_rendered, int* capacity, int* flags) {
    if (!t_lazy.host_vals || t_lazy.cap == 0) return fail(SGR_E_INVALID, "no lazy forward has run on this thread");
    if (num_rendered) *num_rendered = (int)t_lazy.host_vals[4];
    if (capacity) *capacity = (int)t_lazy.cap;
    if (flags) *flags = lazy_flags(t_lazy);
    return 0;
}

int sgr_profile_host_wait_us(int reset) {
    const unsigned long long ns = reset ? g_wait_ns.exchange(0, std::memory_order_relaxed) : g_wait_ns.load(std::memory_order_relaxed);
    return (int)std::min<unsigned long long>(ns / 1000ull, 0x7fffffffull);
}

int sgr_has_variants(void) { return SGR_WITH_VARIANTS ? 1 : 0; }

int sgr_test_switches(int mask) {
    const int prev = switches() | ((SGR_WITH_VARIANTS && sgr_sort_get_one_sweep()) ? 32 : 0);
    if (mask >= 0) {
        if (!SGR_WITH_VARIANTS) mask &= ~SGR_VARIANT_BITS;
        g_switches.store(mask & ~32, std::memory_order_relaxed);
        sgr_sort_set_one_sweep((mask >> 5) & 1);
    }
    return prev;
}

int sgr_profile_enable(int on) {
    if (on && !g_prof.created) {
        for (int i = 0; i < SGR_PROF_SLOTS; i++)
            for (int k = 0; k < 2; k++)
                if (hipEventCreate(&g_prof.ev[i][k]) != hipSuccess) return fail(SGR_E_HIP, "hipEventCreate failed");
        // first use of an event allocates its completion signal; do that here, outside any timed region (observed: a
        // recording pass over fresh events ran at 14 ms per step instead of 5.7)
        for (int i = 0; i < SGR_PROF_SLOTS; i++)
            for (int k = 0; k < 2; k++) (void)hipEventRecord(g_prof.ev[i][k], nullptr);
        (void)hipDeviceSynchronize();
        g_prof.created = true;
    }
    g_prof.n.store(0);
    g_prof.on.store(on != 0);
    return 0;
}

// An event pair costs ~10 us of GPU idle time each side of the bracketed launch (the queue drains at the marker), so
// a timed region that wants its throughput undisturbed records only every k-th occurrence of a stage.
int sgr_profile_sample(int every) {
    const int prev = g_prof.every.load();
    if (every >= 1) {
        g_prof.every.store(every);
        for (int i = 0; i < SGR_PROF_STAGES; i++) g_prof.seen[i].store(0u);
    }
    return prev;
}

int sgr_profile_select(int stage_mask) {
    const int prev = g_prof.mask.load();
    if (stage_mask >= 0) g_prof.mask.store(stage_mask & ((1 << SGR_PROF_STAGES) - 1));
    return prev;
}

// Sums the recorded stage durations (ms) since sgr_profile_enable(1) and resets the recorder.
int sgr_profile_read(double* sum_ms, int* counts) {
    for (int i = 0; i < SGR_PROF_STAGES; i++) { sum_ms[i] = 0.0; counts[i] = 0; }
    if (!g_prof.created) return 0;
    SGR_HIP(hipDeviceSynchronize());
    const int n = std::min(g_prof.n.load(), SGR_PROF_SLOTS);
    for (int i = 0; i < n; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof.ev[i][0], g_prof.ev[i][1]) != hipSuccess) continue;  // slot never closed
        sum_ms[g_prof.stage[i]] += ms;
        counts[g_prof.stage[i]]++;
    }
    g_prof.n.store(0);
    return n;
}

int sgr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 0;
    (void)projmatrix;  // the reference's in_frustum only uses the view transform (auxiliary.h:139-164)
    if (P <= 0) return 0;
    if (!means3D || !viewmatrix || !present) return fail(SGR_E_INVALID, "means3D, viewmatrix and present are required");
    sgr_launch_mark_visible(P, means3D, viewmatrix, present, stream);
    SGR_STAGE("mark_visible");
    return 0;
}

int sgr_visible_filter(int P, int width, int height, const float* means3D, const float* scales, float scale_modifier,
                       const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                       const float* projmatrix, float tan_fovx, float tan_fovy, int prefiltered, int* radii,
                       float* means2D, int debug, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (P <= 0) return 0;
    const int W = width, H = height;
    const int gx = (W + SGR_BLOCK_X - 1) / SGR_BLOCK_X, gy = (H + SGR_BLOCK_Y - 1) / SGR_BLOCK_Y;
    if (gx > SGR_MAX_GRID_DIM || gy > SGR_MAX_GRID_DIM) return fail(SGR_E_INVALID, "image too large");
    if (!means3D || !viewmatrix || !projmatrix || !radii || !means2D)
        return fail(SGR_E_INVALID, "means3D, viewmatrix, projmatrix, radii and means2D are required");
    if (((scales == nullptr) || (rotations == nullptr)) == (cov3D_precomp == nullptr))
        return fail(SGR_E_INVALID, "provide exactly one of scales+rotations / cov3D_precomp");
    SgrCamArgs ca;
    ca.view = viewmatrix; ca.proj = projmatrix;
    ca.tan_fovx = tan_fovx; ca.tan_fovy = tan_fovy;
    ca.focal_y = H / (2.0f * tan_fovy); ca.focal_x = W / (2.0f * tan_fovx);
    ca.W = W; ca.H = H; ca.gx = gx; ca.gy = gy; ca.scale_modifier = scale_modifier;
    SgrGeomView gv;
    memset(&gv, 0, sizeof(gv));
    uint32_t* flag = nullptr;
    SgrFlagBlock fb;
    if (prefiltered) {  // only then can the kernel raise the "filtered although prefiltered" flag
        fb = sgr_acquire_flag_block();
        flag = fb.ptr;
        if (!flag) return fail(SGR_E_HIP, "flag block allocation failed");
        SGR_HIP(hipMemsetAsync(flag, 0, 4, stream));
        gv.header = flag;
    }
    SGR_HIP(hipMemsetAsync(means2D, 0, (size_t)P * 2 * sizeof(float), stream));  // torch::full(0), rasterize_points.cu:271
    sgr_launch_filter(P, means3D, scales, rotations, cov3D_precomp, ca, gv, radii, means2D, prefiltered, stream);
    SGR_STAGE("filter");
    if (prefiltered) {
        uint32_t* host = pinned_pair();
        if (!host) return fail(SGR_E_HIP, "pinned readback slot creation failed");
        SGR_HIP(hipMemcpyAsync(host, flag, 4, hipMemcpyDeviceToHost, stream));
        SGR_HIP(hipStreamSynchronize(stream));
        if (host[0] & 1u) return fail(SGR_E_PREFILTER, "Point is filtered although prefiltered is set. This shouldn't happen!");
    }
    return 0;
}

int sgr_knn(int P, const float* points, float* meanDists, sgr_alloc_fn scratch, void* scratch_user, void* stream_) {
    std::string err;
    const int rc = sgr_knn_impl(P, points, meanDists, scratch, scratch_user, (hipStream_t)stream_, err);
    if (rc < 0) g_err = err;
    return rc;
}

// ------------------------------------------------------------------------------------------------
// introspection
__global__ void sgr_strip_dead_kernel(uint32_t* v, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] &= ~SGR_DEAD;
}
__global__ void sgr_export_kernel(int which, int P, SgrGeomView gv, void* dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    switch (which) {
        case 0: ((float*)dst)[i] = gv.rec[4 * (size_t)i + 2].w; break;
        case 1: {
            const uint32_t c = gv.clamped[i];
            ((uint8_t*)dst)[3 * i] = c & 1u; ((uint8_t*)dst)[3 * i + 1] = (c >> 1) & 1u; ((uint8_t*)dst)[3 * i + 2] = (c >> 2) & 1u;
        } break;
        case 2: { const float4 a = gv.rec[4 * (size_t)i]; ((float*)dst)[2 * i] = a.x; ((float*)dst)[2 * i + 1] = a.y; } break;
        case 4: ((float4*)dst)[i] = gv.rec[4 * (size_t)i + 1]; break;
        case 5: { const float4 c = gv.rec[4 * (size_t)i + 2]; ((float*)dst)[3 * i] = c.x; ((float*)dst)[3 * i + 1] = c.y; ((float*)dst)[3 * i + 2] = c.z; } break;
        // (marked-list mode, header[6] == 3: the list is made of the reference's rects, gv.aux_ref; gv.aux / u0 number the rows)
        case 6: ((uint32_t*)dst)[i] = gv.header[6] == 3u ? gv.aux_ref[i].x : gv.aux[i].x; break;
        case 7: ((uint32_t*)dst)[i] = gv.u0[i] + gv.aux[i].x; break;  // the reference's inclusive index-order scan (mode 3: by a scan, below)
        case 14: { const float4 a = gv.rec[4 * (size_t)i]; ((float*)dst)[2 * i] = a.z; ((float*)dst)[2 * i + 1] = a.w; } break;
        case 16: {  // tile rect {x0, y0, x1, y1} (exclusive upper corner); all zero for a culled Gaussian
            const uint2 a = gv.header[6] == 3u ? make_uint2(gv.aux_ref[i].x, gv.aux_ref[i].y) : gv.aux[i];
            // (a masked rect of at most 64 tiles does not store its height: the highest set bit of the mask gives the last row
            // that matters, and rows above it hold no emitted tile)
            const uint32_t x0 = a.y & 1023u, y0 = (a.y >> 10) & 1023u, w = (a.y >> 20) & 1023u;
            uint32_t h = w ? a.x / w : 0u;
            if (a.y & SGR_RECT_MASKED) h = (63u - (uint32_t)__clzll((long long)gv.tmask[i])) / w + 1u;
            ((uint4*)dst)[i] = a.x ? make_uint4(x0, y0, x0 + w, y0 + h) : make_uint4(0u, 0u, 0u, 0u);
        } break;
        case 18:  // tile mask (0 = every tile of the rect is emitted)
            ((uint64_t*)dst)[i] = (gv.header[6] != 3u && gv.aux[i].x && (gv.aux[i].y & SGR_RECT_MASKED)) ? gv.tmask[i] : 0ull;
            break;
    }
}

int sgr_export_internal(int which, int P, int R, int width, int height, char* geom_buffer, char* binning_buffer,
                        char* image_buffer, void* dst, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 1;
    const int gx = (width + SGR_BLOCK_X - 1) / SGR_BLOCK_X, gy = (height + SGR_BLOCK_Y - 1) / SGR_BLOCK_Y;
    const size_t N = (size_t)width * height, T = (size_t)gx * gy;
    if (which <= 7 || which == 14 || which == 16 || which == 17 || which == 18) {
        if (P <= 0) return 0;
        const SgrGeomView gv = sgr_geom_carve(geom_buffer, (size_t)P);
        if (which == 17) { SGR_HIP(hipMemcpyAsync(dst, gv.header + 5, 4, hipMemcpyDeviceToDevice, stream)); return 0; }
        if (which == 3) return fail(SGR_E_INVALID, "cov3D is not materialised (the backward recomputes it)");
        if (which == 7) {
            uint32_t mode = 0;
            SGR_HIP(hipMemcpyAsync(&mode, gv.header + 6, 4, hipMemcpyDeviceToHost, stream));
            SGR_HIP(hipStreamSynchronize(stream));
            if (mode == 3u) {  // the reference's point_offsets = inclusive scan of ITS tiles_touched (the forward scans them in depth order only)
                sgr_launch_scan(reinterpret_cast<const uint32_t*>(gv.aux_ref), (uint32_t*)dst, (size_t)P, gv.scan_tmp, true, stream,
                                nullptr, nullptr, 4);
                SGR_STAGE("export scan");
                return 0;
            }
        }
        sgr_export_kernel<<<(P + 255) / 256, 256, 0, stream>>>(which, P, gv, dst);
        SGR_STAGE("export");
        return 0;
    }
    if (which == 8 || which == 9 || which == 15 || which == 19) {
        if (R <= 0) return 0;
        const SgrBinView bv = sgr_bin_carve(binning_buffer, (size_t)R);
        if (which == 19) { SGR_HIP(hipMemcpyAsync(dst, bv.hlist, (size_t)R * 4, hipMemcpyDeviceToDevice, stream)); return 0; }
        if (which == 15) { SGR_HIP(hipMemcpyAsync(dst, bv.hit4, (size_t)R, hipMemcpyDeviceToDevice, stream)); return 0; }
        const int cur = sorted_index(width, height), lcur = list_index(width, height);
        if (which == 8) {
            SGR_HIP(hipMemcpyAsync(dst, bv.vals[lcur], (size_t)R * 4, hipMemcpyDeviceToDevice, stream));
            sgr_strip_dead_kernel<<<(R + 255) / 256, 256, 0, stream>>>((uint32_t*)dst, R);  // marked-list mode: bit 31 = cannot blend
            SGR_STAGE("export point_list");
        } else {
            const SgrGeomView gv = sgr_geom_carve(geom_buffer, (size_t)P);
            sgr_launch_compose_keys(R, bv.keys[cur], tile_key16(T), bv.vals[lcur], gv.rec, (uint64_t*)dst, stream);
            SGR_STAGE("export keys");
        }
        return 0;
    }
    const SgrImgView iv = sgr_img_carve(image_buffer, N, T);
    if (which == 12) { SGR_HIP(hipMemcpyAsync(dst, iv.ranges, T * 8, hipMemcpyDeviceToDevice, stream)); return 0; }
    if (which == 13) { SGR_HIP(hipMemcpyAsync(dst, iv.n_contrib, N * 4, hipMemcpyDeviceToDevice, stream)); return 0; }
    if (which == 20) { SGR_HIP(hipMemcpyAsync(dst, iv.n_contrib_k, N * 4, hipMemcpyDeviceToDevice, stream)); return 0; }
    return fail(SGR_E_INVALID, "unknown internal array");
}

// ------------------------------------------------------------------------------------------------
// primitive self-tests
int sgr_test_scan(const uint32_t* in, uint32_t* out, size_t n, int inclusive, uint32_t* tmp, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 1;
    sgr_launch_scan(in, out, n, tmp, inclusive != 0, stream);
    SGR_STAGE("scan");
    return 0;
}
int sgr_test_sort(uint64_t* keys0, uint64_t* keys1, uint32_t* vals0, uint32_t* vals1, uint32_t n, int end_bit,
                  uint32_t* hist, uint32_t* scan_tmp, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 1;
    uint64_t* keys[2] = {keys0, keys1};
    uint32_t* vals[2] = {vals0, vals1};
    const int cur = sgr_launch_sort_pairs(keys, vals, n, end_bit, hist, scan_tmp, stream);
    SGR_STAGE("sort");
    if (sgr_sort_get_one_sweep() && n) {  // one-sweep form: a look-back that gave up leaves its mark in the control block
        uint32_t e = 0;
        SGR_HIP(hipMemcpyAsync(&e, hist + SGR_SORT_MAX_PASS * 256 + SGR_SORT_MAX_PASS, 4, hipMemcpyDeviceToHost, stream));
        SGR_HIP(hipStreamSynchronize(stream));
        if (e) return sgr_set_error(SGR_E_HIP, "radix sort: look-back gave up");
    }
    return cur;
}
int sgr_test_sort32(uint32_t* keys0, uint32_t* keys1, uint32_t* vals0, uint32_t* vals1, uint32_t n, int end_bit,
                    int max_bits, uint32_t* hist, uint32_t* scan_tmp, void* stream_) {
    if (max_bits != 8 && max_bits != 9) return sgr_set_error(SGR_E_INVALID, "sgr_test_sort32: max_bits must be 8 or 9");
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 1;
    uint32_t* keys[2] = {keys0, keys1};
    uint32_t* vals[2] = {vals0, vals1};
    const int cur = sgr_launch_sort_pairs32(keys, vals, n, end_bit, hist, scan_tmp, stream, false, nullptr, nullptr, max_bits);
    SGR_STAGE("sort32");
    if (sgr_sort_get_one_sweep() && n) {  // one-sweep form: a look-back that gave up leaves its mark in the control block
        uint32_t e = 0;
        SGR_HIP(hipMemcpyAsync(&e, hist + SGR_SORT_MAX_PASS * 256 + SGR_SORT_MAX_PASS, 4, hipMemcpyDeviceToHost, stream));
        SGR_HIP(hipStreamSynchronize(stream));
        if (e) return sgr_set_error(SGR_E_HIP, "radix sort: look-back gave up");
    }
    return cur;
}
size_t sgr_test_sort_hist_words(uint32_t n) { return sgr_sort_hist_words(n ? n : 1); }
size_t sgr_test_scan_tmp_words(size_t n) { return sgr_scan_tmp_count(n ? n : 1); }
__global__ void sgr_exact_math_test_kernel(int n, const float* x, float* exp_lib, float* exp_ref, const float* a,
                                           const float* b, float* div_lib, float* div_ref) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    exp_lib[i] = expf(x[i]);
    exp_ref[i] = sgr_expf_ref(x[i]);
    const SgrRcp rc = sgr_rcp_refined(b[i]);
    div_lib[i] = a[i] / b[i];
    div_ref[i] = sgr_div_by(a[i], b[i], rc.y);
}
int sgr_test_exact_math(int n, const float* x, float* exp_lib, float* exp_ref, const float* a, const float* b,
                        float* div_lib, float* div_ref, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 1;
    if (n <= 0) return 0;
    sgr_exact_math_test_kernel<<<(n + 255) / 256, 256, 0, stream>>>(n, x, exp_lib, exp_ref, a, b, div_lib, div_ref);
    SGR_STAGE("exact_math");
    return 0;
}
int sgr_test_lds_atomic_order(const uint32_t* pattern, uint32_t* out, int trials, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 1;
    sgr_launch_lds_atomic_order_test(pattern, out, trials, stream);
    SGR_STAGE("lds_atomic_order");
    return 0;
}
int sgr_test_wave_sum(const float* in, float* out_dpp, float* out_shfl, int nwaves, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int debug = 1;
    sgr_launch_wave_sum_test(in, out_dpp, out_shfl, nwaves, stream);
    SGR_STAGE("wave_sum");
    return 0;
}

}  // extern "C"
