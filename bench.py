#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native Gaussian rasterizer.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): train iters/s (rasterizer forward + backward) at 1M synthetic Gaussians, 1920x1280,
SH degree 3.  One "step" = one `GaussianRasterizer(...)` forward plus the autograd backward through it, driven by
dense random per-pixel upstream gradients dL/dcolor, dL/ddepth, dL/dalpha (the gradients of the scalar loss
sum(out * w); `--loss scalar` builds that loss with torch ops instead, which adds ~17 small torch kernels, ~0.13 ms,
that are not part of the rasterizer -- the CPU and reference-kernel baselines below are fed the same upstream
gradients directly).  Inputs are resident in HBM.  With N > 1 every rank renders
its own camera view of the replicated Gaussian set (weak scaling: per-GPU work fixed) and the per-Gaussian gradients
are summed over RCCL each step (street_gaussians_amd/multiview.py: by default the dense parameter gradients are
all-reduced and the SH gradient, which is rank-1 per view, is rebuilt on every rank from an all-gather of the per-view
dRGB; `--reduce bucket` all-reduces everything as one flat bucket).  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      -- the dominant kernel (blend backward): algorithmic bytes per launch (SURVEY.md 8d B_blend_b)
                   divided by its mean duration measured with HIP events on the launch stream INSIDE the timed region
                   (sgr_profile_*; only that kernel is bracketed there, two events per step), against the 8 TB/s HBM
                   peak.  "stages_ms" (every stage) comes from a short extra pass after the timed region.
                   "traffic" is the rocprofv3 PMC figure of profiles/pmc_blend_bwd.json and is printed only when that
                   file was measured on the kernel sources this build was compiled from.
  sustained     -- (N = 1) an extra region of >= 1 s of back-to-back steps when the K timed steps were shorter.
  other_configs -- (N = 1) BASELINE.json's other single-GPU configurations on the same build: 500 k, 2 M + 19
                   semantic channels, 5 M Gaussians (ms/step, R, V, blend kernel times).
  cpu_baseline  -- the C oracle (a line-by-line port of the reference algorithm; the reference has no CPU path)
                   on the host's cores (all of them, and one), on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from street_gaussians_amd import synthetic as syn  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
# "partials_memset": since the row flags are cleared inside the forward's tile-ranges launch this bracket is empty on
# the shipped path (it reads the event pair's own ~5 us); only the no-cull / no-hit-record A/B switches still clear there
STAGES = ["preprocess", "scan", "duplicate", "sort", "tile_ranges", "blend_fwd", "partials_memset", "blend_bwd",
          "gauss_bwd"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--device-warmup", type=float, default=1.0,
                    help="seconds of untimed steps in front of the --warmup steps (clock ramp of an idle GPU); 0 = none")
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1280)
    ap.add_argument("--semantics", type=int, default=0)
    ap.add_argument("--mode", choices=["strict", "exact", "fast"], default="strict",
                    help="which configuration the timed region (`value`) runs in.  strict (default): parity arithmetic on the "
                         "reference's own tile rects (sgr_test_switches bits 7 + 10) -- the configuration that meets north_star's "
                         "parity sentence as worded (binning arrays the reference's entry for entry, images / gradients within "
                         "rel 1e-4 end to end); exact: parity arithmetic on the cut-down tile lists (bit 7); fast: the library's "
                         "default arithmetic on the cut-down lists.  The other two are measured in extra regions of the same run "
                         "(value_strict / value_exact / value_fast)")
    ap.add_argument("--loss", choices=["grads", "scalar"], default="grads",
                    help="grads: backward from fixed upstream gradients; scalar: torch-built loss sum(out*w)")
    ap.add_argument("--reduce", choices=["factored", "bucket"], default="factored",
                    help="N > 1 exchange: factored = all-reduce of the dense parameter gradients + all-gather of per-view "
                         "dRGB with a local rebuild of dL/dSH (60 B/Gaussian on the wire); bucket = one flat all-reduce "
                         "of everything (236 B/Gaussian)")
    ap.add_argument("--exchange", choices=["overlap", "blocking"], default="blocking",
                    help="N > 1: blocking (default, the reported value) = the exchange completes inside the step, i.e. the "
                         "optimiser sees this step's summed gradients (north_star's semantics); overlap = the exchange runs on "
                         "a side stream under the NEXT step's forward (gradients one step late).  With the default the overlap "
                         "schedule is measured as an extra region and printed under `exchange_overlap`")
    ap.add_argument("--densify-every", type=int, default=10, help="configs[4] loop: one densify_and_prune every k iterations")
    ap.add_argument("--densify-loop", action="store_true",
                    help="N > 1: also run BASELINE configs[4] as worded -- 5 M replicated Gaussians, one view per rank, "
                         "gradient exchange every step, REPLICATED densify / prune every --densify-every steps -- and print it "
                         "under `configs4_densify_loop` (every rank takes part; off by default so that a scaling run stays short)")
    ap.add_argument("--densify-gaussians", type=int, default=5_000_000)
    ap.add_argument("--step-times", action="store_true", help="debug: also print 10 individually synchronised steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the 500k / 2M+S19 / 5M runs (N = 1 only)")
    ap.add_argument("--scene", default=None, help="a scene PLY in the reference's layout (street_gaussian_model.py:94-117, "
                    "read with street_gaussians_amd.plyio) to render instead of the synthetic Gaussians")
    ap.add_argument("--cpu-sample-gaussians", type=int, default=0, help="override the CPU sample size")
    return ap.parse_args()


def host_cpu():
    """(model name, {"threads": hardware threads, "sockets": packages, "physical_cores": distinct (package, core) pairs}) from
    /proc/cpuinfo: `cpu_baseline.cores` counts THREADS the oracle used; SMT siblings are not cores."""
    model, sockets, phys, threads = "unknown", set(), set(), 0
    try:
        pid = None
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name") and model == "unknown":
                    model = ln.split(":", 1)[1].strip()
                elif ln.startswith("processor"):
                    threads += 1
                elif ln.startswith("physical id"):
                    pid = ln.split(":", 1)[1].strip()
                    sockets.add(pid)
                elif ln.startswith("core id"):
                    phys.add((pid, ln.split(":", 1)[1].strip()))
    except OSError:
        pass
    return model, {"threads": threads or (os.cpu_count() or 1), "sockets": len(sockets) or None,
                   "physical_cores": len(phys) or None}


def cpu_baseline(args, cam, sc):
    """Times the oracle (port of the reference algorithm; the reference has no CPU path) with OpenMP on the host's
    cores, on a bounded sample of the SAME workload: first the central 1/16 window of the image (same focal length, so
    the depth complexity per pixel is the workload's; preprocess still visits all P Gaussians), and -- when that
    predicts at most ~30 s -- the full frame, which is then what is reported."""
    from oracle import oracle

    def run(Ws, Hs, c, reps):
        w = syn.loss_weights(c, S=0)
        kw = dict(means3D=sc.means3D, opacities=sc.opacities, viewmatrix=c.viewmatrix, projmatrix=c.projmatrix,
                  campos=c.campos, bg=torch.zeros(3), tanfovx=c.tanfovx, tanfovy=c.tanfovy, image_height=Hs,
                  image_width=Ws, sh_degree=3, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        best, R = None, 0
        for _ in range(reps):
            t0 = time.time()
            fw = oracle.forward(internals=False, **kw)
            oracle.backward(fw, w["color"], w["depth"], w["alpha"], None, parallel=True)
            dt = time.time() - t0
            R = fw.num_rendered
            fw.free()
            best = dt if best is None else min(best, dt)
            if dt > 20:
                break
        return best, R

    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))  # OpenMP THREADS the oracle runs on
    model, topo = host_cpu()
    Ws, Hs = max(16, args.width // 4), max(16, args.height // 4)
    small = syn.make_camera(Ws, Hs, fx=args.width / (2.0 * cam.tanfovx))
    t_small, R_small = run(Ws, Hs, small, 2)
    # one thread (SURVEY 8d asks for both): the central 1/64 window, so that it stays within ~10 s
    one = None
    try:
        gomp = C.CDLL("libgomp.so.1")
        W1, H1 = max(16, args.width // 8), max(16, args.height // 8)
        c1 = syn.make_camera(W1, H1, fx=args.width / (2.0 * cam.tanfovx))
        gomp.omp_set_num_threads(1)
        try:
            t1, R1 = run(W1, H1, c1, 1)
        finally:
            gomp.omp_set_num_threads(cores)
        one = {"seconds": round(t1, 3), "sample": f"central {W1}x{H1} window = 1/64 of the pixels (R={R1}), all {sc.P} "
                                                   "Gaussians preprocessed", "cores": 1,
               "full_frame_iters_per_s_estimate": round(1.0 / (64.0 * t1), 5)}
    except Exception as ex:
        one = {"error": str(ex)[:120]}
    if 16.0 * t_small <= 30.0:  # the whole frame fits the budget: report the real thing
        t_full, R_full = run(args.width, args.height, cam, 2)
        return {"value": round(1.0 / t_full, 4), "unit": "iters/s", "cores": cores, "threads": cores, "host": topo,
                "cores_note": "`cores` = OpenMP threads used (= hardware threads incl. SMT siblings); host.physical_cores / "
                              "host.sockets describe the box", "cpu_model": model, "kind": "port",
                "sample": f"oracle fwd+bwd (OpenMP), the full workload: all {sc.P} Gaussians, {args.width}x{args.height} "
                          f"(R={R_full}), best of 2", "seconds": round(t_full, 3),
                "window_1_16_seconds": round(t_small, 3), "one_thread": one}
    return {"value": round(1.0 / t_small, 4), "unit": "iters/s on the sample", "cores": cores, "threads": cores, "host": topo,
            "cpu_model": model, "kind": "port",
            "sample": f"oracle fwd+bwd (OpenMP), all {sc.P} Gaussians, central {Ws}x{Hs} window = 1/16 of the "
                      f"{args.width}x{args.height} pixels (R={R_small}); full-frame rate ~ value/16",
            "seconds": round(t_small, 3), "one_thread": one}


def cpu_smoke_recipe():
    """BASELINE.json configs[0] on the host's cores: the reference's own smoke test, script/test_gaussian_rasterization.py:44-91
    -- 10 k torch.rand Gaussians (SH degree 0 with M = 4, un-normalised quaternions), its pinhole camera, 256x256, "Test 1" a
    forward without semantics and "Test 2" one with 15 semantic channels -- through the C oracle (the reference has no CPU
    path of its own; tests/test_gpu_parity.py: test_smoke_recipe_config0 is the same recipe through the GPU path)."""
    import math
    from oracle import oracle
    c = syn.smoke_test_camera()
    g = torch.Generator().manual_seed(0)
    n, H, W = 10000, 256, 256
    means3D, _m2d = torch.rand(n, 3, generator=g), torch.rand(n, 3, generator=g)
    shs, opacity = torch.rand(n, 4, 3, generator=g), torch.rand(n, 1, generator=g)
    scales, rotations = torch.rand(n, 3, generator=g), torch.rand(n, 4, generator=g)
    rotations[:, 0] = 1
    semantics = torch.rand(n, 15, generator=g)
    out, R = {}, 0
    for name, sem in (("test1_no_semantics", None), ("test2_15_semantic_channels", semantics)):
        best = None
        for _ in range(3):
            t0 = time.time()
            fw = oracle.forward(means3D=means3D, opacities=opacity, viewmatrix=c["world_view_transform"],
                                projmatrix=c["full_proj_transform"], campos=c["camera_center"], bg=torch.zeros(3),
                                tanfovx=math.tan(c["FoVx"] * 0.5), tanfovy=math.tan(c["FoVy"] * 0.5), image_height=H,
                                image_width=W, sh_degree=0, shs=shs, scales=scales, rotations=rotations, semantics=sem,
                                internals=False)
            d = time.time() - t0
            R = fw.num_rendered
            fw.free()
            best = d if best is None else min(best, d)
        out[name + "_ms"] = round(1e3 * best, 2)
    out.update({"gaussians": n, "image": f"{W}x{H}", "sh_degree": 0, "num_rendered_R": int(R),
                "cores": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)),
                "what": "configs[0]: the forward calls of script/test_gaussian_rasterization.py:44-91 through the C oracle "
                        "(OpenMP), best of 3 each"})
    return out


def reference_kernels_on_gpu(args, cam, sc):
    """Times the reference's OWN kernels (untouched CUDA sources compiled for gfx950, oracle/_ref) on this GPU at
    the same workload: the most meaningful speed-up denominator.  Test infrastructure used as a
    reported baseline only; skipped when the .so did not travel.  Its three scratch buffers come from a pool that is
    warmed by the first call (ref_set_pooled: what torch's caching allocator does for the reference's own glue), so
    the timed calls contain no hipMalloc / hipFree; best of 4."""
    from oracle import ref
    if not ref.available():
        return None
    pooled = hasattr(ref.lib(), "ref_set_pooled")
    if pooled:
        ref.lib().ref_set_pooled(1)
    d = lambda t: t.detach().to("cuda").contiguous()  # inputs resident in HBM, like the timed HIP path
    w = {k: d(v) for k, v in syn.loss_weights(cam, S=0).items()}
    kw = dict(means3D=d(sc.means3D), opacities=d(sc.opacities), viewmatrix=d(cam.viewmatrix),
              projmatrix=d(cam.projmatrix), campos=d(cam.campos), bg=torch.zeros(3, device="cuda"), tanfovx=cam.tanfovx,
              tanfovy=cam.tanfovy, image_height=cam.image_height, image_width=cam.image_width, sh_degree=3,
              shs=d(sc.shs), scales=d(sc.scales), rotations=d(sc.rotations))
    best = None
    for it in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rf = ref.forward(**kw)
        ref.backward(rf, w["color"], w["depth"], w["alpha"], None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rf.free()
        if it > 0:  # the first call sizes the pool
            best = dt if best is None else min(best, dt)
    if pooled:
        ref.lib().ref_set_pooled(0)
    return {"ms_per_step": round(1e3 * best, 3), "iters_per_s": round(1.0 / best, 3),
            "what": "reference CUDA kernels (forward.cu/backward.cu/rasterizer_impl.cu + hipCUB) compiled unmodified "
                    "for gfx950, same inputs resident in HBM, scratch " + ("pre-allocated (pooled)" if pooled else
                    "hipMalloc'ed per call") + "; includes its gradient zero-fills and its blocking D2H of num_rendered"}


class Workload:
    """One rasterizer workload resident in HBM: `step()` = forward + backward (+ the exchange step when N > 1)."""

    def __init__(self, args, P, S, rank, dev, dist=None, force_dist=False, scene_file=None, scene=None):
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
        from street_gaussians_amd import multiview
        self.args, self.P, self.S, self.dev = args, P, S, dev
        W, H = args.width, args.height
        # identical Gaussians on every rank (seed 0); rank r renders view r (yawed r*5 degrees)
        self.cam0 = syn.make_camera(W, H, fx=2050.0 * W / 1920.0)
        if scene is not None:  # rasterizer inputs somebody composed already (street_config)
            self.scene = scene
            self.P = P = scene.P
        elif scene_file:
            self.scene = load_scene_file(scene_file, S)
            self.P = P = self.scene.P
        else:
            self.scene = syn.make_scene(P, self.cam0, sh_degree_max=3, S=S, seed=0)
        scene = self.scene
        self.cam = cam = syn.make_camera(W, H, fx=2050.0 * W / 1920.0, yaw_deg=5.0 * rank)
        self.params = {k: getattr(scene, k).to(dev).requires_grad_(True)
                       for k in ["means3D", "scales", "rotations", "opacities", "shs"]}
        if S:
            self.params["semantics"] = scene.semantics.to(dev).requires_grad_(True)
        self.means2D = torch.zeros(scene.P, 3, device=dev, requires_grad=True)
        self.w = {k: v.to(dev) for k, v in syn.loss_weights(cam, S=S, seed=1 + rank).items()}
        self.st = GaussianRasterizationSettings(
            image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
            bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam.viewmatrix.to(dev),
            projmatrix=cam.projmatrix.to(dev), sh_degree=3, campos=cam.campos.to(dev), prefiltered=False, debug=False)
        self.rast = GaussianRasterizer(self.st)
        self.radii = None
        # exchange step (N > 1): the optimiser's gradients.  dL/dmeans2D is not one of them -- it feeds per-view
        # densification statistics (multiview.reduce_densification_stats) -- so it stays local.
        self.reducer = None
        params = self.params
        if dist is not None:
            def bucket():
                return multiview.GradReducer(list(params.values()), force=force_dist)
            if args.reduce == "factored":
                # no silent downgrade: a broken factored exchange must fail the run (a SCALE line quoted on the 236
                # B/Gaussian bucket while the config says "factored" would be a wrong number)
                dense = [v for k, v in params.items() if k != "shs"]
                self.reducer = multiview.FactoredGradReducer(dense, params["shs"], params["means3D"], force=force_dist)
            else:
                self.reducer = bucket()
            self.reducer.warm_up()  # RCCL's lazy communicator / channel set-up is not part of a training step

    def step(self):
        p, w, S = self.params, self.w, self.S
        for t in list(p.values()) + [self.means2D]:
            t.grad = None
        color, radii, depth, alpha, sem = self.rast(p["means3D"], self.means2D, p["opacities"], shs=p["shs"],
                                                    scales=p["scales"], rotations=p["rotations"],
                                                    semantics=p.get("semantics"))
        if self.reducer is not None and self.args.exchange == "overlap":
            # the previous step's exchange ran under the forward that was just queued; its result is due now (the
            # optimiser step would sit here), then the gradient slots are cleared for this step's backward
            self.reducer.wait()
            for t in list(p.values()):
                t.grad = None
        if self.args.loss == "scalar":
            loss = (color * w["color"]).sum() + (depth * w["depth"]).sum() + (alpha * w["alpha"]).sum()
            if S:
                loss = loss + (sem * w["semantic"]).sum()
            loss.backward()
        else:
            outs, grads = [color, depth, alpha], [w["color"], w["depth"], w["alpha"]]
            if S:
                outs.append(sem)
                grads.append(w["semantic"])
            torch.autograd.backward(outs, grads)
        if self.reducer is not None:
            if self.args.exchange == "overlap":
                self.reducer.begin()
            else:
                self.reducer.all_reduce()
        self.radii = radii

    def drain(self):
        """Joins an exchange that is still in flight (end of a timed region)."""
        if self.reducer is not None:
            self.reducer.wait()

    def counts(self):
        """R (tile instances), V (visible Gaussians) and the sum of n_contrib of this view (R/P and V/P are
        reported with every number).  R is the REFERENCE's num_rendered -- the unit SURVEY 8d's byte formulas are written in;
        the library emits fewer instances (tile rects cut down to where alpha >= 1/255 is possible): self.R_emitted."""
        from street_gaussians_amd import _C as native
        st, p = self.st, self.params
        V = int((self.radii > 0).sum().item())
        out = native.rasterize_gaussians(st.bg, p["means3D"].detach(), torch.Tensor([]),
                                         torch.zeros(self.P, 0, device=self.dev), p["opacities"].detach(),
                                         p["scales"].detach(), p["rotations"].detach(), 1.0, torch.Tensor([]),
                                         st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy, st.image_height,
                                         st.image_width, p["shs"].detach(), 3, st.campos, False, False)
        self.R_emitted = int(out[0])
        exp = lambda name: native.export_internal(name, self.P, self.R_emitted, self.args.height, self.args.width, out[6], out[7], out[8])
        n_contrib = exp("n_contrib")
        R = int(exp("num_rendered_reference")[0].item())
        self.V_in = int(native.mark_visible(p["means3D"].detach(), st.viewmatrix, st.projmatrix).sum().item())
        return R, V, int(n_contrib.to(torch.int64).sum().item())


class GpuLoopBackend:
    """What DensifyLoop runs on: the product path (HIP rasterizer, fused densify kernels, CUDA events).  A test can hand the
    loop another backend with the same members -- tests/test_multiview_gloo.py runs the unchanged control flow (steps,
    statistics reduce, replicated densify, reducer rebuild, thresholds) on CPU tensors across two gloo ranks."""
    reducer_kw = {}  # extra keyword arguments of multiview.FactoredGradReducer (a CPU backend supplies mask_fn / rebuild_fn)

    def __init__(self, dev):
        self.device = dev

    def rasterizer(self, settings):
        from diff_gaussian_rasterization import GaussianRasterizer
        return GaussianRasterizer(settings)

    def settings(self, **kw):
        from diff_gaussian_rasterization import GaussianRasterizationSettings
        return GaussianRasterizationSettings(**kw)

    def densify_and_prune(self, *a, **kw):
        from street_gaussians_amd import densify
        return densify.densify_and_prune(*a, **kw)

    def last_num_rendered(self):
        from street_gaussians_amd import rasterizer
        return rasterizer.last_num_rendered()

    def sync(self):
        torch.cuda.synchronize()

    def mark(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def elapsed_ms(self, a, b):
        return a.elapsed_time(b)

    def reserve(self, n_points):
        from street_gaussians_amd import densify
        self.pool = densify.Pool(self.device, factor=1.25, instances_per_point=8.0)
        nbytes = self.pool.reserve(n_points)
        torch.cuda.reset_peak_memory_stats(self.device)
        return nbytes

    def allocator(self):
        st = torch.cuda.memory_stats(self.device)
        return {k: int(st.get(k, 0)) for k in ("num_device_alloc", "num_alloc_retries", "num_ooms", "num_device_free",
                                               "reserved_bytes.all.peak", "allocated_bytes.all.peak")}

    def host_wait_us(self, reset):
        from street_gaussians_amd import _native
        return _native.lib().sgr_profile_host_wait_us(reset)


class DensifyLoop:
    """BASELINE.json configs[4] as written: 5 M Gaussians, 1920x1280, SH degree 3, with the densify / prune step ACTIVE
    between iterations, so that P, R and the three scratch buffers change under load ("one synthetic densify
    step (clone 5 % + split 5 %, prune 5 %) between iterations so buffers are re-sized").

    The loop is the reference's (train.py:187-210): every iteration the rasterizer forward + backward, with
    `set_max_radii2D` + `add_densification_stats` applied by the backward's fused statistics sink
    (GaussianRasterizer.stats_sink = scene.FlatStats.sink(), street_gaussian_model.py:551-571); every
    `densify_every`-th iteration `densify_and_prune` (gaussian_model.py:522-553) through
    street_gaussians_amd.densify.densify_and_prune on the RAW parameters and their Adam moments, after which the
    activations are re-applied (exp / sigmoid / normalize, gaussian_model.py:224-251), the statistics are re-created
    as zeros (:545-547) and the next forward runs on the changed P.  Thresholds (max_grad, percent_dense * extent,
    min_opacity) are quantiles of the statistics of an untimed calibration interval, chosen for ~5 % clones, ~5 % splits
    and ~5 % pruned per step.  reset_opacity (train.py:207-208) is exercised by tests/test_gpu_densify_loop.py, not here:
    it makes every splat transparent, i.e. it would replace the stress workload by a trivial one."""

    def __init__(self, args, P, dev, densify_every, dist=None, rank=0, force_dist=False, backend=None):
        """dist != None (N > 1, or a forced one-rank group): the configuration as BASELINE.json words it -- Gaussians
        replicated, one camera view per rank, gradients exchanged every step, and the densify step REPLICATED: per-view
        statistics combined across ranks, the same densify_and_prune with the same normals on every rank
        (multiview.densify_replicated), reducers rebuilt for the new P."""
        self.args, self.dev, self.every = args, dev, max(1, densify_every)
        self.dist, self.rank, self.force_dist = dist, rank, force_dist
        self.be = be = backend if backend is not None else GpuLoopBackend(dev)
        self.reducer = self.normals = None
        W, H = args.width, args.height
        cam0 = syn.make_camera(W, H, fx=2050.0 * W / 1920.0)
        cam = self.cam = syn.make_camera(W, H, fx=2050.0 * W / 1920.0, yaw_deg=5.0 * rank)
        sc = syn.make_scene(P, cam0, sh_degree_max=3, S=0, seed=0)  # identical Gaussians on every rank
        d = lambda t: t.to(dev).contiguous()
        op = sc.opacities.clamp(1e-6, 1 - 1e-6)
        # raw parameters as the reference's GaussianModel stores them (gaussian_model.py:120-127)
        self.params = {"xyz": d(sc.means3D), "f_dc": d(sc.shs[:, :1, :]), "f_rest": d(sc.shs[:, 1:, :]),
                       "opacity": d(torch.log(op / (1 - op))), "scaling": d(torch.log(sc.scales)), "rotation": d(sc.rotations)}
        self.states = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in self.params.items()}
        self.w = {k: v.to(dev) for k, v in syn.loss_weights(cam, S=0, seed=1 + rank).items()}
        self.st = be.settings(
            image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(3, device=dev),
            scale_modifier=1.0, viewmatrix=cam.viewmatrix.to(dev), projmatrix=cam.projmatrix.to(dev), sh_degree=3,
            campos=cam.campos.to(dev), prefiltered=False, debug=False)
        self.rast = be.rasterizer(self.st)
        self.kw = None
        self.log = []
        self.max_R = 0
        self._activate()
        if dist is not None:
            from street_gaussians_amd import multiview
            i = self.inputs
            self.reducer = multiview.FactoredGradReducer([i[k] for k in ("means3D", "scales", "rotations", "opacities")],
                                                         i["shs"], i["means3D"], force=force_dist, **be.reducer_kw)
            self.reducer.warm_up()
            self.normals = multiview.ReplicatedNormals(seed=17)

    def _activate(self):
        from street_gaussians_amd import scene
        p = self.params
        self.P = p["xyz"].shape[0]
        self.inputs = {"means3D": p["xyz"], "scales": torch.exp(p["scaling"]),
                       "rotations": torch.nn.functional.normalize(p["rotation"]), "opacities": torch.sigmoid(p["opacity"]),
                       "shs": torch.cat([p["f_dc"], p["f_rest"]], 1).contiguous()}
        for t in self.inputs.values():
            t.requires_grad_(True)
        self.means2D = torch.zeros(self.P, 3, device=self.dev, requires_grad=True)
        self.stats = scene.FlatStats([self.P], self.dev)
        self.rast.stats_sink = self.stats.sink()

    def step(self):
        i, w = self.inputs, self.w
        for t in list(i.values()) + [self.means2D]:
            t.grad = None
        color, radii, depth, alpha, _ = self.rast(i["means3D"], self.means2D, i["opacities"], shs=i["shs"],
                                                  scales=i["scales"], rotations=i["rotations"])
        torch.autograd.backward([color, depth, alpha], [w["color"], w["depth"], w["alpha"]])
        if self.reducer is not None:
            self.reducer.all_reduce()  # blocking exchange: the step's summed gradients
        self.max_R = max(self.max_R, self.be.last_num_rendered())

    def calibrate(self):
        """Thresholds from the statistics accumulated so far (untimed): ~10 % of the points over max_grad, half of those
        small enough to be cloned, ~5 % of the points under min_opacity."""
        st, p = self.stats, self.params
        g = (st.xyz_gradient_accum[:, 0:1] / st.denom).nan_to_num(0.0).flatten()
        q = lambda t, f: float(torch.kthvalue(t.float().flatten(), max(1, int(f * t.numel()))).values)
        max_grad = q(g, 0.90)
        sel = g >= max_grad
        big = torch.exp(p["scaling"]).max(dim=1).values
        thr = q(big[sel], 0.5) if int(sel.sum()) else 1.0
        self.kw = dict(max_grad=max_grad, min_opacity=q(torch.sigmoid(p["opacity"]), 0.05), extent=1.0,
                       percent_dense=thr, percent_big_ws=1e9, prune_big=False)

    def densify(self):
        sync = self.be.sync
        sync()
        t_r = time.perf_counter()
        if self.dist is not None:
            # replicated densify, step 1: the per-view statistics of all ranks become one (sums / max) -- part of the path
            from street_gaussians_amd import multiview
            multiview.reduce_densification_stats(self.stats.xyz_gradient_accum, self.stats.denom, self.stats.max_radii2D)
            sync()
        t_c = time.perf_counter()
        self.calibrate()  # this synthetic schedule's thresholds (quantiles): not part of the path, excluded from the totals
        sync()     # (computed from the combined statistics: the same thresholds on every rank)
        t0 = time.perf_counter()
        self.calib_s += t0 - t_c
        t0 -= t_c - t_r  # the statistics reduce counts as densify time
        # step 2: the same densify_and_prune on every rank, with the same normals (multiview.ReplicatedNormals)
        new_p, new_s, scal, _ = self.be.densify_and_prune(self.params, self.stats.xyz_gradient_accum, self.stats.denom,
                                                          states=self.states, normal_source=self.normals, **self.kw)
        self.params, self.states = new_p, new_s
        self._activate()
        if self.reducer is not None:
            i = self.inputs
            self.reducer.rebuild([i[k] for k in ("means3D", "scales", "rotations", "opacities")], i["shs"], i["means3D"])
        sync()
        scal = dict(scal)
        scal["ms"] = round(1e3 * (time.perf_counter() - t0), 3)
        scal["gaussians_after"] = self.P
        self.log.append(scal)

    def _replicas_identical(self):
        """After the run: are the raw parameters and Adam moments bit-identical on all ranks?  (None with one process.)"""
        if self.dist is None:
            return None
        from street_gaussians_amd import multiview
        ts = [self.params[k] for k in sorted(self.params)] + [t for k in sorted(self.states) for t in self.states[k]]
        return bool(multiview.replicas_identical(ts))

    def run(self, fence, n_densify=3):
        every = self.every
        # A densify step re-sizes every buffer; the library's pool (street_gaussians_amd.densify.Pool) reserves 1.25 x the
        # bytes live at the size the run will reach and hands them to torch's caching allocator (whose requests repeat thanks
        # to the size ladder, street_gaussians_amd/_alloc.py), so that no device allocation happens inside the loop
        # (round 4: a fixed 48 GB block reserved here)
        be = self.be
        pool_bytes = be.reserve(int(self.P * 1.25))
        for _ in range(every):  # untimed calibration interval (also the warm-up)
            self.step()
        self.calibrate()
        P0, self.max_R, self.calib_s = self.P, 0, 0.0
        steps = every * n_densify
        marks = [None] * (steps + 1)
        allocs0 = be.allocator().get("num_device_alloc", 0)
        host_s = 0.0
        fence()
        be.host_wait_us(1)
        t0 = time.perf_counter()
        for it in range(steps):
            marks[it] = be.mark()
            th = time.perf_counter()
            self.step()
            host_s += time.perf_counter() - th  # wall time of the call; the forward's wait for the GPU is subtracted below
            if it == steps - 1:
                marks[steps] = be.mark()
            if (it + 1) % every == 0:
                self.densify()
        fence()
        dt = time.perf_counter() - t0 - self.calib_s
        wait_s = 1e-6 * float(be.host_wait_us(1))  # the host idling at the read-back of num_rendered
        d_ms = sum(x["ms"] for x in self.log)
        # per-iteration GPU time from the event marks: the iteration right after a densify step runs on re-sized buffers
        # (geometry / binning / backward scratch / gradients no longer fit the allocator's cached blocks), the others are
        # steady state
        step_ms = []
        for it in range(steps):
            if (it + 1) % every == 0 and it != steps - 1:
                step_ms.append(None)  # the interval to the next mark contains the densify step
            else:
                step_ms.append(be.elapsed_ms(marks[it], marks[it + 1]))
        first_after = [step_ms[it] for it in range(steps) if it % every == 0 and it > 0 and step_ms[it] is not None]
        steady = sorted(v for it, v in enumerate(step_ms) if v is not None and not (it % every == 0 and it > 0))
        return {"config": "configs[4] 5M + densify/prune active in the loop (train.py:187-210)", "gaussians_start": P0,
                "gaussians_end": self.P, "steps": steps, "densify_every": every, "densify_steps": len(self.log),
                "ms_per_step_amortised": round(1e3 * dt / steps, 4), "iters_per_s_amortised": round(steps / dt, 3),
                "raster_ms_per_step": round((1e3 * dt - d_ms) / steps, 4),
                "raster_ms_steady_median": round(steady[len(steady) // 2], 4) if steady else None,
                "raster_ms_first_iteration_after_densify": [round(v, 3) for v in first_after],
                "densify_ms_mean": round(d_ms / max(1, len(self.log)), 3), "densify_log": self.log,
                "host_ms_to_queue_one_iteration": round(1e3 * max(0.0, host_s - wait_s) / steps, 3),
                "host_ms_in_step_call_incl_gpu_wait": round(1e3 * host_s / steps, 3),
                "host_note": "host_ms_to_queue_one_iteration = wall time inside step() minus the time the forward spent waiting for "
                             "the GPU at its read-back of num_rendered (sgr_profile_host_wait_us): the host's own work per iteration; "
                             "round 4 printed the wall time, which contains the GPU's previous backward",
                "pool_bytes_reserved": int(pool_bytes),
                "device_allocations_in_region": int(be.allocator().get("num_device_alloc", 0) - allocs0),
                "allocator": be.allocator(),
                "ranks": (self.dist.get_world_size() if self.dist is not None else 1),
                "replicas_identical": self._replicas_identical(),
                "max_num_rendered_R": self.max_R, "thresholds_last": {k: (round(v, 6) if isinstance(v, float) else v) for k, v in self.kw.items()},
                "note": "rasterizer forward+backward every iteration with the fused densification-statistics sink; "
                        "densify_and_prune (plan + gather kernels, raw parameters + 12 Adam moment tensors) every "
                        f"{every}th iteration, then exp/sigmoid/normalize re-activation; P, R and the geometry / binning / "
                        "backward scratch buffers change size at each of those steps (the iteration after a densify step "
                        "pays the allocator's fresh device allocations: listed separately); the thresholds are re-derived "
                        "from quantiles before every densify step (~5 % clones, ~5 % splits, ~5 % pruned) -- that "
                        "synthetic-schedule work is excluded from every figure"}


def load_scene_file(path, S):
    """A scene in the reference's on-disk layout (one `vertex_<model>` element per sub-model,
    street_gaussian_model.py:94-117; or a .pth checkpoint, train.py:218-223), flattened to rasterizer inputs: activations
    applied as the model's getters do (gaussian_model.py:224-251), all sub-models concatenated, actors left in their
    stored (object) frame, the first Fourier coefficient as their DC colour."""
    from street_gaussians_amd import checkpoint, plyio
    models = checkpoint.load(path) if path.endswith(".pth") else plyio.read_scene_ply(path)
    ten = lambda v: torch.as_tensor(v).float()
    cat = lambda k: torch.cat([ten(m[k]) for m in models.values()], 0)
    xyz = cat("xyz")
    dc = torch.cat([ten(m["features_dc"])[:, :1, :] for m in models.values()], 0)
    shs = torch.cat([dc, cat("features_rest")], 1)
    sem = torch.zeros(xyz.shape[0], S)
    rot = cat("rotation")
    return syn.Scene(xyz.contiguous(), torch.exp(cat("scaling")).contiguous(),
                     (rot / rot.norm(dim=1, keepdim=True)).contiguous(), torch.sigmoid(cat("opacity")).contiguous(),
                     shs.contiguous(), sem.contiguous())


def profiled_steps(L, wl, fence, steps, stage_mask, every=1):
    """Runs `steps` steps with the chosen stages bracketed by HIP events on every `every`-th step (an event pair
    drains the queue around the launch it brackets, ~10 us each side: the timed region samples); returns
    (seconds, {stage: mean ms})."""
    L.sgr_profile_select(stage_mask)
    L.sgr_profile_sample(every)
    L.sgr_profile_enable(1)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if os.environ.get("SGR_BENCH_REGION_TRACE") else None
    fence()
    t0 = time.perf_counter()
    host, waits = [], []
    for i in range(steps):
        if marks:
            marks[i].record()
            L.sgr_profile_host_wait_us(1)
            th = time.perf_counter()
        wl.step()
        if marks:
            host.append(round(1e3 * (time.perf_counter() - th), 3))
            waits.append(round(1e-3 * L.sgr_profile_host_wait_us(1), 3))
    wl.drain()  # the last step's exchange belongs to the timed region
    if marks:
        marks[steps].record()
    fence()
    dt = time.perf_counter() - t0
    if marks:  # debug: where a short region loses time (GPU time between the starts of consecutive steps)
        print("[region-trace]", steps, round(1e3 * dt, 3), "gpu", [round(marks[i].elapsed_time(marks[i + 1]), 3) for i in range(steps)][:24],
              "host_call", host[:24], "of_which_wait", waits[:24], file=sys.stderr)
    sums = (C.c_double * 9)()
    counts = (C.c_int * 9)()
    L.sgr_profile_read(sums, counts)
    L.sgr_profile_enable(0)
    L.sgr_profile_select(0x1FF)
    L.sgr_profile_sample(1)
    return dt, {STAGES[i]: (sums[i] / counts[i] if counts[i] else None) for i in range(9)}


def count_visits(wl, R):
    """Contributing (quadrant, instance) visits of this view = what the blend backward walks: the bits of the forward's hit
    record at list positions below the tile's highest n_contrib (entries behind the last batch a tile processed are
    undefined).  Untimed; torch ops on the exported internals."""
    from street_gaussians_amd import _C as native
    st, p = wl.st, wl.params
    H, W = wl.args.height, wl.args.width
    out = native.rasterize_gaussians(st.bg, p["means3D"].detach(), torch.Tensor([]), torch.zeros(wl.P, 0, device=wl.dev),
                                     p["opacities"].detach(), p["scales"].detach(), p["rotations"].detach(), 1.0, torch.Tensor([]),
                                     st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy, st.image_height, st.image_width,
                                     p["shs"].detach(), 3, st.campos, False, False)
    R = int(out[0])
    exp = lambda name: native.export_internal(name, wl.P, R, H, W, out[6], out[7], out[8])
    nc = exp("n_contrib").view(H, W).to(torch.int64)
    gy, gx = (H + 15) // 16, (W + 15) // 16
    pad = torch.zeros(gy * 16, gx * 16, dtype=torch.int64, device=nc.device)
    pad[:H, :W] = nc
    maxc = pad.view(gy, 16, gx, 16).amax(dim=(1, 3)).reshape(-1)
    rg = exp("ranges").view(-1, 2).to(torch.int64)
    cnt = rg[:, 1] - rg[:, 0]
    tile = torch.repeat_interleave(torch.arange(gy * gx, device=nc.device), cnt)
    pos = torch.arange(R, device=nc.device) - rg[tile, 0]
    hit = exp("hits")[:R].to(torch.int64)
    bits = (hit & 1) + ((hit >> 1) & 1) + ((hit >> 2) & 1) + ((hit >> 3) & 1)
    return int(bits[pos < maxc[tile]].sum().item())


def blend_bytes(S, R, N, V):
    """SURVEY 8d algorithmic bytes per launch: B_blend_b, B_blend_f."""
    return (44 + 4 * S) * R + (28 + 4 * S) * N + (48 + 4 * S) * V, (44 + 4 * S) * R + (24 + 4 * S) * N


def stage_bytes(P, V_in, V, R, N, T, S, M=16):
    """SURVEY 8d algorithmic (compulsory) bytes of every stage of one forward + backward: B_pre, B_scan, B_dup, B_sort
    (read + write once; the depth pre-sort of this implementation is part of the sort's job, its time is reported under
    "scan" next to B_scan), B_rng, B_blend_f, B_blend_b, B_pre_b.  B_zero does not apply (no gradient zero fills)."""
    bb, bf = blend_bytes(S, R, N, V)
    return {"preprocess": 12 * P + (28 + 12 * M) * V_in + 79 * V, "scan": 8 * P, "duplicate": 12 * R, "sort": 24 * R,
            "tile_ranges": 8 * R + 8 * T, "blend_fwd": bf, "blend_bwd": bb,
            "gauss_bwd": (111 + 12 * M) * V + (64 + 12 * M) * V}


def stage_hbm_frac(stage_ms, sb):
    """Per stage: SURVEY 8d bytes / measured time, as a fraction of the 8 TB/s HBM peak (+ the whole step)."""
    out = {}
    for k, b in sb.items():
        ms = stage_ms.get(k)
        out[k] = round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms else None
    tot_ms = sum(v for k, v in stage_ms.items() if v and k in sb)
    out["all_stages"] = round(sum(sb.values()) / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tot_ms else None
    return out


def launch_plan(n_gpus, env, device_count, argv, port=None):
    """What `bench.py --gpus N` does about its ranks.  Returns None when this process IS a rank (N = 1, or started by
    torch.distributed.run: WORLD_SIZE == N), else the command line that re-executes bench.py as N ranks, one per GPU, under
    torch.distributed.run on this node (rendezvous on 127.0.0.1).  Raises SystemExit (rc != 0) when N GPUs do not exist or
    the launcher's world size contradicts --gpus: a run must never print a line for fewer ranks than it was asked for."""
    world = env.get("WORLD_SIZE")
    if world is not None:
        if int(world) != n_gpus:
            raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world}")
        return None
    if n_gpus < 1:
        raise SystemExit(f"--gpus {n_gpus}: need at least one GPU")
    share = bool(env.get("SGR_BENCH_SHARE_GPU"))
    if share and env.get("SGR_BENCH_BACKEND", "nccl") != "gloo":
        raise SystemExit("SGR_BENCH_SHARE_GPU=1 (all ranks on cuda:0: a TEST of the N > 1 rank program on a one-GPU box) needs "
                         "SGR_BENCH_BACKEND=gloo -- RCCL refuses two ranks on one device")
    if device_count < n_gpus and not (share and device_count >= 1):
        raise SystemExit(f"--gpus {n_gpus} but only {device_count} GPU(s) are visible: refusing to run on fewer ranks")
    if n_gpus == 1:
        return None
    if port is None:
        import socket
        with socket.socket() as sk:  # a free port for the rendezvous
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main():
    args = parse()
    plan = launch_plan(args.gpus, os.environ, torch.cuda.device_count() if torch.cuda.is_available() else 0, sys.argv[1:])
    if plan is not None:  # `python bench.py --gpus N` without a launcher: spawn the N ranks ourselves
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL across processes
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, args.gpus))))
        raise SystemExit(subprocess.call(plan, env=env))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    # SGR_BENCH_SHARE_GPU=1 + SGR_BENCH_BACKEND=gloo: every rank on cuda:0, collectives over gloo (CUDA tensors staged through
    # the host) -- the N > 1 rank program end to end on a one-GPU box (tests/test_gpu_multiview.py); never a measurement
    backend = os.environ.get("SGR_BENCH_BACKEND", "nccl")
    share_gpu = bool(os.environ.get("SGR_BENCH_SHARE_GPU"))
    if share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    force_dist = bool(os.environ.get("SGR_BENCH_FORCE_DIST"))  # exercise the RCCL path on a single GPU (testing)
    if world > 1 or force_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29511"
            os.environ.setdefault("RANK", "0"), os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from street_gaussians_amd import _native
    from street_gaussians_amd import build as sgr_build

    S = args.semantics
    # the configuration of the timed region (every rank sets the same process-wide switches)
    from street_gaussians_amd import _C as native_c
    MODE_MASK = {"strict": native_c.EXACT | native_c.REF_RECT, "exact": native_c.EXACT, "fast": 0}
    base_switches = native_c.test_switches(-1) & ~(native_c.EXACT | native_c.REF_RECT)
    native_c.test_switches(base_switches | MODE_MASK[args.mode])
    wl = Workload(args, args.gaussians, S, rank, dev, dist=dist, force_dist=force_dist, scene_file=args.scene)
    args.gaussians = wl.P
    reducer = wl.reducer

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed, in front of the W warm-up steps: bring the device to its working clocks.  A fresh box starts the run with an
    # idle GPU (the host spent the last seconds generating the scene) and `--warmup 5` is 7 ms of work: the first tens of
    # milliseconds run at ramping clocks (measured: the same build reads 650 it/s in a 20-step region right after start-up and
    # 688 it/s a second later).  Reported as `device_warmup_s`; the timed region is still exactly --steps steps after exactly
    # --warmup steps.
    L = _native.lib()
    # the stage timer's events are created here, in front of the warm-up: their creation (2048 events, each recorded once,
    # a device synchronisation) idles the GPU for tens of milliseconds, after which it needs ~10 steps to come back to its
    # clocks -- round 4 did this between the warm-up steps and the timed region (the 20-step region read 4 % slow)
    L.sgr_profile_enable(1)
    L.sgr_profile_enable(0)
    t_w = time.perf_counter()
    n_w = 0
    while args.device_warmup > 0 and time.perf_counter() - t_w < args.device_warmup:
        wl.step()
        n_w += 1
        if n_w % 32 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        wl.step()
    # ---- the timed region: EXACTLY args.steps steps between two fences; only the dominant kernel carries events
    # (at most every 8th step whatever --steps is: an event pair drains the queue ~10 us each side of the launch it
    # brackets, which taxed a 20-step region by ~1 % when every step carried one; at least 3 samples)
    every = max(1, min(8, (args.steps - 1) // 2))
    dt, timed = profiled_steps(L, wl, fence, args.steps, 1 << 7, every=every)
    if os.environ.get("SGR_BENCH_REGION_TRACE"):  # debug: the same region again, and once more after 50 ms of host sleep
        profiled_steps(L, wl, fence, args.steps, 1 << 7, every=every)
        time.sleep(0.05)
        profiled_steps(L, wl, fence, args.steps, 1 << 7, every=every)
        profiled_steps(L, wl, fence, args.steps, 0, every=every)
    if args.step_times:
        ts = []
        for _ in range(10):
            fence()
            t1 = time.perf_counter()
            wl.step()
            fence()
            ts.append(round(1e3 * (time.perf_counter() - t1), 3))
        print(f"[step-times] {ts}", file=sys.stderr)
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    # after a blocking exchange every rank holds the SAME summed gradients, bit for bit (checksums, MIN == MAX over ranks)
    grads_identical = None
    if dist is not None and reducer is not None and args.exchange == "blocking":
        from street_gaussians_amd import multiview
        gl = [p.grad for p in wl.params.values()]
        grads_identical = bool(all(g is not None for g in gl) and multiview.replicas_identical(gl))

    # ---- N > 1, extra: the OTHER exchange schedule on the same workload (the reported value is the blocking one: the
    # optimiser sees this step's summed gradients; "overlap" hides the exchange under the next step's forward, i.e. the
    # gradients arrive one step late).  Every rank runs this region (it contains the collectives).
    exchange_overlap = None
    if reducer is not None and args.exchange == "blocking":
        args.exchange = "overlap"
        try:
            for _ in range(3):
                wl.step()
            wl.drain()
            odt, _ = profiled_steps(L, wl, fence, args.steps, 0)
            to = torch.tensor([odt], device=dev, dtype=torch.float64)
            if dist is not None:
                dist.all_reduce(to, op=dist.ReduceOp.MAX)
            odt = float(to.item())
            exchange_overlap = {"value": round(world * args.steps / odt, 3), "ms_per_step": round(1e3 * odt / args.steps, 4),
                                "schedule": "exchange on a side stream under the NEXT step's forward: one-step-delayed "
                                            "gradients in training; NOT the reported value"}
        finally:
            args.exchange = "blocking"
            wl.drain()

    # ---- untimed extras: every stage bracketed (a short pass), a >= 1 s sustained region
    _, stage_ms = profiled_steps(L, wl, fence, min(50, max(5, args.steps)), 0x1FF)
    sustained = None
    kernel_samples = (args.steps + every - 1) // every
    kernel_region = f"the {args.steps} timed steps"
    if world == 1 and dist is None and dt < 1.0:
        # >= 1 s of back-to-back steps: a K-step region of a few tens of ms pays its opening fence (an empty queue, one
        # exposed host wait, clocks that dipped during the synchronisation) once per K steps.  The dominant kernel is
        # bracketed on every 8th step here as well, so that a short --steps still has >= 16 duration samples.
        n = int(1.2 / max(dt / args.steps, 1e-5)) + 1
        sdt, s_timed = profiled_steps(L, wl, fence, n, 1 << 7, every=8)
        sustained = {"steps": n, "seconds": round(sdt, 3), "ms_per_step": round(1e3 * sdt / n, 4),
                     "iters_per_s": round(n / sdt, 3)}
        if kernel_samples < 16 and s_timed.get("blend_bwd"):
            timed = dict(timed, blend_bwd=s_timed["blend_bwd"])
            kernel_samples, kernel_region = (n + 7) // 8, f"the sustained region of {n} steps"
    # The three configurations, measured on the same workload.  The timed region above ran in `args.mode` (default strict);
    # the other two get regions of their own so that their cost is in the line next to `value`:
    #   strict = sgr_test_switches bits 7 + 10 (SGR_EXACT=1 SGR_REF_RECT=1): the forward with the reference's power expression,
    #            the device library's expf and the unfused depth / alpha / semantic sums; the backward with the reference's power
    #            expression and every blend / skip decision guarded to be the forward's; K12/K13 without FP contraction; AND the
    #            reference's own tile rects, so that num_rendered / point_list / keys / ranges / n_contrib are the reference's
    #            ENTRY FOR ENTRY, alpha / depth / semantic images bit-identical to the reference's strict build, every gradient
    #            within rel 1e-4 END TO END (tests/test_gpu_fullsize.py: test_threeway_against_reference_kernels_at_baseline_size)
    #   exact  = bit 7 alone: the same arithmetic on the cut-down tile lists (the reference's lists minus the instances that
    #            cannot blend);
    #   fast   = the library's default arithmetic (v_exp_f32 on a pre-scaled conic, v_rcp_f32, contraction) on the cut-down
    #            lists: same algorithm, different last bits.
    modes = {}
    if world == 1 and dist is None:
        for label in ("strict", "exact", "fast"):
            mask = MODE_MASK[label]
            native_c.test_switches(base_switches | mask)
            try:
                for _ in range(5):
                    wl.step()
                if label == args.mode:  # the headline region itself; only its stage pass is added here
                    n_p, pdt = args.steps, dt
                else:
                    n_p = max(args.steps, int(0.6 / max(dt / args.steps, 1e-5)) + 1)  # the same kind of region as `sustained`
                    pdt, _ = profiled_steps(L, wl, fence, n_p, 0)
                _, p_stage = profiled_steps(L, wl, fence, 16, 0x1FF)
                wl.counts()  # under the switch: what this mode emits (wl.R_emitted)
                modes[label] = {"steps": n_p, "ms_per_step": round(1e3 * pdt / n_p, 4), "iters_per_s": round(n_p / pdt, 3),
                                "instances_emitted": wl.R_emitted, "stages_ms": {k: (round(v, 4) if v is not None else None)
                                                                                 for k, v in p_stage.items()},
                                "switches": int(mask)}
            finally:
                native_c.test_switches(base_switches | MODE_MASK[args.mode])
        for _ in range(3):
            wl.step()
    # ---- the step without a host wait (sgr_set_lazy, opt-in: include/sgr.h) and the same step replayed from a hipGraph:
    # extra regions, never the reported value
    lazy_info = None
    if world == 1 and dist is None and S == 0:
        try:
            # host time spent waiting for the device inside sgr_forward (the read-back of num_rendered), blocking vs lazy
            fence()
            L.sgr_profile_host_wait_us(1)
            for _ in range(50):
                wl.step()
            fence()
            wait_blocking = L.sgr_profile_host_wait_us(1) / 50.0
            was = native_c.set_lazy(True)
            try:
                for _ in range(3):
                    wl.step()
                n_l = max(args.steps, int(0.4 / max(dt / args.steps, 1e-5)) + 1)
                fence()
                L.sgr_profile_host_wait_us(1)
                ldt, _ = profiled_steps(L, wl, fence, n_l, 0)
                torch.cuda.synchronize()
                wait_lazy = L.sgr_profile_host_wait_us(1) / float(n_l)
                R_l, cap_l, fl_l = native_c.lazy_status()
                lazy_info = {"ms_per_step_lazy": round(1e3 * ldt / n_l, 4), "steps": n_l, "num_rendered": R_l, "capacity": cap_l,
                             "flags": fl_l,
                             "host_wait_us_per_step": {"blocking": round(wait_blocking, 1), "lazy": round(wait_lazy, 1),
                                                       "what": "time the host spends inside sgr_forward waiting for the device "
                                                               "(sgr_profile_host_wait_us): the GPU runs behind the host, so in the "
                                                               "blocking mode this is mostly the host being AHEAD, not the GPU idling"}}
                p, w = wl.params, wl.w
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    wl.step()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                for t in list(p.values()) + [wl.means2D]:
                    t.grad = None
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    color, radii, depth, alpha, _ = wl.rast(p["means3D"], wl.means2D, p["opacities"], shs=p["shs"],
                                                            scales=p["scales"], rotations=p["rotations"])
                    torch.autograd.backward([color, depth, alpha], [w["color"], w["depth"], w["alpha"]])
                for _ in range(5):
                    g.replay()
                fence()
                t0 = time.perf_counter()
                for _ in range(n_l):
                    g.replay()
                fence()
                gdt = time.perf_counter() - t0
                lazy_info["ms_per_step_graph"] = round(1e3 * gdt / n_l, 4)
                lazy_info["graph_status"] = dict(zip(("num_rendered", "capacity", "flags"), native_c.lazy_status()))
                del g
            finally:
                native_c.set_lazy(was)
            for _ in range(3):
                wl.step()
        except Exception as ex:  # an extra: never lose the headline over it
            lazy_info = dict(lazy_info or {}, error=f"{type(ex).__name__}: {ex}"[:300])
    R, V, pairs_blended = wl.counts()
    wl_R_emitted = wl.R_emitted  # default mode (the mode regions above ran wl.counts() under their own switches)
    N = args.width * args.height
    # N > 1 (or a forced one-rank group) with --densify-loop: configs[4] as BASELINE.json words it, every rank taking part
    densify_multi = None
    if args.densify_loop and dist is not None:
        torch.cuda.empty_cache()
        loop = DensifyLoop(args, args.densify_gaussians, dev, args.densify_every, dist=dist, rank=rank, force_dist=force_dist)
        densify_multi = loop.run(fence)
        t = torch.tensor([densify_multi["ms_per_step_amortised"]], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        densify_multi["ms_per_step_amortised_max_over_ranks"] = round(float(t.item()), 4)
        densify_multi["views_per_s_whole_job"] = round(world * 1e3 / float(t.item()), 3)
        del loop

    if rank == 0:
        bwd_ms = timed["blend_bwd"]
        algo_bytes = blend_bytes(S, R, N, V)[0]
        T_tiles = ((args.width + 15) // 16) * ((args.height + 15) // 16)
        sb = stage_bytes(args.gaussians, wl.V_in, V, R, N, T_tiles, S)
        sb_emitted = stage_bytes(args.gaussians, wl.V_in, V, wl_R_emitted, N, T_tiles, S)
        fwd_ms = stage_ms["blend_fwd"]
        traffic, valu, traffic_note, pmc_derived = None, None, None, None
        tfile = os.path.join(ROOT, "profiles", "pmc_blend_bwd.json")
        exact_arith = args.mode in ("strict", "exact")
        kernel_name = ("sgr_blend_bwd_kernel_exact" if exact_arith else
                       ("sgr_blend_bwd_kernel_s0" if S == 0 else "sgr_blend_bwd_kernel"))
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                sha = sgr_build.source_sha16()
                same_cfg = (tj.get("gaussians") == args.gaussians and tj.get("width") == args.width and
                            tj.get("height") == args.height and kernel_name in str(tj.get("kernel", "")) and
                            tj.get("mode", "fast") == args.mode and tj.get("semantics", 0) == S and not args.scene)
                if not same_cfg:
                    traffic_note = "profiles/pmc_blend_bwd.json was measured on another kernel / configuration"
                elif tj.get("source_sha16") != sha:
                    traffic_note = (f"profiles/pmc_blend_bwd.json was measured on kernel sources {tj.get('source_sha16')}, "
                                    f"this build is {sha}: re-run tools/gpu_evidence.sh + tools/pmc_summary.py")
                else:
                    traffic = tj.get("hbm_bytes_per_launch")
                    pmc_derived = {k: v for k, v in (tj.get("derived") or {}).items() if "nominal" not in k}
                    traffic_note = "rocprofv3 PMC, 2*FETCH_SIZE + WRITE_SIZE per launch, " + str(tj.get("source", ""))[:120]
                    if tj.get("sq_insts_valu") and bwd_ms:
                        # what actually bounds the kernel: VALU wave-instructions (rocprofv3 SQ_INSTS_VALU) against
                        # the issue slots of 256 CUs x 4 SIMDs (one wave64 VALU instruction = 2 cycles) at 2.4 GHz
                        slots = 1024 * (bwd_ms * 1e-3) * 2.4e9 / 2.0
                        valu = {"insts_per_launch": int(tj["sq_insts_valu"]), "issue_slot_frac": round(tj["sq_insts_valu"] / slots, 3),
                                "source": "profiles/pmc_blend_bwd.json (SQ_INSTS_VALU) / live kernel_ms"}
            except Exception as ex:
                traffic, traffic_note = None, f"profiles/pmc_blend_bwd.json unreadable: {ex}"
        # `value`: whole-job throughput of the timed region -- EXACTLY --steps steps between two fences, as the command line
        # says.  (Round 4 printed the >= 1 s sustained region as `value` next to `steps: K`; the two agree to < 0.1 % once
        # the device is at its clocks, and a line whose `steps` and `ms_per_step` describe different regions reads wrong.)
        value = round(world * args.steps / dt, 3)
        ms_per_step = round(1e3 * dt / args.steps, 4)
        # the bound the dominant kernel actually runs against: VALU issue.  Per-visit pipe cycles from the static model of the
        # kernel's ISA priced with instruction costs MEASURED IN CYCLES (tools/valu_model.py over tools/ubench/valu_rates2.hip:
        # s_memtime in the kernel, clock reported), times the visits of this frame, over the 1024 SIMDs at the ubench's clock
        valu_issue = None
        try:
            vm_path = os.path.join("profiles", "r6", "valu_model.json")
            vm = json.load(open(os.path.join(ROOT, vm_path)))
            vm_k = vm["parity_mode" if exact_arith else "default"]
            if S == 0 and bwd_ms and vm_k.get("source_sha16") == sgr_build.source_sha16():
                visits = count_visits(wl, R)
                cyc, ghz = vm_k["valu_pipe_cycles_per_visit"], vm_k["clock_ghz"]
                bound_ms = visits * cyc / (1024 * ghz * 1e9) * 1e3
                valu_issue = {"bound": "valu_issue", "visits": visits, "valu_pipe_cycles_per_visit": cyc, "clock_ghz": ghz,
                              "bound_ms": round(bound_ms, 4), "kernel_ms": round(bwd_ms, 4), "frac": round(bound_ms / bwd_ms, 3),
                              "achieved_gcycles_per_s": round(visits * cyc / (bwd_ms * 1e-3) / 1e9, 1),
                              "peak_gcycles_per_s": round(1024 * ghz, 1),
                              "source": vm_path + " (static ISA model x instruction costs in cycles, "
                                        "profiles/r5/valu_rates2.jsonl); time bound = visits * cycles / (1024 SIMDs * measured clock)"}
            elif S == 0:
                valu_issue = {"note": vm_path + " was made for other kernel sources: re-run tools/valu_model.py"}
        except Exception as ex:
            valu_issue = {"note": f"unavailable: {ex}"[:160]}

        def kernel_roofline(kms, instances):
            """HBM fraction of the blend backward for a launch of `kms` ms: SURVEY 8d's formula on the instances the launch
            actually processed (the basis of `frac`), and on the reference's R (the work the reference's kernel does)."""
            if not kms:
                return None
            be, br = blend_bytes(S, instances, N, V)[0], algo_bytes
            return {"kernel_ms": round(kms, 4), "instances": instances, "algorithmic_bytes_per_launch": be,
                    "achieved": round(be / (kms * 1e-3) / 1e9, 2), "frac": round(be / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "frac_on_reference_R": round(br / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}

        rl_head = kernel_roofline(bwd_ms, wl_R_emitted)  # the timed region's mode
        rl_modes = {k: kernel_roofline(m["stages_ms"].get("blend_bwd"), m["instances_emitted"]) for k, m in modes.items()}
        rl_modes[args.mode] = rl_head
        g = lambda d, k: (d or {}).get(k)
        mode_val = lambda k, f: (value if f == "iters_per_s" else ms_per_step) if k == args.mode else g(modes.get(k), f)
        line = {
            "metric": "train iters/s (fwd+bwd) @1M Gaussians 1920x1280 SH3",
            "value": value,
            "mode": args.mode,
            "conforming": args.mode == "strict",
            "mode_note": {"strict": "`value` is the STRICTLY CONFORMING configuration: parity arithmetic on the reference's own tile "
                                    "rects (SGR_EXACT=1 SGR_REF_RECT=1) -- binning arrays the reference's entry for entry, alpha / depth "
                                    "/ semantic images bit-identical, colour and all nine gradient tensors within rel 1e-4 end to end "
                                    "(gated at all four BASELINE sizes); value_fast = the library's default mode on the same workload",
                          "exact": "`value` is the parity arithmetic on the cut-down tile lists (SGR_EXACT=1)",
                          "fast": "`value` is the library's DEFAULT mode (fast arithmetic, cut-down tile lists), NOT the "
                                  "conforming one: see value_strict"}[args.mode],
            "value_strict": mode_val("strict", "iters_per_s"),
            "value_exact": mode_val("exact", "iters_per_s"),
            "value_fast": mode_val("fast", "iters_per_s"),
            "unit": "iters/s",
            "n_gpus": world,
            "rccl_ranks": (dist.get_world_size() if (dist is not None and backend == "nccl") else 0),
            "dist_ranks": (dist.get_world_size() if dist is not None else 0),
            "dist_backend": (backend if dist is not None else None),
            "ranks_share_one_gpu": share_gpu if dist is not None else None,
            "summed_gradients_identical_on_all_ranks": grads_identical,
            "exchange_direct_bucket_writes": (None if reducer is None else {
                "dense_tensors_copied_into_the_bucket_last_step": getattr(getattr(reducer, "dense", reducer), "copied_last", None),
                "what": "0 = the rasterizer backward wrote every dense gradient straight into the all-reduce bucket "
                        "(multiview.GradReducer._sink); the masked dRGB goes straight into the all-gather payload and this "
                        "view's own dL/dSH is not written (sgr_backward_extras.masked_color_out / skip_sh_grad)"}),
            "steps": args.steps,
            "warmup": args.warmup,
            "device_warmup_s": args.device_warmup,
            "ms_per_step": ms_per_step,
            "ms_per_step_strict": mode_val("strict", "ms_per_step"),
            "ms_per_step_exact": mode_val("exact", "ms_per_step"),
            "ms_per_step_fast": mode_val("fast", "ms_per_step"),
            "blend_bwd_kernel_ms": round(bwd_ms, 4) if bwd_ms else None,
            "blend_bwd_kernel_ms_strict": g(rl_modes.get("strict"), "kernel_ms"),
            "blend_bwd_kernel_ms_exact": g(rl_modes.get("exact"), "kernel_ms"),
            "blend_bwd_kernel_ms_fast": g(rl_modes.get("fast"), "kernel_ms"),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not args.scene else f"scene file {os.path.basename(args.scene)}",
            "config": {"workload": f"{args.gaussians} " + ("synthetic Gaussians (SURVEY 8d recipe, seed 0)" if not args.scene
                                   else f"Gaussians of {os.path.basename(args.scene)}") +
                                   f", {args.width}x{args.height}, SH degree 3, S={S} semantic channels, "
                                   f"rasterizer forward+backward, one camera view per GPU, mode {args.mode}",
                       "gaussians": args.gaussians, "width": args.width, "height": args.height, "sh_degree": 3,
                       "semantic_channels": S, "loss": args.loss, "views_per_step": world, "num_rendered_R": R, "visible_V": V,
                       "R_over_P": round(R / args.gaussians, 3), "V_over_P": round(V / args.gaussians, 3),
                       "instances_emitted": wl_R_emitted,
                       "instances_note": "num_rendered_R = the reference's num_rendered (3-sigma squares, auxiliary.h getRect), the "
                                         "unit of SURVEY 8d's byte formulas; instances_emitted = what this mode duplicates, sorts and "
                                         "blends (strict: the same number; exact / fast: rects cut down to the tiles where the Gaussian "
                                         "can reach alpha >= 1/255, with a tile mask inside them -- bit-identical images)",
                       "parallelism": f"view-dp{world}" + ((" + RCCL all-reduce of Gaussian grads" + (
                           " (dense 44 B/Gaussian; SH gradient rebuilt from an all-gather of per-view dRGB, 12 B/Gaussian/view)"
                           if args.reduce == "factored" else " (one 236 B/Gaussian bucket)")) if world > 1 else ""),
                       "exchange_bytes_per_rank": reducer.nbytes if reducer is not None else 0,
                       "exchange_schedule": (None if reducer is None else (
                           "side stream, overlapped with the next step's forward (one-step-delayed gradients in training)"
                           if args.exchange == "overlap" else "blocking, inside the step")),
                       "kernel_sources_sha16": sgr_build.source_sha16()},
            # The dominant kernel (blend backward) is bound by VALU ISSUE, not by HBM: `bound` says so and the
            # valu_* scalars are its roofline; achieved / peak / unit / frac / traffic are the HBM figures the metric asks for
            # (algorithmic bytes of the instances the launch processes / kernel time, against 8 TB/s).
            "roofline": {"bound": "valu_issue", "kernel": kernel_name, "mode": args.mode,
                         "achieved": g(rl_head, "achieved"), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": g(rl_head, "frac"), "frac_on_reference_R": g(rl_head, "frac_on_reference_R"),
                         "traffic": traffic,
                         "valu_issue_frac": g(valu_issue, "frac"),
                         "valu_achieved_gcycles_per_s": g(valu_issue, "achieved_gcycles_per_s"),
                         "valu_peak_gcycles_per_s": g(valu_issue, "peak_gcycles_per_s"),
                         "kernel_ms": g(rl_head, "kernel_ms"),
                         "kernel_ms_strict": g(rl_modes.get("strict"), "kernel_ms"), "frac_strict": g(rl_modes.get("strict"), "frac"),
                         "kernel_ms_exact": g(rl_modes.get("exact"), "kernel_ms"), "frac_exact": g(rl_modes.get("exact"), "frac"),
                         "kernel_ms_fast": g(rl_modes.get("fast"), "kernel_ms"), "frac_fast": g(rl_modes.get("fast"), "frac"),
                         "algorithmic_bytes_per_launch": g(rl_head, "algorithmic_bytes_per_launch"),
                         "algorithmic_bytes_note": "SURVEY 8d: (44+4S) R + (28+4S) N + (48+4S) V with R = the instances the launch "
                                                   "processes (config.instances_emitted); frac_on_reference_R: the same with the "
                                                   "reference's num_rendered",
                         "bound_note": "`frac` (and achieved / peak / unit / traffic) is the HBM figure BASELINE.json's metric asks "
                                       "for; the kernel's own bound is VALU issue: valu_issue_frac = (visits x pipe cycles per visit "
                                       "of its instruction stream at measured instruction costs) / (1024 SIMDs x measured clock x "
                                       "kernel time).  north_star's '>= 60 % of the HBM roofline in the blend kernel' is not reachable "
                                       "by this algorithm (0.45 GB of compulsory bytes against ~1.6e8 (pixel, Gaussian) pairs of ~80 "
                                       "flops each; SURVEY 8d)",
                         "traffic_note": traffic_note, "valu": valu, "valu_issue": valu_issue, "pmc": pmc_derived,
                         "modes": rl_modes,
                         "kernel_ms_source": f"HIP events around the kernel on its launch stream, mean over "
                                             f"{kernel_samples} launches of {kernel_region} "
                                             f"(every 8th step at most carries the event pair)",
                         "blend_fwd": {"kernel_ms": round(fwd_ms, 4) if fwd_ms else None,
                                       "achieved": round(blend_bytes(S, wl_R_emitted, N, V)[1] / (fwd_ms * 1e-3) / 1e9, 2) if fwd_ms else None,
                                       "algorithmic_bytes_per_launch": blend_bytes(S, wl_R_emitted, N, V)[1]},
                         "stages_ms": {k: (round(v, 4) if v is not None else None) for k, v in stage_ms.items()},
                         "stages_ms_source": "extra pass after the timed region, every stage bracketed by HIP events",
                         "stages_bytes": sb_emitted, "stages_hbm_frac": stage_hbm_frac(stage_ms, sb_emitted),
                         "stages_hbm_frac_on_reference_R": stage_hbm_frac(stage_ms, sb),
                         "stages_bytes_source": "SURVEY 8d algorithmic bytes (B_pre, B_scan, B_dup, B_sort = 24 R, B_rng, "
                                                "B_blend_f, B_blend_b, B_pre_b) with R = the emitted instances; frac = bytes / "
                                                "stage time / 8 TB/s",
                         "sum_n_contrib_pairs": pairs_blended},
        }
        if sustained is not None:
            line["sustained"] = sustained
        if modes:
            line["modes"] = dict(modes, what="strict = sgr_test_switches bits 7 + 10 (SGR_EXACT=1 SGR_REF_RECT=1): north_star's parity "
                                 "sentence as worded; exact = bit 7 alone (cut-down tile lists); fast = the library's default "
                                 "arithmetic on the cut-down lists (DESIGN.md section 4)")
        if lazy_info is not None:
            line["lazy"] = dict(lazy_info, what="sgr_set_lazy(1): list capacity from the previous frames, no wait for the frame's own num_rendered "
                                "(ms_per_step_lazy: eager launches; ms_per_step_graph: forward + backward captured once in a "
                                "hipGraph and replayed); bit-identical outputs (tests/test_gpu_graph.py); opt-in, not `value`")
        if exchange_overlap is not None:
            line["exchange_overlap"] = exchange_overlap
        if world == 1 and dist is None and not args.no_other_configs and not args.scene:
            line["other_configs"] = other_configs(args, L, dev, fence)
        if densify_multi is not None:
            line["configs4_densify_loop"] = densify_multi
        if not args.no_cpu_baseline and world == 1:
            try:
                rk = reference_kernels_on_gpu(args, wl.cam, wl.scene)
                if rk is not None:
                    rk["speedup_vs_reference_kernels"] = round(line["value"] / rk["iters_per_s"], 2)
                    line["reference_kernels_mi355x"] = rk
            except Exception as ex:
                line["reference_kernels_mi355x"] = {"error": str(ex)[:200]}
            try:
                line["cpu_baseline"] = cpu_baseline(args, wl.cam0, wl.scene)
            except Exception as ex:  # the baseline is a reported extra; never lose the GPU number over it
                line["cpu_baseline"] = {"value": None, "unit": "iters/s", "cores": os.cpu_count(), "kind": "port",
                                        "sample": f"failed: {ex}"}
            try:
                line["cpu_baseline"]["configs0_smoke_recipe"] = cpu_smoke_recipe()
            except Exception as ex:
                line["cpu_baseline"]["configs0_smoke_recipe"] = {"error": str(ex)[:160]}
        # LAST key: the figures a reader of the tail of this line needs, in one short object
        line["summary"] = {
            "value": value, "mode": args.mode, "conforming": args.mode == "strict", "ms_per_step": ms_per_step, "steps": args.steps,
            "value_strict": line["value_strict"], "ms_per_step_strict": line["ms_per_step_strict"],
            "value_exact": line["value_exact"], "ms_per_step_exact": line["ms_per_step_exact"],
            "value_fast": line["value_fast"], "ms_per_step_fast": line["ms_per_step_fast"],
            "sustained_ms_per_step": g(sustained, "ms_per_step"),
            "ms_per_step_lazy": g(lazy_info, "ms_per_step_lazy"), "ms_per_step_graph": g(lazy_info, "ms_per_step_graph"),
            "blend_bwd_ms": {k: g(rl_modes.get(k), "kernel_ms") for k in ("strict", "exact", "fast")},
            "blend_bwd_hbm_frac_on_processed_instances": {k: g(rl_modes.get(k), "frac") for k in ("strict", "exact", "fast")},
            "blend_bwd_hbm_frac_on_reference_R": g(rl_head, "frac_on_reference_R"),
            "valu_issue_frac": g(valu_issue, "frac"),
            "stages_ms": line["roofline"]["stages_ms"],
            "instances": {"reference_R": R, "emitted": wl_R_emitted},
            "cpu_baseline_iters_per_s": g(line.get("cpu_baseline"), "value"),
            "reference_kernels_on_this_gpu_iters_per_s": g(line.get("reference_kernels_mi355x"), "iters_per_s")}
        try:  # RCCL prints a version banner through C stdio; flush it so the JSON line stays the last line
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def tile_loads(wl):
    """List length per tile of this view (untimed; the exported ranges of a fresh forward): how unequal the tiles are."""
    from street_gaussians_amd import _C as native
    st, p = wl.st, wl.params
    sem = p["semantics"].detach() if "semantics" in p else torch.zeros(wl.P, 0, device=wl.dev)
    out = native.rasterize_gaussians(st.bg, p["means3D"].detach(), torch.Tensor([]), sem, p["opacities"].detach(),
                                     p["scales"].detach(), p["rotations"].detach(), 1.0, torch.Tensor([]), st.viewmatrix,
                                     st.projmatrix, st.tanfovx, st.tanfovy, st.image_height, st.image_width,
                                     p["shs"].detach(), 3, st.campos, False, False)
    rg = native.export_internal("ranges", wl.P, int(out[0]), wl.args.height, wl.args.width, out[6], out[7], out[8]).view(-1, 2)
    n = (rg[:, 1] - rg[:, 0]).to(torch.float64)
    srt = torch.sort(n).values
    q = lambda f: int(srt[min(srt.numel() - 1, int(f * srt.numel()))].item())
    return {"tiles": int(n.numel()), "empty_tiles": int((n == 0).sum().item()), "min": int(srt[0].item()), "median": q(0.5),
            "mean": round(float(n.mean().item()), 1), "p99": q(0.99), "max": int(srt[-1].item()),
            "max_over_mean": round(float(srt[-1].item() / max(n.mean().item(), 1e-9)), 2)}


def street_config(args, L, dev, fence, P=2_000_000, S=19):
    """BASELINE.json configs[2] as the reference builds such a frame (lib/models/street_gaussian_model.py:219-449): a static
    background model + 12 posed actors with Fourier DC features, composed by street_gaussians_amd.scene into the
    rasterizer's inputs -- road, facades, clutter, EMPTY SKY in the upper middle of the image, the actors' tiles carrying
    lists several times the mean (synthetic.make_street_segments).  Reports how unequal the tiles are, the blend kernels'
    cost per (pixel, Gaussian) pair next to the uniform scene's (other_configs: configs[2] 2M + 19 channels), and the same
    step with the blend launches in LONGEST-FIRST tile order (sgr_test_switches bit 14) -- the experiment the uniform scene
    could not decide (it decided: the forward now picks the order per frame)."""
    from street_gaussians_amd import scene as sg
    from street_gaussians_amd import _C as native_c
    W, H = args.width, args.height
    cam0 = syn.make_camera(W, H, fx=2050.0 * W / 1920.0)
    raw = syn.make_street_segments(P, cam0, S=S, seed=0)
    segs = [sg.Segment(**{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items() if v is not None}) for d in raw]
    with torch.no_grad():
        means, rot, scl, opa, shs, sem = sg.compose(segs, 16, S)
    scene = syn.Scene(means.contiguous(), scl.contiguous(), rot.contiguous(), opa.contiguous(), shs.contiguous(),
                      (sem if S else torch.zeros(means.shape[0], 0, device=dev)).contiguous())
    del segs
    wl = Workload(args, scene.P, S, 0, dev, scene=scene)

    def measure():
        t_w, n_w = time.perf_counter(), 0
        while n_w < 3 or time.perf_counter() - t_w < 0.25:
            wl.step()
            n_w += 1
            if n_w % 16 == 0:
                torch.cuda.synchronize()
        steps = 20
        dt, _ = profiled_steps(L, wl, fence, steps, 0)
        _, st = profiled_steps(L, wl, fence, 10, 0x1FF)
        return round(1e3 * dt / steps, 4), {k: (round(v, 4) if v is not None else None) for k, v in st.items()}

    ms, st = measure()
    R, V, pairs = wl.counts()
    loads = tile_loads(wl)
    cur = native_c.test_switches(-1)
    native_c.test_switches(cur | native_c.NO_LPT)
    try:
        ms_st, st_st = measure()
    finally:
        native_c.test_switches(cur)
    fast = None
    if cur & (native_c.EXACT | native_c.REF_RECT):
        base = cur & ~(native_c.EXACT | native_c.REF_RECT)
        try:
            native_c.test_switches(base)
            f_ms, f_st = measure()
            native_c.test_switches(base | native_c.NO_LPT)
            fl_ms, fl_st = measure()
            fast = {"ms_per_step": f_ms, "stages_ms": f_st, "supertile_order": {"ms_per_step": fl_ms, "stages_ms": fl_st}}
        finally:
            native_c.test_switches(cur)
    return {"config": "configs[2] street-like composed scene: background + 12 posed actors (scene.compose), empty sky, "
                      f"{scene.P} Gaussians + {S} semantic channels", "gaussians": scene.P, "semantic_channels": S, "mode": args.mode,
            "actors": len(raw) - 1, "ms_per_step": ms, "stages_ms": st, "num_rendered_R": R, "instances_emitted": wl.R_emitted,
            "visible_V": V, "sum_n_contrib_pairs": pairs, "tile_list_length": loads,
            "blend_ns_per_1000_pairs": {k: round(1e6 * st[k] / max(pairs, 1) * 1e3, 3) for k in ("blend_fwd", "blend_bwd") if st.get(k)},
            "tile_order": {"default_ms_per_step": ms, "supertile_order_ms_per_step": ms_st, "supertile_order_stages_ms": st_st,
                           "blend_fwd_plus_bwd_ms": {"default": round((st["blend_fwd"] or 0) + (st["blend_bwd"] or 0), 4),
                                                     "supertile_order": round((st_st["blend_fwd"] or 0) + (st_st["blend_bwd"] or 0), 4)},
                           "what": "default: the forward decides per frame (one one-workgroup launch, timed inside blend_fwd) -- this "
                                   "scene's longest list is > 2.5 x the mean, so both blend launches walk the tiles LONGEST LIST "
                                   "FIRST; supertile_order = sgr_test_switches bit 15 (SGR_NO_LPT), the XCD-aware order the uniform "
                                   "scenes keep"},
            "fast": fast}


def other_configs(args, L, dev, fence):
    """BASELINE.json's other single-GPU configurations on the same build (untimed extras of the N = 1 run): configs[1]
    500 k Gaussians, configs[2] 2 M Gaussians + 19 semantic channels, configs[4]'s per-GPU load 5 M Gaussians -- first the
    rasterizer alone, then the loop as configs[4] words it, with densify / prune active between iterations (DensifyLoop)."""
    out = []
    for name, P, S in (("configs[1] 500k", 500_000, 0), ("configs[2] 2M + 19 semantic channels", 2_000_000, 19),
                       ("configs[4] 5M (rasterizer only)", 5_000_000, 0)):
        try:
            torch.cuda.empty_cache()
            wl = Workload(args, P, S, 0, dev)
            # warm-up by TIME: the scene was generated on the host for seconds, the idle GPU has clocked down, and 3 short
            # steps do not bring it back (observed: 4.3 instead of 1.3 ms/step at 500 k right after the idle period)
            t_w = time.perf_counter()
            n_w = 0
            while n_w < 3 or time.perf_counter() - t_w < 0.25:
                wl.step()
                n_w += 1
                if n_w % 16 == 0:
                    torch.cuda.synchronize()
            steps = 20
            dt, _ = profiled_steps(L, wl, fence, steps, 0)
            _, st = profiled_steps(L, wl, fence, 10, 0x1FF)
            R, V, pairs = wl.counts()
            # the same in the library's default (fast) mode, when the run's mode is another one
            fast = None
            from street_gaussians_amd import _C as native_c
            cur = native_c.test_switches(-1)
            if cur & (native_c.EXACT | native_c.REF_RECT):
                native_c.test_switches(cur & ~(native_c.EXACT | native_c.REF_RECT))
                try:
                    for _ in range(5):
                        wl.step()
                    fdt, _ = profiled_steps(L, wl, fence, steps, 0)
                    _, fst = profiled_steps(L, wl, fence, 10, 0x1FF)
                    fast = {"ms_per_step": round(1e3 * fdt / steps, 4),
                            "stages_ms": {k: (round(v, 4) if v is not None else None) for k, v in fst.items()}}
                finally:
                    native_c.test_switches(cur)
            bb, fb = blend_bytes(S, R, args.width * args.height, V)
            sb = stage_bytes(P, wl.V_in, V, R, args.width * args.height, ((args.width + 15) // 16) * ((args.height + 15) // 16), S)
            Npx, Tt = args.width * args.height, ((args.width + 15) // 16) * ((args.height + 15) // 16)
            sbe = stage_bytes(P, wl.V_in, V, wl.R_emitted, Npx, Tt, S)  # on the instances the kernels process
            bbe = blend_bytes(S, wl.R_emitted, Npx, V)[0]
            out.append({"config": name, "gaussians": P, "semantic_channels": S, "steps": steps, "mode": args.mode,
                        "ms_per_step": round(1e3 * dt / steps, 4), "iters_per_s": round(steps / dt, 3),
                        "ms_per_step_fast": (fast or {}).get("ms_per_step"), "fast": fast,
                        "num_rendered_R": R, "instances_emitted": wl.R_emitted, "visible_V": V, "sum_n_contrib_pairs": pairs,
                        "blend_ns_per_1000_pairs": {k: round(1e6 * st[k] / max(pairs, 1) * 1e3, 3) for k in ("blend_fwd", "blend_bwd") if st.get(k)},
                        "stages_hbm_frac": stage_hbm_frac(st, sbe), "stages_hbm_frac_on_reference_R": stage_hbm_frac(st, sb),
                        "blend_bwd_ms": round(st["blend_bwd"], 4) if st["blend_bwd"] else None,
                        "blend_fwd_ms": round(st["blend_fwd"], 4) if st["blend_fwd"] else None,
                        "blend_bwd_hbm_frac": round(bbe / (st["blend_bwd"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if st["blend_bwd"] else None,
                        "blend_bwd_hbm_frac_on_reference_R": round(bb / (st["blend_bwd"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if st["blend_bwd"] else None,
                        "stages_ms": {k: (round(v, 4) if v is not None else None) for k, v in st.items()}})
            del wl
        except Exception as ex:  # an extra: never lose the headline over it
            out.append({"config": name, "error": f"{type(ex).__name__}: {ex}"[:200]})
    try:  # configs[2] as a street scene: composed background + actors, empty sky, clustered tiles
        torch.cuda.empty_cache()
        out.append(street_config(args, L, dev, fence))
    except Exception as ex:
        out.append({"config": "configs[2] street-like composed scene", "error": f"{type(ex).__name__}: {ex}"[:300]})
    try:  # configs[4] as written: the densify / prune step active between iterations
        torch.cuda.empty_cache()
        # twice (fresh state each time), both kept: the region is ~0.15 s long with a host synchronisation at every
        # densify step, so a box whose host cores are busy elsewhere shows up here first (observed once: 55 ms iterations
        # with normal kernels).  The FIRST run is the entry (no selection), the second rides along
        runs = []
        for _ in range(2):
            torch.cuda.empty_cache()
            loop = DensifyLoop(args, 5_000_000, dev, args.densify_every)
            runs.append(loop.run(fence))
            del loop
        runs[0]["other_run"] = {k: runs[1][k] for k in ("ms_per_step_amortised", "raster_ms_steady_median", "densify_ms_mean",
                                                        "host_ms_to_queue_one_iteration", "device_allocations_in_region")}
        out.append(runs[0])
    except Exception as ex:
        out.append({"config": "configs[4] 5M + densify/prune active in the loop", "error": f"{type(ex).__name__}: {ex}"[:300]})
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
