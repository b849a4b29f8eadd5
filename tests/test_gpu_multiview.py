"""`-m gpu` tests of the view-sharded exchange step (street_gaussians_amd/multiview.py + csrc/sgr_multiview.hip):
the SH gradient rebuilt from per-view dRGB must equal the sum of the per-view dL/dSH tensors it replaces."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from gpu_utils import dev, npy, settings
from helpers import oracle_kwargs
from oracle import oracle
from street_gaussians_amd import multiview, rasterizer
from street_gaussians_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _views(V):
    return [syn.make_camera(480, 320, fx=512.5, yaw_deg=4.0 * v) for v in range(V)]


@pytest.mark.parametrize("deg", [3, 1])
def test_factored_sh_gradient_equals_sum_over_views(deg):
    from diff_gaussian_rasterization import GaussianRasterizer
    V = 3
    cams = _views(V)
    sc = syn.make_scene(20000, cams[0], S=0, seed=11)
    sc.shs[::3, 0, :] -= 2.0  # plenty of clamped channels
    t = {k: dev(getattr(sc, k)).requires_grad_(True) for k in ["means3D", "scales", "rotations", "opacities", "shs"]}

    def run_views():
        for v, cam in enumerate(cams):
            st = settings(cam)._replace(sh_degree=deg)
            color, radii, depth, alpha, sem = GaussianRasterizer(st)(t["means3D"], None, t["opacities"], shs=t["shs"],
                                                                     scales=t["scales"], rotations=t["rotations"])
            w = syn.loss_weights(cam, seed=3 + v)
            torch.autograd.backward([color, depth, alpha], [dev(w["color"]), dev(w["depth"]), dev(w["alpha"])])

    run_views()  # plain autograd accumulation: shs.grad = sum of the per-view dL/dSH
    want = {k: p.grad.clone() for k, p in t.items()}
    for p in t.values():
        p.grad = None
    port = socket.socket()
    port.bind(("127.0.0.1", 0))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port.getsockname()[1])
    port.close()
    dist.init_process_group("nccl", rank=0, world_size=1)  # one-rank RCCL group: the collectives really run
    try:
        dense = [t[k] for k in ["means3D", "scales", "rotations", "opacities"]]
        with multiview.FactoredGradReducer(dense, t["shs"], t["means3D"], views_per_rank=V, force=True) as red:
            run_views()
            red.all_reduce()
            assert red.nbytes == 4 * (sum(p.numel() for p in dense) + V * (3 + 3 * sc.P))
        assert len(rasterizer.BACKWARD_OBSERVERS) == 0
    finally:
        dist.destroy_process_group()
    got, ref = npy(t["shs"].grad), npy(want["shs"])
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 2e-6 * scale, np.abs(got - ref).max() / scale
    assert (got[:, (deg + 1) ** 2:, :] == 0).all()
    for k in ["means3D", "scales", "rotations", "opacities"]:
        assert torch.equal(t[k].grad, want[k]), k


def test_masked_color_grad_and_rebuild_match_oracle():
    """One view, through the raw entry points: dRGB = dL/dcolour with the oracle's clamp mask, and the rebuilt dL/dSH
    equals the oracle's (backward.cu:46-105)."""
    from gpu_utils import raw_backward, raw_forward
    from street_gaussians_amd import _C
    cam = _views(1)[0]
    sc = syn.make_scene(5000, cam, S=0, seed=12)
    sc.shs[::2, 0, :] -= 2.0
    kw = oracle_kwargs(cam, sc, deg=2)
    wts = syn.loss_weights(cam)
    fw = oracle.forward(**kw)
    ref = oracle.backward(fw, wts["color"], wts["depth"], wts["alpha"], None)
    res, _ = raw_forward(kw)
    g = raw_backward(kw, res, wts)
    drgb = _C.masked_color_grad(res["geom"], g["colors"], sc.P)
    mask = fw.clamped.reshape(sc.P, 3) != 0
    assert (npy(drgb)[mask] == 0).all() and (npy(drgb)[~mask] == npy(g["colors"])[~mask]).all()
    dsh = _C.sh_grad_from_views(dev(sc.means3D), dev(cam.campos).reshape(1, 3), drgb.reshape(1, sc.P, 3), 2, 16)
    want = ref["sh"].reshape(sc.P, 16, 3)
    assert np.abs(npy(dsh) - want).max() <= 1e-5 * np.abs(want).max()
    assert torch.allclose(dsh, g["sh"].reshape(sc.P, 16, 3), rtol=0, atol=2e-6 * float(g["sh"].abs().max()))
    fw.free()


def test_bench_runs_the_rccl_exchange_on_one_rank():
    """bench.py with a forced one-rank RCCL group (SGR_BENCH_FORCE_DIST=1): the factored exchange, overlapped and
    blocking, runs end to end on the GPU and prints the contract's JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("overlap", "blocking"):
        env = dict(os.environ, SGR_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                              "--gaussians", "100000", "--no-cpu-baseline", "--no-other-configs", "--exchange", mode],
                             env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads(out.stdout.strip().split("\n")[-1])
        assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["exchange_bytes_per_rank"] > 0
        assert (mode == "overlap") == ("overlapped" in line["config"]["exchange_schedule"])
        # the reported value is the blocking schedule; the overlapped one rides along as an extra when blocking is reported
        assert (mode == "blocking") == ("exchange_overlap" in line)


def test_bench_runs_the_replicated_densify_loop_on_a_forced_group():
    """bench.py --densify-loop with a forced one-rank RCCL group: BASELINE configs[4]'s multi-rank program (gradient exchange
    every step, statistics reduce, replicated densify with ReplicatedNormals, reducer rebuild for the new P) end to end on
    the GPU at a small size; the world-2 logic itself is tests/test_multiview_gloo.py."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SGR_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                          "--gaussians", "100000", "--no-cpu-baseline", "--no-other-configs", "--densify-loop",
                          "--densify-gaussians", "200000", "--densify-every", "4"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().split("\n")[-1])
    d = line["configs4_densify_loop"]
    assert d["ranks"] == 1 and d["replicas_identical"] is True and d["densify_steps"] == 3
    assert d["gaussians_end"] != d["gaussians_start"] and d["ms_per_step_amortised"] > 0
    assert line["rccl_ranks"] == 1


def test_bench_two_ranks_share_one_gpu_over_gloo():
    """`bench.py --gpus 2` for real: the launcher spawns TWO rank processes (torch.distributed.run), both on cuda:0, with the
    collectives over gloo on CUDA tensors (SGR_BENCH_BACKEND=gloo SGR_BENCH_SHARE_GPU=1 -- RCCL refuses two ranks on one
    device).  The main rank program -- Workload.step with the factored exchange (all-reduce of the dense bucket the backward
    wrote into directly + all-gather of the per-view dRGB + local SH rebuild), the barrier + MAX-over-ranks timing, the
    overlap region -- runs with world size 2, rank 1 rendering its own view; afterwards both ranks hold the same summed
    gradients bit for bit.  Never a measurement: the first N > 1 execution must not be the driver's 8-GPU run."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SGR_BENCH_BACKEND="gloo", SGR_BENCH_SHARE_GPU="1")
    for reduce in ("factored", "bucket"):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                              "--device-warmup", "0", "--gaussians", "100000", "--no-cpu-baseline", "--no-other-configs",
                              "--reduce", reduce], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
        line = json.loads([ln for ln in out.stdout.strip().split("\n") if ln.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["dist_ranks"] == 2 and line["dist_backend"] == "gloo"
        assert line["ranks_share_one_gpu"] is True and line["rccl_ranks"] == 0
        assert line["value"] > 0 and line["config"]["views_per_step"] == 2
        assert line["summed_gradients_identical_on_all_ranks"] is True
        assert line["config"]["exchange_bytes_per_rank"] > 0 and "exchange_overlap" in line
        # the backward wrote the dense gradients straight into the bucket: nothing left to copy at begin()
        assert line["exchange_direct_bucket_writes"]["dense_tensors_copied_into_the_bucket_last_step"] == 0, line["exchange_direct_bucket_writes"]


def test_direct_bucket_writes_match_the_copying_exchange():
    """The backward writing its dense gradients into the exchange bucket, the masked dRGB into the payload slot and no own
    dL/dSH (rasterizer.BACKWARD_SINKS) gives bit-identical gradients to the round-5 path that allocated, copied, masked in a
    launch of its own and wrote dL/dSH twice (GradReducer.direct = False + the sinks removed)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    cam = _views(1)[0]
    sc = syn.make_scene(30000, cam, S=0, seed=21)
    sc.shs[::3, 0, :] -= 2.0
    names = ["means3D", "scales", "rotations", "opacities", "shs"]
    w = {k: dev(v) for k, v in syn.loss_weights(cam, seed=5).items()}
    port = socket.socket()
    port.bind(("127.0.0.1", 0))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port.getsockname()[1])
    port.close()
    dist.init_process_group("nccl", rank=0, world_size=1)
    res = {}
    try:
        for label in ("direct", "copying"):
            t = {k: dev(getattr(sc, k)).requires_grad_(True) for k in names}
            dense = [t[k] for k in names[:4]]
            with multiview.FactoredGradReducer(dense, t["shs"], t["means3D"], force=True) as red:
                if label == "copying":
                    red.dense.direct = False
                    rasterizer.remove_sink(red._sink)
                for rnd in range(2):
                    for p in t.values():
                        p.grad = None
                    color, radii, depth, alpha, _ = GaussianRasterizer(settings(cam))(t["means3D"], None, t["opacities"], shs=t["shs"],
                                                                                      scales=t["scales"], rotations=t["rotations"])
                    torch.autograd.backward([color, depth, alpha], [w["color"], w["depth"], w["alpha"]])
                    if label == "direct":  # autograd adopted the bucket slices as the parameters' gradients
                        assert all(t[k].grad.data_ptr() == v.data_ptr() for k, v in zip(names[:4], red.dense._views))
                        assert t["shs"].grad is None
                    red.all_reduce()
                    assert red.dense.copied_last == (0 if label == "direct" else 4)
                res[label] = {k: t[k].grad.clone() for k in names}
            assert len(rasterizer.BACKWARD_SINKS) == 0 and len(rasterizer.BACKWARD_OBSERVERS) == 0
    finally:
        dist.destroy_process_group()
    for k in names:
        assert torch.equal(res["direct"][k], res["copying"][k]), k


def test_async_exchange_matches_blocking_exchange():
    """GradReducer / FactoredGradReducer: begin() + wait() give the same gradients as all_reduce(), also when the next
    step's work is queued in between (one-rank RCCL group)."""
    port = socket.socket()
    port.bind(("127.0.0.1", 0))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port.getsockname()[1])
    port.close()
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        _async_body()
    finally:
        dist.destroy_process_group()


def _async_body():
    g = torch.Generator().manual_seed(1)
    params = [torch.zeros(50000, 3, device="cuda", requires_grad=True), torch.zeros(50000, 1, device="cuda", requires_grad=True)]
    grads = [torch.randn(p.shape, generator=g).cuda() for p in params]
    red = multiview.GradReducer(params, force=True)
    for p, gr in zip(params, grads):
        p.grad = gr.clone()
    red.begin()
    junk = torch.randn(4096, 4096, device="cuda") @ torch.randn(4096, 4096, device="cuda")  # "the next forward"
    for p in params:
        p.grad = torch.zeros_like(p)
    red.wait()
    torch.cuda.synchronize()
    for p, gr in zip(params, grads):
        assert torch.equal(p.grad, gr)
    assert torch.isfinite(junk).all()


def test_scene_graph_segments_keep_the_factored_exchange():
    """Background + two posed actors with Fourier DC features: the rasterizer's SH input is scene.compose's NON-leaf cat
    over sub-models (street_gaussian_model.py:287-449).  With `segments=` the reducer rebuilds every sub-model's
    features_dc / features_rest gradient from the view's dRGB (+ the posed models' world positions and IDFT rows carried
    in the payload) -- equal to what autograd sends through compose's backward when the SH tensor is NOT detached."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from street_gaussians_amd import scene
    cam = _views(1)[0]
    g = torch.Generator().manual_seed(31)
    r = lambda *s: torch.randn(*s, generator=g)
    M, C = 16, 3
    base = syn.make_scene(9000, cam, S=0, seed=13)

    def seg(lo, hi, actor):
        n = hi - lo
        op = base.opacities[lo:hi].clamp(1e-4, 1 - 1e-4)
        kw = dict(xyz=base.means3D[lo:hi].clone(), rotation=base.rotations[lo:hi].clone(), scaling=torch.log(base.scales[lo:hi]),
                  opacity=torch.log(op / (1 - op)), features_rest=base.shs[lo:hi, 1:, :].clone())
        if actor:
            kw.update(features_dc=base.shs[lo:hi, :1, :].repeat(1, C, 1) + 0.1 * r(n, C, 3), idft=torch.tensor([0.6, 0.3, 0.1]) + 0.05 * r(C),
                      pose=torch.tensor([1.0, 0.02 * actor, -0.01, 0.03, 0.1 * actor, -0.05, 0.2]))
            kw["xyz"] = kw["xyz"] - kw["pose"][4:]  # keep the posed model inside the view
        else:
            kw.update(features_dc=base.shs[lo:hi, :1, :].clone())
        t = {k: (v.float().cuda().requires_grad_(k != "idft") if torch.is_tensor(v) else v) for k, v in kw.items()}
        return scene.Segment(**t)

    segs = [seg(0, 6000, 0), seg(6000, 7800, 1), seg(7800, 9000, 2)]
    w = {k: dev(v) for k, v in syn.loss_weights(cam, seed=4).items()}

    def run(detach):
        for s in segs:
            for name in ("xyz", "rotation", "scaling", "opacity", "features_dc", "features_rest", "pose"):
                t = getattr(s, name)
                if t is not None:
                    t.grad = None
        means, rot, scl, opa, shs, _ = scene.compose(segs, M, 0)
        color, radii, depth, alpha, _ = GaussianRasterizer(settings(cam))(means, None, opa, shs=shs.detach() if detach else shs,
                                                                          scales=scl, rotations=rot)
        torch.autograd.backward([color, depth, alpha], [w["color"], w["depth"], w["alpha"]])

    run(detach=False)
    want = [(s.features_dc.grad.clone(), s.features_rest.grad.clone(), s.xyz.grad.clone()) for s in segs]
    port = socket.socket()
    port.bind(("127.0.0.1", 0))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port.getsockname()[1])
    port.close()
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        sh_segments = [multiview.SHSegment(s.features_dc, s.features_rest, s.xyz if s.pose is None else None) for s in segs]
        with multiview.FactoredGradReducer([segs[0].xyz], segments=sh_segments, force=True) as red:
            for rnd in range(2):  # twice: the payload buffers alternate
                red.set_frame([0, 1, 2], idft={1: segs[1].idft, 2: segs[2].idft})
                run(detach=True)
                assert not segs[1].features_dc.grad.any()  # nothing of dL/dSH went down the graph (compose wrote zeros)
                red.all_reduce()
                for s, (gdc, grest, gxyz) in zip(segs, want):
                    for got, ref in ((s.features_dc.grad, gdc), (s.features_rest.grad, grest)):
                        scale = float(ref.abs().max())
                        assert float((got - ref).abs().max()) <= 3e-6 * scale, float((got - ref).abs().max()) / scale
                assert torch.equal(segs[0].xyz.grad, want[0][2])  # the dense path: untouched by the factoring
            assert red.nbytes == 4 * (segs[0].xyz.numel() + 3 + 3 * 9000 + 3 * 3000 + 2 * C)
    finally:
        dist.destroy_process_group()


def test_bench_line_names_its_mode_and_carries_the_contracts_objects():
    """`bench.py --mode strict|exact|fast` (N = 1): `value` is the named mode's throughput, the other two ride along, and the
    line carries the objects the contract asks for -- roofline {bound, achieved, peak, unit, frac, traffic} for the dominant
    kernel of THAT mode and cpu_baseline {value, unit, cores, kind, sample}."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for mode, kernel in (("strict", "sgr_blend_bwd_kernel_exact"), ("fast", "sgr_blend_bwd_kernel_s0")):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--device-warmup", "0",
                              "--gaussians", "60000", "--no-other-configs", "--mode", mode],
                             env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads([ln for ln in out.stdout.strip().split("\n") if ln.startswith("{")][-1])
        assert line["mode"] == mode and line["conforming"] == (mode == "strict") and line["value"] > 0
        assert line["value"] == line["value_" + mode] and line["ms_per_step"] == line["ms_per_step_" + mode]
        for k in ("value_strict", "value_exact", "value_fast"):
            assert line[k] and line[k] > 0, k
        r = line["roofline"]
        assert r["bound"] == "valu_issue" and kernel in r["kernel"] and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert r["achieved"] > 0 and 0 < r["frac"] < 1 and "traffic" in r and r["mode"] == mode
        cb = line["cpu_baseline"]
        assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["host"]["threads"] >= cb["cores"] and cb["sample"]
        assert line["config"]["num_rendered_R"] >= line["config"]["instances_emitted"] > 0
        if mode == "strict":
            assert line["config"]["num_rendered_R"] == line["config"]["instances_emitted"]
        assert line["summary"]["mode"] == mode and list(line)[-1] == "summary"


def test_two_views_in_one_autograd_pass_accumulate_correctly_with_direct_bucket_writes():
    """loss = render(view 1) + render(view 2); ONE backward: both rasterizer backwards run before either gradient has been
    installed, so only the first may write into the exchange bucket -- the second is accumulated the ordinary way.  The summed
    gradients equal those of two separate backward passes without any reducer."""
    from diff_gaussian_rasterization import GaussianRasterizer
    cams = _views(2)
    sc = syn.make_scene(20000, cams[0], S=0, seed=23)
    names = ["means3D", "scales", "rotations", "opacities", "shs"]
    ws = [{k: dev(v) for k, v in syn.loss_weights(c, seed=7 + i).items()} for i, c in enumerate(cams)]

    def joint(t):
        loss = 0.0
        for cam, w in zip(cams, ws):
            color, radii, depth, alpha, _ = GaussianRasterizer(settings(cam))(t["means3D"], None, t["opacities"], shs=t["shs"],
                                                                              scales=t["scales"], rotations=t["rotations"])
            loss = loss + (color * w["color"]).sum() + (depth * w["depth"]).sum() + (alpha * w["alpha"]).sum()
        loss.backward()

    t0 = {k: dev(getattr(sc, k)).requires_grad_(True) for k in names}
    joint(t0)
    want = {k: t0[k].grad.clone() for k in names}
    port = socket.socket()
    port.bind(("127.0.0.1", 0))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port.getsockname()[1])
    port.close()
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        t = {k: dev(getattr(sc, k)).requires_grad_(True) for k in names}
        red = multiview.GradReducer([t[k] for k in names], force=True)
        try:
            joint(t)
            red.all_reduce()
            for k in names:
                assert torch.allclose(t[k].grad, want[k], rtol=0, atol=1e-6 * float(want[k].abs().max())), k
        finally:
            red.close()
    finally:
        dist.destroy_process_group()
