"""`-m gpu`: the forward without a host wait (sgr_set_lazy, include/sgr.h) and a training step replayed from a hipGraph.

The reference blocks on num_rendered in every forward (rasterizer_impl.cu:284) and so does the default path here.  In lazy
mode the instance list gets a capacity from the previous frames, nothing on the step path synchronises, and a forward +
backward can be captured once and replayed.  Held here: (1) a lazy step is BIT-identical to the blocking one -- images,
radii, all gradients -- although it sorts and scans a padded list; (2) a captured step replays bit-identically; (3) a frame
that overflows its capacity is reported by the next forward (SGR_E_LAZY), after which the thread re-seeds with a blocking
forward and carries on."""
import pytest
import torch

from gpu_utils import dev, settings
from street_gaussians_amd import _C, synthetic as syn
from street_gaussians_amd._native import SgrError

pytestmark = pytest.mark.gpu
NAMES = ["means3D", "scales", "rotations", "opacities", "shs"]


def _setup(P=150_000, W=960, H=640, seed=4, scale=1.0):
    from diff_gaussian_rasterization import GaussianRasterizer
    cam = syn.make_camera(W, H, fx=1000.0)
    sc = syn.make_scene(P, cam, S=0, seed=seed)
    st = settings(cam, bg=torch.tensor([0.1, 0.3, 0.2]))
    rast = GaussianRasterizer(raster_settings=st)
    t = {k: dev(getattr(sc, k)).requires_grad_(True) for k in NAMES}
    with torch.no_grad():
        t["scales"] *= scale
    m2d = torch.zeros(sc.P, 3, device="cuda", requires_grad=True)
    w = {k: v.cuda() for k, v in syn.loss_weights(cam, S=0).items()}
    return rast, t, m2d, w


def _step(rast, t, m2d, w):
    for x in list(t.values()) + [m2d]:
        x.grad = None
    color, radii, depth, alpha, _ = rast(t["means3D"], m2d, t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    torch.autograd.backward([color, depth, alpha], [w["color"], w["depth"], w["alpha"]])
    return color, radii, depth, alpha


def _snapshot(outs, t, m2d):
    return [o.detach().clone() for o in outs] + [t[k].grad.clone() for k in NAMES] + [m2d.grad.clone()]


def _same(a, b, what):
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), f"{what}: tensor {i} differs"


def test_lazy_step_is_bit_identical_and_replays_from_a_graph():
    from street_gaussians_amd import rasterizer
    rast, t, m2d, w = _setup()
    prev = _C.set_lazy(False)
    try:
        ref = _snapshot(_step(rast, t, m2d, w), t, m2d)
        R_true = rasterizer.last_num_rendered()
        _C.set_lazy(True)
        seed = _snapshot(_step(rast, t, m2d, w), t, m2d)  # (the first forward after the switch blocks and seeds the capacity)
        _same(ref, seed, "seeding step")
        first = _snapshot(_step(rast, t, m2d, w), t, m2d)
        second = _snapshot(_step(rast, t, m2d, w), t, m2d)
        torch.cuda.synchronize()
        _same(ref, first, "first lazy step")
        _same(ref, second, "second lazy step")
        R, cap, flags = _C.lazy_status()
        assert flags == 0 and R == R_true and R <= cap <= int(1.35 * R) + 8192, (R, cap, flags)  # (the blocking call's 25 % of head-room decays)
        assert rasterizer.last_num_rendered() == cap  # what the caller sees (and hands to the backward) is the capacity

        # ---- a captured step: forward + backward in one graph, replayed
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up on the capture side, as torch's graph recipe asks
            _step(rast, t, m2d, w)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        for x in list(t.values()) + [m2d]:
            x.grad = None
        with torch.cuda.graph(g):
            color, radii, depth, alpha, _ = rast(t["means3D"], m2d, t["opacities"], shs=t["shs"], scales=t["scales"],
                                                 rotations=t["rotations"])
            torch.autograd.backward([color, depth, alpha], [w["color"], w["depth"], w["alpha"]])
        for rep in range(3):
            for o in (color, depth, alpha):
                o.zero_()
            for k in NAMES:
                t[k].grad.zero_()
            g.replay()
            torch.cuda.synchronize()
            _same(ref, _snapshot((color, radii, depth, alpha), t, m2d), f"graph replay {rep}")
            R, cap, flags = _C.lazy_status()
            assert flags == 0 and R == R_true
        del g
    finally:
        _C.set_lazy(prev)
        torch.cuda.synchronize()


def test_lazy_overflow_is_reported_one_call_late_and_the_thread_recovers():
    small = _setup(P=60_000, W=640, H=400, seed=7)
    big = _setup(P=60_000, W=640, H=400, seed=7, scale=4.0)  # the same Gaussians four times as large: many more instances
    prev = _C.set_lazy(False)
    try:
        ref_big = _snapshot(_step(*big), big[1], big[2])
        ref_small = _snapshot(_step(*small), small[1], small[2])
        _C.set_lazy(True)
        _step(*small)  # blocking: seeds the capacity with the small frame's R
        _step(*small)  # lazy
        _step(*big)  # overflows the capacity: nobody knows yet
        torch.cuda.synchronize()
        R, cap, flags = _C.lazy_status()
        assert (flags & 1) and R > cap, (R, cap, flags)
        with pytest.raises(SgrError, match="PREVIOUS lazy forward"):
            _step(*big)
        # the next forward is a blocking one (it re-seeds the capacity), then the thread is lazy again -- and right
        got = _snapshot(_step(*big), big[1], big[2])
        _same(ref_big, got, "blocking forward after the overflow")
        got = _snapshot(_step(*big), big[1], big[2])
        torch.cuda.synchronize()
        _same(ref_big, got, "lazy forward with the new capacity")
        assert _C.lazy_status()[2] == 0
        got = _snapshot(_step(*small), small[1], small[2])
        _same(ref_small, got, "a smaller frame inside the capacity")
    finally:
        _C.set_lazy(prev)
        torch.cuda.synchronize()


def test_lazy_frames_with_few_or_no_instances():
    """The capacity has a floor, so a lazy frame with no visible Gaussian at all (everything behind the camera) runs over a
    list of sentinel keys only: images = the background, every gradient zero, no flag raised; and the thread carries on with a
    normal frame afterwards, bit-identical to the blocking result."""
    rast, t, m2d, w = _setup(P=20_000, W=320, H=200, seed=11)
    prev = _C.set_lazy(False)
    try:
        ref = _snapshot(_step(rast, t, m2d, w), t, m2d)
        behind = {k: v.detach().clone().requires_grad_(True) for k, v in t.items()}
        with torch.no_grad():
            behind["means3D"][:, 2] = -behind["means3D"][:, 2].abs() - 1.0  # camera looks down +z (synthetic.make_camera)
        ref_empty = _snapshot(_step(rast, behind, m2d, w), behind, m2d)
        assert all(float(g.abs().max()) == 0.0 for g in ref_empty[4:]), "an empty frame has zero gradients"
        _C.set_lazy(True)
        _step(rast, t, m2d, w)        # blocking: seeds the capacity
        got = _snapshot(_step(rast, behind, m2d, w), behind, m2d)  # lazy, R = 0
        torch.cuda.synchronize()
        _same(ref_empty, got, "lazy frame without instances")
        R, cap, flags = _C.lazy_status()
        assert R == 0 and cap >= 1024 and flags == 0, (R, cap, flags)
        got = _snapshot(_step(rast, t, m2d, w), t, m2d)  # lazy again, a normal frame inside the (decayed) capacity
        torch.cuda.synchronize()
        _same(ref, got, "lazy frame after an empty one")
        assert _C.lazy_status()[2] == 0
    finally:
        _C.set_lazy(prev)
        torch.cuda.synchronize()
