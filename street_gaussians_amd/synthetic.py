"""Synthetic street-like Gaussian scenes and cameras (SURVEY.md 8d recipe).

Used by bench.py and the tests so that every number is quoted on the same inputs.  The camera
matrices follow the reference's convention (/root/reference/lib/utils/camera_utils.py:52-61 and
lib/utils/graphics_utils.py:51-70): ``viewmatrix = W2C.T`` and ``projmatrix = W2C.T @ P.T`` --
i.e. the tensors hold the TRANSPOSED (row-vector) matrices, flat index ``m[4*r+c]`` is element
(c, r) of the column-vector matrix.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class Camera:
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    viewmatrix: torch.Tensor  # [4,4] = W2C.T
    projmatrix: torch.Tensor  # [4,4] = W2C.T @ P.T
    campos: torch.Tensor      # [3]

    def to(self, device):
        return Camera(self.image_height, self.image_width, self.tanfovx, self.tanfovy,
                      self.viewmatrix.to(device), self.projmatrix.to(device), self.campos.to(device))


def projection_matrix(znear: float, zfar: float, tanfovx: float, tanfovy: float) -> torch.Tensor:
    """getProjectionMatrix of the reference (lib/utils/graphics_utils.py:51-70), column-vector form."""
    top = tanfovy * znear
    bottom = -top
    right = tanfovx * znear
    left = -right
    P = torch.zeros(4, 4, dtype=torch.float64)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def make_camera(width: int = 1920, height: int = 1280, fx: float = 2050.0, fy: Optional[float] = None,
                yaw_deg: float = 0.0, znear: float = 0.01, zfar: float = 1000.0,
                translation=(0.0, 0.0, 0.0)) -> Camera:
    """Waymo-front-like pinhole camera; view k of the multi-view benchmark is yawed by k*5 degrees."""
    fy = fx if fy is None else fy
    tanfovx = width / (2.0 * fx)
    tanfovy = height / (2.0 * fy)
    a = math.radians(yaw_deg)
    # world -> camera: rotate about the camera's y axis, then translate
    R = torch.tensor([[math.cos(a), 0.0, -math.sin(a)], [0.0, 1.0, 0.0], [math.sin(a), 0.0, math.cos(a)]],
                     dtype=torch.float64)
    W2C = torch.eye(4, dtype=torch.float64)
    W2C[:3, :3] = R
    W2C[:3, 3] = torch.tensor(translation, dtype=torch.float64)
    P = projection_matrix(znear, zfar, tanfovx, tanfovy)
    view_t = W2C.t().contiguous()
    proj_t = (W2C.t() @ P.t()).contiguous()
    campos = torch.linalg.inv(W2C)[:3, 3]
    return Camera(height, width, tanfovx, tanfovy, view_t.float(), proj_t.float(), campos.float().contiguous())


@dataclass
class Scene:
    means3D: torch.Tensor    # [P,3]
    scales: torch.Tensor     # [P,3]  (post-activation)
    rotations: torch.Tensor  # [P,4]  (w,x,y,z) normalised
    opacities: torch.Tensor  # [P,1]
    shs: torch.Tensor        # [P,M,3]
    semantics: torch.Tensor  # [P,S]

    def to(self, device):
        return Scene(*[t.to(device) for t in (self.means3D, self.scales, self.rotations, self.opacities, self.shs,
                                              self.semantics)])

    @property
    def P(self):
        return self.means3D.shape[0]


def make_scene(P: int, cam: Camera, sh_degree_max: int = 3, S: int = 0, seed: int = 0,
               zmin: float = 1.0, zmax: float = 80.0, scale_px: float = 0.0015, scale_sigma: float = 0.6,
               margin: float = 1.1) -> Scene:
    """means uniform in the view-0 frustum with depth uniform in 1/z over [zmin, zmax],
    x,y = z*tanfov*U(-margin, margin); scales = exp(N(log(scale_px*z), scale_sigma^2)) per axis;
    rotations = normalize(N(0,1)^4); opacity = sigmoid(N(0, 2^2)); shs = N(0, 0.3^2) with higher bands
    damped and a DC offset so colours sit in range; semantics = N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(P, generator=g, dtype=torch.float64)
    inv_z = 1.0 / zmax + u * (1.0 / zmin - 1.0 / zmax)
    z = 1.0 / inv_z
    ux = (torch.rand(P, generator=g, dtype=torch.float64) * 2 - 1) * margin
    uy = (torch.rand(P, generator=g, dtype=torch.float64) * 2 - 1) * margin
    x = z * cam.tanfovx * ux
    y = z * cam.tanfovy * uy
    means = torch.stack([x, y, z], dim=1).float()
    logs = torch.log(scale_px * z)[:, None] + scale_sigma * torch.randn(P, 3, generator=g, dtype=torch.float64)
    scales = torch.exp(logs).float()
    q = torch.randn(P, 4, generator=g, dtype=torch.float64)
    q = (q / q.norm(dim=1, keepdim=True)).float()
    opac = torch.sigmoid(2.0 * torch.randn(P, 1, generator=g, dtype=torch.float64)).float()
    M = (sh_degree_max + 1) ** 2
    shs = 0.3 * torch.randn(P, M, 3, generator=g, dtype=torch.float64)
    if M > 1:
        shs[:, 1:, :] *= 0.3
    shs[:, 0, :] += 0.5  # DC offset
    sem = torch.randn(P, S, generator=g, dtype=torch.float64).float()
    return Scene(means.contiguous(), scales.contiguous(), q.contiguous(), opac.contiguous(),
                 shs.float().contiguous(), sem.contiguous())


def loss_weights(cam: Camera, S: int = 0, seed: int = 1):
    """Fixed random per-pixel weights so every dL_dout_* is dense."""
    g = torch.Generator().manual_seed(seed)
    H, W = cam.image_height, cam.image_width
    return dict(color=torch.rand(3, H, W, generator=g) - 0.5, depth=(torch.rand(1, H, W, generator=g) - 0.5) * 0.1,
                alpha=torch.rand(1, H, W, generator=g) - 0.5, semantic=torch.rand(S, H, W, generator=g) - 0.5)


def smoke_test_camera() -> dict:
    """The hard-coded KITTI-like camera of the reference's only test
    (/root/reference/script/test_gaussian_rasterization.py:6-20)."""
    return dict(
        image_height=375, image_width=1242, FoVx=1.416, FoVy=0.506,
        world_view_transform=torch.tensor([[0.9598, 0.0081, 0.2806, 0.0000], [-0.0123, 0.9998, 0.0134, 0.0000],
                                           [-0.2804, -0.0163, 0.9597, 0.0000], [-2.0954, -0.0935, 4.9320, 1.0000]]),
        full_proj_transform=torch.tensor([[1.1205, 0.0312, 0.2806, 0.2806], [-0.0144, 3.8661, 0.0134, 0.0134],
                                          [-0.3274, -0.0632, 0.9598, 0.9597], [-2.4464, -0.3614, 4.9225, 4.9320]]),
        camera_center=torch.tensor([6.2808e-01, 1.4572e-03, -5.3226e+00]))


def make_street_segments(P: int, cam: Camera, n_actors: int = 12, actor_share: float = 0.12, S: int = 0, seed: int = 0,
                         sh_degree_max: int = 3, fourier_dim: int = 3):
    """A NON-uniform, street-like scene as the reference composes one (lib/models/street_gaussian_model.py:219-449): one
    static background model + ``n_actors`` rigid actors with per-frame poses and Fourier DC features, as RAW parameters
    (log scales, logit opacities, un-normalised quaternions) ready for ``street_gaussians_amd.scene.Segment``.

    Camera frame of view 0: x right, y down, z forward, camera 1.6 m above the road.  Background = road surface (45 %: the
    plane y = 1.6, depth uniform in 1/z), two facade planes at x = -9 / +9 m rising 7 m above the camera (35 %), clutter
    between them below 0.8 m above the camera (20 %) -- nothing above the facades, so the upper middle of the image is EMPTY SKY.  Actors =
    4.5 x 1.8 x 1.5 m boxes of small, dense splats standing on the road between 6 and 45 m (the tiles they cover carry
    lists several times the mean).  Returns a list of dicts with the Segment fields (CPU float32 tensors)."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)
    rn = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    M = (sh_degree_max + 1) ** 2
    n_act_total = int(P * actor_share) if n_actors else 0
    n_bg = P - n_act_total
    n_road, n_fac = int(0.45 * n_bg), int(0.35 * n_bg)
    n_clu = n_bg - n_road - n_fac

    def depth(n, zmin, zmax):
        return 1.0 / (1.0 / zmax + r(n) * (1.0 / zmin - 1.0 / zmax))

    # road
    z = depth(n_road, 2.0, 80.0)
    road = torch.stack([z * cam.tanfovx * (r(n_road) * 2.2 - 1.1), torch.full((n_road,), 1.6, dtype=torch.float64) + 0.02 * rn(n_road), z], 1)
    # facades: x = +-9, y from 1.6 (ground) up to -7
    z = depth(n_fac, 10.0, 80.0)
    side = torch.where(r(n_fac) < 0.5, -1.0, 1.0)
    fac = torch.stack([side * (9.0 + 0.15 * rn(n_fac)), 1.6 - 8.6 * r(n_fac), z], 1)
    # clutter (vegetation, poles, parked things): below 3 m above the camera, between the facades
    z = depth(n_clu, 4.0, 60.0)
    clu = torch.stack([(r(n_clu) * 2 - 1) * 8.5, 1.6 - 2.4 * r(n_clu) ** 2, z], 1)
    bg_xyz = torch.cat([road, fac, clu], 0)
    zb = bg_xyz[:, 2]

    def raw(n, z_for_scale, scale_px, sigma):
        logs = torch.log(scale_px * z_for_scale)[:, None] + sigma * rn(n, 3)
        q = rn(n, 4)
        op = 2.0 * rn(n, 1)  # logit
        dc = 0.3 * rn(n, 1, 3) + 0.5
        rest = 0.09 * rn(n, M - 1, 3)
        return logs, q, op, dc, rest

    segs = []
    logs, q, op, dc, rest = raw(n_bg, zb, 0.0015, 0.6)
    f32 = lambda t: t.float().contiguous()
    segs.append(dict(xyz=f32(bg_xyz), rotation=f32(q), scaling=f32(logs), opacity=f32(op), features_dc=f32(dc),
                     features_rest=f32(rest), semantic=f32(rn(n_bg, S)) if S else None))
    if n_actors:
        per = [n_act_total // n_actors + (1 if i < n_act_total % n_actors else 0) for i in range(n_actors)]
        for i, n in enumerate(per):
            za = float(6.0 + (45.0 - 6.0) * r(1) ** 1.5)
            xa = float((r(1) * 2 - 1) * min(6.0, 0.8 * za * cam.tanfovx))
            yaw = float((r(1) * 2 - 1) * 0.4)
            # points on / in the box (object frame: x length, y up-down, z width), denser near the surfaces
            box = (r(n, 3) - 0.5) * torch.tensor([4.5, 1.5, 1.8], dtype=torch.float64)
            logs = torch.log(torch.full((n, 1), 0.04, dtype=torch.float64)) + 0.5 * rn(n, 3)
            qa = rn(n, 4)
            opa = 1.0 + 1.5 * rn(n, 1)
            dcs = 0.3 * rn(n, fourier_dim, 3) + 0.5 / fourier_dim
            rest = 0.09 * rn(n, M - 1, 3)
            import math
            pose = torch.tensor([math.cos(yaw / 2), 0.0, math.sin(yaw / 2), 0.0, xa, 1.6 - 0.75, za], dtype=torch.float32)
            idft = torch.tensor([1.0] + [0.2 / (k + 1) for k in range(fourier_dim - 1)], dtype=torch.float32)
            segs.append(dict(xyz=f32(box), rotation=f32(qa), scaling=f32(logs), opacity=f32(opa), features_dc=f32(dcs),
                             features_rest=f32(rest), semantic=f32(rn(n, 1)) if S else None, pose=pose, idft=idft,
                             class_label=(i % max(S, 1))))
    return segs
